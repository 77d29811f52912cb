"""Tensor-level wrappers over the C ABI (one function per entry point).

Every wrapper takes contiguous fp32 CUDA(HIP) tensors, allocates outputs with torch
(plumbing only: the caching allocator and the current stream), and enqueues the HIP
kernel on ``torch.cuda.current_stream()``.  There is no CPU or ATen fallback: a CPU
tensor raises.
"""
import ctypes

import torch

from . import _lib
from . import layout
from ._lib import ACT_NONE, ACT_TANH, PAD_REFLECT, PAD_ZERO, SeganSrc, check


import os as _os

PREC_FP32, PREC_BF16, PREC_BF16X3 = 0, 1, 3
_PREC_NAMES = {'fp32': PREC_FP32, 'bf16': PREC_BF16, 'bf16x3': PREC_BF16X3}
_precision = _PREC_NAMES[_os.environ.get('SEGAN_PRECISION', 'fp32')]
_EUNSUPPORTED = -3


_deterministic = _os.environ.get('SEGAN_DETERMINISTIC', '0') == '1'


def set_deterministic(on):
    """Bit-reproducible mode.  The forward / data-gradient contractions always are reproducible
    (their stream-K tail is reduced in a fixed order); this switch makes the weight gradients and
    the dense-head GEMMs reduce their contraction splits in a fixed order too (slabs + a second
    kernel) instead of with fp32 atomics.  Measured cost: 11 % of the step in round 2, 0.6 - 2.2 % in
    rounds 3 - 5, 0.4 % at the end of round 6 (one box; the ordered reductions now issue their loads
    before their stores) — the first reading below the 1 % at which it would become the default, so
    it stays opt-in until that holds across boxes: ``set_deterministic(True)`` /
    SEGAN_DETERMINISTIC=1 / train.py --deterministic.  (The
    bf16 / bf16x3 weight gradients always add their splits with atomics.)"""
    global _deterministic
    _deterministic = bool(on)


def get_deterministic():
    return _deterministic


_ACC_NAMES = ('plain', 'blocked')
_accumulation = _os.environ.get('SEGAN_ACCUMULATION', 'plain')
if _accumulation not in _ACC_NAMES:
    raise ValueError('SEGAN_ACCUMULATION must be one of {}'.format(_ACC_NAMES))
PREC_FP32_BLOCKED = 4


def set_accumulation(mode):
    """How the fp32 forward / data-gradient contractions accumulate: 'plain' (one MFMA
    accumulator over the whole contraction: the default and the benchmarked configuration) or
    'blocked' (SEGAN_PREC_FP32_BLOCKED: 256-term blocks summed in a second register set).
    Blocked accumulation brings the forward error of the deep layers against fp64 from
    1.5-2.2e-6 down to 3e-7 — below torch's own CPU fp32 result.  It costs one of the three
    resident waves per SIMD (~3 % of the contraction rate, 1.5-2.2 % of the step), so it is
    opt-in: ``set_accumulation('blocked')`` / SEGAN_ACCUMULATION=blocked.  What it does NOT
    buy is a reliably smaller batch-300 gradient distance to the oracle: that figure is set by
    which individual PReLU gates flip, not by the size of the roundoff.
    The weight gradients stay plain (their contractions are split across workgroups already;
    blocking them was measured: no change in any parity figure for 4.4 % of the step).  No
    effect on the bf16 modes."""
    global _accumulation
    if mode not in _ACC_NAMES:
        raise ValueError('accumulation must be one of {}'.format(_ACC_NAMES))
    _accumulation = mode


def get_accumulation():
    return _accumulation


def _fp32():
    return PREC_FP32_BLOCKED if _accumulation == 'blocked' else PREC_FP32


def set_precision(mode):
    """Precision of ALL contractions (conv / deconv forward, data gradients and weight
    gradients): 'fp32' (exact fp32 MFMA, the default and the benchmarked configuration), 'bf16'
    (bf16 operands, fp32 accumulate: BASELINE config 5) or 'bf16x3' (exact 3-way bf16 split of
    every fp32 operand, six partial products: fp32-class accuracy on the bf16 matrix cores).
    A geometry the bf16 kernels do not cover falls back to fp32 per call.  The dense head,
    BatchNorm, activations, losses and optimizers always run in fp32."""
    global _precision
    if mode not in _PREC_NAMES:
        raise ValueError('precision must be one of {}'.format(sorted(_PREC_NAMES)))
    _precision = _PREC_NAMES[mode]


def get_precision():
    return [k for k, v in _PREC_NAMES.items() if v == _precision][0]


def _stream():
    # the raw handle of torch's current stream on the current device: two C calls (torch.cuda.
    # current_stream() builds a python Stream object per call: 3 us, twice per library call)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError('{} must be a tensor, got {}'.format(name, type(t)))
    if not t.is_cuda:
        raise RuntimeError(
            '{} is on {}: segan_pytorch_amd runs only on an MI355X (HIP) device; '
            'there is no CPU path'.format(name, t.device))
    if t.dtype != torch.float32:
        raise TypeError('{} must be float32, got {}'.format(name, t.dtype))
    if not t.is_contiguous():
        raise ValueError('{} must be contiguous'.format(name))
    if ndim is not None and t.dim() != ndim:
        raise ValueError('{} must have {} dims, got {}'.format(name, ndim, tuple(t.shape)))
    return t


class Src(object):
    """A logical [B, C0+C1, L] activation with an on-load transform (``segan_src``).

    ``t0``/``t1``: the channel segments; ``scale``/``shift``/``slope``: optional
    per-channel vectors over the concatenated channel axis.
    """

    def __init__(self, t0, t1=None, scale=None, shift=None, slope=None):
        _chk(t0, 'src.t0', 3)
        self.t0, self.t1 = t0, t1
        self.C0 = t0.shape[1]
        self.C1 = 0
        if t1 is not None:
            _chk(t1, 'src.t1', 3)
            if t1.shape[0] != t0.shape[0] or t1.shape[2] != t0.shape[2]:
                raise ValueError('src segments disagree: {} vs {}'.format(
                    tuple(t0.shape), tuple(t1.shape)))
            self.C1 = t1.shape[1]
        self.C = self.C0 + self.C1
        for name, v in (('scale', scale), ('shift', shift), ('slope', slope)):
            if v is not None:
                _chk(v, 'src.' + name)
                if v.numel() != self.C:
                    raise ValueError('src.{} has {} entries for {} channels'.format(
                        name, v.numel(), self.C))
        self.scale, self.shift, self.slope = scale, shift, slope
        self.B, self.L = t0.shape[0], t0.shape[2]

    def c_struct(self):
        s = SeganSrc()
        s.p0 = self.t0.data_ptr()
        s.p1 = self.t1.data_ptr() if self.t1 is not None else None
        s.C0, s.C1 = self.C0, self.C1
        s.scale = self.scale.data_ptr() if self.scale is not None else None
        s.shift = self.shift.data_ptr() if self.shift is not None else None
        s.slope = self.slope.data_ptr() if self.slope is not None else None
        return s


# stream-K scratch of the fp32 contractions (include/segan_hip.h): one buffer per device and
# stream (launches on different streams must not share it), allocated on first use
_corr_scratch = {}


def _scratch():
    """(pointer, bytes) of this device+stream's stream-K scratch."""
    dev = torch._C._cuda_getDevice()
    key = (dev, torch._C._cuda_getCurrentRawStream(dev))
    buf = _corr_scratch.get(key)
    if buf is None:
        nbytes = _lib.load().segan_corr_scratch_bytes()
        buf = torch.empty(nbytes, device=torch.device('cuda', dev), dtype=torch.uint8)
        _corr_scratch[key] = buf
    return ctypes.c_void_p(buf.data_ptr()), buf.numel()


def last_corr_launch():
    """Diagnostics: dict describing this thread's last fp32 conv / deconv forward or data
    gradient launch (segan_debug_last_corr)."""
    arr = (ctypes.c_int * 6)()
    _lib.load().segan_debug_last_corr(arr)
    k = ('kernel', 'workgroups', 'tiles', 'tiles_whole', 'streamk_units', 'xf_mode')
    d = dict(zip(k, list(arr)))
    d['streamk'] = d['streamk_units'] > 0
    return d


def last_wgrad_launch():
    """Diagnostics: how this thread's last fp32 weight gradient was launched."""
    arr = (ctypes.c_int * 6)()
    _lib.load().segan_debug_last_wgrad(arr)
    return dict(zip(('kernel', 'tiles', 'splits', 'chunks_per_split', 'resident_per_cu', 'hi_loads'),
                    list(arr)))


_call_scratch = {}


def _stream_scratch(nbytes, device):
    """A per-call scratch buffer of at least `nbytes` on this device + stream: ONE grow-only
    buffer per (device, stream), re-used by every call — the calls of a stream execute in order
    and none keeps its scratch past its last kernel (round-3 advice: a fresh torch.empty of
    134+ MB per bf16 contraction call went through the caching allocator thousands of times per
    step)."""
    dev = torch.device(device).index
    if dev is None:
        dev = torch._C._cuda_getDevice()
    key = (dev, torch._C._cuda_getCurrentRawStream(dev))
    buf = _call_scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _call_scratch.pop(key, None)          # release before growing
        buf = torch.empty(int(nbytes), device=torch.device('cuda', dev), dtype=torch.uint8)
        _call_scratch[key] = buf
    return buf


def set_reserved_slots(n):
    """Workgroup slots the launch planners of the contraction kernels leave free for other kernels
    (segan_set_reserved_slots; default 0 or $SEGAN_RESERVED_SLOTS): RCCL's channels during a
    data-parallel step.  Returns the previous value."""
    return int(_lib.load().segan_set_reserved_slots(int(n)))


def release_scratch(device=None):
    """Hand the per-stream scratch buffers (the stream-K slabs of the fp32 contractions, 128 MiB
    per device + stream, and the grow-only call scratch of the bf16 / weight-gradient calls, 134+ MB
    per stream) back to the caching allocator — all of them, or those of one device.

    The buffers are keyed by the RAW stream handle, so an entry outlives a destroyed torch stream;
    nothing else ever frees them.  Call it where streams come and go (a server that creates a
    stream per request) or before handing the device to something else; the next contraction call
    re-allocates what it needs.  Safe at any point between calls: the kernels that used a buffer
    were enqueued on its stream, and the caching allocator re-uses the block in stream order
    (the buffers were allocated while their stream was current).

    The invariant the sharing rests on, stated here because nothing else enforces it: a library
    call hands out AT MOST ONE region of a stream's call scratch (`_stream_scratch`), and no call
    keeps it past its last kernel.  A bf16 call that the library declines (SEGAN_EUNSUPPORTED) and
    that is retried in fp32 satisfies it because the fp32 forms use `_scratch()` — the separate
    stream-K buffer — only."""
    dev = None if device is None else torch.device(device).index
    for d in (_corr_scratch, _call_scratch):
        for key in [k for k in d if dev is None or k[0] == dev]:
            del d[key]


def scratch_bytes():
    """Bytes currently held by the per-stream scratch buffers (diagnostics / tests)."""
    return sum(b.numel() for d in (_corr_scratch, _call_scratch) for b in d.values())


def _bf_scratch(op, B, N, M, L, K, S, pad, device):
    """(buffer, pointer, bytes) of the packed-activation scratch of a bf16 / bf16x3 contraction
    call (segan_bf16_scratch_bytes), from the stream's call scratch."""
    planes = 3 if _precision == PREC_BF16X3 else 1
    nbytes = _lib.load().segan_bf16_scratch_bytes(op, B, N, M, L, K, S, pad, planes)
    if nbytes == 0:
        return None, None, 0
    buf = _stream_scratch(nbytes, device)
    return buf, ctypes.c_void_p(buf.data_ptr()), nbytes


def conv_pad(K, S):
    return layout.conv_pad(K, S)


def deconv_pad(K, S):
    return layout.deconv_pad(K, S)


# ---------------------------------------------------------------------------------
# weight packing
# ---------------------------------------------------------------------------------
_weights_epoch = 0


def bump_weights_epoch(params=None):
    """Invalidate packed weights after the weights were modified through raw pointers (the
    fused optimizers, DP broadcast), which torch's version counter cannot see.  With
    `params` only those tensors' packs are invalidated (an optimizer step must not force
    the OTHER network to re-pack); without, every WeightPack is."""
    global _weights_epoch
    if params is None:
        _weights_epoch += 1
        return
    for p in params:
        p._segan_epoch = getattr(p, '_segan_epoch', 0) + 1


class WeightPack(object):
    """Polyphase-packed copies of ONE weight tensor [m, n, K] (see layout.py), owned by
    the module that owns the weight and refreshed when the weight changes."""

    def __init__(self):
        self._buf = {}
        self._key = {}

    def _get(self, w, S, pad_t, want):
        _chk(w, 'weight', 3)
        key = (w.data_ptr(), w._version, _weights_epoch, getattr(w, '_segan_epoch', 0),
               tuple(w.shape), S, pad_t)
        if self._key.get(want) == key:
            return self._buf[want]
        lib = _lib.load()
        M, N, K = w.shape
        nbytes = (lib.segan_packed_f_bytes if want == 'f' else lib.segan_packed_t_bytes)(M, N, S)
        if nbytes == 0:
            raise ValueError('unsupported stride {} (must be 1, 2 or 4)'.format(S))
        buf = self._buf.get(want)
        if buf is None or buf.numel() * 4 != nbytes or buf.device != w.device:
            buf = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
        check(lib.segan_pack_weights(_ptr(w), _ptr(buf) if want == 'f' else None,
                                     _ptr(buf) if want == 't' else None, M, N, K, S, pad_t,
                                     _stream()), 'pack_weights')
        self._buf[want] = buf
        self._key[want] = key
        return buf

    def f(self, w, S):
        return self._get(w, S, 0, 'f')

    def g(self, w, S):
        """"G" packing wg[m][n*32 + r*8 + u] = w[m][n][4u + r] of the short-row data gradient
        (segan_pack_weights_g)."""
        _chk(w, 'weight', 3)
        key = (w.data_ptr(), w._version, _weights_epoch, getattr(w, '_segan_epoch', 0),
               tuple(w.shape), S)
        if self._key.get('g') == key:
            return self._buf['g']
        lib = _lib.load()
        M, N, K = w.shape
        nbytes = lib.segan_packed_g_bytes(M, N, S)
        if nbytes == 0:
            raise ValueError('G packing: stride {} not supported'.format(S))
        buf = self._buf.get('g')
        if buf is None or buf.numel() * 4 != nbytes or buf.device != w.device:
            buf = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
        check(lib.segan_pack_weights_g(_ptr(w), _ptr(buf), M, N, K, S, _stream()), 'pack_weights_g')
        self._buf['g'] = buf
        self._key['g'] = key
        return buf

    def t(self, w, S, pad_t):
        return self._get(w, S, pad_t, 't')

    def bf(self, w, S, pad_t, tform, planes):
        """bf16-plane packing (segan_pack_weights_bf) for the bf16 / bf16x3 kernels."""
        _chk(w, 'weight', 3)
        want = ('bf', tform, planes)
        key = (w.data_ptr(), w._version, _weights_epoch, getattr(w, '_segan_epoch', 0),
               tuple(w.shape), S, pad_t)
        if self._key.get(want) == key:
            return self._buf[want]
        lib = _lib.load()
        M, N, K = w.shape
        nbytes = lib.segan_packed_bf_bytes(M, N, S, tform, planes)
        if nbytes == 0:
            raise ValueError('unsupported geometry for bf16 packing')
        buf = self._buf.get(want)
        if buf is None or buf.numel() * 2 != nbytes or buf.device != w.device:
            buf = torch.empty(nbytes // 2, device=w.device, dtype=torch.bfloat16)
        check(lib.segan_pack_weights_bf(_ptr(w), _ptr(buf), M, N, K, S, tform, pad_t, planes,
                                        _stream()), 'pack_weights_bf')
        self._buf[want] = buf
        self._key[want] = key
        return buf


# ---------------------------------------------------------------------------------
# contractions
# ---------------------------------------------------------------------------------
def conv1d_fwd(src, w, bias, S, roll=0, pad_mode=PAD_REFLECT, padL=None, pack=None):
    """Pre-activation of GConv1DBlock: conv(reflect_pad(roll(src)), w) + bias."""
    M, N, K = w.shape
    if src.C != N:
        raise ValueError('conv1d_fwd: input has {} channels, weight expects {}'.format(src.C, N))
    B, L = src.B, src.L
    if L % S != 0:
        raise ValueError('conv1d_fwd: length {} not divisible by stride {}'.format(L, S))
    if padL is None:
        padL = conv_pad(K, S)[0]
    out = torch.empty((B, M, L // S), device=w.device, dtype=torch.float32)
    cs = src.c_struct()
    pack = pack or WeightPack()
    lib = _lib.load()
    if _precision and pad_mode == PAD_REFLECT:
        keep, sp, sn = _bf_scratch(0, B, N, M, L, K, S, padL, w.device)
        rc = lib.segan_conv1d_fwd(ctypes.byref(cs), _ptr(pack.bf(w, S, 0, 0, _precision)),
                                  _ptr(bias), _ptr(out), B, N, M, L, K, S, padL, pad_mode, roll,
                                  _precision, sp, sn, _stream())
        if rc != _EUNSUPPORTED:
            check(rc, 'conv1d_fwd')
            return out
    check(lib.segan_conv1d_fwd(ctypes.byref(cs), _ptr(pack.f(w, S)), _ptr(bias), _ptr(out), B, N,
                               M, L, K, S, padL, pad_mode, roll, _fp32(), *_scratch(), _stream()),
          'conv1d_fwd')
    return out


def conv1d_dgrad(da, w, L, S, roll=0, padL=None, pack=None):
    """Gradient of conv1d_fwd w.r.t. its (un-rolled, un-padded) input: [B, N, L]."""
    _chk(da, 'da', 3)
    M, N, K = w.shape
    B = da.shape[0]
    if da.shape[1] != M or da.shape[2] * S != L:
        raise ValueError('conv1d_dgrad: da {} inconsistent with weight {} / L {}'.format(
            tuple(da.shape), tuple(w.shape), L))
    if padL is None:
        padL = conv_pad(K, S)[0]
    small = N <= 2          # first layer: direct VALU kernel on the unpacked weight
    _chk(w, 'weight', 3)
    pack = pack or WeightPack()
    if not small and not _precision and short_rows_ok(N, M, L, S):
        # deep layers (at most 64 positions per row after the stride): GEMM + col2im form, no
        # zero-halo columns, fold and roll in the epilogue
        return conv1d_dgrad_short(da, w, L, S, roll=roll, padL=padL, pack=pack)
    dx = torch.empty((B, N, L), device=da.device, dtype=torch.float32)
    halo = torch.empty((B * N * max(K - 1, 1),), device=da.device, dtype=torch.float32)
    lib = _lib.load()
    if small:
        check(lib.segan_conv1d_dgrad(_ptr(da), None, _ptr(w.detach()), _ptr(dx), _ptr(halo), B, N,
                                     M, L, K, S, padL, roll, PREC_FP32, None, 0, _stream()),
              'conv1d_dgrad')
        return dx
    if _precision:
        keep, sp, sn = _bf_scratch(1, B, N, M, L, K, S, padL, da.device)
        rc = lib.segan_conv1d_dgrad(_ptr(da), _ptr(pack.bf(w, S, 0, 1, _precision)), None, _ptr(dx),
                                    _ptr(halo), B, N, M, L, K, S, padL, roll, _precision, sp, sn,
                                    _stream())
        if rc != _EUNSUPPORTED:
            check(rc, 'conv1d_dgrad')
            return dx
    if short_rows_ok(N, M, L, S):        # a bf16 mode whose kernel does not cover this geometry
        return conv1d_dgrad_short(da, w, L, S, roll=roll, padL=padL, pack=pack)
    check(lib.segan_conv1d_dgrad(_ptr(da), _ptr(pack.t(w, S, 0)), None, _ptr(dx), _ptr(halo), B, N,
                                 M, L, K, S, padL, roll, _fp32(), *_scratch(), _stream()),
          'conv1d_dgrad')
    return dx


def conv1d_dgrad_short(da, w, L, S, roll=0, padL=None, pack=None):
    """conv1d_dgrad through segan_conv1d_dgrad_short (L/S in {4, 8, 16, 32, 64}, N % 4 == 0,
    M % 16 == 0); raises for other geometries."""
    _chk(da, 'da', 3)
    _chk(w, 'weight', 3)
    M, N, K = w.shape
    B = da.shape[0]
    if padL is None:
        padL = conv_pad(K, S)[0]
    dx = torch.empty((B, N, L), device=da.device, dtype=torch.float32)
    pack = pack or WeightPack()
    check(_lib.load().segan_conv1d_dgrad_short(_ptr(da), _ptr(pack.g(w, S)), _ptr(dx), B, N,
                                               M, L, K, S, padL, roll, _stream()),
          'conv1d_dgrad_short')
    return dx


_SHORT_LS = (4, 8, 16, 32, 64)


def short_rows_ok(N, M, L, S):
    """Does the fp32 conv data gradient of this geometry run the short-row kernel
    (segan_conv1d_dgrad_short)?"""
    return S in (2, 4) and L % S == 0 and (L // S) in _SHORT_LS and N % 4 == 0 and M % 16 == 0


def wgrad(lo, hi, dw, K, S, padL, pad_mode, roll=0):
    """dw[m,n,k] += sum lo[b,m,t] * pad(roll(hi))[b,n,S*t+k]  (accumulates into dw)."""
    _chk(dw, 'dw', 3)
    M, N = lo.C, hi.C
    if tuple(dw.shape) != (M, N, K):
        raise ValueError('wgrad: dw {} != ({}, {}, {})'.format(tuple(dw.shape), M, N, K))
    if lo.B != hi.B or lo.L * S != hi.L:
        raise ValueError('wgrad: lo [{}x{}] / hi [{}x{}] inconsistent for stride {}'.format(
            lo.B, lo.L, hi.B, hi.L, S))
    cl, ch = lo.c_struct(), hi.c_struct()
    lib = _lib.load()
    if _precision != PREC_FP32:
        nbytes = lib.segan_wgrad_scratch_bytes(lo.B, M, N, lo.L, S, _precision, 0)
        scratch = _stream_scratch(nbytes, dw.device)
        rc = lib.segan_wgrad(ctypes.byref(cl), ctypes.byref(ch), _ptr(dw), lo.B, M, N, lo.L, K, S,
                             padL, pad_mode, roll, _precision, 0,
                             ctypes.c_void_p(scratch.data_ptr()) if scratch is not None else None,
                             nbytes if scratch is not None else 0, _stream())
        if rc != -3:
            check(rc, 'wgrad')
            return
    flags = 1 if _deterministic else 0
    lo_plain = lo.t1 is None and lo.scale is None and lo.shift is None and lo.slope is None
    scratch, nbytes = None, 0
    if flags or not lo_plain:
        nbytes = lib.segan_wgrad_scratch_bytes(lo.B, M, N, lo.L, S, PREC_FP32, flags)
        scratch = _stream_scratch(nbytes, dw.device)
    check(lib.segan_wgrad(ctypes.byref(cl), ctypes.byref(ch), _ptr(dw), lo.B, M, N, lo.L, K, S, padL,
                          pad_mode, roll, PREC_FP32, flags,
                          ctypes.c_void_p(scratch.data_ptr()) if scratch is not None else None, nbytes,
                          _stream()), 'wgrad')


def deconv1d_fwd(src, w, bias, S, act=ACT_NONE, pack=None):
    """GDeconv1DBlock pre-activation (or tanh output): [B, N, S*Ls]."""
    M, N, K = w.shape
    if src.C != M:
        raise ValueError('deconv1d_fwd: input has {} channels, weight expects {}'.format(src.C, M))
    pad = deconv_pad(K, S)
    B, Ls = src.B, src.L
    y = torch.empty((B, N, S * Ls), device=w.device, dtype=torch.float32)
    cs = src.c_struct()
    small = N <= 2          # last generator layer (Cout = 1): direct VALU kernel
    _chk(w, 'weight', 3)
    pack = pack or WeightPack()
    lib = _lib.load()
    if small:
        check(lib.segan_deconv1d_fwd(ctypes.byref(cs), None, _ptr(w.detach()), _ptr(bias), _ptr(y),
                                     B, M, N, Ls, K, S, pad, act, PREC_FP32, None, 0, _stream()),
              'deconv1d_fwd')
        return y
    if _precision and act == ACT_NONE:
        keep, sp, sn = _bf_scratch(2, B, N, M, S * Ls, K, S, pad, w.device)
        rc = lib.segan_deconv1d_fwd(ctypes.byref(cs), _ptr(pack.bf(w, S, pad, 1, _precision)), None,
                                    _ptr(bias), _ptr(y), B, M, N, Ls, K, S, pad, act, _precision,
                                    sp, sn, _stream())
        if rc != _EUNSUPPORTED:
            check(rc, 'deconv1d_fwd')
            return y
    check(lib.segan_deconv1d_fwd(ctypes.byref(cs), _ptr(pack.t(w, S, pad)), None, _ptr(bias),
                                 _ptr(y), B, M, N, Ls, K, S, pad, act, _fp32(), *_scratch(),
                                 _stream()), 'deconv1d_fwd')
    return y


def deconv1d_dgrad(dy, w, S, M0=0, need0=True, need1=True, pack=None):
    """Gradient w.r.t. the deconv input, split at channel M0 into (dx0, dx1)."""
    _chk(dy, 'dy', 3)
    M, N, K = w.shape
    B = dy.shape[0]
    if dy.shape[1] != N or dy.shape[2] % S != 0:
        raise ValueError('deconv1d_dgrad: dy {} inconsistent with weight {}'.format(
            tuple(dy.shape), tuple(w.shape)))
    Ls = dy.shape[2] // S
    pad = deconv_pad(K, S)
    dx0 = dx1 = None
    if M0 > 0 and need0:
        dx0 = torch.empty((B, M0, Ls), device=dy.device, dtype=torch.float32)
    if M - M0 > 0 and need1:
        dx1 = torch.empty((B, M - M0, Ls), device=dy.device, dtype=torch.float32)
    if dx0 is None and dx1 is None:
        return None, None
    pack = pack or WeightPack()
    lib = _lib.load()
    if _precision:
        keep, sp, sn = _bf_scratch(3, B, N, M, S * Ls, K, S, pad, dy.device)
        rc = lib.segan_deconv1d_dgrad(_ptr(dy), _ptr(pack.bf(w, S, 0, 0, _precision)), _ptr(dx0),
                                      _ptr(dx1), B, M, M0, N, Ls, K, S, pad, _precision, sp, sn,
                                      _stream())
        if rc != _EUNSUPPORTED:
            check(rc, 'deconv1d_dgrad')
            return dx0, dx1
    check(lib.segan_deconv1d_dgrad(_ptr(dy), _ptr(pack.f(w, S)), _ptr(dx0), _ptr(dx1), B, M, M0, N,
                                   Ls, K, S, pad, _fp32(), *_scratch(), _stream()), 'deconv1d_dgrad')
    return dx0, dx1


# ---------------------------------------------------------------------------------
# per-channel kernels
# ---------------------------------------------------------------------------------
def _ws(B, C, L, per_split, extra, device):
    ns = _lib.load().segan_bn_nsplit(B, C, L)
    return torch.empty((per_split * ns * C + extra * C,), device=device, dtype=torch.float32)


def bn_stats(x, gamma, beta, eps, momentum, running_mean, running_var):
    """Training-mode BatchNorm1d statistics; returns (mean, rstd, scale, shift)."""
    _chk(x, 'x', 3)
    B, C, L = x.shape
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    scale = torch.empty_like(mean)
    shift = torch.empty_like(mean)
    ws = _ws(B, C, L, 3, 0, x.device)
    check(_lib.load().segan_bn_stats(_ptr(x), _ptr(gamma), _ptr(beta), eps, momentum,
                                     _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(rstd),
                                     _ptr(scale), _ptr(shift), _ptr(ws), B, C, L, _stream()),
          'bn_stats')
    return mean, rstd, scale, shift


def bn_partial(x):
    """This rank's BatchNorm partial statistics ws[nsplit, C, 3] = (count, mean, M2)."""
    _chk(x, 'x', 3)
    B, C, L = x.shape
    ns = _lib.load().segan_bn_nsplit(B, C, L)
    ws = torch.empty((ns, C, 3), device=x.device, dtype=torch.float32)
    check(_lib.load().segan_bn_partial(_ptr(x), _ptr(ws), B, C, L, _stream()), 'bn_partial')
    return ws


def bn_final(ws_all, gamma, beta, eps, momentum, running_mean, running_var):
    """Combine partial statistics [nsplit_total, C, 3] (all ranks); returns (mean, rstd, scale,
    shift) and updates the running statistics like bn_stats."""
    _chk(ws_all, 'ws_all', 3)
    C = ws_all.shape[1]
    mean = torch.empty(C, device=ws_all.device, dtype=torch.float32)
    rstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    check(_lib.load().segan_bn_final(_ptr(ws_all), ws_all.shape[0], _ptr(gamma), _ptr(beta), eps,
                                     momentum, _ptr(running_mean), _ptr(running_var), _ptr(mean),
                                     _ptr(rstd), _ptr(scale), _ptr(shift), C, _stream()), 'bn_final')
    return mean, rstd, scale, shift


def act_bwd_bn_reduce(a, dh, slope, bn, dslope=None, dgamma=None, dbeta=None):
    """First half of the BatchNorm backward: per-channel (sum g, sum g*xhat) of this rank's
    samples -> totals [C, 2]; accumulates dslope / dgamma / dbeta.  Returns (totals, ws)."""
    _chk(a, 'a', 3)
    B, C, L = a.shape
    mean, rstd, gamma, beta = bn
    totals = torch.empty((C, 2), device=a.device, dtype=torch.float32)
    ws = _ws(B, C, L, 4, 2, a.device)
    check(_lib.load().segan_act_bwd_bn_reduce(_ptr(a), _ptr(dh), _ptr(slope), _ptr(mean), _ptr(rstd),
                                              _ptr(gamma), _ptr(beta), _ptr(dslope), _ptr(dgamma),
                                              _ptr(dbeta), _ptr(totals), _ptr(ws), B, C, L, _stream()),
          'act_bwd_bn_reduce')
    return totals, ws


def act_bwd_bn_apply(a, dh, slope, bn, totals, count_total, dbias=None, ws=None):
    """Second half: da from the (globally summed) totals and the global element count."""
    _chk(a, 'a', 3)
    B, C, L = a.shape
    mean, rstd, gamma, beta = bn
    da = torch.empty_like(a)
    if ws is None:
        ws = _ws(B, C, L, 4, 2, a.device)
    check(_lib.load().segan_act_bwd_bn_apply(_ptr(a), _ptr(dh), _ptr(slope), _ptr(mean), _ptr(rstd),
                                             _ptr(gamma), _ptr(beta), _ptr(totals), _ptr(da),
                                             _ptr(dbias), _ptr(ws), B, C, L, float(count_total),
                                             _stream()), 'act_bwd_bn_apply')
    return da


def affine_prelu(x, scale=None, shift=None, slope=None):
    _chk(x, 'x', 3)
    B, C, L = x.shape
    y = torch.empty_like(x)
    check(_lib.load().segan_affine_prelu(_ptr(x), _ptr(scale), _ptr(shift), _ptr(slope), _ptr(y),
                                         B, C, L, _stream()), 'affine_prelu')
    return y


def affine_tanh(x, scale=None, shift=None):
    """tanh(x*scale + shift) with per-channel scale / shift on [B, C, L]."""
    _chk(x, 'x', 3)
    B, C, L = x.shape
    y = torch.empty_like(x)
    check(_lib.load().segan_affine_tanh(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), B, C, L,
                                        _stream()), 'affine_tanh')
    return y


def scale_mask(x, scale, mask):
    """x * scale[c] * mask on [B, C, L] (scale may be None): dropout on the skip path with the
    alpha scale folded in, and (scale None) its backward."""
    _chk(x, 'x', 3)
    _chk(mask, 'mask', 3)
    if mask.shape != x.shape:
        raise ValueError('scale_mask: mask {} vs x {}'.format(tuple(mask.shape), tuple(x.shape)))
    B, C, L = x.shape
    y = torch.empty_like(x)
    check(_lib.load().segan_scale_mask(_ptr(x), _ptr(scale), _ptr(mask), _ptr(y), B, C, L,
                                       _stream()), 'scale_mask')
    return y


def sum_skip(x0, slope0, x1, alpha):
    """prelu(x0, slope0) + alpha * x1 (GSkip merge_mode 'sum')."""
    _chk(x0, 'x0', 3)
    _chk(x1, 'x1', 3)
    if x0.shape != x1.shape:
        raise ValueError('sum_skip: shapes differ {} vs {}'.format(tuple(x0.shape), tuple(x1.shape)))
    B, C, L = x0.shape
    out = torch.empty_like(x0)
    check(_lib.load().segan_sum_skip(_ptr(x0), _ptr(slope0), _ptr(x1), _ptr(alpha), _ptr(out), B, C,
                                     L, _stream()), 'sum_skip')
    return out


def act_bwd(a, dh, dskip=None, slope=None, alpha=None, bn=None, dslope=None, dalpha=None,
            dgamma=None, dbeta=None, dbias=None):
    """Backward of (BN+)PReLU(+alpha skip) on pre-activation a; returns da.

    bn = (mean, rstd, gamma, beta) or None.  The d* tensors are accumulated into."""
    _chk(a, 'a', 3)
    B, C, L = a.shape
    da = torch.empty_like(a)
    ws = _ws(B, C, L, 4, 2, a.device)
    mean = rstd = gamma = beta = None
    if bn is not None:
        mean, rstd, gamma, beta = bn
    check(_lib.load().segan_act_bwd(_ptr(a), _ptr(dh), _ptr(dskip), _ptr(slope), _ptr(alpha),
                                    _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(da),
                                    _ptr(dslope), _ptr(dalpha), _ptr(dgamma), _ptr(dbeta),
                                    _ptr(dbias), _ptr(ws), B, C, L, _stream()), 'act_bwd')
    return da


def tanh_bwd(y, dy, clean=None, l1_scale=0.0, dbias=None):
    _chk(y, 'y', 3)
    B, C, L = y.shape
    da = torch.empty_like(y)
    ws = _ws(B, C, L, 1, 0, y.device)
    check(_lib.load().segan_tanh_bwd(_ptr(y), _ptr(dy), _ptr(clean), float(l1_scale), _ptr(da),
                                     _ptr(dbias), _ptr(ws), B, C, L, _stream()), 'tanh_bwd')
    return da


# ---------------------------------------------------------------------------------
# dense head
# ---------------------------------------------------------------------------------
def gemm(A, sam, sak, Bm, sbk, sbn, C, M, N, K, overwrite):
    lib = _lib.load()
    sp, sn = None, 0
    if _deterministic:      # split-K partials as slabs, added in split order
        sn = lib.segan_gemm_scratch_bytes(M, N, K)
        keep = torch.empty(sn, device=C.device, dtype=torch.uint8)
        sp = ctypes.c_void_p(keep.data_ptr())
    check(lib.segan_gemm(_ptr(A), sam, sak, _ptr(Bm), sbk, sbn, _ptr(C), C.stride(0), M, N,
                         K, 1 if overwrite else 0, 1 if _deterministic else 0, sp, sn, _stream()),
          'gemm')


def linear_fwd(x, w):
    """x [B, I] @ w[O, I]^T -> [B, O] (no bias)."""
    _chk(x, 'x', 2)
    _chk(w, 'w', 2)
    Bn, I = x.shape
    O = w.shape[0]
    y = torch.empty((Bn, O), device=x.device, dtype=torch.float32)
    gemm(x, I, 1, w, 1, I, y, Bn, O, I, True)
    return y


def linear_dgrad(dy, w):
    """dy [B, O] @ w[O, I] -> [B, I]."""
    Bn, O = dy.shape
    I = w.shape[1]
    dx = torch.empty((Bn, I), device=dy.device, dtype=torch.float32)
    gemm(dy, O, 1, w, I, 1, dx, Bn, I, O, True)
    return dx


def linear_wgrad(dy, x, dw):
    """dw[O, I] += dy[B, O]^T @ x[B, I]."""
    Bn, O = dy.shape
    I = x.shape[1]
    gemm(dy, 1, O, x, I, 1, dw, O, I, Bn, False)


def bias_prelu_rows(x, bias, slope):
    _chk(x, 'x', 2)
    y = torch.empty_like(x)
    check(_lib.load().segan_bias_prelu_rows(_ptr(x), _ptr(bias), _ptr(slope), _ptr(y), x.shape[0],
                                            x.shape[1], _stream()), 'bias_prelu_rows')
    return y


def bias_prelu_rows_bwd(x, bias, slope, dy, dslope, dbias):
    dx = torch.empty_like(x)
    check(_lib.load().segan_bias_prelu_rows_bwd(_ptr(x), _ptr(bias), _ptr(slope), _ptr(dy),
                                                _ptr(dx), _ptr(dslope), _ptr(dbias), x.shape[0],
                                                x.shape[1], _stream()), 'bias_prelu_rows_bwd')
    return dx


# ---------------------------------------------------------------------------------
# losses / optimizers / utilities
# ---------------------------------------------------------------------------------
def pool_time_fwd(x, mode):
    """Global pooling over time of x [B, C, L] -> ([B, C], idx): mode 'max' (idx = first argmax,
    int32 [B, C]) or 'avg' (idx None).  AdaptiveMaxPool1d(1) / AdaptiveAvgPool1d(1) of the
    'gmax' / 'gavg' discriminator heads (discriminator.py:128-137)."""
    _chk(x, 'x', 3)
    B, C, L = x.shape
    y = torch.empty((B, C), device=x.device, dtype=torch.float32)
    idx = torch.empty((B, C), device=x.device, dtype=torch.int32) if mode == 'max' else None
    check(_lib.load().segan_pool_time_fwd(_ptr(x), _ptr(y), _ptr(idx), B * C, L,
                                          0 if mode == 'max' else 1, _stream()), 'pool_time_fwd')
    return y, idx


def pool_time_bwd(dy, idx, L, mode):
    """Gradient of pool_time_fwd w.r.t. x: [B, C, L]."""
    _chk(dy, 'dy', 2)
    B, C = dy.shape
    dx = torch.empty((B, C, L), device=dy.device, dtype=torch.float32)
    check(_lib.load().segan_pool_time_bwd(_ptr(dy), _ptr(idx), _ptr(dx), B * C, L,
                                          0 if mode == 'max' else 1, _stream()), 'pool_time_bwd')
    return dx


def mse_const(x, target):
    """mean((x - target)^2) for a constant target (LSGAN labels, model.py:298,305,316)."""
    _chk(x, 'x')
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    check(_lib.load().segan_mse_const(_ptr(x), float(target), _ptr(loss), None, None, 1.0,
                                      x.numel(), _stream()), 'mse_const')
    return loss


def mse_const_bwd(x, target, gout=None, gscale=1.0):
    """Gradient of mse_const times the (device-resident) upstream scalar gout."""
    _chk(x, 'x')
    grad = torch.empty_like(x)
    check(_lib.load().segan_mse_const(_ptr(x), float(target), None, _ptr(grad), _ptr(gout),
                                      float(gscale), x.numel(), _stream()), 'mse_const_bwd')
    return grad


def bce_logits_const(x, target):
    _chk(x, 'x')
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    check(_lib.load().segan_bce_logits_const(_ptr(x), float(target), _ptr(loss), None, None, 1.0,
                                             x.numel(), _stream()), 'bce_logits_const')
    return loss


def bce_logits_const_bwd(x, target, gout=None, gscale=1.0):
    _chk(x, 'x')
    grad = torch.empty_like(x)
    check(_lib.load().segan_bce_logits_const(_ptr(x), float(target), None, _ptr(grad), _ptr(gout),
                                             float(gscale), x.numel(), _stream()),
          'bce_logits_const_bwd')
    return grad


def l1_bwd(x, y, gout=None, gscale=1.0):
    _chk(x, 'x')
    grad = torch.empty_like(x)
    check(_lib.load().segan_l1_bwd(_ptr(x), _ptr(y), _ptr(gout), float(gscale), _ptr(grad),
                                   x.numel(), _stream()), 'l1_bwd')
    return grad


def l1_mean(x, y):
    _chk(x, 'x')
    _chk(y, 'y')
    if x.shape != y.shape:
        raise ValueError('l1_mean: shapes differ {} vs {}'.format(tuple(x.shape), tuple(y.shape)))
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    ws = torch.empty(1024, device=x.device, dtype=torch.float32)
    check(_lib.load().segan_l1_mean(_ptr(x), _ptr(y), _ptr(loss), _ptr(ws), x.numel(), _stream()),
          'l1_mean')
    return loss


def mse_mean(x, y):
    """mean((x - y)^2) of two tensors (F.mse_loss; --reg_loss mse_loss)."""
    _chk(x, 'x')
    _chk(y, 'y')
    if x.shape != y.shape:
        raise ValueError('mse_mean: shapes differ {} vs {}'.format(tuple(x.shape), tuple(y.shape)))
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    ws = torch.empty(1024, device=x.device, dtype=torch.float32)
    check(_lib.load().segan_mse_mean(_ptr(x), _ptr(y), _ptr(loss), _ptr(ws), x.numel(), _stream()),
          'mse_mean')
    return loss


def mse_bwd(x, y, gout=None, gscale=1.0):
    _chk(x, 'x')
    grad = torch.empty_like(x)
    check(_lib.load().segan_mse_bwd(_ptr(x), _ptr(y), _ptr(gout), float(gscale), _ptr(grad),
                                    x.numel(), _stream()), 'mse_bwd')
    return grad


# ---- STFT power loss (model.py:640-653) ----------------------------------------------------
_stft_basis_cache = {}


def stft_pitch(n_fft):
    """Row width of the basis / spectra: 2*(n_fft/2+1) rounded up to a multiple of 4."""
    return (2 * (n_fft // 2 + 1) + 3) // 4 * 4


def stft_basis(n_fft, win, device):
    """[win, pitch] real DFT basis of a centred rectangular window (cached): columns
    [0, nbins) real part, [nbins, 2*nbins) imaginary part, zero padding up to the pitch."""
    key = (n_fft, win, str(device))
    if key not in _stft_basis_cache:
        bs = torch.empty((win, stft_pitch(n_fft)), device=device, dtype=torch.float32)
        check(_lib.load().segan_stft_basis(_ptr(bs), n_fft, win, _stream()), 'stft_basis')
        _stft_basis_cache[key] = bs
    return _stft_basis_cache[key]


def stft_frames(x, n_fft, hop, win):
    """x [B, T] -> frames [B*NF, win] of the reflect-padded signal."""
    _chk(x, 'x', 2)
    B, T = x.shape
    NF = 1 + T // hop
    fr = torch.empty((B * NF, win), device=x.device, dtype=torch.float32)
    check(_lib.load().segan_stft_frames(_ptr(x), _ptr(fr), B, T, n_fft, hop, win, _stream()),
          'stft_frames')
    return fr


def stft_spectrum(frames, basis):
    """frames [R, win] @ basis [win, pitch] -> [R, pitch] (real | imaginary | zero pad)."""
    R, win = frames.shape
    N2 = basis.shape[1]
    S = torch.empty((R, N2), device=frames.device, dtype=torch.float32)
    gemm(frames, win, 1, basis, N2, 1, S, R, N2, win, True)
    return S


def stft_spectrum_bwd(dS, basis):
    """dS [R, pitch] @ basis^T -> dframes [R, win]."""
    R, N2 = dS.shape
    win = basis.shape[0]
    df = torch.empty((R, win), device=dS.device, dtype=torch.float32)
    gemm(dS, N2, 1, basis, 1, N2, df, R, win, N2, True)
    return df


def powdb(S, nbins, eps=10e-20):
    _chk(S, 'S', 2)
    rows, pitch = S.shape
    db = torch.empty((rows, nbins), device=S.device, dtype=torch.float32)
    check(_lib.load().segan_powdb(_ptr(S), _ptr(db), rows, nbins, pitch, eps, _stream()), 'powdb')
    return db


def powdb_bwd(S, ddb, nbins, eps=10e-20):
    _chk(S, 'S', 2)
    _chk(ddb, 'ddb', 2)
    rows, pitch = S.shape
    dS = torch.empty_like(S)
    check(_lib.load().segan_powdb_bwd(_ptr(S), _ptr(ddb), _ptr(dS), rows, nbins, pitch, eps,
                                      _stream()), 'powdb_bwd')
    return dS


def stft_overlap_add(dframes, B, T, n_fft, hop, win):
    _chk(dframes, 'dframes', 2)
    dx = torch.empty((B, T), device=dframes.device, dtype=torch.float32)
    check(_lib.load().segan_stft_overlap_add(_ptr(dframes), _ptr(dx), B, T, n_fft, hop, win,
                                             _stream()), 'stft_overlap_add')
    return dx


# ---- spectral normalisation -----------------------------------------------------------------
def _sn_view(w, dim):
    """[A, Bd, K] view of a weight for the C ABI: Conv1d / ConvTranspose1d [A, Bd, K], Linear
    [A, Bd] (K = 1), PReLU [A] (Bd = K = 1)."""
    shp = tuple(w.shape) + (1,) * (3 - w.dim())
    if w.dim() > 3 or dim not in (0, 1) or (dim == 1 and w.dim() < 2):
        raise ValueError('snorm: unsupported weight shape {} / dim {}'.format(tuple(w.shape), dim))
    return shp


def snorm_fwd(w, u, v, dim, power_iteration, eps=1e-12):
    """(w / sigma, sigma) of torch.nn.utils.spectral_norm; u, v are updated in place when
    `power_iteration` (training mode)."""
    _chk(w, 'w')
    _chk(u, 'u', 1)
    _chk(v, 'v', 1)
    A, Bd, K = _sn_view(w, dim)
    rows, cols = (A, Bd * K) if dim == 0 else (Bd, A * K)
    if u.numel() != rows or v.numel() != cols:
        raise ValueError('snorm: u[{}] / v[{}] do not match the {}x{} matrix view'.format(
            u.numel(), v.numel(), rows, cols))
    lib = _lib.load()
    w_sn = torch.empty_like(w)
    sigma = torch.empty(1, device=w.device, dtype=torch.float32)
    ws = torch.empty(lib.segan_snorm_ws_floats(A, Bd, K, dim), device=w.device, dtype=torch.float32)
    check(lib.segan_snorm_fwd(_ptr(w), _ptr(u), _ptr(v), _ptr(w_sn), _ptr(sigma), _ptr(ws), A, Bd, K,
                              dim, 1 if power_iteration else 0, float(eps), _stream()), 'snorm_fwd')
    return w_sn, sigma


def snorm_bwd(dw_sn, w, u, v, sigma, dim, dw):
    """dw += d(w/sigma)^T dw_sn with u, v held constant (how torch differentiates it)."""
    _chk(dw_sn, 'dw_sn')
    _chk(dw, 'dw')
    A, Bd, K = _sn_view(w, dim)
    lib = _lib.load()
    ws = torch.empty(lib.segan_snorm_ws_floats(A, Bd, K, dim), device=w.device, dtype=torch.float32)
    check(lib.segan_snorm_bwd(_ptr(dw_sn), _ptr(w), _ptr(u), _ptr(v), _ptr(sigma), _ptr(dw), _ptr(ws),
                              A, Bd, K, dim, _stream()), 'snorm_bwd')


def pcm16_prep(pcm, first, coef):
    """int16 slices [B, 2, T+1] (+ first[B] uint8) -> (clean, noisy) fp32 [B, T]: min-max
    normalisation and pre-emphasis of se_dataset.py:108-117 on the GPU, bit-exact."""
    if pcm.dtype != torch.int16 or first.dtype != torch.uint8 or not pcm.is_cuda or not first.is_cuda:
        raise TypeError('pcm16_prep: pcm must be a CUDA int16 tensor and first a CUDA uint8 tensor')
    if pcm.dim() != 3 or pcm.shape[1] != 2 or not pcm.is_contiguous() or first.numel() != pcm.shape[0]:
        raise ValueError('pcm16_prep: pcm must be contiguous [B, 2, T+1], first [B]')
    B, T = pcm.shape[0], pcm.shape[2] - 1
    clean = torch.empty((B, T), device=pcm.device, dtype=torch.float32)
    noisy = torch.empty((B, T), device=pcm.device, dtype=torch.float32)
    check(_lib.load().segan_pcm16_prep(ctypes.c_void_p(pcm.data_ptr()),
                                       ctypes.c_void_p(first.data_ptr()), _ptr(clean), _ptr(noisy),
                                       B, T, float(coef), _stream()), 'pcm16_prep')
    return clean, noisy


def de_emphasize(y, coef):
    """x[n] = coef*x[n-1] + y[n] along the last axis of a CUDA tensor [..., T] (se_dataset.py:
    119-126 on the device)."""
    _chk(y, 'y')
    T = y.shape[-1]
    x = torch.empty_like(y)
    check(_lib.load().segan_deemphasis(_ptr(y), _ptr(x), y.numel() // T, T, float(coef), _stream()),
          'deemphasis')
    return x


def ssnr(ref, deg, srate=16000, eps=1e-10):
    """Segmental SNR of utils.py:350-395 per row of ref / deg [rows, T] on the device.  Returns
    (overall_snr[rows], mean_segmental_snr[rows], segmental[rows, nframes])."""
    _chk(ref, 'ref', 2)
    _chk(deg, 'deg', 2)
    if ref.shape != deg.shape:
        raise ValueError('ssnr: shapes differ {} vs {}'.format(tuple(ref.shape), tuple(deg.shape)))
    rows, T = ref.shape
    lib = _lib.load()
    nf = lib.segan_ssnr_frames(T, srate)
    seg = torch.empty((rows, max(nf, 1)), device=ref.device, dtype=torch.float32)
    out = torch.empty((rows, 2), device=ref.device, dtype=torch.float32)
    check(lib.segan_ssnr(_ptr(ref), _ptr(deg), _ptr(seg), _ptr(out), rows, T, srate, float(eps),
                         _stream()), 'ssnr')
    return out[:, 0], out[:, 1], seg[:, :nf]


def rmsprop_step(p, g, sq, lr, alpha, eps):
    check(_lib.load().segan_rmsprop_step(_ptr(p), _ptr(g), _ptr(sq), lr, alpha, eps, p.numel(),
                                         _stream()), 'rmsprop_step')


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    check(_lib.load().segan_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), lr, beta1, beta2, eps,
                                      step, p.numel(), _stream()), 'adam_step')


def fill_(t, value):
    _chk(t, 't')
    check(_lib.load().segan_fill(_ptr(t), float(value), t.numel(), _stream()), 'fill')
    return t


def scale_(t, s):
    _chk(t, 't')
    check(_lib.load().segan_scale(_ptr(t), float(s), t.numel(), _stream()), 'scale')
    return t


# ---- data-parallel exchange through the C ABI (RCCL bound at run time) ------------------------
def comm_unique_id():
    """Rendezvous id (bytes) for `Comm`: create on rank 0, ship to the other ranks."""
    lib = _lib.load()
    buf = ctypes.create_string_buffer(lib.segan_comm_id_bytes())
    check(lib.segan_comm_unique_id(buf), 'comm_unique_id')
    return buf.raw


class Comm(object):
    """One RCCL communicator owned by libsegan_hip (segan_comm_init ... segan_comm_destroy): the
    data-parallel exchange of the GAN step without torch.distributed in the data path.  Creation
    is a collective over all ranks; the current CUDA device becomes the communicator's."""

    def __init__(self, world, rank, unique_id):
        lib = _lib.load()
        if len(unique_id) != lib.segan_comm_id_bytes():
            raise ValueError('Comm: unique id must be {} bytes'.format(lib.segan_comm_id_bytes()))
        h = ctypes.c_void_p()
        check(lib.segan_comm_init(ctypes.byref(h), int(world), int(rank), unique_id), 'comm_init')
        self._h, self.world, self.rank = h, int(world), int(rank)

    def _s(self, stream):
        return ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)

    def allreduce(self, t, scale=1.0, stream=None):
        """In-place sum over ranks of a contiguous fp32 CUDA tensor, then * scale."""
        _chk(t, 't')
        check(_lib.load().segan_allreduce(self._h, _ptr(t), t.numel(), float(scale), self._s(stream)),
              'allreduce')
        return t

    def broadcast(self, t, root=0, stream=None):
        _chk(t, 't')
        check(_lib.load().segan_broadcast(self._h, _ptr(t), t.numel(), int(root), self._s(stream)),
              'broadcast')
        return t

    def allgather(self, send, stream=None):
        """[world, *send.shape] <- every rank's `send`."""
        _chk(send, 'send')
        recv = torch.empty((self.world,) + tuple(send.shape), device=send.device, dtype=torch.float32)
        check(_lib.load().segan_allgather(self._h, _ptr(send), _ptr(recv), send.numel(),
                                          self._s(stream)), 'allgather')
        return recv

    def destroy(self):
        if self._h is not None:
            check(_lib.load().segan_comm_destroy(self._h), 'comm_destroy')
            self._h = None
