"""Autograd glue: whole-network ``torch.autograd.Function``s that drive the HIP kernels.

The generator and the discriminator are each ONE autograd node.  Inside a node the
layers exchange *pre-activations*; PReLU, BatchNorm-normalise, the alpha skip scale,
the z / skip concatenations and the discriminator's phase shift are applied by the
consuming kernel while it stages its tile (``ops.Src``), so the reference's
intermediate tensors (h, sk_h, torch.cat outputs, padded/rolled copies) never exist.

Parameter gradients are accumulated by the kernels directly into ``param.grad``
(allocated zero-filled on first use) — exactly torch's accumulate-into-.grad
semantics, minus the temporary — and the nodes return ``None`` for parameter inputs.

Reference semantics followed: segan/models/generator.py:180-230,
segan/models/discriminator.py:150-194, segan/models/modules.py:91-141.
"""
import os

import torch

from . import ops
from .ops import ACT_NONE, ACT_TANH, PAD_REFLECT, PAD_ZERO, Src


def grad_buf(p):
    """The tensor the kernels accumulate this parameter's gradient into."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    elif not p.grad.is_contiguous():
        p.grad = p.grad.contiguous()
    return p.grad


def _gb(p, needed=True):
    if p is None or not needed or not p.requires_grad:
        return None
    return grad_buf(p)


def _bn_batch_stats(bn, training):
    """nn.BatchNorm1d.forward's rule: batch statistics in training mode — and in eval mode too when
    the module tracks no running statistics (track_running_stats=False: the buffers are None)."""
    return training or bn.running_mean is None or bn.running_var is None


def _bn_train_stats(c, bn, training=True):
    """BatchNorm batch statistics of pre-activation c: over this rank's batch, or over the
    global batch when synchronised BatchNorm is on (distributed.sync_bn_enabled).  In training
    mode with tracked statistics the batch counter is bumped FIRST and the running statistics move
    by torch's exponential_average_factor: bn.momentum, or — momentum=None — the cumulative average
    1 / num_batches_tracked (one host read of the counter: a corner no train.py flag reaches).
    Without tracked statistics (or in eval mode without them) nothing but the batch statistics is
    touched."""
    from . import distributed as sdist
    fn = sdist.bn_stats_sync if sdist.sync_bn_enabled() else ops.bn_stats
    track = training and bn.running_mean is not None and bn.running_var is not None
    momentum = 0.0
    if track:
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        if bn.momentum is not None:
            momentum = bn.momentum
        else:
            momentum = 1.0 / float(bn.num_batches_tracked) if bn.num_batches_tracked is not None else 0.0
    return fn(c, bn.weight, bn.bias, bn.eps, momentum, bn.running_mean if track else None,
              bn.running_var if track else None)


def _bn_eval_affine(bn):
    """(scale, shift, rstd) of an eval-mode BatchNorm with tracked statistics: a fixed per-channel
    affine map of the running statistics."""
    rstd = torch.rsqrt(bn.running_var + bn.eps)
    gamma = bn.weight.detach() if bn.weight is not None else torch.ones_like(rstd)
    beta = bn.bias.detach() if bn.bias is not None else torch.zeros_like(rstd)
    scale = (gamma * rstd).contiguous()
    shift = (beta - bn.running_mean * scale).contiguous()
    return scale, shift, rstd.contiguous()


def _eval_bn_act_bwd(c, bn, scale, shift, rstd, dh, slope, dslope, y_tanh=None, dbias=None):
    """Backward through act(bn_eval(c)) of a stand-alone block whose BatchNorm is in eval() — a fixed
    affine map, which autograd differentiates like any other (fine-tuning with frozen statistics;
    round-5 advice: this used to raise).  g = dL/d bn(c) through the PReLU (or the Tanh, from its
    output `y_tanh`), dbeta += sum g, dgamma += sum g * xhat with xhat = (c - mean) * rstd of the
    RUNNING statistics, dL/dc = g * scale, and the conv bias takes sum dL/dc.  The two per-channel
    sums for gamma run on ATen reductions: a path the reference's training never takes."""
    if y_tanh is not None:
        g = ops.tanh_bwd(y_tanh, dh, dbias=_gb(bn.bias))
    else:
        v = ops.affine_prelu(c, scale, shift, None)
        g = ops.act_bwd(v, dh, slope=slope, dslope=dslope, dbias=_gb(bn.bias))
    if bn.weight is not None and bn.weight.requires_grad:
        xhat = ops.affine_prelu(c, rstd, (-bn.running_mean * rstd).contiguous(), None)
        grad_buf(bn.weight).add_((g * xhat).sum((0, 2)))
    dc = ops.affine_prelu(g, scale, None, None)
    if dbias is not None:
        ops.act_bwd(dc, dc, dbias=dbias)          # += sum over (b, t)
    return dc


def _act_bwd_bn(a, dh, slope, bn, dslope, dgamma, dbeta, dbias):
    """Backward of BN + PReLU; the per-channel sums are all-reduced when BatchNorm is synced."""
    from . import distributed as sdist
    if sdist.sync_bn_enabled():
        return sdist.act_bwd_bn_sync(a, dh, slope, bn, dslope, dgamma, dbeta, dbias)
    return ops.act_bwd(a, dh, slope=slope, bn=bn, dslope=dslope, dgamma=dgamma, dbeta=dbeta,
                       dbias=dbias)


# Weight gradients on a SIDE stream (round 4, OPT-IN: SEGAN_WGRAD_OVERLAP=1).  Inside a backward
# pass the chain
#   act_bwd(l) -> data gradient(l) -> act_bwd(l-1) -> ...
# is serial, the weight gradient of layer l hangs off it: it needs da(l) and nothing needs IT before
# the optimizer step.  Launched on a second stream it runs beside the chain — beside the next
# data gradient (both on the matrix cores: nothing gained, nothing lost) and beside the next
# act_bwd / BatchNorm-backward passes, which are HBM-bound and leave the matrix cores idle when
# they run alone.  Measured (scripts/r04_overlap_ab.sh, alternating runs on one box): fp32 step
# 87.32 -> 87.10 ms (0.25 %), bf16 24.45 -> 23.92 ms (2.2 %).  The fp32 kernels fill the register
# file (4 x 124 / 3 x 150 VGPRs per SIMD), so a pointwise kernel only gets a slot where a contraction
# workgroup retires: the overlap is the tails, not whole kernels; launching the data gradient
# first changes nothing.  NOT the default: two contraction kernels sharing the machine each take
# about twice as long per launch, so every per-kernel figure (bench.py's event-timed `roofline`
# blocks, a rocprofv3 kernel trace) reads half the rate the kernel has — 0.25 % of the headline is
# not worth measurements that need a footnote.
_WGRAD_OVERLAP = os.environ.get('SEGAN_WGRAD_OVERLAP', '0') == '1'
_side_streams = {}


class _SideStream(object):
    """with _SideStream(t1, t2, ...): launches go to the device's side stream, ordered after
    everything enqueued so far on the current one; the tensors are marked as used there (the
    caching allocator must not hand their memory out again before the side stream is done)."""

    def __init__(self, *tensors):
        # an ops.Src stands for its segments and per-channel vectors: every operand the side
        # stream reads is recorded, not only the gradient (round-4 advice) — the backward passes
        # below also keep ctx.state alive until _join_side, but the allocator no longer depends on it
        flat = []
        for t in tensors:
            if isinstance(t, Src):
                flat.extend((t.t0, t.t1, t.scale, t.shift, t.slope))
            else:
                flat.append(t)
        self.tensors = [t for t in flat if torch.is_tensor(t) and t.is_cuda]
        self.on = _WGRAD_OVERLAP and bool(self.tensors)

    def __enter__(self):
        if not self.on:
            return self
        dev = self.tensors[0].device
        main = torch.cuda.current_stream(dev)
        key = (dev.index, main.cuda_stream)
        side = _side_streams.get(key)
        if side is None:
            side = _side_streams[key] = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        self.side = side
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if not self.on:
            return False
        self.ctx.__exit__(*exc)
        for t in self.tensors:
            t.record_stream(self.side)
        return False


def _join_side(ref):
    """The current stream waits for the side stream's weight gradients (end of a backward pass)."""
    if not _WGRAD_OVERLAP or ref is None or not ref.is_cuda:
        return
    main = torch.cuda.current_stream(ref.device)
    side = _side_streams.get((ref.device.index, main.cuda_stream))
    if side is not None:
        main.wait_stream(side)


def _ready(*mods_or_params):
    """Report the parameters of finished layers to the data-parallel gradient reducer."""
    from . import distributed as sdist
    if sdist._active is None:
        return
    ps = []
    for m in mods_or_params:
        if m is None:
            continue
        if isinstance(m, torch.nn.Module):
            ps.extend(torch.nn.Module.parameters(m))
        else:
            ps.append(m)
    sdist.grad_ready(*ps)


def _ones(n, ref):
    return torch.ones(n, device=ref.device, dtype=torch.float32)


def _cat(a, b):
    return torch.cat((a.detach().reshape(-1), b.detach().reshape(-1)))


# =====================================================================================
# effective weights (spectral normalisation)
# =====================================================================================
import itertools as _it

_sn_epoch = _it.count(1 << 40)


def _is_sn(mod):
    return hasattr(mod, 'weight_orig')


class _Weights(object):
    """Effective weights of ONE forward call.  For a plain parameter container (the nn.Conv1d /
    nn.ConvTranspose1d / nn.Linear / nn.PReLU children) that is its ``.weight``.  For a
    spectrally-normalised one (``torch.nn.utils.spectral_norm`` registration: ``weight_orig``
    parameter, ``weight_u`` / ``weight_v`` buffers; modules.py:12-14, discriminator.py:118-121)
    it is ``weight_orig / sigma``, computed by ``ops.snorm_fwd`` with the power iteration of
    training mode (u, v updated in place, as torch does under no_grad) and remembered with the
    (u, v, sigma) it used, so the backward of THIS call differentiates through exactly that
    normalisation even when D runs several forwards before one backward (model.py:577-631)."""

    def __init__(self):
        self._sn = {}

    def get(self, mod):
        if not _is_sn(mod):
            return mod.weight
        st = self._sn.get(id(mod))
        if st is None:
            dim = 1 if isinstance(mod, torch.nn.ConvTranspose1d) else 0
            w_sn, sigma = ops.snorm_fwd(mod.weight_orig.detach(), mod.weight_u, mod.weight_v, dim,
                                        mod.training)
            w_sn._segan_epoch = next(_sn_epoch)      # never aliases a cached weight pack
            st = (w_sn, mod.weight_u.clone(), mod.weight_v.clone(), sigma, dim)
            self._sn[id(mod)] = st
        return st[0]

    @staticmethod
    def param(mod):
        return mod.weight_orig if _is_sn(mod) else mod.weight

    def needs_grad(self, mod):
        return self.param(mod).requires_grad

    def grad_target(self, mod):
        """Buffer the weight-gradient kernel accumulates into: the parameter's .grad, or for a
        normalised weight a zeroed temporary that `finish` folds into weight_orig.grad."""
        if not _is_sn(mod):
            return grad_buf(mod.weight)
        return torch.zeros_like(self._sn[id(mod)][0])

    def finish(self, mod, tmp):
        if not _is_sn(mod):
            return
        w_sn, u, v, sigma, dim = self._sn[id(mod)]
        ops.snorm_bwd(tmp, mod.weight_orig.detach(), u, v, sigma, dim, grad_buf(mod.weight_orig))


def _skip_conv_bwd(gskip, dsk, src_j, W):
    """Backward of GSkip 'conv' (generator.py:42-49,65-66), sk = conv1d(h_j, Wk, bk, stride 1,
    zero padding kw//2) with h_j given as a Src (the encoder's linear output, normalised on load
    when the block has a BatchNorm): accumulates dWk / dbk, returns the gradient w.r.t. h_j.
    The data gradient of a stride-1 zero-padded conv is the same kernel run with the weight
    transposed and flipped (padding kw - 1 - kw//2).  `W` holds the effective weights of the
    forward call (the spectrally normalised one when the conv carries a spectral norm); the
    flipped copy and its packed form are cached on the module until the weight changes."""
    mod = gskip.skip_k
    kw = mod.kernel_size[0]
    w = W.get(mod)
    if mod.bias is not None and mod.bias.requires_grad:
        ops.act_bwd(dsk, dsk, dbias=grad_buf(mod.bias))          # dbk += sum over (b, t)
    if W.needs_grad(mod):
        gw = W.grad_target(mod)
        ops.wgrad(Src(dsk), src_j, gw, kw, 1, kw // 2, PAD_ZERO)
        W.finish(mod, gw)
    key = (w.data_ptr(), w._version, ops._weights_epoch, getattr(w, '_segan_epoch', 0))
    cached = gskip.__dict__.get('_wt_cache')
    if cached is None or cached[0] != key:
        wt = w.detach().transpose(0, 1).flip(2).contiguous()
        wt._segan_epoch = next(_sn_epoch)    # a new tensor at a recycled address is a new weight
        cached = (key, wt)
        gskip.__dict__['_wt_cache'] = cached
    return ops.conv1d_fwd(Src(dsk), cached[1], None, 1, pad_mode=PAD_ZERO, padL=kw - 1 - kw // 2,
                          pack=gskip._pack_t)


def _block_norm(blk, c, training):
    """BatchNorm of a generator block applied to its conv output c (build_norm_layer 'bnorm',
    modules.py:9-11): (scale, shift) for the consumers' on-load transform and what the backward
    needs — (mean, rstd, gamma, beta) in training mode, 'eval' with the running statistics —
    or (None, None, None) for a block without one."""
    bn = getattr(blk, 'norm', None)
    if bn is None:
        return None, None, None
    if _bn_batch_stats(bn, training):
        mean, rstd, scale, shift = _bn_train_stats(c, bn, training)
        return scale, shift, (mean, rstd, bn.weight, bn.bias)
    scale, shift, _rstd = _bn_eval_affine(bn)
    return scale, shift, 'eval'


def _vec2(v0, n0, v1, n1, fill, ref):
    """Per-channel vector over two concatenated channel groups; None stands for `fill`
    everywhere in a group, and the result is None when both groups are."""
    if v0 is None and v1 is None:
        return None
    parts = []
    for v, n in ((v0, n0), (v1, n1)):
        parts.append(v.detach().reshape(-1) if v is not None else
                     torch.full((n,), fill, device=ref.device, dtype=torch.float32))
    return torch.cat(parts)


def _dropout_mask(shape, p, device):
    """The noise nn.Dropout multiplies with (0 or 1/(1-p)), drawn from torch's global CPU
    generator exactly as the reference's CPU run draws it (F.dropout of a same-shaped tensor),
    then moved to the device."""
    noise = torch.nn.functional.dropout(torch.ones(shape), p, True)
    return noise.to(device)


# =====================================================================================
# Generator
# =====================================================================================
class GeneratorFn(torch.autograd.Function):
    """y = G(x, z) for an encoder/decoder generator (generator.py:180-230)."""

    @staticmethod
    def forward(ctx, gen, want_hid, x, z, *params):
        x = x.contiguous()
        enc, dec = list(gen.enc_blocks), list(gen.dec_blocks)
        n_enc = len(enc)
        training = gen.training
        a_enc, src_enc, xf_enc, bn_enc = [], [], [], []
        src = Src(x)
        W = _Weights()
        for blk in enc:
            a = ops.conv1d_fwd(src, W.get(blk.conv), blk.conv.bias, blk.stride, pack=blk._pack)
            sc, sh, bnsv = _block_norm(blk, a, training)
            src_enc.append(src)
            a_enc.append(a)
            xf_enc.append((sc, sh))
            bn_enc.append(bnsv)
            src = Src(a, scale=sc, shift=sh, slope=blk.act.weight)
        last = a_enc[-1]
        s_last = enc[-1].act.weight
        sc_l, sh_l = xf_enc[-1]
        if not gen.no_z:
            z = z.contiguous()
            nz, nl = z.shape[1], last.shape[1]
            src = Src(z, last, scale=_vec2(None, nz, sc_l, nl, 1.0, z),
                      shift=_vec2(None, nz, sh_l, nl, 0.0, z), slope=_cat(_ones(nz, z), s_last))
        else:
            src = Src(last, scale=sc_l, shift=sh_l, slope=s_last)
        a_dec, src_dec, xf_dec, bn_dec = [], [], [], []
        skip_saved = {}     # enc index -> (dropout mask or None, materialised bn(a_j) or None)
        enc_idx = n_enc - 1
        for li, blk in enumerate(dec):
            is_conv = hasattr(blk, 'conv')       # pooling 1: a GConv1DBlock (generator.py:171-176)
            if li > 0:
                prev = a_dec[-1]
                s_prev = dec[li - 1].act.weight
                sc_p, sh_p = xf_dec[-1]
                np_ = prev.shape[1]
                if gen.skip and enc_idx in gen.skips and gen.dec_poolings[li] > 1:
                    gskip = gen.skips[enc_idx]['alpha']
                    aj = a_enc[enc_idx]
                    sc_j, sh_j = xf_enc[enc_idx]
                    nj = aj.shape[1]
                    ones_j = _ones(nj, aj)
                    drop_p = gskip.skip_dropout.p if (training and hasattr(gskip, 'skip_dropout')) \
                        else 0.0
                    a_norm = None
                    if gskip.skip_type == 'conv':
                        # GSkip 'conv' (generator.py:42-49,65-66): a stride-1 zero-padded conv of
                        # the encoder's linear output, materialised; the merge stays fused
                        kw = gskip.skip_k.kernel_size[0]
                        sk = ops.conv1d_fwd(Src(aj, scale=sc_j, shift=sh_j), W.get(gskip.skip_k),
                                            gskip.skip_k.bias, 1, pad_mode=PAD_ZERO, padL=kw // 2,
                                            pack=gskip._pack)
                        scale_j = ones_j
                    else:
                        if sc_j is not None:     # the linear output is the NORMALISED one
                            a_norm = ops.affine_prelu(aj, sc_j, sh_j, None)
                        sk = aj if a_norm is None else a_norm
                        scale_j = gskip.skip_k
                    mask = None
                    if drop_p > 0:               # nn.Dropout on sk_h (generator.py:70-71)
                        mask = _dropout_mask(tuple(sk.shape), drop_p, sk.device)
                        fold = None if gskip.skip_type == 'conv' else \
                            scale_j.detach().reshape(-1).contiguous()
                        sk = ops.scale_mask(sk, fold, mask)
                        scale_j = ones_j         # alpha is folded into the masked tensor
                    skip_saved[enc_idx] = (mask, a_norm)
                    if gskip.merge_mode == 'concat':
                        src = Src(prev, sk,
                                  scale=_cat(sc_p if sc_p is not None else _ones(np_, prev), scale_j),
                                  shift=_vec2(sh_p, np_, None, nj, 0.0, prev),
                                  slope=_cat(s_prev, ones_j))
                    else:   # 'sum' (generator.py:71-73): prelu(prev) + skip, materialised
                        hp, sl0 = prev, s_prev
                        if sc_p is not None:
                            hp, sl0 = ops.affine_prelu(prev, sc_p, sh_p, s_prev), None
                        src = Src(ops.sum_skip(hp, sl0, sk, scale_j.detach().reshape(-1)))
                        src.sum_of = True
                else:
                    src = Src(prev, scale=sc_p, shift=sh_p, slope=s_prev)
            has_bn = getattr(blk, 'norm', None) is not None
            if is_conv:
                a = ops.conv1d_fwd(src, W.get(blk.conv), blk.conv.bias, blk.stride,
                                   pack=blk._pack)
            else:
                # the Tanh is fused into the deconv's epilogue unless a BatchNorm sits between
                act = ACT_TANH if (blk.is_tanh and not has_bn) else ACT_NONE
                a = ops.deconv1d_fwd(src, W.get(blk.deconv), blk.deconv.bias, blk.stride, act,
                                     pack=blk._pack)
            sc, sh, bnsv = _block_norm(blk, a, training)
            src_dec.append(src)
            a_dec.append(a)
            xf_dec.append((sc, sh))
            bn_dec.append(bnsv)
            enc_idx -= 1
        lastb = dec[-1]
        last_conv = hasattr(lastb, 'conv')
        sc, sh = xf_dec[-1]
        if last_conv:       # a conv block as last layer ends in its PReLU, not in a Tanh
            y = ops.affine_prelu(a_dec[-1], sc, sh, lastb.act.weight)
        elif bn_dec[-1] is not None:
            y = ops.affine_tanh(a_dec[-1], sc, sh) if lastb.is_tanh else \
                ops.affine_prelu(a_dec[-1], sc, sh, lastb.act.weight)
        else:
            y = a_dec[-1]
        keep_last = y is not a_dec[-1]
        ctx.gen = gen
        ctx.set_materialize_grads(False)
        ctx.x_needs = x.requires_grad
        # the output goes through save_for_backward (no ctx -> y -> grad_fn -> ctx cycle that a
        # never-backpropagated grad-mode forward would leak); the rest are plain intermediates
        ctx.save_for_backward(y)
        ctx.keep_last = keep_last
        ctx.state = (x, z, a_enc, src_enc, xf_enc, bn_enc, a_dec if keep_last else a_dec[:-1],
                     src_dec, bn_dec, skip_saved, W)
        hid = None
        if want_hid:
            hid = {}
            for i, blk in enumerate(enc):
                hid['enc_{}'.format(i)] = ops.affine_prelu(a_enc[i], xf_enc[i][0], xf_enc[i][1],
                                                           blk.act.weight)
            if not gen.no_z:
                hid['enc_zc'] = torch.cat((z, hid['enc_{}'.format(n_enc - 1)]), dim=1)
            for i, blk in enumerate(dec):
                if i == len(dec) - 1:
                    hid['dec_{}'.format(i)] = y
                else:
                    hid['dec_{}'.format(i)] = ops.affine_prelu(a_dec[i], xf_dec[i][0],
                                                               xf_dec[i][1], blk.act.weight)
        ctx.mark_non_differentiable(*([] if hid is None else list(hid.values())))
        if hid is None:
            return y
        return (y,) + tuple(hid.values())

    @staticmethod
    def backward(ctx, dy, *unused):
        gen = ctx.gen
        x, z, a_enc, src_enc, xf_enc, bn_enc, a_dec, src_dec, bn_dec, skip_saved, W = ctx.state
        y = ctx.saved_tensors[0]
        a_dec = list(a_dec) if ctx.keep_last else list(a_dec) + [y]
        enc, dec = list(gen.enc_blocks), list(gen.dec_blocks)
        n_enc, n_dec = len(enc), len(dec)
        if dy is None:
            ctx.state = None
            return (None,) * len(ctx.needs_input_grad)
        if 'eval' in bn_enc or 'eval' in bn_dec:
            raise RuntimeError('Generator backward in eval() mode with BatchNorm is not supported; '
                               'call .train()')
        dy = dy.contiguous()
        dh = dy
        dskip = {}          # enc index -> gradient w.r.t. the skip tensor entering the deconv
        dh_last_enc = None  # gradient w.r.t. h of the last encoder layer
        # ---- decoder, last to first ----
        da = None
        for li in range(n_dec - 1, -1, -1):
            blk = dec[li]
            is_conv = hasattr(blk, 'conv')
            mod = blk.conv if is_conv else blk.deconv
            w = W.get(mod)
            K, S = blk.kwidth, blk.stride
            bnsv = bn_dec[li]
            bnm = getattr(blk, 'norm', None)
            if not is_conv and blk.is_tanh:
                if bnsv is None:
                    da = ops.tanh_bwd(a_dec[li], dy, dbias=_gb(mod.bias))
                else:       # y = tanh(bn(c)): through the Tanh, then through the BatchNorm
                    da = _act_bwd_bn(a_dec[li], ops.tanh_bwd(y, dy), None, bnsv, None,
                                     _gb(bnm.weight), _gb(bnm.bias), _gb(mod.bias))
            elif bnsv is None:
                da = ops.act_bwd(a_dec[li], dh, slope=blk.act.weight,
                                 dslope=_gb(blk.act.weight), dbias=_gb(mod.bias))
            else:
                da = _act_bwd_bn(a_dec[li], dh, blk.act.weight, bnsv, _gb(blk.act.weight),
                                 _gb(bnm.weight), _gb(bnm.bias), _gb(mod.bias))
            src = src_dec[li]
            with _SideStream(da, src):
                if W.needs_grad(mod):
                    gw = W.grad_target(mod)
                    if is_conv:
                        ops.wgrad(Src(da), src, gw, K, S, ops.conv_pad(K, S)[0], PAD_REFLECT)
                    else:
                        ops.wgrad(src, Src(da), gw, K, S, ops.deconv_pad(K, S), PAD_ZERO)
                    W.finish(mod, gw)
                _ready(blk)
            if is_conv:
                # a conv decoder level never takes a skip (generator.py:212-213); its input is
                # one tensor, or (z, h_last) for the first layer
                d_in = ops.conv1d_dgrad(da, w, src.L, S, pack=blk._pack)
                if li == 0:
                    dh_last_enc = d_in if gen.no_z else d_in[:, src.C0:].contiguous()
                else:
                    dh = d_in
            elif li == 0:
                if gen.no_z:
                    _d0, dh_last_enc = ops.deconv1d_dgrad(da, w, S, 0, pack=blk._pack)
                else:
                    # the z half of the input needs no gradient: its tiles are skipped
                    _d0, dh_last_enc = ops.deconv1d_dgrad(da, w, S, src.C0, need0=False,
                                                          pack=blk._pack)
            else:
                if src.C1 > 0:
                    dh, dsk = ops.deconv1d_dgrad(da, w, S, src.C0, pack=blk._pack)
                    dskip[n_enc - 1 - li] = dsk
                else:
                    _d0, dh = ops.deconv1d_dgrad(da, w, S, 0, pack=blk._pack)
                    if getattr(src, 'sum_of', False):
                        dskip[n_enc - 1 - li] = dh    # d(prelu(prev) + skip): same gradient
        # ---- encoder, last to first ----
        dh = dh_last_enc
        dx = None
        for l in range(n_enc - 1, -1, -1):
            blk = enc[l]
            w = W.get(blk.conv)
            K, S = blk.kwidth, blk.stride
            bnsv = bn_enc[l]
            bnm = getattr(blk, 'norm', None)
            sc, sh = xf_enc[l]
            gskip = gen.skips[l]['alpha'] if (gen.skip and l in gen.skips) else None
            dsk = dskip.get(l)
            mask, a_norm = skip_saved.get(l, (None, None))
            if dsk is not None and mask is not None:
                dsk = ops.scale_mask(dsk, None, mask)              # back through the dropout
            alpha_p = None
            if gskip is not None and gskip.skip_type == 'conv':
                if dsk is not None:     # -> gradient w.r.t. the (normalised) linear output
                    dsk = _skip_conv_bwd(gskip, dsk, Src(a_enc[l], scale=sc, shift=sh), W)
                alpha_v = _ones(a_enc[l].shape[1], a_enc[l]) if dsk is not None else None
                ready_extra = gskip.skip_k
            else:
                alpha_p = gskip.skip_k if gskip is not None else None
                alpha_v = alpha_p if dsk is not None else None
                ready_extra = alpha_p
            if bnsv is None:
                da = ops.act_bwd(a_enc[l], dh, dskip=dsk, slope=blk.act.weight, alpha=alpha_v,
                                 dslope=_gb(blk.act.weight),
                                 dalpha=_gb(alpha_p, dsk is not None),
                                 dbias=_gb(blk.conv.bias))
            elif dsk is None:
                da = _act_bwd_bn(a_enc[l], dh, blk.act.weight, bnsv, _gb(blk.act.weight),
                                 _gb(bnm.weight), _gb(bnm.bias), _gb(blk.conv.bias))
            else:
                # the skip taps the NORMALISED linear output: PReLU + skip gradients meet on
                # bn(c) first, then go through the BatchNorm together
                if a_norm is None:
                    a_norm = ops.affine_prelu(a_enc[l], sc, sh, None)
                g = ops.act_bwd(a_norm, dh, dskip=dsk, slope=blk.act.weight, alpha=alpha_v,
                                dslope=_gb(blk.act.weight), dalpha=_gb(alpha_p, True))
                da = _act_bwd_bn(a_enc[l], g, None, bnsv, None, _gb(bnm.weight), _gb(bnm.bias),
                                 _gb(blk.conv.bias))
            padL = ops.conv_pad(K, S)[0]
            with _SideStream(da, src_enc[l]):
                if W.needs_grad(blk.conv):
                    gw = W.grad_target(blk.conv)
                    ops.wgrad(Src(da), src_enc[l], gw, K, S, padL, PAD_REFLECT)
                    W.finish(blk.conv, gw)
                _ready(blk, ready_extra)
            if l > 0:
                dh = ops.conv1d_dgrad(da, w, src_enc[l].L, S, pack=blk._pack)
            elif ctx.x_needs:
                dx = ops.conv1d_dgrad(da, w, src_enc[l].L, S, pack=blk._pack)
        _join_side(dy)
        ctx.state = None
        return (None, None, dx, None) + (None,) * (len(ctx.needs_input_grad) - 4)


# =====================================================================================
# Discriminator
# =====================================================================================
class DiscriminatorFn(torch.autograd.Function):
    """logit = D(x) (discriminator.py:150-194) with the phase shifts given as rolls."""

    @staticmethod
    def forward(ctx, disc, rolls, x, x1, *params):
        # x1 (optional): second channel group; D(cat(x, x1)) without materialising the cat
        # (model.py:174): the first conv reads two pointers
        x = x.contiguous()
        blocks = list(disc.enc_blocks)
        training = disc.training
        src = Src(x) if x1 is None else Src(x, x1.contiguous())
        cs, srcs, bns, xfs = [], [], [], []
        W = _Weights()
        for l, blk in enumerate(blocks):
            c = ops.conv1d_fwd(src, W.get(blk.conv), blk.conv.bias, blk.stride, roll=rolls[l],
                               pack=blk._pack)
            srcs.append(src)
            cs.append(c)
            if blk.norm is not None:
                bn = blk.norm
                if _bn_batch_stats(bn, training):
                    mean, rstd, scale, shift = _bn_train_stats(c, bn, training)
                    bns.append((mean, rstd, bn.weight, bn.bias))
                else:
                    scale, shift, _rstd = _bn_eval_affine(bn)
                    bns.append('eval')
                xfs.append((scale, shift))
                src = Src(c, scale=scale, shift=shift, slope=blk.act.weight)
            else:
                bns.append(None)
                xfs.append((None, None))
                src = Src(c, slope=blk.act.weight)
        # head (discriminator.py:175-191) on the last activation
        h = ops.affine_prelu(cs[-1], xfs[-1][0], xfs[-1][1], blocks[-1].act.weight)
        B = h.shape[0]
        fc = disc.fc
        pt = disc.pool_type
        extra = {}
        if pt == 'none':        # dense head on h.view(B, -1) (discriminator.py:180-182)
            hf = h.view(B, -1)
            y1 = ops.linear_fwd(hf, W.get(fc[0]))
            a1 = ops.bias_prelu_rows(y1, fc[0].bias, W.get(fc[1]))
            y2 = ops.linear_fwd(a1, W.get(fc[2]))
            a2 = ops.bias_prelu_rows(y2, fc[2].bias, W.get(fc[3]))
            y3 = ops.linear_fwd(a2, W.get(fc[4]))
            out = ops.bias_prelu_rows(y3, fc[4].bias, None)
            head = (hf, y1, a1, y2, a2, y3)
        else:
            if pt == 'conv':    # 1x1 conv to one channel, Linear over time (:175-179)
                pc = ops.conv1d_fwd(Src(h), W.get(disc.pool_conv), disc.pool_conv.bias, 1,
                                    pad_mode=PAD_ZERO, padL=0, pack=disc._pool_pack)
                hp, idx = pc.view(B, -1), None
                extra['avg_conv_h'] = hp
            else:               # global max / mean over time, Linear over channels (:183-190)
                hp, idx = ops.pool_time_fwd(h, 'max' if pt == 'gmax' else 'avg')
            y3 = ops.linear_fwd(hp, W.get(fc))
            out = ops.bias_prelu_rows(y3, fc.bias, None)
            head = (h, hp, idx, y3)
        ctx.disc = disc
        ctx.rolls = tuple(rolls)
        ctx.x_needs = x.requires_grad or (x1 is not None and x1.requires_grad)
        ctx.split = None if x1 is None else (x.shape[1], x.requires_grad, x1.requires_grad)
        ctx.state = (cs, srcs, bns, head, W)
        disc._last_fwd = (cs, xfs)      # for the lazy int_act dict
        disc._last_extra = extra
        disc._last_head = head          # linear outputs of the head (gate-aligned parity tests)
        return out

    @staticmethod
    def backward(ctx, dout):
        disc = ctx.disc
        cs, srcs, bns, head, W = ctx.state
        blocks = list(disc.enc_blocks)
        fc = disc.fc
        dout = dout.contiguous()
        # ---- head ----
        def lin_wgrad(mod, dy, xin):
            if W.needs_grad(mod):
                gw = W.grad_target(mod)
                ops.linear_wgrad(dy, xin, gw)
                W.finish(mod, gw)

        def prelu_bwd(y, lin, act, dact):
            gs = W.grad_target(act) if W.needs_grad(act) else None
            d = ops.bias_prelu_rows_bwd(y, lin.bias, W.get(act), dact, gs, _gb(lin.bias))
            if gs is not None:
                W.finish(act, gs)
            return d

        if disc.pool_type == 'none':
            hf, y1, a1, y2, a2, y3 = head
            dy3 = ops.bias_prelu_rows_bwd(y3, fc[4].bias, None, dout, None, _gb(fc[4].bias))
            lin_wgrad(fc[4], dy3, a2)
            da2 = ops.linear_dgrad(dy3, W.get(fc[4]))
            dy2 = prelu_bwd(y2, fc[2], fc[3], da2)
            lin_wgrad(fc[2], dy2, a1)
            da1 = ops.linear_dgrad(dy2, W.get(fc[2]))
            dy1 = prelu_bwd(y1, fc[0], fc[1], da1)
            lin_wgrad(fc[0], dy1, hf)
            dh = ops.linear_dgrad(dy1, W.get(fc[0])).view(cs[-1].shape)
            _ready(fc)
        else:
            h, hp, idx, y3 = head
            Lh = h.shape[2]
            dy3 = ops.bias_prelu_rows_bwd(y3, fc.bias, None, dout, None, _gb(fc.bias))
            lin_wgrad(fc, dy3, hp)
            dhp = ops.linear_dgrad(dy3, W.get(fc))
            if disc.pool_type == 'conv':
                pcm = disc.pool_conv
                dpc = dhp.view(h.shape[0], 1, Lh)
                if pcm.bias is not None and pcm.bias.requires_grad:
                    ops.act_bwd(dpc, dpc, dbias=grad_buf(pcm.bias))      # sum over (b, t)
                if W.needs_grad(pcm):
                    gw = W.grad_target(pcm)
                    ops.wgrad(Src(dpc), Src(h), gw, 1, 1, 0, PAD_ZERO)
                    W.finish(pcm, gw)
                dh = ops.conv1d_dgrad(dpc, W.get(pcm), Lh, 1, padL=0, pack=disc._pool_pack)
                _ready(fc, pcm)
            else:
                dh = ops.pool_time_bwd(dhp, idx, Lh, 'max' if disc.pool_type == 'gmax' else 'avg')
                _ready(fc)
        # ---- conv stack ----
        dx = None
        for l in range(len(blocks) - 1, -1, -1):
            blk = blocks[l]
            w = W.get(blk.conv)
            K, S = blk.kwidth, blk.stride
            bn = bns[l]
            if bn == 'eval':
                raise RuntimeError('Discriminator backward in eval() mode with BatchNorm is not '
                                   'supported; call .train() (the reference never does this)')
            if bn is not None:
                dc = _act_bwd_bn(cs[l], dh, blk.act.weight, bn, _gb(blk.act.weight), _gb(bn[2]),
                                 _gb(bn[3]), _gb(blk.conv.bias))
            else:
                dc = ops.act_bwd(cs[l], dh, slope=blk.act.weight, dslope=_gb(blk.act.weight),
                                 dbias=_gb(blk.conv.bias))
            padL = ops.conv_pad(K, S)[0]
            with _SideStream(dc, srcs[l]):
                if W.needs_grad(blk.conv):
                    gw = W.grad_target(blk.conv)
                    ops.wgrad(Src(dc), srcs[l], gw, K, S, padL, PAD_REFLECT, roll=ctx.rolls[l])
                    W.finish(blk.conv, gw)
                _ready(blk)
            if l > 0:
                dh = ops.conv1d_dgrad(dc, w, srcs[l].L, S, roll=ctx.rolls[l], pack=blk._pack)
            elif ctx.x_needs:
                dx = ops.conv1d_dgrad(dc, w, srcs[l].L, S, roll=ctx.rolls[l], pack=blk._pack)
        _join_side(dout)
        ctx.state = None
        dx1 = None
        if dx is not None and ctx.split is not None:
            c0, need0, need1 = ctx.split
            dx1 = dx[:, c0:].contiguous() if need1 else None
            dx = dx[:, :c0].contiguous() if need0 else None
        return (None, None, dx, dx1) + (None,) * (len(ctx.needs_input_grad) - 4)


# =====================================================================================
# stand-alone blocks (modules.py:73-141)
# =====================================================================================
class ConvBlockFn(torch.autograd.Function):
    """(h, a) = GConv1DBlock(x): reflect pad + strided conv (+BatchNorm) + PReLU."""

    @staticmethod
    def forward(ctx, blk, x, *params):
        x = x.contiguous()
        src = Src(x)
        W = _Weights()
        c = ops.conv1d_fwd(src, W.get(blk.conv), blk.conv.bias, blk.stride, pack=blk._pack)
        bn_saved = None
        scale = shift = None
        if blk.norm is not None:
            bn = blk.norm
            if _bn_batch_stats(bn, blk.training):
                mean, rstd, scale, shift = _bn_train_stats(c, bn, blk.training)
                bn_saved = (mean, rstd, bn.weight, bn.bias)
            else:
                scale, shift, rstd = _bn_eval_affine(bn)
                bn_saved = ('eval', scale, shift, rstd)
        h = ops.affine_prelu(c, scale, shift, blk.act.weight)
        a = c if blk.norm is None else ops.affine_prelu(c, scale, shift, None)
        ctx.blk = blk
        ctx.set_materialize_grads(False)
        ctx.x_needs = x.requires_grad
        ctx.state = (src, c, bn_saved, W)
        return h, a

    @staticmethod
    def backward(ctx, dh, da_lin):
        blk = ctx.blk
        src, c, bn, W = ctx.state
        K, S = blk.kwidth, blk.stride
        w = W.get(blk.conv)
        if dh is None and da_lin is None:
            return (None,) * len(ctx.needs_input_grad)
        dh = dh.contiguous() if dh is not None else None
        if bn is None:
            lin = da_lin.contiguous() if da_lin is not None else None
            dc = ops.act_bwd(c, dh, dskip=lin, slope=blk.act.weight,
                             alpha=_ones(c.shape[1], c) if lin is not None else None,
                             dslope=_gb(blk.act.weight), dbias=_gb(blk.conv.bias))
        else:
            if da_lin is not None or dh is None:
                raise RuntimeError('gradient through the linear output of a normalised '
                                   'GConv1DBlock is unsupported')
            if isinstance(bn[0], str):      # ('eval', ...): frozen statistics, a fixed affine map
                dc = _eval_bn_act_bwd(c, blk.norm, bn[1], bn[2], bn[3], dh, blk.act.weight,
                                      _gb(blk.act.weight), dbias=_gb(blk.conv.bias))
            else:
                dc = _act_bwd_bn(c, dh, blk.act.weight, bn, _gb(blk.act.weight), _gb(bn[2]), _gb(bn[3]),
                                 _gb(blk.conv.bias))
        padL = ops.conv_pad(K, S)[0]
        if W.needs_grad(blk.conv):
            gw = W.grad_target(blk.conv)
            ops.wgrad(Src(dc), src, gw, K, S, padL, PAD_REFLECT)
            W.finish(blk.conv, gw)
        dx = ops.conv1d_dgrad(dc, w, src.L, S, pack=blk._pack) if ctx.x_needs else None
        ctx.state = None
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


class DeconvBlockFn(torch.autograd.Function):
    """h = GDeconv1DBlock(x): transposed conv, trim, (BatchNorm,) PReLU or Tanh
    (modules.py:135-141)."""

    @staticmethod
    def forward(ctx, blk, x, *params):
        x = x.contiguous()
        src = Src(x)
        W = _Weights()
        bn = blk.norm if isinstance(blk.norm, torch.nn.BatchNorm1d) else None
        # with a BatchNorm between the deconv and the activation the tanh cannot ride in the
        # contraction's epilogue: c, then bn(c) through tanh / PReLU in one pointwise pass
        act = ACT_TANH if (blk.is_tanh and bn is None) else ACT_NONE
        c = ops.deconv1d_fwd(src, W.get(blk.deconv), blk.deconv.bias, blk.stride, act,
                             pack=blk._pack)
        bn_saved = None
        if bn is None:
            h = c if blk.is_tanh else ops.affine_prelu(c, slope=blk.act.weight)
        else:
            if _bn_batch_stats(bn, blk.training):
                mean, rstd, scale, shift = _bn_train_stats(c, bn, blk.training)
                bn_saved = (mean, rstd, bn.weight, bn.bias)
            else:
                scale, shift, rstd = _bn_eval_affine(bn)
                bn_saved = ('eval', scale, shift, rstd)
            h = ops.affine_tanh(c, scale, shift) if blk.is_tanh else \
                ops.affine_prelu(c, scale, shift, blk.act.weight)
        ctx.blk = blk
        ctx.x_needs = x.requires_grad
        ctx.state = (src, c, h if (blk.is_tanh and bn is not None) else None, bn_saved, W)
        return h

    @staticmethod
    def backward(ctx, dh):
        blk = ctx.blk
        src, c, y, bn, W = ctx.state
        K, S = blk.kwidth, blk.stride
        w = W.get(blk.deconv)
        dh = dh.contiguous()
        bnm = blk.norm if bn is not None else None
        if bn is not None and isinstance(bn[0], str):
            # frozen statistics (fine-tuning in eval()): the BatchNorm is a fixed affine map
            da = _eval_bn_act_bwd(c, bnm, bn[1], bn[2], bn[3], dh,
                                  None if blk.is_tanh else blk.act.weight,
                                  None if blk.is_tanh else _gb(blk.act.weight),
                                  y_tanh=y if blk.is_tanh else None, dbias=_gb(blk.deconv.bias))
        elif blk.is_tanh:
            if bn is None:
                da = ops.tanh_bwd(c, dh, dbias=_gb(blk.deconv.bias))
            else:       # y = tanh(bn(c)): through the Tanh, then through the BatchNorm
                da = _act_bwd_bn(c, ops.tanh_bwd(y, dh), None, bn, None, _gb(bnm.weight),
                                 _gb(bnm.bias), _gb(blk.deconv.bias))
        elif bn is None:
            da = ops.act_bwd(c, dh, slope=blk.act.weight, dslope=_gb(blk.act.weight),
                             dbias=_gb(blk.deconv.bias))
        else:
            da = _act_bwd_bn(c, dh, blk.act.weight, bn, _gb(blk.act.weight), _gb(bnm.weight),
                             _gb(bnm.bias), _gb(blk.deconv.bias))
        if W.needs_grad(blk.deconv):
            gw = W.grad_target(blk.deconv)
            ops.wgrad(src, Src(da), gw, K, S, ops.deconv_pad(K, S), PAD_ZERO)
            W.finish(blk.deconv, gw)
        dx = None
        if ctx.x_needs:
            _d0, dx = ops.deconv1d_dgrad(da, w, S, 0, pack=blk._pack)
        ctx.state = None
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


# =====================================================================================
# losses
# =====================================================================================
class MSEConstFn(torch.autograd.Function):
    """mean((x - c)^2) against a constant label (nn.MSELoss on a filled label tensor,
    train.py:94 + model.py:298,305,316)."""

    @staticmethod
    def forward(ctx, x, target):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.target = float(target)
        return ops.mse_const(x, target)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.mse_const_bwd(x, ctx.target, gout=g.contiguous()), None


class BCELogitsConstFn(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(x, label) for a constant label (WSEGAN
    --vanilla_gan, model.py:582-583)."""

    @staticmethod
    def forward(ctx, x, target):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.target = float(target)
        return ops.bce_logits_const(x, target)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.bce_logits_const_bwd(x, ctx.target, gout=g.contiguous()), None


class L1MeanFn(torch.autograd.Function):
    """F.l1_loss(x, y) (model.py:79,318)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        ctx.save_for_backward(x, y)
        return ops.l1_mean(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = ops.l1_bwd(x, y, gout=g.contiguous())
        if ctx.needs_input_grad[1]:
            gy = ops.l1_bwd(y, x, gout=g.contiguous())
        return gx, gy


class MSEMeanFn(torch.autograd.Function):
    """F.mse_loss(x, y) between two tensors (--reg_loss mse_loss; train.py:179, model.py:79)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        ctx.save_for_backward(x, y)
        return ops.mse_mean(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = ops.mse_bwd(x, y, gout=g.contiguous())
        if ctx.needs_input_grad[1]:
            gy = ops.mse_bwd(y, x, gout=g.contiguous())
        return gx, gy


class STFTPowL1Fn(torch.autograd.Function):
    """F.l1_loss(pow_db(STFT(x)), pow_db(STFT(y))) of the WSEGAN step (model.py:640-653):
    rectangular window `win` centred in n_fft, hop `hop`, normalized, reflect centre padding;
    pow_db = 10*log10(|X|^2 + 10e-20).  x, y: [B, 1, T] (or [B, T]); the gradient flows to x
    only (y is the clean reference)."""

    @staticmethod
    def forward(ctx, x, y, n_fft, hop, win):
        shape = x.shape
        x2 = x.reshape(shape[0], shape[-1]).contiguous()
        y2 = y.reshape(shape[0], shape[-1]).contiguous()
        basis = ops.stft_basis(n_fft, win, x2.device)
        Sx = ops.stft_spectrum(ops.stft_frames(x2, n_fft, hop, win), basis)
        nb = n_fft // 2 + 1
        dbx = ops.powdb(Sx, nb)
        dby = ops.powdb(ops.stft_spectrum(ops.stft_frames(y2, n_fft, hop, win), basis), nb)
        ctx.save_for_backward(Sx, dbx, dby, basis)
        ctx.geom = (shape, n_fft, hop, win)
        return ops.l1_mean(dbx, dby)

    @staticmethod
    def backward(ctx, g):
        Sx, dbx, dby, basis = ctx.saved_tensors
        shape, n_fft, hop, win = ctx.geom
        ddb = ops.l1_bwd(dbx, dby, gout=g.contiguous())
        dfr = ops.stft_spectrum_bwd(ops.powdb_bwd(Sx, ddb, n_fft // 2 + 1), basis)
        dx = ops.stft_overlap_add(dfr, shape[0], shape[-1], n_fft, hop, win)
        return dx.view(shape), None, None, None, None
