"""ctypes binding of libsegan_hip.so (the C ABI declared in include/segan_hip.h).

The product path has no fallback: if the library is missing or a call fails, a
RuntimeError carrying ``segan_last_error()`` is raised.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SEGAN_HIP_LIB') or os.path.join(_HERE, 'libsegan_hip.so')   # override: A/B of two builds

PAD_REFLECT = 0
PAD_ZERO = 1
ACT_NONE = 0
ACT_TANH = 1
ABI_VERSION = 13


class SeganSrc(Structure):
    """Mirror of ``segan_src`` (include/segan_hip.h)."""
    _fields_ = [('p0', c_void_p), ('p1', c_void_p), ('C0', c_int32), ('C1', c_int32),
                ('scale', c_void_p), ('shift', c_void_p), ('slope', c_void_p)]


_P = c_void_p
_SRC = POINTER(SeganSrc)

# name -> (restype, argtypes); mirrors include/segan_hip.h one to one
SIGNATURES = {
    'segan_abi_version': (c_int, []),
    'segan_set_reserved_slots': (c_int, [c_int]),
    'segan_last_error': (c_char_p, []),
    'segan_packed_f_bytes': (c_size_t, [c_int, c_int, c_int]),
    'segan_packed_t_bytes': (c_size_t, [c_int, c_int, c_int]),
    'segan_pack_weights': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'segan_packed_bf_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'segan_pack_weights_bf': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'segan_corr_scratch_bytes': (c_size_t, []),
    'segan_bf16_scratch_bytes': (c_size_t, [c_int] * 9),
    'segan_debug_last_corr': (None, [POINTER(c_int)]),
    'segan_debug_last_wgrad': (None, [POINTER(c_int)]),
    'segan_conv1d_fwd': (c_int, [_SRC, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_int, _P, c_size_t, _P]),
    'segan_conv1d_dgrad': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, _P, c_size_t, _P]),
    'segan_wgrad': (c_int, [_SRC, _SRC, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_int, _P, c_size_t, _P]),
    'segan_wgrad_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'segan_deconv1d_fwd': (c_int, [_SRC, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, _P, c_size_t, _P]),
    'segan_deconv1d_dgrad': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, _P, c_size_t, _P]),
    'segan_bn_nsplit': (c_int, [c_int, c_int, c_int]),
    'segan_bn_stats': (c_int, [_P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, c_int, c_int,
                               c_int, _P]),
    'segan_bn_partial': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'segan_bn_final': (c_int, [_P, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, c_int, _P]),
    'segan_act_bwd_bn_reduce': (c_int, [_P] * 12 + [c_int, c_int, c_int, _P]),
    'segan_act_bwd_bn_apply': (c_int, [_P] * 11 + [c_int, c_int, c_int, c_double, _P]),
    'segan_affine_prelu': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'segan_affine_tanh': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'segan_scale_mask': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'segan_sum_skip': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'segan_bce_logits_const': (c_int, [_P, c_float, _P, _P, _P, c_float, c_int, _P]),
    'segan_act_bwd': (c_int, [_P] * 16 + [c_int, c_int, c_int, _P]),
    'segan_tanh_bwd': (c_int, [_P, _P, _P, c_float, _P, _P, _P, c_int, c_int, c_int, _P]),
    'segan_gemm_scratch_bytes': (c_size_t, [c_int, c_int, c_int]),
    'segan_gemm': (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int, c_int,
                           c_int, c_int, c_int, _P, c_size_t, _P]),
    'segan_bias_prelu_rows': (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    'segan_bias_prelu_rows_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    'segan_mse_const': (c_int, [_P, c_float, _P, _P, _P, c_float, c_int, _P]),
    'segan_l1_bwd': (c_int, [_P, _P, _P, c_float, _P, c_int64, _P]),
    'segan_l1_mean': (c_int, [_P, _P, _P, _P, c_int64, _P]),
    'segan_packed_g_bytes': (c_size_t, [c_int, c_int, c_int]),
    'segan_pack_weights_g': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    'segan_conv1d_dgrad_short': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int, _P]),
    'segan_pool_time_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    'segan_pool_time_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    'segan_mse_mean': (c_int, [_P, _P, _P, _P, c_int64, _P]),
    'segan_mse_bwd': (c_int, [_P, _P, _P, c_float, _P, c_int64, _P]),
    'segan_snorm_ws_floats': (c_size_t, [c_int, c_int, c_int, c_int]),
    'segan_snorm_fwd': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    'segan_snorm_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'segan_pcm16_prep': (c_int, [_P, _P, _P, _P, c_int, c_int, c_double, _P]),
    'segan_stft_basis': (c_int, [_P, c_int, c_int, _P]),
    'segan_stft_frames': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'segan_stft_pitch': (c_int, [c_int]),
    'segan_powdb': (c_int, [_P, _P, c_int64, c_int, c_int, c_float, _P]),
    'segan_powdb_bwd': (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_float, _P]),
    'segan_stft_overlap_add': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'segan_deemphasis': (c_int, [_P, _P, c_int, c_int, c_double, _P]),
    'segan_ssnr_frames': (c_int, [c_int, c_int]),
    'segan_ssnr': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_double, _P]),
    'segan_rmsprop_step': (c_int, [_P, _P, _P, c_float, c_float, c_float, c_int64, _P]),
    'segan_adam_step': (c_int, [_P, _P, _P, _P, c_float, c_float, c_float, c_float, c_int, c_int64,
                                _P]),
    'segan_fill': (c_int, [_P, c_float, c_int64, _P]),
    'segan_scale': (c_int, [_P, c_float, c_int64, _P]),
    'segan_comm_id_bytes': (c_int, []),
    'segan_comm_unique_id': (c_int, [_P]),
    'segan_comm_init': (c_int, [POINTER(c_void_p), c_int, c_int, _P]),
    'segan_comm_destroy': (c_int, [_P]),
    'segan_comm_rank': (c_int, [_P]),
    'segan_comm_world': (c_int, [_P]),
    'segan_allreduce': (c_int, [_P, _P, c_size_t, c_float, _P]),
    'segan_broadcast': (c_int, [_P, _P, c_size_t, c_int, _P]),
    'segan_allgather': (c_int, [_P, _P, _P, c_size_t, _P]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'segan_pytorch_amd: {} not found. Build it with `make` (or '
            '`python -c "import __graft_entry__ as g; g.build()"`) at the repo root. '
            'There is no non-HIP fallback.'.format(LIB_PATH))
    # torch bundles its own HIP runtime; it has to be up before this library (linked against
    # /opt/rocm's) is mapped, otherwise the library's first launch finds no device
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:     # pragma: no cover
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    ver = lib.segan_abi_version()
    if ver != ABI_VERSION:
        raise RuntimeError('libsegan_hip ABI {} != expected {}'.format(ver, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().segan_last_error()
        raise RuntimeError('libsegan_hip {} failed ({}): {}'.format(
            what, rc, msg.decode() if msg else '?'))
