"""The two building blocks of the SEGAN networks, backed by the HIP kernels.

API mirror of ``GConv1DBlock`` / ``GDeconv1DBlock`` / ``build_norm_layer`` in the
reference's ``segan/models/modules.py:9-18,73-141`` (same constructor arguments,
same sub-module names hence the same ``state_dict`` keys: ``conv.weight``,
``conv.bias``, ``norm.*``, ``act.weight``, ``deconv.weight``, ``deconv.bias``).
The ``nn.Conv1d`` / ``nn.ConvTranspose1d`` / ``nn.BatchNorm1d`` / ``nn.PReLU``
children are used ONLY as parameter containers (and initialisers); their forward is
never called.
"""
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops

MAX_KWIDTH = 32
STRIDES = (1, 2, 4)


def build_norm_layer(norm_type, param=None, num_feats=None):
    if norm_type == 'bnorm':
        return nn.BatchNorm1d(num_feats)
    elif norm_type == 'snorm':
        # registers weight_orig / weight_u / weight_v on the parameter container exactly as the
        # reference does (same state_dict keys, same RNG draws for u and v); the hook torch
        # installs never runs because the container's forward is never called — the
        # normalisation itself is done by ops.snorm_fwd / snorm_bwd (functional._Weights)
        torch.nn.utils.spectral_norm(param)
        return None
    elif norm_type is None:
        return None
    else:
        raise TypeError('Unrecognized norm type: ', norm_type)


def _check_geometry(kwidth, stride):
    if not isinstance(kwidth, int) or kwidth < 1 or kwidth > MAX_KWIDTH:
        raise ValueError('kwidth must be an int in [1, {}], got {}'.format(MAX_KWIDTH, kwidth))
    if stride not in STRIDES:
        raise ValueError('stride (pooling) must be one of {}, got {}'.format(STRIDES, stride))


class GConv1DBlock(nn.Module):

    def __init__(self, ninp, fmaps, kwidth, stride=1, bias=True, norm_type=None):
        super().__init__()
        _check_geometry(kwidth, stride)
        self.conv = nn.Conv1d(ninp, fmaps, kwidth, stride=stride, bias=bias)
        self.norm = build_norm_layer(norm_type, self.conv, fmaps)
        self.act = nn.PReLU(fmaps, init=0)
        self.kwidth = kwidth
        self.stride = stride
        self._pack = ops.WeightPack()

    def forward_norm(self, x, norm_layer):
        raise RuntimeError('forward_norm is fused into the HIP kernels in segan_pytorch_amd')

    def forward(self, x, ret_linear=False):
        if x.dim() != 3 or x.shape[1] != self.conv.in_channels:
            raise ValueError('expected input [B, {}, L], got {}'.format(
                self.conv.in_channels, tuple(x.shape)))
        if x.shape[2] % self.stride != 0:
            raise ValueError('input length {} is not divisible by the stride {}'.format(
                x.shape[2], self.stride))
        params = [p for p in self.parameters()]
        h, a = Fn.ConvBlockFn.apply(self, x, *params)
        if ret_linear:
            return h, a
        return h


class GDeconv1DBlock(nn.Module):

    def __init__(self, ninp, fmaps, kwidth, stride=4, bias=True, norm_type=None, act=None):
        super().__init__()
        _check_geometry(kwidth, stride)
        if stride < 2:
            raise ValueError('GDeconv1DBlock needs stride > 1')
        pad = max(0, (stride - kwidth) // -2)
        # NOTE the reference ignores `bias` here: the deconv always has a bias
        # (modules.py:116-119)
        self.deconv = nn.ConvTranspose1d(ninp, fmaps, kwidth, stride=stride, padding=pad)
        self.norm = build_norm_layer(norm_type, self.deconv, fmaps)
        if act is not None:
            if act != 'Tanh':
                raise NotImplementedError("only act=None (PReLU) or 'Tanh' are implemented")
            self.act = getattr(nn, act)()
        else:
            self.act = nn.PReLU(fmaps, init=0)
        self.is_tanh = act is not None
        self.kwidth = kwidth
        self.stride = stride
        self._pack = ops.WeightPack()

    def forward(self, x):
        if x.dim() != 3 or x.shape[1] != self.deconv.in_channels:
            raise ValueError('expected input [B, {}, L], got {}'.format(
                self.deconv.in_channels, tuple(x.shape)))
        params = [p for p in self.parameters()]
        return Fn.DeconvBlockFn.apply(self, x, *params)
