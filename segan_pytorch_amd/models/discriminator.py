"""SEGAN discriminator on the HIP path.

API mirror of ``Discriminator`` (segan/models/discriminator.py:65-194): same
constructor signature, sub-module names (``enc_blocks.N.{conv,norm,act}``,
``fc.{0..4}``) and ``forward(x) -> (logit, int_act)`` contract, including the
per-layer random phase shift drawn from python's ``random`` in the reference's order
(``randint`` then ``random`` per layer, discriminator.py:159-172) so a seeded run
takes the same shifts.
"""
import random

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from .core import Model
from .modules import GConv1DBlock


class _LazyIntAct(dict):
    """int_act of the reference holds every layer's activation; they are only
    materialised (one affine+PReLU kernel each) when somebody actually reads them."""

    def __init__(self, disc, logit):
        super().__init__()
        self._cs, self._xfs = disc._last_fwd
        self._slopes = [blk.act.weight.detach() for blk in disc.enc_blocks]
        dict.__setitem__(self, 'logit', logit)
        for k, v in getattr(disc, '_last_extra', {}).items():      # 'avg_conv_h' of the conv head
            dict.__setitem__(self, k, v.detach())
        self._pending = set('h_{}'.format(i) for i in range(len(self._cs)))

    def __missing__(self, key):
        if key in self._pending:
            i = int(key.split('_')[1])
            with torch.no_grad():
                v = ops.affine_prelu(self._cs[i], self._xfs[i][0], self._xfs[i][1], self._slopes[i])
            dict.__setitem__(self, key, v)
            self._pending.discard(key)
            return v
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._pending

    def keys(self):
        return ['h_{}'.format(i) for i in range(len(self._cs))] + ['logit']


class Discriminator(Model):

    def __init__(self, ninputs, fmaps, kwidth, poolings, pool_type='none', pool_slen=None,
                 norm_type='bnorm', bias=True, phase_shift=None, sinc_conv=False):
        super().__init__(name='Discriminator')
        self.phase_shift = phase_shift
        if phase_shift is not None:
            assert isinstance(phase_shift, int), type(phase_shift)
            assert phase_shift > 1, phase_shift
        if pool_slen is None:
            raise ValueError('Please specify D network pool seq len (pool_slen) in the end of '
                             'the conv stack: [inp_len // (total_pooling_factor)]')
        if sinc_conv:
            raise NotImplementedError('sinc_conv is not implemented in segan_pytorch_amd (and is '
                                      'broken in the reference: discriminator.py:90-95)')
        if norm_type not in ('bnorm', 'snorm', None):
            raise TypeError('Unrecognized norm type: ', norm_type)
        ninp = ninputs
        self.enc_blocks = nn.ModuleList()
        for fmap, pool in zip(fmaps, poolings):
            self.enc_blocks.append(GConv1DBlock(ninp, fmap, kwidth, stride=pool, bias=bias,
                                                norm_type=norm_type))
            ninp = fmap
        self.pool_type = pool_type
        if pool_type == 'none':
            pool_slen *= fmaps[-1]
            self.fc = nn.Sequential(nn.Linear(pool_slen, 256), nn.PReLU(256),
                                    nn.Linear(256, 128), nn.PReLU(128), nn.Linear(128, 1))
            if norm_type == 'snorm':
                # discriminator.py:118-121, literally: the two hidden Linears and fc[3], which
                # is the second PReLU (its [128] slope vector is normalised as a 128x1 matrix)
                torch.nn.utils.spectral_norm(self.fc[0])
                torch.nn.utils.spectral_norm(self.fc[2])
                torch.nn.utils.spectral_norm(self.fc[3])
        elif pool_type == 'conv':
            # discriminator.py:122-127: 1x1 conv to one channel, then a Linear over time
            self.pool_conv = nn.Conv1d(fmaps[-1], 1, 1)
            self.fc = nn.Linear(pool_slen, 1)
            self._pool_pack = ops.WeightPack()
            if norm_type == 'snorm':
                torch.nn.utils.spectral_norm(self.pool_conv)
                torch.nn.utils.spectral_norm(self.fc)
        elif pool_type in ('gmax', 'gavg'):
            # discriminator.py:128-137: global max / mean over time, then a Linear over channels
            if pool_type == 'gmax':
                self.gmax = nn.AdaptiveMaxPool1d(1)
            else:
                self.gavg = nn.AdaptiveAvgPool1d(1)
            self.fc = nn.Linear(fmaps[-1], 1, 1)
            if norm_type == 'snorm':
                torch.nn.utils.spectral_norm(self.fc)
        elif pool_type == 'mlp':
            raise NotImplementedError("Discriminator pool_type 'mlp' (discriminator.py:138-146) "
                                      "gives one logit per time step, which the reference's own "
                                      "training steps cannot consume (model.py:297: a [B*T] "
                                      "logit against a [B] label); not implemented")
        else:
            raise TypeError('Unrecognized pool type: ', pool_type)
        self._total_pool = 1
        for p in poolings:
            self._total_pool *= p

    def _fn_params(self):
        return self.all_params()

    def draw_rolls(self):
        """One signed circular shift per layer, consuming python's `random` exactly as
        discriminator.py:159-163 does (roll > 0: shift right)."""
        rolls = []
        for _ in self.enc_blocks:
            if self.phase_shift is None:
                rolls.append(0)
                continue
            shift = random.randint(1, self.phase_shift)
            right = random.random() > 0.5
            rolls.append(shift if right else -shift)
        return rolls

    def forward(self, x, x1=None):
        """x: [B, C, L]; or the two halves (x, x1) of the channel axis, which is
        D(torch.cat((x, x1), 1)) of the reference (model.py:174) without the copy."""
        nch = x.shape[1] + (x1.shape[1] if x1 is not None else 0) if x.dim() == 3 else -1
        if x.dim() != 3 or nch != self.enc_blocks[0].conv.in_channels or \
                (x1 is not None and x1.shape[::2] != x.shape[::2]):
            raise ValueError('Discriminator expects [B, {}, L], got {}'.format(
                self.enc_blocks[0].conv.in_channels, tuple(x.shape)))
        L = x.shape[2]
        if L % self._total_pool != 0:
            raise ValueError('input length {} is not divisible by the total pooling {}'.format(
                L, self._total_pool))
        Lo, Co = L // self._total_pool, self.enc_blocks[-1].conv.out_channels
        want = {'none': Lo * Co, 'conv': Lo}.get(self.pool_type, Co)
        fc_in = self.fc[0].in_features if self.pool_type == 'none' else self.fc.in_features
        if want != fc_in:
            raise ValueError('input length {} does not match the {!r} head ({} features expected, '
                             '{} after pooling by {})'.format(L, self.pool_type, fc_in, want,
                                                              self._total_pool))
        rolls = self.draw_rolls()
        y = Fn.DiscriminatorFn.apply(self, rolls, x, x1, *self._fn_params())
        return y, _LazyIntAct(self, y)
