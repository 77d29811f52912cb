"""SEGAN generator on the HIP path.

API mirror of ``GSkip`` (segan/models/generator.py:18-78) and ``Generator``
(generator.py:80-230): same constructor signature and defaults, same attribute and
sub-module names (``enc_blocks``, ``dec_blocks``, ``alpha_<i>``, ``skips``, ``z``,
``no_z``, ``z_dim``, ``dec_poolings``) and therefore the same ``state_dict`` keys, same
``forward(x, z=None, ret_hid=False)`` contract.  The forward is ONE autograd node
(``functional.GeneratorFn``) that chains the HIP kernels.
"""
import threading

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from .core import Model
from .modules import GConv1DBlock, GDeconv1DBlock


class GSkip(nn.Module):
    """Skip connection of the generator (generator.py:18-78): a learnable ('alpha') or fixed
    ('constant') per-channel scale, or a kwidth-wide stride-1 conv ('conv', generator.py:42-49).
    Parameter container: the scale and the concat happen inside the consuming deconv kernel; the
    conv skip runs the conv kernels from ``functional.GeneratorFn``."""

    def __init__(self, skip_type, size, skip_init, skip_dropout=0, merge_mode='sum',
                 kwidth=11, bias=True):
        super().__init__()
        self.merge_mode = merge_mode
        if skip_type in ('alpha', 'constant'):
            if skip_init == 'zero':
                alpha_ = torch.zeros(size)
            elif skip_init == 'randn':
                alpha_ = torch.randn(size)
            elif skip_init == 'one':
                alpha_ = torch.ones(size)
            else:
                raise TypeError('Unrecognized alpha init scheme: ', skip_init)
            self.skip_k = nn.Parameter(alpha_.view(1, -1, 1))
            if skip_type == 'constant':
                self.skip_k.requires_grad = False
        elif skip_type == 'conv':
            if not 1 <= kwidth <= 32 or kwidth % 2 == 0:
                raise ValueError('skip_kwidth must be odd and <= 32 (an even width changes the '
                                 'length: generator.py:43-49), got {}'.format(kwidth))
            pad = kwidth // 2 if kwidth > 1 else 0
            self.skip_k = nn.Conv1d(size, size, kwidth, stride=1, padding=pad, bias=bias)
            self._pack = ops.WeightPack()       # forward
            self._pack_t = ops.WeightPack()     # data gradient (flipped, transposed weight)
        else:
            raise TypeError('Unrecognized GSkip scheme: ', skip_type)
        self.skip_type = skip_type
        if skip_dropout > 0:
            # parameter-free container: the mask is drawn and applied inside functional.GeneratorFn
            self.skip_dropout = nn.Dropout(skip_dropout)
        if merge_mode not in ('sum', 'concat'):
            raise TypeError('Unrecognized skip merge mode: ', merge_mode)

    def __repr__(self):
        if self.skip_type == 'alpha':
            return self._get_name() + '(Alpha(1))'
        elif self.skip_type == 'constant':
            return self._get_name() + '(Constant(1))'
        return super().__repr__()

    def forward(self, hj, hi):
        raise RuntimeError('GSkip is applied inside the fused decoder kernels; call the '
                           'Generator instead')


class Generator(Model):

    def __init__(self, ninputs, fmaps, kwidth, poolings, dec_fmaps=None, dec_kwidth=None,
                 dec_poolings=None, z_dim=None, no_z=False, skip=True, bias=False,
                 skip_init='one', skip_dropout=0, skip_type='alpha', norm_type=None,
                 skip_merge='sum', skip_kwidth=11, name='Generator'):
        super().__init__(name=name)
        self.skip = skip
        self.bias = bias
        self.no_z = no_z
        self.z_dim = z_dim
        self.enc_blocks = nn.ModuleList()
        assert isinstance(fmaps, list), type(fmaps)
        assert isinstance(poolings, list), type(poolings)
        if isinstance(kwidth, int):
            kwidth = [kwidth] * len(fmaps)
        assert isinstance(kwidth, list), type(kwidth)
        if norm_type not in (None, 'snorm', 'bnorm'):
            raise TypeError('Unrecognized norm type: ', norm_type)
        skips = {}
        ninp = ninputs
        for pi, (fmap, pool, kw) in enumerate(zip(fmaps, poolings, kwidth), start=1):
            if skip and pi < len(fmaps):
                # a skip connection for all but the last hidden layer (generator.py:113-123)
                gskip = GSkip(skip_type, fmap, skip_init, skip_dropout, merge_mode=skip_merge,
                              kwidth=skip_kwidth, bias=bias)
                l_i = pi - 1
                skips[l_i] = {'alpha': gskip}
                setattr(self, 'alpha_{}'.format(l_i), skips[l_i]['alpha'])
            self.enc_blocks.append(GConv1DBlock(ninp, fmap, kw, stride=pool, bias=bias,
                                                norm_type=norm_type))
            ninp = fmap
        self.skips = skips
        if not no_z and z_dim is None:
            z_dim = fmaps[-1]
            self.z_dim = z_dim
        if not no_z:
            ninp += z_dim
        if dec_fmaps is None:
            dec_fmaps = fmaps[::-1][1:] + [1]
        else:
            assert isinstance(dec_fmaps, list), type(dec_fmaps)
        if dec_poolings is None:
            dec_poolings = poolings[:]
        else:
            assert isinstance(dec_poolings, list), type(dec_poolings)
        self.dec_poolings = dec_poolings
        if dec_kwidth is None:
            dec_kwidth = kwidth[:]
        elif isinstance(dec_kwidth, int):
            dec_kwidth = [dec_kwidth] * len(dec_fmaps)
        assert isinstance(dec_kwidth, list), type(dec_kwidth)
        self.dec_blocks = nn.ModuleList()
        for pi, (fmap, pool, kw) in enumerate(zip(dec_fmaps, dec_poolings, dec_kwidth), start=1):
            if skip and pi > 1 and pool > 1 and skip_merge == 'concat':
                ninp *= 2
            act = 'Tanh' if pi >= len(dec_fmaps) else None
            if pool > 1:
                blk = GDeconv1DBlock(ninp, fmap, kw, stride=pool, norm_type=norm_type, bias=bias,
                                     act=act)
            else:
                # pooling 1: a plain conv block, never a Tanh (generator.py:171-176)
                blk = GConv1DBlock(ninp, fmap, kw, stride=1, bias=bias, norm_type=norm_type)
            self.dec_blocks.append(blk)
            ninp = fmap
        # look-ahead draw of the next step's z on a host thread (_host_z); the training loops set it
        self.z_prefetch = False
        self._skip_dropout = skip_dropout
        self._total_pool = 1
        for p in poolings:
            self._total_pool *= p

    def _host_z(self, shape, device):
        """z drawn on the host from torch's global CPU generator exactly like the reference
        (generator.py:197-199: same RNG stream), moved to the GPU without stalling the launch
        queue: a pageable-memory ``.to(device)`` makes the host wait for everything already
        enqueued on the stream, once per step.  Two pinned staging buffers and two device
        buffers alternate; the copy runs on its own stream as soon as the step before the
        previous one (the last reader of that device buffer) has finished."""
        st = self.__dict__.get('_zstage')
        if st is not None and (st['shape'] != shape or st['device'] != device):
            self.cancel_z_prefetch()        # a look-ahead draw for the old shape: undone
        if st is None or st['shape'] != shape or st['device'] != device:
            st = {'shape': shape, 'device': device, 'i': 0, 'stream': torch.cuda.Stream(device=device),
                  'pin': [torch.empty(shape, pin_memory=True) for _ in range(2)],
                  'dev': [torch.empty(shape, device=device) for _ in range(2)],
                  'copied': [None, None], 'main': None}
            self.__dict__['_zstage'] = st
        i = st['i']
        st['i'] ^= 1
        job = st.pop('job', None)
        if job is not None:
            # the draw for THIS call was started by the previous one (z_prefetch) into pin[i]
            job['thread'].join()
            if job['error'] is not None:
                raise job['error']
            if self._z_prefetch_raced(job):
                job = None      # draw again, from the generator as it stands (no rewind)
            elif job['shape'] != shape or job['i'] != i:
                # not what this call needs (a shorter last batch): put the generator back where
                # the reference's would be and draw again
                torch.set_rng_state(job['state'])
                job = None
        if job is None:
            if st['copied'][i] is not None:
                st['copied'][i].synchronize()      # the pinned buffer is free again (long done)
            torch.randn(shape, out=st['pin'][i])
        main = torch.cuda.current_stream(device)
        if st['main'] is not None:
            st['stream'].wait_event(st['main'])    # readers of dev[i] (two calls ago) are done
        with torch.cuda.stream(st['stream']):
            st['dev'][i].copy_(st['pin'][i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st['stream'])
        st['copied'][i] = ev
        main.wait_event(ev)
        me = torch.cuda.Event()
        me.record(main)
        st['main'] = me
        if self.z_prefetch and not self.__dict__.get('z_prefetch_disabled', False):
            # the NEXT call's z, drawn by a host thread while this step's launches go out (randn
            # releases the GIL): same numbers as the reference's draw as long as nothing else
            # takes from torch's global CPU generator before the next call — the training loop
            # switches z_prefetch off around the draws it knows of (epoch ends; see SEGAN.train)
            j = i ^ 1
            # the generator's state is snapshotted HERE, on the calling thread; the draw thread
            # notes what it found and what it left, and the join checks both (_z_prefetch_raced):
            # any other taker of the global generator in between is detected, not silently mixed in
            nxt = {'shape': shape, 'i': j, 'state': torch.get_rng_state(), 'before': None,
                   'after': None, 'error': None}

            def draw():
                try:
                    if st['copied'][j] is not None:
                        st['copied'][j].synchronize()
                    nxt['before'] = torch.get_rng_state()
                    torch.randn(shape, out=st['pin'][j])
                    nxt['after'] = torch.get_rng_state()
                except BaseException as e:      # surfaced by the next call
                    nxt['error'] = e
            nxt['thread'] = threading.Thread(target=draw, daemon=True)
            nxt['thread'].start()
            st['job'] = nxt
        z = st['dev'][i]
        z._segan_staged = True
        return z

    def _z_prefetch_raced(self, job):
        """True when something else took from torch's global CPU generator while the look-ahead
        draw `job` was pending (a num_workers=0 dataset or collate using torch RNG, a user hook,
        dropout drawn on the host ...): the generator did not stand where the calling thread left
        it when the draw began, or does not stand where the draw left it now.  The order of those
        takes relative to the z draw is then a matter of thread timing, so the look-ahead is
        switched off for good (``z_prefetch_disabled``; SEGAN.train's per-step switch honours it),
        with one warning; the caller draws z synchronously from the generator as it stands."""
        if job['after'] is None:
            return False
        if torch.equal(job['before'], job['state']) and torch.equal(torch.get_rng_state(), job['after']):
            return False
        import warnings
        warnings.warn('Generator.z_prefetch: torch\'s global CPU generator was used by something else '
                      'while the next z was being drawn on the host thread; the look-ahead draw is '
                      'switched off for the rest of the run (same as --no_prefetch_z) so that the draw '
                      'order stays deterministic', RuntimeWarning)
        self.__dict__['z_prefetch_disabled'] = True
        self.z_prefetch = False
        return True

    def cancel_z_prefetch(self):
        """Undo a pending look-ahead draw (the generator goes back to where it was — unless
        something else has used it meanwhile: then it is left alone, see _z_prefetch_raced)."""
        st = self.__dict__.get('_zstage')
        job = st.pop('job', None) if st else None
        if job is not None:
            job['thread'].join()
            if job['state'] is not None and not self._z_prefetch_raced(job):
                torch.set_rng_state(job['state'])

    def _fn_params(self):
        return self.all_params()

    def forward(self, x, z=None, ret_hid=False):
        if x.dim() != 3:
            raise ValueError('Generator expects [B, C, L], got {}'.format(tuple(x.shape)))
        if x.shape[2] % self._total_pool != 0:
            raise ValueError('input length {} is not divisible by the total pooling {}'.format(
                x.shape[2], self._total_pool))
        if not self.no_z:
            if z is None:
                zshape = (x.size(0), self.z_dim, x.shape[2] // self._total_pool)
                if getattr(self, 'z_generator', None) is not None and x.is_cuda:
                    # opt-in (train.py --device_z): drawn on the GPU from a per-rank generator;
                    # no host randn + H2D copy per step, but not the reference's RNG stream
                    z = torch.randn(zshape, device=x.device, generator=self.z_generator)
                elif x.is_cuda:
                    z = self._host_z(zshape, x.device)
                else:
                    z = torch.randn(*zshape)
            if z.dim() != x.dim():
                raise ValueError('len(z.size) {} != len(hi.size) {}'.format(z.dim(), x.dim()))
            if getattr(z, '_segan_staged', False) and (torch.is_grad_enabled() or
                                                       not hasattr(self, 'z')):
                # a staging buffer is re-used two calls later, but the autograd node keeps z for
                # the first decoder layer's weight gradient (several graphs may be alive at once:
                # gradient accumulation, held fake batches): hand it its own copy (20 MB, ~10 us)
                z = z.clone()
            if not hasattr(self, 'z'):
                self.z = z
        else:
            z = None
        out = Fn.GeneratorFn.apply(self, bool(ret_hid), x, z, *self._fn_params())
        if not ret_hid:
            return out
        keys = ['enc_{}'.format(i) for i in range(len(self.enc_blocks))]
        if not self.no_z:
            keys.append('enc_zc')
        keys += ['dec_{}'.format(i) for i in range(len(self.dec_blocks))]
        return out[0], dict(zip(keys, out[1:]))
