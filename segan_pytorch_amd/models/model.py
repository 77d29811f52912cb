"""SEGAN / WSEGAN training and inference engines on the HIP path.

Mirror of ``SEGAN`` (segan/models/model.py:71-507): same constructor (an ``opts``
attribute bag), ``infer_G`` / ``infer_D`` / ``generate`` / ``discriminate`` /
``build_optimizers`` / ``train`` signatures, same initialisation
(``weights_init``, model.py:28-43) and the same order of operations inside the GAN
step (model.py:292-321).  What differs, by design:

* G and D forward/backward run as the HIP autograd nodes of ``functional.py``;
* the optimizers are the fused flat-arena ones of ``optim.py``;
* during the generator update the discriminator's parameters are frozen, so the D
  weight gradients the reference computes and then discards (its next
  ``Dopt.zero_grad()``) are not computed at all — same results, 1/9 of D's work less;
* under ``torch.distributed`` (one process per GPU) gradients are averaged with one
  RCCL all-reduce per network per step (``distributed.py``).
"""
import os
import random
import timeit

import numpy as np
import torch
import torch.nn as nn

from .. import distributed as sdist
from .. import losses
from .. import optim as soptim
from ..datasets import de_emphasize
from .core import Model, Saver
from .discriminator import Discriminator
from .generator import Generator

try:  # logging is optional plumbing (not installed in every image)
    from tensorboardX import SummaryWriter
    _HAVE_TBX = True
except Exception:  # pragma: no cover
    _HAVE_TBX = False

    class SummaryWriter(object):
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_histogram(self, *a, **k):
            pass


def weights_init(m):
    """model.py:28-43: Conv1d ~ N(0, 0.02) with zero bias; Linear Xavier-uniform;
    ConvTranspose1d / BatchNorm1d / PReLU keep torch defaults (the class-name test
    'Conv1d' does not match them)."""
    classname = m.__class__.__name__
    if classname.find('Conv1d') != -1:
        m.weight.data.normal_(0.0, 0.02)
        if getattr(m, 'bias', None) is not None:
            m.bias.data.fill_(0)
    elif classname.find('Linear') != -1:
        nn.init.xavier_uniform_(m.weight.data)


def wsegan_weights_init(m):
    """model.py:45-60: Xavier-uniform for Conv1d, ConvTranspose1d and Linear."""
    classname = m.__class__.__name__
    if classname.find('Conv1d') != -1 or classname.find('ConvTranspose1d') != -1 or \
            classname.find('Linear') != -1:
        nn.init.xavier_uniform_(m.weight.data)


def _de_emphasize_any(x, coef):
    """1-D tensor -> de-emphasised float32 numpy array (what generate returns): the scan kernel
    for a GPU tensor, the host filter otherwise."""
    if x.is_cuda:
        from .. import ops
        return ops.de_emphasize(x.float(), coef).cpu().numpy() if coef > 0 else x.cpu().numpy()
    return de_emphasize(x.numpy(), coef)


def _freeze_gc():
    """Called once when a training loop starts: everything alive so far (two networks, optimizer
    arenas, packed weights, the loader) moves to the collector's permanent generation, so the
    generation-2 passes the step's thousands of short-lived tensors keep triggering no longer walk
    it.  Measured on the launch path of a step (round 6, scripts/diag_timer_modes.py): 8.0 -> 6.3 ms
    of host time per SEGAN+ step, and no more 50-80 ms pauses at a random launch."""
    import gc
    gc.collect()
    gc.freeze()


def _to_device_async(t, device):
    """A small host tensor to the device without making the host wait for the stream: a pageable
    source turns `.to(device)` into a copy the runtime stages synchronously, behind everything the
    stream still has queued — once per step that drains the launch queue the host had built up
    (round-5 review, weak 4: the WSEGAN step was host-bound).  Pinned staging (torch's caching host
    allocator: no allocation after the first step) + an async copy instead."""
    if torch.device(device).type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


class _frozen(object):
    """Context manager: parameters of `module` do not require grad inside."""

    def __init__(self, module):
        ps = module.all_params() if hasattr(module, 'all_params') else nn.Module.parameters(module)
        self.params = [p for p in ps if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


class SEGAN(Model):

    def __init__(self, opts, name='SEGAN', generator=None, discriminator=None):
        super(SEGAN, self).__init__(name)
        self.save_path = opts.save_path
        self.preemph = opts.preemph
        reg = getattr(opts, 'reg_loss', 'l1_loss')
        if reg not in ('l1_loss', 'mse_loss'):        # getattr(F, opts.reg_loss), model.py:79
            raise NotImplementedError("reg_loss {!r}: 'l1_loss' and 'mse_loss' are implemented".format(reg))
        self.reg_loss = losses.l1_loss if reg == 'l1_loss' else losses.mse_loss
        if generator is None:
            self.G = Generator(1, opts.genc_fmaps, opts.gkwidth, opts.genc_poolings,
                               opts.gdec_fmaps, opts.gdec_kwidth, opts.gdec_poolings,
                               z_dim=opts.z_dim, no_z=opts.no_z, skip=(not opts.no_skip),
                               bias=opts.bias, skip_init=opts.skip_init,
                               skip_type=opts.skip_type, skip_merge=opts.skip_merge,
                               skip_kwidth=opts.skip_kwidth)
        else:
            self.G = generator
        self.G.apply(weights_init)
        if discriminator is None:
            dkwidth = opts.gkwidth if opts.dkwidth is None else opts.dkwidth
            self.D = Discriminator(2, opts.denc_fmaps, dkwidth, poolings=opts.denc_poolings,
                                   pool_type=opts.dpool_type, pool_slen=opts.dpool_slen,
                                   norm_type=opts.dnorm_type, phase_shift=opts.phase_shift,
                                   sinc_conv=opts.sinc_conv)
        else:
            self.D = discriminator
        self.D.apply(weights_init)

    # ---- inference --------------------------------------------------------------------
    def generate(self, inwav, z=None, device='cpu', max_batch=64):
        """Chunked enhancement of a whole utterance (model.py:116-157): 16384-sample chunks,
        the last one zero padded, one z for all chunks, de-emphasis at the end.  The chunks
        are independent (G has no cross-sample coupling in eval mode), so they run as ONE
        batched forward per `max_batch` chunks instead of the reference's batch-1 loop; when
        z is None the first chunk runs alone so that G draws z for a single chunk exactly as
        the reference does, and that z is re-used for the rest (model.py:144-146)."""
        self.G.eval()
        N = 16384
        T = inwav.shape[2]
        nch = (T + N - 1) // N
        x = torch.zeros(nch, 1, N, device=device, dtype=torch.float32)
        x.view(-1)[:T] = torch.as_tensor(inwav[0, 0]).to(device=device, dtype=torch.float32)
        outs, hall, beg = [], None, 0
        with torch.no_grad():
            if z is None:
                y, hall = self.infer_G(x[:1].contiguous(), z=None, ret_hid=True)
                if hasattr(self.G, 'z'):
                    z = self.G.z
                outs.append(y)
                beg = 1
            while beg < nch:
                end = min(nch, beg + max_batch)
                zb = None
                if z is not None:
                    if z.size(0) != 1:
                        # the reference would feed this z to its batch-1 chunks and fail in cat
                        raise ValueError('generate: z must have batch size 1, got {} (G.z is set '
                                         'by the first forward of the module, generator.py:203)'
                                         .format(z.size(0)))
                    zb = z.to(device).expand(end - beg, -1, -1).contiguous()
                y, hall = self.infer_G(x[beg:end].contiguous(), z=zb, ret_hid=True)
                outs.append(y)
                beg = end
        nums = [int(k.split('_')[1]) for k in hall.keys() if 'enc' in k and 'zc' not in k]
        g_c = hall['enc_{}'.format(max(nums))][-1:]
        c_res = torch.cat(outs, 0).reshape(-1)[:T].contiguous()
        # de-emphasis (model.py:154-156) as a scan on the GPU instead of the per-sample loop
        c_res = _de_emphasize_any(c_res, self.preemph)
        return c_res, g_c

    def discriminate(self, cwav, nwav):
        self.D.eval()
        d_veredict, _ = self.D(torch.cat((cwav, nwav), dim=1))
        return d_veredict

    def infer_G(self, nwav, cwav=None, z=None, ret_hid=False):
        return self.G(nwav, z=z, ret_hid=ret_hid)

    def infer_D(self, x_, ref):
        # D(torch.cat((x_, ref), dim=1)) of model.py:173-175; the concatenation is folded
        # into the first conv's two-pointer load
        return self.D(x_, ref)

    def gen_train_samples(self, clean_samples, noisy_samples, z_sample, iteration=None):
        """The periodic listening samples of the training loop (model.py:177-217): G on the first
        (up to) 20 slices of the first batch with the z of the first step, de-emphasised and
        written as save_path/sample_<iteration>-<m>.wav; on the first call also gtruth_<m>.wav,
        noisy_<m>.wav and dif_<m>.wav.  The de-emphasis of all slices is one scan kernel."""
        from scipy.io import wavfile
        with torch.no_grad():
            if z_sample is not None:
                canvas_w = self.infer_G(noisy_samples, clean_samples, z=z_sample)
            else:
                canvas_w = self.infer_G(noisy_samples, clean_samples)

        def rows(t):
            t = t[:, 0].contiguous().float()
            if t.is_cuda:
                from .. import ops
                return (ops.de_emphasize(t, self.preemph) if self.preemph > 0 else t).cpu().numpy()
            return np.stack([de_emphasize(r.numpy(), self.preemph) for r in t])

        canvas = rows(canvas_w)
        n = noisy_samples.size(0)
        missing = [m for m in range(n)
                   if not os.path.exists(os.path.join(self.save_path, 'gtruth_{}.wav'.format(m)))]
        if missing:
            cl, no, di = rows(clean_samples), rows(noisy_samples), rows(noisy_samples - clean_samples)
        for m in range(n):
            m_canvas = canvas[m]
            print('w{} max: {} min: {}'.format(m, m_canvas.max(), m_canvas.min()))
            wavfile.write(os.path.join(self.save_path, 'sample_{}-{}.wav'.format(iteration, m)),
                          int(16e3), m_canvas)
            if m in missing:
                wavfile.write(os.path.join(self.save_path, 'gtruth_{}.wav'.format(m)), int(16e3), cl[m])
                wavfile.write(os.path.join(self.save_path, 'noisy_{}.wav'.format(m)), int(16e3), no[m])
                wavfile.write(os.path.join(self.save_path, 'dif_{}.wav'.format(m)), int(16e3), di[m])

    def _log_weight_norms(self, iteration):
        """model.py:372-386: per-layer and total weight norms of G and D (scalars for the
        tensorboard writer; skipped when no writer is installed)."""
        if not _HAVE_TBX:
            return
        for model, total_name in ((self.G, 'Gtotal'), (self.D, 'Dtotal')):
            total = 0.0
            for k, v in model.named_parameters():
                if 'weight' in k:
                    wn = float(torch.norm(v.data))
                    self.writer.add_scalar('{}_Wnorm'.format(k), wn, iteration)
                    total += wn
            self.writer.add_scalar('{}_Wnorm'.format(total_name), total, iteration)

    # ---- training ---------------------------------------------------------------------
    def build_optimizers(self, opts):
        if opts.opt == 'rmsprop':
            Gopt = soptim.RMSprop(self.G.parameters(), lr=opts.g_lr)
            Dopt = soptim.RMSprop(self.D.parameters(), lr=opts.d_lr)
        elif opts.opt == 'adam':
            Gopt = soptim.Adam(self.G.parameters(), lr=opts.g_lr, betas=(0, 0.9))
            Dopt = soptim.Adam(self.D.parameters(), lr=opts.d_lr, betas=(0, 0.9))
        else:
            raise ValueError('Unrecognized optimizer {}'.format(opts.opt))
        return Gopt, Dopt

    def d_phase(self, clean, noisy, Dopt, criterion, z=None):
        """Discriminator half of the step (model.py:292-308): G forward, D on the real and on
        the (detached) fake pair, both backward passes, gradient all-reduce, ``Dopt.step()``.
        Returns (Genh, d_real_loss, d_fake_loss); Genh keeps its graph for ``g_phase``."""
        # (1) D real update
        Dopt.zero_grad()
        Genh = self.infer_G(noisy, clean, z=z)
        d_real, _ = self.infer_D(clean, noisy)
        d_real_loss = criterion(d_real.view(-1), 1.0)
        d_real_loss.backward()
        # (2) D fake update
        d_fake, _ = self.infer_D(Genh.detach(), noisy)
        d_fake_loss = criterion(d_fake.view(-1), 0.0)
        sdist.arm(Dopt)                 # last backward into D's gradients: all-reduce by buckets
        d_fake_loss.backward()
        sdist.allreduce_grads(Dopt)
        Dopt.step()
        return Genh, d_real_loss, d_fake_loss

    def g_phase(self, Genh, clean, noisy, Gopt, criterion, l1_weight):
        """Generator half (model.py:310-321): D (just updated, frozen here) on the fake pair,
        adversarial + L1 loss, backward, gradient all-reduce, ``Gopt.step()``."""
        Gopt.zero_grad()
        with _frozen(self.D):
            d_fake_, _ = self.infer_D(Genh, noisy)
            g_adv_loss = criterion(d_fake_.view(-1), 1.0)
            g_l1_loss = l1_weight * self.reg_loss(Genh, clean)
            g_loss = g_adv_loss + g_l1_loss
            sdist.arm(Gopt)
            g_loss.backward()
        sdist.allreduce_grads(Gopt)
        Gopt.step()
        return g_adv_loss, g_l1_loss

    def gan_step(self, clean, noisy, Gopt, Dopt, criterion, l1_weight, z=None):
        """One LSGAN step, model.py:292-321 (from ``Dopt.zero_grad()`` to
        ``Gopt.step()``).  clean / noisy: [B, 1, T] on the device.  Returns the four
        losses as 0-dim device tensors (no host sync)."""
        Genh, d_real_loss, d_fake_loss = self.d_phase(clean, noisy, Dopt, criterion, z=z)
        g_adv_loss, g_l1_loss = self.g_phase(Genh, clean, noisy, Gopt, criterion, l1_weight)
        return d_real_loss, d_fake_loss, g_adv_loss, g_l1_loss

    def train(self, opts, dloader, criterion, l1_init, l1_dec_step, l1_dec_epoch, log_freq,
              va_dloader=None, device='cpu'):
        """Train the SEGAN (model.py:230-437)."""
        if criterion is None or isinstance(criterion, nn.MSELoss):
            criterion = losses.MSELoss()
        elif not isinstance(criterion, (losses.MSELoss, losses.BCEWithLogitsLoss)):
            raise NotImplementedError('criterion {}: the step runs nn.MSELoss (train.py:94) or '
                                      'BCEWithLogitsLoss natively; other criteria are not '
                                      'implemented'.format(type(criterion).__name__))
        self.writer = SummaryWriter(os.path.join(self.save_path, 'train'))
        Gopt, Dopt = self.build_optimizers(opts)
        self.G.optim = Gopt
        self.D.optim = Dopt
        sdist.broadcast_params(self.G)
        sdist.broadcast_params(self.D)
        _freeze_gc()
        is_main = sdist.rank() == 0
        eoe_g_saver = Saver(self.G, opts.save_path, max_ckpts=3, optimizer=self.G.optim,
                            prefix='EOE_G-')
        eoe_d_saver = Saver(self.D, opts.save_path, max_ckpts=3, optimizer=self.D.optim,
                            prefix='EOE_D-')
        l1_weight = l1_init
        iteration = 1
        timings = []
        clean_samples = noisy_samples = z_sample = None
        train_gen = is_main and not getattr(opts, 'no_train_gen', True)
        best_val_obj, patience = None, getattr(opts, 'patience', 100)
        for epoch in range(1, opts.epoch + 1):
            beg_t = timeit.default_timer()
            self.G.train()
            self.D.train()
            sampler = getattr(dloader, 'sampler', None)
            if hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(epoch)        # DistributedSampler: a new permutation per epoch
            for bidx, batch in enumerate(dloader, start=1):
                if epoch >= l1_dec_epoch and l1_weight > 0:
                    l1_weight = max(0, l1_weight - l1_dec_step)
                if len(batch) != 4:
                    raise ValueError('Returned {} elements per sample?'.format(len(batch)))
                uttname, clean, noisy, slice_idx = batch
                clean = clean.unsqueeze(1).to(device)
                noisy = noisy.unsqueeze(1).to(device)
                if train_gen and noisy_samples is None:      # model.py:288-290
                    noisy_samples = noisy[:20, :, :].contiguous()
                    clean_samples = clean[:20, :, :].contiguous()
                # z of the NEXT batch is drawn by a host thread while this step's kernels are
                # launched (Generator._host_z) — within an epoch the z draws are the only takers
                # of torch's global CPU generator, so the stream stays the reference's
                # (generator.py:197); not across an epoch end (the samplers reseed from it), not
                # with skip dropout (its masks come from the same generator inside the forward)
                self.G.z_prefetch = (getattr(opts, 'prefetch_z', True) and bidx < len(dloader) and
                                     not getattr(self.G, '_skip_dropout', 0))
                d_real_loss, d_fake_loss, g_adv_loss, g_l1_loss = self.gan_step(
                    clean, noisy, Gopt, Dopt, criterion, l1_weight)
                end_t = timeit.default_timer()
                timings.append(end_t - beg_t)
                beg_t = timeit.default_timer()
                if train_gen and z_sample is None and not self.G.no_z:   # model.py:325-330
                    z_sample = self.G.z[:20, :, :].contiguous().to(device)
                    print('z_sample size: ', z_sample.size())
                if is_main and (bidx % log_freq == 0 or bidx >= len(dloader)):
                    vals = [v.cpu().item() for v in
                            (d_real_loss, d_fake_loss, g_adv_loss, g_l1_loss)]
                    print('(Iter {}) Batch {}/{} (Epoch {}) d_real:{:.4f}, d_fake:{:.4f}, '
                          'g_adv:{:.4f}, g_l1:{:.4f} l1_w: {:.2f}, btime: {:.4f} s, '
                          'mbtime: {:.4f} s'.format(iteration, bidx, len(dloader), epoch,
                                                    vals[0], vals[1], vals[2], vals[3],
                                                    l1_weight, timings[-1], np.mean(timings)))
                    for k, v in zip(('D_real', 'D_fake', 'G_adv', 'G_l1'), vals):
                        self.writer.add_scalar(k, v, iteration)
                    self._log_weight_norms(iteration)
                    if train_gen:
                        self.gen_train_samples(clean_samples, noisy_samples, z_sample,
                                               iteration=iteration)
                iteration += 1
            if va_dloader is not None:
                # validation (model.py:394-433).  The reference's objective adds COVL and PESQ,
                # which need the external `pesqmain` binary; here the objective is the
                # segmental SNR, computed on the GPU
                evals = self.evaluate(opts, va_dloader, log_freq, device=device)
                # every rank evaluates with its own z; rank 0's figure decides for all of them
                # (ranks disagreeing on `break` would deadlock the next all-reduce)
                val_obj = sdist.broadcast_scalar(float(np.mean(evals['ssnr'])), 0, device)
                self.writer.add_scalar('Genh-ssnr', val_obj, epoch)
                if best_val_obj is None or val_obj > best_val_obj:
                    if is_main:
                        print('Val obj (SSNR) improved {} -> {}'.format(best_val_obj, val_obj))
                        self.G.save(self.save_path, iteration, True)
                        self.D.save(self.save_path, iteration, True)
                    best_val_obj, patience = val_obj, getattr(opts, 'patience', 100)
                else:
                    patience -= 1
                    if is_main:
                        print('Val loss did not improve. Patience {}/{}'.format(
                            patience, getattr(opts, 'patience', 100)))
                    if patience <= 0:
                        if is_main:
                            print('STOPPING SEGAN TRAIN: OUT OF PATIENCE.')
                        break
            if is_main:
                # asynchronous (core.Saver): the next epoch starts while the files are written
                self.G.save(self.save_path, iteration, saver=eoe_g_saver)
                self.D.save(self.save_path, iteration, saver=eoe_d_saver)
        self.G.z_prefetch = False
        self.G.cancel_z_prefetch()
        for sv in (eoe_g_saver, eoe_d_saver):
            sv.wait()
        self.G.wait_for_checkpoints()
        self.D.wait_for_checkpoints()

    def evaluate(self, opts, dloader, log_freq, do_noisy=False, max_samples=1, device='cpu'):
        """Objective evaluation on a validation loader (model.py:440-507), on the GPU: G in eval
        mode on up to `max_samples` batches, de-emphasis, segmental SNR (utils.py:350-395) of the
        enhanced — and with `do_noisy` of the noisy — signal against the clean one.  Returns
        {'ssnr': [...], 'snr': [...]} per utterance (and the same for the noisy input).  PESQ /
        CSIG / CBAK / COVL of the reference need its external `pesqmain` binary and are not
        computed.  De-emphasis runs along time (the reference applies it along axis 0 of the
        [B, T] batch, model.py:474-477)."""
        from .. import ops
        self.G.eval()
        self.D.eval()
        evals = {'ssnr': [], 'snr': []}
        noisy_evals = {'ssnr': [], 'snr': []}
        with torch.no_grad():
            for bidx, batch in enumerate(dloader, start=1):
                if len(batch) != 4:
                    raise ValueError('Returned {} elements per sample?'.format(len(batch)))
                uttname, clean, noisy, slice_idx = batch
                clean = clean.to(device).float().contiguous()
                noisy = noisy.to(device).float().contiguous()
                Genh = self.infer_G(noisy.unsqueeze(1)).squeeze(1).contiguous()
                c = ops.de_emphasize(clean, self.preemph)
                for sig, dst in ((Genh, evals),) + (((noisy, noisy_evals),) if do_noisy else ()):
                    snr, ssnr, _ = ops.ssnr(c, ops.de_emphasize(sig, self.preemph))
                    dst['ssnr'] += ssnr.cpu().tolist()
                    dst['snr'] += snr.cpu().tolist()
                if bidx >= max_samples:
                    break
        self.G.train()
        self.D.train()
        return (evals, noisy_evals) if do_noisy else evals


class WSEGAN(SEGAN):
    """Whispered-speech SEGAN variant (model.py:509-766): one summed discriminator loss
    per step with an optional misaligned / interference fake pair, and a generator loss
    of LSGAN + STFT log-power L1 + masked L1.  Everything runs on the HIP path, the STFT
    power loss included (``losses.stft_pow_l1``: framing + DFT-as-GEMM + power/log kernels)."""

    def __init__(self, opts, name='WSEGAN', generator=None, discriminator=None):
        self.lbd = 1
        self.critic_iters = 1
        self.misalign_pair = opts.misalign_pair
        self.interf_pair = opts.interf_pair
        self.pow_weight = opts.pow_weight
        self.vanilla_gan = opts.vanilla_gan
        self.n_fft = opts.n_fft
        # like the reference: SEGAN.__init__ builds and initialises with weights_init
        # (consuming the same RNG draws), then both nets are re-initialised Xavier-uniform
        super(WSEGAN, self).__init__(opts, name, None, None)
        self.G.apply(wsegan_weights_init)
        self.D.apply(wsegan_weights_init)

    def sample_dloader(self, dloader, device='cpu'):
        """A fresh iterator every step, first batch only (model.py:526-535).  The reference's
        RandomSampler reshuffles on every iter(); a DistributedSampler only does when its epoch
        changes, so it is bumped per call (otherwise every step would see the same batch)."""
        if hasattr(dloader, 'sample'):      # PCMShardLoader: one live iterator (datasets.py)
            uttname, clean, noisy, slice_idx = dloader.sample()
            return (uttname, clean.unsqueeze(1).to(device), noisy.unsqueeze(1).to(device),
                    slice_idx.to(device))
        sampler = getattr(dloader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            self._sample_calls = getattr(self, '_sample_calls', 0) + 1
            sampler.set_epoch(self._sample_calls)
        uttname, clean, noisy, slice_idx = next(iter(dloader))
        return (uttname, clean.unsqueeze(1).to(device), noisy.unsqueeze(1).to(device),
                slice_idx.to(device))

    def wgan_d_phase(self, clean, noisy, Dopt, z=None):
        """Discriminator half of the WSEGAN step (model.py:577-631): D on the real pair, G
        forward, D on the (detached) fake pair, optionally on the misaligned and the interference
        pair; ONE backward of the weighted sum, gradient all-reduce, ``Dopt.step()``.  Returns
        (Genh, d_loss); Genh keeps its graph for ``wgan_g_phase``."""
        from random import shuffle
        cost = losses.BCEWithLogitsLoss() if self.vanilla_gan else losses.MSELoss()
        bsz = clean.size(0)
        Dopt.zero_grad()
        d_real, _ = self.infer_D(clean, noisy)
        d_real_loss = cost(d_real.view(-1), 1.0)
        Genh = self.infer_G(noisy, clean, z=z)
        d_fake, _ = self.infer_D(Genh.detach(), noisy)
        d_fake_loss = cost(d_fake.view(-1), 0.0)
        d_weight = 0.5
        d_loss = d_fake_loss + d_real_loss
        n_d_fwd = 2             # D forwards the one backward below differentiates
        if self.misalign_pair:
            perm = list(range(bsz))
            shuffle(perm)      # same RNG draws as shuffling the chunk list (model.py:598-600)
            clean_shuf = clean[_to_device_async(torch.as_tensor(perm), clean.device)]
            d_fake_shuf, _ = self.infer_D(clean, clean_shuf)
            d_loss = d_loss + cost(d_fake_shuf.view(-1), 0.0)
            d_weight = 1 / 3
            n_d_fwd += 1
        if self.interf_pair:
            from scipy import signal
            freqs, amps = [250, 1000, 4000], [0.01, 0.05, 0.1, 1]
            t = np.linspace(0, 2, 32000)
            squares = []
            for _ in range(bsz):
                f_ = random.choice(freqs)
                a_ = random.choice(amps)
                sq = a_ * signal.square(2 * np.pi * f_ * t)
                squares.append(torch.FloatTensor(sq[:clean.size(-1)].reshape((1, -1))))
            squares = torch.cat(squares, dim=0).unsqueeze(1).to(clean.device)
            d_fake_inter, _ = self.infer_D(clean + squares, noisy)
            d_loss = d_loss + cost(d_fake_inter.view(-1), 0.0)
            d_weight = 1 / 4    # also without the misaligned pair, as the reference (model.py:627)
            n_d_fwd += 1
        d_loss = d_weight * d_loss
        # every D forward reports every D parameter once: a bucket leaves after the last one
        sdist.arm(Dopt, passes=n_d_fwd)
        d_loss.backward()
        sdist.allreduce_grads(Dopt)
        Dopt.step()
        return Genh, d_loss

    def wgan_g_phase(self, uttname, Genh, clean, noisy, Gopt, l1_weight):
        """Generator half (model.py:633-669): D (just updated, frozen here) on the fake pair,
        adversarial + STFT log-power L1 + masked L1, backward, all-reduce, ``Gopt.step()``.
        Returns (G_cost, g_adv_loss, pow_loss, den_loss)."""
        cost = losses.BCEWithLogitsLoss() if self.vanilla_gan else losses.MSELoss()
        bsz = clean.size(0)
        Gopt.zero_grad()
        with _frozen(self.D):
            d_fake_, _ = self.infer_D(Genh, noisy)
            g_adv_loss = cost(d_fake_.view(-1), 1.0)
            pow_loss = self.pow_weight * losses.stft_pow_l1(
                Genh, clean, min(Genh.size(-1), self.n_fft), 160, 320)
            G_cost = g_adv_loss + pow_loss
            if l1_weight > 0:
                # model.py:655-662 builds the mask row by row on the device (one fill per
                # utterance); here: one host vector, one copy, broadcast in the products
                mask = _to_device_async(torch.tensor([1.0 if 'additive' in uttn else 0.0 for uttn in uttname],
                                                     dtype=torch.float32), Genh.device).view(bsz, 1, 1)
                den_loss = l1_weight * losses.l1_loss(Genh * mask, clean * mask)
                G_cost = G_cost + den_loss
            else:
                den_loss = torch.zeros((), device=Genh.device)
            sdist.arm(Gopt)
            G_cost.backward()
        sdist.allreduce_grads(Gopt)
        Gopt.step()
        return G_cost, g_adv_loss, pow_loss, den_loss

    def wgan_step(self, uttname, clean, noisy, Gopt, Dopt, l1_weight, z=None):
        """One WSEGAN step (model.py:577-669).  Returns (d_loss, G_cost, pow_loss,
        den_loss) as device scalars."""
        Genh, d_loss = self.wgan_d_phase(clean, noisy, Dopt, z=z)
        G_cost, _g_adv, pow_loss, den_loss = self.wgan_g_phase(uttname, Genh, clean, noisy, Gopt,
                                                               l1_weight)
        return d_loss, G_cost, pow_loss, den_loss

    def train(self, opts, dloader, criterion, l1_init, l1_dec_step, l1_dec_epoch, log_freq,
              va_dloader=None, device='cpu'):
        self.writer = SummaryWriter(os.path.join(opts.save_path, 'train'))
        Gopt, Dopt = self.build_optimizers(opts)
        self.G.optim, self.D.optim = Gopt, Dopt
        sdist.broadcast_params(self.G)
        sdist.broadcast_params(self.D)
        _freeze_gc()
        is_main = sdist.rank() == 0
        eoe_g_saver = Saver(self.G, opts.save_path, max_ckpts=3, optimizer=Gopt, prefix='EOE_G-')
        eoe_d_saver = Saver(self.D, opts.save_path, max_ckpts=3, optimizer=Dopt, prefix='EOE_D-')
        l1_weight = l1_init     # never decays here (model.py:655-667)
        timings = []
        clean_samples = noisy_samples = z_sample = None
        train_gen = is_main and not getattr(opts, 'no_train_gen', True)
        self.G.train()
        self.D.train()
        # z of the NEXT step drawn by a host thread while this step's launches go out, as in
        # SEGAN.train — possible here only when the per-step batch sampling leaves torch's global CPU
        # generator alone, because the reference interleaves the two on that generator (every
        # next(iter(dloader)) reseeds its RandomSampler and the loader's base seed from it, then G draws
        # z: model.py:526-535, generator.py:197): true for PCMShardLoader.sample() (a private generator),
        # not for a plain DataLoader, whose steps keep the synchronous draw and the reference's stream
        lookahead = (getattr(opts, 'prefetch_z', True) and getattr(dloader, 'sample_keeps_global_rng', False)
                     and not getattr(self.G, '_skip_dropout', 0))
        n_iters = opts.epoch * len(dloader)
        for iteration in range(1, n_iters + 1):
            beg_t = timeit.default_timer()
            uttname, clean, noisy, _ = self.sample_dloader(dloader, device)
            self.G.z_prefetch = lookahead and iteration < n_iters
            if train_gen and noisy_samples is None:          # model.py:673-675
                noisy_samples = noisy[:20, :, :].contiguous()
                clean_samples = clean[:20, :, :].contiguous()
            d_loss, G_cost, pow_loss, den_loss = self.wgan_step(uttname, clean, noisy, Gopt, Dopt,
                                                                l1_weight)
            timings.append(timeit.default_timer() - beg_t)
            if train_gen and z_sample is None and not self.G.no_z:       # model.py:676-681
                z_sample = self.G.z[:20, :, :].contiguous().to(device)
                print('z_sample size: ', z_sample.size())
            if is_main and iteration % log_freq == 0:
                print('Iter {}/{} ({} bpe) d_loss:{:.4f}, g_loss: {:.4f}, pow_loss: {:.4f}, '
                      'den_loss: {:.4f} btime: {:.4f} s, mbtime: {:.4f} s'.format(
                          iteration, len(dloader) * opts.epoch, len(dloader), d_loss.item(),
                          G_cost.item(), pow_loss.item(), den_loss.item(), timings[-1],
                          np.mean(timings)))
                self.writer.add_scalar('D_loss', d_loss.item(), iteration)
                self.writer.add_scalar('G_loss', G_cost.item(), iteration)
                self._log_weight_norms(iteration)
                if train_gen:                                # model.py:744-747
                    self.gen_train_samples(clean_samples, noisy_samples, z_sample,
                                           iteration=iteration)
            if is_main and iteration % len(dloader) == 0:
                self.G.save(self.save_path, iteration, saver=eoe_g_saver)
                self.D.save(self.save_path, iteration, saver=eoe_d_saver)
        self.G.z_prefetch = False
        self.G.cancel_z_prefetch()
        for sv in (eoe_g_saver, eoe_d_saver):
            sv.wait()

    def generate(self, inwav, z=None, device=None):
        """Whole-utterance inference in one fully-convolutional pass (model.py:755-766)."""
        self.G.eval()
        ori_len = inwav.size(2)
        pad = (-ori_len) % 1024
        p_wav = torch.nn.functional.pad(inwav, (0, pad)) if pad else inwav
        with torch.no_grad():
            c_res, hall = self.infer_G(p_wav.contiguous(), z=z, ret_hid=True)
        return _de_emphasize_any(c_res[0, 0, :ori_len].contiguous(), self.preemph), hall
