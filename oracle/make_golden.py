"""Generate tests/golden/*.pt by running the REAL reference on CPU (build container only).

    python oracle/make_golden.py

Fixtures (all fp32, oneDNN disabled, deterministic seeds):

tiny_step.pt     a 3-layer SEGAN (fmaps 8/16/32, k31, s4, L=1024, B=3): full weights,
                 inputs, z, rolls, every output / loss / gradient of one GAN step made
                 with the reference's own modules + torch.optim.RMSprop in the order of
                 model.py:292-321, and the weights after it.
tiny_train2.pt   the same net driven through the reference's literal ``SEGAN.train``
                 for two batches (its own z draws and phase-shift draws, replayable
                 from the recorded seeds): final weights.
tiny_s2.pt       a stride-2 variant (vanilla-SEGAN style, fmaps 4/8/8/16, L=256) forward
                 and gradients.
tiny_wsegan2.pt  the tiny net as WSEGAN with --misalign_pair through the reference's literal
                 ``WSEGAN.train`` for two iterations (stft / .cuda() patched in the harness).
segan_plus_b2.pt the default SEGAN+ net (ckpt_segan+/train.opts, seed 111) at B=2:
                 per-tensor init checksums, G output, D logits, losses, and
                 checksums/samples of every gradient.
vanilla11_b8.pt  (python oracle/make_golden.py vanilla11) the original 11-layer stride-2 SEGAN
                 shape, one step at B=8, same content as segan_plus_b2.pt.
tiny_variants.pt (python oracle/make_golden.py variants) one GAN step of the tiny net for every
                 architecture switch train.py can reach that the headline nets do not use:
                 --skip_type conv (concat and sum merges), stride-1 (pooling 1) layers in the
                 encoder and the decoder (generator.py:171-176), a conv block as last decoder
                 layer, --dpool_type conv / gmax / gavg (discriminator.py:122-137).
tiny_gvariants.pt (python oracle/make_golden.py gvariants) the Generator options no train.py flag
                 reaches (model.py:82-96 never passes them): norm_type='bnorm' (BatchNorm in
                 every block, generator.py:126,166-176) and skip_dropout (generator.py:53-54,
                 70-71), alone and combined: output, eval-mode output and all gradients of a
                 linear loss, with the torch seed that drives the dropout masks.
tiny_nobias.pt / segan_plus_nobias_b2.pt (python oracle/make_golden.py nobias) --no_bias, the
                 reference's own batch-300 recipe (run_segan+_train.sh:7, train.py:248): G's
                 convs (and conv skips) are built without a bias, the transposed convs keep
                 theirs (modules.py:116-119 ignores the flag), D keeps all of its biases.  One
                 GAN step of the tiny net (full tensors) and of the default net at B=2
                 (checksums, like segan_plus_b2.pt).
tiny_corners.pt  (python oracle/make_golden.py corners) three corners train.py reaches that had no
                 reference golden until round 6: 'vanillagan' = the literal ``WSEGAN.train`` with
                 --vanilla_gan --misalign_pair (BCE-with-logits cost, model.py:582-585), two
                 iterations; 'constantskip' = --skip_type constant (generator.py:25,40,59: a
                 fixed per-channel scale, requires_grad False, so Model.parameters — core.py:
                 196-198 — hides it from the optimizer): one manual GAN step (skip_init randn)
                 and the literal ``SEGAN.train`` for two batches; 'mseloss' = --reg_loss mse_loss
                 (train.py:179, model.py:79): one manual step and the literal loop.
tiny_wsegan_interf.pt (python oracle/make_golden.py interf) the literal ``WSEGAN.train`` with
                 --interf_pair (model.py:606-628), two iterations each: 'both' =
                 --misalign_pair --interf_pair (four D forwards per step, d_weight 1/4),
                 'interf_only' = --interf_pair alone (three forwards, still weighted 1/4 as the
                 reference does).
"""
import json
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')


def base_opts():
    with open(os.path.join(ref_harness.REF_ROOT, 'ckpt_segan+', 'train.opts')) as f:
        o = json.load(f)
    o['reg_loss'] = 'l1_loss'       # the shipped file predates model.py:79
    return o


def tiny_opts(save_path='/tmp/segan_golden_ckpt'):
    o = base_opts()
    o.update(dict(genc_fmaps=[8, 16, 32], denc_fmaps=[8, 16, 32], genc_poolings=[4, 4, 4],
                  denc_poolings=[4, 4, 4], z_dim=32, dpool_slen=16, batch_size=3, epoch=1,
                  save_path=save_path, slice_size=1024, save_freq=1000, no_train_gen=True))
    return o


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def synth(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    clean = torch.rand(B, T, generator=g) * 2 - 1
    noisy = (clean + 0.1 * torch.randn(B, T, generator=g)).clamp(-1, 1)
    return clean, noisy


def clone_sd(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def grads_of(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def manual_step(ref, segan, clean, noisy, z, roll_seed, l1_weight=100.0, lr=5e-5, reg=F.l1_loss):
    """model.py:292-321 with the reference's modules, explicit z, seeded shifts.  `reg`: the
    regression loss getattr(F, opts.reg_loss) of model.py:79 (segan.reg_loss for --reg_loss)."""
    Gopt = torch.optim.RMSprop(segan.G.parameters(), lr=lr)
    Dopt = torch.optim.RMSprop(segan.D.parameters(), lr=lr)
    criterion = nn.MSELoss()
    segan.G.train()
    segan.D.train()
    B = clean.size(0)
    label = torch.ones(B)
    out = {}
    random.seed(roll_seed)
    Dopt.zero_grad()
    Genh = segan.infer_G(noisy, clean, z=z)
    d_real, _ = segan.infer_D(clean, noisy)
    d_real_loss = criterion(d_real.view(-1), label)
    d_real_loss.backward()
    d_fake, _ = segan.infer_D(Genh.detach(), noisy)
    d_fake_loss = criterion(d_fake.view(-1), label.clone().fill_(0))
    d_fake_loss.backward()
    out['d_grads'] = grads_of(segan.D)
    Dopt.step()
    Gopt.zero_grad()
    d_fake_, _ = segan.infer_D(Genh, noisy)
    g_adv = criterion(d_fake_.view(-1), label.clone().fill_(1))
    g_l1 = l1_weight * reg(Genh, clean)
    (g_adv + g_l1).backward()
    out['g_grads'] = grads_of(segan.G)
    Gopt.step()
    out['G_after'] = clone_sd(segan.G)
    out['D_after'] = clone_sd(segan.D)     # incl. BN buffers after all three D forwards
    out.update(Genh=Genh.detach().clone(), d_real=d_real.detach().clone(),
               d_fake=d_fake.detach().clone(), d_fake_=d_fake_.detach().clone(),
               d_real_loss=d_real_loss.detach().clone(), d_fake_loss=d_fake_loss.detach().clone(),
               g_adv_loss=g_adv.detach().clone(), g_l1_loss=g_l1.detach().clone())
    return out


def checksum(t):
    t = t.detach().double().reshape(-1)
    idx = torch.arange(0, t.numel(), max(1, t.numel() // 257))[:257]
    return {'sum': t.sum().item(), 'abs': t.abs().sum().item(), 'sq': (t * t).sum().item(),
            'n': t.numel(), 'sample_idx': idx, 'sample': t[idx].float().clone()}


def make_snorm(ref):
    """tiny_snorm.pt: (1) one SEGAN step with --dnorm_type snorm (spectral norm on D's convs,
    on fc[0], fc[2] and — as discriminator.py:118-121 literally does — on the PReLU fc[3]);
    the three D forwards each run one power iteration, so D_after also pins the u/v buffers.
    (2) a Generator built with norm_type='snorm' (generator.py:94,126,168; not reachable from
    train.py, which never forwards --gnorm_type): output and gradients of a linear loss."""
    o = tiny_opts()
    o['dnorm_type'] = 'snorm'
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**o))
    clean, noisy = synth(3, 1024, 10)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(3, 32, 16, generator=torch.Generator().manual_seed(11))
    fx = {'opts': o, 'G0': clone_sd(segan.G), 'D0': clone_sd(segan.D), 'clean': clean,
          'noisy': noisy, 'z': z, 'roll_seed': 13,
          'rolls': ref_harness.ReplayRandom(13).rolls(3, o['phase_shift'], 3)}
    fx.update(manual_step(ref, segan, clean, noisy, z, 13))
    seed_all(5)
    G = ref.Generator(1, [8, 16, 32], 31, [4, 4, 4], z_dim=32, skip_merge='concat', bias=True,
                      norm_type='snorm')
    G.train()
    g0 = clone_sd(G)
    gen = torch.Generator().manual_seed(12)
    x = torch.rand(2, 1, 1024, generator=gen) * 2 - 1
    zz = torch.randn(2, 32, 16, generator=gen)
    c = torch.randn(2, 1, 1024, generator=gen)
    y = G(x, z=zz)
    (y * c).sum().backward()
    fx['gsn'] = {'G0': g0, 'x': x, 'z': zz, 'c': c, 'y': y.detach().clone(),
                 'grads': grads_of(G), 'G_after_fwd': clone_sd(G)}
    torch.save(fx, os.path.join(OUT, 'tiny_snorm.pt'))
    print('tiny_snorm.pt done', sorted(k for k in fx['D0'] if 'fc.3' in k or 'enc_blocks.0.conv' in k))


def make_wsegan_snorm(ref):
    """tiny_wsegan_snorm.pt: the literal WSEGAN.train in the flavour of run_wsegan_train.sh
    (--wsegan --dnorm_type snorm --opt adam --misalign_pair), two iterations.  Same harness-only
    patches as tiny_wsegan2 (legacy torch.stft call, hard .cuda())."""
    ow = tiny_opts()
    ow.update(dict(wsegan=True, misalign_pair=True, cuda=False, save_freq=1000, dnorm_type='snorm',
                   opt='adam'))
    _stft = torch.stft
    torch.stft = lambda *a, **k: torch.view_as_real(_stft(*a, return_complex=True, **k))
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    # model.py:224-225 builds Adam(betas=(0, 0.9)); torch 2.x rejects the int 0 -> floats
    _Adam = torch.optim.Adam

    class _AdamFloatBetas(_Adam):
        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), **kw):
            super().__init__(params, lr=lr, betas=tuple(float(b) for b in betas), **kw)
    torch.optim.Adam = _AdamFloatBetas
    try:
        seed_all(111)
        wseg = ref.WSEGAN(SimpleNamespace(**ow))
        c1, n1 = synth(3, 1024, 14)
        names = ['utt_additive_0', 'utt_1', 'utt_additive_2']
        loader = [[names, c1, n1, torch.zeros(3)]]
        fxw = {'opts': ow, 'G0': clone_sd(wseg.G), 'D0': clone_sd(wseg.D), 'clean': c1,
               'noisy': n1, 'names': names, 'seed': 37, 'iters': 2}
        ow2 = dict(ow)
        ow2['epoch'] = 2
        seed_all(37)
        wseg.train(SimpleNamespace(**ow2), loader, None, ow['l1_weight'], ow['l1_dec_step'],
                   ow['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
        fxw['G_final'] = clone_sd(wseg.G)
        fxw['D_final'] = clone_sd(wseg.D)
        torch.save(fxw, os.path.join(OUT, 'tiny_wsegan_snorm.pt'))
        print('tiny_wsegan_snorm.pt done')
    finally:
        torch.stft = _stft
        torch.Tensor.cuda = _cuda
        torch.optim.Adam = _Adam


VANILLA11 = dict(genc_fmaps=[16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 1024],
                 denc_fmaps=[16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 1024],
                 genc_poolings=[2] * 11, denc_poolings=[2] * 11, dpool_slen=8)


def make_vanilla11(ref):
    """vanilla11_b8.pt: the original 11-layer stride-2 SEGAN shape (train.py:199-205 flags
    --genc_fmaps 16 32 32 64 64 128 128 256 256 512 1024 --genc_poolings 2 x11, same for D) at
    B=8 (at B=2 the deepest BatchNorm sees 16 values per channel and the step is ill-conditioned), one GAN step: like segan_plus_b2.pt (checksums of the big tensors)."""
    ob = base_opts()
    ob.update(VANILLA11)
    ob['save_path'] = '/tmp/segan_golden_ckpt'
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**ob))
    clean, noisy = synth(8, 16384, 0)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(8, 1024, 8, generator=torch.Generator().manual_seed(0))
    fx = {'opts': ob, 'batch': 8, 'seed': 111, 'roll_seed': 3, 'z_seed': 0, 'data_seed': 0,
          'rolls': ref_harness.ReplayRandom(3).rolls(11, ob['phase_shift'], 3),
          'init_G': {k: checksum(v) for k, v in segan.G.state_dict().items()},
          'init_D': {k: checksum(v) for k, v in segan.D.state_dict().items()}}
    res = manual_step(ref, segan, clean, noisy, z, 3)
    for k in ('Genh', 'd_real', 'd_fake', 'd_fake_', 'd_real_loss', 'd_fake_loss', 'g_adv_loss',
              'g_l1_loss'):
        fx[k] = res[k]
    fx['d_grads'] = {k: checksum(v) for k, v in res['d_grads'].items()}
    fx['g_grads'] = {k: checksum(v) for k, v in res['g_grads'].items()}
    fx['small_d_grads'] = {k: v for k, v in res['d_grads'].items() if v.numel() <= 4096}
    fx['small_g_grads'] = {k: v for k, v in res['g_grads'].items() if v.numel() <= 4096}
    torch.save(fx, os.path.join(OUT, 'vanilla11_b8.pt'))
    print('vanilla11_b8.pt done', res['Genh'].shape, res['g_l1_loss'],
          sum(p.numel() for p in segan.G.parameters()), sum(p.numel() for p in segan.D.parameters()))


VARIANTS = {
    # GSkip with a k=11 conv on the skip path (generator.py:42-49), both merges
    'skipconv_concat': dict(skip_type='conv', skip_merge='concat'),
    'skipconv_sum': dict(skip_type='conv', skip_merge='sum'),
    # stride-1 layers: GConv1DBlock pads (k//2, k//2) (modules.py:96-98); in the decoder a
    # pooling of 1 builds a GConv1DBlock instead of a transposed conv (generator.py:171-176)
    # and drops that level's skip (generator.py:212-213)
    'pool1_mid': dict(genc_poolings=[4, 1, 4], z_len=64),
    # a 4th decoder layer with pooling 1: the LAST block is a conv + PReLU, no Tanh
    'pool1_last': dict(gdec_fmaps=[8, 4, 2, 1], gdec_poolings=[4, 4, 4, 1], gdec_kwidth=31),
    # discriminator heads other than the dense one
    'dpool_conv': dict(dpool_type='conv'),
    'dpool_gmax': dict(dpool_type='gmax'),
    'dpool_gavg': dict(dpool_type='gavg'),
    # spectral norm on the conv head's pool_conv and fc (discriminator.py:125-127)
    'dpool_conv_snorm': dict(dpool_type='conv', dnorm_type='snorm'),
}


def make_variants(ref):
    out = {}
    for i, (name, delta) in enumerate(VARIANTS.items()):
        o = tiny_opts()
        delta = dict(delta)
        z_len = delta.pop('z_len', 16)
        o.update(dict(genc_fmaps=[4, 8, 16], denc_fmaps=[4, 8, 16], z_dim=16))   # small fixture
        o.update(delta)
        seed_all(200 + i)
        segan = ref.SEGAN(SimpleNamespace(**o))
        clean, noisy = synth(3, 1024, 40 + i)
        clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
        z = torch.randn(3, 16, z_len, generator=torch.Generator().manual_seed(60 + i))
        rs = 80 + i
        fx = {'opts': o, 'G0': clone_sd(segan.G), 'D0': clone_sd(segan.D), 'clean': clean,
              'noisy': noisy, 'z': z, 'roll_seed': rs,
              'rolls': ref_harness.ReplayRandom(rs).rolls(3, o['phase_shift'], 3)}
        fx.update(manual_step(ref, segan, clean, noisy, z, rs))
        out[name] = fx
        print(name, 'Genh', tuple(fx['Genh'].shape), 'd_real', tuple(fx['d_real'].shape),
              float(fx['g_l1_loss']), sorted(k for k in fx['G0'] if 'alpha' in k)[:2],
              sorted(k for k in fx['D0'] if not k.startswith('enc_blocks')))
    torch.save(out, os.path.join(OUT, 'tiny_variants.pt'))
    print('tiny_variants.pt done')


GVARIANTS = {
    'bnorm_concat': dict(norm_type='bnorm', skip_merge='concat'),
    'bnorm_sum': dict(norm_type='bnorm', skip_merge='sum'),
    'dropout_alpha': dict(skip_dropout=0.3, skip_merge='concat'),
    'dropout_conv_sum': dict(skip_dropout=0.3, skip_type='conv', skip_merge='sum'),
    'bnorm_dropout': dict(norm_type='bnorm', skip_dropout=0.2, skip_merge='concat'),
}


def make_gvariants(ref):
    out = {}
    for i, (name, kw) in enumerate(GVARIANTS.items()):
        seed_all(300 + i)
        kwargs = dict(z_dim=16, bias=True, skip_init='randn')
        kwargs.update(kw)
        G = ref.Generator(1, [4, 8, 16], 31, [4, 4, 4], **kwargs)
        G.train()
        g0 = clone_sd(G)
        gen = torch.Generator().manual_seed(400 + i)
        x = torch.rand(3, 1, 1024, generator=gen) * 2 - 1
        z = torch.randn(3, 16, 16, generator=gen)
        c = torch.randn(3, 1, 1024, generator=gen)
        fwd_seed = 500 + i
        torch.manual_seed(fwd_seed)                 # drives the dropout masks of this forward
        y = G(x, z=z)
        (y * c).sum().backward()
        fx = {'kwargs': kwargs, 'G0': g0, 'x': x, 'z': z, 'c': c, 'fwd_seed': fwd_seed,
              'y': y.detach().clone(), 'grads': grads_of(G), 'G_after_fwd': clone_sd(G)}
        G.eval()
        with torch.no_grad():
            fx['y_eval'] = G(x, z=z).clone()
        out[name] = fx
        print(name, tuple(y.shape), float(y.abs().mean()), sorted(fx['grads'])[:3])
    torch.save(out, os.path.join(OUT, 'tiny_gvariants.pt'))
    print('tiny_gvariants.pt done')


def make_nobias(ref):
    o = tiny_opts()
    o['no_bias'] = True
    o['bias'] = False                       # train.py:248
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**o))
    clean, noisy = synth(3, 1024, 20)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(3, 32, 16, generator=torch.Generator().manual_seed(21))
    fx = {'opts': o, 'G0': clone_sd(segan.G), 'D0': clone_sd(segan.D), 'clean': clean,
          'noisy': noisy, 'z': z, 'roll_seed': 17,
          'rolls': ref_harness.ReplayRandom(17).rolls(3, o['phase_shift'], 3)}
    fx.update(manual_step(ref, segan, clean, noisy, z, 17))
    torch.save(fx, os.path.join(OUT, 'tiny_nobias.pt'))
    print('tiny_nobias.pt done; G keys with bias:', sorted(k for k in fx['G0'] if 'bias' in k))
    ob = base_opts()
    ob['save_path'] = '/tmp/segan_golden_ckpt'
    ob['no_bias'] = True
    ob['bias'] = False
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**ob))
    clean, noisy = synth(2, 16384, 0)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(0))
    fx4 = {'opts': ob, 'seed': 111, 'roll_seed': 3, 'z_seed': 0, 'data_seed': 0,
           'rolls': ref_harness.ReplayRandom(3).rolls(5, ob['phase_shift'], 3),
           'init_G': {k: checksum(v) for k, v in segan.G.state_dict().items()},
           'init_D': {k: checksum(v) for k, v in segan.D.state_dict().items()}}
    res = manual_step(ref, segan, clean, noisy, z, 3)
    for k in ('Genh', 'd_real', 'd_fake', 'd_fake_', 'd_real_loss', 'd_fake_loss', 'g_adv_loss',
              'g_l1_loss'):
        fx4[k] = res[k]
    fx4['d_grads'] = {k: checksum(v) for k, v in res['d_grads'].items()}
    fx4['g_grads'] = {k: checksum(v) for k, v in res['g_grads'].items()}
    fx4['small_d_grads'] = {k: v for k, v in res['d_grads'].items() if v.numel() <= 4096}
    fx4['small_g_grads'] = {k: v for k, v in res['g_grads'].items() if v.numel() <= 4096}
    fx4['G_after'] = {k: checksum(v) for k, v in res['G_after'].items()}
    fx4['D_after'] = {k: checksum(v) for k, v in res['D_after'].items()}
    torch.save(fx4, os.path.join(OUT, 'segan_plus_nobias_b2.pt'))
    print('segan_plus_nobias_b2.pt done', res['Genh'].shape, res['g_l1_loss'],
          sum(p.numel() for p in segan.G.parameters()))


def make_interf(ref):
    """The literal WSEGAN.train with --interf_pair (same harness-only patches as tiny_wsegan2:
    legacy torch.stft call, hard .cuda())."""
    _stft = torch.stft
    torch.stft = lambda *a, **k: torch.view_as_real(_stft(*a, return_complex=True, **k))
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        for i, (name, mis) in enumerate((('both', True), ('interf_only', False))):
            ow = tiny_opts()
            ow.update(dict(wsegan=True, misalign_pair=mis, interf_pair=True, cuda=False,
                           save_freq=1000))
            seed_all(111)
            wseg = ref.WSEGAN(SimpleNamespace(**ow))
            c1, n1 = synth(3, 1024, 24 + i)
            names = ['utt_additive_0', 'utt_1', 'utt_additive_2']
            loader = [[names, c1, n1, torch.zeros(3)]]
            fxw = {'opts': ow, 'G0': clone_sd(wseg.G), 'D0': clone_sd(wseg.D), 'clean': c1,
                   'noisy': n1, 'names': names, 'seed': 41 + i, 'iters': 2}
            ow2 = dict(ow)
            ow2['epoch'] = 2
            seed_all(41 + i)
            wseg.train(SimpleNamespace(**ow2), loader, None, ow['l1_weight'], ow['l1_dec_step'],
                       ow['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
            fxw['G_final'] = clone_sd(wseg.G)
            fxw['D_final'] = clone_sd(wseg.D)
            out[name] = fxw
            print('interf', name, 'done')
    finally:
        torch.stft = _stft
        torch.Tensor.cuda = _cuda
    torch.save(out, os.path.join(OUT, 'tiny_wsegan_interf.pt'))
    print('tiny_wsegan_interf.pt done')


def _literal_segan_train(ref, o, data_seeds, seed):
    """The reference's literal SEGAN.train over a two-batch loader (its own z / phase-shift draws,
    replayable from `seed`), from a seed-111 init: the sub-fixture the *_train2 tests replay."""
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**o))
    batches = [synth(3, 1024, s_) for s_ in data_seeds]
    loader = [[['u'] * 3, c, n, torch.zeros(3)] for c, n in batches]
    fx = {'batches': batches, 'seed': seed}         # opts / G0 / D0: the parent fixture's (same init)
    g0, d0 = clone_sd(segan.G), clone_sd(segan.D)
    seed_all(seed)
    segan.train(SimpleNamespace(**o), loader, nn.MSELoss(), o['l1_weight'], o['l1_dec_step'],
                o['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
    fx['G_final'] = clone_sd(segan.G)
    fx['D_final'] = clone_sd(segan.D)
    return fx, g0, d0


def make_corners(ref):
    out = {}
    small = dict(genc_fmaps=[4, 8, 16], denc_fmaps=[4, 8, 16], z_dim=16)     # small fixture, like VARIANTS
    # ---- --vanilla_gan through the literal WSEGAN.train (harness patches as for tiny_wsegan2) ----
    _stft = torch.stft
    torch.stft = lambda *a, **k: torch.view_as_real(_stft(*a, return_complex=True, **k))
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ow = tiny_opts()
        ow.update(small)
        ow.update(dict(wsegan=True, misalign_pair=True, vanilla_gan=True, cuda=False, save_freq=1000))
        seed_all(111)
        wseg = ref.WSEGAN(SimpleNamespace(**ow))
        assert wseg.vanilla_gan is True
        c1, n1 = synth(3, 1024, 34)
        names = ['utt_additive_0', 'utt_1', 'utt_additive_2']
        loader = [[names, c1, n1, torch.zeros(3)]]
        fxw = {'opts': ow, 'G0': clone_sd(wseg.G), 'D0': clone_sd(wseg.D), 'clean': c1,
               'noisy': n1, 'names': names, 'seed': 53, 'iters': 2}
        ow2 = dict(ow)
        ow2['epoch'] = 2
        seed_all(53)
        wseg.train(SimpleNamespace(**ow2), loader, None, ow['l1_weight'], ow['l1_dec_step'],
                   ow['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
        fxw['G_final'] = clone_sd(wseg.G)
        fxw['D_final'] = clone_sd(wseg.D)
        out['vanillagan'] = fxw
        print('corners: vanillagan done')
    finally:
        torch.stft = _stft
        torch.Tensor.cuda = _cuda
    # ---- --skip_type constant ----
    o = tiny_opts()
    o.update(small)
    o.update(dict(skip_type='constant', skip_init='randn'))
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**o))
    consts = sorted(k for k in segan.G.state_dict() if k.endswith('skip_k'))
    named = dict(segan.G.named_parameters())
    assert consts and all(not named[k].requires_grad for k in consts)
    # Model.parameters (core.py:196-198) is what build_optimizers hands to the optimizer: no skip_k
    seen = {id(p) for p in segan.G.parameters()}
    assert all(id(named[k]) not in seen for k in consts)
    clean, noisy = synth(3, 1024, 30)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(3, 16, 16, generator=torch.Generator().manual_seed(31))
    fx = {'opts': o, 'G0': clone_sd(segan.G), 'D0': clone_sd(segan.D), 'clean': clean,
          'noisy': noisy, 'z': z, 'roll_seed': 19, 'constants': consts,
          'n_params_G': segan.G.get_n_params(),
          'rolls': ref_harness.ReplayRandom(19).rolls(3, o['phase_shift'], 3)}
    fx.update(manual_step(ref, segan, clean, noisy, z, 19))
    assert all(k not in fx['g_grads'] for k in consts)
    assert all(torch.equal(fx['G_after'][k], fx['G0'][k]) for k in consts)
    fx['train2'], g0, d0 = _literal_segan_train(ref, o, (32, 33), 29)
    assert all(torch.equal(g0[k], fx['G0'][k]) for k in g0) and all(torch.equal(d0[k], fx['D0'][k]) for k in d0)
    assert all(torch.equal(fx['train2']['G_final'][k], fx['G0'][k]) for k in consts)
    out['constantskip'] = fx
    print('corners: constantskip done', consts, fx['n_params_G'])
    # ---- --reg_loss mse_loss ----
    o = tiny_opts()
    o.update(small)
    o['reg_loss'] = 'mse_loss'
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**o))
    assert segan.reg_loss is F.mse_loss
    clean, noisy = synth(3, 1024, 36)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(3, 16, 16, generator=torch.Generator().manual_seed(37))
    fx = {'opts': o, 'G0': clone_sd(segan.G), 'D0': clone_sd(segan.D), 'clean': clean,
          'noisy': noisy, 'z': z, 'roll_seed': 23,
          'rolls': ref_harness.ReplayRandom(23).rolls(3, o['phase_shift'], 3)}
    fx.update(manual_step(ref, segan, clean, noisy, z, 23, reg=segan.reg_loss))
    fx['train2'], g0, d0 = _literal_segan_train(ref, o, (38, 39), 43)
    assert all(torch.equal(g0[k], fx['G0'][k]) for k in g0) and all(torch.equal(d0[k], fx['D0'][k]) for k in d0)
    out['mseloss'] = fx
    print('corners: mseloss done', float(fx['g_l1_loss']))
    torch.save(out, os.path.join(OUT, 'tiny_corners.pt'))
    print('tiny_corners.pt done')


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_harness.import_reference()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    if len(sys.argv) > 1 and sys.argv[1] == 'corners':
        make_corners(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'nobias':
        make_nobias(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'interf':
        make_interf(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'snorm':     # only the spectral-norm fixtures
        make_snorm(ref)
        make_wsegan_snorm(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'vanilla11':
        make_vanilla11(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'variants':
        make_variants(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'gvariants':
        make_gvariants(ref)
        return

    # ---------------- tiny_step ----------------
    o = tiny_opts()
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**o))
    clean, noisy = synth(3, 1024, 0)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(3, 32, 16, generator=torch.Generator().manual_seed(5))
    fx = {'opts': o, 'G0': clone_sd(segan.G), 'D0': clone_sd(segan.D), 'clean': clean,
          'noisy': noisy, 'z': z, 'roll_seed': 7,
          'rolls': ref_harness.ReplayRandom(7).rolls(3, o['phase_shift'], 3)}
    # forward-only extras: hidden activations of G and D in train mode (before the step)
    with torch.no_grad():
        y, hall = segan.G(noisy, z=z, ret_hid=True)
        fx['G_hall'] = {k: v.clone() for k, v in hall.items()}
    d_tmp = ref.Discriminator(2, o['denc_fmaps'], o['gkwidth'], o['denc_poolings'],
                              pool_type='none', pool_slen=16, norm_type='bnorm', phase_shift=5)
    d_tmp.load_state_dict(segan.D.state_dict())
    d_tmp.train()
    random.seed(7)
    with torch.no_grad():
        yd, acts = d_tmp(torch.cat((clean, noisy), 1))
    fx['D_acts'] = {k: v.clone() for k, v in acts.items()}
    fx.update(manual_step(ref, segan, clean, noisy, z, 7))
    torch.save(fx, os.path.join(OUT, 'tiny_step.pt'))
    print('tiny_step.pt', {k: (tuple(v.shape) if torch.is_tensor(v) else type(v).__name__)
                           for k, v in fx.items() if k in ('Genh', 'd_real', 'g_l1_loss')})

    # ---------------- tiny_train2: the literal SEGAN.train ----------------
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**o))
    c1, n1 = synth(3, 1024, 1)
    c2, n2 = synth(3, 1024, 2)
    loader = [[['u'] * 3, c1, n1, torch.zeros(3)], [['u'] * 3, c2, n2, torch.zeros(3)]]
    fx2 = {'opts': o, 'G0': clone_sd(segan.G), 'D0': clone_sd(segan.D),
           'batches': [(c1, n1), (c2, n2)], 'seed': 23}
    seed_all(23)
    opts_ns = SimpleNamespace(**o)
    segan.train(opts_ns, loader, nn.MSELoss(), o['l1_weight'], o['l1_dec_step'],
                o['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
    fx2['G_final'] = clone_sd(segan.G)
    fx2['D_final'] = clone_sd(segan.D)
    torch.save(fx2, os.path.join(OUT, 'tiny_train2.pt'))
    print('tiny_train2.pt done')

    # ---------------- tiny_s2: stride-2 (vanilla SEGAN style) ----------------
    o2 = tiny_opts()
    o2.update(dict(genc_fmaps=[4, 8, 8, 16], denc_fmaps=[4, 8, 8, 16], genc_poolings=[2] * 4,
                   denc_poolings=[2] * 4, z_dim=16, dpool_slen=16, slice_size=256))
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**o2))
    clean, noisy = synth(2, 256, 3)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(2, 16, 16, generator=torch.Generator().manual_seed(6))
    fx3 = {'opts': o2, 'G0': clone_sd(segan.G), 'D0': clone_sd(segan.D), 'clean': clean,
           'noisy': noisy, 'z': z, 'roll_seed': 9,
           'rolls': ref_harness.ReplayRandom(9).rolls(4, o2['phase_shift'], 3)}
    fx3.update(manual_step(ref, segan, clean, noisy, z, 9))
    torch.save(fx3, os.path.join(OUT, 'tiny_s2.pt'))
    print('tiny_s2.pt done')

    # ---------------- tiny_wsegan2: the literal WSEGAN.train (misalign pair) ----------------
    # The reference cannot run as written on torch 2.x / CPU: the
    # legacy torch.stft call needs return_complex, and labels are hard-.cuda()'d.  Both are
    # patched IN THE HARNESS ONLY (the reference sources are untouched): stft returns the
    # legacy real view [..., 2] so torch.norm(x, 2, dim=3) is |STFT| as before.
    ow = tiny_opts()
    ow.update(dict(wsegan=True, misalign_pair=True, cuda=False, save_freq=1000))
    _stft = torch.stft
    torch.stft = lambda *a, **k: torch.view_as_real(_stft(*a, return_complex=True, **k))
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        seed_all(111)
        wseg = ref.WSEGAN(SimpleNamespace(**ow))
        c1, n1 = synth(3, 1024, 4)
        names = ['utt_additive_0', 'utt_1', 'utt_additive_2']
        loader = [[names, c1, n1, torch.zeros(3)]]
        fxw = {'opts': ow, 'G0': clone_sd(wseg.G), 'D0': clone_sd(wseg.D), 'clean': c1,
               'noisy': n1, 'names': names, 'seed': 31, 'iters': 2}
        ow2 = dict(ow)
        ow2['epoch'] = 2          # 2 iterations over the 1-batch loader
        seed_all(31)
        wseg.train(SimpleNamespace(**ow2), loader, None, ow['l1_weight'], ow['l1_dec_step'],
                   ow['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
        fxw['G_final'] = clone_sd(wseg.G)
        fxw['D_final'] = clone_sd(wseg.D)
        torch.save(fxw, os.path.join(OUT, 'tiny_wsegan2.pt'))
        print('tiny_wsegan2.pt done')
    finally:
        torch.stft = _stft
        torch.Tensor.cuda = _cuda

    # ---------------- segan_plus_b2: the default net ----------------
    ob = base_opts()
    ob['save_path'] = '/tmp/segan_golden_ckpt'
    seed_all(111)
    segan = ref.SEGAN(SimpleNamespace(**ob))
    clean, noisy = synth(2, 16384, 0)
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(0))
    fx4 = {'opts': ob, 'seed': 111, 'roll_seed': 3, 'z_seed': 0, 'data_seed': 0,
           'rolls': ref_harness.ReplayRandom(3).rolls(5, ob['phase_shift'], 3),
           'init_G': {k: checksum(v) for k, v in segan.G.state_dict().items()},
           'init_D': {k: checksum(v) for k, v in segan.D.state_dict().items()}}
    res = manual_step(ref, segan, clean, noisy, z, 3)
    for k in ('Genh', 'd_real', 'd_fake', 'd_fake_', 'd_real_loss', 'd_fake_loss', 'g_adv_loss',
              'g_l1_loss'):
        fx4[k] = res[k]
    fx4['d_grads'] = {k: checksum(v) for k, v in res['d_grads'].items()}
    fx4['g_grads'] = {k: checksum(v) for k, v in res['g_grads'].items()}
    fx4['small_d_grads'] = {k: v for k, v in res['d_grads'].items() if v.numel() <= 4096}
    fx4['small_g_grads'] = {k: v for k, v in res['g_grads'].items() if v.numel() <= 4096}
    fx4['G_after'] = {k: checksum(v) for k, v in res['G_after'].items()}
    fx4['D_after'] = {k: checksum(v) for k, v in res['D_after'].items()}
    torch.save(fx4, os.path.join(OUT, 'segan_plus_b2.pt'))
    print('segan_plus_b2.pt done', res['Genh'].shape, res['g_l1_loss'])
    make_snorm(ref)
    make_wsegan_snorm(ref)


if __name__ == '__main__':
    main()
