"""BASELINE config 5 (bf16 MFMA mixed precision) and config 4 (WSEGAN --misalign_pair) at the
parity bar of the fp32 path: gradients, not only forwards (round-4 review, "Next round" item 1).

The re-stated tolerance for bf16 GRADIENTS:

* A bf16 contraction rounds both operands to 8 significand bits; a pre-activation of a
  BatchNorm-normalised (unit-scale) layer therefore differs from its fp64 value by ~2e-3..1e-2,
  and every pre-activation closer to zero than that takes the other PReLU side.  At the initial
  PReLU slope of 0 (model.py:28-43: the gates are ReLUs) that is a share of ~1e-3..1e-2 of ALL
  gates, each of which switches the gradient through its unit on or off: the free-running
  gradient distance to fp64 is sqrt(share of flipped gates, compounded over the layers) ~ 0.1,
  for ANY bf16 evaluation of this non-smooth function.  It is not a property of the kernels.
* What IS a property of the kernels is the distance with the derivative discontinuity removed:
  the fp64 oracle evaluated with the GPU's own gate sides (oracle `gates=`).  That figure is
  pure operand rounding and is asserted here: see BF16_ALIGNED_TOL.
* And what a user of config 5 needs is that training on those gradients follows the fp32
  trajectory: a 20-step run from identical state, losses and weights compared step by step,
  with the bf16x3 mode as the control for the trajectory's own sensitivity to rounding.
"""
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import segan_oracle as O
from conftest import max_rel

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# ---- the re-stated bf16 gradient tolerances (relative L2 per tensor, worst tensor) -----------
# measured on MI355X (round 5, gpurun_out -> DESIGN.md section 6): D at batch 300, slopes init /
# trained: 5.7e-4 / 5.2e-4 of the gates flipped (per layer 8e-4 .. 2.3e-3 from the second conv on),
# largest |a| at a flip 0.025 / 0.021, logits 8.2e-3 / 6.1e-3, gradients aligned 7.4e-3 / 6.3e-3
# (worst tensor enc_blocks.4.conv.weight / enc_blocks.3.act.weight) against 0.14 / 0.11
# free-running; generator phase at batch 16: aligned 1.15e-2 (enc_blocks.4.act.weight) against
# 0.18 free-running.
BF16_FLIP_SHARE = 2e-3        # share of PReLU gates that may differ from the fp64 run's sides
BF16_FLIP_ABS = 0.08          # ... every one of them a unit-scale value this close to zero
BF16_ALIGNED_TOL = 2e-2       # gradient distance to fp64 with the GPU's gate sides imposed
BF16_LOGITS_TOL = 2e-2        # D's logits (max-abs relative to the largest logit)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def l2_rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


# 'trained' (PReLU slopes 0.05..0.3; measured 5.2e-4 flips, 6.3e-3 aligned) costs another two fp64
# oracle evaluations at batch 300 (~40 s of host time): run with SEGAN_TEST_FULL=1
@pytest.mark.parametrize('slopes', ['init'] + (['trained'] if os.environ.get('SEGAN_TEST_FULL') == '1' else []))
def test_discriminator_bf16_gradients_with_aligned_gates(slopes):
    """test_discriminator_gradients_with_aligned_gates with ops.set_precision('bf16'): one D
    forward + backward at batch 300 (BatchNorm over 300 x L) on the bf16 matrix cores against the
    fp64 oracle.  Counted: the gates whose side differs (bounded share, all near zero).  Asserted:
    with the GPU's sides imposed every parameter gradient is within BF16_ALIGNED_TOL relative L2
    — the worst tensor is named in the failure message and printed."""
    from test_gpu_kernels import discriminator_aligned_gates_run
    r = discriminator_aligned_gates_run(slopes, 'bf16')
    assert r['logits_max_rel'] < BF16_LOGITS_TOL
    assert r['flip_share'] < BF16_FLIP_SHARE, r['flip_share_per_layer']
    assert r['worst_abs_a_at_a_flip'] < BF16_FLIP_ABS
    assert r['zero_grads_ok']
    assert r['aligned'] < BF16_ALIGNED_TOL, (r['aligned_worst_tensor'], r['aligned'])
    # the free-running distance is the gates: it must be explained, i.e. much larger than the
    # aligned one only where gates are ReLU-like (slope 0) — recorded, bounded loosely
    assert r['free_running'] < 0.5


def _default_model(B, seed=111):
    bench = _bench()
    from segan_pytorch_amd.models import SEGAN
    from segan_pytorch_amd.datasets import synthetic_pairs
    opts = bench.default_opts()
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    m = SEGAN(SimpleNamespace(**opts))
    gsd0 = {k: v.detach().clone() for k, v in m.G.state_dict().items()}
    dsd0 = {k: v.detach().clone() for k, v in m.D.state_dict().items()}
    clean, noisy = synthetic_pairs(B, 16384, 0)
    return opts, m, gsd0, dsd0, clean.unsqueeze(1), noisy.unsqueeze(1)


def test_generator_phase_bf16_gradients_with_aligned_gates():
    """The generator phase (model.py:310-321: D on (G(noisy), noisy), adversarial + 100 * L1,
    backward into G) of the default SEGAN+ net at batch 16 on the bf16 matrix cores against the
    fp64 oracle with the GPU's PReLU sides imposed on BOTH networks: every generator gradient
    within BF16_ALIGNED_TOL relative L2.  Free-running figures are printed."""
    from segan_pytorch_amd import losses, ops
    from test_gpu_kernels import gpu_discriminator_gates
    B = 16
    opts, m, gsd0, dsd0, clean, noisy = _default_model(B)
    st = opts['genc_poolings']
    z = torch.randn(B, 1024, 16, generator=torch.Generator().manual_seed(3))
    rolls = [2, -5, 1, -1, 4]
    m = m.to(DEV)
    m.G.train(); m.D.train()
    Gopt, Dopt = m.build_optimizers(SimpleNamespace(**opts))
    m.D.draw_rolls = lambda: list(rolls)
    cg, ng, zg = clean.to(DEV), noisy.to(DEV), z.to(DEV)
    old, oldp = ops.get_deterministic(), ops.get_precision()
    ops.set_deterministic(True)
    ops.set_precision('bf16')
    try:
        with torch.no_grad():
            _, hall = m.G(ng, z=zg, ret_hid=True)
        n_dec = len(m.G.dec_blocks)
        gg = {k: (v > 0).cpu() for k, v in hall.items()
              if k != 'enc_zc' and k != 'dec_{}'.format(n_dec - 1)}
        Genh = m.infer_G(ng, cg, z=zg)
        # gradients are read after g_phase's optimizer step (the step does not touch .grad)
        g_adv, g_l1 = m.g_phase(Genh, cg, ng, Gopt, losses.MSELoss(), 100.0)
        torch.cuda.synchronize()
    finally:
        ops.set_deterministic(old)
        ops.set_precision(oldp)
    gd = gpu_discriminator_gates(m.D)
    gn = dict(m.G.named_parameters())

    def oracle64(g_gates, d_gates):
        G64 = {k: v.double().requires_grad_(True) for k, v in gsd0.items()}
        D64 = {k: (v.double() if torch.is_floating_point(v) else v.clone()) for k, v in dsd0.items()}
        genh = O.generator_forward(G64, noisy.double(), z.double(), st, gates=g_gates)
        d = O.discriminator_forward(D64, torch.cat((genh, noisy.double()), 1), rolls, st, gates=d_gates)
        adv = F.mse_loss(d.view(-1), torch.ones(B, dtype=torch.float64))
        l1 = 100.0 * F.l1_loss(genh, clean.double())
        keys = list(G64.keys())
        return genh, adv, l1, dict(zip(keys, torch.autograd.grad(adv + l1, [G64[k] for k in keys])))

    genh_a, adv_a, l1_a, ga = oracle64(gg, gd)
    _, _, _, gf = oracle64(None, None)
    free = max(l2_rel(gn[k].grad, g) for k, g in gf.items())
    worst, worst_k = 0.0, None
    for k, g in ga.items():
        e = l2_rel(gn[k].grad, g)
        if e > worst:
            worst, worst_k = e, k
    mse = ((Genh.detach().cpu().double() - genh_a) ** 2).mean().item()
    print(dict(batch=B, g_mse=mse, g_adv_rel=max_rel(g_adv, adv_a), g_l1_rel=max_rel(g_l1, l1_a),
               free_running=free, aligned=worst, aligned_worst_tensor=worst_k))
    assert mse < 1e-4                       # the north-star bar, bf16
    assert max_rel(g_l1, l1_a) < 5e-3 and max_rel(g_adv, adv_a) < 5e-2
    assert worst < BF16_ALIGNED_TOL, (worst_k, worst)


def test_bf16_trajectory_tracks_fp32_over_20_steps():
    """Config 5 as it is USED (model.py:298-321 trains on these gradients): 20 full GAN steps of
    the default SEGAN+ net at batch 32 from identical weights, data, z and phase shifts in exact
    fp32, with the bf16x3 contractions (fp32-class rounding in a different summation order: the
    CONTROL for how far two correct runs of this GAN drift apart on their own) and with the bf16
    contractions.

    The first steps of this training are violent by construction — RMSprop's first step moves
    every weight by +-10 lr, D's losses jump to ~500 at step 1 and oscillate between 0.1 and 3
    afterwards — and the trajectory amplifies rounding: at batch 8 even the CONTROL is off by
    factors of 2-3 on single adversarial losses (tests/diag/diag_bf16_trajectory.py).  At batch 32
    the control tracks fp32 closely, which makes a band meaningful.  Measured (round 5):

                         per-step ratio to fp32, worst of 20     geometric mean over 20 steps
                         d_real  d_fake  g_adv   g_l1            d_real  d_fake  g_adv   g_l1
      bf16x3 (control)   1.10    1.06    1.07    1.002           1.003   1.004   1.005   1.0005
      bf16               2.10    1.54    2.13    1.002           1.035   0.984   1.029   0.9987

      total weight update after 20 steps against fp32's: cosine G 0.73 / D 0.885 (control 0.88 /
      0.96 — RMSprop's early steps are sign-like, so elements whose gradient is at the noise
      level step at random), length within 0.5 %.

    Stated band (asserted): bf16 — the L1 term (the quantity the generator is trained on, 100 x)
    within 1 % at EVERY step; every adversarial / discriminator loss within a factor 4 of fp32's
    and their geometric mean over the 20 steps within 15 %; update cosine G > 0.5, D > 0.7, length
    within 2 %.  Control — factor 1.5, mean within 3 %, cosine G > 0.7, D > 0.9."""
    import math
    from segan_pytorch_amd import losses, ops
    from segan_pytorch_amd.models import SEGAN
    B, STEPS = 32, 20
    opts, m0, gsd0, dsd0, clean, noisy = _default_model(B)
    del m0
    zs = [torch.randn(B, 1024, 16, generator=torch.Generator().manual_seed(100 + i)) for i in range(STEPS)]
    rng = random.Random(5)
    rolls = [[[rng.randint(1, 5) * (1 if rng.random() > 0.5 else -1) for _ in range(5)] for _ in range(3)]
             for _ in range(STEPS)]

    def run(prec):
        oldp, oldd = ops.get_precision(), ops.get_deterministic()
        ops.set_precision(prec)
        ops.set_deterministic(True)
        try:
            m = SEGAN(SimpleNamespace(**opts))
            m.G.load_state_dict(gsd0)
            m.D.load_state_dict(dsd0)
            m = m.to(DEV)
            Gopt, Dopt = m.build_optimizers(SimpleNamespace(**opts))
            m.G.train(); m.D.train()
            flat = iter([r for step in rolls for r in step])
            m.D.draw_rolls = lambda: list(next(flat))
            cg, ng = clean.to(DEV), noisy.to(DEV)
            log = []
            for i in range(STEPS):
                out = m.gan_step(cg, ng, Gopt, Dopt, losses.MSELoss(), 100.0, z=zs[i].to(DEV))
                log.append([float(v) for v in out])
            torch.cuda.synchronize()
            g = torch.cat([(v.detach().cpu().double() - gsd0[k].double()).flatten()
                           for k, v in m.G.state_dict().items()])
            d = torch.cat([(v.detach().cpu().double() - dsd0[k].double()).flatten()
                           for k, v in m.D.state_dict().items()
                           if torch.is_floating_point(v) and k.split('.')[-1] not in O._BUFFERS])
            return log, g, d
        finally:
            ops.set_precision(oldp)
            ops.set_deterministic(oldd)

    ref = run('fp32')
    bands = {'bf16x3': dict(step=1.5, mean=0.03, l1=0.005, cos_g=0.7, cos_d=0.9, norm=0.02),
             'bf16': dict(step=4.0, mean=0.15, l1=0.01, cos_g=0.5, cos_d=0.7, norm=0.02)}
    for prec, bd in bands.items():
        log, g, d = run(prec)
        assert all(math.isfinite(v) and v > 0 for row in log for v in row)
        worst, mean = [0.0] * 4, [0.0] * 4
        for a, b in zip(ref[0], log):
            for j in range(4):
                r = math.log(b[j] / a[j])
                worst[j] = max(worst[j], abs(r))
                mean[j] += r / STEPS
        fig = dict(precision=prec, step_ratio_worst=[round(math.exp(w), 4) for w in worst],
                   geo_mean_ratio=[round(math.exp(v), 4) for v in mean],
                   G_update_cosine=(torch.dot(ref[1], g) / (ref[1].norm() * g.norm())).item(),
                   D_update_cosine=(torch.dot(ref[2], d) / (ref[2].norm() * d.norm())).item(),
                   G_update_norm_ratio=(g.norm() / ref[1].norm()).item(),
                   D_update_norm_ratio=(d.norm() / ref[2].norm()).item())
        print(fig)
        for j in range(3):
            assert math.exp(worst[j]) < bd['step'], (prec, j, fig)
            assert abs(math.exp(mean[j]) - 1) < bd['mean'], (prec, j, fig)
        assert math.exp(worst[3]) - 1 < bd['l1'], (prec, fig)
        assert fig['G_update_cosine'] > bd['cos_g'] and fig['D_update_cosine'] > bd['cos_d'], fig
        assert abs(fig['G_update_norm_ratio'] - 1) < bd['norm'] and abs(fig['D_update_norm_ratio'] - 1) < bd['norm'], fig


def test_wsegan_step_at_batch_300_matches_the_oracle():
    """BASELINE config 4 at its benchmarked batch inside `-m gpu` (round-4 review, weak point 2:
    this comparison used to exist only as the builder-run `bench.py --wsegan` side line): one
    WSEGAN step (--misalign_pair: three D forwards under one backward, LSGAN cost, STFT log-power
    L1 + masked L1; model.py:572-669) of the default net at batch 300 in exact fp32 against one
    step of the CPU oracle from the same weights / inputs / z / phase shifts / misalign
    permutation — bench.wsegan_parity with its figures asserted.  One oracle WSEGAN step at
    batch 300 costs ~1-2 minutes of host time."""
    bench = _bench()
    opts = bench.default_opts()
    opts.update(dict(misalign_pair=True, interf_pair=False, pow_weight=0.001, vanilla_gan=False,
                     n_fft=2048))          # bench.py --wsegan (run_wsegan_train.sh's loss flags)
    base, par = bench.wsegan_parity(opts, 300, torch.device(DEV))
    print(base, par)
    for mode in ('fp32_deterministic', 'fp32_default'):
        p = par[mode]
        assert 'error' not in p, p
        assert p['batch'] == 300
        assert p['g_mse'] < 1e-12 and p['g_max_abs'] < 1e-5, p
        for k in ('d_loss_rel', 'g_adv_loss_rel', 'pow_loss_rel', 'den_loss_rel'):
            assert p[k] < 2e-5, (mode, k, p[k])
        assert p['d_grad_rel_l2_worst_tensor'] < 6e-3 and p['g_grad_rel_l2_worst_tensor'] < 6e-3, p
