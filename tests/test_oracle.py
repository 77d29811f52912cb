"""Pin the CPU oracle (oracle/segan_oracle.py) against outputs of the REAL reference
(tests/golden/*.pt, produced by oracle/make_golden.py from /root/reference)."""
import random
from types import SimpleNamespace

import torch

import segan_oracle as O
import pytest

from conftest import VARIANT_NAMES, max_rel, oracle_kwargs

TOL = 2e-5   # fp32 restatement vs fp32 reference modules: same ops, same order


def _check_dict(got, want, tol=TOL, what=''):
    assert set(want.keys()) <= set(got.keys()), (what, set(want) - set(got))
    for k, v in want.items():
        if not torch.is_floating_point(v):
            continue
        err = max_rel(got[k], v)
        assert err < tol, '{} {}: rel err {:.3e}'.format(what, k, err)


def test_tiny_forward_hidden(tiny_step):
    fx = tiny_step
    st = fx['opts']['genc_poolings']
    y, hall = O.generator_forward(fx['G0'], fx['noisy'], fx['z'], st, ret_hid=True)
    _check_dict(hall, fx['G_hall'], what='G hall')
    d0 = {k: v.clone() for k, v in fx['D0'].items()}
    yd, acts = O.discriminator_forward(d0, torch.cat((fx['clean'], fx['noisy']), 1),
                                       fx['rolls'][0], st, ret_act=True)
    _check_dict(acts, fx['D_acts'], what='D acts')


def _check_step(fx, tol=TOL, step_tol=2e-6):
    st = fx['opts']['genc_poolings']
    res = O.gan_step(fx['G0'], fx['D0'], fx['clean'], fx['noisy'], fx['z'], fx['rolls'], st,
                     l1_weight=100.0, lr=5e-5, **oracle_kwargs(fx['opts']))
    for k in ('Genh', 'd_real', 'd_fake', 'd_fake_', 'd_real_loss', 'd_fake_loss', 'g_adv_loss',
              'g_l1_loss'):
        assert max_rel(res[k], fx[k]) < tol, k
    _check_dict(res['d_grads'], fx['d_grads'], tol=1e-4, what='d_grads')
    _check_dict(res['g_grads'], fx['g_grads'], tol=1e-4, what='g_grads')
    # RMSprop normalises the gradient, so roundoff-level gradients (conv biases in front of
    # BatchNorm are mathematically zero) move by +-lr: compare with an absolute tolerance of
    # a fraction of one lr step.
    for name, got, want in (('G', res['G'], fx['G_after']), ('D', res['D'], fx['D_after'])):
        for k, v in want.items():
            if not torch.is_floating_point(v):
                continue
            if name == 'D' and k.endswith('conv.bias'):
                continue
            err = (got[k] - v).abs().max().item()
            assert err < step_tol, '{} after-step {}: abs err {:.3e}'.format(name, k, err)


def test_tiny_step(tiny_step):
    _check_step(tiny_step)


def test_tiny_stride2_step(tiny_s2):
    _check_step(tiny_s2)


@pytest.mark.parametrize('name', VARIANT_NAMES)
def test_tiny_architecture_variants(tiny_variants, name):
    """--skip_type conv, pooling-1 layers, a conv block as last decoder layer, --dpool_type
    conv / gmax / gavg: one step of the real reference each (oracle/make_golden.py variants).
    The small heads leave more of the first RMSprop step in the ill-conditioned regime."""
    _check_step(tiny_variants[name], step_tol=1e-5)


def test_tiny_spectral_norm_step(tiny_snorm):
    """--dnorm_type snorm: spectral norm on D's convs, fc[0], fc[2] and the PReLU fc[3]; the u/v
    buffers after the three D forwards are part of D_after.  Without BatchNorm the adversarial
    gradient reaching G is small, so more of G's first RMSprop step sits in the
    ill-conditioned |g| ~ 1e-7 regime: weights are compared to 20 % of a step."""
    _check_step(tiny_snorm, step_tol=1e-5)


def test_generator_with_spectral_norm(tiny_snorm):
    """Generator(norm_type='snorm'): conv (dim 0) and transposed-conv (dim 1) weights."""
    g = tiny_snorm['gsn']
    sd = O._leafs(g['G0'])
    y = O.generator_forward(sd, g['x'], g['z'], [4, 4, 4])
    assert max_rel(y, g['y']) < TOL
    keys = [k for k in sd if sd[k].requires_grad]
    grads = torch.autograd.grad((y * g['c']).sum(), [sd[k] for k in keys])
    for k, gr in zip(keys, grads):
        assert max_rel(gr, g['grads'][k]) < 1e-4, k
    for k, v in g['G_after_fwd'].items():            # u / v after the power iteration
        assert max_rel(sd[k], v) < TOL, k


GVARIANT_NAMES = ('bnorm_concat', 'bnorm_sum', 'dropout_alpha', 'dropout_conv_sum', 'bnorm_dropout')


@pytest.mark.parametrize('name', GVARIANT_NAMES)
def test_generator_options_no_flag_reaches(name):
    """Generator(norm_type='bnorm') and skip_dropout (generator.py:53-54,126,166-176): output,
    gradients, BatchNorm buffers after the forward and the eval-mode output of the REAL reference
    (oracle/make_golden.py gvariants); dropout masks replayed from the recorded torch seed."""
    from conftest import load_golden
    g = load_golden('tiny_gvariants.pt')[name]
    kw = g['kwargs']
    sd = O._leafs(g['G0'])
    torch.manual_seed(g['fwd_seed'])
    y = O.generator_forward(sd, g['x'], g['z'], [4, 4, 4], skip_merge=kw['skip_merge'],
                            skip_dropout=kw.get('skip_dropout', 0.0))
    assert max_rel(y, g['y']) < TOL
    keys = [k for k in sd if sd[k].requires_grad]
    grads = torch.autograd.grad((y * g['c']).sum(), [sd[k] for k in keys])
    for k, gr in zip(keys, grads):
        assert max_rel(gr, g['grads'][k]) < 1e-4, k
    for k, v in g['G_after_fwd'].items():            # running statistics after one forward
        if torch.is_floating_point(v):
            assert max_rel(sd[k], v) < TOL, k
    with torch.no_grad():
        ye = O.generator_forward(sd, g['x'], g['z'], [4, 4, 4], training=False,
                                 skip_merge=kw['skip_merge'],
                                 skip_dropout=kw.get('skip_dropout', 0.0))
    assert max_rel(ye, g['y_eval']) < TOL


def test_tiny_literal_train_replay(tiny_train2):
    """Replay the reference's literal SEGAN.train (two batches): z comes from the global
    torch RNG (generator.py:197), the phase shifts from python's random
    (discriminator.py:159-163)."""
    _replay_literal_train(tiny_train2, tiny_train2['opts'], tiny_train2['G0'], tiny_train2['D0'])


def _replay_literal_train(fx, o, G, D, **kw):
    st = o['genc_poolings']
    random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    g_sq = d_sq = None
    for clean, noisy in fx['batches']:
        clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
        z = torch.randn(clean.size(0), o['z_dim'], 16)
        rolls = []
        for _ in range(3):
            r = []
            for _ in st:
                s = random.randint(1, o['phase_shift'])
                r.append(s if random.random() > 0.5 else -s)
            rolls.append(r)
        res = O.gan_step(G, D, clean, noisy, z, rolls, st, l1_weight=o['l1_weight'],
                         lr=o['g_lr'], g_sq=g_sq, d_sq=d_sq, **kw)
        G, D, g_sq, d_sq = res['G'], res['D'], res['g_sq'], res['d_sq']
    for k, v in fx['G_final'].items():
        assert (G[k] - v).abs().max().item() < 4e-6, k
    for k, v in fx['D_final'].items():
        if not torch.is_floating_point(v) or k.endswith('conv.bias'):
            continue
        assert (D[k] - v).abs().max().item() < 4e-6, k


def _chk(t, c, tol):
    t = t.detach().double().reshape(-1)
    assert t.numel() == c['n']
    scale = max(c['abs'], 1e-30)
    assert abs(t.sum().item() - c['sum']) / scale < tol
    assert abs(t.abs().sum().item() - c['abs']) / scale < tol
    got = t[c['sample_idx']].float()
    den = max(c['sample'].abs().max().item(), 1e-30)
    assert (got - c['sample']).abs().max().item() / den < max(tol, 1e-5)


def _default_net_init_and_step(fx):
    import random as pyrandom
    import numpy as np
    from segan_pytorch_amd.models import SEGAN
    from segan_pytorch_amd.datasets import synthetic_pairs
    pyrandom.seed(fx['seed'])
    np.random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    m = SEGAN(SimpleNamespace(**fx['opts']))
    gsd, dsd = m.G.state_dict(), m.D.state_dict()
    assert list(gsd.keys()) == list(fx['init_G'].keys())
    assert list(dsd.keys()) == list(fx['init_D'].keys())
    for k, c in fx['init_G'].items():
        _chk(gsd[k], c, 1e-12)
    for k, c in fx['init_D'].items():
        if torch.is_floating_point(dsd[k]):
            _chk(dsd[k], c, 1e-12)
    clean, noisy = synthetic_pairs(2, 16384, fx['data_seed'])
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(2, 1024, 16, generator=torch.Generator().manual_seed(fx['z_seed']))
    st = fx['opts']['genc_poolings']
    res = O.gan_step(gsd, dsd, clean, noisy, z, fx['rolls'], st, 100.0, 5e-5)
    for k in ('Genh', 'd_real', 'd_fake', 'd_fake_', 'g_l1_loss', 'g_adv_loss'):
        assert max_rel(res[k], fx[k]) < TOL, k
    for k, c in fx['d_grads'].items():
        if k.endswith('conv.bias'):
            continue
        _chk(res['d_grads'][k], c, 1e-4)
    for k, c in fx['g_grads'].items():
        _chk(res['g_grads'][k], c, 1e-4)
    return gsd


def test_default_net_init_and_step(segan_plus_b2):
    """The default SEGAN+ net: OUR constructors under seed 111 must reproduce the
    reference's initial weights, and the oracle its outputs/gradients at B=2."""
    _default_net_init_and_step(segan_plus_b2)


def test_default_net_no_bias_init_and_step():
    """--no_bias, the reference's own batch-300 recipe (run_segan+_train.sh:7, train.py:248): G's
    convs have no bias (the transposed convs keep theirs, modules.py:116-119), D is unchanged;
    same checks as the default net at B=2 (oracle/make_golden.py nobias)."""
    from conftest import load_golden
    gsd = _default_net_init_and_step(load_golden('segan_plus_nobias_b2.pt'))
    assert not any(k.endswith('.conv.bias') for k in gsd)
    assert sum(k.endswith('deconv.bias') for k in gsd) == 5


def test_tiny_no_bias_step():
    from conftest import load_golden
    fx = load_golden('tiny_nobias.pt')
    assert not any(k.endswith('.conv.bias') for k in fx['G0'])
    _check_step(fx)


def test_vanilla11_net_init_and_step(vanilla11_b8):
    """The original 11-layer stride-2 SEGAN shape (train.py:199-205 flags): our constructors
    reproduce the reference's seed-111 initial weights, the oracle its step at B=8."""
    import random as pyrandom
    import numpy as np
    from segan_pytorch_amd.models import SEGAN
    from segan_pytorch_amd.datasets import synthetic_pairs
    fx = vanilla11_b8
    pyrandom.seed(fx['seed'])
    np.random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    m = SEGAN(SimpleNamespace(**fx['opts']))
    gsd, dsd = m.G.state_dict(), m.D.state_dict()
    assert list(gsd.keys()) == list(fx['init_G'].keys())
    assert list(dsd.keys()) == list(fx['init_D'].keys())
    for k, c in fx['init_G'].items():
        _chk(gsd[k], c, 1e-12)
    for k, c in fx['init_D'].items():
        if torch.is_floating_point(dsd[k]):
            _chk(dsd[k], c, 1e-12)
    clean, noisy = synthetic_pairs(fx['batch'], 16384, fx['data_seed'])
    clean, noisy = clean.unsqueeze(1), noisy.unsqueeze(1)
    z = torch.randn(fx['batch'], 1024, 8, generator=torch.Generator().manual_seed(fx['z_seed']))
    st = fx['opts']['genc_poolings']
    res = O.gan_step(gsd, dsd, clean, noisy, z, fx['rolls'], st, 100.0, 5e-5)
    for k in ('Genh', 'd_real', 'd_fake', 'd_fake_', 'g_l1_loss', 'g_adv_loss'):
        # d_fake_ / g_adv go through D after its first RMSprop step, which is ill-conditioned
        # where |g| is at roundoff level (thread-count dependent summation order on the CPU)
        assert max_rel(res[k], fx[k]) < (2e-4 if k in ('d_fake_', 'g_adv_loss') else TOL), k
    for k, c in fx['d_grads'].items():
        if k.endswith('conv.bias'):
            continue
        _chk(res['d_grads'][k], c, 1e-4)
    # generator-phase gradients pass through D after its first, ill-conditioned RMSprop step:
    # for this 11-BatchNorm-layer discriminator two CPU runs of the SAME arithmetic (the
    # recorded reference run and this one) already differ by 1e-2..4e-2 there
    for k, c in fx['g_grads'].items():
        _chk(res['g_grads'][k], c, 1e-1)


def test_wsegan_literal_train_replay(tiny_wsegan2):
    """The oracle's WSEGAN step replayed against the reference's literal WSEGAN.train
    (--misalign_pair, two iterations)."""
    _replay_wsegan_train(tiny_wsegan2)


def test_wsegan_vanilla_gan_literal_train_replay(tiny_corners):
    """--vanilla_gan (model.py:582-585: binary_cross_entropy_with_logits for every adversarial
    term) through the reference's literal WSEGAN.train, two iterations."""
    assert tiny_corners['vanillagan']['opts']['vanilla_gan'] is True
    _replay_wsegan_train(tiny_corners['vanillagan'], vanilla_gan=True)


def test_constant_skip_step_and_literal_train(tiny_corners):
    """--skip_type constant (generator.py:25,40-41,59): the per-channel skip scale is a fixed
    (here randn-initialised) constant — no gradient, not in the optimizer (core.py:196-198), unchanged
    by the step and by two batches of the literal SEGAN.train."""
    fx = tiny_corners['constantskip']
    kw = oracle_kwargs(fx['opts'])
    assert kw['frozen'] == tuple(fx['constants'])
    _check_step(fx)
    st = fx['opts']['genc_poolings']
    res = O.gan_step(fx['G0'], fx['D0'], fx['clean'], fx['noisy'], fx['z'], fx['rolls'], st, **kw)
    for k in fx['constants']:
        assert k not in res['g_grads'] and torch.equal(res['G'][k], fx['G0'][k])
        assert float(fx['G0'][k].std()) > 0.1           # a non-trivial constant
    _replay_literal_train(fx['train2'], fx['opts'], fx['G0'], fx['D0'], frozen=kw['frozen'])


def test_mse_reg_loss_step_and_literal_train(tiny_corners):
    """--reg_loss mse_loss (train.py:179; model.py:79 getattr(F, opts.reg_loss)): one step and the
    literal two-batch loop."""
    fx = tiny_corners['mseloss']
    assert oracle_kwargs(fx['opts'])['reg_loss'] == 'mse_loss'
    _check_step(fx)
    _replay_literal_train(fx['train2'], fx['opts'], fx['G0'], fx['D0'], reg_loss='mse_loss')


def _replay_wsegan_train(fx, **kw):
    from conftest import draw_rolls
    o = fx['opts']
    st = o['genc_poolings']
    random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    clean, noisy = fx['clean'].unsqueeze(1), fx['noisy'].unsqueeze(1)
    G, D, g_sq, d_sq = fx['G0'], fx['D0'], None, None
    for _ in range(fx['iters']):
        r0 = draw_rolls(len(st), o['phase_shift'])
        z = torch.randn(clean.size(0), o['z_dim'], 16)
        r1 = draw_rolls(len(st), o['phase_shift'])
        perm = list(range(clean.size(0)))
        random.shuffle(perm)
        r2 = draw_rolls(len(st), o['phase_shift'])
        r3 = draw_rolls(len(st), o['phase_shift'])
        res = O.wsegan_step(G, D, clean, noisy, z, [r0, r1, r2, r3], perm, fx['names'], st,
                            l1_weight=o['l1_weight'], pow_weight=o['pow_weight'], lr=o['g_lr'],
                            n_fft=o['n_fft'], g_sq=g_sq, d_sq=d_sq, **kw)
        G, D, g_sq, d_sq = res['G'], res['D'], res['g_sq'], res['d_sq']
    for k, v in fx['G_final'].items():
        assert (G[k] - v).abs().max().item() < 5e-5, k   # 10 % of an RMSprop step
    for k, v in fx['D_final'].items():
        if not torch.is_floating_point(v) or k.endswith(('conv.bias', 'norm.running_mean')):
            continue
        assert (D[k] - v).abs().max().item() < 5e-5, k


@pytest.mark.parametrize('flavour', ['both', 'interf_only'])
def test_wsegan_interf_pair_literal_train_replay(flavour):
    """--interf_pair (model.py:606-628), with and without --misalign_pair: the oracle's step
    replayed against the reference's literal WSEGAN.train, two iterations; the python `random`
    stream interleaves phase shifts, the misalign shuffle and the per-sample (frequency,
    amplitude) choices exactly as the reference draws them."""
    from conftest import draw_rolls, load_golden
    fx = load_golden('tiny_wsegan_interf.pt')[flavour]
    o = fx['opts']
    st = o['genc_poolings']
    random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    clean, noisy = fx['clean'].unsqueeze(1), fx['noisy'].unsqueeze(1)
    G, D, g_sq, d_sq = fx['G0'], fx['D0'], None, None
    for _ in range(fx['iters']):
        rolls = [draw_rolls(len(st), o['phase_shift'])]
        z = torch.randn(clean.size(0), o['z_dim'], 16)
        rolls.append(draw_rolls(len(st), o['phase_shift']))
        perm = None
        if o['misalign_pair']:
            perm = list(range(clean.size(0)))
            random.shuffle(perm)
            rolls.append(draw_rolls(len(st), o['phase_shift']))
        squares = O.interf_squares(clean.size(0), clean.size(-1))
        rolls.append(draw_rolls(len(st), o['phase_shift']))
        rolls.append(draw_rolls(len(st), o['phase_shift']))
        res = O.wsegan_step(G, D, clean, noisy, z, rolls, perm, fx['names'], st,
                            l1_weight=o['l1_weight'], pow_weight=o['pow_weight'], lr=o['g_lr'],
                            n_fft=o['n_fft'], g_sq=g_sq, d_sq=d_sq, squares=squares)
        G, D, g_sq, d_sq = res['G'], res['D'], res['g_sq'], res['d_sq']
    for k, v in fx['G_final'].items():
        assert (G[k] - v).abs().max().item() < 5e-5, k
    for k, v in fx['D_final'].items():
        if not torch.is_floating_point(v) or k.endswith(('conv.bias', 'norm.running_mean')):
            continue
        assert (D[k] - v).abs().max().item() < 5e-5, k


def test_gate_hook_with_own_gates_is_the_identity(tiny_step):
    """oracle `gates=` (test hook of the gate-aligned gradient comparisons, segan_oracle._prelu):
    imposing the sides the oracle takes by itself changes neither outputs nor gradients — the
    hook alters the PReLU side of an element and nothing else — and imposing a flipped side on
    one element changes the gradients (the hook is live)."""
    fx = tiny_step
    st = fx['opts']['genc_poolings']

    def run(gg=None, gd=None):
        G = {k: v.clone().requires_grad_(True) for k, v in fx['G0'].items()}
        D = {k: (v.clone().requires_grad_(True) if k.split('.')[-1] not in O._BUFFERS else v.clone())
             for k, v in fx['D0'].items()}
        y, hall = O.generator_forward(G, fx['noisy'], fx['z'], st, ret_hid=True, gates=gg)
        d, acts = O.discriminator_forward(D, torch.cat((y, fx['noisy']), 1), fx['rolls'][2], st,
                                          ret_act=True, gates=gd)
        loss = (d.view(-1) - 1).pow(2).mean() + 100.0 * (y - fx['clean']).abs().mean()
        ps = [v for v in list(G.values()) + list(D.values()) if v.requires_grad]
        return y, d, hall, acts, torch.autograd.grad(loss, ps, allow_unused=True)

    y, d, hall, acts, gr = run()
    gg = {k: v > 0 for k, v in hall.items() if k != 'enc_zc' and k != 'dec_{}'.format(len(st) - 1)}
    gd = {k: acts['a_' + k[2:]] > 0 for k in acts if k.startswith('h_')}
    gd['fc_1'], gd['fc_3'] = acts['fc_a1'] > 0, acts['fc_a3'] > 0
    y2, d2, _, _, gr2 = run(gg, gd)
    assert torch.equal(y, y2) and torch.equal(d, d2)
    for a, b in zip(gr, gr2):
        assert (a is None and b is None) or torch.equal(a, b)
    gd['h_1'] = gd['h_1'].clone()
    gd['h_1'][0, 0, 0] = ~gd['h_1'][0, 0, 0]
    _, _, _, _, gr3 = run(gg, gd)
    assert any(a is not None and not torch.equal(a, b) for a, b in zip(gr, gr3))
