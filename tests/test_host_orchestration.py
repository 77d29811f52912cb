"""Host logic end to end on CPU: the autograd nodes, the flat-arena optimizers, the GAN
step and SEGAN.train, run with the TEST-ONLY CPU emulation of the kernel entry points
(tests/emu_ops.py) against the golden outputs of the real reference.  The HIP kernels
themselves are checked on the GPU box by tests/test_gpu_*.py."""
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import emu_ops
from conftest import GVARIANT_NAMES, VARIANT_NAMES, check_gvariant, max_rel

# RMSprop's first step moves every weight by lr*g/(0.1|g|+1e-8) = +-10*lr = 5e-4 wherever
# |g| >> 1e-7 and is ill-conditioned where the gradient is at roundoff level (|g| ~ 1e-8):
# weights after a step are compared to 10 % of a full step.
STEP_TOL = 5e-5
# A conv bias in front of BatchNorm has a mathematically zero gradient; what autograd returns
# is roundoff noise that RMSprop normalises into +-10*lr random steps, so that bias (which
# BatchNorm cancels exactly) and the running_mean that tracks it are implementation noise.
NOISE_KEYS = ('conv.bias', 'norm.running_mean')


def assert_weights_after_step(sd, after, grads=None, skip=()):
    """Weights after RMSprop step(s) against the reference's.  The first RMSprop steps move a
    weight by lr*g/(0.1|g|+1e-8): +-10*lr = 5e-4 wherever |g| >> 1e-7, but where the gradient
    is at roundoff level (|g| <~ 1e-7) the SIGN of the step is implementation noise.  So:
    elements with a well-conditioned reference gradient (|g| > 1e-6) must agree to 10 % of a
    step; every element to within ~2 full steps; and at most 0.1 % of the elements may be
    off by more than 10 % of a step."""
    for k, v in after.items():
        if not torch.is_floating_point(v) or k.endswith(tuple(skip)):
            continue
        err = (sd[k].detach().cpu().float() - v).abs()
        assert err.max().item() < 1.2e-3, (k, err.max().item())
        bad = (err > STEP_TOL).float().mean().item()
        assert bad < 1e-3, (k, bad)
        if grads is not None and k in grads:
            well = grads[k].abs() > 1e-6
            if well.any():
                assert err[well].max().item() < STEP_TOL, (k, err[well].max().item())


@pytest.fixture(autouse=True)
def _emulate():
    emu_ops.install()
    yield
    emu_ops.uninstall()


def build(fx):
    from segan_pytorch_amd.models import SEGAN
    m = SEGAN(SimpleNamespace(**fx['opts']))
    m.G.load_state_dict(fx['G0'])
    m.D.load_state_dict(fx['D0'])
    return m


def check_step(fx):
    from segan_pytorch_amd import losses
    m = build(fx)
    Gopt, Dopt = m.build_optimizers(SimpleNamespace(**fx['opts']))
    m.G.train()
    m.D.train()
    random.seed(fx['roll_seed'])
    out = m.gan_step(fx['clean'], fx['noisy'], Gopt, Dopt, losses.MSELoss(), 100.0, z=fx['z'])
    for got, key in zip(out, ('d_real_loss', 'd_fake_loss', 'g_adv_loss', 'g_l1_loss')):
        assert max_rel(got, fx[key]) < 2e-5, key
    for name, net, grads in (('D', m.D, fx['d_grads']), ('G', m.G, fx['g_grads'])):
        named = dict(net.named_parameters())
        for k, g in grads.items():
            if name == 'D' and k.endswith('conv.bias'):
                continue
            assert max_rel(named[k].grad, g) < 1e-4, (name, k)
    assert_weights_after_step(m.G.state_dict(), fx['G_after'], fx['g_grads'])
    assert_weights_after_step(m.D.state_dict(), fx['D_after'], fx['d_grads'], skip=NOISE_KEYS)


def test_gan_step_orchestration(tiny_step):
    check_step(tiny_step)


def test_gan_step_orchestration_stride2(tiny_s2):
    check_step(tiny_s2)


@pytest.mark.parametrize('name', [n for n in VARIANT_NAMES if not n.endswith('_snorm')])
def test_gan_step_orchestration_variants(tiny_variants, name):
    """Host side of the architecture switches (conv skips, pooling-1 layers, conv last block,
    the conv / gmax / gavg discriminator heads) against one step of the real reference each."""
    check_step(tiny_variants[name])


def test_gan_step_orchestration_no_bias():
    """--no_bias (run_segan+_train.sh:7): G's convs without a bias."""
    from conftest import load_golden
    check_step(load_golden('tiny_nobias.pt'))


def _literal_train(fx, o, G0, D0, tmp_path, device='cpu'):
    o = dict(o)
    o['save_path'] = str(tmp_path)
    m = build({'opts': o, 'G0': G0, 'D0': D0})
    if device != 'cpu':
        m = m.to(device)
    loader = [[['u'] * 3, c, n, torch.zeros(3)] for c, n in fx['batches']]
    random.seed(fx['seed'])
    np.random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    m.train(SimpleNamespace(**o), loader, None, o['l1_weight'], o['l1_dec_step'],
            o['l1_dec_epoch'], 1000, va_dloader=None, device=device)
    assert_weights_after_step(m.G.state_dict(), fx['G_final'])
    assert_weights_after_step(m.D.state_dict(), fx['D_final'], skip=NOISE_KEYS)
    return m


def test_constant_skip_is_never_trained(tiny_corners, tmp_path):
    """--skip_type constant (generator.py:25,40-41,59; round-5 review, missing 3): the scale rides in
    the consuming deconv's load like an alpha, but it is not a trainable parameter —
    Model.parameters (core.py:196-198) hides it, so the flat optimizer arena does not hold it,
    get_n_params does not count it, no gradient buffer is ever allocated for it and neither one
    step nor the literal two-batch SEGAN.train moves it."""
    fx = tiny_corners['constantskip']
    m = build(fx)
    named = dict(m.G.named_parameters())
    assert all(not named[k].requires_grad for k in fx['constants'])
    assert m.G.get_n_params() == fx['n_params_G']
    Gopt, _ = m.build_optimizers(SimpleNamespace(**fx['opts']))
    held = {id(p) for p in Gopt._params}
    assert all(id(named[k]) not in held for k in fx['constants'])
    assert sum(p.numel() for p in Gopt._params) == fx['n_params_G']    # the arena pads between views
    check_step(fx)
    m2 = _literal_train(fx['train2'], fx['opts'], fx['G0'], fx['D0'], tmp_path)
    for k in fx['constants']:
        assert torch.equal(m2.G.state_dict()[k], fx['G0'][k]), k
        assert dict(m2.G.named_parameters())[k].grad is None


def test_mse_reg_loss_step_and_train(tiny_corners, tmp_path):
    """--reg_loss mse_loss (train.py:179, model.py:79) at model level: one step and the literal loop."""
    fx = tiny_corners['mseloss']
    check_step(fx)
    _literal_train(fx['train2'], fx['opts'], fx['G0'], fx['D0'], tmp_path)


def test_hidden_outputs(tiny_step):
    fx = tiny_step
    m = build(fx)
    m.G.train()
    m.D.train()
    with torch.no_grad():
        y, hall = m.G(fx['noisy'], z=fx['z'], ret_hid=True)
    assert list(hall.keys()) == list(fx['G_hall'].keys())
    for k, v in fx['G_hall'].items():
        assert max_rel(hall[k], v) < 2e-5, k
    random.seed(fx['roll_seed'])
    with torch.no_grad():
        yd, acts = m.D(torch.cat((fx['clean'], fx['noisy']), 1))
    assert sorted(acts.keys()) == sorted(fx['D_acts'].keys())
    for k, v in fx['D_acts'].items():
        assert max_rel(acts[k], v) < 2e-5, k


def test_literal_train_and_checkpoints(tiny_train2, tmp_path):
    fx = tiny_train2
    o = dict(fx['opts'])
    o['save_path'] = str(tmp_path)
    m = build({'opts': o, 'G0': fx['G0'], 'D0': fx['D0']})
    loader = [[['u'] * 3, c, n, torch.zeros(3)] for c, n in fx['batches']]
    random.seed(fx['seed'])
    np.random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    m.train(SimpleNamespace(**o), loader, None, o['l1_weight'], o['l1_dec_step'],
            o['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
    assert_weights_after_step(m.G.state_dict(), fx['G_final'])
    assert_weights_after_step(m.D.state_dict(), fx['D_final'], skip=NOISE_KEYS)
    # reference checkpoint format (core.py:61-70) and index (core.py:26-59)
    import json
    import os
    idx = json.load(open(os.path.join(str(tmp_path), 'EOE_G-checkpoints')))
    assert idx['current'] == 'EOE_G-Generator-3.ckpt' and idx['latest'] == [idx['current']]
    ck = torch.load(os.path.join(str(tmp_path), 'weights_' + idx['current']), weights_only=False)
    assert set(ck.keys()) == {'step', 'state_dict', 'optimizer'}
    assert set(ck['optimizer']['state'][0].keys()) == {'step', 'square_avg'}
    # a fresh generator restores from it (Model.load_pretrained, core.py:187-190)
    from segan_pytorch_amd.models import SEGAN
    m2 = SEGAN(SimpleNamespace(**o))
    m2.G.load_pretrained(os.path.join(str(tmp_path), 'weights_' + idx['current']), True)
    for k, v in m.G.state_dict().items():
        assert torch.equal(m2.G.state_dict()[k], v), k


def test_blocks_standalone(tiny_step):
    import segan_oracle as O
    from segan_pytorch_amd.models import GConv1DBlock, GDeconv1DBlock
    torch.manual_seed(5)
    blk = GConv1DBlock(6, 10, 31, stride=4, bias=True, norm_type='bnorm')
    blk.act.weight.data.uniform_(0.1, 0.3)
    x = torch.randn(3, 6, 128)
    xg = x.clone().requires_grad_(True)
    h = blk(xg)
    h.square().sum().backward()
    sd = {k: v.detach().double() for k, v in blk.state_dict().items()}
    for k in ('conv.weight', 'conv.bias', 'act.weight', 'norm.weight', 'norm.bias'):
        sd[k].requires_grad_(True)
    xd = x.double().requires_grad_(True)
    bn = {'weight': sd['norm.weight'], 'bias': sd['norm.bias'],
          'running_mean': torch.zeros(10, dtype=torch.float64),
          'running_var': torch.ones(10, dtype=torch.float64)}
    hr, _ = O.gconv_block(xd, sd['conv.weight'], sd['conv.bias'], sd['act.weight'], 4, bn=bn)
    hr.square().sum().backward()
    assert max_rel(h, hr) < 2e-5 and max_rel(xg.grad, xd.grad) < 1e-4
    assert max_rel(blk.conv.weight.grad, sd['conv.weight'].grad) < 1e-4
    assert max_rel(blk.norm.weight.grad, sd['norm.weight'].grad) < 1e-4
    db = GDeconv1DBlock(10, 4, 31, stride=4)
    db.act.weight.data.uniform_(0.1, 0.3)
    xq = torch.randn(2, 10, 32)
    xqg = xq.clone().requires_grad_(True)
    db(xqg).square().sum().backward()
    sd = {k: v.detach().double().requires_grad_(True) for k, v in db.state_dict().items()}
    xqd = xq.double().requires_grad_(True)
    O.gdeconv_block(xqd, sd['deconv.weight'], sd['deconv.bias'], sd['act.weight'],
                    4).square().sum().backward()
    assert max_rel(xqg.grad, xqd.grad) < 1e-4
    assert max_rel(db.deconv.weight.grad, sd['deconv.weight'].grad) < 1e-4
    # deconv block WITH BatchNorm, stand-alone (modules.py:109-141 with norm_type='bnorm'): PReLU and
    # Tanh flavours — the Tanh cannot ride in the contraction's epilogue behind a BatchNorm
    for act in (None, 'Tanh'):
        torch.manual_seed(11)
        dbn = GDeconv1DBlock(10, 4, 31, stride=4, norm_type='bnorm', act=act)
        if act is None:
            dbn.act.weight.data.uniform_(0.1, 0.3)
        dbn.norm.weight.data.uniform_(0.5, 1.5)
        dbn.norm.bias.data.uniform_(-0.2, 0.2)
        xb = torch.randn(3, 10, 32)
        xbg = xb.clone().requires_grad_(True)
        cw = torch.randn(3, 4, 128)
        yb = dbn(xbg)
        (yb * cw).sum().backward()
        sd = {k: v.detach().double() for k, v in dbn.state_dict().items()}
        keys = ['deconv.weight', 'deconv.bias', 'norm.weight', 'norm.bias'] + (['act.weight'] if act is None else [])
        for k in keys:
            sd[k].requires_grad_(True)
        bn = {'weight': sd['norm.weight'], 'bias': sd['norm.bias'],
              'running_mean': torch.zeros(4, dtype=torch.float64),
              'running_var': torch.ones(4, dtype=torch.float64)}
        xbd = xb.detach().double().requires_grad_(True)
        yr = O.gdeconv_block(xbd, sd['deconv.weight'], sd['deconv.bias'], sd.get('act.weight'), 4,
                             tanh=act is not None, bn=bn)
        (yr * cw.double()).sum().backward()
        assert max_rel(yb, yr) < 2e-5, act
        assert max_rel(xbg.grad, xbd.grad) < 1e-4, act
        for k in keys:
            if k == 'deconv.bias':
                continue        # cancelled by the BatchNorm: roundoff on both sides
            mod, name = k.split('.')
            assert max_rel(getattr(getattr(dbn, mod), name).grad, sd[k].grad) < 1e-4, (act, k)
        assert max_rel(dbn.norm.running_mean, bn['running_mean']) < 1e-4
        assert max_rel(dbn.norm.running_var, bn['running_var']) < 1e-4


def _torch_block(kind, blk, x, act_tanh=False):
    """The same block composed from torch.nn.functional ops on the block's own parameters."""
    import torch.nn.functional as F
    K, S = blk.kwidth, blk.stride
    if kind == 'conv':
        c = F.conv1d(F.pad(x, (K // 2 - 1, K // 2), mode='reflect'), blk.conv.weight, blk.conv.bias, stride=S)
    else:
        pad = max(0, (S - K) // -2)
        c = F.conv_transpose1d(x, blk.deconv.weight, blk.deconv.bias, stride=S, padding=pad)
        c = c[:, :, :-1] if K % 2 else c
    bn = blk.norm
    c = F.batch_norm(c, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training or
                     bn.running_mean is None, bn.momentum if bn.momentum is not None else 0.0, bn.eps)
    return torch.tanh(c) if act_tanh else F.prelu(c, blk.act.weight)


@pytest.mark.parametrize('kind,act', [('conv', None), ('deconv', None), ('deconv', 'Tanh')])
def test_standalone_block_backward_with_batchnorm_in_eval(kind, act):
    """Round-5 advice: a stand-alone block whose BatchNorm is in eval() — fine-tuning with frozen
    statistics — is differentiable in the reference (autograd goes through the fixed affine map);
    here it used to raise.  Output, input gradient and every parameter gradient (gamma and beta
    included) against plain torch autograd of the same composition."""
    import copy
    from segan_pytorch_amd.models import GConv1DBlock, GDeconv1DBlock
    torch.manual_seed(3)
    if kind == 'conv':
        blk = GConv1DBlock(6, 10, 31, stride=4, bias=True, norm_type='bnorm')
        x = torch.randn(3, 6, 128)
    else:
        blk = GDeconv1DBlock(10, 4, 31, stride=4, norm_type='bnorm', act=act)
        x = torch.randn(2, 10, 32)
    if act is None:
        blk.act.weight.data.uniform_(0.1, 0.3)
    blk.norm.weight.data.uniform_(0.5, 1.5)
    blk.norm.bias.data.uniform_(-0.2, 0.2)
    blk.norm.running_mean.uniform_(-0.3, 0.3)
    blk.norm.running_var.uniform_(0.5, 2.0)
    ref = copy.deepcopy(blk)
    blk.eval()
    ref.eval()
    xg, xr = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    w = torch.randn(1)
    h = blk(xg)
    h = h[0] if isinstance(h, tuple) else h
    hr = _torch_block(kind, ref, xr, act_tanh=act == 'Tanh')
    assert max_rel(h, hr) < 2e-5
    c = torch.randn_like(hr)
    (h * c).sum().backward()
    (hr * c).sum().backward()
    assert max_rel(xg.grad, xr.grad) < 1e-4
    for (k, p), (_k, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and max_rel(p.grad, q.grad) < 1e-4, k
    # the running statistics did not move
    assert torch.equal(blk.norm.running_mean, ref.norm.running_mean)
    assert int(blk.norm.num_batches_tracked) == 0


def test_batchnorm_without_tracked_statistics_and_cumulative_momentum():
    """nn.BatchNorm1d corners the blocks inherit from torch: track_running_stats=False (no buffers:
    batch statistics also in eval mode — this used to dereference running_var) and momentum=None (the
    running statistics are the cumulative average 1 / num_batches_tracked, not a 0.1 EMA)."""
    from segan_pytorch_amd.models import GDeconv1DBlock
    torch.manual_seed(4)
    blk = GDeconv1DBlock(10, 4, 31, stride=4, norm_type='bnorm')
    blk.act.weight.data.uniform_(0.1, 0.3)
    blk.norm = torch.nn.BatchNorm1d(4, track_running_stats=False)
    x = torch.randn(3, 10, 32)
    for mode in (True, False):
        blk.train(mode)
        h = blk(x)
        assert max_rel(h, _torch_block('deconv', blk, x)) < 2e-5
    blk.norm = torch.nn.BatchNorm1d(4, momentum=None)
    ref = torch.nn.BatchNorm1d(4, momentum=None)
    blk.train()
    ref.train()
    import torch.nn.functional as F
    for i in range(3):
        xi = torch.randn(3, 10, 32)
        blk(xi)
        c = F.conv_transpose1d(xi, blk.deconv.weight, blk.deconv.bias, stride=4, padding=13)[:, :, :-1]
        ref(c.detach())
    assert int(blk.norm.num_batches_tracked) == 3
    assert max_rel(blk.norm.running_mean, ref.running_mean) < 1e-5
    assert max_rel(blk.norm.running_var, ref.running_var) < 1e-5


@pytest.mark.parametrize('golden', ['tiny_wsegan2.pt', 'tiny_wsegan_snorm.pt', 'vanillagan'])
def test_wsegan_literal_train(golden, tmp_path):
    """WSEGAN.train (misalign pair, STFT power loss, masked L1) against the reference's
    literal WSEGAN.train; the second fixture is the run_wsegan_train.sh flavour
    (--dnorm_type snorm --opt adam), the third --vanilla_gan (BCE cost, model.py:582-585)."""
    from conftest import load_golden
    from segan_pytorch_amd.models import WSEGAN
    fx = load_golden('tiny_corners.pt')[golden] if golden == 'vanillagan' else load_golden(golden)
    o = dict(fx['opts'])
    o['save_path'] = str(tmp_path)
    o['epoch'] = fx['iters']
    m = WSEGAN(SimpleNamespace(**o))
    m.G.load_state_dict(fx['G0'])
    m.D.load_state_dict(fx['D0'])
    loader = [[fx['names'], fx['clean'], fx['noisy'], torch.zeros(3)]]
    random.seed(fx['seed'])
    np.random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    m.train(SimpleNamespace(**o), loader, None, o['l1_weight'], o['l1_dec_step'],
            o['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
    assert_weights_after_step(m.G.state_dict(), fx['G_final'])
    assert_weights_after_step(m.D.state_dict(), fx['D_final'], skip=NOISE_KEYS)


@pytest.mark.parametrize('flavour', ['both', 'interf_only'])
def test_wsegan_interf_pair_literal_train(flavour, tmp_path):
    """WSEGAN.train with --interf_pair (with and without --misalign_pair) against the
    reference's literal loop (oracle/make_golden.py interf)."""
    from conftest import load_golden
    from segan_pytorch_amd.models import WSEGAN
    fx = load_golden('tiny_wsegan_interf.pt')[flavour]
    o = dict(fx['opts'])
    o['save_path'] = str(tmp_path)
    o['epoch'] = fx['iters']
    m = WSEGAN(SimpleNamespace(**o))
    m.G.load_state_dict(fx['G0'])
    m.D.load_state_dict(fx['D0'])
    loader = [[fx['names'], fx['clean'], fx['noisy'], torch.zeros(3)]]
    random.seed(fx['seed'])
    np.random.seed(fx['seed'])
    torch.manual_seed(fx['seed'])
    m.train(SimpleNamespace(**o), loader, None, o['l1_weight'], o['l1_dec_step'],
            o['l1_dec_epoch'], 1000, va_dloader=None, device='cpu')
    assert_weights_after_step(m.G.state_dict(), fx['G_final'])
    assert_weights_after_step(m.D.state_dict(), fx['D_final'], skip=NOISE_KEYS)


def test_wsegan_train_keeps_the_z_lookahead_only_with_a_private_sampling_generator(tmp_path):
    """WSEGAN.train draws a fresh batch every step.  With a plain loader that draw reseeds from
    torch's global generator between two z draws (model.py:526-535, generator.py:197), so the next z
    cannot be drawn ahead without changing the reference's stream: look-ahead off.  A loader whose
    sample() uses its own generator (PCMShardLoader: `sample_keeps_global_rng`) leaves the global
    generator to the z draws alone: look-ahead on for every step but the last, off again at the end."""
    from conftest import load_golden
    from segan_pytorch_amd.models import WSEGAN
    fx = load_golden('tiny_wsegan2.pt')
    o = dict(fx['opts'])
    o['save_path'] = str(tmp_path)
    o['epoch'] = 3

    class Loader(object):
        def __init__(self, private):
            if private:
                self.sample_keeps_global_rng = True
            self.seen = []

        def __len__(self):
            return 1

        def sample(self):
            self.seen.append(m.G.z_prefetch)
            return [fx['names'], fx['clean'], fx['noisy'], torch.zeros(3)]

    for private, want in ((True, [False, True, True]), (False, [False, False, False])):
        m = WSEGAN(SimpleNamespace(**o))
        m.G.load_state_dict(fx['G0'])
        m.D.load_state_dict(fx['D0'])
        ld = Loader(private)
        random.seed(1)
        torch.manual_seed(1)
        m.train(SimpleNamespace(**o), ld, None, o['l1_weight'], o['l1_dec_step'], o['l1_dec_epoch'], 1000,
                va_dloader=None, device='cpu')
        # sample() of step i sees the flag step i - 1 left: on after every step but the last
        assert ld.seen == want, (private, ld.seen)
        assert m.G.z_prefetch is False


def _sum_merge_reference(sd, x, z):
    """generator.py:180-230 with skip_merge='sum' (GSkip.forward 64-74)."""
    import segan_oracle as O
    hi, skips = x, {}
    for l in range(3):
        p = 'enc_blocks.%d.' % l
        hi, lin = O.gconv_block(hi, sd[p + 'conv.weight'], sd[p + 'conv.bias'], sd[p + 'act.weight'], 4)
        if l < 2:
            skips[l] = lin
    hi = torch.cat((z, hi), 1)
    e = 2
    for l in range(3):
        if e in skips:
            hi = sd['alpha_%d.skip_k' % e] * skips[e] + hi
        p = 'dec_blocks.%d.' % l
        hi = O.gdeconv_block(hi, sd[p + 'deconv.weight'], sd[p + 'deconv.bias'],
                             sd.get(p + 'act.weight'), 4, tanh=(l == 2))
        e -= 1
    return hi


def make_sum_generator(device='cpu'):
    from segan_pytorch_amd.models import Generator
    torch.manual_seed(0)
    g = Generator(1, [8, 16, 32], 31, [4, 4, 4], z_dim=32, skip_merge='sum', bias=True)
    for p in g.parameters():
        if p.dim() == 1:
            p.data.uniform_(0.05, 0.3)
    for i in range(2):
        getattr(g, 'alpha_%d' % i).skip_k.data.uniform_(0.5, 1.5)
    return g.to(device)


def test_generator_sum_merge():
    """skip_merge='sum' is the Generator's own default (generator.py:95)."""
    g = make_sum_generator()
    x, z = torch.randn(2, 1, 1024), torch.randn(2, 32, 16)
    y = g(x, z=z)
    y.square().sum().backward()
    sd = {k: v.detach().double().requires_grad_(True) for k, v in g.state_dict().items()}
    yr = _sum_merge_reference(sd, x.double(), z.double())
    yr.square().sum().backward()
    assert max_rel(y, yr) < 2e-5
    for k, p in g.named_parameters():
        assert max_rel(p.grad, sd[k].grad) < 1e-4, k


def test_bce_cost():
    import torch.nn.functional as F
    from segan_pytorch_amd import losses
    d = torch.randn(7, 1, requires_grad=True)
    l = losses.BCEWithLogitsLoss()(d.view(-1), 1.0)
    (0.5 * l).backward()
    dd = d.detach().double().requires_grad_(True)
    lr = F.binary_cross_entropy_with_logits(dd.view(-1), torch.ones(7, dtype=torch.float64))
    (0.5 * lr).backward()
    assert abs(l.item() - lr.item()) < 1e-6 and max_rel(d.grad, dd.grad) < 1e-5


def test_legacy_generator_checkpoint_loads(tiny_step, tmp_path):
    """A checkpoint with the old gen_enc / gen_dec key names (what weightG_fmt_converter.py
    rewrites offline in the reference) loads directly through load_pretrained."""
    from segan_pytorch_amd.models.core import convert_legacy_generator_keys
    fx = tiny_step
    m = build(fx)
    legacy = {}
    for k, v in fx['G0'].items():
        if k.startswith('enc_blocks'):
            k = k.replace('enc_blocks', 'gen_enc')
        elif k.startswith('dec_blocks'):
            k = k.replace('dec_blocks', 'gen_dec').replace('deconv', 'conv')
        legacy[k] = v + 0.5
    assert any('gen_dec' in k for k in legacy)
    assert set(convert_legacy_generator_keys(legacy).keys()) == set(fx['G0'].keys())
    path = str(tmp_path / 'old_G.ckpt')
    torch.save({'state_dict': legacy}, path)
    m.G.load_pretrained(path, load_last=True)
    for k, v in m.G.state_dict().items():
        assert torch.equal(v.cpu(), fx['G0'][k] + 0.5), k


def test_weightG_fmt_converter_cli(tiny_step, tmp_path):
    """weightG_fmt_converter.py <file> writes <file>.v2 with today's key names and every other
    top-level entry carried over (weightG_fmt_converter.py:18-44 of the reference); the result
    loads as a pretrained generator."""
    import os, subprocess, sys
    fx = tiny_step
    legacy = {}
    for k, v in fx['G0'].items():
        if k.startswith('enc_blocks'):
            k = k.replace('enc_blocks', 'gen_enc')
        elif k.startswith('dec_blocks'):
            k = k.replace('dec_blocks', 'gen_dec').replace('deconv', 'conv')
        legacy[k] = v + 0.25
    path = str(tmp_path / 'old_G.ckpt')
    torch.save({'state_dict': legacy, 'step': 17}, path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'weightG_fmt_converter.py'), path],
                         capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr
    assert 'gen_dec.0.conv.weight -> dec_blocks.0.deconv.weight' in out.stdout
    ck = torch.load(path + '.v2', weights_only=False)
    assert ck['step'] == 17 and list(ck['state_dict'].keys()) == list(fx['G0'].keys())
    m = build(fx)
    m.G.load_pretrained(path + '.v2', load_last=True)
    for k, v in m.G.state_dict().items():
        assert torch.equal(v.cpu(), fx['G0'][k] + 0.25), k


def test_gan_step_with_spectral_norm(tiny_snorm):
    """--dnorm_type snorm through the host logic (weights from ops.snorm_fwd per forward call,
    gradients folded back into weight_orig by ops.snorm_bwd), against the reference."""
    fx = tiny_snorm
    m = build(fx)
    assert sorted(m.D.state_dict().keys()) == sorted(fx['D0'].keys())
    check_step(fx)


def test_generator_with_spectral_norm(tiny_snorm):
    from segan_pytorch_amd.models import Generator
    g = tiny_snorm['gsn']
    G = Generator(1, [8, 16, 32], 31, [4, 4, 4], z_dim=32, skip_merge='concat', bias=True,
                  norm_type='snorm')
    assert sorted(G.state_dict().keys()) == sorted(g['G0'].keys())
    G.load_state_dict(g['G0'])
    G.train()
    y = G(g['x'], z=g['z'])
    assert max_rel(y, g['y']) < 2e-5
    (y * g['c']).sum().backward()
    named = dict(G.named_parameters())
    for k, gr in g['grads'].items():
        assert max_rel(named[k].grad, gr) < 1e-4, k
    for k, v in g['G_after_fwd'].items():
        assert max_rel(G.state_dict()[k], v) < 2e-5, k


@pytest.mark.parametrize('name', GVARIANT_NAMES)
def test_generator_batchnorm_and_skip_dropout(name):
    """Generator(norm_type='bnorm') and skip_dropout, host side: against the REAL reference's
    output, gradients, running statistics and eval-mode output (tests/golden/tiny_gvariants.pt)."""
    from conftest import load_golden
    check_gvariant(load_golden('tiny_gvariants.pt')[name], 'cpu', 2e-5, 1e-4)


def test_distributed_sampler_epochs_advance(tiny_wsegan2):
    """Under data parallelism the loaders use a DistributedSampler, which only reshuffles when
    its epoch changes: WSEGAN.sample_dloader (a fresh iterator per step, model.py:526-535) must
    not replay the same first batch forever, and SEGAN.train must reshuffle every epoch."""
    from torch.utils.data import DataLoader, Dataset
    from torch.utils.data.distributed import DistributedSampler
    from segan_pytorch_amd.models import WSEGAN

    class DS(Dataset):
        def __len__(self):
            return 64

        def __getitem__(self, i):
            return 'u%d' % i, torch.full((8,), float(i)), torch.full((8,), float(i)), 0

    ds = DS()
    dl = DataLoader(ds, batch_size=4, sampler=DistributedSampler(ds, num_replicas=2, rank=0, shuffle=True))
    m = WSEGAN(SimpleNamespace(**tiny_wsegan2['opts']))
    firsts = [tuple(m.sample_dloader(dl)[1][:, 0, 0].tolist()) for _ in range(4)]
    assert len(set(firsts)) > 1, firsts
