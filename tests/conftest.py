import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location='cpu', weights_only=False)


@pytest.fixture(scope='session')
def tiny_step():
    return load_golden('tiny_step.pt')


@pytest.fixture(scope='session')
def tiny_train2():
    return load_golden('tiny_train2.pt')


@pytest.fixture(scope='session')
def tiny_s2():
    return load_golden('tiny_s2.pt')


@pytest.fixture(scope='session')
def tiny_snorm():
    return load_golden('tiny_snorm.pt')


@pytest.fixture(scope='session')
def tiny_wsegan2():
    return load_golden('tiny_wsegan2.pt')


@pytest.fixture(scope='session')
def tiny_variants():
    return load_golden('tiny_variants.pt')


VARIANT_NAMES = ('skipconv_concat', 'skipconv_sum', 'pool1_mid', 'pool1_last', 'dpool_conv',
                 'dpool_gmax', 'dpool_gavg', 'dpool_conv_snorm')


def oracle_kwargs(opts):
    """oracle.gan_step / generator_forward keyword arguments for a train.py option dict."""
    kw = dict(dec_strides=opts.get('gdec_poolings') or None, d_strides=opts['denc_poolings'],
              skip_merge=opts['skip_merge'], pool_type=opts['dpool_type'])
    if opts.get('reg_loss', 'l1_loss') != 'l1_loss':
        kw['reg_loss'] = opts['reg_loss']
    if opts.get('skip_type') == 'constant':      # generator.py:40-41: skip_k.requires_grad = False
        kw['frozen'] = tuple('alpha_{}.skip_k'.format(i) for i in range(len(opts['genc_fmaps']) - 1))
    return kw


@pytest.fixture(scope='session')
def tiny_corners():
    return load_golden('tiny_corners.pt')


GVARIANT_NAMES = ('bnorm_concat', 'bnorm_sum', 'dropout_alpha', 'dropout_conv_sum', 'bnorm_dropout')


def check_gvariant(g, device, act_tol, grad_tol):
    """A Generator option no train.py flag reaches (BatchNorm in G, skip dropout) against the
    REAL reference's forward / backward (tests/golden/tiny_gvariants.pt) on `device`."""
    from segan_pytorch_amd.models import Generator
    G = Generator(1, [4, 8, 16], 31, [4, 4, 4], **g['kwargs'])
    assert sorted(G.state_dict().keys()) == sorted(g['G0'].keys())
    G.load_state_dict(g['G0'])
    G = G.to(device)
    G.train()
    torch.manual_seed(g['fwd_seed'])          # the dropout masks of the reference's forward
    y = G(g['x'].to(device), z=g['z'].to(device))
    assert max_rel(y, g['y']) < act_tol
    (y * g['c'].to(device)).sum().backward()
    named = dict(G.named_parameters())
    for k, gr in g['grads'].items():
        if g['kwargs'].get('norm_type') == 'bnorm' and k.endswith(('conv.bias', 'deconv.bias')):
            continue    # a bias in front of a BatchNorm: mathematically zero gradient, roundoff
        assert max_rel(named[k].grad, gr) < grad_tol, k
    for k, v in g['G_after_fwd'].items():     # BatchNorm running statistics after the forward
        if torch.is_floating_point(v):
            assert max_rel(G.state_dict()[k], v) < act_tol, k
        else:
            assert torch.equal(G.state_dict()[k].cpu(), v), k
    G.eval()
    with torch.no_grad():
        ye = G(g['x'].to(device), z=g['z'].to(device))
    assert max_rel(ye, g['y_eval']) < act_tol



def draw_rolls(n_layers, phase_shift):
    """The python-`random` draws of one Discriminator.forward (discriminator.py:159-163)."""
    import random
    out = []
    for _ in range(n_layers):
        s = random.randint(1, phase_shift)
        out.append(s if random.random() > 0.5 else -s)
    return out


@pytest.fixture(scope='session')
def segan_plus_b2():
    return load_golden('segan_plus_b2.pt')


@pytest.fixture(scope='session')
def vanilla11_b8():
    return load_golden('vanilla11_b8.pt')


def max_rel(a, b):
    """max |a-b| / max(|b|) — scale-aware error for tensors."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)
