"""Drop-in boundary checks that need no GPU: the C-ABI library loads and exports every
symbol include/segan_hip.h declares, the Python surface mirrors the reference's
(constructor signatures, state_dict keys, CLI flags, checkpoint format), and the product
path refuses to run without a HIP device (no fallback)."""
import inspect
import json
import os
import re
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from segan_pytorch_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'segan_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(segan_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'segan_src'}
    assert len(declared) >= 25
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), 'libsegan_hip.so does not export ' + name
        assert name in _lib.SIGNATURES, 'no ctypes signature for ' + name
    assert set(_lib.SIGNATURES) == declared
    assert lib.segan_abi_version() == _lib.ABI_VERSION
    # argument validation happens before any launch and reports through segan_last_error
    assert lib.segan_packed_f_bytes(64, 64, 3) == 0
    rc = lib.segan_fill(None, 0.0, 10, None)
    assert rc != 0 and b'fill' in lib.segan_last_error()


def test_constructor_signatures_match_the_reference():
    from segan_pytorch_amd.models import Discriminator, GConv1DBlock, GDeconv1DBlock, Generator
    g = list(inspect.signature(Generator.__init__).parameters)
    assert g == ['self', 'ninputs', 'fmaps', 'kwidth', 'poolings', 'dec_fmaps', 'dec_kwidth',
                 'dec_poolings', 'z_dim', 'no_z', 'skip', 'bias', 'skip_init', 'skip_dropout',
                 'skip_type', 'norm_type', 'skip_merge', 'skip_kwidth', 'name']
    d = list(inspect.signature(Discriminator.__init__).parameters)
    assert d == ['self', 'ninputs', 'fmaps', 'kwidth', 'poolings', 'pool_type', 'pool_slen',
                 'norm_type', 'bias', 'phase_shift', 'sinc_conv']
    assert list(inspect.signature(GConv1DBlock.__init__).parameters) == \
        ['self', 'ninp', 'fmaps', 'kwidth', 'stride', 'bias', 'norm_type']
    assert list(inspect.signature(GDeconv1DBlock.__init__).parameters) == \
        ['self', 'ninp', 'fmaps', 'kwidth', 'stride', 'bias', 'norm_type', 'act']
    assert list(inspect.signature(Generator.forward).parameters) == ['self', 'x', 'z', 'ret_hid']


def test_state_dict_keys_and_shapes_match_reference(tiny_step, segan_plus_b2):
    from segan_pytorch_amd.models import SEGAN
    for fx, has_w in ((tiny_step, True), (segan_plus_b2, False)):
        m = SEGAN(SimpleNamespace(**fx['opts']))
        if has_w:
            assert list(m.G.state_dict().keys()) == list(fx['G0'].keys())
            assert list(m.D.state_dict().keys()) == list(fx['D0'].keys())
            for k, v in fx['G0'].items():
                assert m.G.state_dict()[k].shape == v.shape, k
            for k, v in fx['D0'].items():
                assert m.D.state_dict()[k].shape == v.shape, k
        else:
            assert list(m.G.state_dict().keys()) == list(fx['init_G'].keys())
            assert list(m.D.state_dict().keys()) == list(fx['init_D'].keys())
            assert m.G.get_n_params() == 64770561 and m.D.get_n_params() == 25825793


def test_cli_flags_cover_the_reference_train_opts(segan_plus_b2):
    import train
    ns = vars(train.build_parser().parse_args([]))
    ref_keys = set(segan_plus_b2['opts'].keys()) - {'l1_loss', 'bias', 'reg_loss', 'save_path'}
    missing = ref_keys - set(ns.keys())
    assert not missing, missing
    assert ns['batch_size'] == 100 and ns['gkwidth'] == 31 and ns['genc_poolings'] == [4] * 5
    assert ns['skip_merge'] == 'concat' and ns['dnorm_type'] == 'bnorm' and ns['phase_shift'] == 5
    import clean
    c = vars(clean.build_parser().parse_args([]))
    assert set(c) == {'g_pretrained_ckpt', 'test_files', 'h5', 'seed', 'synthesis_path', 'cuda',
                      'soundfile', 'cfg_file'}


def test_no_cpu_fallback(tiny_step):
    from segan_pytorch_amd import ops
    from segan_pytorch_amd.models import SEGAN
    m = SEGAN(SimpleNamespace(**tiny_step['opts']))
    with pytest.raises(RuntimeError, match='MI355X'):
        m.G(tiny_step['noisy'], z=tiny_step['z'])
    with pytest.raises(RuntimeError, match='MI355X'):
        m.D(torch.cat((tiny_step['clean'], tiny_step['noisy']), 1))
    Gopt, Dopt = m.build_optimizers(SimpleNamespace(**tiny_step['opts']))
    with pytest.raises(RuntimeError, match='MI355X'):
        Gopt.step()
    with pytest.raises(RuntimeError):
        ops.Src(torch.zeros(1, 1, 64))


def test_argument_validation_mirrors_reference_errors():
    from segan_pytorch_amd.models import Discriminator, GConv1DBlock, Generator
    with pytest.raises(ValueError):          # discriminator.py:83-86
        Discriminator(2, [8, 16], 31, [4, 4], pool_slen=None)
    with pytest.raises(TypeError):           # modules.py:17-18
        GConv1DBlock(1, 4, 31, stride=4, norm_type='bogus')
    with pytest.raises(AssertionError):      # generator.py:105
        Generator(1, (8, 16), 31, [4, 4])
    g = Generator(1, [8, 16], 31, [4, 4], z_dim=8)        # default skip_merge='sum' builds
    with pytest.raises(ValueError):
        g(torch.zeros(1, 1, 30))             # not divisible by the pooling
    with pytest.raises(ValueError):          # generator.py:200-202
        g(torch.zeros(1, 1, 32), z=torch.zeros(1, 8))
    for bad in (dict(sinc_conv=True), dict(pool_type='mlp')):
        with pytest.raises(NotImplementedError):
            Discriminator(2, [8, 16], 31, [4, 4], pool_slen=2, **bad)
    with pytest.raises(TypeError):           # discriminator.py:147-148
        Discriminator(2, [8, 16], 31, [4, 4], pool_slen=2, pool_type='bogus')
    # the other heads build with the reference's state_dict keys (discriminator.py:122-137)
    for pt, keys in (('conv', {'pool_conv.weight', 'pool_conv.bias', 'fc.weight', 'fc.bias'}),
                     ('gmax', {'fc.weight', 'fc.bias'}), ('gavg', {'fc.weight', 'fc.bias'})):
        dd = Discriminator(2, [8, 16], 31, [4, 4], pool_slen=2, pool_type=pt)
        assert {k for k in dd.state_dict() if not k.startswith('enc_blocks')} == keys, pt
    gc = Generator(1, [8, 16, 32], 31, [4, 1, 4], z_dim=8, skip_type='conv', skip_merge='concat',
                   bias=True)
    assert 'alpha_0.skip_k.weight' in gc.state_dict() and 'dec_blocks.1.conv.weight' in gc.state_dict()
    d = Discriminator(2, [8, 16], 31, [4, 4], pool_slen=2, norm_type='snorm')
    assert 'enc_blocks.0.conv.weight_orig' in d.state_dict() and 'fc.3.weight_u' in d.state_dict()


def test_flat_arena_optimizer_state_dict_format():
    from segan_pytorch_amd import optim as soptim
    ps = [torch.nn.Parameter(torch.randn(3, 2, 5)), torch.nn.Parameter(torch.randn(7))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ours, ref = soptim.RMSprop(qs, lr=5e-5), torch.optim.RMSprop(ps, lr=5e-5)
    # parameters became views of one arena, values preserved, grads pre-allocated
    assert all(torch.equal(p.detach(), q.detach()) for p, q in zip(ps, qs))
    assert qs[1].data_ptr() == ours.flat_param.data_ptr() + 4 * ours._offsets[1]
    assert all(q.grad is not None and float(q.grad.abs().sum()) == 0 for q in qs)
    for p in ps:
        p.grad = torch.zeros_like(p)
    ref.step()
    sd, rsd = ours.state_dict(), ref.state_dict()
    assert set(sd['state'][0].keys()) == set(rsd['state'][0].keys()) == {'step', 'square_avg'}
    for k in ('lr', 'alpha', 'eps', 'momentum', 'centered', 'weight_decay'):
        assert sd['param_groups'][0][k] == rsd['param_groups'][0][k]
    ours.load_state_dict(rsd)           # a torch RMSprop checkpoint loads
    assert ours.state[qs[0]]['square_avg'].data_ptr() == \
        ours._flat_state['square_avg'].data_ptr()
    ours.zero_grad()
    # a gradient re-allocated behind our back is folded into the arena again
    qs[0].grad = torch.ones_like(qs[0])
    ours._resync()
    assert qs[0].grad.data_ptr() == ours.flat_grad.data_ptr()
    assert float(ours.flat_grad[:30].sum()) == 30.0


def test_saver_rotation_matches_reference_format(tmp_path):
    from segan_pytorch_amd.models import Generator, Saver
    g = Generator(1, [4, 8], 31, [4, 4], z_dim=8, skip_merge='concat')
    sv = Saver(g, str(tmp_path), max_ckpts=2, prefix='EOE_G-')
    for step in (1, 2, 3, 4, 5):
        sv.save('Generator', step)
    sv.wait()       # writing is asynchronous: join the writer before looking at the files
    idx = json.load(open(os.path.join(str(tmp_path), 'EOE_G-checkpoints')))
    assert idx['current'] == 'EOE_G-Generator-5.ckpt'
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.startswith('weights_'))
    # the reference drops the oldest once MORE than max_ckpts are listed (core.py:40-51)
    assert files == ['weights_EOE_G-Generator-{}.ckpt'.format(s) for s in (3, 4, 5)]
    ck = torch.load(os.path.join(str(tmp_path), files[-1]), weights_only=False)
    assert set(ck.keys()) == {'step', 'state_dict'} and ck['step'] == 5


def test_weight_pack_cache_follows_the_optimizer_epoch(monkeypatch):
    """A packed weight copy is refreshed when the fused optimizer has touched the weight through
    raw pointers (ops.bump_weights_epoch marks the PARAMETER object): callers must hand the
    parameter itself to the pack, not a detached alias, which carries no mark."""
    from segan_pytorch_amd import _lib, ops
    calls = []

    class FakeLib(object):
        def segan_packed_g_bytes(self, M, N, S):
            return M * N * 32 * 4

        def segan_pack_weights_g(self, w, wg, M, N, K, S, stream):
            calls.append((M, N, K, S))
            return 0

    monkeypatch.setattr(_lib, 'load', lambda: FakeLib())
    monkeypatch.setattr(ops, '_chk', lambda t, name, ndim=None: t)
    monkeypatch.setattr(ops, '_stream', lambda: None)
    w = torch.nn.Parameter(torch.randn(16, 8, 31))
    pack = ops.WeightPack()
    a = pack.g(w, 4)
    assert pack.g(w, 4) is a and len(calls) == 1            # cached
    ops.bump_weights_epoch([w])                              # what the fused optimizers do
    pack.g(w, 4)
    assert len(calls) == 2                                   # re-packed
    with torch.no_grad():
        w.add_(1.0)                                          # an in-place torch update: version counter
    pack.g(w, 4)
    assert len(calls) == 3
    ops.bump_weights_epoch()                                 # global invalidation (DP broadcast)
    pack.g(w, 4)
    assert len(calls) == 4


def test_async_checkpoint_snapshot_is_taken_at_save_time(tmp_path):
    """Saver.save returns before the file is written; what is written is the state AT THE CALL,
    whatever happens to the weights afterwards, in exactly the synchronous format."""
    import torch
    from segan_pytorch_amd.models import Generator, Saver
    g = Generator(1, [4, 8], 31, [4, 4], z_dim=8, skip_merge='concat')
    want = {k: v.clone() for k, v in g.state_dict().items()}
    sv = Saver(g, str(tmp_path), prefix='A-')
    sv.save('Generator', 7)
    with torch.no_grad():
        for p in g.parameters():
            p.add_(1.0)                      # training goes on
    sv.wait()
    ck = torch.load(os.path.join(str(tmp_path), 'weights_A-Generator-7.ckpt'), weights_only=False)
    assert ck['step'] == 7 and list(ck['state_dict'].keys()) == list(want.keys())
    for k, v in want.items():
        assert torch.equal(ck['state_dict'][k], v), k
    sync = Saver(g, str(tmp_path), prefix='S-', async_save=False)
    sync.save('Generator', 8)
    ck2 = torch.load(os.path.join(str(tmp_path), 'weights_S-Generator-8.ckpt'), weights_only=False)
    assert list(ck2['state_dict'].keys()) == list(ck['state_dict'].keys())
    # nn.Module.state_dict()'s per-module version record survives the snapshot (round-3 advice)
    assert ck2['state_dict']._metadata and ck['state_dict']._metadata == ck2['state_dict']._metadata
    assert not [f for f in os.listdir(str(tmp_path)) if f.endswith('.tmp')]


def test_checkpoint_index_follows_the_weights_file(tmp_path, monkeypatch):
    """The index names a checkpoint only once its weights file is complete, and the rotated-out
    file goes after that: a write that dies leaves the previous index and every listed file
    (round-3 advice); load_weights falls back to the newest listed file that exists."""
    import torch
    from segan_pytorch_amd.models.core import Saver
    net = torch.nn.Linear(3, 2)
    sv = Saver(net, str(tmp_path), max_ckpts=1, prefix='X-')
    for step in range(3):
        sv.save('Net', step)
    sv.wait()
    idx0 = json.load(open(str(tmp_path / 'X-checkpoints')))
    files0 = sorted(os.listdir(str(tmp_path)))
    real_save = torch.save

    def dying_save(obj, path, *a, **k):
        raise OSError('disk full')
    monkeypatch.setattr(torch, 'save', dying_save)
    sv.save('Net', 3)
    with pytest.raises(OSError):
        sv.wait()
    monkeypatch.setattr(torch, 'save', real_save)
    assert json.load(open(str(tmp_path / 'X-checkpoints'))) == idx0
    assert sorted(os.listdir(str(tmp_path))) == files0          # nothing deleted, nothing half-written
    assert Saver(torch.nn.Linear(3, 2), str(tmp_path), prefix='X-').load_weights()
    # an index whose `current` is missing (written by the reference's order of operations)
    idx = dict(idx0, current='X-Net-99.ckpt', latest=idx0['latest'] + ['X-Net-99.ckpt'])
    with open(str(tmp_path / 'X-checkpoints'), 'w') as f:
        json.dump(idx, f)
    assert Saver(torch.nn.Linear(3, 2), str(tmp_path), prefix='X-').load_weights()


@pytest.mark.parametrize('async_save', [False, True])
def test_resaving_the_same_step_names_keeps_every_new_file(tmp_path, async_save):
    """A second run in the same save_path restarts at iteration 1 (model.py:271), so the
    '<name>-<step>' checkpoint names repeat.  The rotated-out file is deleted AFTER the new one is
    written (round-3 order), so a checkpoint must never be its own victim (round-4 advice): after
    the re-run `current` names a file that exists and that holds the NEW weights."""
    from segan_pytorch_amd.models.core import Saver
    net = torch.nn.Linear(3, 2)
    sv = Saver(net, str(tmp_path), max_ckpts=3, prefix='EOE_G-', async_save=async_save)
    for step in (101, 201, 301, 401, 501):
        sv.save('Net', step)
    sv.wait()
    net2 = torch.nn.Linear(3, 2)
    sv2 = Saver(net2, str(tmp_path), max_ckpts=3, prefix='EOE_G-', async_save=async_save)
    for step in (101, 201, 301, 401, 501, 601):
        with torch.no_grad():
            net2.weight.fill_(float(step))
        sv2.save('Net', step)
        sv2.wait()
        idx = json.load(open(str(tmp_path / 'EOE_G-checkpoints')))
        cur = str(tmp_path / ('weights_' + idx['current']))
        assert idx['current'] == 'EOE_G-Net-{}.ckpt'.format(step) and os.path.exists(cur)
        assert float(torch.load(cur)['state_dict']['weight'][0, 0]) == float(step)
        assert len(idx['latest']) == len(set(idx['latest'])) <= 4
        for n in idx['latest']:
            assert os.path.exists(str(tmp_path / ('weights_' + n))), n
    probe = torch.nn.Linear(3, 2)
    assert Saver(probe, str(tmp_path), prefix='EOE_G-').load_weights()
    assert float(probe.weight.detach()[0, 0]) == 601.0


def test_reserved_slots_setter_round_trips():
    """segan_set_reserved_slots (ABI v13): process-wide, clamps to [0, 512], returns the previous
    value; 0 by default (a static reserve costs more than the overlap it protects)."""
    from segan_pytorch_amd import ops
    first = ops.set_reserved_slots(32)
    try:
        assert first == int(os.environ.get('SEGAN_RESERVED_SLOTS', '0'))
        assert ops.set_reserved_slots(-5) == 32
        assert ops.set_reserved_slots(100000) == 0
        assert ops.set_reserved_slots(0) == 512
    finally:
        ops.set_reserved_slots(first)
    hdr = open(os.path.join(ROOT, 'include', 'segan_hip.h')).read()
    assert 'int segan_set_reserved_slots(int n);' in hdr


def test_accumulation_mode_switch():
    """ops.set_accumulation: 'plain' (default) / 'blocked' select SEGAN_PREC_FP32 /
    SEGAN_PREC_FP32_BLOCKED for the fp32 forward / data-gradient entry points."""
    from segan_pytorch_amd import ops
    assert ops.get_accumulation() == 'plain' and ops._fp32() == ops.PREC_FP32 == 0
    ops.set_accumulation('blocked')
    try:
        assert ops.get_accumulation() == 'blocked' and ops._fp32() == ops.PREC_FP32_BLOCKED == 4
        with pytest.raises(ValueError):
            ops.set_accumulation('kahan')
        assert ops.get_accumulation() == 'blocked'
    finally:
        ops.set_accumulation('plain')
    hdr = open(os.path.join(ROOT, 'include', 'segan_hip.h')).read()
    assert '#define SEGAN_PREC_FP32_BLOCKED 4' in hdr


def test_purge_checkpoints_keeps_the_newest(tmp_path):
    """purge_ckpts.py (reference purge_ckpts.py:7-29): after four saves only the last checkpoint
    and a one-entry index remain, the survivor still loads, and an index that lists a missing
    file is refused before anything is deleted."""
    from segan_pytorch_amd.models.core import Saver, purge_checkpoints
    net = torch.nn.Linear(3, 2)
    sv = Saver(net, str(tmp_path), max_ckpts=5, prefix='EOE_G-')
    for step in range(4):
        sv.save('Generator', step)
    sv.wait()
    (tmp_path / 'weights_EOE_G-Generator-9.ckpt.tmp').write_bytes(b'partial')
    removed = purge_checkpoints(str(tmp_path), verbose=False)
    assert len(removed) == 4
    left = sorted(n for n in os.listdir(str(tmp_path)) if n.startswith('weights_'))
    assert left == ['weights_EOE_G-Generator-3.ckpt']
    idx = json.load(open(str(tmp_path / 'EOE_G-checkpoints')))
    assert idx['latest'] == ['EOE_G-Generator-3.ckpt'] and idx['current'] == 'EOE_G-Generator-3.ckpt'
    assert Saver(torch.nn.Linear(3, 2), str(tmp_path), prefix='EOE_G-').load_weights()
    sv.save('Generator', 4)
    sv.wait()
    os.unlink(str(tmp_path / 'weights_EOE_G-Generator-3.ckpt'))
    with pytest.raises(FileNotFoundError):
        purge_checkpoints(str(tmp_path), verbose=False)
    assert os.path.exists(str(tmp_path / 'weights_EOE_G-Generator-4.ckpt'))
