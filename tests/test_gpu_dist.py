"""RCCL on the GPU box.  A one-GPU box cannot host two RCCL ranks, so the data-parallel path is
driven at world size 1 with the collectives forced on (segan_pytorch_amd/distributed.py,
SEGAN_DP_SINGLE): communicator set-up, bucketed async all-reduces issued from inside the backward
passes, the waits before the optimizer steps and the 1/world scale all run on RCCL, and — the mean
over one rank being the identity — the weights after two GAN steps must equal the plain run's
bit for bit (deterministic reductions).  world_size-2 semantics are covered on CPU with gloo
(tests/test_dist_cpu.py)."""
import os
import subprocess
import sys

import pytest
import torch

HELPER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'dp_single_rank.py')


def _run(mode, path):
    r = subprocess.run([sys.executable, HELPER, mode, path], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(path, map_location='cpu', weights_only=False)


@pytest.mark.gpu
def test_rccl_single_rank_step_equals_plain_step(tmp_path):
    plain = _run('plain', str(tmp_path / 'plain.pt'))
    rccl = _run('rccl', str(tmp_path / 'rccl.pt'))
    assert rccl['info']['backend'] == 'nccl' and rccl['info']['world'] == 1
    # 8 MiB buckets: the big weights are buckets of their own, ~10 all-reduces per step
    assert rccl['info']['all_reduce_calls'] >= 2 * 8, rccl['info']
    for k, v in plain['sd'].items():
        assert torch.equal(v, rccl['sd'][k]), k
    assert plain['info']['losses'] == rccl['info']['losses']


@pytest.mark.gpu
def test_weight_gradients_on_the_side_stream_change_nothing(tmp_path):
    """SEGAN_WGRAD_OVERLAP=1 (functional._SideStream: the weight gradients of a backward pass are
    launched on a second stream beside the data-gradient chain, opt-in): the same kernels on the
    same data in deterministic mode, so the weights after two GAN steps equal the one-stream run's
    bit for bit — alone and under the RCCL gradient reducer, whose gradient-ready reports then
    come from the side stream."""
    plain = _run('plain', str(tmp_path / 'plain.pt'))
    for mode in ('overlap', 'rccl_overlap'):
        got = _run(mode, str(tmp_path / (mode + '.pt')))
        for k, v in plain['sd'].items():
            assert torch.equal(v, got['sd'][k]), (mode, k)
        assert plain['info']['losses'] == got['info']['losses']


@pytest.mark.gpu
def test_native_comm_single_rank_step_equals_plain_step(tmp_path):
    """SEGAN_COMM=native: the gradient buckets travel through libsegan_hip's OWN communicator
    (C ABI segan_comm_init / segan_allreduce / segan_comm_destroy; RCCL bound at run time), issued
    on a side stream from inside the backward passes with the 1/world scale folded in; at world
    size 1 the weights after two steps must equal the plain run's bit for bit, and no gradient
    may have gone through torch.distributed."""
    plain = _run('plain', str(tmp_path / 'plain.pt'))
    nat = _run('native', str(tmp_path / 'native.pt'))
    assert nat['info']['native_all_reduce_calls'] >= 2 * 8, nat['info']
    assert nat['info']['all_reduce_calls'] == 0, nat['info']
    # two steps x (D arena + G arena) floats went through segan_allreduce
    # (the arenas pad every parameter to a 16-byte boundary)
    assert 0 <= nat['info']['native_floats'] - 2 * (25825793 + 64770561) < 4096, nat['info']
    for k, v in plain['sd'].items():
        assert torch.equal(v, nat['sd'][k]), k
    assert plain['info']['losses'] == nat['info']['losses']


@pytest.mark.gpu
def test_native_comm_collectives(tmp_path):
    """segan_allreduce (sum + scale), segan_broadcast, segan_allgather on a one-rank communicator."""
    from segan_pytorch_amd import ops
    c = ops.Comm(1, 0, ops.comm_unique_id())
    try:
        x = torch.arange(1000, dtype=torch.float32, device='cuda')
        y = x.clone()
        c.allreduce(y, 0.5)
        assert torch.equal(y, x * 0.5)
        c.broadcast(y, 0)
        g = c.allgather(x)
        torch.cuda.synchronize()
        assert g.shape == (1, 1000) and torch.equal(g[0], x) and torch.equal(y, x * 0.5)
        with pytest.raises(RuntimeError):
            c.broadcast(y, 3)
    finally:
        c.destroy()


@pytest.mark.gpu
def test_rccl_single_rank_sync_batchnorm(tmp_path):
    plain = _run('plain', str(tmp_path / 'plain.pt'))
    sync = _run('syncbn', str(tmp_path / 'sync.pt'))
    worst = 0.0
    for k, v in plain['sd'].items():
        if not v.dtype.is_floating_point:
            assert torch.equal(v, sync['sd'][k]), k
            continue
        d = (v.double() - sync['sd'][k].double()).norm() / max(float(v.double().norm()), 1e-30)
        worst = max(worst, float(d))
    # statistics over the "global" batch of one rank: same numbers through all_gather/all_reduce,
    # summed in a different order
    assert worst < 1e-5, worst


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 8])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` (N = 2, and N = 8 with the `other_workloads` block: the command of
    the driver's scaling run) as the driver may call it (no WORLD_SIZE in the environment): the
    script re-executes itself under torch.distributed.run, both ranks initialise a process group,
    shard the batch, all-reduce their gradients from inside the backward passes and rank 0 prints
    ONE JSON line for the whole job.  A one-GPU box cannot host two RCCL ranks, so the two ranks
    share cuda:0 over gloo (SEGAN_DIST_BACKEND / SEGAN_LOCAL_DEVICE test hooks); everything else —
    the exec, the rendezvous on 127.0.0.1, the rank bookkeeping, the barriers and the max-over-ranks
    timing — is the code the 8-GPU run executes."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEGAN_DIST_BACKEND='gloo', SEGAN_LOCAL_DEVICE='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    batch = 12 if world == 2 else 8
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', '2',
                        '--warmup', '1', '--batch', str(batch), '--no-cpu-baseline', '--no-modes'] +
                       (['--no-side-workloads'] if world == 2 else ['--side-steps', '1']),
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == world and d['ranks_seen'] == list(range(world)) and d['backend'] == 'gloo'
    assert d['losses_finite'] is True and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == batch * world and d['config']['parallelism'] == 'dp{}'.format(world)
    assert abs(d['value'] - batch * world * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    if world == 8:
        # BASELINE configs 3 / 4 are 8-GPU configurations: the side workloads run on all ranks too
        ow = d['other_workloads']
        for k in ('wsegan', 'vanilla11'):
            assert 'error' not in ow[k], ow[k]
            assert ow[k]['n_gpus'] == 8 and ow[k]['losses_finite'] is True
            assert abs(ow[k]['value'] - batch * world * 1e3 / ow[k]['ms_per_step']) < 1e-6 * ow[k]['value']
    # the multi-GPU line diagnoses itself: transport, SyncBN switch, bucket layout of the two
    # gradient arenas (G 64.8 M floats, D 25.8 M), how long the compute stream waited for the
    # collectives per step and how many buckets only left at the optimizer step
    c = d['comm']
    assert c['backend'].startswith('torch.distributed (gloo') and c['sync_bn'] is False
    assert sorted(a['arena_floats'] // 1000000 for a in c['arenas']) == [25, 64]
    for a in c['arenas']:
        assert a['buckets'] == len(a['bucket_mb']) >= 2 and a['finishes'] == 2
        assert abs(sum(a['bucket_mb']) - a['arena_floats'] * 4 / 2 ** 20) < 0.006 * a['buckets']   # sizes rounded to 0.01
        assert a['late_buckets'] <= a['buckets'] * a['finishes']
        # buckets run from the last parameter backwards (the order the backward finishes them in), so
        # every bucket closes inside the backward pass: none — and in particular none of the big ones
        # (G: dec0 124 MB, enc4 62 MB; D: enc4 62 MB) — is left for the optimizer step
        assert len(a['late_by_bucket']) == a['buckets'] and sum(a['late_by_bucket']) == a['late_buckets']
        for mb, late in zip(a['bucket_mb'], a['late_by_bucket']):
            if mb > 30:
                assert late == 0, (a['bucket_mb'], a['late_by_bucket'])
        assert a['wait_device_ms_per_step'] >= 0.0 and a['wait_host_ms_per_step'] > 0.0
    # the host side of every rank, all ranks at it at once: z draw (single-threaded randn) and its H2D
    h = c['host']
    assert len(h['z_draw_ms_per_rank']) == len(h['z_h2d_ms_per_rank']) == world
    assert all(v > 0 for v in h['z_draw_ms_per_rank'] + h['z_h2d_ms_per_rank'])
    assert h['pinning'] is None or any(k in h['pinning'] for k in ('cpus', 'skipped', 'error'))
    assert d['comm_wait_ms_per_step'] == c['comm_wait_ms_per_step'] >= 0.0
    assert list(c['ms_per_step']) == ['torch.distributed']
