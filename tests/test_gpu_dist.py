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
