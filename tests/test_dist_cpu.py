"""Data-parallel plumbing on CPU: two gloo ranks (world_size 2) exercise the gradient
all-reduce on the flat arena, parameter broadcast and batch sharding — the same code
paths RCCL runs on the GPUs (segan_pytorch_amd/distributed.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from segan_pytorch_amd import distributed as sdist
    from segan_pytorch_amd import optim as soptim
    from segan_pytorch_amd.models import Generator
    rk, ws, _ = sdist.init_from_env(backend='gloo')
    assert (rk, ws) == (rank, world) and sdist.is_dist()
    torch.manual_seed(100 + rank)                 # replicas start different on purpose
    g = Generator(1, [4, 8], 31, [4, 4], z_dim=8, skip_merge='concat', bias=True)
    sdist.broadcast_params(g, src=0)
    opt = soptim.RMSprop(g.parameters(), lr=1e-3)
    # every rank writes a rank-dependent gradient; the mean must come back everywhere
    for i, p in enumerate(g.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    sdist.allreduce_grads(opt)
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1)))
             for i, p in enumerate(g.parameters()))
    flat = torch.cat([p.detach().reshape(-1) for p in g.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    batch = torch.arange(12.).view(6, 2)
    shard = sdist.shard_batch(batch)
    q.put((rank, ok, same, shard[:, 0].tolist()))
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_broadcast_shard():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], 'gradient mean wrong'
    assert res[0][2] and res[1][2], 'parameters differ after broadcast'
    assert res[0][3] == [0.0, 2.0, 4.0] and res[1][3] == [6.0, 8.0, 10.0]


def test_single_process_is_a_noop():
    from segan_pytorch_amd import distributed as sdist
    assert sdist.world_size() == 1 and sdist.rank() == 0
    t = torch.ones(4)
    assert sdist.allreduce_mean_(t) is t and sdist.shard_batch(t) is t
