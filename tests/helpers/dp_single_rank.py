"""Two GAN steps of the SEGAN+ default net at batch 6 on cuda:0, either plainly or inside an
initialised ONE-rank RCCL process group with the gradient collectives forced on
(SEGAN_DP_SINGLE=1).  Writes the weights after the steps to argv[2].  Run by
tests/test_gpu_dist.py; world size 1 is all a one-GPU box can give RCCL (it refuses two ranks
on one device), and it still drives the whole RCCL path: communicator set-up, async bucket
all-reduces issued from inside the backward passes on RCCL's stream, waits, scale."""
import os
import random
import socket
import sys
from types import SimpleNamespace

mode, out = sys.argv[1], sys.argv[2]
os.environ['SEGAN_DETERMINISTIC'] = '1'
if mode.endswith('overlap'):        # 'overlap' / 'rccl_overlap': weight gradients on the side stream
    os.environ['SEGAN_WGRAD_OVERLAP'] = '1'
    mode = 'plain' if mode == 'overlap' else mode[:-len('_overlap')]
if mode != 'plain':
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.update(SEGAN_DP_SINGLE='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0',
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      SEGAN_DP_BUCKET_MB='8')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
if mode in ('syncbn', 'native_syncbn'):
    os.environ['SEGAN_SYNC_BN'] = '1'
if mode.startswith('native'):
    os.environ['SEGAN_COMM'] = 'native'      # gradients through libsegan_hip's own communicator

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from segan_pytorch_amd import distributed as sdist
from segan_pytorch_amd import losses
from segan_pytorch_amd.datasets import synthetic_pairs
from segan_pytorch_amd.models import SEGAN

rank, world, local = sdist.init_from_env()
info = {'mode': mode, 'world': world}
if mode != 'plain':
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_backend() == 'nccl', 'RCCL group not initialised'
    info['backend'] = dist.get_backend()
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
opts = bench.default_opts()
random.seed(111); np.random.seed(111); torch.manual_seed(111)
model = SEGAN(SimpleNamespace(**opts)).to(dev)
Gopt, Dopt = model.build_optimizers(SimpleNamespace(**opts))
sdist.broadcast_params(model.G)
sdist.broadcast_params(model.D)
model.G.train(); model.D.train()
clean, noisy = synthetic_pairs(6, 16384, seed=0, device=dev)
clean, noisy = clean.unsqueeze(1).contiguous(), noisy.unsqueeze(1).contiguous()
random.seed(1000); torch.manual_seed(2000)
crit = losses.MSELoss()
if mode != 'plain':
    # count the collectives RCCL is asked for
    n = {'all_reduce': 0, 'native_all_reduce': 0, 'native_floats': 0}
    real = dist.all_reduce

    def counted(*a, **k):
        n['all_reduce'] += 1
        return real(*a, **k)
    dist.all_reduce = counted
    if mode.startswith('native'):
        from segan_pytorch_amd import ops
        assert sdist.native_comm() is not None and sdist.native_comm().world == 1
        real_native = ops.Comm.allreduce

        def counted_native(self, t, *a, **k):
            n['native_all_reduce'] += 1
            n['native_floats'] += t.numel()
            return real_native(self, t, *a, **k)
        ops.Comm.allreduce = counted_native
for _ in range(2):
    ls = model.gan_step(clean, noisy, Gopt, Dopt, crit, 100.0, z=None)
torch.cuda.synchronize()
if mode != 'plain':
    info['all_reduce_calls'] = n['all_reduce']
    info['native_all_reduce_calls'] = n['native_all_reduce']
    info['native_floats'] = n['native_floats']
    dist.all_reduce = real
info['losses'] = [float(x) for x in ls]
sd = {'G.' + k: v.detach().cpu() for k, v in model.G.state_dict().items()}
sd.update({'D.' + k: v.detach().cpu() for k, v in model.D.state_dict().items()})
torch.save({'info': info, 'sd': sd}, out)
if mode != 'plain':
    sdist.destroy_native()
    dist.destroy_process_group()
