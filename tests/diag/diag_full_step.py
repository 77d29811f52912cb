import sys, os
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests'); sys.path.insert(0,ROOT+'/oracle')
import torch, random, numpy as np
from types import SimpleNamespace
from conftest import load_golden
from test_gpu_model import build, run_step
from segan_pytorch_amd.datasets import synthetic_pairs
fx=load_golden('segan_plus_b2.pt')
m=build(fx, seed=fx['seed'])
clean,noisy=synthetic_pairs(2,16384,0); clean,noisy=clean.unsqueeze(1),noisy.unsqueeze(1)
z=torch.randn(2,1024,16,generator=torch.Generator().manual_seed(0))
out,Gopt,Dopt=run_step(m,fx,clean,noisy,z)
gn=dict(m.G.named_parameters()); dn=dict(m.D.named_parameters())
def rel(t,c):
    t=t.detach().double().cpu().reshape(-1)
    got=t[c['sample_idx']].float(); den=max(c['sample'].abs().max().item(),1e-30)
    return (got-c['sample']).abs().max().item()/den, abs(t.sum().item()-c['sum'])/max(c['abs'],1e-30)
for k,c in fx['g_grads'].items():
    print('G %-32s sample_rel %.2e sum_rel %.2e'%(k,*rel(gn[k].grad,c)))
for k,c in fx['d_grads'].items():
    print('D %-32s sample_rel %.2e sum_rel %.2e'%(k,*rel(dn[k].grad,c)))
