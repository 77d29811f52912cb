"""Diagnostic: per-step wall time (device-synchronised) of bf16 steps with the sampled kernel timer,
after an fp32 phase in the same process (the `other_precisions` leg of bench.py)."""
import sys, time, os, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from segan_pytorch_amd import ops
dev = torch.device('cuda', 0)
wl = bench.make_workload('segan_plus', False, dev, 0, 300, False)
for _ in range(5):
    wl.one_step()
torch.cuda.synchronize()
wl.model.G.cancel_z_prefetch(); wl.model.G.z_prefetch = False
wl.model.G.z_generator = torch.Generator(device=dev).manual_seed(0)
for prec in ('bf16x3', 'bf16'):
    ops.set_precision(prec)
    for mode in ('timer', 'notimer', 'timer', 'timer'):
        t = None
        if mode == 'timer':
            t = bench.KernelTimer(); t.install(); t.active = False
        for i in range(2):
            if t is not None and i == 1:
                n = t.prepare(wl.one_step, 3)
            else:
                wl.one_step()
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        per = []
        samp = bench.sample_steps(10)
        for i in range(10):
            if t is not None:
                t.begin_step(i in samp)
            t0 = time.perf_counter()
            wl.one_step()
            h = time.perf_counter() - t0
            torch.cuda.synchronize()
            per.append((round(1e3 * h, 1), round(1e3 * (time.perf_counter() - t0), 1)))
        gc.enable()
        if t is not None:
            t.uninstall()
            s = t.finish()
            print(prec, mode, 'events/step', n, {k: (round(v['ms_per_step'], 2), v['launches_per_step']) for k, v in s.items()})
        print(prec, mode, per)
