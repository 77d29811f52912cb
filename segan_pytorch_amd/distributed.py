"""Data-parallel plumbing: one process per GPU, RCCL (backend 'nccl' on ROCm) over
xGMI.  The reference is single-process (README.md:79, "Multi-GPU is not supported
yet"); this is the new data-parallel path SURVEY.md section 8(e) specifies:

* every rank holds a full replica of G and D and its own optimizer state;
* the batch is sharded across ranks (rank r trains on its own 1/world slice);
* gradients are averaged on the flat gradient arena of each optimizer (D: 25.8 M floats
  between the D backward passes and ``Dopt.step``; G: 64.8 M floats between the G backward
  and ``Gopt.step``) in BUCKETS of consecutive parameters: the backward pass reports every
  layer whose gradients are final (``grad_ready``) and a bucket's all-reduce is issued — async,
  on RCCL's own stream — as soon as its last parameter is, so the collective of the decoder /
  the deep discriminator layers (the big weights, finished first) runs under the rest of the
  backward pass; ``allreduce_grads`` waits for the buckets and applies the 1/world scale, so
  every rank applies the gradient of the mean loss over the global batch;
* BatchNorm statistics stay local to a rank (standard DDP semantics: each replica is
  the reference at its per-GPU batch size).

On CPU tensors (gloo) the same code runs with torch arithmetic, which is what the
world_size-2 tests exercise.
"""
import os
import time

import torch
import torch.distributed as dist

from . import ops


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _collectives_on():
    """Do the gradient collectives run?  Yes with more than one rank; SEGAN_DP_SINGLE=1 (a test
    hook) also issues them in an initialised ONE-rank group, which is how the RCCL code path —
    async bucket all-reduces from inside the backward, the waits, the scale — is exercised on a
    one-GPU box (tests/test_gpu_dist.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('SEGAN_DP_SINGLE') == '1'


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialise the default process group from the torchrun environment
    (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT).  No-op when
    WORLD_SIZE is absent or 1.  Returns (rank, world_size, local_rank)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    lr = int(os.environ.get('LOCAL_RANK', '0'))
    if ws <= 1 and os.environ.get('SEGAN_DP_SINGLE') != '1':
        return 0, 1, lr
    # testing hooks: SEGAN_DIST_BACKEND=gloo and SEGAN_LOCAL_DEVICE=<i> let several ranks share
    # one GPU (RCCL refuses duplicate devices); never set in production
    backend = backend or os.environ.get('SEGAN_DIST_BACKEND')
    if 'SEGAN_LOCAL_DEVICE' in os.environ:
        lr = int(os.environ['SEGAN_LOCAL_DEVICE'])
    # before any communicator exists: threads created from here on (RCCL's proxies, the z-draw thread,
    # loader workers) inherit this rank's CPU slice
    pin_host_threads(int(os.environ.get('LOCAL_RANK', lr)), int(os.environ.get('LOCAL_WORLD_SIZE', ws)))
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(lr)
            # RCCL's kernels on a HIGH-PRIORITY stream: the contraction kernels fill every CU (3
            # workgroups of 4 waves each), so a bucket's all-reduce can only start where a compute
            # workgroup retires; with priority the hardware dispatcher hands the first freed slots
            # to RCCL's few workgroups (one per channel) instead of the next compute workgroups
            opts = None
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
            except Exception:       # pragma: no cover - option absent in this torch build
                opts = None
            kw = {'pg_options': opts} if opts is not None else {}
            dist.init_process_group(backend, rank=rk, world_size=ws,
                                    device_id=torch.device('cuda', lr), **kw)
        else:
            dist.init_process_group(backend, rank=rk, world_size=ws)
    _init_native(rk, ws)
    return rk, ws, lr


_host_pin = None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


def host_topology(cpus):
    """{cpu: (numa node, package, core id)} of the given logical CPUs from sysfs (None entries where the
    kernel does not say).  Linux enumerates the physical cores first and their SMT siblings after
    them on most two-socket hosts (cpu 0-127 = cores, 128-255 = siblings), so nothing about the
    topology can be read off the CPU numbers themselves (round-5 advice)."""
    import glob
    topo = {}
    for c in cpus:
        base = '/sys/devices/system/cpu/cpu{}'.format(c)
        node = None
        for nd in glob.glob(base + '/node[0-9]*'):
            node = int(os.path.basename(nd)[4:])
        topo[c] = (node, _read_int(base + '/topology/physical_package_id'),
                   _read_int(base + '/topology/core_id'))
    return topo


def gpu_numa_nodes(n):
    """NUMA node of local GPUs 0..n-1 (/sys/bus/pci/devices/<bdf>/numa_node), None where unknown."""
    out = []
    for i in range(n):
        node = None
        try:
            pr = torch.cuda.get_device_properties(i)
            bdf = '{:04x}:{:02x}:{:02x}.0'.format(pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            node = _read_int('/sys/bus/pci/devices/{}/numa_node'.format(bdf))
            if node is not None and node < 0:
                node = None
        except Exception:
            node = None
        out.append(node)
    return out


def plan_host_slices(local_world, cpus, topo=None, gpu_nodes=None):
    """Disjoint CPU lists, one per local rank — or None when there are fewer CPUs than ranks.

    With the topology known the unit is the PHYSICAL core (all its SMT siblings go to one rank: two
    ranks never share a core's execution units) and a rank whose GPU reports a NUMA node gets cores of
    that node (its pinned H2D staging buffers, the z draw and the loader workers stay on the socket
    the GPU hangs off); the ranks of one node split its cores evenly.  Where a node has fewer cores
    than ranks that want it, or the topology is unknown, the allowed CPUs are ordered by (node,
    package, core, cpu) and cut into equal runs — whole cores when there are enough, logical CPUs
    otherwise.  Pure function of its arguments (tests/test_dist_cpu.py)."""
    cpus = sorted(cpus)
    if local_world < 1 or len(cpus) < local_world:
        return None
    topo = topo or {}

    def key(c):
        n, p, k = topo.get(c, (None, None, None))
        return (n if n is not None else 1 << 20, p if p is not None else 1 << 20,
                k if k is not None else c, c)

    cores = {}
    for c in sorted(cpus, key=key):
        n, p, k = topo.get(c, (None, None, None))
        cores.setdefault((n, p, k if k is not None else ('cpu', c)), []).append(c)
    core_list = list(cores.items())          # ordered by node, package, core
    gpu_nodes = list(gpu_nodes) if gpu_nodes is not None else [None] * local_world
    gpu_nodes = (gpu_nodes + [None] * local_world)[:local_world]
    if all(n is not None for n in gpu_nodes):
        by_node = {}
        for r, n in enumerate(gpu_nodes):
            by_node.setdefault(n, []).append(r)
        out, ok = [None] * local_world, True
        for n, ranks in by_node.items():
            mine = [cs for (nn, _p, _k), cs in core_list if nn == n]
            if len(mine) < len(ranks):
                ok = False
                break
            per = len(mine) // len(ranks)
            for j, r in enumerate(ranks):
                out[r] = sorted(c for cs in mine[j * per:(j + 1) * per] for c in cs)
        if ok:
            return out
    if len(core_list) >= local_world:
        per = len(core_list) // local_world
        return [sorted(c for _k, cs in core_list[r * per:(r + 1) * per] for c in cs)
                for r in range(local_world)]
    flat = [c for _k, cs in core_list for c in cs]
    per = len(flat) // local_world
    return [sorted(flat[r * per:(r + 1) * per]) for r in range(local_world)]


def _ranges(cpus):
    """'0-15,128-143' for a sorted CPU list."""
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else '{}-{}'.format(cpus[i], cpus[j]))
        i = j + 1
    return ','.join(out)


def pin_host_threads(local_rank, local_world):
    """One slice of the host's cores per rank of this node (round-4 review, item 7): N ranks that
    each start torch's default intra-op pool (= all cores) and a z-draw thread oversubscribe the
    host N times over, and the single-threaded randn of the next z (~25 ms for 300 x 1024 x 16)
    then competes with 8 x 128 idle-spinning OpenMP threads.  The slices come from the real topology
    (`plan_host_slices`: whole physical cores, on the NUMA node of the rank's GPU where sysfs names
    it; round-5 advice — contiguous CPU ranges put rank 4 of an SMT host on the hyperthreads of rank
    0's cores, on the other socket from its GPU).  EVERY thread the process already has is bound
    (/proc/self/task: an OpenMP pool or runtime helper threads started before this call keep the old
    mask under a plain sched_setaffinity(0, ...)), threads created later inherit the mask — which is
    why `init_from_env` calls this BEFORE the process group (RCCL's proxy threads) exists.  torch's
    intra-op pool is sized to the slice's physical cores (at most 16: the host side of a step is
    launches and one randn).  `SEGAN_NO_PIN=1` leaves the process alone.  Returns what was done
    (bench.py's `comm.host.pinning` carries it, CPU list included), or None."""
    global _host_pin
    if local_world <= 1 or os.environ.get('SEGAN_NO_PIN') == '1' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        topo = host_topology(cpus)
        nodes = gpu_numa_nodes(local_world) if torch.cuda.is_available() and \
            'SEGAN_LOCAL_DEVICE' not in os.environ else None
        plan = plan_host_slices(local_world, cpus, topo, nodes)
        if plan is None or not plan[local_rank]:
            _host_pin = {'skipped': '{} CPUs for {} ranks'.format(len(cpus), local_world)}
            return _host_pin
        mine = plan[local_rank]
        tids = [0]
        try:
            tids = [int(t) for t in os.listdir('/proc/self/task')] or [0]
        except Exception:
            pass
        bound = 0
        for tid in tids:
            try:
                os.sched_setaffinity(tid, mine)
                bound += 1
            except Exception:       # a thread that exited meanwhile
                pass
        ncores = len({topo[c][1:] if topo[c][2] is not None else c for c in mine})
        torch.set_num_threads(max(1, min(ncores, 16)))
        _host_pin = {'cpus': _ranges(mine), 'logical_cpus': len(mine), 'physical_cores': ncores,
                     'numa_node': topo[mine[0]][0], 'gpu_numa_node': nodes[local_rank] if nodes else None,
                     'threads_bound': bound, 'torch_threads': torch.get_num_threads()}
    except Exception as e:      # pragma: no cover - a container that forbids it
        _host_pin = {'error': repr(e)}
    return _host_pin


def host_pin():
    return _host_pin


# ---- the exchange through libsegan_hip's own RCCL communicators (SEGAN_COMM=native) -----------
_native = None          # ops.Comm of the gradient buckets: used on _native_stream ONLY
_native_cur = None      # ops.Comm of everything issued on the CURRENT stream (synchronised
                        # BatchNorm, allreduce_mean_, broadcast_params)
_native_stream = None   # side stream the bucket all-reduces run on


def native_comm():
    """The library-owned RCCL communicator of the gradient buckets (C ABI: segan_comm_init /
    segan_allreduce / segan_comm_destroy) when SEGAN_COMM=native, else None.  torch.distributed
    then only carries the 128-byte rendezvous ids (and the host-side barriers of the launch
    scripts); gradients, initial weights and synchronised-BatchNorm statistics travel through
    the C ABI.  TWO communicators exist in that mode, one per stream that issues collectives:
    the bucket all-reduces run on a side stream under the backward pass while the Sync-BN
    exchanges of that same backward run on the compute stream — one communicator on two streams
    would leave their relative order to RCCL's internal serialisation, which has to be identical
    on every rank (round-3 advice); with a communicator per stream each one sees one program
    order."""
    return _native


def _init_native(rk, ws, force=False):
    global _native, _native_cur, _native_stream
    if _native is not None or not torch.cuda.is_available():
        return
    if not force and os.environ.get('SEGAN_COMM') != 'native':
        return
    ident = [(ops.comm_unique_id(), ops.comm_unique_id()) if rk == 0 else None]
    if ws > 1:
        dist.broadcast_object_list(ident, src=0)
    _native = ops.Comm(ws, rk, ident[0][0])
    _native_cur = ops.Comm(ws, rk, ident[0][1])
    _native_stream = torch.cuda.Stream(priority=-1)     # high priority: see init_from_env


def set_native(on):
    """Switch the data path between the library's own communicators and torch.distributed at
    run time (bench.py times both in one process); creating them is a collective over all
    ranks, so every rank must make the same call."""
    if on:
        _init_native(rank(), world_size(), force=True)
    else:
        destroy_native()
    _reducers.clear()


def destroy_native():
    global _native, _native_cur, _native_stream
    if _native is not None:
        torch.cuda.synchronize()
        _native.destroy()
        _native_cur.destroy()
        _native, _native_cur, _native_stream = None, None, None


import atexit
atexit.register(destroy_native)


def allreduce_mean_(flat):
    """In-place mean over ranks of a flat fp32 tensor (one collective)."""
    ws = world_size()
    if ws <= 1:
        return flat
    if _native_cur is not None and flat.is_cuda:
        return _native_cur.allreduce(flat, 1.0 / ws)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if flat.is_cuda:
        ops.scale_(flat, 1.0 / ws)
    else:
        flat.mul_(1.0 / ws)
    return flat


_bucket_bytes = int(float(os.environ.get('SEGAN_DP_BUCKET_MB', '16')) * (1 << 20))


def set_bucket_bytes(n):
    """Target size of a gradient bucket (default 16 MiB, SEGAN_DP_BUCKET_MB: D = {fc} 16 MB, {enc4} 62, {enc3, enc2} 19, {enc1, enc0} 1; G = {dec1..dec4} 41, {dec0} 124, {enc4, alphas} 62, {enc3, enc2} 19, {enc1, enc0} 1); a parameter is
    never split, so the big weights form buckets of their own."""
    global _bucket_bytes
    _bucket_bytes = int(n)
    _reducers.clear()


class GradReducer(object):
    """Bucketed gradient averaging of ONE optimizer's flat arena, overlapped with the backward
    pass that produces the gradients (module docstring)."""

    def __init__(self, optimizer, bucket_bytes):
        self.opt = optimizer
        self.buckets = []                 # (first float, end float)
        self.members = []                 # number of parameters per bucket
        self.bucket_of = {}
        # Buckets are runs of consecutive parameters built from the LAST parameter backwards: the
        # backward passes finish the layers in reverse registration order (G: dec4 .. dec0, enc4 ..
        # enc0; D: fc, enc4 .. enc0), so a run that ENDS at a late layer closes as soon as its
        # earliest member is done, and what is left for the end of the pass — the only part of the
        # exchange nothing can hide — is the small early layers (D: enc1 + enc0, 1 MB; built
        # front to back the first bucket was enc0 .. enc4 = 82 MB of D's 98, all of it exposed on the
        # critical path before Dopt.step, model.py:308)
        params, offs = optimizer._params, optimizer._offsets
        i = len(params) - 1
        while i >= 0:
            hi = offs[i + 1] if i + 1 < len(params) else optimizer._total
            nb, j = 0, i
            while True:
                nb += params[j].numel() * 4
                if nb >= bucket_bytes or j == 0:
                    break
                j -= 1
            for k in range(j, i + 1):
                self.bucket_of[id(params[k])] = len(self.buckets)
            self.buckets.append((offs[j], hi))
            self.members.append(i - j + 1)
            i = j - 1
        self.armed = False
        self.pending, self.sent, self.works = [], [], []
        self.wait_events, self.wait_host_s, self.finishes = [], 0.0, 0

    def arm(self, passes=1):
        """Call before the LAST backward() that adds to these gradients.  `passes`: how many
        times that backward runs the network's node — WSEGAN's summed discriminator loss holds
        2 to 4 D forwards (model.py:577-631), so every D parameter is written (and reported)
        that many times and a bucket may only leave after the LAST report of its members."""
        self.opt._resync()
        self.passes = int(passes)
        self.pending = [n * self.passes for n in self.members]
        self.sent = [False] * len(self.buckets)
        self.seen = {}
        self.works = []
        self.armed = True

    def _send(self, b):
        lo, hi = self.buckets[b]
        flat = self.opt.flat_grad
        if _native is not None and flat.is_cuda:
            # C-ABI path: the bucket's sum AND its 1/world scale on the side stream, behind the
            # kernels that produced it (event), under the rest of the backward pass
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            _native_stream.wait_event(ev)
            _native.allreduce(flat[lo:hi], 1.0 / world_size(), stream=_native_stream)
            self.native = True
        else:
            self.works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.sent[b] = True

    def ready(self, p):
        b = self.bucket_of.get(id(p))
        if b is None or not self.armed or self.sent[b] or self.seen.get(id(p), 0) >= self.passes:
            return
        self.seen[id(p)] = self.seen.get(id(p), 0) + 1
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._send(b)

    def finish(self):
        """All buckets reduced (those never reported ready are sent now), then the mean."""
        if not self.armed:
            self.arm()
        flat = self.opt.flat_grad
        t0 = time.perf_counter()
        e0 = None
        if _profile and flat.is_cuda:
            # how long the COMPUTE stream sits here for the collectives (device time between
            # these two events): ~0 when the buckets finished under the backward pass
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        late = 0
        lb = self.__dict__.setdefault('late_by_bucket', [0] * len(self.buckets))
        for b in range(len(self.buckets)):
            if not self.sent[b]:
                self._send(b)
                late += 1
                if _profile:
                    lb[b] += 1
        for w in self.works:
            w.wait()
        self.armed = False
        self.works = []
        if getattr(self, 'native', False):
            done = torch.cuda.Event()
            done.record(_native_stream)
            torch.cuda.current_stream().wait_event(done)     # the optimizer step follows
            self.native = False                              # already scaled per bucket
        elif flat.is_cuda:
            ops.scale_(flat, 1.0 / world_size())
        else:
            flat.mul_(1.0 / world_size())
        if _profile:
            self.finishes += 1
            self.late_buckets = getattr(self, 'late_buckets', 0) + late
            self.wait_host_s += time.perf_counter() - t0
            if e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.wait_events.append((e0, e1))


_reducers = {}
_active = None
_profile = False


def set_profile(on):
    """Collect per-reducer wait statistics (comm_stats); off by default."""
    global _profile
    _profile = bool(on)
    for r in _reducers.values():
        r.wait_events, r.wait_host_s, r.finishes, r.late_buckets = [], 0.0, 0, 0
        r.late_by_bucket = [0] * len(r.buckets)


def comm_stats():
    """What the gradient exchange cost since set_profile(True), per optimizer arena (in creation
    order: SEGAN.build_optimizers makes G's first): number of buckets and their sizes, how many
    all-reduce rounds (`finishes`), how many buckets were only sent at finish() (not overlapped
    with the backward), the device time the compute stream waited in finish() and the host time
    spent there.  Synchronises the device."""
    out = []
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    for r in _reducers.values():
        out.append({
            'arena_floats': int(r.opt._total),
            'buckets': len(r.buckets),
            'bucket_mb': [round((hi - lo) * 4 / 2 ** 20, 2) for lo, hi in r.buckets],
            'finishes': r.finishes,
            'late_buckets': getattr(r, 'late_buckets', 0),
            'late_by_bucket': list(getattr(r, 'late_by_bucket', [0] * len(r.buckets))),
            'wait_device_ms': sum(a.elapsed_time(b) for a, b in r.wait_events),
            'wait_host_ms': 1e3 * r.wait_host_s,
        })
    return {'backend': 'native (libsegan_hip RCCL communicators)' if _native is not None else
            'torch.distributed ({})'.format(dist.get_backend() if dist.is_initialized() else 'none'),
            'sync_bn': sync_bn_enabled(), 'bucket_target_mb': _bucket_bytes / 2 ** 20, 'arenas': out,
            # what bounds RCCL's footprint beside the contraction kernels (the
            # measured 1.9 % is for 32 channels = 32 workgroups of 256 threads): whatever the
            # environment sets is recorded with the line
            'rccl_env': {k: v for k, v in os.environ.items()
                         if k.startswith(('NCCL_', 'RCCL_')) and 'CHANNEL' in k or k in ('NCCL_ALGO', 'NCCL_PROTO')}}


def _reducer(optimizer):
    r = _reducers.get(id(optimizer))
    if r is None or r.opt is not optimizer:
        r = GradReducer(optimizer, _bucket_bytes)
        _reducers[id(optimizer)] = r
    return r


def drop_reducer(optimizer):
    """Forget the reducer of an optimizer that is going away (it holds the optimizer — and through it
    the flat parameter / gradient / state arenas — alive)."""
    global _active
    r = _reducers.pop(id(optimizer), None)
    if r is not None and _active is r:
        _active = None


def arm(optimizer, passes=1):
    """Announce that the next backward() is the last one into `optimizer`'s gradients before
    its step: from here on ``grad_ready`` starts the all-reduce of every bucket that completes.
    `passes` = number of forwards of the network that backward() differentiates (each one
    reports every parameter once; a bucket leaves after the last report)."""
    global _active
    if not _collectives_on():
        return
    _active = _reducer(optimizer)
    _active.arm(passes)


def grad_ready(*params):
    """Called by the backward passes (functional.py) once the kernels that write the gradients
    of `params` are enqueued and nothing later adds to them."""
    if _active is None:
        return
    for p in params:
        if p is not None:
            _active.ready(p)


def allreduce_grads(optimizer):
    """Average the gradients of every parameter the optimizer owns across ranks (waits for the
    buckets already in flight, sends the rest)."""
    global _active
    if not _collectives_on():
        return
    r = _reducer(optimizer)
    if _active is not None and _active is not r:
        _active.armed = False
    r.finish()
    _active = None


def broadcast_params(module, src=0):
    """Make every replica start from rank `src`'s weights and buffers."""
    if not _collectives_on():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if _native_cur is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
                _native_cur.broadcast(t.data, src)
            else:
                dist.broadcast(t.data, src)
    ops.bump_weights_epoch()


def broadcast_scalar(value, src=0, device=None):
    """Rank `src`'s python float on every rank (the validation objective: every rank must take
    the same early-stopping / best-checkpoint branch, or the ranks that go on hang in the next
    gradient all-reduce).  Identity without data parallelism."""
    if not is_dist():
        return float(value)
    if device is None or dist.get_backend() != 'nccl':
        device = torch.device('cpu') if dist.get_backend() != 'nccl' else \
            torch.device('cuda', torch.cuda.current_device())
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.broadcast(t, src)
    return float(t.item())


def shard_batch(t, dim=0):
    """This rank's contiguous slice of a global batch."""
    ws, rk = world_size(), rank()
    if ws <= 1:
        return t
    n = t.shape[dim]
    if n % ws != 0:
        raise ValueError('global batch {} is not divisible by world size {}'.format(n, ws))
    per = n // ws
    return t.narrow(dim, rk * per, per)


# ---- synchronised BatchNorm (optional) --------------------------------
def sync_bn_enabled():
    """SEGAN_SYNC_BN=1 under data parallelism: D's BatchNorm statistics (forward) and the
    per-channel gradient sums (backward) are taken over the GLOBAL batch, so N ranks of batch
    b reproduce one process at batch N*b.  Default off: every replica normalises with its own
    batch, like the reference at its per-GPU batch size."""
    return _collectives_on() and os.environ.get("SEGAN_SYNC_BN", "0") == "1"


def bn_stats_sync(x, gamma, beta, eps, momentum, running_mean, running_var):
    ws = ops.bn_partial(x)                                   # [nsplit, C, 3]
    if _native_cur is not None:
        allp = _native_cur.allgather(ws).view(-1, ws.shape[1], 3)
    else:
        parts = [torch.empty_like(ws) for _ in range(world_size())]
        dist.all_gather(parts, ws)
        allp = torch.cat(parts, 0).contiguous()
    return ops.bn_final(allp, gamma, beta, eps, momentum, running_mean, running_var)


def act_bwd_bn_sync(a, dh, slope, bn, dslope=None, dgamma=None, dbeta=None, dbias=None):
    totals, ws = ops.act_bwd_bn_reduce(a, dh, slope, bn, dslope, dgamma, dbeta)
    if _native_cur is not None:
        _native_cur.allreduce(totals, 1.0)
    else:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    count = float(a.shape[0]) * float(a.shape[2]) * world_size()
    return ops.act_bwd_bn_apply(a, dh, slope, bn, totals, count, dbias, ws)
