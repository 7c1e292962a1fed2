"""Data parallelism over RCCL: one process per GPU, per-rank minibatch shards, ONE flat
all-reduce of the weight gradients and one of the architecture gradients per step.

Replaces the reference's torch.nn.DataParallel sites (mmimdb_darts_searchable.py:36-37,
ntu_darts_searchable.py:50-52, ego_darts_searchable.py:51-53).  DataParallel semantics kept:
loss = mean over the GLOBAL batch (== mean of equal-shard means), BatchNorm statistics per
replica (unsynchronised), architecture tensors shared.  Unlike DataParallel nothing is
re-broadcast per step: replicas start identical and apply identical Adam updates to the
all-reduced gradients, so they stay bit-identical.

The hooks ride on ``optimizer.register_step_pre_hook`` so the reference's unchanged
trainers (loss.backward(); optimizer.step()) pick them up.  backend 'nccl' is RCCL on ROCm;
'gloo' is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1'))


def init_from_env(backend=None):
    """torch.distributed rendezvous from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    world = env_world()
    if world <= 1:
        return 0, 0, 1
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    backend = os.environ.get('BMNAS_DIST_BACKEND', backend)       # testing hook (gloo on one GPU)
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if 'BMNAS_FORCE_DEVICE' in os.environ:                        # testing hook: all ranks on one GPU
        local = int(os.environ['BMNAS_FORCE_DEVICE'])
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def _staged(t, group=None):
    """gloo (the CPU test backend, also used to put two ranks on ONE GPU in the -m gpu tests) is
    given host copies of device tensors; RCCL works on the device buffers in place."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
    if _staged(t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.detach().copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def broadcast(t, src=0, group=None):
    if _staged(t, group):
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.detach().copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


class NativeComm:
    """RCCL through the C ABI (bmnas_comm_* / bmnas_allreduce_f32, csrc/comm.hip) instead of
    torch.distributed's process group: the collective is a plain asynchronous launch on the current
    HIP stream, which is what lets GraphedTrainStep capture it BETWEEN the backward and the Adam
    launch — fwd + bwd + all-reduce + Adam as one hipGraph replay.  The unique id travels over the
    already-initialised torch.distributed group (any backend).  Tried by default under an RCCL process group
    (FlatGradAllReducer.plan: every rank must succeed, else all of them keep the host-issued all-reduce);
    BMNAS_NATIVE_RCCL=0 turns it off."""

    _instance = None

    def __init__(self, group=None):
        from . import lib
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.comm = None
        # Rank 0 ALWAYS reaches the broadcast: it ships the unique id or, when librccl cannot be bound /
        # ncclGetUniqueId fails, an error sentinel — so the other ranks, which are waiting in the same broadcast, get
        # an answer and every rank raises together instead of being left in a collective nobody else will enter.
        msg = [None]
        if self.rank == 0:
            try:
                msg[0] = ('uid', lib.comm_get_unique_id())
            except Exception as e:                          # noqa: BLE001
                msg[0] = ('error', f'{type(e).__name__}: {e}')
        if self.world > 1:
            dist.broadcast_object_list(msg, src=0, group=group)
        kind, payload = msg[0]
        if kind != 'uid':
            raise RuntimeError(f'rank 0 could not create the RCCL unique id ({payload})')
        self.comm = lib.comm_init_rank(self.world, self.rank, payload)
        # the first collective on a communicator sets up its channels (host-side handshakes between ranks):
        # do it here, where every rank is known to be present, not inside somebody's stream capture
        warm = torch.ones(1, device='cuda')
        lib.allreduce_f32(warm, self.comm, True)
        torch.cuda.current_stream().synchronize()
        self.info = lib.comm_info(self.comm)

    @classmethod
    def get(cls, group=None):
        if cls._instance is None:
            cls._instance = cls(group)
        return cls._instance

    def all_reduce(self, flat, average=True):
        from . import lib
        lib.allreduce_f32(flat, self.comm, average)

    def destroy(self):
        from . import lib
        if self.comm is not None:
            lib.comm_destroy(self.comm)
            self.comm = None
            NativeComm._instance = None


def native_rccl_enabled():
    """Default ON: under N > 1 the trainers run the step bench.py reports — fwd + bwd + all-reduce + Adam inside one
    hipGraph replay — whenever every rank can create the C-ABI communicator; BMNAS_NATIVE_RCCL=0 keeps the
    host-issued torch.distributed all-reduce between the replay and an eager Adam launch."""
    return os.environ.get('BMNAS_NATIVE_RCCL', '1') not in ('0', '', 'false', 'False')


class _Watchdog:
    """`with _Watchdog(what):` — a rendezvous that never returns on some rank (ncclCommInitRank waiting for a peer
    that died) cannot be turned into an exception from inside: after BMNAS_COMM_WATCHDOG_S seconds (default 180) the
    process says what it was waiting for and exits with status 3, so that the launcher tears the job down instead of
    hanging forever.  Nothing is retried or re-executed from here."""

    def __init__(self, what):
        self.what, self.timer = what, None

    def __enter__(self):
        import sys
        import threading

        def bail():
            print(f'bmnas.dist: {self.what} did not finish within the watchdog; leaving (exit 3)', file=sys.stderr,
                  flush=True)
            os._exit(3)

        self.timer = threading.Timer(float(os.environ.get('BMNAS_COMM_WATCHDOG_S', '180')), bail)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


_AVG = {}


def avg_supported(device, group=None):
    """True when the backend can all-reduce with ReduceOp.AVG (RCCL / NCCL >= 2.10 can, gloo
    cannot): the mean over ranks then needs neither a pre-scaled loss nor a scale kernel.
    Probed once per process with a 1-element collective — every rank must call this."""
    key = (str(device), id(group))
    if key not in _AVG:
        ok = dist.is_initialized() and dist.get_backend(group) == 'nccl'
        if ok:
            try:
                t = torch.ones(1, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
                ok = abs(float(t) - 1.0) < 1e-6
            except Exception:                    # noqa: BLE001 — any refusal means "not supported"
                ok = False
        _AVG[key] = ok
    return _AVG[key]


def shard(t, rank, world, dim=0):
    """rank's contiguous shard of the global batch (equal shards: mean of means == global mean)."""
    n = t.shape[dim]
    if n % world != 0:
        raise ValueError(f'global batch {n} is not divisible by world size {world}')
    per = n // world
    return t.narrow(dim, rank * per, per)


def uneven_bounds(n, rank, world):
    """(start, length) of rank's slice when n samples are scattered the way nn.DataParallel does it
    (mmimdb_darts_searchable.py:36-37 -> torch.nn.parallel.scatter -> comm.scatter -> Tensor.chunk(world)):
    contiguous slices of ceil(n / world) samples until the batch is used up — 100 over 8 is 13 x 7 + 9, 9 over 8 is
    2, 2, 2, 2, 1 and three idle replicas (length 0).  Which samples share a replica matters beyond bookkeeping:
    train-mode BatchNorm takes its batch statistics per replica."""
    c = -(-n // world)
    start = min(rank * c, n)
    return start, max(0, min(c, n - start))


# The weight of THIS rank's shard in the current global batch: n_rank * world / n (1.0 for equal shards).  The
# reference's criterion sees the gathered output of all replicas and takes ONE mean over the n samples
# (mmimdb_darts_searchable.py:114 after DataParallel's gather); a rank here takes the mean over its own n_rank
# samples, so its gradient enters the average over ranks with this weight.  Set per batch by the trainer loop
# (`_loop._shard_batch`), read by every reducer.
_SHARD_WEIGHT = [1.0]


def set_shard_weight(w):
    _SHARD_WEIGHT[0] = float(w)


def shard_weight():
    return _SHARD_WEIGHT[0]


class FlatGradAllReducer:
    """Averages the .grad of a fixed tensor list across ranks with ONE all-reduce on a flat
    fp32 bucket (4.2 / 6.3 / 9.3 MB of weights, or the 42 / 70 / 94-float arch vector).

    Whatever path a rank is on — a captured step that produced its gradients straight in the bucket, or an
    eager step whose .grad tensors are copied in — the step's communication is the SAME call, `reduce_bucket()`:
    same communicator, same reduction, same scaling (`plan()`, decided once, collectively).  A rank that could
    not capture its step and a rank replaying its graph therefore still pair up."""

    def __init__(self, tensors, group=None, selftest=False):
        """selftest: on ONE GPU, go through everything an N > 1 step goes through — the flat bucket, the C-ABI
        communicator (of world size 1) and its all-reduce captured inside the step's hipGraph — so that the trainers'
        per-phase buckets can be exercised and timed without a second GPU (bench.py --dp-selftest)."""
        self.tensors = [t for t in tensors]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.selftest = bool(selftest) and self.world <= 1
        self._plan = None
        self.flat = None
        self.reduced = False

    def plan(self):
        """How the bucket is averaged: 'native' — RCCL through the C ABI, ncclAvg (the default under an RCCL group; a plain
        launch on the current stream, capturable); 'avg' — torch.distributed with ReduceOp.AVG (RCCL);
        'presum' — the bucket holds gradients pre-scaled by 1/world and is summed (gloo has no AVG).
        Probing AVG support is a 1-element collective: every rank calls plan() at its first step."""
        if self._plan is None:
            dev = self.tensors[0].device
            if self.world <= 1:
                self._plan = 'single'
                if self.selftest and dev.type == 'cuda':
                    from . import lib
                    if lib.comm_available():
                        NativeComm.get(self.group)
                        self._plan = 'native'
            elif native_rccl_enabled() and dev.type == 'cuda' and dist.get_backend(self.group) == 'nccl':
                # collective init, here rather than inside a capture; a rank that cannot bind librccl / create the
                # communicator takes every rank back to the host-issued plan (all ranks agree, or none goes native)
                # Decide locally and AGREE FIRST: whether librccl binds at all is a per-rank fact; a rank
                # where it does not must not leave the others inside the unique-id broadcast.  Only when every rank
                # can bind the library does anybody enter the rendezvous; its outcome is agreed on again.
                from . import lib
                try:
                    can = bool(lib.comm_available())
                except Exception:                           # noqa: BLE001
                    can = False
                ok = all_ranks_agree(can, dev, self.group)
                if ok:
                    try:
                        with _Watchdog('the C-ABI RCCL communicator (ncclCommInitRank)'):
                            NativeComm.get(self.group)
                    except Exception as e:                  # noqa: BLE001
                        import warnings
                        warnings.warn(f'bmnas.dist: C-ABI RCCL communicator unavailable ({type(e).__name__}: {e}); '
                                      'host-issued all-reduce', RuntimeWarning)
                        ok = False
                if all_ranks_agree(ok, dev, self.group):
                    self._plan = 'native'
                else:
                    self._plan = 'avg' if avg_supported(dev, self.group) else 'presum'
            else:
                self._plan = 'avg' if avg_supported(dev, self.group) else 'presum'
        return self._plan

    @property
    def loss_scale(self):
        """What a step that writes its gradients straight into the bucket multiplies its loss by: 1 / world under a
        summing collective, times the weight of this rank's shard in the global batch (uneven scatter)."""
        return (1.0 / self.world if self.plan() == 'presum' else 1.0) * shard_weight()

    # -- persistent bucket: gradients are PRODUCED in the flat buffer (GraphedTrainStep writes them
    # there inside the captured step), so a step's communication is one all-reduce and nothing else:
    # no flatten, no scale, no copy back
    def ensure_bucket(self):
        if self.flat is None:
            dev = self.tensors[0].device
            self.flat = torch.zeros(sum(t.numel() for t in self.tensors), device=dev, dtype=torch.float32)
            self.views, off = [], 0
            for t in self.tensors:
                self.views.append(self.flat[off:off + t.numel()].view(t.shape))
                off += t.numel()
        return self.views

    def reduce_bucket(self):
        """The ONE collective of a step.  The bucket holds this rank's gradients (pre-scaled by `loss_scale`)."""
        plan = self.plan()
        if plan == 'native':
            NativeComm.get(self.group).all_reduce(self.flat, average=True)
        elif plan == 'avg':
            all_reduce(self.flat, dist.ReduceOp.AVG, self.group)
        elif plan == 'presum':
            all_reduce(self.flat, dist.ReduceOp.SUM, self.group)

    def all_reduce_bucket(self):
        """After a step that wrote its gradients into the bucket: reduce, and tell the optimizer pre-hook."""
        self.reduce_bucket()
        self.reduced = True                      # the optimizer pre-hook must not average again

    def __call__(self):
        """Eager path (optimizer.step pre-hook): average .grad across ranks.  The collective always
        spans the WHOLE tensor list (missing gradients travel as zeros) and goes through reduce_bucket(),
        so that a rank on the eager path and a rank replaying its captured step issue the same all-reduce."""
        if self.reduced:
            self.reduced = False
            return
        if self.world <= 1 and not self.selftest:
            return
        views = self.ensure_bucket()
        have = [(v, t.grad) for v, t in zip(views, self.tensors) if t.grad is not None]
        if len(have) < len(views):
            self.flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if self.loss_scale != 1.0:                # 1 / world of a summing collective x this rank's shard weight
            self.flat.mul_(self.loss_scale)
        self.reduce_bucket()
        if have:
            torch._foreach_copy_([g for _, g in have], [v for v, _ in have])


def all_ranks_agree(ok, device, group=None):
    """True iff `ok` is true on EVERY rank (a MIN all-reduce of one flag): used where ranks decide locally
    whether they can take a path whose collectives differ (capturing a step)."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return bool(ok)
    t = torch.tensor([1.0 if ok else 0.0], device=device)
    all_reduce(t, dist.ReduceOp.MIN, group)
    return bool(t.item() > 0.5)


def attach(optimizer, tensors=None, group=None, selftest=False):
    """Average gradients across ranks right before ``optimizer.step()``.  tensors defaults
    to every tensor in the optimizer's param groups.  selftest: see FlatGradAllReducer."""
    if tensors is None:
        tensors = [p for g in optimizer.param_groups for p in g['params']]
    reducer = FlatGradAllReducer(tensors, group, selftest=selftest)
    handle = optimizer.register_step_pre_hook(lambda opt, args, kwargs: reducer())
    optimizer._bmnas_reducer = reducer           # GraphedTrainStep writes gradients into its bucket
    return reducer, handle


def broadcast_state(module, arch_tensors=(), src=0, group=None):
    """Make replicas identical once at start-up (DataParallel re-broadcasts every step)."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return
    for t in list(module.state_dict().values()) + list(arch_tensors):
        broadcast(t.data if hasattr(t, 'data') else t, src=src, group=group)
