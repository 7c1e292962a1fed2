"""CPU, world_size 2 over gloo: the data-parallel plumbing that replaces nn.DataParallel
(bmnas.dist).  The model here is a plain torch stand-in (the HIP hypernet needs a GPU); what is
under test is the sharding, the flat gradient all-reduce riding on optimizer.step pre-hooks for
BOTH the weight and the architecture optimizer, and that replicas stay identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    """weights + unregistered 'arch' tensors, like FusionNetwork."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.l1 = torch.nn.Linear(6, 5)
        self.l2 = torch.nn.Linear(5, 3)
        self.alpha = (1e-3 * torch.randn(4, 2)).requires_grad_(True)

    def arch_parameters(self):
        return [self.alpha]

    def forward(self, x):
        w = torch.softmax(self.alpha, -1)[:, 1].sum()
        return self.l2(torch.relu(self.l1(x))) * w


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bm-nas_amd'))
    from bmnas import dist as bdist
    r, l, w = bdist.init_from_env('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(1)
    X, Y = torch.randn(8, 6), torch.randn(8, 3)
    model = Toy()
    if rank == 1:                      # start different on purpose; broadcast_state must fix it
        with torch.no_grad():
            model.l1.weight.add_(1.0)
            model.alpha.add_(1.0)
    bdist.broadcast_state(model, model.arch_parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    aopt = torch.optim.Adam(model.arch_parameters(), lr=1e-2, betas=(0.5, 0.999))
    bdist.attach(opt)
    bdist.attach(aopt)
    # single-process reference on the FULL batch
    ref = Toy()
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    raopt = torch.optim.Adam(ref.arch_parameters(), lr=1e-2, betas=(0.5, 0.999))
    crit = torch.nn.MSELoss()
    for it in range(3):
        xs, ys = bdist.shard(X, rank, world), bdist.shard(Y, rank, world)
        assert xs.shape[0] == 4
        opt.zero_grad()
        crit(model(xs), ys).backward()
        opt.step()                     # pre-hook averages the shard gradients
        aopt.zero_grad()
        crit(model(xs), ys).backward()
        aopt.step()
        ropt.zero_grad()
        crit(ref(X), Y).backward()
        ropt.step()
        raopt.zero_grad()
        crit(ref(X), Y).backward()
        raopt.step()
    for a, b in zip(list(model.parameters()) + model.arch_parameters(),
                    list(ref.parameters()) + ref.arch_parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a - b).abs().max()
    # replicas identical
    flat = torch.cat([p.detach().reshape(-1) for p in list(model.parameters()) + model.arch_parameters()])
    other = flat.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(flat, other)
    with pytest.raises(ValueError):
        bdist.shard(torch.zeros(7, 2), rank, world)
    # persistent-bucket mode (what the captured hipGraph step uses): gradients are written into the
    # reducer's flat buffer pre-scaled by 1/world, one all-reduce(sum), and the optimizer pre-hook
    # must NOT average a second time.  gloo has no ReduceOp.AVG: avg_supported() must say so.
    assert bdist.avg_supported(torch.device('cpu')) is False
    red = opt._bmnas_reducer
    # the collective plan is decided once, for captured and eager steps alike (gloo: pre-scaled sum)
    assert red.plan() == 'presum' and red.loss_scale == 1.0 / world
    # ranks agree on a locally decided flag: true only if true everywhere (capture success, communicator creation)
    assert bdist.all_ranks_agree(True, torch.device('cpu')) is True
    assert bdist.all_ranks_agree(rank == 0, torch.device('cpu')) is False
    views = red.ensure_bucket()
    xs, ys = bdist.shard(X, rank, world), bdist.shard(Y, rank, world)
    grads = torch.autograd.grad(crit(model(xs), ys) / world, list(model.parameters()))
    for p, v, g in zip(model.parameters(), views, grads):
        v.copy_(g)
        p.grad = v
    red.all_reduce_bucket()
    want = torch.autograd.grad(crit(ref(X), Y), list(ref.parameters()))
    for v, w_ in zip(views, want):
        assert torch.allclose(v, w_, rtol=1e-5, atol=1e-6)
    before = [v.clone() for v in views]
    opt.step()                         # hook sees `reduced` and leaves the bucket alone
    assert red.reduced is False
    for v, b_ in zip(views, before):
        assert torch.equal(v, b_)
    # ranks on different paths must still issue the SAME collective: rank 1 has no gradient for one
    # tensor (e.g. it ran an eager step that did not touch it) — the eager reducer spans the whole
    # bucket, a missing gradient travels as zeros
    params = list(model.parameters())
    for p_ in params:
        p_.grad = torch.full_like(p_, float(rank + 1))
    if rank == 1:
        params[0].grad = None
    red.reduced = False
    red()
    assert torch.allclose(params[1].grad, torch.full_like(params[1], 1.5))
    if rank == 0:
        assert torch.allclose(params[0].grad, torch.full_like(params[0], 0.5))      # (1 + 0) / 2
    else:
        assert params[0].grad is None
    # uneven scatter (DataParallel / Tensor.chunk: 7 samples over 2 ranks = 4 + 3; 1 sample over 2 = 1 + an idle replica):
    # each rank's shard-mean gradient is weighted by n_rank * world / n inside the reducer, the result is the gradient of
    # ONE mean over all n samples; a rank without samples joins with zeros and ends with the same averaged gradient
    torch.manual_seed(7)
    for n in (7, 1):
        Xu, Yu = torch.randn(n, 6), torch.randn(n, 3)
        start, length = bdist.uneven_bounds(n, rank, world)
        assert (start, length) == ((0, (n + 1) // 2) if rank == 0 else ((n + 1) // 2, n // 2))
        bdist.set_shard_weight(length * world / n)
        red.reduced = False
        for p_ in params:
            p_.grad = None
        if length:
            crit(model(Xu.narrow(0, start, length)), Yu.narrow(0, start, length)).backward()
        else:
            for p_ in params:
                p_.grad = torch.zeros_like(p_)
        assert abs(red.loss_scale - (length * world / n) / world) < 1e-12
        red()
        # (the replicas are identical; `ref` has not followed the last opt.step above: the full-batch gradient of `model`)
        want = torch.autograd.grad(crit(model(Xu), Yu), params)
        for a, b in zip(params, want):
            assert torch.allclose(a.grad, b, rtol=1e-5, atol=1e-6), (n, (a.grad - b).abs().max())
    bdist.set_shard_weight(1.0)
    # ... and a whole optimizer step with an IDLE replica, the way the trainer loop runs it (`_loop._idle_step`): a global
    # batch of ONE sample over two ranks leaves rank 1 without samples; it joins the step's collective with zero gradients
    # and applies the same averaged update — both replicas end where a single process stepping on that sample ends
    import models.search.train_searchable._loop as loop
    torch.manual_seed(11)
    X1, Y1 = torch.randn(1, 6), torch.randn(1, 3)
    solo = Toy()
    solo.load_state_dict(model.state_dict())
    with torch.no_grad():
        solo.alpha.copy_(model.alpha)
    sopt = torch.optim.Adam(solo.parameters(), lr=1e-2)
    opt2 = torch.optim.Adam(model.parameters(), lr=1e-2)          # fresh Adam state on both sides
    bdist.attach(opt2)
    start, length = bdist.uneven_bounds(1, rank, world)
    bdist.set_shard_weight(length * world / 1)
    if length:
        opt2.zero_grad()
        crit(model(X1), Y1).backward()
        opt2.step()
    else:
        loop._idle_step(opt2)
    bdist.set_shard_weight(1.0)
    sopt.zero_grad()
    crit(solo(X1), Y1).backward()
    sopt.step()
    for a, b in zip(model.parameters(), solo.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a - b).abs().max()
    # the C-ABI communicator's rendezvous must not strand ranks.  (1) rank 0 cannot create the unique id:
    # it still reaches the broadcast and ships an error sentinel; EVERY rank raises the same RuntimeError, nobody is
    # left inside a collective.  (2) plan(): librccl binds on one rank only -> all ranks agree BEFORE any rendezvous and
    # fall back to the host-issued plan together (gloo here, so the plan is 'presum' either way; what is checked is that
    # the agreement precedes the rendezvous: comm_get_unique_id must never be called).
    from bmnas import lib as blib

    def boom():
        raise OSError('librccl.so: cannot open shared object file (simulated)')
    real_uid = blib.comm_get_unique_id
    blib.comm_get_unique_id = boom
    try:
        with pytest.raises(RuntimeError, match='rank 0 could not create the RCCL unique id'):
            bdist.NativeComm(None)
    finally:
        blib.comm_get_unique_id = real_uid
    t = torch.ones(1)
    dist.all_reduce(t)                 # both ranks are out of the rendezvous and still paired up
    assert float(t) == world
    calls = []
    real_av, real_backend = blib.comm_available, dist.get_backend
    blib.comm_available = lambda: rank == 0
    blib.comm_get_unique_id = lambda: calls.append(1) or boom()

    class _CudaLike:
        type = 'cuda'
    red2 = bdist.FlatGradAllReducer(list(model.parameters()))
    real_all_agree = bdist.all_ranks_agree
    bdist.all_ranks_agree = lambda ok, dev, group=None: real_all_agree(ok, torch.device('cpu'), group)
    real_avg = bdist.avg_supported
    bdist.avg_supported = lambda dev, group=None: False
    dist.get_backend = lambda group=None: 'nccl'
    try:
        red2.tensors = [type('T', (), {'device': _CudaLike()})()]
        assert red2.plan() == 'presum' and not calls
    finally:
        dist.get_backend = real_backend
        blib.comm_available, blib.comm_get_unique_id = real_av, real_uid
        bdist.all_ranks_agree, bdist.avg_supported = real_all_agree, real_avg
    dist.destroy_process_group()
    open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')


def test_two_rank_gradient_averaging_matches_full_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def test_single_process_is_a_no_op():
    import sys
    from bmnas import dist as bdist
    assert bdist.env_world() == int(os.environ.get('WORLD_SIZE', '1'))
    m = Toy()
    red = bdist.FlatGradAllReducer(list(m.parameters()))
    red()                              # world 1: nothing to do, must not need a process group
