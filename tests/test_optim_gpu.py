"""-m gpu: bmnas.optim.Adam (one HIP launch) against torch.optim.Adam on the CPU — the
optimizer the reference builds at mmimdb_darts_searchable.py:28-33 — same state layout, same
arithmetic.  Tolerance: 2e-6 relative to the parameter scale after 6 steps (fp32, the only
differences are fused-multiply-add contractions)."""
import copy

import pytest
import torch

from gpu_util import dev

pytestmark = pytest.mark.gpu


def _make(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) for s in shapes]


SHAPES = [(384, 384, 1), (384,), (7,), (1,), (13, 2), (3, 5, 11), (4099,), (192, 16)]


@pytest.mark.parametrize('betas,wd', [((0.9, 0.999), 1e-4), ((0.5, 0.999), 1e-3), ((0.9, 0.999), 0.0)])
def test_adam_matches_torch(betas, wd):
    from bmnas.optim import Adam
    init = _make(1, SHAPES)
    cpu = [t.clone().requires_grad_(True) for t in init]
    gpu = [t.clone().to(dev()).requires_grad_(True) for t in init]
    groups = lambda ps: [{'params': ps[:3]}, {'params': ps[3:], 'lr': 3e-3}]
    ref = torch.optim.Adam(groups(cpu), lr=1e-2, betas=betas, weight_decay=wd)
    opt = Adam(groups(gpu), lr=1e-2, betas=betas, weight_decay=wd)
    for step in range(6):
        grads = _make(100 + step, SHAPES)
        for i, (c, g_, gr) in enumerate(zip(cpu, gpu, grads)):
            if i == 2 and step < 2:
                c.grad, g_.grad = None, None          # joins later: its own step count
                continue
            c.grad, g_.grad = gr.clone(), gr.clone().to(dev())
        for grp_c, grp_g in zip(ref.param_groups, opt.param_groups):
            grp_c['lr'] = grp_g['lr'] = grp_c['lr'] * 0.9       # per-batch schedule
        ref.step()
        opt.step()
    for c, g_ in zip(cpu, gpu):
        scale = float(c.detach().abs().max()) + 1e-3
        assert float((g_.detach().cpu() - c.detach()).abs().max()) <= 2e-6 * scale + 1e-7
    sd_ref, sd = ref.state_dict(), opt.state_dict()
    for k in sd_ref['state']:
        assert float(sd['state'][k]['step']) == float(sd_ref['state'][k]['step'])
        a, b = sd['state'][k]['exp_avg_sq'].cpu(), sd_ref['state'][k]['exp_avg_sq']
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12


def test_adam_state_dict_roundtrip_with_torch():
    """A torch.optim.Adam checkpoint loads into bmnas.optim.Adam and training continues identically."""
    from bmnas.optim import Adam
    init = _make(2, SHAPES[:4])
    cpu = [t.clone().requires_grad_(True) for t in init]
    ref = torch.optim.Adam(cpu, lr=1e-2, weight_decay=1e-4)
    for step in range(3):
        for c, gr in zip(cpu, _make(50 + step, SHAPES[:4])):
            c.grad = gr
        ref.step()
    gpu = [c.detach().clone().to(dev()).requires_grad_(True) for c in cpu]
    opt = Adam(gpu, lr=1e-2, weight_decay=1e-4)
    opt.load_state_dict(copy.deepcopy(ref.state_dict()))
    for step in range(3, 6):
        for c, g_, gr in zip(cpu, gpu, _make(50 + step, SHAPES[:4])):
            c.grad, g_.grad = gr.clone(), gr.clone().to(dev())
        ref.step()
        opt.step()
    for c, g_ in zip(cpu, gpu):
        assert float((g_.detach().cpu() - c.detach()).abs().max()) <= 2e-6 * (float(c.detach().abs().max()) + 1e-3)
    assert float(opt.state_dict()['state'][0]['step']) == 6.0


def test_adam_in_graph_replays_with_new_rates():
    from bmnas.optim import Adam
    init = _make(3, SHAPES)
    cpu = [t.clone().requires_grad_(True) for t in init]
    gpu = [t.clone().to(dev()).requires_grad_(True) for t in init]
    ref = torch.optim.Adam(cpu, lr=1e-2, betas=(0.5, 0.999), weight_decay=1e-3)
    opt = Adam(gpu, lr=1e-2, betas=(0.5, 0.999), weight_decay=1e-3)
    for g_ in gpu:
        g_.grad = torch.zeros_like(g_)
    opt.capture_safe()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(graph, stream=s):
        opt.step()
    torch.cuda.current_stream().wait_stream(s)
    for step in range(5):
        lr = 1e-2 * (0.8 ** step)
        for grp in list(ref.param_groups) + list(opt.param_groups):
            grp['lr'] = lr
        for c, g_, gr in zip(cpu, gpu, _make(70 + step, SHAPES)):
            c.grad = gr.clone()
            g_.grad.copy_(gr)
        ref.step()
        opt.wait_staging()
        opt.prepare_replay()
        graph.replay()
        opt.mark_launched()
    torch.cuda.synchronize()
    for c, g_ in zip(cpu, gpu):
        assert float((g_.detach().cpu() - c.detach()).abs().max()) <= 2e-6 * (float(c.detach().abs().max()) + 1e-3)
    assert float(opt.state_dict()['state'][0]['step']) == 5.0


def test_adam_refuses_cpu_parameters():
    from bmnas import lib
    from bmnas.optim import Adam
    p = torch.zeros(4, requires_grad=True)
    p.grad = torch.ones(4)
    with pytest.raises(lib.BmnasError):
        Adam([p]).step()


def test_eager_step_between_replays_does_not_poison_the_graph():
    """graph replay, eager step() with fresh gradient tensors (same parameters), graph replay:
    the captured plan keeps its own staging buffers and pointer table."""
    from bmnas.optim import Adam
    init = _make(4, SHAPES)
    cpu = [t.clone().requires_grad_(True) for t in init]
    gpu = [t.clone().to(dev()).requires_grad_(True) for t in init]
    ref = torch.optim.Adam(cpu, lr=1e-2, weight_decay=1e-3)
    opt = Adam(gpu, lr=1e-2, weight_decay=1e-3)
    static = [torch.zeros_like(g_) for g_ in gpu]
    for g_, s_ in zip(gpu, static):
        g_.grad = s_
    opt.capture_safe()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(graph, stream=s):
        opt.step()
    torch.cuda.current_stream().wait_stream(s)
    plan = opt.captured_plan()
    for step in range(6):
        grads = _make(90 + step, SHAPES)
        for c, gr in zip(cpu, grads):
            c.grad = gr.clone()
        ref.step()
        if step in (2, 4):                       # eager, on brand-new gradient tensors
            for g_, gr in zip(gpu, grads):
                g_.grad = gr.clone().to(dev())
            opt.step()
        else:
            for s_, gr in zip(static, grads):
                s_.copy_(gr)
            opt.activate(plan)
            opt.prepare_replay()
            graph.replay()
            opt.mark_launched()
    torch.cuda.synchronize()
    for c, g_ in zip(cpu, gpu):
        assert float((g_.detach().cpu() - c.detach()).abs().max()) <= 2e-6 * (float(c.detach().abs().max()) + 1e-3)
    assert float(opt.state_dict()['state'][0]['step']) == 6.0
    # a checkpoint restored after the capture: the replay must follow the NEW moment tensors
    sd = copy.deepcopy(opt.state_dict())
    opt.load_state_dict(sd)
    grads = _make(200, SHAPES)
    for c, s_, gr in zip(cpu, static, grads):
        c.grad = gr.clone()
        s_.copy_(gr)
    ref.step()
    opt.activate(plan)
    opt.prepare_replay()
    graph.replay()
    opt.mark_launched()
    torch.cuda.synchronize()
    for c, g_ in zip(cpu, gpu):
        assert float((g_.detach().cpu() - c.detach()).abs().max()) <= 2e-6 * (float(c.detach().abs().max()) + 1e-3)
    assert float(opt.state_dict()['state'][0]['step']) == 7.0


def test_native_rccl_allreduce_single_rank_and_under_graph_capture():
    """bmnas_comm_* / bmnas_allreduce_f32 (csrc/comm.hip): RCCL bound lazily behind the C ABI.  One GPU
    is all a gpurun box has, so this is a world-size-1 communicator: the collective must run, leave the
    bucket unchanged (sum and average over one rank), and be CAPTURABLE into a hipGraph next to other
    launches on the stream — the property the in-graph all-reduce of a data-parallel step relies on."""
    from bmnas import lib
    if not lib.comm_available():
        pytest.skip('librccl not loadable in this process')
    uid = lib.comm_get_unique_id()
    assert len(uid) == 128
    comm = lib.comm_init_rank(1, 0, uid)
    try:
        x = torch.arange(4096, device=dev(), dtype=torch.float32)
        want = x.clone()
        lib.allreduce_f32(x, comm, average=False)
        lib.allreduce_f32(x, comm, average=True)
        torch.cuda.synchronize()
        assert torch.equal(x, want)
        # what the communicator reports about itself (bmnas_comm_info: the N > 1 bench line prints it)
        info = lib.comm_info(comm)
        assert info['ranks'] == 1 and info['rank'] == 0 and info['hip_device'] == torch.cuda.current_device()
        assert info['rccl_version'] > 20000
        # captured: y = 2 * x ; all-reduce(y) ; z = y + 1  replayed on fresh data
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            lib.allreduce_f32(y, comm, average=True)           # warm up RCCL's lazy setup outside capture
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                torch.mul(x, 2.0, out=y)
                lib.allreduce_f32(y, comm, average=True)
                torch.add(y, 1.0, out=z)
        torch.cuda.current_stream().wait_stream(s)
        for k in range(3):
            x.fill_(float(k))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(z, torch.full_like(z, 2.0 * k + 1.0))
    finally:
        lib.comm_destroy(comm)


def test_bench_dp_step_shapes_on_one_gpu(tmp_path):
    """bench.py --dp-selftest: the N > 1 step shapes — gradients produced in the flat bucket (last cell step's
    conv / BatchNorm gradients first), the in-graph RCCL all-reduce, and the forked-stream early all-reduce
    joined before the step ends — captured and replayed through a world-size-1 communicator (one GPU is all a
    test box has).  The line must carry the rccl evidence object and the per-shape timings."""
    import json
    import os
    import subprocess
    import sys
    from bmnas import lib
    if not lib.comm_available():
        pytest.skip('librccl not loadable in this process')
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')
    r = subprocess.run([sys.executable, bench, '--dp-selftest', '--steps', '5', '--warmup', '2', '--regions', '3',
                        '--no-full-step', '--no-roofline', '--no-cpu-baseline'], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert set(d['step_shapes']) >= {'single', 'graph', 'overlap', 'headline'}, d['step_shapes']
    assert d['rccl']['comm_ranks'] == 1 and d['rccl']['distinct_devices'] == 1
    assert d['rccl']['allreduce_bytes'] > d['rccl']['allreduce_early_bytes'] > 0
    assert d['timed_regions']['n'] == 3 and len(d['timed_regions']['ms_per_step']) == 3
    assert abs(d['value'] - 5 / (d['ms_per_step'] * 5e-3)) / d['value'] < 1e-3


def test_overlapped_bucket_reduction_gives_the_same_gradients():
    """The 'overlap' step shape against the plain one, same model, same batch, dropout off: every gradient in
    the bucket the same up to the run-to-run order of the fp32 atomics that accumulate them (2e-5 of scale; world
    size 1: the two all-reduces are identities, so a real difference would come from the fork / join — a part
    copied before its gradients were final, or a view the final copy overwrote — and be O(1))."""
    import os
    import sys
    from bmnas import lib
    if not lib.comm_available():
        pytest.skip('librccl not loadable in this process')
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as B
    for cname, batch in (('mmimdb', 32), ('ntu', 8)):
        got = B.dp_shapes_selfcheck(cname, batch)
        from gpu_util import assert_close_scaled
        for k, (plain, over) in got.items():
            # (fp32 atomics accumulate these in a run-dependent order: 1e-5 of scale, not bit equality; a part
            # copied before its gradients were final, or overwritten by the final copy, would be off by O(1))
            assert torch.isfinite(over).all(), (cname, k)
            if k.endswith('conv.bias'):
                continue                     # mathematically zero in front of a train-mode BatchNorm: pure round-off
            # (the arch gradients are small differences of sharded atomic sums: 1e-4-level run-to-run)
            assert_close_scaled(f'{cname}:{k}', over, plain, rel=1e-3 if k.startswith('arch.') else 2e-5, floor=1e-9)
