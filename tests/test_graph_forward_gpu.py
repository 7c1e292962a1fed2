"""bmnas.graph.GraphedForward: the gradient-free passes of the loops (metric pass of the dev phase, eval / test) as
one hipGraph replay each."""
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _case(cname, batch, training):
    import bench as B
    from bmnas import nn as bnn
    c = dict(B.CONFIGS[cname], drpt=0.0)                  # deterministic: dropout parity has its own tests
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    model = B.HyperNet(c, 'R', cname).to(dev)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train(training)
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    return model, crit, dev, c


@pytest.mark.parametrize('cname,batch', [('mmimdb', 16), ('ntu', 8)])
@pytest.mark.parametrize('training', [False, True])
def test_graphed_forward_equals_the_eager_forward(cname, batch, training):
    """Replays on NEW batches (copied into the static inputs) against the eager no-grad forward of an identical
    module: outputs and loss to fp32 round-off (the classifier's split-K partials and the BatchNorm batch sums are
    added with atomics: their order differs between any two runs); in train mode also the BatchNorm running statistics
    after every pass (the replays keep updating them, like the reference's dev phase does under model.train()), and
    the warm-up passes of the capture must not have moved them."""
    import bench as B
    from bmnas.graph import GraphedForward
    from gpu_util import assert_close_scaled
    model, crit, dev, c = _case(cname, batch, training)
    twin = copy.deepcopy(model)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    xs, y = B.synth_batch(c, batch, dev, 0, 'R', cname)
    fwd = GraphedForward.try_build(model, crit, xs, y)
    assert fwd, 'capture failed'
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), f'the capture moved {k}'
    for it in range(3):
        xs, y = B.synth_batch(c, batch, dev, 10 + it, 'R', cname)
        assert fwd.matches(model, xs, y)
        loss, out = fwd(xs, y)
        with torch.no_grad():
            want = twin(xs)
            wloss = crit(want, y)
        torch.cuda.synchronize()
        assert_close_scaled(f'pass {it} output', out, want, rel=1e-5)
        assert_close_scaled(f'pass {it} loss', loss.reshape(1), wloss.reshape(1), rel=1e-5)
        for (k, a), (_, b) in zip(model.state_dict().items(), twin.state_dict().items()):
            if a.dtype.is_floating_point:
                assert_close_scaled(f'pass {it} {k}', a, b, rel=1e-5)
            else:
                assert torch.equal(a, b), (it, k)
    # another mode or another batch size is not this graph's
    model.train(not training)
    assert not fwd.matches(model, xs, y)
    model.train(training)
    xs2, y2 = B.synth_batch(c, batch + 1, dev, 0, 'R', cname)
    assert not fwd.matches(model, xs2, y2)


def test_graphed_forward_declines_what_it_cannot_capture():
    """Host tensors, or a module that synchronises with the host: try_build returns False, the module's state is as
    before and the GPU is usable."""
    import bench as B
    from bmnas.graph import GraphedForward
    model, crit, dev, c = _case('mmimdb', 8, True)
    xs, y = B.synth_batch(c, 8, dev, 0, 'R', 'mmimdb')
    assert GraphedForward.try_build(model, crit, [x.cpu() for x in xs], y) is False

    class Syncing(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, inputs):
            out = self.inner(inputs)
            float(out.sum())                               # a host read inside the forward pass
            return out

    bad = Syncing(model)
    before = {k: v.clone() for k, v in bad.state_dict().items()}
    assert GraphedForward.try_build(bad, crit, xs, y) is False
    for k, v in bad.state_dict().items():
        assert torch.equal(v, before[k]), k
    with torch.no_grad():
        assert torch.isfinite(model(xs)).all()


@pytest.mark.parametrize('cname,batch,foreign_allowed', [('mmimdb', 32, 0), ('ntu', 16, 0)])
def test_captured_found_stage_step_holds_only_this_repos_launches(cname, batch, foreign_allowed):
    """no aten / runtime launches inside the captured found-stage step.  Its zero-filled accumulators
    come from ONE persistent arena that the batch-copy launch in front of every replay clears (bmnas.functions
    _StepArena), the dropout step counter is advanced by that launch too, Adam's scalars ride in its arguments: a replay
    of the MM-IMDB found network contains this repository's kernels only, and so does the NTU genotype's (round 6: a
    state read by two consumers travels through the first one, bmnas.functions.ConvBnActThruFn — the second reader's
    gradient is accumulated by the first one's data-gradient launch, not by an autograd add)."""
    import bench as B
    from bmnas import nn as bnn
    from bmnas.graph import GraphedTrainStep
    from bmnas.optim import Adam
    from torch.profiler import ProfilerActivity, profile
    c = B.CONFIGS[cname]
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    model = B.FoundNet(c, cname).to(dev).train()
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    xs, y = B.synth_batch(c, batch, dev, 0)
    xs = [x.detach() for x in xs]
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    g = GraphedTrainStep(model, crit, opt, xs, y)
    ref = [p.detach().clone() for p in model.parameters()]
    for _ in range(3):
        g(xs, y)
    torch.cuda.synchronize()
    assert any(not torch.equal(a, b) for a, b in zip(ref, model.parameters()))     # the replays do train
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loss, _ = g(xs, y)[:2]
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    foreign = [n for n in names if 'at::native' in n or 'Memcpy' in n or 'Memset' in n or 'rocclr' in n]
    assert len(foreign) <= foreign_allowed, foreign
    assert any('copy_batch_k' in n for n in names), names
    assert torch.isfinite(loss).all()


def test_captured_step_with_an_unfused_classifier_whose_output_is_not_a_multiple_of_four(monkeypatch):
    """with the classifier outside the fused head (BMNAS_FUSE_HEAD=0) LinearFn carves its (b, O) output from
    the captured step's arena; Ego's per-GPU shard is 6 x 83 = 498 floats — not a multiple of four, which the arena's
    rounding turned into a 500-float slice that `view(6, 83)` refused INSIDE the capture.  The step must capture and its
    replays must train like the eager step."""
    import bench as B
    from bmnas import cell as K
    from bmnas import nn as bnn
    from bmnas.graph import GraphedTrainStep
    from bmnas.optim import Adam
    monkeypatch.setattr(K, 'FUSE_HEAD', False)
    c = dict(B.CONFIGS['ego'], drpt=0.0)
    dev = torch.device('cuda:0')
    nets = []
    for _ in range(2):
        torch.manual_seed(5)
        m = B.HyperNet(c, 'F', 'ego').to(dev).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        nets.append(m)
    crit = bnn.CrossEntropyLoss()
    xs, y = B.synth_batch(c, 6, dev, 0)
    xs = [x.detach() for x in xs]
    assert (6 * c['nout']) % 4 != 0
    opts = [Adam(m.parameters(), lr=1e-3, weight_decay=1e-4) for m in nets]
    g = GraphedTrainStep(nets[0], crit, opts[0], xs, y)
    first = None
    for _ in range(4):                         # (losses of later steps depend on the earlier updates: they pin those too)
        loss_g = float(g(xs, y)[0])
        opts[1].zero_grad()
        loss_e = crit(nets[1](xs), y)
        loss_e.backward()
        opts[1].step()
        loss_e = float(loss_e.detach())
        assert abs(loss_g - loss_e) <= 2e-4 * max(1.0, abs(loss_e)), (loss_g, loss_e)
        first = loss_e if first is None else first
    assert loss_e < first                      # and the replays do train


@pytest.mark.parametrize('cname,batch', [('mmimdb', 32), ('ntu', 8)])
def test_k_steps_per_replay_train_like_single_steps(cname, batch):
    """GraphedTrainStep(k=4): four consecutive optimisation steps — each over a batch of its own, each with the learning
    rate of ITS step — as one hipGraph replay.  Against the same eight batches taken one by one by a
    single-step graph on an identically initialised copy: every step's loss agrees (later losses depend on the earlier
    updates, so the per-step Adam scalars and gradient tensors of all four slots are pinned), and so do the step counts."""
    import bench as B
    from bmnas import nn as bnn
    from bmnas.graph import GraphedTrainStep
    from bmnas.optim import Adam
    c = dict(B.CONFIGS[cname], drpt=0.0)
    dev = torch.device('cuda:0')
    nets = []
    for _ in range(2):
        torch.manual_seed(7)
        m = B.HyperNet(c, 'F', cname).to(dev).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        nets.append(m)
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    batches = []
    for i in range(8):
        xs, y = B.synth_batch(c, batch, dev, 10 + i)
        batches.append(([x.detach() for x in xs], y))
    lrs = [2e-3 * (0.8 ** i) for i in range(8)]                  # a schedule that moves every step
    opts = [Adam(m.parameters(), lr=lrs[0], weight_decay=1e-4) for m in nets]
    g4 = GraphedTrainStep(nets[0], crit, opts[0], *batches[0], k=4)
    g1 = GraphedTrainStep(nets[1], crit, opts[1], *batches[0])
    got, want = [], []
    for r in range(2):
        for j in range(4):
            i = 4 * r + j
            for g in opts[0].param_groups:
                g['lr'] = lrs[i]
            g4.stage(j, *batches[i])
        got += [float(l) for l, _ in g4.replay_staged()]
    for i in range(8):
        for g in opts[1].param_groups:
            g['lr'] = lrs[i]
        want.append(float(g1(*batches[i])[0]))
    for i, (a, b_) in enumerate(zip(got, want)):
        assert abs(a - b_) <= 3e-4 * max(1.0, abs(b_)), (i, got, want)
    assert want[-1] != want[0]
    s0 = {float(st['step']) for st in opts[0].state_dict()['state'].values()}
    s1 = {float(st['step']) for st in opts[1].state_dict()['state'].values()}
    assert s0 == s1 == {8.0}
    with pytest.raises(RuntimeError):
        g4(*batches[0])                                          # a k-step graph is driven by stage() / replay_staged()


def test_k_architecture_steps_with_metric_forward_per_replay():
    """Architect.step_k: k x (architecture step + the dev phase's metric forward) as one replay — against Architect.step
    (..., metric=True) batch by batch on an identically initialised copy: the metric losses (each seen AFTER its own alpha
    update) and the final alphas agree."""
    import types
    import bench as B
    from bmnas import nn as bnn
    from bmnas.optim import Adam
    from models.search.darts.architect import Architect
    c = dict(B.CONFIGS['mmimdb'], drpt=0.0)
    dev = torch.device('cuda:0')
    args = types.SimpleNamespace(weight_decay=1e-4, hip_graph=True)
    crit = bnn.BCEWithLogitsLoss()
    runs = []
    batches = []
    for i in range(4):
        xs, y = B.synth_batch(c, 16, dev, 30 + i)
        batches.append(([x.detach() for x in xs], y))
    for mode in ('k', 'one'):
        torch.manual_seed(9)
        m = B.HyperNet(c, 'F', 'mmimdb').to(dev).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        opt = Adam(m.arch_parameters(), lr=3e-2, betas=(0.5, 0.999), weight_decay=1e-3)
        arch = Architect(m, args, crit, opt)
        if mode == 'k':
            outs = arch.step_k(batches, None)
            assert outs is not None and len(outs) == 4
            losses = [float(l) for l, _ in outs]
        else:
            losses = []
            for x, y in batches:
                got = arch.step(x, y, None, metric=True)
                assert got is not None
                losses.append(float(got[0]))
        runs.append((losses, [p.detach().clone() for p in m.arch_parameters()]))
    for a, b_ in zip(*[r[0] for r in runs]):
        assert abs(a - b_) <= 2e-4 * max(1.0, abs(b_)), runs
    for pa, pb in zip(runs[0][1], runs[1][1]):
        assert torch.allclose(pa, pb, rtol=1e-3, atol=1e-5)
