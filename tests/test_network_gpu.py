"""-m gpu: the product modules (models.search.darts.*) on cuda:0 against (1) the golden
vectors captured from the reference and (2) the CPU oracle, whole network fwd + bwd."""
import json

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import synth
from gpu_util import (Args, compare_search_step, assert_close_scaled, assert_summary_scaled, build_found_net, build_search_net, dev,
                      set_mode)
from util import case_id, cfg_of, golden_files, load_npz, summarize

pytestmark = pytest.mark.gpu


def _run_search_case(meta, head=None):
    """head: None -> the reference's composition cls(net(xs)) with torch's Linear / criterion;
    'fused' -> FusionNetwork.forward_classified with bmnas.nn.Linear (K7 + classifier in one launch,
    csrc/head.hip) and the bmnas criterion kernel; 'deferred' -> the same with the criterion evaluated
    by the head's backward launch (bmnas.nn.fused_criterion)."""
    from bmnas import nn as bnn
    cfg = cfg_of(meta)
    seed, batch, nout = meta['seed'], meta['batch'], meta['num_outputs']
    net = build_search_net(cfg, seed, meta['mode'])
    cls = (torch.nn.Linear if head is None else bnn.Linear)(cfg.M * cfg.C * cfg.L, nout)
    cw, cb = synth.make_classifier(cfg, nout, seed)
    cls.weight.data.copy_(cw)
    cls.bias.data.copy_(cb)
    cls.to(dev())
    xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
    y = synth.make_labels(meta['loss'], batch, nout, seed).to(dev())
    if head is None:
        crit = torch.nn.BCEWithLogitsLoss() if meta['loss'] == 'bce' else torch.nn.CrossEntropyLoss()
    else:
        crit = bnn.BCEWithLogitsLoss() if meta['loss'] == 'bce' else bnn.CrossEntropyLoss()
    with torch.set_grad_enabled(meta['has_grads']):
        if head is None:
            feat = net(xs)
            logits = cls(feat)
            loss = crit(logits, y)
        else:
            feat = None
            from bmnas import cell as K
            if not K.FUSE_HEAD:
                pytest.skip('BMNAS_FUSE_HEAD=0')
            assert net.cell.head_fusable(cls)
            with bnn.fused_criterion(head == 'deferred'):
                logits = net.forward_classified(xs, cls)
                loss = crit(logits, y)
            if head == 'deferred' and meta['has_grads']:
                assert type(loss.grad_fn).__name__ == 'DeferredLossFnBackward'
    if meta['has_grads']:
        loss.backward()
    return net, cls, xs, feat, logits, loss


@pytest.mark.parametrize('head', [None, 'fused', 'deferred'])
# ('train_drop' fixtures were drawn under torch-side seeded masks: they pin the ORACLE's dropout sites,
# tests/test_oracle_golden.py; the HIP path with dropout on is checked in tests/test_dropout_gpu.py)
@pytest.mark.parametrize('path', [p for p in golden_files('hypernet_*.npz') if 'train_drop' not in p], ids=case_id)
def test_search_hypernet_matches_reference_golden(path, head):
    meta, z = load_npz(path)
    if head is not None and meta['cfg']['M'] > meta['cfg']['S']:
        pytest.skip('the cell concatenates an input state: no fused head (FusionCell.head_fusable)')
    if head == 'deferred' and not meta['has_grads']:
        pytest.skip('a deferred criterion needs a backward pass')
    net, cls, xs, feat, logits, loss = _run_search_case(meta, head)
    assert_close_scaled('logits', logits, z['logits'])
    assert_close_scaled('loss', loss, z['loss'])
    full = meta['full']
    got = {}
    if meta['has_grads']:
        for k, v in net.named_parameters():
            got['grad:' + k] = v.grad
        got['grad:central_classifier.weight'] = cls.weight.grad
        got['grad:central_classifier.bias'] = cls.bias.grad
        for i, a in enumerate(net.arch_parameters()):
            got[f'grad:arch.{i}'] = a.grad
        for i, x in enumerate(xs):
            got[f'grad:input.{i}'] = x.grad
    for k, v in net.state_dict().items():
        if fo.is_buffer(k):
            got['buf:' + k] = v
    got['feat'] = feat
    for k in z.files:
        if k in ('meta', 'logits', 'loss') or (k == 'feat' and feat is None):
            continue
        g = got[k]
        assert g is not None, k
        if k.endswith('conv.bias') and k.startswith('grad:') and meta['mode'] != 'eval':
            assert float(g.abs().max()) < 1e-4, k     # mathematically zero (BN removes the mean)
            continue
        if full or k.startswith('grad:arch.') or g.dim() == 0:
            assert_close_scaled(k, g.double() if g.dtype != torch.float32 else g, z[k], rel=2e-4)
        else:
            s = summarize(g)
            want = z[k]
            n = g.numel()
            l2 = max(abs(float(want[1])), 1e-12)
            assert abs(s[1] - want[1]) <= 2e-4 * l2 + 1e-7, (k, 'l2', s[1], want[1])
            assert abs(s[0] - want[0]) <= 2e-4 * l2 * np.sqrt(n) + 1e-6, (k, 'sum', s[0], want[0])
            assert np.all(np.abs(s[2:] - want[2:]) <= 2e-4 * (np.abs(want[2:]) + l2 / np.sqrt(n)) + 1e-7), \
                (k, 'head', s[2:], want[2:])


@pytest.mark.parametrize('name,batch,nout,loss_kind', [('mmimdb', 32, 23, 'bce'), ('ntu', 16, 60, 'ce'),
                                                      ('ego', 7, 83, 'ce'),
                                                      # BASELINE.json per-GPU sizes (configs 2-5)
                                                      ('mmimdb', 128, 23, 'bce'), ('ntu', 8, 60, 'ce'),
                                                      ('ntu', 64, 60, 'ce'), ('ego', 6, 83, 'ce'), ('ego', 48, 83, 'ce'),
                                                      # ragged production batches: the merged / pipelined
                                                      # launches see partial tiles
                                                      ('mmimdb', 100, 23, 'bce'), ('mmimdb', 250, 23, 'bce'),
                                                      ('ntu', 250, 60, 'ce'), ('ego', 97, 83, 'ce')])
@pytest.mark.parametrize('head', [None, 'fused', 'deferred'])
def test_search_hypernet_matches_oracle_real_configs(name, batch, nout, loss_kind, head):
    """Full tensors (every gradient element) against the oracle at the three real configs,
    train-mode BN, dropout identity; ragged batches (odd batch with L=8 packs two samples per
    MFMA tile; 100 / 250 / 97 leave partial tiles in the merged and pipelined launches).

    The whole step must match ONE evaluation of the oracle at full tolerance (logits 1e-4 of scale,
    gradients 2e-4): the fp32 op sequence, the same in float64, or float64 with an explicit, verified set of
    ReLU decisions on inputs within 2e-5 of zero taken the other way (gpu_util.match_step; the two CPU
    evaluations differ from EACH OTHER that way at batch 128-250)."""
    cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.0})
    seed = 31
    meta = dict(cfg=dict(cfg), seed=seed, batch=batch, num_outputs=nout, loss=loss_kind,
                mode='train_nodrop', has_grads=True)
    net, cls, xs, feat, logits, loss = _run_search_case(meta, head)
    compare_search_step(cfg, batch, nout, loss_kind, net, cls, [x.grad for x in xs], logits, loss, masks=None,
                        seed=seed, label=f'{name} b{batch} head={head}', attn_drop=0.0)
    arch = synth.make_arch(cfg, seed)
    # genotype parity on the same arch parameters
    got = fo.genotype_to_jsonable(net.genotype())
    assert got == fo.genotype_to_jsonable(fo.network_genotype(arch, cfg))


@pytest.mark.parametrize('batch', [512, 1024])
def test_search_hypernet_matches_oracle_above_250_samples(batch):
    """BASELINE.json config 3's global batch (1024) and half of it on ONE GPU, every gradient element against the
    oracle (the streaming LayerNorm kernels and the weight-gradient tiles see 8 / 16 times the samples of config 2;
    VERDICT r03 weak 11: nothing above 250 samples was compared with the oracle).  The ReLU-decision matcher is
    capped at 16 ambiguous elements here: beyond that the case fails instead of exploring for minutes."""
    name, nout, loss_kind, head = 'mmimdb', 23, 'bce', 'deferred'
    cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.0})
    seed = 31
    meta = dict(cfg=dict(cfg), seed=seed, batch=batch, num_outputs=nout, loss=loss_kind,
                mode='train_nodrop', has_grads=True)
    net, cls, xs, feat, logits, loss = _run_search_case(meta, head)
    compare_search_step(cfg, batch, nout, loss_kind, net, cls, [x.grad for x in xs], logits, loss, masks=None,
                        seed=seed, label=f'{name} b{batch} head={head}', attn_drop=0.0, max_ambiguous=16)


@pytest.mark.parametrize('path', golden_files('found_*.npz'), ids=case_id)
def test_found_network_matches_reference_golden(path):
    meta, z = load_npz(path)
    cfg = cfg_of(meta)
    g = fo.genotype_from_jsonable(meta['genotype'])
    seed, batch = meta['seed'], meta['batch']
    net = build_found_net(cfg, g, seed, meta['mode'])
    xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
    with torch.set_grad_enabled(meta['has_grads']):
        feat = net(xs)
    # found_mm / found_nt (round 2): production feature sizes C192/L16 and C128/L8, summary form
    full = meta.get('full', True)
    close = assert_close_scaled if full else assert_summary_scaled
    close('feat', feat, z['feat'])
    if meta['has_grads']:
        w = torch.from_numpy(np.random.Generator(np.random.PCG64(seed))
                             .standard_normal(tuple(feat.shape)).astype(np.float32)).to(dev())
        (feat * w).sum().backward()
        params = dict(net.named_parameters())
        for k in z.files:
            if k.startswith('grad:input.'):
                x = xs[int(k.split('.')[-1])]
                got = x.grad if x.grad is not None else torch.zeros_like(x)
                close(k, got, z[k], rel=2e-4)
            elif k.startswith('grad:'):
                t = params[k[5:]]
                got = t.grad if t.grad is not None else torch.zeros_like(t)
                if k.endswith('conv.bias') and meta['mode'] != 'eval':
                    assert float(got.abs().max()) < 1e-4, k
                else:
                    close(k, got, z[k], rel=2e-4)
    for k, v in net.state_dict().items():
        if fo.is_buffer(k):
            if full or v.dim() == 0:
                assert_close_scaled('buf:' + k, v.float(), z['buf:' + k])
            else:
                assert_summary_scaled('buf:' + k, v.float(), z['buf:' + k])


@pytest.mark.parametrize('path', golden_files('prims_*.npz'), ids=case_id)
def test_edited_primitives_match_reference_golden(path):
    """PRIMITIVES edited to ['none', 'fc_relu', 'fc_mish', 'skip'] (reference
    operations.py:9-10, 22-65).  The mixed edges leave the one-launch HIP path and are composed op
    by op (FusionMixedOp's generic branch, FusionCell.forward's unfused loop, NodeCell's inner
    edges); the step nodes' NodeMixedOps stay on the kernels.  Against the reference's outputs."""
    import models.search.darts.genotypes as gt
    from models.search.darts.model_search import FusionNetwork
    meta, z = load_npz(path)
    cfg = cfg_of(meta)
    prims = meta['primitives']
    saved = list(gt.PRIMITIVES)
    gt.PRIMITIVES[:] = prims
    try:
        seed, batch, nout = meta['seed'], meta['batch'], meta['num_outputs']
        net = FusionNetwork(cfg.S, cfg.M, cfg.N, 2, Args(cfg), criterion=None)
        shapes = fo.param_shapes(cfg, prims)
        assert set(net.state_dict().keys()) == set(shapes.keys())
        net.load_state_dict(synth.make_params(cfg, seed, shapes))
        for dst, src in zip(net.arch_parameters(), synth.make_arch(cfg, seed, 0.5, prims)):
            assert dst.shape == src.shape
            dst.data.copy_(src)
        net.to(dev())
        set_mode(net, meta['mode'])
        cls = torch.nn.Linear(cfg.M * cfg.C * cfg.L, nout)
        cw, cb = synth.make_classifier(cfg, nout, seed)
        cls.weight.data.copy_(cw)
        cls.bias.data.copy_(cb)
        cls.to(dev())
        xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
        y = synth.make_labels(meta['loss'], batch, nout, seed).to(dev())
        crit = torch.nn.BCEWithLogitsLoss() if meta['loss'] == 'bce' else torch.nn.CrossEntropyLoss()
        with torch.set_grad_enabled(meta['has_grads']):
            feat = net(xs)
            logits = cls(feat)
            loss = crit(logits, y)
        assert_close_scaled('feat', feat, z['feat'])
        assert_close_scaled('logits', logits, z['logits'])
        assert_close_scaled('loss', loss, z['loss'])
        if meta['has_grads']:
            loss.backward()
            params = dict(net.named_parameters())
            for k in z.files:
                if not k.startswith('grad:'):
                    continue
                name = k[5:]
                if name.startswith('arch.'):
                    got = net.arch_parameters()[int(name.split('.')[1])].grad
                elif name.startswith('input.'):
                    got = xs[int(name.split('.')[1])].grad
                elif name.startswith('central_classifier.'):
                    got = getattr(cls, name.split('.')[1]).grad
                else:
                    got = params[name].grad
                got = got if got is not None else torch.zeros(z[k].shape)
                if (name.endswith('conv.bias') or name.endswith('linear.bias')) and meta['mode'] != 'eval' \
                        and float(np.abs(z[k]).max()) < 1e-4:
                    assert float(got.abs().max()) < 1e-4, k
                else:
                    assert_close_scaled(k, got, z[k], rel=3e-4)
        for k, v in net.state_dict().items():
            if fo.is_buffer(k):
                assert_close_scaled('buf:' + k, v.float(), z['buf:' + k])
        assert fo.genotype_to_jsonable(net.genotype()) == json.loads(str(z['genotype']))
    finally:
        gt.PRIMITIVES[:] = saved


@pytest.mark.parametrize('optim', ['torch', 'bmnas', 'graph', 'graph_metric'])
@pytest.mark.parametrize('path', golden_files('traj_*.npz'), ids=case_id)
def test_search_trajectory_matches_reference_golden(path, optim):
    """3 iterations of w-step + Architect.step on the product modules reproduce the reference's
    logits, arch parameters and weights — with stock torch.optim.Adam and with the one-launch
    bmnas.optim.Adam that search_setup installs, and with both phases replayed as hipGraphs
    (bmnas.graph.GraphedTrainStep: fwd + criterion + bwd + Adam in one launch)."""
    import bmnas.optim
    Adam = torch.optim.Adam if optim == 'torch' else bmnas.optim.Adam
    from gpu_util import dev
    from models.search.darts.architect import Architect
    from models.search.darts.model_search import FusionNetwork
    meta, z = load_npz(path)
    cfg = cfg_of(meta)
    seed, batch, nout, iters = meta['seed'], meta['batch'], meta['num_outputs'], meta['iters']

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fusion_net = FusionNetwork(cfg.S, cfg.M, cfg.N, 2, Args(cfg), criterion=None)
            self.central_classifier = torch.nn.Linear(cfg.M * cfg.C * cfg.L, nout)

        def forward(self, xs):
            return self.central_classifier(self.fusion_net(list(xs)))

        def arch_parameters(self):
            return self.fusion_net.arch_parameters()

    model = Net()
    model.fusion_net.load_state_dict(synth.make_params(cfg, seed))
    for dst, src in zip(model.arch_parameters(), synth.make_arch(cfg, seed, 1e-3)):
        dst.data.copy_(src)
    cw, cb = synth.make_classifier(cfg, nout, seed)
    model.central_classifier.weight.data.copy_(cw)
    model.central_classifier.bias.data.copy_(cb)
    crit = torch.nn.BCEWithLogitsLoss() if meta['loss'] == 'bce' else torch.nn.CrossEntropyLoss()
    # optimizers created BEFORE .to(device), like the reference's train_darts_model
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    aopt = Adam(model.arch_parameters(), lr=3e-4, betas=(0.5, 0.999), weight_decay=1e-3)
    model.to(dev())
    set_mode(model, 'train_nodrop')
    architect = Architect(model, Args(cfg), crit, aopt)
    w_graph = a_graph = None
    for it in range(iters):
        xs = [x.to(dev()) for x in synth.make_inputs(cfg, batch, seed + 10 * it)]
        y = synth.make_labels(meta['loss'], batch, nout, seed + 10 * it).to(dev())
        xv = [x.to(dev()) for x in synth.make_inputs(cfg, batch, seed + 10 * it + 5)]
        yv = synth.make_labels(meta['loss'], batch, nout, seed + 10 * it + 5).to(dev())
        if optim in ('graph', 'graph_metric'):
            from bmnas.graph import GraphedTrainStep
            if w_graph is None:
                w_graph = GraphedTrainStep(model, crit, opt, xs, y)
                # 'graph_metric' (round 5): the dev phase's gradient-free forward rides at the end of the architecture
                # step's replay, behind the Adam launch — it must see the UPDATED alphas, like the reference's
                a_graph = GraphedTrainStep(model, crit, aopt, xv, yv, metric_forward=optim == 'graph_metric')
            assert w_graph.matches(xs, y)
            _, logits = w_graph(xs, y)[:2]
            assert_close_scaled(f'train_logits.{it}', logits, z[f'train_logits.{it}'], rel=5e-4)
            got = a_graph(xv, yv)
            if optim == 'graph_metric':
                assert len(got) == 4
                assert_close_scaled(f'dev_logits.{it}', got[3], z[f'dev_logits.{it}'], rel=5e-4)
                assert_close_of_scale_loss = float(crit(got[3], yv))
                assert abs(float(got[2]) - assert_close_of_scale_loss) <= 1e-5 * max(1.0, abs(assert_close_of_scale_loss))
                continue
        else:
            opt.zero_grad()
            logits = model(xs)
            crit(logits, y).backward()
            opt.step()
            assert_close_scaled(f'train_logits.{it}', logits, z[f'train_logits.{it}'], rel=5e-4)
            architect.step(xv, yv, None)
        with torch.no_grad():
            assert_close_scaled(f'dev_logits.{it}', model(xv), z[f'dev_logits.{it}'], rel=5e-4)
    for i, a in enumerate(model.arch_parameters()):
        assert a.is_cuda
        assert_close_scaled(f'arch.{i}', a, z[f'arch.{i}'], rel=5e-4)
    assert fo.genotype_to_jsonable(model.fusion_net.genotype()) == json.loads(str(z['genotype']))


@pytest.mark.parametrize('name,batch', [('mmimdb', 16), ('ntu', 8)])
def test_cell_called_with_softmaxed_weights(name, batch):
    """FusionCell.forward(input_features, weights) with weights = softmax(alphas), the
    reference's own call (model_search.py:95-96): same output and the same alpha gradient as
    the one-launch path that hands the cell the raw alphas."""
    cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.0})
    net = build_search_net(cfg, 5, 'train_nodrop')
    xs = [x.to(dev()) for x in synth.make_inputs(cfg, batch, 5)]
    feat = net(xs)
    w = torch.randn_like(feat)
    (feat * w).sum().backward()
    want_alpha = net.alphas_edges.grad.clone()
    want_beta = net.arch_parameters()[1].grad.clone()
    for a in net.arch_parameters():
        a.grad = None
    net2 = build_search_net(cfg, 5, 'train_nodrop')
    weights = torch.softmax(net2.alphas_edges, dim=-1)
    feat2 = net2.cell(xs, weights)
    (feat2 * w).sum().backward()
    assert_close_scaled('feat', feat2, feat.detach().cpu())
    assert_close_scaled('alpha grad', net2.alphas_edges.grad, want_alpha.cpu(), rel=3e-4)
    assert_close_scaled('beta grad', net2.arch_parameters()[1].grad, want_beta.cpu(), rel=3e-4)


@pytest.mark.parametrize('name,batch', [('mmimdb', 1024), ('ntu', 512)])
def test_full_size_batch_properties(name, batch):
    """BASELINE.json's largest global batch on one GPU, checked through properties that do not
    need the oracle at that size (eval mode: BatchNorm uses running statistics, dropout is off):
    (1) samples are independent — row i of the batch-1024 output equals the output of sample i
        alone and of the 128-sample shard that contains it;
    (2) data parallelism reproduces the single-device result — the gradient of the mean loss over the full batch equals the
        mean of the 8 shard gradients (what the flat RCCL all-reduce computes)."""
    cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.1})
    net = build_search_net(cfg, 7, 'eval')
    nout = 23
    cls = torch.nn.Linear(cfg.M * cfg.C * cfg.L, nout).to(dev())
    xs = [x.to(dev()) for x in synth.make_inputs(cfg, batch, 3)]
    y = synth.make_labels('bce', batch, nout, 3).to(dev())
    crit = torch.nn.BCEWithLogitsLoss()
    params = [p for p in net.parameters()] + list(net.arch_parameters()) + list(cls.parameters())

    def grads_of(lo, hi):
        for p in params:
            p.grad = None
        out = cls(net([x[lo:hi] for x in xs]))
        crit(out, y[lo:hi]).backward()
        return out.detach(), [None if p.grad is None else p.grad.clone() for p in params]

    full_out, full_g = grads_of(0, batch)
    shards = 8
    per = batch // shards
    acc = None
    for r in range(shards):
        out, g = grads_of(r * per, (r + 1) * per)
        assert_close_scaled(f'shard {r} rows', out, full_out[r * per:(r + 1) * per].cpu(), rel=2e-5)
        acc = g if acc is None else [a if b is None else a + b for a, b in zip(acc, g)]
    for i in (0, 1, per - 1, per, batch // 2 + 3, batch - 1):
        one = cls(net([x[i:i + 1] for x in xs])).detach()
        assert_close_scaled(f'sample {i} alone', one, full_out[i:i + 1].cpu(), rel=2e-5)
    for p, a, f in zip(params, acc, full_g):
        if f is None:
            continue
        # different summation orders over 16 k (sample, l) terms that largely cancel: fp32 noise
        # relative to the tensor's scale, not to the element
        assert_close_scaled('mean of shard gradients', a / shards, f.cpu(), rel=5e-3)


def test_graph_replays_draw_fresh_consistent_dropout_masks():
    """Train mode with dropout under hipGraph replay: every replay must use new masks — the device
    step counter advances by the step's span once per replay, inside the cell prologue launch (no
    add kernel of its own) — and the replayed autograd step stays well-formed (the gradient of
    sum(feat * s) with respect to the scalar s is sum(feat), exactly)."""
    from bmnas.graph import GraphedStep
    cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'drpt': 0.3})
    net = build_search_net(cfg, 11, 'train')
    xs = [x.to(dev()) for x in synth.make_inputs(cfg, 8, 4)]
    scale = torch.ones((), device=dev(), requires_grad=True)
    params = [p for p in net.parameters()]

    def fn():
        feat = net(xs)
        loss = (feat * scale).sum()
        grads = torch.autograd.grad(loss, [scale] + params)
        return feat, loss, grads[0], grads[1]

    g = GraphedStep(fn, warmup=2)
    feats, dscale = [], []
    for _ in range(3):
        feat, loss, ds, _ = g.replay()
        torch.cuda.synchronize()
        feats.append(feat.clone())
        # d/dscale sum(feat * scale) = sum(feat): exact
        assert_close_scaled('dscale == sum(feat)', ds, feat.sum().cpu(), rel=1e-5)
        dscale.append(float(ds))
    # fresh masks per replay: the outputs differ (same inputs, same weights, BN statistics aside the
    # dropout pattern is the only thing that changes this much)
    assert float((feats[0] - feats[1]).abs().max()) > 1e-3
    assert float((feats[1] - feats[2]).abs().max()) > 1e-3
    assert int(g.counter) - g.counter_base == 3 * g.span and g.span > 0


def _interleave_model(cfg, seed, nout):
    from models.search.darts.model_search import FusionNetwork
    from bmnas import nn as bnn

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fusion_net = FusionNetwork(cfg.S, cfg.M, cfg.N, 2, Args(cfg), criterion=None)
            self.central_classifier = bnn.Linear(cfg.M * cfg.C * cfg.L, nout)

        def forward(self, xs):
            return self.central_classifier(self.fusion_net(list(xs)))

        def arch_parameters(self):
            return self.fusion_net.arch_parameters()

    model = Net()
    model.fusion_net.load_state_dict(synth.make_params(cfg, seed))
    for dst, src in zip(model.arch_parameters(), synth.make_arch(cfg, seed, 1e-3)):
        dst.data.copy_(src)
    cw, cb = synth.make_classifier(cfg, nout, seed)
    model.central_classifier.weight.data.copy_(cw)
    model.central_classifier.bias.data.copy_(cb)
    model.to(dev())
    set_mode(model, 'train_nodrop')
    return model


def test_graph_steps_survive_eager_steps_in_between():
    """A captured step, then an EAGER step on a ragged batch (what the trainer loops do with the
    last batch of an epoch, `drop_last=False`), then captured steps again — for the weight AND the
    architecture optimizer — must give the parameters that the same sequence gives when every
    step runs eagerly.  (An eager step() that staged its pointers through the captured plan's
    pinned buffer would make every later replay apply the ragged batch's gradient.)"""
    import bmnas.optim
    from bmnas.graph import GraphedTrainStep
    cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'C': 32, 'drpt': 0.0})
    seed, nout = 4, 23
    seq = [8, 8, 5, 8, 3, 8]                      # batch sizes: 8 = the captured shape
    crit_of = lambda: __import__('bmnas.nn', fromlist=['x']).BCEWithLogitsLoss()
    finals = {}
    for mode in ('eager', 'graph'):
        model = _interleave_model(cfg, seed, nout)
        crit = crit_of()
        opt = bmnas.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        aopt = bmnas.optim.Adam(model.arch_parameters(), lr=3e-3, betas=(0.5, 0.999), weight_decay=1e-3)
        wg = ag = None
        for it, b in enumerate(seq):
            xs = [x.to(dev()) for x in synth.make_inputs(cfg, b, seed + it)]
            y = synth.make_labels('bce', b, nout, seed + it).to(dev())
            for o in (opt, aopt):
                for g in o.param_groups:
                    g['lr'] *= 0.9                 # a per-batch schedule must reach the replays
            if mode == 'graph' and b == 8:
                if wg is None:
                    wg = GraphedTrainStep(model, crit, opt, xs, y)
                    ag = GraphedTrainStep(model, crit, aopt, xs, y)
                wg(xs, y)
                ag(xs, y)
            else:
                for o in (opt, aopt):
                    o.zero_grad()
                    crit(model(xs), y).backward()
                    o.step()
        torch.cuda.synchronize()
        finals[mode] = [p.detach().cpu().clone() for p in list(model.parameters()) + list(model.arch_parameters())]
        finals[mode + '_steps'] = [float(opt.state_dict()['state'][0]['step']),
                                   float(aopt.state_dict()['state'][0]['step'])]
    assert finals['eager_steps'] == finals['graph_steps'] == [len(seq), len(seq)]
    for i, (a, b) in enumerate(zip(finals['graph'], finals['eager'])):
        assert_close_scaled(f'param {i} after graph/eager interleave', a, b, rel=2e-4)
    # and the updates were not trivially small: the comparison has teeth
    model0 = _interleave_model(cfg, seed, nout)
    moved = max(float((a - p.detach().cpu()).abs().max()) for a, p in zip(finals['eager'], model0.parameters()))
    assert moved > 1e-3


@pytest.mark.parametrize('name,batch,nout,loss_kind', [('mmimdb', 32, 23, 'bce'), ('ntu', 8, 60, 'ce'), ('ego', 6, 83, 'ce')])
@pytest.mark.parametrize('head', [None, 'deferred'])
def test_architecture_step_backward_skips_the_weight_gradients(name, batch, nout, loss_kind, head, monkeypatch):
    """Architect.step differentiates alpha / beta / gamma only (architect.py:21-29; the captured step calls
    torch.autograd.grad(loss, arch_parameters) inside bmnas.cell.arch_grads_only()).  The backward then carries no weight-gradient tiles, no
    LayerNorm-affine reductions and no classifier weight-gradient product — and the architecture gradients are the
    ones a full backward gives."""
    from bmnas import lib
    cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.0})
    meta = dict(cfg=dict(cfg), seed=17, batch=batch, num_outputs=nout, loss=loss_kind, mode='train_nodrop',
                has_grads=False)

    def forward():
        from bmnas import nn as bnn
        net = build_search_net(cfg, 17, 'train_nodrop')
        cls = (torch.nn.Linear if head is None else bnn.Linear)(cfg.M * cfg.C * cfg.L, nout).to(dev())
        cw, cb = synth.make_classifier(cfg, nout, 17)
        cls.weight.data.copy_(cw)
        cls.bias.data.copy_(cb)
        xs = [x.to(dev()) for x in synth.make_inputs(cfg, batch, 17)]
        y = synth.make_labels(loss_kind, batch, nout, 17).to(dev())
        if head is None:
            crit = torch.nn.BCEWithLogitsLoss() if loss_kind == 'bce' else torch.nn.CrossEntropyLoss()
            return net, cls, crit(cls(net(xs)), y)
        crit = bnn.BCEWithLogitsLoss() if loss_kind == 'bce' else bnn.CrossEntropyLoss()
        with bnn.fused_criterion(True):
            return net, cls, crit(net.forward_classified(xs, cls), y)

    net, cls, loss = forward()
    full = torch.autograd.grad(loss, list(net.arch_parameters()) + list(net.parameters()) + list(cls.parameters()))
    want = [g.clone() for g in full[:len(net.arch_parameters())]]
    seen = {'dW_none': 0, 'dW': 0, 'ln': 0, 'part_none': 0}
    real_all, real_ln, real_ep, real_head = lib.conv1x1_bwd_all_sdpa, lib.ln_affine_bwd_multi, lib.backward_epilogue, lib.head_bwd

    def spy_all(*a, **k):
        seen['dW_none' if a[11] is None else 'dW'] += 1
        return real_all(*a, **k)

    def spy_ln(*a, **k):
        seen['ln'] += 1
        return real_ln(*a, **k)

    def spy_ep(probs, *a, **k):
        seen['ln'] += len(probs)
        return real_ep(probs, *a, **k)

    def spy_head(*a, **k):
        seen['part_none'] += a[14] is None
        return real_head(*a, **k)

    monkeypatch.setattr(lib, 'conv1x1_bwd_all_sdpa', spy_all)
    monkeypatch.setattr(lib, 'ln_affine_bwd_multi', spy_ln)
    monkeypatch.setattr(lib, 'backward_epilogue', spy_ep)
    monkeypatch.setattr(lib, 'head_bwd', spy_head)
    real_head_lazy = lib.head_bwd_lazy                   # (node_multiplier == 1: the lazy-LayerNorm form, `part` at 14 too)
    monkeypatch.setattr(lib, 'head_bwd_lazy', lambda *a, **k: (seen.__setitem__('part_none', seen['part_none'] + (a[14] is None)), real_head_lazy(*a, **k))[1])
    from bmnas import cell as K
    net2, cls2, loss2 = forward()
    with K.arch_grads_only():                             # what GraphedTrainStep does for the arch optimizer
        got = torch.autograd.grad(loss2, list(net2.arch_parameters()))
    assert seen['dW'] == 0 and seen['ln'] == 0, seen
    if K.FUSE_ATTN_GEMM:                                  # (default dispatch: one merged backward launch per inner step)
        assert seen['dW_none'] == cfg.S * cfg.ns, seen
    assert seen['part_none'] == (1 if (head is not None and K.FUSE_HEAD) else 0), seen
    for i, (g, w) in enumerate(zip(got, want)):
        assert_close_scaled(f'arch.{i}', g, w, rel=1e-4)
    # ... and a full backward afterwards still produces every gradient (the switch is per backward)
    net3, cls3, loss3 = forward()
    loss3.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net3.parameters())


@pytest.mark.parametrize('name,batch,nout,loss_kind', [('mmimdb', 32, 23, 'bce'), ('mmimdb', 128, 23, 'bce'),
                                                       ('ntu', 8, 60, 'ce'), ('ego', 6, 83, 'ce')])
@pytest.mark.parametrize('head', [None, 'deferred'])
def test_weight_step_backward_skips_the_architecture_gradients(name, batch, nout, loss_kind, head, monkeypatch):
    """The weight step of the search loop differentiates the network weights only (train_searchable/*.py: its optimizer
    holds model.parameters(); the captured step calls torch.autograd.grad(loss, those) inside
    bmnas.cell.weight_grads_only()).  The cell backward then runs no
    arch-softmax backward, its cell-level K1 pair launches get dw = dw2 = NULL (no dot products, inputs not loaded) —
    and the weight / input gradients are the ones a full backward gives (the streams that are still read are the same,
    so are the additions: equal up to the order of the batch-reduction atomics)."""
    from bmnas import lib, nn as bnn
    cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.0})

    def forward():
        net = build_search_net(cfg, 17, 'train_nodrop')
        cls = (torch.nn.Linear if head is None else bnn.Linear)(cfg.M * cfg.C * cfg.L, nout).to(dev())
        cw, cb = synth.make_classifier(cfg, nout, 17)
        cls.weight.data.copy_(cw)
        cls.bias.data.copy_(cb)
        xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, 17)]
        y = synth.make_labels(loss_kind, batch, nout, 17).to(dev())
        if head is None:
            crit = torch.nn.BCEWithLogitsLoss() if loss_kind == 'bce' else torch.nn.CrossEntropyLoss()
            return net, cls, xs, crit(cls(net(xs)), y)
        crit = bnn.BCEWithLogitsLoss() if loss_kind == 'bce' else bnn.CrossEntropyLoss()
        with bnn.fused_criterion(True):
            return net, cls, xs, crit(net.forward_classified(xs, cls), y)

    net, cls, xs, loss = forward()
    leaves = list(net.parameters()) + list(cls.parameters()) + xs
    want = torch.autograd.grad(loss, leaves + list(net.arch_parameters()))[:len(leaves)]
    seen = {'pair': 0, 'pair_null': 0, 'softmax_bwd': 0}

    def spy_pair(real):
        def f(*a, **k):
            seen['pair'] += 1
            seen['pair_null'] += a[9] is None and a[10] is None
            return real(*a, **k)
        return f

    for fn in ('mixsum_pair_bwd', 'mixsum_pair_bwd_x', 'mixsum_pair_bwd_lazy'):
        monkeypatch.setattr(lib, fn, spy_pair(getattr(lib, fn)))
    real_sm, real_ep = lib.arch_softmax_multi, lib.backward_epilogue
    monkeypatch.setattr(lib, 'arch_softmax_multi',
                        lambda ws, dws, outs, bwd, *a, **k: (seen.__setitem__('softmax_bwd', seen['softmax_bwd'] + bool(bwd)),
                                                             real_sm(ws, dws, outs, bwd, *a, **k))[1])
    monkeypatch.setattr(lib, 'backward_epilogue',
                        lambda probs, b, L, ws, *a, **k: (seen.__setitem__('softmax_bwd', seen['softmax_bwd'] + bool(ws)),
                                                          real_ep(probs, b, L, ws, *a, **k))[1])
    from bmnas import cell as K
    net2, cls2, xs2, loss2 = forward()
    with K.weight_grads_only():                          # what GraphedTrainStep does for the weight optimizer
        got = torch.autograd.grad(loss2, list(net2.parameters()) + list(cls2.parameters()) + xs2)
    assert seen['softmax_bwd'] == 0, seen
    assert seen['pair'] == seen['pair_null'], seen           # every stand-alone K1 pair backward went without dots
    if cfg.nm == 1:
        assert seen['pair'] == cfg.S, seen                    # (node_multiplier != 1: all but step 0's ride in tail launches)
    for i, (g, w) in enumerate(zip(got, want)):
        assert_close_scaled(f'grad {i}', g, w, rel=1e-5)


@pytest.mark.parametrize('ns,nm', [(1, 1), (2, 2)])
def test_strict_zero_propagates_non_finite_inputs_like_the_reference(ns, nm):
    """BMNAS_STRICT_ZERO (debug): the reference's Zero primitive is x.mul(0.) (operations.py:18-20), so a NaN / Inf in
    a cell input reaches every step's mixed sum as NaN; the kernels drop the exactly-zero term.  With the switch the
    non-finite PATTERN of the cell output equals the oracle's (which keeps the term), the finite samples agree to the
    usual tolerance; without it the deviation documented in include/bmnas_hip.h shows (an Inf input stays Inf-scaled
    instead of turning into NaN — still non-finite, but the affected samples are the same)."""
    from bmnas import cell as K
    # (L = 16: one sample per 16-column MFMA n-group.  With L = 8 / 4 two / four samples share an attention tile whose
    # block-diagonal mask is a multiplication, so a non-finite sample also takes its tile partners with it)
    cfg = fo.make_cfg(N=3, C=32, L=16, S=2, M=2, ns=ns, nm=nm, drpt=0.0)
    seed, batch = 7, 6
    xs_cpu = synth.make_inputs(cfg, batch, seed)
    xs_cpu[1][2, 5, 3] = float('nan')
    xs_cpu[2][4, 0, 0] = float('inf')
    # eval mode: train-mode BatchNorm statistics would spread one sample's NaN over the whole batch (in the reference too)
    want = fo.fusion_cell([x.clone() for x in xs_cpu], synth.make_arch(cfg, seed), synth.make_params(cfg, seed), cfg,
                          False, attn_drop=0.0)
    net = build_search_net(cfg, seed, 'eval')
    prev = K.STRICT_ZERO
    K.STRICT_ZERO = True
    try:
        with torch.no_grad():
            got = net([x.to(dev()) for x in xs_cpu]).cpu()
    finally:
        K.STRICT_ZERO = prev
    assert torch.equal(torch.isfinite(got), torch.isfinite(want)), (int((~torch.isfinite(got)).sum()),
                                                                   int((~torch.isfinite(want)).sum()))
    ok = torch.isfinite(want)
    assert bool(ok.any()) and bool((~ok).any())
    assert_close_scaled('finite part', got[ok], want[ok])
