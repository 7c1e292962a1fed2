"""-m gpu: parity of the HIP path with every dropout LIVE — the mode the search runs in (model.train() in both
phases, train_searchable/mmimdb.py:47,54) and the mode bench.py times.

torch's generator cannot be matched bit for bit, so the comparison is made under the SAME masks from the other
side: the kernels draw from a counter-based Philox stream; `bmnas_dropout_mask` (include/bmnas_hip.h) writes out
the multipliers of a site exactly as the kernels apply them (checked here against oracle/philox.py, an
independent numpy restatement pinned by the Random123 known-answer vectors), and the CPU oracle replays the step
with those masks injected at its dropout sites (oracle.fusion_oracle.injected_masks — whose site positions and
order are pinned against the REFERENCE run with live dropout, tests/golden/make_golden_r03.py).
Every tensor of forward and backward is compared: rel 1e-4 of the tensor's scale forward, 2e-4 gradients."""
import contextlib
import json

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import philox, synth
from gpu_util import (Args, assert_close_of_scale, assert_close_scaled, build_found_net, build_search_net,
                      compare_search_step, dev)
from util import golden_files, load_npz

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def recorded_sites():
    """Collects (descriptor, numel) of every live dropout site the HIP path issues, in issue order."""
    from bmnas import cell as K
    assert K.DROP.record is None
    K.DROP.record = rec = []
    try:
        yield rec
    finally:
        K.DROP.record = None


def site_masks(rec, step_value=None):
    """The multipliers each recorded site applied, as CPU tensors (bmnas_dropout_mask)."""
    from bmnas import lib
    return [lib.dropout_mask(d, n, dev(), step_value).cpu() for d, n in rec]


# ---------------------------------------------------------------- the exported mask itself
@pytest.mark.parametrize('p,seed,offset,n', [(0.1, 2, 0, 8 * 192 * 16), (0.2, 12345678901234567, 977, 1003),
                                             (0.5, 0xFFFFFFFFFFFFFFFF, (1 << 60) + 5, 4), (0.03, 7, 1 << 40, 1),
                                             (0.9, 31, 3, 70001)])
def test_exported_mask_is_the_documented_philox_stream(p, seed, offset, n):
    from bmnas import lib
    d = lib.make_dropout(p, seed, offset)
    got = lib.dropout_mask(d, n, dev()).cpu().numpy()
    want = philox.dropout_multipliers(p, seed, offset, n)
    assert np.array_equal(got, want)
    # ... and with the device step counter of a captured step added to the offset
    ctr = torch.full((1,), (1 << 60) + 12345, dtype=torch.int64, device=dev())
    d2 = lib.make_dropout(p, seed, offset, ctr.data_ptr())
    got2 = lib.dropout_mask(d2, n, dev()).cpu().numpy()
    assert np.array_equal(got2, philox.dropout_multipliers(p, seed, offset, n, step=(1 << 60) + 12345))
    assert np.array_equal(lib.dropout_mask(d2, n, dev(), step_value=(1 << 60) + 12345).cpu().numpy(), got2)
    if n > 1000:
        assert not np.array_equal(got, got2)
        assert abs(float((got == 0).mean()) - p) < 0.02


def test_identity_descriptor_exports_ones():
    from bmnas import lib
    assert torch.equal(lib.dropout_mask(lib.NO_DROP, 37, dev()).cpu(), torch.ones(37))


# ------------------------------------------------------------------ modules, one site each
def _gen(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _rand(g, *shape):
    return torch.from_numpy(g.standard_normal(shape).astype(np.float32))


@pytest.mark.parametrize('b,C,L,same', [(5, 16, 8, True), (3, 32, 16, False), (2, 192, 16, True), (9, 128, 8, False),
                                        (7, 48, 4, False)])
def test_attention_dropout_sits_before_the_layernorm_like_the_reference(b, C, L, same):
    """ScaledDotAttn with its Dropout(0.1) live (node_operations.py:104-106: softmax(qk)v -> dropout -> LayerNorm):
    the mask cannot be read off the output (a LayerNorm follows), so output AND all four gradients are compared
    with the oracle under the exported mask — which is also what shows that the backward regenerated the
    forward's mask."""
    from bmnas.functions import SdpaLnFn
    g = _gen(1200 + b + C + L)
    x = _rand(g, b, C, L)
    y = x if same else _rand(g, b, C, L)
    w, bb, go = 1 + 0.1 * _rand(g, C, L), 0.1 * _rand(g, C, L), _rand(g, b, C, L)

    def run(device, fn):
        xd = x.clone().to(device).requires_grad_(True)
        yd = xd if same else y.clone().to(device).requires_grad_(True)
        wd, bd = w.clone().to(device).requires_grad_(True), bb.clone().to(device).requires_grad_(True)
        out = fn(xd, yd, wd, bd)
        out.backward(go.to(device))
        return out, [xd.grad, yd.grad, wd.grad, bd.grad]

    with recorded_sites() as rec:
        ho, hg = run(dev(), lambda a, b_, c, d: SdpaLnFn.apply(a, b_, c, d, 0.1, True))
    assert len(rec) == 1 and rec[0][1] == b * C * L
    masks = site_masks(rec)
    assert 0 < float((masks[0] == 0).float().mean()) < 0.3
    with fo.injected_masks(masks) as inj:
        ro, rg = run('cpu', lambda a, b_, c, d: fo.op_scaled_dot_attn(a, b_, c, d, True))
    assert inj.used == 1
    assert_close_scaled('out', ho, ro)
    for n, a, b_ in zip(['dx', 'dy', 'dln_w', 'dln_b'], hg, rg):
        assert_close_scaled(n, a, b_, rel=2e-4)


@pytest.mark.parametrize('kind', ['glu', 'fc'])
@pytest.mark.parametrize('b,C,L,p', [(4, 16, 8, 0.25), (3, 32, 16, 0.1), (8, 192, 16, 0.1), (9, 128, 8, 0.2)])
def test_conv_bn_act_modules_with_live_dropout(kind, b, C, L, p):
    """LinearGLU / ConcatFC (x != y) with their Dropout(drpt) live: kept values are scaled by 1/(1-p), dropped
    ones are exactly zero, and the backward regenerates exactly the forward's mask (the gradients match the
    oracle under the exported mask; a different mask in the backward would move them by O(1))."""
    from models.search.darts.node_operations import ConcatFC, LinearGLU
    cfg = fo.make_cfg(N=2, C=C, L=L, drpt=p)
    g = _gen(1300 + b + C + L)
    M = 2 * C if kind == 'glu' else C
    shapes = {'conv.weight': (M, 2 * C, 1), 'conv.bias': (M,), 'bn.weight': (M,), 'bn.bias': (M,),
              'bn.running_mean': (M,), 'bn.running_var': (M,), 'bn.num_batches_tracked': ()}
    prm = synth.make_params(cfg, 5 + b, shapes)
    x, y, go = _rand(g, b, C, L), _rand(g, b, C, L), _rand(g, b, C, L)
    mod = (LinearGLU if kind == 'glu' else ConcatFC)(C, Args(cfg, p))
    mod.load_state_dict({k: v.clone() for k, v in prm.items()})
    mod.to(dev()).train()
    xd, yd = x.to(dev()).requires_grad_(True), y.to(dev()).requires_grad_(True)
    with recorded_sites() as rec:
        out = mod(xd, yd)
        out.backward(go.to(dev()))
    assert len(rec) == 1
    mask = site_masks(rec)[0].view(b, C, L)
    po = {'op.' + k: (v.clone() if fo.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in prm.items()}
    xo, yo = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    with fo.injected_masks([mask]):
        ref = (fo.op_linear_glu if kind == 'glu' else fo.op_concat_fc)(xo, yo, po, 'op', True, p)
    ref.backward(go)
    oc = out.detach().cpu()
    assert torch.all(oc[mask == 0] == 0)                       # dropped elements are exact zeros
    assert_close_scaled('out', out, ref)
    assert_close_scaled('dx', xd.grad, xo.grad, rel=2e-4)
    assert_close_scaled('dy', yd.grad, yo.grad, rel=2e-4)
    assert torch.all(xd.grad.isfinite())
    assert_close_scaled('dconv.weight', mod.conv.weight.grad, po['op.conv.weight'].grad, rel=2e-4)
    assert_close_scaled('dbn.weight', mod.bn.weight.grad, po['op.bn.weight'].grad, rel=2e-4)
    assert_close_scaled('dbn.bias', mod.bn.bias.grad, po['op.bn.bias'].grad, rel=2e-4)
    # a second call draws a new mask
    with recorded_sites() as rec2:
        mod(xd, yd)
    assert not torch.equal(site_masks(rec2)[0], mask.reshape(-1))


@pytest.mark.parametrize('b,C,L,same', [(4, 16, 8, True), (5, 16, 8, False), (6, 192, 16, True), (7, 128, 8, False)])
def test_node_mixed_op_with_live_dropout(b, C, L, same):
    """NodeMixedOp in train mode, three live sites (attention 0.1, LinearGLU / ConcatFC drpt), issued in the
    reference's evaluation order (node_operations.py:119)."""
    from models.search.darts.node_operations import NodeMixedOp
    drpt = 0.2
    cfg = fo.make_cfg(N=2, C=C, L=L, S=1, M=1, ns=1, nm=1, drpt=drpt)
    g = _gen(1400 + b + C)
    prefix = 'cell._step_nodes.0.node_cell.node_ops.0._ops'
    prm = {k: v for k, v in synth.make_params(cfg, 11).items() if k.startswith(prefix)}
    x = _rand(g, b, C, L)
    y = x if same else _rand(g, b, C, L)
    gam, go = torch.softmax(_rand(g, 4), -1), _rand(g, b, C, L)
    op = NodeMixedOp(C, L, Args(cfg, drpt))
    sd = op.state_dict()
    op.load_state_dict({k: prm[prefix[:-len('_ops')] + k].clone() for k in sd})
    op.to(dev()).train()
    xd = x.to(dev()).requires_grad_(True)
    yd = xd if same else y.to(dev()).requires_grad_(True)
    gd = gam.to(dev()).requires_grad_(True)
    with recorded_sites() as rec:
        out = op(xd, yd, gd)
        out.backward(go.to(dev()))
    assert [n for _, n in rec] == [b * C * L] * 3
    assert [round(1 - 1 / d.scale, 3) for d, _ in rec] == [0.1, drpt, drpt]
    po = {k: (v.clone() if fo.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in prm.items()}
    xo = x.clone().requires_grad_(True)
    yo = xo if same else y.clone().requires_grad_(True)
    go_ = gam.clone().requires_grad_(True)
    with fo.injected_masks(site_masks(rec)) as inj:
        ref = fo.node_mixed_op(xo, yo, go_, po, prefix, True, drpt)
    assert inj.used == 3
    ref.backward(go)
    assert_close_scaled('out', out, ref)
    assert_close_scaled('dgamma', gd.grad, go_.grad, rel=2e-4)
    assert_close_scaled('dx', xd.grad, xo.grad, rel=2e-4)
    if not same:
        assert_close_scaled('dy', yd.grad, yo.grad, rel=2e-4)
    for k, v in op.named_parameters():
        want = po[prefix[:-len('_ops')] + k].grad
        if k.endswith('conv.bias'):
            assert float(v.grad.abs().max()) < 1e-4
        else:
            assert_close_scaled('d' + k, v.grad, want, rel=2e-4)


def test_reshape_layers_with_live_dropout():
    """ReshapeInputLayer{,_MMIMDB} (aux_models.py:51-76, 87-115) in train mode with dropout live, the cases of
    aux_layers_drop.npz: HIP vs oracle.reshape_layer under the exported mask."""
    import models.auxiliary.aux_models as aux
    z = np.load(golden_files('aux_layers_drop.npz')[0])
    for m in json.loads(str(z['meta'])):
        rng = _gen(77)
        C, c_in, L = m['C'], m['c_in'], m['L']
        sd = {'conv.weight': (rng.uniform(-1, 1, (C, c_in, 1)) / np.sqrt(c_in)).astype(np.float32),
              'conv.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
              'bn.weight': (1 + 0.1 * rng.standard_normal(C)).astype(np.float32),
              'bn.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
              'bn.running_mean': (0.1 * rng.standard_normal(C)).astype(np.float32),
              'bn.running_var': (1 + 0.2 * np.abs(rng.standard_normal(C))).astype(np.float32),
              'bn.num_batches_tracked': np.zeros((), np.int64)}
        sd = {k: torch.from_numpy(v) for k, v in sd.items()}

        class A:
            drpt = m['p']

        layer = getattr(aux, m['cls'])(c_in, C, L, A())
        layer.load_state_dict({k: v.clone() for k, v in sd.items()})
        layer.to(dev()).train()
        x = torch.from_numpy(rng.standard_normal(tuple(m['shape'])).astype(np.float32))
        xd = x.to(dev()).requires_grad_(True)
        with recorded_sites() as rec:
            yd = layer(xd)
            w = torch.from_numpy(rng.standard_normal(tuple(yd.shape)).astype(np.float32))
            (yd * w.to(dev())).sum().backward()
        assert len(rec) == 1, m['key']
        pp = {'r.' + k: (v.clone() if fo.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        kind = 'mmimdb' if m['cls'] == 'ReshapeInputLayer_MMIMDB' else 'video'
        with fo.injected_masks(site_masks(rec)):
            yo = fo.reshape_layer(xo, pp, 'r', L, kind, True, m['p'])
        (yo * w).sum().backward()
        k = m['key']
        assert_close_scaled(k + ':y', yd, yo)
        assert_close_scaled(k + ':dx', xd.grad, xo.grad, rel=2e-4)
        assert_close_scaled(k + ':dconv_w', layer.conv.weight.grad, pp['r.conv.weight'].grad, rel=2e-4)
        assert_close_scaled(k + ':dbn_w', layer.bn.weight.grad, pp['r.bn.weight'].grad, rel=2e-4)
        assert_close_scaled(k + ':dbn_b', layer.bn.bias.grad, pp['r.bn.bias'].grad, rel=2e-4)
        assert_close_scaled(k + ':rm', layer.bn.running_mean, pp['r.bn.running_mean'])
        assert_close_scaled(k + ':rv', layer.bn.running_var, pp['r.bn.running_var'])


# ----------------------------------------------------------------------- found networks
@pytest.mark.parametrize('path', golden_files('found_*_train_nodrop.npz'),
                         ids=lambda p: p.split('/')[-1].replace('_train_nodrop.npz', ''))
def test_found_networks_with_live_dropout(path):
    """Found_FusionNetwork (model.py:133-160, node.py:45-76; x != y) in train mode with dropout live, on the
    genotypes / shapes of the found_* fixtures, against the oracle under the exported masks."""
    meta, _ = load_npz(path)
    cfg = fo.Cfg({**meta['cfg'], 'drpt': 0.15})
    g = fo.genotype_from_jsonable(meta['genotype'])
    seed, batch = meta['seed'], meta['batch']
    net = build_found_net(cfg, g, seed, 'train')
    xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
    with recorded_sites() as rec:
        feat = net(xs)
        w = torch.from_numpy(_gen(seed).standard_normal(tuple(feat.shape)).astype(np.float32))
        (feat * w.to(dev())).sum().backward()
    p = synth.make_params(cfg, seed, fo.found_param_shapes(cfg, g))
    pp = {k: (v if fo.is_buffer(k) else v.requires_grad_(True)) for k, v in p.items()}
    xo = [x.requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
    with fo.injected_masks(site_masks(rec)) as inj:
        ofeat = fo.found_cell(xo, g, pp, cfg, True)
    assert inj.used == len(rec)
    (ofeat * w).sum().backward()
    assert_close_scaled('feat', feat, ofeat)
    for i, (a, b_) in enumerate(zip(xs, xo)):
        if b_.grad is not None:
            assert_close_scaled(f'grad:input.{i}', a.grad, b_.grad, rel=2e-4)
    for k, v in net.named_parameters():
        want = pp[k].grad
        if want is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
        elif k.endswith('conv.bias'):
            assert float(v.grad.abs().max()) < 1e-4, k
        else:
            assert_close_scaled('grad:' + k, v.grad, want, rel=2e-4)
    for k, v in net.state_dict().items():
        if fo.is_buffer(k):
            assert_close_scaled('buf:' + k, v.float(), p[k].float())


# ------------------------------------------------------------- the hypernet, real configs
def _hypernet_case(name, batch, nout, loss_kind, drpt, head, seed=31):
    """-> cfg, net, cls, xs, crit, labels with dropout live."""
    from bmnas import nn as bnn
    cfg = fo.Cfg({**fo.CONFIGS[name], **({} if drpt is None else {'drpt': drpt})})
    net = build_search_net(cfg, seed, 'train')
    cls = (torch.nn.Linear if head is None else bnn.Linear)(cfg.M * cfg.C * cfg.L, nout)
    cw, cb = synth.make_classifier(cfg, nout, seed)
    cls.weight.data.copy_(cw)
    cls.bias.data.copy_(cb)
    cls.to(dev())
    xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
    y = synth.make_labels(loss_kind, batch, nout, seed).to(dev())
    if head is None:
        crit = torch.nn.BCEWithLogitsLoss() if loss_kind == 'bce' else torch.nn.CrossEntropyLoss()
    else:
        crit = bnn.BCEWithLogitsLoss() if loss_kind == 'bce' else bnn.CrossEntropyLoss()
    return cfg, net, cls, xs, crit, y


def _forward(net, cls, xs, crit, y, head):
    from bmnas import nn as bnn
    if head is None:
        logits = cls(net(xs))
        return logits, crit(logits, y)
    with bnn.fused_criterion(head == 'deferred'):
        logits = net.forward_classified(xs, cls)
        return logits, crit(logits, y)


def _compare_step(cfg, batch, nout, loss_kind, masks, net, cls, grads_in, logits, loss, label=''):
    """Every tensor of the step against the oracle under `masks` (gpu_util.compare_search_step: the whole step
    must match one evaluation of the reference math — fp32, float64, or float64 with a verified set of ReLU
    decisions on inputs within round-off of zero taken the other way)."""
    return compare_search_step(cfg, batch, nout, loss_kind, net, cls, grads_in, logits, loss, masks=masks, label=label)


REAL = [('mmimdb', 128, 23, 'bce', None), ('ntu', 8, 60, 'ce', None), ('ego', 6, 83, 'ce', None),
        ('ego', 6, 83, 'ce', 0.15),         # the Ego main's drpt is 0: 0.15 makes its out_conv dropout live too
        ('mmimdb', 32, 23, 'bce', None), ('ntu', 64, 60, 'ce', None), ('ego', 48, 83, 'ce', 0.1),
        ('mmimdb', 100, 23, 'bce', 0.3), ('ntu', 250, 60, 'ce', None)]


@pytest.mark.parametrize('head', [None, 'fused', 'deferred'])
@pytest.mark.parametrize('name,batch,nout,loss_kind,drpt', REAL)
def test_search_step_with_live_dropout_matches_oracle(name, batch, nout, loss_kind, drpt, head):
    """Forward + backward of the hypernet at BASELINE.json's per-GPU sizes with every dropout live (the mains'
    drpt: MM-IMDB 0.1, NTU 0.2, Ego 0), eager."""
    cfg, net, cls, xs, crit, y = _hypernet_case(name, batch, nout, loss_kind, drpt, head)
    with recorded_sites() as rec:
        logits, loss = _forward(net, cls, xs, crit, y, head)
        loss.backward()
    per_node = cfg.ns * (1 + (2 if cfg.drpt > 0 else 0)) + (1 if cfg.nm != 1 and cfg.drpt > 0 else 0)
    assert len(rec) == cfg.S * per_node
    _compare_step(cfg, batch, nout, loss_kind, site_masks(rec), net, cls, [x.grad for x in xs], logits, loss,
                  f'drop: {name} b{batch} head={head}')


@pytest.mark.parametrize('name,batch,nout,loss_kind,drpt', REAL[:4])
def test_search_step_with_live_dropout_under_graph_replay(name, batch, nout, loss_kind, drpt):
    """The same step as ONE hipGraph replay (what bench.py times): the sites' Philox offsets restart at 0 in the
    capture and the device step counter — advanced by the cell prologue launch at the start of every replay — is
    added in the kernels.  Replays 1 and 3 are compared with the oracle under the masks exported for the counter
    value that replay read; the two replays must have drawn different masks."""
    from bmnas.graph import GraphedStep
    cfg, net, cls, xs, crit, y = _hypernet_case(name, batch, nout, loss_kind, drpt, 'deferred')
    state = {k: v.clone() for k, v in net.state_dict().items()}
    params = list(net.parameters()) + list(cls.parameters()) + list(net.arch_parameters()) + xs
    from bmnas.functions import unit_grad

    def fn():
        logits, loss = _forward(net, cls, xs, crit, y, 'deferred')
        grads = torch.autograd.grad(loss, params, grad_outputs=unit_grad(loss.device), allow_unused=True)
        return (loss, logits, *grads)

    with recorded_sites() as rec:
        g = GraphedStep(fn, warmup=2)
    rec = [r for r in rec if r[0].step]                  # the captured step's sites (warm-up passes are eager)
    from bmnas import cell as K
    assert g.span == sum((n + 3) // 4 for _, n in rec)
    # default switches: the cell prologue launch advances the step counter in front of every site
    assert g.advanced_first == (K.FUSE_PROLOGUE and len(rec) > 0)
    seen = []
    for replay in range(3):
        net.load_state_dict(state)                       # BatchNorm running statistics back to the start
        out = g.replay()
        torch.cuda.synchronize()
        masks = site_masks(rec, g.site_step_value())
        seen.append(masks)
        if replay == 1:
            continue
        loss, logits, grads = out[0], out[1], out[2:]
        for p, gr in zip(params, grads):
            p.grad = gr
        _compare_step(cfg, batch, nout, loss_kind, masks, net, cls, [x.grad for x in xs], logits, loss,
                      f'drop: {name} b{batch} replay {replay}')
    assert not torch.equal(seen[0][0], seen[2][0]) and not torch.equal(seen[0][0], seen[1][0])


@pytest.mark.parametrize('grouped', [False, True])
def test_dropout_in_front_of_the_cell_prologue_keeps_its_mask_in_a_captured_step(grouped):
    """Reshape layers -> hypernet -> classifier captured as one step (bench.py --tier R, the trainers' graphed
    step): the reshape layers' dropout sites are issued BEFORE the cell prologue.  The prologue must then not be
    the one that advances the step counter (their backward would regenerate a mask the forward never applied,
    ADVICE r02): the add stays at the end of the graph and every site, in forward and backward, reads the same
    value.  grouped: the reshape layers as ONE grouped set of launches (aux.reshape_all, what the drivers call):
    its first launch — the zero-fill of the group's accumulation buffers, in front of every site — advances the
    counter instead.  Compared with the oracle (reshape layers + hypernet) under the exported masks."""
    import models.auxiliary.aux_models as aux
    from bmnas import nn as bnn
    from bmnas.functions import unit_grad
    from bmnas.graph import GraphedStep
    cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'drpt': 0.2})
    batch, nout, seed = 16, 23, 31
    c_ins = [64, 128, 32, 64, 16, 48]
    net = build_search_net(cfg, seed, 'train')
    cls = bnn.Linear(cfg.M * cfg.C * cfg.L, nout)
    cw, cb = synth.make_classifier(cfg, nout, seed)
    cls.weight.data.copy_(cw)
    cls.bias.data.copy_(cb)
    cls.to(dev())
    rng = _gen(501)
    layers, lparams, raws = [], [], []
    for i, c_in in enumerate(c_ins):
        shapes = {'conv.weight': (cfg.C, c_in, 1), 'conv.bias': (cfg.C,), 'bn.weight': (cfg.C,), 'bn.bias': (cfg.C,),
                  'bn.running_mean': (cfg.C,), 'bn.running_var': (cfg.C,), 'bn.num_batches_tracked': ()}
        prm = synth.make_params(cfg, 900 + i, shapes)
        layer = aux.ReshapeInputLayer_MMIMDB(c_in, cfg.C, cfg.L, Args(cfg))
        layer.load_state_dict({k: v.clone() for k, v in prm.items()})
        layers.append(layer.to(dev()).train())
        lparams.append(prm)
        raws.append(torch.from_numpy(rng.standard_normal((batch, c_in, 8, 8)).astype(np.float32)))
    raw_d = [r.to(dev()) for r in raws]
    y = synth.make_labels('bce', batch, nout, seed).to(dev())
    crit = bnn.BCEWithLogitsLoss()
    lp = [p for layer in layers for p in layer.parameters()]
    params = lp + list(net.parameters()) + list(cls.parameters()) + list(net.arch_parameters())
    states = [{k: v.clone() for k, v in m.state_dict().items()} for m in layers + [net]]

    def fn():
        with bnn.fused_criterion(True):
            feats = aux.reshape_all(layers, raw_d) if grouped else [layer(r) for layer, r in zip(layers, raw_d)]
            logits = net.forward_classified(feats, cls)
            loss = crit(logits, y)
        return (loss, logits, *torch.autograd.grad(loss, params, grad_outputs=unit_grad(loss.device)))

    with recorded_sites() as rec:
        g = GraphedStep(fn, warmup=2)
    rec = [r for r in rec if r[0].step]
    assert len(rec) == len(c_ins) + 2 * 3
    # per layer: sites in front of the prologue -> the add ends the graph; grouped: the group's first launch advances
    assert g.advanced_first == grouped
    for replay in range(2):
        for m, st in zip(layers + [net], states):
            m.load_state_dict(st)
        out = g.replay()
        torch.cuda.synchronize()
    masks = site_masks(rec, g.site_step_value())
    # oracle: reshape layers -> hypernet -> criterion
    pl = [{f'r.{k}': (v.clone() if fo.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in prm.items()}
          for prm in lparams]
    p = {k: (v if fo.is_buffer(k) else v.requires_grad_(True)) for k, v in synth.make_params(cfg, seed).items()}
    arch = [a.requires_grad_(True) for a in synth.make_arch(cfg, seed)]
    cwo, cbo = cw.clone().requires_grad_(True), cb.clone().requires_grad_(True)
    with fo.injected_masks(masks) as inj:
        feats = [fo.reshape_layer(r, q, 'r', cfg.L, 'mmimdb', True, cfg.drpt) for r, q in zip(raws, pl)]
        ologits = fo.hypernet_logits(feats, arch, p, cwo, cbo, cfg, True)
    assert inj.used == len(masks)
    oloss = fo.loss_fn('bce')(ologits, y.cpu())
    oloss.backward()
    assert_close_of_scale('logits', out[1], ologits.detach(), rel=1e-4)
    assert_close_of_scale('loss', out[0], oloss.detach(), rel=1e-4)
    grads = dict(zip([id(t) for t in params], out[2:]))
    for i, (layer, q) in enumerate(zip(layers, pl)):
        for k, v in layer.named_parameters():
            if k == 'conv.bias':
                assert float(grads[id(v)].abs().max()) < 1e-4
            else:
                assert_close_scaled(f'reshape.{i}.{k}', grads[id(v)], q['r.' + k].grad, rel=2e-4)
    for k, v in net.named_parameters():
        if not k.endswith('conv.bias'):
            assert_close_scaled('grad:' + k, grads[id(v)], p[k].grad, rel=2e-4)
    for i, a in enumerate(net.arch_parameters()):
        assert_close_scaled(f'grad:arch.{i}', grads[id(a)], arch[i].grad, rel=2e-4)
