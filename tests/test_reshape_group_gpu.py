"""-m gpu: the reshape layers of all modalities as ONE grouped set of launches (SURVEY.md row f1;
bmnas.functions.ReshapeGroupFn, bmnas_conv1x1_{fwd,bwd}_group, bmnas_bn_relu_{fwd,bwd}_group) against the CPU oracle
(oracle.reshape_layer, pinned by the reference's aux_layers*.npz goldens) — the real channel widths of the three
datasets (mmimdb_darts_searchable.py:86, ntu_darts_searchable.py:104, ego_darts_searchable.py:104), large and small
grids (pipelined tiles / multi-round split-K inside the group), training with live dropout (exported masks), training
without, eval — and against the per-layer path it replaces."""
import numpy as np
import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import synth
from gpu_util import assert_close_scaled, dev

pytestmark = pytest.mark.gpu

C_INS = {'mmimdb': [512, 512, 512, 512, 64, 128],
         'ntu': [512, 1024, 2048, 2048, 128, 256, 1024, 512],
         'ego': [512, 1024, 2048, 2048, 512, 1024, 2048, 2048],
         'mixed': [64, 16, 48, 2064, 32]}                  # off-grid widths: not a multiple of 32, K > 2048


class _A:
    def __init__(self, drpt):
        self.drpt = drpt


def _build(kind, C, L, drpt, seed):
    import models.auxiliary.aux_models as aux
    cls = aux.ReshapeInputLayer_MMIMDB if kind == 'mmimdb' else aux.ReshapeInputLayer
    layers, params = [], []
    for i, c_in in enumerate(C_INS[kind]):
        shapes = {'conv.weight': (C, c_in, 1), 'conv.bias': (C,), 'bn.weight': (C,), 'bn.bias': (C,),
                  'bn.running_mean': (C,), 'bn.running_var': (C,), 'bn.num_batches_tracked': ()}
        prm = synth.make_params(fo.make_cfg(N=2, C=C, L=L), seed + i, shapes)
        layer = cls(c_in, C, L, _A(drpt))
        layer.load_state_dict({k: v.clone() for k, v in prm.items()})
        layers.append(layer.to(dev()))
        params.append(prm)
    return layers, params


@pytest.mark.parametrize('kind,b,C,L,mode', [
    ('mmimdb', 128, 192, 16, 'train_drop'), ('mmimdb', 128, 192, 16, 'eval'), ('mmimdb', 16, 192, 16, 'train'),
    ('ntu', 64, 128, 8, 'train_drop'), ('ntu', 8, 128, 8, 'train'), ('ntu', 250, 128, 8, 'train'),
    ('mixed', 37, 32, 8, 'train_drop'), ('mixed', 300, 48, 4, 'train'), ('ntu', 6, 128, 8, 'eval'),
    ('ego', 48, 128, 8, 'train'), ('ego', 45, 128, 8, 'eval')])
def test_grouped_reshape_layers_match_oracle(kind, b, C, L, mode):
    import models.auxiliary.aux_models as aux
    from bmnas import cell as K
    from bmnas import lib
    drpt = 0.2 if mode == 'train_drop' else 0.0
    layers, params = _build(kind, C, L, drpt, 700)
    for m in layers:
        m.train(mode != 'eval')
    rng = np.random.Generator(np.random.PCG64(11))
    xs = [torch.from_numpy(np.maximum(rng.standard_normal((b, c, L)), 0).astype(np.float32)) for c in C_INS[kind]]
    ws = [torch.from_numpy(rng.standard_normal((b, C, L)).astype(np.float32)) for _ in xs]
    xd = [x.to(dev()).requires_grad_(True) for x in xs]
    lib.conv_family_calls(reset=True)
    assert K.DROP.record is None
    K.DROP.record = rec = []
    try:
        outs = aux.reshape_tails(layers, xd)
        sum((o * w.to(dev())).sum() for o, w in zip(outs, ws)).backward()
    finally:
        K.DROP.record = None
    calls = lib.conv_family_calls()
    assert calls['fwd_group'] == 1 and calls['bwd_group'] == 1, calls
    # long contractions (K >= 1024) on about one tile per CU: the group's forward is the multi-quad kernel, every
    # tile's contraction split over the quads of one workgroup (conv_fwd_group_q_k)
    assert calls['fwd_quads_group'] == (1 if (kind, b) in (('ntu', 64), ('ego', 48), ('ego', 45)) else 0), calls
    assert sum(v for k, v in calls.items() if not k.endswith('_group')) == 0, calls      # nothing per layer
    assert len(rec) == (len(layers) if mode == 'train_drop' else 0)
    masks = [lib.dropout_mask(d, n, dev()).cpu() for d, n in rec]
    # oracle, layer by layer, under the same masks
    xo = [x.clone().requires_grad_(True) for x in xs]
    pls = [{'r.' + k: (v.clone() if fo.is_buffer(k) else v.clone().requires_grad_(True)) for k, v in prm.items()}
           for prm in params]
    import contextlib
    with (fo.injected_masks(masks) if masks else contextlib.nullcontext()):
        oo = [fo._dropout(fo._relu(fo._conv_bn(x, q['r.conv.weight'], q['r.conv.bias'], q['r.bn.weight'],
                                               q['r.bn.bias'], q['r.bn.running_mean'], q['r.bn.running_var'],
                                               mode != 'eval')), drpt, mode != 'eval') for x, q in zip(xo, pls)]
    sum((o * w).sum() for o, w in zip(oo, ws)).backward()
    for i, (layer, q) in enumerate(zip(layers, pls)):
        tag = f'{kind}[{i}] C_in {C_INS[kind][i]}'
        assert_close_scaled(tag + ' out', outs[i], oo[i])
        assert_close_scaled(tag + ' dx', xd[i].grad, xo[i].grad, rel=2e-4)
        assert_close_scaled(tag + ' dconv.weight', layer.conv.weight.grad, q['r.conv.weight'].grad, rel=2e-4)
        assert_close_scaled(tag + ' dbn.weight', layer.bn.weight.grad, q['r.bn.weight'].grad, rel=2e-4)
        assert_close_scaled(tag + ' dbn.bias', layer.bn.bias.grad, q['r.bn.bias'].grad, rel=2e-4)
        if mode == 'eval':
            assert_close_scaled(tag + ' dconv.bias', layer.conv.bias.grad, q['r.conv.bias'].grad, rel=2e-4)
        else:
            # mathematically zero (BatchNorm removes the mean): what is left is the round-off of a sum of b * L
            # terms of O(1) (the upstream gradients here are N(0, 1), not a loss's 1e-3)
            assert float(layer.conv.bias.grad.abs().max()) < 2e-6 * b * L + 1e-4
            assert_close_scaled(tag + ' rm', layer.bn.running_mean, q['r.bn.running_mean'])
            assert_close_scaled(tag + ' rv', layer.bn.running_var, q['r.bn.running_var'])
            assert int(layer.bn.num_batches_tracked) == 1


def test_grouped_path_equals_the_per_layer_path_and_serves_the_drivers():
    """reshape_tails vs [layer._tail(x)]: same results (same tile bodies; 1e-5 of scale covers the different
    atomics order), a found net's nn.ReLU placeholders pass through, inputs without gradient get none."""
    import models.auxiliary.aux_models as aux
    layers, _ = _build('mmimdb', 192, 16, 0.0, 900)
    for m in layers:
        m.train()
    rng = np.random.Generator(np.random.PCG64(5))
    xs = [torch.from_numpy(rng.standard_normal((32, c, 16)).astype(np.float32)).to(dev()) for c in C_INS['mmimdb']]
    state = [{k: v.clone() for k, v in m.state_dict().items()} for m in layers]
    xa = [x.clone().requires_grad_(i != 1) for i, x in enumerate(xs)]            # input 1: no gradient wanted
    mixed = list(layers)
    mixed[3] = torch.nn.ReLU()                                                   # a found net's placeholder
    outs = aux.reshape_tails(mixed, xa)
    sum(o.sum() * (i + 1) for i, o in enumerate(outs)).backward()
    assert xa[1].grad is None and torch.equal(outs[3], torch.relu(xa[3]))
    got = [(o.detach().clone(), None if x.grad is None else x.grad.clone(), m.conv.weight.grad.clone()
            if isinstance(m, aux._ReshapeBase) else None) for o, x, m in zip(outs, xa, mixed)]
    for m, st in zip(layers, state):
        m.load_state_dict(st)
        m.zero_grad()
    xb = [x.clone().requires_grad_(i != 1) for i, x in enumerate(xs)]
    ref = [m._tail(x) if isinstance(m, aux._ReshapeBase) else m(x) for m, x in zip(mixed, xb)]
    sum(o.sum() * (i + 1) for i, o in enumerate(ref)).backward()
    for i, (m, x) in enumerate(zip(mixed, xb)):
        assert_close_scaled(f'out {i}', got[i][0], ref[i], rel=1e-5)
        if x.grad is not None:
            assert_close_scaled(f'dx {i}', got[i][1], x.grad, rel=1e-5)
        if got[i][2] is not None:
            assert_close_scaled(f'dW {i}', got[i][2], m.conv.weight.grad, rel=1e-5)


@pytest.mark.parametrize('kind,L,shapes', [
    # MM-IMDB: VGG feature maps + a vector (mmimdb_darts_searchable.py:99-111); 4x4 pooling of 20x32, 10x16, 5x8 maps
    ('mmimdb', 16, [(7, 64, 20, 32), (7, 64, 10, 16), (7, 32, 5, 8), (7, 48), (7, 16, 3, 3), (7, 16, 4, 4)]),
    # NTU / Ego: video maps (T, H, W), skeleton maps (T, J), vectors; T > L, T < L, T == L
    ('video', 8, [(3, 32, 8, 6, 6), (3, 64, 13, 3, 3), (3, 16, 3, 4), (3, 48), (3, 16, 4, 4), (3, 32, 1, 5, 5)]),
    ('video', 4, [(2, 16, 9, 7), (2, 32, 4, 2, 2)])])
def test_grouped_adaptive_max_pool_matches_torch(kind, L, shapes):
    """PoolGroupFn (csrc/pool.hip) against the reference's own pooling ops (layer.pooled = the aten
    AdaptiveMaxPool2d + F.interpolate of aux_models.py:62-70 / 101-108): values bit for bit, and the input gradient
    routed to the same elements (ties included: duplicated maxima are planted), overlapping and repeating windows."""
    import models.auxiliary.aux_models as aux
    from bmnas.functions import PoolGroupFn
    cls = aux.ReshapeInputLayer_MMIMDB if kind == 'mmimdb' else aux.ReshapeInputLayer
    rng = np.random.Generator(np.random.PCG64(3))
    layers = [cls(s[1], 16, L, _A(0.0)).to(dev()) for s in shapes]
    xs = []
    for s in shapes:
        x = rng.standard_normal(s).astype(np.float32)
        x = np.round(x * 2) / 2                               # few distinct values: many ties inside every window
        xs.append(torch.from_numpy(x).to(dev()))
    xa = [x.clone().requires_grad_(True) for x in xs]
    xb = [x.clone().requires_grad_(True) for x in xs]
    outs = PoolGroupFn.apply([m.pool_dims(x) for m, x in zip(layers, xa)], *xa)
    ref = [m.pooled(x) for m, x in zip(layers, xb)]
    ws = [torch.from_numpy(rng.standard_normal(tuple(r.shape)).astype(np.float32)).to(dev()) for r in ref]
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    sum((o * w).sum() for o, w in zip(ref, ws)).backward()
    for i, (o, r) in enumerate(zip(outs, ref)):
        assert o.shape == r.shape and torch.equal(o, r), (kind, i)
        # an input element that is the argmax of several windows (repeating / overlapping windows) receives a SUM:
        # same terms as torch's scatter, another order -> equal up to fp32 round-off; where it is zero it is exactly zero
        assert torch.equal(xa[i].grad == 0, xb[i].grad == 0), (kind, i)
        assert_close_scaled(f'{kind} dx {i}', xa[i].grad, xb[i].grad, rel=1e-6)


def test_reshape_all_is_pool_plus_grouped_convs():
    """aux.reshape_all on raw feature maps = [layer(f)]: 1 pooling launch + the grouped conv launches."""
    import models.auxiliary.aux_models as aux
    layers, _ = _build('mmimdb', 192, 16, 0.0, 950)
    for m in layers:
        m.train()
    rng = np.random.Generator(np.random.PCG64(8))
    shapes = [(16, 512, 20, 32), (16, 512, 20, 32), (16, 512, 10, 16), (16, 512, 5, 8), (16, 64), (16, 128)]
    raws = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev()) for s in shapes]
    state = [{k: v.clone() for k, v in m.state_dict().items()} for m in layers]
    got = aux.reshape_all(layers, raws)
    for m, st in zip(layers, state):
        m.load_state_dict(st)
    ref = [m(f) for m, f in zip(layers, raws)]
    for i, (a, b_) in enumerate(zip(got, ref)):
        assert_close_scaled(f'modality {i}', a, b_, rel=1e-4)        # (different tile families: other summation orders)
