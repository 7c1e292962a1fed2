"""-m gpu: the one-pass variances where they are weakest — activations whose mean is large
against their spread.

The forward GEMM epilogues accumulate, per output channel, the batch sums of d = u - bias and d^2 (fp32 atomics)
and the consumer turns them into a variance as E[d^2] - E[d]^2 (csrc/bn_fin.hpp); the head takes the K7 LayerNorm's
statistics the same way from per-sample (sum, sum of squares) of the node outputs (csrc/head.hip).  That form loses
(mean / std)^2 of relative precision.  These tests pin what is claimed: the forward error follows ~2 eps r^2 with
r = |mean| / std of the BatchNorm input — inside the 1e-4 parity bound up to r = 20 (reached at r ~ 27), 3e-4 at 47,
~1e-3 at 91 — where torch's two-pass form stays at 1e-5 (reference math: nn.BatchNorm1d at aux_models.py:58-60 /
node_operations.py:34,53; nn.LayerNorm at model_search.py:27,65)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fusion_oracle as fo, synth
from gpu_util import build_search_net, dev


def _rel(got, want):
    got = got.detach().double().cpu().numpy()
    want = want.detach().double().cpu().numpy()
    return float(np.abs(got - want).max() / np.abs(want).max())


# (input offset mu, weight row sum c): the pre-BatchNorm activation d = W x has mean mu c and spread sqrt(1 + c^2 / C_in)
CASES = [(512, 192, 16, 128, 1.0, 0.0), (512, 192, 16, 128, 1.0, 10.0), (512, 192, 16, 128, 2.0, 17.0),
         (512, 192, 16, 128, 3.0, 22.0), (512, 192, 16, 128, 4.0, 30.0), (2048, 128, 8, 64, 1.0, 0.0), (2048, 128, 8, 64, 1.0, 10.0),
         (2048, 128, 8, 64, 1.0, 32.0), (2048, 128, 8, 64, 3.0, 40.0)]


@pytest.mark.parametrize('c_in,C,L,batch,mu,rowsum', CASES)
def test_batchnorm_after_conv_with_offset_activations(c_in, C, L, batch, mu, rowsum):
    """Conv1d(k=1) -> BatchNorm (train) -> ReLU of a reshape layer whose pre-BatchNorm activations sit at
    |mean| / std = 0 ... ~100 in every channel (input offset + unit noise, weights with a controlled row sum).
    Bound: 2.5e-7 r^2 + 2e-6 of scale at the ratio r MEASURED on the float64 evaluation (1e-4 at r = 20, 2.1e-3 at
    r = 91)."""
    import models.auxiliary.aux_models as aux

    class A:
        drpt = 0.0

    rng = np.random.Generator(np.random.PCG64(int(rowsum) + c_in))
    W = rng.standard_normal((C, c_in)).astype(np.float64) / np.sqrt(c_in)       # row norm ~ 1 -> std of d ~ 1
    W -= W.mean(1, keepdims=True)                                               # row sum 0 ...
    sign = np.where(rng.uniform(size=(C, 1)) < 0.5, -1.0, 1.0)
    W += sign * rowsum / c_in                                                   # ... then row sum = +-rowsum
    sd = {'conv.weight': W.astype(np.float32)[:, :, None],
          'conv.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
          'bn.weight': (1 + 0.1 * rng.standard_normal(C)).astype(np.float32),
          'bn.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
          'bn.running_mean': np.zeros(C, np.float32), 'bn.running_var': np.ones(C, np.float32),
          'bn.num_batches_tracked': np.zeros((), np.int64)}
    layer = aux._ReshapeBase(c_in, C, L, A())           # conv -> bn -> relu (-> dropout) on pooled (b, C_in, L) features
    shape = (batch, c_in, L)
    layer.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    layer.to(dev()).train()
    x = torch.from_numpy((mu + rng.standard_normal(shape)).astype(np.float32))       # offset mu, unit spread
    xg = x.to(dev()).requires_grad_(True)
    y = layer._tail(xg)
    wgt = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    (y * wgt.to(dev())).sum().backward()
    # float64 evaluation of the same math
    xd = x.double().reshape(batch, c_in, L).requires_grad_(True)
    Wd = torch.from_numpy(W.astype(np.float32)).double().requires_grad_(True)
    u = torch.einsum('oc,bcl->bol', Wd, xd) + torch.from_numpy(sd['conv.bias']).double()[None, :, None]
    mean, var = u.mean((0, 2), keepdim=True), u.var((0, 2), unbiased=False, keepdim=True)
    got_ratio = float((mean.abs() / var.sqrt()).median())
    yd = torch.relu((u - mean) / torch.sqrt(var + 1e-5) * torch.from_numpy(sd['bn.weight']).double()[None, :, None]
                    + torch.from_numpy(sd['bn.bias']).double()[None, :, None])
    (yd * wgt.double().reshape(yd.shape)).sum().backward()
    assert got_ratio <= 110.0, got_ratio
    # the law the one-pass form follows (measured: ~1.3e-7 r^2 = 2 eps r^2 at r = 9 ... 91, both widths): pinned with
    # a factor of two.  1e-4 is reached at r ~ 27 by measurement and guaranteed by this bound up to r = 20
    bound = 2.5e-7 * got_ratio ** 2 + 2e-6
    errs = {'y': _rel(y.reshape(yd.shape), yd), 'dx': _rel(xg.grad.reshape(xd.shape), xd.grad),
            'dW': _rel(layer.conv.weight.grad.reshape(C, c_in), Wd.grad),
            'running_var': _rel(layer.bn.running_var, 0.9 + 0.1 * u.var((0, 2), unbiased=True))}
    # the reference's OWN arithmetic (torch CPU fp32: nn.Conv1d -> nn.BatchNorm1d -> relu, aux_models.py:58-60) against
    # the same float64 evaluation: far out, where a ReLU input within round-off of zero decides a gradient element,
    # fp32 itself is the limit — the kernels have to be as good as that, not better
    ref = torch.nn.Sequential(torch.nn.Conv1d(c_in, C, 1), torch.nn.BatchNorm1d(C))
    ref.load_state_dict({'0.weight': torch.from_numpy(sd['conv.weight']), '0.bias': torch.from_numpy(sd['conv.bias']),
                         '1.weight': torch.from_numpy(sd['bn.weight']), '1.bias': torch.from_numpy(sd['bn.bias']),
                         '1.running_mean': torch.zeros(C), '1.running_var': torch.ones(C),
                         '1.num_batches_tracked': torch.zeros((), dtype=torch.long)})
    ref.train()
    xr = x.clone().requires_grad_(True)
    yr = torch.relu(ref(xr))
    (yr * wgt).sum().backward()
    ref_errs = {'y': _rel(yr, yd), 'dx': _rel(xr.grad, xd.grad), 'dW': _rel(ref[0].weight.grad.reshape(C, c_in), Wd.grad),
                'running_var': _rel(ref[1].running_var, 0.9 + 0.1 * u.var((0, 2), unbiased=True))}
    flips = int(((y.reshape(yd.shape).detach().cpu() > 0) != (yd > 0)).sum())
    print(f'|mean|/std = {got_ratio:.1f} (C_in {c_in}): ' + ', '.join(f'{k} {v:.1e} (torch fp32 {ref_errs[k]:.1e})'
                                                                      for k, v in errs.items())
          + f'; ReLU decisions differing from float64: {flips} (torch fp32: {int(((yr > 0) != (yd > 0)).sum())})')
    ref_flips = int(((yr > 0) != (yd > 0)).sum())
    for k, v in errs.items():
        if k in ('dx', 'dW') and (flips or ref_flips):
            # a ReLU input within round-off of zero decided differently (by the kernels OR by torch's own fp32
            # arithmetic): whole gradient rows move by 1e-2 ... 1e-1, in both implementations alike — nothing to pin
            continue
        assert v <= max(bound * (3.0 if k in ('dx', 'dW') else 1.0), 3.0 * ref_errs[k]), (k, v, bound, ref_errs[k])


@pytest.mark.parametrize('offset,bound', [(0.0, 1e-4), (10.0, 1e-4), (30.0, 1e-4)])
def test_k7_layernorm_statistics_with_offset_node_outputs(offset, bound):
    """The head's K7 LayerNorm takes its statistics from per-sample (sum, sum of squares) of the step nodes' outputs:
    node LayerNorm biases of `offset` put those outputs at |mean| / std ~ offset."""
    from bmnas import nn as bnn
    cfg = fo.make_cfg(N=3, C=64, L=16, S=2, M=2, ns=1, nm=1, drpt=0.0)
    seed, batch, nout = 9, 24, 7
    net = build_search_net(cfg, seed, 'train_nodrop')
    p = synth.make_params(cfg, seed)
    for k in list(p):
        if k.endswith('node_cell.ln.bias'):
            p[k] = p[k] + offset
    net.load_state_dict(p)
    net.to(dev())
    cw, cb = synth.make_classifier(cfg, nout, seed)
    cls = bnn.Linear(cfg.M * cfg.C * cfg.L, nout).to(dev())
    cls.weight.data.copy_(cw)
    cls.bias.data.copy_(cb)
    xs = [x.to(dev()) for x in synth.make_inputs(cfg, batch, seed)]
    logits = net.forward_classified(xs, cls)
    f = lambda t: t.double() if t.is_floating_point() else t
    want = fo.hypernet_logits([f(x) for x in synth.make_inputs(cfg, batch, seed)], [f(a) for a in synth.make_arch(cfg, seed)],
                              {k: f(v) for k, v in p.items()}, f(cw), f(cb), cfg, True, attn_drop=0.0)
    err = _rel(logits, want)
    print(f'node LayerNorm bias + {offset}: logits off by {err:.1e} of scale')
    assert err <= bound, err


@pytest.mark.parametrize('mu,rowsum,warns', [(1.0, 0.0, False), (4.0, 30.0, True)])
def test_bn_ratio_debug_check_reports_offset_activations(mu, rowsum, warns):
    """BMNAS_BN_RATIO_CHECK: nothing at run time noticed a BatchNorm input outside the range the
    one-pass variance is pinned for.  With the switch on every training-mode BatchNorm of the path reports
    r = |mean - bias| / std per layer (bmnas.cell.bn_ratio_report) and warns above 20."""
    import warnings
    import models.auxiliary.aux_models as aux
    from bmnas import cell as K

    class A:
        drpt = 0.0

    c_in, C, L, batch = 512, 64, 16, 32
    rng = np.random.Generator(np.random.PCG64(5))
    W = rng.standard_normal((C, c_in)) / np.sqrt(c_in)
    W -= W.mean(1, keepdims=True)
    W += rowsum / c_in
    layer = aux._ReshapeBase(c_in, C, L, A())
    with torch.no_grad():
        layer.conv.weight.copy_(torch.from_numpy(W.astype(np.float32))[:, :, None])
    layer.to(dev()).train()
    x = torch.from_numpy((mu + rng.standard_normal((batch, c_in, L))).astype(np.float32)).to(dev())
    u = torch.einsum('oc,bcl->bol', torch.from_numpy(W.astype(np.float32)).double(), x.double().cpu())
    want = float((u.mean((0, 2)).abs() / u.var((0, 2), unbiased=False).sqrt()).max())
    prev = K.BN_RATIO_CHECK
    K.BN_RATIO_CHECK = True
    K.BN_RATIOS.clear()
    try:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter('always')
            layer._tail(x)
            got = K.bn_ratio_report()
    finally:
        K.BN_RATIO_CHECK = prev
    assert len(got) == 1, got
    r = next(iter(got.values()))
    assert abs(r - want) <= 0.05 * want + 0.05, (r, want)
    assert any('|mean| / std' in str(w.message) for w in rec) == warns, (r, [str(w.message) for w in rec])
    assert (r > 20.0) == warns
