"""-m gpu: the step node's LayerNorm applied by its consumers (csrc/lazyln.hip: streaming producer, K1 pair sum / head
as the normalising consumers, partial-sum LayerNorm backward) against (a) the one-workgroup-per-sample kernels it
replaces — same network, same inputs, same dropout masks, switch flipped — and (b) the CPU oracle.

Reference math: models/search/darts/node_search.py:57-68 (NodeCell tail), model_search.py:58-67 (the consumers)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fusion_oracle as fo, synth
from gpu_util import assert_close_scaled, build_search_net, dev


def _step(cfg, batch, seed, nout, lazy, kind='bce', mode='train_nodrop', count=None):
    from bmnas import cell as K, nn as bnn, lib
    K.LAZY_LN = lazy
    net = build_search_net(cfg, seed, mode)
    cw, cb = synth.make_classifier(cfg, nout, seed)
    cls = bnn.Linear(cfg.M * cfg.C * cfg.L, nout).to(dev())
    cls.weight.data.copy_(cw)
    cls.bias.data.copy_(cb)
    xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, seed)]
    y = synth.make_labels(kind, batch, nout, seed).to(dev())
    torch.manual_seed(1234)                      # same Philox seed for both variants
    K.DROP.offset = 0
    if count is not None:
        calls = []
        orig = {n: getattr(lib, n) for n in ('node_mix_pre_fwd', 'mixsum_pair_fwd_lazy', 'head_fwd_lazy',
                                             'head_bwd_lazy', 'mixsum_pair_bwd_lazy', 'node_mix_lnp_bwd',
                                             'node_mix_ln_fwd', 'node_mix_ln_bwd')}
        for n, f in orig.items():
            setattr(lib, n, (lambda n_, f_: lambda *a, **k: (calls.append(n_), f_(*a, **k))[1])(n, f))
    try:
        with bnn.fused_criterion():
            logits = net.forward_classified(xs, cls)
            crit = bnn.BCEWithLogitsLoss() if kind == 'bce' else bnn.CrossEntropyLoss()
            loss = crit(logits, y)
        loss.backward()
    finally:
        if count is not None:
            for n, f in orig.items():
                setattr(lib, n, f)
            count.extend(calls)
        K.LAZY_LN = True
    out = {'logits': logits.detach().clone(), 'loss': loss.detach().clone()}
    for i, a in enumerate(net.arch_parameters()):
        out[f'arch.{i}'] = a.grad.clone()
    for i, x in enumerate(xs):
        out[f'input.{i}'] = x.grad.clone()
    for n, p in net.named_parameters():
        out['p.' + n] = p.grad.clone()
    out['cls.w'], out['cls.b'] = cls.weight.grad.clone(), cls.bias.grad.clone()
    torch.cuda.synchronize()
    return out


CASES = [
    # (N, C, L, S, M, ns, batch, nout, kind)
    (6, 192, 16, 2, 2, 1, 128, 23, 'bce'),        # MM-IMDB, the headline shape: 3 parts per sample
    (6, 192, 16, 2, 2, 1, 37, 23, 'bce'),         # ragged batch
    (3, 32, 16, 2, 2, 1, 6, 5, 'bce'),            # C L / 4 = 128 < one part: half-empty workgroups
    (4, 128, 8, 2, 2, 2, 9, 60, 'ce'),            # two inner steps before the tail; one part
    (3, 64, 16, 3, 2, 1, 10, 7, 'ce'),            # three cell steps: node 0 feeds two later K1 sums, M < S
    (3, 96, 16, 1, 1, 1, 5, 4, 'bce'),            # one step: the head is the only consumer; 384 float4 = 2 parts
    (2, 64, 4, 3, 3, 1, 18, 9, 'ce'),             # L = 4, every node in the head's tail
]


@pytest.mark.parametrize('case', CASES, ids=[f'N{c[0]}C{c[1]}L{c[2]}S{c[3]}M{c[4]}ns{c[5]}b{c[6]}' for c in CASES])
@pytest.mark.parametrize('mode', ['train_nodrop', 'train', 'eval'])
def test_lazy_layernorm_equals_the_per_sample_kernels(case, mode):
    N, C, L, S, M, ns, batch, nout, kind = case
    from bmnas import cell as K
    if not K.FUSE_HEAD:
        pytest.skip('BMNAS_FUSE_HEAD=0 (switch matrix): the streaming LayerNorm exists under the fused head only')
    cfg = fo.make_cfg(N=N, C=C, L=L, S=S, M=M, ns=ns, nm=1, drpt=0.2 if mode == 'train' else 0.0)
    calls = []
    lazy = _step(cfg, batch, 5, nout, True, kind, mode, count=calls)
    assert 'node_mix_pre_fwd' in calls and 'node_mix_lnp_bwd' in calls and 'head_fwd_lazy' in calls, calls
    assert 'node_mix_ln_fwd' not in calls and 'node_mix_ln_bwd' not in calls, calls
    if S > 1:
        assert calls.count('mixsum_pair_fwd_lazy') == S - 1 and calls.count('mixsum_pair_bwd_lazy') == S - 1, calls
    eager = _step(cfg, batch, 5, nout, False, kind, mode)
    for k in eager:
        if k.endswith('conv.bias'):
            continue             # in front of a train-mode BatchNorm: mathematically zero, round-off in both
        # same math, different summation order.  Forward: far inside the parity tolerance.  Gradients: both runs
        # accumulate their BatchNorm sums with atomics, and a ReLU input within round-off of zero may fall on
        # either side from one run to the next (gpu_util.match_step) — the bound is what one such element moves;
        # a wrong partial sum or a missed piece of gradient is off by O(1)
        if k in ('logits', 'loss'):
            assert_close_scaled(k, lazy[k], eager[k], rel=2e-5)
            continue
        # gradients: the tensor as a whole within 1e-2 (relative l2; one flipped unit moves it by ~1e-3 at these
        # sizes — and ONE element of an input gradient by 13 % of the tensor's scale, seen at b = 37 inside the full
        # suite), no element further off than a third of the tensor's scale
        a, w = lazy[k].double(), eager[k].double()
        l2 = float((a - w).norm() / w.norm().clamp_min(1e-30))
        assert l2 <= 1e-2, (k, 'relative l2', l2)
        assert_close_scaled(k, lazy[k], eager[k], rel=0.3)


@pytest.mark.parametrize('case', CASES[:5], ids=[f'N{c[0]}C{c[1]}L{c[2]}S{c[3]}M{c[4]}ns{c[5]}b{c[6]}' for c in CASES[:5]])
def test_lazy_layernorm_forward_against_the_oracle(case):
    """(The gradients of this path against the oracle, with the ReLU-decision matcher: every fused-head case of
    tests/test_network_gpu.py and tests/test_dropout_gpu.py with node_multiplier == 1 runs through it.)"""
    N, C, L, S, M, ns, batch, nout, kind = case
    cfg = fo.make_cfg(N=N, C=C, L=L, S=S, M=M, ns=ns, nm=1, drpt=0.0)
    seed = 5
    got = _step(cfg, batch, seed, nout, True, kind, 'train_nodrop')
    cw, cb = synth.make_classifier(cfg, nout, seed)
    ologits, oloss, og = fo.search_step(synth.make_inputs(cfg, batch, seed), synth.make_labels(kind, batch, nout, seed),
                                        synth.make_arch(cfg, seed), synth.make_params(cfg, seed), cw, cb, cfg, kind,
                                        training=True, attn_drop=0.0)
    assert_close_scaled('logits', got['logits'], ologits)
    assert_close_scaled('loss', got['loss'], oloss)
