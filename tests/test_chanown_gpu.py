"""csrc/chanown.hip — the channel-owner launches for small per-GPU shards (b L <= 64 columns: NTU b8, Ego b6).

One inner step of a search-mode NodeCell (reference models/search/darts/node_search.py:52-57 with NodeMixedOp.forward
node_operations.py:118-120) as ONE launch per direction.  Checked here against the launches they replace — the
conv + attention | mix (+ next inner sum) pairs, themselves pinned against the oracle by tests/test_kernels_gpu.py and
tests/test_dropout_gpu.py — on the same inputs and the same dropout descriptors, element by element; the whole-network
suites (tests/test_network_gpu.py, test_dropout_gpu.py, test_poison_gpu.py at NTU b8 / Ego b6) run THROUGH these
launches by default and hold them to the oracle / the reference goldens.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.gpu_util import assert_close_scaled, dev  # noqa: E402

pytestmark = pytest.mark.gpu


def _case(b, C, L, seed, n_prev, training, drop):
    from bmnas import lib
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev())
    z = r(b, C, L)
    M = 3 * C
    P = dict(Weff=r(M, C) * (C ** -0.5), bias=r(M) * 0.1, bn_w=1 + 0.1 * r(M), bn_b=0.1 * r(M),
             rm=0.1 * r(M), rv=1 + 0.1 * r(M).abs(), ln_w=1 + 0.1 * r(C, L), ln_b=0.1 * r(C, L),
             gamma=torch.softmax(r(4), -1), w=torch.softmax(r(n_prev + 1, 2), -1),
             prev=[z] * min(2, n_prev) + [r(b, C, L) for _ in range(max(0, n_prev - 2))])
    mk = lambda p, off: lib.make_dropout(p, 1234, off, None) if (training and drop) else lib.NO_DROP
    n4 = (b * C * L + 3) // 4
    P['drops'] = (mk(0.1, 0), mk(0.2, n4), mk(0.2, 2 * n4))
    return z, P


def _alloc(z, P, b, C, L, zero):
    M = 3 * C
    mk = torch.zeros_like if zero else torch.empty_like
    return dict(U=torch.empty(b, M, L, device=dev()), chan=torch.empty(4 * M, device=dev()),
                p1=torch.empty_like(z), xhat=torch.empty_like(z), st=torch.empty(2 * b, device=dev()),
                s=mk(z), zn=mk(z), rm=P['rm'].clone(), rv=P['rv'].clone(),
                nbt=torch.zeros(2, dtype=torch.int64, device=dev()))


def _two_launch_call(z, P, out, part, b, C, L, training):
    """bmnas_conv1x1_fwd_sdpa (batch sums into zero-filled shards) + bmnas_node_mix_fwd_next (finalises them)."""
    from bmnas import cell as K
    from bmnas import lib
    M = 3 * C
    d_attn, d_glu, d_fc = P['drops']
    lib.conv1x1_fwd_sdpa([z], C, P['Weff'], C, P['bias'], out['U'], part, b, L, M, 0, z, z, P['ln_w'], P['ln_b'],
                         out['p1'], out['xhat'], out['st'], C, d_attn, K.STAT_SHARDS if training else 0)
    fin = lib.make_bn_fin(part, K.STAT_SHARDS if training else 0, P['bias'], P['bn_w'], P['bn_b'], out['rm'],
                          out['rv'], out['nbt'], training)
    lib.node_mix_fwd(z, z, out['p1'], out['U'], out['chan'], P['gamma'], out['s'], b, C, L, d_glu, d_fc, fin,
                     (P['prev'], P['w'][:, 1], 2, out['zn']))


def _two_launch_fwd(z, P, b, C, L, training):
    from bmnas import cell as K
    out = _alloc(z, P, b, C, L, False)
    part = torch.zeros(K.STAT_SHARDS * 3 * C * 2, device=dev()) if training else None
    _two_launch_call(z, P, out, part, b, C, L, training)
    return out


def _co_call(z, P, out, b, C, L, training):
    from bmnas import lib
    d_attn, d_glu, d_fc = P['drops']
    bn = lib.make_bn_fin(None, 0, P['bias'], P['bn_w'], P['bn_b'], out['rm'], out['rv'], out['nbt'], training)
    lib.co_inner_fwd(z, P['Weff'], bn, P['gamma'], P['ln_w'], P['ln_b'], out['p1'], out['xhat'], out['st'], out['U'],
                     out['chan'], out['s'], b, C, L, d_attn, d_glu, d_fc, (P['prev'], P['w'][:, 1], 2, out['zn']))


def _co_fwd(z, P, b, C, L, training):
    out = _alloc(z, P, b, C, L, True)
    _co_call(z, P, out, b, C, L, training)
    return out


SHAPES = [(8, 128, 8), (6, 128, 8), (4, 192, 16), (3, 64, 16), (5, 64, 4), (1, 128, 8), (16, 64, 4)]


@pytest.mark.parametrize('b,C,L', SHAPES)
@pytest.mark.parametrize('mode', ['train_drop', 'train', 'eval'])
@pytest.mark.parametrize('n_prev', [2, 4])
def test_channel_owner_forward_equals_the_two_launches_it_replaces(b, C, L, mode, n_prev):
    from bmnas import lib
    assert lib.co_inner_ok(b, C, L)
    training, drop = mode != 'eval', mode == 'train_drop'
    z, P = _case(b, C, L, 100 * b + C + L, n_prev, training, drop)
    want = _two_launch_fwd(z, P, b, C, L, training)
    got = _co_fwd(z, P, b, C, L, training)
    torch.cuda.synchronize()
    for k in ('U', 'p1', 'xhat', 'st', 's', 'zn', 'rm', 'rv'):
        assert_close_scaled(k, got[k], want[k], rel=2e-5)
    M = 3 * C
    for i, name in enumerate(('mean', 'rstd', 'scale', 'shift')):
        assert_close_scaled('chan.' + name, got['chan'][i * M:(i + 1) * M], want['chan'][i * M:(i + 1) * M], rel=2e-5)
    assert torch.equal(got['nbt'], want['nbt'])
    if drop:            # dropped elements are exact zeros of the same positions: compare the GLU / ConcatFC zero pattern
        assert float((got['s'] - want['s']).abs().max()) <= 2e-5 * float(want['s'].abs().max())


def test_channel_owner_refuses_what_it_does_not_cover():
    from bmnas import lib
    assert not lib.co_inner_ok(9, 128, 8)          # 72 columns
    assert not lib.co_inner_ok(8, 256, 8)          # channel count outside the instantiations
    assert not lib.co_inner_ok(8, 128, 5)
    z, P = _case(9, 128, 8, 1, 2, True, False)
    with pytest.raises(lib.BmnasError):
        _co_fwd(z, P, 9, 128, 8, True)


@pytest.mark.parametrize('cname,batch', [('ntu', 8), ('ego', 6)])
def test_search_step_with_and_without_channel_owner_launches(cname, batch, monkeypatch):
    """The whole hypernet step (dropout live, same seed) through the channel-owner launches and through the launches they
    replace: logits, loss and every gradient agree to fp32 round-off — and the channel-owner path is what runs by default
    at these shapes (launch names seen by the profiler)."""
    import bench as B
    from bmnas import cell as K
    from bmnas import nn as bnn
    from torch.profiler import ProfilerActivity, profile
    c = B.CONFIGS[cname]
    res = {}
    for on in (True, False):
        monkeypatch.setattr(K, 'CO_INNER', on)
        torch.manual_seed(2)
        model = B.HyperNet(c, 'F', cname).to(dev()).train()
        crit = bnn.CrossEntropyLoss()
        xs, y = B.synth_batch(c, batch, dev(), 0)
        K.DROP.offset = 0
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            logits = model(xs)
            loss = crit(logits, y)
            params = [p for p in model.parameters()] + list(model.arch_parameters())
            grads = torch.autograd.grad(loss, params + xs)
            torch.cuda.synchronize()
        names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        res[on] = (logits.detach(), loss.detach(), [g.detach() for g in grads], names)
    assert any('co_inner_fwd_k' in n for n in res[True][3]), res[True][3]
    assert not any('co_inner' in n for n in res[False][3])
    assert len(res[True][3]) < len(res[False][3])
    assert_close_scaled('logits', res[True][0], res[False][0], rel=2e-5)
    assert abs(float(res[True][1]) - float(res[False][1])) <= 2e-5 * abs(float(res[False][1]))
    for i, (a, b_) in enumerate(zip(res[True][2], res[False][2])):
        assert_close_scaled(f'grad[{i}]', a, b_, rel=1e-4)
