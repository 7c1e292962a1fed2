"""-m gpu: which GEMM kernel family serves which shape ("prove reachable or delete").

The conv entry points pick a family by shape; every family must be reached by a shape that the
reference's configurations produce, and the parity of each case is checked against a plain
fp32 torch contraction in the same test, so no family runs unverified:

    pipe_fwd / pipe_bwd     LDS-tiled, K-chunked (standalone convs with >= 96 tiles: out_conv and
                            reshape layers at production batch)
    ksplit                  split-K, one memory round trip (small grids, K <= 768) or several register rounds
                            in one launch (longer K at <= 512 workgroups: the C_in = 1024 / 2048 reshape layers
                            at <= 64 samples per GPU)
    lds                     whole-K LDS tiles: the ONE generic fallback (channel counts that the pipelined kernels
                            do not take at large grids, a fused torch.cat of two sources with K > 768 at small
                            ones) — no reference shape needs it (round 3 removed the second fallback, conv_nj_k)
    fwd_sdpa_pipe / _ksplit conv + attention in one launch (search NodeMixedOp, large / small batch)
    bwd_all_pipe / _ksplit  data-gradient + weight-gradient + attention backward in one launch
    conv_w                  weight-gradient GEMM alone
"""
import numpy as np
import pytest
import torch

from oracle import fusion_oracle as fo
from oracle import synth
from gpu_util import assert_close_scaled, build_search_net, dev

import os

# the expectations below describe the DEFAULT dispatch; tools/test_matrix.sh forces other kernel families
# through these switches on purpose (their results are checked by the parity tests, not here)
_FORCED = [k for k in ('BMNAS_CONV_PIPE', 'BMNAS_FUSE_ATTN_GEMM', 'BMNAS_FUSE_BWD_PAIR') if os.environ.get(k) is not None]
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(bool(_FORCED), reason=f'kernel family forced by {_FORCED}')]


def _conv_case(b, C_in, M, L, seed=0, n_src=1):
    """fwd + data-gradient + weight-gradient of one 1x1 conv through the C ABI vs torch fp32.
    n_src > 1: the input is handed over as n_src channel slices (torch.cat fused into the GEMM)."""
    from bmnas import lib
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, C_in, L, generator=g).to(dev())
    if n_src > 1:
        return _conv_case_cat(lib, g, x, b, C_in, M, L, n_src)
    W = (torch.randn(M, C_in, generator=g) / C_in ** 0.5).to(dev())
    bias = torch.randn(M, generator=g).to(dev())
    dU = torch.randn(b, M, L, generator=g).to(dev())
    U = torch.empty(b, M, L, device=dev())
    n_part = lib.conv1x1_num_partials(b, L)
    part = torch.empty(n_part * M * 2, device=dev())
    lib.conv1x1_fwd([x], C_in, W, C_in, bias, U, part, b, L, M, 0)
    want = torch.einsum('mk,bkl->bml', W.double(), x.double()) + bias.double()[None, :, None]
    assert_close_scaled('U', U, want.float())
    dx = torch.empty_like(x)
    lib.conv1x1_bwd_data(dU, W, C_in, [dx], C_in, 0, b, L, M, 0)
    assert_close_scaled('dx', dx, torch.einsum('mk,bml->bkl', W.double(), dU.double()).float(), rel=2e-4)
    dW = torch.zeros(M, C_in, device=dev())
    db = torch.zeros(M, device=dev())
    lib.conv1x1_bwd_weight(dU, [x], C_in, dW, C_in, db, 0, b, L, M)
    assert_close_scaled('dW', dW, torch.einsum('bml,bkl->mk', dU.double(), x.double()).float(), rel=2e-4)
    assert_close_scaled('db', db, dU.double().sum((0, 2)).float(), rel=2e-4)


def _conv_case_cat(lib, g, x, b, C_in, M, L, n_src):
    Cs = C_in // n_src
    xs = [x[:, q * Cs:(q + 1) * Cs].contiguous() for q in range(n_src)]
    W = (torch.randn(M, C_in, generator=g) / C_in ** 0.5).to(dev())
    bias = torch.randn(M, generator=g).to(dev())
    U = torch.empty(b, M, L, device=dev())
    part = torch.empty(lib.conv1x1_num_partials(b, L) * M * 2, device=dev())
    lib.conv1x1_fwd(xs, Cs, W, C_in, bias, U, part, b, L, M, 0)
    want = torch.einsum('mk,bkl->bml', W.double(), x.double()) + bias.double()[None, :, None]
    assert_close_scaled('U', U, want.float())


CASES = [
    # (what, batch, C_in, M, L) -> families that must serve it (fwd, bwd-data)
    ('out_conv NTU b512', 512, 256, 128, 8, {'pipe_fwd', 'ksplit'}),
    ('reshape MM-IMDB C_in 512 b128', 128, 512, 192, 16, {'pipe_fwd', 'pipe_bwd'}),
    ('out_conv NTU b8', 8, 256, 128, 8, {'ksplit'}),
    ('reshape NTU C_in 2048 b64', 64, 2048, 128, 8, {'ksplit'}),        # multi-round split-K (10.7 us; lds: 51 us)
    ('reshape NTU C_in 2048 b6', 6, 2048, 128, 8, {'ksplit'}),
    ('reshape NTU C_in 2048 b256', 256, 2048, 128, 8, {'pipe_fwd'}),
    # generic fallbacks for shapes outside the reference's configurations:
    ('channels not a multiple of 32, large grid: C_in 2064 b512', 512, 2064, 128, 8, {'lds'}),
]


@pytest.mark.parametrize('what,b,C_in,M,L,expect', CASES, ids=[c[0] for c in CASES])
def test_standalone_conv_families(what, b, C_in, M, L, expect):
    from bmnas import lib
    lib.conv_family_calls(reset=True)
    _conv_case(b, C_in, M, L)
    got = {k for k, v in lib.conv_family_calls().items() if v > 0}
    assert expect <= got, (what, got)
    assert 'conv_w' in got


@pytest.mark.parametrize('name,batch,expect', [
    ('mmimdb', 128, {'fwd_sdpa_pipe', 'bwd_all_pipe'}),
    ('mmimdb', 8, {'fwd_sdpa_ksplit', 'bwd_all_ksplit'}),
    ('ntu', 64, {'fwd_sdpa_ksplit', 'bwd_all_ksplit', 'bwd_pair'}),
])
def test_merged_launch_families(name, batch, expect):
    """One search step; the parity of these same shapes is pinned by test_network_gpu.py
    (test_search_hypernet_matches_oracle_real_configs) — here: which kernels ran."""
    from bmnas import lib
    cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.0})
    net = build_search_net(cfg, 3, 'train_nodrop')
    xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, 3)]
    lib.conv_family_calls(reset=True)
    net(xs).sum().backward()
    torch.cuda.synchronize()
    got = {k for k, v in lib.conv_family_calls().items() if v > 0}
    assert expect <= got, (name, batch, got)


def test_cat_of_two_sources_with_long_contraction_takes_the_generic_kernel():
    """lds, the generic fallback, also at a tiny grid: a two-source (fused torch.cat) K = 2048 contraction,
    which the multi-round split-K kernel (single source) does not cover."""
    from bmnas import lib
    lib.conv_family_calls(reset=True)
    _conv_case(6, 2048, 128, 8, n_src=2)
    got = {k for k, v in lib.conv_family_calls().items() if v > 0}
    assert 'lds' in got, got


def test_every_family_is_reachable(monkeypatch):
    """Union over the cases above: no dead GEMM family."""
    from bmnas import cell, lib
    lib.conv_family_calls(reset=True)
    _conv_case(6, 2048, 128, 8, n_src=2)
    for what, b, C_in, M, L, _ in CASES:
        _conv_case(b, C_in, M, L)
    for name, batch in (('mmimdb', 128), ('mmimdb', 8), ('ntu', 8)):      # ntu: out_conv -> bwd_pair
        cfg = fo.Cfg({**fo.CONFIGS[name], 'drpt': 0.0})
        net = build_search_net(cfg, 3, 'train_nodrop')
        xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, batch, 3)]
        net(xs).sum().backward()
    # the grouped reshape-layer launches (fwd_group / bwd_group; parity: tests/test_reshape_group_gpu.py)
    import models.auxiliary.aux_models as aux

    class A:
        drpt = 0.0

    layers = [aux.ReshapeInputLayer(c_in, 32, 8, A()).to(dev()).train() for c_in in (64, 32)]
    feats = [torch.randn(4, c_in, 8, device=dev(), requires_grad=True) for c_in in (64, 32)]
    sum(o.sum() for o in aux.reshape_tails(layers, feats)).backward()
    # ... and their multi-quad forward: long contractions (K >= 1024) on about one tile per CU (NTU at 64 samples)
    layers = [aux.ReshapeInputLayer(c_in, 128, 8, A()).to(dev()).train() for c_in in (1024, 2048, 1024)]
    feats = [torch.randn(64, c_in, 8, device=dev(), requires_grad=True) for c_in in (1024, 2048, 1024)]
    sum(o.sum() for o in aux.reshape_tails(layers, feats)).backward()
    torch.cuda.synchronize()
    calls = lib.conv_family_calls()
    dead = [k for k, v in calls.items() if v == 0]
    assert not dead, (dead, calls)


_ANY_TOGGLE = [k for k in os.environ if k.startswith('BMNAS_') and k not in ('BMNAS_DEFAULT',)]


@pytest.mark.skipif(bool(_ANY_TOGGLE), reason=f'launch counts describe the default switches ({_ANY_TOGGLE})')
@pytest.mark.parametrize('name,batch', [('ntu', 8), ('ego', 6)])
def test_small_batch_launch_merges_are_taken(name, batch, monkeypatch):
    """At 6-8 samples per GPU (BASELINE configs 4 / 5) the step is its launch count: the merges of round 2 must
    actually be dispatched there — the next cell step's K1 pair sum / its backward inside the NodeCell tail
    launches (bmnas_bn_relu_ln_fwd_pair / _bwd_pair) and the last inner step's mix backward inside the out_conv
    data-gradient tiles (bmnas_conv1x1_bwd_all_mix).  Counted at the Python boundary, with the merges on and off;
    parity of both forms is what the network tests check."""
    import torch
    from bmnas import lib, cell as K
    from gpu_util import build_search_net
    from oracle import fusion_oracle as fo, synth
    cfg = fo.CONFIGS[name]
    net = build_search_net(cfg, 5, 'train')
    xs = [x.cuda().requires_grad_(True) for x in synth.make_inputs(cfg, batch, 5)]
    counted = ('node_mix_bwd', 'mixsum_pair_fwd', 'mixsum_pair_bwd', 'bn_relu_ln_fwd', 'bn_relu_ln_bwd',
               'conv1x1_bwd_all', 'node_mix_fwd', 'node_mix_conv_fwd', 'conv1x1_fwd')
    calls = {}

    def wrap(fn_name):
        fn = getattr(lib, fn_name)

        def w(*a, **k):
            calls[fn_name] = calls.get(fn_name, 0) + 1
            return fn(*a, **k)
        return w
    for n in counted:
        monkeypatch.setattr(lib, n, wrap(n))

    def step():
        calls.clear()
        for p in list(net.parameters()) + list(net.arch_parameters()) + xs:
            p.grad = None
        net(xs).square().mean().backward()
        torch.cuda.synchronize()
        return dict(calls)
    S, ns = cfg.S, cfg.ns
    on = step()
    assert on.get('mixsum_pair_fwd', 0) == 0                 # first in the prologue launch, the rest in the tails
    assert on['mixsum_pair_bwd'] == 1                        # only the first cell step's (no node before it)
    assert on['node_mix_bwd'] == S * (ns - 1)                # the last inner step's rides in conv1x1_bwd_all
    assert on['bn_relu_ln_fwd'] == S and on['bn_relu_ln_bwd'] == S and on['conv1x1_bwd_all'] == S
    # round 3: the last inner step's mix forward rides in the out_conv launch (bmnas_node_mix_conv_fwd)
    if K.FUSE_MIX_GEMM and K.FUSE_PROLOGUE and K.FUSE_BN_FINALIZE:
        assert on['node_mix_conv_fwd'] == S and on.get('conv1x1_fwd', 0) == 0 and on['node_mix_fwd'] == S * (ns - 1)
    monkeypatch.setattr(K, 'FUSE_NEXT_PAIR', False)
    monkeypatch.setattr(K, 'FUSE_MIX_EPILOGUE', False)
    monkeypatch.setattr(K, 'FUSE_MIX_GEMM', False)
    off = step()
    assert off['mixsum_pair_fwd'] == S - 1 and off['mixsum_pair_bwd'] == S
    assert off['node_mix_bwd'] == S * ns
    assert off.get('node_mix_conv_fwd', 0) == 0 and off['conv1x1_fwd'] == S and off['node_mix_fwd'] == S * ns
