"""CPU: the C-ABI shared library builds for gfx950, loads without a GPU, and exports every
entry point that include/bmnas_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'bmnas_hip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(bmnas_\w+)\s*\(', text)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for must in ('bmnas_mixsum_fwd', 'bmnas_mixsum_bwd', 'bmnas_sdpa_ln_fwd', 'bmnas_sdpa_ln_bwd',
                 'bmnas_conv1x1_fwd', 'bmnas_conv1x1_bwd_data', 'bmnas_conv1x1_bwd_weight',
                 'bmnas_cat_ln_fwd', 'bmnas_cat_ln_bwd', 'bmnas_node_mix_fwd', 'bmnas_node_mix_bwd',
                 'bmnas_node_mix_ln_bwd', 'bmnas_bn_relu_ln_fwd_pair', 'bmnas_bn_relu_ln_bwd_pair',
                 'bmnas_bn_finalize', 'bmnas_arch_softmax_multi', 'bmnas_version'):
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    from bmnas import build, lib
    path = build.build()
    assert os.path.exists(path)
    so = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(so, s)]
    assert not missing, missing
    # the ctypes binding covers the same set
    assert sorted(lib.SIGNATURES) == declared_symbols()
    assert lib.load().bmnas_version() >= 100


def test_library_is_gfx950_code():
    from bmnas import build
    data = open(build.build(), 'rb').read()
    assert b'gfx950' in data


def test_product_path_fails_loudly_without_gpu_tensors():
    """No CPU fallback: the nn.Module mirror refuses CPU tensors instead of computing."""
    import torch
    from models.search.darts.model_search import FusionNetwork

    class A:
        C, L, drpt, num_input_nodes, node_steps, node_multiplier = 16, 8, 0.1, 3, 1, 1

    net = FusionNetwork(2, 2, 3, 2, A())
    xs = [torch.randn(2, 16, 8) for _ in range(3)]
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net(xs)


def test_missing_library_raises(monkeypatch, tmp_path):
    from bmnas import lib
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(lib.BmnasError, match='no fallback'):
        lib.load()
