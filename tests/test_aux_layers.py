"""Reshape layers and the cosine scheduler against the reference's golden
outputs: scheduler on CPU; the layers on the GPU (their conv+BN+ReLU tail runs on the HIP GEMM)."""
import json

import numpy as np
import pytest
import torch

from util import golden_files


def _load():
    z = np.load(golden_files('aux_layers.npz')[0])
    return z, json.loads(str(z['meta']))


def test_cosine_scheduler_matches_reference():
    import models.auxiliary.scheduler as sc
    z, _ = _load()
    s = sc.LRCosineAnnealingScheduler(1e-3, 1e-6, 1, 2, 7.5)
    got = np.array([s.step() for _ in range(60)])
    assert np.allclose(got, z['sched'], rtol=1e-12, atol=0)
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(2))], lr=1.0)
    s.update_optimizer(opt)
    assert opt.param_groups[0]['lr'] == s.eta


@pytest.mark.gpu
def test_reshape_layers_match_reference_golden():
    import models.auxiliary.aux_models as aux
    from gpu_util import assert_close_scaled
    z, meta = _load()

    class A:
        drpt = 0.1

    for m in meta:
        layer = getattr(aux, m['cls'])(m['c_in'], m['C'], m['L'], A())
        rng = np.random.Generator(np.random.PCG64(77))
        C, c_in = m['C'], m['c_in']
        sd = {'conv.weight': (rng.uniform(-1, 1, (C, c_in, 1)) / np.sqrt(c_in)).astype(np.float32),
              'conv.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
              'bn.weight': (1 + 0.1 * rng.standard_normal(C)).astype(np.float32),
              'bn.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
              'bn.running_mean': (0.1 * rng.standard_normal(C)).astype(np.float32),
              'bn.running_var': (1 + 0.2 * np.abs(rng.standard_normal(C))).astype(np.float32),
              'bn.num_batches_tracked': np.zeros((), np.int64)}
        layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        layer.cuda()
        layer.train(m['mode'] != 'eval')
        if m['mode'] == 'train_nodrop':
            layer.dropout.p = 0.0
        x = torch.from_numpy(rng.standard_normal(tuple(m['shape'])).astype(np.float32)).cuda().requires_grad_(True)
        y = layer(x)
        w = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
        (y * w).sum().backward()
        k = m['key']
        assert_close_scaled(k + ':y', y, z[k + ':y'])
        assert_close_scaled(k + ':dx', x.grad, z[k + ':dx'], rel=2e-4)
        assert_close_scaled(k + ':dconv_w', layer.conv.weight.grad, z[k + ':dconv_w'], rel=2e-4)
        assert_close_scaled(k + ':dbn_w', layer.bn.weight.grad, z[k + ':dbn_w'], rel=2e-4)
        assert_close_scaled(k + ':dbn_b', layer.bn.bias.grad, z[k + ':dbn_b'], rel=2e-4)
        assert_close_scaled(k + ':rm', layer.bn.running_mean, z[k + ':rm'])
        assert_close_scaled(k + ':rv', layer.bn.running_var, z[k + ':rv'])


@pytest.mark.gpu
def test_reshape_layers_at_production_widths_match_reference_golden():
    """C_in in {512, 1024, 2048} -> C 192 / 128 (the real backbones' feature widths,
    mmimdb_darts_searchable.py:86, ntu_darts_searchable.py:104, ego_darts_searchable.py:104):
    K = C_in up to 2048 through the conv + BN + ReLU kernels; fixtures in summary form."""
    import models.auxiliary.aux_models as aux
    from gpu_util import assert_close_scaled, assert_summary_scaled
    z = np.load(golden_files('aux_layers_big.npz')[0])
    meta = json.loads(str(z['meta']))

    class A:
        drpt = 0.1

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)       # leaving the HIP path would warn: make it fail
        for m in meta:
            layer = getattr(aux, m['cls'])(m['c_in'], m['C'], m['L'], A())
            rng = np.random.Generator(np.random.PCG64(m['seed']))
            C, c_in = m['C'], m['c_in']
            sd = {'conv.weight': (rng.uniform(-1, 1, (C, c_in, 1)) / np.sqrt(c_in)).astype(np.float32),
                  'conv.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
                  'bn.weight': (1 + 0.1 * rng.standard_normal(C)).astype(np.float32),
                  'bn.bias': (0.1 * rng.standard_normal(C)).astype(np.float32),
                  'bn.running_mean': (0.1 * rng.standard_normal(C)).astype(np.float32),
                  'bn.running_var': (1 + 0.2 * np.abs(rng.standard_normal(C))).astype(np.float32),
                  'bn.num_batches_tracked': np.zeros((), np.int64)}
            layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            layer.cuda()
            layer.train(m['mode'] != 'eval')
            if m['mode'] == 'train_nodrop':
                layer.dropout.p = 0.0
            x = torch.from_numpy(rng.standard_normal(tuple(m['shape'])).astype(np.float32)).cuda().requires_grad_(True)
            y = layer(x)
            w = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
            (y * w).sum().backward()
            k = m['key']
            assert_summary_scaled(k + ':y', y, z[k + ':y'])
            assert_summary_scaled(k + ':dx', x.grad, z[k + ':dx'], rel=3e-4)
            assert_summary_scaled(k + ':dconv_w', layer.conv.weight.grad, z[k + ':dconv_w'], rel=3e-4)
            assert_close_scaled(k + ':dbn_w', layer.bn.weight.grad, z[k + ':dbn_w'], rel=3e-4)
            assert_close_scaled(k + ':dbn_b', layer.bn.bias.grad, z[k + ':dbn_b'], rel=3e-4)
            assert_close_scaled(k + ':rm', layer.bn.running_mean, z[k + ':rm'])
            assert_close_scaled(k + ':rv', layer.bn.running_var, z[k + ':rv'])
