"""Deterministic synthetic parameters / inputs for the golden fixtures.  TEST
INFRASTRUCTURE ONLY (see oracle/fusion_oracle.py header).

Everything is drawn from ``numpy.random.Generator(PCG64(seed))`` in a fixed key
order, so the golden generator (build container, imports the reference) and the
parity tests (GPU box, reference absent) regenerate bit-identical tensors without
shipping them; torch RNG streams are never relied on.
"""
from __future__ import annotations

import numpy as np
import torch

from . import fusion_oracle as fo


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _fill(rng, key, shape):
    """Non-trivial values for every tensor kind so no term of the math is hidden
    (LN/BN affines away from 1/0, running stats away from 0/1)."""
    if key.endswith('num_batches_tracked'):
        return np.zeros((), dtype=np.int64)
    if key.endswith('running_mean'):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if key.endswith('running_var'):
        return (1.0 + 0.2 * np.abs(rng.standard_normal(shape))).astype(np.float32)
    if key.endswith('conv.weight'):
        fan_in = shape[1]
        return (rng.uniform(-1.0, 1.0, shape) / np.sqrt(fan_in)).astype(np.float32)
    if key.endswith('conv.bias'):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if key.endswith('linear.weight'):    # FC_Relu / FC_Mish (round-2 fixtures)
        return (rng.uniform(-1.0, 1.0, shape) / np.sqrt(shape[1])).astype(np.float32)
    if key.endswith('linear.bias'):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if key.endswith('.weight'):          # LN / BN scale
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if key.endswith('.bias'):            # LN / BN shift
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    raise KeyError(key)


def make_params(cfg, seed, shapes=None):
    """state_dict-keyed float32 tensors for the search hypernet (or any shapes dict)."""
    rng = _rng(seed)
    shapes = fo.param_shapes(cfg) if shapes is None else shapes
    return {k: torch.from_numpy(_fill(rng, k, s)) for k, s in shapes.items()}


def make_arch(cfg, seed, scale=0.5, primitives=None):
    """alphas/betas/gammas; scale 0.5 (not the reference's 1e-3 init) so the softmax
    weights are far from uniform and every arch-gradient term is exercised."""
    rng = _rng(seed + 1000003)
    return [torch.from_numpy((scale * rng.standard_normal(s)).astype(np.float32))
            for s in fo.arch_shapes(cfg, primitives)]


def make_inputs(cfg, batch, seed):
    """N tensors (batch, C, L): relu(N(0,1)) like the reshape layers' outputs
    (aux_models.py:112-114)."""
    rng = _rng(seed + 2000003)
    return [torch.from_numpy(np.maximum(rng.standard_normal((batch, cfg.C, cfg.L)), 0.0)
                             .astype(np.float32)) for _ in range(cfg.N)]


def make_classifier(cfg, num_outputs, seed):
    rng = _rng(seed + 3000003)
    fan_in = cfg.M * cfg.C * cfg.L
    w = (rng.uniform(-1, 1, (num_outputs, fan_in)) / np.sqrt(fan_in)).astype(np.float32)
    b = (rng.uniform(-1, 1, (num_outputs,)) / np.sqrt(fan_in)).astype(np.float32)
    return torch.from_numpy(w), torch.from_numpy(b)


def make_labels(kind, batch, num_outputs, seed):
    """'bce': multi-hot float (batch, num_outputs) Bernoulli(0.2) (MM-IMDB);
    'ce': int64 (batch,) uniform class ids (NTU / EgoGesture)."""
    rng = _rng(seed + 4000003)
    if kind == 'bce':
        return torch.from_numpy((rng.uniform(size=(batch, num_outputs)) < 0.2).astype(np.float32))
    return torch.from_numpy(rng.integers(0, num_outputs, size=(batch,)).astype(np.int64))


def make_drop_mask(seed, site, shape, p):
    """Multipliers of dropout site number `site` of a step (0 for a dropped element, 1/(1-p) for a kept one):
    what nn.Dropout(p) applies in train mode, with the Bernoulli(1-p) draw coming from this module's
    counter-keyed generator instead of torch's stream (round-3 'train_drop' fixtures)."""
    rng = _rng(seed + 5000003 + 7919 * site)
    keep = rng.uniform(size=tuple(shape)) >= p
    return torch.from_numpy(keep.astype(np.float32) * np.float32(1.0 / (1.0 - p)))
