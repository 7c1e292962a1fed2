"""Shared helpers for the parity tests (golden loading, oracle drivers)."""
import glob
import json
import os

import numpy as np
import torch

from oracle import fusion_oracle as fo
from oracle import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_files(pattern):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load_npz(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z['meta']))
    return meta, z


def case_id(path):
    return os.path.basename(path).replace('.npz', '')


def cfg_of(meta, mode=None):
    cfg = fo.Cfg(meta['cfg'])
    return cfg


def mode_flags(mode):
    """-> (training, drpt_override, attn_drop): golden 'train_nodrop' = train-mode BN with
    every dropout an identity; 'eval' = eval mode."""
    if mode == 'eval':
        return False, None, fo.ATTN_DROP
    if mode == 'train_drop':             # round 3: every dropout live, masks injected (golden_masks)
        return True, None, fo.ATTN_DROP
    return True, 0.0, 0.0


def golden_masks(meta):
    """The dropout multipliers of a 'train_drop' fixture's live sites, in execution order, regenerated from
    the seeds the generator used (tests/golden/make_golden_r03.py: site k of the step ->
    synth.make_drop_mask(seed, k, shape, p); sites with p = 1e-12 are identities and are skipped)."""
    return [synth.make_drop_mask(meta['seed'], k, s['shape'], s['p'])
            for k, s in enumerate(meta['sites']) if s['p'] > 1e-6]


def summarize(t, nsample=8):
    f = t.detach().double().reshape(-1).cpu()
    head = f[:nsample]
    if head.numel() < nsample:
        head = torch.cat([head, torch.zeros(nsample - head.numel(), dtype=torch.float64)])
    return torch.cat([f.sum()[None], f.norm()[None], head]).numpy()


def assert_close(name, got, want, rtol=1e-4, atol=1e-5):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    if not (err <= tol).all():
        i = int(np.argmax(err - tol))
        raise AssertionError(f'{name}: max|err|={err.max():.3e} at flat {i}: got {got.reshape(-1)[i]!r} '
                             f'want {want.reshape(-1)[i]!r} (rtol={rtol}, atol={atol})')


def assert_summary_close(name, got_tensor, want_summary, rtol=1e-4):
    """Compare against a [sum, l2, first-8] summary; sums of many terms get an atol scaled
    by the tensor's l2 norm (cancellation)."""
    got = summarize(got_tensor, len(want_summary) - 2)
    l2 = max(abs(float(want_summary[1])), 1e-12)
    n = got_tensor.numel()
    assert_close(name + ':sum', got[0], want_summary[0], rtol=rtol, atol=rtol * l2 * max(1.0, np.sqrt(n)) * 1e-1 + 1e-6)
    assert_close(name + ':l2', got[1], want_summary[1], rtol=rtol, atol=1e-6)
    assert_close(name + ':head', got[2:], want_summary[2:], rtol=rtol, atol=rtol * l2 / np.sqrt(max(n, 1)) + 1e-6)


def grad_atol(key, default=2e-6):
    """The conv bias in front of a train-mode BatchNorm has a mathematically ZERO gradient
    (BN subtracts the batch mean); what the reference stores there is fp32 round-off of
    size ~1e-6, so those entries are compared with an absolute tolerance only."""
    if key.endswith('conv.bias'):
        return 1e-4
    return default
