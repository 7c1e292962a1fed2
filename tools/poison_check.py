"""Stale-read check: every fresh float32 device allocation (torch.empty / empty_like — what the kernel sequencing
uses for outputs and gradient slots) is filled with NaN before use, then one MM-IMDB step runs through the
lazy-LayerNorm path and through the per-sample path.  A kernel that reads a destination it was told to overwrite
(an accumulate flag set on a fresh slot, an old value fetched before an aliasing store) turns its results into NaN.
    python tools/poison_check.py            (one GPU; expected: "non-finite tensors: [] 0" twice)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT,'bm-nas_amd'), os.path.join(ROOT,'tests')):
    sys.path.insert(0,p)
import torch
_el, _e = torch.empty_like, torch.empty
def el(t, *a, **k):
    r=_el(t,*a,**k)
    if r.is_cuda and r.dtype==torch.float32: r.fill_(float('nan'))
    return r
def e(*a, **k):
    r=_e(*a,**k)
    if r.is_cuda and r.dtype==torch.float32: r.fill_(float('nan'))
    return r
torch.empty_like, torch.empty = el, e
from oracle import fusion_oracle as fo
from test_lazy_ln_gpu import _step
cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'drpt': 0.0})
for lazy in (True, False):
    out=_step(cfg, 32, 11, 23, lazy, 'bce', 'train_nodrop')
    bad=[k for k,v in out.items() if not torch.isfinite(v).all()]
    print('lazy', lazy, 'non-finite tensors:', bad[:8], len(bad))
