"""Lazy-LayerNorm path vs the per-sample kernels vs the float64 oracle, per tensor (diagnostics; prints, no asserts)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
from oracle import fusion_oracle as fo, synth
import test_lazy_ln_gpu as T


def rel(a, b):
    a, b = a.double().cpu().numpy(), np.asarray(b.detach().double().cpu().numpy() if torch.is_tensor(b) else b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


for case in [T.CASES[i] for i in (int(x) for x in (sys.argv[1:] or ['1']))]:
    N, C, L, S, M, ns, batch, nout, kind = case
    cfg = fo.make_cfg(N=N, C=C, L=L, S=S, M=M, ns=ns, nm=1, drpt=0.0)
    lz = T._step(cfg, batch, 5, nout, True, kind, 'train_nodrop')
    lz2 = T._step(cfg, batch, 5, nout, True, kind, 'train_nodrop')
    e1 = T._step(cfg, batch, 5, nout, False, kind, 'train_nodrop')
    e2 = T._step(cfg, batch, 5, nout, False, kind, 'train_nodrop')
    cw, cb = synth.make_classifier(cfg, nout, 5)
    args = (synth.make_inputs(cfg, batch, 5), synth.make_labels(kind, batch, nout, 5), synth.make_arch(cfg, 5),
            synth.make_params(cfg, 5), cw, cb, cfg, kind)
    _, _, og = fo.search_step(*args, training=True, attn_drop=0.0)
    print(case)
    for k in sorted(lz):
        ok = og.get(k if not k.startswith('p.') else k[2:])
        line = f'  {k:45s} lazy-eager {rel(lz[k], e1[k]):.1e}  eager-eager {rel(e1[k], e2[k]):.1e}  lazy-lazy {rel(lz[k], lz2[k]):.1e}'
        if ok is not None:
            line += f'  lazy-oracle {rel(lz[k], ok):.1e}  eager-oracle {rel(e1[k], ok):.1e}'
        if rel(lz[k], e1[k]) > 2e-5:
            print(line)
