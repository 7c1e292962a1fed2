"""Where the weight step's time beyond fwd + bwd goes: GraphedTrainStep (fwd + criterion + bwd + Adam) replayed
with (a) the full optimizer launch, (b) the Adam kernel alone (poke-mode plans hold no H2D copy node any more: same as (a)), (c) no optimizer work.
    python tools/time_wstep.py [config] [batch] [variant ...]     (one variant: a clean rocprofv3 --kernel-trace --stats)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd')):
    sys.path.insert(0, p)
import torch

import bench as B
from bmnas import nn as bnn
from bmnas.graph import GraphedTrainStep
from bmnas.optim import Adam

cname = sys.argv[1] if len(sys.argv) > 1 else 'mmimdb'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
c = B.CONFIGS[cname]
dev = torch.device('cuda:0')


def build(variant):
    torch.manual_seed(2)
    model = B.HyperNet(c, 'F', cname).to(dev).train()
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    xs, y = B.synth_batch(c, batch, dev, 0)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    if variant == 'nocopy':
        opt._launch = lambda: __import__('bmnas').lib.adam_multi(opt._plan['dev_tab'], opt._plan['chunks'],
                                                                 opt._plan['n_chunks'], opt._plan['dev_hyp'])
    elif variant == 'nostep':
        opt._launch = lambda: None
    g = GraphedTrainStep(model, crit, opt, xs, y)
    return g, xs, y


for variant in (sys.argv[3:] or ('full', 'nocopy', 'nostep')):
    g, xs, y = build(variant)
    for _ in range(20):
        g(xs, y)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            g(xs, y)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
    # no input copy: the step called on its own static tensors.  (Not `g._g.replay()`: a captured step depends on the launch
    # in front of every replay — it clears the accumulation arena, advances the dropout counter and delivers Adam's
    # scalars; raw replays accumulate into an arena nobody clears and reach Inf / NaN.)
    sb = g.static_batch()
    for _ in range(20):
        g(*sb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g(*sb)
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 200 * 1e3
    print(f'{cname} b{batch} {variant:7s}: call {best:.4f} ms   without input copy {rep:.4f} ms', flush=True)
