import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd')):
    sys.path.insert(0, p)
import torch
import bench as B
from bmnas import nn as bnn
from bmnas.graph import GraphedStep
for cname, batch in (('mmimdb', 128), ('ntu', 8)):
    c = B.CONFIGS[cname]; dev = torch.device('cuda:0')
    torch.manual_seed(2)
    model = B.HyperNet(c, 'R', cname).to(dev).train()
    crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
    xs, y = B.synth_batch(c, batch, dev, 0, 'R', cname)
    def fwd():
        with torch.no_grad():
            out = model(xs)
            return crit(out, y), out
    for _ in range(20): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fwd()
    torch.cuda.synchronize(); e = (time.perf_counter() - t0) / 200 * 1e3
    g = GraphedStep(fwd)
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): g.replay()
    torch.cuda.synchronize(); r = (time.perf_counter() - t0) / 500 * 1e3
    print(cname, batch, 'no-grad forward (tier R): eager %.3f ms, graph %.3f ms' % (e, r), flush=True)
