"""cProfile of the eager (no hipGraph) search step: where the host time goes."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd')):
    sys.path.insert(0, p)
import torch
import bench

c = bench.CONFIGS['mmimdb']
dev = torch.device('cuda', 0)
torch.manual_seed(2)
model = bench.HyperNet(c).to(dev).train()
from bmnas import nn as bnn
crit = bnn.BCEWithLogitsLoss()
xs, y = bench.synth_batch(c, 128, dev, 0)
leaves = list(model.parameters()) + list(model.arch_parameters()) + xs


def step():
    for t in leaves:
        t.grad = None
    loss = crit(model(xs), y)
    loss.backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print('eager ms/step', (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
