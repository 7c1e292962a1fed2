"""Quick eager timing of one search step (fwd+bwd) of the product modules on cuda:0."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch

from oracle import fusion_oracle as fo, synth
from gpu_util import build_search_net

name = sys.argv[1] if len(sys.argv) > 1 else 'mmimdb'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nout = {'mmimdb': 23, 'ntu': 60, 'ego': 83}[name]
cfg = fo.CONFIGS[name]
net = build_search_net(cfg, 2, 'train', arch_scale=1e-3)
cls = torch.nn.Linear(cfg.M * cfg.C * cfg.L, nout).cuda()
xs = [x.cuda().requires_grad_(True) for x in synth.make_inputs(cfg, B, 0)]
kind = 'bce' if name == 'mmimdb' else 'ce'
y = synth.make_labels(kind, B, nout, 0).cuda()
crit = torch.nn.BCEWithLogitsLoss() if kind == 'bce' else torch.nn.CrossEntropyLoss()


def step():
    for p in net.parameters():
        p.grad = None
    loss = crit(cls(net(xs)), y)
    loss.backward()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.time()
n = 30
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f'{name} B={B}: {dt*1e3:.3f} ms/step eager  ({1/dt:.1f} steps/s, {B/dt:.0f} samples/s)')
