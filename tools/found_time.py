"""Timing of a Found_FusionNetwork (discrete net, x != y) training step on cuda:0:
eager fwd+bwd, and fwd + criterion + bwd + Adam as one hipGraph replay."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch

from oracle import fusion_oracle as fo, synth
from gpu_util import build_found_net, build_search_net

name = sys.argv[1] if len(sys.argv) > 1 else 'mmimdb'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nout = {'mmimdb': 23, 'ntu': 60, 'ego': 83}[name]
cfg = fo.CONFIGS[name]
search = build_search_net(cfg, 2, 'train', arch_scale=1e-1)
geno = search.genotype()
print(geno)
net = build_found_net(cfg, geno, 2, 'train')
from bmnas import nn as bnn
from bmnas.graph import GraphedTrainStep
from bmnas.optim import Adam


class Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fusion_net = net
        self.central_classifier = bnn.Linear(cfg.M * cfg.C * cfg.L, nout)

    def forward(self, xs):
        return self.central_classifier(self.fusion_net(list(xs)))


model = Model().cuda().train()
xs = [x.cuda() for x in synth.make_inputs(cfg, B, 0)]
kind = 'bce' if name == 'mmimdb' else 'ce'
y = synth.make_labels(kind, B, nout, 0).cuda()
crit = bnn.BCEWithLogitsLoss() if kind == 'bce' else bnn.CrossEntropyLoss()
opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)


def step():
    opt.zero_grad()
    loss = crit(model(xs), y)
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


print(f'{name} found net B={B}: eager {timeit(step):.3f} ms/step')
g = GraphedTrainStep(model, crit, opt, xs, y)
print(f'{name} found net B={B}: graph {timeit(lambda: g(xs, y), 200):.3f} ms/step')
