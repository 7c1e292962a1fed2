import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import faulthandler; faulthandler.enable()
import torch
import bench
from bmnas import nn as bnn
from bmnas.optim import Adam
from bmnas.graph import GraphedTrainStep
variant = sys.argv[1] if len(sys.argv) > 1 else 'plain'
c = bench.CONFIGS['mmimdb']
dev = torch.device('cuda', 0)
model = bench.HyperNet(c).to(dev).train()
crit = bnn.BCEWithLogitsLoss()
if 'aten' in variant:
    model.central_classifier = torch.nn.Linear(c['M'] * c['C'] * c['L'], c['nout']).to(dev)
    crit = torch.nn.BCEWithLogitsLoss()
if 'atenloss' in variant:
    crit = torch.nn.BCEWithLogitsLoss()
if 'atenlin' in variant:
    model.central_classifier = torch.nn.Linear(c['M'] * c['C'] * c['L'], c['nout']).to(dev)
xs, y = bench.synth_batch(c, 16, dev, 0)
xs = [x.detach() for x in xs]
opt = Adam(model.parameters(), lr=1e-3)
aopt = Adam(model.arch_parameters(), lr=1e-3, betas=(0.5, 0.999))
if 'nobackward' not in variant:
    opt.zero_grad()
    crit(model(xs), y).backward()
    opt.step()
if variant == 'gradnone':
    for p in model.parameters():
        p.grad = None
torch.cuda.synchronize()
print('building graph, variant', variant, flush=True)
g = GraphedTrainStep(model, crit, aopt, xs, y)
print('built', flush=True)
g(xs, y)
torch.cuda.synchronize()
print('replayed ok', flush=True)
