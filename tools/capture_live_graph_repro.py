"""Reproduces the condition GraphedTrainStep guards against: capturing a step while tensors of an
earlier eager autograd graph are still referenced (flag `del` drops them -> capture succeeds).  Without
the guard HIP's capture_end segfaults (the stale AccumulateGrad nodes pull the default stream into the
capture).  Flags: nostep noopt torchw bnnlin bnnloss bigC bigB drop sync mm L16 N6 late del."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import faulthandler; faulthandler.enable()
import torch
from oracle import fusion_oracle as fo, synth
from gpu_util import Args, dev, set_mode
from util import cfg_of, load_npz
import bmnas.optim
from bmnas import nn as bnn
from bmnas.graph import GraphedTrainStep
from models.search.darts.model_search import FusionNetwork
flags = set(sys.argv[1:])
meta, z = load_npz(os.path.join(ROOT, 'tests/golden/traj_a.npz'))
cfg = cfg_of(meta)
if 'mm' in flags:
    cfg = fo.Cfg({**fo.CONFIGS['mmimdb'], 'drpt': 0.1})
if 'L16' in flags:
    cfg = fo.Cfg({**dict(cfg), 'L': 16})
if 'N6' in flags:
    cfg = fo.Cfg({**dict(cfg), 'N': 6})
if 'bigC' in flags:
    cfg = fo.Cfg({**dict(cfg), 'C': 64})
seed, batch, nout = meta['seed'], (32 if 'bigB' in flags else meta['batch']), meta['num_outputs']

class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fusion_net = FusionNetwork(cfg.S, cfg.M, cfg.N, 2, Args(cfg), criterion=None)
        self.central_classifier = (bnn.Linear if 'bnnlin' in flags else torch.nn.Linear)(cfg.M * cfg.C * cfg.L, nout)
    def forward(self, xs):
        return self.central_classifier(self.fusion_net(list(xs)))
    def arch_parameters(self):
        return self.fusion_net.arch_parameters()

model = Net()
crit = bnn.BCEWithLogitsLoss() if 'bnnloss' in flags else torch.nn.BCEWithLogitsLoss()
if 'late' in flags:
    model.to(dev())
WAdam = torch.optim.Adam if 'torchw' in flags else bmnas.optim.Adam
opt = WAdam(model.parameters(), lr=1e-3, weight_decay=1e-4)
aopt = bmnas.optim.Adam(model.arch_parameters(), lr=3e-4, betas=(0.5, 0.999), weight_decay=1e-3)
if 'late' not in flags:
    model.to(dev())
set_mode(model, 'train' if 'drop' in flags else 'train_nodrop')
xs = [x.to(dev()) for x in synth.make_inputs(cfg, batch, seed)]
y = synth.make_labels('bce', batch, nout, seed).to(dev())
if 'nostep' not in flags:
    opt.zero_grad()
    logits = model(xs)
    crit(logits, y).backward()
    if 'noopt' not in flags:
        opt.step()
    if 'del' in flags:
        del logits
if 'sync' in flags:
    torch.cuda.synchronize()
print('capturing', sorted(flags), flush=True)
g = GraphedTrainStep(model, crit, aopt, xs, y)
print('built', flush=True)
g(xs, y); torch.cuda.synchronize()
print('replayed ok', flush=True)
