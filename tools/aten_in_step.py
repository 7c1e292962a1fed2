"""Which launches of a captured training step are NOT this repo's kernels, and who issues them.

    python tools/aten_in_step.py found mmimdb 128      # bench.py --stage found's step (FoundNet + Adam over everything)
    python tools/aten_in_step.py found ntu 64
    python tools/aten_in_step.py search mmimdb 128     # the w-step of full_search_step

Runs the step's function eagerly (what GraphedTrainStep captures: forward, criterion, torch.autograd.grad, Adam)
under torch.profiler and prints every device kernel / memcpy that does not come from libbmnas_hip.so with the aten op
and the Python frames of this repository that led to it.  One GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from bmnas import nn as bnn
from bmnas.functions import unit_grad
from bmnas.optim import Adam

stage, cname, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
c = bench.CONFIGS[cname]
dev = torch.device('cuda', 0)
torch.manual_seed(2)
model = (bench.FoundNet(c, cname) if stage == 'found' else bench.HyperNet(c, 'F', cname)).to(dev).train()
crit = bnn.BCEWithLogitsLoss() if c['loss'] == 'bce' else bnn.CrossEntropyLoss()
xs, y = bench.synth_batch(c, B, dev, 0)
xs = [x.detach() for x in xs]
params = list(model.parameters())
opt = Adam(params, lr=1e-3, weight_decay=1e-4)


def fn():
    with bnn.fused_criterion():
        logits = model(xs)
        loss = crit(logits, y)
    grads = torch.autograd.grad(loss, params, grad_outputs=unit_grad(dev), allow_unused=True)
    for t, g in zip(params, grads):
        t.grad = g
    opt.step()
    return loss


for _ in range(3):
    fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    fn()
    torch.cuda.synchronize()
ours = 0
seen = {}
for e in prof.events():
    ks = getattr(e, 'kernels', None) or []
    if not ks:
        continue
    for k in ks:
        name = k.name
        if '_k(' in name or '_k<' in name or name.endswith('_k'):
            ours += 1
            continue
        frames = [f for f in (e.stack or []) if 'site-packages' not in f and 'dist-packages' not in f
                  and 'tools/aten_in_step' not in f and not f.startswith('<built-in')][:5]
        key = (e.name, name[:60], tuple(frames))
        seen[key] = seen.get(key, 0) + 1
print(f'{stage} {cname} b{B}: {ours} launches of this repo\'s kernels; foreign launches:')
for (op, kname, frames), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f'  x{n}  {op}  ->  {kname}')
    for f in frames:
        print(f'        {f}')
