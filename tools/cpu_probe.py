import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import fusion_oracle as fo, synth
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads(), flush=True)
cfg = fo.CONFIGS['mmimdb']
p = synth.make_params(cfg, 2); arch = synth.make_arch(cfg, 2, 1e-3)
cw, cb = synth.make_classifier(cfg, 23, 2)
xs = synth.make_inputs(cfg, 128, 0); y = synth.make_labels('bce', 128, 23, 0)
for nt in [int(a) for a in sys.argv[1:]]:
    torch.set_num_threads(nt)
    ts = []
    t_all = time.time()
    for i in range(12):
        t0 = time.perf_counter()
        fo.search_step(xs, y, arch, p, cw, cb, cfg, 'bce', training=True)
        ts.append(time.perf_counter() - t0)
        if time.time() - t_all > 25: break
    ts = sorted(ts[2:]) if len(ts) > 4 else ts
    print(f'threads {nt}: median {ts[len(ts)//2]*1e3:.1f} ms over {len(ts)} steps', flush=True)
