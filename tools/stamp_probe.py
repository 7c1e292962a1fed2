"""Where the streaming kernels of the lazy-LayerNorm path spend their time, from INSIDE the kernels.

Needs a timing build:   BMNAS_HIPCC_EXTRA=-DBMNAS_BODY_PROBES=1 python -m bmnas.build   (build.py notices the flag change
and recompiles; a later plain build restores the production library — no stamp executes there).
Thread 0 of every workgroup stamps the shader clock (s_memtime) at a few points and the 100 MHz wall clock at entry /
exit (csrc/common.hpp STAMP).  Reported per kernel, over its workgroups:
  * dispatch skew: wall clock at entry relative to the first workgroup's (how long the grid takes to get going);
  * the kernel's duration seen from inside: last exit - first entry;
  * per segment (stamp i -> i + 1): median / 90th percentile shader cycles.
    python tools/stamp_probe.py [batch]         (MM-IMDB shape, eager steps, one GPU)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch

from bmnas import lib, nn as bnn
from oracle import fusion_oracle as fo, synth
from gpu_util import build_search_net, dev

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SLOTS = 4096
NAMES = {0: ('node_mix_pre_fwd_k', ['entry -> loads issued', 'BatchNorm finalise (+ barrier)', 'mix + store pre',
                                    'two block reductions + record']),
         1: ('mixsum_pair_fwd_lazy_k', ['whole body']),
         2: ('mixsum_pair_bwd_lazy_k', ['loads + wait', 'dots + dx stores', 'block reduction + atomics / partial stores']),
         3: ('node_mix_lnp_bwd_k', ['partials + operand loads + wave sums', 'LayerNorm + mix backward + stores',
                                    'BatchNorm sums: LDS + atomics', 'dgamma: block sum + atomics']),
         4: ('head_fwd_k', ['loads issued + statistics', 'MFMAs + tiles to LDS', 'barrier + workgroup sum + atomics']),
         5: ('head_bwd_k', ['scrub + loads + criterion prologue + barrier', 'GEMM 1 + LayerNorm backward + state gradients',
                            'LayerNorm partials: barrier + stores', 'affine partials + GEMM 2 + dW partial stores'])}

cfg = fo.CONFIGS['mmimdb']
net = build_search_net(cfg, 2, 'train')
cls = bnn.Linear(cfg.M * cfg.C * cfg.L, 23).to(dev())
xs = [x.to(dev()).requires_grad_(True) for x in synth.make_inputs(cfg, B, 0)]
y = synth.make_labels('bce', B, 23, 0).to(dev())
buf = torch.zeros(8 * SLOTS * 8, dtype=torch.int64, device=dev())
rc = lib.load().bmnas_debug_stamps(C.c_void_p(buf.data_ptr()), SLOTS) or \
    lib.load().bmnas_debug_stamps_head(C.c_void_p(buf.data_ptr()), SLOTS) or \
    lib.load().bmnas_debug_stamps_conv(C.c_void_p(buf.data_ptr()), SLOTS)
if rc != 0:
    raise SystemExit('this library has no stamps: build with BMNAS_HIPCC_EXTRA=-DBMNAS_BODY_PROBES=1 (rc %d)' % rc)


def step():
    for t in list(net.parameters()) + list(net.arch_parameters()) + xs + list(cls.parameters()):
        t.grad = None
    with bnn.fused_criterion():
        loss = bnn.BCEWithLogitsLoss()(net.forward_classified(xs, cls), y)
    loss.backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
buf.zero_()
step()
torch.cuda.synchronize()
data = buf.cpu().numpy().astype(np.uint64).reshape(8, SLOTS, 8)
# the shader clock against the wall clock, from the longest kernel: cycles per 10 ns tick
print(f'# in-kernel stamps, MM-IMDB b{B}, one eager step (the LAST launch of each instrumented kernel is what the buffer '
      f'holds: cell step 0 for the forward kernels\' second launch... see below)')
for slot, (name, segs) in NAMES.items():
    d = data[slot]
    used = d[:, 6] != 0
    if not used.any():
        print(f'{name}: not launched')
        continue
    d = d[used]
    n = len(d)
    w0 = d[:, 6].astype(np.int64)
    w1 = d[:, 7].astype(np.int64)
    first = w0.min()
    skew = (w0 - first) * 0.01                      # us
    dur = (w1.max() - first) * 0.01
    last = len(segs)
    cyc_total = (d[:, last].astype(np.int64) - d[:, 0].astype(np.int64))
    own = (w1 - w0) * 0.01
    ghz = np.median(cyc_total[own > 0] / (own[own > 0] * 1e3)) if (own > 0).any() else float('nan')
    print(f'{name}: {n} workgroups; seen from inside {dur:.2f} us (first entry -> last exit); shader clock ~{ghz:.2f} GHz')
    print(f'   dispatch skew (entry after the first workgroup): median {np.median(skew):.2f} us, 90% {np.percentile(skew, 90):.2f}, '
          f'max {skew.max():.2f}')
    print(f'   a workgroup\'s own duration: median {np.median(own):.2f} us, 90% {np.percentile(own, 90):.2f}, max {own.max():.2f}')
    for i, sname in enumerate(segs):
        c = d[:, i + 1].astype(np.int64) - d[:, i].astype(np.int64)
        print(f'   {sname:45s} median {np.median(c):8.0f} cyc ({np.median(c) / (ghz * 1e3):5.2f} us)   90% {np.percentile(c, 90):8.0f}')
# slot 6: the data-gradient tiles of the merged backward GEMM launch — cycles summed over a tile's chunks
d = data[6]
d = d[d[:, 6] != 0]
if len(d):
    tot = d[:, 3].astype(np.float64)
    # (since the k-block loop is software-pipelined the operand reads are no longer a separate phase: column 0 stays 0)
    print(f'conv_bwd_all_pipe_k data-gradient tiles: {len(d)} workgroups x {int(d[0, 7])} chunks; per tile (median cycles): '
          f'LDS reads + MFMAs + stash slices {np.median(d[:, 0] + d[:, 1]):.0f} ({np.median((d[:, 0] + d[:, 1]) / tot):.0%}), '
          f'fetch issue + barrier + rest {np.median(d[:, 2]):.0f} '
          f'({np.median(d[:, 2] / tot):.0%}), whole chunk loop {np.median(tot):.0f} cycles')
