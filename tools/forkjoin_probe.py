"""What a fork / join pair inside a hipGraph costs on MI355X, separated from what runs on the forked branch.

Round 3 found the 'overlap' data-parallel shape (the last cell step's gradients all-reduced on a forked stream while
the first step's backward runs) 28 us slower per step than the in-graph all-reduce at the end, on one GPU — a single
number that did not say whether the graph BRANCH or RCCL's launch on the side stream was to blame.  This probe
captures a chain of 16 trivial kernels on the capture stream and adds P fork/join pairs whose side branch carries
  (a) one trivial kernel,
  (b) bmnas_allreduce_f32 on a world-size-1 communicator, 4 KB and 4.2 MB (the step's gradient bucket),
and, for each, the SERIAL twin: the same side work issued on the capture stream itself (no branch).  Per pair:
  branch cost = (forked - serial) / P.
    python tools/forkjoin_probe.py > profiles/r04_forkjoin_probe.txt      (one GPU)
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bm-nas_amd'))
import torch

from bmnas import dist as bdist

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
x = torch.zeros(4096, device=dev)
y = torch.zeros(4096, device=dev)
side = torch.cuda.Stream(dev)
small = torch.ones(1024, device=dev)
bucket = torch.ones(1_050_000, device=dev)          # 4.2 MB: the MM-IMDB step's flat gradient bucket
comm = bdist.NativeComm.get()                       # world-size-1 communicator (no process group)


def trivial():
    y.add_(1.0)


def build(pairs, work, forked, chain=16):
    """A captured graph: `chain` trivial kernels on the capture stream with `pairs` side jobs spread between them."""
    every = chain // max(pairs, 1)
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(dev)
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for _ in range(3):                          # warm-up outside the capture (lazy module loads)
            x.add_(1.0)
            work()
        cap.synchronize()
        with torch.cuda.graph(g, stream=cap):
            done = 0
            for i in range(chain):
                x.add_(1.0)
                if pairs and i % every == 0 and done < pairs:
                    done += 1
                    if forked:
                        side.wait_stream(cap)                       # fork
                        with torch.cuda.stream(side):
                            work()
                    else:
                        work()
            if forked and pairs:
                cap.wait_stream(side)                               # join (once: every branch ends on `side`)
    torch.cuda.current_stream().wait_stream(cap)
    return g


def timed(g, n=400, rounds=7):
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e6)
    return statistics.median(out)


works = [('trivial kernel', trivial),
         ('all-reduce 4 KB (world 1)', lambda: comm.all_reduce(small, average=True)),
         ('all-reduce 4.2 MB (world 1)', lambda: comm.all_reduce(bucket, average=True))]
print('# hipGraph fork/join probe (MI355X, ROCm %s): 16 trivial kernels on the capture stream + P side jobs' % torch.version.hip)
print('# us per replay (median of 7 x 400 replays); branch cost per pair = (forked - serial) / P')
base = timed(build(0, trivial, False))
print(f'chain alone: {base:.2f} us  ({base / 16:.2f} us per trivial kernel)')
for name, fn in works:
    for pairs in (1, 2, 4):
        ser = timed(build(pairs, fn, False))
        frk = timed(build(pairs, fn, True))
        print(f'{name:30s} P={pairs}: serial {ser:7.2f}  forked {frk:7.2f}  side work {(ser - base) / pairs:6.2f} us each  '
              f'branch cost {(frk - ser) / pairs:+6.2f} us per pair')
