"""A grid-wide barrier inside one launch against a launch boundary, at this path's sizes (the persistent
cell-step question).  Two dependent streaming phases over a buffer (phase B reads what OTHER workgroups wrote in
phase A): two launches vs one launch with an agent-scope release / counter / poll / acquire barrier, 256 workgroups,
replayed from a hipGraph like the real step.  Prints us per pair of phases.
    python tools/barrier_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bm-nas_amd'))
import torch

from bmnas import lib

L = lib.load()
dev = torch.device('cuda:0')
st = lambda: torch.cuda.current_stream().cuda_stream
print('# MB per phase | two launches (us) | one launch + grid barrier (us) | barrier - boundary (us)')
for mb in (0.05, 0.4, 1.5, 4.7, 9.4, 28.0):
    n = int(mb * 1e6 / 4) // 1024 * 1024
    a, t, o = (torch.randn(n, device=dev) for _ in range(3))
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    res = {}
    for mode in (0, 1):
        reps = 40
        ctr.zero_()
        rnd = [0]

        def body():
            for _ in range(reps):
                rnd[0] += 1
                rc = L.bmnas_probe_barrier(a.data_ptr(), t.data_ptr(), o.data_ptr(), n, 256, mode, ctr.data_ptr(), rnd[0], st())
                assert rc == 0, rc
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body()                                   # warm-up (eager); the counter keeps counting
            torch.cuda.synchronize()
            # capture `reps` calls; replays must see the counter where the capture's round numbers expect it
            base = rnd[0]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                body()
            best = 1e9
            for _ in range(5):
                ctr.fill_(256 * base)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        res[mode] = best
        # correctness of the fused form: out = 2 * (0.5 * in + 1)[shifted] + 1
        if mode == 1:
            want = (2 * (0.5 * a + 1) + 1).roll(-(n // 2))
            assert torch.allclose(o, want, atol=1e-5), float((o - want).abs().max())
    print(f'{mb:6.2f} | {res[0]:8.2f} | {res[1]:8.2f} | {res[1] - res[0]:+6.2f}')
