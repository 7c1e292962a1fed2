"""Where the time of a channel-owner launch goes: a hipGraph of 20 bmnas_co_inner_fwd launches (NTU b8 shape by default),
50 replays, microseconds per launch — one BMNAS_CO_PROBE variant per child process (the probe value is read once).
    python tools/co_probe.py [b C L] [probe ...]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd')):
    sys.path.insert(0, p)


def child(b, C, L, which):
    import torch
    from tests import test_chanown_gpu as T
    from bmnas import lib
    from bmnas import cell as K
    z, P = T._case(b, C, L, 7, 2, True, True)
    out = T._alloc(z, P, b, C, L, True)
    part = torch.zeros(K.STAT_SHARDS * 3 * C * 2, device=z.device)
    # (timing only: nothing re-zeroes the accumulate-into buffers between the launches of the graph)
    fn = {'fwd': lambda: T._co_call(z, P, out, b, C, L, True),
          'two': lambda: T._two_launch_call(z, P, out, part, b, C, L, True)}[which]
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                fn()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print(f'{which} b{b} C{C} L{L} probe {os.environ.get("BMNAS_CO_PROBE", "0"):>3s}: '
          f'{(time.perf_counter() - t0) / 50 / 20 * 1e6:.2f} us per call', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    args = sys.argv[1:]
    shape = args[:3] if len(args) >= 3 else ['8', '128', '8']
    probes = args[3:] or ['0', '1', '2', '3', '4', '8', '12', '16', '32', '63']
    which = os.environ.get('CO_WHICH', 'fwd')
    for pr in probes:
        env = dict(os.environ, BMNAS_CO_PROBE=pr)
        subprocess.run([sys.executable, __file__, '--child', *shape, which], env=env, check=False)
    if which == 'fwd':
        subprocess.run([sys.executable, __file__, '--child', *shape, 'two'], env=dict(os.environ, BMNAS_CO_PROBE='0'))
