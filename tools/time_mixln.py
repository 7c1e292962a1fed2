"""Back-to-back timing of bmnas_node_mix_ln_bwd against the two launches it replaces (MM-IMDB shapes)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'bm-nas_amd')):
    sys.path.insert(0, p)
import torch
from bmnas import lib

b, C, L = int(os.environ.get('B', 128)), 192, 16
d = 'cuda'
M = 3 * C
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(d)
x, p1, U, pre, gy = r(b, C, L), r(b, C, L), r(b, M, L), r(b, C, L), r(b, C, L)
ln_w, ln_b = r(C, L), r(C, L)
gamma = torch.softmax(r(4), 0)
chan = torch.cat([r(M) * 0.1, r(M).abs() + 0.5, r(M).abs() + 0.5, r(M) * 0.1])
stats = torch.stack([pre.mean(dim=(1, 2)), 1.0 / pre.std(dim=(1, 2))], 1).contiguous()
dglu, dfc = lib.make_dropout(0.1, 1, 0), lib.make_dropout(0.1, 1, b * C * L // 4)
gin, dres, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
dV, bn_grad, dgam = torch.empty(b, M, L, device=d), torch.zeros(2 * M, device=d), torch.zeros(16 * 64, device=d)
# spoil the caches between launches like a real step does (other kernels' data in between)
junk = torch.empty(64 << 20, device=d)


def fused():
    lib.node_mix_ln_bwd(gy, pre, ln_w, stats, gin, dres, 0, x, x, p1, U, chan, gamma, dgam, dx, None, 0, dV, bn_grad,
                        b, C, L, dglu, dfc, 16, 64)


def split():
    lib.cat_ln_bwd(gy, [pre], None, ln_w, ln_b, stats, [gin], dres, 0, None, None, b, C, L, False)
    lib.node_mix_bwd(gin, x, x, p1, U, chan, gamma, dgam, dx, None, 0, dV, bn_grad, b, C, L, dglu, dfc, 16, 64)


def timeit(fn, n=200, spoil=False):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    if not spoil:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / n
    for _ in range(n // 4):
        junk.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e) * 1e3
    return tot / (n // 4)


print('probe', os.environ.get('BMNAS_MIXLN_PROBE', '0'), 'b', b,
      'fused %.2f us  split %.2f us | cold: fused %.2f  split %.2f' % (timeit(fused), timeit(split),
                                                                      timeit(fused, spoil=True), timeit(split, spoil=True)))
