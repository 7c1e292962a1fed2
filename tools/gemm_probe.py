"""Micro-benchmark of the conv1x1 GEMM entry points in isolation (HIP events, tight loop)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bm-nas_amd'))
import torch
from bmnas import lib
b, C, L = int(os.environ.get('B', 128)), 192, 16
dev = 'cuda'
z = torch.randn(b, C, L, device=dev)
Wf = torch.randn(3 * C, C, device=dev)
bias = torch.randn(3 * C, device=dev)
U = torch.empty(b, 3 * C, L, device=dev)
P = lib.conv1x1_num_partials(b, L)
part = torch.empty(P * 3 * C * 2, device=dev)
dU = torch.randn(b, 3 * C, L, device=dev)
dz = torch.empty(b, C, L, device=dev)
dW = torch.zeros(3 * C, 2 * C, device=dev)
db = torch.zeros(3 * C, device=dev)

_blk = torch.randn(8192, 8192, device=dev)

def timeit(fn, n=50):
    # queue the calls behind ~30 ms of GEMMs so the GPU runs them back to back (the Python/ctypes
    # launch path costs ~8 us per call and would otherwise be what is measured)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    b = _blk
    for _ in range(4): b = (b @ b) * 1e-4
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

print('probe', os.environ.get('BMNAS_CONV_PROBE', '0'), 'B', b)
print('fwd      us', round(timeit(lambda: lib.conv1x1_fwd([z], C, Wf, C, bias, U, part, b, L, 3 * C)), 2))
stat = torch.zeros(4 * 3 * C * 2, device=dev)
print('fwd stat us', round(timeit(lambda: lib.conv1x1_fwd([z], C, Wf, C, bias, U, stat, b, L, 3 * C, stat_shards=4)), 2))
print('fwd nost us', round(timeit(lambda: lib.conv1x1_fwd([z], C, Wf, C, bias, U, None, b, L, 3 * C)), 2))
print('bwd_data us', round(timeit(lambda: lib.conv1x1_bwd_data(dU, Wf, C, [dz], C, 0, b, L, 3 * C)), 2))
print('bwd_wght us', round(timeit(lambda: lib.conv1x1_bwd_weight(dU, [z], C, dW, 2 * C, db, C, b, L, 3 * C)), 2))
x = torch.randn(b, C, L, device=dev); o = torch.empty_like(x)
print('copy 1.5MB us', round(timeit(lambda: o.copy_(x)), 2))
