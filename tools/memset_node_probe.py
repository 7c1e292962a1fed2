"""Does a hipMemsetAsync captured into a hipGraph clear its buffer on EVERY replay?  (ROCm 7.2 on MI355X, round 3:
no — 1920 ... 11776 bytes are cleared on the first replay and hold 1e26 ... 1e32 / inf afterwards (1920 bytes passed
on one box and failed on another); 1 MB is fine.  torch's zero_() / torch.zeros inside a capture are fill KERNELS
and clear on every replay: second half of the output.)
Why csrc/linear.hip zero-fills with a kernel.  Run on the GPU box:  python tools/memset_node_probe.py"""
import ctypes

import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for n in (1920, 1992, 4096, 11776, 1 << 20):
    buf = torch.full((n // 4,), 7.0, device='cuda')
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, n, torch.cuda.current_stream().cuda_stream)
        buf.add_(1.0)
    seen = []
    for _ in range(4):
        graph.replay()
        torch.cuda.synchronize()
        seen.append((float(buf.min()), float(buf.max())))
    print(f'memset node of {n} bytes (rc {rc}), then +1: min/max per replay {seen}', flush=True)

# the same through torch: tensor.zero_() / torch.zeros inside a capture (what bmnas.functions._ZeroPool and the arena
# fallbacks issue) — a fill KERNEL on this build, so every replay clears
for n in (1920, 4096, 11776, 1 << 20):
    buf = torch.full((n // 4,), 7.0, device='cuda')
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        buf.zero_()
        z = torch.zeros(n // 4, device='cuda')
        buf.add_(1.0)
        z.add_(2.0)
    seen = []
    for _ in range(4):
        graph.replay()
        torch.cuda.synchronize()
        seen.append((float(buf.min()), float(buf.max()), float(z.min()), float(z.max())))
    print(f'torch zero_() / zeros of {n} bytes, then +1 / +2: per replay {seen}', flush=True)
