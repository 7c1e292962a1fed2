import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/bm-nas_amd')
import torch, torch.distributed as dist
from bmnas import dist as bdist
rank, local, world = bdist.init_from_env('nccl')
dev = torch.device('cuda', local); torch.cuda.set_device(dev)
flat = torch.zeros(624000, device=dev)
small = torch.zeros(64, device=dev)
x = torch.randn(4096, 4096, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
r = {}
r['allreduce_2.5MB'] = t(lambda: dist.all_reduce(flat))
r['allreduce_small'] = t(lambda: dist.all_reduce(small))
r['matmul'] = t(lambda: x @ x)
r['matmul+allreduce'] = t(lambda: (x @ x, dist.all_reduce(flat)))
if rank == 0: print(r)
dist.barrier(); dist.destroy_process_group()
