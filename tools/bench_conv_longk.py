"""Forward 1x1 conv of the reshape layers with a long contraction (C_in 2048) at small grids: time per launch
(HIP events over 200 launches) and parity against torch.  BMNAS_KSPLIT_MULTI=0 selects the old kernels."""
import sys
import torch
sys.path.insert(0, 'bm-nas_amd')
from bmnas import lib

for (b, C_in, M, L) in [(64, 2048, 128, 8), (8, 2048, 128, 8), (6, 2048, 128, 8), (48, 2048, 128, 8), (32, 1024, 192, 16)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, C_in, L, generator=g).cuda()
    W = (torch.randn(M, C_in, generator=g) / C_in ** 0.5).cuda()
    bias = torch.randn(M, generator=g).cuda()
    U = torch.empty(b, M, L, device='cuda')
    n_part = lib.conv1x1_num_partials(b, L)
    part = torch.empty(n_part * M * 2, device='cuda')
    lib.conv_family_calls(reset=True)
    lib.conv1x1_fwd([x], C_in, W, C_in, bias, U, part, b, L, M, 0)
    fam = {k: v for k, v in lib.conv_family_calls().items() if v}
    want = torch.einsum('mk,bkl->bml', W.double(), x.double()) + bias.double()[None, :, None]
    err = (U.double() - want).abs().max().item() / want.abs().max().item()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        lib.conv1x1_fwd([x], C_in, W, C_in, bias, U, part, b, L, M, 0)
    s.record()
    for _ in range(200):
        lib.conv1x1_fwd([x], C_in, W, C_in, bias, U, part, b, L, M, 0)
    e.record()
    torch.cuda.synchronize()
    print(f'b {b:4d} C_in {C_in} M {M} L {L}: {s.elapsed_time(e) / 200 * 1e3:7.2f} us/launch  family {fam}  rel err {err:.2e}')
