import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from neighborretr_amd import ops, hip
import numpy as np
def run(A, Bv, N=64, prec=hip.PREC_BF16X3):
    g = torch.Generator().manual_seed(1)
    t = torch.randn(A, N, 512, generator=g).cuda(); v = torch.randn(Bv, N, 512, generator=g).cuda()
    pt, pv = ops.prepare_tokens(t), ops.prepare_tokens(v)
    wt = torch.full((A, N), 1.0 / N).cuda(); wv = torch.full((Bv, N), 1.0 / N).cuda()
    for _ in range(5): ops.local_level(pt, pv, wt, wv, A, N, Bv, N, prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.local_level(pt, pv, wt, wv, A, N, Bv, N, prec)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    print(f"{os.path.basename(hip.LIB_PATH)}  x3 {A}x{Bv} 64-token: {us:8.1f} us   {2*3*A*Bv*N*N*512/us/1e6:7.1f} TFLOP/s issued")
if len(sys.argv) > 1 and sys.argv[1] == "24":
    for A_, B_ in ((1024, 1024), (1000, 1000), (1024, 512), (256, 256)):
        Nt, Nv = 24, 12
        g = torch.Generator().manual_seed(1)
        t = torch.randn(A_, Nt, 512, generator=g).cuda(); v = torch.randn(B_, Nv, 512, generator=g).cuda()
        pt, pv = ops.prepare_tokens(t), ops.prepare_tokens(v)
        wt = torch.full((A_, Nt), 1.0 / Nt).cuda(); wv = torch.full((B_, Nv), 1.0 / Nv).cuda()
        for _ in range(5): ops.local_level(pt, pv, wt, wv, A_, Nt, B_, Nv, hip.PREC_BF16X3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ops.local_level(pt, pv, wt, wv, A_, Nt, B_, Nv, hip.PREC_BF16X3)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 * 1e6
        print(f"{os.path.basename(hip.LIB_PATH)}  x3 {A_}x{B_} 24x12-token: {us:8.1f} us   {2*3*A_*B_*Nt*Nv*512/us/1e6:7.1f} TFLOP/s issued")
else:
    run(128, 128); run(1000, 1000)
