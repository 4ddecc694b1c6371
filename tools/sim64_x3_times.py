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
run(128, 128); run(1000, 1000)
