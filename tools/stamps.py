#!/usr/bin/env python
"""Phase cycle sums of the K loop (wave 0 of workgroup 0) from a -DNR_STAMP build of the fused local_level kernel:
cycles waiting for the slice's own DMA, at the barrier, and computing (fragment reads + MFMAs + DMA issue)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import hip, ops  # noqa: E402

DEV = "cuda"


def main():
    A, Nt, Bv, Nv = 128, 24, 512, 12
    prec = hip.PREC_BF16X3 if "x3" in sys.argv[1:] else hip.PREC_BF16
    g = torch.Generator().manual_seed(0)
    t = torch.randn(A, Nt, 512, generator=g).to(DEV)
    v = torch.randn(Bv, Nv, 512, generator=g).to(DEV)
    pt = ops.prepare_tokens(t, torch.ones(A, Nt, device=DEV))
    pv = ops.prepare_tokens(v, torch.ones(Bv, Nv, device=DEV))
    wt = torch.full((A, Nt), 1.0 / Nt, device=DEV)
    wv = torch.full((Bv, Nv), 1.0 / Nv, device=DEV)
    lib = hip.lib()
    buf = (ctypes.c_ulonglong * 256)()
    for _ in range(5):
        ops.local_level(pt, pv, wt, wv, A, Nt, Bv, Nv, prec, hip.OUT_ROWSUM)
    lib.nr_debug_stamps(buf, 1)
    wait, bar, comp, n, total = buf[0], buf[1], buf[2], max(buf[3], 1), buf[4]
    print(f"tiles {hip.local_level_tiles(A, Nt, Bv, Nv, prec)}  slices {n}")
    print(f"per slice: wait for own DMA {wait / n:7.0f}   barrier {bar / n:7.0f}   compute {comp / n:7.0f}   sum {(wait + bar + comp) / n:7.0f} cycles")
    print(f"whole kernel (this wave): {total} cycles; K loop {wait + bar + comp} ({100.0 * (wait + bar + comp) / max(total, 1):.0f} %)")
    print(f"wave 0: setup before the K loop {buf[5]} cycles, K loop {buf[6] - buf[5]}, epilogue {buf[4] - buf[6]}")
    if buf[8] or buf[10]:
        print(f"ping-pong loop: group 0 wave: work {buf[8]} + barrier wait {buf[9]} = {buf[8] + buf[9]} cycles;  "
              f"group 1 wave: work {buf[10]} + barrier wait {buf[11]} = {buf[10] + buf[11]}")
        print(f"  of the work: MFMA phases {buf[12]} (group 0) / {buf[13]} (group 1) cycles, i.e. per phase "
              f"{buf[12] / 16:.0f} / {buf[13] / 16:.0f}; memory phases per phase {(buf[8] - buf[12]) / 16:.0f} / {(buf[10] - buf[13]) / 16:.0f}")


if __name__ == "__main__":
    main()
