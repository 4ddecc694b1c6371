#!/usr/bin/env python
"""Per-kernel register / scratch usage of csrc/*.hip as the compiler reports it (-Rpass-analysis=kernel-resource-usage):
lists every kernel with VGPR spills or a private segment (scratch).  CPU only (hipcc cross-compiles gfx950).

    python tools/resource_usage.py [file.hip ...] [--all]
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neighborretr_amd", "csrc")


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def usage(src):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-c", src,
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("NR_EXTRA_FLAGS", "").split()
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    out = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        out.append(dict(name=demangle(b.split()[0]), vgprs=g(r" VGPRs"), agprs=g(r"AGPRs"), spill=g(r"VGPRs Spill"),
                        scratch=g(r"ScratchSize \[bytes/lane\]"), occupancy=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]")))
    return out


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    show_all = "--all" in sys.argv
    n = bad = 0
    for f in files:
        for k in usage(f):
            n += 1
            if k["spill"] > 0 or k["scratch"] > 0:
                bad += 1
            if show_all or k["spill"] > 0 or k["scratch"] > 0:
                print(f"{os.path.basename(f):22s} VGPRs {k['vgprs']:3d} AGPRs {k['agprs']:3d} spill {k['spill']:4d} scratch {k['scratch']:5d} B/lane  "
                      f"occ {k['occupancy']}  LDS {k['lds']:6d}  {k['name'][:140]}")
    print(f"{n} kernels, {bad} with spills / scratch")


if __name__ == "__main__":
    main()
