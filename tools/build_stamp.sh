#!/bin/bash
# diagnostic build of the library (-DNR_STAMP: in-kernel phase clocks) -> neighborretr_amd/libnr_stamp.so; use with NR_HIP_LIB
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/nr_stamp_obj
ls neighborretr_amd/csrc/*.hip | xargs -P 8 -I{} sh -c '/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNR_STAMP -DNR_TUNE -Wno-unused-function -I include -c {} -o /tmp/nr_stamp_obj/$(basename {} .hip).o'
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o neighborretr_amd/libnr_stamp.so /tmp/nr_stamp_obj/*.o
ls -la neighborretr_amd/libnr_stamp.so
