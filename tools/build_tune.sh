#!/bin/bash
# tuning build of the library (-DNR_TUNE: the NR_* environment hooks are live) -> neighborretr_amd/libnr_tune.so
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/nr_tune_obj
for s in neighborretr_amd/csrc/*.hip; do
  o=/tmp/nr_tune_obj/$(basename "${s%.hip}").o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ -n "$(find neighborretr_amd/csrc include -name '*.h' -newer "$o")" ]; then
    echo "$s"
  fi
done | xargs -P 8 -I{} sh -c '/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNR_TUNE -Wno-unused-function -I include -c {} -o /tmp/nr_tune_obj/$(basename {} .hip).o'
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o neighborretr_amd/libnr_tune.so /tmp/nr_tune_obj/*.o
ls -la neighborretr_amd/libnr_tune.so
