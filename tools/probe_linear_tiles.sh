export NR_HIP_LIB=$PWD/neighborretr_amd/libnr_tune.so
echo "default tiles"; python tools/linear_bound_probe.py 2>&1 | grep "shape"
for t in "2,2,2,4" "4,2,2,4" "6,2,2,4"; do echo "NR_LINEAR_TILE=$t (8 waves, two-deep ring, ping-pong loop)"; NR_LINEAR_TILE=$t python tools/linear_bound_probe.py 2>&1 | grep "shape" | sed 's/one pass.*//'; done
