#!/bin/bash
# the two-graph form of an overlapped owned step (tools/round_profile.py two): others fused or not, 1 / 2 slots, host issue order
cd "$(dirname "$0")/.."
for v in "single 1 AB" "single 2 AB" "single 2 AO" "fused 1 AB" "fused 2 AB" "fused 2 AO" "fused 2 AB" "single 2 AB"; do
  echo "== $v"; python tools/round_profile.py 8 two $v 2>&1 | grep -E "round us|Error" | tail -2
done
