#!/bin/bash
# the two-graph form of an overlapped owned step (tools/round_profile.py W two ...): others fused or not, 1 / 2 slots, host issue order,
# one loss stream or one per slot (own / chain)
cd "$(dirname "$0")/.."
W="${1:-8}"
for v in "single 1 AB one" "single 2 AB one" "single 2 AB own" "single 2 AB chain" "single 3 AB own" "single 3 AB chain"; do
  echo "== W=$W $v"; python tools/round_profile.py $W two $v 2>&1 | grep -E "round us|Error" | tail -2
done
