#!/bin/bash
# A/B of two BUILDS of the library on the headline bench line, alternating REPS times on one box (boxes differ by up to 10 %):
#   bash tools/ab_library.sh REPS path/to/libremixt_hip_prev.so      (the other side is the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
REPS=$1; PREV=$2
F="--steps 20 --warmup 5 --no-cpu-baseline --no-fit-from-init --no-extra-states"
for rep in $(seq 1 $REPS); do
for side in prev new; do
  if [ $side = prev ]; then export RMX_LIB_PATH=$PREV; else unset RMX_LIB_PATH; fi
  python3 $ROOT/bench.py $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %.1f it/s %.2f ms | fb %.3f ms frac %.3f' % ('$side', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
done
