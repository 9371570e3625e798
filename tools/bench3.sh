#!/bin/bash
# the headline bench three times (20 steps each), value and ms per step
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-states --no-fit-from-init"
for i in 1 2 3; do python3 $ROOT/bench.py $F "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f it/s %.2f ms/step frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; done
