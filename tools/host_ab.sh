#!/bin/bash
# A/B of host-side M-step options on the headline bench (each twice)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-states --no-fit-from-init"
run() { tag=$1; shift; for i in 1 2; do python3 $ROOT/bench.py $F "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.1f it/s %.2f ms/step' % ('$tag', d['value'], d['ms_per_step']))"; done; }
run default
run sample_prep=0 --host-option sample_prep=0
run mstep_threads=1 --host-option mstep_threads=1
run mstep_threads=2 --host-option mstep_threads=2
run prep0_threads1 --host-option sample_prep=0 --host-option mstep_threads=1
run switch50 --switch-interval-us 50
run switch50_threads2 --switch-interval-us 50 --host-option mstep_threads=2
