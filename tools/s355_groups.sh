#!/bin/bash
# restart groups sharing the GPU: one group of 16, two / three / four free-running groups, the same paced (paced: a group reaches a sweep's
# forward-backward point only after its previous forward-backward finished);
# at 355 states (default) or with "165" as first argument at the headline grid
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
if [ "${1:-355}" = "165" ]; then F="--steps 20 --warmup 5"; else F="--steps 10 --warmup 3 --max-cn 12"; fi
F="$F --no-cpu-baseline --no-extra-states --no-fit-from-init"
run() { tag=$1; shift; for i in 1 2; do python3 $ROOT/bench.py $F "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%-22s %.1f it/s %.2f ms/step | k_fb %s joint %s' % ('$tag', d['value'], d['ms_per_step'], k.get('k_fb'), k.get('k_fb_joint')))"; done; }
run one_group --groups 1
run two_groups --groups 2 --host-option paced=0
run two_groups_paced --groups 2 --host-option paced=1
run three_groups_paced --groups 3 --host-option paced=1
run four_groups_paced --groups 4 --host-option paced=1
