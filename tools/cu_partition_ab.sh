#!/bin/bash
# restart groups confined to disjoint CU ranges (library option cu_partition through RestartGroups(cu_partition=True)) against groups sharing the chip:
#   bash tools/cu_partition_ab.sh [165|355]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
if [ "${1:-355}" = "165" ]; then F="--steps 20 --warmup 5"; else F="--steps 10 --warmup 3 --max-cn 12"; fi
F="$F --no-cpu-baseline --no-extra-states --no-fit-from-init"
run() { tag=$1; shift; for i in 1 2; do python3 $ROOT/bench.py $F "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
f=k.get('k_fb') or {}
print('%-34s %.1f it/s %.2f ms/step | k_fb %.2f ms x %d' % ('$tag', d['value'], d['ms_per_step'], f.get('ms',0)/max(f.get('n',1),1), f.get('n',0)))"; done; }
if [ "${1:-355}" = "165" ]; then
run two_groups --groups 2
run two_groups_cu --groups 2 --host-option cu_partition=1
run two_groups_cu_nv4 --groups 2 --host-option cu_partition=1 --option fb_nv=4
run four_groups --groups 4
run four_groups_cu --groups 4 --host-option cu_partition=1
run four_groups_cu_paced --groups 4 --host-option cu_partition=1 --host-option paced=1
else
run two_groups_paced --groups 2 --host-option paced=1
run two_groups_cu --groups 2 --host-option paced=0 --host-option cu_partition=1
run two_groups_cu_paced --groups 2 --host-option paced=1 --host-option cu_partition=1
run four_groups_cu --groups 4 --host-option paced=0 --host-option cu_partition=1
run four_groups_cu_paced --groups 4 --host-option paced=1 --host-option cu_partition=1
fi
