#!/bin/bash
# A/B of the paired forward-backward launches (rmx_pair_batches) on the headline bench, each twice, with the kernel table
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-states --no-fit-from-init"
run() { tag=$1; shift; for i in 1 2; do python3 $ROOT/bench.py $F "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%-22s %.1f it/s %.2f ms/step | k_fb %s k_fb_joint %s marg %s' % ('$tag', d['value'], d['ms_per_step'], k.get('k_fb'), k.get('k_fb_joint'), k.get('k_marginals<true>')))"; done; }
run pair_fb=0 --host-option pair_fb=0
run joint
for q in "" 8; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; fi
  for us in 800 1500 2200; do RMX_PAIR_STAGGER_US=$us run "q${q:-4}_stagger$us"; done
done
export GPU_MAX_HW_QUEUES=8
run q8_pair_fb=0 --host-option pair_fb=0
run q8_joint
