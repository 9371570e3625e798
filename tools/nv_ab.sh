#!/bin/bash
# A/B of the forward-backward workgroup shape on the headline bench (one box): automatic choice vs pinned 4 / 2 restarts per workgroup
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-states --no-fit-from-init $EXTRA"
run() { tag=$1; shift; for i in 1 2; do python3 $ROOT/bench.py $F "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%-16s %.1f it/s %.2f ms/step | k_fb avg %.3f ms n=%d | marg avg %.3f | pairwise avg %.3f' % ('$tag', d['value'], d['ms_per_step'], k['k_fb']['ms']/k['k_fb']['n'], k['k_fb']['n'], k['k_marginals<true>']['ms']/k['k_marginals<true>']['n'], k['k_pairwise']['ms']/k['k_pairwise']['n']))"; done; }
run auto
run fb_nv=4 --option fb_nv=4
run fb_nv=2 --option fb_nv=2
