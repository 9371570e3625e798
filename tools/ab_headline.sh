#!/bin/bash
# A/B of library options on the headline bench line only, alternating the variants REPS times on one box:
#   bash tools/ab_headline.sh [REPS] "" "trial_kernel=1" "search_mode=6 trial_kernel=1" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
REPS=$1; shift
F="--steps 20 --warmup 5 --no-cpu-baseline --no-fit-from-init --no-extra-states --profile-all"
for rep in $(seq 1 $REPS); do
for o in "$@"; do
  opts=""; for kv in $o; do case $kv in host:*) opts="$opts --host-option ${kv#host:}";; groups=*) opts="$opts --groups ${kv#groups=}";; *) opts="$opts --option $kv";; esac; done
  python3 $ROOT/bench.py $F $opts 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
def g(n):
    v=k.get(n); return '%s %.2f' % (n, v['ms']/max(v['n'],1)) if v else ''
print('%-40s %.1f it/s %.2f ms | fb %.3f ms frac %.3f | %s' % ('${o:-default}', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], ' '.join(g(n) for n in ('k_marginals<false>','k_ell_list','k_pairwise','k_marginals<true>','k_framelogprob'))))"
done
done
