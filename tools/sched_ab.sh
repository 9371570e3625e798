#!/bin/bash
# A/B of the restart-group scheduling knobs on one box: hardware queues, groups, GIL switch interval.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sched_ab
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-states --no-fit-from-init"
run() { # tag, env assignments..., -- bench flags
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 $ROOT/bench.py $F "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python3 - "$OUT/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d['kernels']
    print('%-28s %7.1f it/s %6.2f ms/step  fb avg %.3f ms  marg %.3f' % (sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'],
          k.get('k_marginals<true>',{'ms':0})['ms']/max(1,k.get('k_marginals<true>',{'n':1})['n'])), flush=True)
except Exception as e:
    print(sys.argv[2], 'FAILED', e, flush=True)
PY
}
run g2_default X=1 --
run g2_q8 GPU_MAX_HW_QUEUES=8 --
run g2_q16 GPU_MAX_HW_QUEUES=16 --
run g2_q8_si GPU_MAX_HW_QUEUES=8 -- --switch-interval-us 50
run g3_q16 GPU_MAX_HW_QUEUES=16 -- --groups 3
run g3_q16_si GPU_MAX_HW_QUEUES=16 -- --groups 3 --switch-interval-us 50
run g4_q16_si GPU_MAX_HW_QUEUES=16 -- --groups 4 --switch-interval-us 50
run g1 X=1 -- --groups 1
# timeline of the 2-group run with 8 queues
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace -d $OUT/prof_q8 -o b --output-format csv -- python3 $ROOT/bench.py $F > $OUT/q8_rocprof.json 2> $OUT/q8_rocprof.err
python3 $ROOT/tools/timeline.py $OUT/prof_q8/b_kernel_trace.csv "k_fbm<42>" > $OUT/timeline_q8.txt 2>&1
cat $OUT/timeline_q8.txt
rm -rf $OUT/prof_q8
