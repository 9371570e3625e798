#!/bin/bash
# the driver's default bench invocation, timed, with a digest of the extra objects of its JSON line
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
t0=$(date +%s.%N)
python3 $ROOT/bench.py "$@" > $OUT/bench_full.json 2> $OUT/bench_full.err
t1=$(date +%s.%N)
echo "wall $(echo "$t1 - $t0" | bc) s"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_full.json").read().strip().splitlines()[-1])
print(d["metric"], round(d["value"],1), "it/s", round(d["ms_per_step"],2), "ms/step; roofline frac", round(d["roofline"]["frac"],3), "avg launch ms", round(d["roofline"]["avg_launch_ms"],3))
for k in ("roofline_one_group","states_355","strong_scaling_proxy","fit_from_init","cpu_baseline"):
    v=d.get(k)
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk not in ("kernels","note","sample")}
    print(k, json.dumps(v)[:1100])
PY
