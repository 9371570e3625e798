#!/bin/bash
# Everything under profiles/ for one round, on a GPU box: bash tools/collect_profiles.sh r02
# (rocprofv3 gets the program itself after `--`; counters are collected in their own passes, without trace domains.)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
F="--steps 20 --warmup 5"
python3 $ROOT/bench.py $F > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench done: $(cut -c1-120 $OUT/bench_line.json)"
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o b --output-format csv -- python3 $ROOT/bench.py $F --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2> $OUT/prof_bench.err
python3 $ROOT/tools/timeline.py $OUT/prof_bench/b_kernel_trace.csv "k_fbm<42>" > $OUT/timeline.txt 2>&1
export ITERS=3
export RST=8
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- python3 $ROOT/tools/fb_only.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- python3 $ROOT/tools/fb_only.py > $OUT/pmc_write.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write 50000 165 8 $OUT/traffic.json > $OUT/pmc_summary.log 2>&1
export ITERS=5
{ echo "# 16 restarts per launch (one restart group)"; RST=16 FB_DEBUG=1 python3 $ROOT/tools/fb_only.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# 8 restarts per launch (the bench: two restart groups of 8)"; RST=8 FB_DEBUG=1 python3 $ROOT/tools/fb_only.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# 355 states (max_cn = 12), 16 restarts per launch"; ITERS=3 RST=16 MAXCN=12 python3 $ROOT/tools/fb_only.py 2>&1 | grep -v amdgpu.ids; } > $OUT/fb_launch_shapes.txt
python3 $ROOT/tools/mstep_marks.py 2>&1 | grep -v amdgpu.ids > $OUT/mstep_stage_times.txt
python3 $ROOT/tools/h_rounds.py 2>&1 | grep -v amdgpu.ids >> $OUT/mstep_stage_times.txt
python3 $ROOT/tools/e2e_time.py 2>&1 | grep -v amdgpu.ids > $OUT/e2e_wall.txt
ls -la $OUT
