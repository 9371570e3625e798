#!/bin/bash
# Everything under profiles/ for one round, on a GPU box: bash tools/collect_profiles.sh r03
# (rocprofv3 gets the program itself after `--`; counters are collected in their own passes, without trace domains.)
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
F="--steps 20 --warmup 5"
python3 $ROOT/bench.py $F > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench done: $(cut -c1-120 $OUT/bench_line.json)"
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o b --output-format csv -- python3 $ROOT/bench.py $F --no-cpu-baseline --no-extra-states --no-fit-from-init > $OUT/bench_line_under_rocprof.json 2> $OUT/prof_bench.err
python3 $ROOT/tools/timeline.py $(find $OUT/prof_bench -name "b_kernel_trace.csv" | head -1) "k_fbm<42>" > $OUT/timeline.txt 2>&1
cp $(find $OUT/prof_bench -name "b_kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
pmc() { # tag, restarts, max_cn, states
  export ITERS=3 RST=$2 MAXCN=$3
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$1 -o p --output-format csv -- python3 $ROOT/tools/fb_only.py > $OUT/pmc_fetch_$1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$1 -o p --output-format csv -- python3 $ROOT/tools/fb_only.py > $OUT/pmc_write_$1.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc_fetch_$1 $OUT/pmc_write_$1 50000 $4 $2 $OUT/traffic_$1.json > $OUT/pmc_summary_$1.log 2>&1
}
pmc 8 8 8 165
pmc 16 16 8 165
pmc s355 16 12 355
pmc s355_8 8 12 355
rocprofv3 --kernel-trace --stats -d $OUT/prof_s355 -o s --output-format csv -- python3 $ROOT/tools/fb_only.py > $OUT/prof_s355.log 2>&1
cp $(find $OUT/prof_s355 -name "s_kernel_stats.csv" | head -1) $OUT/s355_kernel_stats.csv
unset MAXCN
export ITERS=5
{ echo "# 16 restarts per launch (one restart group)"; RST=16 FB_DEBUG=1 python3 $ROOT/tools/fb_only.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# 8 restarts per launch (the bench: two restart groups of 8)"; RST=8 FB_DEBUG=1 python3 $ROOT/tools/fb_only.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# 8 restarts per launch, one breakpoint (plain steps only)"; RST=8 NBRK=1 FB_DEBUG=1 python3 $ROOT/tools/fb_only.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# 355 states (max_cn = 12), 16 restarts per launch"; ITERS=3 RST=16 MAXCN=12 FB_DEBUG=1 python3 $ROOT/tools/fb_only.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# 355 states, one breakpoint (plain steps only)"; ITERS=3 RST=16 MAXCN=12 NBRK=1 FB_DEBUG=1 python3 $ROOT/tools/fb_only.py 2>&1 | grep -v amdgpu.ids; } > $OUT/fb_launch_shapes.txt
python3 $ROOT/tools/mstep_marks.py 2>&1 | grep -v amdgpu.ids > $OUT/mstep_stage_times.txt
python3 $ROOT/tools/h_rounds.py 2>&1 | grep -v amdgpu.ids >> $OUT/mstep_stage_times.txt
python3 $ROOT/tools/prep_time.py 2>&1 | grep -v amdgpu.ids >> $OUT/mstep_stage_times.txt
python3 $ROOT/tools/e2e_time.py 2>&1 | grep -v amdgpu.ids > $OUT/e2e_wall.txt
python3 $ROOT/tools/large_grids.py 2>&1 | grep -v amdgpu.ids > $OUT/large_grids.txt
(cd $ROOT/tools/micro && ./mfma64_bench throughput) > $OUT/mfma64_throughput.txt 2>&1
rm -rf $OUT/prof_bench $OUT/prof_s355 $OUT/pmc_fetch_* $OUT/pmc_write_*
ls -la $OUT
