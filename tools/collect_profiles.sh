#!/bin/bash
# Everything under profiles/ for one round, on a GPU box: bash tools/collect_profiles.sh r04
# (rocprofv3 gets the program itself after `--`; counters are collected in their own passes, without trace domains.)
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
F="--steps 20 --warmup 5"
python3 $ROOT/bench.py $F > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench done: $(cut -c1-120 $OUT/bench_line.json)"
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o b --output-format csv -- python3 $ROOT/bench.py $F --no-cpu-baseline --no-extra-states --no-fit-from-init > $OUT/bench_line_under_rocprof.json 2> $OUT/prof_bench.err
python3 $ROOT/tools/timeline.py $(find $OUT/prof_bench -name "b_kernel_trace.csv" | head -1) "k_fbm<42" > $OUT/timeline.txt 2>&1
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
unset MAXCN ITERS RST
bash $ROOT/tools/fb_shapes.sh 8 2>&1 | grep -v amdgpu.ids > $OUT/fb_launch_shapes.txt
{ echo; echo "# 355 states (max_cn = 12)"; bash $ROOT/tools/fb_shapes.sh 12 2>&1 | grep -v amdgpu.ids; } >> $OUT/fb_launch_shapes.txt
# FETCH_SIZE / WRITE_SIZE on kernels of known bytes (8- and 16-byte loads per lane, the row pattern of the strip kernels)
rocprofv3 --pmc FETCH_SIZE -d $OUT/calib_f -o c --output-format csv -- $ROOT/tools/micro/pmc_calib > $OUT/calib_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/calib_w -o c --output-format csv -- $ROOT/tools/micro/pmc_calib > $OUT/calib_w.log 2>&1
python3 $ROOT/tools/pmc_calib_summary.py $OUT/calib_f $OUT/calib_w > $OUT/pmc_calibration.txt 2>&1
rm -rf $OUT/calib_f $OUT/calib_w
python3 $ROOT/tools/mstep_marks.py 2>&1 | grep -v amdgpu.ids > $OUT/mstep_stage_times.txt
python3 $ROOT/tools/h_rounds.py 2>&1 | grep -v amdgpu.ids >> $OUT/mstep_stage_times.txt
python3 $ROOT/tools/prep_time.py 2>&1 | grep -v amdgpu.ids >> $OUT/mstep_stage_times.txt
python3 $ROOT/tools/e2e_time.py 2>&1 | grep -v amdgpu.ids > $OUT/e2e_wall.txt
python3 $ROOT/tools/large_grids.py 2>&1 | grep -v amdgpu.ids > $OUT/large_grids.txt
(cd $ROOT/tools/micro && ./mfma64_bench throughput) > $OUT/mfma64_throughput.txt 2>&1
# the memory system's answer to the fused marginal pass's access pattern (tools/micro/stream_rows.hip), and the pass itself beside it
{ $ROOT/tools/micro/stream_rows; echo; echo "# the passes themselves on the same box (8 restarts, tools/fb_only.py: k_marginals<true> = four fused passes and the call's last, unfused one)"; RST=8 ITERS=5 python3 $ROOT/tools/fb_only.py 2>&1 | grep "k_marginals\|k_framelog"; } > $OUT/stream_rows.txt 2>&1
# SQ counters of the sweep kernels (their own run: no trace domains)
export ITERS=3 RST=8
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d $OUT/pmc_sq -o p --output-format csv -- python3 $ROOT/tools/fb_only.py > $OUT/pmc_sq.log 2>&1
python3 $ROOT/tools/sq_counters.py $OUT/pmc_sq > $OUT/sweep_sq_counters.txt 2>&1
rm -rf $OUT/pmc_sq
unset ITERS RST
# the two groups' M-step stages with the search rounds driven by the device (search_mode 5) beside the default above
{ echo "# OPTS=search_mode=5"; OPTS=search_mode=5 python3 $ROOT/tools/mstep_marks.py 2>&1 | grep -v amdgpu.ids; echo "# RST=8 NGROUPS=1 (RestartGroups picks search_mode 5 for a single group)"; RST=8 NGROUPS=1 python3 $ROOT/tools/mstep_marks.py 2>&1 | grep -v amdgpu.ids; } > $OUT/mstep_stage_times_search5.txt 2>&1
REPS=6 python3 $ROOT/tools/s355_repeat.py 2>&1 | grep "^run" > $OUT/s355_repeat.txt
rm -rf $OUT/prof_bench $OUT/prof_s355 $OUT/pmc_fetch_* $OUT/pmc_write_*
# round 5: Gantt of one EM period, decode timing, repeated measurements in one process (stream pool), the 8-restart share's M-step stages
rocprofv3 --kernel-trace -d $OUT/prof_g -o g --output-format csv -- python3 $ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra-states --no-fit-from-init > /dev/null 2>&1
python3 $ROOT/tools/gantt.py $(find $OUT/prof_g -name "g_kernel_trace.csv" | head -1) 80 120 > $OUT/gantt.txt 2>&1
rm -rf $OUT/prof_g
{ python3 $ROOT/tools/decode_time.py 8; python3 $ROOT/tools/decode_time.py 8 2; python3 $ROOT/tools/decode_time.py 12; } 2>&1 | grep "rep \|k_viterbi\|k_backtrace\|results()\|collect_fit" > $OUT/decode_time.txt
{ RST=8 python3 $ROOT/tools/mstep_marks.py; } 2>&1 | grep -v amdgpu.ids > $OUT/mstep_stage_times_share.txt
ls -la $OUT
