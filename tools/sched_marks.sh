#!/bin/bash
# stage times of the M-steps (tools/mstep_marks.py) with 2 / 3 / 4 restart groups and more hardware queues
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sched_marks
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for g in 2 3 4; do
  for q in 4 16; do
    echo "### groups $g hw queues $q"
    GPU_MAX_HW_QUEUES=$q NGROUPS=$g python3 $ROOT/tools/mstep_marks.py 2>&1 | grep -v amdgpu.ids
  done
done > $OUT/marks.txt
