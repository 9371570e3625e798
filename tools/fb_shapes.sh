#!/bin/bash
# forward-backward launch shapes at the bench workload: restarts per launch x restarts per workgroup (k_fbm<., NV>), with the kernel's
# cycle counters.  Usage: tools/fb_shapes.sh [MAXCN [variant]]   (variant: an alternative build tools/micro/lib_<variant>.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
MAXCN=${1:-8}
for rst in 4 8 16; do
  for nv in 1 2 4; do
    if [ $((rst / nv * 46)) -gt 400 ]; then continue; fi
    echo "== restarts per launch $rst, per workgroup $nv ${2:+(library $2)}"
    STAMPS=$2 RST=$rst NV=$nv MAXCN=$MAXCN FB_DEBUG=1 ITERS=3 python3 $ROOT/tools/fb_only.py 2>&1 | grep -E "k_fb |debug|raised" || exit 1
  done
done
echo "== chains in the proportions of the human chromosomes (5 : 1), 8 and 16 restarts per launch: shapes chosen per chain"
for rst in 8 16; do RST=$rst MAXCN=$MAXCN UNEQUAL=1 FB_DEBUG=1 ITERS=3 python3 $ROOT/tools/fb_only.py 2>&1 | grep -E "k_fb |debug|shapes|raised"; done
for nv in 2 4; do echo "   (pinned: $nv per workgroup, 8 per launch)"; RST=8 NV=$nv MAXCN=$MAXCN UNEQUAL=1 FB_DEBUG=1 ITERS=3 python3 $ROOT/tools/fb_only.py 2>&1 | grep -E "k_fb |debug|raised"; done
