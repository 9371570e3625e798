#!/bin/bash
# per-wave cycle stamps of a plain forward-backward step (libraries built with -DRMX_FB_PSTAMPS as tools/micro/lib_pstamps_<variant>.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
for v in "$@"; do
  for nv in 1 2 4; do
    echo "== variant $v, restarts per workgroup $nv (8 per launch)"
    STAMPS=pstamps_$v PSTAMPS=1 RST=8 NV=$nv FB_DEBUG=1 ITERS=2 NBRK=1 python3 $ROOT/tools/fb_only.py 2>&1 | grep -E "k_fb |debug|plain-step|raised" || exit 1
  done
done
