#!/bin/bash
# timing of experimental builds of the library (tools/micro/lib_<name>.so) on the forward-backward launch: fb_only.py per variant
for v in "$@"; do
  echo "== $v"
  if [ "$v" = base ]; then env NBRK=${NBRK:-1} ITERS=3 RST=${RST:-16} MAXCN=${MAXCN:-12} FB_DEBUG=1 python3 tools/fb_only.py 2>&1 | grep -E "k_fb |debug|raised"
  else env STAMPS=$v NBRK=${NBRK:-1} ITERS=3 RST=${RST:-16} MAXCN=${MAXCN:-12} FB_DEBUG=1 python3 tools/fb_only.py 2>&1 | grep -E "k_fb |debug|raised"; fi
done
