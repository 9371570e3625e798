#!/bin/bash
# A/B of a library option on the headline bench and its 355-state line: bash tools/opt_ab.sh "two_streams=0" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-fit-from-init"
for o in "" "$@"; do
  opts=""; for kv in $o; do opts="$opts --option $kv"; done
  python3 $ROOT/bench.py $F $opts 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d.get('states_355',{})
print('%-28s headline %.1f it/s %.2f ms | 355: %.1f it/s %.1f ms' % ('${o:-default}', d['value'], d['ms_per_step'], s.get('value',0), s.get('ms_per_step',0)))"
done
