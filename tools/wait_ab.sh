#!/bin/bash
# host wait policy A/B: the default (blocked wait on the completion signal) vs active waiting
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-states --no-fit-from-init"
for v in "X=1" "ROC_ACTIVE_WAIT_TIMEOUT=100" "ROC_ACTIVE_WAIT_TIMEOUT=1000"; do
  echo "== $v"
  env $v python3 $ROOT/tools/h_rounds.py 2>&1 | grep rounds
  env $v python3 $ROOT/bench.py $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s %.2f ms/step' % (d['value'], d['ms_per_step']))"
  env $v python3 $ROOT/bench.py $F --groups 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one group: %.1f it/s %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
