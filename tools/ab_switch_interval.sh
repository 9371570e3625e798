cd /tmp
ROOT=$GRAFT_REPO_ROOT
F="--steps 20 --warmup 5 --no-cpu-baseline --no-fit-from-init --no-extra-states"
for rep in 1 2; do for sw in 0 50 200 1000; do
python3 $ROOT/bench.py $F --switch-interval-us $sw 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('switch interval $sw us: %.1f it/s %.2f ms' % (d['value'], d['ms_per_step']))"
done; done
