export TMPDIR=/tmp
run() { (timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1) | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print('$*', '| fwd+bwd %.0f steps/s'%d['value'], '| fwd us/step %.2f'%d['fwd_us_per_time_step'], '| bwd us/step %.2f'%d['bwd_us_per_time_step'])
except Exception as e: print('$*', 'ERR', l[-400:])
"; }
run --workload gs3d_128
run --workload gs3d_128 --opt fuse_wgrad=1
run --workload gs3d_128 --opt fuse_wgrad=1 --opt stream3d=0
run --workload gs3d_128 --reaction factored --opt fuse_wgrad=1
