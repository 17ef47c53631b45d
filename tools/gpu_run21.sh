export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
run() { (timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1) | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print('$*', '| fwd+bwd %.0f steps/s'%d['value'], '| fwd us/step %.2f'%d['fwd_us_per_time_step'], '| bwd us/step %.2f'%d['bwd_us_per_time_step'])
except Exception as e: print('$*', 'ERR', l[-400:])
"; }
run --workload gs2d_512
run --workload gs2d_512 --opt overlap=0
run --workload gs2d_512 --opt overlap_chunk=64
run --workload gs2d_512 --opt overlap_chunk=256
run --workload gs3d_128
run --workload gs3d_128 --opt overlap=0
run --workload lo2d_512
run --workload lo2d_512 --opt overlap=0
run --workload gs2d_512 --reaction factored
run --workload gs2d_512 --reaction factored --opt overlap=0
