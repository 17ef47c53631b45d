export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
run() { (timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1) | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print('$*', '| fwd+bwd %.0f steps/s'%d['value'], '| fwd us/step %.2f'%d['fwd_us_per_time_step'], '| bwd us/step %.2f'%d['bwd_us_per_time_step'], '|', ' '.join('%s %.2fus %.0fGB/s'%(k['kernel'][:14],k['avg_launch_us'],k['achieved']) for k in d['roofline']['all_kernels']))
except Exception as e: print('$*', 'ERR', l[-400:])
"; }
run --workload gs2d_512
run --workload gs2d_512 --reaction factored
run --workload lo2d_512
