mkdir -p gpurun_out
export TMPDIR=/tmp
run() { (timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --T 400 "$@" 2>&1 | tail -1) | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print('$*', '| fwd+bwd %.0f steps/s'%d['value'], '| fwd us/step %.2f'%d['fwd_us_per_time_step'], '| bwd us/step %.2f'%d['bwd_us_per_time_step'])
except Exception as e: print('$*', 'ERR', l[-300:])
"; }
run --workload gs2d_512
run --workload gs2d_512 --opt tile_nt=512
run --workload gs2d_512 --opt tile_k=2
run --workload gs2d_512 --opt tile=0
run --workload gs2d_512 --opt tile=0 --opt vec=1
run --workload gs2d_512 --opt tile=0 --opt block=64
run --workload gs2d_512 --opt tile=0 --opt block=128
run --workload lo2d_512 --opt tile_nt=512
run --workload gs3d_128 --opt vec=1
run --workload gs3d_128 --opt block=128
run --workload gs3d_128 --opt block=64
