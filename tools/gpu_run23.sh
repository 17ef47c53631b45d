export TMPDIR=/tmp
for wl in gs2d_512 gs3d_128 lo2d_512; do
(timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl 2>&1 | tail -1) | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', 'value %.0f'%d['value'], json.dumps(d.get('physics_residual')))
"
done
