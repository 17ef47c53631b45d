export TMPDIR=/tmp
timeout 600 python tools/tile_timing.py 2>&1 | grep -v amdgpu.ids | tail -16
(timeout 2400 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
for wl in gs2d_512 gs3d_128; do
python bench.py --workload $wl --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl value %.0f steps/s  fwd %.2f us  bwd %.2f us'%(d['value'], d['fwd_us_per_time_step'], d['bwd_us_per_time_step']))
for k in d['roofline']['all_kernels']: print('     ', k['kernel'], '%.2f us'%k['avg_launch_us'], '%.0f GB/s'%k['achieved'])
"
done
