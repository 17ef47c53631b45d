export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout=900 -k "tile or golden or full_size or ragged or rollout" 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
for wl in gs2d_512 lo2d_512; do
for r in poly factored; do
python bench.py --workload $wl --reaction $r --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl $r value %.0f steps/s  fwd %.2f us  bwd %.2f us'%(d['value'], d['fwd_us_per_time_step'], d['bwd_us_per_time_step']))
for k in d['roofline']['all_kernels']: print('     ', k['kernel'], '%.2f us'%k['avg_launch_us'], '%.0f GB/s'%k['achieved'])
"
done; done
