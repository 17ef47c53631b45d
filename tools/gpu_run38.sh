export TMPDIR=/tmp
run() { python bench.py --workload gs3d_128 --no-cpu-baseline --T 200 "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*'.ljust(40), 'value %.0f steps/s  fwd %.2f us  bwd %.2f us'%(d['value'], d['fwd_us_per_time_step'], d['bwd_us_per_time_step']), ' | '.join('%s %.2f us'%(k['kernel'][:12], k['avg_launch_us']) for k in d['roofline']['all_kernels'][:2]))
"; }
run
for zc in 1 2 4 8; do run --opt stream3d=2 --opt zc=$zc; done
