export TMPDIR=/tmp
(timeout 2400 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout=900 -k "tile or full_size or golden" 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*'.ljust(44), 'value %.0f steps/s  fwd %.2f us  bwd %.2f us'%(d['value'], d['fwd_us_per_time_step'], d['bwd_us_per_time_step']), ' | '.join('%s %.2f us'%(k['kernel'][:12], k['avg_launch_us']) for k in d['roofline']['all_kernels'][:2]))
"; }
run --workload gs2d_512
run --workload gs2d_512 --opt tile_xcd=0
run --workload gs2d_512 --reaction factored
run --workload lo2d_512
run --workload lo2d_512 --opt tile_xcd=0
cd /tmp
for x in 1 0; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcout
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmcout -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload gs2d_512 --T 100 --opt tile_xcd=$x > /tmp/pmc.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pmcout -name "*.db" | head -1) "tile_xcd=$x" | grep "tile_kernel" | cut -c1-150
done; done
