# full GPU suite + Stage-1 bench lines / kernel stats + size sweep with the tile heuristic
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
cd /tmp
for wl in bur1_100 lo1_100 bur1_512; do
  (timeout 900 python $R/bench.py --workload $wl 2>&1 | tail -1) > $R/gpurun_out/final_bench_$wl.json
done
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --workload bur1_100 --no-cpu-baseline --steps 3 --warmup 1 > /tmp/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) > $R/gpurun_out/final_kernel_stats_bur1_100.txt 2>&1
cd $R
head -8 gpurun_out/final_kernel_stats_bur1_100.txt | cut -c1-160
for f in gpurun_out/final_bench_bur1_100.json gpurun_out/final_bench_lo1_100.json gpurun_out/final_bench_bur1_512.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('  value %.0f steps/s  fwd %.2f us  bwd %.2f us'%(d['value'], d['fwd_us_per_time_step'], d['bwd_us_per_time_step']), ' dominant', d['roofline']['kernel'], 'frac %.3f'%d['roofline']['frac'], 'cpu', d.get('cpu_baseline',{}).get('value'))
for k in d['roofline']['all_kernels']: print('     ', k['kernel'], '%.2f us'%k['avg_launch_us'], '%.1f TF'%k['achieved'], 'frac %.3f'%k['frac'], 'share %.2f'%k['share_of_pass'])
"; done
timeout 900 python tools/size_sweep.py --out gpurun_out/size_sweep.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/size_sweep.txt
