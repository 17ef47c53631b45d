# Round-end evidence: bench lines, rocprofv3 kernel-trace summaries of the same commands, PMC traffic.
#   ROUND=r03 bash tools/gpu_final_profiles.sh        (run on the GPU box, e.g. through gpurun)
ROUND=${ROUND:-r03}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
# the driver's command first: default bench line (headline + also + cpu baselines + module path)
(timeout 900 python $R/bench.py 2>&1 | tail -1) > $O/${ROUND}_final_bench_default.json
for wl in gs3d_128 lo2d_512 gs2d_100 gs3d_48 bur1_100 lo1_100; do
  (timeout 900 python $R/bench.py --workload $wl --no-also 2>&1 | tail -1) > $O/${ROUND}_final_bench_$wl.json
done
for wl in gs2d_512 gs3d_128 lo2d_512 gs2d_100; do
  rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extras --no-also --steps 3 --warmup 1 > /tmp/kt.log 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) > $O/${ROUND}_final_rocprofv3_kernel_stats_$wl.txt 2>&1
  grep "^{" /tmp/kt.log | tail -1 > $O/${ROUND}_final_bench_under_rocprof_$wl.json
done
(timeout 900 python $R/bench.py --workload gs2d_512 --reaction factored --no-cpu-baseline --no-also 2>&1 | tail -1) > $O/${ROUND}_final_bench_gs2d_512_factored.json
# counters at the workloads' own horizons (round 6: the committed traffic figures are no longer T = 100 runs scaled up)
declare -A TDEF=([gs2d_512]=1000 [gs3d_128]=500 [lo2d_512]=400 [gs2d_100]=200)
: > $O/${ROUND}_final_pmc_fetch_write_summary.txt
for wl in gs2d_512 gs3d_128 lo2d_512 gs2d_100; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcout
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmcout -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-also --workload $wl > /tmp/pmc.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pmcout -name "*.db" | head -1) "$wl T=${TDEF[$wl]}" | grep "pi::" >> $O/${ROUND}_final_pmc_fetch_write_summary.txt 2>&1
done
done
cd $R
timeout 900 python tools/size_sweep.py --out gpurun_out/${ROUND}_size_sweep.json 2>&1 | grep -v amdgpu.ids > gpurun_out/${ROUND}_size_sweep.txt
(PERCNN_FORCE_P2P=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-also --slab-extra 2>/dev/null | tail -1) > gpurun_out/${ROUND}_final_bench_gs2d_512_torchrun1_rccl_self.json
for f in gpurun_out/${ROUND}_final_bench_[a-z0-9_]*.json; do case $f in *under_rocprof*) continue;; esac; echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('  value %.0f steps/s  fwd %.2f us  bwd %.2f us'%(d['value'], d['fwd_us_per_time_step'], d['bwd_us_per_time_step']), ' dominant', d['roofline']['kernel'], 'frac %.3f'%d['roofline']['frac'], 'cpu', d.get('cpu_baseline',{}).get('value'), 'also', (d.get('also') or {}).get('gs3d_128',{}).get('value'), 'slab', [(k, round(v.get('us_per_time_step_fwd_bwd', 0), 1)) for k, v in ((d.get('slab_3d') or {}).get('weak_scaling', {}).get('by_transport', {})).items() if isinstance(v, dict)])
"; done
