mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -15) > gpurun_out/test2.log 2>&1
cd /tmp
for wl in gs2d_512 gs3d_128 lo2d_512; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v1_$wl -o v1_$wl -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl > $GRAFT_REPO_ROOT/gpurun_out/prof_v1_$wl.log 2>&1
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/test2.log
find gpurun_out/prof_v1_gs2d_512 -type f | head
