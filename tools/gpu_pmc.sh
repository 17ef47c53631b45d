# HBM traffic of the step kernels from the TCC counters (separate passes, kernel-trace only)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
: > $GRAFT_REPO_ROOT/gpurun_out/pmc_summary.txt
for wl in gs2d_512 gs3d_128 lo2d_512 gs2d_100; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcout
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmcout -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --workload $wl --T 100 > /tmp/pmc.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pmcout -name "*.db" | head -1) "$wl T=100" >> $GRAFT_REPO_ROOT/gpurun_out/pmc_summary.txt 2>&1
done
done
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_summary.txt
