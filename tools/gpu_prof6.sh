mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for wl in gs2d_512 gs3d_128; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v4_$wl -o v4_$wl -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl --opt tile_nt=512 > $GRAFT_REPO_ROOT/gpurun_out/prof_v4_$wl.log 2>&1
done
