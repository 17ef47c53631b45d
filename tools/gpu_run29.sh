export TMPDIR=/tmp
timeout 900 python tools/s1_bench.py --cpu 2>&1 | grep -v amdgpu.ids
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_s1 -o s1 -- python $GRAFT_REPO_ROOT/tools/s1_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $(find /tmp/prof_s1 -name "*.db" | head -1) 2>&1 | tee gpurun_out/s1_kernel_stats.txt | head -20
