export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_stage1.py -m gpu -q --timeout=600 2>&1 | tail -3
timeout 900 python tools/s1_bench.py 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/s1_timing.py 2>&1 | grep -v amdgpu.ids
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_s1 -o s1 -- python $GRAFT_REPO_ROOT/tools/s1_bench.py --only 100 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $(find /tmp/prof_s1 -name "*.db" | head -1) 2>&1 | tee gpurun_out/s1_kernel_stats_100.txt | head -8
