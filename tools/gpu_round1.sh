mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -40) > gpurun_out/test1.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke1.log 2>&1
(timeout 600 python bench.py --steps 3 --warmup 1 2>&1 | tail -3) > gpurun_out/bench_gs2d.log 2>&1
(timeout 600 python bench.py --steps 2 --warmup 1 --workload gs3d_128 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_gs3d.log 2>&1
(timeout 600 python bench.py --steps 2 --warmup 1 --workload lo2d_512 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_lo2d.log 2>&1
cat gpurun_out/test1.log gpurun_out/smoke1.log gpurun_out/bench_*.log
