mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -15) > gpurun_out/test3.log 2>&1
(PERCNN_FORCE_P2P=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --slab-extra 2>&1 | tail -3) > gpurun_out/bench_slab1.log 2>&1
(timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -2) > gpurun_out/bench3_gs2d.log 2>&1
cat gpurun_out/test3.log gpurun_out/bench_slab1.log gpurun_out/bench3_gs2d.log
