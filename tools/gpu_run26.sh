export TMPDIR=/tmp
timeout 1500 python tools/size_sweep.py --out gpurun_out/size_sweep.json 2>&1 | tail -40
