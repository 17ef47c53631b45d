export TMPDIR=/tmp
timeout 600 python tools/tile_timing.py 2>&1 | grep -v amdgpu.ids | tail -16
bash tools/gpu_run34.sh
