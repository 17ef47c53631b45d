export TMPDIR=/tmp
timeout 600 python tools/tile_timing.py 2>&1 | grep -v amdgpu.ids
