export TMPDIR=/tmp
timeout 900 python tools/upscaler_share.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/upscaler_share.txt
