export TMPDIR=/tmp
timeout 900 python tools/upscaler_variants.py 2>&1 | grep -v amdgpu.ids
