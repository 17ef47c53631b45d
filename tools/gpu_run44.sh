export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
timeout 900 python tools/upscaler_share.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/upscaler_share.txt
