export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "large_2d or tile_variants" 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
echo "== default"; timeout 900 python tools/size_sweep.py --out gpurun_out/size_sweep.json 2>&1 | grep -v amdgpu.ids
for o in "tile=0" "tile_k=2"; do
echo "== $o"; timeout 600 python tools/size_sweep.py --family gs2d --min-points 1000000 --opt $o --out gpurun_out/size_sweep_$o.json 2>&1 | grep -v amdgpu.ids
done
