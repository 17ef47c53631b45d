# 2D tile cliff past 512^2: 32x32 tiles (tile_wide=0) vs the wide-tile rule, same box, interleaved rounds
mkdir -p gpurun_out
O=gpurun_out/${ROUND:-r03}_tile_cliff.txt
: > $O
for n in 512 528 544 560 576 608 640 672 704; do
  timeout 300 python tools/opt_sweep.py --family gs2d --shape $n $n --T 200 --reps 3 --rounds 5 --check --opts "tile_wide=0" "" 2>&1 | grep -v amdgpu.ids >> $O
done
timeout 300 python tools/opt_sweep.py --family gs2d --shape 544 544 --T 200 --reps 3 --rounds 5 --check --opts "tile_wide=0" "tile_wide=1" "tile_wide=2" 2>&1 | grep -v amdgpu.ids >> $O
timeout 300 python tools/opt_sweep.py --family gs2d --shape 512 512 --T 200 --reps 3 --rounds 5 --check --opts "tile_wide=0" "tile_wide=1" "tile_wide=2" 2>&1 | grep -v amdgpu.ids >> $O
cat $O
