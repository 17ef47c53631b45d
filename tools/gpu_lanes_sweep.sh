# lanes-along-x rule of the direct kernels (set_blockmap): pre-round-2 rule (lane_x=-1) vs the fitted one (default), many widths
for s in "48 48 48" "64 64 64" "80 80 80" "96 96 96" "112 112 112" "128 128 128" "144 144 144" "160 160 160" "176 176 176" "192 192 192" "200 200 200" "208 208 208" "224 224 224" "240 240 240" "288 288 288" "320 320 320" "352 352 352" "384 384 384" "100 100 100" "32 160 160"; do
  python tools/opt_sweep.py --family gs3d --shape $s --T 20 --reps 3 --rounds 5 --check --opts "lane_x=-1" "" 2>&1 | grep "gs3d  "
done
for s in "2048 2000" "1800 1800" "2048 2048" "3000 1200" "1600 2400"; do
  python tools/opt_sweep.py --family gs2d --shape $s --T 20 --reps 3 --rounds 5 --check --opts "lane_x=-1" "" 2>&1 | grep "gs2d  "
done
for s in "1200 1200" "2048 2048" "1504 1504"; do
  python tools/opt_sweep.py --family lo2d --shape $s --T 20 --reps 3 --rounds 5 --check --opts "lane_x=-1" "" 2>&1 | grep "lo2d  "
done
