for shp in "256 256 256|8" "32 256 256|40" "128 128 128|40" "64 256 256|20" "192 192 192|12" "16 256 256|40"; do
  shape=$(echo "$shp" | cut -d'|' -f1); T=$(echo "$shp" | cut -d'|' -f2)
  python tools/opt_sweep.py --family gs3d --shape $shape --T $T --reps 3 --rounds 3 --check --opts "brick_xny=-1" "" "brick_xny=2" "brick_xny=4" "brick_xny=8" "stream3d=0" "stream3d=0,brick_xny=8" "stream3d=0,brick_xny=4" 2>&1 | grep gs3d
done
