# Access-pattern ceiling of the 3D brick kernels (VERDICT r4 #1c): the same launches with the reaction term, the Jacobian, the
# moments and the coefficient sums compiled out (-DPI_BRICK_SKELETON: window / halo loads, LDS staging, barrier, the 13 stencil
# taps per species and the stores remain; results are WRONG by construction) against the product library, same box, interleaved.
# Build the skeleton library first (CPU container):
#   python -c "from percnn_amd import _lib; _lib.build(extra_flags=['-DPI_BRICK_SKELETON'], out='tools/scratch/libpercnn_pi_skel.so')"
R=${GRAFT_REPO_ROOT:-.}
for shp in "128 128 128|40" "32 256 256|40" "256 256 256|8" "48 48 48|100"; do
  shape=$(echo "$shp" | cut -d'|' -f1); T=$(echo "$shp" | cut -d'|' -f2)
  for i in 1 2; do
    echo "product  : $(python $R/tools/opt_sweep.py --family gs3d --shape $shape --T $T --reps 3 --rounds 3 --opts "stream3d=0" 2>&1 | grep gs3d)"
    echo "skeleton : $(PERCNN_PI_LIB=$R/tools/scratch/libpercnn_pi_skel.so python $R/tools/opt_sweep.py --family gs3d --shape $shape --T $T --reps 3 --rounds 3 --opts "stream3d=0" 2>&1 | grep gs3d)"
  done
  ./tools/ubench/stream_mix $shape $T 5 2>&1 | grep -v "^#" | sed 's/^/floor    : /'
done
