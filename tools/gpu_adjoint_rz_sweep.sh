# planes per pass of the direct 3D adjoint kernel (direct_rz): rz=1 vs the default rule vs rz=2 (the forward keeps its own rule:
# read the bwd column)
for s in "48 48 48" "64 64 64" "80 80 80" "96 96 96" "100 100 100" "112 112 112" "128 128 128" "144 144 144" "160 160 160" "176 176 176" "192 192 192" "200 200 200" "32 256 256" "64 128 128" "96 128 128" "120 128 128" "125 128 128" "130 126 128" "136 128 128" "160 128 128" "100 160 128" "128 120 136" "32 160 160" "24 256 256" "48 256 256"; do
  python tools/opt_sweep.py --family gs3d --shape $s --T 30 --reps 3 --rounds 5 --check --opts "rz=1" "" "rz=2" 2>&1 | grep "gs3d  "
done
