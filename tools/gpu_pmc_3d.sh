export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
: > $R/gpurun_out/r2_pmc_3d_big.txt
for cfg in "384 384 384|4|" "384 384 384|4|l2_tile_kb=0" "256 256 256|6|stream3d=0" "256 256 256|6|" "128 128 128|20|"; do
  shape=$(echo "$cfg" | cut -d'|' -f1); T=$(echo "$cfg" | cut -d'|' -f2); opt=$(echo "$cfg" | cut -d'|' -f3)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcout
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmcout -o pmc -- python $R/tools/opt_sweep.py --family gs3d --shape $shape --T $T --reps 1 --opts "$opt" > /tmp/pmc.log 2>&1
    python $R/tools/pmc_summary.py $(find /tmp/pmcout -name "*.db" | head -1) "$shape T=$T opt=$opt" | grep "pi::pi_\(fwd\|bwd\|stream\)" >> $R/gpurun_out/r2_pmc_3d_big.txt 2>&1
  done
done
cat $R/gpurun_out/r2_pmc_3d_big.txt | cut -c1-220
