# Round 6, resident 3D rollouts (tools/res3d/): same-box A/B against the product's brick kernels at 128^3 + SQ counters of the
# resident kernels.  Output: gpurun_out/r06_res3d/.  Build the harness first (see res3d_dev.hip).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_res3d
mkdir -p $O
H=$R/tools/scratch/res3d_dev512
{
echo "## product brick kernels, 128^3 (tools/opt_sweep.py)"
python $R/tools/opt_sweep.py --family gs3d --shape 128 128 128 --T 40 --reps 5 2>&1 | tail -6
echo "## resident harness: all faces handed over / regions 2,2,2; forward T = 500, sweep T = 100"
R3D_ADJ=1 R3D_TADJ=100 R3D_REGIONS=2,2,2 timeout 200 $H 128 500 5 0 2>&1 | grep -v "^  "
echo "## the same, every face written through (no XCD regions)"
R3D_ADJ=1 R3D_TADJ=100 R3D_REGIONS=1,1,1 timeout 200 $H 128 500 5 0 2>&1 | grep "resident\|bitwise\|adjoint dL"
echo "## no hand-over at all (skip 4: WRONG results, compute + LDS + frame / operand traffic only)"
R3D_SKIP=4 R3D_ADJ=1 R3D_TADJ=100 R3D_REGIONS=2,2,2 timeout 200 $H 128 500 5 0 2>&1 | grep "resident"
} > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp
rocprofv3 -L > $O/avail.txt 2>&1
: > $O/records.jsonl; : > $O/log.txt
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"
 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "TCC_EA0_WRREQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
for pass in "${PASSES[@]}"; do
  keep=""
  for ctr in $pass; do
    if grep -qw "$ctr" $O/avail.txt; then keep="$keep $ctr"; else echo "counter $ctr not offered" >> $O/log.txt; fi
  done
  [ -z "$keep" ] && continue
  rm -rf /tmp/pmcout
  R3D_ADJ=1 R3D_TADJ=100 R3D_REGIONS=2,2,2 timeout 300 rocprofv3 --kernel-trace --pmc $keep -d /tmp/pmcout -o pmc -- $H 128 100 1 0 > /tmp/pmc.log 2>&1
  db=$(find /tmp/pmcout -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_dump.py $db res3d_128 $O/records.jsonl 2>> $O/log.txt
  else echo "pass [$keep] produced no database" >> $O/log.txt; tail -5 /tmp/pmc.log >> $O/log.txt; fi
done
python $R/tools/counters_table.py $O/records.jsonl > $O/summary.txt 2>> $O/log.txt
cut -c1-260 $O/summary.txt
tail -5 $O/log.txt
