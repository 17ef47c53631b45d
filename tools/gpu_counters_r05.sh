# Round 5 counter campaign: SQ / TCP / TCC counters of the dominant kernels at the BASELINE sizes and of the brick
# kernels at 256^3 and on the 32 x 256^2 slab shape.  Separate rocprofv3 passes (kernel-trace + pmc only), names filtered against
# `rocprofv3 -L` so that an unknown counter cannot void a pass.  Output: gpurun_out/r05_counters/{records.jsonl, avail.txt, log.txt}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_counters
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/avail.txt 2>&1
[ -n "$PMC_APPEND" ] || : > $O/records.jsonl
[ -n "$PMC_APPEND" ] || : > $O/log.txt
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"
 "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "TCC_EA0_WRREQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
if [ -n "$PMC_PASSES" ]; then IFS=';' read -r -a PASSES <<< "$PMC_PASSES"; fi
declare -A CMD
# calibration: a device-to-device copy of 128 MiB (known bytes: 134.2 MB read + 134.2 MB written per launch)
CMD[copy_128MiB]="python -c \"import torch; a=torch.rand(1<<25,device='cuda'); b=torch.empty_like(a); [b.copy_(a) for _ in range(6)]; torch.cuda.synchronize()\""
CMD[gs2d_512]="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-also --workload gs2d_512 --T 100"
CMD[gs3d_128]="python $R/tools/opt_sweep.py --family gs3d --shape 128 128 128 --T 20 --reps 1"
CMD[gs3d_256]="python $R/tools/opt_sweep.py --family gs3d --shape 256 256 256 --T 6 --reps 1 --opts \"$OPT256\""
CMD[gs3d_32x256x256]="python $R/tools/opt_sweep.py --family gs3d --shape 32 256 256 --T 20 --reps 1"
CMD[lo2d_512]="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-also --workload lo2d_512 --T 100"
for wl in ${WORKLOADS:-gs2d_512 gs3d_128 gs3d_256 gs3d_32x256x256 lo2d_512}; do
  for pass in "${PASSES[@]}"; do
    keep=""
    for ctr in $pass; do
      if grep -qw "$ctr" $O/avail.txt; then keep="$keep $ctr"; else echo "$wl: counter $ctr not offered by rocprofv3 -L" >> $O/log.txt; fi
    done
    [ -z "$keep" ] && continue
    rm -rf /tmp/pmcout
    eval "timeout 600 rocprofv3 --kernel-trace --pmc $keep -d /tmp/pmcout -o pmc -- ${CMD[$wl]}" > /tmp/pmc.log 2>&1
    rc=$?
    db=$(find /tmp/pmcout -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/pmc_dump.py $db "$wl" $O/records.jsonl 2>> $O/log.txt
    else echo "$wl: pass [$keep] rc=$rc produced no database" >> $O/log.txt; tail -5 /tmp/pmc.log >> $O/log.txt; fi
  done
done
python $R/tools/counters_table.py $O/records.jsonl > $O/summary.txt 2>> $O/log.txt
cat $O/summary.txt | cut -c1-250
tail -30 $O/log.txt
