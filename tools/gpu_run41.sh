# hardware counters of the direct 3D kernels at 128^3 (separate passes, kernel-trace only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
: > $R/gpurun_out/pmc_3d.txt
for ctrs in "GRBM_GUI_ACTIVE TA_BUSY_avr MemUnitStalled" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "MeanOccupancyPerCU TCP_TA_TCP_STATE_READ_sum TA_TA_BUSY_sum"; do
  rm -rf /tmp/pmcout
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmcout -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload gs3d_128 --T 40 > /tmp/pmc.log 2>&1
  python - <<PY >> $R/gpurun_out/pmc_3d.txt
import sqlite3, glob
db = glob.glob('/tmp/pmcout/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
name = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "counter" not in x][0]
for r in c.execute(f"select {name}, counter_name, count(*), avg(value) from counters_collection where {name} like '%pi_fwd_kernel%' or {name} like '%pi_bwd_kernel%' or {name} like '%pi_moments%' group by {name}, counter_name"):
    print(f"{r[1]:40s} n={r[2]:5d} avg={r[3]:14.1f} | {r[0][:60]}")
PY
done
cat $R/gpurun_out/pmc_3d.txt
