mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -30) > gpurun_out/test4.log 2>&1
cat gpurun_out/test4.log
cd /tmp
for wl in gs2d_512 lo2d_512; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v3_$wl -o v3_$wl -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl > $GRAFT_REPO_ROOT/gpurun_out/prof_v3_$wl.log 2>&1
done
cd $GRAFT_REPO_ROOT
for wl in gs2d_512 lo2d_512 gs3d_128; do
(timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $wl 2>&1 | tail -1) > gpurun_out/bench4_$wl.log 2>&1
done
python - <<'PY'
import json
for wl in ("gs2d_512","lo2d_512","gs3d_128"):
    try:
        d=json.loads(open(f"gpurun_out/bench4_{wl}.log").read().strip().splitlines()[-1])
        print(wl, "fwd+bwd steps/s %.0f"%d["value"], "fwd us %.2f"%d["roofline"]["fwd_kernel"]["avg_launch_us"], "bwd us/step %.2f"%d["roofline"]["avg_launch_us"])
    except Exception as e: print(wl, "ERR", e, open(f"gpurun_out/bench4_{wl}.log").read()[-500:])
PY
