export TMPDIR=/tmp
show='
import json,sys
d=json.loads(sys.stdin.read()); s=d.get("slab_3d",{})
def line(tag, r):
    print(tag, r.get("ms_per_time_step_fwd_bwd"), "verified" if r.get("forward_state_equals_single_domain_rollout") else "NOT VERIFIED", r.get("error"), r.get("timed_out_exchange", ""), r.get("workload","")[-40:])
line(sys.argv[1], s)
if "peer_mailbox" in s: line(sys.argv[1] + " [peer mailboxes]", s["peer_mailbox"])
'
for ov in 0 1; do
PERCNN_SLAB_OVERLAP=$ov PERCNN_FORCE_P2P=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-also --slab-extra 2>/dev/null | tail -1 | python -c "$show" "self overlap=$ov"
done
