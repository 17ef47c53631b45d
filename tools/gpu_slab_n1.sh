# slab path on ONE rank (32 x 256^2 slab, T = 40 fwd+bwd): device-copy wrap vs RCCL send/recv-to-self vs the peer-mailbox
# transport through the rank's own mailbox; streaming vs direct kernels.  Every line also says whether the forward state
# equals the single-domain rollout bit for bit.
export TMPDIR=/tmp
show='
import json,sys
d=json.loads(sys.stdin.read()); s=d.get("slab_3d",{})
def line(tag, r):
    print(tag, r.get("ms_per_time_step_fwd_bwd"), "verified" if r.get("forward_state_equals_single_domain_rollout") else "NOT VERIFIED", r.get("error"), r.get("timed_out_exchange", ""), r.get("workload","")[-40:])
line(sys.argv[1], s)
if "peer_mailbox" in s: line(sys.argv[1] + " [peer mailboxes]", s["peer_mailbox"])
'
for o in "" "--opt stream3d=0" "--opt stream3d=2"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-also --slab-extra $o 2>/dev/null | tail -1 | python -c "$show" "local-wrap $o"
done
for o in "" "--opt stream3d=0"; do
PERCNN_FORCE_P2P=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-also --slab-extra $o 2>/dev/null | tail -1 | python -c "$show" "rccl-self $o"
done
