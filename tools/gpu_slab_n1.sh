export TMPDIR=/tmp
for o in "" "--opt stream3d=0" "--opt stream3d=2"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-also --slab-extra $o 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('slab_3d',{}); print('local-wrap', '$o', s.get('ms_per_time_step_fwd_bwd'), s.get('error'))"
done
for o in "" "--opt stream3d=0"; do
PERCNN_FORCE_P2P=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-also --slab-extra $o 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('slab_3d',{}); print('rccl-self', '$o', s.get('ms_per_time_step_fwd_bwd'), s.get('error'), s.get('workload','')[-60:])"
done
