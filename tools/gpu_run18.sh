export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -k "rccl or slab" 2>&1 | tail -15)
(PERCNN_FORCE_P2P=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --slab-extra 2>&1 | tail -2) | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
d=json.loads(l); print('value %.0f'%d['value'], json.dumps(d.get('slab_3d'))[-330:])
"
