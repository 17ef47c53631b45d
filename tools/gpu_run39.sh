export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*'.ljust(44), 'value %.0f steps/s  fwd %.2f us  bwd %.2f us'%(d['value'], d['fwd_us_per_time_step'], d['bwd_us_per_time_step']), ' | '.join('%s %.2f us'%(k['kernel'][:12], k['avg_launch_us']) for k in d['roofline']['all_kernels'][:2]))
"; }
run --workload gs3d_128 --T 200
run --workload gs3d_128 --T 200 --reaction factored
run --workload gs2d_512 --opt tile=0 --T 400
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
import percnn_amd as pa
from bench import load_params, make_cell
dev=torch.device('cuda:0')
cell=make_cell('gs3d', load_params('gs3d_big_128x128x128.npz'), dev, 'poly')
with torch.no_grad(): P=cell.param_block().contiguous()
for shape,T in (((48,48,48),100),((256,256,256),24),((384,384,384),8),((100,100,100),40)):
    traj=torch.rand((T+1,2)+shape,device=dev)*0.1+0.45
    g=torch.randn_like(traj)*1e-6
    pa.rollout_fwd_(traj,P); pa.rollout_bwd(traj,g,P); torch.cuda.synchronize()
    e=[torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(3): pa.rollout_fwd_(traj,P)
    e[1].record()
    for _ in range(3): pa.rollout_bwd(traj,g,P)
    e[2].record(); torch.cuda.synchronize()
    n=shape[0]*shape[1]*shape[2]
    f=e[0].elapsed_time(e[1])/3/T*1e3; b=e[1].elapsed_time(e[2])/3/T*1e3
    print(shape,'fwd %.2f us/step (%.0f GB/s)  bwd %.2f us/step'%(f, 16*n/f/1e3, b), flush=True)
    del traj,g
PY
