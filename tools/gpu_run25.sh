export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -8)
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
import percnn_amd as pa, numpy as np
from bench import load_params, make_cell
dev=torch.device('cuda:0')
for fam,gold in (('gs2d','gs2d_big_512x512.npz'),):
  for reaction in ('poly','factored'):
    cell=make_cell(fam, load_params(gold), dev, reaction)
    with torch.no_grad(): P=cell.param_block().contiguous()
    T=200
    traj=torch.rand((T+1,2,100,100),device=dev)*0.1+0.5
    g=torch.randn_like(traj)*1e-6
    for tile in (0,1):
        pa.set_option('tile',tile)
        pa.rollout_fwd_(traj,P); pa.rollout_bwd(traj,g,P); torch.cuda.synchronize()
        e=[torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(5): pa.rollout_fwd_(traj,P)
        e[1].record()
        for _ in range(5): pa.rollout_bwd(traj,g,P)
        e[2].record(); torch.cuda.synchronize()
        print('100^2 T=200',reaction,'tile',tile,'fwd %.2f us/step  bwd %.2f us/step'%(e[0].elapsed_time(e[1])/5/T*1e3, e[1].elapsed_time(e[2])/5/T*1e3), flush=True)
    pa.set_option('tile',1)
PY
