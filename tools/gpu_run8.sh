export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -k "stream3d or full_size or slab" 2>&1 | grep -E "^FAILED|passed|failed|Error" | head)
python - <<'PY'
import torch, time, sys
sys.path.insert(0,'.')
import percnn_amd as pa, numpy as np
from bench import load_params, make_cell
dev=torch.device('cuda:0')
cell=make_cell('gs3d', load_params('gs3d_big_128x128x128.npz'), dev)
with torch.no_grad(): P=cell.param_block().contiguous()
for shape in ((256,256,256),(64,256,256),(128,128,128)):
    T=10
    traj=torch.rand((T+1,2)+shape,device=dev)*0.1+0.5
    g=torch.randn_like(traj)*1e-6
    for st in (0,1):
        for zc in ((8,) if st==0 else (4,8,16)):
            pa.set_option('stream3d',st); pa.set_option('zc',zc)
            pa.rollout_fwd_(traj,P); torch.cuda.synchronize()
            e=[torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); 
            for _ in range(3): pa.rollout_fwd_(traj,P)
            e[1].record()
            pa.set_option('skip_wgrad',1)
            for _ in range(3): pa.rollout_bwd(traj,g,P)
            e[2].record(); torch.cuda.synchronize(); pa.set_option('skip_wgrad',0)
            n=np.prod(shape)
            tf=e[0].elapsed_time(e[1])/30*1e3; tb=e[1].elapsed_time(e[2])/30*1e3
            print(shape,'stream3d',st,'zc',zc,'fwd %.1f us (%.0f GB/s)  sweep %.1f us (%.0f GB/s)'%(tf,16*n/tf/1e3,tb,32*n/tb/1e3), flush=True)
PY
