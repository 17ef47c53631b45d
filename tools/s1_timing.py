"""Debug aid: per-phase device timestamps of the Stage-1 kernels (needs the -DPI_S1_TIMING build)."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ["PERCNN_PI_LIB"] = os.path.join(ROOT, "percnn_amd", "csrc", "libpercnn_pi_dbg.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import percnn_amd as pa
from s1_bench import load_cell
dev = torch.device("cuda:0")
cell, _ = load_cell("bur1", dev)
with torch.no_grad():
    P = cell.param_block().contiguous()
L = pa.lib()
L.percnn_pi_s1_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for shape in ((100, 100),):
    T = 20
    traj = torch.rand((T + 1, 2) + shape, device=dev) * 0.2
    g = torch.randn_like(traj) * 1e-4
    nblk = min(384, (((shape[0] + 3) // 4) * ((shape[1] + 3) // 4) + 3) // 4)
    def stamps():
        buf = (ctypes.c_longlong * (2 * 512 * 8))()
        assert L.percnn_pi_s1_debug_stamps(buf, 2 * 512 * 8) == 0
        st = np.array(buf, dtype=np.int64).reshape(2, 512, 8)[:, :nblk]
        return (st - st[:, :, 0].min()) / 100.0
    for _ in range(3):
        pa.stage1.rollout_fwd_(traj, P)
    torch.cuda.synchronize()
    rel = stamps()
    print(f"fwd {shape}: us since first block start, median | max over {nblk} workgroups (last patch of wave 0)")
    for i, n in enumerate(["start", "weights", "window", "branches", "end"]):
        print(f"   {n:9s} s0/w0 {np.median(rel[0, :, i]):7.2f} {rel[0, :, i].max():7.2f}   s1/w3 {np.median(rel[1, :, i]):7.2f} {rel[1, :, i].max():7.2f}")
    # backward: stop after the sweep's middle launches by running T=2 (launches t=2 (top), t=1 (generic), t=0 (gather only))
    tr2, g2 = traj[:4].contiguous(), g[:4].contiguous()
    for _ in range(3):
        pa.stage1.rollout_bwd(tr2, g2, P)
    torch.cuda.synchronize()
    rel = stamps()
    print(f"adj {shape}: (the stamps are those of the LAST launch that wrote them: t=1 generic for 1..4, t=0 for 0,2)")
    for i, n in enumerate(["start", "weights", "a_t", "branches", "D gemm", "E stored"]):
        print(f"   {n:9s} s0/w0 {np.median(rel[0, :, i]):7.2f} {rel[0, :, i].max():7.2f}   s1/w3 {np.median(rel[1, :, i]):7.2f} {rel[1, :, i].max():7.2f}")
