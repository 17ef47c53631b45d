"""Debug aid: per-phase device timestamps of pi_fwd2d_tile_kernel (needs the -DPI_TILE_TIMING build)."""
import ctypes, os, sys
os.environ["PERCNN_PI_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "percnn_amd", "csrc",
                                           sys.argv[1] if len(sys.argv) > 1 else "libpercnn_pi_dbg.so")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import percnn_amd as pa
from bench import load_params, make_cell
dev = torch.device("cuda:0")
for reaction in ("poly",):
    cell = make_cell("gs2d", load_params("gs2d_big_512x512.npz"), dev, reaction)
    with torch.no_grad():
        P = cell.param_block().contiguous()
    traj = torch.rand((41, 2, 512, 512), device=dev) * 0.1 + 0.5
    for nt in (512,):
        pa.set_option("tile_nt", nt)
        for _ in range(3):
            pa.rollout_fwd_(traj, P)
        torch.cuda.synchronize()
        L = pa.lib()
        buf = (ctypes.c_longlong * (256 * 16))()
        L.percnn_pi_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert L.percnn_pi_debug_stamps(buf, 256 * 16) == 0
        st = np.array(buf, dtype=np.int64).reshape(256, 16)
        t0 = st[:, 0].min()
        rel = (st - t0) / 100.0            # wall_clock64: 100 MHz constant clock -> 10 ns ticks -> us/100
        names = ["start", "loaded", "c0", "s0", "c1", "s1", "c2", "s2", "c3", "s3"] + ["-"] * 5 + ["end"]
        print(f"{reaction} NT={nt}: last launch, us since first block start (median over 256 blocks | max)")
        for i in [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15]:
            print(f"   {names[i]:7s} median {np.median(rel[:, i]):7.2f}  min {rel[:, i].min():7.2f}  max {rel[:, i].max():7.2f}")
        print(f"   launch boundary: first start of this launch - last end of the previous one = {(st[:, 0].min() - st[:, 14].max()) / 100.0:.2f} us; "
              f"launch period (end - previous end, median) = {np.median(st[:, 15] - st[:, 14]) / 100.0:.2f} us")
        # adjoint sweep (sweep only)
        g = torch.randn_like(traj) * 1e-6
        pa.set_option("skip_wgrad", 1)
        for _ in range(2):
            pa.rollout_bwd(traj, g, P)
        torch.cuda.synchronize()
        pa.set_option("skip_wgrad", 0)
        assert L.percnn_pi_debug_stamps(buf, 256 * 16) == 0
        st = np.array(buf, dtype=np.int64).reshape(256, 16)
        rel = (st - st[:, 0].min()) / 100.0
        an = ["start", "window"] + [f"{w}{m}" for m in range(4) for w in ("comp", "barr", "stor")] + ["-", "end"]
        print(f"{reaction} NT={nt}: ADJOINT last launch")
        for i in list(range(14)) + [15]:
            print(f"   {an[i]:7s} median {np.median(rel[:, i]):7.2f}  min {rel[:, i].min():7.2f}  max {rel[:, i].max():7.2f}")
        print(f"   launch boundary: {(st[:, 0].min() - st[:, 14].max()) / 100.0:.2f} us; launch period (median) = "
              f"{np.median(st[:, 15] - st[:, 14]) / 100.0:.2f} us")
        # per-wave view: when does each of the 8 waves finish the compute phase of every sub-step (relative to its block's start)?
        wb = (ctypes.c_longlong * (256 * 16 * 16))()
        L.percnn_pi_debug_wave_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert L.percnn_pi_debug_wave_stamps(wb, 256 * 16 * 16) == 0
        ws = np.array(wb, dtype=np.int64).reshape(256, 16, 16)[:, :nt // 64]
        wrel = (ws - ws[:, :, 0:1].min(axis=1, keepdims=True)) / 100.0
        print("   per-wave medians over the 256 workgroups (us since the block's first wave started): window | comp0 barr0 | comp1 barr1 | comp2 barr2 | comp3 barr3 | end")
        for w in range(nt // 64):
            r = np.median(wrel[:, w], axis=0)
            print(f"     wave {w}: {r[1]:5.2f} | {r[2]:5.2f} {r[3]:5.2f} | {r[5]:5.2f} {r[6]:5.2f} | {r[8]:5.2f} {r[9]:5.2f} | {r[11]:5.2f} {r[12]:5.2f} | {r[15]:5.2f}")
