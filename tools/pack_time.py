"""Host + device cost of assembling the parameter block (RCNNCell.param_block) on the GPU box: the stock tensor-op assembly,
the registered operator (torch.ops.percnn.pack_block), a plain autograd.Function over the same two kernels, the bare C call."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench, percnn_amd as pa
from percnn_amd import functional as F_pi
dev = torch.device("cuda:0")


class PackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, *tensors):
        ctx.meta = meta
        ctx.save_for_backward(*tensors)
        return F_pi.pack_fwd_hip(list(tensors), *meta)

    @staticmethod
    def backward(ctx, g):
        t = list(ctx.saved_tensors)
        gr = F_pi.pack_bwd_hip(t, g, *ctx.meta)
        return (None, gr[0], gr[1], None, *gr[2:])


def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for wl in ("gs2d_512", "lo2d_512"):
    family, shape, hc, dtype, T, golden = bench.WORKLOADS[wl]
    cell = bench.make_cell(family, bench.load_params(golden), dev, "poly")
    meta = (cell.hidden_channels, cell.ndim, float(cell.dt), float(cell.mu_up or 0.0), cell.diffusion == "sigmoid", True)
    tensors = cell._pack_tensors()

    def old():
        w = cell.W_laplace.weight
        dt_t = torch.tensor([cell.dt], dtype=w.dtype, device=w.device)
        cu, cv = cell.coefficients()
        return F_pi.contract_block(F_pi.pack_params(dt_t, cu, cv, w, tensors[3:]))

    variants = {"tensor ops": old, "param_block()": cell.param_block, "registered op": lambda: torch.ops.percnn.pack_block(tensors, *meta), "autograd.Function": lambda: PackFn.apply(meta, *tensors),
                "C call only": lambda: F_pi.pack_fwd_hip(tensors, *meta)}
    for name, f in variants.items():
        with torch.no_grad():
            a = timeit(f)
        if name == "C call only":
            print(f"{wl} {name:18s}: no_grad {a:6.1f} us")
            continue
        def fb():
            P = f()
            P.backward(torch.ones_like(P))
        b = timeit(fb)
        print(f"{wl} {name:18s}: no_grad {a:6.1f} us   forward+backward {b:6.1f} us (wall, per call)")
