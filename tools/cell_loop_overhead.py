"""Host cost of the reference's own per-step loop (train_2drd.py:169-188 with `cell(h)` swapped in):
per time step, forward loop alone and the whole iteration (cat + dense loss + backward)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import percnn_amd as pa
from percnn_amd import synthetic
dev = torch.device("cuda:0")
for n, T in ((100, 200), (512, 100)):
    cell = pa.gs2d_cell(8).to(dev)
    for f in cell.filter_list:
        f.weight.data.mul_(12.0)
    h0 = synthetic.gs_initial_state((n, n), seed=0).to(dev)
    opt = torch.optim.Adam(cell.parameters(), lr=1e-4)

    def loop():
        h, outs = h0, [h0]
        for _ in range(T):
            h, _ = cell(h)
            outs.append(h)
        return outs

    def iteration():
        opt.zero_grad(set_to_none=True)
        loss = (torch.cat(loop(), 0) ** 2).mean()
        loss.backward()
        opt.step()

    def timeit(fn, k=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / k
    with torch.no_grad():
        t_ng = timeit(loop)
    t_fw = timeit(loop)
    t_it = timeit(iteration)
    print(f"{n}^2 x {T}: forward loop {t_ng / T:6.2f} us/step (no_grad)  {t_fw / T:6.2f} us/step (autograd recording)   "
          f"whole iteration {t_it / T:6.2f} us/step ({t_it / 1e3:.2f} ms)", flush=True)
