#!/usr/bin/env python3
"""What does the host wait of the resident launches cost a training iteration?  

The resident forward / sweep launches make their entry point wait for the kernel's roll call (a spin on a host-mapped word: the
call returns when the stream has REACHED the launch).  Here: wall time per training iteration -- model() + loss + backward() +
Adam step, the reference's loop body (train_2drd.py:393-409) -- with the wait (persist_handshake=1, default) and without
(persist_handshake=0: fire and forget, an abort surfaces at the next entry point), interleaved on the same box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import percnn_amd as pa


def run(shape, T, iters=30):
    dev = torch.device("cuda:0")
    sd = bench.load_params(bench.WORKLOADS["gs2d_512"][5])
    cell = bench.make_cell("gs2d", sd, dev, "poly")
    h0 = bench.initial_state("gs2d", shape).to(dev)
    model = pa.RCNN(cell, step=T, effective_step=list(range(T)), init_state=h0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-6)

    def iteration():
        opt.zero_grad(set_to_none=True)
        outs, _ = model()
        loss = (outs.stacked ** 2).mean()
        loss.backward()
        opt.step()

    res = {0: [], 1: []}
    for hs in (1, 0):
        pa.set_option("persist_handshake", hs)
        for _ in range(3):
            iteration()
    for rnd in range(5):
        for hs in (1, 0):
            pa.set_option("persist_handshake", hs)
            iteration()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                iteration()
            torch.cuda.synchronize()
            res[hs].append((time.perf_counter() - t0) / iters * 1e3)
    pa.set_option("persist_handshake", 1)
    a, b = float(np.median(res[1])), float(np.median(res[0]))
    print(f"{shape[0]}x{shape[1]} T={T}: training iteration {a:.3f} ms with the host wait, {b:.3f} ms without "
          f"({100.0 * (a - b) / b:+.2f} %; rounds with {['%.3f' % x for x in res[1]]} without {['%.3f' % x for x in res[0]]}); "
          f"persist_status {pa._lib.persist_status()}", flush=True)


if __name__ == "__main__":
    run((100, 100), 200, iters=60)
    run((512, 512), 1000, iters=10)
    run((256, 256), 400, iters=20)
