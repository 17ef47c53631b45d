"""Shared helpers for the test-suite (golden fixtures, error norms)."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 tolerances (BASELINE.json: <= 1e-5 rel-L2 vs the reference path); fp64 runs are held to 1e-12.
TOL_TRAJ = {np.dtype("float32"): 1e-5, np.dtype("float64"): 1e-12}
TOL_GRAD = {np.dtype("float32"): 2e-5, np.dtype("float64"): 1e-10}


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def small_cases():
    fns = sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.npz"))
                 if "_big_" not in f and "_biggrad_" not in f and "_longgrad" not in f and "harness" not in f and "stage3" not in f and "stage1" not in f
                 and "train_iter" not in f)
    assert fns, "golden fixtures missing"
    return fns


def case_id(fn):
    return os.path.basename(fn)[:-4]


def family(fn):
    return os.path.basename(fn).split("_")[0]


class Golden:
    def __init__(self, fn):
        self.fn = fn
        self.z = np.load(fn)
        self.family = family(fn)
        self.sd = {k[6:]: self.z[k] for k in self.z.files if k.startswith("param/")}
        self.h0 = self.z["h0"]                       # [1,2,*S]
        self.dtype = self.h0.dtype
        self.ndim = self.h0.ndim - 2
        self.hc = self.sd["Wh1_u.weight"].shape[0]
        self.steps = int(self.z["steps"])
        self.keep_t = [int(t) for t in self.z["keep_t"]]
        self.stride_t = int(self.z["stride_t"])
        self.dt = float(self.z["dt"])
        self.dx = float(self.z["dx"])
        self.mu_up = float(self.z["mu_up"])

    def traj(self, t):
        return self.z[f"traj/{t}"]

    def grads(self, loss):
        pre = f"grad_{loss}/"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}

    def oracle_cell(self):
        from oracle import restatement as R
        cell = {"gs2d": R.gs2d_cell, "gs3d": R.gs3d_cell, "lo2d": R.lo2d_cell}[self.family]()
        cell.load_state_dict({k: torch.tensor(v) for k, v in self.sd.items()})
        return cell

    def product_cell(self, device, reaction="poly"):
        import percnn_amd as pa
        cell = {"gs2d": pa.gs2d_cell, "gs3d": pa.gs3d_cell, "lo2d": pa.lo2d_cell}[self.family](reaction=reaction)
        cell.load_state_dict({k: torch.tensor(v) for k, v in self.sd.items()})
        return cell.to(device)

    def packed(self, reaction="factored"):
        """Parameter block for the plain-C oracle, coefficient computed as the reference does."""
        from oracle import pi_oracle as O
        cell = self.oracle_cell()
        cu, cv = [c.detach().numpy() for c in cell.coefficients()]
        if reaction == "poly":
            return O.pack_poly(self.sd, self.dt, cu, cv, self.dtype)
        return O.pack_params(self.sd, self.dt, cu, cv, self.dtype)

    def named_grads_from_packed(self, pg):
        """Packed gradient block -> reference parameter names (chain through sigmoid where needed)."""
        from oracle import pi_oracle as O
        G = O.unpack_grads(np.asarray(pg, dtype=np.float64), self.hc, self.ndim)
        out = {k: v for k, v in G.items() if not k.startswith("coef_")}
        if "CA" in self.sd:
            for n, c in (("CA", "coef_u"), ("CB", "coef_v")):
                s = 1.0 / (1.0 + np.exp(-self.sd[n].astype(np.float64)))
                out[n] = G[c] * self.mu_up * s * (1 - s)
        else:
            out["DA"], out["DB"] = G["coef_u"], G["coef_v"]
        return out


def data_loss(traj, stride_t, ndim):
    sl = (slice(0, -1, stride_t), slice(None)) + (slice(None, None, 4),) * ndim
    return ((traj[sl] - 0.5) ** 2).mean()


# ---- oracle dispatch on the block kind (36 entries = pre-contracted polynomial block, "hc = 0") ----
def hc_of(P):
    return 0 if len(P) == 36 else ((len(P) - 16) // 2 - 1) // 10


def o_step_fwd(h, P):
    from oracle import pi_oracle as O
    return O.poly_step_fwd(h, P) if len(P) == 36 else O.step_fwd(h, P, hc_of(P))


def o_step_bwd(h, G, inj, P):
    from oracle import pi_oracle as O
    return O.poly_step_bwd(h, G, inj, P) if len(P) == 36 else O.step_bwd(h, G, inj, P, hc_of(P))


def o_rollout_fwd(h0, P, T):
    from oracle import pi_oracle as O
    return O.poly_rollout_fwd(h0, P, T) if len(P) == 36 else O.rollout_fwd(h0, P, hc_of(P), T)


def o_rollout_bwd(traj, g, P):
    from oracle import pi_oracle as O
    return O.poly_rollout_bwd(traj, g, P) if len(P) == 36 else O.rollout_bwd(traj, g, P, hc_of(P))


def random_block(hc, ndim, dtype, seed, scale=0.5):
    """Random but well-conditioned parameter block (hc = 0: polynomial block of 36 entries)."""
    rs = np.random.RandomState(seed)
    n = 36 if hc == 0 else 16 + 2 * (10 * hc + 1)
    P = np.zeros(n, dtype=dtype)
    P[0] = 0.1
    P[1:3] = rs.uniform(0.01, 0.05, 2)
    P[3] = -2.0 * ndim * 1.25
    for a in range(ndim):
        P[4 + 4 * a:8 + 4 * a] = (-1 / 12, 4 / 3, 4 / 3, -1 / 12) + rs.uniform(-0.01, 0.01, 4)  # asymmetric on purpose
    P[16:] = rs.uniform(-scale, scale, n - 16)
    return P
