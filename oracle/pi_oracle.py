"""ctypes front-end of the plain-C oracle (TEST INFRASTRUCTURE ONLY).

Independent restatement of the parameter-block layout and of the map from packed
gradients back to the reference's ``state_dict`` names; cites the same reference lines as
``pi_oracle_impl.h``.  Never imported by ``percnn_amd``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libpi_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(so)
    return _LIB


def star_taps(w_laplace: np.ndarray):
    """Extract (centre, taps[ndim][4]) from a dense [1,1,5,5(,5)] stencil; assert star support."""
    w = np.asarray(w_laplace)
    ndim = w.ndim - 2
    w = w.reshape(w.shape[2:])
    c = (2,) * ndim
    taps = np.zeros((3, 4), dtype=w.dtype)
    mask = np.zeros_like(w, dtype=bool)
    mask[c] = True
    for ax in range(ndim):
        for i, off in enumerate((-2, -1, 1, 2)):
            idx = list(c)
            idx[ax] += off
            taps[ax, i] = w[tuple(idx)]
            mask[tuple(idx)] = True
    assert np.all(w[~mask] == 0), "stencil is not star-shaped"
    return w[c], taps


def pack_params(sd: dict, dt: float, coef_u: float, coef_v: float, dtype) -> np.ndarray:
    """state_dict-like {name: ndarray} -> parameter block P (layout in pi_oracle_impl.h)."""
    hc = sd["Wh1_u.weight"].shape[0]
    P = np.zeros(16 + 2 * (10 * hc + 1), dtype=dtype)
    P[0], P[1], P[2] = dt, coef_u, coef_v
    c, taps = star_taps(sd["W_laplace.weight"])
    P[3] = c
    P[4:16] = taps.reshape(-1)
    for s, name in enumerate("uv"):
        base = 16 + s * (10 * hc + 1)
        for k in (1, 2, 3):
            w = np.asarray(sd[f"Wh{k}_{name}.weight"]).reshape(hc, 2)
            b = np.asarray(sd[f"Wh{k}_{name}.bias"]).reshape(hc)
            for j in range(hc):
                P[base + 10 * j + 3 * (k - 1) + 0] = w[j, 0]
                P[base + 10 * j + 3 * (k - 1) + 1] = w[j, 1]
                P[base + 10 * j + 3 * (k - 1) + 2] = b[j]
        w4 = np.asarray(sd[f"Wh4_{name}.weight"]).reshape(hc)
        for j in range(hc):
            P[base + 10 * j + 9] = w4[j]
        P[base + 10 * hc] = np.asarray(sd[f"Wh4_{name}.bias"]).reshape(())
    return P


def unpack_grads(pg: np.ndarray, hc: int, ndim: int) -> dict:
    """packed gradient block -> {reference parameter name: ndarray}; 'coef_u','coef_v' are the
    gradients w.r.t. the diffusion coefficients (chain through sigmoid is the caller's)."""
    out = {"coef_u": pg[1], "coef_v": pg[2]}
    one = (1,) * ndim
    for s, name in enumerate("uv"):
        base = 16 + s * (10 * hc + 1)
        blk = pg[base:base + 10 * hc].reshape(hc, 10)
        for k in (1, 2, 3):
            out[f"Wh{k}_{name}.weight"] = blk[:, 3 * (k - 1):3 * (k - 1) + 2].reshape((hc, 2) + one).copy()
            out[f"Wh{k}_{name}.bias"] = blk[:, 3 * (k - 1) + 2].copy()
        out[f"Wh4_{name}.weight"] = blk[:, 9].reshape((1, hc) + one).copy()
        out[f"Wh4_{name}.bias"] = pg[base + 10 * hc].reshape(1).copy()
    return out


def _ct(dtype):
    return (ctypes.c_float, "f32") if np.dtype(dtype) == np.float32 else (ctypes.c_double, "f64")


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _shape(S):
    return (ctypes.c_long * len(S))(*S)


def step_fwd(h: np.ndarray, P: np.ndarray, hc: int) -> np.ndarray:
    """h: [2,*S] -> next state [2,*S]."""
    ct, suf = _ct(h.dtype)
    h = np.ascontiguousarray(h)
    out = np.empty_like(h)
    S = h.shape[1:]
    getattr(lib(), "pi_oracle_step_fwd_" + suf)(_ptr(h, ct), _ptr(out, ct), _ptr(P, ct), hc, len(S), _shape(S))
    return out


def step_bwd(h, G, inj, P, hc):
    """-> (Gprev [2,*S], pg double[len(P)])"""
    ct, suf = _ct(h.dtype)
    h, G = np.ascontiguousarray(h), np.ascontiguousarray(G)
    S = h.shape[1:]
    Gp = np.empty_like(h)
    pg = np.zeros(P.shape[0], dtype=np.float64)
    injp = _ptr(np.ascontiguousarray(inj), ct) if inj is not None else None
    getattr(lib(), "pi_oracle_step_bwd_" + suf)(_ptr(h, ct), _ptr(G, ct), injp, _ptr(Gp, ct),
                                                _ptr(pg, ctypes.c_double), _ptr(P, ct), hc, len(S), _shape(S))
    return Gp, pg


def rollout_fwd(h0: np.ndarray, P: np.ndarray, hc: int, T: int) -> np.ndarray:
    """h0: [2,*S] -> traj [T+1,2,*S]"""
    ct, suf = _ct(h0.dtype)
    S = h0.shape[1:]
    traj = np.empty((T + 1,) + h0.shape, dtype=h0.dtype)
    traj[0] = h0
    getattr(lib(), "pi_oracle_rollout_fwd_" + suf)(_ptr(traj, ct), _ptr(P, ct), hc, len(S), _shape(S), T)
    return traj


def rollout_bwd(traj: np.ndarray, gtraj: np.ndarray, P: np.ndarray, hc: int):
    """-> (dL/dh0 [2,*S], pg double[len(P)])"""
    ct, suf = _ct(traj.dtype)
    T = traj.shape[0] - 1
    S = traj.shape[2:]
    traj, gtraj = np.ascontiguousarray(traj), np.ascontiguousarray(gtraj)
    g0 = np.empty(traj.shape[1:], dtype=traj.dtype)
    work = np.empty((2,) + traj.shape[1:], dtype=traj.dtype)
    pg = np.zeros(P.shape[0], dtype=np.float64)
    getattr(lib(), "pi_oracle_rollout_bwd_" + suf)(_ptr(traj, ct), _ptr(gtraj, ct), _ptr(g0, ct),
                                                   _ptr(pg, ctypes.c_double), _ptr(work, ct), _ptr(P, ct),
                                                   hc, len(S), _shape(S), T)
    return g0, pg


# ---- range-restricted steps (used by the slab / halo-exchange tests) -------------------------------
def step_fwd_range(h: np.ndarray, out: np.ndarray, P: np.ndarray, hc: int, lo: int, hi: int) -> np.ndarray:
    """Planes [lo, hi) of axis 0 of ``out`` are overwritten with the step applied to ``h``."""
    ct, suf = _ct(h.dtype)
    assert h.flags.c_contiguous and out.flags.c_contiguous
    S = h.shape[1:]
    getattr(lib(), "pi_oracle_step_fwd_range_" + suf)(_ptr(h, ct), _ptr(out, ct), _ptr(P, ct), hc, len(S), _shape(S),
                                                      ctypes.c_long(lo), ctypes.c_long(hi))
    return out


def step_bwd_range(h, G, inj, Gp, pg, P, hc, lo, hi):
    """Adjoint restricted to planes [lo, hi): writes those planes of ``Gp``, accumulates into ``pg``."""
    ct, suf = _ct(h.dtype)
    assert all(a.flags.c_contiguous for a in (h, G, Gp)) and pg.dtype == np.float64
    S = h.shape[1:]
    injp = _ptr(inj, ct) if inj is not None else None
    getattr(lib(), "pi_oracle_step_bwd_range_" + suf)(_ptr(h, ct), _ptr(G, ct), injp, _ptr(Gp, ct),
                                                      _ptr(pg, ctypes.c_double), _ptr(P, ct), hc, len(S), _shape(S),
                                                      ctypes.c_long(lo), ctypes.c_long(hi))
    return Gp, pg


# ---- pre-contracted polynomial reaction ("poly" mode) ----------------------------------------------
MONOMIALS = {(0, 0): 0, (1, 0): 1, (0, 1): 2, (2, 0): 3, (1, 1): 4, (0, 2): 5, (3, 0): 6, (2, 1): 7, (1, 2): 8, (0, 3): 9}


def expand_poly(sd: dict, name: str) -> np.ndarray:
    """float64 coefficients c[10] of r(u,v) = Wh4(Wh1(h)*Wh2(h)*Wh3(h)) for species `name`
    (train_2drd.py:115-116; cf. the reference's own symbolic read-out train_3drd.py:442-468).
    Plain triple loop over the (u, v, 1) factors of the three linear forms."""
    hc = sd[f"Wh1_{name}.weight"].shape[0]
    L = [np.concatenate([np.asarray(sd[f"Wh{k}_{name}.weight"], dtype=np.float64).reshape(hc, 2),
                         np.asarray(sd[f"Wh{k}_{name}.bias"], dtype=np.float64).reshape(hc, 1)], 1) for k in (1, 2, 3)]
    w4 = np.asarray(sd[f"Wh4_{name}.weight"], dtype=np.float64).reshape(hc)
    ex = [(1, 0), (0, 1), (0, 0)]
    c = np.zeros(10)
    for a in range(3):
        for b in range(3):
            for d in range(3):
                e = (ex[a][0] + ex[b][0] + ex[d][0], ex[a][1] + ex[b][1] + ex[d][1])
                c[MONOMIALS[e]] += np.sum(w4 * L[0][:, a] * L[1][:, b] * L[2][:, d])
    c[0] += float(np.asarray(sd[f"Wh4_{name}.bias"], dtype=np.float64).reshape(()))
    return c


def pack_poly(sd: dict, dt, coef_u, coef_v, dtype) -> np.ndarray:
    P = pack_params(sd, dt, coef_u, coef_v, dtype)
    Q = np.zeros(36, dtype=dtype)
    Q[:16] = P[:16]
    Q[16:26] = expand_poly(sd, "u").astype(dtype)
    Q[26:36] = expand_poly(sd, "v").astype(dtype)
    return Q


def poly_step_fwd(h, Q):
    ct, suf = _ct(h.dtype)
    h = np.ascontiguousarray(h)
    out = np.empty_like(h)
    S = h.shape[1:]
    getattr(lib(), "pi_oracle_poly_step_fwd_range_" + suf)(_ptr(h, ct), _ptr(out, ct), _ptr(Q, ct), len(S), _shape(S),
                                                           ctypes.c_long(0), ctypes.c_long(S[0]))
    return out


def poly_step_bwd(h, G, inj, Q):
    ct, suf = _ct(h.dtype)
    h, G = np.ascontiguousarray(h), np.ascontiguousarray(G)
    S = h.shape[1:]
    Gp = np.empty_like(h)
    qg = np.zeros(36, dtype=np.float64)
    injp = _ptr(np.ascontiguousarray(inj), ct) if inj is not None else None
    getattr(lib(), "pi_oracle_poly_step_bwd_range_" + suf)(_ptr(h, ct), _ptr(G, ct), injp, _ptr(Gp, ct),
                                                           _ptr(qg, ctypes.c_double), _ptr(Q, ct), len(S), _shape(S),
                                                           ctypes.c_long(0), ctypes.c_long(S[0]))
    return Gp, qg


def poly_rollout_fwd(h0, Q, T):
    ct, suf = _ct(h0.dtype)
    S = h0.shape[1:]
    traj = np.empty((T + 1,) + h0.shape, dtype=h0.dtype)
    traj[0] = h0
    getattr(lib(), "pi_oracle_poly_rollout_fwd_" + suf)(_ptr(traj, ct), _ptr(Q, ct), len(S), _shape(S), T)
    return traj


def poly_rollout_bwd(traj, gtraj, Q):
    ct, suf = _ct(traj.dtype)
    T = traj.shape[0] - 1
    S = traj.shape[2:]
    traj, gtraj = np.ascontiguousarray(traj), np.ascontiguousarray(gtraj)
    g0 = np.empty(traj.shape[1:], dtype=traj.dtype)
    work = np.empty((2,) + traj.shape[1:], dtype=traj.dtype)
    qg = np.zeros(36, dtype=np.float64)
    getattr(lib(), "pi_oracle_poly_rollout_bwd_" + suf)(_ptr(traj, ct), _ptr(gtraj, ct), _ptr(g0, ct),
                                                        _ptr(qg, ctypes.c_double), _ptr(work, ct), _ptr(Q, ct), len(S),
                                                        _shape(S), T)
    return g0, qg


def poly_grads_to_params(qg: np.ndarray, sd: dict) -> dict:
    """Chain rule dL/dc -> dL/d{Wh1..4 weights, biases} by finite-difference-free analytic
    differentiation of expand_poly (multilinear in the factors): float64, independent of torch."""
    out = {"coef_u": qg[1], "coef_v": qg[2]}
    ex = [(1, 0), (0, 1), (0, 0)]
    for s, name in enumerate("uv"):
        hc = sd[f"Wh1_{name}.weight"].shape[0]
        ndim = np.asarray(sd[f"Wh1_{name}.weight"]).ndim - 2
        one = (1,) * ndim
        L = [np.concatenate([np.asarray(sd[f"Wh{k}_{name}.weight"], dtype=np.float64).reshape(hc, 2),
                             np.asarray(sd[f"Wh{k}_{name}.bias"], dtype=np.float64).reshape(hc, 1)], 1) for k in (1, 2, 3)]
        w4 = np.asarray(sd[f"Wh4_{name}.weight"], dtype=np.float64).reshape(hc)
        M = qg[16 + 10 * s:26 + 10 * s]
        gL = [np.zeros((hc, 3)) for _ in range(3)]
        gw4 = np.zeros(hc)
        for a in range(3):
            for b in range(3):
                for d in range(3):
                    e = (ex[a][0] + ex[b][0] + ex[d][0], ex[a][1] + ex[b][1] + ex[d][1])
                    m = M[MONOMIALS[e]]
                    gw4 += m * L[0][:, a] * L[1][:, b] * L[2][:, d]
                    gL[0][:, a] += m * w4 * L[1][:, b] * L[2][:, d]
                    gL[1][:, b] += m * w4 * L[0][:, a] * L[2][:, d]
                    gL[2][:, d] += m * w4 * L[0][:, a] * L[1][:, b]
        for k in (1, 2, 3):
            out[f"Wh{k}_{name}.weight"] = gL[k - 1][:, :2].reshape((hc, 2) + one).copy()
            out[f"Wh{k}_{name}.bias"] = gL[k - 1][:, 2].copy()
        out[f"Wh4_{name}.weight"] = gw4.reshape((1, hc) + one)
        out[f"Wh4_{name}.bias"] = np.array([M[0]])
    return out


# ---- advective polynomial block (Stage-3 physics-based cells) -------------------------------------------
def adv_rollout_fwd(h0, A, T):
    ct, suf = _ct(h0.dtype)
    S = h0.shape[1:]
    traj = np.empty((T + 1,) + h0.shape, dtype=h0.dtype)
    traj[0] = h0
    getattr(lib(), "pi_oracle_adv_rollout_fwd_" + suf)(_ptr(traj, ct), _ptr(A, ct), len(S), _shape(S), T)
    return traj


def adv_rollout_bwd(traj, gtraj, A):
    ct, suf = _ct(traj.dtype)
    T = traj.shape[0] - 1
    S = traj.shape[2:]
    traj, gtraj = np.ascontiguousarray(traj), np.ascontiguousarray(gtraj)
    g0 = np.empty(traj.shape[1:], dtype=traj.dtype)
    work = np.empty((2,) + traj.shape[1:], dtype=traj.dtype)
    ag = np.zeros(60, dtype=np.float64)
    getattr(lib(), "pi_oracle_adv_rollout_bwd_" + suf)(_ptr(traj, ct), _ptr(gtraj, ct), _ptr(g0, ct),
                                                       _ptr(ag, ctypes.c_double), _ptr(work, ct), _ptr(A, ct), len(S),
                                                       _shape(S), T)
    return g0, ag


def pack_burgers_stage3(sd: dict, dx: float, dt: float, dtype=np.float64) -> np.ndarray:
    """Advective block of the Stage-3 Burgers cell (bur3:154-157) from its state_dict-like mapping."""
    A = np.zeros(60, dtype=dtype)
    lap = np.asarray(sd["laplace_op.filter.weight"], dtype=np.float64).reshape(5, 5) / dx ** 2
    A[0], A[1], A[2], A[3] = dt, float(sd["nu_u"]), float(sd["nu_v"]), lap[2, 2]
    d0 = np.asarray(sd["dx_op.filter.weight"], dtype=np.float64).reshape(5, 5) / dx      # rows: axis 0
    d1 = np.asarray(sd["dy_op.filter.weight"], dtype=np.float64).reshape(5, 5) / dx      # columns: axis 1
    for i, o in enumerate((-2, -1, 1, 2)):
        A[4 + i], A[8 + i] = lap[2 + o, 2], lap[2, 2 + o]
        A[36 + i], A[40 + i] = d0[2 + o, 2], d1[2, 2 + o]
    A[48 + 0], A[48 + 3] = float(sd["C1_u"]), float(sd["C2_u"])          # u: C1_u*u*D0(u) + C2_u*v*D1(u)
    A[54 + 0], A[54 + 3] = float(sd["C1_v"]), float(sd["C2_v"])          # v: C1_v*u*D0(v) + C2_v*v*D1(v)
    return A


# ---------------------------------------------------------------------------------------------
# Stage-1 Pi-block (5x5 conv branches 2 -> 16; float32).  Block layout: see pi_oracle.c / include/percnn_pi_stage1.h
# ---------------------------------------------------------------------------------------------
S1_NP = 16 + 6 * 16 * 52 + 32 + 2


def s1_pack(sd: dict, dt: float, coef_u: float, coef_v: float) -> np.ndarray:
    """state_dict of the reference's Stage-1 RCNNCell (numpy arrays) -> float32 block;
    coef_* = nu_up * sigmoid(CA / CB) evaluated by the caller in float32 as the reference does (bur1:172-173)."""
    P = np.zeros(S1_NP, dtype=np.float32)
    P[0], P[1], P[2] = dt, coef_u, coef_v
    c, taps = star_taps(np.asarray(sd["W_laplace.weight"], dtype=np.float32))
    P[3] = c
    P[4:8], P[8:12] = taps[0], taps[1]
    for s, sp in enumerate("uv"):
        for k in range(3):
            w = np.asarray(sd[f"Wh{k + 1}_{sp}.weight"], dtype=np.float32).reshape(16, 50)
            b = np.asarray(sd[f"Wh{k + 1}_{sp}.bias"], dtype=np.float32)
            blk = P[16 + (s * 3 + k) * 16 * 52: 16 + (s * 3 + k + 1) * 16 * 52].reshape(16, 52)
            blk[:, :50], blk[:, 50] = w, b
        P[16 + 4992 + s * 16: 16 + 4992 + (s + 1) * 16] = np.asarray(sd[f"Wh4_{sp}.weight"], dtype=np.float32).reshape(16)
        P[16 + 4992 + 32 + s] = np.asarray(sd[f"Wh4_{sp}.bias"], dtype=np.float32).reshape(())
    return P


def s1_step_fwd(h: np.ndarray, P: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.float32)
    out = np.empty_like(h)
    H, W = h.shape[1:]
    lib().pi_oracle_s1_step_fwd_f32(_ptr(h, ctypes.c_float), _ptr(out, ctypes.c_float), _ptr(P, ctypes.c_float),
                                    ctypes.c_long(H), ctypes.c_long(W))
    return out


def s1_rollout_fwd(h0: np.ndarray, P: np.ndarray, T: int) -> np.ndarray:
    H, W = h0.shape[1:]
    traj = np.empty((T + 1,) + h0.shape, dtype=np.float32)
    traj[0] = h0
    lib().pi_oracle_s1_rollout_fwd_f32(_ptr(traj, ctypes.c_float), _ptr(P, ctypes.c_float), ctypes.c_long(H),
                                       ctypes.c_long(W), ctypes.c_int(T))
    return traj
