/* Plain-C restatement of one Pi-block step and its adjoint.  TEST INFRASTRUCTURE ONLY.
 *
 * Included twice by pi_oracle.c with REAL = float / double and SUF = f32 / f64.
 *
 * What it restates (paths relative to /root/reference):
 *   periodic pad           DataDrivenModeling/2d_gs_rd/train_2drd.py:108-109, 3d_gs_rd/train_3drd.py:125-127
 *   Laplacian conv         train_2drd.py:115-116 (W_laplace, weights pre-scaled by 1/dx^2 at :66)
 *   1x1 branches + product train_2drd.py:115-116 (Wh1*Wh2*Wh3 -> Wh4)
 *   Euler update           train_2drd.py:117-119;  lambda-omega twin percnn_LO_eqn.py:98-112
 *   backward               implicit autograd of the above (train_2drd.py:407)
 *
 * Parameter block P (array of REAL) -- the same layout the product's C-ABI documents in
 * include/percnn_pi.h (restated here independently):
 *   P[0]=dt  P[1]=coef_u  P[2]=coef_v  P[3]=centre tap
 *   P[4+4*ax+i]: tap of spatial axis ax (0 = slowest) at offset {-2,-1,+1,+2}[i]   (12 slots)
 *   P[16 + s*(10*hc+1) + 10*j + {0..9}] = {w1u,w1v,b1, w2u,w2v,b2, w3u,w3v,b3, w4} of hidden j, species s
 *   P[16 + s*(10*hc+1) + 10*hc]         = b4 of species s
 * Gradient block pg (double) uses the same indexing (slots 0 and 3..15 stay untouched).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

static inline long FN(wrap_)(long i, long n) { i %= n; return i < 0 ? i + n : i; }

/* star stencil applied at point (z,y,x); flip = +1 forward (cross-correlation), -1 adjoint */
static REAL FN(star_)(const REAL *f, const REAL *P, int ndim, const long *S, const long *idx, int flip)
{
    static const int offs[4] = {-2, -1, 1, 2};
    long str[3], lin = 0;
    long acc = 1;
    for (int a = ndim - 1; a >= 0; --a) { str[a] = acc; acc *= S[a]; }
    for (int a = 0; a < ndim; ++a) lin += idx[a] * str[a];
    REAL lap = P[3] * f[lin];
    for (int a = 0; a < ndim; ++a)
        for (int i = 0; i < 4; ++i) {
            long j = FN(wrap_)(idx[a] + flip * offs[i], S[a]);
            lap = FMA(P[4 + 4 * a + i], f[lin + (j - idx[a]) * str[a]], lap);
        }
    return lap;
}

/* planes [lo0, hi0) of axis 0 are computed, the rest of `out` is left untouched (slab tests) */
void FN(pi_oracle_step_fwd_range_)(const REAL *h, REAL *out, const REAL *P, int hc, int ndim, const long *S,
                                   long lo0, long hi0)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    const REAL dt = P[0];
    long idx[3] = {0, 0, 0};
    for (long p = 0; p < n; ++p) {
        long r = p;
        for (int a = ndim - 1; a >= 0; --a) { idx[a] = r % S[a]; r /= S[a]; }
        if (idx[0] < lo0 || idx[0] >= hi0) continue;
        const REAL u = h[p], v = h[n + p];
        for (int s = 0; s < 2; ++s) {
            const REAL *W = P + 16 + s * (10 * hc + 1);
            REAL lap = FN(star_)(h + s * n, P, ndim, S, idx, +1);
            REAL rr = W[10 * hc];
            for (int j = 0; j < hc; ++j) {
                const REAL *w = W + 10 * j;
                REAL a1 = FMA(w[0], u, FMA(w[1], v, w[2]));
                REAL a2 = FMA(w[3], u, FMA(w[4], v, w[5]));
                REAL a3 = FMA(w[6], u, FMA(w[7], v, w[8]));
                rr = FMA(w[9], (a1 * a2) * a3, rr);
            }
            REAL res = P[1 + s] * lap + rr;
            REAL t = res * dt;
            out[s * n + p] = h[s * n + p] + t;
        }
    }
}

void FN(pi_oracle_step_fwd_)(const REAL *h, REAL *out, const REAL *P, int hc, int ndim, const long *S)
{
    FN(pi_oracle_step_fwd_range_)(h, out, P, hc, ndim, S, 0, S[0]);
}

/* Adjoint of one step.  G = dL/d(next state) [2][n]; inj (nullable) = dL/d(out_{t-1}) added at the end;
 * Gprev = dL/d(prev state); pg accumulates parameter gradients (double).
 * Only planes [lo0, hi0) of axis 0 are computed / accumulated (slab tests). */
void FN(pi_oracle_step_bwd_range_)(const REAL *h, const REAL *G, const REAL *inj, REAL *Gprev, double *pg,
                                   const REAL *P, int hc, int ndim, const long *S, long lo0, long hi0)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    const REAL dt = P[0];
    long idx[3] = {0, 0, 0};
    for (long p = 0; p < n; ++p) {
        long r = p;
        for (int a = ndim - 1; a >= 0; --a) { idx[a] = r % S[a]; r /= S[a]; }
        if (idx[0] < lo0 || idx[0] >= hi0) continue;
        const REAL u = h[p], v = h[n + p];
        REAL du = 0, dv = 0;
        REAL dl[2];
        for (int s = 0; s < 2; ++s) {
            const REAL *W = P + 16 + s * (10 * hc + 1);
            double *gW = pg + 16 + s * (10 * hc + 1);
            REAL lg = FN(star_)(G + s * n, P, ndim, S, idx, -1);
            dl[s] = lg * dt;
            pg[1 + s] += (double)(dl[s] * h[s * n + p]);
            const REAL gr = G[s * n + p] * dt;
            gW[10 * hc] += (double)gr;
            for (int j = 0; j < hc; ++j) {
                const REAL *w = W + 10 * j;
                double *g = gW + 10 * j;
                REAL a1 = FMA(w[0], u, FMA(w[1], v, w[2]));
                REAL a2 = FMA(w[3], u, FMA(w[4], v, w[5]));
                REAL a3 = FMA(w[6], u, FMA(w[7], v, w[8]));
                REAL p12 = a1 * a2;
                REAL gw = gr * w[9];
                REAL q1 = gw * (a2 * a3), q2 = gw * (a1 * a3), q3 = gw * p12;
                g[9] += (double)(gr * (p12 * a3));
                g[0] += (double)(q1 * u); g[1] += (double)(q1 * v); g[2] += (double)q1;
                g[3] += (double)(q2 * u); g[4] += (double)(q2 * v); g[5] += (double)q2;
                g[6] += (double)(q3 * u); g[7] += (double)(q3 * v); g[8] += (double)q3;
                du = FMA(q1, w[0], FMA(q2, w[3], FMA(q3, w[6], du)));
                dv = FMA(q1, w[1], FMA(q2, w[4], FMA(q3, w[7], dv)));
            }
        }
        REAL tu = P[1] * dl[0] + du;
        REAL tv = P[2] * dl[1] + dv;
        REAL gu = G[p] + tu, gv = G[n + p] + tv;
        if (inj) { gu += inj[p]; gv += inj[n + p]; }
        Gprev[p] = gu;
        Gprev[n + p] = gv;
    }
}

void FN(pi_oracle_step_bwd_)(const REAL *h, const REAL *G, const REAL *inj, REAL *Gprev, double *pg,
                             const REAL *P, int hc, int ndim, const long *S)
{
    FN(pi_oracle_step_bwd_range_)(h, G, inj, Gprev, pg, P, hc, ndim, S, 0, S[0]);
}

/* traj: [T+1][2][n], frame 0 filled by the caller (2dgs:162-190 rollout loop) */
void FN(pi_oracle_rollout_fwd_)(REAL *traj, const REAL *P, int hc, int ndim, const long *S, int T)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    for (int t = 0; t < T; ++t)
        FN(pi_oracle_step_fwd_)(traj + (long)t * 2 * n, traj + (long)(t + 1) * 2 * n, P, hc, ndim, S);
}

/* gtraj: dL/dtraj [T+1][2][n]; g0 out = dL/dh0 [2][n]; work: 2 x [2][n] scratch */
void FN(pi_oracle_rollout_bwd_)(const REAL *traj, const REAL *gtraj, REAL *g0, double *pg, REAL *work,
                                const REAL *P, int hc, int ndim, const long *S, int T)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    REAL *A = work, *B = work + 2 * n;
    for (long i = 0; i < 2 * n; ++i) A[i] = gtraj[(long)T * 2 * n + i];
    for (int t = T; t >= 1; --t) {
        REAL *dst = (t == 1) ? g0 : B;
        FN(pi_oracle_step_bwd_)(traj + (long)(t - 1) * 2 * n, A, gtraj + (long)(t - 1) * 2 * n, dst, pg,
                                P, hc, ndim, S);
        if (t > 1) { REAL *tmp = A; A = B; B = tmp; }
    }
    if (T == 0) for (long i = 0; i < 2 * n; ++i) g0[i] = A[i];
}

/* ---- pre-contracted ("poly") reaction --------------------------------------------------------
 * The Hadamard product of the three 1x1 branches followed by the 1x1 aggregation
 * (train_2drd.py:115-116) is, per species, a cubic polynomial in (u, v):
 *     r_s(u,v) = sum_m c[s][m] * phi_m(u,v),  phi = {1,u,v,u^2,uv,v^2,u^3,u^2 v,u v^2,v^3}
 * (the reference itself prints this expansion: train_3drd.py:442-468 get_expression).
 * Block Q: Q[0..15] as P[0..15]; Q[16 + 10*s + m] = c[s][m].  Gradient block: same indexing,
 * qg[16+10*s+m] = dL/dc[s][m] = sum_x g_s(x) dt phi_m(x)  (the "moments"). */
static inline REAL FN(poly_r_)(const REAL *c, REAL u, REAL v)
{
    REAL A0 = FMA(v, FMA(v, FMA(v, c[9], c[5]), c[2]), c[0]);
    REAL A1 = FMA(v, FMA(v, c[8], c[4]), c[1]);
    REAL A2 = FMA(v, c[7], c[3]);
    return FMA(u, FMA(u, FMA(u, c[6], A2), A1), A0);
}

void FN(pi_oracle_poly_step_fwd_range_)(const REAL *h, REAL *out, const REAL *Q, int ndim, const long *S,
                                        long lo0, long hi0)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    const REAL dt = Q[0];
    long idx[3] = {0, 0, 0};
    for (long p = 0; p < n; ++p) {
        long r = p;
        for (int a = ndim - 1; a >= 0; --a) { idx[a] = r % S[a]; r /= S[a]; }
        if (idx[0] < lo0 || idx[0] >= hi0) continue;
        const REAL u = h[p], v = h[n + p];
        for (int s = 0; s < 2; ++s) {
            REAL lap = FN(star_)(h + s * n, Q, ndim, S, idx, +1);
            REAL rr = FN(poly_r_)(Q + 16 + 10 * s, u, v);
            REAL res = Q[1 + s] * lap + rr;
            REAL t = res * dt;
            out[s * n + p] = h[s * n + p] + t;
        }
    }
}

void FN(pi_oracle_poly_step_bwd_range_)(const REAL *h, const REAL *G, const REAL *inj, REAL *Gprev, double *qg,
                                        const REAL *Q, int ndim, const long *S, long lo0, long hi0)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    const REAL dt = Q[0];
    long idx[3] = {0, 0, 0};
    for (long p = 0; p < n; ++p) {
        long r = p;
        for (int a = ndim - 1; a >= 0; --a) { idx[a] = r % S[a]; r /= S[a]; }
        if (idx[0] < lo0 || idx[0] >= hi0) continue;
        const REAL u = h[p], v = h[n + p];
        const REAL u2 = u * u, uv = u * v, v2 = v * v;
        const REAL phi[10] = {1, u, v, u2, uv, v2, u2 * u, u2 * v, u * v2, v2 * v};
        REAL du = 0, dv = 0, dl[2];
        for (int s = 0; s < 2; ++s) {
            const REAL *c = Q + 16 + 10 * s;
            REAL lg = FN(star_)(G + s * n, Q, ndim, S, idx, -1);
            dl[s] = lg * dt;
            qg[1 + s] += (double)(dl[s] * h[s * n + p]);
            const REAL gr = G[s * n + p] * dt;
            qg[16 + 10 * s] += (double)gr;
            for (int m = 1; m < 10; ++m) qg[16 + 10 * s + m] += (double)(gr * phi[m]);
            /* dr/du = A1 + u (2 A2 + 3 c6 u);  dr/dv = B0 + u (B1 + c7 u) */
            REAL A1 = FMA(v, FMA(v, c[8], c[4]), c[1]);
            REAL A2x2 = FMA(v, 2 * c[7], 2 * c[3]);
            REAL ru = FMA(u, FMA(u, 3 * c[6], A2x2), A1);
            REAL B0 = FMA(v, FMA(v, 3 * c[9], 2 * c[5]), c[2]);
            REAL B1 = FMA(v, 2 * c[8], c[4]);
            REAL rv = FMA(u, FMA(u, c[7], B1), B0);
            du = FMA(gr, ru, du);
            dv = FMA(gr, rv, dv);
        }
        REAL tu = Q[1] * dl[0] + du;
        REAL tv = Q[2] * dl[1] + dv;
        REAL gu = G[p] + tu, gv = G[n + p] + tv;
        if (inj) { gu += inj[p]; gv += inj[n + p]; }
        Gprev[p] = gu;
        Gprev[n + p] = gv;
    }
}

void FN(pi_oracle_poly_rollout_fwd_)(REAL *traj, const REAL *Q, int ndim, const long *S, int T)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    for (int t = 0; t < T; ++t)
        FN(pi_oracle_poly_step_fwd_range_)(traj + (long)t * 2 * n, traj + (long)(t + 1) * 2 * n, Q, ndim, S, 0, S[0]);
}

void FN(pi_oracle_poly_rollout_bwd_)(const REAL *traj, const REAL *gtraj, REAL *g0, double *qg, REAL *work,
                                     const REAL *Q, int ndim, const long *S, int T)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    REAL *A = work, *B = work + 2 * n;
    for (long i = 0; i < 2 * n; ++i) A[i] = gtraj[(long)T * 2 * n + i];
    for (int t = T; t >= 1; --t) {
        REAL *dst = (t == 1) ? g0 : B;
        FN(pi_oracle_poly_step_bwd_range_)(traj + (long)(t - 1) * 2 * n, A, gtraj + (long)(t - 1) * 2 * n, dst, qg,
                                           Q, ndim, S, 0, S[0]);
        if (t > 1) { REAL *tmp = A; A = B; B = tmp; }
    }
    if (T == 0) for (long i = 0; i < 2 * n; ++i) g0[i] = A[i];
}

/* ---- advective polynomial block ("adv", 60 entries) -- Stage-3 physics-based cells --------
 * A[0..35] as the poly block Q; A[36 + 4*a + i] = first-derivative tap of axis a at offset {-2,-1,+1,+2}[i];
 * A[48 + 6*s + 2*a + {0,1}] = coefficients (cu, cv) of the advective term (cu*u + cv*v) * D_a(h_s) in species s:
 *   rhs_s = coef_s Lap(h_s) + r_s(u,v) + sum_a (cu_{s,a} u + cv_{s,a} v) D_a(h_s);   next = h + dt*rhs
 * Reference: Burgers Stage-3 f_rhs, DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-3/fine_tuning_[5%noise,41x51x51].py:154-157
 * (dx_2d_op differentiates along tensor dim 2 = axis 0, dy_2d_op along axis 1; :20-30). */
static REAL FN(nbr_)(const REAL *f, int ndim, const long *S, const long *idx, int a, int off)
{
    long str = 1, lin = 0, acc = 1;
    for (int b = ndim - 1; b >= 0; --b) { if (b == a) str = acc; lin += idx[b] * acc; acc *= S[b]; }
    long j = FN(wrap_)(idx[a] + off, S[a]);
    return f[lin + (j - idx[a]) * str];
}

void FN(pi_oracle_adv_step_fwd_)(const REAL *h, REAL *out, const REAL *A, int ndim, const long *S)
{
    static const int offs[4] = {-2, -1, 1, 2};
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    const REAL dt = A[0];
    long idx[3] = {0, 0, 0};
    for (long p = 0; p < n; ++p) {
        long r = p;
        for (int a = ndim - 1; a >= 0; --a) { idx[a] = r % S[a]; r /= S[a]; }
        const REAL u = h[p], v = h[n + p];
        for (int s = 0; s < 2; ++s) {
            REAL lap = FN(star_)(h + s * n, A, ndim, S, idx, +1);
            REAL rr = FN(poly_r_)(A + 16 + 10 * s, u, v);
            REAL adv = 0;
            for (int a = 0; a < ndim; ++a) {
                REAL d = A[36 + 4 * a] * FN(nbr_)(h + s * n, ndim, S, idx, a, offs[0]);
                for (int i = 1; i < 4; ++i) d = FMA(A[36 + 4 * a + i], FN(nbr_)(h + s * n, ndim, S, idx, a, offs[i]), d);
                REAL c = FMA(A[48 + 6 * s + 2 * a], u, A[48 + 6 * s + 2 * a + 1] * v);
                adv = FMA(c, d, adv);
            }
            REAL res = A[1 + s] * lap + (rr + adv);
            REAL t = res * dt;
            out[s * n + p] = h[s * n + p] + t;
        }
    }
}

/* ag: double[60] accumulated gradient block (slots 1,2 coef; 16..35 moments; 48..59 advection coefficients) */
void FN(pi_oracle_adv_step_bwd_)(const REAL *h, const REAL *G, const REAL *inj, REAL *Gprev, double *ag,
                                 const REAL *A, int ndim, const long *S)
{
    static const int offs[4] = {-2, -1, 1, 2};
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    const REAL dt = A[0];
    long idx[3] = {0, 0, 0};
    for (long p = 0; p < n; ++p) {
        long r = p;
        for (int a = ndim - 1; a >= 0; --a) { idx[a] = r % S[a]; r /= S[a]; }
        const REAL u = h[p], v = h[n + p];
        const REAL u2 = u * u, uv = u * v, v2 = v * v;
        const REAL phi[10] = {1, u, v, u2, uv, v2, u2 * u, u2 * v, u * v2, v2 * v};
        REAL dsp[2] = {0, 0};             /* pointwise contributions to (du, dv) */
        REAL advT[2] = {0, 0};            /* transposed first-derivative stencils, per species */
        REAL dl[2];
        for (int s = 0; s < 2; ++s) {
            const REAL *c = A + 16 + 10 * s;
            REAL lg = FN(star_)(G + s * n, A, ndim, S, idx, -1);
            dl[s] = lg * dt;
            ag[1 + s] += (double)(dl[s] * h[s * n + p]);
            const REAL gr = G[s * n + p] * dt;
            ag[16 + 10 * s] += (double)gr;
            for (int m = 1; m < 10; ++m) ag[16 + 10 * s + m] += (double)(gr * phi[m]);
            REAL A1 = FMA(v, FMA(v, c[8], c[4]), c[1]);
            REAL A2x2 = FMA(v, 2 * c[7], 2 * c[3]);
            REAL ru = FMA(u, FMA(u, 3 * c[6], A2x2), A1);
            REAL B0 = FMA(v, FMA(v, 3 * c[9], 2 * c[5]), c[2]);
            REAL B1 = FMA(v, 2 * c[8], c[4]);
            REAL rv = FMA(u, FMA(u, c[7], B1), B0);
            dsp[0] = FMA(gr, ru, dsp[0]);
            dsp[1] = FMA(gr, rv, dsp[1]);
            for (int a = 0; a < ndim; ++a) {
                const REAL cu = A[48 + 6 * s + 2 * a], cv = A[48 + 6 * s + 2 * a + 1];
                REAL d = A[36 + 4 * a] * FN(nbr_)(h + s * n, ndim, S, idx, a, offs[0]);
                for (int i = 1; i < 4; ++i) d = FMA(A[36 + 4 * a + i], FN(nbr_)(h + s * n, ndim, S, idx, a, offs[i]), d);
                const REAL gd = gr * d;
                ag[48 + 6 * s + 2 * a] += (double)(gd * u);
                ag[48 + 6 * s + 2 * a + 1] += (double)(gd * v);
                dsp[0] = FMA(gd, cu, dsp[0]);
                dsp[1] = FMA(gd, cv, dsp[1]);
                /* transposed stencil: sum_i tap_i * m(x - offs_i e_a),  m = (cu u + cv v) * G_s * dt */
                for (int i = 0; i < 4; ++i) {
                    REAL un = FN(nbr_)(h, ndim, S, idx, a, -offs[i]), vn = FN(nbr_)(h + n, ndim, S, idx, a, -offs[i]);
                    REAL gn = FN(nbr_)(G + s * n, ndim, S, idx, a, -offs[i]);
                    REAL m = FMA(cu, un, cv * vn) * (gn * dt);
                    advT[s] = FMA(A[36 + 4 * a + i], m, advT[s]);
                }
            }
        }
        for (int s = 0; s < 2; ++s) {
            REAL t = A[1 + s] * dl[s] + (dsp[s] + advT[s]);
            REAL g = G[s * n + p] + t;
            if (inj) g += inj[s * n + p];
            Gprev[s * n + p] = g;
        }
    }
}

void FN(pi_oracle_adv_rollout_fwd_)(REAL *traj, const REAL *A, int ndim, const long *S, int T)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    for (int t = 0; t < T; ++t)
        FN(pi_oracle_adv_step_fwd_)(traj + (long)t * 2 * n, traj + (long)(t + 1) * 2 * n, A, ndim, S);
}

void FN(pi_oracle_adv_rollout_bwd_)(const REAL *traj, const REAL *gtraj, REAL *g0, double *ag, REAL *work,
                                    const REAL *A, int ndim, const long *S, int T)
{
    long n = 1;
    for (int a = 0; a < ndim; ++a) n *= S[a];
    REAL *X = work, *Y = work + 2 * n;
    for (long i = 0; i < 2 * n; ++i) X[i] = gtraj[(long)T * 2 * n + i];
    for (int t = T; t >= 1; --t) {
        REAL *dst = (t == 1) ? g0 : Y;
        FN(pi_oracle_adv_step_bwd_)(traj + (long)(t - 1) * 2 * n, X, gtraj + (long)(t - 1) * 2 * n, dst, ag, A, ndim, S);
        if (t > 1) { REAL *tmp = X; X = Y; Y = tmp; }
    }
    if (T == 0) for (long i = 0; i < 2 * n; ++i) g0[i] = X[i];
}

#undef FN
#undef CAT
#undef CAT_
