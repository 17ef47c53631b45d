// pi_contract.h -- factored parameter block -> pre-contracted polynomial block, and its chain rule (gfx950).
//
// The Hadamard product of the three 1x1 branches followed by the 1x1 aggregation (train_2drd.py:115-116) is the cubic
//   r_s(u,v) = sum_m c[s][m] phi_m,   c[s][m] = sum_j w4[s][j] * sum_{abc : e_a e_b e_c == phi_m} L1[j][a] L2[j][b] L3[j][c]
// with e = (u, v, 1), L_k[j] = (Wh_k.weight[j,0], Wh_k.weight[j,1], Wh_k.bias[j]) and Wh4.bias added to c[s][0] -- the
// expansion the reference prints symbolically (train_3drd.py:442-468).  Every training iteration needs it once forward
// and once backward; as stock tensor ops that was ~25 + ~35 tiny launches (0.15 ms + 0.2 ms of host-bound time).
// One single-workgroup launch each here; float64 arithmetic, rounded once to the compute type.
#pragma once
#include "pi_device.h"

namespace pi {

// index of the monomial e_a e_b e_c in phi = {1, u, v, u2, uv, v2, u3, u2v, uv2, v3}
__device__ __forceinline__ int mono_of(int a, int b, int c)
{
    const int nu = (a == 0) + (b == 0) + (c == 0), nv = (a == 1) + (b == 1) + (c == 1);
    const int deg = nu + nv;                              // 0: {0}, 1: {1,2}, 2: {3,4,5}, 3: {6,7,8,9}
    return deg * (deg + 1) / 2 + nv;
}

constexpr int CONTRACT_LDS_HC = 64;                       // widest block staged in LDS (wider ones read global memory)

template <typename T>
__global__ void __launch_bounds__(128) pi_contract_fwd_kernel(const T* __restrict__ Pg, int hc, T* __restrict__ Q)
{
    // the whole block comes in with ONE round trip (the per-channel loop below otherwise pays a cold ~2 us load per
    // hidden channel: 20 us at Hc = 8)
    __shared__ T stage[P_W + 2 * (10 * CONTRACT_LDS_HC + 1)];
    // per (species, channel): its ten monomial sums, computed by ONE thread each in parallel (was: twenty threads walking all
    // channels and all 27 index triples each -- 13 us of dependent float64 arithmetic per launch at Hc = 8, 4 % of a
    // 100^2 x 200-step training iteration); the sums over the channels keep their order, results are bit-identical
    __shared__ double mono[2 * CONTRACT_LDS_HC][10];
    const int t = threadIdx.x;
    const bool staged = hc <= CONTRACT_LDS_HC;
    if (staged) {
        for (int i = t; i < nparams(hc); i += blockDim.x) stage[i] = Pg[i];
        __syncthreads();
    }
    const T* P = staged ? stage : Pg;
    if (t < P_W) Q[t] = P[t];
    auto channel_sums = [&](const T* w, double (&sj)[10]) {
#pragma unroll
        for (int m = 0; m < 10; ++m) sj[m] = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    sj[mono_of(a, b, c)] += ((double)w[a] * (double)w[3 + b]) * (double)w[6 + c];
    };
    if (staged) {
        for (int idx = t; idx < 2 * hc; idx += blockDim.x) {
            const int s = idx / hc, j = idx - s * hc;
            double sj[10];
            channel_sums(P + P_W + s * species_block(hc) + 10 * j, sj);
#pragma unroll
            for (int m = 0; m < 10; ++m) mono[idx][m] = sj[m];
        }
        __syncthreads();
    }
    if (t >= 20) return;
    const int s = t / 10, m = t % 10;
    const T* B = P + P_W + s * species_block(hc);
    double acc = 0.0;
    for (int j = 0; j < hc; ++j) {
        double sjm;
        if (staged) {
            sjm = mono[s * hc + j][m];
        } else {
            double sj[10];
            channel_sums(B + 10 * j, sj);
            sjm = sj[m];
        }
        acc += (double)B[10 * j + 9] * sjm;
    }
    if (m == 0) acc += (double)B[10 * hc];
    Q[P_W + t] = (T)acc;
}

// gQ = dL/dQ (36 entries) -> gP = dL/dP (16 + 2*(10*hc+1) entries); thread = (species, hidden channel)
template <typename T>
__global__ void __launch_bounds__(256) pi_contract_bwd_kernel(const T* __restrict__ P, int hc, const T* __restrict__ gQ,
                                                              T* __restrict__ gP)
{
    const int t = threadIdx.x;
    if (t < P_W) gP[t] = gQ[t];
    if (t < 2) gP[P_W + t * species_block(hc) + 10 * hc] = gQ[P_W + 10 * t];      // Wh4.bias rides on c[s][0]
    for (int idx = t; idx < 2 * hc; idx += blockDim.x) {
        const int s = idx / hc, j = idx - s * hc;
        const T* w = P + P_W + s * species_block(hc) + 10 * j;
        const T* gc = gQ + P_W + 10 * s;
        double L[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int a = 0; a < 3; ++a) L[k][a] = (double)w[3 * k + a];
        const double w4 = (double)w[9];
        double g1[3] = {0, 0, 0}, g2[3] = {0, 0, 0}, g3[3] = {0, 0, 0}, gw4 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double G = (double)gc[mono_of(a, b, c)];
                    gw4 += G * ((L[0][a] * L[1][b]) * L[2][c]);
                    g1[a] += G * (L[1][b] * L[2][c]);
                    g2[b] += G * (L[0][a] * L[2][c]);
                    g3[c] += G * (L[0][a] * L[1][b]);
                }
        T* o = gP + P_W + s * species_block(hc) + 10 * j;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[a] = (T)(w4 * g1[a]);
            o[3 + a] = (T)(w4 * g2[a]);
            o[6 + a] = (T)(w4 * g3[a]);
        }
        o[9] = (T)gw4;
    }
}


// ------------------------------------------------------------------------------------------------
// pack (+ contract) in ONE launch: the reference's parameter tensors -> the packed block.
//
// RCNNCell.param_block() used to assemble the block with stock tensor ops (2 sigmoid, 2 mul, a 23-way cat, an
// index_select, then the contraction kernel): 79 us per call on MI355X in no_grad mode and 540 us forward + backward
// through autograd -- as long as the 200-step rollout of the reference's own 100^2 grid.  Here one single-workgroup kernel
// reads the 19 tensors through a pointer table, builds the factored block in LDS (layout of functional._gather_index) and
// writes it or its contraction; a second one maps dL/d(block) back onto the 18 trainable tensors (contraction chain rule,
// sigmoid derivative).  W_laplace is frozen in the reference (train_2drd.py:65-67): it gets no gradient here.
// ------------------------------------------------------------------------------------------------
struct PackPtrs {
    const void* c[2];       // CA, CB (sigmoid mode: coef = mu_up * sigmoid(C), train_2drd.py:115) or DA, DB (raw, percnn_LO_eqn.py:107)
    const void* w;          // W_laplace.weight: 5^ndim taps
    const void* br[16];     // Wh1_u.weight, Wh1_u.bias, ..., Wh4_u.bias, Wh1_v.weight, ..., Wh4_v.bias
};

// forward value: ATen's own formula in the parameter dtype (1 / (1 + exp(-x)) in T, UnarySpecialOpsKernel.cu), so that the
// one-launch packing gives the same diffusion coefficient as `mu_up * torch.sigmoid(CA)` on the device (the
// float64 evaluation rounded to T could differ from it by one ulp); the derivative below stays in float64
template <typename T>
__device__ __forceinline__ T pack_sigmoid_t(T x) { return T(1) / (T(1) + exp(-x)); }
template <typename T>
__device__ __forceinline__ double pack_sigmoid(T x) { return 1.0 / (1.0 + exp(-(double)x)); }

// factored block of `pp` into `stage` (LDS); all threads of the workgroup take part, ends with a barrier
template <typename T>
__device__ __forceinline__ void pack_stage(const PackPtrs& pp, int hc, int ndim, double dt, double mu_up, int sigmoid, T* stage)
{
    const int t = threadIdx.x;
    if (t == 0) stage[P_DT] = (T)dt;
    if (t < 2) {
        const T x = *static_cast<const T*>(pp.c[t]);
        // the reference rounds the sigmoid to the parameter dtype and multiplies by the Python float in that dtype
        stage[P_COEF + t] = sigmoid ? (T)(pack_sigmoid_t(x) * (T)mu_up) : x;
    }
    if (t >= 32 && t < 32 + 13) {
        const T* w = static_cast<const T*>(pp.w);
        const int k = t - 32;                                  // 0: centre, 1 + 4a + i: centre + offs[i] along axis a
        int pos[3] = {2, 2, 2};
        if (k > 0) {
            const int a = (k - 1) / 4, i = (k - 1) % 4;
            if (a < ndim) pos[a] += (i < 2 ? i - 2 : i - 1);
        }
        int lin = 0;
        for (int d = 0; d < ndim; ++d) lin = lin * 5 + pos[d];
        stage[P_C0 + k] = w[lin];
    }
    for (int idx = t; idx < 2 * hc; idx += blockDim.x) {
        const int s = idx / hc, j = idx - s * hc;
        T* o = stage + P_W + s * species_block(hc) + 10 * j;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T* wk = static_cast<const T*>(pp.br[8 * s + 2 * k]);
            const T* bk = static_cast<const T*>(pp.br[8 * s + 2 * k + 1]);
            o[3 * k + 0] = wk[2 * j]; o[3 * k + 1] = wk[2 * j + 1]; o[3 * k + 2] = bk[j];
        }
        o[9] = static_cast<const T*>(pp.br[8 * s + 6])[j];
        if (j == 0) stage[P_W + s * species_block(hc) + 10 * hc] = *static_cast<const T*>(pp.br[8 * s + 7]);
    }
    __syncthreads();
}

// Conditioning guard of the pre-contracted form (RCNNCell docstring "WHEN 'factored' IS REQUIRED"): the expanded cubic cancels
// its monomials only to rounding, so its per-step noise relative to the state is eps * A,
//   A = |dt| * max_s sum_m |c[s][m]| * phi_m(u_max, v_max) / max(u_max, v_max).
// The pack launch computes A from the coefficients it forms anyway and leaves {A, seq} in a HOST-MAPPED slot: the module reads
// it without synchronising (one training iteration late) and packs the factored block instead while A is above its bound.
struct PackGuard {
    double u_max, v_max;   // state bound the amplification is priced at
    double* slot;          // host-mapped {A, seq}; null: no guard
    double seq;            // written AFTER A (release): the host knows which pack a value belongs to
};

template <typename T>
__global__ void __launch_bounds__(128) pi_pack_fwd_kernel(PackPtrs pp, int hc, int ndim, double dt, double mu_up, int sigmoid,
                                                          int contract, T* __restrict__ out, PackGuard guard)
{
    __shared__ T stage[P_W + 2 * (10 * CONTRACT_LDS_HC + 1)];
    __shared__ double mono[2 * CONTRACT_LDS_HC][10];
    __shared__ double amp[20];
    const int t = threadIdx.x;
    pack_stage<T>(pp, hc, ndim, dt, mu_up, sigmoid, stage);
    if (!contract) {
        for (int i = t; i < nparams(hc); i += blockDim.x) out[i] = stage[i];
        if (!guard.slot) return;                          // (the factored block also prices its contraction while guarded)
    }
    // same arithmetic and order as pi_contract_fwd_kernel (bit-identical blocks)
    if (contract && t < P_W) out[t] = stage[t];
    for (int idx = t; idx < 2 * hc; idx += blockDim.x) {
        const int s = idx / hc, j = idx - s * hc;
        const T* w = stage + P_W + s * species_block(hc) + 10 * j;
        double sj[10];
#pragma unroll
        for (int m = 0; m < 10; ++m) sj[m] = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    sj[mono_of(a, b, c)] += ((double)w[a] * (double)w[3 + b]) * (double)w[6 + c];
#pragma unroll
        for (int m = 0; m < 10; ++m) mono[idx][m] = sj[m];
    }
    __syncthreads();
    if (t < 20) {
        const int s = t / 10, m = t % 10;
        const T* B = stage + P_W + s * species_block(hc);
        double acc = 0.0;
        for (int j = 0; j < hc; ++j) acc += (double)B[10 * j + 9] * mono[s * hc + j][m];
        if (m == 0) acc += (double)B[10 * hc];
        if (contract) out[P_W + t] = (T)acc;
        if (guard.slot) {
            const double u = guard.u_max, v = guard.v_max;
            const double phi[10] = {1.0, u, v, u * u, u * v, v * v, u * u * u, u * u * v, u * v * v, v * v * v};
            amp[t] = fabs((double)(T)acc) * phi[m];       // the ROUNDED coefficient is what the kernels evaluate
        }
    }
    if (!guard.slot) return;
    __syncthreads();
    if (t == 0) {
        double a[2] = {0.0, 0.0};
        for (int s = 0; s < 2; ++s)
            for (int m = 0; m < 10; ++m) a[s] += amp[10 * s + m];
        const double bound = guard.u_max > guard.v_max ? guard.u_max : guard.v_max;
        const double A = fabs(dt) * (a[0] > a[1] ? a[0] : a[1]) / (bound > 0.0 ? bound : 1.0);
        __hip_atomic_store(guard.slot, A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(guard.slot + 1, guard.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// g_block = dL/d(block) (contracted: 36 entries, else nparams(hc)) -> gradients of the 18 trainable tensors (`gp`, same slots
// as `pp`; gp.w is ignored; NULL slots are skipped)
template <typename T>
__global__ void __launch_bounds__(128) pi_pack_bwd_kernel(PackPtrs pp, PackPtrs gp, int hc, int ndim, double dt, double mu_up,
                                                          int sigmoid, int contract, const T* __restrict__ g_block)
{
    __shared__ T stage[P_W + 2 * (10 * CONTRACT_LDS_HC + 1)];
    const int t = threadIdx.x;
    pack_stage<T>(pp, hc, ndim, dt, mu_up, sigmoid, stage);
    if (t < 2 && gp.c[t]) {
        const double g = (double)g_block[P_COEF + t];
        double d = g;
        if (sigmoid) {
            const double sg = pack_sigmoid(*static_cast<const T*>(pp.c[t]));
            d = g * mu_up * sg * (1.0 - sg);
        }
        *static_cast<T*>(const_cast<void*>(gp.c[t])) = (T)d;
    }
    for (int idx = t; idx < 2 * hc; idx += blockDim.x) {
        const int s = idx / hc, j = idx - s * hc;
        const T* w = stage + P_W + s * species_block(hc) + 10 * j;
        T go[10];
        if (contract) {
            // chain rule of the contraction, as pi_contract_bwd_kernel
            const T* gc = g_block + P_W + 10 * s;
            double L[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int a = 0; a < 3; ++a) L[k][a] = (double)w[3 * k + a];
            const double w4 = (double)w[9];
            double g1[3] = {0, 0, 0}, g2[3] = {0, 0, 0}, g3[3] = {0, 0, 0}, gw4 = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double G = (double)gc[mono_of(a, b, c)];
                        gw4 += G * ((L[0][a] * L[1][b]) * L[2][c]);
                        g1[a] += G * (L[1][b] * L[2][c]);
                        g2[b] += G * (L[0][a] * L[2][c]);
                        g3[c] += G * (L[0][a] * L[1][b]);
                    }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                go[a] = (T)(w4 * g1[a]); go[3 + a] = (T)(w4 * g2[a]); go[6 + a] = (T)(w4 * g3[a]);
            }
            go[9] = (T)gw4;
        } else {
            const T* g = g_block + P_W + s * species_block(hc) + 10 * j;
#pragma unroll
            for (int i = 0; i < 10; ++i) go[i] = g[i];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T* gw = static_cast<T*>(const_cast<void*>(gp.br[8 * s + 2 * k]));
            T* gb = static_cast<T*>(const_cast<void*>(gp.br[8 * s + 2 * k + 1]));
            if (gw) { gw[2 * j] = go[3 * k]; gw[2 * j + 1] = go[3 * k + 1]; }
            if (gb) gb[j] = go[3 * k + 2];
        }
        if (T* g4 = static_cast<T*>(const_cast<void*>(gp.br[8 * s + 6]))) g4[j] = go[9];
        if (j == 0)
            if (T* gb4 = static_cast<T*>(const_cast<void*>(gp.br[8 * s + 7])))
                *gb4 = contract ? g_block[P_W + 10 * s] : g_block[P_W + s * species_block(hc) + 10 * hc];   // Wh4.bias rides on c[s][0]
    }
}

}  // namespace pi
