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

}  // namespace pi
