// mfma_branch_ubench.hip -- can the matrix cores evaluate the Hc = 8 branch product of the Pi-block faster than the vector ALU?
// (VERDICT r3 next #3 / BASELINE north_star: "MFMA only if the 1x1 branch product is recast as a small dense contraction".)
//
// Per point and species the reference evaluates  r = b4 + sum_j w4[j] * a1[j] * a2[j] * a3[j],  a_k[j] = wu*u + wv*v + b  (2dgs:115-116):
// 72 fused operations per species, 144 per point.  Recast as a contraction: rows = (species, j) = 16, K = (u, v, 1, 0), one
// v_mfma_f32_16x16x4_f32 per branch k and 16 points; the Hadamard product is in-lane (the three D tiles share their layout), the
// aggregation is 4 FMAs per lane + one cross-lane add.  The kernels below keep everything in registers (no memory traffic) and
// give the matrix-core flavour the most favourable layout possible: its B operand (u, v, 1, 0 per 16 points) is ASSUMED to sit in
// the right lanes already and its result is not moved back to the stencil's strip layout -- both would cost VALU / LDS work in
// the real kernel.
//   valu      packed fp32 (v_pk_fma_f32), the tile kernels' factored body
//   mfma      3 x v_mfma_f32_16x16x4_f32 + products + aggregation per 16 points
//   both      the two in ONE wave on independent data: do the pipes overlap (time ~ max) or serialise (time ~ sum)?
// Build / run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o ubench tools/mfma_branch_ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int HC = 8, REPS = 256;

// ---- VALU: each lane owns 4 points (two 2-vectors), both species; the loop structure of the tile kernels' factored body
// (pi_tile2d.h, fwd_substep: species / hidden-channel loops ROLLED, the next channel's 10 wave-uniform scalars prefetched into
// SGPRs -- fully unrolled the 162 scalars do not fit the SGPR file and the compiler parks them in VGPR lanes: 653 v_readlane +
// 190 s_nop per call in a first version of this benchmark, three times slower) ----
struct W10 { float w[10]; };
__device__ __forceinline__ W10 load_w10(const float* __restrict__ p)
{
    W10 r;
#pragma unroll
    for (int i = 0; i < 10; ++i) r.w[i] = p[i];
    return r;
}
__device__ __forceinline__ void valu_body(const float* __restrict__ W, v2f (&u)[2], v2f (&v)[2])
{
    v2f r[2][2];
#pragma clang loop unroll(disable)
    for (int s = 0; s < 2; ++s) {
        const float* w = W + s * (10 * HC + 1);
        v2f rr[2] = {v2f{w[10 * HC], w[10 * HC]}, v2f{w[10 * HC], w[10 * HC]}};
        W10 nx = load_w10(w);
#pragma clang loop unroll(disable)
        for (int j = 0; j < HC; ++j) {
            const W10 c = nx;
            if (j + 1 < HC) nx = load_w10(w + 10 * (j + 1));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v2f a1 = __builtin_elementwise_fma(v2f{c.w[0], c.w[0]}, u[h], __builtin_elementwise_fma(v2f{c.w[1], c.w[1]}, v[h], v2f{c.w[2], c.w[2]}));
                const v2f a2 = __builtin_elementwise_fma(v2f{c.w[3], c.w[3]}, u[h], __builtin_elementwise_fma(v2f{c.w[4], c.w[4]}, v[h], v2f{c.w[5], c.w[5]}));
                const v2f a3 = __builtin_elementwise_fma(v2f{c.w[6], c.w[6]}, u[h], __builtin_elementwise_fma(v2f{c.w[7], c.w[7]}, v[h], v2f{c.w[8], c.w[8]}));
                rr[h] = __builtin_elementwise_fma(v2f{c.w[9], c.w[9]}, (a1 * a2) * a3, rr[h]);
            }
        }
        r[s][0] = rr[0]; r[s][1] = rr[1];
        if (s == 0) { asm volatile("" : "+v"(r[0][0]), "+v"(r[0][1])); }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // feed the result back so that nothing is hoisted out of the loop
        u[h] = __builtin_elementwise_fma(r[0][h], v2f{1e-7f, 1e-7f}, u[h]);
        v[h] = __builtin_elementwise_fma(r[1][h], v2f{1e-7f, 1e-7f}, v[h]);
    }
}

// ---- VALU with the weights read from LDS (all lanes the same address: a broadcast read): the scalars arrive in VGPRs, where a
// v_pk_fma_f32 takes any number of them (one SGPR per instruction is the constant-bus limit: the SGPR flavour above moves the
// second scalar of every inner FMA into a VGPR pair first), and the reads can run several channels ahead ----
struct W12 { v4f a, b, c; };       // {w1u,w1v,b1,w2u} {w2v,b2,w3u,w3v} {b3,w4,-,-}
__device__ __forceinline__ W12 lds_w12(const float* p)
{
    const v4f* q = reinterpret_cast<const v4f*>(p);
    return W12{q[0], q[1], q[2]};
}
__device__ __forceinline__ void valu_lds_body(const float* Wl /* LDS: [2][HC][12] + [2] biases */, v2f (&u)[2], v2f (&v)[2])
{
    v2f r[2][2];
#pragma clang loop unroll(disable)
    for (int s = 0; s < 2; ++s) {
        const float* w = Wl + s * (12 * HC);
        const float b4 = Wl[2 * 12 * HC + s];
        v2f rr[2] = {v2f{b4, b4}, v2f{b4, b4}};
        W12 n0 = lds_w12(w), n1 = lds_w12(w + 12);
#pragma clang loop unroll(disable)
        for (int j = 0; j < HC; ++j) {
            const W12 c = n0;
            n0 = n1;
            if (j + 2 < HC) n1 = lds_w12(w + 12 * (j + 2));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v2f a1 = __builtin_elementwise_fma(v2f{c.a[0], c.a[0]}, u[h], __builtin_elementwise_fma(v2f{c.a[1], c.a[1]}, v[h], v2f{c.a[2], c.a[2]}));
                const v2f a2 = __builtin_elementwise_fma(v2f{c.a[3], c.a[3]}, u[h], __builtin_elementwise_fma(v2f{c.b[0], c.b[0]}, v[h], v2f{c.b[1], c.b[1]}));
                const v2f a3 = __builtin_elementwise_fma(v2f{c.b[2], c.b[2]}, u[h], __builtin_elementwise_fma(v2f{c.b[3], c.b[3]}, v[h], v2f{c.c[0], c.c[0]}));
                rr[h] = __builtin_elementwise_fma(v2f{c.c[1], c.c[1]}, (a1 * a2) * a3, rr[h]);
            }
        }
        r[s][0] = rr[0]; r[s][1] = rr[1];
        if (s == 0) { asm volatile("" : "+v"(r[0][0]), "+v"(r[0][1])); }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u[h] = __builtin_elementwise_fma(r[0][h], v2f{1e-7f, 1e-7f}, u[h]);
        v[h] = __builtin_elementwise_fma(r[1][h], v2f{1e-7f, 1e-7f}, v[h]);
    }
}

// ---- MFMA: per 16 points three 16x16x4 products; lane l holds rows 4*(l/16)+i of column l%16 ----
// A[k]: lane l supplies A[row = l%16][kk = l/16] of branch k;  B: lane l supplies B[kk = l/16][n = l%16] = (u, v, 1, 0)[kk] of point n
__device__ __forceinline__ float mfma_body(const float (&A)[3], const float (&w4)[4], float bias, float b)
{
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    const v4f d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0], b, z, 0, 0, 0);
    const v4f d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[1], b, z, 0, 0, 0);
    const v4f d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2], b, z, 0, 0, 0);
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) part = fmaf(w4[i], (d1[i] * d2[i]) * d3[i], part);
    part += __shfl_xor(part, 16);                           // rows 0-7 = species u (lane groups 0, 1), rows 8-15 = species v (2, 3)
    return part + bias;
}

template <int MODE>   // 0 valu, 1 mfma, 2 both, 3 valu with the weights in LDS
__global__ void __launch_bounds__(256) k_branch(const float* __restrict__ W, float* __restrict__ out, int reps)
{
    __shared__ __attribute__((aligned(16))) float Wl[2 * 12 * HC + 4];
    if (MODE == 3) {
        for (int i = threadIdx.x; i < 2 * HC * 10; i += blockDim.x) {
            const int sp = i / (10 * HC), j = (i % (10 * HC)) / 10, k = i % 10;
            Wl[sp * 12 * HC + 12 * j + k] = W[sp * (10 * HC + 1) + 10 * j + k];
        }
        if (threadIdx.x < 2) Wl[2 * 12 * HC + threadIdx.x] = W[threadIdx.x * (10 * HC + 1) + 10 * HC];
        __syncthreads();
    }
    const int lane = threadIdx.x % 64;
    v2f u[2] = {{0.3f + 1e-3f * lane, 0.31f}, {0.32f, 0.33f}}, v[2] = {{0.2f, 0.21f + 1e-3f * lane}, {0.22f, 0.23f}};
    // MFMA operands of this lane: 3 A values (one per branch), 4 aggregation weights, and the B value of each of the 16 groups of
    // 16 points a wave's 256 points form -- here 4 groups per iteration are live (register budget), 4 iterations per "strip"
    float A[3], w4[4];
    const int row = lane % 16, kk = lane / 16;
    for (int k = 0; k < 3; ++k) A[k] = kk < 3 ? W[(row / 8) * (10 * HC + 1) + 10 * (row % 8) + 3 * k + kk] : 0.f;
    for (int i = 0; i < 4; ++i) { const int r = 4 * kk + i; w4[i] = W[(r / 8) * (10 * HC + 1) + 10 * (r % 8) + 9]; }
    const float bias = W[10 * HC];
    float bv[4];
    for (int g = 0; g < 4; ++g) bv[g] = kk == 0 ? 0.3f + 0.01f * (row + g) : (kk == 1 ? 0.2f + 0.01f * row : (kk == 2 ? 1.f : 0.f));
    float acc = 0.f;
    for (int it = 0; it < reps; ++it) {
        if (MODE == 0 || MODE == 2) valu_body(W, u, v);      // 256 points per wave and call
        if (MODE == 3) valu_lds_body(Wl, u, v);
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q)                      // 16 groups of 16 points = the same 256 points per wave
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float r = mfma_body(A, w4, bias, bv[g]);
                    bv[g] = kk < 2 ? fmaf(r, 1e-7f, bv[g]) : bv[g];
                    acc += r;
                }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + u[0].x + u[1].y + v[0].y + v[1].x + bv[0] + bv[3];
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<float> hW(2 * (10 * HC + 1));
    for (size_t i = 0; i < hW.size(); ++i) hW[i] = 0.05f * (float)((int)(i * 2654435761u % 41) - 20) / 20.f;
    float *dW, *dout;
    CK(hipMalloc(&dW, hW.size() * sizeof(float)));
    CK(hipMemcpy(dW, hW.data(), hW.size() * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, (size_t)cus * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::printf("%s, %d CUs; Hc = %d; one call = 256 points per wave, both species; %d calls per launch\n", prop.name, cus, HC, REPS);
    std::printf("%-28s %8s %14s %16s\n", "waves per SIMD", "variant", "us per launch", "ns per call-wave");
    for (int wps : {1, 2, 4}) {                               // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
        const int blocks = cus * wps;
        float t[4];
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_branch<0>, dim3(blocks), dim3(256), 0, 0, dW, dout, REPS);
                if (mode == 1) hipLaunchKernelGGL(k_branch<1>, dim3(blocks), dim3(256), 0, 0, dW, dout, REPS);
                if (mode == 2) hipLaunchKernelGGL(k_branch<2>, dim3(blocks), dim3(256), 0, 0, dW, dout, REPS);
                if (mode == 3) hipLaunchKernelGGL(k_branch<3>, dim3(blocks), dim3(256), 0, 0, dW, dout, REPS);
            };
            launch(); launch();
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, 0));
                launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            t[mode] = best;
            // one SIMD runs wps waves, each REPS calls: time per call and wave slot on that SIMD
            std::printf("%-28d %8s %14.2f %16.1f\n", wps, mode == 0 ? "valu" : (mode == 1 ? "mfma" : (mode == 2 ? "both" : "valu_lds")), 1e3 * best,
                        1e6 * best / (REPS * wps));
        }
        std::printf("   -> mfma / valu = %.2f;  both / (valu + mfma) = %.2f  (1.0 = the pipes serialise, max/sum = %.2f = perfect overlap);  "
                    "valu_lds / valu = %.2f\n",
                    t[1] / t[0], t[2] / (t[0] + t[1]), (t[0] > t[1] ? t[0] : t[1]) / (t[0] + t[1]), t[3] / t[0]);
    }
    return 0;
}
