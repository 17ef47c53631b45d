// Micro-benchmarks used to understand where a Pi-block step spends its time on gfx950.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o ubench ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../percnn_amd/csrc/pi_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

// (1) raw VALU issue rate: NI independent fma chains per lane, ITER iterations
template <int NI, bool PACKED>
__global__ void valu_kernel(float* out, float a, float b, int iters, long long* cyc)
{
    float x[NI];
    for (int i = 0; i < NI; ++i) x[i] = threadIdx.x * 0.001f + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NI; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NI; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 1; }

__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* buf; CK(hipMalloc(&buf, 64 << 20));
    long long* cyc; CK(hipMalloc(&cyc, 64));
    float ms;

    // empty kernel back-to-back
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, buf);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("empty kernel 256x256: %.2f us per launch\n", ms * 1e3 / 2000);

    // copy 2 MiB -> 2 MiB (ping-pong) back-to-back: the memory-side floor of one 512^2 step
    {
        float4 *a = (float4*)buf, *b = (float4*)(buf + (4 << 20) / 4 * 2);
        int n = (2 << 20) / 16;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 2000; ++i) { hipLaunchKernelGGL(copy_kernel, dim3(n / 256), dim3(256), 0, st, a, b, n); std::swap(a, b); }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("dependent 2 MiB copy kernels (512 blocks): %.2f us per launch\n", ms * 1e3 / 2000);
    }

    // VALU rate
    for (int blocks : {256, 1024, 4096}) {
        const int iters = 2000;
        long long c = 0;
        hipLaunchKernelGGL((valu_kernel<16, false>), dim3(blocks), dim3(256), 0, st, buf, 1.0001f, 0.5f, iters, cyc);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((valu_kernel<16, false>), dim3(blocks), dim3(256), 0, st, buf, 1.0001f, 0.5f, iters, cyc);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        double instr = (double)iters * 16;
        printf("valu fma x16 chains, %4d blocks: %.1f us, wave0 %.2f clk64-ticks/instr, chip %.1f TFLOP/s\n", blocks,
               ms * 1e3, c / instr, 2.0 * instr * 64 * 4 * blocks / (ms * 1e-3) / 1e12);
    }

    // (2) the real forward kernel, 512^2, Hc=8
    {
        const int H = 512, W = 512, hc = 8;
        const long n = (long)H * W;
        float *h, *P;
        CK(hipMalloc(&h, 2 * n * 4 * 2));
        CK(hipMalloc(&P, 256 * 4));
        std::vector<float> hp(256, 0.01f), hh(2 * n, 0.5f);
        hp[0] = 0.5f; hp[1] = hp[2] = 1e-5f; hp[3] = -50000.f;
        for (int i = 4; i < 16; ++i) hp[i] = 13333.f;
        CK(hipMemcpy(P, hp.data(), 256 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(h, hh.data(), 2 * n * 4, hipMemcpyHostToDevice));
        pi::Geom g; g.n0 = H; g.n1 = 1; g.W = W; g.rows = H; g.s0 = W; g.ss = n; g.off = 0; g.wrap0 = 1;
        float* a = h; float* b = h + 2 * n;
        for (int block : {64, 256}) {
            const int grid = (int)((n / 4 + block - 1) / block);
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 1000; ++i) {
                    hipLaunchKernelGGL((pi::pi_fwd_kernel<float, 2, 8, 4>), dim3(grid), dim3(block), 0, st, a, b, P, g, hc);
                    std::swap(a, b);
                }
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("pi_fwd_kernel<f32,2D,Hc8,vec4> 512^2 block %d: %.2f us per step\n", block, ms * 1e3 / 1000);
        }
        // same kernel, independent launches (no ping-pong dependency: same in/out every time)
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 1000; ++i)
            hipLaunchKernelGGL((pi::pi_fwd_kernel<float, 2, 8, 4>), dim3(256), dim3(256), 0, st, a, b, P, g, hc);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pi_fwd_kernel same buffers every launch: %.2f us per step\n", ms * 1e3 / 1000);
        // 8x larger grid (2048 rows): throughput mode
        {
            float* big; CK(hipMalloc(&big, 2L * 4096 * 512 * 4 * 2));
            CK(hipMemset(big, 0, 2L * 4096 * 512 * 4 * 2));
            pi::Geom g2 = g; g2.n0 = 4096; g2.rows = 4096; g2.ss = 4096L * 512;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 200; ++i)
                hipLaunchKernelGGL((pi::pi_fwd_kernel<float, 2, 8, 4>), dim3(2048), dim3(256), 0, st, big, big + 2L * 4096 * 512, P, g2, hc);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("pi_fwd_kernel 4096x512 (8x points): %.2f us per launch = %.2f us per 512^2-equivalent\n", ms * 1e3 / 200, ms * 1e3 / 200 / 8);
        }
    }
    return 0;
}
