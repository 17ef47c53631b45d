// valu_issue_ubench.hip -- how fast ONE wave (and 2 / 4 per SIMD) issues packed / scalar fp32 FMAs on gfx950, dependent vs independent.
// Answers: is a pass of the tile kernels with one wave per SIMD slow because of dependency latency (fixable by interleaving
// chains) or because a lone wave cannot fill the VALU whatever it issues?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valu_issue_ubench tools/valu_issue_ubench.hip && ./valu_issue_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

// CH independent chains, each ITER x UNROLL dependent instructions
template <int CH, bool PK>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float a, float b)
{
    v2f x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = v2f{(float)threadIdx.x + c, 1.0f + c};
    const v2f va{a, a}, vb{b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(va), "v"(vb));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c].x) : "v"(a), "v"(b));
            }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c].x + x[c].y;
    if (s == 12345.678f) out[0] = s;
}

template <int CH, bool PK>
void run(int waves_per_simd, float* out)
{
    const int iters = 2000, nt = 256 * waves_per_simd;      // one workgroup per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<CH, PK><<<256, nt>>>(out, 10, 1.0f, 0.5f);
    CK(hipEventRecord(e0));
    k<CH, PK><<<256, nt>>>(out, iters, 1.0f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)iters * 16 * CH;
    std::printf("%-10s chains %d  waves/SIMD %d : %.2f ns per instruction per wave, %.2f ns per instruction per SIMD\n", PK ? "v_pk_fma" : "v_fma", CH,
                waves_per_simd, ms * 1e6 / instr, ms * 1e6 / instr / waves_per_simd);
}

int main()
{
    float* out; CK(hipMalloc(&out, 64));
    for (int w : {1, 2, 4}) {
        run<1, true>(w, out); run<2, true>(w, out); run<4, true>(w, out); run<8, true>(w, out);
        run<1, false>(w, out); run<2, false>(w, out); run<4, false>(w, out);
    }
    return 0;
}
