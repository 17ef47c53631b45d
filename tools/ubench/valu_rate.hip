// VALU issue rate on gfx950 for the instruction kinds the 2D tile sub-steps are made of ("the tile
// sub-steps are VALU-issue-bound"): v_fma_f32 vs v_pk_fma_f32 vs v_fma_f64, with NI independent chains per lane
// (NI = 1: dependent-issue latency; NI = 8: throughput), at 1 and 2 waves per SIMD (256 / 512 threads, one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e) { printf("HIP error %d line %d\n", (int)e, __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND, int NI>   // 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_fma_f64, 3: v_pk_mul_f32 + v_pk_add_f32 alternating
__global__ void rate_kernel(float* out, float a, float b, int iters, long long* cyc)
{
    f2 x[NI]; double d[NI];
    for (int i = 0; i < NI; ++i) { x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i}; d[i] = x[i].x; }
    const f2 va = {a, a * 1.0001f}, vb = {b, b * 1.01f};
    const double da = a, db = b;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
                if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
                if (KIND == 3) {
                    if (r & 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(va));
                    else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(vb));
                }
            }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NI; ++i) s += x[i].x + x[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NI>
int run(const char* name, int threads, float* buf, long long* cyc)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000; float ms = 0; long long c = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((rate_kernel<KIND, NI>), dim3(256), dim3(threads), 0, 0, buf, 1.0001f, 0.5f, iters, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double instr = (double)iters * 8 * NI;                  // per wave
    const double wps = threads / 256.0;                           // waves per SIMD
    // kernel time / (instructions issued per SIMD) in ns -> cycles at the clock the part actually ran (reported both ways)
    printf("%-22s NI=%d threads=%d: %.3f ns per wave-instruction, %.3f ns per SIMD-instruction slot, s_memtime ticks/instr %.3f\n",
           name, NI, threads, ms * 1e6 / instr, ms * 1e6 / (instr * wps), (double)c / instr);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    float* buf; CK(hipMalloc(&buf, 64 << 20)); long long* cyc; CK(hipMalloc(&cyc, 64));
    for (int threads : {256, 512, 1024}) {
        if (run<0, 1>("v_fma_f32 dependent", threads, buf, cyc)) return 1;
        if (run<0, 8>("v_fma_f32", threads, buf, cyc)) return 1;
        if (run<1, 1>("v_pk_fma_f32 dependent", threads, buf, cyc)) return 1;
        if (run<1, 8>("v_pk_fma_f32", threads, buf, cyc)) return 1;
        if (run<3, 8>("v_pk_mul/add_f32", threads, buf, cyc)) return 1;
        if (run<2, 1>("v_fma_f64 dependent", threads, buf, cyc)) return 1;
        if (run<2, 8>("v_fma_f64", threads, buf, cyc)) return 1;
    }
    return 0;
}
