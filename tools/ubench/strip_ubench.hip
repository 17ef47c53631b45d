#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../percnn_amd/csrc/pi_tile2d.h"   // (tools/ubench/ is two levels below the repo root, like tools/scratch/)
using namespace pi;
template <int NT, int MODE>
__global__ void __launch_bounds__(NT) k(const float* __restrict__ P_in, float* out, unsigned long long* t, int n)
{
    constexpr int K = 4, B = 32;
    using TL = Tile<K, B, B>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* b0 = reinterpret_cast<float*>(smem_raw) + lds_pad0<float>::value;
    float* b1 = reinterpret_cast<float*>(smem_raw) + 2 * TL::PLANE + lds_pad1<float>::value;
    for (int i = threadIdx.x; i < 4 * TL::PLANE; i += NT) reinterpret_cast<float*>(smem_raw)[i + 2] = 0.001f * (i % 97);
    float Ph[NPOLY];
#pragma unroll
    for (int i = 0; i < NPOLY; ++i) { float x = P_in[i]; asm volatile("" : "+v"(x)); Ph[i] = x; }
    // an interior strip: rows 8 .. 8 + NT/8, 8 strips per row
    const int sy = 8 + (int)threadIdx.x / 8, sx = 8 + 4 * ((int)threadIdx.x % 8);
    unsigned w = (unsigned)(sy * TL::LX + sx) | (1u << 16) | (0xFu << 17);
    asm volatile("" : "+v"(w));
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { fwd_strip_geo<float, K, B, B>(b0, b1, Ph, w); fwd_strip_geo<float, K, B, B>(b1, b0, Ph, w); }
        if (MODE == 1) { fwd_strip_geo<float, K, B, B>(b0, b1, Ph, w); lds_barrier(); fwd_strip_geo<float, K, B, B>(b1, b0, Ph, w); lds_barrier(); }
        if (MODE == 2) { fwd_strip_geo_loads_first<float, K, B, B>(b0, b1, Ph, w); fwd_strip_geo_loads_first<float, K, B, B>(b1, b0, Ph, w); }
    }
    const unsigned long long t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * NT] = b0[sy * TL::LX + sx];
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
template <int NT, int MODE> void run(const float* P, float* o, unsigned long long* t, int blocks, const char* name)
{
    const size_t lds = tile_state_bytes<float, 4, 32, 32>() + 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int n = 2000;
    for (int r = 0; r < 2; ++r) { k<NT, MODE><<<blocks, NT, lds>>>(P, o, t, n); hipDeviceSynchronize(); }
    unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-28s NT %4d blocks %3d: %.3f us per strip per wave\n", name, NT, blocks, h / 100.0 / (2.0 * n));
}
int main(int argc, char** argv)
{
    float hP[36]; for (int i = 0; i < 36; ++i) hP[i] = 0.01f * (i + 1);
    float *P, *o; unsigned long long* t;
    hipMalloc(&P, 256); hipMalloc(&o, 1 << 22); hipMalloc(&t, 64);
    hipMemcpy(P, hP, sizeof hP, hipMemcpyHostToDevice);
    if (argc > 1) {                                          // one configuration only (counter runs)
        const int nt = std::atoi(argv[1]);
        if (nt == 64) run<64, 0>(P, o, t, 256, "one wave alone");
        if (nt == 256) run<256, 0>(P, o, t, 256, "strips back to back");
        if (nt == 512) run<512, 0>(P, o, t, 256, "strips back to back");
        return 0;
    }
    for (int blocks : {1, 256}) {
        run<256, 0>(P, o, t, blocks, "strips back to back");
        run<512, 0>(P, o, t, blocks, "strips back to back");
        run<256, 1>(P, o, t, blocks, "strip + barrier");
        run<512, 1>(P, o, t, blocks, "strip + barrier");
        run<256, 2>(P, o, t, blocks, "loads first, back to back");
        run<64, 0>(P, o, t, blocks, "one wave alone");
    }
    return 0;
}
