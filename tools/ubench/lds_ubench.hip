#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
// MODE 0: b128 reads, MODE 1: b64 reads.  pitch (bytes) between rows of 8 lanes; lane l reads row l/8, column (l%8)*16 bytes
template <int NT, int MODE>
__global__ void __launch_bounds__(NT) k(float* out, unsigned long long* t, int n, int pitch, int lanes_per_row, int colstride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += NT) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
    unsigned base = (unsigned)((lane / lanes_per_row) * pitch + (lane % lanes_per_row) * colstride + wave * 8 * pitch) % 32768u;
    f4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned a = base + (unsigned)(j * 1024 % 16384);
            asm volatile("" : "+v"(a));
            if (MODE == 0) { const f4 v = *reinterpret_cast<const f4*>(smem + a); acc += v; }
            else { const f2 v = *reinterpret_cast<const f2*>(smem + a); acc.x += v.x; acc.y += v.y; }
        }
    }
    const unsigned long long t1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * NT] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
template <int NT, int MODE> void run(float* o, unsigned long long* t, int pitch, int lpr, int cs, const char* name)
{
    const int n = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int r = 0; r < 2; ++r) { k<NT, MODE><<<1, NT, 65536>>>(o, t, n, pitch, lpr, cs); hipDeviceSynchronize(); }
    unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    const double ns = h * 10.0 / (8.0 * n);                 // per read instruction per wave
    const double bytes = (MODE == 0 ? 16.0 : 8.0) * NT;     // per instruction round of the whole workgroup
    printf("%-34s %s NT %4d: %.2f ns per read per wave -> CU rate %.1f B/ns (~%.0f B/clk at 2.4 GHz)\n", name, MODE == 0 ? "b128" : "b64 ", NT, ns,
           bytes / ns, bytes / ns / 2.4);
}
int main()
{
    float* o; unsigned long long* t; hipMalloc(&o, 1 << 22); hipMalloc(&t, 64);
#define ALL(P, L, C, NAME) run<64, 0>(o, t, P, L, C, NAME); run<256, 0>(o, t, P, L, C, NAME); run<512, 0>(o, t, P, L, C, NAME); run<256, 1>(o, t, P, L, C, NAME); run<512, 1>(o, t, P, L, C, NAME);
    ALL(1024, 64, 16, "linear (lane * 16 B)")
    ALL(192, 8, 16, "8 lanes per row, pitch 192 B")
    ALL(208, 8, 16, "8 lanes per row, pitch 208 B")
    ALL(256, 8, 16, "8 lanes per row, pitch 256 B")
    ALL(144, 8, 16, "8 lanes per row, pitch 144 B (3D)")
    return 0;
}
