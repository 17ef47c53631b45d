// stream_mix.hip -- what the memory system of one MI355X gives a launch-per-step sweep over a trajectory, by access pattern
// (VERDICT r4 #1b/#1c: "access-pattern ceiling").  No Pi-block arithmetic: each kernel issues the loads / stores of one pattern
// with 16 bytes per lane and one add per loaded value, on the frames of [T+1][2][n0][n1][W] float32 trajectories exactly as
// the rollout lays them out (species planes n0*n1*W apart, frames twice that), one launch per time step, launches dependent.
//   copy      out[t+1] = in[t]                                  1 read stream, 1 write   (the forward step's algorithmic bytes)
//   r3w1      Gp = G + h + inj                                  3 reads, 1 write         (the adjoint step's algorithmic bytes)
//   zwin      out = sum of the chunk in planes z-2..z+2         forward step's plane-neighbour loads, no in-plane neighbours
//   zwin_r3w1 Gp = sum_z G + h + inj                            adjoint step's loads without the in-plane neighbours
// build: hipcc --offload-arch=gfx950 -O3 -o stream_mix tools/ubench/stream_mix.hip ; run: ./stream_mix n0 n1 W T [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct G3 { int n0; long plane4; long species4; };   // planes, float4 per plane, float4 per species

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// lane -> (plane z, chunk c): blocks walk planes in order; XCD-contiguous remap so that an XCD owns a range of planes
__device__ __forceinline__ unsigned remap(unsigned b, unsigned n) { const unsigned q = n / 8; return (n % 8 == 0) ? (b % 8) * q + b / 8 : b; }

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                         float4* __restrict__ o, G3 g)
{
    const unsigned blk = remap(blockIdx.x, gridDim.x);
    const long i = (long)blk * 256 + threadIdx.x;               // chunk of one species
    if (i >= g.species4) return;
    const int z = (int)(i / g.plane4);
    const long r = i - (long)z * g.plane4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float4* as = a + s * g.species4;
        float4 v;
        if (MODE == 0 || MODE == 1) v = as[i];
        else {
            v = as[i];
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                if (d == 0) continue;
                int zz = z + d; zz += zz < 0 ? g.n0 : 0; zz -= zz >= g.n0 ? g.n0 : 0;
                v = add4(v, as[(long)zz * g.plane4 + r]);
            }
        }
        if (MODE == 1 || MODE == 3) v = add4(v, add4(b[s * g.species4 + i], c[s * g.species4 + i]));
        o[s * g.species4 + i] = v;
    }
}

int main(int argc, char** argv)
{
    const int n0 = argc > 1 ? atoi(argv[1]) : 128, n1 = argc > 2 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 128;
    const int T = argc > 4 ? atoi(argv[4]) : 16, reps = argc > 5 ? atoi(argv[5]) : 5;
    G3 g{n0, (long)n1 * W / 4, (long)n0 * n1 * W / 4};
    const size_t frame4 = 2 * (size_t)g.species4, bytes = (size_t)(T + 1) * frame4 * 16;
    float4 *traj, *gtraj, *adj;
    CK(hipMalloc(&traj, bytes)); CK(hipMalloc(&gtraj, bytes)); CK(hipMalloc(&adj, bytes));
    CK(hipMemset(traj, 0, bytes)); CK(hipMemset(gtraj, 0, bytes)); CK(hipMemset(adj, 0, bytes));
    const unsigned grid = (unsigned)((g.species4 + 255) / 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"copy (1r 1w)", "r3w1 pointwise", "zwin (5 planes r, 1w)", "zwin + 2 pointwise r, 1w"};
    const double alg[4] = {2.0, 4.0, 2.0, 4.0};
    std::printf("# %d x %d x %d float32, 2 species, T = %d frames, one launch per step, %u workgroups of 256\n", n0, n1, W, T, grid);
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<float> us;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0));
            for (int t = 0; t < T; ++t) {
                const float4* A = (mode == 0 || mode == 2) ? traj + (size_t)t * frame4 : adj + (size_t)(T - t) * frame4;
                float4* O = (mode == 0 || mode == 2) ? traj + (size_t)(t + 1) * frame4 : adj + (size_t)(T - t - 1) * frame4;
                const float4* B = traj + (size_t)(T - t - 1) * frame4;
                const float4* C = gtraj + (size_t)(T - t - 1) * frame4;
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, A, B, C, O, g); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, A, B, C, O, g); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, A, B, C, O, g); break;
                default: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, A, B, C, O, g); break;
                }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) us.push_back(ms * 1e3f / T);
        }
        std::sort(us.begin(), us.end());
        const double u = us[us.size() / 2], gb = alg[mode] * (double)g.species4 * 2 * 16 / 1e9;   // alg[] frames
        std::printf("%-28s %9.2f us per step  %7.0f GB/s on %.0f algorithmic MB\n", names[mode], u, gb / (u * 1e-6), gb * 1e3);
    }
    return 0;
}
