// Feasibility microbenchmark #2 for a persistent 2D tile kernel: halo hand-over between
// resident workgroups with DATA-TAGGED GRANULES (8-byte {epoch, value} words, agent-scope relaxed atomics = sc1
// stores / loads: the data is the flag; guide "Guideline 16", recipe R2) instead of the flag protocol priced in
// tools/handover_microbench.hip (4.8 us per hand-over).
//
// Models the 512^2 headline problem: 256 workgroups x 512 threads, one 32x32 two-species tile each, XCD-aware tile
// placement (one 8x4 rectangle of tiles per XCD), K = 4 sub-steps per hand-over.  Per iteration a workgroup
//   1. "computes" (a dependent LDS + FMA chain of a configurable length standing in for the K sub-steps),
//   2. stores its tile to the trajectory (plain 16-byte stores, 8 KiB),
//   3. publishes the 8-wide border band of its tile as granules (1536 values = 12 KiB),
//   4. gathers its 8-wide halo ring (2560 values) from the 8 neighbours' bands, polling until every tag matches.
// Reported: microseconds per iteration with and without steps 3-4 -> cost of one hand-over, to be compared with the
// launch boundary it would replace (2.3 us launch/drain + 1.4-2.4 us window load of the K = 4 tile kernels).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hg tools/handover_granule_microbench.hip && /tmp/hg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int TX = 16, TY = 16, NB = TX * TY, NT = 512, B = 32, HW = 8;      // tiles, threads, tile edge, halo width
constexpr int BAND = 2 * (B * B - (B - 2 * HW) * (B - 2 * HW));              // 1536 border values (both species)
constexpr int RING = 2 * ((B + 2 * HW) * (B + 2 * HW) - B * B);              // 2560 halo values
constexpr long SPIN_MAX = 4000000;
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

// XCD-aware block -> tile map of the tile kernels (pi_tile2d.h: tile_of_block): rectangles of 8 x 4 tiles per XCD
__device__ __host__ inline int tile_of_block(int b, int xcd_aware)
{
    if (!xcd_aware) return b;
    const int rx = 2, rw = 8, rh = 4;
    const int xcd = b % 8, j = b / 8;
    const int ty = (xcd / rx) * rh + j / rw, tx = (xcd % rx) * rw + j % rw;
    return ty * TX + tx;
}

// position p (0..B*B-1 row-major inside the tile) -> index inside the border band, -1 if interior
__device__ inline int band_index(int y, int x)
{
    if (y < HW) return y * B + x;
    if (y >= B - HW) return HW * B + (y - (B - HW)) * B + x;
    const int r = y - HW;                                     // middle rows: 2*HW values per row
    if (x < HW) return 2 * HW * B + r * 2 * HW + x;
    if (x >= B - HW) return 2 * HW * B + r * 2 * HW + HW + (x - (B - HW));
    return -1;
}

template <int EXCH>   // 0: compute + trajectory store only; 1: + granule publish / gather
__global__ __launch_bounds__(NT) void persistent_kernel(gu64* outbox /*[2][NB][BAND]*/, float4* traj, int iters, int work,
                                                        int xcd_aware, long* stamps, int* errors, const int* tile_to_block)
{
    __shared__ float lds[2 * (B + 2 * HW) * (B + 2 * HW)];
    const int tid = threadIdx.x, tile = tile_of_block(blockIdx.x, xcd_aware);
    const int ty = tile / TX, tx = tile % TX;
    for (int i = tid; i < 2 * (B + 2 * HW) * (B + 2 * HW); i += NT) lds[i] = 0.001f * i;
    __syncthreads();
    unsigned bad = 0;
    const long t0 = wall_clock64();
    float acc = 1.0f + tid * 1e-6f;
    for (int it = 0; it < iters; ++it) {
        const unsigned epoch = (unsigned)it + 1u;
        // 1. stand-in for K sub-steps: dependent LDS reads + FMAs, LDS-only barriers between "sub-steps"
        for (int s = 0; s < 4; ++s) {
            for (int w = 0; w < work; ++w) acc = __builtin_fmaf(acc, 0.999f, lds[(tid * 5 + w * 37 + s) % (2 * 48 * 48)]);
            lds[(tid + s * NT) % (2 * 48 * 48)] = acc * 1e-3f;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // 2. trajectory store of the tile (plain, 16 B per lane: 2048 floats = 512 x float4)
        traj[((size_t)(it & 3) * NB + tile) * 512 + tid] = make_float4(acc, acc + 1, acc + 2, acc + 3);
        if (EXCH) {
            // 3. publish the border band: 1536 granules, 3 per thread
            gu64* mine = outbox + ((size_t)(epoch & 1) * NB + tile) * BAND;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = tid + q * NT;
                const unsigned val = (unsigned)tile * 4096u + (unsigned)i;
                __hip_atomic_store(mine + i, ((u64)epoch << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // 4. gather the halo ring: 2560 granules, 5 per thread, re-read until every tag of the wave matches
            gu64* src[5]; unsigned want[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int r = tid + q * NT;                   // ring element: species, then position around the tile
                const int sp = r / (RING / 2), e = r % (RING / 2);
                // enumerate ring positions row-major over the 48 x 48 window skipping the centre
                int wy, wx;
                if (e < HW * 48) { wy = e / 48; wx = e % 48; }
                else if (e < HW * 48 + B * 2 * HW) { const int m = e - HW * 48; wy = HW + m / (2 * HW); const int c = m % (2 * HW); wx = c < HW ? c : B + c; }
                else { const int m = e - HW * 48 - B * 2 * HW; wy = HW + B + m / 48; wx = m % 48; }
                const int gy = ty * B + wy - HW, gx = tx * B + wx - HW;            // global point (may wrap)
                const int nty = ((gy + TY * B) / B) % TY, ntx = ((gx + TX * B) / B) % TX;
                const int ly = (gy + TY * B) % B, lx = (gx + TX * B) % B;
                const int ntile = nty * TX + ntx;
                const int bi = band_index(ly, lx);
                src[q] = outbox + ((size_t)(epoch & 1) * NB + ntile) * BAND + sp * (BAND / 2) + bi;
                want[q] = (unsigned)ntile * 4096u + (unsigned)(sp * (BAND / 2) + bi);
            }
            long spins = 0;
            unsigned v[5];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const u64 x = __hip_atomic_load(src[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[q] = (unsigned)x;
                    ok &= (unsigned)(x >> 32) == epoch;
                }
                if (__all(ok)) break;
                if (++spins > SPIN_MAX) { atomicAdd(errors, 1); break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                bad += v[q] != want[q];
                lds[(tid + q * NT) % (2 * 48 * 48)] = __uint_as_float(v[q]) * 0.0f + acc;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    const long t1 = wall_clock64();
    if (bad) atomicAdd(errors + 1, (int)bad);
    if (tid == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
    if (acc == 12345.678f) traj[0].x = acc;
    (void)tile_to_block;
}

int main()
{
    gu64* outbox; float4* traj; long* stamps; int* errors;
    CK(hipMalloc((void**)&outbox, sizeof(u64) * 2 * NB * BAND));
    CK(hipMalloc(&traj, sizeof(float4) * 4 * NB * 512));
    CK(hipMalloc(&stamps, sizeof(long) * 2 * NB)); CK(hipMalloc(&errors, 8));
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    int occ0 = 0, occ1 = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, persistent_kernel<0>, NT, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, persistent_kernel<1>, NT, 0));
    printf("CUs %d, resident workgroups per CU: %d / %d -> %s\n", prop.multiProcessorCount, occ0, occ1,
           prop.multiProcessorCount * occ1 >= NB ? "all 256 workgroups co-resident" : "NOT co-resident: would deadlock");
    if (prop.multiProcessorCount * occ1 < NB) return 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 1000;
    for (int xcd_aware : {1, 0}) for (int work : {0, 40, 120}) {
        float us[2] = {0, 0};
        int err[2] = {0, 0};
        for (int exch = 0; exch < 2; ++exch) {
            for (int rep = 0; rep < 2; ++rep) {              // second repetition is the one reported
                CK(hipMemsetAsync((void*)outbox, 0, sizeof(u64) * 2 * NB * BAND));
                CK(hipMemsetAsync(errors, 0, 8));
                CK(hipEventRecord(e0));
                if (exch) persistent_kernel<1><<<NB, NT>>>(outbox, traj, iters, work, xcd_aware, stamps, errors, nullptr);
                else      persistent_kernel<0><<<NB, NT>>>(outbox, traj, iters, work, xcd_aware, stamps, errors, nullptr);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                us[exch] = ms * 1e3f / iters;
                if (exch) CK(hipMemcpy(err, errors, 8, hipMemcpyDeviceToHost));
            }
        }
        printf("xcd_aware %d work %3d: compute+store %.3f us/iter, with granule hand-over %.3f us/iter -> hand-over %.3f us"
               "  (spin timeouts %d, wrong values %d)\n", xcd_aware, work, us[0], us[1], us[1] - us[0], err[0], err[1]);
    }
    return 0;
}
