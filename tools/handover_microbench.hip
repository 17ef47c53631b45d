// feasibility microbenchmark: neighbour-flag hand-over between persistent workgroups (one per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int TX = 16, TY = 16, NB = TX * TY, NT = 256;
constexpr int PER = 2;                    // uint4 per thread per frame  (256 thr * 2 * 16 B = 8 KiB per block per iteration)
constexpr long SPIN_MAX = 2000000;

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ inline uint4 ld_sc1(const uint4* p) {
    v4u v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline void st_sc1(uint4* p, uint4 u) {
    v4u v = {u.x, u.y, u.z, u.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

template <int MODE>   // 0: compiler memory model (release/acquire, agent scope); 1: write-through stores + bypass loads, relaxed flags
__global__ __launch_bounds__(NT) void pp_kernel(uint4* buf, int* flags, int iters, long* stamps, int* errors) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int bx = b % TX, by = b / TX;
    int nb[8]; int k = 0;
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) if (dx || dy)
        nb[k++] = ((by + dy + TY) % TY) * TX + (bx + dx + TX) % TX;
    long t0 = wall_clock64();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        // ---- wait for the 8 neighbours to have published frame `it`
        if (it > 0) {
            if (tid < 8) {
                long spin = 0;
                if (MODE == 0) { while (__hip_atomic_load(&flags[nb[tid]], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < it && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1); }
                else           { while (__hip_atomic_load(&flags[nb[tid]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1); }
                if (spin >= SPIN_MAX) atomicAdd(errors, 1);
            }
            __syncthreads();
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // read the neighbours' frame `it` and check it
            for (int n = 0; n < 8; ++n) {
                const uint4* src = buf + ((size_t)(it & 1) * NB + nb[n]) * (NT * PER) + tid;
                uint4 v;
                if (MODE == 0) v = *src;
                else { const unsigned* s32 = reinterpret_cast<const unsigned*>(src);      // pipelined device-coherent loads
                       v.x = __hip_atomic_load(s32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                       v.w = __hip_atomic_load(s32 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                unsigned want = (unsigned)it * 65536u + (unsigned)nb[n];
                bad += (v.x != want) + (v.w != want + (unsigned)tid);
            }
        }
        // ---- publish own frame it+1
        uint4* dst = buf + ((size_t)((it + 1) & 1) * NB + b) * (NT * PER);
        unsigned tag = (unsigned)(it + 1) * 65536u + (unsigned)b;
        for (int p = 0; p < PER; ++p) {
            uint4 v = make_uint4(tag, p, 7u, tag + (unsigned)tid);
            if (MODE == 0) dst[p * NT + tid] = v; else st_sc1(dst + p * NT + tid, v);
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (MODE == 0) __hip_atomic_store(&flags[b], it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else           __hip_atomic_store(&flags[b], it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    long t1 = wall_clock64();
    if (bad) atomicAdd(errors + 1, (int)bad);
    if (tid == 0) { stamps[2 * b] = t0; stamps[2 * b + 1] = t1; }
}

__global__ __launch_bounds__(NT) void one_kernel(uint4* buf, int it) {      // launch-per-iteration comparison
    const int b = blockIdx.x, tid = threadIdx.x;
    const int bx = b % TX, by = b / TX;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) if (dx || dy) {
        int n = ((by + dy + TY) % TY) * TX + (bx + dx + TX) % TX;
        uint4 v = buf[((size_t)(it & 1) * NB + n) * (NT * PER) + tid];
        acc.x += v.x; acc.w += v.w;
    }
    uint4* dst = buf + ((size_t)((it + 1) & 1) * NB + b) * (NT * PER);
    for (int p = 0; p < PER; ++p) dst[p * NT + tid] = make_uint4(acc.x + it, p, 7u, acc.w);
}

int main() {
    uint4* buf; int* flags; long* stamps; int* errors;
    CK(hipMalloc(&buf, sizeof(uint4) * 2 * NB * NT * PER));
    CK(hipMalloc(&flags, sizeof(int) * NB)); CK(hipMalloc(&stamps, sizeof(long) * 2 * NB)); CK(hipMalloc(&errors, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) for (int iters : {100, 1000}) {
        CK(hipMemset(flags, 0, sizeof(int) * NB)); CK(hipMemset(errors, 0, 8)); CK(hipMemset(buf, 0, sizeof(uint4) * 2 * NB * NT * PER));
        CK(hipEventRecord(e0));
        if (mode == 0) pp_kernel<0><<<NB, NT>>>(buf, flags, iters, stamps, errors);
        else           pp_kernel<1><<<NB, NT>>>(buf, flags, iters, stamps, errors);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long> st(2 * NB); int err[2];
        CK(hipMemcpy(st.data(), stamps, sizeof(long) * 2 * NB, hipMemcpyDeviceToHost)); CK(hipMemcpy(err, errors, 8, hipMemcpyDeviceToHost));
        long lo = st[0], hi = st[1];
        for (int b = 0; b < NB; ++b) { lo = std::min(lo, st[2 * b]); hi = std::max(hi, st[2 * b + 1]); }
        printf("mode %d iters %d: event %.3f us/iter, device clock %.3f us/iter, spin timeouts %d, stale words %d\n", mode, iters,
               ms * 1e3 / iters, (hi - lo) * 0.01 / iters, err[0], err[1]);
    }
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) one_kernel<<<NB, NT>>>(buf, it);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("launch per iteration: %.3f us/iter\n", ms * 1e3 / iters);
    }
    return 0;
}
