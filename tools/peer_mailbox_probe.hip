// Feasibility probe for the slab path's peer-mailbox halo transport: two PROCESSES exchange halo
// faces by one-sided stores into each other's fine-grained device memory (hipIpc handles) and signal with epoch flags,
// everything stream-ordered, no host synchronisation and no RCCL call.  On a one-GPU box both processes use device 0
// (that is also how tests/test_slab_dist_gpu.py exercises the product); with more GPUs pass the device of each process.
//
// Per exchange and process:   put kernel (faces -> peer mailbox slot [epoch & 1], last block raises the peer's flag)
//                             take kernel (every block waits for its flag >= epoch, then mailbox -> halo planes)
// Reported: microseconds per exchange for 2 x 2 faces of `bytes` each, and payload verification.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pmp tools/peer_mailbox_probe.hip && /tmp/pmp [bytes_per_face] [dev0 dev1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] hip error %s line %d\n", getpid(), hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

typedef unsigned long long u64;
typedef float v4f __attribute__((ext_vector_type(4)));

struct Box {                       // header of a mailbox allocation (fine-grained device memory of the RECEIVER)
    u64 flag[2][16];               // [from prev / from next], one 128-byte line each: epoch of the newest complete put
    u64 error[16];
    unsigned count[2][32];         // block counters of the sender's put kernels (per direction)
};

__global__ void put_kernel(const v4f* face, v4f* slot, size_t n16, u64* peer_flag, unsigned* counter, u64 epoch)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(face[i], slot + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(peer_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void take_kernel(const v4f* slot, v4f* halo, size_t n16, const u64* my_flag, u64 epoch, u64* error)
{
    if (threadIdx.x == 0) {
        const u64 t0 = wall_clock64();
        while (__hip_atomic_load(my_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 200000000ull) { error[0] = epoch; break; }       // 2 s at 100 MHz
        }
    }
    __syncthreads();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        halo[i] = __builtin_nontemporal_load(slot + i);
}

__global__ void fill_kernel(float* p, size_t n, float v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023); }
__global__ void check_kernel(const float* p, size_t n, float v, unsigned* bad) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != v + (float)(i & 1023)) atomicAdd(bad, 1u); }

static void xfer(int wfd, int rfd, const void* mine, void* theirs, size_t n)
{
    if (write(wfd, mine, n) != (ssize_t)n) { perror("write"); exit(3); }
    size_t got = 0;
    while (got < n) { ssize_t r = read(rfd, (char*)theirs + got, n - got); if (r <= 0) { perror("read"); exit(3); } got += r; }
}

int run(int rank, int dev, int wfd, int rfd, size_t face_bytes)
{
    CK(hipSetDevice(dev));
    const size_t slot_bytes = 2 * face_bytes;                       // both species of one face
    const size_t total = 4096 + 2 /*parity*/ * 2 /*direction*/ * slot_bytes;
    char* box = nullptr;
    CK(hipExtMallocWithFlags((void**)&box, total, hipDeviceMallocFinegrained));
    CK(hipMemset(box, 0, total));
    hipIpcMemHandle_t mine, theirs;
    CK(hipIpcGetMemHandle(&mine, box));
    xfer(wfd, rfd, &mine, &theirs, sizeof(mine));
    char* peer = nullptr;
    CK(hipIpcOpenMemHandle((void**)&peer, theirs, hipIpcMemLazyEnablePeerAccess));
    // world size 2: the peer is both my prev and my next
    float *faces, *halos; unsigned* bad;
    CK(hipMalloc(&faces, 2 * slot_bytes)); CK(hipMalloc(&halos, 2 * slot_bytes)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto slot = [&](char* b, int parity, int dir) { return (v4f*)(b + 4096 + ((size_t)parity * 2 + dir) * slot_bytes); };
    const size_t n16 = slot_bytes / 16;
    const int blocks = (int)((n16 + 255) / 256 < 64 ? (n16 + 255) / 256 : 64);
    const int iters = 300, check_every = 50;
    int token = 0, dummy = 0;
    xfer(wfd, rfd, &token, &dummy, sizeof(int));                     // both mapped
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        const u64 base = (u64)rep * iters;
        CK(hipEventRecord(e0, st));
        for (int it = 1; it <= iters; ++it) {
            const u64 epoch = base + it;
            const bool chk = it % check_every == 0;
            if (chk) for (int dir = 0; dir < 2; ++dir)
                hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, st, faces + dir * (slot_bytes / 4), slot_bytes / 4,
                                   (float)(1000 * (rank + 1) + 10 * dir + (int)(epoch % 7)));
            Box* pb = (Box*)peer; Box* mb = (Box*)box;
            for (int dir = 0; dir < 2; ++dir)                        // dir 0: to "next" (arrives as from-prev), 1: to "prev"
                hipLaunchKernelGGL(put_kernel, dim3(blocks), dim3(256), 0, st, (const v4f*)(faces + dir * (slot_bytes / 4)),
                                   slot(peer, (int)(epoch & 1), dir), n16, &pb->flag[dir][0], &pb->count[dir][0], epoch);
            for (int dir = 0; dir < 2; ++dir)
                hipLaunchKernelGGL(take_kernel, dim3(blocks), dim3(256), 0, st, (const v4f*)slot(box, (int)(epoch & 1), dir),
                                   (v4f*)(halos + dir * (slot_bytes / 4)), n16, &mb->flag[dir][0], epoch, &mb->error[0]);
            if (chk) for (int dir = 0; dir < 2; ++dir)
                hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, st, halos + dir * (slot_bytes / 4), slot_bytes / 4,
                                   (float)(1000 * (2 - rank) + 10 * dir + (int)(epoch % 7)), bad);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned hbad = 0; u64 herr = 0;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&herr, &((Box*)box)->error[0], 8, hipMemcpyDeviceToHost));
    printf("[rank %d dev %d] %zu B per face and species: %.2f us per exchange (2 puts + 2 takes; incl. the verification passes), wrong values %u, timeouts %llu\n",
           rank, dev, face_bytes, ms * 1e3 / iters, hbad, herr);
    xfer(wfd, rfd, &token, &dummy, sizeof(int));                     // nobody unmaps while the other still runs
    CK(hipIpcCloseMemHandle(peer));
    CK(hipFree(box));
    return (hbad || herr) ? 1 : 0;
}

int main(int argc, char** argv)
{
    const size_t face_bytes = argc > 1 ? (size_t)atol(argv[1]) : (size_t)4 * 256 * 256 * 4;   // 4 planes of 256^2 floats
    const int dev0 = argc > 3 ? atoi(argv[2]) : 0, dev1 = argc > 3 ? atoi(argv[3]) : 0;
    int a2b[2], b2a[2];
    if (pipe(a2b) || pipe(b2a)) { perror("pipe"); return 3; }
    const pid_t pid = fork();                                        // before any HIP call
    if (pid == 0) return run(1, dev1, b2a[1], a2b[0], face_bytes);
    const int rc = run(0, dev0, a2b[1], b2a[0], face_bytes);
    int status = 0; waitpid(pid, &status, 0);
    return rc | (WIFEXITED(status) ? WEXITSTATUS(status) : 9);
}
