// pi_peer.h -- peer-mailbox halo transport of the slab path (gfx950, xGMI).
//
// A ring exchange through RCCL costs one ncclGroup call per exchange (~19-25 us on MI355X for four sends and four
// receives of <= 2 MiB, measured send/recv-to-self) -- half of what a 32 x 256^2 slab spends computing a time step.  xGMI
// peers are load/store addressable, so the faces can travel as plain stores instead: every rank owns a MAILBOX in
// fine-grained device memory (uncached at the receiver: remote stores bypass its L2), maps the mailboxes of its two ring
// neighbours once (hipIpc handles), and an exchange is two kernels on the compute stream, no host synchronisation:
//
//   put   my first / last `width` interior planes -> slot [epoch & 1] of the prev / next rank's mailbox; the last
//         workgroup of each direction then publishes `epoch` in that mailbox's arrival flag (system-scope release);
//   take  every workgroup waits (bounded) until its arrival flag reaches `epoch`, then copies the slot into my halo planes.
//
// Slots are double-buffered by exchange parity.  Why that suffices on a lock-step ring: my put of exchange e+2 is ordered
// (same stream) behind my take of e+1, which saw the neighbour's flag e+1, which the neighbour raised in its put e+1,
// which is ordered behind ITS take of exchange e -- the last reader of the slot I am about to overwrite.
// A take that times out (neighbour dead / ranks out of step) records the epoch in the mailbox's error word and every
// later take returns at once: wrong halos, but no hang; the host reads the word with percnn_pi_peer_box_status().
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

namespace pi {

struct PeerBox {                                 // header of a mailbox; slots follow at PEER_HDR
    unsigned long long flag[2][16];              // [0] written by my prev neighbour, [1] by my next: newest complete exchange
    unsigned long long error[16];                // != 0: epoch of the first take that timed out
    unsigned count[2][32];                       // block counters of MY put kernels (per direction)
    unsigned face[2][32];                        // fused put: bricks of MY step kernel that have stored their share of face [dir]
};
constexpr size_t PEER_HDR = 4096;
static_assert(sizeof(PeerBox) <= PEER_HDR, "mailbox header");

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t peer_round16(size_t b) { return (b + 15) & ~(size_t)15; }
inline size_t peer_box_bytes(size_t slot_bytes) { return PEER_HDR + 4 * ((slot_bytes + 4095) & ~(size_t)4095); }
inline char* peer_slot(void* box, size_t slot_bytes, int parity, int dir)
{
    return static_cast<char*>(box) + PEER_HDR + (size_t)(parity * 2 + dir) * ((slot_bytes + 4095) & ~(size_t)4095);
}

// copy `bytes` (both pointers 16-byte aligned iff VEC) with this direction's share of the grid; four loads in flight
// per lane (the source of a take is uncached memory, the destination of a put is on the far side of an xGMI link)
template <bool VEC>
__device__ __forceinline__ void peer_copy(const char* __restrict__ src, char* __restrict__ dst, size_t bytes, int lb, int nb)
{
    using U = typename std::conditional<VEC, v4u, unsigned>::type;     // element sizes are 4 or 8 bytes
    const size_t n = bytes / sizeof(U), stride = (size_t)nb * blockDim.x;
    const U* s = reinterpret_cast<const U*>(src);
    U* d = reinterpret_cast<U*>(dst);
    size_t i = (size_t)lb * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const U a = s[i], b = s[i + stride], c = s[i + 2 * stride], e = s[i + 3 * stride];
        d[i] = a; d[i + stride] = b; d[i + 2 * stride] = c; d[i + 3 * stride] = e;
    }
    for (; i < n; i += stride) d[i] = s[i];
}

struct PeerXfer {
    const char* src[2][2];       // [direction][species]: faces inside my slab (put) / slots inside my mailbox (take)
    char* dst[2][2];             //                        slots inside the neighbours' mailboxes (put) / my halo planes (take)
    size_t bytes;                // per species and direction
    PeerBox* signal[2];          // put: the mailbox whose flag[direction] announces the arrival
    PeerBox* mine;               // my mailbox (put: block counters; take: arrival flags, error word)
    unsigned long long epoch;
    int blocks_per_dir;
    unsigned wire_ticks;         // MEASUREMENT AID (option peer_wire_us, default 0): the put holds the arrival flag back this many
                                 // 10 ns ticks after its stores have left -- a stand-in for the link time of a real xGMI hop when
                                 // the ring runs through one device's own mailbox (examples/slab_delay_ring.cpp)
};

// the last put workgroup of a direction, just before it announces the slot
__device__ __forceinline__ void peer_wire_delay(unsigned ticks)
{
    if (ticks == 0u) return;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
}

template <bool VEC>
__global__ void __launch_bounds__(256) peer_put_kernel(PeerXfer x)
{
    const int dir = (int)blockIdx.x / x.blocks_per_dir, lb = (int)blockIdx.x % x.blocks_per_dir;
    for (int s = 0; s < 2; ++s) peer_copy<VEC>(x.src[dir][s], x.dst[dir][s], x.bytes, lb, x.blocks_per_dir);
    __threadfence_system();                                      // my stores have left for the neighbour's memory
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(&x.mine->count[dir][0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)x.blocks_per_dir - 1) {
            __hip_atomic_store(&x.mine->count[dir][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            peer_wire_delay(x.wire_ticks);
            __hip_atomic_store(&x.signal[dir]->flag[dir][0], x.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- put fused into the step kernel (round 3, VERDICT r2 #2c) ----------------------------------------------------------------
// The separate put kernel could only start when the whole step kernel had finished, and the take behind it waits for the
// neighbour's put: the wire time of an exchange (2 MiB per direction over one xGMI link: ~30 us) sat fully exposed between two
// steps.  Fused: the step kernel itself is launched with `nput` extra workgroups (the lowest block ids).  A brick that stores
// planes of a face (write-through stores, drained) bumps that face's counter in my mailbox; a put workgroup waits until its
// face's counter has reached `expect` -- every brick holding a share of the face has stored it --, acquires, and copies its
// share into the neighbour's slot while the bricks of the other planes are still computing; the last put workgroup of a
// direction publishes the exchange number exactly as peer_put_kernel does and clears the face counter.  The take stays a
// launch of its own.  No deadlock whatever the dispatch order: bricks never wait for anything, and the put workgroups are a
// handful against >= 1024 resident slots, so a brick they wait for always finds a slot; their wait is bounded like a take's.
struct PeerPutFused {
    PeerXfer x;                  // as for peer_put_kernel (src = face planes of the frame the step kernel writes)
    int nput;                    // workgroups that do the put (2 * x.blocks_per_dir, a multiple of 8); 0 = no fused put
    int lo[2], hi[2];            // plane ranges [lo, hi) of the two faces, in the step kernel's own plane numbering
    unsigned expect[2];          // bricks that store planes of face [dir]
    unsigned long long timeout_ticks;
    int vec16;                   // all face / slot pointers 16-byte aligned, sizes multiples of 16
};

// does a brick of planes [z0, z1) hold a share of one of the two faces?  (block-uniform)
__device__ __forceinline__ bool peer_holds_face(const PeerPutFused& f, int z0, int z1)
{
    return (z0 < f.hi[0] && z1 > f.lo[0]) || (z0 < f.hi[1] && z1 > f.lo[1]);
}

// a brick that has stored planes [z0, z1): all its (write-through) stores are drained before the counters move
__device__ __forceinline__ void peer_face_stored(const PeerPutFused& f, int z0, int z1)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
            if (z0 < f.hi[d] && z1 > f.lo[d])
                __hip_atomic_fetch_add(&f.x.mine->face[d][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// put workgroup b of the fused launch (b < f.nput)
template <bool VEC>
__device__ __forceinline__ void peer_put_block(const PeerPutFused& f, int b)
{
    const PeerXfer& x = f.x;
    const int dir = b / x.blocks_per_dir, lb = b % x.blocks_per_dir;
    __shared__ int ready;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(&x.mine->face[dir][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < f.expect[dir]) {
            if (wall_clock64() - t0 > f.timeout_ticks) {                 // a brick that never came: record it like a take's time-out
                unsigned long long expect = 0;
                __hip_atomic_compare_exchange_strong(&x.mine->error[0], &expect, x.epoch, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");               // the bricks' written-through planes, not my L1's idea of them
        ready = ok;
    }
    __syncthreads();
    if (ready)
        for (int s = 0; s < 2; ++s) peer_copy<VEC>(x.src[dir][s], x.dst[dir][s], x.bytes, lb, x.blocks_per_dir);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(&x.mine->count[dir][0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)x.blocks_per_dir - 1) {
            __hip_atomic_store(&x.mine->count[dir][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&x.mine->face[dir][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (a put that gave up does not announce the slot: the neighbour's take times out and poisons its halo)
            const bool clean = __hip_atomic_load(&x.mine->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
            peer_wire_delay(x.wire_ticks);
            if (ready && clean) __hip_atomic_store(&x.signal[dir]->flag[dir][0], x.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// the same segmented copy without any signalling: packs the faces of both species into one staging message per direction
// (and unpacks the received ones) for the RCCL ring -- an ncclGroup costs per operation (~3 us each on MI355X), so 2 sends +
// 2 receives of packed faces beat 4 + 4 of per-species planes even with the two copy launches
template <bool VEC>
__global__ void __launch_bounds__(256) face_copy_kernel(PeerXfer x)
{
    const int dir = (int)blockIdx.x / x.blocks_per_dir, lb = (int)blockIdx.x % x.blocks_per_dir;
    for (int s = 0; s < 2; ++s) peer_copy<VEC>(x.src[dir][s], x.dst[dir][s], x.bytes, lb, x.blocks_per_dir);
}

template <bool VEC>
__global__ void __launch_bounds__(256) peer_take_kernel(PeerXfer x, unsigned long long timeout_ticks)
{
    const int dir = (int)blockIdx.x / x.blocks_per_dir, lb = (int)blockIdx.x % x.blocks_per_dir;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();            // 100 MHz
        while (__hip_atomic_load(&x.mine->flag[dir][0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < x.epoch) {
            if (__hip_atomic_load(&x.mine->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (wall_clock64() - t0 > timeout_ticks) {
                unsigned long long expect = 0;
                __hip_atomic_compare_exchange_strong(&x.mine->error[0], &expect, x.epoch, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __shared__ int arrived;
    if (threadIdx.x == 0)
        arrived = __hip_atomic_load(&x.mine->flag[dir][0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= x.epoch;
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                // every wave: nothing older than the flag is read below
    if (arrived) {
        for (int s = 0; s < 2; ++s) peer_copy<VEC>(x.src[dir][s], x.dst[dir][s], x.bytes, lb, x.blocks_per_dir);
    } else {
        // timed out (now or in an earlier exchange: the error word is sticky): whatever sits in the slot is NOT this
        // exchange's face.  Poison the halo planes instead of copying it, so that a late or dead neighbour shows up as
        // NaNs in the very next step rather than as silently wrong halos (all-ones bit patterns are NaNs in float32 and
        // float64 alike); the hosts' status checks report the exchange number.
        const size_t words = x.bytes / 4, per = (words + (size_t)x.blocks_per_dir - 1) / (size_t)x.blocks_per_dir;
        const size_t w0 = per * (size_t)lb, w1 = w0 + per < words ? w0 + per : words;
        for (int s = 0; s < 2; ++s) {
            unsigned* d = reinterpret_cast<unsigned*>(x.dst[dir][s]);
            for (size_t i = w0 + threadIdx.x; i < w1; i += blockDim.x) d[i] = 0xffffffffu;
        }
    }
}

}  // namespace pi
