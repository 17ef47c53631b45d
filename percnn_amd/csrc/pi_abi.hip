// pi_abi.hip -- C-ABI entry points of libpercnn_pi.so (declared in include/percnn_pi.h).
// Host side only validates arguments, picks a kernel instantiation and enqueues launches on the
// caller's stream: no allocation, no synchronisation, no exceptions across the boundary.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>

#include "../../include/percnn_pi.h"
#include "pi_kernels.h"
#include "pi_tile2d.h"
#include "pi_stream3d.h"
#include "pi_brick3d.h"
#include "pi_res3d.h"
#include "pi_contract.h"
#include "pi_peer.h"
#include "pi_adv.h"
#include "pi_host.h"

namespace {

using pi::Geom;

struct Options {
    int block = 256;
    int vec = 0;            // 0 = widest legal (16 B per lane); 1/2/4 = cap (tuning aid)
    int wgrad_blocks = 1024;
    int tile = 1;           // 2D: temporally blocked LDS kernels where the shape allows
    int tile_k = 4;         // sub-steps per launch (2 or 4)
    int tile_nt = 512;      // workgroup size of the tile kernels (256 or 512)
    int stream3d = 1;       // 3D: plane-streaming kernels where the shape allows (W = 64*VEC)
    int tile_fuse = 1;      // 2D tile sweep (float32 poly, K = 4, 32x32 tiles, 512 threads): reduce the 20 coefficient moments
                            // inside the sweep launches and keep only the hand-over adjoint frames (no moments pass)
    int bwd_cpl = 2;        // direct adjoint kernel: chunks per lane (grid-stride) while >= 512 workgroups remain; measured
                            // 1 -> 2: 128^3 22.9 -> 21.6, 192^3 76.9 -> 64.7, 2048^2 43.8 -> 37.9 us per fused backward step
    int zc = 8;             // planes per workgroup of the streaming kernels
    int overlap = 0;        // 1: run the gradient reduction on a side stream, chunk by chunk, under the sweep (measured: no gain on MI355X)
    int overlap_chunk = 128; // time steps per chunk
    int fuse_wgrad = 2;     // rollout sweep with the fused per-step kernel (all gradients reduced in the sweep launches)
                            // instead of sweep + one time-parallel reduction: 0 never, 1 whenever the per-step direct kernels
                            // sweep, 2 = where it measured faster: float32 poly mode on the direct-kernel path (128^3:
                            // 26.0 -> 23.3 us per step, 2048^2: 45.4 -> 39.8; fp64 and the factored mode lose)
    int skip_wgrad = 0;     // diagnostics: rollout_bwd runs the adjoint sweep only (bench uses it to time the sweep alone)
    int tile_xcd = 1;       // XCD-aware block -> tile map of the 2D tile kernels (0 = identity)
    int tile_by = 0;        // tile height of the 2D tile kernels: 32, 16, or 0 = by grid size (see tile_by_for)
    int slab_fused_put_adj = 0; // ... and the adjoint sweep's faces by the sweep launch (to self: no gain; across xGMI: bench.py decides)
    int peer_upb = 2048;    // mailbox put / take launches: 16-byte units per workgroup and species (peer_prepare)
    int peer_upb_take = 0;  // ... of the take alone (0 = as the put)
    int tile_persist = 2;   // float32 poly blocks on whole 32 x 32 tiles, <= one tile per CU: the whole tile sweep
                            // of a rollout in ONE launch of resident workgroups (pi_adj2d_persist_kernel).  2 = plain launch (one
                            // workgroup fills a CU's LDS, so a grid of <= #CUs is resident as long as no OTHER kernel holds whole
                            // CUs: calls on other streams of this process are detected and take the launch-per-group path;
                            // processes that share one GPU must set 0); 1 = cooperative launch (residency guaranteed by the
                            // runtime, but ~0.4 ms per launch on ROCm 7.2: slower than what it saves at T = 1000); 0 = off
    int persist_handshake = 1;  // persistent sweep: the entry point waits (host spin on a host-mapped word, no stream sync) until
                            // the launch reports that every workgroup is resident -- or that it aborted, in which case the
                            // launch-per-group sweep is enqueued in the same call and the persistent path is switched off for
                            // this device (percnn_pi_persist_status).  0: no wait; an aborted launch is reported by the NEXT
                            // entry point (PERCNN_PI_EASYNC) instead
    int fwd_small_half = 1;     // small-tile resident forward (32 x 8 tiles) on half-strips: 512 lanes, two waves per SIMD
    int adj_small_half = 1;     // small-tile resident sweep (32 x 8 tiles) on half-strips: 512 lanes, two waves per SIMD
    // small-tile resident kernels: 64-clock units between publish and the first ring request (PersistArgs::pause); -1 = by tile count.
    // Re-swept with the half-strip / pair-granule kernels of round 6 (us per step at pause 12 | 16 | 20 | 24 | 28 | 32):
    //   forward  100^2 (52 tiles) .873 .867 .821 .819 .843 .871 | 160^2 .921 .916 .901 .886 .874 .874 | 256^2 .978 .941 .921 .865 .851 .878
    //   sweep    100^2 1.159 1.145 1.123 1.130 1.155 1.182 | 160^2 1.216 1.206 1.201 1.183 1.179 1.206 | 256^2 1.539 1.517 1.480 1.432 1.373 1.377
    int adj_small_pause = -1;   // -> 20 up to 64 tiles, 28 above
    int fwd_small_pause = -1;   // -> 24 up to 64 tiles, 28 above
    int persist_small = 1;      // the 32 x 8-tile regime (grids below ~300^2, split schedule) as one persistent launch too
    int fwd_persist_per_cu = 1; // ... on grids of up to this many tiles per CU (1 or 2)
    int fwd_persist_f64 = 1;    // ... float64 too (lambda-omega; 16-byte granules)
    int adj_persist_f64 = 1;    // the float64 tile SWEEP as one resident launch as well (split flavour, moments in shared LDS rows)
    int fwd_persist = 1;        // the FORWARD rollout of such a grid as one launch of resident workgroups too (pi_fwd2d_persist_kernel;
                            // same residency check / abort / fallback; its granule outbox is a per-device scratch of the library)
    int persist_split = 1;      // persistent sweep: 1 = split flavour (pi_adj2d_persist_split_kernel: the halo-independent
                            // "pyramid" of the next group runs while the granules of the hand-over travel), 0 = round 3's kernel
    int persist_timeout_ms = 2000;       // bound of one hand-over wait inside the persistent sweep
    int persist_first_timeout_ms = 100;  // ... of the first one (residency check)
    int tile_wide = 3;      // float32 poly blocks: 3 = 32x40 / 40x40 tiles where they keep the grid in one round (tile_wide_for),
                            // 0 = never, 1 / 2 = force 32x40 / 40x40
    int fwd_blocks = 0;     // direct forward kernel: grid cap (0 = none; measured: a bounded persistent grid loses, 384^3 376 -> 416 us)
    int xcd_window = 0;     // direct forward kernel: XCD-contiguous block remap inside windows of this many blocks (0 = whole grid)
    int block_small = 1;    // direct kernels: 64-thread workgroups while 256-thread ones would leave CUs idle (small grids)
    int rz = 0;             // direct 3D kernels, pre-contracted blocks: planes per workgroup pass sharing their plane neighbours in
                            // registers (1, 2, 4; 0 = by size, see direct_rz)
    int l2_tile_kb = 128;   // direct 3D kernels: y-tile of a plane (both species, KiB) whose five stencil planes stay in the L2
                            // (0 = whole planes, the pre-round-2 order): see set_blockmap
    int l2_tile_min_kb = 1536;  // ... applied once four neighbour planes x two species exceed this many KiB (0: always; tests)
    int peer_wire_us = 0;   // MEASUREMENT AID: every put over the peer mailboxes holds its arrival flag back this many microseconds
                            // (pi_peer.h PeerXfer::wire_ticks): a stand-in for the link time of an xGMI hop when the ring runs to self
    int slab_put_blocks = 16; // ... workgroups per direction of that fused put (a multiple of 4)
    int slab_fused_put = 1; // native slab rollouts over the peer mailboxes: the step kernel that writes a frame about to be exchanged
                            // also carries its faces into the neighbours' mailboxes (pi_peer.h "put fused into the step kernel")
    int slab_local_index = 1;   // native slab rollouts of ONE rank (ring == NULL): 1 = the periodic wrap is resolved by index inside the
                            // step launches (no face copies, no recomputed halo planes: 32 x 256^2, us per fwd+bwd step: 34.6 with one
                            // copy launch per exchange -> see profiles/r05_slab_local_wrap.txt); 0 = by face copies into the halo
                            // planes -- the launches of a multi-rank run minus its transport (what bench.py's compute / exchange
                            // split times; also selected per call by bit 1 of the `overlap` argument)
    int slab_wide_adjoint = 0;  // native slab backward over RCCL: one exchange per TWO adjoint steps (4 adjoint planes + 2 dL/dtraj
                            // planes per side in one group call), the 2-plane strips next to the faces recomputed locally.
                            // Built for VERDICT r1 #3, measured SLOWER on MI355X (RCCL to self, 32 x 256^2 slab: 74.0 -> 79.6 us
                            // per fwd+bwd step): an ncclGroup costs per operation (~3 us each of its 8 sends / receives), not per
                            // call, so 16 operations every second step save nothing and the two strip launches come on top
    int lane_x = 0;         // direct kernels: log2 of the lanes along x per row segment (2..6), 0 = fewest idle lanes (set_blockmap),
                            // -1 = the pre-round-2 rule (next power of two >= chunks per row)
    int brick3d = 1;        // 3D: brick kernels (pi_brick3d.h) for one-step launches where the shape allows: 0 never, 1 by
                            // size (brick_ok), 2 whenever eligible
    int res3d = 1;          // 3D float32 pre-contracted blocks: the whole reverse sweep of a rollout as ONE launch of resident
                            // workgroups with the adjoint state in LDS (pi_res3d.h; round 6): 0 never, 1 where it measured faster than
                            // the brick sweep (whole 16 x 16 x 32 blocks, at least 7/8 of the CUs busy: 128^3, 112 x 128^2), 2 on
                            // every grid of whole blocks that fits the device (tests)
    int brick_rz = 0;       // planes per brick (1, 2, 4; 0 = by size)
    int brick_xcd = 1;      // brick kernels: XCD regions split in y as well as in z where the counts divide (BrickGeom::xny):
                            // 0 never, 1 where it measured no worse (make_brick_geom), 2 always
    int brick_xny = 0;      // ... 0 = regions sized by the L2 (make_brick_geom), 1 = contiguous plane ranges, 2 / 4 / 8 = force that many
                            // row strips (tuning aid); -1 = the round-4 rule for forward steps of >= 8 M points (contiguous)
    int brick_wide = 1;     // 3D rows of 65 .. 128 chunks on 512-lane bricks (0: the direct kernels, as before round 4)
    int brick_nt = 0;       // lanes per brick workgroup: 512 = the wide flavour (pre-contracted blocks; see brick_nt_for), else 256
    int brick_wgs = 0;      // adjoint brick kernel: resident workgroups per CU that walk the bricks (0 = 4 / 2 by planes per brick)
    int brick_wt = 1;       // brick kernels store their output frame write-through (BrickGeom::wt)
    int lds_pad = 0;        // extra dynamic LDS per workgroup (bytes): lowers workgroups/CU so that a
                            // small grid is spread over all CUs instead of being packed onto a few
};
// The ONLY process-wide mutable state of the library: the table of tuning defaults.  Every entry point copies it once,
// under the lock, into its Problem (p.opt), overlays the call's own overrides ("key=value,..." of the *_opt entry
// points) and reads nothing else afterwards -- concurrent calls with different options do not interact.
Options g_defaults;
std::mutex g_defaults_mu;
int apply_option(Options& o, const char* key, long value);
int apply_overrides(Options& o, const char* spec);
void persist_reset();
int persist_async_error();

template <typename F>
hipError_t allow_lds(F* f, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// One side stream + a small event ring per host thread and device for sweep || reduction overlap (created lazily, released when
// the thread ends: applications that churn worker threads do not accumulate streams; fork/join through events only -- no host
// synchronisation).
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    hipEvent_t done = nullptr;
    int next = 0;
    bool ok = false;
    SideStream() = default;
    SideStream(const SideStream&) = delete;
    SideStream& operator=(const SideStream&) = delete;
    ~SideStream()
    {
        // (thread-local destructors of the main thread run inside exit() before the runtime's own teardown; errors are ignored --
        // work still queued on the stream completes, hipStreamDestroy only releases the handle)
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        if (done) (void)hipEventDestroy(done);
        if (stream) (void)hipStreamDestroy(stream);
        (void)hipGetLastError();
    }
};
SideStream* side_stream()
{
    // per host THREAD and device: two threads that drive two streams of one device must not share the event ring
    static thread_local SideStream per_dev[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream& s = per_dev[dev];
    if (!s.ok) {
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (auto& e : s.ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return nullptr;
        s.ok = true;
    }
    return &s;
}

constexpr int MAX_BWD_BLOCKS = 4096;   // bounds the per-workgroup gradient partials (grid-stride beyond)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Problem {
    int ndim, hc;
    int64_t n0, n1, W;
    int64_t n;          // points per species (interior)
    bool slab;
    int halo = 2;       // slab layout: planes present on each side of axis 0 (even, >= 2)
    int skip = 0;       // slab layout: outermost planes per side that this call neither reads nor writes
    int lo = -1, hi = -1;   // slab layout, plane-range calls: padded plane indices [lo, hi) this call computes
    bool slab_periodic = false; // slab layout of ONE rank: the n0 interior planes are a periodic domain of their own -- the wrap is
                                // resolved by index inside the step launches (plane -1 = interior plane n0 - 1), the halo planes
                                // are neither read nor written
    Options opt;            // this call's tuning options: process defaults at entry + per-call overrides
    pi::LossInj loss{0.0, nullptr, 0};   // adjoint calls: what the injection pointer means (pi_device.h); mode 0 = dL/dout itself
};

int make_problem(int hc, int ndim, const int64_t* shape, bool slab, Problem& p, const char* overrides = nullptr,
                 bool launches = true)                      // launches = false: a pure size / plan query
{
    {
        std::lock_guard<std::mutex> lk(g_defaults_mu);
        p.opt = g_defaults;
    }
    if (int rc = apply_overrides(p.opt, overrides)) return rc;
    if (launches)
        if (int rc = persist_async_error()) return rc;      // an earlier persistent sweep aborted and nobody has been told yet
    if (!shape || (ndim != 2 && ndim != 3) || hc < -1 || hc > 64) return PERCNN_PI_EINVAL;   // hc == 0: poly, -1: advective
    if (hc == -1 && slab) return PERCNN_PI_EINVAL;
    for (int a = 0; a < ndim; ++a)
        if (shape[a] < 2 || shape[a] > (1 << 30)) return PERCNN_PI_EINVAL;
    p.ndim = ndim; p.hc = hc; p.slab = slab;
    p.n0 = shape[0];
    p.n1 = ndim == 3 ? shape[1] : 1;
    p.W = shape[ndim - 1];
    p.n = p.n0 * p.n1 * p.W;
    if (p.n > (int64_t(1) << 40)) return PERCNN_PI_EINVAL;
    return 0;
}

pi::FastDiv make_fastdiv(unsigned d)
{
    pi::FastDiv f{0u, 0u};
    if (d <= 1) return f;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                                   // l = ceil(log2 d) >= 1
    f.m = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    f.s = l - 1;
    return f;
}

// chunk-id decomposition of the time-parallel / residual / advective kernels without integer division (vec = points per lane)
void set_fastdiv(Geom& g, int vec)
{
    const long nchunks = (long)g.rows * (g.W / vec);
    g.fastdiv = nchunks < (1L << 31) ? 1 : 0;
    g.dcpr = make_fastdiv((unsigned)(g.W / vec));
    g.dn1 = make_fastdiv((unsigned)g.n1);
}

// block-uniform decomposition of the direct step kernels (pi_kernels.h "Direct step kernels, addressing"): lanes along x
// = the power of two that wastes the fewest lanes on this row length; false if the grid is outside what 32-bit byte
// offsets / 31-bit block ids address (2D: the whole local field + 4 rows, 3D: one plane, must stay below 4 GiB)
bool set_blockmap(Geom& g, int ndim, int vec, int block, size_t elem, int l2_tile_bytes = 128 * 1024, int rz = 1,
                  long l2_tile_min_bytes = 3L << 19, int lane_x = 0)
{
    const long cpr = g.W / vec;
    // lanes along x: 2^lxs consecutive chunks of a row per wave row-segment.  A row of cpr chunks is covered by
    // ceil(cpr / 2^lxs) segments, so a width that is not a power of two leaves lanes idle (cpr = 48 on 64 lanes: 25 %; the
    // reference's 48^3: 12 chunks on 16 lanes).  Pick the power of two with the most useful lanes, weighted by what a
    // narrower contiguous segment costs (fitted to 96^3 .. 224^3 on MI355X, profiles/r02_lanes_along_x.txt: 128-byte
    // segments run at ~0.8 of 1 KiB ones, 256 / 512-byte ones at ~0.95): 160^3 13.8 k -> 17.0 k, 192^3 8.4 k -> 10.6 k steps/s.
    static const double seg_weight[5] = {0.70, 0.82, 0.95, 0.95, 1.0};     // 2^lxs = 4, 8, 16, 32, 64 chunks of 16 bytes
    int lxs = 2;
    double best_pow2 = 1.0;                               // score of the chosen power of two (forced / old rule: never flat)
    if (lane_x >= 2 && lane_x <= 6) {
        lxs = lane_x;                                     // tuning aid / A-B tests
    } else if (lane_x == -1) {                            // the pre-round-2 rule
        if (cpr <= 64) { while ((1L << lxs) < cpr) ++lxs; }
        else {
            lxs = 6;
            double best = 0.0;
            for (int c = 6; c >= 4; --c) {
                const long lx = 1L << c;
                const double eff = (double)cpr / (double)(((cpr + lx - 1) / lx) * lx);
                if (eff > best + 1e-9) { best = eff; lxs = c; }
            }
        }
    } else {
        const long rows = ndim == 3 ? g.n1 : g.n0;
        double best = 0.0;
        for (int c = 6; c >= 2; --c) {
            const long lx = 1L << c, rb = block >> c;     // rb rows of lx chunks per workgroup
            if (lx > block) continue;
            double eff = (double)cpr / (double)(((cpr + lx - 1) / lx) * lx) *
                         (double)rows / (double)(((rows + rb - 1) / rb) * rb);
            eff *= seg_weight[c - 2];
            // a row pitch that is not a multiple of 128 bytes leaves the segments straddling cache lines: S bytes touch
            // (S + 128) / 128 lines on average instead of S / 128 (200^3 with 128-byte segments: forward 39 -> 66 us)
            const long seg = lx * 16;
            if (((long)g.W * (long)elem) % 128 != 0) eff *= (double)seg / (double)(seg + 128);
            if (eff > best + 1e-9) { best = eff; lxs = c; }
        }
        best_pow2 = best;
    }
    while ((1 << lxs) > block) --lxs;
    const long nrow = ndim == 3 ? g.n1 : g.n0;
    // ... or none of them: consecutive chunks of the plane across row ends (`flat`; one multiply-shift per lane, rows follow
    // each other in memory so a wave still moves 1 KiB pieces) -- taken where the best power of two leaves > ~10 % idle
    bool flat = lane_x == 7;
    if (lane_x == 0 && nrow * cpr < (1L << 31)) {
        const long total = nrow * cpr, nb = (total + block - 1) / block;
        double eff = 0.97 * (double)total / (double)(nb * block);
        if (((long)g.W * (long)elem) % 128 != 0) eff *= 1024.0 / (1024.0 + 128.0);
        // (not with four planes per pass: 208^3 forward 41 -> 51 us, 240^3 65 -> 69 us when forced, while the two-plane
        // adjoint of the same grids gains 7 %)
        // and only below 8 M points, where the kernels are latency-bound and idle lanes are what costs: beyond that the
        // outcome follows the DRAM access pattern instead (same-box A/B: 288^3 +10 %, 224^3 -3 %, 352^3 -5 %)
        const long npts = (long)g.n0 * g.n1 * g.W;
        flat = eff > best_pow2 + 0.08 && rz <= 2 && npts < (8L << 20);
    }
    if (flat && nrow * cpr >= (1L << 31)) return false;
    const long lx = flat ? block : 1L << lxs, rpb = flat ? 1 : block >> lxs;
    g.lxs = flat ? -1 : lxs;
    g.nxb = flat ? 1 : (int)((cpr + lx - 1) / lx);
    g.nrg = flat ? (int)((nrow * cpr + block - 1) / block) : (int)((nrow + rpb - 1) / rpb);
    if (flat) g.dcpr = make_fastdiv((unsigned)cpr);
    g.rz = ndim == 3 ? rz : 1;
    const long ngroups = ndim == 3 ? (g.n0 + g.rz - 1) / g.rz : 1;          // plane groups of rz planes
    const long nblk = (long)g.nxb * g.nrg * ngroups;
    const unsigned long long span = (unsigned long long)(ndim == 3 ? (long)g.n1 : (long)g.n0 + 4) * g.W * elem;
    if (nblk <= 0 || nblk >= (1L << 31) || span >= (1ull << 32)) return false;
    g.nblk = (unsigned)nblk;
    g.dnxb = make_fastdiv((unsigned)g.nxb);
    g.dnrg = make_fastdiv((unsigned)g.nrg);
    // 3D y-tiling for L2 residency (Geom::rgt): a tile's plane section, both species, is held to ~128 KiB so that the four
    // neighbour planes of everything in flight on an XCD (~2 MiB) fit its 4 MiB L2 next to the streams themselves
    g.rgt = 0; g.nlast = 0; g.per_tile = 0;
    // (only where whole planes do not fit anyway: four neighbour planes x two species above ~1.5 MiB; measured on
    // MI355X: 384^3 backward 726 -> 513 us, 256^3 171 -> 148 us per step, 192^3 -- 1.2 MB of neighbour planes -- loses 3-8 %)
    if (ndim == 3 && l2_tile_bytes > 0 && 8L * g.n1 * g.W * (long)elem > l2_tile_min_bytes) {
        const long tile_rows = (long)l2_tile_bytes / (2 * (long)g.W * (long)elem);
        long rgt = flat ? (long)l2_tile_bytes / (2 * (long)block * vec * (long)elem) : tile_rows / rpb;
        if (rgt < 1) rgt = 1;
        if (rgt < g.nrg && (long)rgt * ngroups < (1L << 31)) {
            const long ntile = (g.nrg + rgt - 1) / rgt;
            g.rgt = (int)rgt;
            g.nlast = (int)(g.nrg - (ntile - 1) * rgt);
            g.per_tile = (unsigned)(rgt * ngroups);
            g.dper = make_fastdiv(g.per_tile);
            g.drgt = make_fastdiv((unsigned)g.rgt);
            g.dlast = make_fastdiv((unsigned)g.nlast);
        }
    }
    return true;
}

Geom make_geom(const Problem& p)
{
    Geom g;
    g.fastdiv = 0; g.dcpr = pi::FastDiv{0u, 0u}; g.dn1 = pi::FastDiv{0u, 0u};
    g.lxs = 0; g.nxb = g.nrg = 0; g.nblk = 0; g.dnxb = pi::FastDiv{0u, 0u}; g.dnrg = pi::FastDiv{0u, 0u};
    g.rgt = g.nlast = 0; g.per_tile = 0; g.dper = g.drgt = g.dlast = pi::FastDiv{0u, 0u};
    g.xwin = 0; g.rz = 1;
    g.loss = p.loss;
    g.n0 = (int)p.n0; g.n1 = (int)p.n1; g.W = (int)p.W;
    g.rows = (int)(p.n0 * p.n1);
    g.s0 = (long)(p.n1 * p.W);
    if (p.slab && p.slab_periodic) {
        g.ss = (long)(p.n0 + 2 * p.halo) * g.s0;
        g.off = (long)p.halo * g.s0;
        g.wrap0 = 1;
    } else if (p.slab) {
        // local array: n0 + 2*halo planes; this call computes planes [skip+2, n0+2*halo-skip-2)
        g.ss = (long)(p.n0 + 2 * p.halo) * g.s0;
        if (p.lo >= 0) {                                   // explicit plane range (communication overlap: faces first)
            g.off = (long)p.lo * g.s0;
            g.n0 = p.hi - p.lo;
        } else {
            g.off = (long)(p.skip + 2) * g.s0;
            g.n0 = (int)(p.n0 + 2 * p.halo - 2 * p.skip - 4);
        }
        g.rows = g.n0 * (int)p.n1;
        g.wrap0 = 0;
    } else {
        g.ss = (long)p.n0 * g.s0; g.off = 0; g.wrap0 = 1;
    }
    return g;
}

template <typename T>
int pick_vec(const Problem& p, std::initializer_list<const void*> ptrs)
{
    constexpr int V = pi::vec_width<T>::value;
    if (p.opt.vec == 1) return 1;
    if (p.W % V) return 1;
    for (const void* q : ptrs)
        if (q && (reinterpret_cast<uintptr_t>(q) % 16)) return 1;
    return V;
}

// workgroup size of the direct kernels: the "block" option, or 128 threads while 256-thread workgroups would leave
// CUs idle (a 48^3 grid is 108 workgroups of 256 threads on 256 CUs)
int direct_block(const Problem& p, const Geom& g, int vec)
{
    const long chunks = (long)g.rows * (g.W / vec);
    if (p.opt.block_small && p.opt.block == 256 && chunks < 1024L * 256) return 128;   // measured 48^3 .. 96^3: +0-10 %
    return p.opt.block;
}

// ---- kernel instantiation dispatch ------------------------------------------------------------
// planes per workgroup pass of the direct 3D kernels (the RZ > 1 flavours exist for pre-contracted blocks on 16-byte lanes)
// Measured on MI355X (profiles/r02_direct_kernel_option_sweeps.txt, us per step rz = 1 / 2 / 4): forward 384^3 363 / 312 /
// 283, 256^3 101 / 84 / 81, 128^3 12.3 / 11.1 / 11.8, 96^3 8.0 / 8.6 / 9.5, 48^3 4.7 / 5.5 / 7.0; backward 384^3 533 /
// 482 / 529, 256^3 152 / 141 / 153, 128^3 21.9 / 24.2 / 21.9, 96^3 15.8 / 15.0 / 17.8 -- sharing plane neighbours pays
// once the grid no longer fits the caches; small grids need the workgroups more than the reuse.
template <typename T>
int direct_rz(const Problem& p, int vec, bool adjoint)
{
    if (p.ndim != 3 || p.hc != 0 || vec != pi::vec_width<T>::value) return 1;
    if (p.opt.rz) return p.opt.rz;
    Geom g = make_geom(p);
    const int64_t pts = (int64_t)g.rows * p.W;
    if (adjoint) {
        // Two planes per pass save two of eleven neighbour loads per output but halve the workgroups.  Measured on MI355X
        // (profiles/r02_adjoint_rz.txt): they win from ~0.4 M points on (80^3 .. 112^3 -7 %, 144^3 -12 %, 160^3 / 192^3
        // -5 %) -- except where one plane per pass is exactly one full resident round of the chip (2048 threads per CU:
        // 120..128 x 128^2, 32 x 256^2: +8 % with two planes); one block more than that round and a plane per pass needs a
        // second round (130 x 126 x 128: 25.8 vs 23.3 us).
        if (pts >= ((int64_t)8 << 20)) return 2;
        if (pts < ((int64_t)3 << 17)) return 1;
        const int block = direct_block(p, g, vec);
        if (!set_blockmap(g, p.ndim, vec, block, sizeof(T), p.opt.l2_tile_kb * 1024, 1, (long)p.opt.l2_tile_min_kb * 1024,
                          p.opt.lane_x)) return 1;
        const int64_t threads = (int64_t)g.nblk * block, round = (int64_t)256 * 2048;
        return (threads <= round && threads > round - round / 8) ? 1 : 2;
    }
    return pts >= ((int64_t)8 << 20) ? 4 : (pts >= ((int64_t)3 << 19) ? 2 : 1);
}

template <typename T, int NDIM, int HC, int VEC, int RZ = 1>
hipError_t launch_fwd(const T* h, T* out, const T* P, const Problem& p, hipStream_t st)
{
    Geom g = make_geom(p);
    const int block = direct_block(p, g, VEC);
    if (g.rows <= 0) return hipSuccess;
    if (!set_blockmap(g, NDIM, VEC, block, sizeof(T), p.opt.l2_tile_kb * 1024, RZ, (long)p.opt.l2_tile_min_kb * 1024, p.opt.lane_x))
        return (hipError_t)PERCNN_PI_ETOOLARGE;              // 32-bit plane / field offsets (was an unspecific hipErrorInvalidValue)
    const unsigned grid = (p.opt.fwd_blocks > 0 && g.nblk > (unsigned)p.opt.fwd_blocks) ? (unsigned)p.opt.fwd_blocks : g.nblk;
    g.xwin = (unsigned)p.opt.xcd_window;
    auto* k = pi::pi_fwd_kernel<T, NDIM, HC, VEC, RZ>;
    if (hipError_t e = allow_lds(k, (size_t)p.opt.lds_pad)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), (size_t)p.opt.lds_pad, st, h, out, P, g, p.hc);
    return hipGetLastError();
}

unsigned bwd_grid(const Problem& p, int vec, size_t elem, int rz)
{
    Geom g = make_geom(p);
    if (g.rows <= 0 || !set_blockmap(g, p.ndim, vec, direct_block(p, g, vec), elem, p.opt.l2_tile_kb * 1024, rz, (long)p.opt.l2_tile_min_kb * 1024, p.opt.lane_x)) return 0;
    long need = g.nblk;
    const int cpl = rz > 1 ? (p.opt.bwd_cpl + rz - 1) / rz : p.opt.bwd_cpl;     // a pass already covers rz chunks per lane
    if (cpl > 1 && need >= 512L * cpl) need = (need + cpl - 1) / cpl;            // chunks per lane
    return (unsigned)(need < MAX_BWD_BLOCKS ? need : MAX_BWD_BLOCKS);
}

template <typename T, int NDIM, int HC, int VEC, bool WGRAD, int RZ = 1>
hipError_t launch_bwd(const T* h, const T* G, const T* inj, T* Gp, double* partials, const T* P,
                      const Problem& p, hipStream_t st)
{
    Geom g = make_geom(p);
    const int block = direct_block(p, g, VEC);
    const unsigned grid = bwd_grid(p, VEC, sizeof(T), RZ);
    if (g.rows <= 0) return hipSuccess;
    if (!grid || !set_blockmap(g, NDIM, VEC, block, sizeof(T), p.opt.l2_tile_kb * 1024, RZ, (long)p.opt.l2_tile_min_kb * 1024, p.opt.lane_x))
        return (hipError_t)PERCNN_PI_ETOOLARGE;
    const size_t lds = align_up((size_t)(block / pi::WAVE) * pi::nparams(p.hc) * sizeof(T), 16) +
                       (size_t)(block / pi::WAVE) * 2 * sizeof(double) +
                       ((WGRAD && HC == pi::POLY) ? (size_t)20 * (block + 8) * sizeof(T) : 0) +   // moment transpose scratch
                       (size_t)p.opt.lds_pad;
    auto* k = pi::pi_bwd_kernel<T, NDIM, HC, VEC, WGRAD, RZ>;
    if (hipError_t e = allow_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, st, h, G, inj, Gp, partials, P, g, p.hc);
    return hipGetLastError();
}

// time-parallel weight gradients over steps (t_lo, t_hi]
template <typename T, int JC, int NS, int VEC>
hipError_t launch_wgrad_pass(const T* traj, const T* adj, double* partials, const T* P, const Problem& p, int t_lo,
                             int t_hi, int j0, unsigned nb, hipStream_t st)
{
    const int block = 256;
    const size_t lds = (size_t)(block / pi::WAVE) * NS * (10 * JC + 1) * sizeof(T);
    const Geom g = make_geom(p);
    const long ss = p.slab ? g.ss : (long)p.n, off = p.slab ? (long)p.halo * g.s0 : 0;
    hipLaunchKernelGGL((pi::pi_wgrad_kernel<T, JC, NS, VEC>), dim3(nb, NS == 2 ? 1 : 2), dim3(block), lds, st, traj,
                       adj, partials, P, (long)p.n, ss, off, t_lo, t_hi, p.hc, j0);
    return hipGetLastError();
}

template <typename T, int VEC>
hipError_t launch_wgrad(const T* traj, const T* adj, double* partials, const T* P, const Problem& p, int t_lo,
                        int t_hi, unsigned* rows_out, hipStream_t st)
{
    const long total = (long)(t_hi - t_lo) * (p.n / VEC);
    long nb = (total + 256L * 16 - 1) / (256L * 16);          // >= 16 chunks per lane before adding blocks
    if (nb > p.opt.wgrad_blocks) nb = p.opt.wgrad_blocks;
    if (nb < 1) nb = 1;
    *rows_out = (unsigned)(2 * nb);
    if (p.hc == 0) {                                           // pre-contracted mode: coefficient moments
        const size_t lds = (size_t)(256 / pi::WAVE) * 20 * sizeof(T);
        const Geom g = make_geom(p);
        const long ss = p.slab ? g.ss : (long)p.n, off = p.slab ? (long)p.halo * g.s0 : 0;
        hipLaunchKernelGGL((pi::pi_moments_kernel<T, VEC>), dim3((unsigned)nb), dim3(256), lds, st, traj, adj, partials,
                           P, (long)p.n, ss, off, t_lo, t_hi);
        return hipGetLastError();
    }
    // small hidden widths: one workgroup handles both species (the state is streamed once)
    if (p.hc == 2) return launch_wgrad_pass<T, 2, 2, VEC>(traj, adj, partials, P, p, t_lo, t_hi, 0, (unsigned)nb, st);
    if (p.hc == 4) return launch_wgrad_pass<T, 4, 2, VEC>(traj, adj, partials, P, p, t_lo, t_hi, 0, (unsigned)nb, st);
    const int jc = p.hc % 8 == 0 ? 8 : p.hc % 4 == 0 ? 4 : p.hc % 2 == 0 ? 2 : 1;
    for (int j0 = 0; j0 < p.hc; j0 += jc) {
        hipError_t e;
        switch (jc) {
            case 8:  e = launch_wgrad_pass<T, 8, 1, VEC>(traj, adj, partials, P, p, t_lo, t_hi, j0, (unsigned)nb, st); break;
            case 4:  e = launch_wgrad_pass<T, 4, 1, VEC>(traj, adj, partials, P, p, t_lo, t_hi, j0, (unsigned)nb, st); break;
            case 2:  e = launch_wgrad_pass<T, 2, 1, VEC>(traj, adj, partials, P, p, t_lo, t_hi, j0, (unsigned)nb, st); break;
            default: e = launch_wgrad_pass<T, 1, 1, VEC>(traj, adj, partials, P, p, t_lo, t_hi, j0, (unsigned)nb, st); break;
        }
        if (e) return e;
    }
    return hipSuccess;
}

#define PI_DISPATCH_HC(CALL, NDIM, VEC)                         \
    switch (p.hc) {                                             \
        case 0:  return CALL(NDIM, pi::POLY, VEC);              \
        case 2:  return CALL(NDIM, 2, VEC);                     \
        case 4:  return CALL(NDIM, 4, VEC);                     \
        case 8:  return CALL(NDIM, 8, VEC);                     \
        default: return CALL(NDIM, 0, VEC);                     \
    }
#define PI_DISPATCH(CALL)                                                   \
    do {                                                                    \
        constexpr int V = pi::vec_width<T>::value;                          \
        if (p.ndim == 2) {                                                  \
            if (vec == 1) { PI_DISPATCH_HC(CALL, 2, 1) } else { PI_DISPATCH_HC(CALL, 2, V) } \
        } else {                                                            \
            if (vec == 1) { PI_DISPATCH_HC(CALL, 3, 1) } else { PI_DISPATCH_HC(CALL, 3, V) } \
        }                                                                   \
    } while (0)



// ---- advective polynomial blocks (hc == -1): one fused launch per step / adjoint step -----------------
template <typename T>
hipError_t adv_fwd(const T* h, T* out, const T* A, const Problem& p, hipStream_t st)
{
    const Geom g = make_geom(p);
    const unsigned grid = (unsigned)((p.n + 255) / 256);
    if (p.ndim == 2) hipLaunchKernelGGL((pi::pi_adv_fwd_kernel<T, 2>), dim3(grid), dim3(256), 0, st, h, out, A, g);
    else             hipLaunchKernelGGL((pi::pi_adv_fwd_kernel<T, 3>), dim3(grid), dim3(256), 0, st, h, out, A, g);
    return hipGetLastError();
}

template <typename T>
hipError_t adv_bwd(const T* h, const T* G, const T* inj, T* Gp, double* partials, const T* A, const Problem& p,
                   hipStream_t st, unsigned* rows)
{
    const Geom g = make_geom(p);
    long need = (p.n + 255) / 256;
    const unsigned grid = (unsigned)(need < MAX_BWD_BLOCKS ? need : MAX_BWD_BLOCKS);
    if (rows) *rows = grid;
    const size_t lds = (size_t)(256 / pi::WAVE) * pi::NADV * sizeof(double);
    if (p.ndim == 2) hipLaunchKernelGGL((pi::pi_adv_bwd_kernel<T, 2>), dim3(grid), dim3(256), lds, st, h, G, inj, Gp, partials, A, g);
    else             hipLaunchKernelGGL((pi::pi_adv_bwd_kernel<T, 3>), dim3(grid), dim3(256), lds, st, h, G, inj, Gp, partials, A, g);
    return hipGetLastError();
}

// ---- plane-streaming 3D path ---------------------------------------------------------------------
constexpr int STREAM_TY = 4;

// VEC such that one wave spans a full row (W == 64*VEC), 0 if the shape does not qualify
template <typename T>
int stream3d_vec(const Problem& p, std::initializer_list<const void*> ptrs, bool adjoint)
{
    if (!p.opt.stream3d || p.ndim != 3) return 0;
    // Round 3 (brick kernels, pi_brick3d.h): the z-march keeps the forward step from ~8 M points on (256^3: 63.8 vs 68.0 us);
    // below that and for every adjoint step the bricks are faster (64 x 256^2: 17.2 + 37.5 vs 19.1 + 39.6 us, 256^3 adjoint
    // 139.8 vs 144.2) -- which also retires the streaming adjoint's register spills
    if (p.opt.stream3d == 1 && p.opt.brick3d && (adjoint || (p.n0 + (p.slab ? 2 * p.halo : 0)) * p.n1 * p.W < ((int64_t)1 << 23)))
        return 0;
    // measured on MI355X: the plane-streaming kernels win from ~4M points per rank upwards (256^3: 4.1 vs
    // 2.7 TB/s forward); at 128^3 there are too few waves to cover their per-plane barrier chain.
    // Rows as wide as a full 16-B/lane wave (W = 256 fp32 -- the 32 x 256^2 slabs of the 8-GPU 256^3 problem) win
    // from ~2M points already (slab rollout 69 -> 63 us per step).
    // (plane-range calls are judged by the whole local slab -- the range in between the faces is most of it -- except
    // the few-plane faces themselves, which are not worth a z-march)
    if (p.lo >= 0 && p.hi - p.lo < 8 && p.opt.stream3d == 1) return 0;
    const int64_t pts = (p.n0 + (p.slab ? 2 * p.halo : 0)) * p.n1 * p.W;
    // Round 2 (direct kernels with scalar plane addressing + L2 y-tiling): a z-march only pays with enough planes per
    // workgroup chain -- the 32-plane slabs of the 8-GPU 256^3 problem now run faster on the direct kernels (47.9 vs
    // 58.7 us per fwd+bwd step, profiles/r02_slab_n1.txt); the whole 256^3 domain still prefers streaming forward
    // (66 vs 101 us) and ties backward (143 vs 147 us)
    if (p.opt.stream3d == 1 && (pts < ((int64_t)3 << 20) || p.n0 < 64)) return 0;
    if (p.hc != 0 && p.hc != 2 && p.hc != 4 && p.hc != 8) return 0;
    if (p.n1 % STREAM_TY) return 0;
    int vec = 0;
    for (int v = pi::vec_width<T>::value; v >= 1; v /= 2)
        if (p.W == (int64_t)pi::WAVE * v) { vec = v; break; }
    if (!vec) return 0;
    for (const void* q : ptrs)
        if (q && (reinterpret_cast<uintptr_t>(q) % (vec * sizeof(T)))) return 0;
    return vec;
}

inline int stream3d_zc(const Problem& p, const Geom& g, bool adj)
{
    int zc = adj ? 2 * p.opt.zc : p.opt.zc;               // the heavier adjoint body amortises its prologue over more planes
    const long ytiles = p.n1 / STREAM_TY;
    // forward, round 5 (profiles/r05_counters_summary.txt): a chain of 8 planes fetches 12 -- memory-side reads at 256^3 were
    // 199 MB for a 134 MB state; 16 planes per chain while >= 1024 workgroups remain: 172 MB, 64.8 -> 61.9 us per step
    if (!adj && p.opt.zc == 8 && ((g.n0 + 15) / 16) * ytiles >= 1024) zc = 16;
    while ((long)((g.n0 + zc - 1) / zc) * ytiles > MAX_BWD_BLOCKS) zc *= 2;
    return zc;
}

template <typename T, int HC, int VEC, bool ADJ>
hipError_t launch_stream3d(const T* f, T* out, const T* h, const T* inj, double* partials, const T* P, const Problem& p,
                           hipStream_t st, unsigned* rows_out, int with_mom)
{
    const Geom g = make_geom(p);
    const int zc = stream3d_zc(p, g, ADJ);
    const unsigned grid = (unsigned)(((g.n0 + zc - 1) / zc) * (p.n1 / STREAM_TY));
    if (rows_out) *rows_out = grid;
    if (g.n0 <= 0) return hipSuccess;
    const size_t lds = (size_t)4 * pi::Strip<T, VEC, STREAM_TY>::PLANE * sizeof(T);
    hipLaunchKernelGGL((pi::pi_stream3d_kernel<T, HC, VEC, STREAM_TY, ADJ>), dim3(grid), dim3(pi::WAVE * STREAM_TY), lds,
                       st, f, out, h, inj, partials, P, g, zc, p.hc, with_mom);
    return hipGetLastError();
}

template <typename T, bool ADJ>
hipError_t stream3d(int vec, const T* f, T* out, const T* h, const T* inj, double* partials, const T* P,
                    const Problem& p, hipStream_t st, unsigned* rows_out, int with_mom = 0)
{
#define CALL_S3(HC, VEC) launch_stream3d<T, HC, VEC, ADJ>(f, out, h, inj, partials, P, p, st, rows_out, with_mom)
#define S3_HC(VEC)                                                  \
    switch (p.hc) {                                                 \
        case 0:  return CALL_S3(pi::POLY, VEC);                     \
        case 2:  return CALL_S3(2, VEC);                            \
        case 4:  return CALL_S3(4, VEC);                            \
        default: return CALL_S3(8, VEC);                            \
    }
    if (vec == 1) { S3_HC(1) }
    if (vec == 2) { S3_HC(2) }
    if constexpr (pi::vec_width<T>::value == 4) { S3_HC(4) }
    return hipErrorInvalidValue;
#undef S3_HC
#undef CALL_S3
}

// ---- 3D brick kernels (pi_brick3d.h) --------------------------------------------------------------
// rows of up to 64 sixteen-byte chunks, planes below 4 GiB; returns planes per brick, 0 = not this path
// (wide = false: callers whose kernels exist with 256 lanes only -- the residual loss pass)
template <typename T>
int brick_rz_for(const Problem& p, int vec, bool adjoint, bool wide = true)
{
    if (!p.opt.brick3d || p.ndim != 3 || p.hc < 0 || (size_t)vec * sizeof(T) != 16) return 0;
    const int64_t cpr = p.W / vec;
    // rows of 65 .. 128 chunks (W = 260 .. 512 float32): the 512-lane flavours, which exist for pre-contracted blocks with the
    // plain injection form
    const int64_t cpr_max = (wide && p.opt.brick_wide && p.hc == 0 && p.loss.mode == 0) ? pi::brick_cpr_max(512) : pi::BRICK_CPR_MAX;
    if (cpr > cpr_max || p.n1 < 2 || p.n1 * p.W * (int64_t)sizeof(T) >= (int64_t(1) << 32)) return 0;
    // wide rows, same-box A/B (us per step, direct -> 512-lane bricks; profiles/r04_wide_bricks.txt): forward 384^3 281 -> 253,
    // 320^3 172 -> 146, 192 x 192 x 512 97 -> 86, 64 x 384^2 a tie; adjoint 64 x 384^2 125 -> 81, 192 x 192 x 512 198 -> 173, but
    // 320^3 288 -> 295, 384^3 475 -> 493 (one 512-lane workgroup per CU next to its 42 KB moment scratch)
    // -> the forward takes them from 16 M points on, the adjoint below 25 M
    const bool wide_row = cpr > pi::BRICK_CPR_MAX;
    if (wide_row && !p.opt.brick_rz && (adjoint ? p.n >= (int64_t(3) << 23) : p.n < (int64_t(1) << 24))) return 0;
    // planes per brick (profiles/r03_brick_sweeps.txt): sharing plane neighbours pays in the forward kernel from ~2 M points on
    // (128^3: 8.8 -> 8.4 us, 200^3: 33 -> 30); the adjoint keeps one plane (more workgroups in flight beats the reuse: 144^3 25.4 vs
    // 26.1 us, 200^3 59.3 vs 60.9; 128^3 ties); small grids need the workgroups (48^3: 99 k vs 89 k steps/s)
    // (beyond ~12 M points the adjoint is DRAM-bound and takes two planes as well: 256^3 172 -> 140 us)
    int rz = p.opt.brick_rz ? p.opt.brick_rz : (p.n >= (adjoint ? (int64_t(3) << 22) : (int64_t(1) << 21)) ? 2 : 1);
    if (rz > 1 && p.hc != 0) rz = 1;                        // the multi-plane flavours exist for pre-contracted blocks
    if (wide_row && rz > 2) rz = 2;                         // the 512-lane flavours: one or two planes per brick
    return rz;
}

// lanes per brick workgroup.  The four halo rows of a brick cost as many fetches as the brick owns chunks once a row is 64
// chunks wide (W = 256 float32: 256^3, and the 32 x 256^2 slabs of its 8-GPU decomposition); 512 lanes halve that share.
// (the 512-lane flavours exist for pre-contracted blocks and the plain injection form)
// Measured (same-box A/B over 14 shapes, T = 12): 512 lanes win 2-6 % where their bricks fill exactly one or two resident rounds
// (128^3, 32 x 256^2, 16 x 256^2) and lose 8-20 % wherever the coarser granule leaves a round partly empty (112^3, 144^3,
// 64 x 256^2, the 256^3 adjoint); in the driver's 500-step 128^3 run the gain was inside the box-to-box spread (forward 8.40 ->
// 7.84 us, adjoint 17.78 -> 18.29).  Not a default: option brick_nt = 512 selects them.
// Round 5: FORWARD steps on rows of exactly 64 chunks (W = 256 float32) take the 512-lane bricks by default -- with the L2-sized XCD
// regions, same box (us per forward step, 256 -> 512 lanes): 16 x 256^2 5.97 -> 5.62, 24 x 256^2 7.95 -> 7.49, 32 x 256^2 9.45 ->
// 8.50, 64 x 256^2 16.06 -> 15.12; the adjoint loses or ties on all four and stays on 256 lanes (profiles/r05_brick_option_sweeps.txt)
int brick_nt_for(const Problem& p, int vec, bool adjoint = true)
{
    if (p.hc == 0 && p.loss.mode == 0 && p.W / std::max(1, vec) > pi::BRICK_CPR_MAX) return 512;     // wide rows (brick_rz_for)
    // (while at least one 512-lane workgroup per CU remains: thinner slabs need the workgroups more than the shorter halo)
    if (!adjoint && p.opt.brick_nt == 0 && p.hc == 0 && p.loss.mode == 0 && p.W / std::max(1, vec) == pi::BRICK_CPR_MAX &&
        p.n1 % 8 == 0 && p.n / (4 * 512) >= 256) return 512;
    // Round 6: rows of 32 chunks (W = 128 float32) whose 512-lane two-plane bricks fill whole resident rounds (128^3: 512 bricks =
    // two per CU): forward 8.24 -> 8.03 us per step (same box, interleaved rounds; profiles/r06_forward_128_options.txt).  The
    // adjoint of that class is the resident sweep (pi_res3d.h) and, where that does not run, stays on 256 lanes (round 5: 17.8 -> 18.3)
    if (!adjoint && p.opt.brick_nt == 0 && p.hc == 0 && p.loss.mode == 0 && p.W / std::max(1, vec) == 32 && p.n1 % 16 == 0 &&
        p.n0 % 2 == 0 && p.n >= (int64_t(1) << 21) && ((p.n1 * 32 / 512) * (p.n0 / 2)) % 256 == 0) return 512;
    return (p.opt.brick_nt == 512 && p.hc == 0 && p.loss.mode == 0) ? 512 : 256;
}

pi::BrickGeom make_brick_geom(const Problem& p, int vec, int rz, int nt = pi::BRICK_NT, bool adjoint = true, bool faces_first = false)
{
    const Geom g = make_geom(p);
    pi::BrickGeom b;
    b.n0 = g.n0; b.n1 = g.n1; b.cpr = g.W / vec; b.total = b.n1 * b.cpr;
    b.nrg = (b.total + nt - 1) / nt;
    b.nblk = (unsigned)((long)b.nrg * ((g.n0 + rz - 1) / rz));
    b.wrap0 = g.wrap0; b.s0 = g.s0; b.ss = g.ss; b.off = g.off;
    b.dnrg = make_fastdiv((unsigned)b.nrg); b.dcpr = make_fastdiv((unsigned)b.cpr);
    b.nseg = (4 * b.cpr + 63) / 64; b.ntask = 2 * rz * b.nseg; b.dnseg = make_fastdiv((unsigned)b.nseg);
    b.wt = p.opt.brick_wt;
    b.loss = p.loss;
    // XCD regions (BrickGeom::xny): split the brick rows of a plane group between 2 or 4 XCDs where the counts divide and the
    // regions get closer to square in (planes, rows) -- fewer halo planes fetched by two L2s
    b.xny = 0; b.xpg = 0; b.xrg = 0; b.dxrg = pi::FastDiv{0u, 0u};
    const long npg = (g.n0 + rz - 1) / rz;
    // Measured (gpurun_out -> profiles/r03_brick_xcd_regions.txt; PMC at 128^3: forward 37.9 -> 36.8 MB per launch = 1.10x the
    // algorithmic bytes, adjoint 75.9 -> 71.0 MB = 1.06x): 128^3 34.2 -> 36.1 k steps/s, the 32 x 256^2 slab of the 8-GPU 256^3
    // problem -- four planes per XCD before, as many halo planes as own ones -- 29.1 -> 34.5 k; neutral at 64^3 .. 160^3; the
    // 256^3 FORWARD loses (66 -> 76 us) while its adjoint gains a little, hence: option 1 = everywhere but forward steps of
    // 8 M points and more, 2 = everywhere
    // Round 5 (profiles/r05_counters_summary.txt, TCC_EA0_RDREQ_128B): at 256^3 the memory-side reads were 1.44-1.48x the
    // algorithmic ones -- the adjoint brick sweep asked the fabric for 579 MB where 403 MB are needed, ALL of the excess on the
    // stencil-read field (2.3x): a region as tall as half the grid streams 2 MB of new lines through a 4 MB L2 per plane group,
    // so the plane neighbours the next group would have found there are gone.  The regions are now sized by what an L2 holds:
    // the three plane groups that share plane neighbours (footprint = rows x row bytes x planes per brick x arrays touched) must
    // fit L2_KEEP; among the maps that do, the smallest halo share wins; strips of 1/8 of the rows (ny = 8) are a candidate too.
    const bool split = p.opt.brick_xcd == 2 || (p.opt.brick_xcd == 1 && (adjoint || p.n < (int64_t(1) << 23) || p.opt.brick_xny >= 0));
    if (split && b.nblk % pi::NXCD == 0 && p.opt.brick_xny != 1) {
        const long rows_per_brick = std::max<long>(1, nt / std::max(1, b.cpr));
        const double row_bytes = 16.0 * b.cpr;
        const double arrays = adjoint ? 8.0 : 4.0;                           // species-planes streamed per output plane
        constexpr double L2_KEEP = 2.5 * 1024 * 1024;                        // of the 4 MiB per XCD
        auto footprint = [&](double rows) { return 3.0 * rows * row_bytes * rz * arrays; };
        // the contiguous map: an XCD takes npg / 8 plane groups of ALL rows
        const double all_rows = (double)b.nrg * rows_per_brick;
        double best = 4.0 * rz / (double)std::max<long>(1, (npg / pi::NXCD) * rz);
        bool best_fits = footprint(all_rows) <= L2_KEEP;
        for (int ny : {2, 4, 8}) {
            if (p.opt.brick_xny > 1 && ny != p.opt.brick_xny) continue;
            const int nz = pi::NXCD / ny;
            if (npg % nz || b.nrg % ny) continue;
            // a launch with a fused peer put attached sends BOTH faces: Brick::locate computes them first only when the planes are
            // split between at least two regions (its upper half walks top-down); strips of 1/8 of the rows (nz = 1) would store the
            // top face last and expose that put behind the whole launch
            if (faces_first && nz < 2 && p.opt.brick_xny <= 1) continue;
            const double planes = (double)(npg / nz) * rz, rows = (double)(b.nrg / ny) * rows_per_brick;
            const double share = 4.0 / planes + 4.0 / rows;
            const bool fits = footprint(rows) <= L2_KEEP;
            const bool take = p.opt.brick_xny > 1 || (fits && !best_fits) || (fits == best_fits && (fits ? share < best + 1e-9 : true));   // (ties: the narrower strips)
            if (take) { best = share; best_fits = fits; b.xny = ny; b.xpg = (int)(npg / nz); b.xrg = b.nrg / ny; }
        }
        if (b.xny) b.dxrg = make_fastdiv((unsigned)b.xrg);
    }
    return b;
}

// what a fused put (pi_peer.h) of a slab step needs besides the transfer itself: the two faces as plane ranges of the PADDED
// local array; launch_brick_* turn them into the kernel's own plane numbering and count the bricks that hold them
struct FusedPut {
    pi::PeerXfer x;
    bool vec16;
    int face_lo[2], face_hi[2];          // padded plane indices [lo, hi) of the faces sent to next / prev
    unsigned long long timeout_ticks;
};

template <int RZ>
pi::PeerPutFused fused_put_args(const FusedPut* fp, const Problem& p, const pi::BrickGeom& b)
{
    pi::PeerPutFused f{};
    if (!fp) return f;
    f.x = fp->x;
    f.vec16 = fp->vec16 ? 1 : 0;
    f.timeout_ticks = fp->timeout_ticks;
    // a few workgroups per direction: what they feed is one xGMI link (~64 GB/s each way), not HBM; 2 x blocks is a multiple
    // of 8 (XCD placement of the bricks behind them)
    f.x.blocks_per_dir = std::min(p.opt.slab_put_blocks, (fp->x.blocks_per_dir + 3) / 4 * 4);
    f.nput = 2 * f.x.blocks_per_dir;
    const int first = p.lo >= 0 ? p.lo : p.skip + 2;                     // padded index of the kernel's plane 0
    for (int d = 0; d < 2; ++d) {
        f.lo[d] = std::max(0, fp->face_lo[d] - first);
        f.hi[d] = std::min(b.n0, fp->face_hi[d] - first);
        const int groups = f.hi[d] > f.lo[d] ? (f.hi[d] - 1) / RZ - f.lo[d] / RZ + 1 : 0;
        f.expect[d] = (unsigned)(groups * b.nrg);
    }
    return f;
}

template <typename T, int HC, int RZ, int NT = pi::BRICK_NT>
hipError_t launch_brick_fwd(const T* h, T* out, const T* P, const Problem& p, hipStream_t st, const FusedPut* fp = nullptr)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    pi::BrickGeom b = make_brick_geom(p, VEC, RZ, NT, false, fp != nullptr);
    if (b.n0 <= 0) return hipSuccess;
    const size_t lds = (size_t)2 * RZ * pi::brick_wb(NT) + (size_t)p.opt.lds_pad;
    const pi::PeerPutFused put = fused_put_args<RZ>(fp, p, b);
    if (fp) b.wt = 1;                                                    // the put workgroups read what the bricks wrote THROUGH
    auto* k = pi::pi_fwd3d_brick_kernel<T, HC, RZ, NT>;
    if (hipError_t e = allow_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(b.nblk + (unsigned)put.nput), dim3(NT), lds, st, h, out, P, b, p.hc, put);
    return hipGetLastError();
}

template <typename T>
hipError_t brick_fwd(int rz, const T* h, T* out, const T* P, const Problem& p, hipStream_t st, const FusedPut* fp = nullptr)
{
    if (p.hc == 0) {
        if (rz <= 2 && brick_nt_for(p, 16 / (int)sizeof(T), false) == 512)
            return rz == 2 ? launch_brick_fwd<T, pi::POLY, 2, 512>(h, out, P, p, st, fp) : launch_brick_fwd<T, pi::POLY, 1, 512>(h, out, P, p, st, fp);
        if (rz == 4) return launch_brick_fwd<T, pi::POLY, 4>(h, out, P, p, st, fp);
        if (rz == 2) return launch_brick_fwd<T, pi::POLY, 2>(h, out, P, p, st, fp);
        return launch_brick_fwd<T, pi::POLY, 1>(h, out, P, p, st, fp);
    }
    switch (p.hc) {
        case 2:  return launch_brick_fwd<T, 2, 1>(h, out, P, p, st, fp);
        case 4:  return launch_brick_fwd<T, 4, 1>(h, out, P, p, st, fp);
        case 8:  return launch_brick_fwd<T, 8, 1>(h, out, P, p, st, fp);
        default: return launch_brick_fwd<T, 0, 1>(h, out, P, p, st, fp);
    }
}

// The adjoint brick kernel pays a per-workgroup tail (block reduction of the 22 sums, ~1.4 us of a 5 us pass at 128^3): a
// grid of what is resident at once (4 workgroups per CU with one-plane bricks, 2 with more) walks the bricks instead, every
// workgroup the same number of them where the count allows.  Probe (128^3, us per adjoint step): 2048 one-brick workgroups
// 17.4, 1024 two-brick ones 17.2, 768 (uneven) 17.8; two-plane bricks 1024 / 512 workgroups 18.1 / 17.0.
unsigned brick_bwd_grid(const Problem& p, int vec, int rz)
{
    const int elem = 16 / vec;                              // bricks run on 16-byte lanes: 4 = float32, 8 = float64
    const int nt = rz <= 2 ? brick_nt_for(p, vec) : 256;
    const pi::BrickGeom b = make_brick_geom(p, vec, rz, nt);
    static int cu_count[16] = {};                           // per device, asked once (benign race: same value)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) {
        if (!cu_count[dev]) {
            int n = 0;
            cu_count[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        }
        cus = cu_count[dev];
    }

    int per_cu = p.opt.brick_wgs ? p.opt.brick_wgs
                                 : ((rz == 1 && p.hc == 0 && p.loss.mode != 2 && elem == 4) ? 4 : (rz == 1 ? 3 : 2));
    if (nt == 512 && !p.opt.brick_wgs) per_cu = (per_cu + 1) / 2;      // the same waves per CU in half as many workgroups
    unsigned cap = (unsigned)(cus * per_cu);
    if (cap > (unsigned)MAX_BWD_BLOCKS) cap = MAX_BWD_BLOCKS;
    if (b.nblk <= cap) return b.nblk;
    // ... while that is at most two bricks each: a workgroup takes its bricks one after the other, nothing of the next one is
    // in flight while it computes (200^3, one-plane bricks: 4000 two-brick workgroups 59.3 us per step, 1000 eight-brick ones 69.6)
    if (!p.opt.brick_wgs && b.nblk > 2 * cap) cap = MAX_BWD_BLOCKS;
    const unsigned k = (b.nblk + cap - 1) / cap;            // bricks per workgroup
    return (b.nblk + k - 1) / k;
}

template <typename T, int HC, int RZ, bool MOM, int NT = pi::BRICK_NT>
hipError_t launch_brick_bwd(const T* h, const T* G, const T* inj, T* Gp, double* partials, const T* P, const Problem& p,
                            hipStream_t st, const FusedPut* fp = nullptr)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    pi::BrickGeom b = make_brick_geom(p, VEC, RZ, NT, true, fp != nullptr);
    if (b.n0 <= 0) return hipSuccess;
    const unsigned grid = brick_bwd_grid(p, VEC, RZ);
    const size_t head = (size_t)(NT / pi::WAVE) * 2 * sizeof(double);
    const size_t windows = (size_t)2 * RZ * pi::brick_wb(NT), scratch = MOM ? (size_t)(32 + 20 * (NT + 8)) * sizeof(T) : 0;
    const size_t lds = head + (windows > scratch ? windows : scratch) + (size_t)p.opt.lds_pad;
    if constexpr (HC == pi::POLY && RZ == 1 && NT == pi::BRICK_NT) {
        if (fp && p.loss.mode == 0) {                        // slab sweep over the mailboxes: faces put by this launch
            const pi::PeerPutFused put = fused_put_args<RZ>(fp, p, b);
            b.wt = 1;                                        // the put workgroups read what the bricks wrote THROUGH
            auto* k = pi::pi_adj3d_brick_kernel<T, HC, RZ, MOM, 0, NT, true>;
            if (hipError_t e = allow_lds(k, lds)) return e;
            hipLaunchKernelGGL(k, dim3(grid + (unsigned)put.nput), dim3(NT), lds, st, h, G, inj, Gp, partials, P, b, p.hc, put);
            return hipGetLastError();
        }
    }
    if (fp) return hipErrorInvalidValue;                     // (the caller checks brick_bwd_can_put first)
    if constexpr (NT != pi::BRICK_NT) {
        auto* k = pi::pi_adj3d_brick_kernel<T, HC, RZ, MOM, 0, NT>;
        if (hipError_t e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, h, G, inj, Gp, partials, P, b, p.hc, pi::NoPut{});
        return hipGetLastError();
    }
    if (p.loss.mode == 1) {
        auto* k = pi::pi_adj3d_brick_kernel<T, HC, RZ, MOM, 1>;
        if (hipError_t e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(pi::BRICK_NT), lds, st, h, G, inj, Gp, partials, P, b, p.hc, pi::NoPut{});
        return hipGetLastError();
    }
    if (p.loss.mode == 2) {
        auto* k = pi::pi_adj3d_brick_kernel<T, HC, RZ, MOM, 2>;
        if (hipError_t e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(pi::BRICK_NT), lds, st, h, G, inj, Gp, partials, P, b, p.hc, pi::NoPut{});
        return hipGetLastError();
    }
    auto* k = pi::pi_adj3d_brick_kernel<T, HC, RZ, MOM>;
    if (hipError_t e = allow_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(pi::BRICK_NT), lds, st, h, G, inj, Gp, partials, P, b, p.hc, pi::NoPut{});
    return hipGetLastError();
}

// mom: all gradients of a pre-contracted block in the sweep launch; else adjoint state + diffusion-coefficient sums
// the flavours that exist with the put inside: pre-contracted blocks, one-plane bricks of 256 lanes, plain injection
template <typename T>
bool brick_bwd_can_put(int rz, const Problem& p)
{
    return p.hc == 0 && rz == 1 && p.loss.mode == 0 && brick_nt_for(p, 16 / (int)sizeof(T)) != 512;
}

template <typename T>
hipError_t brick_bwd(int rz, bool mom, const T* h, const T* G, const T* inj, T* Gp, double* partials, const T* P,
                     const Problem& p, hipStream_t st, const FusedPut* fp = nullptr)
{
    if (fp) {
        if (!brick_bwd_can_put<T>(rz, p)) return hipErrorInvalidValue;
        return mom ? launch_brick_bwd<T, pi::POLY, 1, true>(h, G, inj, Gp, partials, P, p, st, fp)
                   : launch_brick_bwd<T, pi::POLY, 1, false>(h, G, inj, Gp, partials, P, p, st, fp);
    }
#define CALL_BB(HC, RZ, MOM) launch_brick_bwd<T, HC, RZ, MOM>(h, G, inj, Gp, partials, P, p, st)
    if (p.hc == 0) {
        if (rz <= 2 && brick_nt_for(p, 16 / (int)sizeof(T)) == 512) {
#define CALL_BB5(RZ, MOM) launch_brick_bwd<T, pi::POLY, RZ, MOM, 512>(h, G, inj, Gp, partials, P, p, st)
            if (mom) return rz == 2 ? CALL_BB5(2, true) : CALL_BB5(1, true);
            return rz == 2 ? CALL_BB5(2, false) : CALL_BB5(1, false);
#undef CALL_BB5
        }
        if (mom) return rz == 4 ? CALL_BB(pi::POLY, 4, true) : (rz == 2 ? CALL_BB(pi::POLY, 2, true) : CALL_BB(pi::POLY, 1, true));
        return rz == 4 ? CALL_BB(pi::POLY, 4, false) : (rz == 2 ? CALL_BB(pi::POLY, 2, false) : CALL_BB(pi::POLY, 1, false));
    }
    switch (p.hc) {
        case 2:  return CALL_BB(2, 1, false);
        case 4:  return CALL_BB(4, 1, false);
        case 8:  return CALL_BB(8, 1, false);
        default: return CALL_BB(0, 1, false);
    }
#undef CALL_BB
}

template <typename T>
hipError_t step_fwd(const T* h, T* out, const T* P, const Problem& p, hipStream_t st)
{
    if (p.hc == -1) return adv_fwd<T>(h, out, P, p, st);
    if (const int sv = stream3d_vec<T>(p, {h, out}, false))
        return stream3d<T, false>(sv, h, out, nullptr, nullptr, nullptr, P, p, st, nullptr);
    const int vec = pick_vec<T>(p, {h, out});
    if (const int brz = brick_rz_for<T>(p, vec, false)) return brick_fwd<T>(brz, h, out, P, p, st);
    {
        constexpr int V = pi::vec_width<T>::value;
        const int rz = direct_rz<T>(p, vec, false);
        if (rz == 2) return launch_fwd<T, 3, pi::POLY, V, 2>(h, out, P, p, st);
        if (rz == 4) return launch_fwd<T, 3, pi::POLY, V, 4>(h, out, P, p, st);
    }
#define CALL_FWD(NDIM, HC, VEC) launch_fwd<T, NDIM, HC, VEC>(h, out, P, p, st)
    PI_DISPATCH(CALL_FWD);
#undef CALL_FWD

}

// WGRAD=true: fused single-step adjoint incl. all parameter gradients (step API);
// WGRAD=false: adjoint state + diffusion-coefficient gradients only (rollout sweep)
template <typename T, bool WGRAD>
hipError_t step_bwd(const T* h, const T* G, const T* inj, T* Gp, double* partials, const T* P, const Problem& p,
                    hipStream_t st, unsigned* grid_out)
{
    if (p.hc == -1) return adv_bwd<T>(h, G, inj, Gp, partials, P, p, st, grid_out);   // always fused (tiny grids)
    if (!WGRAD || (p.hc == 0 && sizeof(T) == 4))            // fused flavour of the streaming kernel: float32 poly mode only
        if (const int sv = stream3d_vec<T>(p, {h, G, inj, Gp}, true))
            return stream3d<T, true>(sv, G, Gp, h, inj, partials, P, p, st, grid_out, WGRAD ? 1 : 0);
    const int vec = pick_vec<T>(p, {h, G, inj, Gp});
    if (!WGRAD || p.hc == 0)                                  // factored blocks with all gradients in the launch: direct kernel
        if (const int brz = brick_rz_for<T>(p, vec, true)) {
            if (grid_out) *grid_out = brick_bwd_grid(p, vec, brz);
            return brick_bwd<T>(brz, WGRAD, h, G, inj, Gp, partials, P, p, st);
        }
    const int rz = direct_rz<T>(p, vec, true);
    if (grid_out) *grid_out = bwd_grid(p, vec, sizeof(T), rz);
    {
        constexpr int V = pi::vec_width<T>::value;
        if (rz == 2) return launch_bwd<T, 3, pi::POLY, V, WGRAD, 2>(h, G, inj, Gp, partials, P, p, st);
        if (rz == 4) return launch_bwd<T, 3, pi::POLY, V, WGRAD, 4>(h, G, inj, Gp, partials, P, p, st);
    }
#define CALL_BWD(NDIM, HC, VEC) launch_bwd<T, NDIM, HC, VEC, WGRAD>(h, G, inj, Gp, partials, P, p, st)
    PI_DISPATCH(CALL_BWD);
#undef CALL_BWD
}


// ---- temporally blocked 2D path -----------------------------------------------------------------
constexpr int TILE_B = 32;

// Tile height of the poly-mode tile kernels.  32x32 tiles give a 512^2 grid exactly one workgroup per CU; smaller grids
// leave CUs idle and the launch time is the dependent chain of ONE workgroup, so they get 32x16 tiles (twice the
// workgroups, shorter chain): measured 243 k vs 208 k steps/s on the reference's 100^2 grid, 135 k vs 171 k at 512^2.
// Round 2: the sub-steps are VALU-issue-bound, so what counts is waves per SIMD -- 32x8 tiles with 256 threads are ONE wave
// per SIMD (a 32x16 tile's 308 strips need five waves, i.e. two on one SIMD): while they fit one per CU (<= 256 tiles)
// they win everywhere, 100^2 310 k -> 337 k steps/s, 256^2 279 k -> 304 k, lambda-omega 256^2 194 k -> 220 k; 320^2
// (400 tiles) loses 8 %.
int tile_by_for(const Problem& p)
{
    if (p.hc != 0) return TILE_B;
    if (p.opt.tile_by == 8 || p.opt.tile_by == 16 || p.opt.tile_by == 32) return p.opt.tile_by;
    if (p.opt.tile_k != 4 || p.opt.tile_nt != 512) return TILE_B;
    const int64_t tiles32 = ((p.n0 + TILE_B - 1) / TILE_B) * ((p.W + TILE_B - 1) / TILE_B);
    const int64_t tiles8 = ((p.n0 + 7) / 8) * ((p.W + TILE_B - 1) / TILE_B);
    // Round 5: whole-tile grids of 113 .. 128 tiles take 32 x 32 tiles -- the resident pyramid kernels beat the 16-row ones there
    // (352^2: 262.8 -> 276.3 k steps/s, 256 x 512: 270.1 -> 280.6 k; 288^2 / 320^2 tie: profiles/r05_tile_height_mid_sizes.txt)
    if (tiles8 > 256 && tiles32 > 112 && tiles32 <= 128 && p.n0 % TILE_B == 0 && p.W % TILE_B == 0) return TILE_B;
    return tiles8 <= 256 ? 8 : (tiles32 <= 128 ? 16 : TILE_B);
}

// Wide tiles.  Past 512^2 the 32 x 32 tiles no longer fit one per CU (544^2 = 289 tiles on 256 CUs) and
// the launch takes two rounds of a latency-bound workgroup chain: 512^2 218 k steps/s, 544^2 130 k.  Float32 poly blocks
// therefore grow the TILE while that keeps the grid in one resident round: 32 x 40 tiles on 640 lanes (first sub-step's
// region 44 x 52 = 572 strips), then 40 x 40 on 768 lanes (52 x 52 = 676 strips) -- one strip per lane either way, which is
// what the sweep's operand pipeline is written for, and 10 / 12 waves at <= 168 registers (the fused-moments sweep holds
// 164).  0: 32-wide tiles by tile_by_for; 1: 32 x 40 / 640; 2: 40 x 40 / 768.  Option tile_wide = 0 switches the rule off,
// 1 / 2 force a shape (tests).  Float64 keeps 32 x 32: its fused sweep's [20][lanes] LDS accumulators do not fit next to a
// larger window.
// The two directions choose independently (frames have one layout whatever tile wrote them): the sweep is register-bound
// at one workgroup per CU and gains from both shapes (same box, us per step 32 x 32 -> wide: 544^2 5.43 -> 3.83, 640^2 5.31 ->
// 3.82); the forward is bound by waves per SIMD and sub-step (50 registers: its second round of 32 x 32 tiles is co-resident),
// a 32 x 40 tile's nine wave slots per launch against seven still beat two rounds (544^2 2.47 -> 2.24) but a 40 x 40 tile's do
// not (640^2 2.41 -> 2.91), so the forward stops at 32 x 40 (profiles/r03_tile_cliff.txt).
template <typename T>
int tile_wide_for(const Problem& p, bool adjoint)
{
    if (sizeof(T) != 4 || p.hc != 0 || p.opt.tile_wide == 0 || p.opt.tile_k != 4 || p.opt.tile_nt != 512 || p.opt.tile_by != 0)
        return 0;
    if (p.opt.tile_wide == 1 || p.opt.tile_wide == 2) return p.opt.tile_wide;
    auto tiles = [&](int64_t bx, int64_t by) { return ((p.n0 + by - 1) / by) * ((p.W + bx - 1) / bx); };
    const int cus = 256;
    if (tiles(32, 32) <= cus) return 0;
    if (tiles(32, 40) <= cus) return 1;
    if (adjoint && tiles(40, 40) <= cus) return 2;
    return 0;
}

struct TileShape { int bx, by, nt; };
template <typename T>
TileShape tile_shape_for(const Problem& p, bool adjoint)
{
    switch (tile_wide_for<T>(p, adjoint)) {
        case 1: return {32, 40, 640};
        case 2: return {40, 40, 768};
        default: return {TILE_B, tile_by_for(p), 0};
    }
}
template <typename T>
int64_t tile_count(const Problem& p, bool adjoint)
{
    const TileShape s = tile_shape_for<T>(p, adjoint);
    return ((p.n0 + s.by - 1) / s.by) * ((p.W + s.bx - 1) / s.bx);
}

template <typename T>
bool tile_eligible(const Problem& p, std::initializer_list<const void*> ptrs, bool adjoint)
{
    if (!p.opt.tile || p.opt.vec == 1 || p.ndim != 2 || p.slab) return false;
    if (p.hc != 0 && p.hc != 2 && p.hc != 4 && p.hc != 8) return false;
    // ragged grids (e.g. the reference's 100^2) run with partial edge tiles; the window must not wrap onto itself
    const TileShape shp = tile_shape_for<T>(p, adjoint);
    auto fits = [](int64_t n, int64_t b) { return (n + b - 1) / b * b + 16 <= 2 * n; };        // one wrap per window coordinate
    if (p.W % pi::vec_width<T>::value || !fits(p.n0, shp.by < TILE_B ? TILE_B : shp.by) || !fits(p.W, shp.bx)) return false;
    // the adjoint tile kernel owns one partial row per workgroup: beyond MAX_BWD_BLOCKS tiles the grid-stride
    // direct kernels take over (4096^2 = 16384 tiles)
    if (tile_count<T>(p, true) > MAX_BWD_BLOCKS) return false;
    // temporal blocking pays while launches are latency-bound; beyond, the halo ring's redundant traffic costs more than
    // the launches it saves.  Round 2 (write-through frame stores, cheaper tails; profiles/r02_direct_kernel_option_sweeps.txt,
    // us per step tiles / direct): forward 1024^2 4.5 / 6.4, 1536^2 8.6 / 9.9, 2048^2 14.4 / 14.9; backward 1024^2 11.0 / 12.2,
    // 1280^2 18.0 / 16.4, 2048^2 41.3 / 31.1 -> the forward keeps tiles below 3 M points, the sweep below 1.25 M
    // (tile = 2 forces the tile path)
    if (p.opt.tile == 1 && p.n >= (adjoint ? (5 << 18) : (3 << 20))) return false;
    for (const void* q : ptrs)
        if (q && (reinterpret_cast<uintptr_t>(q) % 16)) return false;
    return true;
}

// XCD-aware tile map (pi::tile_of_block): split the tiles_y x tiles_x tile grid into 8 equal rectangles, as square as
// possible; identity when the counts do not divide (ragged grids, small grids)
pi::TileGeom make_tile_geom(const Problem& p, int by, int bx = TILE_B)
{
    const int tiles_x = (int)((p.W + bx - 1) / bx), tiles_y = (int)((p.n0 + by - 1) / by);
    pi::TileGeom g{(int)p.n0, (int)p.W, (long)p.n, tiles_x, 0, 0, 0, p.loss};
    if (!p.opt.tile_xcd) return g;
    int best = -1;
    for (int rx : {1, 2, 4, 8}) {
        const int ry = pi::NXCD / rx;
        if (tiles_x % rx || tiles_y % ry) continue;
        const int rw = tiles_x / rx, rh = tiles_y / ry;
        const int perim = rw + rh;                          // halo traffic of a rectangle ~ its perimeter
        if (best < 0 || perim < best) { best = perim; g.rx = rx; g.rw = rw; g.rh = rh; }
    }
    return g;
}

template <typename T, int HC, int K, int NT, int BY = TILE_B, int BX = TILE_B>
hipError_t launch_fwd_tile(T* frame_t, const T* P, const Problem& p, hipStream_t st)
{
    using TL = pi::Tile<K, BX, BY>;
    const pi::TileGeom g = make_tile_geom(p, BY, BX);
    const unsigned grid = (unsigned)(((p.n0 + BY - 1) / BY) * g.tiles_x);
    const size_t lds = (size_t)4 * TL::PLANE * sizeof(T) + 32 /* lds_pad0/1 */ + (size_t)p.opt.lds_pad;
    auto* k = pi::pi_fwd2d_tile_kernel<T, HC, K, BX, BY, NT>;
    if (hipError_t e = allow_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, frame_t, (long)(2 * p.n), P, g);
    return hipGetLastError();
}

template <typename T, int HC, int K, int NT, int BY = TILE_B, bool MOM = false, int BX = TILE_B>
hipError_t launch_adj_tile(const T* hframe_t, const T* gframe_t, T* aframe_t, unsigned inj_mask, T* g_h0,
                           int steps_to_zero, double* partials, const T* P, const Problem& p, hipStream_t st)
{
    using TL = pi::Tile<K, BX, BY>;
    const pi::TileGeom g = make_tile_geom(p, BY, BX);
    const unsigned grid = (unsigned)(((p.n0 + BY - 1) / BY) * g.tiles_x);
    size_t lds = (size_t)4 * TL::PLANE * sizeof(T) + 32 /* lds_pad0/1 */;
    if (MOM && sizeof(T) == 4) {                // the tail reduction's scratch (pi_tile2d.h): doubles, then [20][NT + 16] values
        const size_t tail = (size_t)(2 * (NT / pi::WAVE) + 20 + 20 * (NT / pi::WAVE)) * sizeof(double) +
                            (size_t)20 * (NT + 16) * sizeof(T);
        if (tail > lds) lds = tail;
    }
    if (MOM && sizeof(T) == 8)                  // float64: [20][NT] per-lane moment accumulators behind the state buffers
        lds = pi::tile_state_bytes<T, K, BX, BY>() + (size_t)20 * NT * sizeof(double);
    lds += (size_t)p.opt.lds_pad;
    auto* k = pi::pi_adj2d_tile_kernel<T, HC, K, BX, BY, NT, MOM>;
    if (hipError_t e = allow_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, hframe_t, gframe_t, aframe_t, (long)(2 * p.n), inj_mask, g_h0,
                       steps_to_zero, partials, pi::nparams(p.hc), P, g);
    return hipGetLastError();
}

#define PI_TILE_VARIANTS(CALL, HC)                                              \
    do {                                                                        \
        if (p.opt.tile_k == 2) return CALL(HC, 2, 256);                         \
        if constexpr (HC == pi::POLY) {                                         \
            if (p.opt.tile_k == 8) return CALL(HC, 8, 1024);                    \
            if (p.opt.tile_nt == 1024) return CALL(HC, 4, 1024);                \
            if (tile_by_for(p) == 16) return CALL(HC, 4, 320, 16);              \
            if (tile_by_for(p) == 8) return CALL(HC, 4, 256, 8);                \
        }                                                                       \
        if (p.opt.tile_nt == 256) return CALL(HC, 4, 256);                      \
        return CALL(HC, 4, 512);                                                \
    } while (0)
#define PI_TILE_DISPATCH(CALL)                                                  \
    do {                                                                        \
        switch (p.hc) {                                                         \
            case 0: PI_TILE_VARIANTS(CALL, pi::POLY);                           \
            case 2: PI_TILE_VARIANTS(CALL, 2);                                  \
            case 4: PI_TILE_VARIANTS(CALL, 4);                                  \
            default: PI_TILE_VARIANTS(CALL, 8);                                 \
        }                                                                       \
    } while (0)

template <typename T>
hipError_t fwd_tile(T* frame_t, const T* P, const Problem& p, hipStream_t st)
{
    if constexpr (sizeof(T) == 4) {
        switch (tile_wide_for<T>(p, false)) {
            case 1: return launch_fwd_tile<T, pi::POLY, 4, 640, 40, 32>(frame_t, P, p, st);
            case 2: return launch_fwd_tile<T, pi::POLY, 4, 768, 40, 40>(frame_t, P, p, st);
            default: break;
        }
    }
#define CALL_FT(HC, K, NT, ...) launch_fwd_tile<T, HC, K, NT, ##__VA_ARGS__>(frame_t, P, p, st)
    PI_TILE_DISPATCH(CALL_FT);
#undef CALL_FT
}

// the fused-moments flavour exists for one variant: float32 poly blocks, K = 4, 32 x 32 tiles, 512 threads
template <typename T>
bool tile_fuse_ok(const Problem& p)
{
    // measured on MI355X (backward us per step, split -> fused): 384^2 3.37 -> 3.20, 512^2 3.90 -> 3.29, 1000^2 13.4 -> 11.3;
    // in the 16-row-tile regime (<= 128 tiles of 32x32, e.g. the reference's 100^2) the extra VALU work sits on the one
    // critical workgroup chain and loses (2.26 -> 2.66), so it keeps the split schedule
    // float64 (lambda-omega): the per-lane moment sums live in LDS and are updated with ds_add_f64 (pi_tile2d.h) -- a register
    // flavour needed more than the 256 registers of a 512-thread workgroup (57 spilled doubles, 5.98 -> 9.13 us per step,
    // profiles/r02_fp64_fused_tile_sweep.txt); with LDS accumulators: 186 VGPRs, no scratch, lambda-omega 512^2 backward
    // 5.45 -> 4.48 us per step, 1024^2 19.1 -> 15.9 (profiles/r02_fp64_fused_lds_accumulators.txt)
    return p.opt.tile_fuse && !p.opt.skip_wgrad && p.hc == 0 &&
           p.opt.tile_k == 4 && p.opt.tile_nt == 512 && (tile_by_for(p) == TILE_B || tile_wide_for<T>(p, true) != 0);
}

template <typename T>
hipError_t adj_tile(const T* hframe_t, const T* gframe_t, T* aframe_t, unsigned inj_mask, T* g_h0, int steps_to_zero,
                    double* partials, const T* P, const Problem& p, hipStream_t st)
{
    if constexpr (sizeof(T) == 4) {
        const bool fused = tile_fuse_ok<T>(p);
#define CALL_WIDE(NT, BY, BX, MOM) \
    launch_adj_tile<T, pi::POLY, 4, NT, BY, MOM, BX>(hframe_t, gframe_t, aframe_t, inj_mask, g_h0, steps_to_zero, partials, P, p, st)
        switch (tile_wide_for<T>(p, true)) {
            case 1: return fused ? CALL_WIDE(640, 40, 32, true) : CALL_WIDE(640, 40, 32, false);
            case 2: return fused ? CALL_WIDE(768, 40, 40, true) : CALL_WIDE(768, 40, 40, false);
            default: break;
        }
#undef CALL_WIDE
    }
    if (tile_fuse_ok<T>(p))
        return launch_adj_tile<T, pi::POLY, 4, 512, TILE_B, true>(hframe_t, gframe_t, aframe_t, inj_mask, g_h0,
                                                                  steps_to_zero, partials, P, p, st);
#define CALL_AT(HC, K, NT, ...) \
    launch_adj_tile<T, HC, K, NT, ##__VA_ARGS__>(hframe_t, gframe_t, aframe_t, inj_mask, g_h0, steps_to_zero, partials, P, p, st)
    PI_TILE_DISPATCH(CALL_AT);
#undef CALL_AT
}

// ---- persistent fused sweep (pi_tile2d.h "PERSISTENT fused sweep") ---------------------------------------------------------
int device_cu_count()
{
    static int cu_count[16] = {};                           // per device, asked once (benign race: same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (!cu_count[dev]) {
        int n = 0;
        cu_count[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : -1;
    }
    return cu_count[dev] > 0 ? cu_count[dev] : 0;
}

bool persist_disabled_here();
constexpr int PERSIST_BAND = 2 * (TILE_B * TILE_B - (TILE_B - 16) * (TILE_B - 16));     // granules per tile and parity (K = 4)
size_t persist_outbox_bytes(const Problem& p, int elem = 4)     // per band value: 8 bytes (float32: 16-byte granules of two values -- or the
                                                                // 8-byte words of the unsplit sweep), 16 bytes (float64: one value per granule)
{
    return (size_t)2 * (size_t)((p.n0 / TILE_B) * (p.W / TILE_B)) * PERSIST_BAND * (elem == 8 ? 16 : 8);
}

// can the tile sweep of this rollout run as one cooperative launch?  (everything the kernel assumes, checked here)
template <typename T>
bool persist_ok(const Problem& p, const unsigned char* mask, int t_top, int ngroups, hipStream_t st)
{
    // (float64 since round 5: the split flavour on 16-byte granules, moments in shared LDS rows; option adj_persist_f64)
    if (!p.opt.tile_persist || ngroups < 2) return false;
    if (sizeof(T) != 4 && (!p.opt.adj_persist_f64 || !p.opt.persist_split)) return false;
    if (persist_disabled_here()) return false;               // a launch aborted on this device (see PersistGuard)
    if (mask && t_top >= 4096) return false;                 // the frame mask travels as a kernel argument (4096 bits)
    if (!tile_fuse_ok<T>(p) || tile_wide_for<T>(p, true) != 0 || tile_by_for(p) != TILE_B) return false;
    if (p.n0 % TILE_B || p.W % TILE_B) return false;
    const int64_t tiles = (p.n0 / TILE_B) * (p.W / TILE_B);
    const int cus = device_cu_count();
    if (tiles < 16 || cus <= 0 || tiles > cus) return false;          // (tiny grids: the 8-row tile kernels are faster anyway)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
    return true;
}

// Two persistent sweeps in flight on ONE device would each hold part of the CUs and wait for workgroups that cannot become
// resident.  Within a process: the last persistent launch per device leaves an event behind; a call on ANOTHER stream while that
// event is not ready takes the launch-per-group path (same stream: ordered behind it anyway).
// Anything ELSE that keeps workgroups from becoming resident -- another process on the GPU, a CU mask (HSA_CU_MASK, a CU-masked
// stream), a long kernel on another stream -- is caught by the launch itself: its workgroups give up within a bound, write
// nothing, and say so in a host-mapped status slot (pi_tile2d.h, PersistArgs::host).  With the handshake (default) the entry
// point waits for that word -- "all resident" normally arrives a few microseconds after the kernel starts -- and enqueues the
// launch-per-group sweep itself on an abort; the device then keeps the launch-per-group path until persist_reset.
constexpr int PERSIST_SLOTS = 32;
struct PersistHost {                                      // host-mapped (hipHostMalloc): written by the device, read here
    volatile int slot[PERSIST_SLOTS][4];                  // {roll call complete, group, tile, aborted}: see PersistArgs::host
};
struct PersistGuard {
    std::mutex mu;
    hipEvent_t ev[16] = {};
    hipStream_t st[16] = {};
    bool armed[16] = {};
    bool disabled[16] = {};                               // a launch aborted on this device: launch-per-group until persist_reset
    PersistHost* host = nullptr;
    bool host_tried = false;
    bool watch[PERSIST_SLOTS] = {};                       // slots of launches nobody has looked at since (no-handshake mode)
    int next_slot = 0;
    long launches = 0, aborts = 0;
    int last_group = -1, last_tile = -1;
    bool warned = false;
    void* fwd_scratch[16] = {};                           // per device: sync words + granule outbox of the resident forward
    size_t fwd_scratch_bytes[16] = {};
};
PersistGuard g_persist;

bool persist_disabled_here()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return true;
    std::lock_guard<std::mutex> lk(g_persist.mu);
    return g_persist.disabled[dev];
}

void persist_reset()
{
    std::lock_guard<std::mutex> lk(g_persist.mu);
    for (bool& d : g_persist.disabled) d = false;
    for (bool& w : g_persist.watch) w = false;
    g_persist.warned = false;
}

// the launches of no-handshake callers: has one of them aborted since anybody looked?  (reads of cached host memory)
int persist_async_error()
{
    if (!g_persist.host) return 0;
    std::lock_guard<std::mutex> lk(g_persist.mu);
    int rc = 0;
    for (int i = 0; i < PERSIST_SLOTS; ++i)
        if (g_persist.watch[i] && g_persist.host->slot[i][3] != 0) {
            g_persist.watch[i] = false;
            g_persist.last_group = g_persist.host->slot[i][1];
            g_persist.last_tile = g_persist.host->slot[i][2];
            ++g_persist.aborts;
            for (bool& d : g_persist.disabled) d = true;
            rc = PERCNN_PI_EASYNC;
        }
    return rc;
}

bool persist_enter(hipStream_t st, int& dev)
{
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    std::lock_guard<std::mutex> lk(g_persist.mu);
    if (g_persist.disabled[dev]) return false;
    if (g_persist.armed[dev] && g_persist.st[dev] != st && hipEventQuery(g_persist.ev[dev]) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (!g_persist.host_tried) {
        g_persist.host_tried = true;
        void* hp = nullptr;
        if (hipHostMalloc(&hp, sizeof(PersistHost), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hp) {
            std::memset(hp, 0, sizeof(PersistHost));
            g_persist.host = static_cast<PersistHost*>(hp);
        } else (void)hipGetLastError();
    }
    return g_persist.host != nullptr;                      // no status word, no persistent launch
}
void persist_leave(hipStream_t st, int dev)
{
    std::lock_guard<std::mutex> lk(g_persist.mu);
    if (!g_persist.ev[dev] && hipEventCreateWithFlags(&g_persist.ev[dev], hipEventDisableTiming) != hipSuccess) return;
    if (hipEventRecord(g_persist.ev[dev], st) == hipSuccess) { g_persist.st[dev] = st; g_persist.armed[dev] = true; }
}

// ---- RESIDENT 3D reverse sweep (pi_res3d.h, round 6) ------------------------------------------------------------------
// The whole sweep of a float32 pre-contracted 3D rollout as one launch: one resident workgroup per 16 x 16 x 32 block keeps the
// adjoint state in LDS, reads only h_{t-1} and the injected dL/dtraj_{t-1} per step, hands its two-deep faces to the six
// neighbours as data-tagged granules and carries the 22 gradient sums; only dL/dh0 and one partial row per block are written.
// Same-box A/B at 128^3 (profiles/r06_resident3d_ab.txt): 16.4 us per step against the brick sweep's 19.0; dL/dh0 bit-identical.
struct Res3dPlan { int gz, gy, gx, rz, ry, rx, nblk; };
bool res3d_plan(const Problem& p, int nsteps, size_t elem, Res3dPlan& pl)
{
    using namespace pi::r3d;
    if (elem != 4 || p.ndim != 3 || p.hc != 0 || p.slab || p.loss.mode != 0 || !p.opt.res3d || !p.opt.tile_persist) return false;
    if (nsteps < 16 || nsteps >= 4096) return false;                     // (the frame bitset of AdjArgs; short sweeps: bricks)
    if (persist_disabled_here()) return false;                           // a resident launch aborted on this device (persist_reset re-arms)
    if (p.n0 % BZ || p.n1 % BY || p.W % BX || p.n > (int64_t(1) << 28)) return false;
    pl.gz = (int)(p.n0 / BZ); pl.gy = (int)(p.n1 / BY); pl.gx = (int)(p.W / BX);
    pl.nblk = pl.gz * pl.gy * pl.gx;
    if (pl.gz < 2 || pl.gy < 2 || pl.gx < 2) return false;              // (a block that is its own neighbour: never exercised)
    const int cus = device_cu_count();
    if (cus <= 0 || pl.nblk > cus) return false;
    // the resident sweep takes ~15 us per step whatever the block count (a block's step + one hand-over); the bricks scale with the
    // points: 192 blocks 14.5 (bricks) vs 14.9, 224 blocks 16.9 vs 15.2, 256 blocks 17.3 vs 14.8 (profiles/r06_forward_128_options.txt)
    if (p.opt.res3d == 1 && pl.nblk * 8 < cus * 7) return false;         // fewer than 7/8 of the CUs busy: the brick sweep wins
    // XCD regions (workgroup b runs on XCD b % 8 and takes a block of region b % 8): the split of 8 = rz * ry * rx that divides
    // the block counts with the fewest faces between regions; none divides: linear order, every face written through
    pl.rz = pl.ry = pl.rx = 1;
    long best = -1;
    for (int rz : {1, 2, 4, 8})
        for (int ry : {1, 2, 4, 8}) {
            if (rz * ry > 8) continue;
            const int rx = 8 / (rz * ry);
            if (pl.gz % rz || pl.gy % ry || pl.gx % rx) continue;
            const long dz = pl.gz / rz, dy = pl.gy / ry, dx = pl.gx / rx;
            // face area (in points) a region exposes per axis it is cut along
            const long cut = (rz > 1 ? dy * dx * BY * BX : 0) + (ry > 1 ? dz * dx * BZ * BX : 0) + (rx > 1 ? dz * dy * BZ * BY : 0);
            if (best < 0 || cut < best) { best = cut; pl.rz = rz; pl.ry = ry; pl.rx = rx; }
        }
    return true;
}

// hipSuccess: the sweep t_top -> 0 has been enqueued (dL/dh0 and nblk partial rows); hipErrorLaunchFailure: it ran and aborted
// (nothing it was asked for is valid: zero the partial rows and take the launch-per-step path); hipErrorLaunchTimeOut: fatal;
// anything else: nothing was launched, take the launch-per-step path
hipError_t launch_adj_res3d(const float* traj, const float* g_traj, const unsigned char* mask, const float* g_top, float* g_h0,
                            int t_top, double* partials, const float* P, const Problem& p, const Res3dPlan& pl, hipStream_t st)
{
    using namespace pi::r3d;
    constexpr int NT = 512;
    auto* k = pi_adj3d_resident_kernel<NT>;
    if (hipError_t e = allow_lds(k, (size_t)LDS_BYTES)) return e;
    static int blocks_per_cu[16] = {};                      // per device, asked once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return hipErrorNotSupported; }
    if (!blocks_per_cu[dev]) {
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k, NT, (size_t)LDS_BYTES) != hipSuccess || q < 1) { (void)hipGetLastError(); q = -1; }
        blocks_per_cu[dev] = q;
    }
    if (blocks_per_cu[dev] < 1) return hipErrorCooperativeLaunchTooLarge;
    pi_host::Resident r;
    const size_t outbox_bytes = (size_t)2 * (size_t)pl.nblk * Shape<NT>::BOX_BYTES;
    if (hipError_t e = pi_host::resident_begin(st, outbox_bytes, r)) return e == hipErrorLaunchFailure ? hipErrorNotSupported : e;
    Args a{};
    a.n0 = (int)p.n0; a.n1 = (int)p.n1; a.n2 = (int)p.W;
    a.gz = pl.gz; a.gy = pl.gy; a.gx = pl.gx;
    a.rz = pl.rz; a.ry = pl.ry; a.rx = pl.rx;
    a.ss = (long)p.n; a.frame_stride = (long)(2 * p.n);
    a.outbox = r.scratch + 256;
    a.sync = reinterpret_cast<unsigned*>(r.scratch);
    a.host = const_cast<int*>(r.hs);
    a.nsteps = t_top;
    a.skip = 0; a.pause = 0;
    a.timeout_ticks = r.timeout_ticks; a.first_timeout_ticks = r.first_timeout_ticks;
    AdjArgs aa{};
    aa.traj = traj; aa.gtraj = g_traj; aa.gtop = g_top; aa.gout = g_h0; aa.partials = partials;
    aa.np = pi::nparams(p.hc); aa.t_top = t_top;
    for (int f = 0; f < t_top; ++f)
        if (!mask || mask[f]) aa.frames[f >> 5] |= 1u << (f & 31);
    hipLaunchKernelGGL(k, dim3((unsigned)pl.nblk), dim3(NT), (size_t)LDS_BYTES, st, P, a, aa, (unsigned long long*)nullptr);
    if (hipError_t e = hipGetLastError()) return e;
    return pi_host::resident_launched(st, r, (unsigned)pl.nblk, "the resident 3D adjoint sweep");
}

// groups of 4 steps from frame t_top down; g_h0 only if the last group ends at frame 0.  Returns hipErrorCooperativeLaunchTooLarge
// (or whatever the runtime says) WITHOUT having launched anything if the workgroups cannot all be resident, and
// hipErrorLaunchFailure if the launch ran and ABORTED (handshake mode; nothing it was asked for has been written): the caller
// then runs the launch-per-group path.  `sync`: three zeroed device words (PersistArgs::sync).
template <typename T>
hipError_t launch_adj_persist(const T* hframe_t, const T* gframe_t, T* aframe_t, T* g_h0, int t_top, const unsigned char* mask,
                              int ngroups, double* partials, unsigned long long* outbox, unsigned* sync, const T* P,
                              const Problem& p, int dev, hipStream_t st)
{
    constexpr int K = 4, NT = 512;
    using TL = pi::Tile<K, TILE_B, TILE_B>;
    pi::TileGeom g = make_tile_geom(p, TILE_B);
    const unsigned grid = (unsigned)((p.n0 / TILE_B) * g.tiles_x);
    (void)sizeof(TL);
    // state buffers | [20][NT] double moment sums per lane | publish / gather tables (3 + 5 + 5 ints per lane) and, split sweep,
    // 6 of strip geometry | abort word
    constexpr int LACC = sizeof(T) == 8 ? NT / 2 : NT;      // rows of the moment accumulators (pi_tile2d.h)
    const size_t lds = pi::tile_state_bytes<T, K, TILE_B, TILE_B>() + (size_t)20 * LACC * sizeof(double) +
                       (size_t)(p.opt.persist_split ? pi::PERSIST_SPLIT_TABLE_ROWS : 13) * NT * sizeof(int) + 16;
    auto* k = pi::pi_adj2d_persist_split_kernel<T, K, TILE_B, TILE_B, NT>;
    if constexpr (sizeof(T) == 4) {
        if (!p.opt.persist_split) k = pi::pi_adj2d_persist_kernel<T, K, TILE_B, TILE_B, NT>;
    }
    if (hipError_t e = allow_lds(k, lds)) return e;
    static int resident[16][2] = {};                        // per device and flavour (per value type: a template): one workgroup per CU?
    int& res = resident[dev][p.opt.persist_split ? 1 : 0];
    if (!res) {
        int nb = 0;
        res = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, NT, lds) == hipSuccess && nb >= 1) ? 1 : -1;
    }
    if (res < 0) return hipErrorCooperativeLaunchTooLarge;
    if (hipError_t e = hipMemsetAsync(outbox, 0, persist_outbox_bytes(p, (int)sizeof(T)), st)) return e;
    long frame_stride = (long)(2 * p.n);
    int np = pi::nparams(p.hc);
    pi::PersistArgs pa{};
    int slot;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        slot = g_persist.next_slot++ % PERSIST_SLOTS;
        g_persist.watch[slot] = false;
        ++g_persist.launches;
    }
    volatile int* hs = g_persist.host->slot[slot];
    hs[0] = 0; hs[1] = -1; hs[2] = -1; hs[3] = 0;
    pa.outbox = outbox; pa.sync = sync; pa.ngroups = ngroups;
    pa.host = const_cast<int*>(hs);
    pa.timeout_ticks = (unsigned long long)p.opt.persist_timeout_ms * 100000ull;             // 100 MHz clock
    pa.first_timeout_ticks = (unsigned long long)p.opt.persist_first_timeout_ms * 100000ull;
    pa.t_top = t_top;
    pa.masked = mask ? 1 : 0;
    if (mask)
        for (int t = 0; t < t_top && t < 4096; ++t)
            if (mask[t]) pa.frames[t >> 5] |= 1u << (t & 31);
    void* args[] = {(void*)&hframe_t, (void*)&gframe_t, (void*)&aframe_t, (void*)&frame_stride, (void*)&g_h0, (void*)&partials,
                    (void*)&np, (void*)&P, (void*)&g, (void*)&pa};
    hipError_t e;
    if (p.opt.tile_persist == 2) {                          // plain launch: one workgroup per CU fits (residency: see above)
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, hframe_t, gframe_t, aframe_t, frame_stride, g_h0, partials, np, P, g, pa);
        e = hipGetLastError();
    } else {
        e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k), dim3(grid), dim3(NT), args, (unsigned)lds, st);
    }
    if (e != hipSuccess) return e;
    {   // whoever launched it: an abort nobody has dealt with surfaces at the next entry point (PERCNN_PI_EASYNC)
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[slot] = true;
    }
    if (!p.opt.persist_handshake) return hipSuccess;        // fire and forget
    // wait for the roll call (not for the sweep): normally a few microseconds after the kernel starts, i.e. this returns when the
    // stream has reached the sweep.  The bound only guards against a device that never runs the launch at all.
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (hs[0] == 0 && hs[3] == 0) {
        if ((++spins & 0x3ff) == 0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(600)) return hipErrorLaunchTimeOut;
            std::this_thread::yield();
        }
    }
    if (hs[3] != 0) {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[slot] = false;                      // dealt with here: the caller re-runs the sweep launch by launch
        ++g_persist.aborts;
        g_persist.last_group = hs[1];
        g_persist.last_tile = hs[2];
        g_persist.disabled[dev] = true;
        if (!g_persist.warned) {
            g_persist.warned = true;
            std::fprintf(stderr, "percnn_pi: the persistent tile sweep could not keep all %u workgroups resident on device %d "
                                 "(group %d, tile %d: another process / kernel holds CUs, or a CU mask is set); using one launch "
                                 "per group of steps from now on (percnn_pi_set_option(\"persist_reset\", 1) re-arms it)\n",
                         grid, dev, (int)hs[1], (int)hs[2]);
        }
        return hipErrorLaunchFailure;
    }
    return hipSuccess;
}

// ---- resident FORWARD (pi_fwd2d_persist_kernel): the same grids, the same residency protocol ----------------------------------
template <typename T>
bool fwd_persist_ok(const Problem& p, int ngroups, hipStream_t st)
{
    // (eight groups at least: short rollouts -- the step loop's speculative groups of 8 / 16 steps among them -- keep the
    // launch-per-group kernel and with it a host that never waits for the stream)
    // (float64 since round 5 -- lambda-omega, BASELINE configs[2] -- on 16-byte granules; option fwd_persist_f64)
    if (!p.opt.tile_persist || !p.opt.fwd_persist || (sizeof(T) != 4 && !p.opt.fwd_persist_f64) || ngroups < 8) return false;
    if (persist_disabled_here()) return false;
    // what the launch-per-group path would run as pi_fwd2d_tile_kernel<float, poly, 4, 32, 32, 512> (the trajectory is that kernel's)
    if (p.hc != 0 || p.opt.tile_k != 4 || p.opt.tile_nt != 512 || tile_by_for(p) != TILE_B || tile_wide_for<T>(p, false) != 0) return false;
    if (p.n0 % TILE_B || p.W % TILE_B) return false;
    const int64_t tiles = (p.n0 / TILE_B) * (p.W / TILE_B);
    const int cus = device_cu_count();
    // (its 77 KB workgroups fit two per CU: up to 2 x #CUs tiles, option fwd_persist_per_cu; the launch asks the runtime)
    if (tiles < 16 || cus <= 0 || tiles > (int64_t)cus * p.opt.fwd_persist_per_cu) return false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) return false;
    return true;
}

// frames t0 + 1 .. t0 + 4 * ngroups from frame t0.  Return values as launch_adj_persist: hipErrorLaunchFailure = it ran and
// ABORTED (frames may be partly written: the caller recomputes them launch by launch, which is deterministic).
template <typename T>
hipError_t launch_fwd_persist(T* frame_t0, int ngroups, const T* P, const Problem& p, int dev, hipStream_t st)
{
    constexpr int K = 4, NT = 512;
    pi::TileGeom g = make_tile_geom(p, TILE_B);
    const unsigned grid = (unsigned)((p.n0 / TILE_B) * g.tiles_x);
    // state buffers | publish / gather tables and 6 rows of strip geometry | abort word
    const size_t lds = pi::tile_state_bytes<T, K, TILE_B, TILE_B>() + (size_t)pi::PERSIST_SPLIT_TABLE_ROWS * NT * sizeof(int) + 16;
    // one workgroup per CU: the flavour that holds the parameter block in vector registers (142 of them); a grid that needs two
    // per CU (option fwd_persist_per_cu = 2): the one that holds it in scalar registers (83 vector registers)
    const int two = (int64_t)grid > (int64_t)device_cu_count() ? 1 : 0;
    auto* k = two ? pi::pi_fwd2d_persist_kernel<T, K, TILE_B, TILE_B, NT, 2> : pi::pi_fwd2d_persist_kernel<T, K, TILE_B, TILE_B, NT, 1>;
    if (hipError_t e = allow_lds(k, lds)) return e;
    static int resident[16][2] = {};                        // per device and flavour: workgroups that fit a CU (asked once; -1: none)
    if (!resident[dev][two]) {
        int nb = 0;
        resident[dev][two] = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, NT, lds) == hipSuccess && nb >= 1) ? nb : -1;
    }
    if (resident[dev][two] < 0 || (int64_t)grid > (int64_t)device_cu_count() * resident[dev][two]) return hipErrorCooperativeLaunchTooLarge;
    // per-device scratch: 256 B of sync words | granule outbox (allocated once, sized for this grid or larger)
    const size_t need = 256 + persist_outbox_bytes(p, (int)sizeof(T));
    unsigned char* scratch;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        if (g_persist.fwd_scratch_bytes[dev] < need) {
            if (g_persist.fwd_scratch[dev]) {               // (a launch that still uses the old one is ordered before this call's
                (void)hipFree(g_persist.fwd_scratch[dev]);  // work only on its own stream: hipFree synchronises the device)
                g_persist.fwd_scratch[dev] = nullptr;
                g_persist.fwd_scratch_bytes[dev] = 0;
            }
            void* q = nullptr;
            if (hipMalloc(&q, need) != hipSuccess || !q) { (void)hipGetLastError(); return hipErrorOutOfMemory; }
            g_persist.fwd_scratch[dev] = q;
            g_persist.fwd_scratch_bytes[dev] = need;
        }
        scratch = static_cast<unsigned char*>(g_persist.fwd_scratch[dev]);
    }
    if (hipError_t e = hipMemsetAsync(scratch, 0, need, st)) return e;
    long frame_stride = (long)(2 * p.n);
    pi::PersistArgs pa{};
    int slot;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        slot = g_persist.next_slot++ % PERSIST_SLOTS;
        g_persist.watch[slot] = false;
        ++g_persist.launches;
    }
    volatile int* hs = g_persist.host->slot[slot];
    hs[0] = 0; hs[1] = -1; hs[2] = -1; hs[3] = 0;
    pa.outbox = reinterpret_cast<unsigned long long*>(scratch + 256);
    pa.sync = reinterpret_cast<unsigned*>(scratch);
    pa.ngroups = ngroups;
    pa.host = const_cast<int*>(hs);
    pa.timeout_ticks = (unsigned long long)p.opt.persist_timeout_ms * 100000ull;             // 100 MHz clock
    pa.first_timeout_ticks = (unsigned long long)p.opt.persist_first_timeout_ms * 100000ull;
    void* args[] = {(void*)&frame_t0, (void*)&frame_stride, (void*)&P, (void*)&g, (void*)&pa};
    hipError_t e;
    if (p.opt.tile_persist == 2) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, frame_t0, frame_stride, P, g, pa);
        e = hipGetLastError();
    } else {
        e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k), dim3(grid), dim3(NT), args, (unsigned)lds, st);
    }
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[slot] = true;
    }
    if (!p.opt.persist_handshake) return hipSuccess;        // fire and forget (an abort: PERCNN_PI_EASYNC at the next entry point)
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (hs[0] == 0 && hs[3] == 0) {
        if ((++spins & 0x3ff) == 0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(600)) return hipErrorLaunchTimeOut;
            std::this_thread::yield();
        }
    }
    if (hs[3] != 0) {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[slot] = false;
        ++g_persist.aborts;
        g_persist.last_group = hs[1];
        g_persist.last_tile = hs[2];
        g_persist.disabled[dev] = true;
        if (!g_persist.warned) {
            g_persist.warned = true;
            std::fprintf(stderr, "percnn_pi: the resident forward rollout could not keep all %u workgroups resident on device %d "
                                 "(group %d, tile %d: another process / kernel holds CUs, or a CU mask is set); using one launch "
                                 "per group of steps from now on (percnn_pi_set_option(\"persist_reset\", 1) re-arms it)\n",
                         grid, dev, (int)hs[1], (int)hs[2]);
        }
        return hipErrorLaunchFailure;
    }
    return hipSuccess;
}

// ---- small-tile persistent sweep (pi_adj2d_persist_small_kernel): the 32 x 8 / 256-lane regime with the split schedule --------
// granule outbox of the small-tile persistent sweep (pi_adj2d_persist_small_kernel): two parities x tiles x the 2 x 32 x 8 values
// a tile publishes; behind the partial rows (every adjoint frame is in use in the split schedule it serves)
// (the 32 x 8 / 256-lane and the 32 x 16 / 320-lane regimes: whatever tile_by_for picks below 32 rows)
int64_t persist_small_tiles(const Problem& p, int by) { return ((p.n0 + by - 1) / by) * ((p.W + TILE_B - 1) / TILE_B); }
size_t persist_small_outbox_bytes(const Problem& p, int elem)
{
    if (p.ndim != 2 || p.hc != 0 || elem != 4 || p.slab) return 0;
    const int by = tile_by_for(p);
    if (by >= TILE_B || persist_small_tiles(p, by) > 256) return 0;
    return align_up((size_t)2 * (size_t)persist_small_tiles(p, by) * (2 * TILE_B * by) * sizeof(unsigned long long), 256);
}

template <typename T>
bool persist_small_ok(const Problem& p, const unsigned char* mask, int t_top, int ngroups, hipStream_t st)
{
    if (!p.opt.tile_persist || !p.opt.persist_small || sizeof(T) != 4 || ngroups < 2 || p.hc != 0) return false;
    if (persist_disabled_here()) return false;
    if (mask && t_top >= 4096) return false;
    const int by = tile_by_for(p);
    if (p.opt.tile_k != 4 || p.opt.tile_nt != 512 || tile_wide_for<T>(p, true) != 0 || by >= TILE_B || tile_fuse_ok<T>(p)) return false;
    const int64_t tiles = persist_small_tiles(p, by);
    const int cus = device_cu_count();
    if (tiles < 2 || cus <= 0 || tiles > cus || tiles > 256) return false;    // one workgroup per CU at most: resident for sure
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
    return true;
}

// as launch_adj_persist; every adjoint frame of the groups it runs is written
template <typename T, int BY, int NT, bool HALFS = false>
hipError_t launch_adj_persist_small_t(const T* hframe_t, const T* gframe_t, T* aframe_t, T* g_h0, int t_top, const unsigned char* mask,
                                    int ngroups, double* partials, unsigned long long* outbox, unsigned* sync, const T* P,
                                    const Problem& p, int dev, hipStream_t st)
{
    constexpr int K = 4;
    using TL = pi::Tile<K, TILE_B, BY>;
    pi::TileGeom g = make_tile_geom(p, BY);
    const unsigned grid = (unsigned)(((p.n0 + BY - 1) / BY) * g.tiles_x);
    constexpr int RINGH = TL::LX * TL::LY - TILE_B * BY, NGAT = (2 * RINGH + NT - 1) / NT;
    // state buffers | gather tables | K rows of strip geometry | abort word
    const size_t lds = pi::tile_state_bytes<T, K, TILE_B, BY>() + (size_t)(2 * NGAT + K) * NT * sizeof(int) + 16;
    auto* k = pi::pi_adj2d_persist_small_kernel<T, K, TILE_B, BY, NT, HALFS>;
    if (hipError_t e = allow_lds(k, lds)) return e;
    if (hipError_t e = hipMemsetAsync(outbox, 0, persist_small_outbox_bytes(p, (int)sizeof(T)), st)) return e;
    long frame_stride = (long)(2 * p.n);
    int np = pi::nparams(p.hc);
    pi::PersistArgs pa{};
    int slot;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        slot = g_persist.next_slot++ % PERSIST_SLOTS;
        g_persist.watch[slot] = false;
        ++g_persist.launches;
    }
    volatile int* hs = g_persist.host->slot[slot];
    hs[0] = 0; hs[1] = -1; hs[2] = -1; hs[3] = 0;
    pa.outbox = outbox; pa.sync = sync; pa.ngroups = ngroups;
    pa.host = const_cast<int*>(hs);
    pa.timeout_ticks = (unsigned long long)p.opt.persist_timeout_ms * 100000ull;
    pa.first_timeout_ticks = (unsigned long long)p.opt.persist_first_timeout_ms * 100000ull;
    pa.t_top = t_top;
    pa.pause = p.opt.adj_small_pause >= 0 ? p.opt.adj_small_pause : (grid <= 64 ? 20 : 28);
    pa.masked = mask ? 1 : 0;
    if (mask)
        for (int t = 0; t < t_top && t < 4096; ++t)
            if (mask[t]) pa.frames[t >> 5] |= 1u << (t & 31);
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, hframe_t, gframe_t, aframe_t, frame_stride, g_h0, partials, np, P, g, pa);
    if (hipError_t e = hipGetLastError()) return e;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[slot] = true;
    }
    if (!p.opt.persist_handshake) return hipSuccess;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (hs[0] == 0 && hs[3] == 0) {
        if ((++spins & 0x3ff) == 0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(600)) return hipErrorLaunchTimeOut;
            std::this_thread::yield();
        }
    }
    if (hs[3] != 0) {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[slot] = false;
        ++g_persist.aborts;
        g_persist.last_group = hs[1];
        g_persist.last_tile = hs[2];
        g_persist.disabled[dev] = true;
        if (!g_persist.warned) {
            g_persist.warned = true;
            std::fprintf(stderr, "percnn_pi: the persistent tile sweep could not keep all %u workgroups resident on device %d "
                                 "(group %d, tile %d: another process / kernel holds CUs, or a CU mask is set); using one launch "
                                 "per group of steps from now on (percnn_pi_set_option(\"persist_reset\", 1) re-arms it)\n",
                         grid, dev, (int)hs[1], (int)hs[2]);
        }
        return hipErrorLaunchFailure;
    }
    return hipSuccess;
}

// ---- small-tile / ragged resident FORWARD (pi_fwd2d_persist_small_kernel) ------------------------------------------------------
// tile height the resident small forward runs on: what the launch-per-group forward would use (8 / 16 rows), or 32 rows for the
// grids the 32 x 32 resident forward does not take (ragged, fewer than 16 tiles); 0 = not this path
// Grids of 113 .. 128 whole 32 x 32 tiles, where round 5 moved the SWEEP from 16-row to 32-row tiles (tile_by_for): since the
// 16-row resident forward works on half-strips and pair granules it beats the 32 x 32 resident forward there (us per step: 256 x 512
// 1.20 -> 1.01, 384 x 320 1.20 -> 1.01, 352^2 1.33 -> 1.20) while the 32-row sweep still wins (1.93 vs 2.04): the forward alone
// takes 16 rows.  Every kernel's trajectory is the same bit for bit.
template <typename T>
bool fwd_prefers_16_rows(const Problem& p)
{
    if (sizeof(T) != 4 || !p.opt.persist_small || !p.opt.fwd_small_half || p.opt.tile_by != 0 || p.hc != 0) return false;
    if (p.n0 % TILE_B || p.W % TILE_B || tile_by_for(p) != TILE_B) return false;
    const int64_t tiles32 = (p.n0 / TILE_B) * (p.W / TILE_B), tiles16 = (p.n0 / 16) * (p.W / TILE_B);
    const int cus = device_cu_count();
    return tiles32 > 112 && tiles32 <= 128 && cus > 0 && tiles16 <= std::min<int64_t>(cus, 256);
}

template <typename T>
int fwd_persist_small_by(const Problem& p, int ngroups, hipStream_t st)
{
    // (eight groups at least, as the 32 x 32 resident forward: short rollouts -- the step loop's speculative groups among them --
    // keep the launch-per-group kernel and a host that never waits for the stream)
    if (!p.opt.tile_persist || !p.opt.persist_small || !p.opt.fwd_persist || sizeof(T) != 4 || ngroups < 8 || p.hc != 0) return 0;
    if (persist_disabled_here()) return 0;
    if (p.opt.tile_k != 4 || p.opt.tile_nt != 512 || tile_wide_for<T>(p, false) != 0) return 0;
    const int by = tile_by_for(p);
    // Measured (profiles/r05_small_tile_resident_forward.txt, us per forward step, launch per group -> resident): 32 x 8 tiles
    // 100^2 1.23 -> 1.08, 256^2 1.22 -> 1.14; 32 x 16 tiles (300 x 320) 1.46 -> 1.63 and ragged 32 x 32 tiles (500^2) 1.92 -> 2.28:
    // whole-tile granules and an un-overlapped hand-over only pay where the sub-steps are short -> the 8-row regime by default,
    // persist_small = 2 takes the others too (tests)
    // Round 6, half-strips (twice the lanes, both waves of a SIMD busy): 32 x 8 tiles 100^2 1.05 -> 0.93, 256^2 1.08 -> 1.01; 32 x 16
    // tiles 288^2 1.53 -> 1.12, 300 x 320 1.47 -> 1.22, 260^2 1.43 -> 1.12 -> the 16-row regime joins the default; ragged 32 x 32
    // tiles (500^2: 1.90 -> 2.49) stay on the launch-per-group kernel -- until the same kernel on 1024 lanes (half-strips, pair granules,
    // a longer pause): 500^2 1.93 -> 1.60: default as well.  (512^2 on it: 1.40 against the 32 x 32 pyramid kernel's 1.35.)
    const bool by_default = by == 8 || ((by == 16 || by == TILE_B) && p.opt.fwd_small_half);
    if (!by_default && p.opt.persist_small < 2) return 0;
    const int64_t tiles = ((p.n0 + by - 1) / by) * ((p.W + TILE_B - 1) / TILE_B);
    const int cus = device_cu_count();
    // one workgroup per CU at most: resident for sure (persist_small = 2, experiments: up to fwd_persist_per_cu x 2 per CU --
    // the launch asks the runtime how many fit, the roll call decides)
    const int64_t max_tiles = p.opt.persist_small >= 2 ? (int64_t)cus * 2 * p.opt.fwd_persist_per_cu : std::min<int64_t>(cus, 256);
    if (tiles < 2 || cus <= 0 || tiles > max_tiles) return 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) return 0;
    return by;
}

// wait for the roll call of a resident launch (or its abort); what launch_fwd_persist / launch_adj_persist do inline
hipError_t persist_wait_roll_call(volatile int* hs, int slot, int dev, unsigned grid, const char* what)
{
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (hs[0] == 0 && hs[3] == 0) {
        if ((++spins & 0x3ff) == 0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(600)) return hipErrorLaunchTimeOut;
            std::this_thread::yield();
        }
    }
    if (hs[3] != 0) {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[slot] = false;
        ++g_persist.aborts;
        g_persist.last_group = hs[1];
        g_persist.last_tile = hs[2];
        g_persist.disabled[dev] = true;
        if (!g_persist.warned) {
            g_persist.warned = true;
            std::fprintf(stderr, "percnn_pi: %s could not keep all %u workgroups resident on device %d (group %d, tile %d: another "
                                 "process / kernel holds CUs, or a CU mask is set); using one launch per group of steps from now on "
                                 "(percnn_pi_set_option(\"persist_reset\", 1) re-arms it)\n", what, grid, dev, (int)hs[1], (int)hs[2]);
        }
        return hipErrorLaunchFailure;
    }
    return hipSuccess;
}

// frames t0 + 1 .. t0 + 4 * ngroups from frame t0; return values as launch_fwd_persist
template <typename T, int BY, int NT, bool HALFS = false>
hipError_t launch_fwd_persist_small_t(T* frame_t0, int ngroups, const T* P, const Problem& p, int dev, hipStream_t st)
{
    constexpr int K = 4;
    using TL = pi::Tile<K, TILE_B, BY>;
    pi::TileGeom g = make_tile_geom(p, BY);
    const unsigned grid = (unsigned)(((p.n0 + BY - 1) / BY) * g.tiles_x);
    constexpr int RINGH = TL::LX * TL::LY - TILE_B * BY, NGAT = (2 * RINGH + NT - 1) / NT;
    // state buffers | gather tables | K rows of strip geometry | abort word
    const size_t lds = pi::tile_state_bytes<T, K, TILE_B, BY>() + (size_t)(2 * NGAT + K) * NT * sizeof(int) + 16;
    auto* k = pi::pi_fwd2d_persist_small_kernel<T, K, TILE_B, BY, NT, HALFS>;
    if (hipError_t e = allow_lds(k, lds)) return e;
    {
        static int blocks_per_cu[16] = {};                  // per device (per tile height / value type: a template), asked once
        int dv = 0;
        if (hipGetDevice(&dv) != hipSuccess || dv < 0 || dv >= 16) { (void)hipGetLastError(); return hipErrorCooperativeLaunchTooLarge; }
        if (!blocks_per_cu[dv]) {
            int q = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k, NT, lds) != hipSuccess || q < 1) { (void)hipGetLastError(); q = -1; }
            blocks_per_cu[dv] = q;
        }
        const int nb = blocks_per_cu[dv];
        if (nb < 1 || (int64_t)grid > (int64_t)device_cu_count() * nb) return hipErrorCooperativeLaunchTooLarge;
    }
    // per-device scratch (shared with the 32 x 32 resident forward): 256 B of sync words | granule outbox: 2 parities x tiles x 2 x 32 x BY
    const size_t outbox_bytes = (size_t)2 * grid * (2 * TILE_B * BY) * sizeof(unsigned long long);
    const size_t need = 256 + outbox_bytes;
    unsigned char* scratch;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        if (g_persist.fwd_scratch_bytes[dev] < need) {
            if (g_persist.fwd_scratch[dev]) {
                (void)hipFree(g_persist.fwd_scratch[dev]);
                g_persist.fwd_scratch[dev] = nullptr;
                g_persist.fwd_scratch_bytes[dev] = 0;
            }
            void* q = nullptr;
            if (hipMalloc(&q, need) != hipSuccess || !q) { (void)hipGetLastError(); return hipErrorOutOfMemory; }
            g_persist.fwd_scratch[dev] = q;
            g_persist.fwd_scratch_bytes[dev] = need;
        }
        scratch = static_cast<unsigned char*>(g_persist.fwd_scratch[dev]);
    }
    if (hipError_t e = hipMemsetAsync(scratch, 0, need, st)) return e;
    long frame_stride = (long)(2 * p.n);
    pi::PersistArgs pa{};
    int slot;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        slot = g_persist.next_slot++ % PERSIST_SLOTS;
        g_persist.watch[slot] = false;
        ++g_persist.launches;
    }
    volatile int* hs = g_persist.host->slot[slot];
    hs[0] = 0; hs[1] = -1; hs[2] = -1; hs[3] = 0;
    pa.outbox = reinterpret_cast<unsigned long long*>(scratch + 256);
    pa.sync = reinterpret_cast<unsigned*>(scratch);
    pa.ngroups = ngroups;
    pa.host = const_cast<int*>(hs);
    pa.timeout_ticks = (unsigned long long)p.opt.persist_timeout_ms * 100000ull;             // 100 MHz clock
    pa.first_timeout_ticks = (unsigned long long)p.opt.persist_first_timeout_ms * 100000ull;
    // (32-row tiles -- ragged grids -- at pause 28 | 40 | 48 | 56 | 64 | 80: 500^2 1.77 1.66 1.60 1.57 1.61 1.73, 400^2 1.54 1.51 1.49 1.49 1.56 1.71)
    pa.pause = p.opt.fwd_small_pause >= 0 ? p.opt.fwd_small_pause : (BY == TILE_B ? 52 : (grid <= 64 ? 24 : 28));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, frame_t0, frame_stride, P, g, pa);
    if (hipError_t e = hipGetLastError()) return e;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[slot] = true;
    }
    if (!p.opt.persist_handshake) return hipSuccess;        // fire and forget (an abort: PERCNN_PI_EASYNC at the next entry point)
    return persist_wait_roll_call(hs, slot, dev, grid, "the resident forward rollout (small tiles)");
}

template <typename T>
hipError_t launch_fwd_persist_small(int by, T* frame_t0, int ngroups, const T* P, const Problem& p, int dev, hipStream_t st)
{
    // (round 6: 32 x 8 tiles on half-strips, 512 lanes; fwd_small_half=0: whole strips on 256 lanes)
    if (by == 8 && p.opt.fwd_small_half) return launch_fwd_persist_small_t<T, 8, 512, true>(frame_t0, ngroups, P, p, dev, st);
    if (by == 8) return launch_fwd_persist_small_t<T, 8, 256>(frame_t0, ngroups, P, p, dev, st);
    if (by == 16 && p.opt.fwd_small_half) return launch_fwd_persist_small_t<T, 16, 640, true>(frame_t0, ngroups, P, p, dev, st);
    if (by == 16) return launch_fwd_persist_small_t<T, 16, 320>(frame_t0, ngroups, P, p, dev, st);
    if (p.opt.fwd_small_half) return launch_fwd_persist_small_t<T, TILE_B, 1024, true>(frame_t0, ngroups, P, p, dev, st);
    return launch_fwd_persist_small_t<T, TILE_B, 512>(frame_t0, ngroups, P, p, dev, st);
}

template <typename T>
hipError_t launch_adj_persist_small(const T* hframe_t, const T* gframe_t, T* aframe_t, T* g_h0, int t_top, const unsigned char* mask,
                                    int ngroups, double* partials, unsigned long long* outbox, unsigned* sync, const T* P,
                                    const Problem& p, int dev, hipStream_t st)
{
    if (tile_by_for(p) == 16 && p.opt.adj_small_half)
        return launch_adj_persist_small_t<T, 16, 640, true>(hframe_t, gframe_t, aframe_t, g_h0, t_top, mask, ngroups, partials, outbox, sync, P, p, dev, st);
    if (tile_by_for(p) == 16)
        return launch_adj_persist_small_t<T, 16, 320>(hframe_t, gframe_t, aframe_t, g_h0, t_top, mask, ngroups, partials, outbox, sync, P, p, dev, st);
    // (round 6: half-strips on 512 lanes -- two waves per SIMD; adj_small_half=0: whole strips on 256 lanes)
    if (p.opt.adj_small_half)
        return launch_adj_persist_small_t<T, 8, 512, true>(hframe_t, gframe_t, aframe_t, g_h0, t_top, mask, ngroups, partials, outbox, sync, P, p, dev, st);
    return launch_adj_persist_small_t<T, 8, 256>(hframe_t, gframe_t, aframe_t, g_h0, t_top, mask, ngroups, partials, outbox, sync, P, p, dev, st);
}

// ---- workspace carving -------------------------------------------------------------------------
struct Workspace {
    void* adj[2];
    double* partials;
    size_t partials_bytes;
};

size_t partials_bytes_for(int hc) { return align_up((size_t)MAX_BWD_BLOCKS * pi::nparams(hc) * sizeof(double), 256); }

size_t workspace_bytes(const Problem& p, int elem)
{
    return 2 * align_up((size_t)2 * p.n * elem, 256) + partials_bytes_for(p.hc);
}

bool carve(void* ws, size_t bytes, const Problem& p, int elem, Workspace& w)
{
    if (!ws || bytes < workspace_bytes(p, elem) || (reinterpret_cast<uintptr_t>(ws) % 16)) return false;
    auto* b = static_cast<unsigned char*>(ws);
    const size_t a = align_up((size_t)2 * p.n * elem, 256);
    w.adj[0] = b; w.adj[1] = b + a;
    w.partials = reinterpret_cast<double*>(b + 2 * a);
    w.partials_bytes = partials_bytes_for(p.hc);
    return true;
}

hipError_t finish_grads(const Workspace& w, unsigned nblocks, int hc, double* param_grad, hipStream_t st)
{
    const int np = pi::nparams(hc);
    hipLaunchKernelGGL(pi::pi_reduce_partials_kernel, dim3(np), dim3(pi::WAVE), 0, st, w.partials, (int)nblocks, np,
                       param_grad);
    return hipGetLastError();
}

// ---- typed implementations -----------------------------------------------------------------------
int set_slab(Problem& p, int halo, int skip)
{
    if (halo < 2 || (halo & 1) || skip < 0 || (skip & 1) || skip > halo - 2) return PERCNN_PI_EINVAL;
    p.halo = halo; p.skip = skip;
    return 0;
}

int set_slab_range(Problem& p, int halo, int lo, int hi)
{
    if (halo < 2 || (halo & 1) || lo < 2 || hi <= lo || hi > p.n0 + 2 * halo - 2) return PERCNN_PI_EINVAL;
    p.halo = halo; p.skip = 0; p.lo = lo; p.hi = hi;
    return 0;
}

template <typename T>
int slab_step_fwd_range_impl(const T* h, T* out, const T* P, int hc, int ndim, const int64_t* shape, int halo, int lo,
                             int hi, void* stream)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, true, p)) return rc;
    if (int rc = set_slab_range(p, halo, lo, hi)) return rc;
    if (!h || !out || !P || h == out) return PERCNN_PI_EINVAL;
    return (int)step_fwd<T>(h, out, P, p, static_cast<hipStream_t>(stream));
}

template <typename T>
int step_fwd_impl(const T* h, T* out, const T* P, int hc, int ndim, const int64_t* shape, void* stream, bool slab,
                  int halo = 2, int skip = 0, const char* options = nullptr)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, slab, p, options)) return rc;
    if (slab) if (int rc = set_slab(p, halo, skip)) return rc;
    if (!h || !out || !P || h == out) return PERCNN_PI_EINVAL;
    return (int)step_fwd<T>(h, out, P, p, static_cast<hipStream_t>(stream));
}

template <typename T>
int step_bwd_impl(const T* h, const T* g_out, const T* g_inj, T* g_in, double* param_grad, void* ws, size_t ws_bytes,
                  const T* P, int hc, int ndim, const int64_t* shape, void* stream, bool slab, int halo = 2,
                  int flags = 0, int lo = -1, int hi = -1, const char* options = nullptr)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, slab, p, options)) return rc;
    if (slab) {
        if (lo >= 0) { if (int rc = set_slab_range(p, halo, lo, hi)) return rc; }
        else if (int rc = set_slab(p, halo, halo - 2)) return rc;     // adjoint: interior planes only
    }
    if (!h || !g_out || !g_in || (!param_grad && !(flags & PERCNN_PI_NO_FINISH)) || !P || g_in == g_out) return PERCNN_PI_EINVAL;
    Workspace w;
    if (!carve(ws, ws_bytes, p, sizeof(T), w)) return PERCNN_PI_EWORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    // a sweep split into several launches (plane ranges, time steps) keeps its sums in the partial rows:
    // NO_RESET = rows already hold earlier launches, NO_FINISH = a later launch reduces them into param_grad
    if (!(flags & PERCNN_PI_NO_RESET))
        if (hipError_t e = hipMemsetAsync(w.partials, 0, w.partials_bytes, st)) return (int)e;
    unsigned grid = 0;
    const bool sweep_only = flags & PERCNN_PI_SWEEP_ONLY;
    hipError_t e = sweep_only ? step_bwd<T, false>(h, g_out, g_inj, g_in, w.partials, P, p, st, &grid)
                              : step_bwd<T, true>(h, g_out, g_inj, g_in, w.partials, P, p, st, &grid);
    if (e) return (int)e;
    if (flags & PERCNN_PI_NO_FINISH) return 0;
    return (int)finish_grads(w, (flags & PERCNN_PI_NO_RESET) ? (unsigned)MAX_BWD_BLOCKS : grid, hc, param_grad, st);
}

// the partial rows of a workspace that earlier step_bwd launches (NO_FINISH) left their sums in -> param_grad (+=)
template <typename T>
int bwd_rows_finish_impl(void* ws, size_t ws_bytes, int hc, int ndim, const int64_t* shape, double* param_grad, void* stream)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, false, p)) return rc;
    if (!param_grad) return PERCNN_PI_EINVAL;
    Workspace w;
    if (!carve(ws, ws_bytes, p, sizeof(T), w)) return PERCNN_PI_EWORKSPACE;
    return (int)finish_grads(w, (unsigned)MAX_BWD_BLOCKS, hc, param_grad, static_cast<hipStream_t>(stream));
}

// branch-weight / coefficient-moment gradients of T steps over the interior of LOCAL slab trajectories
template <typename T>
int slab_wgrad_impl(const T* traj, const T* adj, double* param_grad, void* ws, size_t ws_bytes, const T* P, int hc,
                    int ndim, const int64_t* shape, int halo, int T_steps, void* stream)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, true, p)) return rc;
    if (int rc = set_slab(p, halo, halo - 2)) return rc;
    if (!traj || !adj || !param_grad || !P || T_steps < 0) return PERCNN_PI_EINVAL;
    Workspace w;
    if (!carve(ws, ws_bytes, p, sizeof(T), w)) return PERCNN_PI_EWORKSPACE;
    if (T_steps == 0) return 0;
    auto st = static_cast<hipStream_t>(stream);
    if (hipError_t e = hipMemsetAsync(w.partials, 0, w.partials_bytes, st)) return (int)e;
    unsigned rows = 0;
    const Geom g = make_geom(p);
    const bool vec_ok = (p.W % pi::vec_width<T>::value == 0) && p.opt.vec != 1 &&
                        (reinterpret_cast<uintptr_t>(traj) % 16 == 0) && (reinterpret_cast<uintptr_t>(adj) % 16 == 0);
    hipError_t e = vec_ok ? launch_wgrad<T, pi::vec_width<T>::value>(traj, adj, w.partials, P, p, 0, T_steps, &rows, st)
                          : launch_wgrad<T, 1>(traj, adj, w.partials, P, p, 0, T_steps, &rows, st);
    if (e) return (int)e;
    return (int)finish_grads(w, rows, hc, param_grad, st);
}

// ---- native slab rollouts: T-step loops with ring halo exchange (RCCL through function pointers) ------------------
template <typename T> int ring_dtype(const percnn_pi_halo_ring* r);
template <> int ring_dtype<float>(const percnn_pi_halo_ring* r) { return r->dtype_f32; }
template <> int ring_dtype<double>(const percnn_pi_halo_ring* r) { return r->dtype_f64; }

// the same exchange through the peer mailboxes (pi_peer.h): put into the neighbours' slots, take from mine
// bounded wait of a take when the ring does not say: PERCNN_PEER_TIMEOUT_S seconds (default 300 -- a rank that compiles the
// library, writes a checkpoint or loads data may arrive minutes late; the bound is there against DEAD neighbours), in 10 ns ticks
unsigned long long peer_default_timeout_ticks()
{
    static const unsigned long long ticks = [] {
        double s = 300.0;
        if (const char* e = std::getenv("PERCNN_PEER_TIMEOUT_S")) {
            char* end = nullptr;
            const double v = std::strtod(e, &end);
            if (end != e && v > 0.0) s = v;
        }
        return (unsigned long long)(s * 1e8);
    }();
    return ticks;
}

// one exchange through the mailboxes, described but not launched: the put and the take of `width` face planes of `slab`
// (advances the ring's exchange counter)
template <typename T>
int peer_prepare(T* slab, const Problem& p, int width, percnn_pi_peer_ring* pr, pi::PeerXfer& put, pi::PeerXfer& take, bool& vec)
{
    if (!pr || !pr->my_box || !pr->prev_box || !pr->next_box || width < 1 || width > p.halo || width > p.n0)
        return PERCNN_PI_EINVAL;
    const size_t plane = (size_t)(p.n1 * p.W), bytes = (size_t)width * plane * sizeof(T);
    const size_t ss = (size_t)(p.n0 + 2 * p.halo) * plane;
    const size_t soff = pi::peer_round16(bytes);                 // species 1 inside a slot
    if (2 * soff > pr->slot_bytes) return PERCNN_PI_EINVAL;
    const int64_t n = p.n0, halo = p.halo;
    auto at = [&](int s, int64_t pl) { return reinterpret_cast<char*>(slab + (size_t)s * ss + (size_t)pl * plane); };
    const uint64_t epoch = ++pr->epoch;
    const int parity = (int)(epoch & 1);
    put = pi::PeerXfer{};
    take = pi::PeerXfer{};
    // direction 0: my LAST interior planes -> the next rank (arrive there as "from prev", its lower halo);
    // direction 1: my FIRST interior planes -> the prev rank (arrive as "from next", its upper halo)
    char* to_next = pi::peer_slot(pr->next_box, pr->slot_bytes, parity, 0);
    char* to_prev = pi::peer_slot(pr->prev_box, pr->slot_bytes, parity, 1);
    char* from_prev = pi::peer_slot(pr->my_box, pr->slot_bytes, parity, 0);
    char* from_next = pi::peer_slot(pr->my_box, pr->slot_bytes, parity, 1);
    vec = bytes % 16 == 0;
    for (int s = 0; s < 2; ++s) {
        put.src[0][s] = at(s, halo + n - width); put.dst[0][s] = to_next + s * soff;
        put.src[1][s] = at(s, halo);             put.dst[1][s] = to_prev + s * soff;
        take.src[0][s] = from_prev + s * soff;   take.dst[0][s] = at(s, halo - width);
        take.src[1][s] = from_next + s * soff;   take.dst[1][s] = at(s, halo + n);
        for (int d = 0; d < 2; ++d)
            vec = vec && reinterpret_cast<uintptr_t>(put.src[d][s]) % 16 == 0 && reinterpret_cast<uintptr_t>(take.dst[d][s]) % 16 == 0;
    }
    const size_t units = bytes / (vec ? 16 : 4);
    // units per workgroup and species.  Measured through the rank's own mailbox (32 x 256^2 slab, us per fwd+bwd step, put and
    // take alike): 256 -> 89, 512 -> 67, 1024 -> 57.3, 2048 -> 52.4, 4096 -> 52.7, 8192 -> 61, 16384 -> 76: uncached stores / loads
    // from MANY workgroups at once contend (profiles/r03_peer_units_per_block.txt)
    const size_t upb = (size_t)p.opt.peer_upb, upt = (size_t)(p.opt.peer_upb_take ? p.opt.peer_upb_take : p.opt.peer_upb);
    const int bpd = (int)std::min<size_t>(256, std::max<size_t>(1, (units + upb - 1) / upb));
    const int bpt = (int)std::min<size_t>(256, std::max<size_t>(1, (units + upt - 1) / upt));
    put.bytes = take.bytes = bytes;
    put.epoch = take.epoch = epoch;
    put.blocks_per_dir = bpd;
    take.blocks_per_dir = bpt;
    put.wire_ticks = (unsigned)p.opt.peer_wire_us * 100u;
    put.mine = take.mine = static_cast<pi::PeerBox*>(pr->my_box);
    put.signal[0] = static_cast<pi::PeerBox*>(pr->next_box);
    put.signal[1] = static_cast<pi::PeerBox*>(pr->prev_box);
    return 0;
}

inline unsigned long long peer_ticks(const percnn_pi_peer_ring* pr)
{
    return pr->timeout_ticks ? pr->timeout_ticks : peer_default_timeout_ticks();
}

inline int peer_launch_take(const pi::PeerXfer& take, bool vec, const percnn_pi_peer_ring* pr, hipStream_t st)
{
    if (vec) hipLaunchKernelGGL(pi::peer_take_kernel<true>, dim3(2 * take.blocks_per_dir), dim3(256), 0, st, take, peer_ticks(pr));
    else     hipLaunchKernelGGL(pi::peer_take_kernel<false>, dim3(2 * take.blocks_per_dir), dim3(256), 0, st, take, peer_ticks(pr));
    return (int)hipGetLastError();
}

template <typename T>
int peer_exchange(T* slab, const Problem& p, int width, percnn_pi_peer_ring* pr, hipStream_t st)
{
    pi::PeerXfer put, take;
    bool vec = false;
    if (int rc = peer_prepare<T>(slab, p, width, pr, put, take, vec)) return rc;
    if (vec) hipLaunchKernelGGL(pi::peer_put_kernel<true>, dim3(2 * put.blocks_per_dir), dim3(256), 0, st, put);
    else     hipLaunchKernelGGL(pi::peer_put_kernel<false>, dim3(2 * put.blocks_per_dir), dim3(256), 0, st, put);
    return peer_launch_take(take, vec, pr, st);
}

template <typename T>
int peer_exchange_impl(T* slab, int ndim, const int64_t* shape, int halo, int width, percnn_pi_peer_ring* ring, void* stream)
{
    Problem p;
    if (int rc = make_problem(0, ndim, shape, true, p)) return rc;
    if (halo < 1 || !slab) return PERCNN_PI_EINVAL;
    p.halo = halo;
    return peer_exchange<T>(slab, p, width, ring, static_cast<hipStream_t>(stream));
}

// faces of `slab` ([2][n0 + 2*halo][plane]) -> the neighbours' halo planes, `width` planes per side, on stream st;
// slab2 / width2: a second array of the same layout exchanged in the SAME ncclGroup (one call's latency for both)
template <typename T>
int ring_exchange(T* slab, const Problem& p, int width, const percnn_pi_halo_ring* r, hipStream_t st, T* slab2 = nullptr,
                  int width2 = 0)
{
    if (r && r->peer) {
        if (int rc = peer_exchange<T>(slab, p, width, r->peer, st)) return rc;
        return slab2 ? peer_exchange<T>(slab2, p, width2, r->peer, st) : 0;
    }
    const size_t plane = (size_t)(p.n1 * p.W), cnt = (size_t)width * plane;
    const size_t ss = (size_t)(p.n0 + 2 * p.halo) * plane;
    const int64_t n = p.n0, halo = p.halo;
    auto at = [&](int s, int64_t pl) { return slab + (size_t)s * ss + (size_t)pl * plane; };
    if (!r) {                                              // single rank: periodic wrap, ONE copy launch for the four faces
        pi::PeerXfer cp{};                                 // (was four hipMemcpyAsync: 48.2 -> 44 us per fwd+bwd step on the 32 x 256^2 slab)
        const size_t fb = cnt * sizeof(T);
        bool vec = fb % 16 == 0;
        for (int s = 0; s < 2; ++s) {
            cp.src[0][s] = reinterpret_cast<const char*>(at(s, halo + n - width)); cp.dst[0][s] = reinterpret_cast<char*>(at(s, halo - width));
            cp.src[1][s] = reinterpret_cast<const char*>(at(s, halo));             cp.dst[1][s] = reinterpret_cast<char*>(at(s, halo + n));
            for (int d = 0; d < 2; ++d)
                vec = vec && reinterpret_cast<uintptr_t>(cp.src[d][s]) % 16 == 0 && reinterpret_cast<uintptr_t>(cp.dst[d][s]) % 16 == 0;
        }
        cp.bytes = fb;
        cp.blocks_per_dir = (int)std::min<size_t>(256, std::max<size_t>(1, (fb / (vec ? 16 : 4) + 1023) / 1024));
        if (vec) hipLaunchKernelGGL(pi::face_copy_kernel<true>, dim3(2 * cp.blocks_per_dir), dim3(256), 0, st, cp);
        else hipLaunchKernelGGL(pi::face_copy_kernel<false>, dim3(2 * cp.blocks_per_dir), dim3(256), 0, st, cp);
        if (hipError_t e = hipGetLastError()) return (int)e;
        return slab2 ? ring_exchange<T>(slab2, p, width2, r, st) : 0;
    }
    const int dt = ring_dtype<T>(r);
    if (!slab2 && r->stage && (size_t)8 * cnt * sizeof(T) <= r->stage_bytes) {
        // packed faces: [to next | to prev | from prev | from next], both species each
        char* stg = static_cast<char*>(r->stage);
        const size_t fb = cnt * sizeof(T), msg = 2 * fb;
        pi::PeerXfer pk{}, un{};
        bool vec = fb % 16 == 0 && reinterpret_cast<uintptr_t>(stg) % 16 == 0;
        for (int s = 0; s < 2; ++s) {
            pk.src[0][s] = reinterpret_cast<const char*>(at(s, halo + n - width)); pk.dst[0][s] = stg + s * fb;
            pk.src[1][s] = reinterpret_cast<const char*>(at(s, halo));             pk.dst[1][s] = stg + msg + s * fb;
            un.src[0][s] = stg + 2 * msg + s * fb;  un.dst[0][s] = reinterpret_cast<char*>(at(s, halo - width));
            un.src[1][s] = stg + 3 * msg + s * fb;  un.dst[1][s] = reinterpret_cast<char*>(at(s, halo + n));
            for (int d = 0; d < 2; ++d)
                vec = vec && reinterpret_cast<uintptr_t>(pk.src[d][s]) % 16 == 0 && reinterpret_cast<uintptr_t>(un.dst[d][s]) % 16 == 0;
        }
        const size_t units = fb / (vec ? 16 : 4);
        pk.bytes = un.bytes = fb;
        pk.blocks_per_dir = un.blocks_per_dir = (int)std::min<size_t>(256, std::max<size_t>(1, (units + 1023) / 1024));
        if (vec) hipLaunchKernelGGL(pi::face_copy_kernel<true>, dim3(2 * pk.blocks_per_dir), dim3(256), 0, st, pk);
        else hipLaunchKernelGGL(pi::face_copy_kernel<false>, dim3(2 * pk.blocks_per_dir), dim3(256), 0, st, pk);
        if (int rc = r->group_start()) return rc;
        if (int rc = r->send(stg, 2 * cnt, dt, r->next, r->comm, st)) return rc;
        if (int rc = r->send(stg + msg, 2 * cnt, dt, r->prev, r->comm, st)) return rc;
        if (int rc = r->recv(stg + 2 * msg, 2 * cnt, dt, r->prev, r->comm, st)) return rc;
        if (int rc = r->recv(stg + 3 * msg, 2 * cnt, dt, r->next, r->comm, st)) return rc;
        if (int rc = r->group_end()) return rc;
        if (vec) hipLaunchKernelGGL(pi::face_copy_kernel<true>, dim3(2 * un.blocks_per_dir), dim3(256), 0, st, un);
        else hipLaunchKernelGGL(pi::face_copy_kernel<false>, dim3(2 * un.blocks_per_dir), dim3(256), 0, st, un);
        return (int)hipGetLastError();
    }
    if (int rc = r->group_start()) return rc;
    // per peer, sends and receives pair up in issue order (matters when prev == next: 2 ranks)
    for (int pass = 0; pass < (slab2 ? 2 : 1); ++pass) {
        T* base = pass ? slab2 : slab;
        const int wd = pass ? width2 : width;
        const size_t c = (size_t)wd * plane;
        auto a2 = [&](int s, int64_t pl) { return base + (size_t)s * ss + (size_t)pl * plane; };
        for (int s = 0; s < 2; ++s) {
            if (int rc = r->send(a2(s, halo + n - wd), c, dt, r->next, r->comm, st)) return rc;
            if (int rc = r->send(a2(s, halo), c, dt, r->prev, r->comm, st)) return rc;
        }
        for (int s = 0; s < 2; ++s) {
            if (int rc = r->recv(a2(s, halo - wd), c, dt, r->prev, r->comm, st)) return rc;
            if (int rc = r->recv(a2(s, halo + n), c, dt, r->next, r->comm, st)) return rc;
        }
    }
    return r->group_end();
}

// exchange on the side stream, ordered after everything enqueued on `st` so far; *done is recorded behind it
template <typename T>
int ring_exchange_async(T* slab, const Problem& p, int width, const percnn_pi_halo_ring* r, hipStream_t st, SideStream* ss,
                        hipEvent_t* done)
{
    hipEvent_t ready = ss->ev[ss->next++ % 8];
    if (hipError_t e = hipEventRecord(ready, st)) return (int)e;
    if (hipError_t e = hipStreamWaitEvent(ss->stream, ready, 0)) return (int)e;
    if (int rc = ring_exchange<T>(slab, p, width, r, ss->stream)) return rc;
    *done = ss->ev[ss->next++ % 8];
    return (int)hipEventRecord(*done, ss->stream);
}

template <typename T>
int slab_rollout_fwd_impl(T* traj, const T* P, int hc, int ndim, const int64_t* shape, int halo, int T_steps,
                          const percnn_pi_halo_ring* ring, int overlap, void* stream)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, true, p)) return rc;
    if (int rc = set_slab(p, halo, 0)) return rc;
    if (!traj || !P || T_steps < 0) return PERCNN_PI_EINVAL;
    // the exchange sends `halo` INTERIOR planes per side: a thinner slab would forward halo planes as if they were data
    if (p.n0 < halo) return PERCNN_PI_EINVAL;
    auto st = static_cast<hipStream_t>(stream);
    const int k = halo / 2;
    const int64_t n = p.n0;
    const size_t frame = (size_t)2 * (p.n0 + 2 * halo) * p.n1 * p.W;
    if (!ring && !(overlap & 2) && p.opt.slab_local_index) {
        // one rank: the interior IS the periodic domain -- no exchange, no halo planes, one launch per step
        Problem q = p;
        q.slab_periodic = true;
        for (int t = 0; t < T_steps; ++t)
            if (hipError_t e = step_fwd<T>(traj + (size_t)t * frame, traj + (size_t)(t + 1) * frame, P, q, st)) return (int)e;
        return 0;
    }
    overlap &= 1;
    SideStream* side = overlap && n >= 2 * halo ? side_stream() : nullptr;
    hipEvent_t pending = nullptr;
    auto range = [&](T* in, T* out, int lo, int hi) -> int {
        Problem q = p;
        if (int rc = set_slab_range(q, halo, lo, hi)) return rc;
        return (int)step_fwd<T>(in, out, P, q, st);
    };
    // peer mailboxes + brick kernels: the step that writes a frame about to be exchanged puts its faces itself
    const int vecw = pick_vec<T>(p, {traj, traj + frame});
    const bool fuse_put = ring && ring->peer && p.opt.slab_fused_put && !side && (size_t)vecw * sizeof(T) == 16;
    pi::PeerXfer take_pending{};
    bool have_take = false, take_vec = false;
    for (int t = 0; t < T_steps; ++t) {
        T* cur = traj + (size_t)t * frame;
        T* nxt = cur + frame;
        const int m = t % k;
        if (m == 0) {
            if (have_take) {                                 // the put of this exchange rode on the previous step's launch
                if (int rc = peer_launch_take(take_pending, take_vec, ring->peer, st)) return rc;
                have_take = false;
            } else if (pending) {
                if (hipError_t e = hipStreamWaitEvent(st, pending, 0)) return (int)e;
                pending = nullptr;
            } else if (int rc = ring_exchange<T>(cur, p, halo, ring, st)) return rc;
        }
        if (fuse_put && m == k - 1 && t + 1 < T_steps) {
            Problem q = p;
            if (int rc = set_slab(q, halo, 2 * m)) return rc;
            const int brz = stream3d_vec<T>(q, {cur, nxt}, false) ? 0 : brick_rz_for<T>(q, vecw, false);   // step_fwd's own choice
            if (brz) {
                FusedPut fp{};
                if (int rc = peer_prepare<T>(nxt, p, halo, ring->peer, fp.x, take_pending, take_vec)) return rc;
                fp.vec16 = take_vec;
                fp.timeout_ticks = peer_ticks(ring->peer);
                fp.face_lo[0] = halo + (int)n - halo; fp.face_hi[0] = halo + (int)n;     // my LAST interior planes -> next
                fp.face_lo[1] = halo;                 fp.face_hi[1] = 2 * halo;          // my FIRST interior planes -> prev
                if (hipError_t e = brick_fwd<T>(brz, cur, nxt, P, q, st, &fp)) return (int)e;
                have_take = true;
                continue;
            }
        }
        if (side && m == k - 1 && t + 1 < T_steps) {        // frame t+1 is exchanged next: faces first
            if (int rc = range(cur, nxt, halo, 2 * halo)) return rc;
            if (int rc = range(cur, nxt, (int)n, (int)n + halo)) return rc;
            if (int rc = ring_exchange_async<T>(nxt, p, halo, ring, st, side, &pending)) return rc;
            if (n > 2 * halo)
                if (int rc = range(cur, nxt, 2 * halo, (int)n)) return rc;
        } else {
            Problem q = p;
            if (int rc = set_slab(q, halo, 2 * m)) return rc;
            if (hipError_t e = step_fwd<T>(cur, nxt, P, q, st)) return (int)e;
        }
    }
    if (pending)
        if (hipError_t e = hipStreamWaitEvent(st, pending, 0)) return (int)e;
    return 0;
}

template <typename T>
int slab_rollout_bwd_impl(const T* traj, const T* g_traj, T* adj, double* param_grad, void* ws, size_t ws_bytes,
                          const T* P, int hc, int ndim, const int64_t* shape, int halo, int T_steps,
                          const percnn_pi_halo_ring* ring, int overlap, void* stream)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, true, p)) return rc;
    if (int rc = set_slab(p, halo, halo - 2)) return rc;
    if (!traj || !g_traj || !adj || !param_grad || !P || T_steps < 0) return PERCNN_PI_EINVAL;
    if (p.n0 < 2) return PERCNN_PI_EINVAL;                  // the adjoint exchange sends 2 interior planes per side
    Workspace w;
    if (!carve(ws, ws_bytes, p, sizeof(T), w)) return PERCNN_PI_EWORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    const int64_t n = p.n0;
    const size_t plane = (size_t)(p.n1 * p.W), ss = (size_t)(n + 2 * halo) * plane, frame = 2 * ss;
    // frame 0 of `adj` is returned to the caller as dL/dh0 (padded layout): its halo planes are never written by the
    // sweep (no exchange for frame 0), so define them here -- zeros, as the portable orchestration returns
    for (int s = 0; s < 2; ++s) {
        if (hipError_t e = hipMemsetAsync(adj + (size_t)s * ss, 0, (size_t)halo * plane * sizeof(T), st)) return (int)e;
        if (hipError_t e = hipMemsetAsync(adj + (size_t)s * ss + (size_t)(halo + n) * plane, 0,
                                          (size_t)halo * plane * sizeof(T), st)) return (int)e;
    }
    const bool by_index = !ring && !(overlap & 2) && p.opt.slab_local_index;     // one rank: wrap by index, no exchange
    // float32 poly mode: the 20 coefficient moments are reduced inside the sweep launches (no slab_wgrad pass)
    const bool fuse = hc == 0 && sizeof(T) == 4 && p.opt.fuse_wgrad != 0;
    // adj[T] interior = dL/dtraj[T] interior (one rank, fused sums: the first sweep launch reads dL/dtraj[T] where it lies --
    // nobody exchanges or re-reads adj[T])
    const bool top_in_place = by_index && fuse && T_steps > 0;
    if (!top_in_place)
        for (int s = 0; s < 2; ++s) {
            const size_t o = (size_t)T_steps * frame + (size_t)s * ss + (size_t)halo * plane;
            if (hipError_t e = hipMemcpyAsync(adj + o, g_traj + o, (size_t)n * plane * sizeof(T), hipMemcpyDeviceToDevice, st))
                return (int)e;
        }
    if (T_steps == 0) return 0;
    if (hipError_t e = hipMemsetAsync(w.partials, 0, w.partials_bytes, st)) return (int)e;
    overlap &= 1;
    SideStream* side = overlap && n >= 4 && !by_index ? side_stream() : nullptr;
    hipEvent_t pending = nullptr;
    auto sweep = [&](int t, int lo, int hi, bool strip = false) -> int {   // adjoint planes [lo, hi) of frame t-1 from frame t
        Problem q = p;
        if (by_index) q.slab_periodic = true;
        else if (int rc = set_slab_range(q, halo, lo, hi)) return rc;
        unsigned grid = 0;
        const T* hf = traj + (size_t)(t - 1) * frame;
        const T* gf = (top_in_place && t == T_steps) ? g_traj + (size_t)t * frame : adj + (size_t)t * frame;
        const T* jf = g_traj + (size_t)(t - 1) * frame;
        T* of = adj + (size_t)(t - 1) * frame;
        // strips outside the interior (wide-halo blocks below) belong to the neighbour's sums: their partial rows go to a
        // scratch area nobody reads
        if (strip) return (int)step_bwd<T, false>(hf, gf, jf, of, static_cast<double*>(w.adj[0]), P, q, st, &grid);
        return fuse ? (int)step_bwd<T, true>(hf, gf, jf, of, w.partials, P, q, st, &grid)
                    : (int)step_bwd<T, false>(hf, gf, jf, of, w.partials, P, q, st, &grid);
    };
    // Wide-halo adjoint over RCCL (option slab_wide_adjoint, off: measured slower, see Options): TWO adjoint steps share one
    // exchange: 4 planes of adj[t] and 2 planes of dL/dtraj[t-1] per side travel together, adj[t-1] is then also computed on the two
    // planes next to each face (two small launches; the trajectory halos are at least 2 planes valid in every frame the
    // forward wrote), and step t-1 -> t-2 needs no exchange.  The halo planes of g_traj are used as receive buffers.
    // Not with the peer mailboxes (their per-exchange cost is two small launches already), not in overlap mode.
    const bool wide = ring && !ring->peer && !side && p.opt.slab_wide_adjoint && halo >= 4 && n >= 4 &&
                      (size_t)2 * p.n * sizeof(T) >= w.partials_bytes;
    T* g_mut = const_cast<T*>(g_traj);
    const bool fuse_put = ring && ring->peer && p.opt.slab_fused_put_adj && !side && !wide;
    pi::PeerXfer take_pending{};
    bool have_take = false, take_vec = false;
    for (int t = T_steps; t >= 1; --t) {
        if (wide && t >= 2) {
            if (int rc = ring_exchange<T>(adj + (size_t)t * frame, p, 4, ring, st, g_mut + (size_t)(t - 1) * frame, 2)) return rc;
            if (int rc = sweep(t, halo - 2, halo, true)) return rc;
            if (int rc = sweep(t, halo + (int)n, halo + (int)n + 2, true)) return rc;
            if (int rc = sweep(t, halo, halo + (int)n)) return rc;
            --t;                                           // adj[t-1] now has 2 valid halo planes per side
            if (int rc = sweep(t, halo, halo + (int)n)) return rc;
            continue;
        }
        if (by_index) {
            if (int rc = sweep(t, halo, halo + (int)n)) return rc;
            continue;
        }
        if (have_take) {                                   // the put of this exchange rode on the previous sweep launch
            if (int rc = peer_launch_take(take_pending, take_vec, ring->peer, st)) return rc;
            have_take = false;
        } else if (pending) {
            if (hipError_t e = hipStreamWaitEvent(st, pending, 0)) return (int)e;
            pending = nullptr;
        } else if (int rc = ring_exchange<T>(adj + (size_t)t * frame, p, 2, ring, st)) return rc;
        if (fuse_put && t > 1) {
            // peer mailboxes + brick kernels: the sweep launch that writes adj[t-1] puts its faces itself
            Problem q = p;
            if (int rc = set_slab_range(q, halo, halo, halo + (int)n)) return rc;
            const T* hf = traj + (size_t)(t - 1) * frame;
            const T* gf = adj + (size_t)t * frame;
            const T* jf = g_traj + (size_t)(t - 1) * frame;
            T* of = adj + (size_t)(t - 1) * frame;
            const int vq = pick_vec<T>(q, {hf, gf, jf, of});
            const bool streams = stream3d_vec<T>(q, {hf, gf, jf, of}, true) != 0;
            const int brz = streams ? 0 : brick_rz_for<T>(q, vq, true);
            if (brz && brick_bwd_can_put<T>(brz, q) && (size_t)vq * sizeof(T) == 16) {
                FusedPut fp{};
                if (int rc = peer_prepare<T>(of, p, 2, ring->peer, fp.x, take_pending, take_vec)) return rc;
                fp.vec16 = take_vec;
                fp.timeout_ticks = peer_ticks(ring->peer);
                fp.face_lo[0] = halo + (int)n - 2; fp.face_hi[0] = halo + (int)n;        // my LAST interior planes -> next
                fp.face_lo[1] = halo;              fp.face_hi[1] = halo + 2;              // my FIRST interior planes -> prev
                if (hipError_t e = brick_bwd<T>(brz, fuse, hf, gf, jf, of, w.partials, P, q, st, &fp)) return (int)e;
                have_take = true;
                continue;
            }
        }
        if (side && t > 1) {
            if (int rc = sweep(t, halo, halo + 2)) return rc;
            if (int rc = sweep(t, halo + (int)n - 2, halo + (int)n)) return rc;
            if (int rc = ring_exchange_async<T>(adj + (size_t)(t - 1) * frame, p, 2, ring, st, side, &pending)) return rc;
            if (n > 4)
                if (int rc = sweep(t, halo + 2, halo + (int)n - 2)) return rc;
        } else if (int rc = sweep(t, halo, halo + (int)n)) return rc;
    }
    // diffusion-coefficient sums of the whole sweep (kept in the partial rows across all launches) ...
    if (hipError_t e = finish_grads(w, MAX_BWD_BLOCKS, hc, param_grad, st)) return (int)e;
    if (fuse) return 0;
    // ... then the branch gradients of all steps in one time-parallel reduction over the local interior
    return slab_wgrad_impl<T>(traj, adj, param_grad, ws, ws_bytes, P, hc, ndim, shape, halo, T_steps, stream);
}

template <typename T>
int rollout_fwd_impl(T* traj, const T* P, int hc, int ndim, const int64_t* shape, int T_steps, void* stream,
                     const char* options = nullptr)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, false, p, options)) return rc;
    if (!traj || !P || T_steps < 0) return PERCNN_PI_EINVAL;
    auto st = static_cast<hipStream_t>(stream);
    const size_t frame = (size_t)2 * p.n;
    int t = 0;
    if (tile_eligible<T>(p, {traj}, false)) {
        const int K = (p.opt.tile_k == 8 && p.hc != 0) ? 4 : p.opt.tile_k;
        // whole groups of four steps as ONE launch of resident workgroups where the grid allows (pi_fwd2d_persist_kernel: the
        // launch-per-group kernel's trajectory bit for bit); an aborted launch is recomputed below, launch by launch
        {
            const int ngroups = K == 4 ? T_steps / K : 0;
            int pdev = 0;
            const bool big = fwd_persist_ok<T>(p, ngroups, st);
            // ... and the grids the 32 x 32 flavour does not take -- small-tile regime, ragged grids, fewer than 16 tiles -- on
            // pi_fwd2d_persist_small_kernel (round 5; float32)
            int small_by = 0;
            if constexpr (sizeof(T) == 4) small_by = big ? (fwd_prefers_16_rows<T>(p) ? 16 : 0) : fwd_persist_small_by<T>(p, ngroups, st);
            if ((big || small_by) && persist_enter(st, pdev)) {
                hipError_t e;
                if constexpr (sizeof(T) == 4)
                    e = (big && !small_by) ? launch_fwd_persist<T>(traj, ngroups, P, p, pdev, st)
                            : launch_fwd_persist_small<T>(small_by, traj, ngroups, P, p, pdev, st);
                else
                    e = launch_fwd_persist<T>(traj, ngroups, P, p, pdev, st);
                if (e == hipSuccess) { t = K * ngroups; persist_leave(st, pdev); }
                else if (e == hipErrorLaunchTimeOut) return (int)e;
                else {
                    (void)hipGetLastError();                // not resident / not supported / aborted
                    if (e == hipErrorLaunchFailure) persist_leave(st, pdev);
                }
            }
        }
        for (; t + K <= T_steps; t += K)
            if (hipError_t e = fwd_tile<T>(traj + (size_t)t * frame, P, p, st)) return (int)e;
    }
    for (; t < T_steps; ++t)
        if (hipError_t e = step_fwd<T>(traj + (size_t)t * frame, traj + (size_t)(t + 1) * frame, P, p, st)) return (int)e;
    return 0;
}

size_t rollout_workspace_bytes(const Problem& p, int T_steps, int elem)
{
    return align_up((size_t)(T_steps + 1) * 2 * p.n * elem, 256) + partials_bytes_for(p.hc) + persist_small_outbox_bytes(p, elem);
}

template <typename T>
int rollout_bwd_impl(const T* traj, const T* g_traj, const unsigned char* mask, T* g_h0, double* param_grad, void* ws,
                     size_t ws_bytes, const T* P, int hc, int ndim, const int64_t* shape, int T_steps, void* stream,
                     const char* options = nullptr, const pi::LossInj* loss = nullptr, const T* g_top = nullptr)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, false, p, options)) return rc;
    // g_top: dL/d(frame T_steps) lives in a buffer of its own (frame T_steps of g_traj is then never read; needs mask[T_steps])
    if (g_top && (loss || (mask && !mask[T_steps]))) return PERCNN_PI_EINVAL;
    // loss != nullptr: no dL/dtraj exists; `g_traj` is the TARGET trajectory (mode 2) or ignored (mode 1) and the sweep forms
    // the gradient of frame t from the state it reads anyway (pi::LossInj); `mask` then selects the frames inside the loss
    if (loss) {
        if ((loss->mode != 1 && loss->mode != 2) || (loss->mode == 2 && !g_traj)) return PERCNN_PI_EINVAL;
        p.loss = *loss;
        if (p.loss.mode == 1) g_traj = traj;
    }
    if (!traj || !g_traj || !g_h0 || !param_grad || !P || T_steps < 0) return PERCNN_PI_EINVAL;
    // kernel families without the in-kernel form (advective blocks, a forced plane-streaming adjoint): the caller materialises
    if (loss && (hc == -1 || stream3d_vec<T>(p, {traj, g_traj, g_h0}, true))) return PERCNN_PI_EINVAL;
    if (!ws || ws_bytes < rollout_workspace_bytes(p, T_steps, sizeof(T)) || (reinterpret_cast<uintptr_t>(ws) % 16))
        return PERCNN_PI_EWORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    const size_t frame = (size_t)2 * p.n;
    const size_t frame_bytes = frame * sizeof(T);
    T* adj = static_cast<T*>(ws);                       // adjoint trajectory: adj[t] = dL/dh_t (all consumers)
    Workspace w;
    w.partials = reinterpret_cast<double*>(static_cast<unsigned char*>(ws) +
                                           align_up((size_t)(T_steps + 1) * frame_bytes, 256));
    w.partials_bytes = partials_bytes_for(p.hc);
    auto has = [&](int t) { return !mask || mask[t]; };

    // frames after the last one carrying gradient contribute nothing: start the sweep there
    int t_top = T_steps;
    while (t_top > 0 && !has(t_top)) --t_top;
    // dL/dh of one frame that no later step injects (the top frame): a copy, or -- loss form -- a * (h - target)
    auto top_frame = [&](int t, T* dst) -> hipError_t {
        if (!loss)
            return hipMemcpyAsync(dst, (g_top && t == T_steps) ? g_top : g_traj + (size_t)t * frame, frame_bytes,
                                  hipMemcpyDeviceToDevice, st);
        const T* tg = p.loss.mode == 2 ? g_traj + (size_t)t * frame : nullptr;
        const bool v16 = frame % pi::vec_width<T>::value == 0 && reinterpret_cast<uintptr_t>(traj) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(dst) % 16 == 0 && (!tg || reinterpret_cast<uintptr_t>(g_traj) % 16 == 0) &&
                         frame_bytes % 16 == 0;
        const unsigned nb = (unsigned)std::min<size_t>(2048, (frame + 1023) / 1024);
        if (v16) hipLaunchKernelGGL((pi::pi_loss_grad_kernel<T, pi::vec_width<T>::value>), dim3(nb), dim3(256), 0, st,
                                    traj + (size_t)t * frame, tg, dst, (long)frame, p.loss);
        else     hipLaunchKernelGGL((pi::pi_loss_grad_kernel<T, 1>), dim3(nb), dim3(256), 0, st,
                                    traj + (size_t)t * frame, tg, dst, (long)frame, p.loss);
        return hipGetLastError();
    };
    if (t_top == 0) {
        if (has(0)) return (int)top_frame(0, g_h0);
        return (int)hipMemsetAsync(g_h0, 0, frame_bytes, st);
    }
    if (hipError_t e = hipMemsetAsync(w.partials, 0, w.partials_bytes, st)) return (int)e;

    // 1) sequential reverse sweep: adjoint states (+ diffusion-coefficient gradients), with
    // 2) the time-parallel branch-gradient reduction of every finished chunk of steps running UNDER it on a side
    //    stream (the sweep is latency-bound, the reduction HBM-bound); partial rows are disjoint columns.
    const bool vec_ok = (p.n % pi::vec_width<T>::value == 0) && p.opt.vec != 1 &&
                        (reinterpret_cast<uintptr_t>(traj) % 16 == 0);
    // fused gradient reduction only where the per-step direct kernels sweep EVERY step (no tile launches, no plane
    // streaming): the other kernel families have no fused flavour
    // (the streaming kernel's fused flavour exists for float32 poly mode)
    const bool f32poly = hc == 0 && sizeof(T) == 4;
    const bool direct_sweep = !tile_eligible<T>(p, {traj, g_traj, g_h0, adj}, true) &&
                              (f32poly || !stream3d_vec<T>(p, {traj, g_traj, g_h0, adj}, true));
    const bool tile_fused = !direct_sweep && tile_eligible<T>(p, {traj, g_traj, g_h0, adj}, true) && tile_fuse_ok<T>(p);
    // (float64 pre-contracted blocks too since round 2: per-lane double accumulators; lambda-omega beyond the tile regime,
    // backward per step 1200^2 29.0 -> 25.1 us, 2048^2 71.5 -> 60.4, 3072^2 187 -> 143 -- no pi_moments_kernel pass)
    const bool fuse = tile_fused || (direct_sweep && !p.opt.skip_wgrad && hc != -1 &&
                                     (p.opt.fuse_wgrad == 1 || (p.opt.fuse_wgrad == 2 && hc == 0)));
    // The top frame's dL/dh is dL/dtraj[t_top] itself.  Where every step is its own launch and nobody else reads the adjoint
    // trajectory (fused gradient sums), the first step reads it where it lies instead of from a copy in adj[t_top] -- at 256^3
    // that copy is 128 MiB, ~45 us of a 10-step rollout (round 5)
    const T* top_in_place = (direct_sweep && fuse && !loss) ? ((g_top && t_top == T_steps) ? g_top : g_traj + (size_t)t_top * frame) : nullptr;
    if (!top_in_place)
        if (hipError_t e = top_frame(t_top, adj + (size_t)t_top * frame)) return (int)e;
    unsigned rows = 0, wrows = 0;
    auto reduce_range = [&](int lo, int hi, hipStream_t s2) -> hipError_t {      // steps (lo, hi]
        if (hi <= lo || p.opt.skip_wgrad || fuse) return hipSuccess;
        unsigned r = 0;
        hipError_t e = vec_ok ? launch_wgrad<T, pi::vec_width<T>::value>(traj, adj, w.partials, P, p, lo, hi, &r, s2)
                              : launch_wgrad<T, 1>(traj, adj, w.partials, P, p, lo, hi, &r, s2);
        if (r > wrows) wrows = r;
        return e;
    };
    SideStream* ss = (p.opt.overlap && !p.opt.skip_wgrad && !fuse && t_top >= 2 * p.opt.overlap_chunk)
                         ? side_stream() : nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (ss && (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) ss = nullptr;
    int reduced_above = t_top;                       // steps (reduced_above, t_top] are already handed to the reduction
    auto hand_over = [&](int t_done) -> hipError_t { // adj frames > t_done ... are final: reduce steps (t_done, reduced_above]
        if (!ss || reduced_above - t_done < p.opt.overlap_chunk) return hipSuccess;
        hipEvent_t ev = ss->ev[ss->next++ % 8];
        if (hipError_t e = hipEventRecord(ev, st)) return e;
        if (hipError_t e = hipStreamWaitEvent(ss->stream, ev, 0)) return e;
        if (hipError_t e = reduce_range(t_done, reduced_above, ss->stream)) return e;
        reduced_above = t_done;
        return hipSuccess;
    };
    if (ss) {                                         // the side stream must see the memset / top-frame copy
        hipEvent_t ev = ss->ev[ss->next++ % 8];
        if (hipError_t e = hipEventRecord(ev, st)) return (int)e;
        if (hipError_t e = hipStreamWaitEvent(ss->stream, ev, 0)) return (int)e;
    }
    int t_cur = t_top;
    if (tile_eligible<T>(p, {traj, g_traj, g_h0, adj}, true)) {
        const int K = (p.opt.tile_k == 8 && p.hc != 0) ? 4 : p.opt.tile_k;
        rows = (unsigned)tile_count<T>(p, true);
        // the whole tile sweep as ONE cooperative launch where that applies (pi_adj2d_persist_kernel); its granule outbox
        // lives in the three adjoint frames between the hand-over frame and the group above it, which nobody touches then
        {
            const int ngroups = K == 4 ? t_cur / K : 0;
            if (tile_fused && ngroups >= 2 && persist_ok<T>(p, mask, t_cur, ngroups, st) &&
                persist_outbox_bytes(p, (int)sizeof(T)) <= (size_t)(K - 1) * frame_bytes) {
                const int t_end = t_cur - K * ngroups;
                auto* outbox = reinterpret_cast<unsigned long long*>(adj + (size_t)(t_end + 1) * frame);
                // roll-call / abort words: the last partial row (zeroed above; tiles <= #CUs << MAX_BWD_BLOCKS rows are in use)
                unsigned* sync = reinterpret_cast<unsigned*>(w.partials + (size_t)(MAX_BWD_BLOCKS - 1) * pi::nparams(p.hc));
                int pdev = 0;
                if (persist_enter(st, pdev)) {
                    const hipError_t e = launch_adj_persist<T>(traj + (size_t)t_cur * frame, g_traj + (size_t)t_cur * frame,
                                                               adj + (size_t)t_cur * frame, t_end == 0 ? g_h0 : nullptr, t_cur,
                                                               mask, ngroups, w.partials, outbox, sync, P, p, pdev, st);
                    if (e == hipSuccess) { t_cur = t_end; persist_leave(st, pdev); }
                    else if (e == hipErrorLaunchTimeOut) return (int)e;
                    else {
                        (void)hipGetLastError();            // not resident / not supported / aborted: the launch-per-group path
                        if (e == hipErrorLaunchFailure) {   // it RAN and gave up: workgroups that were already through may have
                            persist_leave(st, pdev);        // added to their partial rows -- start the rows over
                            if (hipError_t e2 = hipMemsetAsync(w.partials, 0, w.partials_bytes, st)) return (int)e2;
                        }
                    }
                }
            }
        }
        // the small-tile regime (32 x 8 tiles, split schedule): the same, with every adjoint frame written and the granule
        // outbox behind the partial rows
        if constexpr (sizeof(T) == 4) {
            const int ngroups = K == 4 ? t_cur / K : 0;
            if (t_cur == t_top && !tile_fused && !loss && ngroups >= 2 && persist_small_ok<T>(p, mask, t_cur, ngroups, st) &&
                ws_bytes >= rollout_workspace_bytes(p, T_steps, sizeof(T))) {
                const int t_end = t_cur - K * ngroups;
                auto* outbox = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(w.partials) + w.partials_bytes);
                unsigned* sync = reinterpret_cast<unsigned*>(w.partials + (size_t)(MAX_BWD_BLOCKS - 1) * pi::nparams(p.hc));
                int pdev = 0;
                if (persist_enter(st, pdev)) {
                    const hipError_t e = launch_adj_persist_small<T>(traj + (size_t)t_cur * frame, g_traj + (size_t)t_cur * frame,
                                                                     adj + (size_t)t_cur * frame, t_end == 0 ? g_h0 : nullptr, t_cur,
                                                                     mask, ngroups, w.partials, outbox, sync, P, p, pdev, st);
                    if (e == hipSuccess) { t_cur = t_end; persist_leave(st, pdev); }
                    else if (e == hipErrorLaunchTimeOut) return (int)e;
                    else {
                        (void)hipGetLastError();
                        if (e == hipErrorLaunchFailure) {
                            persist_leave(st, pdev);
                            if (hipError_t e2 = hipMemsetAsync(w.partials, 0, w.partials_bytes, st)) return (int)e2;
                        }
                    }
                }
            }
        }
        for (; t_cur - K >= 0; t_cur -= K) {
            unsigned m = 0;
            for (int q = 0; q < K; ++q) if (has(t_cur - 1 - q)) m |= 1u << q;
            if (hipError_t e = adj_tile<T>(traj + (size_t)t_cur * frame, g_traj + (size_t)t_cur * frame,
                                           adj + (size_t)t_cur * frame, m, g_h0, t_cur == K ? K : 0, w.partials, P, p, st))
                return (int)e;
            // frames t_cur-K .. t_cur-1 were just written; steps (t_cur - K, ...] need adj frames > t_cur - K: all final
            if (hipError_t e = hand_over(t_cur - K)) return (int)e;
        }
    }
    // 3D float32 pre-contracted blocks on whole 16 x 16 x 32 blocks: the whole sweep as ONE resident launch (pi_res3d.h), the
    // gradient sums carried along; aborted / not resident / not supported: the launch-per-step bricks below
    if constexpr (sizeof(T) == 4) {
        Res3dPlan rp;
        if (t_cur == t_top && direct_sweep && fuse && !loss && res3d_plan(p, t_top, sizeof(T), rp)) {
            const T* gin = top_in_place ? top_in_place : adj + (size_t)t_top * frame;
            const hipError_t e = launch_adj_res3d(traj, g_traj, mask, gin, g_h0, t_top, w.partials, P, p, rp, st);
            if (e == hipSuccess) { t_cur = 0; rows = (unsigned)rp.nblk; }
            else if (e == hipErrorLaunchTimeOut) return (int)e;
            else {
                (void)hipGetLastError();
                if (e == hipErrorLaunchFailure)             // it ran and gave up: blocks that were through may have added to their rows
                    if (hipError_t e2 = hipMemsetAsync(w.partials, 0, w.partials_bytes, st)) return (int)e2;
            }
        }
    }
    for (int t = t_cur; t >= 1; --t) {
        T* dst = (t == 1) ? g_h0 : adj + (size_t)(t - 1) * frame;
        const T* inj = has(t - 1) ? g_traj + (size_t)(t - 1) * frame : nullptr;
        unsigned r2 = 0;
        const T* gin = (top_in_place && t == t_top) ? top_in_place : adj + (size_t)t * frame;
        hipError_t e = fuse
            ? step_bwd<T, true>(traj + (size_t)(t - 1) * frame, gin, inj, dst, w.partials, P, p, st, &r2)
            : step_bwd<T, false>(traj + (size_t)(t - 1) * frame, gin, inj, dst, w.partials, P, p, st, &r2);
        if (e) return (int)e;
        if (r2 > rows) rows = r2;
        if (hipError_t e2 = hand_over(t - 1)) return (int)e2;
    }
    if (p.opt.skip_wgrad || hc == -1 || fuse)
        return (int)finish_grads(w, rows, hc, param_grad, st);
    if (ss) {
        // remaining steps (0, reduced_above] on the side stream too (ordered behind the earlier chunks), then join
        hipEvent_t ev = ss->ev[ss->next++ % 8];
        if (hipError_t e = hipEventRecord(ev, st)) return (int)e;
        if (hipError_t e = hipStreamWaitEvent(ss->stream, ev, 0)) return (int)e;
        if (hipError_t e = reduce_range(0, reduced_above, ss->stream)) return (int)e;
        if (hipError_t e = hipEventRecord(ss->done, ss->stream)) return (int)e;
        if (hipError_t e = hipStreamWaitEvent(st, ss->done, 0)) return (int)e;
    } else {
        if (hipError_t e = reduce_range(0, t_top, st)) return (int)e;
    }
    return (int)finish_grads(w, rows > wrows ? rows : wrows, hc, param_grad, st);
}


// ---- physics residual over a trajectory (time-parallel; pre-contracted block of the TRUE equation) ----
pi::FrameGrid make_frame_grid(long nchunks, unsigned frame_slots)
{
    const unsigned gx = (unsigned)((nchunks + 255) / 256);
    return pi::FrameGrid{gx, frame_slots, (gx + pi::NXCD - 1) / pi::NXCD};
}

template <typename T, bool ADJ>
int residual_impl(const T* traj, const T* G, T* out, const T* Q, int ndim, const int64_t* shape, int nframes, void* stream)
{
    Problem p;
    if (int rc = make_problem(0, ndim, shape, false, p)) return rc;
    if (!traj || !out || !Q || (ADJ && !G) || nframes < 0 || nframes > 65535) return PERCNN_PI_EINVAL;
    if (nframes == 0) return 0;
    const Geom g = make_geom(p);
    const int vec = pick_vec<T>(p, {traj, G, out});
    const long nchunks = (long)g.rows * (g.W / vec);
    const pi::FrameGrid fg = make_frame_grid(nchunks, (unsigned)nframes);
    const dim3 grid(pi::frame_grid_blocks(fg)), block(256);
    auto st = static_cast<hipStream_t>(stream);
    constexpr int V = pi::vec_width<T>::value;
#define PI_RES(NDIM, VEC)                                                                                      \
    do {                                                                                                       \
        if constexpr (ADJ) hipLaunchKernelGGL((pi::pi_residual_adj_kernel<T, NDIM, VEC>), grid, block, 0, st, traj, G, out, Q, g, fg, 0); \
        else hipLaunchKernelGGL((pi::pi_residual_kernel<T, NDIM, VEC>), grid, block, 0, st, traj, out, Q, g, fg);    \
    } while (0)
    if (ndim == 2) { if (vec == 1) PI_RES(2, 1); else PI_RES(2, V); }
    else           { if (vec == 1) PI_RES(3, 1); else PI_RES(3, V); }
#undef PI_RES
    return (int)hipGetLastError();
}

// ---- the residual LOSS (MSE of f_u + MSE of f_v, optionally with the reference's padded-grid weighting) and its gradient
// w.r.t. the trajectory, without materialising the residual or the autograd temporaries of the loss expression ----
constexpr unsigned RESLOSS_SLOTS = 16384;           // per-block partial sums of the loss pass (doubles)
// 2D tile flavour: workgroups in flight ~ twice what the chip holds; each walks its frames y, y + gy, ... (512^2 x 200 frames:
// 203 -> 196 us against 16384 short-lived workgroups; PMC: 832 VALU instructions per wave of which a third were prologue)
constexpr unsigned RESLOSS_WGS = 4096;

double resloss_scale(const Problem& p, int ndim, const int64_t* shape, int nframes, int weighted)
{
    double pts = 1.0;
    for (int a = 0; a < ndim; ++a) pts *= (double)(shape[a] + (weighted ? 1 : 0));
    (void)p;
    return 1.0 / ((double)nframes * pts);
}

template <typename T>
int residual_sqloss_impl(const T* traj, const T* Q, int ndim, const int64_t* shape, int nframes, int weighted, T* loss_out,
                         void* ws, size_t ws_bytes, void* stream)
{
    Problem p;
    if (int rc = make_problem(0, ndim, shape, false, p)) return rc;
    if (!traj || !Q || !loss_out || nframes < 1) return PERCNN_PI_EINVAL;
    if (!ws || ws_bytes < RESLOSS_SLOTS * sizeof(double) || reinterpret_cast<uintptr_t>(ws) % 8) return PERCNN_PI_EWORKSPACE;
    const Geom g = make_geom(p);
    const int vec = pick_vec<T>(p, {traj});
    auto st0 = static_cast<hipStream_t>(stream);
    // 3D grids the brick kernels take: plane neighbours from a register window, in-plane neighbours from LDS (pi_brick3d.h)
    if (ndim == 3 && p.opt.brick3d) {
        const int brz = brick_rz_for<T>(p, vec, false, false);
        if (brz == 1 || brz == 2) {
            pi::BrickGeom b = make_brick_geom(p, pi::vec_width<T>::value, brz, pi::BRICK_NT, false);
            if (b.nblk > 0 && b.nblk <= RESLOSS_SLOTS) {
                unsigned gy = RESLOSS_SLOTS / b.nblk;                    // (4096 workgroups measured slower here: 1.61 -> 1.69 ms at 128^3)
                if (gy > (unsigned)nframes) gy = (unsigned)nframes;
                if (gy > 65535u) gy = 65535u;
                const size_t lds = (size_t)2 * brz * pi::brick_wb(pi::BRICK_NT);
                const double scale = resloss_scale(p, ndim, shape, nframes, weighted);
                double* partials = static_cast<double*>(ws);
                const long frame = 2 * p.n;
                hipError_t e = hipSuccess;
                if (brz == 2) {
                    auto* k = pi::pi_res3d_brick_kernel<T, 2>;
                    e = allow_lds(k, lds);
                    if (e == hipSuccess) hipLaunchKernelGGL(k, dim3(b.nblk, gy), dim3(pi::BRICK_NT), lds, st0, traj, partials, Q, b, frame, nframes, weighted);
                } else {
                    auto* k = pi::pi_res3d_brick_kernel<T, 1>;
                    e = allow_lds(k, lds);
                    if (e == hipSuccess) hipLaunchKernelGGL(k, dim3(b.nblk, gy), dim3(pi::BRICK_NT), lds, st0, traj, partials, Q, b, frame, nframes, weighted);
                }
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL((pi::pi_sqerr_finish_kernel<T>), dim3(1), dim3(64), 0, st0, partials, (int)(b.nblk * gy), scale, loss_out);
                return (int)hipGetLastError();
            }
        }
    }
    // 2D grids the tile machinery takes: the tile + its 2-wide ring once through LDS (pi_tile2d.h)
    if (ndim == 2 && p.opt.tile && vec == pi::vec_width<T>::value && p.n0 >= TILE_B + 2 && p.W >= TILE_B + 2 &&
        reinterpret_cast<uintptr_t>(traj) % 16 == 0) {
        const pi::TileGeom tg = make_tile_geom(p, TILE_B, TILE_B);
        const unsigned tiles = (unsigned)(((p.n0 + TILE_B - 1) / TILE_B) * tg.tiles_x);
        if (tiles <= RESLOSS_SLOTS) {
            unsigned gy = std::max(1u, RESLOSS_WGS / tiles);
            if (gy > (unsigned)nframes) gy = (unsigned)nframes;
            if (gy > 65535u) gy = 65535u;
            const size_t lds = (size_t)2 * (TILE_B + 8) * (TILE_B + 4) * sizeof(T);
            double* partials = static_cast<double*>(ws);
            auto* k = pi::pi_res2d_tile_kernel<T, TILE_B, TILE_B, 256>;
            hipLaunchKernelGGL(k, dim3(tiles, gy), dim3(256), lds, st0, traj, partials, Q, tg, (long)(2 * p.n), nframes, weighted);
            hipLaunchKernelGGL((pi::pi_sqerr_finish_kernel<T>), dim3(1), dim3(64), 0, st0, partials, (int)(tiles * gy),
                               resloss_scale(p, ndim, shape, nframes, weighted), loss_out);
            return (int)hipGetLastError();
        }
    }
    const long nchunks = (long)g.rows * (g.W / vec);
    const unsigned gx = (unsigned)((nchunks + 255) / 256);
    if (gx > RESLOSS_SLOTS) return PERCNN_PI_ETOOLARGE;
    unsigned gy = RESLOSS_SLOTS / gx;
    if (gy > (unsigned)nframes) gy = (unsigned)nframes;
    const pi::FrameGrid fg = make_frame_grid(nchunks, gy);
    auto st = static_cast<hipStream_t>(stream);
    double* partials = static_cast<double*>(ws);
    const pi::ResLoss rl{resloss_scale(p, ndim, shape, nframes, weighted), nullptr, weighted};
    constexpr int V = pi::vec_width<T>::value;
#define PI_RSQ(NDIM, VEC) hipLaunchKernelGGL((pi::pi_residual_sq_kernel<T, NDIM, VEC, false>), dim3(pi::frame_grid_blocks(fg)), \
                                             dim3(256), 0, st, traj, (T*)nullptr, partials, Q, g, nframes, rl, fg)
    if (ndim == 2) { if (vec == 1) PI_RSQ(2, 1); else PI_RSQ(2, V); }
    else           { if (vec == 1) PI_RSQ(3, 1); else PI_RSQ(3, V); }
#undef PI_RSQ
    hipLaunchKernelGGL((pi::pi_sqerr_finish_kernel<T>), dim3(1), dim3(64), 0, st, partials, (int)(gx * gy), rl.scale, loss_out);
    return (int)hipGetLastError();
}

// g_traj[0 .. nout) = g_loss * dL/dtraj for L over the residuals of frames 0 .. nframes-1 (nout >= nframes + 1; frames past
// nframes are zeroed).  scratch: nframes frames (the scaled residual G).  Two launches.
template <typename T>
int residual_sqloss_bwd_impl(const T* traj, const T* g_loss, const T* Q, int ndim, const int64_t* shape, int nframes, int nout,
                             int weighted, T* scratch, T* g_traj, void* stream)
{
    Problem p;
    if (int rc = make_problem(0, ndim, shape, false, p)) return rc;
    if (!traj || !Q || !scratch || !g_traj || nframes < 1 || nout < nframes + 1 || nout > 65535) return PERCNN_PI_EINVAL;
    const Geom g = make_geom(p);
    const int vec = pick_vec<T>(p, {traj, scratch, g_traj});
    const long nchunks = (long)g.rows * (g.W / vec);
    const pi::FrameGrid fg1 = make_frame_grid(nchunks, (unsigned)nframes), fg2 = make_frame_grid(nchunks, (unsigned)nout);
    auto st = static_cast<hipStream_t>(stream);
    const pi::ResLoss rl{resloss_scale(p, ndim, shape, nframes, weighted), g_loss, weighted};
    constexpr int V = pi::vec_width<T>::value;
#define PI_RSB(NDIM, VEC)                                                                                                   \
    do {                                                                                                                    \
        hipLaunchKernelGGL((pi::pi_residual_sq_kernel<T, NDIM, VEC, true>), dim3(pi::frame_grid_blocks(fg1)), dim3(256), 0, \
                           st, traj, scratch, (double*)nullptr, Q, g, nframes, rl, fg1);                                    \
        hipLaunchKernelGGL((pi::pi_residual_adj_kernel<T, NDIM, VEC, true>), dim3(pi::frame_grid_blocks(fg2)), dim3(256),   \
                           0, st, traj, (const T*)scratch, g_traj, Q, g, fg2, nframes);                                     \
    } while (0)
    if (ndim == 2) { if (vec == 1) PI_RSB(2, 1); else PI_RSB(2, V); }
    else           { if (vec == 1) PI_RSB(3, 1); else PI_RSB(3, V); }
#undef PI_RSB
    return (int)hipGetLastError();
}

}  // namespace

namespace {

int apply_option(Options& o, const char* key, long value)
{
    if (!key) return PERCNN_PI_EINVAL;
    if (!std::strcmp(key, "vec")) {
        if (value != 0 && value != 1) return PERCNN_PI_EINVAL;
        o.vec = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "tile")) {                       // 0 = never, 1 = size heuristic, 2 = whenever eligible
        if (value < 0 || value > 2) return PERCNN_PI_EINVAL;
        o.tile = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "tile_xcd")) { o.tile_xcd = value != 0; return 0; }
    if (!std::strcmp(key, "tile_by")) {
        if (value != 0 && value != 8 && value != 16 && value != 32) return PERCNN_PI_EINVAL;
        o.tile_by = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "slab_fused_put_adj")) { o.slab_fused_put_adj = value != 0; return 0; }
    if (!std::strcmp(key, "peer_upb") || !std::strcmp(key, "peer_upb_take")) {
        const bool take = key[8] == '_';
        if ((value < 256 && !(take && value == 0)) || value > 65536) return PERCNN_PI_EINVAL;
        (take ? o.peer_upb_take : o.peer_upb) = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "tile_persist")) {
        if (value < 0 || value > 2) return PERCNN_PI_EINVAL;
        o.tile_persist = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "persist_handshake")) { o.persist_handshake = value != 0; return 0; }
    if (!std::strcmp(key, "persist_split")) { o.persist_split = value != 0; return 0; }
    if (!std::strcmp(key, "persist_small")) { if (value < 0 || value > 2) return PERCNN_PI_EINVAL; o.persist_small = (int)value; return 0; }
    if (!std::strcmp(key, "fwd_persist")) { o.fwd_persist = value != 0; return 0; }
    if (!std::strcmp(key, "fwd_persist_f64")) { o.fwd_persist_f64 = value != 0; return 0; }
    if (!std::strcmp(key, "adj_persist_f64")) { o.adj_persist_f64 = value != 0; return 0; }
    if (!std::strcmp(key, "fwd_persist_per_cu")) {
        if (value < 1 || value > 2) return PERCNN_PI_EINVAL;
        o.fwd_persist_per_cu = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "persist_timeout_ms") || !std::strcmp(key, "persist_first_timeout_ms")) {
        if (value < 1 || value > 600000) return PERCNN_PI_EINVAL;
        (key[8] == 'f' ? o.persist_first_timeout_ms : o.persist_timeout_ms) = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "persist_reset")) { persist_reset(); return 0; }
    if (!std::strcmp(key, "tile_wide")) {
        if (value < 0 || value > 3) return PERCNN_PI_EINVAL;
        o.tile_wide = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "skip_wgrad")) { o.skip_wgrad = value != 0; return 0; }
    if (!std::strcmp(key, "fuse_wgrad")) {
        if (value < 0 || value > 2) return PERCNN_PI_EINVAL;
        o.fuse_wgrad = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "overlap")) { o.overlap = value != 0; return 0; }
    if (!std::strcmp(key, "overlap_chunk")) {
        if (value < 1 || value > (1 << 20)) return PERCNN_PI_EINVAL;
        o.overlap_chunk = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "tile_fuse")) {                        // 0 split, 1 fused for float32 (default), 2 fused for float64 too
        if (value < 0 || value > 2) return PERCNN_PI_EINVAL;
        o.tile_fuse = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "bwd_cpl")) {
        if (value < 1 || value > 16) return PERCNN_PI_EINVAL;
        o.bwd_cpl = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "stream3d")) {                         // 0 = never, 1 = size heuristic, 2 = whenever eligible
        if (value < 0 || value > 2) return PERCNN_PI_EINVAL;
        o.stream3d = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "zc")) {
        if (value < 1 || value > 1024) return PERCNN_PI_EINVAL;
        o.zc = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "fwd_blocks")) {
        if (value < 0 || value > (1 << 24)) return PERCNN_PI_EINVAL;
        o.fwd_blocks = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "xcd_window")) {
        if (value < 0 || value > (1 << 24) || value % 8) return PERCNN_PI_EINVAL;
        o.xcd_window = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "block_small")) { o.block_small = value != 0; return 0; }
    if (!std::strcmp(key, "slab_wide_adjoint")) { o.slab_wide_adjoint = value != 0; return 0; }
    if (!std::strcmp(key, "slab_local_index")) { o.slab_local_index = value != 0; return 0; }
    if (!std::strcmp(key, "fwd_small_half")) { if (value < 0 || value > 1) return PERCNN_PI_EINVAL; o.fwd_small_half = (int)value; return 0; }
    if (!std::strcmp(key, "adj_small_half")) { if (value < 0 || value > 1) return PERCNN_PI_EINVAL; o.adj_small_half = (int)value; return 0; }
    if (!std::strcmp(key, "adj_small_pause")) { if (value < -1 || value > 200) return PERCNN_PI_EINVAL; o.adj_small_pause = (int)value; return 0; }
    if (!std::strcmp(key, "fwd_small_pause")) { if (value < -1 || value > 200) return PERCNN_PI_EINVAL; o.fwd_small_pause = (int)value; return 0; }
    if (!std::strcmp(key, "brick_xny")) { if (value < -1 || value > 8 || value == 3 || (value > 4 && value < 8)) return PERCNN_PI_EINVAL; o.brick_xny = (int)value; return 0; }
    if (!std::strcmp(key, "slab_fused_put")) { o.slab_fused_put = value != 0; return 0; }
    if (!std::strcmp(key, "peer_wire_us")) {
        if (value < 0 || value > 100000) return PERCNN_PI_EINVAL;
        o.peer_wire_us = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "slab_put_blocks")) {
        if (value < 4 || value > 256 || value % 4) return PERCNN_PI_EINVAL;
        o.slab_put_blocks = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "lane_x")) {
        if (value != 0 && value != -1 && (value < 2 || value > 7)) return PERCNN_PI_EINVAL;     // 7 = flat
        o.lane_x = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "rz")) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return PERCNN_PI_EINVAL;
        o.rz = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "l2_tile_min_kb")) {
        if (value < 0 || value > (1 << 22)) return PERCNN_PI_EINVAL;
        o.l2_tile_min_kb = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "l2_tile_kb")) {
        if (value < 0 || value > 16384) return PERCNN_PI_EINVAL;
        o.l2_tile_kb = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "brick3d")) {                          // 0 = never, 1 = size heuristic, 2 = whenever eligible
        if (value < 0 || value > 2) return PERCNN_PI_EINVAL;
        o.brick3d = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "res3d")) {                            // 0 = never, 1 = where it measured faster, 2 = whenever eligible
        if (value < 0 || value > 2) return PERCNN_PI_EINVAL;
        o.res3d = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "brick_wt")) { o.brick_wt = value != 0; return 0; }
    if (!std::strcmp(key, "brick_xcd")) {
        if (value < 0 || value > 2) return PERCNN_PI_EINVAL;
        o.brick_xcd = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "brick_wide")) { o.brick_wide = value != 0; return 0; }
    if (!std::strcmp(key, "brick_nt")) {
        if (value != 0 && value != 256 && value != 512) return PERCNN_PI_EINVAL;
        o.brick_nt = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "brick_wgs")) {
        if (value < 0 || value > 16) return PERCNN_PI_EINVAL;
        o.brick_wgs = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "brick_rz")) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return PERCNN_PI_EINVAL;
        o.brick_rz = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "lds_pad")) {
        if (value < 0 || value > 80 * 1024 || value % 16) return PERCNN_PI_EINVAL;
        o.lds_pad = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "tile_k")) {
        if (value != 2 && value != 4 && value != 8) return PERCNN_PI_EINVAL;   // 8: poly mode only (else 4)
        o.tile_k = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "tile_nt")) {
        if (value != 256 && value != 512 && value != 1024) return PERCNN_PI_EINVAL;   // 1024: poly mode only
        o.tile_nt = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "wgrad_blocks")) {
        if (value < 1 || value > MAX_BWD_BLOCKS / 2) return PERCNN_PI_EINVAL;
        o.wgrad_blocks = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "block")) {
        if (value < 64 || value > 256 || value % 64) return PERCNN_PI_EINVAL;
        o.block = (int)value;
        return 0;
    }
    if (!std::strcmp(key, "graph")) return value == 0 ? 0 : PERCNN_PI_EINVAL;   // reserved
    return PERCNN_PI_EINVAL;
}

// "key=value,key=value" (whitespace-free); empty / NULL = no overrides
int apply_overrides(Options& o, const char* spec)
{
    if (!spec) return 0;
    char key[32];
    while (*spec) {
        const char* eq = std::strchr(spec, '=');
        if (!eq || eq == spec || (size_t)(eq - spec) >= sizeof(key)) return PERCNN_PI_EINVAL;
        std::memcpy(key, spec, (size_t)(eq - spec));
        key[eq - spec] = 0;
        char* end = nullptr;
        const long v = std::strtol(eq + 1, &end, 10);
        if (end == eq + 1 || (*end && *end != ',')) return PERCNN_PI_EINVAL;
        if (int rc = apply_option(o, key, v)) return rc;
        spec = *end ? end + 1 : end;
    }
    return 0;
}

}  // namespace


namespace {
// loss value of the squared-error losses the sweep differentiates in place (pi::LossInj): scale * sum over the frames with
// mask[f] != 0 of sum_x (traj - target)^2, written to out[0] (compute type).  ws: >= 1024 doubles.
template <typename T>
int sqerr_impl(const T* traj, const T* target, const unsigned char* mask, int nframes, int ndim, const int64_t* shape,
               double scale, T* out, void* ws, size_t ws_bytes, void* stream)
{
    Problem p;
    if (int rc = make_problem(0, ndim, shape, false, p)) return rc;
    if (!traj || !out || nframes < 0) return PERCNN_PI_EINVAL;
    constexpr unsigned NB = 1024;
    if (!ws || ws_bytes < NB * sizeof(double) || reinterpret_cast<uintptr_t>(ws) % 8) return PERCNN_PI_EWORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    const long frame = 2 * p.n;
    // the mask lives on the HOST (as in rollout_bwd); the kernel wants it on the device: pass it by value in 64-bit words
    // would cap T; instead launch once per run of selected frames (runs are few: a slice, or everything)
    double* partials = static_cast<double*>(ws);
    if (hipError_t e = hipMemsetAsync(partials, 0, NB * sizeof(double), st)) return (int)e;
    const bool v16 = frame % pi::vec_width<T>::value == 0 && reinterpret_cast<uintptr_t>(traj) % 16 == 0 &&
                     (!target || reinterpret_cast<uintptr_t>(target) % 16 == 0) && (frame * sizeof(T)) % 16 == 0;
    // runs of consecutive selected frames (a slice, or everything: one run); each gets an equal stripe of the partial slots
    int nruns = 0;
    for (int f = 0; f < nframes; ++f)
        if ((!mask || mask[f]) && (f == 0 || (mask && !mask[f - 1]))) ++nruns;
    // (more than 64 runs -- e.g. every 10th frame of 1000: the runs take the 64 stripes in turn and ADD to them; launches of
    // one stream are ordered, so the sums stay deterministic)
    const unsigned stripes = (unsigned)(nruns > 64 ? 64 : nruns);
    const unsigned nb = stripes ? NB / stripes : NB;
    int f = 0, run = 0;
    while (f < nframes) {
        while (f < nframes && mask && !mask[f]) ++f;
        int g = f;
        while (g < nframes && (!mask || mask[g])) ++g;
        if (g > f) {
            double* dst = partials + (size_t)(run++ % (int)stripes) * nb;
            const T* tr = traj + (size_t)f * frame;
            const T* tg = target ? target + (size_t)f * frame : nullptr;
            if (v16) hipLaunchKernelGGL((pi::pi_sqerr_kernel<T, pi::vec_width<T>::value>), dim3(nb), dim3(256), 0, st, tr, tg,
                                        (long)(g - f) * frame, dst);
            else     hipLaunchKernelGGL((pi::pi_sqerr_kernel<T, 1>), dim3(nb), dim3(256), 0, st, tr, tg,
                                        (long)(g - f) * frame, dst);
        }
        f = g;
    }
    hipLaunchKernelGGL((pi::pi_sqerr_finish_kernel<T>), dim3(1), dim3(64), 0, st, partials, (int)NB, scale, out);
    return (int)hipGetLastError();
}
}  // namespace

namespace {
// Which kernel family a rollout of this problem runs on and how its backward is scheduled -- the library's own dispatch
// rules, evaluated for 16-byte-aligned buffers (bench.py labels its roofline entries with it instead of mirroring the rules).
// out = {forward family, adjoint family, gradients reduced inside the sweep launches (0 / 1), time steps per forward launch,
//        per adjoint launch, planes per pass forward, adjoint, lanes per brick workgroup (0: no bricks), 2D tile width,
//        height, lanes per tile workgroup of the adjoint (0: no tiles), the same three of the forward, 1 if the tile sweep of a
//        long unmasked rollout runs as ONE persistent launch}; families: 0 direct, 1 2D tiles, 2 plane streaming, 3 3D bricks,
//        4 advective block
template <typename T>
int debug_plan_impl(int hc, int ndim, const int64_t* shape, const char* options, int* out)
{
    Problem p;
    if (int rc = make_problem(hc, ndim, shape, false, p, options, false)) return rc;
    const int vec = pick_vec<T>(p, {});
    auto family = [&](bool adjoint, bool wgrad, int& planes) {
        planes = 1;
        if (p.hc == -1) return 4;
        if (tile_eligible<T>(p, {}, adjoint)) return 1;
        if (!adjoint || !wgrad || (p.hc == 0 && sizeof(T) == 4))
            if (stream3d_vec<T>(p, {}, adjoint)) return 2;
        if (!adjoint || !wgrad || p.hc == 0)
            if (const int brz = brick_rz_for<T>(p, vec, adjoint)) { planes = brz; return 3; }
        planes = direct_rz<T>(p, vec, adjoint);
        return 0;
    };
    const bool f32poly = hc == 0 && sizeof(T) == 4;
    const bool tiled_adj = tile_eligible<T>(p, {}, true);
    const bool direct_sweep = !tiled_adj && (f32poly || !stream3d_vec<T>(p, {}, true));
    const bool tile_fused = tiled_adj && tile_fuse_ok<T>(p);
    const bool fuse = tile_fused || (direct_sweep && !p.opt.skip_wgrad && hc != -1 &&
                                     (p.opt.fuse_wgrad == 1 || (p.opt.fuse_wgrad == 2 && hc == 0)));
    const int K = (p.opt.tile_k == 8 && p.hc != 0) ? 4 : p.opt.tile_k;
    out[0] = family(false, false, out[5]);
    out[1] = family(true, fuse, out[6]);
    out[2] = fuse ? 1 : 0;
    out[3] = out[0] == 1 ? K : 1;
    out[4] = out[1] == 1 ? K : 1;
    out[7] = (out[0] == 3 || out[1] == 3) ? brick_nt_for(p, vec, out[1] == 3) : 0;     // lanes per brick workgroup (the adjoint's if it runs on bricks)
    for (int i = 8; i < 15; ++i) out[i] = 0;
    // the whole tile sweep as one launch of resident workgroups (needs a device to ask for its CU count: 0 without one)
    if constexpr (sizeof(T) == 4)
        out[14] = ((out[1] == 1 && (persist_ok<T>(p, nullptr, 1 << 20, 1 << 18, nullptr) ||
                                    persist_small_ok<T>(p, nullptr, 1 << 20, 1 << 18, nullptr))) ? 1 : 0) |
                  ((out[0] == 1 && (fwd_persist_ok<T>(p, 1 << 18, nullptr) || fwd_persist_small_by<T>(p, 1 << 18, nullptr) != 0)) ? 2 : 0);
    else
        out[14] = ((out[1] == 1 && persist_ok<T>(p, nullptr, 1 << 20, 1 << 18, nullptr)) ? 1 : 0) |
                  ((out[0] == 1 && fwd_persist_ok<T>(p, 1 << 18, nullptr)) ? 2 : 0);
    {                                                                      // 3D: the resident sweep (pi_res3d.h) where it applies
        Res3dPlan rp;
        if (out[1] == 3 && direct_sweep && fuse && res3d_plan(p, 1 << 10, sizeof(T), rp)) out[14] |= 1;
    }
    for (int dir = 0; dir < 2; ++dir) {                                    // 2D tiles: width, height, lanes per workgroup
        if (out[dir] != 1) continue;
        const TileShape ts = tile_shape_for<T>(p, dir == 1);
        int* o = out + (dir == 1 ? 8 : 11);
        o[0] = ts.bx; o[1] = ts.by;
        o[2] = ts.nt ? ts.nt : (p.opt.tile_k == 2 ? 256 : (p.hc == 0 && p.opt.tile_k == 8) ? 1024 : (p.hc == 0 && p.opt.tile_nt == 1024) ? 1024 :
                                (p.hc == 0 && ts.by == 16) ? 320 : (p.hc == 0 && ts.by == 8) ? 256 : p.opt.tile_nt == 256 ? 256 : 512);
    }
    return 0;
}

}  // namespace

// ---- exported symbols ---------------------------------------------------------------------------
// ---- resident launches of the Stage-1 block (pi_s1_abi.hip, the library's second translation unit) ---------------------------
// They share the residency guard of the 2D resident kernels: one resident grid per device at a time (calls on other streams
// are detected and refused), host-mapped status slots, a per-device scratch (sync words + granule outbox), the handshake on the
// roll call and the "disabled after an abort" state (percnn_pi_persist_status / persist_reset report and re-arm both).
namespace pi_host {
int resident_async_error() { return persist_async_error(); }
int resident_cu_count() { return device_cu_count(); }
// hipSuccess: launch; anything else: take the launch-per-step path (nothing has been enqueued but the memset)
hipError_t resident_begin(void* stream, size_t outbox_bytes, Resident& r)
{
    auto st = static_cast<hipStream_t>(stream);
    Options o;
    { std::lock_guard<std::mutex> lk(g_defaults_mu); o = g_defaults; }
    if (!o.tile_persist || !o.fwd_persist) return hipErrorNotSupported;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return hipErrorNotSupported; }
    if (!persist_enter(st, r.dev)) return hipErrorNotSupported;
    const size_t need = 256 + outbox_bytes;
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        if (g_persist.fwd_scratch_bytes[r.dev] < need) {
            if (g_persist.fwd_scratch[r.dev]) {
                (void)hipFree(g_persist.fwd_scratch[r.dev]);
                g_persist.fwd_scratch[r.dev] = nullptr;
                g_persist.fwd_scratch_bytes[r.dev] = 0;
            }
            void* q = nullptr;
            if (hipMalloc(&q, need) != hipSuccess || !q) { (void)hipGetLastError(); return hipErrorOutOfMemory; }
            g_persist.fwd_scratch[r.dev] = q;
            g_persist.fwd_scratch_bytes[r.dev] = need;
        }
        r.scratch = static_cast<unsigned char*>(g_persist.fwd_scratch[r.dev]);
        r.slot = g_persist.next_slot++ % PERSIST_SLOTS;
        g_persist.watch[r.slot] = false;
    }
    if (hipError_t e = hipMemsetAsync(r.scratch, 0, need, st)) return e;
    r.hs = g_persist.host->slot[r.slot];
    r.hs[0] = 0; r.hs[1] = -1; r.hs[2] = -1; r.hs[3] = 0;
    r.timeout_ticks = (unsigned long long)o.persist_timeout_ms * 100000ull;                 // 100 MHz clock
    r.first_timeout_ticks = (unsigned long long)o.persist_first_timeout_ms * 100000ull;
    return hipSuccess;
}
// after the launch: hipSuccess = resident and running; hipErrorLaunchFailure = it aborted (the caller recomputes launch by launch)
hipError_t resident_launched(void* stream, Resident& r, unsigned grid, const char* what)
{
    auto st = static_cast<hipStream_t>(stream);
    int handshake;
    { std::lock_guard<std::mutex> lk(g_defaults_mu); handshake = g_defaults.persist_handshake; }
    {
        std::lock_guard<std::mutex> lk(g_persist.mu);
        g_persist.watch[r.slot] = true;
        ++g_persist.launches;                               // (counted here: the caller's launch has been enqueued without error)
    }
    hipError_t e = hipSuccess;
    if (handshake) e = persist_wait_roll_call(r.hs, r.slot, r.dev, grid, what);
    if (e == hipSuccess || e == hipErrorLaunchFailure) persist_leave(st, r.dev);
    return e;
}
}  // namespace pi_host

extern "C" {

#ifdef PI_TILE_TIMING
int percnn_pi_debug_stamps(long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pi::pi_tile_stamps), (size_t)n * sizeof(long long));
}
int percnn_pi_debug_wave_stamps(long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pi::pi_tile_wave_stamps), (size_t)n * sizeof(long long));
}
#endif

int percnn_pi_abi_version(void) { return PERCNN_PI_ABI_VERSION; }
size_t percnn_pi_halo_ring_bytes(void) { return sizeof(percnn_pi_halo_ring); }

// ---- peer mailboxes (pi_peer.h) ----
size_t percnn_pi_peer_box_bytes(size_t slot_bytes) { return pi::peer_box_bytes(slot_bytes); }
int percnn_pi_peer_box_alloc(void** box, size_t slot_bytes)
{
    if (!box || !slot_bytes) return PERCNN_PI_EINVAL;
    const size_t total = pi::peer_box_bytes(slot_bytes);
    // fine-grained: stores that arrive over xGMI bypass this device's L2, so its own reads must not be served from there
    if (hipError_t e = hipExtMallocWithFlags(box, total, hipDeviceMallocFinegrained)) return (int)e;
    if (hipError_t e = hipMemset(*box, 0, total)) return (int)e;
    return (int)hipDeviceSynchronize();
}
int percnn_pi_peer_box_free(void* box) { return box ? (int)hipFree(box) : 0; }
int percnn_pi_peer_box_export(void* box, void* handle64)
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t");
    if (!box || !handle64) return PERCNN_PI_EINVAL;
    hipIpcMemHandle_t h;
    if (hipError_t e = hipIpcGetMemHandle(&h, box)) return (int)e;
    std::memcpy(handle64, &h, sizeof(h));
    return 0;
}
int percnn_pi_peer_box_open(const void* handle64, void** mapped)
{
    if (!handle64 || !mapped) return PERCNN_PI_EINVAL;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof(h));
    return (int)hipIpcOpenMemHandle(mapped, h, hipIpcMemLazyEnablePeerAccess);
}
int percnn_pi_peer_box_close(void* mapped) { return mapped ? (int)hipIpcCloseMemHandle(mapped) : 0; }
int percnn_pi_peer_box_status(const void* box, uint64_t* error_epoch, void* stream)
{
    if (!box || !error_epoch) return PERCNN_PI_EINVAL;
    auto st = static_cast<hipStream_t>(stream);
    unsigned long long e = 0;
    if (hipError_t rc = hipMemcpyAsync(&e, &static_cast<const pi::PeerBox*>(box)->error[0], sizeof(e), hipMemcpyDeviceToHost, st))
        return (int)rc;
    if (hipError_t rc = hipStreamSynchronize(st)) return (int)rc;
    *error_epoch = e;
    return 0;
}
int percnn_pi_peer_exchange_f32(float* slab, int ndim, const int64_t* shape, int halo, int width, percnn_pi_peer_ring* ring,
                                void* stream)
{ return peer_exchange_impl<float>(slab, ndim, shape, halo, width, ring, stream); }
int percnn_pi_peer_exchange_f64(double* slab, int ndim, const int64_t* shape, int halo, int width, percnn_pi_peer_ring* ring,
                                void* stream)
{ return peer_exchange_impl<double>(slab, ndim, shape, halo, width, ring, stream); }

// Host-only view of the direct kernels' block decomposition (set_blockmap / direct_rz) for a shape -- what tests/
// test_host_logic.py checks without a GPU: out = {lxs (-1 = flat), nxb, nrg, nblk, rz chosen for the adjoint, block}.
int percnn_pi_debug_blockmap(int ndim, const int64_t* shape, int elem_size, const char* options, int* out)
{
    Problem p;
    if (int rc = make_problem(0, ndim, shape, false, p, options)) return rc;
    if (!out || (elem_size != 4 && elem_size != 8)) return PERCNN_PI_EINVAL;
    const int vec = (p.W % (16 / elem_size) == 0 && p.opt.vec != 1) ? 16 / elem_size : 1;
    Geom g = make_geom(p);
    const int block = direct_block(p, g, vec);
    const int rz = elem_size == 4 ? direct_rz<float>(p, vec, true) : direct_rz<double>(p, vec, true);
    if (!set_blockmap(g, p.ndim, vec, block, (size_t)elem_size, p.opt.l2_tile_kb * 1024, rz, (long)p.opt.l2_tile_min_kb * 1024,
                      p.opt.lane_x)) return PERCNN_PI_EINVAL;
    out[0] = g.lxs; out[1] = g.nxb; out[2] = g.nrg; out[3] = (int)g.nblk; out[4] = rz; out[5] = block;
    return 0;
}

int percnn_pi_debug_plan(int hc, int ndim, const int64_t* shape, int elem_size, const char* options, int* out)
{
    if (!out || (elem_size != 4 && elem_size != 8)) return PERCNN_PI_EINVAL;
    return elem_size == 4 ? debug_plan_impl<float>(hc, ndim, shape, options, out) : debug_plan_impl<double>(hc, ndim, shape, options, out);
}

size_t percnn_pi_param_count(int hc) { return hc < -1 ? 0 : (size_t)pi::nparams(hc); }

#define PI_CONTRACT(SUF, T)                                                                                             \
    int percnn_pi_contract_fwd_##SUF(const T* params, int hc, T* poly, void* stream)                                   \
    {                                                                                                                   \
        if (!params || !poly || hc < 1) return PERCNN_PI_EINVAL;                                                        \
        hipLaunchKernelGGL((pi::pi_contract_fwd_kernel<T>), dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream),    \
                           params, hc, poly);                                                                           \
        return (int)hipGetLastError();                                                                                  \
    }                                                                                                                   \
    int percnn_pi_contract_bwd_##SUF(const T* params, int hc, const T* g_poly, T* g_params, void* stream)              \
    {                                                                                                                   \
        if (!params || !g_poly || !g_params || hc < 1) return PERCNN_PI_EINVAL;                                         \
        hipLaunchKernelGGL((pi::pi_contract_bwd_kernel<T>), dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),   \
                           params, hc, g_poly, g_params);                                                               \
        return (int)hipGetLastError();                                                                                  \
    }
PI_CONTRACT(f32, float)
PI_CONTRACT(f64, double)
#undef PI_CONTRACT

namespace {
bool pack_ptrs(const percnn_pi_param_ptrs* in, bool need_all, pi::PackPtrs& out)
{
    static_assert(sizeof(percnn_pi_param_ptrs) == sizeof(pi::PackPtrs), "pointer tables");
    if (!in) return false;
    out.c[0] = in->c[0]; out.c[1] = in->c[1]; out.w = in->w;
    for (int i = 0; i < 16; ++i) out.br[i] = in->branch[i];
    if (!need_all) return true;
    if (!out.c[0] || !out.c[1] || !out.w) return false;
    for (int i = 0; i < 16; ++i) if (!out.br[i]) return false;
    return true;
}
}  // namespace

#define PI_PACK(SUF, T)                                                                                                 \
    int percnn_pi_pack_fwd_##SUF(const percnn_pi_param_ptrs* params, int hc, int ndim, double dt, double mu_up,         \
                                 int sigmoid, int contract, T* block, void* stream)                                     \
    {                                                                                                                   \
        pi::PackPtrs pp;                                                                                                \
        if (!pack_ptrs(params, true, pp) || !block || hc < 1 || hc > pi::CONTRACT_LDS_HC || (ndim != 2 && ndim != 3))  \
            return PERCNN_PI_EINVAL;                                                                                    \
        hipLaunchKernelGGL((pi::pi_pack_fwd_kernel<T>), dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream), pp, hc, \
                           ndim, dt, mu_up, sigmoid, contract, block, pi::PackGuard{1.0, 1.0, nullptr, 0.0});           \
        return (int)hipGetLastError();                                                                                  \
    }                                                                                                                   \
    int percnn_pi_pack_fwd_guard_##SUF(const percnn_pi_param_ptrs* params, int hc, int ndim, double dt, double mu_up,   \
                                       int sigmoid, int contract, T* block, double u_max, double v_max,                 \
                                       double* host_slot, double seq, void* stream)                                     \
    {                                                                                                                   \
        pi::PackPtrs pp;                                                                                                \
        if (!pack_ptrs(params, true, pp) || !block || hc < 1 || hc > pi::CONTRACT_LDS_HC || (ndim != 2 && ndim != 3) || \
            !(u_max >= 0.0) || !(v_max >= 0.0))                                                                         \
            return PERCNN_PI_EINVAL;                                                                                    \
        hipLaunchKernelGGL((pi::pi_pack_fwd_kernel<T>), dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream), pp, hc, \
                           ndim, dt, mu_up, sigmoid, contract, block, pi::PackGuard{u_max, v_max, host_slot, seq});     \
        return (int)hipGetLastError();                                                                                  \
    }                                                                                                                   \
    int percnn_pi_pack_bwd_##SUF(const percnn_pi_param_ptrs* params, const percnn_pi_param_ptrs* grads, int hc, int ndim, \
                                 double dt, double mu_up, int sigmoid, int contract, const T* g_block, void* stream)    \
    {                                                                                                                   \
        pi::PackPtrs pp, gp;                                                                                            \
        if (!pack_ptrs(params, true, pp) || !pack_ptrs(grads, false, gp) || !g_block || hc < 1 ||                       \
            hc > pi::CONTRACT_LDS_HC || (ndim != 2 && ndim != 3))                                                       \
            return PERCNN_PI_EINVAL;                                                                                    \
        hipLaunchKernelGGL((pi::pi_pack_bwd_kernel<T>), dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream), pp, gp, \
                           hc, ndim, dt, mu_up, sigmoid, contract, g_block);                                            \
        return (int)hipGetLastError();                                                                                  \
    }
PI_PACK(f32, float)
PI_PACK(f64, double)
#undef PI_PACK

size_t percnn_pi_bwd_workspace_bytes(int hc, int ndim, const int64_t* shape, int elem_size)
{
    Problem p;
    if (make_problem(hc, ndim, shape, false, p, nullptr, false) || (elem_size != 4 && elem_size != 8)) return 0;
    // sized for the padded slab layout as well: (n0+4) planes
    Problem q = p;
    q.n = (p.n0 + 4) * p.n1 * p.W;
    return workspace_bytes(q, elem_size);
}

size_t percnn_pi_rollout_bwd_workspace_bytes(int hc, int ndim, const int64_t* shape, int T_steps, int elem_size)
{
    Problem p;
    if (make_problem(hc, ndim, shape, false, p, nullptr, false) || (elem_size != 4 && elem_size != 8) || T_steps < 0) return 0;
    // a per-call `tile_by` override (percnn_pi_rollout_bwd_opt_*) may pick a tile height whose resident sweep needs the
    // outbox although the default height does not: size for every height a call may ask for
    size_t need = rollout_workspace_bytes(p, T_steps, elem_size);
    for (int by : {8, 16}) {
        Problem q = p;
        q.opt.tile_by = by;
        need = std::max(need, rollout_workspace_bytes(q, T_steps, elem_size));
    }
    return need;
}

int percnn_pi_set_option(const char* key, long value)
{
    std::lock_guard<std::mutex> lk(g_defaults_mu);
    return apply_option(g_defaults, key, value);
}

// host-mapped words the device can write and the host can read without synchronising (pack guard slots)
int percnn_pi_host_words_alloc(void** p, size_t bytes)
{
    if (!p || bytes == 0) return PERCNN_PI_EINVAL;
    void* hp = nullptr;
    if (hipError_t e = hipHostMalloc(&hp, bytes, hipHostMallocMapped | hipHostMallocCoherent)) return (int)e;
    std::memset(hp, 0, bytes);
    *p = hp;
    return 0;
}
int percnn_pi_host_words_free(void* p) { return p ? (int)hipHostFree(p) : 0; }

// diagnostics (tests of the persistent sweep's abort path): `blocks` workgroups that each hold `lds_bytes` of a CU's LDS for
// `ms` milliseconds on `stream` -- what "another kernel holds whole CUs" looks like
namespace {
__global__ void pi_debug_hog_kernel(unsigned long long ticks, int* started)
{
    extern __shared__ unsigned char hog_lds[];
    hog_lds[threadIdx.x] = (unsigned char)threadIdx.x;
    // roll call into host-mapped words (plain system-scope stores, no PCIe atomics): the host returns from the entry point
    // only once every hog workgroup holds its CU -- otherwise whether the hog or the kernel under test is dispatched first
    // is up to the queue scheduler and the test it serves is a coin toss
    if (threadIdx.x == 0 && started) __hip_atomic_store(started + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (hog_lds[(threadIdx.x + 1) % 64] == 255 && started) started[blockIdx.x] = 2;
}
}  // namespace
int percnn_pi_debug_hog(int blocks, int lds_bytes, int ms, void* stream)
{
    if (blocks < 1 || blocks > 4096 || lds_bytes < 64 || lds_bytes > 160 * 1024 || ms < 1 || ms > 10000) return PERCNN_PI_EINVAL;
    if (hipError_t e = allow_lds(pi_debug_hog_kernel, (size_t)lds_bytes)) return (int)e;
    static int* started = nullptr;                          // (tests only: one hog at a time)
    if (!started) {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 4096 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            hp = nullptr;
        }
        started = static_cast<int*>(hp);
    }
    if (started) std::memset(started, 0, (size_t)blocks * sizeof(int));
    hipLaunchKernelGGL(pi_debug_hog_kernel, dim3((unsigned)blocks), dim3(64), (size_t)lds_bytes, static_cast<hipStream_t>(stream),
                       (unsigned long long)ms * 100000ull, started);
    if (hipError_t e = hipGetLastError()) return (int)e;
    if (started) {                                          // wait (bounded: 2 s) until every hog workgroup is resident
        const auto t0 = std::chrono::steady_clock::now();
        volatile int* vs = started;
        for (int b = 0; b < blocks;) {
            if (vs[b] != 0) { ++b; continue; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
        }
    }
    return 0;
}

int percnn_pi_persist_status(long* info)
{
    if (!info) return PERCNN_PI_EINVAL;
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16;
    if (!have_dev) (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_persist.mu);
    info[0] = g_persist.launches;
    info[1] = g_persist.aborts;
    info[2] = have_dev && g_persist.disabled[dev] ? 1 : 0;
    info[3] = g_persist.last_group;
    info[4] = g_persist.last_tile;
    // state word of the most recent launch as the device left it: 0 not started, 1 every workgroup resident, 2 aborted
    if (g_persist.host && g_persist.next_slot > 0) {
        volatile int* hs = g_persist.host->slot[(g_persist.next_slot - 1) % PERSIST_SLOTS];
        info[5] = hs[3] ? 2 : hs[0];
    } else info[5] = -1;
    return 0;
}

// Fence for callers that run the resident launches fire-and-forget (persist_handshake = 0) and hand their outputs to code
// that is not this library: waits for `stream`, then reports -- once -- whether a resident launch aborted since anybody looked
int percnn_pi_persist_fence(void* stream)
{
    if (hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream))) return (int)e;
    return persist_async_error();
}

#define PI_EXPORT(SUF, T)                                                                                           \
    int percnn_pi_step_fwd_##SUF(const T* h, T* out, const T* params, int hc, int ndim, const int64_t* shape,      \
                                 void* stream)                                                                      \
    { return step_fwd_impl<T>(h, out, params, hc, ndim, shape, stream, false); }                                    \
    int percnn_pi_slab_step_fwd_##SUF(const T* h, T* out, const T* params, int hc, int ndim, const int64_t* shape, \
                                      int halo, int skip, void* stream)                                             \
    { return step_fwd_impl<T>(h, out, params, hc, ndim, shape, stream, true, halo, skip); }                                     \
    int percnn_pi_step_bwd_##SUF(const T* h, const T* g_out, const T* g_inject, T* g_in, double* param_grad,       \
                                 void* workspace, size_t workspace_bytes, const T* params, int hc, int ndim,       \
                                 const int64_t* shape, void* stream)                                                \
    { return step_bwd_impl<T>(h, g_out, g_inject, g_in, param_grad, workspace, workspace_bytes, params, hc, ndim,  \
                              shape, stream, false); }                                                              \
    int percnn_pi_slab_step_bwd_##SUF(const T* h, const T* g_out, const T* g_inject, T* g_in, double* param_grad,  \
                                      void* workspace, size_t workspace_bytes, const T* params, int hc, int ndim,  \
                                      const int64_t* shape, int halo, int flags, void* stream)                      \
    { return step_bwd_impl<T>(h, g_out, g_inject, g_in, param_grad, workspace, workspace_bytes, params, hc, ndim,  \
                              shape, stream, true, halo, flags); }                                                  \
    int percnn_pi_slab_step_fwd_range_##SUF(const T* h, T* out, const T* params, int hc, int ndim,                 \
                                            const int64_t* shape, int halo, int lo, int hi, void* stream)          \
    { return slab_step_fwd_range_impl<T>(h, out, params, hc, ndim, shape, halo, lo, hi, stream); }                  \
    int percnn_pi_slab_step_bwd_range_##SUF(const T* h, const T* g_out, const T* g_inject, T* g_in,                \
                                            double* param_grad, void* workspace, size_t workspace_bytes,           \
                                            const T* params, int hc, int ndim, const int64_t* shape, int halo,     \
                                            int lo, int hi, int flags, void* stream)                                \
    { return step_bwd_impl<T>(h, g_out, g_inject, g_in, param_grad, workspace, workspace_bytes, params, hc, ndim,  \
                              shape, stream, true, halo, flags, lo, hi); }                                          \
    int percnn_pi_slab_rollout_fwd_##SUF(T* traj, const T* params, int hc, int ndim, const int64_t* shape, int halo, \
                                         int T_steps, const percnn_pi_halo_ring* ring, int overlap, void* stream)   \
    { return slab_rollout_fwd_impl<T>(traj, params, hc, ndim, shape, halo, T_steps, ring, overlap, stream); }       \
    int percnn_pi_slab_rollout_bwd_##SUF(const T* traj, const T* g_traj, T* adj, double* param_grad, void* workspace, \
                                         size_t workspace_bytes, const T* params, int hc, int ndim,                \
                                         const int64_t* shape, int halo, int T_steps,                               \
                                         const percnn_pi_halo_ring* ring, int overlap, void* stream)                \
    { return slab_rollout_bwd_impl<T>(traj, g_traj, adj, param_grad, workspace, workspace_bytes, params, hc, ndim,  \
                                      shape, halo, T_steps, ring, overlap, stream); }                               \
    int percnn_pi_slab_wgrad_##SUF(const T* traj, const T* adj, double* param_grad, void* workspace,               \
                                   size_t workspace_bytes, const T* params, int hc, int ndim, const int64_t* shape, \
                                   int halo, int T_steps, void* stream)                                             \
    { return slab_wgrad_impl<T>(traj, adj, param_grad, workspace, workspace_bytes, params, hc, ndim, shape, halo,  \
                                T_steps, stream); }                                                               \
    int percnn_pi_rollout_fwd_##SUF(T* traj, const T* params, int hc, int ndim, const int64_t* shape, int T_steps, \
                                    void* stream)                                                                   \
    { return rollout_fwd_impl<T>(traj, params, hc, ndim, shape, T_steps, stream); }                                 \
    int percnn_pi_rollout_bwd_##SUF(const T* traj, const T* g_traj, const unsigned char* frame_mask, T* g_h0,      \
                                    double* param_grad, void* workspace, size_t workspace_bytes, const T* params,  \
                                    int hc, int ndim, const int64_t* shape, int T_steps, void* stream)              \
    { return rollout_bwd_impl<T>(traj, g_traj, frame_mask, g_h0, param_grad, workspace, workspace_bytes, params,   \
                                 hc, ndim, shape, T_steps, stream); }                                               \
    /* the same four calls with per-call tuning overrides ("key=value,..."; NULL / "" = process defaults) */        \
    int percnn_pi_step_fwd_opt_##SUF(const T* h, T* out, const T* params, int hc, int ndim, const int64_t* shape,  \
                                     const char* options, void* stream)                                             \
    { return step_fwd_impl<T>(h, out, params, hc, ndim, shape, stream, false, 2, 0, options); }                     \
    int percnn_pi_step_bwd_opt_##SUF(const T* h, const T* g_out, const T* g_inject, T* g_in, double* param_grad,   \
                                     void* workspace, size_t workspace_bytes, const T* params, int hc, int ndim,   \
                                     const int64_t* shape, const char* options, void* stream)                       \
    { return step_bwd_impl<T>(h, g_out, g_inject, g_in, param_grad, workspace, workspace_bytes, params, hc, ndim,  \
                              shape, stream, false, 2, 0, -1, -1, options); }                                       \
    int percnn_pi_step_bwd_rows_##SUF(const T* h, const T* g_out, const T* g_inject, T* g_in, void* workspace,      \
                                      size_t workspace_bytes, const T* params, int hc, int ndim,                    \
                                      const int64_t* shape, int flags, void* stream)                                \
    { return step_bwd_impl<T>(h, g_out, g_inject, g_in, nullptr, workspace, workspace_bytes, params, hc, ndim,     \
                              shape, stream, false, 2, (flags & PERCNN_PI_NO_RESET) | PERCNN_PI_NO_FINISH); }       \
    int percnn_pi_bwd_rows_finish_##SUF(void* workspace, size_t workspace_bytes, int hc, int ndim,                  \
                                        const int64_t* shape, double* param_grad, void* stream)                     \
    { return bwd_rows_finish_impl<T>(workspace, workspace_bytes, hc, ndim, shape, param_grad, stream); }            \
    int percnn_pi_rollout_fwd_opt_##SUF(T* traj, const T* params, int hc, int ndim, const int64_t* shape,          \
                                        int T_steps, const char* options, void* stream)                             \
    { return rollout_fwd_impl<T>(traj, params, hc, ndim, shape, T_steps, stream, options); }                        \
    int percnn_pi_rollout_bwd_top_##SUF(const T* traj, const T* g_traj, const T* g_top, const unsigned char* frame_mask, \
                                        T* g_h0, double* param_grad, void* workspace, size_t workspace_bytes,      \
                                        const T* params, int hc, int ndim, const int64_t* shape, int T_steps,      \
                                        const char* options, void* stream)                                          \
    { return rollout_bwd_impl<T>(traj, g_traj, frame_mask, g_h0, param_grad, workspace, workspace_bytes, params,   \
                                 hc, ndim, shape, T_steps, stream, options, nullptr, g_top); }                      \
    int percnn_pi_rollout_bwd_opt_##SUF(const T* traj, const T* g_traj, const unsigned char* frame_mask, T* g_h0,  \
                                        double* param_grad, void* workspace, size_t workspace_bytes,               \
                                        const T* params, int hc, int ndim, const int64_t* shape, int T_steps,      \
                                        const char* options, void* stream)                                          \
    { return rollout_bwd_impl<T>(traj, g_traj, frame_mask, g_h0, param_grad, workspace, workspace_bytes, params,   \
                                 hc, ndim, shape, T_steps, stream, options); }

PI_EXPORT(f32, float)
PI_EXPORT(f64, double)

// squared-error losses differentiated inside the sweep (include/percnn_pi.h)
#define PI_EXPORT_LOSS(SUF, T)                                                                                      \
    int percnn_pi_rollout_bwd_sqerr_##SUF(const T* traj, const T* target, const unsigned char* frame_mask, double scale, \
                                          const T* dev_scale, T* g_h0, double* param_grad, void* workspace,        \
                                          size_t workspace_bytes, const T* params, int hc, int ndim,               \
                                          const int64_t* shape, int T_steps, const char* options, void* stream)    \
    {                                                                                                               \
        const pi::LossInj l{scale, dev_scale, target ? 2 : 1};                                                      \
        return rollout_bwd_impl<T>(traj, target, frame_mask, g_h0, param_grad, workspace, workspace_bytes, params, \
                                   hc, ndim, shape, T_steps, stream, options, &l);                                  \
    }                                                                                                               \
    int percnn_pi_traj_sqerr_##SUF(const T* traj, const T* target, const unsigned char* frame_mask, int nframes,   \
                                   int ndim, const int64_t* shape, double scale, T* out, void* workspace,          \
                                   size_t workspace_bytes, void* stream)                                            \
    { return sqerr_impl<T>(traj, target, frame_mask, nframes, ndim, shape, scale, out, workspace, workspace_bytes, stream); }

PI_EXPORT_LOSS(f32, float)
PI_EXPORT_LOSS(f64, double)

#define PI_EXPORT_RES(SUF, T)                                                                                       \
    int percnn_pi_residual_fwd_##SUF(const T* traj, T* resid, const T* params, int ndim, const int64_t* shape,     \
                                     int nframes, void* stream)                                                     \
    { return residual_impl<T, false>(traj, nullptr, resid, params, ndim, shape, nframes, stream); }                 \
    int percnn_pi_residual_bwd_##SUF(const T* traj, const T* g_resid, T* g_state, const T* params, int ndim,       \
                                     const int64_t* shape, int nframes, void* stream)                               \
    { return residual_impl<T, true>(traj, g_resid, g_state, params, ndim, shape, nframes, stream); }

PI_EXPORT_RES(f32, float)
PI_EXPORT_RES(f64, double)

size_t percnn_pi_residual_sqloss_workspace_bytes(void) { return RESLOSS_SLOTS * sizeof(double); }

#define PI_EXPORT_RESLOSS(SUF, T)                                                                                   \
    int percnn_pi_residual_sqloss_##SUF(const T* traj, const T* params, int ndim, const int64_t* shape, int nframes, \
                                        int weighted, T* loss_out, void* workspace, size_t workspace_bytes,         \
                                        void* stream)                                                               \
    { return residual_sqloss_impl<T>(traj, params, ndim, shape, nframes, weighted, loss_out, workspace,             \
                                     workspace_bytes, stream); }                                                    \
    int percnn_pi_residual_sqloss_bwd_##SUF(const T* traj, const T* g_loss, const T* params, int ndim,              \
                                            const int64_t* shape, int nframes, int nout_frames, int weighted,       \
                                            T* scratch, T* g_traj, void* stream)                                    \
    { return residual_sqloss_bwd_impl<T>(traj, g_loss, params, ndim, shape, nframes, nout_frames, weighted, scratch, \
                                         g_traj, stream); }

PI_EXPORT_RESLOSS(f32, float)
PI_EXPORT_RESLOSS(f64, double)

}  // extern "C"
