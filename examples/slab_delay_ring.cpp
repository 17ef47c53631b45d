// slab_delay_ring.cpp -- does the slab loops' overlap schedule hide a wire?  
//
// No multi-GPU node has been available to the builder, and "to self" every message arrives as fast as a device copy: nothing to
// hide, so the faces-first / side-stream schedule has only ever shown its own cost.  This host drives the NATIVE slab loops
// (percnn_pi_slab_rollout_fwd / _bwd: the per-rank code of the 8-GPU decomposition of BASELINE configs[4]) on ONE rank of the
// configs[4] slab shape (32 x 256 x 256 float32, halo 4) through a percnn_pi_halo_ring whose four function pointers are C++
// callbacks with RCCL's group semantics and an INJECTED WIRE (see Fabric below): the sends copy the faces into staging buffers on
// the stream the loop hands in, group_end holds that stream for `delay` microseconds (one small spinning kernel: a workgroup slot,
// no bandwidth) and then copies the faces into the halo planes.  prev == next == this rank (a ring of one): sends and receives pair
// up in issue order, RCCL's rule for two operations on the same peer.  Nothing synchronises with the host inside a rollout.
//
// It prints, for delay = 0 and for the budgeted link times of DESIGN.md 6 (2 MiB faces at ~64 GB/s per direction: 30 us per
// forward exchange; 1 MiB: 16 us per adjoint exchange), the time per fwd+bwd step of the PLAIN schedule (exchange between
// steps) and of the OVERLAP schedule (faces first, exchange on a side stream under the interior), and checks that both
// schedules produce the same bits.  tests/test_slab_dist_gpu.py asserts what the numbers must show.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude examples/slab_delay_ring.cpp -Lpercnn_amd/csrc -lpercnn_pi \
//         -Wl,-rpath,$PWD/percnn_amd/csrc -o slab_delay_ring && ./slab_delay_ring [T=40] [reps=5] [fwd_us=30] [bwd_us=16] [halo=4] [ring|peer]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#include "percnn_pi.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)
#define PK(x) do { int r_ = (x); if (r_ != 0) { std::printf("percnn_pi error %d at %s:%d\n", r_, __FILE__, __LINE__); std::exit(1); } } while (0)

namespace {

__global__ void wire_kernel(unsigned long long ticks)          // 100 MHz wall clock
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

struct Message { void* stage; size_t bytes; };
struct PendingRecv { void* buf; size_t bytes; hipStream_t st; };

// The wire model: every operation of a group runs on the stream the caller hands in (RCCL's send / receive kernels do too).  A send
// copies the face into a staging buffer; the receives are held back until group_end, which first holds the stream for `delay`
// microseconds -- ONE interval per exchange: the two directions of a ring exchange use two links and travel concurrently -- and
// then copies the faces into the halo planes.  No other stream, no event: what the caller's schedule does not overlap itself is
// exposed in full, what it puts on a side stream runs beside its compute stream.
struct Fabric {
    std::deque<Message> fifo;                                    // the ordered pair (me -> me)
    std::vector<PendingRecv> pending;
    std::vector<void*> stages;                                   // staging buffers, reused round-robin
    size_t stage_bytes = 0;
    size_t next = 0;
    bool in_group = false;
    double delay_us = 0.0;                                       // per exchange
    long sends = 0, recvs = 0, groups = 0, errors = 0;
} fab;

int g_start() { if (fab.in_group) { ++fab.errors; return 4; } fab.in_group = true; return 0; }

int ring_send(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream)
{
    if (!fab.in_group || peer != 0 || comm != (void*)0xC0FFEE || dtype != 7) { ++fab.errors; return 4; }
    const size_t bytes = count * 4;
    if (bytes > fab.stage_bytes) { ++fab.errors; return 4; }
    void* stage = fab.stages[fab.next++ % fab.stages.size()];
    if (hipMemcpyAsync(stage, buf, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)) != hipSuccess) return 1;
    fab.fifo.push_back(Message{stage, bytes});
    ++fab.sends;
    return 0;
}

int ring_recv(void* buf, size_t count, int dtype, int peer, void* comm, void* stream)
{
    if (!fab.in_group || peer != 0 || comm != (void*)0xC0FFEE || dtype != 7) { ++fab.errors; return 4; }
    fab.pending.push_back(PendingRecv{buf, count * 4, static_cast<hipStream_t>(stream)});
    return 0;
}

int g_end()
{
    if (!fab.in_group) { ++fab.errors; return 4; }
    fab.in_group = false;
    ++fab.groups;
    if (fab.pending.empty()) return 0;
    hipStream_t st = fab.pending.front().st;
    if (fab.delay_us > 0.0) hipLaunchKernelGGL(wire_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)(fab.delay_us * 100.0));
    for (const PendingRecv& r : fab.pending) {                   // receives pair up with the sends in issue order
        if (fab.fifo.empty() || r.st != st) { ++fab.errors; return 4; }
        const Message m = fab.fifo.front();
        fab.fifo.pop_front();
        if (m.bytes != r.bytes) { ++fab.errors; return 4; }
        if (hipMemcpyAsync(r.buf, m.stage, m.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
        ++fab.recvs;
    }
    fab.pending.clear();
    return 0;
}

struct Rng {
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    float uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xFFFFFF) / 16777216.0f; }
};

}  // namespace

int main(int argc, char** argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 40;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    const double fwd_us = argc > 3 ? std::atof(argv[3]) : 30.0, bwd_us = argc > 4 ? std::atof(argv[4]) : 16.0;
    const int halo = argc > 5 ? std::atoi(argv[5]) : 4;             // forward: 2 MiB faces every two steps; adjoint: 1 MiB every step
    // transport: "ring" = the RCCL-shaped callbacks above; "peer" = the peer mailboxes (pi_peer.h) through this device's own
    // mailbox, the wire injected by the library's measurement option peer_wire_us (the put holds its arrival flag back)
    const bool peer = argc > 6 && !std::strcmp(argv[6], "peer");
    const int64_t shape[3] = {32, 256, 256};                     // one rank's interior of 256^3 / 8
    const size_t plane = (size_t)shape[1] * shape[2], padded = (size_t)(shape[0] + 2 * halo) * plane, frame = 2 * padded;
    const int np = (int)percnn_pi_param_count(0);
    if (percnn_pi_abi_version() != PERCNN_PI_ABI_VERSION) { std::printf("ABI mismatch\n"); return 1; }

    std::vector<float> hP(np, 0.0f);
    Rng r;
    hP[0] = 0.05f; hP[1] = 0.02f; hP[2] = 0.01f; hP[3] = -7.5f;
    const float taps[4] = {-1.0f / 12, 4.0f / 3, 4.0f / 3, -1.0f / 12};
    for (int a = 0; a < 3; ++a) for (int i = 0; i < 4; ++i) hP[4 + 4 * a + i] = taps[i];
    for (int i = 16; i < 36; ++i) hP[i] = 0.05f * (r.uni() - 0.5f);
    std::vector<float> h0(frame);
    for (auto& x : h0) x = r.uni();

    float *dP, *traj, *gtraj, *adj;
    double* pg;
    void* ws;
    const size_t ws_bytes = percnn_pi_bwd_workspace_bytes(0, 3, shape, 4);
    CK(hipMalloc(&dP, np * sizeof(float)));
    CK(hipMalloc(&traj, (size_t)(T + 1) * frame * sizeof(float)));
    CK(hipMalloc(&gtraj, (size_t)(T + 1) * frame * sizeof(float)));
    CK(hipMalloc(&adj, (size_t)(T + 1) * frame * sizeof(float)));
    CK(hipMalloc(&pg, np * sizeof(double)));
    CK(hipMalloc(&ws, ws_bytes));
    CK(hipMemcpy(dP, hP.data(), np * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(traj, h0.data(), frame * sizeof(float), hipMemcpyHostToDevice));
    {
        std::vector<float> g(frame);
        for (int t = 0; t <= T; ++t) {
            for (auto& x : g) x = 1e-3f * (r.uni() - 0.5f);
            CK(hipMemcpy(gtraj + (size_t)t * frame, g.data(), frame * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    fab.stage_bytes = (size_t)8 * halo * plane * sizeof(float);   // both species of a face of `halo` planes, with room to spare
    fab.stages.resize(16);
    for (auto& s : fab.stages) CK(hipMalloc(&s, fab.stage_bytes));
    void* pack_stage;
    CK(hipMalloc(&pack_stage, fab.stage_bytes));

    percnn_pi_halo_ring ring;
    std::memset(&ring, 0, sizeof ring);
    if (percnn_pi_halo_ring_bytes() != sizeof ring) { std::printf("halo ring layout mismatch\n"); return 1; }
    ring.comm = (void*)0xC0FFEE; ring.prev = 0; ring.next = 0; ring.dtype_f32 = 7; ring.dtype_f64 = 8;
    ring.group_start = g_start; ring.group_end = g_end; ring.send = ring_send; ring.recv = ring_recv;
    ring.peer = nullptr; ring.stage = pack_stage; ring.stage_bytes = fab.stage_bytes;
    percnn_pi_peer_ring pr;
    std::memset(&pr, 0, sizeof pr);
    if (peer) {
        void* box = nullptr;
        const size_t slot = (size_t)2 * halo * plane * sizeof(float);
        PK(percnn_pi_peer_box_alloc(&box, slot));
        pr.my_box = pr.prev_box = pr.next_box = box;
        pr.slot_bytes = slot;
        ring.peer = &pr;
    }

    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    std::vector<float> ref_last(frame), ref_g0(frame), got(frame);
    bool have_ref = false;
    int mismatches = 0;
    struct Row { const char* name; int overlap; double f_us, b_us; double fwd, bwd; int fused; };
    std::vector<Row> rows = {{"plain, no wire", 0, 0.0, 0.0, 0, 0, 0}, {"overlap, no wire", 1, 0.0, 0.0, 0, 0, 0},
                             {"plain, wire", 0, fwd_us, bwd_us, 0, 0, 0}, {"overlap, wire", 1, fwd_us, bwd_us, 0, 0, 0}};
    if (peer) {                                                  // the puts fused into the step / sweep launches (no side stream)
        rows.push_back({"fused put, no wire", 0, 0.0, 0.0, 0, 0, 1});
        rows.push_back({"fused put, wire", 0, fwd_us, bwd_us, 0, 0, 1});
    }
    for (auto& row : rows) {
        std::vector<double> tf, tb;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CK(hipMemsetAsync(pg, 0, np * sizeof(double), st));
            fab.delay_us = row.f_us;
            if (peer) {
                PK(percnn_pi_set_option("slab_fused_put", row.fused));
                PK(percnn_pi_set_option("slab_fused_put_adj", row.fused));
                PK(percnn_pi_set_option("peer_wire_us", (long)row.f_us));
            }
            CK(hipEventRecord(e0, st));
            PK(percnn_pi_slab_rollout_fwd_f32(traj, dP, 0, 3, shape, halo, T, &ring, row.overlap, st));
            CK(hipEventRecord(e1, st));
            fab.delay_us = row.b_us;
            if (peer) PK(percnn_pi_set_option("peer_wire_us", (long)row.b_us));
            PK(percnn_pi_slab_rollout_bwd_f32(traj, gtraj, adj, pg, ws, ws_bytes, dP, 0, 3, shape, halo, T, &ring, row.overlap, st));
            CK(hipEventRecord(e2, st));
            CK(hipStreamSynchronize(st));
            float a, b;
            CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
            if (rep) { tf.push_back(a * 1e3 / T); tb.push_back(b * 1e3 / T); }
        }
        std::sort(tf.begin(), tf.end()); std::sort(tb.begin(), tb.end());
        row.fwd = tf[tf.size() / 2]; row.bwd = tb[tb.size() / 2];
        // same bits whatever the schedule and the wire: last state frame and dL/dh0 (interior + halo planes as the loops leave them)
        CK(hipMemcpy(got.data(), traj + (size_t)T * frame, frame * sizeof(float), hipMemcpyDeviceToHost));
        if (!have_ref) ref_last = got; else mismatches += std::memcmp(got.data(), ref_last.data(), frame * sizeof(float)) != 0;
        CK(hipMemcpy(got.data(), adj, frame * sizeof(float), hipMemcpyDeviceToHost));
        // (compare the interior planes only: halo planes of the adjoint frame are scratch)
        const size_t lo = (size_t)halo * plane, n_int = (size_t)shape[0] * plane;
        if (!have_ref) ref_g0 = got;
        else for (int s = 0; s < 2; ++s) mismatches += std::memcmp(got.data() + s * padded + lo, ref_g0.data() + s * padded + lo, n_int * sizeof(float)) != 0;
        have_ref = true;
    }
    if (fab.errors || !fab.fifo.empty()) { std::printf("ring protocol errors: %ld, unmatched sends: %zu\n", fab.errors, fab.fifo.size()); return 1; }
    if (peer) {
        unsigned long long bad = 0;
        PK(percnn_pi_peer_box_status(pr.my_box, (uint64_t*)&bad, st));
        if (bad) { std::printf("a take timed out at exchange %llu\n", bad); return 1; }
        PK(percnn_pi_set_option("peer_wire_us", 0));
    }
    const double runs = (double)rows.size() * (reps + 1);
    const double ex_f = fab.groups ? (double)fab.sends / 2.0 / runs / T : 0.0;   // exchanges per time step (fwd + bwd), two sends each
    std::printf("slab 32x256x256 f32, halo %d, T = %d, ring of one, transport %s; messages per fwd+bwd step: %.2f sends (%ld groups)\n", halo, T,
                peer ? "peer mailboxes" : "ring callbacks", (double)fab.sends / runs / T, fab.groups);
    (void)ex_f;
    for (const auto& row : rows)
        std::printf("RESULT %-18s fwd_wire_us %5.1f bwd_wire_us %5.1f | fwd %7.2f bwd %7.2f total %7.2f us per time step\n", row.name, row.f_us, row.b_us,
                    row.fwd, row.bwd, row.fwd + row.bwd);
    std::printf("bitwise: %s\n", mismatches ? "DIFFERENT" : "identical");
    if (mismatches) return 1;
    std::printf("slab_delay_ring ok\n");
    return 0;
}
