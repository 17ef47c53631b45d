// C-ABI of the Stage-1 Pi-block (include/percnn_pi_stage1.h); second translation unit of libpercnn_pi.so.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

#include "../../include/percnn_pi.h"
#include "../../include/percnn_pi_stage1.h"
#include "pi_s1.h"
#include "pi_host.h"

namespace {

using pi::s1::Geom;

struct S1Options {
    int etile = 1;      // hand the input-gradient contributions over as 8x8 footprint tiles when H, W are multiples of 4
    int skip_wgrad = 0; // diagnostics: run the adjoint sweep only (parameter gradients are returned as zeros)
    int persist = 1;    // whole rollouts (forward: frames 1..T; sweep: t = T..0) as ONE launch of resident waves where every
                        // (patch, species) task fits on the device at once (s1_fwd_persist_kernel / s1_adj_persist_kernel);
                        // 0 = one launch per step.  Residency guard, handshake and abort -> fallback: pi_host.h
    int persist_min_steps = 8;
    int pause_fwd = 24; // s_sleep units between publishing a step's granules and asking for the neighbours' (asked for too early
                        // they come back stale and cost a second round trip: 100^2 forward 4.45 us per step at 0, 3.4 at 24, 3.9 at 48)
    int pause_adj = 0;  // ... of the sweep: no effect measured (0 .. 32) -- it waits for the tasks that share a SIMD anyway
} g_s1;

constexpr int MAX_GRID_X = 384;       // workgroups per species (x 2 species = 3 per CU); waves grid-stride over patches beyond that

bool make_geom(const int64_t* shape, Geom& g)
{
    if (!shape || shape[0] < 8 || shape[1] < 8 || shape[0] > (1 << 15) || shape[1] > (1 << 15)) return false;
    g.H = (int)shape[0];
    g.W = (int)shape[1];
    g.px = (g.W + 3) / 4;
    g.npy = (g.H + 3) / 4;
    g.npatch = g.npy * g.px;
    g.n = (long)g.H * g.W;
    return true;
}

unsigned grid_x(const Geom& g)
{
    const int need = (g.npatch + pi::s1::WAVES - 1) / pi::s1::WAVES;
    return (unsigned)(need < MAX_GRID_X ? need : MAX_GRID_X);
}

hipError_t step_fwd(const float* h, float* out, const float* P, const Geom& g, hipStream_t st)
{
    hipLaunchKernelGGL(pi::s1::s1_fwd_kernel, dim3(grid_x(g), 2), dim3(64 * pi::s1::WAVES), 0, st, h, out, P, g);
    return hipGetLastError();
}

// ---- resident rollouts ---------------------------------------------------------------------------------------
// one task per wave: every workgroup of the grid must be on the device at once
template <typename K>
bool resident_fits(const Geom& g, int T, K* kernel, unsigned& gx)
{
    if (!g_s1.persist || T < g_s1.persist_min_steps || T >= 4096 || g.H % 4 || g.W % 4) return false;
    gx = (unsigned)((g.npatch + pi::s1::WAVES - 1) / pi::s1::WAVES);
    // workgroups of this kernel one CU holds: asked once per device and kernel (a template instance per kernel; benign race, same
    // value) -- these calls exist to take microseconds of host time off a rollout
    static int blocks_per_cu[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return false; }
    if (!blocks_per_cu[dev]) {
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kernel, 64 * pi::s1::WAVES, 0) != hipSuccess || q < 1) {
            (void)hipGetLastError();
            q = -1;
        }
        blocks_per_cu[dev] = q;
    }
    const int nb = blocks_per_cu[dev];
    if (nb < 1) return false;
    const int cus = pi_host::resident_cu_count();
    return cus > 0 && 2u * gx <= (unsigned)cus * (unsigned)(nb < 4 ? nb : 4);
}

pi::s1::ResArgs resident_args(const pi_host::Resident& r, unsigned nwg, int pause)
{
    pi::s1::ResArgs ra{};
    ra.outbox = reinterpret_cast<unsigned long long*>(r.scratch + 256);
    ra.sync = reinterpret_cast<unsigned*>(r.scratch);
    ra.host = const_cast<int*>(r.hs);
    ra.timeout_ticks = r.timeout_ticks;
    ra.first_timeout_ticks = r.first_timeout_ticks;
    ra.nwg = (int)nwg;
    ra.pause = pause;
    return ra;
}

// hipSuccess: frames 1..T are being written by one resident launch; hipErrorLaunchTimeOut: fatal; else: run the per-step path
hipError_t rollout_fwd_resident(float* traj, const float* P, const Geom& g, int T, hipStream_t st)
{
    unsigned gx = 0;
    if (!resident_fits(g, T, pi::s1::s1_fwd_persist_kernel, gx)) return hipErrorNotSupported;
    pi_host::Resident r;
    if (hipError_t e = pi_host::resident_begin(st, 2 * pi::s1::res_fwd_half(g) * sizeof(unsigned long long), r)) return e;
    const pi::s1::ResArgs ra = resident_args(r, 2 * gx, g_s1.pause_fwd);
    hipLaunchKernelGGL(pi::s1::s1_fwd_persist_kernel, dim3(gx, 2), dim3(64 * pi::s1::WAVES), 0, st, traj, T, P, g, ra);
    if (hipError_t e = hipGetLastError()) return e;
    return pi_host::resident_launched(st, r, 2 * gx, "the resident Stage-1 forward rollout");
}

hipError_t sweep_resident(const float* traj, const float* g_traj, const unsigned char* frame_mask, float* adj, float* g_h0,
                          const float* P, const Geom& g, int T, hipStream_t st)
{
    unsigned gx = 0;
    if (!resident_fits(g, T, pi::s1::s1_adj_persist_kernel, gx)) return hipErrorNotSupported;
    pi_host::Resident r;
    if (hipError_t e = pi_host::resident_begin(st, 2 * pi::s1::res_adj_half(g) * sizeof(unsigned long long), r)) return e;
    pi::s1::ResArgs ra = resident_args(r, 2 * gx, g_s1.pause_adj);
    if (frame_mask) {
        ra.masked = 1;
        for (int t = 0; t <= T; ++t)
            if (frame_mask[t]) ra.frames[t >> 5] |= 1u << (t & 31);
    }
    hipLaunchKernelGGL(pi::s1::s1_adj_persist_kernel, dim3(gx, 2), dim3(64 * pi::s1::WAVES), 0, st, traj, g_traj, adj, g_h0, T, P, g, ra);
    if (hipError_t e = hipGetLastError()) return e;
    return pi_host::resident_launched(st, r, 2 * gx, "the resident Stage-1 adjoint sweep");
}

constexpr int WGRAD_GRID_X = 256;     // x 2 species = 2 workgroups per CU; one float row of partials each

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Workspace {
    float* adj;        // [T+1][2][n]
    float* D;          // [2 (step parity)][2 species][50 taps][n]
    float* partials;   // [WGRAD_GRID_X][2][ROW]
    double* partials_d;// [WGRAD_GRID_X][2][ROWD]
    size_t bytes;
};

Workspace carve(void* base, const Geom& g, int T)
{
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t nfloat) { float* p = base ? reinterpret_cast<float*>((char*)base + off) : nullptr;
                                     off += align_up(nfloat * sizeof(float), 256); return p; };
    w.adj = take((size_t)(T + 1) * 2 * g.n);
    w.D = take((size_t)2 * 2 * pi::s1::NTAP * g.n);
    w.partials = take((size_t)WGRAD_GRID_X * 2 * pi::s1::ROW);
    w.partials_d = reinterpret_cast<double*>(take((size_t)WGRAD_GRID_X * 2 * pi::s1::ROWD * 2));
    w.bytes = off;
    return w;
}

hipError_t adj_step(const float* h_prev, const float* inj, const float* adj_next, const float* D_next, float* adj_out,
                    float* D_out, const float* P, const Geom& g, hipStream_t st)
{
    const bool etile = g_s1.etile && g.H % 4 == 0 && g.W % 4 == 0;
    if (etile)
        hipLaunchKernelGGL(pi::s1::s1_adj_kernel<true>, dim3(grid_x(g), 2), dim3(64 * pi::s1::WAVES), 0, st, h_prev, inj,
                           adj_next, D_next, adj_out, D_out, P, g);
    else
        hipLaunchKernelGGL(pi::s1::s1_adj_kernel<false>, dim3(grid_x(g), 2), dim3(64 * pi::s1::WAVES), 0, st, h_prev, inj,
                           adj_next, D_next, adj_out, D_out, P, g);
    return hipGetLastError();
}

}  // namespace

extern "C" {

size_t percnn_pi_s1_param_count(void) { return pi::s1::NP; }

int percnn_pi_s1_set_option(const char* key, long value)
{
    if (!key) return PERCNN_PI_EINVAL;
    if (!std::strcmp(key, "etile")) { g_s1.etile = value != 0; return 0; }
    if (!std::strcmp(key, "skip_wgrad")) { g_s1.skip_wgrad = value != 0; return 0; }
    if (!std::strcmp(key, "persist")) { g_s1.persist = value != 0; return 0; }
    if (!std::strcmp(key, "persist_min_steps") && value >= 1) { g_s1.persist_min_steps = (int)value; return 0; }
    if (!std::strcmp(key, "pause_fwd") && value >= 0 && value <= 4096) { g_s1.pause_fwd = (int)value; return 0; }
    if (!std::strcmp(key, "pause_adj") && value >= 0 && value <= 4096) { g_s1.pause_adj = (int)value; return 0; }
    return PERCNN_PI_EINVAL;
}

int percnn_pi_s1_step_fwd_f32(const float* h, float* h_next, const float* params, const int64_t* shape, void* stream)
{
    Geom g;
    if (!h || !h_next || !params || h == h_next || !make_geom(shape, g)) return PERCNN_PI_EINVAL;
    return (int)step_fwd(h, h_next, params, g, static_cast<hipStream_t>(stream));
}

int percnn_pi_s1_rollout_fwd_f32(float* traj, const float* params, const int64_t* shape, int T_steps, void* stream)
{
    Geom g;
    if (!traj || !params || T_steps < 0 || !make_geom(shape, g)) return PERCNN_PI_EINVAL;
    if (int rc = pi_host::resident_async_error()) return rc;
    const size_t frame = (size_t)2 * g.n;
    {
        const hipError_t e = rollout_fwd_resident(traj, params, g, T_steps, static_cast<hipStream_t>(stream));
        if (e == hipSuccess) return 0;
        if (e == hipErrorLaunchTimeOut) return (int)e;
        (void)hipGetLastError();                            // not eligible / not resident / aborted: launch by launch (deterministic)
    }
    for (int t = 0; t < T_steps; ++t)
        if (hipError_t e = step_fwd(traj + t * frame, traj + (t + 1) * frame, params, g, static_cast<hipStream_t>(stream)))
            return (int)e;
    return 0;
}

size_t percnn_pi_s1_rollout_bwd_workspace_bytes(const int64_t* shape, int T_steps)
{
    Geom g;
    if (T_steps < 0 || !make_geom(shape, g)) return 0;
    return carve(nullptr, g, T_steps).bytes;
}

int percnn_pi_s1_rollout_bwd_f32(const float* traj, const float* g_traj, const unsigned char* frame_mask, float* g_h0,
                                 double* param_grad, void* workspace, size_t workspace_bytes, const float* params,
                                 const int64_t* shape, int T_steps, void* stream)
{
    Geom g;
    if (!traj || !g_traj || !g_h0 || !param_grad || !params || T_steps < 0 || !make_geom(shape, g))
        return PERCNN_PI_EINVAL;
    const Workspace w = carve(workspace, g, T_steps);
    if (!workspace || workspace_bytes < w.bytes || (reinterpret_cast<uintptr_t>(workspace) % 16)) return PERCNN_PI_EWORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    const size_t frame = (size_t)2 * g.n, dsz = (size_t)2 * pi::s1::NTAP * g.n;
    auto inj = [&](int t) { return (!frame_mask || frame_mask[t]) ? g_traj + t * frame : nullptr; };
    auto D = [&](int t) { return w.D + (size_t)(t & 1) * dsz; };
    if (int rc = pi_host::resident_async_error()) return rc;
    bool swept = false;
    if (T_steps >= 1) {
        const hipError_t e = sweep_resident(traj, g_traj, frame_mask, w.adj, g_h0, params, g, T_steps, st);
        if (e == hipSuccess) swept = true;
        else if (e == hipErrorLaunchTimeOut) return (int)e;
        else (void)hipGetLastError();
    }
    for (int t = swept ? -1 : T_steps; t >= 0; --t) {
        const bool top = t == T_steps;
        hipError_t e = adj_step(t > 0 ? traj + (size_t)(t - 1) * frame : nullptr, inj(t),
                                top ? nullptr : w.adj + (size_t)(t + 1) * frame, top ? nullptr : D(t + 1),
                                t > 0 ? w.adj + (size_t)t * frame : g_h0, t > 0 ? D(t) : nullptr, params, g, st);
        if (e) return (int)e;
    }
    if (g_s1.skip_wgrad) return (int)hipMemsetAsync(param_grad, 0, sizeof(double) * pi::s1::NP, st);
    const long ntask = (long)T_steps * g.npatch;
    long gx = (ntask + pi::s1::WAVES - 1) / pi::s1::WAVES;
    gx = gx < 1 ? 1 : (gx > WGRAD_GRID_X ? WGRAD_GRID_X : gx);
    hipLaunchKernelGGL(pi::s1::s1_wgrad_kernel, dim3((unsigned)gx, 2), dim3(64 * pi::s1::WAVES), 0, st, traj, w.adj,
                       w.partials, w.partials_d, params, g, T_steps);
    if (hipError_t e = hipGetLastError()) return (int)e;
    hipLaunchKernelGGL(pi::s1::s1_reduce_kernel, dim3((pi::s1::NP + 63) / 64), dim3(64, 4), 0, st, w.partials, w.partials_d,
                       (int)gx, param_grad);
    return (int)hipGetLastError();
}

#ifdef PI_S1_TIMING
int percnn_pi_s1_debug_stamps(long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pi::s1::s1_stamps), (size_t)n * sizeof(long long));
}
#endif

}  // extern "C"
