// step3d_probe.hip -- standalone probe of the direct 3D step kernels at ONE shape (default 128^3, float32, pre-contracted
// block): per-step time of kernel variants in a ping-pong rollout, bit-identity between variants, and (built with
// -DPI_3D_TIMING) a per-wave device timeline.  The host side re-states just enough of pi_abi.hip's set_blockmap for a
// power-of-two row (whole rows per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DPI_3D_TIMING] -o step3d_probe step3d_probe.hip && ./step3d_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../percnn_amd/csrc/pi_kernels.h"
#include "../../percnn_amd/csrc/pi_brick3d.h"

#define CK(x) do { hipError_t e_ = (x); if (e_) { printf("HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static pi::FastDiv fastdiv(unsigned d)
{
    pi::FastDiv f{0u, 0u};
    if (d <= 1) return f;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.m = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    f.s = l - 1;
    return f;
}

static pi::Geom geom(int n0, int n1, int W, int block, int rz, bool lw)
{
    pi::Geom g;
    memset(&g, 0, sizeof g);
    g.n0 = n0; g.n1 = n1; g.W = W; g.rows = n0 * n1; g.s0 = (long)n1 * W; g.ss = (long)n0 * g.s0; g.off = 0; g.wrap0 = 1;
    const int cpr = W / 4;
    int lxs = 0;
    while ((1 << lxs) < cpr) ++lxs;
    if ((1 << lxs) != cpr || cpr > block) { printf("probe: W/4 must be a power of two <= block\n"); exit(1); }
    g.lxs = lxs; g.nxb = 1;
    const int rpb = block >> lxs;
    g.nrg = (n1 + rpb - 1) / rpb;
    g.rz = rz;
    g.nblk = (unsigned)(g.nxb * g.nrg * ((n0 + rz - 1) / rz));
    g.dnxb = fastdiv(1); g.dnrg = fastdiv((unsigned)g.nrg); g.dcpr = fastdiv((unsigned)cpr); g.dn1 = fastdiv((unsigned)n1);
    g.fastdiv = 1;
    (void)lw;
    return g;
}

struct Variant { const char* name; int rz; bool lw; int block; bool brick = false; bool wt = false; };

static pi::BrickGeom brick_geom(int n0, int n1, int W, int rz)
{
    pi::BrickGeom g;
    memset(&g, 0, sizeof g);
    g.n0 = n0; g.n1 = n1; g.cpr = W / 4; g.total = n1 * g.cpr;
    g.nrg = (g.total + 255) / 256;
    g.nblk = (unsigned)(g.nrg * ((n0 + rz - 1) / rz));
    g.wrap0 = 1; g.s0 = (long)n1 * W; g.ss = (long)n0 * g.s0; g.off = 0;
    g.dnrg = fastdiv((unsigned)g.nrg); g.dcpr = fastdiv((unsigned)g.cpr);
    g.nseg = (4 * g.cpr + 63) / 64; g.ntask = 2 * rz * g.nseg; g.dnseg = fastdiv((unsigned)g.nseg);
    return g;
}
template <int RZ>
static void launch_brick(const float* h, float* out, const float* P, const pi::BrickGeom& g, hipStream_t st)
{
    hipLaunchKernelGGL((pi::pi_fwd3d_brick_kernel<float, pi::POLY, RZ>), dim3(g.nblk), dim3(256), (size_t)2 * RZ * pi::BRICK_WB, st, h, out, P, g, 0, pi::PeerPutFused{});
}

template <int RZ>
static void launch(const float* h, float* out, const float* P, const pi::Geom& g, int block, hipStream_t st)
{
    hipLaunchKernelGGL((pi::pi_fwd_kernel<float, 3, pi::POLY, 4, RZ>), dim3(g.nblk), dim3(block), 0, st, h, out, P, g, 0);
}

static int gN0, gN1, gW;
static void run(const Variant& v, const float* h, float* out, const float* P, const pi::Geom& g, hipStream_t st)
{
    if (v.brick) {
        pi::BrickGeom bg = brick_geom(gN0, gN1, gW, v.rz);
        bg.wt = v.wt;
        if (v.rz == 1) launch_brick<1>(h, out, P, bg, st);
        if (v.rz == 2) launch_brick<2>(h, out, P, bg, st);
        if (v.rz == 4) launch_brick<4>(h, out, P, bg, st);
        return;
    }
    if (v.rz == 1) launch<1>(h, out, P, g, v.block, st);
    if (v.rz == 2) launch<2>(h, out, P, g, v.block, st);
    if (v.rz == 4) launch<4>(h, out, P, g, v.block, st);
}

int main(int argc, char** argv)
{
    const int n0 = argc > 1 ? atoi(argv[1]) : 128, n1 = argc > 2 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 128;
    gN0 = n0; gN1 = n1; gW = W;
    const int T = argc > 4 ? atoi(argv[4]) : 100;
    const size_t n = (size_t)2 * n0 * n1 * W;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // frames of ONE trajectory buffer, as the rollout writes them (every step stores into memory no launch has touched);
    // PROBE_PINGPONG=1: two buffers (stores hit lines that are still in the caches -- 1.5 us per step faster at 128^3)
    const bool pingpong = getenv("PROBE_PINGPONG") != nullptr;
    float *traj, *P;
    CK(hipMalloc(&traj, (size_t)(pingpong ? 2 : T + 1) * n * 4)); CK(hipMalloc(&P, 64 * 4));
    float* a = traj;
    std::vector<float> h0(n), hp(64, 0.f);
    srand(1);
    for (auto& x : h0) x = 0.4f + 0.2f * (float)rand() / RAND_MAX;
    // dt, coefficients, star taps of a 4th-order Laplacian / dx^2, small cubic coefficients: magnitudes of the Gray-Scott block
    hp[pi::P_DT] = 0.5f; hp[pi::P_COEF] = 0.1f; hp[pi::P_COEF + 1] = 0.05f;
    const float idx2 = 0.2304f;
    hp[pi::P_C0] = -7.5f * idx2;
    for (int ax = 0; ax < 3; ++ax) { hp[pi::P_TAPS + 4 * ax + 0] = hp[pi::P_TAPS + 4 * ax + 3] = -idx2 / 12; hp[pi::P_TAPS + 4 * ax + 1] = hp[pi::P_TAPS + 4 * ax + 2] = idx2 * 4 / 3; }
    for (int i = 0; i < 20; ++i) hp[pi::P_W + i] = 0.01f * (float)((i * 7) % 11 - 5);
    CK(hipMemcpy(P, hp.data(), 64 * 4, hipMemcpyHostToDevice));
    std::vector<Variant> vars = {{"direct rz=1", 1, false, 256}, {"direct rz=2", 2, false, 256}, {"direct rz=4", 4, false, 256},
                                 {"brick rz=1", 1, false, 256, true}, {"brick rz=2", 2, false, 256, true}, {"brick rz=4", 4, false, 256, true},
                                 {"brick rz=1 wt", 1, false, 256, true, true}, {"brick rz=2 wt", 2, false, 256, true, true}};
    std::vector<float> first, cur(n);
    for (const auto& v : vars) {
        const pi::Geom g = geom(n0, n1, W, v.block, v.rz, v.lw);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemcpy(a, h0.data(), n * 4, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, st));
            for (int t = 0; t < T; ++t) {
                const float* x = pingpong ? traj + (size_t)(t & 1) * n : traj + (size_t)t * n;
                float* y = pingpong ? traj + (size_t)((t + 1) & 1) * n : traj + (size_t)(t + 1) * n;
                run(v, x, y, P, g, st);
            }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float m; CK(hipEventElapsedTime(&m, e0, e1));
            ms = rep == 0 ? m : std::min(ms, m);
        }
        CK(hipMemcpy(cur.data(), pingpong ? traj : traj + (size_t)T * n, n * 4, hipMemcpyDeviceToHost));   // T even
        if (first.empty()) first = cur;
        const bool same = !memcmp(first.data(), cur.data(), n * 4);
        printf("%-26s %4u blocks x %3d  %7.2f us per step  %s\n", v.name, g.nblk, v.block, ms * 1e3 / T, same ? "bit-identical" : "DIFFERS");
#ifdef PI_3D_TIMING
        {
            static long long st8[4096 * 8 * 8];
            CK(hipMemcpyFromSymbol(st8, HIP_SYMBOL(pi::pi_3d_stamps), sizeof st8));
            const int nb = (int)std::min(g.nblk, 4096u), nw = v.block / 64;
            long long t0 = st8[0];
            for (int bq = 0; bq < nb; ++bq) for (int w = 0; w < nw; ++w) t0 = std::min(t0, st8[(bq * 8 + w) * 8]);
            const char* names[8] = {"start", "requested", "committed", "barrier", "plane0 done", "planes done", "-", "end"};
            for (int s = 0; s < 8; ++s) {
                if (s == 6 || (!v.lw && !v.brick && s >= 1 && s <= 3) || (v.rz == 1 && s == 5)) continue;
                std::vector<double> x;
                for (int bq = 0; bq < nb; ++bq) for (int w = 0; w < nw; ++w) x.push_back((st8[(bq * 8 + w) * 8 + s] - t0) / 100.0);
                std::sort(x.begin(), x.end());
                printf("      %-12s  min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", names[s], x[0], x[x.size() / 10], x[x.size() / 2], x[x.size() * 9 / 10], x.back());
            }
        }
#endif
    }
    // ---- adjoint sweep (brick kernel, fused moments): h = trajectory frames, G ping-pong, dL/dtraj of T+1 frames ----
    if (!pingpong) {
        float *inj, *ga, *gb; double* partials;
        CK(hipMalloc(&inj, (size_t)(T + 1) * n * 4)); CK(hipMalloc(&ga, n * 4)); CK(hipMalloc(&gb, n * 4));
        CK(hipMalloc(&partials, (size_t)4096 * 36 * 8));
        CK(hipMemset(inj, 0, (size_t)(T + 1) * n * 4)); CK(hipMemset(partials, 0, (size_t)4096 * 36 * 8));
        for (int rz : {1, 2}) for (unsigned cap : {4096u, 1024u, 768u, 512u}) for (int noinj : {0, 1}) {
            const int wt = 1;
            pi::BrickGeom bg = brick_geom(n0, n1, W, rz);
            bg.wt = wt;
            if (cap < 4096u && cap >= bg.nblk) continue;
            const unsigned grid = std::min(bg.nblk, cap);
            const size_t lds = 64 + std::max((size_t)2 * rz * pi::BRICK_WB, (size_t)(32 + 20 * 264) * 4);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(ga, 0, n * 4, st));
                CK(hipEventRecord(e0, st));
                float *x = ga, *y = gb;
                for (int t = T; t >= 1; --t) {
                    const float* hf = traj + (size_t)(t - 1) * n;
                    const float* jf = noinj ? nullptr : inj + (size_t)(t - 1) * n;
                    if (rz == 1) hipLaunchKernelGGL((pi::pi_adj3d_brick_kernel<float, pi::POLY, 1, true>), dim3(grid), dim3(256), lds, st, hf, x, jf, y, partials, P, bg, 0, pi::NoPut{});
                    else         hipLaunchKernelGGL((pi::pi_adj3d_brick_kernel<float, pi::POLY, 2, true>), dim3(grid), dim3(256), lds, st, hf, x, jf, y, partials, P, bg, 0, pi::NoPut{});
                    std::swap(x, y);
                }
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
                float m; CK(hipEventElapsedTime(&m, e0, e1));
                ms = rep == 0 ? m : std::min(ms, m);
            }
            printf("adjoint brick rz=%d wt=%d %s  %4u blocks  %7.2f us per step\n", rz, wt, noinj ? "no dL/dtraj" : "dense dL/dtraj", grid, ms * 1e3 / T);
#ifdef PI_3D_TIMING
            {
                static long long st8[4096 * 8 * 8];
                CK(hipMemcpyFromSymbol(st8, HIP_SYMBOL(pi::pi_3d_stamps), sizeof st8));
                const int nb = (int)grid, nw = 4;
                long long t0 = st8[0];
                for (int bq = 0; bq < nb; ++bq) for (int w = 0; w < nw; ++w) t0 = std::min(t0, st8[(bq * 8 + w) * 8]);
                const char* names[8] = {"start", "requested", "committed", "barrier", "plane0 done", "planes done", "sweep done", "end"};
                for (int s2 = 0; s2 < 8; ++s2) {
                    if (rz == 1 && s2 == 5) continue;
                    std::vector<double> x;
                    for (int bq = 0; bq < nb; ++bq) for (int w = 0; w < nw; ++w) x.push_back((st8[(bq * 8 + w) * 8 + s2] - t0) / 100.0);
                    std::sort(x.begin(), x.end());
                    printf("      %-12s  min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", names[s2], x[0], x[x.size() / 10], x[x.size() / 2], x[x.size() * 9 / 10], x.back());
                }
            }
#endif
        }
    }
    return 0;
}
