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
    if (lw) {
        g.lw_nwin = (unsigned)(rpb * cpr + 4 * cpr);
        g.lw_base = 0;
        g.d4cpr = fastdiv((unsigned)(4 * cpr));
    }
    return g;
}

struct Variant { const char* name; int rz; bool lw; int block; };

template <int RZ, bool LW>
static void launch(const float* h, float* out, const float* P, const pi::Geom& g, int block, hipStream_t st)
{
    const size_t lds = LW ? (size_t)2 * RZ * g.lw_nwin * 16 : 0;
    hipLaunchKernelGGL((pi::pi_fwd_kernel<float, 3, pi::POLY, 4, RZ, LW>), dim3(g.nblk), dim3(block), lds, st, h, out, P, g, 0);
}

static void run(const Variant& v, const float* h, float* out, const float* P, const pi::Geom& g, hipStream_t st)
{
    if (v.rz == 1) { if (v.lw) launch<1, true>(h, out, P, g, v.block, st); else launch<1, false>(h, out, P, g, v.block, st); }
    if (v.rz == 2) { if (v.lw) launch<2, true>(h, out, P, g, v.block, st); else launch<2, false>(h, out, P, g, v.block, st); }
    if (v.rz == 4) { if (v.lw) launch<4, true>(h, out, P, g, v.block, st); else launch<4, false>(h, out, P, g, v.block, st); }
}

int main(int argc, char** argv)
{
    const int n0 = argc > 1 ? atoi(argv[1]) : 128, n1 = argc > 2 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 128;
    const int T = 200;
    const size_t n = (size_t)2 * n0 * n1 * W;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *a, *b, *ref, *P;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&ref, n * 4)); CK(hipMalloc(&P, 64 * 4));
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
                                 {"ldswin rz=1", 1, true, 256}, {"ldswin rz=2", 2, true, 256}, {"ldswin rz=4", 4, true, 256},
                                 {"ldswin rz=1 block=128", 1, true, 128}, {"ldswin rz=2 block=128", 2, true, 128}};
    std::vector<float> first, cur(n);
    for (const auto& v : vars) {
        const pi::Geom g = geom(n0, n1, W, v.block, v.rz, v.lw);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemcpy(a, h0.data(), n * 4, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, st));
            float *x = a, *y = b;
            for (int t = 0; t < T; ++t) { run(v, x, y, P, g, st); std::swap(x, y); }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float m; CK(hipEventElapsedTime(&m, e0, e1));
            ms = rep == 0 ? m : std::min(ms, m);
        }
        CK(hipMemcpy(cur.data(), a, n * 4, hipMemcpyDeviceToHost));     // T even: the last frame is in a
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
                if (s == 6 || (!v.lw && s >= 1 && s <= 3) || (v.rz == 1 && s == 5)) continue;
                std::vector<double> x;
                for (int bq = 0; bq < nb; ++bq) for (int w = 0; w < nw; ++w) x.push_back((st8[(bq * 8 + w) * 8 + s] - t0) / 100.0);
                std::sort(x.begin(), x.end());
                printf("      %-12s  min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", names[s], x[0], x[x.size() / 10], x[x.size() / 2], x[x.size() * 9 / 10], x.back());
            }
        }
#endif
    }
    return 0;
}
