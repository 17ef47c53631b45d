// persist_dev.hip -- development harness for the persistent 2D tile sweep (round 4).
//
// The persistent kernels compile in seconds on their own (the library's pi_abi.hip takes minutes), so variants of
// pi_adj2d_persist_split_kernel are A/B-timed here against round 3's pi_adj2d_persist_kernel and checked bit for bit against
// the launch-per-group kernel (pi_adj2d_tile_kernel<MOM>) on the headline geometry: 512^2, float32 pre-contracted block,
// groups of four steps, XCD-aware tile map.  Not part of the product; build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DVARIANT_FLAGS] -o persist_dev tools/persist_dev.hip
//   ./persist_dev [T=200] [reps=5]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../percnn_amd/csrc/pi_tile2d.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

namespace {
constexpr int N = 512, B = 32, K = 4, NT = 512;
using TL = pi::Tile<K, B, B>;

pi::TileGeom make_geom()
{
    pi::TileGeom g{N, N, (long)N * N, N / B, 2, 8, 4, pi::LossInj{0.0, nullptr, 0}};   // 2 x 4 rectangles of 8 x 4 tiles
    return g;
}

struct Rng {
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    float uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xFFFFFF) / 16777216.0f; }
};
}  // namespace

int main(int argc, char** argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 200;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    const int ngroups = T / K;
    const size_t frame = (size_t)2 * N * N;
    const int np = 36, tiles = (N / B) * (N / B);
    std::vector<float> hP(np, 0.0f);
    Rng r;
    hP[0] = 0.1f; hP[1] = 0.02f; hP[2] = 0.03f; hP[3] = -5.0f;
    const float taps[4] = {-1.0f / 12, 4.0f / 3, 4.0f / 3, -1.0f / 12};
    for (int a = 0; a < 3; ++a) for (int i = 0; i < 4; ++i) hP[4 + 4 * a + i] = taps[i] + 0.01f * (r.uni() - 0.5f);
    for (int i = 16; i < 36; ++i) hP[i] = 0.2f * (r.uni() - 0.5f);
    std::vector<float> hh((size_t)(T + 1) * frame), hg((size_t)(T + 1) * frame);
    for (auto& x : hh) x = r.uni();
    for (auto& x : hg) x = (r.uni() - 0.5f) * 1e-3f;
    float *dP, *dh, *dg, *da;
    CK(hipMalloc(&dP, np * sizeof(float)));
    CK(hipMalloc(&dh, hh.size() * sizeof(float)));
    CK(hipMalloc(&dg, hg.size() * sizeof(float)));
    CK(hipMalloc(&da, hh.size() * sizeof(float)));
    CK(hipMemcpy(dP, hP.data(), np * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, hh.data(), hh.size() * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), hg.size() * sizeof(float), hipMemcpyHostToDevice));
    double* dpart;
    CK(hipMalloc(&dpart, (size_t)tiles * np * sizeof(double)));
    constexpr int BAND = 2 * (B * B - (B - 16) * (B - 16));
    const size_t outbox_bytes = (size_t)2 * tiles * BAND * sizeof(unsigned long long);
    unsigned long long* outbox;
    CK(hipMalloc(&outbox, outbox_bytes));
    unsigned* sync;
    CK(hipMalloc(&sync, 64));
    int* host;
    CK(hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const pi::TileGeom g = make_geom();
    const long fs = (long)frame;
    const float* hfr = dh + (size_t)T * frame;
    const float* gfr = dg + (size_t)T * frame;
    float* afr = da + (size_t)T * frame;
    const int t_end = T - K * ngroups;

    auto reset = [&]() {
        CK(hipMemsetAsync(dpart, 0, (size_t)tiles * np * sizeof(double), st));
        CK(hipMemcpyAsync(afr, dg + (size_t)T * frame, frame * sizeof(float), hipMemcpyDeviceToDevice, st));   // top frame
    };
    // ---- launch per group (reference) ----
    auto* kt = pi::pi_adj2d_tile_kernel<float, pi::POLY, K, B, B, NT, true>;
    size_t lds_t = (size_t)4 * TL::PLANE * sizeof(float) + 32;
    {
        const size_t tail = (size_t)(2 * (NT / 64) + 20 + 20 * (NT / 64)) * sizeof(double) + (size_t)20 * (NT + 16) * sizeof(float);
        if (tail > lds_t) lds_t = tail;
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    auto run_tile = [&]() {
        for (int grp = 0; grp < ngroups; ++grp) {
            const long go = -(long)grp * K * fs;
            hipLaunchKernelGGL(kt, dim3(tiles), dim3(NT), lds_t, st, hfr + go, gfr + go, afr + go, fs, 0xFu, (float*)nullptr, 0, dpart, np,
                               dP, g);
        }
    };
    // ---- persistent flavours ----
    const size_t lds_p = pi::tile_state_bytes<float, K, B, B>() + (size_t)20 * NT * sizeof(double) + (size_t)pi::PERSIST_SPLIT_TABLE_ROWS * NT * sizeof(int) + 16;
    auto* kp0 = pi::pi_adj2d_persist_kernel<float, K, B, B, NT>;
    auto* kp1 = pi::pi_adj2d_persist_split_kernel<float, K, B, B, NT>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kp0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kp1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    // what-if (timing only, results wrong): argv[3] = "l2" makes every group of the split sweep read the SAME operand frames, i.e. from
    // the L2 instead of HBM -- what the streaming of 16 MB of operands per group costs the sweep
    const long fs_split = (argc > 3 && !std::strcmp(argv[3], "l2")) ? 0 : fs;
    auto run_persist = [&](int split) {
        CK(hipMemsetAsync(outbox, 0, outbox_bytes, st));
        CK(hipMemsetAsync(sync, 0, 64, st));
        host[0] = 0; host[3] = 0;
        pi::PersistArgs pa{};
        pa.outbox = outbox; pa.sync = sync; pa.host = host; pa.ngroups = ngroups;
        pa.timeout_ticks = 200000000ull; pa.first_timeout_ticks = 200000000ull;
        pa.t_top = T; pa.masked = 0;
        if (split) hipLaunchKernelGGL(kp1, dim3(tiles), dim3(NT), lds_p, st, hfr, gfr, afr, fs_split, (float*)nullptr, dpart, np, dP, g, pa);
        else       hipLaunchKernelGGL(kp0, dim3(tiles), dim3(NT), lds_p, st, hfr, gfr, afr, fs, (float*)nullptr, dpart, np, dP, g, pa);
    };

    std::vector<float> ref(frame), out(frame);
    std::vector<double> pref((size_t)tiles * np), pout((size_t)tiles * np);
    auto fetch = [&](std::vector<float>& f, std::vector<double>& p) {
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(f.data(), da + (size_t)t_end * frame, frame * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(p.data(), dpart, p.size() * sizeof(double), hipMemcpyDeviceToHost));
    };
    reset(); run_tile(); fetch(ref, pref);
    CK(hipGetLastError());
    for (int split = 0; split < 2; ++split) {
        CK(hipMemsetAsync(da + (size_t)t_end * frame, 0xFF, frame * sizeof(float), st));
        reset(); run_persist(split); fetch(out, pout);
        CK(hipGetLastError());
        size_t bad = 0;
        for (size_t i = 0; i < frame; ++i) bad += std::memcmp(&ref[i], &out[i], 4) != 0;
        double num = 0, den = 0;
        for (int c = 0; c < np; ++c) {
            double a = 0, b = 0;
            for (int tl = 0; tl < tiles; ++tl) { a += pref[(size_t)tl * np + c]; b += pout[(size_t)tl * np + c]; }
            num += (a - b) * (a - b); den += a * a;
        }
        std::printf("%-14s state[%d]: %zu of %zu values differ from the launch-per-group sweep; gradient sums rel-L2 %.2e; host state %d\n",
                    split ? "persist_split" : "persist", t_end, bad, frame, den > 0 ? std::sqrt(num / den) : 0.0, host[0]);
    }
    // ---- timing: interleaved rounds ----
    for (int rep = 0; rep < reps; ++rep) {
        float ms[3];
        for (int v = 0; v < 3; ++v) {
            reset();
            CK(hipEventRecord(e0, st));
            if (v == 0) run_tile(); else run_persist(v - 1);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[v], e0, e1));
        }
        std::printf("round %d: us per group of %d steps: launch-per-group %.2f | persist %.2f | persist_split %.2f   (us per step %.3f | %.3f | %.3f)\n",
                    rep, K, 1e3 * ms[0] / ngroups, 1e3 * ms[1] / ngroups, 1e3 * ms[2] / ngroups, 1e3 * ms[0] / (ngroups * K),
                    1e3 * ms[1] / (ngroups * K), 1e3 * ms[2] / (ngroups * K));
    }
#ifdef PI_PERSIST_STAMPS
    // device timeline of one group of the split kernel (the last timed launch): medians over the 256 workgroups of wave 0's
    // stamps, microseconds since the group's first stamp
    {
        static long long hs[256 * 8 * 16];
        CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(pi::pi_persist_stamps), sizeof(hs)));
        const char* names[9] = {"group start", "P0", "loads requested", "P1", "ring in LDS", "P2", "P3", "P4", "P5"};
        const char* last = "published";
        for (int w = 0; w < 8; w += 7) {
            std::printf("wave %d:", w);
            for (int i = 0; i < 9; ++i) {
                std::vector<double> v;
                for (int b = 0; b < 256; ++b) v.push_back((hs[(b * 8 + w) * 16 + i] - hs[(b * 8 + 0) * 16 + 0]) * 0.01);
                std::sort(v.begin(), v.end());
                std::printf(" %s %.2f", i == 8 ? last : names[i == 0 ? 0 : (i == 1 ? 1 : (i == 2 ? 3 : (i == 3 ? 4 : i + 1)))], v[128]);
            }
            std::printf("\n");
        }
    }
#endif
    return 0;
}
