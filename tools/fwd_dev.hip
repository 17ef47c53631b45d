// fwd_dev.hip -- development harness for the 2D forward tile kernel (round 4): variants of pi_fwd2d_tile_kernel are A/B-timed on the
// headline geometry (512^2, float32 pre-contracted block, K = 4 steps per launch, XCD-aware tile map) and their whole trajectory
// is checked bit for bit against the shipped variant's; -DPI_TILE_TIMING adds the device timeline of a launch.  Not part of the product; build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o fwd_dev tools/fwd_dev.hip && ./fwd_dev [T=1000] [reps=5]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../percnn_amd/csrc/pi_tile2d.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

namespace {
constexpr int N = 512, B = 32, K = 4;
using TL = pi::Tile<K, B, B>;
struct Rng {
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    float uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xFFFFFF) / 16777216.0f; }
};
template <int NT>
void launch(float* frames, long fs, const float* P, const pi::TileGeom& g, int T, hipStream_t st)
{
    auto* k = pi::pi_fwd2d_tile_kernel<float, pi::POLY, K, B, B, NT>;
    const size_t lds = (size_t)4 * TL::PLANE * sizeof(float) + 32;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
    for (int t = 0; t + K <= T; t += K)
        hipLaunchKernelGGL(k, dim3((N / B) * (N / B)), dim3(NT), lds, st, frames + (size_t)t * fs, fs, P, g);
}
}  // namespace

int main(int argc, char** argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 1000;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    const size_t frame = (size_t)2 * N * N;
    const int np = 36;
    std::vector<float> hP(np, 0.0f);
    Rng r;
    hP[0] = 0.1f; hP[1] = 0.02f; hP[2] = 0.03f; hP[3] = -5.0f;
    const float taps[4] = {-1.0f / 12, 4.0f / 3, 4.0f / 3, -1.0f / 12};
    for (int a = 0; a < 3; ++a) for (int i = 0; i < 4; ++i) hP[4 + 4 * a + i] = taps[i] + 0.01f * (r.uni() - 0.5f);
    for (int i = 16; i < 36; ++i) hP[i] = 0.2f * (r.uni() - 0.5f);
    std::vector<float> h0(frame);
    for (auto& x : h0) x = r.uni();
    float *dP, *dA, *dB;
    CK(hipMalloc(&dP, np * sizeof(float)));
    CK(hipMalloc(&dA, (size_t)(T + 1) * frame * sizeof(float)));
    CK(hipMalloc(&dB, (size_t)(T + 1) * frame * sizeof(float)));
    CK(hipMemcpy(dP, hP.data(), np * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, h0.data(), frame * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, h0.data(), frame * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const pi::TileGeom g{N, N, (long)N * N, N / B, 2, 8, 4, pi::LossInj{0.0, nullptr, 0}};
    const long fs = (long)frame;
    // reference trajectory: the shipped variant
    launch<512>(dA, fs, dP, g, T, st);
    CK(hipStreamSynchronize(st));
    std::vector<float> ref((size_t)(T + 1) * frame), out((size_t)(T + 1) * frame);
    CK(hipMemcpy(ref.data(), dA, ref.size() * sizeof(float), hipMemcpyDeviceToHost));
    auto check = [&](const char* name) {
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        CK(hipMemcpy(out.data(), dB, out.size() * sizeof(float), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < out.size(); ++i) bad += std::memcmp(&ref[i], &out[i], 4) != 0;
        std::printf("%-22s %zu of %zu trajectory values differ from the shipped variant\n", name, bad, out.size());
        CK(hipMemsetAsync(dB + frame, 0xFF, (size_t)T * frame * sizeof(float), st));
    };
    launch<1024>(dB, fs, dP, g, T, st); check("1024 lanes");
    for (int rep = 0; rep < reps; ++rep) {
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0, st));
            if (v == 0) launch<512>(dB, fs, dP, g, T, st);
            if (v == 1) launch<1024>(dB, fs, dP, g, T, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[v], e0, e1));
        }
        const int nl = T / K;
        std::printf("round %d: us per launch of %d steps: 512 lanes %.2f | 1024 lanes %.2f   (us per step %.3f | %.3f)\n", rep, K, 1e3 * ms[0] / nl,
                    1e3 * ms[1] / nl, 1e3 * ms[0] / T, 1e3 * ms[1] / T);
    }
    // ---- the whole rollout as ONE launch of resident workgroups (pi_fwd2d_persist_kernel) ----
    {
        constexpr int NT = 512, BAND = 2 * (B * B - (B - 16) * (B - 16));
        const int tiles = (N / B) * (N / B), ngroups = T / K;
        const size_t outbox_bytes = (size_t)2 * tiles * BAND * sizeof(unsigned long long);
        unsigned long long* outbox; unsigned* sync; int* host;
        CK(hipMalloc(&outbox, outbox_bytes));
        CK(hipMalloc(&sync, 64));
        CK(hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocCoherent));
        auto* kp = pi::pi_fwd2d_persist_kernel<float, K, B, B, NT>;
        const size_t lds_p = pi::tile_state_bytes<float, K, B, B>() + (size_t)pi::PERSIST_SPLIT_TABLE_ROWS * NT * sizeof(int) + 16;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
        auto run_p = [&]() {
            CK(hipMemsetAsync(outbox, 0, outbox_bytes, st));
            CK(hipMemsetAsync(sync, 0, 64, st));
            host[0] = 0; host[3] = 0;
            pi::PersistArgs pa{};
            pa.outbox = outbox; pa.sync = sync; pa.host = host; pa.ngroups = ngroups;
            pa.timeout_ticks = 200000000ull; pa.first_timeout_ticks = 200000000ull;
            hipLaunchKernelGGL(kp, dim3(tiles), dim3(NT), lds_p, st, dB, fs, dP, g, pa);
        };
        run_p();
        check("persistent forward");
        std::printf("  host state: roll call %d, aborted %d\n", host[0], host[3]);
        for (int rep = 0; rep < reps; ++rep) {
            float ms0, ms1;
            CK(hipEventRecord(e0, st)); launch<512>(dB, fs, dP, g, T, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms0, e0, e1));
            CK(hipEventRecord(e0, st)); run_p(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms1, e0, e1));
            std::printf("persistent round %d: us per group of %d steps: launch per group %.2f | one resident launch %.2f   (us per step %.3f | %.3f)\n",
                        rep, K, 1e3 * ms0 / ngroups, 1e3 * ms1 / ngroups, 1e3 * ms0 / T, 1e3 * ms1 / T);
        }
    }
#ifdef PI_PERSIST_STAMPS
    {   // device timeline of one group of the resident launch: medians over the workgroups of wave 0's / wave 7's stamps
        static long long hs[256 * 8 * 16];
        CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(pi::pi_persist_stamps), sizeof(hs)));
        const char* names[9] = {"group start", "P0", "P1 computed", "ring in LDS", "P2", "P3", "P4", "P5", "published"};
        for (int w = 0; w < 8; w += 7) {
            std::printf("wave %d:", w);
            for (int i = 0; i < 9; ++i) {
                std::vector<double> v;
                for (int b = 0; b < 256; ++b) v.push_back((hs[(b * 8 + w) * 16 + i] - hs[(b * 8 + 0) * 16 + 0]) * 0.01);
                std::sort(v.begin(), v.end());
                std::printf(" %s %.2f", names[i], v[128]);
            }
            std::printf("\n");
        }
        // what pass P3 (A_1: one strip on waves 0-3, the rest of level 1 stored by waves 4-7) is made of, per wave: us since the
        // barrier that ended P2 -- frame stores issued | strip computed and stored to LDS | through the barrier
        for (int w = 0; w < 8; ++w) {
            std::printf("P3 wave %d:", w);
            for (int i : {9, 10, 5}) {
                std::vector<double> v;
                for (int b = 0; b < 256; ++b) v.push_back((hs[(b * 8 + w) * 16 + i] - hs[(b * 8 + w) * 16 + 4]) * 0.01);
                std::sort(v.begin(), v.end());
                std::printf("  %s %.2f", i == 9 ? "stores issued" : i == 10 ? "strip done" : "barrier passed", v[128]);
            }
            std::printf("\n");
        }
    }
#endif
    // the same 250 launches as ONE graph launch (does a captured chain shorten the dependent-kernel boundary?)
    {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        launch<512>(dB, fs, dP, g, T, st);
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < reps; ++rep) {
            float ms;
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(exec, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("graph round %d: %.2f us per launch of %d steps (%.3f us per step)\n", rep, 1e3 * ms / (T / K), K, 1e3 * ms / T);
        }
        check("512 lanes, graph");
    }
#ifdef PI_TILE_TIMING
    // device timeline of the shipped variant (-DPI_TILE_TIMING): per-workgroup 100 MHz stamps of the LAST launch, medians over the
    // workgroups, microseconds since the workgroup's start; "boundary" = start - latest end of the previous launch
    {
        launch<512>(dB, fs, dP, g, T, st);
        CK(hipStreamSynchronize(st));
        static long long hs[4096 * 16];
        CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(pi::pi_tile_stamps), sizeof(hs)));
        const int nb = (N / B) * (N / B);
        long long prev_end = 0, first_start = 1ll << 62;
        for (int b = 0; b < nb; ++b) { if (hs[b * 16 + 14] > prev_end) prev_end = hs[b * 16 + 14]; if (hs[b * 16 + 0] < first_start) first_start = hs[b * 16 + 0]; }
        const char* names[16] = {"start", "window in LDS", "s0 computed", "s0 barrier+store", "s1 computed", "s1 barrier+store", "s2 computed", "s2 barrier+store",
                                 "s3 computed", "s3 barrier+store", "", "", "", "", "", "end"};
        for (int i : {1, 2, 3, 4, 5, 6, 7, 8, 9, 15}) {
            std::vector<double> v;
            for (int b = 0; b < nb; ++b) v.push_back((hs[b * 16 + i] - hs[b * 16 + 0]) * 0.01);
            std::sort(v.begin(), v.end());
            std::printf("  %-18s median %.2f  p90 %.2f  max %.2f\n", names[i], v[nb / 2], v[nb * 9 / 10], v[nb - 1]);
        }
        std::vector<double> st0, en;
        for (int b = 0; b < nb; ++b) { st0.push_back((hs[b * 16 + 0] - prev_end) * 0.01); en.push_back((hs[b * 16 + 15] - first_start) * 0.01); }
        std::sort(st0.begin(), st0.end()); std::sort(en.begin(), en.end());
        std::printf("  start - latest end of the previous launch: median %.2f  min %.2f  max %.2f;  last end - first start %.2f\n", st0[nb / 2], st0[0],
                    st0[nb - 1], en[nb - 1]);
    }
#endif
    return 0;
}
