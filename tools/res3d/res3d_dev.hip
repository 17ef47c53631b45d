// res3d_dev.hip -- development harness for the RESIDENT 3D rollouts (round 6, percnn_amd/csrc/pi_res3d.h): the one-launch forward
// (and reverse sweep) against a naive launch-per-step kernel of the same operation order -- whole trajectory compared bit for bit,
// interleaved timing, device timeline of a step (-DPI_R3D_STAMPS=<workgroup>).  Not part of the product; build here, run on the box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/scratch/res3d_dev tools/res3d/res3d_dev.hip
//   ./tools/scratch/res3d_dev [N=128] [T=500] [reps=5] [pause=0]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>

#include "../../percnn_amd/csrc/pi_res3d.h"
#ifndef R3D_NT
#define R3D_NT 512
#endif
#define R3D_NT_ R3D_NT

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

namespace {
struct Rng {
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    float uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xFFFFFF) / 16777216.0f; }
};

// naive reference: one point per lane, operands from global memory, pi::star's operation order
__global__ void __launch_bounds__(256) ref_fwd(const float* __restrict__ h, float* __restrict__ out, const float* __restrict__ P, int n0, int n1, int n2)
{
    using namespace pi;
    const long n = (long)n0 * n1 * n2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % n2), y = (int)((i / n2) % n1), z = (int)(i / ((long)n1 * n2));
    float c[2], lap[2];
    for (int s = 0; s < 2; ++s) {
        const float* f = h + s * n;
        auto at = [&](int zz, int yy, int xx) { return f[((long)((zz + n0) % n0) * n1 + (yy + n1) % n1) * n2 + (xx + n2) % n2]; };
        c[s] = at(z, y, x);
        float l = P[P_C0] * c[s];
        for (int t = 0; t < 4; ++t) { const int k = t < 2 ? t - 2 : t - 1; l = fma_(P[P_TAPS + t], at(z + k, y, x), l); }
        for (int t = 0; t < 4; ++t) { const int k = t < 2 ? t - 2 : t - 1; l = fma_(P[P_TAPS + 4 + t], at(z, y + k, x), l); }
        for (int t = 0; t < 4; ++t) { const int k = t < 2 ? t - 2 : t - 1; l = fma_(P[P_TAPS + 8 + t], at(z, y, x + k), l); }
        lap[s] = l;
    }
    const float dt = P[P_DT];
    for (int s = 0; s < 2; ++s) {
        const float rr = poly_r(P + P_W + 10 * s, c[0], c[1]);
        const float res = P[P_COEF + s] * lap[s] + rr;
        const float inc = res * dt;
        out[s * n + i] = c[s] + inc;
    }
}

// naive reference of one adjoint step (pi_adj3d_brick_kernel's operation order), sums by double atomics
__global__ void __launch_bounds__(256) ref_adj(const float* __restrict__ h, const float* __restrict__ G, const float* __restrict__ inj,
                                               float* __restrict__ Gp, double* __restrict__ sums, const float* __restrict__ P, int n0, int n1, int n2)
{
    using namespace pi;
    const long n = (long)n0 * n1 * n2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % n2), y = (int)((i / n2) % n1), z = (int)(i / ((long)n1 * n2));
    const float dt = P[P_DT];
    float gc[2], dl[2];
    for (int s = 0; s < 2; ++s) {
        const float* f = G + s * n;
        auto at = [&](int zz, int yy, int xx) { return f[((long)((zz + n0) % n0) * n1 + (yy + n1) % n1) * n2 + (xx + n2) % n2]; };
        gc[s] = at(z, y, x);
        float l = P[P_C0] * gc[s];
        for (int t = 0; t < 4; ++t) { const int k = -(t < 2 ? t - 2 : t - 1); l = fma_(P[P_TAPS + t], at(z + k, y, x), l); }
        for (int t = 0; t < 4; ++t) { const int k = -(t < 2 ? t - 2 : t - 1); l = fma_(P[P_TAPS + 4 + t], at(z, y + k, x), l); }
        for (int t = 0; t < 4; ++t) { const int k = -(t < 2 ? t - 2 : t - 1); l = fma_(P[P_TAPS + 8 + t], at(z, y, x + k), l); }
        dl[s] = l * dt;
    }
    const float u = h[i], v = h[n + i];
    float du = 0.f, dv = 0.f;
    for (int s = 0; s < 2; ++s) {
        const float gr = gc[s] * dt;
        float ru, rv;
        poly_dr(P + P_W + 10 * s, u, v, ru, rv);
        du = fma_(gr, ru, du);
        dv = fma_(gr, rv, dv);
        atomicAdd(&sums[s], (double)(dl[s] * (s ? v : u)));
        const double g = gr, U = u, V = v;
        const double ph[10] = {1, U, V, U * U, U * V, V * V, U * U * U, U * U * V, U * V * V, V * V * V};
        for (int m = 0; m < 10; ++m) atomicAdd(&sums[2 + 10 * s + m], g * ph[m]);
    }
    const float tu = P[P_COEF + 0] * dl[0] + du, tv = P[P_COEF + 1] * dl[1] + dv;
    float ou = gc[0] + tu, ov = gc[1] + tv;
    if (inj) { ou += inj[i]; ov += inj[n + i]; }
    Gp[i] = ou; Gp[n + i] = ov;
}
}  // namespace

int main(int argc, char** argv)
{
    using namespace pi::r3d;
    const int N = argc > 1 ? std::atoi(argv[1]) : 128;
    const int T = argc > 2 ? std::atoi(argv[2]) : 500;
    const int reps = argc > 3 ? std::atoi(argv[3]) : 5;
    const int pause = argc > 4 ? std::atoi(argv[4]) : 0;
    const int n0 = N, n1 = N, n2 = N;
    const size_t npts = (size_t)n0 * n1 * n2, frame = 2 * npts;
    std::vector<float> hP(36, 0.0f);
    Rng r;
    hP[0] = 0.05f; hP[1] = 0.02f; hP[2] = 0.01f; hP[3] = -7.5f;
    const float taps[4] = {-1.0f / 12, 4.0f / 3, 4.0f / 3, -1.0f / 12};
    for (int a = 0; a < 3; ++a) for (int i = 0; i < 4; ++i) hP[4 + 4 * a + i] = taps[i] + 0.01f * (r.uni() - 0.5f);
    for (int i = 16; i < 36; ++i) hP[i] = 0.05f * (r.uni() - 0.5f);
    std::vector<float> h0(frame);
    for (auto& x : h0) x = r.uni();
    float *dP, *dA, *dB;
    CK(hipMalloc(&dP, 36 * sizeof(float)));
    CK(hipMalloc(&dA, (size_t)(T + 1) * frame * sizeof(float)));
    CK(hipMalloc(&dB, (size_t)(T + 1) * frame * sizeof(float)));
    CK(hipMemcpy(dP, hP.data(), 36 * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, h0.data(), frame * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, h0.data(), frame * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    Args a{};
    a.n0 = n0; a.n1 = n1; a.n2 = n2;
    a.gz = n0 / BZ; a.gy = n1 / BY; a.gx = n2 / BX;
    a.ss = (long)npts; a.frame_stride = (long)frame;
    a.rz = a.ry = a.rx = 1;
    if (const char* e = std::getenv("R3D_REGIONS")) std::sscanf(e, "%d,%d,%d", &a.rz, &a.ry, &a.rx);
    if (const char* e = std::getenv("R3D_SKIP")) a.skip = std::atoi(e);
    const int nwg = a.gz * a.gy * a.gx;
    const size_t obytes = (size_t)2 * nwg * Shape<R3D_NT_>::BOX_BYTES;
    void* outbox;
    unsigned* sync;
    unsigned long long* stamps;
    CK(hipMalloc(&outbox, obytes));
    CK(hipMalloc(&sync, 64));
    CK(hipMalloc(&stamps, 64 * sizeof(unsigned long long)));
    CK(hipMemset(stamps, 0, 64 * sizeof(unsigned long long)));
    a.outbox = outbox; a.sync = sync; a.host = nullptr;
    a.nsteps = T; a.pause = pause;
    a.timeout_ticks = 2000000ull; a.first_timeout_ticks = 2000000ull;      // 20 ms
#ifndef R3D_NT
#define R3D_NT 512
#endif
    constexpr int NT = R3D_NT;
    constexpr int BOX_BYTES = Shape<NT>::BOX_BYTES;
    const size_t lds = LDS_BYTES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pi_fwd3d_resident_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pi_fwd3d_resident_kernel<NT>, NT, lds));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("regions %d x %d x %d; skip %d; ", a.rz, a.ry, a.rx, a.skip);
    std::printf("grid %d^3, %d workgroups of %d lanes, LDS %zu B, occupancy %d per CU x %d CUs, outbox %.1f MB\n", N, nwg, NT, lds, occ, prop.multiProcessorCount, obytes / 1e6);
    if (occ * prop.multiProcessorCount < nwg) { std::printf("does not fit\n"); return 1; }

    // reference trajectory
    for (int t = 0; t < T; ++t)
        hipLaunchKernelGGL(ref_fwd, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, dA + (size_t)t * frame, dA + (size_t)(t + 1) * frame, dP, n0, n1, n2);
    CK(hipStreamSynchronize(st));

    auto run = [&]() {
        CK(hipMemsetAsync(outbox, 0, obytes, st));
        CK(hipMemsetAsync(sync, 0, 64, st));
        hipLaunchKernelGGL(pi_fwd3d_resident_kernel<NT>, dim3(nwg), dim3(NT), lds, st, dB, dP, a, stamps);
    };
    run();
    CK(hipStreamSynchronize(st));
    unsigned hs[4];
    CK(hipMemcpy(hs, sync, sizeof hs, hipMemcpyDeviceToHost));
    std::printf("sync: started %u abort %u timeouts %u\n", hs[0], hs[1], hs[2]);
    // compare
    {
        std::vector<float> A(frame), B(frame);
        size_t bad_total = 0;
        int first_bad = -1;
        for (int t = 1; t <= T; t += (t < 8 ? 1 : std::max(1, T / 16))) {
            CK(hipMemcpy(A.data(), dA + (size_t)t * frame, frame * sizeof(float), hipMemcpyDeviceToHost));
            CK(hipMemcpy(B.data(), dB + (size_t)t * frame, frame * sizeof(float), hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < frame; ++i) bad += std::memcmp(&A[i], &B[i], 4) != 0;
            if (bad && first_bad < 0) {
                first_bad = t;
                {   // where inside a block do the differences sit?
                    std::vector<size_t> hz(BZ, 0), hy(BY, 0), hx(BX, 0);
                    for (size_t i = 0; i < frame; ++i)
                        if (std::memcmp(&A[i], &B[i], 4) != 0) {
                            const size_t p = i % npts;
                            ++hz[(p / ((size_t)n1 * n2)) % BZ]; ++hy[((p / n2) % n1) % BY]; ++hx[(p % n2) % BX];
                        }
                    std::printf("  %zu differences in frame %d; by local z:", bad, t);
                    for (auto c : hz) std::printf(" %zu", c);
                    std::printf("\n  by local y:");
                    for (auto c : hy) std::printf(" %zu", c);
                    std::printf("\n  by local x:");
                    for (auto c : hx) std::printf(" %zu", c);
                    std::printf("\n");
                }
                int shown = 0;
                for (size_t i = 0; i < frame && shown < 8; ++i)
                    if (std::memcmp(&A[i], &B[i], 4) != 0) {
                        const size_t p = i % npts;
                        std::printf("  frame %d s %zu z %zu y %zu x %zu: ref %.9g got %.9g\n", t, i / npts, p / ((size_t)n1 * n2), (p / n2) % n1, p % n2, A[i], B[i]);
                        ++shown;
                    }
            }
            bad_total += bad;
        }
        {   // last frame always
            CK(hipMemcpy(A.data(), dA + (size_t)T * frame, frame * sizeof(float), hipMemcpyDeviceToHost));
            CK(hipMemcpy(B.data(), dB + (size_t)T * frame, frame * sizeof(float), hipMemcpyDeviceToHost));
            size_t bad = 0;
            double nrm = 0;
            for (size_t i = 0; i < frame; ++i) { bad += std::memcmp(&A[i], &B[i], 4) != 0; nrm += (double)A[i] * A[i]; }
            std::printf("last frame: %zu differing values, |ref|^2 = %.6g\n", bad, nrm);
            bad_total += bad;
        }
        std::printf("bitwise: %s (first bad frame %d)\n", bad_total ? "DIFFERENT" : "identical", first_bad);
    }
    // timing
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) {
        CK(hipMemsetAsync(outbox, 0, obytes, st));
        CK(hipMemsetAsync(sync, 0, 64, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(pi_fwd3d_resident_kernel<NT>, dim3(nwg), dim3(NT), lds, st, dB, dP, a, stamps);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float m;
        CK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    std::printf("resident forward: median %.3f ms = %.3f us per step (min %.3f); frame bytes per step %.1f MB -> %.2f TB/s of stores\n", ms[ms.size() / 2],
                1e3 * ms[ms.size() / 2] / T, 1e3 * ms[0] / T, frame * 4 / 1e6, frame * 4 / (1e9 * ms[ms.size() / 2] / T));
    ms.clear();
    for (int i = 0; i < std::min(reps, 3); ++i) {
        CK(hipEventRecord(e0, st));
        for (int t = 0; t < T; ++t)
            hipLaunchKernelGGL(ref_fwd, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, dA + (size_t)t * frame, dA + (size_t)(t + 1) * frame, dP, n0, n1, n2);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float m;
        CK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    std::printf("naive launch per step: %.3f us per step\n", 1e3 * ms[ms.size() / 2] / T);
#ifdef PI_R3D_STAMPS
    unsigned long long hst[16];
    CK(hipMemcpy(hst, stamps, sizeof hst, hipMemcpyDeviceToHost));
    std::printf("timeline of step 20, workgroup %d (us from its start): S strips %.2f | barrier %.2f | published %.2f | I strips %.2f | barrier %.2f | written back %.2f | ring landed + unpacked %.2f\n",
                PI_R3D_STAMPS, (hst[1] - hst[0]) / 100.0, (hst[2] - hst[0]) / 100.0, (hst[3] - hst[0]) / 100.0, (hst[4] - hst[0]) / 100.0, (hst[5] - hst[0]) / 100.0, (hst[6] - hst[0]) / 100.0, (hst[7] - hst[0]) / 100.0);
#endif
    if (std::getenv("R3D_ADJ")) {
        // ---- reverse sweep: dA = the reference trajectory, dL/dtraj = a fixed multiple of it with every third frame left out ----
        const int TA = std::min(T, std::getenv("R3D_TADJ") ? std::atoi(std::getenv("R3D_TADJ")) : T);
        float *dG, *dG0, *dG1, *dR0;
        double *dSums, *dPart;
        CK(hipMalloc(&dG, (size_t)(TA + 1) * frame * sizeof(float)));
        CK(hipMalloc(&dG0, frame * sizeof(float))); CK(hipMalloc(&dG1, frame * sizeof(float))); CK(hipMalloc(&dR0, frame * sizeof(float)));
        CK(hipMalloc(&dSums, 22 * sizeof(double))); CK(hipMalloc(&dPart, (size_t)nwg * 36 * sizeof(double)));
        {
            std::vector<float> g(frame);
            Rng r2; r2.s = 12345;
            for (int t = 0; t <= TA; ++t) {
                for (auto& x : g) x = 1e-3f * (r2.uni() - 0.5f);
                CK(hipMemcpy(dG + (size_t)t * frame, g.data(), frame * sizeof(float), hipMemcpyHostToDevice));
            }
        }
        AdjArgs aa{};
        aa.traj = dA; aa.gtraj = dG; aa.gtop = dG + (size_t)TA * frame; aa.gout = dR0; aa.partials = dPart; aa.np = 36; aa.t_top = TA;
        for (int f = 0; f < TA; ++f) if (f % 3 != 1) aa.frames[f >> 5] |= 1u << (f & 31);
        // reference: launch per step, ping-pong
        CK(hipMemset(dSums, 0, 22 * sizeof(double)));
        {
            const float* gin = dG + (size_t)TA * frame;
            float* pp[2] = {dG0, dG1};
            for (int t = TA; t >= 1; --t) {
                const int f = t - 1;
                const bool h = (aa.frames[f >> 5] >> (f & 31)) & 1u;
                float* dst = pp[t & 1];
                hipLaunchKernelGGL(ref_adj, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, dA + (size_t)f * frame, gin, h ? dG + (size_t)f * frame : nullptr, dst, dSums, dP, n0, n1, n2);
                gin = dst;
            }
            CK(hipStreamSynchronize(st));
            Args b2 = a;
            b2.nsteps = TA;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pi_adj3d_resident_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            std::vector<float> msv;
            for (int i = 0; i < reps + 1; ++i) {
                CK(hipMemsetAsync(outbox, 0, obytes, st));
                CK(hipMemsetAsync(sync, 0, 64, st));
                CK(hipMemsetAsync(dPart, 0, (size_t)nwg * 36 * sizeof(double), st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(pi_adj3d_resident_kernel<NT>, dim3(nwg), dim3(NT), lds, st, dP, b2, aa, stamps);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float m;
                CK(hipEventElapsedTime(&m, e0, e1));
                if (i) msv.push_back(m);
            }
            CK(hipMemcpy(hs, sync, sizeof hs, hipMemcpyDeviceToHost));
            std::printf("adjoint sync: started %u abort %u timeouts %u\n", hs[0], hs[1], hs[2]);
            std::vector<float> A(frame), B(frame);
            CK(hipMemcpy(A.data(), gin, frame * sizeof(float), hipMemcpyDeviceToHost));
            CK(hipMemcpy(B.data(), dR0, frame * sizeof(float), hipMemcpyDeviceToHost));
            size_t bad = 0;
            double nrm = 0;
            for (size_t i = 0; i < frame; ++i) { bad += std::memcmp(&A[i], &B[i], 4) != 0; nrm += (double)A[i] * A[i]; }
            std::printf("adjoint dL/dh0 (T = %d): %zu differing values, |ref|^2 = %.6g -> %s\n", TA, bad, nrm, bad ? "DIFFERENT" : "identical");
            std::vector<double> hsum(22), part((size_t)nwg * 36);
            CK(hipMemcpy(hsum.data(), dSums, 22 * sizeof(double), hipMemcpyDeviceToHost));
            CK(hipMemcpy(part.data(), dPart, part.size() * sizeof(double), hipMemcpyDeviceToHost));
            double worst = 0;
            for (int k = 0; k < 22; ++k) {
                const int slot = k < 2 ? 1 + k : 16 + k - 2;
                double tot = 0;
                for (int w = 0; w < nwg; ++w) tot += part[(size_t)w * 36 + slot];
                const double rel = std::fabs(tot - hsum[k]) / (std::fabs(hsum[k]) + 1e-30);
                worst = std::max(worst, rel);
                if (k < 4 || rel > 1e-4) std::printf("  sum %2d: ref %.9e got %.9e rel %.2e\n", k, hsum[k], tot, rel);
            }
            std::printf("adjoint sums: worst relative difference %.2e\n", worst);
            std::sort(msv.begin(), msv.end());
            std::printf("resident adjoint: median %.3f ms = %.3f us per step (min %.3f)\n", msv[msv.size() / 2], 1e3 * msv[msv.size() / 2] / TA, 1e3 * msv[0] / TA);
#ifdef PI_R3D_STAMPS
            unsigned long long hst2[16];
            CK(hipMemcpy(hst2, stamps, sizeof hst2, hipMemcpyDeviceToHost));
            std::printf("adjoint timeline of step 20: S strips %.2f | barrier %.2f | published %.2f | I strips %.2f | barrier %.2f | written back %.2f | ring landed + unpacked %.2f\n",
                        (hst2[1] - hst2[0]) / 100.0, (hst2[2] - hst2[0]) / 100.0, (hst2[3] - hst2[0]) / 100.0, (hst2[4] - hst2[0]) / 100.0, (hst2[5] - hst2[0]) / 100.0, (hst2[6] - hst2[0]) / 100.0, (hst2[7] - hst2[0]) / 100.0);
#endif
        }
    }
    return 0;
}
