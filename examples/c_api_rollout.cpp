// The C-ABI of libpercnn_pi.so driven from a plain C++ host -- no Python, no PyTorch: what a non-Python caller of the
// drop-in boundary links against (include/percnn_pi.h).  Builds a pre-contracted parameter block ("hc = 0": Laplacian
// taps + 2 x 10 cubic coefficients, here the true 2D Gray-Scott equation), rolls a 64 x 48 state out for T steps with
// percnn_pi_rollout_fwd_f32, checks the last frame against a scalar host loop in the same operation order (bit for bit),
// then runs percnn_pi_rollout_bwd_f32 for L = sum(traj) and checks dL/dh0 against finite differences of that loop.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -Iinclude examples/c_api_rollout.cpp \
//         -Lpercnn_amd/csrc -lpercnn_pi -Wl,-rpath,$PWD/percnn_amd/csrc -o c_api_rollout && ./c_api_rollout
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "percnn_pi.h"

#define CHECK(x) do { int rc_ = (int)(x); if (rc_) { std::fprintf(stderr, "%s -> %d (line %d)\n", #x, rc_, __LINE__); return 1; } } while (0)

static const int H = 64, W = 48, T = 6;
static const long N = (long)H * W;

static inline int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// one step on the host, same order as the kernels: centre tap, axis-0 taps (-2,-1,+1,+2), axis-1 taps; Horner cubic
static void host_step(const std::vector<float>& h, std::vector<float>& o, const float* P)
{
    static const int offs[4] = {-2, -1, 1, 2};
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float u = h[y * W + x], v = h[N + y * W + x];
            for (int s = 0; s < 2; ++s) {
                const float* f = h.data() + s * N;
                float lap = P[3] * f[y * W + x];
                for (int i = 0; i < 4; ++i) lap = std::fmaf(P[4 + i], f[wrapi(y + offs[i], H) * W + x], lap);
                for (int i = 0; i < 4; ++i) lap = std::fmaf(P[8 + i], f[y * W + wrapi(x + offs[i], W)], lap);
                const float* c = P + 16 + 10 * s;          // 1, u, v, u2, uv, v2, u3, u2v, uv2, v3
                const float A0 = std::fmaf(v, std::fmaf(v, std::fmaf(v, c[9], c[5]), c[2]), c[0]);
                const float A1 = std::fmaf(v, std::fmaf(v, c[8], c[4]), c[1]);
                const float A2 = std::fmaf(v, c[7], c[3]);
                const float rr = std::fmaf(u, std::fmaf(u, std::fmaf(u, c[6], A2), A1), A0);
                const float res = P[1 + s] * lap + rr;
                const float inc = res * P[0];
                o[s * N + y * W + x] = f[y * W + x] + inc;
            }
        }
}

static double host_loss(std::vector<float> h0, const float* P)
{
    std::vector<float> a = h0, b(2 * N);
    double L = 0;
    for (float x : a) L += x;
    for (int t = 0; t < T; ++t) { host_step(a, b, P); a.swap(b); for (float x : a) L += x; }
    return L;
}

int main()
{
    if (percnn_pi_abi_version() != PERCNN_PI_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    // parameter block: dt, Du, Dv, 4th-order Laplacian / dx^2, Gray-Scott reaction  -u v^2 + f (1-u),  u v^2 - (f+k) v
    const float dx = 0.02f, f = 0.04f, k = 0.06f;
    float P[36] = {0};
    P[0] = 0.5f; P[1] = 2e-5f; P[2] = 5e-6f;
    P[3] = -5.0f / (dx * dx);
    const float taps[4] = {-1.0f / 12, 4.0f / 3, 4.0f / 3, -1.0f / 12};
    for (int a = 0; a < 3; ++a) for (int i = 0; i < 4; ++i) P[4 + 4 * a + i] = taps[i] / (dx * dx);
    P[16 + 0] = f; P[16 + 1] = -f; P[16 + 8] = -1.0f;        // species u: f - f u - u v^2
    P[26 + 2] = -(f + k); P[26 + 8] = 1.0f;                  // species v: -(f+k) v + u v^2
    if (percnn_pi_param_count(0) != 36) return 1;

    std::vector<float> h0(2 * N);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            h0[y * W + x] = 0.6f + 0.3f * std::sin(0.3f * x) * std::cos(0.2f * y);
            h0[N + y * W + x] = 0.25f + 0.2f * std::cos(0.25f * x + 0.1f * y);
        }
    const size_t frame = 2 * N;
    float *traj, *gtraj, *g0, *dP;
    double* pg;
    void* ws;
    const int64_t shape[2] = {H, W};
    const size_t wsb = percnn_pi_rollout_bwd_workspace_bytes(0, 2, shape, T, 4);
    CHECK(hipMalloc(&traj, (T + 1) * frame * 4)); CHECK(hipMalloc(&gtraj, (T + 1) * frame * 4));
    CHECK(hipMalloc(&g0, frame * 4)); CHECK(hipMalloc(&dP, sizeof(P))); CHECK(hipMalloc(&pg, 36 * 8)); CHECK(hipMalloc(&ws, wsb));
    CHECK(hipMemcpy(traj, h0.data(), frame * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dP, P, sizeof(P), hipMemcpyHostToDevice));
    std::vector<float> ones((T + 1) * frame, 1.0f);
    CHECK(hipMemcpy(gtraj, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));

    CHECK(percnn_pi_rollout_fwd_f32(traj, dP, 0, 2, shape, T, nullptr));
    CHECK(percnn_pi_rollout_bwd_f32(traj, gtraj, nullptr, g0, pg, ws, wsb, dP, 0, 2, shape, T, nullptr));
    CHECK(hipDeviceSynchronize());

    std::vector<float> last(frame), grad(frame);
    CHECK(hipMemcpy(last.data(), traj + T * frame, frame * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(grad.data(), g0, frame * 4, hipMemcpyDeviceToHost));
    std::vector<float> a = h0, b(frame);
    for (int t = 0; t < T; ++t) { host_step(a, b, P); a.swap(b); }
    for (size_t i = 0; i < frame; ++i)
        if (a[i] != last[i]) { std::fprintf(stderr, "forward differs at %zu: %g vs %g\n", i, a[i], last[i]); return 1; }
    // dL/dh0 at a few points by central differences of the host loop (float32 loop: loose tolerance)
    for (long idx : {5L * W + 7, 40L * W + 30, N + 17L * W + 3}) {
        std::vector<float> hp = h0, hm = h0;
        const float eps = 1e-2f;
        hp[idx] += eps; hm[idx] -= eps;
        const double fd = (host_loss(hp, P) - host_loss(hm, P)) / (2.0 * eps);
        if (std::fabs(fd - grad[idx]) > 2e-2 * std::fabs(fd) + 1e-3) { std::fprintf(stderr, "dL/dh0[%ld] %g vs fd %g\n", idx, grad[idx], fd); return 1; }
    }
    std::printf("c_api_rollout ok: %d steps on %dx%d, forward bit-identical to the host loop, dL/dh0 matches finite differences\n", T, H, W);
    return 0;
}
