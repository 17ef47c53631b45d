/* Plain-C oracle for the Pi-block step (TEST INFRASTRUCTURE ONLY -- see pi_oracle_impl.h).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: every rounding is the one written down). */
#include <math.h>

#define REAL float
#define SUF f32
#define FMA(a, b, c) fmaf((a), (b), (c))
#include "pi_oracle_impl.h"
#undef REAL
#undef SUF
#undef FMA

#define REAL double
#define SUF f64
#define FMA(a, b, c) fma((a), (b), (c))
#include "pi_oracle_impl.h"
#undef REAL
#undef SUF
#undef FMA

/* ------------------------------------------------------------------------------------------------
 * Stage-1 Pi-block forward (float32 only, as the reference's Stage-1 scripts):
 *   DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-1/rcnn_Burgers_[...].py:160-176
 *   DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-1/rcnn_LO_[...].py:161-170
 * Block layout (include/percnn_pi_stage1.h, restated independently):
 *   P[0..15]  header as above (dt, coef_u, coef_v, centre tap, taps of axis 0 / axis 1)
 *   P[16 + ((s*3+k)*16 + j)*52 + kk]  branch k of species s, hidden channel j;
 *             kk = c*25 + dy*5 + dx  (cross-correlation tap (dy-2, dx-2) on input channel c), kk=50 bias, kk=51 = 0
 *   P[16+4992 + s*16 + j] = Wh4 weight,  P[16+4992+32 + s] = Wh4 bias
 * Summation order = an fmaf chain over kk ascending from 0 -- what a chain of v_mfma_f32_16x16x4_f32 computes
 * (MI355X guide: "bit-for-bit a k-ordered f32 fmaf chain"); hidden channels are contracted as four
 * chains of four (j = 4g..4g+3) summed (c0+c1)+(c2+c3).  The torch restatement agrees to float32 round-off.
 * ------------------------------------------------------------------------------------------------ */
#define S1_HC 16
#define S1_KK 52
#define S1_OFF_W 16
#define S1_OFF_W4 (16 + 6 * 16 * 52)
#define S1_OFF_B4 (S1_OFF_W4 + 32)

static inline long s1_wrap(long i, long n) { i %= n; return i < 0 ? i + n : i; }

void pi_oracle_s1_step_fwd_f32(const float *h, float *out, const float *P, long H, long W)
{
    static const int offs[4] = {-2, -1, 1, 2};
    const long n = H * W;
    const float dt = P[0];
    for (long y = 0; y < H; ++y)
        for (long x = 0; x < W; ++x) {
            float col[S1_KK];
            for (int c = 0; c < 2; ++c)
                for (int dy = 0; dy < 5; ++dy)
                    for (int dx = 0; dx < 5; ++dx)
                        col[c * 25 + dy * 5 + dx] = h[c * n + s1_wrap(y + dy - 2, H) * W + s1_wrap(x + dx - 2, W)];
            col[50] = 1.0f;
            col[51] = 0.0f;
            for (int s = 0; s < 2; ++s) {
                float part[4];
                for (int g = 0; g < 4; ++g) {
                    float t = 0.0f;
                    for (int r = 0; r < 4; ++r) {
                        const int j = 4 * g + r;
                        float br[3];
                        for (int k = 0; k < 3; ++k) {
                            const float *w = P + S1_OFF_W + ((s * 3 + k) * 16 + j) * S1_KK;
                            float acc = 0.0f;
                            for (int kk = 0; kk < S1_KK; ++kk) acc = fmaf(w[kk], col[kk], acc);
                            br[k] = acc;
                        }
                        t = fmaf(P[S1_OFF_W4 + s * 16 + j], (br[0] * br[1]) * br[2], t);
                    }
                    part[g] = t;
                }
                const float rr = ((part[0] + part[1]) + (part[2] + part[3])) + P[S1_OFF_B4 + s];
                const float *f = h + s * n;
                float lap = P[3] * f[y * W + x];
                for (int i = 0; i < 4; ++i) lap = fmaf(P[4 + i], f[s1_wrap(y + offs[i], H) * W + x], lap);
                for (int i = 0; i < 4; ++i) lap = fmaf(P[8 + i], f[y * W + s1_wrap(x + offs[i], W)], lap);
                const float res = P[1 + s] * lap + rr;
                const float t = res * dt;
                out[s * n + y * W + x] = f[y * W + x] + t;
            }
        }
}

void pi_oracle_s1_rollout_fwd_f32(float *traj, const float *P, long H, long W, int T)
{
    const long frame = 2 * H * W;
    for (int t = 0; t < T; ++t) pi_oracle_s1_step_fwd_f32(traj + t * frame, traj + (t + 1) * frame, P, H, W);
}
