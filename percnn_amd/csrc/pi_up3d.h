// 'same' 5x5x5 cross-correlation 8 -> 8 channels, zero padding: the contraction inside the 3D IC generator's second
// layer (ConvTranspose3d(8, 8, 5, padding=2), train_3drd.py:41-56) -- forward AND input gradient are this operation
// with differently arranged weights.  SURVEY 8f rank 4: MIOpen needs 107 ms per pass at 128^3 on MI355X, the
// im2col + rocBLAS formulation 7 ms; K = 8*125 = 1000 with N = 8 is too narrow for a library GEMM tile and too wide
// for the matrix cores to pay (M = 8 of 16 rows), so this is a register-blocked VALU kernel:
//   * a workgroup owns one z-plane tile of TY x TX outputs; a lane owns 4 consecutive x of one row and all 8 output
//     channels (32 accumulators = 16 v_pk_fma_f32 destinations);
//   * the input window (5 planes x (TY+4) rows x (TX+4) columns) of TWO input channels at a time is staged in LDS with
//     the zero padding resolved at staging time;
//   * per (ci, dz, dy) a lane reads ONE 8-float row segment (two ds_read_b128) and applies 5 taps x 8 channels =
//     40 wave-uniform weights (scalar loads from a [ci][dz][dy][dx][co] table) as 80 packed FMAs.
#pragma once
#include <hip/hip_runtime.h>
#include "pi_device.h"

namespace pi {
namespace up3d {

constexpr int C = 8;                  // channels in and out
constexpr int TY = 16, TX = 64;       // outputs per workgroup: 16 rows x 64 columns of one z-plane
constexpr int NT = TY * TX / 4;       // 256 lanes, 4 x-outputs each
constexpr int WY = TY + 4, WX = TX + 8;   // window rows / padded row length (72 floats: 16-B aligned segments)
constexpr int CPC = 2;                // input channels staged per pass
constexpr int NW = C * 125 * C;       // weight table entries

struct Geom {
    int D, H, W;
    long plane;                       // H * W
    long cs;                          // channel stride D * H * W
    int tiles_x, tiles_y;
};

using f2 = float __attribute__((ext_vector_type(2)));

// out[co](p) = bias[co] + sum_{ci, d} Wt[ci][dz][dy][dx][co] * in[ci](p + d - 2)      (zero outside the grid)
// grid = tiles_x * tiles_y * D workgroups
__global__ __launch_bounds__(NT) void conv5_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                   const float* __restrict__ Wt, const float* __restrict__ bias, Geom g)
{
    __shared__ __attribute__((aligned(16))) float win[CPC][5][WY][WX];
    const int tid = threadIdx.x;
    const int ty = tid / (TX / 4), tx4 = (tid % (TX / 4)) * 4;
    int b = blockIdx.x;
    const int bx = b % g.tiles_x; b /= g.tiles_x;
    const int by = b % g.tiles_y;
    const int z = b / g.tiles_y;
    const int y0 = by * TY, x0 = bx * TX;

    f2 acc[C][2];
#pragma unroll
    for (int co = 0; co < C; ++co) {
        const float bv = bias ? bias[co] : 0.f;
        acc[co][0] = f2{bv, bv};
        acc[co][1] = f2{bv, bv};
    }

    for (int c0 = 0; c0 < C; c0 += CPC) {
        __syncthreads();
        // stage CPC channels x 5 planes x WY rows x (TX + 4) columns; column j of the window <-> x0 - 2 + j
        constexpr int ROW = TX + 4;
        for (int i = tid; i < CPC * 5 * WY * ROW; i += NT) {
            const int j = i % ROW;
            int r = i / ROW;
            const int wy = r % WY; r /= WY;
            const int dz = r % 5;
            const int cc = r / 5;
            const int zz = z + dz - 2, yy = y0 + wy - 2, xx = x0 + j - 2;
            float v = 0.f;
            if ((unsigned)zz < (unsigned)g.D && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W)
                v = in[(long)(c0 + cc) * g.cs + (long)zz * g.plane + (long)yy * g.W + xx];
            win[cc][dz][wy][j] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int cc = 0; cc < CPC; ++cc)
#pragma unroll 1
            for (int dz = 0; dz < 5; ++dz)
#pragma unroll 1
                for (int dy = 0; dy < 5; ++dy) {
                    const float* row = &win[cc][dz][ty + dy][tx4];          // 16-B aligned: tx4 % 4 == 0, WX % 4 == 0
                    const float4 s0 = *reinterpret_cast<const float4*>(row);
                    const float4 s1 = *reinterpret_cast<const float4*>(row + 4);
                    const float seg[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    const float* w = Wt + ((((c0 + cc) * 5 + dz) * 5 + dy) * 5) * C;      // [dx][co], wave-uniform
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) {
                        const f2 a = f2{seg[dx], seg[dx + 1]}, c = f2{seg[dx + 2], seg[dx + 3]};
#pragma unroll
                        for (int co = 0; co < C; ++co) {
                            const float wv = w[dx * C + co];
                            const f2 ww = f2{wv, wv};
                            acc[co][0] = __builtin_elementwise_fma(ww, a, acc[co][0]);
                            acc[co][1] = __builtin_elementwise_fma(ww, c, acc[co][1]);
                        }
                    }
                }
    }
    const int y = y0 + ty, x = x0 + tx4;
    if (y < g.H && x < g.W) {
#pragma unroll
        for (int co = 0; co < C; ++co) {
            float* o = out + (long)co * g.cs + (long)z * g.plane + (long)y * g.W + x;
            const float v[4] = {acc[co][0].x, acc[co][0].y, acc[co][1].x, acc[co][1].y};
            if (x + 3 < g.W && ((g.W & 3) == 0)) {
                *reinterpret_cast<float4*>(o) = float4{v[0], v[1], v[2], v[3]};
            } else {
                for (int i = 0; i < 4 && x + i < g.W; ++i) o[i] = v[i];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// weight gradient of the same contraction:  dWt[ci][dz][dy][dx][co] = sum_p in[ci](p + d - 2) * g[co](p)
// Lanes own weights, not points: lane (ci, dz, dy) keeps its 5 taps x 8 channels = 40 sums in registers while the
// workgroup walks over its tiles (persistent: gridDim.x workgroups share all tiles); per 4 points a lane reads one
// 8-float input segment (distinct per lane) and the 4 x 8 output gradients (the same for every lane: LDS broadcast)
// and issues 80 packed FMAs.  One float row of 8000 sums per workgroup; wgrad_reduce_kernel adds the rows in a fixed
// order.
// ------------------------------------------------------------------------------------------------
constexpr int GY = 8, GX = 32;                 // points per tile (one z-plane)
constexpr int GWY = GY + 4, GWX = GX + 8;      // window rows / padded row length
constexpr int NOWN = C * 25;                   // lanes that own weights (the rest only help staging)

struct WGeom {
    int D, H, W;
    long plane, cs;
    int tiles_x, tiles_y;
    long ntiles;
};

__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ in, const float* __restrict__ g,
                                                    float* __restrict__ partials, WGeom q)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*win)[5][GWY][GWX] = reinterpret_cast<float (*)[5][GWY][GWX]>(smem);              // [C]
    float (*gt)[GX][C] = reinterpret_cast<float (*)[GX][C]>(smem + C * 5 * GWY * GWX);       // [GY][GX][co]
    const int tid = threadIdx.x;
    const bool own = tid < NOWN;
    const int ci = own ? tid / 25 : 0, dz = own ? (tid % 25) / 5 : 0, dy = own ? tid % 5 : 0;

    f2 acc[5][C / 2];
#pragma unroll
    for (int dx = 0; dx < 5; ++dx)
#pragma unroll
        for (int cp = 0; cp < C / 2; ++cp) acc[dx][cp] = f2{0.f, 0.f};

    for (long tile = blockIdx.x; tile < q.ntiles; tile += gridDim.x) {
        long b = tile;
        const int bx = (int)(b % q.tiles_x); b /= q.tiles_x;
        const int by = (int)(b % q.tiles_y);
        const int z = (int)(b / q.tiles_y);
        const int y0 = by * GY, x0 = bx * GX;
        __syncthreads();
        // input window: C x 5 x GWY x (GX + 4) values, column j <-> x0 - 2 + j, staged as 8-byte pairs (x0 - 2 is even)
        constexpr int PAIRS = (GX + 4) / 2;
        for (int i = tid; i < C * 5 * GWY * PAIRS; i += 256) {
            const int jp = i % PAIRS;
            int r = i / PAIRS;
            const int wy = r % GWY; r /= GWY;
            const int pz = r % 5;
            const int cc = r / 5;
            const int zz = z + pz - 2, yy = y0 + wy - 2, xx = x0 + 2 * jp - 2;
            float v0 = 0.f, v1 = 0.f;
            if ((unsigned)zz < (unsigned)q.D && (unsigned)yy < (unsigned)q.H) {
                const float* src = in + (long)cc * q.cs + (long)zz * q.plane + (long)yy * q.W;
                if ((unsigned)xx < (unsigned)q.W) v0 = src[xx];
                if ((unsigned)(xx + 1) < (unsigned)q.W) v1 = src[xx + 1];
            }
            win[cc][pz][wy][2 * jp] = v0;
            win[cc][pz][wy][2 * jp + 1] = v1;
        }
        for (int i = tid; i < GY * GX * C; i += 256) {                   // gt[py][px][co]
            const int co = i % C;
            const int px = (i / C) % GX, py = i / (C * GX);
            const int yy = y0 + py, xx = x0 + px;
            gt[py][px][co] = (yy < q.H && xx < q.W) ? g[(long)co * q.cs + (long)z * q.plane + (long)yy * q.W + xx] : 0.f;
        }
        __syncthreads();
        if (own) {
#pragma unroll 1
            for (int py = 0; py < GY; ++py)
#pragma unroll 1
                for (int px = 0; px < GX; px += 4) {
                    const float* row = &win[ci][dz][py + dy][px];
                    const float4 s0 = *reinterpret_cast<const float4*>(row);
                    const float4 s1 = *reinterpret_cast<const float4*>(row + 4);
                    const float seg[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 ga = *reinterpret_cast<const float4*>(&gt[py][px + i][0]);
                        const float4 gb = *reinterpret_cast<const float4*>(&gt[py][px + i][4]);
                        const f2 gp[4] = {f2{ga.x, ga.y}, f2{ga.z, ga.w}, f2{gb.x, gb.y}, f2{gb.z, gb.w}};
#pragma unroll
                        for (int dx = 0; dx < 5; ++dx) {
                            const f2 sv = f2{seg[dx + i], seg[dx + i]};
#pragma unroll
                            for (int cp = 0; cp < C / 2; ++cp) acc[dx][cp] = __builtin_elementwise_fma(gp[cp], sv, acc[dx][cp]);
                        }
                    }
                }
        }
    }
    if (own) {
        float* row = partials + (long)blockIdx.x * NW + (long)tid * 40;      // [ci][dz][dy] = tid, then [dx][co]
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
#pragma unroll
            for (int cp = 0; cp < C / 2; ++cp) {
                row[dx * C + 2 * cp] = acc[dx][cp].x;
                row[dx * C + 2 * cp + 1] = acc[dx][cp].y;
            }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partials, int nrows, float* __restrict__ gw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NW) return;
    double s = 0.0;
    for (int r = 0; r < nrows; ++r) s += (double)partials[(long)r * NW + i];
    gw[i] = (float)s;
}

}  // namespace up3d
}  // namespace pi
