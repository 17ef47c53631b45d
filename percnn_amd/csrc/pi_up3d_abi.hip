// C-ABI of the 3D IC generator's 5^3 contraction (include/percnn_pi_stage1.h is unrelated; declared in percnn_pi.h).
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/percnn_pi.h"
#include "pi_up3d.h"

extern "C" {

int percnn_pi_conv3d_k5c8_f32(const float* in, float* out, const float* weights, const float* bias, const int64_t* shape,
                              void* stream)
{
    if (!in || !out || !weights || !shape || in == out) return PERCNN_PI_EINVAL;
    for (int a = 0; a < 3; ++a)
        if (shape[a] < 1 || shape[a] > (1 << 12)) return PERCNN_PI_EINVAL;
    pi::up3d::Geom g;
    g.D = (int)shape[0]; g.H = (int)shape[1]; g.W = (int)shape[2];
    g.plane = (long)g.H * g.W;
    g.cs = (long)g.D * g.plane;
    g.tiles_x = (g.W + pi::up3d::TX - 1) / pi::up3d::TX;
    g.tiles_y = (g.H + pi::up3d::TY - 1) / pi::up3d::TY;
    const long blocks = (long)g.tiles_x * g.tiles_y * g.D;
    if (blocks > 0x7fffffffL) return PERCNN_PI_EINVAL;
    hipLaunchKernelGGL(pi::up3d::conv5_kernel, dim3((unsigned)blocks), dim3(pi::up3d::NT), 0, static_cast<hipStream_t>(stream),
                       in, out, weights, bias, g);
    return (int)hipGetLastError();
}

constexpr int WGRAD_ROWS = 512;

size_t percnn_pi_conv3d_k5c8_wgrad_workspace_bytes(void) { return (size_t)WGRAD_ROWS * pi::up3d::NW * sizeof(float); }

int percnn_pi_conv3d_k5c8_wgrad_f32(const float* in, const float* g_out, float* g_weights, void* workspace,
                                    size_t workspace_bytes, const int64_t* shape, void* stream)
{
    if (!in || !g_out || !g_weights || !shape) return PERCNN_PI_EINVAL;
    for (int a = 0; a < 3; ++a)
        if (shape[a] < 1 || shape[a] > (1 << 12)) return PERCNN_PI_EINVAL;
    if (!workspace || workspace_bytes < percnn_pi_conv3d_k5c8_wgrad_workspace_bytes()) return PERCNN_PI_EWORKSPACE;
    namespace U = pi::up3d;
    U::WGeom q;
    q.D = (int)shape[0]; q.H = (int)shape[1]; q.W = (int)shape[2];
    q.plane = (long)q.H * q.W;
    q.cs = (long)q.D * q.plane;
    q.tiles_x = (q.W + U::GX - 1) / U::GX;
    q.tiles_y = (q.H + U::GY - 1) / U::GY;
    q.ntiles = (long)q.tiles_x * q.tiles_y * q.D;
    const unsigned rows = (unsigned)(q.ntiles < WGRAD_ROWS ? q.ntiles : WGRAD_ROWS);
    const size_t lds = (size_t)(U::C * 5 * U::GWY * U::GWX + U::GY * U::GX * U::C) * sizeof(float);
    auto st = static_cast<hipStream_t>(stream);
    static bool lds_ok = false;
    if (!lds_ok) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(U::wgrad_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return (int)e;
        lds_ok = true;
    }
    hipLaunchKernelGGL(U::wgrad_kernel, dim3(rows), dim3(256), lds, st, in, g_out, static_cast<float*>(workspace), q);
    if (hipError_t e = hipGetLastError()) return (int)e;
    hipLaunchKernelGGL(U::wgrad_reduce_kernel, dim3((U::NW + 255) / 256), dim3(256), 0, st,
                       static_cast<const float*>(workspace), (int)rows, g_weights);
    return (int)hipGetLastError();
}

}  // extern "C"
