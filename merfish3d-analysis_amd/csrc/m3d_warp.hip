// Resampling kernels of the registration / decode-warp path (utils/multiview_registration.py).
//
//   m3d_max_z        _max_z_projection_gpu                                        (:57-77)
//   m3d_affine_warp  warp_array_to_reference_gpu = ndimage.affine_transform,
//                    order 1, output dtype = input dtype                         (:797-902)
//   m3d_flow_warp    warp_array_to_reference_with_affine_and_sofima_flow_gpu:
//                    3 flow interpolations + affine compose + ONE image sample   (:905-1131)
//
// The reference evaluates these with cupyx.scipy.ndimage whose documented semantics are
// SciPy's: coordinates and spline weights in double, product order value*wz*wy*wx summed
// z-major, mode "constant" = cval for any coordinate outside [0, n-1] (no edge blending),
// mode "nearest" = clamp, unsigned-integer outputs rounded as (t + 0.5) truncated and
// clipped.  The kernels mirror that order so float32 results match the SciPy oracle to the
// last bit.  The flow warp fuses what the reference does in Z/4 batches of 5
// map_coordinates calls with 12 temporaries each into one pass: 4 B read + 4 B written per
// voxel (flow grid and gathers are cache resident).

#include <cmath>

#include "m3d_common.h"
#include "m3d_index.h"
#include "m3d_sample.h"

using namespace m3d;

namespace {

template <typename T>
__device__ __forceinline__ T cast_out(double t);
template <>
__device__ __forceinline__ float cast_out<float>(double t)
{
    return (float)t;
}
template <>
__device__ __forceinline__ uint16_t cast_out<uint16_t>(double t)
{
    t = t > 0.0 ? t + 0.5 : 0.0;  // scipy CASE_INTERP_OUT_UINT
    t = t > 65535.0 ? 65535.0 : t;
    return (uint16_t)t;
}

struct Affine {
    double m[9];
    double off[3];
};

template <typename T>
__global__ void k_affine_warp(const T *__restrict__ in, int Zi, int Yi, int Xi, T *__restrict__ out,
                              int Zo, int Yo, int Xo, Affine A, int mode, double cval)
{
    const int64_t rows = (int64_t)Zo * Yo;
    for (int64_t r = (int64_t)blockIdx.x * 4 + threadIdx.y; r < rows; r += (int64_t)gridDim.x * 4) {
        const int z = (int)(r / Yo), y = (int)(r % Yo);
        for (int x = threadIdx.x; x < Xo; x += 64) {
            // scipy: cc = offset + sum_j matrix[i][j] * o_j, accumulated j = 0..2
            double c[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double v = A.off[i];
                v += A.m[3 * i + 0] * (double)z;
                v += A.m[3 * i + 1] * (double)y;
                v += A.m[3 * i + 2] * (double)x;
                c[i] = v;
            }
            out[r * Xo + x] = cast_out<T>(sample_linear3(in, Zi, Yi, Xi, c[0], c[1], c[2], mode, cval));
        }
    }
}

struct FlowArgs {
    const float *flow;  // [3][fz][fy][fx], channels X,Y,Z
    int fz, fy, fx;
    float stride[3];     // zyx
    float box_start[3];  // zyx
    float spacing[3];
    float origin[3];
    float T[12];  // rows 0..2 of the 4x4 physical transform
};

template <typename T>
__global__ void k_flow_warp(const T *__restrict__ in, int Zi, int Yi, int Xi, T *__restrict__ out,
                            int Zo, int Yo, int Xo, FlowArgs f, int mode, float cval)
{
    const int64_t rows = (int64_t)Zo * Yo;
    const int64_t fplane = (int64_t)f.fz * f.fy * f.fx;
    for (int64_t r = (int64_t)blockIdx.x * 4 + threadIdx.y; r < rows; r += (int64_t)gridDim.x * 4) {
        const int z = (int)(r / Yo), y = (int)(r % Yo);
        const float gz = (float)z, gy = (float)y;
        for (int x = threadIdx.x; x < Xo; x += 64) {
            const float gx = (float)x;
            // flow sampled at (idx - box_start) / stride, float32 like the reference (:1042-1049)
            const float qz = (gz - f.box_start[0]) / f.stride[0];
            const float qy = (gy - f.box_start[1]) / f.stride[1];
            const float qx = (gx - f.box_start[2]) / f.stride[2];
            const float ux = (float)sample_linear3(f.flow + 0 * fplane, f.fz, f.fy, f.fx, (double)qz,
                                                   (double)qy, (double)qx, mode, (double)cval);
            const float uy = (float)sample_linear3(f.flow + 1 * fplane, f.fz, f.fy, f.fx, (double)qz,
                                                   (double)qy, (double)qx, mode, (double)cval);
            const float uz = (float)sample_linear3(f.flow + 2 * fplane, f.fz, f.fy, f.fx, (double)qz,
                                                   (double)qy, (double)qx, mode, (double)cval);
            const float ax = gx + ux, ay = gy + uy, az = gz + uz;  // :1062
            const float pz = az * f.spacing[0] + f.origin[0];      // :1064-1066
            const float py = ay * f.spacing[1] + f.origin[1];
            const float px = ax * f.spacing[2] + f.origin[2];
            const float mz = f.T[0] * pz + f.T[1] * py + f.T[2] * px + f.T[3];  // :1068-1085
            const float my = f.T[4] * pz + f.T[5] * py + f.T[6] * px + f.T[7];
            const float mx = f.T[8] * pz + f.T[9] * py + f.T[10] * px + f.T[11];
            const float sz = (mz - f.origin[0]) / f.spacing[0];  // :1086-1093
            const float sy = (my - f.origin[1]) / f.spacing[1];
            const float sx = (mx - f.origin[2]) / f.spacing[2];
            out[r * Xo + x] = cast_out<T>(
                sample_linear3(in, Zi, Yi, Xi, (double)sz, (double)sy, (double)sx, mode, (double)cval));
        }
    }
}

template <typename T>
__global__ void k_max_z(const T *__restrict__ in, int Z, int64_t plane, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane;
         i += (int64_t)gridDim.x * blockDim.x) {
        float m = (float)in[i];
        for (int z = 1; z < Z; ++z) {
            const float v = (float)in[(int64_t)z * plane + i];
            m = (v > m || v != v) ? v : m;  // np.max propagates NaN
        }
        out[i] = m;
    }
}

inline int warp_grid(int64_t rows)
{
    int64_t g = (rows + 3) / 4;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return (int)g;
}

}  // namespace

extern "C" {

int m3d_max_z(const void *in, int dtype, const int32_t shape_zyx[3], float *out, void *stream)
{
    M3D_REQUIRE(in && shape_zyx && out, "null pointer");
    M3D_REQUIRE(dtype == M3D_U16 || dtype == M3D_F32, "dtype must be u16 or f32");
    const int Z = shape_zyx[0];
    const int64_t plane = (int64_t)shape_zyx[1] * shape_zyx[2];
    M3D_REQUIRE(Z > 0 && plane > 0, "shape must be positive");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == M3D_U16)
        hipLaunchKernelGGL(k_max_z<uint16_t>, dim3(grid_for(plane)), dim3(kBlock), 0, st,
                           (const uint16_t *)in, Z, plane, out);
    else
        hipLaunchKernelGGL(k_max_z<float>, dim3(grid_for(plane)), dim3(kBlock), 0, st,
                           (const float *)in, Z, plane, out);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_affine_warp(const void *in, int dtype, const int32_t in_shape_zyx[3],
                    const double matrix_px_host[9], const double offset_px_host[3], void *out,
                    const int32_t out_shape_zyx[3], int mode, double cval, void *stream)
{
    M3D_REQUIRE(in && out && in_shape_zyx && out_shape_zyx && matrix_px_host && offset_px_host,
                "null pointer");
    M3D_REQUIRE(dtype == M3D_U16 || dtype == M3D_F32, "dtype must be u16 or f32");
    M3D_REQUIRE(mode == MODE_CONSTANT || mode == MODE_NEAREST, "mode must be constant(0) or nearest(1)");
    for (int a = 0; a < 3; ++a)
        M3D_REQUIRE(in_shape_zyx[a] > 0 && out_shape_zyx[a] > 0, "shape must be positive");
    Affine A;
    for (int i = 0; i < 9; ++i) A.m[i] = matrix_px_host[i];
    for (int i = 0; i < 3; ++i) A.off[i] = offset_px_host[i];
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(64, 4);
    const int grid = warp_grid((int64_t)out_shape_zyx[0] * out_shape_zyx[1]);
    if (dtype == M3D_U16)
        hipLaunchKernelGGL(k_affine_warp<uint16_t>, dim3(grid), blk, 0, st, (const uint16_t *)in,
                           in_shape_zyx[0], in_shape_zyx[1], in_shape_zyx[2], (uint16_t *)out,
                           out_shape_zyx[0], out_shape_zyx[1], out_shape_zyx[2], A, mode, cval);
    else
        hipLaunchKernelGGL(k_affine_warp<float>, dim3(grid), blk, 0, st, (const float *)in,
                           in_shape_zyx[0], in_shape_zyx[1], in_shape_zyx[2], (float *)out,
                           out_shape_zyx[0], out_shape_zyx[1], out_shape_zyx[2], A, mode, cval);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_flow_warp(const void *in, int dtype, const int32_t in_shape_zyx[3], void *out,
                  const int32_t out_shape_zyx[3], const float *flow_xyz,
                  const int32_t flow_shape_zyx[3], const float stride_zyx_host[3],
                  const float box_start_zyx_host[3], const float spacing_zyx_host[3],
                  const float origin_zyx_host[3], const float transform_zyx_um_host[16], int mode,
                  float cval, void *stream)
{
    M3D_REQUIRE(in && out && flow_xyz && in_shape_zyx && out_shape_zyx && flow_shape_zyx, "null pointer");
    M3D_REQUIRE(stride_zyx_host && box_start_zyx_host && spacing_zyx_host && origin_zyx_host &&
                    transform_zyx_um_host,
                "null pointer");
    M3D_REQUIRE(dtype == M3D_U16 || dtype == M3D_F32, "dtype must be u16 or f32");
    M3D_REQUIRE(mode == MODE_CONSTANT || mode == MODE_NEAREST, "mode must be constant(0) or nearest(1)");
    for (int a = 0; a < 3; ++a)
        M3D_REQUIRE(in_shape_zyx[a] > 0 && out_shape_zyx[a] > 0 && flow_shape_zyx[a] > 0,
                    "shape must be positive");
    FlowArgs f;
    f.flow = flow_xyz;
    f.fz = flow_shape_zyx[0];
    f.fy = flow_shape_zyx[1];
    f.fx = flow_shape_zyx[2];
    for (int a = 0; a < 3; ++a) {
        f.stride[a] = stride_zyx_host[a];
        f.box_start[a] = box_start_zyx_host[a];
        f.spacing[a] = spacing_zyx_host[a];
        f.origin[a] = origin_zyx_host[a];
    }
    for (int i = 0; i < 12; ++i) f.T[i] = transform_zyx_um_host[i];
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(64, 4);
    const int grid = warp_grid((int64_t)out_shape_zyx[0] * out_shape_zyx[1]);
    if (dtype == M3D_U16)
        hipLaunchKernelGGL(k_flow_warp<uint16_t>, dim3(grid), blk, 0, st, (const uint16_t *)in,
                           in_shape_zyx[0], in_shape_zyx[1], in_shape_zyx[2], (uint16_t *)out,
                           out_shape_zyx[0], out_shape_zyx[1], out_shape_zyx[2], f, mode, cval);
    else
        hipLaunchKernelGGL(k_flow_warp<float>, dim3(grid), blk, 0, st, (const float *)in,
                           in_shape_zyx[0], in_shape_zyx[1], in_shape_zyx[2], (float *)out,
                           out_shape_zyx[0], out_shape_zyx[1], out_shape_zyx[2], f, mode, cval);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

}  // extern "C"
