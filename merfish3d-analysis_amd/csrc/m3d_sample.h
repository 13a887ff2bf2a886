// SciPy-exact order-1 sampling shared by the resampling and flow-estimation kernels:
// coordinates and weights in double, product order value*wz*wy*wx summed z-major,
// mode "constant" = cval for any coordinate outside [0, n-1], mode "nearest" = clamp.
#pragma once

#include <cmath>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace m3d {

enum { MODE_CONSTANT = 0, MODE_NEAREST = 1 };

template <typename T>
__device__ __forceinline__ double sample_linear3(const T *__restrict__ img, int Z, int Y, int X,
                                                 double cz, double cy, double cx, int mode, double cval)
{
    if (mode == MODE_CONSTANT) {
        if (cz < 0.0 || cz > (double)(Z - 1) || cy < 0.0 || cy > (double)(Y - 1) || cx < 0.0 ||
            cx > (double)(X - 1))
            return cval;
        if (!(cz == cz) || !(cy == cy) || !(cx == cx)) return cval;
    } else {
        cz = cz < 0.0 ? 0.0 : (cz > (double)(Z - 1) ? (double)(Z - 1) : cz);
        cy = cy < 0.0 ? 0.0 : (cy > (double)(Y - 1) ? (double)(Y - 1) : cy);
        cx = cx < 0.0 ? 0.0 : (cx > (double)(X - 1) ? (double)(X - 1) : cx);
        if (!(cz == cz) || !(cy == cy) || !(cx == cx)) return cval;
    }
    const double fz = floor(cz), fy = floor(cy), fx = floor(cx);
    const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
    const double tz = cz - fz, ty = cy - fy, tx = cx - fx;
    const double wz[2] = {1.0 - tz, tz}, wy[2] = {1.0 - ty, ty}, wx[2] = {1.0 - tx, tx};
    const int zi[2] = {z0, z0 + 1 < Z ? z0 + 1 : Z - 1};
    const int yi[2] = {y0, y0 + 1 < Y ? y0 + 1 : Y - 1};
    const int xi[2] = {x0, x0 + 1 < X ? x0 + 1 : X - 1};
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double coeff = (double)img[((int64_t)zi[a] * Y + yi[b]) * X + xi[c]];
                coeff *= wz[a];
                coeff *= wy[b];
                coeff *= wx[c];
                t += coeff;
            }
    return t;
}

}  // namespace m3d
