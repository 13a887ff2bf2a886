// Residual (deformable) flow estimation between the round-1 fiducial volume and an
// affine-initialised moving round  (utils/sofima_registration.py:524-757).
//
// The reference delegates the arithmetic to the `sofima` package running on JAX
// (JAXMaskedXCorrWithStatsCalculator.flow_field, mesh.relax_mesh + elastic_mesh_3d) and to
// jax.scipy map_coordinates for its own sub-pixel search and flow composition.  Here:
//
//   m3d_patch_pack            gather one patch pair per flow-grid node, subtract the patch
//                             means, pack pre + i*post zero-padded to the FFT box
//   (rocFFT, batched c2c)     one forward transform per node (both patches at once)
//   m3d_patch_cross_spectrum  split the packed spectrum (Hermitian symmetry) and form
//                             PRE * conj(POST) / N in place
//   (rocFFT, batched c2c)     inverse -> full linear cross-correlation, real part
//   m3d_xcorr_peaks           integer peak (= flow vector post -> pre), peak sharpness,
//                             peak ratio, one workgroup per node
//   m3d_subpixel_refine       27-candidate fractional search by normalised SSD, patches held
//                             in LDS                                           (:243-417)
//   m3d_mesh_relax            elastic mesh, velocity-Verlet + FIRE, two launches per step
//   m3d_flow_compose          residual o base on the same grid                (:178-240)
//   m3d_moments, m3d_normalized_sq_diff   _normalized_mean_squared_error      (:149-175)
//
//   m3d_patch_xcorr_fused     the first five steps in one call with the transforms in LDS / registers, for the
//                             transform box 24 x 64 x 64 of the configured 10 x 32 x 32 patches (k_xc_*)
//
// Flow grids are tiny next to the images (11 x 127 x 127 nodes for a 64 x 2048 x 2048 tile);
// the cost is the 2 x 177k 24 x 64 x 64 transforms per pass: through rocFFT they stream through HBM about 25 times
// (0.71 s per pass), the fused route moves 3 MB per node (0.15 s).

#include <algorithm>
#include <cmath>
#include <cstring>

#include "m3d_common.h"
#include "m3d_fft.h"
#include "m3d_sample.h"

using namespace m3d;

namespace {

struct I3 {
    int z, y, x;
};

// ---------------------------------------------------------------------------------------------
// patch gather + mean subtraction + complex packing
// ---------------------------------------------------------------------------------------------
__global__ void k_patch_pack(const float *__restrict__ pre, const float *__restrict__ post, I3 shape,
                             I3 patch, I3 step, I3 grid, int64_t node_begin, I3 F,
                             float2 *__restrict__ out)
{
    __shared__ double sm[4];
    __shared__ float means[2];
    const int64_t node = node_begin + blockIdx.x;
    const int gx = (int)(node % grid.x), gy = (int)((node / grid.x) % grid.y),
              gz = (int)(node / ((int64_t)grid.x * grid.y));
    const int z0 = gz * step.z, y0 = gy * step.y, x0 = gx * step.x;
    const int pv = patch.z * patch.y * patch.x;
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < pv; i += blockDim.x) {
        const int ux = i % patch.x, uy = (i / patch.x) % patch.y, uz = i / (patch.x * patch.y);
        const int64_t src = ((int64_t)(z0 + uz) * shape.y + (y0 + uy)) * shape.x + (x0 + ux);
        sa += (double)pre[src];
        sb += (double)post[src];
    }
    const double ta = block_sum(sa, sm);
    const double tb = block_sum(sb, sm);
    if (threadIdx.x == 0) {
        means[0] = (float)(ta / (double)pv);
        means[1] = (float)(tb / (double)pv);
    }
    __syncthreads();
    const float ma = means[0], mb = means[1];
    const int fv = F.z * F.y * F.x;
    float2 *dst = out + (int64_t)blockIdx.x * fv;
    for (int i = threadIdx.x; i < fv; i += blockDim.x) {
        const int ux = i % F.x, uy = (i / F.x) % F.y, uz = i / (F.x * F.y);
        float2 v = make_float2(0.f, 0.f);
        if (ux < patch.x && uy < patch.y && uz < patch.z) {
            const int64_t src = ((int64_t)(z0 + uz) * shape.y + (y0 + uy)) * shape.x + (x0 + ux);
            v.x = pre[src] - ma;
            v.y = post[src] - mb;
        }
        dst[i] = v;
    }
}

// Z = FFT(a + i b), a and b real:  A(k) = (Z(k) + conj Z(-k)) / 2,  B(k) = (Z(k) - conj Z(-k)) / 2i.
// C(k) = A(k) conj(B(k)) / N is the spectrum of c[d] = sum_u b(u) a(u + d); C(-k) = conj C(k).
__global__ void k_cross_spectrum(float2 *__restrict__ buf, int count, I3 F, float inv_n)
{
    const int fv = F.z * F.y * F.x;
    const int64_t total = (int64_t)count * fv;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t % fv);
        const int kx = i % F.x, ky = (i / F.x) % F.y, kz = i / (F.x * F.y);
        const int mx = kx ? F.x - kx : 0, my = ky ? F.y - ky : 0, mz = kz ? F.z - kz : 0;
        const int j = (mz * F.y + my) * F.x + mx;
        if (j < i) continue;  // the partner thread writes both
        float2 *p = buf + (t - i);
        const float2 zk = p[i], zm = p[j];
        const float ar = 0.5f * (zk.x + zm.x), ai = 0.5f * (zk.y - zm.y);
        const float br = 0.5f * (zk.y + zm.y), bi = -0.5f * (zk.x - zm.x);
        const float cr = (ar * br + ai * bi) * inv_n, ci = (ai * br - ar * bi) * inv_n;
        p[i] = make_float2(cr, ci);
        if (j != i) p[j] = make_float2(cr, -ci);
    }
}

// ---------------------------------------------------------------------------------------------
// peak statistics of one full cross-correlation volume (lags -(p-1) .. p-1 per axis)
// ---------------------------------------------------------------------------------------------
struct PeakGeom {
    I3 F, patch, L;  // FFT box, patch, lag-volume extents 2p-1
};
__device__ __forceinline__ float real_of(float2 v) { return v.x; }
__device__ __forceinline__ float real_of(float v) { return v; }
template <typename T>
__device__ __forceinline__ float lag_value(const T *__restrict__ xc, const PeakGeom &g, int iz, int iy, int ix)
{
    int dz = iz - (g.patch.z - 1), dy = iy - (g.patch.y - 1), dx = ix - (g.patch.x - 1);
    dz = dz < 0 ? dz + g.F.z : dz;
    dy = dy < 0 ? dy + g.F.y : dy;
    dx = dx < 0 ? dx + g.F.x : dx;
    return real_of(xc[((int64_t)dz * g.F.y + dy) * g.F.x + dx]);
}

template <typename T>  // float2: the inverse transform's complex output (real part used); float: the real volume
__global__ void k_xcorr_peaks(const T *__restrict__ buf, PeakGeom g, int min_distance, int radius,
                              float threshold_rel, float *__restrict__ out, int64_t channel_stride,
                              int64_t node_begin)
{
    __shared__ float smv[4];
    __shared__ int smi[4];
    __shared__ float s_max, s_min;
    __shared__ int s_arg;
    const T *xc = buf + (int64_t)blockIdx.x * g.F.z * g.F.y * g.F.x;
    const int nL = g.L.z * g.L.y * g.L.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

    // global maximum, first index on ties (np.argmax)
    // one wave per lag row (iz, iy): the row's index arithmetic is done once, the lanes run along x; four rows are
    // requested before the first is compared (one load in flight per lane made this kernel latency bound)
    const int nrows = g.L.z * g.L.y, nwaves = (int)(blockDim.x >> 6);
    auto row_of = [&](int r) -> const T * {
        const int iz = r / g.L.y, iy = r - iz * g.L.y;
        int dz = iz - (g.patch.z - 1), dy = iy - (g.patch.y - 1);
        dz = dz < 0 ? dz + g.F.z : dz;
        dy = dy < 0 ? dy + g.F.y : dy;
        return xc + ((int64_t)dz * g.F.y + dy) * g.F.x;
    };
    constexpr int kRowsInFlight = 4;
    float bv = -INFINITY;
    int bi = nL;
    for (int ixb = 0; ixb < g.L.x; ixb += 64) {
        const int ix = ixb + lane;
        int dx = ix - (g.patch.x - 1);
        dx = dx < 0 ? dx + g.F.x : dx;
        const bool x_ok = ix < g.L.x;
        for (int r0 = w; r0 < nrows; r0 += nwaves * kRowsInFlight) {
            float vv[kRowsInFlight];
#pragma unroll
            for (int u = 0; u < kRowsInFlight; ++u) {
                const int r = min(r0 + u * nwaves, nrows - 1);
                vv[u] = x_ok ? real_of(row_of(r)[dx]) : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < kRowsInFlight; ++u) {
                const int r = r0 + u * nwaves;
                if (r >= nrows || !x_ok) continue;
                const int i = r * g.L.x + ix;
                if (vv[u] > bv || (vv[u] == bv && i < bi)) {
                    bv = vv[u];
                    bi = i;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(bv, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        smv[w] = bv;
        smi[w] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (smv[k] > bv || (smv[k] == bv && smi[k] < bi)) {
                bv = smv[k];
                bi = smi[k];
            }
        s_max = bv;
        s_arg = bi < nL ? bi : 0;
    }
    __syncthreads();
    const float vmax = s_max;
    const int px = s_arg % g.L.x, py = (s_arg / g.L.x) % g.L.y, pz = s_arg / (g.L.x * g.L.y);

    // minimum within +-radius of the peak (clipped to the volume)
    const int wz0 = max(0, pz - radius), wz1 = min(g.L.z, pz + radius + 1);
    const int wy0 = max(0, py - radius), wy1 = min(g.L.y, py + radius + 1);
    const int wx0 = max(0, px - radius), wx1 = min(g.L.x, px + radius + 1);
    const int wnx = wx1 - wx0, wny = wy1 - wy0, wn = (wz1 - wz0) * wny * wnx;
    float mn = INFINITY;
    for (int i = threadIdx.x; i < wn; i += blockDim.x) {
        const int ix = wx0 + i % wnx, iy = wy0 + (i / wnx) % wny, iz = wz0 + i / (wnx * wny);
        mn = fminf(mn, lag_value(xc, g, iz, iy, ix));
    }
    mn = block_min(mn, smv);
    if (threadIdx.x == 0) s_min = mn;

    // best other local maximum: v > threshold, no larger neighbour within +-min_distance,
    // outside the peak window
    const float thr = threshold_rel * vmax;
    float second = -INFINITY;
    for (int ixb = 0; ixb < g.L.x; ixb += 64) {
        const int ix = ixb + lane;
        int dx = ix - (g.patch.x - 1);
        dx = dx < 0 ? dx + g.F.x : dx;
        const bool x_ok = ix < g.L.x;
        for (int r0 = w; r0 < nrows; r0 += nwaves * kRowsInFlight) {
            float vv[kRowsInFlight];
#pragma unroll
            for (int u = 0; u < kRowsInFlight; ++u) {
                const int r = min(r0 + u * nwaves, nrows - 1);
                vv[u] = x_ok ? real_of(row_of(r)[dx]) : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < kRowsInFlight; ++u) {
                const int r = r0 + u * nwaves;
                if (r >= nrows || !x_ok) continue;
                const int iz = r / g.L.y, iy = r - iz * g.L.y;
                if (iz >= wz0 && iz < wz1 && iy >= wy0 && iy < wy1 && ix >= wx0 && ix < wx1) continue;
                const float v = vv[u];
                if (!(v > thr) || !(v > second)) continue;
                bool is_peak = true;
                const int nz0 = max(0, iz - min_distance), nz1 = min(g.L.z, iz + min_distance + 1);
                const int ny0 = max(0, iy - min_distance), ny1 = min(g.L.y, iy + min_distance + 1);
                const int nx0 = max(0, ix - min_distance), nx1 = min(g.L.x, ix + min_distance + 1);
                for (int a = nz0; a < nz1 && is_peak; ++a)
                    for (int b = ny0; b < ny1 && is_peak; ++b)
                        for (int c = nx0; c < nx1; ++c)
                            if (lag_value(xc, g, a, b, c) > v) {
                                is_peak = false;
                                break;
                            }
                if (is_peak) second = v;
            }
        }
    }
    second = block_max(second, smv);
    if (threadIdx.x == 0) {
        const float sharp = vmax / s_min;
        const float ratio = second > -INFINITY ? vmax / second : 0.0f;
        const int64_t n = node_begin + blockIdx.x;
        out[0 * channel_stride + n] = (float)(px - (g.patch.x - 1));
        out[1 * channel_stride + n] = (float)(py - (g.patch.y - 1));
        out[2 * channel_stride + n] = (float)(pz - (g.patch.z - 1));
        out[3 * channel_stride + n] = sharp;
        out[4 * channel_stride + n] = ratio;
    }
}


// ---------------------------------------------------------------------------------------------
// Fused patch cross-correlation for the common box 24 x 64 x 64 (patch <= 12 x 32 x 32): the rocFFT route above moves a
// node's 786-KB box through HBM about 25 times (two batched 3-D transforms of five kernels each, pack, cross spectrum,
// peaks).  Here the transforms run in LDS / registers and a node's data crosses memory four times, in pieces small
// enough to stay in the infinity cache between kernels:
//   k_xc_means      patch means (double sums, as k_patch_pack)
//   k_xc_xy_forward one workgroup per (node, patch plane): gather + mean-subtract + pack pre + i post, 64 x 64 2-D FFT
//                   in LDS (rows of the zero half skipped), plane -> G[node][pz][ky][kx]
//   k_xc_z_cross    one workgroup per (node, pair of rows ky, -ky): 24-point FFT along z in registers (planes >= pz are
//                   zero), Hermitian split + PRE conj(POST) / N against the partner column through LDS, inverse
//                   24-point FFT -> G2[node][Fz][ky][kx]
//   k_xc_xy_inverse one workgroup per (node, plane): inverse 2-D FFT in LDS, real part -> X[node][Fz][y][x] float
//   k_xcorr_peaks<float> on X.
// Same arithmetic as the rocFFT route up to the rounding of the transforms.
// ---------------------------------------------------------------------------------------------
__device__ __constant__ float kCos64[64] = {1.0f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f, 6.123234e-17f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f, -1.0f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f, -0.773010453f, -0.707106781f, -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f, -1.8369702e-16f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f, 0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f};
__device__ __constant__ float kSin64[64] = {0.0f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f, 0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f, 1.0f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f, 1.2246468e-16f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f, -1.0f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f, -0.773010453f, -0.707106781f, -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f};
constexpr int kXcF = 64, kXcFz = 24, kXcLS = 65;  // lateral box, axial box, LDS row stride (float2)

// x[3a + b] (a < 8, b < 3) -> X[c + 8d] (c < 8, d < 3)
template <bool INV>
__device__ __forceinline__ void dft24(float2 *v)
{
    const float cs[15][2] = {{1.0f, 0.0f}, {0.965925826f, 0.258819045f}, {0.866025404f, 0.5f}, {0.707106781f, 0.707106781f}, {0.5f, 0.866025404f}, {0.258819045f, 0.965925826f}, {6.123234e-17f, 1.0f}, {-0.258819045f, 0.965925826f}, {-0.5f, 0.866025404f}, {-0.707106781f, 0.707106781f}, {-0.866025404f, 0.5f}, {-0.965925826f, 0.258819045f}, {-1.0f, 1.2246468e-16f}, {-0.965925826f, -0.258819045f}, {-0.866025404f, -0.5f}};
    float2 u[3][8];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
        for (int a = 0; a < 8; ++a) u[b][a] = v[3 * a + b];
        dft8<INV>(u[b]);
    }
#pragma unroll
    for (int c = 1; c < 8; ++c) {
        u[1][c] = ctw<INV>(u[1][c], cs[c][0], cs[c][1]);
        u[2][c] = ctw<INV>(u[2][c], cs[2 * c][0], cs[2 * c][1]);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float2 a0 = u[0][c], a1 = u[1][c], a2 = u[2][c];
        dft3<INV>(a0, a1, a2);
        v[c] = a0;
        v[c + 8] = a1;
        v[c + 16] = a2;
    }
}

// 64-point FFT of up to 64 lines held in LDS, four threads per line (line = tid % 64, l = tid / 64: l is uniform over a
// wave): element n of a line at base[n * es].  n = l + 4 m, k = k1 + 16 k2:
//   X[k1 + 16 k2] = sum_l W64^(l k1) W4^(l k2) sum_m x[l + 4 m] W16^(m k1).
// Natural order in, natural order out, three barriers (all threads of the block must call; `active` masks the work).
template <bool INV>
__device__ __forceinline__ void fft64_lines(float2 *base, int es, int l, bool active)
{
    float2 v[16];
    if (active) {
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = base[(l + 4 * m) * es];
        dft16<INV>(v);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            const int j = l * k1;  // < 64, uniform over the wave
            base[(l + 4 * k1) * es] = k1 == 0 ? v[0] : ctw<INV>(v[k1], kCos64[j], kSin64[j]);
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k1 = 4 * q + l;
#pragma unroll
            for (int ll = 0; ll < 4; ++ll) v[4 * q + ll] = base[(ll + 4 * k1) * es];
            dft4<INV>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);  // X[k1 + 16 k2] at v[4 q + k2]
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) base[(4 * q + l + 16 * k2) * es] = v[4 * q + k2];
    }
    __syncthreads();
}

__global__ void k_xc_means(const float *__restrict__ pre, const float *__restrict__ post, I3 shape, I3 patch, I3 step,
                           I3 grid, int64_t node_begin, float2 *__restrict__ means)
{
    __shared__ double sm[4];
    const int64_t node = node_begin + blockIdx.x;
    const int gx = (int)(node % grid.x), gy = (int)((node / grid.x) % grid.y), gz = (int)(node / ((int64_t)grid.x * grid.y));
    const int z0 = gz * step.z, y0 = gy * step.y, x0 = gx * step.x;
    const int pv = patch.z * patch.y * patch.x;
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < pv; i += blockDim.x) {
        const int ux = i % patch.x, uy = (i / patch.x) % patch.y, uz = i / (patch.x * patch.y);
        const int64_t src = ((int64_t)(z0 + uz) * shape.y + (y0 + uy)) * shape.x + (x0 + ux);
        sa += (double)pre[src];
        sb += (double)post[src];
    }
    const double ta = block_sum(sa, sm);
    const double tb = block_sum(sb, sm);
    if (threadIdx.x == 0) means[blockIdx.x] = make_float2((float)(ta / (double)pv), (float)(tb / (double)pv));
}

__global__ __launch_bounds__(256) void k_xc_xy_forward(const float *__restrict__ pre, const float *__restrict__ post,
                                                       I3 shape, I3 patch, I3 step, I3 grid, int64_t node_begin,
                                                       const float2 *__restrict__ means, float2 *__restrict__ G)
{
    __shared__ float2 pl[kXcF * kXcLS];
    const int tid = threadIdx.x, line = tid & 63, l = tid >> 6;
    const int nb = blockIdx.x / patch.z, uz = blockIdx.x % patch.z;
    const int64_t node = node_begin + nb;
    const int gx = (int)(node % grid.x), gy = (int)((node / grid.x) % grid.y), gz = (int)(node / ((int64_t)grid.x * grid.y));
    const int z0 = gz * step.z + uz, y0 = gy * step.y, x0 = gx * step.x;
    const float2 mean = means[nb];
    for (int i = tid; i < kXcF * kXcF; i += 256) {
        const int x = i & 63, y = i >> 6;
        float2 v = make_float2(0.f, 0.f);
        if (x < patch.x && y < patch.y) {
            const int64_t src = ((int64_t)z0 * shape.y + (y0 + y)) * shape.x + (x0 + x);
            v = make_float2(pre[src] - mean.x, post[src] - mean.y);
        }
        pl[y * kXcLS + x] = v;
    }
    __syncthreads();
    fft64_lines<false>(pl + line * kXcLS, 1, l, line < patch.y);  // rows: the zero rows transform to zero
    fft64_lines<false>(pl + line, kXcLS, l, true);                // columns
    float2 *dst = G + (int64_t)blockIdx.x * (kXcF * kXcF);
    for (int i = tid; i < kXcF * kXcF; i += 256) dst[i] = pl[(i >> 6) * kXcLS + (i & 63)];
}

// Z = FFT(a + i b), a and b real:  A(k) = (Z(k) + conj Z(-k)) / 2,  B(k) = (Z(k) - conj Z(-k)) / 2i,  C = A conj(B) / N
__global__ __launch_bounds__(128) void k_xc_z_cross(const float2 *__restrict__ G, float2 *__restrict__ G2, int pz,
                                                    float inv_n)
{
    __shared__ float2 ex[kXcFz * 128];
    const int tid = threadIdx.x, kx = tid & 63, rho = tid >> 6;
    const int nb = blockIdx.x >> 5, r = blockIdx.x & 31;
    const int ky = r == 0 ? (rho ? 32 : 0) : (rho ? 64 - r : r);
    const float2 *src = G + ((int64_t)nb * pz * kXcF + ky) * kXcF + kx;
    float2 v[kXcFz];
#pragma unroll
    for (int z = 0; z < kXcFz; ++z) v[z] = z < pz ? src[(int64_t)z * (kXcF * kXcF)] : make_float2(0.f, 0.f);
    dft24<false>(v);
#pragma unroll
    for (int kz = 0; kz < kXcFz; ++kz) ex[kz * 128 + tid] = v[kz];
    __syncthreads();
    const int prho = r == 0 ? rho : 1 - rho, pkx = (64 - kx) & 63;
#pragma unroll
    for (int kz = 0; kz < kXcFz; ++kz) {
        const float2 zk = v[kz], zm = ex[((kXcFz - kz) % kXcFz) * 128 + prho * 64 + pkx];
        const float ar = 0.5f * (zk.x + zm.x), ai = 0.5f * (zk.y - zm.y);
        const float br = 0.5f * (zk.y + zm.y), bi = -0.5f * (zk.x - zm.x);
        v[kz] = make_float2((ar * br + ai * bi) * inv_n, (ai * br - ar * bi) * inv_n);
    }
    dft24<true>(v);
    float2 *dst = G2 + ((int64_t)nb * kXcFz * kXcF + ky) * kXcF + kx;
    // only the planes that hold lags -(pz - 1) .. pz - 1 are read again
#pragma unroll
    for (int z = 0; z < kXcFz; ++z)
        if (z < pz || z > kXcFz - pz) dst[(int64_t)z * (kXcF * kXcF)] = v[z];
}

__global__ __launch_bounds__(256) void k_xc_xy_inverse(const float2 *__restrict__ G2, float *__restrict__ X, int pz)
{
    __shared__ float2 pl[kXcF * kXcLS];
    const int tid = threadIdx.x, line = tid & 63, l = tid >> 6;
    // the 2 pz - 1 planes of lags -(pz - 1) .. pz - 1: box planes 0 .. pz - 1 and Fz - (pz - 1) .. Fz - 1
    const int nlag = 2 * pz - 1, nb = blockIdx.x / nlag, q = blockIdx.x % nlag;
    const int64_t plane = (int64_t)nb * kXcFz + (q < pz ? q : kXcFz - nlag + q);
    const float2 *src = G2 + plane * (kXcF * kXcF);
    for (int i = tid; i < kXcF * kXcF; i += 256) pl[(i >> 6) * kXcLS + (i & 63)] = src[i];
    __syncthreads();
    fft64_lines<true>(pl + line, kXcLS, l, true);
    fft64_lines<true>(pl + line * kXcLS, 1, l, true);
    float *dst = X + plane * (kXcF * kXcF);
    for (int i = tid; i < kXcF * kXcF; i += 256) dst[i] = pl[(i >> 6) * kXcLS + (i & 63)].x;
}

// ---------------------------------------------------------------------------------------------
// sub-pixel refinement: argmin over K^3 fractional candidates of the normalised SSD between the
// post patch and the order-1 sampled pre patch (zero outside the image)
// ---------------------------------------------------------------------------------------------
constexpr int kMaxOffsets = 5;
struct RefineArgs {
    I3 shape, patch, step, region;  // region = extents of the pre window held in LDS
    int n_off;
    float off[kMaxOffsets];
    float off_min, off_max;
    float eps;
};

// order-1 sample of patch voxel i from the LDS window (weights identical for the whole patch)
__device__ __forceinline__ float tri_sample(const float *__restrict__ origin, const I3 &patch, const I3 &region,
                                            int i, const float (&wz)[2], const float (&wy)[2],
                                            const float (&wx)[2])
{
    const int ux = i % patch.x, uy = (i / patch.x) % patch.y, uz = i / (patch.x * patch.y);
    const float *p = origin + (uz * region.y + uy) * region.x + ux;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int d = 0; d < 2; ++d) s += p[(a * region.y + b) * region.x + d] * wz[a] * wy[b] * wx[d];
    return s;
}

__global__ void k_subpixel_refine(const float *__restrict__ pre, const float *__restrict__ post,
                                  RefineArgs A, const int *__restrict__ nodes,
                                  const float *__restrict__ vec, float *__restrict__ out)
{
    extern __shared__ float lds[];
    __shared__ float smf[4];
    __shared__ float s_scalar[2];
    __shared__ float s_scores[kMaxOffsets * kMaxOffsets * kMaxOffsets];
    const int n = blockIdx.x;
    const int pv = A.patch.z * A.patch.y * A.patch.x;
    float *postn = lds;     // normalised post patch
    float *reg = lds + pv;  // pre window
    const int gz = nodes[3 * n + 0], gy = nodes[3 * n + 1], gx = nodes[3 * n + 2];
    const float vx = vec[3 * n + 0], vy = vec[3 * n + 1], vz = vec[3 * n + 2];
    const int sz = gz * A.step.z, sy = gy * A.step.y, sx = gx * A.step.x;
    const float bz = (float)sz + vz, by = (float)sy + vy, bx = (float)sx + vx;

    // post patch -> LDS, normalised
    float acc = 0.f;
    for (int i = threadIdx.x; i < pv; i += blockDim.x) {
        const int ux = i % A.patch.x, uy = (i / A.patch.x) % A.patch.y, uz = i / (A.patch.x * A.patch.y);
        const int z = sz + uz, y = sy + uy, x = sx + ux;
        float v = 0.f;
        if (z < A.shape.z && y < A.shape.y && x < A.shape.x)
            v = post[((int64_t)z * A.shape.y + y) * A.shape.x + x];
        postn[i] = v;
        acc += v;
    }
    float t = block_sum(acc, smf);
    if (threadIdx.x == 0) s_scalar[0] = t / (float)pv;
    __syncthreads();
    float mean = s_scalar[0];
    acc = 0.f;
    for (int i = threadIdx.x; i < pv; i += blockDim.x) {
        const float c = postn[i] - mean;
        acc += c * c;
    }
    t = block_sum(acc, smf);
    if (threadIdx.x == 0) s_scalar[1] = fmaxf(sqrtf(t / (float)pv), A.eps);
    __syncthreads();
    float den = s_scalar[1];
    for (int i = threadIdx.x; i < pv; i += blockDim.x) postn[i] = (postn[i] - mean) / den;

    // pre window -> LDS (zero outside the image: map_coordinates mode="constant", cval=0)
    const int rz0 = (int)floorf(bz + A.off_min), ry0 = (int)floorf(by + A.off_min),
              rx0 = (int)floorf(bx + A.off_min);
    const int rv = A.region.z * A.region.y * A.region.x;
    for (int i = threadIdx.x; i < rv; i += blockDim.x) {
        const int ux = i % A.region.x, uy = (i / A.region.x) % A.region.y, uz = i / (A.region.x * A.region.y);
        const int z = rz0 + uz, y = ry0 + uy, x = rx0 + ux;
        float v = 0.f;
        if (z >= 0 && y >= 0 && x >= 0 && z < A.shape.z && y < A.shape.y && x < A.shape.x)
            v = pre[((int64_t)z * A.shape.y + y) * A.shape.x + x];
        reg[i] = v;
    }
    __syncthreads();

    const int K = A.n_off, nc = K * K * K;
    for (int c = 0; c < nc; ++c) {
        const float oz = A.off[c / (K * K)], oy = A.off[(c / K) % K], ox = A.off[c % K];
        const float cz = bz + oz, cy = by + oy, cx = bx + ox;  // patch start in the pre image
        const float fz = floorf(cz), fy = floorf(cy), fx = floorf(cx);
        const float tz = cz - fz, ty = cy - fy, tx = cx - fx;
        const int lz = (int)fz - rz0, ly = (int)fy - ry0, lx = (int)fx - rx0;
        const float wz[2] = {1.f - tz, tz}, wy[2] = {1.f - ty, ty}, wx[2] = {1.f - tx, tx};
        const float *origin = reg + (lz * A.region.y + ly) * A.region.x + lx;
#define M3D_SAMPLE(i) tri_sample(origin, A.patch, A.region, (i), wz, wy, wx)
        acc = 0.f;
        for (int i = threadIdx.x; i < pv; i += blockDim.x) acc += M3D_SAMPLE(i);
        t = block_sum(acc, smf);
        if (threadIdx.x == 0) s_scalar[0] = t / (float)pv;
        __syncthreads();
        mean = s_scalar[0];
        acc = 0.f;
        for (int i = threadIdx.x; i < pv; i += blockDim.x) {
            const float q = M3D_SAMPLE(i) - mean;
            acc += q * q;
        }
        t = block_sum(acc, smf);
        if (threadIdx.x == 0) s_scalar[1] = fmaxf(sqrtf(t / (float)pv), A.eps);
        __syncthreads();
        den = s_scalar[1];
        acc = 0.f;
        for (int i = threadIdx.x; i < pv; i += blockDim.x) {
            const float q = (M3D_SAMPLE(i) - mean) / den - postn[i];
            acc += q * q;
        }
        t = block_sum(acc, smf);
        if (threadIdx.x == 0) s_scores[c] = t / (float)pv;
        __syncthreads();
#undef M3D_SAMPLE
    }
    if (threadIdx.x == 0) {
        int best = 0;
        for (int c = 1; c < nc; ++c)
            if (s_scores[c] < s_scores[best]) best = c;  // first minimum (argmin)
        out[3 * n + 0] = vx + A.off[best % K];
        out[3 * n + 1] = vy + A.off[(best / K) % K];
        out[3 * n + 2] = vz + A.off[best / (K * K)];
    }
}

// ---------------------------------------------------------------------------------------------
// elastic mesh relaxation: velocity-Verlet + FIRE
// ---------------------------------------------------------------------------------------------
struct FireState {
    float dt, alpha;
    int n_pos, pad;
};
struct MeshGeom {
    I3 grid;
    float stride[3];  // xyz
    float k0, k;
};
__constant__ int c_links[9][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, -1, 0},
                                  {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1}};  // xyz

constexpr int kMeshSlots = 16;  // records (4 doubles each) the per-step sums are spread over (64: the readers get slower)
// mode 0: a = F(x) only (initialisation); mode 1: a_new = F(x), v += dt/2 (a + a_new), sums
__global__ void k_mesh_force(const float *__restrict__ x, const float *__restrict__ prev,
                             float *__restrict__ a, float *__restrict__ v, MeshGeom g, float cap,
                             const FireState *__restrict__ st, double *__restrict__ sums, int mode)
{
    __shared__ double sm[4];
    const int64_t N = (int64_t)g.grid.z * g.grid.y * g.grid.x;
    const float dt = st->dt;
    double sp = 0.0, svv = 0.0, saa = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N;
         n += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(n % g.grid.x), iy = (int)((n / g.grid.x) % g.grid.y),
                  iz = (int)(n / ((int64_t)g.grid.x * g.grid.y));
        const float x0 = x[n], x1 = x[N + n], x2 = x[2 * N + n];
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
        for (int l = 0; l < 9; ++l) {
            const int dx = c_links[l][0], dy = c_links[l][1], dz = c_links[l][2];
            const float l0x = g.stride[0] * (float)dx, l0y = g.stride[1] * (float)dy,
                        l0z = g.stride[2] * (float)dz;
            const float l0n = sqrtf(l0x * l0x + l0y * l0y + l0z * l0z);
            const float kl = g.k * g.stride[0] / l0n;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // s = 0: this node is the base of the link, partner at +d; s = 1: partner at -d
                const int jx = s ? ix - dx : ix + dx, jy = s ? iy - dy : iy + dy, jz = s ? iz - dz : iz + dz;
                if (jx < 0 || jy < 0 || jz < 0 || jx >= g.grid.x || jy >= g.grid.y || jz >= g.grid.z) continue;
                const int64_t m = ((int64_t)jz * g.grid.y + jy) * g.grid.x + jx;
                const float y0 = x[m], y1 = x[N + m], y2 = x[2 * N + m];
                const float ex = s ? (x0 - y0) + l0x : (y0 - x0) + l0x;
                const float ey = s ? (x1 - y1) + l0y : (y1 - x1) + l0y;
                const float ez = s ? (x2 - y2) + l0z : (y2 - x2) + l0z;
                const float ln = sqrtf(ex * ex + ey * ey + ez * ez);
                const float fac = kl * (1.0f - l0n / ln);
                if (s) {
                    f0 -= fac * ex;
                    f1 -= fac * ey;
                    f2 -= fac * ez;
                } else {
                    f0 += fac * ex;
                    f1 += fac * ey;
                    f2 += fac * ez;
                }
            }
        }
        const float p0 = prev[n], p1 = prev[N + n], p2 = prev[2 * N + n];
        if (isfinite(p0)) f0 += g.k0 * (p0 - x0);
        if (isfinite(p1)) f1 += g.k0 * (p1 - x1);
        if (isfinite(p2)) f2 += g.k0 * (p2 - x2);
        f0 = fminf(fmaxf(f0, -cap), cap);
        f1 = fminf(fmaxf(f1, -cap), cap);
        f2 = fminf(fmaxf(f2, -cap), cap);
        if (mode == 1) {
            const float h = 0.5f * dt;
            const float v0 = v[n] + h * (a[n] + f0), v1 = v[N + n] + h * (a[N + n] + f1),
                        v2 = v[2 * N + n] + h * (a[2 * N + n] + f2);
            v[n] = v0;
            v[N + n] = v1;
            v[2 * N + n] = v2;
            sp += (double)f0 * v0 + (double)f1 * v1 + (double)f2 * v2;
            svv += (double)v0 * v0 + (double)v1 * v1 + (double)v2 * v2;
            saa += (double)f0 * f0 + (double)f1 * f1 + (double)f2 * f2;
        }
        a[n] = f0;
        a[N + n] = f1;
        a[2 * N + n] = f2;
    }
    if (mode == 1) {
        const double t0 = block_sum(sp, sm);
        const double t1 = block_sum(svv, sm);
        const double t2 = block_sum(saa, sm);
        if (threadIdx.x == 0) {
            // one of kMeshSlots records: 693 workgroups adding to ONE address serialise (26 of this kernel's 35 us)
            double *slot = sums + 4 * (blockIdx.x % kMeshSlots);
            atomicAdd(&slot[0], t0);
            atomicAdd(&slot[1], t1);
            atomicAdd(&slot[2], t2);
        }
    }
}

// FIRE velocity mixing and time-step control, then the position update of the NEXT step into
// x_next (x_cur stays the answer if the run stops here).
// mode 0: position update only (first step).  collect: accumulate max|v| and sum v^2.
__global__ void k_mesh_fire(const float *__restrict__ x_cur, float *__restrict__ x_next,
                            float *__restrict__ v, const float *__restrict__ a, int64_t N3,
                            const FireState *__restrict__ st_in, FireState *__restrict__ st_out,
                            const double *__restrict__ sums, double *__restrict__ sums_next, float dt_max,
                            int mode, int collect, unsigned int *__restrict__ vmax_bits,
                            unsigned int *__restrict__ amax_bits, double *__restrict__ ekin)
{
    __shared__ double sm[4];
    __shared__ float smf[4];
    FireState s = *st_in;
    float mix = 0.f;
    bool zero_v = false;
    if (mode == 1) {
        double power = 0.0, svv = 0.0, saa = 0.0;
        for (int k = 0; k < kMeshSlots; ++k) {
            power += sums[4 * k];
            svv += sums[4 * k + 1];
            saa += sums[4 * k + 2];
        }
        const float v_norm = (float)sqrt(svv);
        const float a_norm = (float)sqrt(saa + 1e-6);
        mix = v_norm / a_norm;
        const float alpha = s.alpha;  // mixing uses the pre-update alpha
        s.n_pos = power >= 0.0 ? s.n_pos + 1 : 0;
        if (power > 0.0 && s.n_pos > 5) {
            s.dt = fminf(s.dt * 1.1f, dt_max);
            s.alpha = s.alpha * 0.99f;
        }
        if (power < 0.0) {
            s.dt = s.dt * 0.5f;
            s.alpha = 0.1f;
            zero_v = true;
        }
        mix *= alpha;
        // v <- v + alpha (a |v|/|a| - v)
        const float keep = 1.0f - alpha;
        float vm = 0.f, am = 0.f;
        double e = 0.0;
        const float dt = s.dt, hdt2 = 0.5f * s.dt * s.dt;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N3;
             i += (int64_t)gridDim.x * blockDim.x) {
            const float ai = a[i];
            float vi = zero_v ? 0.f : keep * v[i] + mix * ai;
            v[i] = vi;
            x_next[i] = x_cur[i] + dt * vi + hdt2 * ai;
            vm = fmaxf(vm, fabsf(vi));
            am = fmaxf(am, fabsf(ai));
            e += (double)vi * vi;
        }
        if (collect) {
            vm = block_max(vm, smf);
            am = block_max(am, smf);
            const double et = block_sum(e, sm);
            if (threadIdx.x == 0) {
                atomicMax(vmax_bits, __float_as_uint(vm));
                atomicMax(amax_bits, __float_as_uint(am));
                atomicAdd(ekin, 0.5 * et);
            }
        }
    } else {
        const float dt = s.dt, hdt2 = 0.5f * s.dt * s.dt;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N3;
             i += (int64_t)gridDim.x * blockDim.x)
            x_next[i] = x_cur[i] + dt * v[i] + hdt2 * a[i];
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) *st_out = s;
        for (int k = threadIdx.x; k < 4 * kMeshSlots; k += blockDim.x) sums_next[k] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void k_flow_compose(const float *__restrict__ base, const float *__restrict__ res, I3 grid,
                               float sz, float sy, float sx, float bz, float by, float bx,
                               float *__restrict__ out)
{
    const int64_t N = (int64_t)grid.z * grid.y * grid.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N;
         n += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(n % grid.x), iy = (int)((n / grid.x) % grid.y),
                  iz = (int)(n / ((int64_t)grid.x * grid.y));
        const float rx = res[n], ry = res[N + n], rz = res[2 * N + n];
        // float32 like the reference: shifted = box + grid * stride + residual; (shifted - box) / stride
        const float cz = ((bz + (float)iz * sz + rz) - bz) / sz;
        const float cy = ((by + (float)iy * sy + ry) - by) / sy;
        const float cx = ((bx + (float)ix * sx + rx) - bx) / sx;
        out[n] = rx + (float)sample_linear3(base, grid.z, grid.y, grid.x, (double)cz, (double)cy, (double)cx,
                                            MODE_NEAREST, 0.0);
        out[N + n] = ry + (float)sample_linear3(base + N, grid.z, grid.y, grid.x, (double)cz, (double)cy,
                                                (double)cx, MODE_NEAREST, 0.0);
        out[2 * N + n] = rz + (float)sample_linear3(base + 2 * N, grid.z, grid.y, grid.x, (double)cz,
                                                    (double)cy, (double)cx, MODE_NEAREST, 0.0);
    }
}

__global__ void k_moments(const float *__restrict__ src, int64_t n, double *__restrict__ out)
{
    __shared__ double sm[4];
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double v = src[i];
        s += v;
        q += v * v;
    }
    const double ts = block_sum(s, sm);
    const double tq = block_sum(q, sm);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], ts);
        atomicAdd(&out[1], tq);
    }
}

__global__ void k_norm_sq_diff(const float *__restrict__ a, const float *__restrict__ b, int64_t n, float ma,
                               float sa, float mb, float sb, double *__restrict__ out)
{
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float d = (a[i] - ma) / sa - (b[i] - mb) / sb;
        s += (double)(d * d);
    }
    const double t = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(out, t);
}

inline bool i3_positive(const int32_t *v) { return v && v[0] > 0 && v[1] > 0 && v[2] > 0; }
inline I3 to_i3(const int32_t *v) { return I3{v[0], v[1], v[2]}; }

}  // namespace

extern "C" {

int m3d_patch_pack(const float *pre, const float *post, const int32_t shape_zyx[3],
                   const int32_t patch_zyx[3], const int32_t step_zyx[3], const int32_t grid_zyx[3],
                   int64_t node_begin, int32_t node_count, const int32_t fft_zyx[3], void *out_c64,
                   void *stream)
{
    M3D_REQUIRE(pre && post && out_c64, "null pointer");
    M3D_REQUIRE(i3_positive(shape_zyx) && i3_positive(patch_zyx) && i3_positive(step_zyx) &&
                    i3_positive(grid_zyx) && i3_positive(fft_zyx),
                "extents must be positive");
    const int64_t nodes = (int64_t)grid_zyx[0] * grid_zyx[1] * grid_zyx[2];
    M3D_REQUIRE(node_begin >= 0 && node_count >= 0 && node_begin + node_count <= nodes,
                "node range outside the flow grid");
    for (int d = 0; d < 3; ++d) {
        M3D_REQUIRE((int64_t)(grid_zyx[d] - 1) * step_zyx[d] + patch_zyx[d] <= shape_zyx[d],
                    "last patch reaches outside the image");
        M3D_REQUIRE(fft_zyx[d] >= 2 * patch_zyx[d] - 1, "FFT box smaller than the linear correlation");
    }
    if (node_count == 0) return M3D_OK;
    hipLaunchKernelGGL(k_patch_pack, dim3(node_count), dim3(kBlock), 0, (hipStream_t)stream, pre, post,
                       to_i3(shape_zyx), to_i3(patch_zyx), to_i3(step_zyx), to_i3(grid_zyx), node_begin,
                       to_i3(fft_zyx), (float2 *)out_c64);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_patch_cross_spectrum(void *buf_c64, int32_t count, const int32_t fft_zyx[3], void *stream)
{
    M3D_REQUIRE(buf_c64 && i3_positive(fft_zyx) && count >= 0, "bad arguments");
    if (count == 0) return M3D_OK;
    const int64_t fv = (int64_t)fft_zyx[0] * fft_zyx[1] * fft_zyx[2];
    M3D_REQUIRE(fv < (int64_t)1 << 31, "FFT box too large");
    hipLaunchKernelGGL(k_cross_spectrum, dim3(grid_for(fv * count, 4)), dim3(kBlock), 0,
                       (hipStream_t)stream, (float2 *)buf_c64, count, to_i3(fft_zyx), 1.0f / (float)fv);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_xcorr_peaks(const void *buf_c64, int32_t count, const int32_t fft_zyx[3],
                    const int32_t patch_zyx[3], int32_t min_distance, int32_t peak_radius,
                    float threshold_rel, float *flow_out, int64_t channel_stride, int64_t node_begin,
                    void *stream)
{
    M3D_REQUIRE(buf_c64 && flow_out && i3_positive(fft_zyx) && i3_positive(patch_zyx) && count >= 0,
                "bad arguments");
    M3D_REQUIRE(min_distance >= 0 && peak_radius >= 0, "peak windows must be non-negative");
    M3D_REQUIRE(node_begin >= 0 && node_begin + count <= channel_stride, "output range");
    PeakGeom g;
    g.F = to_i3(fft_zyx);
    g.patch = to_i3(patch_zyx);
    g.L = I3{2 * patch_zyx[0] - 1, 2 * patch_zyx[1] - 1, 2 * patch_zyx[2] - 1};
    M3D_REQUIRE(g.F.z >= g.L.z && g.F.y >= g.L.y && g.F.x >= g.L.x, "FFT box smaller than the lag volume");
    M3D_REQUIRE((int64_t)g.L.z * g.L.y * g.L.x < (int64_t)1 << 31, "lag volume too large");
    if (count == 0) return M3D_OK;
    hipLaunchKernelGGL(k_xcorr_peaks<float2>, dim3(count), dim3(kBlock), 0, (hipStream_t)stream,
                       (const float2 *)buf_c64, g, min_distance, peak_radius, threshold_rel, flow_out,
                       channel_stride, node_begin);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

size_t m3d_patch_xcorr_fused_workspace_bytes(const int32_t patch_zyx[3], int32_t count)
{
    if (!i3_positive(patch_zyx) || count <= 0) return 0;
    const size_t plane = (size_t)kXcF * kXcF;
    const size_t a = std::max((size_t)patch_zyx[0] * plane * sizeof(float2), (size_t)kXcFz * plane * sizeof(float));
    return (size_t)count * (a + (size_t)kXcFz * plane * sizeof(float2) + sizeof(float2)) + 512;
}

int m3d_patch_xcorr_fused_supported(const int32_t patch_zyx[3], const int32_t fft_zyx[3])
{
    return i3_positive(patch_zyx) && i3_positive(fft_zyx) && fft_zyx[0] == kXcFz && fft_zyx[1] == kXcF &&
           fft_zyx[2] == kXcF && 2 * patch_zyx[0] - 1 <= kXcFz && 2 * patch_zyx[1] - 1 <= kXcF &&
           2 * patch_zyx[2] - 1 <= kXcF;
}

int m3d_patch_xcorr_fused(const float *pre, const float *post, const int32_t shape_zyx[3], const int32_t patch_zyx[3],
                          const int32_t step_zyx[3], const int32_t grid_zyx[3], int64_t node_begin, int32_t count,
                          int32_t min_distance, int32_t peak_radius, float threshold_rel, void *work, size_t work_bytes,
                          float *flow_out, int64_t channel_stride, void *stream)
{
    M3D_REQUIRE(pre && post && work && flow_out, "null pointer");
    M3D_REQUIRE(i3_positive(shape_zyx) && i3_positive(patch_zyx) && i3_positive(step_zyx) && i3_positive(grid_zyx),
                "bad extents");
    const int32_t fft[3] = {kXcFz, kXcF, kXcF};
    if (!m3d_patch_xcorr_fused_supported(patch_zyx, fft)) {
        set_err("m3d_patch_xcorr_fused", "patch does not fit the 24 x 64 x 64 box");
        return M3D_ERR_UNSUPPORTED;
    }
    M3D_REQUIRE(min_distance >= 0 && peak_radius >= 0, "peak windows must be non-negative");
    M3D_REQUIRE(node_begin >= 0 && count >= 0 && node_begin + count <= channel_stride, "output range");
    const int64_t nodes = (int64_t)grid_zyx[0] * grid_zyx[1] * grid_zyx[2];
    M3D_REQUIRE(node_begin + count <= nodes, "node range outside the grid");
    for (int a = 0; a < 3; ++a)
        M3D_REQUIRE((int64_t)(grid_zyx[a] - 1) * step_zyx[a] + patch_zyx[a] <= shape_zyx[a], "grid leaves the volume");
    if (work_bytes < m3d_patch_xcorr_fused_workspace_bytes(patch_zyx, count)) {
        set_err("workspace", "smaller than m3d_patch_xcorr_fused_workspace_bytes");
        return M3D_ERR_WORKSPACE;
    }
    if (count == 0) return M3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const I3 shape = to_i3(shape_zyx), patch = to_i3(patch_zyx), step = to_i3(step_zyx), grid = to_i3(grid_zyx);
    const size_t plane = (size_t)kXcF * kXcF;
    const size_t a_bytes = std::max((size_t)patch.z * plane * sizeof(float2), (size_t)kXcFz * plane * sizeof(float));
    char *w = (char *)work;
    float2 *means = (float2 *)w;
    w += (((size_t)count * sizeof(float2) + 255) / 256) * 256;
    float2 *G = (float2 *)w;   // [count][pz][64][64] c64, later X [count][24][64][64] f32
    float *X = (float *)w;
    w += (size_t)count * a_bytes;
    float2 *G2 = (float2 *)w;  // [count][24][64][64] c64
    hipLaunchKernelGGL(k_xc_means, dim3(count), dim3(kBlock), 0, st, pre, post, shape, patch, step, grid, node_begin,
                       means);
    hipLaunchKernelGGL(k_xc_xy_forward, dim3(count * patch.z), dim3(256), 0, st, pre, post, shape, patch, step, grid,
                       node_begin, means, G);
    hipLaunchKernelGGL(k_xc_z_cross, dim3(count * 32), dim3(128), 0, st, G, G2, patch.z,
                       1.0f / (float)(kXcFz * kXcF * kXcF));
    hipLaunchKernelGGL(k_xc_xy_inverse, dim3(count * (2 * patch.z - 1)), dim3(256), 0, st, G2, X, patch.z);
    PeakGeom g;
    g.F = I3{kXcFz, kXcF, kXcF};
    g.patch = patch;
    g.L = I3{2 * patch.z - 1, 2 * patch.y - 1, 2 * patch.x - 1};
    hipLaunchKernelGGL(k_xcorr_peaks<float>, dim3(count), dim3(kBlock), 0, st, (const float *)X, g, min_distance,
                       peak_radius, threshold_rel, flow_out, channel_stride, node_begin);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

// Three offsets per axis whose span fits a three-point window (the configured -0.5, 0, 0.5): all 27 candidates share
// their gathers.  Order-1 interpolation is separable and its weights are the same for every voxel of a patch, so a
// voxel's 27 samples come from its 3 x 3 x 3 window neighbourhood through three small contractions (x, y, z: 81 fused
// multiply-adds each) instead of 27 x 8 gathers.  Two sweeps over the patch: the first collects sum and sum of squares
// per candidate (mean, deviation), the second the normalised squared difference itself,
//     score = mean(((S - mean S) / sd_s - (P - mean P) / sd_p)^2),    sd = max(sqrt(var), eps)     (:373-399)
// in that form - NOT expanded into 2 - 2 corr, which loses the tiny positive scores of almost-identical candidates to
// rounding and then prefers them to an exactly matching one.  Values are shifted by the window mean first so that
// the variances are not differences of large numbers.
__device__ __forceinline__ void sample27(const float *__restrict__ w0, int Ry, int Rx, const float (&wz)[3][3],
                                         const float (&wy)[3][3], const float (&wx)[3][3], float (&S)[27])
{
    float B[3][3][3];  // [window z][candidate y][candidate x]
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float Ar[3][3];  // [window y][candidate x]
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float *r = w0 + (a * Ry + b) * Rx;
            const float c0 = r[0], c1 = r[1], c2 = r[2];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) Ar[b][kx] = c0 * wx[kx][0] + c1 * wx[kx][1] + c2 * wx[kx][2];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                B[a][ky][kx] = Ar[0][kx] * wy[ky][0] + Ar[1][kx] * wy[ky][1] + Ar[2][kx] * wy[ky][2];
    }
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                S[(kz * 3 + ky) * 3 + kx] =
                    B[0][ky][kx] * wz[kz][0] + B[1][ky][kx] * wz[kz][1] + B[2][ky][kx] * wz[kz][2];
}

__global__ __launch_bounds__(256) void k_subpixel_refine3(const float *__restrict__ pre, const float *__restrict__ post,
                                                          RefineArgs A, const int *__restrict__ nodes,
                                                          const float *__restrict__ vec, float *__restrict__ out)
{
    extern __shared__ float win[];  // pre window minus its mean
    __shared__ double smd[4];
    __shared__ float s_m0;
    __shared__ float s_part[4][56];
    __shared__ float s_mean[28], s_rsd[28];  // per candidate, [27] = the post patch
    const int n = blockIdx.x, tid = threadIdx.x;
    const int pv = A.patch.z * A.patch.y * A.patch.x;
    const int gz = nodes[3 * n + 0], gy = nodes[3 * n + 1], gx = nodes[3 * n + 2];
    const float vx = vec[3 * n + 0], vy = vec[3 * n + 1], vz = vec[3 * n + 2];
    const int sz = gz * A.step.z, sy = gy * A.step.y, sx = gx * A.step.x;
    const float bz = (float)sz + vz, by = (float)sy + vy, bx = (float)sx + vx;
    const int rz0 = (int)floorf(bz + A.off_min), ry0 = (int)floorf(by + A.off_min), rx0 = (int)floorf(bx + A.off_min);
    const int Ry = A.region.y, Rx = A.region.x;
    const int rv = A.region.z * Ry * Rx;

    // pre window -> LDS (zero outside the image: map_coordinates mode="constant", cval=0), then minus its mean
    float acc = 0.f;
    {
        int ux = tid % Rx, uy = (tid / Rx) % Ry, uz = tid / (Rx * Ry);
        const int qx = 256 % Rx, qy = (256 / Rx) % Ry, qz = 256 / (Rx * Ry);
        for (int i = tid; i < rv; i += 256) {
            const int z = rz0 + uz, y = ry0 + uy, x = rx0 + ux;
            float v = 0.f;
            if (z >= 0 && y >= 0 && x >= 0 && z < A.shape.z && y < A.shape.y && x < A.shape.x)
                v = pre[((int64_t)z * A.shape.y + y) * A.shape.x + x];
            win[i] = v;
            acc += v;
            ux += qx;
            int c = ux >= Rx ? 1 : 0;
            ux -= c ? Rx : 0;
            uy += qy + c;
            c = uy >= Ry ? 1 : 0;
            uy -= c ? Ry : 0;
            uz += qz + c;
        }
    }
    const double wsum = block_sum((double)acc, smd);
    if (tid == 0) s_m0 = (float)(wsum / (double)rv);
    __syncthreads();
    const float m0 = s_m0;
    for (int i = tid; i < rv; i += 256) win[i] -= m0;
    __syncthreads();

    // per-axis weights of the three candidates over the three window points (uniform over the block)
    float wz[3][3], wy[3][3], wx[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float cz = bz + A.off[k], cy = by + A.off[k], cx = bx + A.off[k];
        const float fz = floorf(cz), fy = floorf(cy), fx = floorf(cx);
        const float tz = cz - fz, ty = cy - fy, tx = cx - fx;
        const int lz = (int)fz - rz0, ly = (int)fy - ry0, lx = (int)fx - rx0;  // 0 or 1
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            wz[k][j] = j == lz ? 1.f - tz : (j == lz + 1 ? tz : 0.f);
            wy[k][j] = j == ly ? 1.f - ty : (j == ly + 1 ? ty : 0.f);
            wx[k][j] = j == lx ? 1.f - tx : (j == lx + 1 ? tx : 0.f);
        }
    }
    const int px = A.patch.x, py = A.patch.y;
    const int qx = 256 % px, qy = (256 / px) % py, qz = 256 / (px * py);
    const int ux0 = tid % px, uy0 = (tid / px) % py, uz0 = tid / (px * py);
    auto advance = [&](int &ux, int &uy, int &uz) {
        ux += qx;
        int c = ux >= px ? 1 : 0;
        ux -= c ? px : 0;
        uy += qy + c;
        c = uy >= py ? 1 : 0;
        uy -= c ? py : 0;
        uz += qz + c;
    };
    auto post_at = [&](int uz, int uy, int ux) -> float {
        const int z = sz + uz, y = sy + uy, x = sx + ux;
        float pval = 0.f;
        if (z < A.shape.z && y < A.shape.y && x < A.shape.x) pval = post[((int64_t)z * A.shape.y + y) * A.shape.x + x];
        return pval - m0;
    };
    const int lane = tid & 63, wv = tid >> 6;
    const float inv = 1.0f / (float)pv;

    // sweep 1: sum and sum of squares of every candidate and of the post patch
    {
        float s1[28], s2[28];
#pragma unroll
        for (int c = 0; c < 28; ++c) s1[c] = s2[c] = 0.f;
        int ux = ux0, uy = uy0, uz = uz0;
        for (int i = tid; i < pv; i += 256) {
            float S[27];
            sample27(win + (uz * Ry + uy) * Rx + ux, Ry, Rx, wz, wy, wx, S);
            const float P = post_at(uz, uy, ux);
#pragma unroll
            for (int c = 0; c < 27; ++c) {
                s1[c] += S[c];
                s2[c] += S[c] * S[c];
            }
            s1[27] += P;
            s2[27] += P * P;
            advance(ux, uy, uz);
        }
#pragma unroll
        for (int c = 0; c < 28; ++c) {
            s1[c] = wave_sum(s1[c]);
            s2[c] = wave_sum(s2[c]);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 28; ++c) {
                s_part[wv][c] = s1[c];
                s_part[wv][28 + c] = s2[c];
            }
        }
        __syncthreads();
        if (tid < 28) {
            const double a = (double)s_part[0][tid] + (double)s_part[1][tid] + (double)s_part[2][tid] + (double)s_part[3][tid];
            const double b = (double)s_part[0][28 + tid] + (double)s_part[1][28 + tid] + (double)s_part[2][28 + tid] +
                             (double)s_part[3][28 + tid];
            const double mean = a * (double)inv, var = fmax(b * (double)inv - mean * mean, 0.0);
            s_mean[tid] = (float)mean;
            s_rsd[tid] = 1.0f / fmaxf((float)sqrt(var), A.eps);
        }
        __syncthreads();
    }

    // sweep 2: the normalised squared differences
    float mean_c[27], rsd_c[27];
#pragma unroll
    for (int c = 0; c < 27; ++c) {
        mean_c[c] = s_mean[c];
        rsd_c[c] = s_rsd[c];
    }
    const float mean_p = s_mean[27], rsd_p = s_rsd[27];
    float sc[27];
#pragma unroll
    for (int c = 0; c < 27; ++c) sc[c] = 0.f;
    {
        int ux = ux0, uy = uy0, uz = uz0;
        for (int i = tid; i < pv; i += 256) {
            float S[27];
            sample27(win + (uz * Ry + uy) * Rx + ux, Ry, Rx, wz, wy, wx, S);
            const float pn = (post_at(uz, uy, ux) - mean_p) * rsd_p;
#pragma unroll
            for (int c = 0; c < 27; ++c) {
                const float q = (S[c] - mean_c[c]) * rsd_c[c] - pn;
                sc[c] += q * q;
            }
            advance(ux, uy, uz);
        }
    }
#pragma unroll
    for (int c = 0; c < 27; ++c) sc[c] = wave_sum(sc[c]);
    __syncthreads();  // s_part is reused
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 27; ++c) s_part[wv][c] = sc[c];
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0;
        float best_score = 0.f;
        for (int c = 0; c < 27; ++c) {
            const float score = (float)(((double)s_part[0][c] + (double)s_part[1][c] + (double)s_part[2][c] +
                                         (double)s_part[3][c]) * (double)inv);
            if (c == 0 || score < best_score) {  // first minimum (argmin)
                best = c;
                best_score = score;
            }
        }
        out[3 * n + 0] = vx + A.off[best % 3];
        out[3 * n + 1] = vy + A.off[(best / 3) % 3];
        out[3 * n + 2] = vz + A.off[best / 9];
    }
}

int m3d_subpixel_refine(const float *pre, const float *post, const int32_t shape_zyx[3],
                        const int32_t patch_zyx[3], const int32_t step_zyx[3], const int32_t *nodes_zyx,
                        const float *vectors_xyz, int32_t n, const float *offsets_host, int32_t n_offsets,
                        float epsilon, float *out_xyz, void *stream)
{
    M3D_REQUIRE(pre && post && nodes_zyx && vectors_xyz && out_xyz && offsets_host, "null pointer");
    M3D_REQUIRE(i3_positive(shape_zyx) && i3_positive(patch_zyx) && i3_positive(step_zyx) && n >= 0,
                "bad extents");
    M3D_REQUIRE(n_offsets >= 1 && n_offsets <= kMaxOffsets, "1..5 sub-pixel offsets");
    if (n == 0) return M3D_OK;
    RefineArgs A;
    A.shape = to_i3(shape_zyx);
    A.patch = to_i3(patch_zyx);
    A.step = to_i3(step_zyx);
    A.n_off = n_offsets;
    A.eps = epsilon;
    A.off_min = offsets_host[0];
    A.off_max = offsets_host[0];
    for (int i = 0; i < kMaxOffsets; ++i) {
        A.off[i] = i < n_offsets ? offsets_host[i] : 0.f;
        if (i < n_offsets) {
            A.off_min = fminf(A.off_min, offsets_host[i]);
            A.off_max = fmaxf(A.off_max, offsets_host[i]);
        }
    }
    // window extent per axis: floor(b + off_min) .. floor(b + off_max + p - 1) + 1 for any start b
    const int extra = (int)ceilf(A.off_max - A.off_min) + 2;
    A.region = I3{patch_zyx[0] + extra, patch_zyx[1] + extra, patch_zyx[2] + extra};
    const size_t lds_bytes = sizeof(float) * ((size_t)patch_zyx[0] * patch_zyx[1] * patch_zyx[2] +
                                              (size_t)A.region.z * A.region.y * A.region.x);
    if (lds_bytes > 160 * 1024 - 2048) {
        set_err("m3d_subpixel_refine", "patch does not fit the 160 KiB LDS of one CU");
        return M3D_ERR_UNSUPPORTED;
    }
    // three offsets over a three-point window: the one-pass kernel (window only in LDS: two workgroups per CU)
    constexpr bool one_pass_ok = true;
    const size_t win_bytes = sizeof(float) * (size_t)A.region.z * A.region.y * A.region.x;
    if (one_pass_ok && n_offsets == 3 && extra == 3 && 
        offsets_host[0] <= offsets_host[1] && offsets_host[1] <= offsets_host[2]) {
        M3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_subpixel_refine3,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes));
        hipLaunchKernelGGL(k_subpixel_refine3, dim3(n), dim3(256), win_bytes, (hipStream_t)stream, pre, post, A,
                           nodes_zyx, vectors_xyz, out_xyz);
        M3D_KERNEL_CHECK();
        return M3D_OK;
    }
    M3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_subpixel_refine,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k_subpixel_refine, dim3(n), dim3(kBlock), lds_bytes, (hipStream_t)stream, pre, post,
                       A, nodes_zyx, vectors_xyz, out_xyz);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

constexpr size_t kMeshScalarBytes = 256 + 2 * 32 * kMeshSlots;  // states, energy / maxima, two sets of sum records
size_t m3d_mesh_relax_workspace_bytes(const int32_t grid_zyx[3])
{
    if (!i3_positive(grid_zyx)) return 0;
    const size_t n3 = (size_t)3 * grid_zyx[0] * grid_zyx[1] * grid_zyx[2];
    return 3 * n3 * sizeof(float) + kMeshScalarBytes;  // x ping-pong partner, v, a + scalars
}

int m3d_mesh_relax(float *x_xyz, const float *prev_xyz, const int32_t grid_zyx[3],
                   const float stride_xyz[3], const m3d_mesh_params *p, void *work, size_t work_bytes,
                   m3d_mesh_result *result, void *stream)
{
    M3D_REQUIRE(x_xyz && prev_xyz && p && work && result && stride_xyz, "null pointer");
    M3D_REQUIRE(i3_positive(grid_zyx), "grid extents must be positive");
    M3D_REQUIRE(p->num_iters > 0 && p->max_iters > 0 && p->dt > 0.f, "bad integrator parameters");
    M3D_REQUIRE(stride_xyz[0] > 0.f && stride_xyz[1] > 0.f && stride_xyz[2] > 0.f, "stride must be positive");
    if (work_bytes < m3d_mesh_relax_workspace_bytes(grid_zyx)) {
        set_err("m3d_mesh_relax", "work buffer smaller than m3d_mesh_relax_workspace_bytes");
        return M3D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)grid_zyx[0] * grid_zyx[1] * grid_zyx[2], N3 = 3 * N;
    char *scal = (char *)work;  // scalars first: the doubles stay 8-byte aligned
    float *xb = (float *)(scal + kMeshScalarBytes), *v = xb + N3, *a = v + N3;
    FireState *state[2] = {(FireState *)scal, (FireState *)(scal + 16)};
    double *sums[2] = {(double *)(scal + 256), (double *)(scal + 256 + 32 * kMeshSlots)};
    double *ekin = (double *)(scal + 96);
    unsigned int *vmax_bits = (unsigned int *)(scal + 104);
    unsigned int *amax_bits = (unsigned int *)(scal + 108);

    // FIRE grows dt until the stiffest spring goes unstable; with large force caps that throws
    // nodes across their neighbours (a folded mesh is another equilibrium).  Bound dt by half the
    // velocity-Verlet limit 2 / omega_max, omega_max^2 <= k0 + 2 * (18 link constants) (Gershgorin).
    double link_sum = 0.0;
    {
        static const int links[9][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, -1, 0},
                                        {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1}};
        for (int l = 0; l < 9; ++l) {
            double q = 0.0;
            for (int d = 0; d < 3; ++d) q += (double)stride_xyz[d] * links[l][d] * stride_xyz[d] * links[l][d];
            link_sum += (double)p->k * stride_xyz[0] / sqrt(q);
        }
    }
    const float dt_max = fminf(p->dt_max, (float)(1.0 / sqrt((double)p->k0 + 4.0 * link_sum)));

    MeshGeom g;
    g.grid = to_i3(grid_zyx);
    g.stride[0] = stride_xyz[0];
    g.stride[1] = stride_xyz[1];
    g.stride[2] = stride_xyz[2];
    g.k0 = p->k0;
    g.k = p->k;

    M3D_HIP_CHECK(hipMemsetAsync(v, 0, (size_t)N3 * sizeof(float), st));
    M3D_HIP_CHECK(hipMemsetAsync(scal, 0, kMeshScalarBytes, st));
    FireState init = {p->dt, 0.1f, 0, 0};
    M3D_HIP_CHECK(hipMemcpyAsync(state[0], &init, sizeof(init), hipMemcpyHostToDevice, st));
    M3D_HIP_CHECK(hipStreamSynchronize(st));  // `init` is a stack variable

    const int gridN = grid_for(N), grid3 = grid_for(N3);
    float cap = p->start_cap;
    float *xc = x_xyz, *xn = xb;  // current / next positions
    int si = 0;                   // state / sums parity
    // a0 = F(x0); x1 = x0 + dt v0 + dt^2/2 a0
    hipLaunchKernelGGL(k_mesh_force, dim3(gridN), dim3(kBlock), 0, st, xc, prev_xyz, a, v, g, cap, state[0],
                       sums[0], 0);
    hipLaunchKernelGGL(k_mesh_fire, dim3(grid3), dim3(kBlock), 0, st, xc, xn, v, a, N3, state[0], state[0],
                       sums[0], sums[0], dt_max, 0, 0, vmax_bits, amax_bits, ekin);
    {
        float *t = xc;
        xc = xn;
        xn = t;
    }
    int iterations = 0, blocks = 0;
    float vmax = 0.f;
    double e_last = 0.0;
    while (iterations < p->max_iters) {
        M3D_HIP_CHECK(hipMemsetAsync(ekin, 0, 16, st));  // ekin, max|v|, max|F|
        for (int it = 0; it < p->num_iters; ++it) {
            const int last = it == p->num_iters - 1;
            hipLaunchKernelGGL(k_mesh_force, dim3(gridN), dim3(kBlock), 0, st, xc, prev_xyz, a, v, g, cap,
                               state[si], sums[si], 1);
            hipLaunchKernelGGL(k_mesh_fire, dim3(grid3), dim3(kBlock), 0, st, xc, xn, v, a, N3, state[si],
                               state[si ^ 1], sums[si], sums[si ^ 1], dt_max, 1, last, vmax_bits, amax_bits, ekin);
            si ^= 1;
            if (!last) {
                float *t = xc;
                xc = xn;
                xn = t;
            }
        }
        M3D_KERNEL_CHECK();
        iterations += p->num_iters;
        ++blocks;
        struct {
            double e;
            unsigned int vb, ab;
        } rb;
        M3D_HIP_CHECK(hipMemcpyAsync(&rb, ekin, 16, hipMemcpyDeviceToHost, st));
        M3D_HIP_CHECK(hipStreamSynchronize(st));
        float amax;
        memcpy(&vmax, &rb.vb, sizeof(float));
        memcpy(&amax, &rb.ab, sizeof(float));
        e_last = rb.e;
        bool stop = false;
        // a FIRE reset zeroes v for one step, so the residual force has to be small as well
        if (fmaxf(vmax, amax) < p->stop_v_max) {
            if (cap >= p->final_cap)
                stop = true;
            else
                cap = fminf(cap * p->cap_upscale, p->final_cap);
        }
        if (stop || iterations >= p->max_iters) break;
        // continue: the advanced positions become current
        float *t = xc;
        xc = xn;
        xn = t;
    }
    if (xc != x_xyz)
        M3D_HIP_CHECK(hipMemcpyAsync(x_xyz, xc, (size_t)N3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    FireState fin;
    M3D_HIP_CHECK(hipMemcpyAsync(&fin, state[si], sizeof(fin), hipMemcpyDeviceToHost, st));
    M3D_HIP_CHECK(hipStreamSynchronize(st));
    result->iterations = iterations;
    result->blocks = blocks;
    result->final_kinetic_energy = e_last;
    result->final_v_max = vmax;
    result->final_cap = cap;
    result->final_dt = fin.dt;
    return M3D_OK;
}

int m3d_flow_compose(const float *base, const float *residual, const int32_t grid_zyx[3],
                     const float stride_zyx[3], const float box_start_xyz[3], float *out, void *stream)
{
    M3D_REQUIRE(base && residual && out && stride_zyx && box_start_xyz && i3_positive(grid_zyx),
                "bad arguments");
    M3D_REQUIRE(base != out, "compose cannot run in place on the base field");
    const int64_t N = (int64_t)grid_zyx[0] * grid_zyx[1] * grid_zyx[2];
    hipLaunchKernelGGL(k_flow_compose, dim3(grid_for(N)), dim3(kBlock), 0, (hipStream_t)stream, base,
                       residual, to_i3(grid_zyx), stride_zyx[0], stride_zyx[1], stride_zyx[2],
                       box_start_xyz[2], box_start_xyz[1], box_start_xyz[0], out);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_moments(const float *src, int64_t n, double *out2, void *stream)
{
    M3D_REQUIRE(src && out2 && n >= 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    M3D_HIP_CHECK(hipMemsetAsync(out2, 0, 2 * sizeof(double), st));
    if (n == 0) return M3D_OK;
    hipLaunchKernelGGL(k_moments, dim3(grid_for(n, 8)), dim3(kBlock), 0, st, src, n, out2);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_normalized_sq_diff(const float *a, const float *b, int64_t n, float mean_a, float scale_a,
                           float mean_b, float scale_b, double *out1, void *stream)
{
    M3D_REQUIRE(a && b && out1 && n >= 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    M3D_HIP_CHECK(hipMemsetAsync(out1, 0, sizeof(double), st));
    if (n == 0) return M3D_OK;
    hipLaunchKernelGGL(k_norm_sq_diff, dim3(grid_for(n, 8)), dim3(kBlock), 0, st, a, b, n, mean_a, scale_a,
                       mean_b, scale_b, out1);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

}  // extern "C"
