// Fused 3-D FFT convolution for RLGC on MI355X: 5 kernels, 44 B of HBM traffic per voxel.
//
//   k_x_forward   two real rows -> one packed complex FFT (in LDS) -> two Hermitian half rows,
//                 written into the tiled spectrum  S[z][kx/TW][y][kx%TW]      (r 4 B, w 4 B)
//   k_y_pass      FFT along y of one contiguous [Yp][TW] column tile, in place  (r 4 B, w 4 B)
//   k_z_conv      FFT along z, multiply by the OTF (or conj / |.|^2), inverse FFT along z,
//                 in place; P adjacent spectrum positions per block             (r 8 B, w 4 B)
//   k_y_pass      inverse                                                        (r 4 B, w 4 B)
//   k_x_inverse   two Hermitian half rows -> packed complex inverse FFT -> two real rows
//                                                                                (r 4 B, w 4 B)
// The spectrum keeps the digit-reversed order of the decimation-in-frequency passes along
// y and z (see m3d_fft.h) - only the OTF, which went through the same passes, ever meets it.
// Replaces cp.fft.rfftn / "*= H" / cp.fft.irfftn of fft_conv (utils/rlgc.py:316-350).

#include "m3d_fftconv.h"

#include <cmath>
#include <vector>

#include "m3d_common.h"
#include "m3d_fft.h"

using namespace m3d;

namespace m3d {

struct FftConvPlan {
    int Zp, Yp, Xp, Xc;
    int TW, nT;          // kx tile width / number of tiles
    int RB;              // real rows per X-pass block (even)
    int P;               // spectrum positions per Z-pass block
    int64_t plane;       // complex elements per z plane of the tiled spectrum = nT*Yp*TW
    AxisPlan ax, ay, az;
    void *tables = nullptr;  // one device allocation holding all twiddle / pos tables
    size_t lds_x, lds_y, lds_z;
};

}  // namespace m3d

namespace {

struct ConvGeo {
    int Zp, Yp, Xp, Xc, TW, nT, RB, P;
    int64_t plane;
};

__device__ __forceinline__ int64_t spec_index(const ConvGeo &g, int z, int y, int k)
{
    return (((int64_t)z * g.nT + k / g.TW) * g.Yp + y) * g.TW + (k % g.TW);
}

// ---------------------------------------------------------------- X forward (R2C)
__global__ void k_x_forward(const float *__restrict__ src, float2 *__restrict__ spec, ConvGeo g,
                            AxisPlan ax)
{
    extern __shared__ float2 lds_fx[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int groups = (g.Yp + g.RB - 1) / g.RB;
    const int z = blockIdx.x / groups, y0 = (blockIdx.x % groups) * g.RB;
    const int nb = g.RB / 2, N = g.Xp;
    for (int i = tid; i < nb * N; i += nt) {
        const int bi = i / N, n = i - bi * N;
        const int ya = y0 + 2 * bi, yb = ya + 1;
        const float a = ya < g.Yp ? src[((int64_t)z * g.Yp + ya) * N + n] : 0.0f;
        const float b = yb < g.Yp ? src[((int64_t)z * g.Yp + yb) * N + n] : 0.0f;
        lds_fx[i] = make_float2(a, b);
    }
    __syncthreads();
    lds_fft<false>(lds_fx, ax, 1, nb, N, tid, nt);
    // Hermitian split: A[k] = (C[k] + conj(C[N-k])) / 2, B[k] = (C[k] - conj(C[N-k])) / (2i)
    const int total = g.nT * nb * g.TW;
    for (int e = tid; e < total; e += nt) {
        const int kk = e % g.TW, bi = (e / g.TW) % nb, kt = e / (g.TW * nb);
        const int k = kt * g.TW + kk;
        if (k >= g.Xc) continue;
        const float2 c1 = lds_fx[bi * N + ax.pos[k]];
        const float2 c2 = lds_fx[bi * N + ax.pos[k == 0 ? 0 : N - k]];
        const float2 A = make_float2(0.5f * (c1.x + c2.x), 0.5f * (c1.y - c2.y));
        const float2 B = make_float2(0.5f * (c1.y + c2.y), 0.5f * (c2.x - c1.x));
        const int ya = y0 + 2 * bi;
        if (ya < g.Yp) spec[spec_index(g, z, ya, k)] = A;
        if (ya + 1 < g.Yp) spec[spec_index(g, z, ya + 1, k)] = B;
    }
}

// ---------------------------------------------------------------- X inverse (C2R)
__global__ void k_x_inverse(const float2 *__restrict__ spec, float *__restrict__ dst, ConvGeo g,
                            AxisPlan ax)
{
    extern __shared__ float2 lds_ix[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int groups = (g.Yp + g.RB - 1) / g.RB;
    const int z = blockIdx.x / groups, y0 = (blockIdx.x % groups) * g.RB;
    const int nb = g.RB / 2, N = g.Xp;
    const int total = g.nT * nb * g.TW;
    for (int e = tid; e < total; e += nt) {
        const int kk = e % g.TW, bi = (e / g.TW) % nb, kt = e / (g.TW * nb);
        const int k = kt * g.TW + kk;
        if (k >= g.Xc) continue;
        const int ya = y0 + 2 * bi;
        const float2 A = ya < g.Yp ? spec[spec_index(g, z, ya, k)] : make_float2(0.f, 0.f);
        const float2 B = ya + 1 < g.Yp ? spec[spec_index(g, z, ya + 1, k)] : make_float2(0.f, 0.f);
        // C[k] = A + iB ; C[N-k] = conj(A) + i conj(B)
        lds_ix[bi * N + ax.pos[k]] = make_float2(A.x - B.y, A.y + B.x);
        if (k != 0 && 2 * k != N) lds_ix[bi * N + ax.pos[N - k]] = make_float2(A.x + B.y, B.x - A.y);
    }
    __syncthreads();
    lds_fft<true>(lds_ix, ax, 1, nb, N, tid, nt);
    for (int i = tid; i < nb * N; i += nt) {
        const int bi = i / N, n = i - bi * N;
        const int ya = y0 + 2 * bi, yb = ya + 1;
        const float2 v = lds_ix[i];
        if (ya < g.Yp) dst[((int64_t)z * g.Yp + ya) * N + n] = v.x;
        if (yb < g.Yp) dst[((int64_t)z * g.Yp + yb) * N + n] = v.y;
    }
}

// ---------------------------------------------------------------- Y pass (in place)
template <bool INV>
__global__ void k_y_pass(float2 *__restrict__ spec, ConvGeo g, AxisPlan ay)
{
    extern __shared__ float2 lds_y[];
    const int tid = threadIdx.x, nt = blockDim.x;
    float2 *tile = spec + (int64_t)blockIdx.x * g.Yp * g.TW;  // (z, kt) tiles are consecutive
    const int n = g.Yp * g.TW;
    for (int i = tid; i < n; i += nt) lds_y[i] = tile[i];
    __syncthreads();
    lds_fft<INV>(lds_y, ay, g.TW, g.TW, 1, tid, nt);
    for (int i = tid; i < n; i += nt) tile[i] = lds_y[i];
}

// ---------------------------------------------------------------- Z pass
// mode 0: S*H  1: S*conj(H)  2: S*|H|^2  (then * scale), -1: forward only (OTF construction)
__global__ void k_z_conv(float2 *__restrict__ spec, const float2 *__restrict__ otf, ConvGeo g,
                         AxisPlan az, int mode, float scale)
{
    extern __shared__ float2 lds_z[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * g.P;
    const int P = g.P, n = g.Zp * P;
    for (int i = tid; i < n; i += nt) {
        const int z = i / P, p = i - z * P;
        lds_z[i] = (p0 + p < g.plane) ? spec[(int64_t)z * g.plane + p0 + p] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    lds_fft<false>(lds_z, az, P, P, 1, tid, nt);
    if (mode < 0) {
        for (int i = tid; i < n; i += nt) {
            const int z = i / P, p = i - z * P;
            if (p0 + p < g.plane) spec[(int64_t)z * g.plane + p0 + p] = lds_z[i];
        }
        return;
    }
    for (int i = tid; i < n; i += nt) {
        const int z = i / P, p = i - z * P;
        if (p0 + p >= g.plane) continue;
        float2 h = otf[(int64_t)z * g.plane + p0 + p];
        if (mode == 1) h.y = -h.y;
        if (mode == 2) h = make_float2(h.x * h.x + h.y * h.y, 0.0f);
        const float2 v = cmul(lds_z[i], h);
        lds_z[i] = make_float2(v.x * scale, v.y * scale);
    }
    __syncthreads();
    lds_fft<true>(lds_z, az, P, P, 1, tid, nt);
    for (int i = tid; i < n; i += nt) {
        const int z = i / P, p = i - z * P;
        if (p0 + p < g.plane) spec[(int64_t)z * g.plane + p0 + p] = lds_z[i];
    }
}

bool factor_23(int n, int *radix, int *nstages)
{
    int c3 = 0, c2 = 0, m = n;
    while (m % 3 == 0) {
        m /= 3;
        ++c3;
    }
    while (m % 2 == 0) {
        m /= 2;
        ++c2;
    }
    if (m != 1) return false;
    int s = 0;
    for (int i = 0; i < c3 / 2; ++i) radix[s++] = 9;
    if (c3 % 2) radix[s++] = 3;
    for (int i = 0; i < c2 / 2; ++i) radix[s++] = 4;
    if (c2 % 2) radix[s++] = 2;
    *nstages = s;
    return s <= kMaxStages;
}

ConvGeo geo_of(const FftConvPlan *p)
{
    ConvGeo g;
    g.Zp = p->Zp;
    g.Yp = p->Yp;
    g.Xp = p->Xp;
    g.Xc = p->Xc;
    g.TW = p->TW;
    g.nT = p->nT;
    g.RB = p->RB;
    g.P = p->P;
    g.plane = p->plane;
    return g;
}

}  // namespace

namespace m3d {

int fftconv_create(int Zp, int Yp, int Xp, FftConvPlan **out)
{
    *out = nullptr;
    FftConvPlan *p = new FftConvPlan();
    p->Zp = Zp;
    p->Yp = Yp;
    p->Xp = Xp;
    p->Xc = Xp / 2 + 1;
    if (!factor_23(Xp, p->ax.radix, &p->ax.nstages) || !factor_23(Yp, p->ay.radix, &p->ay.nstages) ||
        !factor_23(Zp, p->az.radix, &p->az.nstages)) {
        delete p;
        return M3D_ERR_UNSUPPORTED;
    }
    p->ax.n = Xp;
    p->ay.n = Yp;
    p->az.n = Zp;
    // tile width: widest of 4/2/1 whose column tile fits a 72-KB LDS buffer (2 blocks / CU)
    p->TW = 4;
    while (p->TW > 1 && (size_t)Yp * p->TW * sizeof(float2) > 72 * 1024) p->TW /= 2;
    p->lds_y = (size_t)Yp * p->TW * sizeof(float2);
    // rows per X block: 4 real rows (2 packed) when they fit 40 KB, else 2
    p->RB = ((size_t)2 * Xp * sizeof(float2) <= 40 * 1024) ? 4 : 2;
    p->lds_x = (size_t)(p->RB / 2) * Xp * sizeof(float2);
    p->P = 32;
    while (p->P > 1 && (size_t)Zp * p->P * sizeof(float2) > 64 * 1024) p->P /= 2;
    p->lds_z = (size_t)Zp * p->P * sizeof(float2);
    if (p->lds_y > 150 * 1024 || p->lds_x > 150 * 1024 || p->lds_z > 150 * 1024) {
        delete p;
        return M3D_ERR_UNSUPPORTED;
    }
    p->nT = (p->Xc + p->TW - 1) / p->TW;
    p->plane = (int64_t)p->nT * Yp * p->TW;

    // tables: twiddles for x, y, z + pos for x
    const size_t nb = sizeof(float2) * ((size_t)Xp + Yp + Zp) + sizeof(int) * (size_t)Xp;
    std::vector<char> host(nb);
    float2 *twx = (float2 *)host.data(), *twy = twx + Xp, *twz = twy + Yp;
    int *posx = (int *)(twz + Zp);
    auto fill_tw = [](float2 *t, int n) {
        for (int k = 0; k < n; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n;
            t[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    };
    fill_tw(twx, Xp);
    fill_tw(twy, Yp);
    fill_tw(twz, Zp);
    for (int k = 0; k < Xp; ++k) {
        int rem = k, pos = 0, M = Xp;
        for (int s = 0; s < p->ax.nstages; ++s) {
            const int r = p->ax.radix[s];
            const int d = rem % r;
            rem /= r;
            M /= r;
            pos += d * M;
        }
        posx[k] = pos;
    }
    if (hipMalloc(&p->tables, nb) != hipSuccess) {
        delete p;
        return M3D_ERR_OUT_OF_MEMORY;
    }
    if (hipMemcpy(p->tables, host.data(), nb, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->tables);
        delete p;
        return M3D_ERR_HIP;
    }
    float2 *d = (float2 *)p->tables;
    p->ax.tw = d;
    p->ay.tw = d + Xp;
    p->az.tw = d + Xp + Yp;
    p->ax.pos = (const int *)(d + Xp + Yp + Zp);
    p->ay.pos = nullptr;
    p->az.pos = nullptr;
    // allow > 64 KB of dynamic LDS where needed
    (void)hipFuncSetAttribute((const void *)k_x_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_x);
    (void)hipFuncSetAttribute((const void *)k_x_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_x);
    (void)hipFuncSetAttribute((const void *)k_y_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_y);
    (void)hipFuncSetAttribute((const void *)k_y_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_y);
    (void)hipFuncSetAttribute((const void *)k_z_conv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_z);
    *out = p;
    return M3D_OK;
}

void fftconv_destroy(FftConvPlan *p)
{
    if (!p) return;
    if (p->tables) (void)hipFree(p->tables);
    delete p;
}

int64_t fftconv_spectrum_elems(const FftConvPlan *p) { return (int64_t)p->Zp * p->plane; }

static int launch_forward_xy(const FftConvPlan *p, const float *src, float2 *spec, hipStream_t st,
                             EventLog *log = nullptr)
{
    const ConvGeo g = geo_of(p);
    const int groups = (p->Yp + p->RB - 1) / p->RB;
    if (log && log->record()) return M3D_ERR_HIP;
    hipLaunchKernelGGL(k_x_forward, dim3((unsigned)(p->Zp * groups)), dim3(256), p->lds_x, st, src, spec, g, p->ax);
    if (log && log->record()) return M3D_ERR_HIP;
    hipLaunchKernelGGL(k_y_pass<false>, dim3((unsigned)(p->Zp * p->nT)), dim3(512), p->lds_y, st, spec, g, p->ay);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int fftconv_make_otf(const FftConvPlan *p, const float *psf_padded, float2 *otf, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_forward_xy(p, psf_padded, otf, st);
    if (rc) return rc;
    const ConvGeo g = geo_of(p);
    const unsigned nblk = (unsigned)((p->plane + p->P - 1) / p->P);
    hipLaunchKernelGGL(k_z_conv, dim3(nblk), dim3(256), p->lds_z, st, otf, (const float2 *)nullptr, g, p->az, -1, 1.0f);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int fftconv_conv(const FftConvPlan *p, const float *src, const float2 *otf, int mode, float2 *spec,
                 float *dst, void *stream, EventLog *log)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_forward_xy(p, src, spec, st, log);
    if (rc) return rc;
    if (log && log->record()) return M3D_ERR_HIP;
    const ConvGeo g = geo_of(p);
    const float scale = 1.0f / ((float)p->Zp * (float)p->Yp * (float)p->Xp);
    const unsigned nblk = (unsigned)((p->plane + p->P - 1) / p->P);
    hipLaunchKernelGGL(k_z_conv, dim3(nblk), dim3(256), p->lds_z, st, spec, otf, g, p->az, mode, scale);
    if (log && log->record()) return M3D_ERR_HIP;
    hipLaunchKernelGGL(k_y_pass<true>, dim3((unsigned)(p->Zp * p->nT)), dim3(512), p->lds_y, st, spec, g, p->ay);
    if (log && log->record()) return M3D_ERR_HIP;
    const int groups = (p->Yp + p->RB - 1) / p->RB;
    hipLaunchKernelGGL(k_x_inverse, dim3((unsigned)(p->Zp * groups)), dim3(256), p->lds_x, st, spec, dst, g, p->ax);
    if (log && log->record()) return M3D_ERR_HIP;
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

}  // namespace m3d
