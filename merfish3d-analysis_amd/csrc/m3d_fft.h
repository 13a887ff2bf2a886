// In-LDS mixed-radix FFT building blocks for the fused RLGC convolution (gfx950).
//
// Why not rocFFT: for the headline volume 128 x 2187 x 2187 rocFFT expands the odd-length
// real transform to full complex data and runs 8 full passes (copy, 3 FFT kernels, 3
// transposes, hermitian copy) per direction - ~78 GB of HBM traffic per transform for a
// 2.45-GB real volume (profiles/r01_rocfft_kernel_stats.csv).  The convolution engine built
// from these blocks moves 44 B per voxel per convolution:
//
//   X pass  two real rows are packed as one complex row (a + i b), ONE complex FFT of the
//           odd length serves both, the Hermitian halves are separated on the way out;
//   Y pass  works on a [z][kx-tile][y][4] tiled spectrum so each column tile is one
//           contiguous 70-KB run (no transposes, fully coalesced);
//   Z pass  forward FFT, OTF multiply and inverse FFT fused in one kernel.
//
// Transforms are in place in LDS: decimation-in-frequency forward (natural in ->
// digit-reversed out) and decimation-in-time inverse (digit-reversed in -> natural out).
// The permuted order never has to be undone: the spectrum is only ever multiplied by an
// OTF that went through the same forward pipeline.  One barrier per stage, one buffer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace m3d {

constexpr int kMaxStages = 16;

struct AxisPlan {
    int n;                 // transform length (2^a 3^b)
    int nstages;
    int radix[kMaxStages];  // DIF order: first stage splits the full length
    const float2 *tw;       // device: exp(-2 pi i k / n), k in [0, n)
    const int *pos;         // device: position of frequency k after the DIF pass
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)  // a * conj(b)
{
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot90(float2 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <bool INV>
__device__ __forceinline__ void dft2(float2 *v)
{
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <bool INV>
__device__ __forceinline__ void dft3(float2 &v0, float2 &v1, float2 &v2)
{
    const float s = 0.86602540378443864676f;
    const float2 t1 = cadd(v1, v2);
    const float2 t2 = make_float2(v0.x - 0.5f * t1.x, v0.y - 0.5f * t1.y);
    const float2 d = csub(v1, v2);
    const float2 t3 = rot90<INV>(make_float2(s * d.x, s * d.y));
    v0 = cadd(v0, t1);
    v1 = cadd(t2, t3);
    v2 = csub(t2, t3);
}
template <bool INV>
__device__ __forceinline__ void dft4(float2 *v)
{
    const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
    const float2 c = cadd(v[1], v[3]), d = rot90<INV>(csub(v[1], v[3]));
    v[0] = cadd(a, c);
    v[2] = csub(a, c);
    v[1] = cadd(b, d);
    v[3] = csub(b, d);
}
template <bool INV>
__device__ __forceinline__ void dft9(float2 *v)
{
    // x[3a + b] -> X[c + 3d]
    float2 u[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        u[b][0] = v[b];
        u[b][1] = v[3 + b];
        u[b][2] = v[6 + b];
        dft3<INV>(u[b][0], u[b][1], u[b][2]);
    }
    // twiddles W9^(b*c), W9 = exp(-+ 2 pi i / 9)
    const float c1 = 0.76604444311897803520f, s1 = 0.64278760968653932632f;
    const float c2 = 0.17364817766693034885f, s2 = 0.98480775301220805937f;
    const float c4 = -0.93969262078590838405f, s4 = 0.34202014332566873304f;
    const float sg = INV ? 1.0f : -1.0f;
    u[1][1] = cmul(u[1][1], make_float2(c1, sg * s1));
    u[1][2] = cmul(u[1][2], make_float2(c2, sg * s2));
    u[2][1] = cmul(u[2][1], make_float2(c2, sg * s2));
    u[2][2] = cmul(u[2][2], make_float2(c4, sg * s4));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float2 a0 = u[0][c], a1 = u[1][c], a2 = u[2][c];
        dft3<INV>(a0, a1, a2);
        v[c] = a0;
        v[c + 3] = a1;
        v[c + 6] = a2;
    }
}

template <int R, bool INV>
__device__ __forceinline__ void dft_r(float2 *v)
{
    if constexpr (R == 2) dft2<INV>(v);
    if constexpr (R == 3) dft3<INV>(v[0], v[1], v[2]);
    if constexpr (R == 4) dft4<INV>(v);
    if constexpr (R == 9) dft9<INV>(v);
}

// One in-place stage over `batch` interleaved transforms of length n held in LDS.
// element k of transform b lives at data[k * estride + b * bstride]; M = current block size.
template <int R, bool INV>
__device__ __forceinline__ void lds_stage(float2 *data, int n, int M, int estride, int batch,
                                          int bstride, const float2 *__restrict__ tw, int tid,
                                          int nthreads)
{
    const int sub = M / R;        // butterflies per block
    const int per_fft = n / R;    // butterflies per transform
    const int total = per_fft * batch;
    const int twstep = n / M;
    for (int i = tid; i < total; i += nthreads) {
        int b, t;
        if (bstride == 1) {  // batch-fastest: consecutive lanes touch consecutive LDS words
            b = i % batch;
            t = i / batch;
        } else {
            t = i % per_fft;
            b = i / per_fft;
        }
        const int blk = t / sub, j = t - blk * sub;
        float2 *p = data + (size_t)(blk * M + j) * estride + (size_t)b * bstride;
        const int step = sub * estride;
        float2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = p[q * step];
        if (INV) {
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul_conj(v[q], tw[j * q * twstep]);
            dft_r<R, true>(v);
        } else {
            dft_r<R, false>(v);
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul(v[q], tw[j * q * twstep]);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) p[q * step] = v[q];
    }
}

// Full transform: DIF forward (natural -> permuted) or DIT inverse (permuted -> natural,
// unnormalised).  Every thread of the block must call it; ends with a barrier.
template <bool INV>
__device__ __forceinline__ void lds_fft(float2 *data, const AxisPlan &ap, int estride, int batch,
                                        int bstride, int tid, int nthreads)
{
    const int n = ap.n;
    if (!INV) {
        int M = n;
        for (int s = 0; s < ap.nstages; ++s) {
            const int r = ap.radix[s];
            if (r == 9) lds_stage<9, false>(data, n, M, estride, batch, bstride, ap.tw, tid, nthreads);
            else if (r == 4) lds_stage<4, false>(data, n, M, estride, batch, bstride, ap.tw, tid, nthreads);
            else if (r == 3) lds_stage<3, false>(data, n, M, estride, batch, bstride, ap.tw, tid, nthreads);
            else lds_stage<2, false>(data, n, M, estride, batch, bstride, ap.tw, tid, nthreads);
            M /= r;
            __syncthreads();
        }
    } else {
        int M = 1;
        for (int s = ap.nstages - 1; s >= 0; --s) {
            const int r = ap.radix[s];
            M *= r;
            if (r == 9) lds_stage<9, true>(data, n, M, estride, batch, bstride, ap.tw, tid, nthreads);
            else if (r == 4) lds_stage<4, true>(data, n, M, estride, batch, bstride, ap.tw, tid, nthreads);
            else if (r == 3) lds_stage<3, true>(data, n, M, estride, batch, bstride, ap.tw, tid, nthreads);
            else lds_stage<2, true>(data, n, M, estride, batch, bstride, ap.tw, tid, nthreads);
            __syncthreads();
        }
    }
}

}  // namespace m3d
