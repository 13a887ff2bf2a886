// PixelDecoder device stages for MI355X / gfx950 (PixelDecoder.py of the reference).
//
//   m3d_weight_readout    float32(readout) * float32(predictor)        (:1879-1881)
//   m3d_gaussian_lowpass  cupyx gaussian_filter, reflect, truncate 4   (:1948-1980)
//   m3d_decode_pixels     scale -> clip -> L2 normalise -> distance to every codeword ->
//                         argmin -> two-threshold call -> f16/i16 result volumes, one pass
//                         (:2354-2643); the reference runs ~12 CuPy kernels + a cuVS
//                         pairwise_distance that materialises a [pixels, C] matrix per plane.
//
// Layout: the bit stack is f32 [B, N] bit-major (N = Z*Y*X), so lane l of a wave reads
// voxel p0+l of every bit plane: 256-B coalesced segments per bit.  The distance step is the
// only dense contraction on the hot path and runs on the matrix cores with exact-f32
// v_mfma_f32_32x32x2_f32 (no bf16/xf32: decoded indices must be bit-exact).  Codewords are
// the A operand (rows), pixels the B operand (columns), so each lane ends up holding the
// distances of ONE pixel to 16 codewords per tile and the argmin needs no cross-lane
// traffic except one half-wave swap.

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "m3d_common.h"
#include "m3d_index.h"

using namespace m3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------
__global__ void k_weight_readout(const uint16_t *__restrict__ readout,
                                 const float *__restrict__ predictor, float *__restrict__ out,
                                 int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)readout[i] * predictor[i];
}

// ---------------------------------------------------------------------------------------
// Gaussian low-pass.  scipy/cupyx "reflect" = half-sample symmetric: (d c b a | a b c d | d c b a)
constexpr int kMaxRadius = 32;
struct GaussW {
    int radius;
    int offset;  // integer shift added to every tap index (0 for the Gaussian)
    int wrap;    // 0: scipy "reflect", 1: scipy "grid-wrap" (periodic)
    float w[2 * kMaxRadius + 1];
};

__device__ __forceinline__ int tap_idx(int i, int n, int wrap)
{
    if (!wrap) return reflect_idx(i, n);
    int m = i % n;
    return m < 0 ? m + n : m;
}

// Walk one line (length len, element stride `stride`) per thread with a register window:
// every input is read once, every output written once; lanes are adjacent lines, so all
// accesses are coalesced.  Used for the Z and Y axes.
template <int R>
__global__ void k_gauss_walk(const float *__restrict__ in, float *__restrict__ out, int64_t n_lines,
                             int64_t inner, int64_t outer_stride, int len, int64_t stride,
                             GaussW gw)
{
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_lines;
         c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = (c / inner) * outer_stride + (c % inner);
        float win[2 * R + 1];
#pragma unroll
        for (int k = 0; k < 2 * R; ++k)
            win[k] = in[base + (int64_t)tap_idx(k - R + gw.offset, len, gw.wrap) * stride];
        for (int t = 0; t < len; ++t) {
            win[2 * R] = in[base + (int64_t)tap_idx(t + R + gw.offset, len, gw.wrap) * stride];
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k <= 2 * R; ++k) acc = rn_fma(gw.w[k], win[k], acc);
            out[base + (int64_t)t * stride] = acc;
#pragma unroll
            for (int k = 0; k < 2 * R; ++k) win[k] = win[k + 1];
        }
    }
}

// Generic-radius fallback for the strided axes (taps re-read through L1/L2).
__global__ void k_gauss_direct(const float *__restrict__ in, float *__restrict__ out,
                               int64_t n_lines, int64_t inner, int64_t outer_stride, int len,
                               int64_t stride, GaussW gw)
{
    const int R = gw.radius;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_lines;
         c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = (c / inner) * outer_stride + (c % inner);
        for (int t = 0; t < len; ++t) {
            float acc = 0.0f;
            for (int k = 0; k <= 2 * R; ++k)
                acc = rn_fma(gw.w[k], in[base + (int64_t)tap_idx(t + k - R + gw.offset, len, gw.wrap) * stride], acc);
            out[base + (int64_t)t * stride] = acc;
        }
    }
}

// X axis (contiguous): one block per row, row staged in LDS, so in-place is safe.
__global__ void k_gauss_x(const float *__restrict__ in, float *__restrict__ out, int64_t rows, int X,
                          GaussW gw)
{
    extern __shared__ float row_lds[];
    const int R = gw.radius;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float *src = in + r * X;
        float *dst = out + r * X;
        __syncthreads();
        for (int x = threadIdx.x; x < X; x += blockDim.x) row_lds[x] = src[x];
        __syncthreads();
        for (int x = threadIdx.x; x < X; x += blockDim.x) {
            float acc = 0.0f;
            for (int k = 0; k <= 2 * R; ++k)
                acc = rn_fma(gw.w[k], row_lds[tap_idx(x + k - R + gw.offset, X, gw.wrap)], acc);
            dst[x] = acc;
        }
    }
}

GaussW make_weights(float sigma)
{
    // scipy _gaussian_kernel1d: radius = int(truncate * sigma + 0.5), float64 weights
    GaussW gw;
    gw.offset = 0;
    gw.wrap = 0;
    const double sd = (double)sigma;
    int radius = (int)(4.0 * sd + 0.5);
    gw.radius = radius;
    if (radius > kMaxRadius) return gw;
    double w[2 * kMaxRadius + 1], sum = 0.0;
    for (int k = -radius; k <= radius; ++k) {
        w[k + radius] = std::exp(-0.5 / (sd * sd) * (double)k * (double)k);
        sum += w[k + radius];
    }
    for (int k = 0; k <= 2 * radius; ++k) gw.w[k] = (float)(w[k] / sum);
    for (int k = 2 * radius + 1; k < 2 * kMaxRadius + 1; ++k) gw.w[k] = 0.0f;
    return gw;
}

template <int R>
void launch_walk(const float *in, float *out, int64_t n_lines, int64_t inner, int64_t outer_stride,
                 int len, int64_t stride, const GaussW &gw, hipStream_t st)
{
    hipLaunchKernelGGL(k_gauss_walk<R>, dim3(grid_for(n_lines)), dim3(kBlock), 0, st, in, out,
                       n_lines, inner, outer_stride, len, stride, gw);
}

void gauss_strided(const float *in, float *out, int64_t n_lines, int64_t inner,
                   int64_t outer_stride, int len, int64_t stride, const GaussW &gw, hipStream_t st)
{
    switch (gw.radius) {
#define M3D_CASE(R)                                                                    \
    case R:                                                                            \
        launch_walk<R>(in, out, n_lines, inner, outer_stride, len, stride, gw, st);    \
        break;
        M3D_CASE(1) M3D_CASE(2) M3D_CASE(3) M3D_CASE(4) M3D_CASE(5) M3D_CASE(6) M3D_CASE(7) M3D_CASE(8)
        M3D_CASE(9) M3D_CASE(10) M3D_CASE(11) M3D_CASE(12) M3D_CASE(13) M3D_CASE(14) M3D_CASE(15) M3D_CASE(16)
#undef M3D_CASE
        default:
            hipLaunchKernelGGL(k_gauss_direct, dim3(grid_for(n_lines)), dim3(kBlock), 0, st, in,
                               out, n_lines, inner, outer_stride, len, stride, gw);
    }
}

// ---------------------------------------------------------------------------------------
// cp.round(x, 5) of a float32 (CuPy evaluates in double), then the float16 store
// (PixelDecoder.py:2621-2632 into the float16 arrays of :2540-2543).
// The product v * 1e5 is exact in double (24 x 17 significant bits) and rint() of it is an integer q; for |q| < 2^24
// the quotient is taken in float32: float32(double(q) / 1e5) == float32(q) / 100000.0f, because rounding a quotient
// first to 53 bits and then to 24 is innocuous (53 >= 2 * 24 + 2) - one correctly rounded float32 division instead
// of a double-precision one, same bits.
__device__ __forceinline__ uint16_t round5_f16(float v)
{
    const double q = rint((double)v * 1e5);
    if (fabs(q) < 16777216.0) return f32_to_f16_bits(rn_div((float)q, 100000.0f));
    return f32_to_f16_bits((float)(q / 1e5));
}

struct DecodeArgs {
    const float *stack;
    const float *background;
    const float *normalization;
    const float *codebook;
    const int32_t *excluded;
    int16_t *decoded;
    uint16_t *magnitude_f16;
    uint16_t *distance_f16;
    uint16_t *scaled_f16;
    float *magnitude_f32;
    float *distance_f32;
    int64_t N;
    int32_t B, C, n_excluded, NT, prenormalized;
    float thr, lo, hi;
};

// KS = k-steps of 2 bits each (B <= 2*KS).  Block = 4 waves; a wave owns 64 pixels per trip:
// lane l loads/normalises/stores pixel p0+l, the two 32-pixel MFMA column tiles fetch their
// operands from the owner lanes with ds_bpermute.
template <int KS>
__global__ __launch_bounds__(256) void k_decode(DecodeArgs a)
{
    extern __shared__ float lds[];
    const int NT = a.NT;
    float *afrag = lds;                      // [NT][KS][64]
    float *yn = afrag + (size_t)NT * KS * 64;  // [NT*32] codeword squared norms, in epilogue order (see ynq)
    int *excl = (int *)(yn + NT * 32);       // [NT*32] 1 if excluded
    const float4 *ynq = (const float4 *)yn;  // [NT][2 halves][4 row groups] x 4 rows
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int half = lane >> 5;
    const int col = lane & 31;

    // ---- stage codebook fragments: afrag[t][s][l] = cb[32t + (l&31)][2s + (l>>5)]
    for (int i = tid; i < NT * KS * 64; i += blockDim.x) {
        const int l = i & 63, s = (i >> 6) % KS, t = (i >> 6) / KS;
        const int c = 32 * t + (l & 31), k = 2 * s + (l >> 5);
        afrag[i] = (c < a.C && k < a.B) ? a.codebook[(size_t)c * a.B + k] : 0.0f;
    }
    for (int c = tid; c < NT * 32; c += blockDim.x) {
        float acc = 0.0f;
        if (c < a.C)
            for (int b = 0; b < a.B; ++b) {
                const float v = a.codebook[(size_t)c * a.B + b];
                acc = rn_fma(v, v, acc);
            }
        // slot of codeword c = 32 t + j + 8 g + 4 h  ->  ((t * 2 + h) * 4 + g) * 4 + j
        {
            const int t = c >> 5, w = c & 31, j = w & 3, h = (w >> 2) & 1, gq = w >> 3;
            yn[(((t * 2 + h) * 4 + gq) << 2) + j] = c < a.C ? acc : INFINITY;
        }
        int e = 0;
        for (int j = 0; j < a.n_excluded; ++j) e |= (a.excluded[j] == c);
        excl[c] = e;
    }
    __syncthreads();

    const bool use_norm = a.background != nullptr && a.normalization != nullptr;
    float bk[2 * KS], nm[2 * KS];
#pragma unroll
    for (int b = 0; b < 2 * KS; ++b) {
        bk[b] = (use_norm && b < a.B) ? a.background[b] : 0.0f;
        nm[b] = (use_norm && b < a.B) ? a.normalization[b] : 1.0f;
    }

    const int64_t n_trips = (a.N + 63) / 64;
    const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
    const int64_t wave_stride = (int64_t)gridDim.x * (blockDim.x >> 6);
    // The pixel of the NEXT trip is requested while the current one goes through the matrix cores: with four waves
    // per SIMD and no prefetch the waves sat in s_waitcnt for 43 % of their cycles (profiles/r02_pmc_decode.json).
    // Clamped addresses, so the request is unconditional and lands in the loop-carried registers.
    float raw[2 * KS];
    {
        const int64_t p0 = min(wave_global * 64 + lane, a.N - 1);
#pragma unroll
        for (int b = 0; b < 2 * KS; ++b) raw[b] = a.stack[(size_t)min(b, a.B - 1) * a.N + p0];
    }
    for (int64_t trip = wave_global; trip < n_trips; trip += wave_stride) {
        const int64_t p = trip * 64 + lane;
        const bool valid = p < a.N;
        // ---- load + scale + clip (PixelDecoder.py:2399-2401, 2429)
        float s[2 * KS];
#pragma unroll
        for (int b = 0; b < 2 * KS; ++b) {
            float v = (valid && b < a.B) ? raw[b] : 0.0f;
            if (!a.prenormalized) {
                if (use_norm) v = rn_div(rn_sub(v, bk[b]), nm[b]);
                v = fminf(fmaxf(v, 0.0f), 1.0f);
            }
            s[b] = (b < a.B) ? v : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const int64_t pn = min((trip + wave_stride) * 64 + lane, a.N - 1);
#pragma unroll
            for (int b = 0; b < 2 * KS; ++b) raw[b] = a.stack[(size_t)min(b, a.B - 1) * a.N + pn];
        }
        __builtin_amdgcn_sched_barrier(0);
        // the clipped traces leave now (float16), so their registers are free during the matrix phase
        if (valid) {
#pragma unroll
            for (int b = 0; b < 2 * KS; ++b)
                if (b < a.B) a.scaled_f16[(size_t)b * a.N + p] = round5_f16(s[b]);
        }
        // ---- L2 norm over bits, sequential like add.reduce(axis=0) (:2459-2462)
        float acc = rn_mul(s[0], s[0]);
#pragma unroll
        for (int b = 1; b < 2 * KS; ++b) acc = rn_add(acc, rn_mul(s[b], s[b]));
        const float nrm = rn_sqrt(acc);
        const float mag = (nrm == 0.0f) ? -1.0f : nrm;
        const float den = (nrm == 0.0f) ? INFINITY : nrm;
        float xn[2 * KS];
        float XN = 0.0f;
#pragma unroll
        for (int b = 0; b < 2 * KS; ++b) {
            xn[b] = a.prenormalized ? s[b] : rn_div(s[b], den);
            XN = rn_fma(xn[b], xn[b], XN);
        }

        float my_d = INFINITY;
        int my_i = 0;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            // B operand of this column tile: pixel 32*pt + col, k = 2s + half
            const int src = 32 * pt + col;
            float bop[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const float e0 = __shfl(xn[2 * k], src, 64);
                const float e1 = __shfl(xn[2 * k + 1], src, 64);
                bop[k] = half ? e1 : e0;
            }
            const float xn_t = __shfl(XN, src, 64);
            float best2 = INFINITY;
            int besti = 0;
            for (int t = 0; t < NT; ++t) {
                f32x16 d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const float *af = afrag + (size_t)t * KS * 64 + lane;
#pragma unroll
                for (int k = 0; k < KS; ++k)
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k * 64], bop[k], d, 0, 0, 0);
                // d[r] = <codeword 32t + row(r), pixel>, row(r) = (r&3) + 8(r>>2) + 4*half.  The four codeword norms
                // of a row group are one 16-byte LDS read (ynq: [tile][half][group][4], +inf for rows past C, so
                // padding rows lose every comparison without a test).
                const float4 *yq = ynq + (t * 2 + half) * 4;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 y4 = yq[gq];
                    const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * gq + j;
                        const int c = 32 * t + j + 8 * gq + 4 * half;
                        const float tt = rn_add(xn_t, yv[j]);
                        const float d2 = fmaxf(rn_fma(-2.0f, d[r], tt), 0.0f);  // 2 d is exact: == tt - 2 d rounded once
                        const float b2o = best2;
                        const int bio = besti;
                        const bool lt = d2 < b2o;
                        best2 = lt ? d2 : b2o;
                        besti = lt ? c : bio;
                        // d = sqrt_rn(d2) is monotone: a smaller d2 wins unless it rounds to the same distance, where
                        // the earlier codeword must be kept.  Such near ties are rare: one wave-uniform test per
                        // element, the square roots only when some lane has one.
                        const bool near = lt && d2 >= b2o * 0.9999995f;
                        if (__builtin_amdgcn_ballot_w64(near)) {
                            if (near && rn_sqrt(d2) == rn_sqrt(b2o)) besti = bio;
                        }
                    }
                }
            }
            // merge the two half-waves (rows +0 / +4 of every 8)
            float bd = rn_sqrt(best2);
            const float od = __shfl_xor(bd, 32, 64);
            const int oi = __shfl_xor(besti, 32, 64);
            if (od < bd || (od == bd && oi < besti)) {
                bd = od;
                besti = oi;
            }
            if ((lane >> 5) == pt) {
                my_d = bd;
                my_i = besti;
            }
        }

        if (valid) {
            // two-threshold call + exclusions (:2610-2619, :863-877)
            int dec = (my_d <= a.thr) ? my_i : -1;
            if (mag < a.lo) dec = -1;
            if (mag > a.hi) dec = -1;
            if (excl[my_i]) dec = -1;
            a.decoded[p] = (int16_t)dec;
            a.magnitude_f16[p] = round5_f16(mag);
            a.distance_f16[p] = round5_f16(my_d);
            if (a.magnitude_f32) a.magnitude_f32[p] = mag;
            if (a.distance_f32) a.distance_f32[p] = my_d;
        }
    }
}

}  // namespace

extern "C" {

int m3d_weight_readout(const uint16_t *readout, const float *predictor, float *out, int64_t n,
                       void *stream)
{
    M3D_REQUIRE(readout && predictor && out && n >= 0, "null pointer");
    if (n == 0) return M3D_OK;
    hipLaunchKernelGGL(k_weight_readout, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       readout, predictor, out, n);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_gaussian_lowpass(const float *in, float *out, float *scratch, const int32_t shape_zyx[3],
                         const float sigma_zyx[3], int is_3d, void *stream)
{
    M3D_REQUIRE(in && out && scratch && shape_zyx && sigma_zyx, "null pointer");
    M3D_REQUIRE(scratch != in && scratch != out, "scratch must not alias in/out");
    const int Z = shape_zyx[0], Y = shape_zyx[1], X = shape_zyx[2];
    M3D_REQUIRE(Z > 0 && Y > 0 && X > 0, "shape must be positive");
    M3D_REQUIRE(sigma_zyx[0] > 0 && sigma_zyx[1] > 0 && sigma_zyx[2] > 0,
                "sigma must be > 0 (callers skip the filter when any sigma is 0)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t plane = (int64_t)Y * X;
    const GaussW gy = make_weights(sigma_zyx[1]), gx = make_weights(sigma_zyx[2]);
    M3D_REQUIRE(gy.radius <= kMaxRadius && gx.radius <= kMaxRadius, "sigma too large (radius > 32)");
    M3D_REQUIRE(gx.radius >= 1 && gy.radius >= 1, "sigma too small (radius 0)");
    M3D_REQUIRE((size_t)X * sizeof(float) <= 64 * 1024, "X extent too large for the row kernel");
    const float *cur = in;
    if (is_3d) {
        const GaussW gz = make_weights(sigma_zyx[0]);
        M3D_REQUIRE(gz.radius <= kMaxRadius && gz.radius >= 1, "sigma_z out of range");
        gauss_strided(cur, scratch, plane, plane, 0, Z, plane, gz, st);  // Z: in -> scratch
        gauss_strided(scratch, out, (int64_t)Z * X, X, plane, Y, X, gy, st);  // Y: scratch -> out
        cur = out;
    } else {
        gauss_strided(cur, scratch, (int64_t)Z * X, X, plane, Y, X, gy, st);  // Y: in -> scratch
        cur = scratch;
    }
    const int64_t rows = (int64_t)Z * Y;
    const int grid = (int)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL(k_gauss_x, dim3(grid), dim3(kBlock), (size_t)X * sizeof(float), st, cur, out,
                       rows, X, gx);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_separable_filter(const float *in, float *out, float *scratch, const int32_t shape_zyx[3],
                         const float *weights_host[3], const int32_t radius[3],
                         const int32_t offset[3], int wrap, void *stream)
{
    M3D_REQUIRE(in && out && scratch && shape_zyx && weights_host && radius && offset, "null pointer");
    M3D_REQUIRE(scratch != in && scratch != out, "scratch must not alias in/out");
    const int Z = shape_zyx[0], Y = shape_zyx[1], X = shape_zyx[2];
    M3D_REQUIRE(Z > 0 && Y > 0 && X > 0, "shape must be positive");
    M3D_REQUIRE((size_t)X * sizeof(float) <= 64 * 1024, "X extent too large for the row kernel");
    hipStream_t st = (hipStream_t)stream;
    const int64_t plane = (int64_t)Y * X;
    GaussW g[3];
    for (int a = 0; a < 3; ++a) {
        g[a].radius = radius[a];
        g[a].offset = offset[a];
        g[a].wrap = wrap ? 1 : 0;
        if (radius[a] < 0) continue;  // axis skipped
        M3D_REQUIRE(radius[a] <= kMaxRadius && weights_host[a], "filter radius must be <= 32");
        for (int k = 0; k < 2 * kMaxRadius + 1; ++k)
            g[a].w[k] = k <= 2 * radius[a] ? weights_host[a][k] : 0.0f;
    }
    // ping-pong so the last written buffer is `out`: count the active strided axes
    const float *cur = in;
    const bool do_z = radius[0] >= 0, do_y = radius[1] >= 0, do_x = radius[2] >= 0;
    auto strided = [&](int axis, float *dst) {
        if (axis == 0)
            gauss_strided(cur, dst, plane, plane, 0, Z, plane, g[0], st);
        else
            gauss_strided(cur, dst, (int64_t)Z * X, X, plane, Y, X, g[1], st);
        cur = dst;
    };
    if (do_z && do_y) {
        strided(0, scratch);
        strided(1, out);
    } else if (do_z || do_y) {
        strided(do_z ? 0 : 1, (do_x || in == out) ? scratch : out);
    }
    if (do_x) {
        const int64_t rows = (int64_t)Z * Y;
        const int grid = (int)(rows < 4096 ? rows : 4096);
        hipLaunchKernelGGL(k_gauss_x, dim3(grid), dim3(kBlock), (size_t)X * sizeof(float), st, cur, out,
                           rows, X, g[2]);
    } else if (cur != out) {
        M3D_HIP_CHECK(hipMemcpyAsync(out, cur, sizeof(float) * (size_t)Z * plane,
                                     hipMemcpyDeviceToDevice, st));
    }
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_decode_pixels(const float *stack, int32_t B, int64_t N, const float *background,
                      const float *normalization, const float *codebook, int32_t C,
                      float pixel_threshold, float magnitude_lo, float magnitude_hi,
                      const int32_t *excluded, int32_t n_excluded, int16_t *decoded,
                      uint16_t *magnitude_f16, uint16_t *distance_f16, uint16_t *scaled_f16,
                      float *magnitude_f32, float *distance_f32, int32_t flags, void *stream)
{
    M3D_REQUIRE(stack && codebook && decoded && magnitude_f16 && distance_f16 && scaled_f16,
                "null pointer");
    M3D_REQUIRE(B >= 1 && C >= 1 && N >= 0, "B, C must be positive");
    M3D_REQUIRE((background == nullptr) == (normalization == nullptr),
                "background and normalization must both be given or both be NULL");
    M3D_REQUIRE(n_excluded == 0 || excluded, "excluded indices missing");
    if (N == 0) return M3D_OK;
    if (B > 32) {
        set_err("m3d_decode_pixels", "more than 32 bits is not supported");
        return M3D_ERR_UNSUPPORTED;
    }
    const int KS = B <= 16 ? 8 : (B <= 22 ? 11 : 16);
    const int NT = (C + 31) / 32;
    const size_t lds = ((size_t)NT * KS * 64 + (size_t)NT * 32 * 2) * sizeof(float);
    if (lds > 64 * 1024) {
        set_err("m3d_decode_pixels", "codebook too large for the LDS-resident fragments");
        return M3D_ERR_UNSUPPORTED;
    }
    DecodeArgs a;
    a.stack = stack;
    a.background = background;
    a.normalization = normalization;
    a.codebook = codebook;
    a.excluded = excluded;
    a.decoded = decoded;
    a.magnitude_f16 = magnitude_f16;
    a.distance_f16 = distance_f16;
    a.scaled_f16 = scaled_f16;
    a.magnitude_f32 = magnitude_f32;
    a.distance_f32 = distance_f32;
    a.N = N;
    a.B = B;
    a.C = C;
    a.n_excluded = n_excluded;
    a.NT = NT;
    a.prenormalized = (flags & M3D_DECODE_PRENORMALIZED) ? 1 : 0;
    a.thr = pixel_threshold;
    a.lo = magnitude_lo;
    a.hi = magnitude_hi;
    const int64_t trips = (N + 63) / 64;
    int64_t blocks = (trips + 3) / 4;
    // blocks per CU of the grid-stride launch: 116 registers per lane allow four waves per SIMD = four resident blocks
    constexpr int per_cu = 4;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * (int64_t)per_cu) blocks = 256 * (int64_t)per_cu;
    hipStream_t st = (hipStream_t)stream;
    if (KS == 8)
        hipLaunchKernelGGL(k_decode<8>, dim3((int)blocks), dim3(256), lds, st, a);
    else if (KS == 11)
        hipLaunchKernelGGL(k_decode<11>, dim3((int)blocks), dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL(k_decode<16>, dim3((int)blocks), dim3(256), lds, st, a);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

}  // extern "C"
