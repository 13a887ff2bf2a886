// Phase-correlation building blocks for register_pair_to_fixed
// (utils/multiview_registration.py:208-320; cucim phase_cross_correlation with
// upsample_factor=10, disambiguate=True at :265-270 and :296-301).
//
// The transform sizes are whatever overlap crop the first stage leaves (e.g. 64 x 2040 x 2041),
// not 2-3-smooth, so the complex FFTs go through rocFFT (the fused in-LDS engine of
// m3d_fftconv.hip only exists for RLGC's padded sizes).  Everything around them is one pass
// each: real->complex, cross-power spectrum with phase normalisation, |.| argmax, the
// upsampled-DFT contraction of the last axis and the overlap correlation sums of the
// +-wrap disambiguation.  Scalars (shift estimate, 15-point DFT kernels) stay on the host.

#include <rocfft/rocfft.h>

#include <new>

#include "m3d_common.h"

using namespace m3d;

struct m3d_fft_plan {
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    size_t work = 0;
    int64_t n = 0;
};

namespace {

template <typename T>
__global__ void k_real_to_complex(const T *__restrict__ in, int64_t n, float2 *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = make_float2((float)in[i], 0.0f);
}

// image_product = src * conj(tgt); phase normalisation divides by max(|p|, 100 eps)
__global__ void k_cross_power(const float2 *__restrict__ a, const float2 *__restrict__ b, int64_t n,
                              int normalize, float2 *__restrict__ out)
{
    const float floor_v = 100.0f * 1.1920929e-07f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float2 x = a[i], y = b[i];
        float2 p = make_float2(x.x * y.x + x.y * y.y, x.y * y.x - x.x * y.y);
        if (normalize) {
            const float m = fmaxf(hypotf(p.x, p.y), floor_v);
            p.x /= m;
            p.y /= m;
        }
        out[i] = p;
    }
}

struct ArgMax {
    float v;
    long long i;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b)
{
    // larger magnitude wins, first (smallest index) on ties; NaN propagates like np.argmax
    if (b.v > a.v || (b.v == a.v && b.i < a.i) || (b.v != b.v && a.v == a.v)) return b;
    return a;
}
__global__ void k_argmax_abs(const float2 *__restrict__ d, int64_t n, ArgMax *__restrict__ partial)
{
    __shared__ ArgMax sm[4];
    ArgMax best = {-1.0f, (long long)n};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float2 x = d[i];
        ArgMax c = {hypotf(x.x, x.y), (long long)i};
        best = better(best, c);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ArgMax o;
        o.v = __shfl_down(best.v, off, 64);
        o.i = __shfl_down(best.i, off, 64);
        best = better(best, o);
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) best = better(best, sm[w]);
        partial[blockIdx.x] = best;
    }
}
__global__ void k_argmax_final(const ArgMax *__restrict__ partial, int nblocks, long long *idx,
                               float *val)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ArgMax best = partial[0];
    for (int i = 1; i < nblocks; ++i) best = better(best, partial[i]);
    *idx = best.i;
    *val = best.v;
}

// out[a, r] = sum_x K[a, x] * (conj?) D[r, x]     (np.tensordot(kernel, data, axes=(1, -1)))
constexpr int kMaxKa = 16;
__global__ void k_contract_last(const float2 *__restrict__ data, int64_t rows, int X,
                                const float2 *__restrict__ kern, int Ka, int conj_data,
                                float2 *__restrict__ out)
{
    __shared__ float smr[4], smi[4];
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        float2 acc[kMaxKa];
#pragma unroll
        for (int a = 0; a < kMaxKa; ++a) acc[a] = make_float2(0.f, 0.f);
        const float2 *row = data + r * X;
        for (int x = threadIdx.x; x < X; x += blockDim.x) {
            float2 d = row[x];
            if (conj_data) d.y = -d.y;
#pragma unroll
            for (int a = 0; a < kMaxKa; ++a) {
                if (a < Ka) {
                    const float2 k = kern[(size_t)a * X + x];
                    acc[a].x += k.x * d.x - k.y * d.y;
                    acc[a].y += k.x * d.y + k.y * d.x;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < kMaxKa; ++a) {
            if (a >= Ka) break;
            const float re = block_sum(acc[a].x, smr);
            const float im = block_sum(acc[a].y, smi);
            if (threadIdx.x == 0) out[(size_t)a * rows + r] = make_float2(re, im);
        }
    }
}

// sums over the box [lo, hi): n, sum a, sum b, sum aa, sum bb, sum ab  (np.corrcoef inputs)
__global__ void k_box_corr_sums(const float *__restrict__ a, const float *__restrict__ b, int Y, int X,
                                int z0, int z1, int y0, int y1, int x0, int x1,
                                double *__restrict__ sums)
{
    __shared__ double sm[4];
    double s[5] = {0, 0, 0, 0, 0};
    const int64_t rows = (int64_t)(z1 - z0) * (y1 - y0);
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int z = z0 + (int)(r / (y1 - y0)), y = y0 + (int)(r % (y1 - y0));
        const int64_t base = ((int64_t)z * Y + y) * X;
        for (int x = x0 + threadIdx.x; x < x1; x += blockDim.x) {
            const double u = a[base + x], v = b[base + x];
            s[0] += u;
            s[1] += v;
            s[2] += u * u;
            s[3] += v * v;
            s[4] += u * v;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double t = block_sum(s[k], sm);
        if (threadIdx.x == 0 && t != 0.0) atomicAdd(&sums[k + 1], t);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0)
        atomicAdd(&sums[0], (double)rows * (double)(x1 - x0));
}

}  // namespace

#define M3D_ROCFFT_TRY(expr)                               \
    do {                                                   \
        rocfft_status _s = (expr);                         \
        if (_s != rocfft_status_success) {                 \
            char _b[64];                                   \
            snprintf(_b, 64, "rocfft_status %d", (int)_s); \
            m3d::set_err(#expr, _b);                       \
            return M3D_ERR_ROCFFT;                         \
        }                                                  \
    } while (0)

extern "C" {

int m3d_fft_plan_create(int ndim, const int32_t *shape, m3d_fft_plan **out)
{
    return m3d_fft_plan_create_batched(ndim, shape, 1, out);
}

int m3d_fft_plan_create_batched(int ndim, const int32_t *shape, int32_t batch, m3d_fft_plan **out)
{
    M3D_REQUIRE(out && shape && ndim >= 1 && ndim <= 3, "ndim must be 1..3");
    M3D_REQUIRE(batch >= 1, "batch must be positive");
    size_t lengths[3];
    int64_t n = 1;
    for (int d = 0; d < ndim; ++d) {
        M3D_REQUIRE(shape[d] > 0, "shape must be positive");
        lengths[ndim - 1 - d] = (size_t)shape[d];  // rocFFT wants the fastest axis first
        n *= shape[d];
    }
    m3d_fft_plan *p = new (std::nothrow) m3d_fft_plan();
    if (!p) return M3D_ERR_OUT_OF_MEMORY;
    p->n = n * batch;
    static bool setup_done = false;
    if (!setup_done) {
        rocfft_setup();
        setup_done = true;
    }
    rocfft_status s = rocfft_plan_create(&p->fwd, rocfft_placement_inplace,
                                         rocfft_transform_type_complex_forward,
                                         rocfft_precision_single, (size_t)ndim, lengths, (size_t)batch,
                                         nullptr);
    if (s == rocfft_status_success)
        s = rocfft_plan_create(&p->inv, rocfft_placement_inplace,
                               rocfft_transform_type_complex_inverse, rocfft_precision_single,
                               (size_t)ndim, lengths, (size_t)batch, nullptr);
    size_t wf = 0, wi = 0;
    if (s == rocfft_status_success) s = rocfft_plan_get_work_buffer_size(p->fwd, &wf);
    if (s == rocfft_status_success) s = rocfft_plan_get_work_buffer_size(p->inv, &wi);
    if (s == rocfft_status_success) s = rocfft_execution_info_create(&p->info);
    if (s != rocfft_status_success) {
        set_err("rocfft plan creation", "complex transform");
        if (p->fwd) rocfft_plan_destroy(p->fwd);
        if (p->inv) rocfft_plan_destroy(p->inv);
        delete p;
        return M3D_ERR_ROCFFT;
    }
    p->work = wf > wi ? wf : wi;
    *out = p;
    return M3D_OK;
}

size_t m3d_fft_plan_workspace_bytes(const m3d_fft_plan *p) { return p ? p->work : 0; }

int m3d_fft_plan_destroy(m3d_fft_plan *p)
{
    if (!p) return M3D_OK;
    if (p->info) rocfft_execution_info_destroy(p->info);
    if (p->fwd) rocfft_plan_destroy(p->fwd);
    if (p->inv) rocfft_plan_destroy(p->inv);
    delete p;
    return M3D_OK;
}

int m3d_fft_exec(m3d_fft_plan *p, void *data_c64, int inverse, void *work, size_t work_bytes,
                 void *stream)
{
    M3D_REQUIRE(p && data_c64, "null pointer");
    if (p->work > work_bytes || (p->work && !work)) {
        set_err("m3d_fft_exec", "work buffer smaller than m3d_fft_plan_workspace_bytes");
        return M3D_ERR_WORKSPACE;
    }
    M3D_ROCFFT_TRY(rocfft_execution_info_set_stream(p->info, (hipStream_t)stream));
    if (p->work) M3D_ROCFFT_TRY(rocfft_execution_info_set_work_buffer(p->info, work, p->work));
    void *buf[1] = {data_c64};
    M3D_ROCFFT_TRY(rocfft_execute(inverse ? p->inv : p->fwd, buf, nullptr, p->info));
    return M3D_OK;  // unnormalised like rocFFT; callers scale the inverse by 1/n where it matters
}

int m3d_real_to_complex(const void *in, int dtype, int64_t n, void *out_c64, void *stream)
{
    M3D_REQUIRE(in && out_c64 && n >= 0, "null pointer");
    M3D_REQUIRE(dtype == M3D_U16 || dtype == M3D_F32, "dtype must be u16 or f32");
    if (n == 0) return M3D_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == M3D_U16)
        hipLaunchKernelGGL(k_real_to_complex<uint16_t>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                           (const uint16_t *)in, n, (float2 *)out_c64);
    else
        hipLaunchKernelGGL(k_real_to_complex<float>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                           (const float *)in, n, (float2 *)out_c64);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_cross_power(const void *src_freq, const void *target_freq, int64_t n, int normalize,
                    void *out, void *stream)
{
    M3D_REQUIRE(src_freq && target_freq && out && n >= 0, "null pointer");
    if (n == 0) return M3D_OK;
    hipLaunchKernelGGL(k_cross_power, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const float2 *)src_freq, (const float2 *)target_freq, n, normalize,
                       (float2 *)out);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_argmax_abs(const void *data_c64, int64_t n, void *scratch, int64_t *index_out,
                   float *value_out, void *stream)
{
    M3D_REQUIRE(data_c64 && scratch && index_out && value_out && n > 0, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(n);  // scratch must hold kMaxGrid * 16 bytes
    hipLaunchKernelGGL(k_argmax_abs, dim3(grid), dim3(kBlock), 0, st, (const float2 *)data_c64, n,
                       (ArgMax *)scratch);
    hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(64), 0, st, (const ArgMax *)scratch, grid,
                       (long long *)index_out, value_out);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_contract_last(const void *data_c64, int64_t rows, int32_t X, const void *kernel_c64,
                      int32_t Ka, int conj_data, void *out_c64, void *stream)
{
    M3D_REQUIRE(data_c64 && kernel_c64 && out_c64 && rows > 0 && X > 0, "null pointer");
    M3D_REQUIRE(Ka >= 1 && Ka <= kMaxKa, "kernel rows must be 1..16");
    const int grid = (int)(rows < 65535 ? rows : 65535);
    hipLaunchKernelGGL(k_contract_last, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       (const float2 *)data_c64, rows, X, (const float2 *)kernel_c64, Ka, conj_data,
                       (float2 *)out_c64);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_box_corr_sums(const float *a, const float *b, const int32_t shape_zyx[3],
                      const int32_t lo_zyx[3], const int32_t hi_zyx[3], double *sums6, void *stream)
{
    M3D_REQUIRE(a && b && shape_zyx && lo_zyx && hi_zyx && sums6, "null pointer");
    for (int d = 0; d < 3; ++d)
        M3D_REQUIRE(lo_zyx[d] >= 0 && hi_zyx[d] <= shape_zyx[d], "box outside the volume");
    hipStream_t st = (hipStream_t)stream;
    M3D_HIP_CHECK(hipMemsetAsync(sums6, 0, 6 * sizeof(double), st));
    if (hi_zyx[0] <= lo_zyx[0] || hi_zyx[1] <= lo_zyx[1] || hi_zyx[2] <= lo_zyx[2]) return M3D_OK;
    const int64_t rows = (int64_t)(hi_zyx[0] - lo_zyx[0]) * (hi_zyx[1] - lo_zyx[1]);
    const int grid = (int)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL(k_box_corr_sums, dim3(grid), dim3(kBlock), 0, st, a, b, shape_zyx[1],
                       shape_zyx[2], lo_zyx[0], hi_zyx[0], lo_zyx[1], hi_zyx[1], lo_zyx[2],
                       hi_zyx[2], sums6);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

}  // extern "C"
