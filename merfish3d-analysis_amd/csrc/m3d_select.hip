// Exact order statistics by 8-bit radix select, for the global normalisation vectors
// (PixelDecoder._global_normalization_vectors, PixelDecoder.py:981-1199: cp.percentile at
// :1113/:1157, cp.median at :1072/:1127/:1177).
//
// The reference sorts/partitions whole volumes on one GPU.  Here every rank histograms the
// next key byte of its own voxels (one streaming pass, 4 B/voxel read), the 256-bin int64
// histograms are summed (RCCL all-reduce over xGMI when tiles are sharded across GPUs,
// ~2 KB per bit - latency bound) and after four passes the k-th smallest value is known
// exactly, so percentiles / medians over tiles held by different GPUs are identical to the
// single-GPU result.  Values can be filtered (strictly below / above a cut-off) and
// transformed (max(x - background, 0)) on the fly, which is what the pooled-tail medians of
// :1113-1127 and :1144-1177 need - no compaction, no temporary copies.

#include "m3d_common.h"

using namespace m3d;

namespace {

__device__ __forceinline__ uint32_t sortable_key(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// transform: 0 none, 1 max(x - sub, 0).  filter: 0 none, 1 keep x < thr, 2 keep x > thr
// (tested after the transform).  NaNs are dropped.
__global__ void k_radix_hist(const float *__restrict__ values, int64_t n, int transform, float sub,
                             int filter, double thr, int pass, uint32_t prefix,
                             unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int lh[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const uint32_t mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v = values[i];
        if (transform == 1) {
            v = v - sub;
            v = v < 0.0f ? 0.0f : v;
        }
        if (v != v) continue;
        if (filter == 1 && !((double)v < thr)) continue;
        if (filter == 2 && !((double)v > thr)) continue;
        const uint32_t k = sortable_key(v);
        if ((k & mask) != (prefix & mask)) continue;
        atomicAdd(&lh[(k >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// current_image[current_image > thr] = value   (hot-pixel repair, PixelDecoder.py:1072-1074)
__global__ void k_replace_above(float *__restrict__ v, int64_t n, float thr, float value)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        if (v[i] > thr) v[i] = value;
}

}  // namespace

extern "C" {

int m3d_radix_hist(const float *values, int64_t n, int transform, float sub, int filter, double thr,
                   int pass, uint32_t prefix, int64_t *hist256, void *stream)
{
    M3D_REQUIRE(hist256 && (values || n == 0) && n >= 0, "null pointer");
    M3D_REQUIRE(pass >= 0 && pass <= 3, "pass must be 0..3");
    M3D_REQUIRE(transform >= 0 && transform <= 1 && filter >= 0 && filter <= 2, "bad transform/filter");
    if (n == 0) return M3D_OK;
    hipLaunchKernelGGL(k_radix_hist, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, values, n,
                       transform, sub, filter, thr, pass, prefix, (unsigned long long *)hist256);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_replace_above(float *values, int64_t n, float threshold, float replacement, void *stream)
{
    M3D_REQUIRE((values || n == 0) && n >= 0, "null pointer");
    if (n == 0) return M3D_OK;
    hipLaunchKernelGGL(k_replace_above, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, values,
                       n, threshold, replacement);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

}  // extern "C"
