// Connected components of equal codeword ids + per-component statistics on MI355X.
//
// Replaces, inside PixelDecoder._extract_barcodes (PixelDecoder.py:2908-3201):
//   cucim label(background=-1, connectivity=3 | per-plane connectivity=2)      :2946-2972
//   bincount / size filters (applied on the compact table by the host)          :2976-2989
//   cucim regionprops_table(label, intensity_min, area) on the distance image   :2991-2995
//   CPU skimage regionprops_table(area, centroid, intensity_mean x B, inertia)  :3025-3042
//   the full-volume host argsort that maps label -> codeword                    :3066-3078
//
// Decoded voxels are a few percent of the volume, so everything after the labelling pass
// works on the sparse foreground through atomics: lock-free union-find (atomicMin keeps
// the raster-first voxel as root, which is also scikit-image's label order), one slot per
// root from a counter, then double-precision atomic sums per slot.  The sums are integers
// or float16 values (multiples of 2^-24 below 2^16): they are EXACT in double whatever
// the arrival order, so the table is bit-reproducible.

#include "m3d_common.h"

using namespace m3d;

namespace {

__device__ __forceinline__ int uf_find(const int32_t *__restrict__ parent, int x)
{
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

__device__ __forceinline__ void uf_union(int32_t *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        // a > b: hang a under b if a is still a root
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;  // someone re-parented a meanwhile: retry from there
    }
}

__global__ void k_ccl_init(const int16_t *__restrict__ img, int32_t *__restrict__ parent, int64_t n,
                           int16_t background)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        parent[i] = (img[i] == background) ? -1 : (int32_t)i;
}

// 13 "earlier" neighbours; order = number of non-zero offsets must be <= connectivity.
__global__ void k_ccl_union(const int16_t *__restrict__ img, int32_t *__restrict__ parent, int Z, int Y,
                            int X, int16_t background, int connectivity, int per_plane)
{
    const int64_t n = (int64_t)Z * Y * X;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int16_t v = img[i];
        if (v == background) continue;
        const int x = (int)(i % X);
        const int64_t t = i / X;
        const int y = (int)(t % Y), z = (int)(t / Y);
        for (int dz = per_plane ? 0 : -1; dz <= 0; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dz == 0 && (dy > 0 || (dy == 0 && dx >= 0))) continue;
                    const int order = (dz != 0) + (dy != 0) + (dx != 0);
                    if (order > connectivity) continue;
                    const int zz = z + dz, yy = y + dy, xx = x + dx;
                    if (zz < 0 || yy < 0 || yy >= Y || xx < 0 || xx >= X) continue;
                    const int64_t j = ((int64_t)zz * Y + yy) * X + xx;
                    if (img[j] != v) continue;
                    uf_union(parent, (int)i, (int)j);
                }
    }
}

// Flatten to the root and hand every root a dense slot.
__global__ void k_ccl_flatten(int32_t *__restrict__ parent, int32_t *__restrict__ slot, int64_t n,
                              int32_t *__restrict__ counter)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int p = parent[i];
        if (p < 0) continue;
        if (p == (int)i) {
            slot[i] = atomicAdd(counter, 1);
        }
    }
}
__global__ void k_ccl_resolve(int32_t *__restrict__ parent, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int p = parent[i];
        if (p < 0 || p == (int)i) continue;
        // roots never change after k_ccl_union finished, so racing writers store the same value
        parent[i] = uf_find(parent, p);
    }
}

__device__ __forceinline__ float f16_bits_to_f32(uint16_t b)
{
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}

constexpr int kSumCols = 11;  // area, sz, sy, sx, szz, syy, sxx, szy, szx, syx, magnitude sum

template <typename TI>
__global__ void k_region_stats(const int16_t *__restrict__ img, const int32_t *__restrict__ label,
                               const int32_t *__restrict__ slot, int Z, int Y, int X, int B,
                               const TI *__restrict__ intensity, const uint16_t *__restrict__ magnitude,
                               const uint16_t *__restrict__ distance, int32_t *__restrict__ comp_root,
                               int32_t *__restrict__ comp_code, unsigned int *__restrict__ comp_mind,
                               double *__restrict__ sums)
{
    const int64_t n = (int64_t)Z * Y * X;
    const int ncol = kSumCols + B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = label[i];
        if (r < 0) continue;
        const int s = slot[r];
        const int x = (int)(i % X);
        const int64_t t = i / X;
        const int y = (int)(t % Y), z = (int)(t / Y);
        double *row = sums + (size_t)s * ncol;
        if (r == (int)i) {
            comp_root[s] = r;
            comp_code[s] = (int32_t)img[i];
        }
        const double dz = z, dy = y, dx = x;
        atomicAdd(&row[0], 1.0);
        atomicAdd(&row[1], dz);
        atomicAdd(&row[2], dy);
        atomicAdd(&row[3], dx);
        atomicAdd(&row[4], dz * dz);
        atomicAdd(&row[5], dy * dy);
        atomicAdd(&row[6], dx * dx);
        atomicAdd(&row[7], dz * dy);
        atomicAdd(&row[8], dz * dx);
        atomicAdd(&row[9], dy * dx);
        atomicAdd(&row[10], (double)f16_bits_to_f32(magnitude[i]));
        const float d = f16_bits_to_f32(distance[i]);
        atomicMin(&comp_mind[s], __float_as_uint(d < 0.0f ? 0.0f : d));
        for (int b = 0; b < B; ++b) {
            float v;
            if constexpr (sizeof(TI) == 2)
                v = f16_bits_to_f32(((const uint16_t *)intensity)[(size_t)b * n + i]);
            else
                v = ((const float *)intensity)[(size_t)b * n + i];
            atomicAdd(&row[kSumCols + b], (double)v);
        }
    }
}

__global__ void k_fill_u32(unsigned int *p, unsigned int v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

}  // namespace

extern "C" {

int m3d_ccl_label(const int16_t *decoded, const int32_t shape_zyx[3], int connectivity,
                  int per_plane, int32_t *labels, int32_t *slot, int32_t *n_components,
                  void *stream)
{
    M3D_REQUIRE(decoded && shape_zyx && labels && slot && n_components, "null pointer");
    M3D_REQUIRE(connectivity >= 1 && connectivity <= 3, "connectivity must be 1, 2 or 3");
    const int Z = shape_zyx[0], Y = shape_zyx[1], X = shape_zyx[2];
    M3D_REQUIRE(Z > 0 && Y > 0 && X > 0, "shape must be positive");
    const int64_t n = (int64_t)Z * Y * X;
    M3D_REQUIRE(n < 2147483647LL, "volume too large for int32 labels");
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(n);
    M3D_HIP_CHECK(hipMemsetAsync(n_components, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_ccl_init, dim3(grid), dim3(kBlock), 0, st, decoded, labels, n, (int16_t)-1);
    hipLaunchKernelGGL(k_ccl_union, dim3(grid), dim3(kBlock), 0, st, decoded, labels, Z, Y, X,
                       (int16_t)-1, connectivity, per_plane);
    hipLaunchKernelGGL(k_ccl_flatten, dim3(grid), dim3(kBlock), 0, st, labels, slot, n, n_components);
    hipLaunchKernelGGL(k_ccl_resolve, dim3(grid), dim3(kBlock), 0, st, labels, n);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_region_stats(const int16_t *decoded, const int32_t *labels, const int32_t *slot,
                     const int32_t shape_zyx[3], int32_t B, const void *intensity,
                     int intensity_dtype, const uint16_t *magnitude_f16,
                     const uint16_t *distance_f16, int32_t n_components, int32_t *comp_root,
                     int32_t *comp_code, float *comp_min_distance, double *comp_sums, void *stream)
{
    M3D_REQUIRE(decoded && labels && slot && shape_zyx && intensity && magnitude_f16 && distance_f16,
                "null pointer");
    M3D_REQUIRE(intensity_dtype == M3D_F16 || intensity_dtype == M3D_F32,
                "intensity must be float16 or float32");
    M3D_REQUIRE(B >= 0 && n_components >= 0, "negative count");
    if (n_components == 0) return M3D_OK;
    M3D_REQUIRE(comp_root && comp_code && comp_min_distance && comp_sums, "null output");
    const int Z = shape_zyx[0], Y = shape_zyx[1], X = shape_zyx[2];
    const int64_t n = (int64_t)Z * Y * X;
    hipStream_t st = (hipStream_t)stream;
    M3D_HIP_CHECK(hipMemsetAsync(comp_sums, 0, sizeof(double) * (size_t)n_components * (kSumCols + B), st));
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n_components)), dim3(kBlock), 0, st,
                       (unsigned int *)comp_min_distance, 0x7F800000u, (int64_t)n_components);
    if (intensity_dtype == M3D_F16)
        hipLaunchKernelGGL(k_region_stats<uint16_t>, dim3(grid_for(n)), dim3(kBlock), 0, st, decoded,
                           labels, slot, Z, Y, X, B, (const uint16_t *)intensity, magnitude_f16,
                           distance_f16, comp_root, comp_code, (unsigned int *)comp_min_distance,
                           comp_sums);
    else
        hipLaunchKernelGGL(k_region_stats<float>, dim3(grid_for(n)), dim3(kBlock), 0, st, decoded,
                           labels, slot, Z, Y, X, B, (const float *)intensity, magnitude_f16,
                           distance_f16, comp_root, comp_code, (unsigned int *)comp_min_distance,
                           comp_sums);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

}  // extern "C"
