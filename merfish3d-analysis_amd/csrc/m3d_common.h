// Shared host/device helpers for libm3d_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/m3d_hip.h"

namespace m3d {

// ---- per-thread error text -------------------------------------------------------------
inline char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}
inline void set_err(const char *what, const char *detail)
{
    snprintf(err_buf(), 512, "%s: %s", what, detail ? detail : "");
}

#define M3D_HIP_CHECK(expr)                                                       \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            m3d::set_err(#expr, hipGetErrorString(_e));                           \
            return (_e == hipErrorOutOfMemory) ? M3D_ERR_OUT_OF_MEMORY : M3D_ERR_HIP; \
        }                                                                         \
    } while (0)

#define M3D_KERNEL_CHECK() M3D_HIP_CHECK(hipGetLastError())

#define M3D_REQUIRE(cond, msg)                     \
    do {                                           \
        if (!(cond)) {                             \
            m3d::set_err("invalid argument", msg); \
            return M3D_ERR_INVALID_ARG;            \
        }                                          \
    } while (0)

// ---- launch geometry --------------------------------------------------------------------
constexpr int kBlock = 256;        // 4 waves of 64
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 blocks: grid-stride the rest

inline int grid_for(int64_t n, int per_thread = 1)
{
    int64_t blocks = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
    if (blocks < 1) blocks = 1;
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    return (int)blocks;
}

// ---- wave / block reductions (wave = 64 lanes) -------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_down(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// Block-wide reductions (1-D or 2-D blocks whose x extent is a multiple of 64, at most
// 256 threads); result valid in the thread with flat id 0.
__device__ __forceinline__ int flat_tid() { return threadIdx.y * blockDim.x + threadIdx.x; }
__device__ __forceinline__ int flat_nwaves() { return (blockDim.x * blockDim.y + 63) >> 6; }
template <typename T>
__device__ __forceinline__ T block_sum(T v, T *smem /* >= 4 */)
{
    v = wave_sum(v);
    const int lane = flat_tid() & 63, w = flat_tid() >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    T r = 0;
    if (flat_tid() == 0) {
        const int nw = flat_nwaves();
        for (int i = 0; i < nw; ++i) r += smem[i];  // fixed order: deterministic
    }
    return r;
}
template <typename T>
__device__ __forceinline__ T block_max(T v, T *smem)
{
    v = wave_max(v);
    const int lane = flat_tid() & 63, w = flat_tid() >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    T r = v;
    if (flat_tid() == 0) {
        const int nw = flat_nwaves();
        r = smem[0];
        for (int i = 1; i < nw; ++i) r = smem[i] > r ? smem[i] : r;
    }
    return r;
}
template <typename T>
__device__ __forceinline__ T block_min(T v, T *smem)
{
    v = wave_min(v);
    const int lane = flat_tid() & 63, w = flat_tid() >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    T r = v;
    if (flat_tid() == 0) {
        const int nw = flat_nwaves();
        r = smem[0];
        for (int i = 1; i < nw; ++i) r = smem[i] < r ? smem[i] : r;
    }
    return r;
}

// ---- IEEE-exact float32 primitives -----------------------------------------------------------
// hipcc's __fsqrt_rn is the approximate native sqrt unless OCML_BASIC_ROUNDED_OPERATIONS is
// defined, and __fadd_rn / __fmul_rn are plain operators: with -ffp-contract=off (set in the
// Makefile) the plain operators are already single-rounded, and `/` and sqrtf() are correctly
// rounded by default (-fhip-fp32-correctly-rounded-divide-sqrt).  These wrappers document
// every place where the result must match the CPU oracle bit for bit.
__device__ __forceinline__ float rn_add(float a, float b) { return a + b; }
__device__ __forceinline__ float rn_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float rn_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float rn_div(float a, float b) { return a / b; }
__device__ __forceinline__ float rn_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float rn_sqrt(float a) { return __builtin_sqrtf(a); }

// ---- float16 (IEEE binary16, round-to-nearest-even) as raw bits ---------------------------
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f)
{
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32: RNE, overflow -> inf, NaN kept
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}

}  // namespace m3d
