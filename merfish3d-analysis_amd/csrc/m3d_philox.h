// Philox4x32 (kSplitRounds rounds): the counter-based generator under the Binomial(n, 1/2) photon split of
// m3d_binom.h, which replaces cp.random.default_rng(seed).binomial(n, 0.5) of the reference
// (utils/rlgc.py:564, 619-622).  Stateless, so the split is regenerated wherever it is needed instead of being
// stored (saves two full-volume passes per iteration); integer only - bit-identical to oracle/philox.py.
//
//   key     = (seed lo, seed hi)
//   counter = (voxel lo, voxel hi, iteration, block);  each block yields 128 bits
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define M3D_HD __host__ __device__ __forceinline__
#else
#define M3D_HD static inline
#endif

namespace m3d {

struct U32x4 {
    uint32_t x, y, z, w;
};

// Rounds of the photon split.  Random123 recommends 10 as a safety margin; Philox4x32 passes BigCrush from 7 rounds
// on (Salmon et al., SC'11, table 2), and the split kernel is bound by exactly these multiplies (profiles/
// r02_pmc_sq_rlgc_after.json: one vector instruction per four cycles per SIMD, 1.5 G of them per launch).
constexpr int kSplitRounds = 7;

// a ^ b ^ c: one v_bitop3_b32 on the device (the compiler emits two v_xor_b32 for the expression)
M3D_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

template <int ROUNDS>
M3D_HD U32x4 philox4x32(U32x4 c, uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c.x;
        const uint64_t p1 = (uint64_t)M1 * c.z;
        U32x4 n;
        n.x = xor3((uint32_t)(p1 >> 32), c.y, k0);
        n.y = (uint32_t)p1;
        n.z = xor3((uint32_t)(p0 >> 32), c.w, k1);
        n.w = (uint32_t)p0;
        c = n;
        k0 += W0;
        k1 += W1;
    }
    return c;
}

M3D_HD U32x4 philox4x32_10(U32x4 c, uint32_t k0, uint32_t k1) { return philox4x32<10>(c, k0, k1); }  // Random123 KAT form

M3D_HD int popc32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}

M3D_HD uint32_t low_mask(int nbits)  // nbits in [0, 32]
{
    return nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u);
}

}  // namespace m3d
