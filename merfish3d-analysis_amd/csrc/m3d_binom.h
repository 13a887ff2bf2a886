// Exact Binomial(n, 1/2) of bounded work: the photon split of RLGC (utils/rlgc.py:564, 619-622,
// cp.random.default_rng(seed).binomial(n, 0.5) in the reference) from ONE Philox block per voxel whatever n.
//
//   n = 64 q + r.   Binomial(n, 1/2) = Binomial(r, 1/2) + Binomial(64 q, 1/2), independent.
//   r < 64 : pop-count of r fresh bits (words x, y of the voxel's Philox block 0)      - exact
//   q = 1  : pop-count of 64 fresh bits (words z, w)                                    - exact
//   q >= 2 : inversion of the 64-bit uniform (z:w).  X = 32 q +- D with D = |X - 32 q|; the sign is the lowest
//            bit, D is the number of thresholds T_q[i] = floor(2^63 P(D <= i)) that are <= the other 63 bits v.
//            Every outcome's probability is its exact value rounded to a multiple of 2^-63 - the resolution of a
//            double-precision BTPE sampler (what the reference's CuPy draws with) is coarser than that.
//   q > 1023 (counts above 65535 + ..., float images only): chunks of 64 * 1023 photons, one more 64-bit uniform
//            each (two per further Philox block).
//
// The thresholds are not computed in floating point: `binom_tables_build` evaluates C(64 q, 32 q + d) / 2^(64 q) in
// 191-bit fixed point with every division rounded down and carries the bound of the accumulated truncation error
// along; a threshold is accepted only if floor() of both ends of its error interval agree, so the tables hold the
// mathematically exact integers (oracle/philox.py certifies the same way and must - and does - arrive at the same
// numbers; tests/test_binomial_split.py compares them entry by entry).
//
// The search for D starts from an arithmetic guess - the normal quantile of the tail mass with the binomial's
// kurtosis and lattice corrections, off by one threshold about once in 10^4 draws - and is then decided by the
// exact integers alone: one 16-byte load of the two thresholds around the guess settles almost every draw, a wrong
// guess falls back to a bisection over the whole row, so the float arithmetic of the guess, which differs between
// host and device, cannot change a result.  A per-lane gather costs the L1 one tag look-up per distinct line and,
// when the rows in use exceed its 32 KB, a line from L2: six of them per draw (a guide table + bisection, the first
// version) made the kernel 3.9 x slower on a saturated image than on a dim one, the single one 2 x
// (profiles/r03_split_brightness.txt; the pop-count sampler this replaces was 110 x slower there).
// Tables: 1024 rows of 1168 entries (9.6 MB, 6.4 MB of it thresholds), built once per process (a few tens of ms),
// one copy per device.  A row starts with a sentinel 0 and ends in 2^64 - 1s, so that the pair load needs no edge
// cases, and sits at q * 1168 so that finding it needs no load.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "m3d_philox.h"

#include <vector>

namespace m3d {

constexpr uint32_t kBinomQMax = 1023;   // largest table row: Binomial(65472, 1/2)
constexpr uint32_t kBinomRowStride = 1168;  // 8-byte entries per row: 1 + 1162 thresholds at most, padded to 128-byte lines
constexpr uint64_t kBinomVMax = 0x7FFFFFFFFFFFFFFFull;

struct BinomRow {
    uint32_t off;  // the row's leading sentinel in `thr` (= q * kBinomRowStride); threshold i sits at off + 1 + i
    uint32_t len;  // thresholds below 2^63 - 1; the rest of the row holds 2^64 - 1
};

// A guide for the SMALL rows (q = 2 .. kBinomGuideQHi: counts of 128 to 64 (kBinomGuideQHi + 1) - 1, the table draws of
// a dim image), small enough to sit in LDS next to a kernel's own buffers: 256 cells per row over the top eight bits of
// v, a cell = g0 | min(g1 - g0, 3) << 12 with g0 = #{thresholds <= the cell's lower end}, g1 the same count at the next
// cell's lower end (the row's length after the last cell).  For a v inside the cell, g0 <= D <= g1.
constexpr uint32_t kBinomGuideQLo = 2, kBinomGuideQHi = 8;
constexpr int kBinomGuideBits = 8;
constexpr uint32_t kBinomGuideCells = 1u << kBinomGuideBits;
constexpr uint32_t kBinomGuideEntries = (kBinomGuideQHi - kBinomGuideQLo + 1) * kBinomGuideCells;  // 1792 x 2 B = 3.5 KB

struct BinomTables {  // a view; host or device pointers
    const uint64_t *thr;
    const BinomRow *rows;     // [kBinomQMax + 1], rows 0 and 1 empty
    const uint16_t *guide;    // [kBinomGuideEntries], device copies only (global memory; kernels stage it into LDS)
};

struct BinomPair {
    uint64_t below, above;
};

M3D_HD BinomPair binom_load_pair(const uint64_t *p)  // 8-byte aligned
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2), aligned(8)));
    const u64x2 v = *(const u64x2 *)p;  // one global_load_dwordx4 (through L1: with `nt` the kernel is 1.5 x slower)
    return BinomPair{v.x, v.y};
#else
    return BinomPair{p[0], p[1]};
#endif
}

M3D_HD float binom_tail_quantile(float tail, float q)
{
    // t with P(|X - 32 q| > sigma t - 1/2) ~ tail, sigma = 4 sqrt(q).  Normal quantile z of the upper tail mass
    // tail / 2 as w + P12(u(1 / w)), w = sqrt(-2 ln(tail / 2)) (|error| < 4e-6 on [2^-64, 1/2],
    // scripts/fit_tail_quantile.py), then the binomial's two 1 / n terms: excess kurtosis -2 / n,
    // -(z^3 - 3 z) / (12 n), and the lattice (Sheppard) term -z / (24 sigma^2); together -(z^3 - z) / (768 q).
    // Against the exact thresholds the guess floor(sigma t + 1/2) is off for about one draw in 10^4.
    // Raw v_sqrt_f32 / v_log_f32 / v_rcp_f32 and explicit fmas on the device (1 ulp is plenty for a starting point; the
    // file is compiled with contraction off and correctly rounded sqrt, which would triple the instruction count).
#if defined(__HIP_DEVICE_COMPILE__)
#define M3D_BINOM_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
    const float w = __builtin_amdgcn_sqrtf(-1.3862944f * (__builtin_amdgcn_logf(tail) - 1.0f));
    const float u = M3D_BINOM_FMA(2.6912229f, __builtin_amdgcn_rcpf(w), -1.2857143f);
    const float rn = __builtin_amdgcn_rcpf(768.0f * q);
#else
#define M3D_BINOM_FMA(a, b, c) ((a) * (b) + (c))
    const float w = __builtin_sqrtf(-1.3862944f * (__builtin_log2f(tail) - 1.0f));
    const float u = 2.6912229f / w - 1.2857143f;
    const float rn = 1.0f / (768.0f * q);
#endif
    float g = 0.00241146086f;
    g = M3D_BINOM_FMA(g, u, -0.00129357789f);
    g = M3D_BINOM_FMA(g, u, -0.00481956166f);
    g = M3D_BINOM_FMA(g, u, 0.000649343032f);
    g = M3D_BINOM_FMA(g, u, 0.00615365055f);
    g = M3D_BINOM_FMA(g, u, -0.00262752589f);
    g = M3D_BINOM_FMA(g, u, 0.00122392602f);
    g = M3D_BINOM_FMA(g, u, -0.00655751221f);
    g = M3D_BINOM_FMA(g, u, 0.0146817541f);
    g = M3D_BINOM_FMA(g, u, -0.0344220156f);
    g = M3D_BINOM_FMA(g, u, 0.098453992f);
    g = M3D_BINOM_FMA(g, u, -0.374889319f);
    g = M3D_BINOM_FMA(g, u, -0.876371303f);
    const float z = w + g;
    return M3D_BINOM_FMA(-(M3D_BINOM_FMA(z * z, z, -z)), rn, z);
#undef M3D_BINOM_FMA
}

M3D_HD float binom_sigma(float q)  // 4 sqrt(q)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 4.0f * __builtin_amdgcn_sqrtf(q);
#else
    return 4.0f * __builtin_sqrtf(q);
#endif
}

// Binomial(64 q, 1/2), 2 <= q <= kBinomQMax, from the 64-bit uniform (hi:lo).  `lds_guide` (device, optional): the
// small rows' guide staged in LDS by the calling kernel - the search then starts from the cell of v's top eight bits
// instead of the arithmetic quantile (a logarithm, two square roots, two reciprocals, a degree-12 polynomial: 50 of the
// draw's 75 vector instructions).  The cell brackets D between g0 and g1: equal - D is known without a threshold; one or
// two apart - the thresholds g0 and g0 + 1 (one 16-byte load) decide, and a threshold past g1 lies above the cell and so
// above v, no masking needed; wider cells (the far tails) and rows above kBinomGuideQHi take the arithmetic route.  The
// exact integers decide every draw either way.  (A guide in GLOBAL memory was measured in round 4 and lost on bright
// images - 64 lanes, 64 cache lines; LDS has no lines.  profiles/r04_split_chain.txt)
M3D_HD uint32_t binom_table_draw(uint32_t q, uint32_t hi, uint32_t lo, const BinomTables &t,
                                 const uint16_t *lds_guide = nullptr)
{
    const uint64_t u = ((uint64_t)hi << 32) | lo;
    const uint64_t v = u >> 1;
#if defined(__HIP_DEVICE_COMPILE__)
    if (lds_guide && q <= kBinomGuideQHi) {
        const uint32_t cell = lds_guide[(q - kBinomGuideQLo) * kBinomGuideCells + (hi >> (32 - kBinomGuideBits))];
        const uint32_t g0 = cell & 0xFFFu, spread = cell >> 12;
        if (spread <= 2u) {
            uint32_t d = g0;
            if (spread != 0u) {
                const BinomPair p = binom_load_pair(t.thr + (size_t)q * kBinomRowStride + 1 + g0);  // thresholds g0, g0 + 1
                d += (p.below <= v ? 1u : 0u) + (p.above <= v ? 1u : 0u);
            }
            const uint32_t m = 32u * q;
            return (u & 1u) ? m + d : m - d;
        }
    }
#endif
    // the upper tail mass 1 - P(D <= d) that v leaves, from the top 32 of its 63 bits (tails below 2^-33 start
    // the search in the wrong place and fall through to the bisection)
    const float tail = ((float)(0x7FFFFFFFu - (hi >> 1)) + 0.5f) * 0x1p-31f;
    const float fq = (float)q;
    const float guess = binom_sigma(fq) * binom_tail_quantile(tail, fq) + 0.5f;
    const uint64_t *T = t.thr + (size_t)q * kBinomRowStride;  // T[0] = 0, T[1 + i] = threshold i, then 2^64 - 1s
    int d = (int)(guess > 0.0f ? guess : 0.0f);               // D = #{i : threshold i <= v}, guessed (< 7 sigma)
    const BinomPair p = binom_load_pair(T + d);               // thresholds d - 1 and d
    if (p.below > v || p.above <= v) {                        // about one draw in 10^4: bisection over the row
        int a = 0, b = (int)t.rows[q].len;
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (T[1 + mid] <= v)
                a = mid + 1;
            else
                b = mid;
        }
        d = a;
    }
    const uint32_t m = 32u * q;
    return (u & 1u) ? m + (uint32_t)d : m - (uint32_t)d;
}

// Exact Binomial(n, 1/2) for voxel `voxel` at `iteration`.  kChunked = false: n < 65536 (uint16 images), one Philox
// block and at most one table draw; kChunked = true: any n, further chunks of 64 * 1023 from blocks 1, 2, ...
template <bool kChunked>
M3D_HD uint32_t binomial_half(uint32_t n, uint64_t voxel, uint32_t iteration, uint64_t seed, const BinomTables &t,
                              const uint16_t *lds_guide = nullptr)
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    U32x4 c = {(uint32_t)voxel, (uint32_t)(voxel >> 32), iteration, 0u};
    U32x4 r = philox4x32<kSplitRounds>(c, k0, k1);
    const int rb = (int)(n & 63u);
    uint32_t q = n >> 6;
#if defined(__HIP_DEVICE_COMPILE__)
    // the low rb bits of (y:x) as one 64-bit mask: seven instructions instead of the twelve of the two-word form below
    // (same bits, same count)
    const uint64_t keep = ~(~0ull << rb);  // rb < 64
    uint32_t acc = (uint32_t)(popc32(r.x & (uint32_t)keep) + popc32(r.y & (uint32_t)(keep >> 32)));
#else
    uint32_t acc = popc32(r.x & low_mask(rb > 32 ? 32 : rb)) + popc32(r.y & low_mask(rb > 32 ? rb - 32 : 0));
#endif
    const uint32_t q0 = q < kBinomQMax ? q : kBinomQMax;
    const uint32_t ones = (uint32_t)(popc32(r.z) + popc32(r.w));
    if (q0 >= 2u)
        acc += binom_table_draw(q0, r.z, r.w, t, lds_guide);
    else if (q0 == 1u)
        acc += ones;
    if (kChunked) {
        q -= q0;
        for (uint32_t j = 1; q; ++j) {  // uniform j: block (j + 1) / 2, words (x, y) for odd j, (z, w) for even j
            const uint32_t qc = q < kBinomQMax ? q : kBinomQMax;
            if (j & 1u) {
                c.w = (j + 1u) >> 1;
                r = philox4x32<kSplitRounds>(c, k0, k1);
            }
            const uint32_t hi = (j & 1u) ? r.x : r.z, lo = (j & 1u) ? r.y : r.w;
            acc += qc == 1u ? (uint32_t)(popc32(hi) + popc32(lo)) : binom_table_draw(qc, hi, lo, t, lds_guide);
            q -= qc;
        }
    }
    return acc;
}

// ---------------------------------------------------------------------------------------- table construction (host)
namespace binom_detail {

struct U256 {  // little-endian limbs
    uint64_t w[4];
};
inline void mul_small(U256 &a, uint32_t m)
{
    unsigned __int128 carry = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned __int128 p = (unsigned __int128)a.w[i] * m + carry;
        a.w[i] = (uint64_t)p;
        carry = p >> 64;
    }
}
inline void div_small(U256 &a, uint32_t d)  // floor
{
    unsigned __int128 rem = 0;
    for (int i = 3; i >= 0; --i) {
        const unsigned __int128 cur = (rem << 64) | a.w[i];
        a.w[i] = (uint64_t)(cur / d);
        rem = cur % d;
    }
}
inline void add(U256 &a, const U256 &b)
{
    unsigned __int128 carry = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned __int128 s = (unsigned __int128)a.w[i] + b.w[i] + carry;
        a.w[i] = (uint64_t)s;
        carry = s >> 64;
    }
}
inline void add_u64(U256 &a, uint64_t v)
{
    unsigned __int128 carry = v;
    for (int i = 0; i < 4 && carry; ++i) {
        const unsigned __int128 s = (unsigned __int128)a.w[i] + carry;
        a.w[i] = (uint64_t)s;
        carry = s >> 64;
    }
}
// value >> 128, saturated to 2^64 - 1 (the sums stay below 2^192)
inline uint64_t top_after_128(const U256 &a) { return a.w[3] ? ~0ull : a.w[2]; }

}  // namespace binom_detail

struct BinomTablesHost {
    std::vector<uint64_t> thr;
    std::vector<BinomRow> rows;
    std::vector<uint16_t> guide;  // [kBinomGuideEntries], derived from thr (see BinomTables)
    bool certified = false;  // every threshold's error interval fell inside one integer
    // (the host searches with the arithmetic guess; the guide is a device short cut to the same integer)
    BinomTables view() const { return BinomTables{thr.data(), rows.data(), nullptr}; }
};

// Fixed point: x = floor(X * 2^191) for a probability X; thresholds are (sum >> 128).
inline BinomTablesHost binom_tables_build()
{
    using namespace binom_detail;
    BinomTablesHost out;
    out.rows.assign(kBinomQMax + 1, BinomRow{0, 0});
    out.thr.assign((size_t)(kBinomQMax + 1) * kBinomRowStride, ~0ull);
    bool ok = true;
    U256 central = {{0, 0, 1ull << 63, 0}};  // C(0, 0) / 2^0 = 1 = 2^191 / 2^191
    uint32_t k = 0;                          // central = floor-chain of C(2k, k) / 4^k, error < k ulps
    for (uint32_t q = 1; q <= kBinomQMax; ++q) {
        const uint32_t m = 32u * q;
        for (; k < m;) {
            ++k;
            mul_small(central, 2u * k - 1u);
            div_small(central, 2u * k);
        }
        if (q < 2) continue;
        BinomRow &row = out.rows[q];
        row.off = q * kBinomRowStride;
        uint64_t *dst = out.thr.data() + row.off;
        dst[0] = 0;  // leading sentinel: "threshold -1" is below every v; the row's tail stays 2^64 - 1
        uint32_t count = 0;
        U256 term = central, sum = central;
        uint64_t err = m;  // sum true value in [sum, sum + err)
        for (uint32_t i = 0;; ++i) {
            if (i > 0) {
                mul_small(term, m - i + 1u);
                div_small(term, m + i);  // error < m + i ulps
                add(sum, term);
                add(sum, term);
                err += 2ull * (m + i);
            }
            const uint64_t lo = top_after_128(sum);
            if (lo >= kBinomVMax || i >= m) break;  // floor(2^63 P(D <= i)) >= 2^63 - 1: never <= v, not stored
            U256 upper = sum;
            add_u64(upper, err);
            if (top_after_128(upper) != lo) ok = false;
            if (count + 3u > kBinomRowStride) {
                ok = false;
                break;
            }
            dst[1 + count++] = lo;
        }
        row.len = count;
    }
    // the small rows' guide: counts of thresholds at the cells' lower ends, from the certified integers
    out.guide.assign(kBinomGuideEntries, 0);
    for (uint32_t q = kBinomGuideQLo; q <= kBinomGuideQHi; ++q) {
        const uint64_t *T = out.thr.data() + out.rows[q].off + 1;  // threshold i at T[i]
        const uint32_t len = out.rows[q].len;
        uint16_t *g = out.guide.data() + (size_t)(q - kBinomGuideQLo) * kBinomGuideCells;
        uint32_t count = 0, prev = 0;  // #{thresholds <= lower end of cell k}: monotone in k
        for (uint32_t k = 0; k <= kBinomGuideCells; ++k) {
            uint32_t c = len;
            if (k < kBinomGuideCells) {
                const uint64_t lo_end = (uint64_t)k << (63 - kBinomGuideBits);
                while (count < len && T[count] <= lo_end) ++count;
                c = count;
            }
            if (k > 0) {
                const uint32_t spread = c - prev;
                if (prev > 0xFFFu) ok = false;
                g[k - 1] = (uint16_t)((prev & 0xFFFu) | ((spread < 3u ? spread : 3u) << 12));
            }
            prev = c;
        }
    }
    out.certified = ok;
    return out;
}

inline const BinomTablesHost &binom_tables_host()
{
    static const BinomTablesHost tables = binom_tables_build();
    return tables;
}
}  // namespace m3d
