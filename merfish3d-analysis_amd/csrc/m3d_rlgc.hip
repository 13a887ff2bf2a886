// RLGC (Richardson-Lucy gradient consensus) deconvolution for MI355X / gfx950.
//
// Device-resident restatement of utils/rlgc.py:497-757 of the reference: one call uploads
// nothing and downloads nothing; per iteration the host reads back one small status
// record.  Data layout in HBM (all Z,Y,X C-contiguous, X fastest):
//
//   image      u16|f32 [Z,Y,X]        caller buffer, never padded: observed_image =
//                                     image * mask (rlgc.py:588) kills the symmetric pad,
//                                     so the padded copy of rlgc.py:571-576 is never built
//   otf        c64 [Zp,Yp,Xp/2+1]     rfftn of the embedded PSF; conj / |.|^2 formed on the
//                                     fly (rlgc.py:582-584 keeps three arrays)
//   update_norm (kept as 1 / update_norm), reconA, reconB, hu, r1, r2, w1, w2   f32 [Zp,Yp,Xp]
//   cbuf       c64 [Zp,Yp,Xp/2+1]     spectrum scratch (tiled layout of m3d_fftconv.hip; plain for the rocFFT path)
//
// Per iteration (rlgc.py:616-736) the ~40 CuPy elementwise/reduction launches become five
// fused kernels around four FFT convolutions.  This is the unfused form (rocFFT path, or M3D_FFT_PRUNE=0);
// with the fused FFT engine the consensus / update arithmetic runs inside the X inverse passes (ConvEpilogue,
// m3d_fftconv.h) and k_consensus_in, k_update, k_boundary are not launched:
//   conv(recon, otf) -> hu
//   k_split_ratio : Philox split regenerated in registers, KL partial sums for the three
//                   divergences, both ratio volumes written (core region only)
//   conv(r1, otf^*) -> w1 ; conv(r2, otf^*) -> w2
//   k_consensus_in: HT1, HT2 = w * (1 / update_norm); (HT1-1)(HT2-1) -> hu ; HT1+HT2 -> w1 (core)
//   conv(hu, |otf|^2) -> w2
//   k_update      : gated multiplicative update on the core + stop-test reductions
//   k_boundary    : symmetric-extension gather into the pad region (the three
//                   axis-sequential passes of rlgc.py:248-271 compose to one gather)
// Elementwise kernels are HBM-bound: rows are walked by one wave each (64 lanes x dword,
// 256-B coalesced segments), grid-stride over rows, reductions wave-shuffle -> LDS ->
// per-block partial -> single-block deterministic finalize.

#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#include "m3d_common.h"
#include "m3d_fftconv.h"
#include "m3d_index.h"
#include "m3d_binom.h"
#include "m3d_philox.h"

using namespace m3d;

namespace {

struct Geo {
    int Z, Y, X;     // observed image
    int Zp, Yp, Xp;  // padded (FFT) volume
    int bz, by, bx;  // pad before
    int Xc;          // Xp/2+1
    int64_t N, Np, Nc;
};

struct KlPartial {
    double A, Bim, B1, B2, Tim, T1, T2;
};
struct UpdPartial {
    unsigned long long updated;
    float recon_max, max_abs_delta;
};
struct MinMaxPartial {
    float mn, mx;
};

struct Status {  // one record read back per iteration
    double A, Bim, B1, B2, Tim, T1, T2;
    unsigned long long updated;
    float recon_max, max_abs_delta;
    float ht_min, ht_max;
    float mean;
    float pad;
    double psf_sum;
};

constexpr int kRowsPerBlock = 4;  // blockDim = (64, 4): one wave per row
constexpr int kRowGrid = 2048;

inline int row_grid(int64_t rows)
{
    int64_t g = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
    if (g < 1) g = 1;
    if (g > kRowGrid) g = kRowGrid;
    return (int)g;
}

template <typename T>
__device__ __forceinline__ float load_as_f32(const T *p, int64_t i);
template <>
__device__ __forceinline__ float load_as_f32<uint16_t>(const uint16_t *p, int64_t i)
{
    return (float)p[i];
}
template <>
__device__ __forceinline__ float load_as_f32<float>(const float *p, int64_t i)
{
    return p[i];
}

// ---------------------------------------------------------------- setup kernels
__global__ void k_fill_from(float *__restrict__ dst, const float *__restrict__ value, int64_t n)
{
    const float v = *value;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = v;
}

__global__ void k_psf_sum(const float *__restrict__ psf, int64_t n, Status *st)
{
    __shared__ double sm[4];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += (double)psf[i];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) st->psf_sum = acc;
}

// pad_psf (rlgc.py:274-313): embed + 6 rolls + ifftshift == place psf[k] at (k - p/2) mod n.
__global__ void k_psf_embed(const float *__restrict__ psf, float *__restrict__ dst, int pz, int py,
                            int px, Geo g, const Status *st, int normalize)
{
    const int64_t n = (int64_t)pz * py * px;
    float s = (float)st->psf_sum;
    if (!normalize || s == 0.0f) s = 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int x = (int)(i % px);
        int64_t t = i / px;
        int y = (int)(t % py);
        int z = (int)(t / py);
        int zz = ((z - pz / 2) % g.Zp + g.Zp) % g.Zp;
        int yy = ((y - py / 2) % g.Yp + g.Yp) % g.Yp;
        int xx = ((x - px / 2) % g.Xp + g.Xp) % g.Xp;
        // PSF supports larger than the volume wrap and accumulate (np.roll semantics keep
        // one value per cell only when the PSF fits, which rlgc geometry guarantees).
        dst[((int64_t)zz * g.Yp + yy) * g.Xp + xx] = psf[i] / s;
    }
}

__global__ void k_mask(float *__restrict__ dst, Geo g)
{
    const int64_t rows = (int64_t)g.Zp * g.Yp;
    for (int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.y; r < rows;
         r += (int64_t)gridDim.x * kRowsPerBlock) {
        const int z = (int)(r / g.Yp), y = (int)(r % g.Yp);
        const bool in_zy = z >= g.bz && z < g.bz + g.Z && y >= g.by && y < g.by + g.Y;
        float *row = dst + r * g.Xp;
        for (int x = threadIdx.x; x < g.Xp; x += 64)
            row[x] = (in_zy && x >= g.bx && x < g.bx + g.X) ? 1.0f : 0.0f;
    }
}

// update_norm is constant over a run and only ever divides: it is kept as its correctly rounded reciprocal, so
// the per-iteration HT = w / update_norm (rlgc.py:661,671) is one multiply (<= 1 ulp from the quotient, inside
// the FFT's own rounding).
__global__ void k_floor_reciprocal(float *__restrict__ p, float floor_v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        p[i] = rn_div(1.0f, fmaxf(p[i], floor_v));
}

template <typename T>
__global__ void k_image_sum(const T *__restrict__ img, int64_t n, double *__restrict__ partial)
{
    __shared__ double sm[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        acc += (double)load_as_f32(img, i);
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void k_image_mean_finalize(const double *__restrict__ partial, int nblocks, int64_t n,
                                      Status *st)
{
    __shared__ double sm[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) acc += partial[i];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) st->mean = (float)(acc / (double)n);
}

// ---------------------------------------------------------------- spectrum multiply
// mode 0: X*H   1: X*conj(H)   2: X*(H*conj(H))     then * scale (= 1/Np of irfftn)
__global__ void k_cmul(float2 *__restrict__ x, const float2 *__restrict__ h, int64_t n, int mode,
                       float scale)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float2 a = x[i];
        float2 b = h[i];
        if (mode == 1) {
            b.y = -b.y;
        } else if (mode == 2) {
            b.x = b.x * b.x + b.y * b.y;
            b.y = 0.0f;
        }
        float2 r;
        r.x = (a.x * b.x - a.y * b.y) * scale;
        r.y = (a.x * b.y + a.y * b.x) * scale;
        x[i] = r;
    }
}

// ---------------------------------------------------------------- iteration kernels
// rlgc.py:619-628 + 654-672 (core region): split, KL partial sums, both ratio volumes.
template <typename T>
__global__ void k_split_ratio(const float *__restrict__ hu, const T *__restrict__ image,
                              float *__restrict__ r1, float *__restrict__ r2, Geo g,
                              uint32_t iteration, uint64_t seed, BinomTables tables,
                              const float *__restrict__ injected, KlPartial *__restrict__ partial)
{
    __shared__ double sm[4];
    // the sampler's guide for the rows of dim voxels, staged in LDS (m3d_binom.h; uint16 images: float images keep the
    // arithmetic route of their chunked draws)
    __shared__ uint16_t lguide[kBinomGuideEntries];
    const bool use_guide = sizeof(T) == 2 && !injected;
    if (use_guide) {
        for (uint32_t i = threadIdx.y * blockDim.x + threadIdx.x; i < kBinomGuideEntries; i += blockDim.x * blockDim.y)
            lguide[i] = tables.guide[i];
        __syncthreads();
    }
    const float eps = 1e-4f;
    // per-thread partial sums in float32 (<= a few hundred terms each), promoted to double
    // for the cross-thread reduction: keeps the kernel HBM-bound instead of f64-VALU-bound
    float A = 0, Bim = 0, B1 = 0, B2 = 0, Tim = 0, T1 = 0, T2 = 0;
    double dA = 0, dBim = 0, dB1 = 0, dB2 = 0, dTim = 0, dT1 = 0, dT2 = 0;
    const int64_t rows = (int64_t)g.Z * g.Y;
    for (int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.y; r < rows;
         r += (int64_t)gridDim.x * kRowsPerBlock) {
        const int z = (int)(r / g.Y), y = (int)(r % g.Y);
        const int64_t prow = ((int64_t)(z + g.bz) * g.Yp + (y + g.by)) * g.Xp + g.bx;
        const int64_t irow = r * g.X;
        // loads first, four row segments at a time (independent requests in flight), then the arithmetic
        // in the original element order
        constexpr int kU = 4;
        for (int x0 = threadIdx.x; x0 < g.X; x0 += 64 * kU) {
            float hbuf[kU], vbuf[kU], ibuf[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int xc = min(x0 + 64 * u, g.X - 1);
                hbuf[u] = hu[prow + xc];
                vbuf[u] = load_as_f32(image, irow + xc);
                ibuf[u] = injected ? injected[irow + xc] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int x = x0 + 64 * u;
                if (x >= g.X) continue;
                const float h = hbuf[u];
                const float v = vbuf[u];
                float s1;
                if (injected) {
                    s1 = ibuf[u];
                } else {
                    const float vt = v > 0.0f ? v : 0.0f;
                    // astype(int64) truncation, rlgc.py:619; counts above 2^20 are clamped for
                    // the draw only (bounds the chunk loop; u16 data never reaches it)
                    const uint32_t n = (uint32_t)fminf(vt, 1048576.0f);
                    // a uint16 count needs one Philox block and one table row at most
                    s1 = (float)binomial_half<!(sizeof(T) == 2)>(n, (uint64_t)(irow + x), iteration, seed, tables,
                                                                 use_guide ? lguide : nullptr);
                }
                const float s2 = v - s1;
                // kl_div, rlgc.py:401-412, as raw sums (normalisers applied on the host)
                const float a = h + eps;
                const float bim = v + eps, b1 = s1 + eps, b2 = s2 + eps;
                A += a;
                Bim += bim;
                B1 += b1;
                B2 += b2;
                if (a > 0.0f) {
                    // base-2 logarithms straight from v_log_f32 (arguments >= 1e-4: no denormal path); the sums are
                    // scaled by ln 2 once, in double, when the block's partials are written
                    const float la = __builtin_amdgcn_logf(a);
                    Tim += a * (la - __builtin_amdgcn_logf(bim));
                    T1 += a * (la - __builtin_amdgcn_logf(b1));
                    T2 += a * (la - __builtin_amdgcn_logf(b2));
                }
                // mask * split / (0.5 * (Hu + 1e-12)), rlgc.py:654-668: one correctly rounded reciprocal serves both
                // quotients (<= 1 ulp from the two divisions, below the rounding of the transforms that consume them)
                const float inv = rn_div(1.0f, 0.5f * (h + 1e-12f));
                r1[prow + x] = s1 * inv;
                r2[prow + x] = s2 * inv;
            }
        }
        // fold the row's float32 partials into double every row (bounded error growth)
        dA += A; dBim += Bim; dB1 += B1; dB2 += B2; dTim += Tim; dT1 += T1; dT2 += T2;
        A = Bim = B1 = B2 = Tim = T1 = T2 = 0.0f;
    }
    // flatten (64,4) block for the reductions
    dA = block_sum(dA, sm);
    dBim = block_sum(dBim, sm);
    dB1 = block_sum(dB1, sm);
    dB2 = block_sum(dB2, sm);
    dTim = block_sum(dTim, sm);
    dT1 = block_sum(dT1, sm);
    dT2 = block_sum(dT2, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const double ln2 = 0.69314718055994530942;
        KlPartial p = {dA, dBim, dB1, dB2, dTim * ln2, dT1 * ln2, dT2 * ln2};
        partial[blockIdx.x] = p;
    }
}

// First level for the fused split kernel's one-partial-per-workgroup array (65 536 records at the headline size): block
// b sums records b, b + gridDim.x, ... in a fixed order into record b of `out`; k_kl_finalize folds those.
__global__ void k_kl_reduce(const KlPartial *__restrict__ partial, int n, KlPartial *__restrict__ out)
{
    __shared__ double sm[4];
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x + (int)gridDim.x * (int)threadIdx.x; i < n; i += (int)gridDim.x * (int)blockDim.x) {
        const KlPartial p = partial[i];
        v[0] += p.A;
        v[1] += p.Bim;
        v[2] += p.B1;
        v[3] += p.B2;
        v[4] += p.Tim;
        v[5] += p.T1;
        v[6] += p.T2;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = block_sum(v[k], sm);
    if (threadIdx.x == 0) {
        const KlPartial p = {v[0], v[1], v[2], v[3], v[4], v[5], v[6]};
        out[blockIdx.x] = p;
    }
}

__global__ void k_kl_finalize(const KlPartial *__restrict__ partial, int nblocks, Status *st)
{
    __shared__ double sm[4];
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        const KlPartial p = partial[i];
        v[0] += p.A;
        v[1] += p.Bim;
        v[2] += p.B1;
        v[3] += p.B2;
        v[4] += p.Tim;
        v[5] += p.T1;
        v[6] += p.T2;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = block_sum(v[k], sm);
    if (threadIdx.x == 0) {
        st->A = v[0];
        st->Bim = v[1];
        st->B1 = v[2];
        st->B2 = v[3];
        st->Tim = v[4];
        st->T1 = v[5];
        st->T2 = v[6];
    }
}

// rlgc.py:661,671,674,678: HT = w / update_norm; product -> prod (full volume, it feeds
// the consensus FFT); HT1+HT2 kept only where k_update reads it (core), in place over w1.
__global__ void k_consensus_in(float *__restrict__ w1, const float *__restrict__ w2,
                               const float *__restrict__ un, float *__restrict__ prod, Geo g,
                               MinMaxPartial *__restrict__ partial)
{
    __shared__ float smf[4];
    float mn = INFINITY, mx = -INFINITY;
    const int64_t rows = (int64_t)g.Zp * g.Yp;
    for (int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.y; r < rows;
         r += (int64_t)gridDim.x * kRowsPerBlock) {
        const int z = (int)(r / g.Yp), y = (int)(r % g.Yp);
        const bool in_zy = z >= g.bz && z < g.bz + g.Z && y >= g.by && y < g.by + g.Y;
        const int64_t base = r * g.Xp;
        for (int x = threadIdx.x; x < g.Xp; x += 64) {
            const float u = un[base + x];
            const float h1 = w1[base + x] * u;  // u = 1 / update_norm
            const float h2 = w2[base + x] * u;
            prod[base + x] = (h1 - 1.0f) * (h2 - 1.0f);
            const float hs = h1 + h2;
            mn = fminf(mn, hs);
            mx = fmaxf(mx, hs);
            if (in_zy && x >= g.bx && x < g.bx + g.X) w1[base + x] = hs;
        }
    }
    mn = block_min(mn, smf);
    mx = block_max(mx, smf);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        MinMaxPartial p = {mn, mx};
        partial[blockIdx.x] = p;
    }
}

// filter_update (rlgc.py:23-31, 682) + stop-test reductions (rlgc.py:685-694), core only.
__global__ void k_update(const float *__restrict__ consensus, const float *__restrict__ recon,
                         const float *__restrict__ hsum, float *__restrict__ recon_next, Geo g,
                         UpdPartial *__restrict__ partial)
{
    __shared__ float smf[4];
    __shared__ unsigned long long smu[4];
    unsigned long long updated = 0;
    float rmax = -INFINITY, dmax = 0.0f;
    const int64_t rows = (int64_t)g.Z * g.Y;
    for (int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.y; r < rows;
         r += (int64_t)gridDim.x * kRowsPerBlock) {
        const int z = (int)(r / g.Y), y = (int)(r % g.Y);
        const int64_t prow = ((int64_t)(z + g.bz) * g.Yp + (y + g.by)) * g.Xp + g.bx;
        for (int x = threadIdx.x; x < g.X; x += 64) {
            const float c = consensus[prow + x];
            const float old = recon[prow + x];
            const bool skip = c < 0.0f;
            const float nw = skip ? old : old * hsum[prow + x];
            recon_next[prow + x] = nw;
            updated += (c >= 0.0f) ? 1ull : 0ull;
            rmax = fmaxf(rmax, nw);
            dmax = fmaxf(dmax, fabsf(nw - old));
        }
    }
    updated = block_sum(updated, smu);
    rmax = block_max(rmax, smf);
    dmax = block_max(dmax, smf);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        UpdPartial p = {updated, rmax, dmax};
        partial[blockIdx.x] = p;
    }
}

__global__ void k_update_finalize(const UpdPartial *__restrict__ up, int n_up,
                                  const MinMaxPartial *__restrict__ mm, int n_mm, Status *st)
{
    __shared__ float smf[4];
    __shared__ unsigned long long smu[4];
    unsigned long long updated = 0;
    float rmax = -INFINITY, dmax = 0.0f, mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < n_up; i += blockDim.x) {
        updated += up[i].updated;
        rmax = fmaxf(rmax, up[i].recon_max);
        dmax = fmaxf(dmax, up[i].max_abs_delta);
    }
    for (int i = threadIdx.x; i < n_mm; i += blockDim.x) {
        mn = fminf(mn, mm[i].mn);
        mx = fmaxf(mx, mm[i].mx);
    }
    updated = block_sum(updated, smu);
    rmax = block_max(rmax, smf);
    dmax = block_max(dmax, smf);
    mn = block_min(mn, smf);
    mx = block_max(mx, smf);
    if (threadIdx.x == 0) {
        st->updated = updated;
        st->recon_max = rmax;
        st->max_abs_delta = dmax;
        st->ht_min = mn;
        st->ht_max = mx;
    }
}

// Fused-tail variant (ConvEpilogue in m3d_fftconv.h): the X inverse passes of the last two
// convolutions leave min / max / count as atomics; this reads them into the status record and
// re-arms them for the next iteration.
__device__ __forceinline__ void arm_fused_reductions(unsigned int *cons, unsigned int *upd)
{
    cons[0] = m3d::float_key(INFINITY);   // min HT1 + HT2
    cons[1] = m3d::float_key(-INFINITY);  // max HT1 + HT2
    upd[0] = 0u;                          // updated voxels (low / high word)
    upd[1] = 0u;
    upd[2] = m3d::float_key(-INFINITY);   // max of the new estimate
    upd[3] = 0u;                          // max |delta| (non-negative: raw bits order like floats)
}
// one thread per record (m3d::kRedSlots of them, m3d_fftconv.h)
__global__ void k_arm_fused_reductions(unsigned int *cons, unsigned int *upd)
{
    const int s = threadIdx.x;
    if (blockIdx.x == 0 && s < m3d::kRedSlots) arm_fused_reductions(cons + s * m3d::kRedStride, upd + s * m3d::kRedStride);
}
__global__ void k_fused_finalize(unsigned int *cons, unsigned int *upd, Status *st)
{
    __shared__ unsigned long long smu[4];
    __shared__ float smf[4];
    if (blockIdx.x != 0) return;
    const int s = threadIdx.x;
    const bool live = s < m3d::kRedSlots;
    unsigned int *c = cons + (live ? s : 0) * m3d::kRedStride, *u = upd + (live ? s : 0) * m3d::kRedStride;
    unsigned long long updated = live ? *(const unsigned long long *)u : 0ull;
    float recon_max = live ? m3d::key_float(u[2]) : -INFINITY;
    float delta = live ? __uint_as_float(u[3]) : 0.0f;
    float ht_min = live ? m3d::key_float(c[0]) : INFINITY;
    float ht_max = live ? m3d::key_float(c[1]) : -INFINITY;
    __syncthreads();
    if (live) arm_fused_reductions(c, u);
    updated = block_sum(updated, smu);
    recon_max = block_max(recon_max, smf);
    delta = block_max(delta, smf);
    ht_min = block_min(ht_min, smf);
    ht_max = block_max(ht_max, smf);
    if (s == 0) {
        st->updated = updated;
        st->recon_max = recon_max;
        st->max_abs_delta = delta;
        st->ht_min = ht_min;
        st->ht_max = ht_max;
    }
}

// order-independent 64-bit fingerprint of the PSF values (position-salted splitmix64 of the bit patterns)
__global__ void k_psf_hash(const float *__restrict__ psf, int64_t n, unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)__float_as_uint(psf[i]) ^ ((unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull);
        x ^= x >> 30;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 27;
        x *= 0x94D049BB133111EBull;
        x ^= x >> 31;
        acc += x;
    }
    atomicAdd(out, acc);
}

// enforce_symmetric_boundary (rlgc.py:230-271): z, then y, then x passes compose to
// recon[z,y,x] = recon[s(z), s(y), s(x)] with every source inside the core.
__global__ void k_boundary(float *__restrict__ recon, Geo g)
{
    const int64_t rows = (int64_t)g.Zp * g.Yp;
    for (int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.y; r < rows;
         r += (int64_t)gridDim.x * kRowsPerBlock) {
        const int z = (int)(r / g.Yp), y = (int)(r % g.Yp);
        const int sz = sym_src(z, g.bz, g.Z), sy = sym_src(y, g.by, g.Y);
        const bool row_is_core = (sz == z) && (sy == y);
        const float *src = recon + ((int64_t)sz * g.Yp + sy) * g.Xp;
        float *dst = recon + r * g.Xp;
        for (int x = threadIdx.x; x < g.Xp; x += 64) {
            const int sx = sym_src(x, g.bx, g.X);
            if (row_is_core && sx == x) continue;
            dst[x] = src[sx];
        }
    }
}

__global__ void k_crop(const float *__restrict__ recon, float *__restrict__ out, Geo g)
{
    const int64_t rows = (int64_t)g.Z * g.Y;
    for (int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.y; r < rows;
         r += (int64_t)gridDim.x * kRowsPerBlock) {
        const int z = (int)(r / g.Y), y = (int)(r % g.Y);
        const int64_t prow = ((int64_t)(z + g.bz) * g.Yp + (y + g.by)) * g.Xp + g.bx;
        for (int x = threadIdx.x; x < g.X; x += 64) out[r * g.X + x] = recon[prow + x];
    }
}

template <typename T>
__global__ void k_debug_split(const T *__restrict__ image, int64_t n, uint64_t seed,
                              uint32_t iteration, BinomTables tables, float *__restrict__ split1)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float v = load_as_f32(image, i);
        const float vt = v > 0.0f ? v : 0.0f;
        split1[i] = (float)binomial_half<!(sizeof(T) == 2)>((uint32_t)fminf(vt, 1048576.0f), (uint64_t)i, iteration,
                                                            seed, tables);
    }
}

__global__ void k_debug_table_draw(const uint32_t *__restrict__ q, const uint32_t *__restrict__ hi,
                                   const uint32_t *__restrict__ lo, int64_t n, BinomTables tables,
                                   uint32_t *__restrict__ out)
{
    // the small rows' guide staged in LDS, as the fused split kernel does: rows 2 .. kBinomGuideQHi take the guide route,
    // the others the arithmetic one - tests/test_gpu_rlgc.py probes on, below and above every threshold of both kinds
    __shared__ uint16_t lg[kBinomGuideEntries];
    for (uint32_t i = threadIdx.x; i < kBinomGuideEntries; i += blockDim.x) lg[i] = tables.guide[i];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = binom_table_draw(q[i], hi[i], lo[i], tables, lg);
}

__global__ void k_clip_u16(const float *__restrict__ in, uint16_t *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v = in[i];
        v = fminf(fmaxf(v, 0.0f), 65535.0f);  // np.clip; NaN -> 0 through fmaxf
        out[i] = (uint16_t)v;                 // astype(uint16): truncation
    }
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// The split sampler's threshold tables (m3d_binom.h: 1024 rows of exact integer thresholds, 9.6 MB): built once per
// process on the host, one copy per device, kept for the life of the process.  `on` (optional): a device pointer of the
// caller (the image): the tables are taken for the device that owns it, which must be the current one - a kernel
// launched on this device's stream with another device's table pointer would fault.
int device_binom_tables(BinomTables *out, const void *on = nullptr)
{
    static std::mutex mu;
    static BinomTables per_device[64] = {};
    int dev = 0;
    M3D_HIP_CHECK(hipGetDevice(&dev));
    M3D_REQUIRE(dev >= 0 && dev < 64, "device index");
    if (on) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, on) == hipSuccess && attr.type == hipMemoryTypeDevice)
            M3D_REQUIRE(attr.device == dev, "the image lives on another device than the current one");
        else
            (void)hipGetLastError();  // (host / unregistered pointers: nothing to check)
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[dev].thr) {
        const BinomTablesHost &h = binom_tables_host();
        if (!h.certified) {
            set_err("binomial split tables", "a threshold could not be certified");
            return M3D_ERR_HIP;
        }
        uint64_t *thr = nullptr;
        BinomRow *rows = nullptr;
        uint16_t *guide = nullptr;
        hipError_t e = hipMalloc(&thr, h.thr.size() * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc(&rows, h.rows.size() * sizeof(BinomRow));
        if (e == hipSuccess) e = hipMalloc(&guide, h.guide.size() * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemcpy(thr, h.thr.data(), h.thr.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(rows, h.rows.data(), h.rows.size() * sizeof(BinomRow), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(guide, h.guide.data(), h.guide.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {  // nothing half-made stays behind
            if (thr) (void)hipFree(thr);
            if (rows) (void)hipFree(rows);
            if (guide) (void)hipFree(guide);
            set_err("binomial split tables", hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? M3D_ERR_OUT_OF_MEMORY : M3D_ERR_HIP;
        }
        per_device[dev] = BinomTables{thr, rows, guide};
    }
    *out = per_device[dev];
    return M3D_OK;
}

}  // namespace

// =========================================================================================
struct m3d_rlgc_plan {
    Geo g;
    int pz, py, px;
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info_f = nullptr, info_i = nullptr;
    size_t fft_work = 0;
    size_t spec_bytes = 0;
    size_t workspace = 0;
    // workspace offsets
    size_t o_otf2 = 0;  // |OTF|^2 as float (fused engine with the register Z pass), 0: not held
    size_t o_otf, o_cbuf, o_un, o_ra, o_rb, o_hu, o_r1, o_r2, o_w1, o_w2, o_part, o_status,
        o_fftwork;
    Status *status_host = nullptr;  // pinned
    std::vector<hipEvent_t> events;  // reused across runs when params->profile is set
    hipEvent_t loop_ev[2] = {nullptr, nullptr};
    m3d::FftConvPlan *fc = nullptr;  // fused in-LDS FFT convolution engine (default)
    void *zeroed_workspace = nullptr;  // spectrum tile padding already cleared in this buffer
    // OTF and update_norm depend only on the PSF (and the plan's geometry): a tile's bits share one or two
    // PSFs, so they are kept between runs on the same workspace and reused while the PSF fingerprint matches
    void *setup_workspace = nullptr;
    unsigned long long setup_psf_hash = 0;
    int setup_normalize = -1;
};

#define M3D_ROCFFT_CHECK(expr)                                   \
    do {                                                         \
        rocfft_status _s = (expr);                               \
        if (_s != rocfft_status_success) {                       \
            char _b[64];                                         \
            snprintf(_b, 64, "rocfft_status %d", (int)_s);       \
            m3d::set_err(#expr, _b);                             \
            return M3D_ERR_ROCFFT;                               \
        }                                                        \
    } while (0)

static int g_fft_engine = 0;  // 0: fused in-LDS engine (rocFFT only as fallback), 1: rocFFT

extern "C" {

int m3d_version(void) { return 102; }

int m3d_set_fft_engine(int engine)
{
    M3D_REQUIRE(engine == 0 || engine == 1, "engine must be 0 (fused) or 1 (rocFFT)");
    g_fft_engine = engine;
    return M3D_OK;
}

int m3d_rlgc_plan_uses_rocfft(const m3d_rlgc_plan *p) { return (p && !p->fc) ? 1 : 0; }

const char *m3d_last_error(void) { return err_buf(); }

const char *m3d_status_string(int s)
{
    switch (s) {
        case M3D_OK: return "ok";
        case M3D_ERR_INVALID_ARG: return "invalid argument";
        case M3D_ERR_OUT_OF_MEMORY: return "out of memory";
        case M3D_ERR_HIP: return "HIP runtime error";
        case M3D_ERR_ROCFFT: return "rocFFT error";
        case M3D_ERR_WORKSPACE: return "workspace too small";
        case M3D_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

int m3d_next_fft_size(int x)
{
    if (x <= 1) return 1;
    int n = x;
    for (;;) {
        int m = n;
        while (m % 2 == 0) m /= 2;
        while (m % 3 == 0) m /= 3;
        if (m == 1) return n;
        ++n;
    }
}

int m3d_rlgc_geometry(const int32_t image_zyx[3], const int32_t psf_zyx[3], int pad_yx,
                      int32_t padded_zyx[3], int32_t pad_before[3], int32_t pad_after[3])
{
    M3D_REQUIRE(image_zyx && psf_zyx && padded_zyx && pad_before && pad_after, "null pointer");
    for (int a = 0; a < 3; ++a) {
        M3D_REQUIRE(image_zyx[a] > 0 && psf_zyx[a] > 0, "shapes must be positive");
        if (a > 0 && !pad_yx) {
            pad_before[a] = pad_after[a] = 0;
            padded_zyx[a] = image_zyx[a];
            continue;
        }
        const int halo = psf_zyx[a] / 2;
        const int with_halo = image_zyx[a] + 2 * halo;
        const int nl = m3d_next_fft_size(with_halo);
        const int extra = nl - with_halo;
        pad_before[a] = halo + extra / 2;
        pad_after[a] = halo + extra - extra / 2;
        padded_zyx[a] = nl;
    }
    return M3D_OK;
}

int m3d_rlgc_plan_create(const int32_t image_zyx[3], const int32_t psf_zyx[3], int pad_yx,
                         m3d_rlgc_plan **out)
{
    M3D_REQUIRE(out, "null plan pointer");
    int32_t padded[3], pb[3], pa[3];
    int rc = m3d_rlgc_geometry(image_zyx, psf_zyx, pad_yx, padded, pb, pa);
    if (rc != M3D_OK) return rc;
    for (int a = 0; a < 3; ++a)
        M3D_REQUIRE(psf_zyx[a] <= padded[a], "PSF support exceeds the padded volume");

    m3d_rlgc_plan *p = new (std::nothrow) m3d_rlgc_plan();
    if (!p) return M3D_ERR_OUT_OF_MEMORY;
    Geo &g = p->g;
    g.Z = image_zyx[0];
    g.Y = image_zyx[1];
    g.X = image_zyx[2];
    g.Zp = padded[0];
    g.Yp = padded[1];
    g.Xp = padded[2];
    g.bz = pb[0];
    g.by = pb[1];
    g.bx = pb[2];
    g.Xc = g.Xp / 2 + 1;
    g.N = (int64_t)g.Z * g.Y * g.X;
    g.Np = (int64_t)g.Zp * g.Yp * g.Xp;
    g.Nc = (int64_t)g.Zp * g.Yp * g.Xc;
    p->pz = psf_zyx[0];
    p->py = psf_zyx[1];
    p->px = psf_zyx[2];

    // FFT engine: the fused in-LDS convolution unless rocFFT is requested (m3d_set_fft_engine)
    // or a padded length cannot be handled (then rocFFT is the library fallback).
    if (g_fft_engine != 1) {
        rc = m3d::fftconv_create(g.Zp, g.Yp, g.Xp, &p->fc);
        if (rc != M3D_OK && rc != M3D_ERR_UNSUPPORTED) {
            if (rc == M3D_ERR_OUT_OF_MEMORY) set_err("fftconv_create", "table allocation failed");
            m3d_rlgc_plan_destroy(p);
            return rc;
        }
    }
    if (!p->fc) {
        static bool setup_done = false;
        if (!setup_done) {
            rocfft_setup();
            setup_done = true;
        }
        const size_t lengths[3] = {(size_t)g.Xp, (size_t)g.Yp, (size_t)g.Zp};
        rocfft_status s;
        s = rocfft_plan_create(&p->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                               rocfft_precision_single, 3, lengths, 1, nullptr);
        if (s == rocfft_status_success)
            s = rocfft_plan_create(&p->inv, rocfft_placement_notinplace,
                                   rocfft_transform_type_real_inverse, rocfft_precision_single, 3,
                                   lengths, 1, nullptr);
        size_t wf = 0, wi = 0;
        if (s == rocfft_status_success) s = rocfft_plan_get_work_buffer_size(p->fwd, &wf);
        if (s == rocfft_status_success) s = rocfft_plan_get_work_buffer_size(p->inv, &wi);
        if (s == rocfft_status_success) s = rocfft_execution_info_create(&p->info_f);
        if (s == rocfft_status_success) s = rocfft_execution_info_create(&p->info_i);
        if (s != rocfft_status_success) {
            char b[64];
            snprintf(b, 64, "rocfft_status %d", (int)s);
            set_err("rocfft plan creation", b);
            m3d_rlgc_plan_destroy(p);
            return M3D_ERR_ROCFFT;
        }
        p->fft_work = wf > wi ? wf : wi;
    }

    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    const size_t fb = sizeof(float) * (size_t)g.Np;
    const size_t cb = sizeof(float2) * (size_t)(p->fc ? m3d::fftconv_spectrum_elems(p->fc) : g.Nc);
    p->spec_bytes = cb;
    p->o_otf = take(cb);
    p->o_cbuf = take(cb);
    p->o_un = take(fb);
    p->o_ra = take(fb);
    p->o_rb = take(fb);
    p->o_hu = take(fb);
    p->o_r1 = take(fb);
    p->o_r2 = take(fb);
    p->o_w1 = take(fb);
    p->o_w2 = take(fb > cb ? fb : cb);  // doubles as the second scratch spectrum of the dual X inverse pass
    if (p->fc && m3d::fftconv_real_otf_supported(p->fc)) p->o_otf2 = take(cb / 2);
    p->o_part = take(sizeof(KlPartial) * (size_t)(kRowGrid + kMaxGrid) * 2);
    p->o_status = take(sizeof(Status));
    p->o_fftwork = take(p->fft_work ? p->fft_work : 256);
    p->workspace = off;

    if (hipHostMalloc((void **)&p->status_host, sizeof(Status), hipHostMallocDefault) !=
        hipSuccess) {
        set_err("hipHostMalloc", "status record");
        m3d_rlgc_plan_destroy(p);
        return M3D_ERR_HIP;
    }
    *out = p;
    return M3D_OK;
}

size_t m3d_rlgc_plan_workspace_bytes(const m3d_rlgc_plan *p) { return p ? p->workspace : 0; }

int m3d_rlgc_plan_destroy(m3d_rlgc_plan *p)
{
    if (!p) return M3D_OK;
    if (p->info_f) rocfft_execution_info_destroy(p->info_f);
    if (p->info_i) rocfft_execution_info_destroy(p->info_i);
    if (p->fwd) rocfft_plan_destroy(p->fwd);
    if (p->inv) rocfft_plan_destroy(p->inv);
    if (p->status_host) (void)hipHostFree(p->status_host);
    m3d::fftconv_destroy(p->fc);
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->loop_ev)
        if (e) (void)hipEventDestroy(e);
    delete p;
    return M3D_OK;
}

int m3d_rlgc_run(m3d_rlgc_plan *p, const void *image, int image_dtype, const float *psf,
                 float *out, void *workspace, size_t workspace_bytes,
                 const m3d_rlgc_params *params, const float *injected, int n_injected,
                 m3d_rlgc_result *res, void *stream_v)
{
    M3D_REQUIRE(p && image && psf && out && workspace && params, "null pointer");
    M3D_REQUIRE(image_dtype == M3D_U16 || image_dtype == M3D_F32, "image dtype must be u16 or f32");
    if (workspace_bytes < p->workspace) {
        set_err("workspace", "smaller than m3d_rlgc_plan_workspace_bytes");
        return M3D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream_v;
    BinomTables binom;
    if (int rc_tables = device_binom_tables(&binom, image)) return rc_tables;
    // (Round 3 could replay the loop body from hipGraphs - bit-equal and measured no faster at any size on MI355X / ROCm 7.2,
    // profiles/r03_graph_replay.txt: a launch-bound iteration waits for the dispatch latency between dependent kernels and
    // the stop test's read-back, neither of which a graph removes.  The path was taken out in round 4.)
    const Geo g = p->g;
    char *ws = (char *)workspace;
    float2 *otf = (float2 *)(ws + p->o_otf);
    float *otf2 = p->o_otf2 ? (float *)(ws + p->o_otf2) : nullptr;
    float2 *cbuf = (float2 *)(ws + p->o_cbuf);
    float *un = (float *)(ws + p->o_un);
    float *recon_cur = (float *)(ws + p->o_ra);
    float *recon_prev = (float *)(ws + p->o_rb);
    float *hu = (float *)(ws + p->o_hu);
    float *r1 = (float *)(ws + p->o_r1);
    float *r2 = (float *)(ws + p->o_r2);
    float *w1 = (float *)(ws + p->o_w1);
    float *w2 = (float *)(ws + p->o_w2);
    KlPartial *kl_part = (KlPartial *)(ws + p->o_part);
    UpdPartial *up_part = (UpdPartial *)(kl_part + kRowGrid);
    MinMaxPartial *mm_part = (MinMaxPartial *)(up_part + kRowGrid);
    double *sum_part = (double *)(mm_part + kRowGrid);
    Status *dstat = (Status *)(ws + p->o_status);
    void *fftwork = ws + p->o_fftwork;

    if (!p->fc) {
        M3D_ROCFFT_CHECK(rocfft_execution_info_set_stream(p->info_f, st));
        M3D_ROCFFT_CHECK(rocfft_execution_info_set_stream(p->info_i, st));
        if (p->fft_work) {
            M3D_ROCFFT_CHECK(rocfft_execution_info_set_work_buffer(p->info_f, fftwork, p->fft_work));
            M3D_ROCFFT_CHECK(rocfft_execution_info_set_work_buffer(p->info_i, fftwork, p->fft_work));
        }
    } else if (p->zeroed_workspace != workspace) {
        // tile padding of the two spectra (kx >= Xc in the last tile) must hold finite values
        M3D_HIP_CHECK(hipMemsetAsync(otf, 0, p->spec_bytes, st));
        M3D_HIP_CHECK(hipMemsetAsync(cbuf, 0, p->spec_bytes, st));
        M3D_HIP_CHECK(hipMemsetAsync(w2, 0, p->spec_bytes, st));  // second scratch spectrum (dual X inverse)
        p->zeroed_workspace = workspace;
    }
    const float inv_n = 1.0f / (float)g.Np;
    const dim3 rb(64, kRowsPerBlock);
    const int g_core = row_grid((int64_t)g.Z * g.Y);
    const int g_pad = row_grid((int64_t)g.Zp * g.Yp);

    auto fft_r2c = [&](float *src, float2 *dst) -> int {
        void *in[1] = {src}, *o[1] = {dst};
        M3D_ROCFFT_CHECK(rocfft_execute(p->fwd, in, o, p->info_f));
        return M3D_OK;
    };
    auto fft_c2r = [&](float2 *src, float *dst) -> int {
        void *in[1] = {src}, *o[1] = {dst};
        M3D_ROCFFT_CHECK(rocfft_execute(p->inv, in, o, p->info_i));
        return M3D_OK;
    };
    // fft_conv (rlgc.py:316-350) with H in {otf, conj(otf), |otf|^2}
    // optional HIP-event timing of every convolution (bench.py roofline; params->profile)
    m3d::EventLog elog;
    elog.events = &p->events;
    elog.stream = st;
    elog.enabled = (params->profile & 1) != 0;  // bit 0: events around every pass; bit 1: the loop's two events only
    const bool time_loop = params->profile != 0;
    auto stamp = [&](int kind) -> int {
        if (elog.mark(kind)) {
            set_err("hipEventRecord", "profiling event");
            return M3D_ERR_HIP;
        }
        return M3D_OK;
    };
    // Every convolution of the iteration either takes an input that is zero outside the observed
    // core (mask, ratio volumes) or produces an output that is only read inside it (Hu, consensus
    // map): the engine skips the rows / planes that carry nothing (M3D_FFT_PRUNE=0 disables).
    const bool prune = !(getenv("M3D_FFT_PRUNE") && atoi(getenv("M3D_FFT_PRUNE")) == 0);
    const m3d::ConvBox core = {g.bz, g.Z, g.by, g.Y, g.bx, g.X};
    // The estimate itself is only ever written on the core; its symmetric boundary (rlgc.py:230-271)
    // is gathered by the X pass (kMirrorIn) instead of being materialised by k_boundary.
    enum { kFull = 0, kInCore = 1, kOutCore = 2, kMirrorIn = 4 };
    const bool virtual_boundary = p->fc && prune;
    // ... and the elementwise kernels that follow the second ratio convolution and the consensus
    // convolution run inside those convolutions' X inverse store loops (ConvEpilogue).
    const bool fused_tail = p->fc && prune;
    // the fused tails' reduction records (m3d_fftconv.h) live in the partial-sum arrays of the unfused kernels
    static_assert(sizeof(MinMaxPartial) * kRowGrid >= sizeof(unsigned int) * m3d::kRedSlots * m3d::kRedStride,
                  "consensus reduction records do not fit the min/max partials they alias");
    static_assert(sizeof(UpdPartial) * kRowGrid >= sizeof(unsigned int) * m3d::kRedSlots * m3d::kRedStride,
                  "update reduction records do not fit the update partials they alias");
    unsigned int *cons_red = (unsigned int *)mm_part, *upd_red = (unsigned int *)up_part;
    auto conv = [&](float *src, int mode, float *dst, int support = kFull,
                    const m3d::ConvEpilogue *epi = nullptr, bool x_forward_done = false) -> int {
        ++elog.convolutions;
        if (p->fc)  // every kernel's time is attributed to its pass
        {
            // the consensus convolution (zero-extended input, output read on the core): |OTF|^2 from its float volume
            const bool real_h = mode == 2 && otf2 && prune && (support & kOutCore) && !(support & kMirrorIn);
            return m3d::fftconv_conv(p->fc, src, real_h ? (const float2 *)otf2 : otf, real_h ? 3 : mode, cbuf, dst, st,
                                     &elog,
                                     prune && (support & kInCore) ? &core : nullptr,
                                     prune && (support & kOutCore) ? &core : nullptr,
                                     prune && (support & kMirrorIn) ? &core : nullptr, epi,
                                     fused_tail && (support & kMirrorIn) ? (float2 *)w2 : nullptr, x_forward_done);
        }
        int rc = stamp(m3d::EventLog::kStart);
        if (rc) return rc;
        if ((rc = fft_r2c(src, cbuf))) return rc;
        hipLaunchKernelGGL(k_cmul, dim3(grid_for(g.Nc)), dim3(kBlock), 0, st, cbuf, otf, g.Nc,
                           mode, inv_n);
        M3D_KERNEL_CHECK();
        if ((rc = fft_c2r(cbuf, dst))) return rc;
        return stamp(m3d::EventLog::kWhole);
    };
    int rc;

    // ---- setup: OTF (rlgc.py:578-584) and update_norm (rlgc.py:587-590), unless this workspace still holds
    // them for the same PSF
    M3D_HIP_CHECK(hipMemsetAsync(dstat, 0, sizeof(Status), st));
    const int64_t npsf = (int64_t)p->pz * p->py * p->px;
    unsigned long long *hash_dev = (unsigned long long *)kl_part, psf_hash = 0;
    M3D_HIP_CHECK(hipMemsetAsync(hash_dev, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_psf_hash, dim3(grid_for(npsf)), dim3(kBlock), 0, st, psf, npsf, hash_dev);
    M3D_HIP_CHECK(hipMemcpyAsync(&psf_hash, hash_dev, sizeof(psf_hash), hipMemcpyDeviceToHost, st));
    M3D_HIP_CHECK(hipStreamSynchronize(st));
    const bool reuse_setup = p->setup_workspace == workspace && p->setup_psf_hash == psf_hash &&
                             p->setup_normalize == (params->normalize_psf ? 1 : 0) &&
                             !(getenv("M3D_RLGC_NO_SETUP_CACHE") && atoi(getenv("M3D_RLGC_NO_SETUP_CACHE")));
    if (!reuse_setup) {
        p->setup_workspace = nullptr;  // invalid until both volumes are complete
        M3D_HIP_CHECK(hipMemsetAsync(hu, 0, sizeof(float) * (size_t)g.Np, st));
        hipLaunchKernelGGL(k_psf_sum, dim3(1), dim3(kBlock), 0, st, psf, npsf, dstat);
        hipLaunchKernelGGL(k_psf_embed, dim3(grid_for(npsf)), dim3(kBlock), 0, st, psf, hu, p->pz,
                           p->py, p->px, g, dstat, params->normalize_psf);
        M3D_KERNEL_CHECK();
        if (p->fc) {
            if ((rc = m3d::fftconv_make_otf(p->fc, hu, otf, st))) return rc;
            if (otf2 && (rc = m3d::fftconv_otf_abs2(p->fc, otf, otf2, st))) return rc;
        } else if ((rc = fft_r2c(hu, otf))) {
            return rc;
        }
        hipLaunchKernelGGL(k_mask, dim3(g_pad), rb, 0, st, hu, g);
        M3D_KERNEL_CHECK();
        if ((rc = conv(hu, 1, un, kInCore))) return rc;
        hipLaunchKernelGGL(k_floor_reciprocal, dim3(grid_for(g.Np)), dim3(kBlock), 0, st, un, 1e-6f, g.Np);
        M3D_KERNEL_CHECK();
        p->setup_workspace = workspace;
        p->setup_psf_hash = psf_hash;
        p->setup_normalize = params->normalize_psf ? 1 : 0;
    }

    // ---- recon = mean(observed) everywhere (rlgc.py:600-603)
    const int g_sum = grid_for(g.N);
    if (image_dtype == M3D_U16)
        hipLaunchKernelGGL(k_image_sum<uint16_t>, dim3(g_sum), dim3(kBlock), 0, st,
                           (const uint16_t *)image, g.N, sum_part);
    else
        hipLaunchKernelGGL(k_image_sum<float>, dim3(g_sum), dim3(kBlock), 0, st,
                           (const float *)image, g.N, sum_part);
    hipLaunchKernelGGL(k_image_mean_finalize, dim3(1), dim3(kBlock), 0, st, sum_part, g_sum, g.N,
                       dstat);
    hipLaunchKernelGGL(k_fill_from, dim3(grid_for(g.Np)), dim3(kBlock), 0, st, recon_cur,
                       &dstat->mean, g.Np);
    M3D_KERNEL_CHECK();
    if (fused_tail) hipLaunchKernelGGL(k_arm_fused_reductions, dim3(1), dim3(kBlock), 0, st, cons_red, upd_red);

    // The update pass can hand the next forward-model convolution its X forward transform (ConvEpilogue::chain).
    const bool chain_update = fused_tail && m3d::fftconv_update_chain_supported(p->fc) &&
                              !(getenv("M3D_RLGC_CHAIN_UPDATE") && atoi(getenv("M3D_RLGC_CHAIN_UPDATE")) == 0);
    bool hu_x_ready = false;  // cbuf holds the X transforms of recon_cur's core rows
    // The split stage can sit between the X inverse pass of the Hu convolution and the X forward passes of the two
    // ratio convolutions (m3d::fftconv_split_chain): Hu and both ratio volumes then never reach HBM.  The per-workgroup
    // KL partials of that kernel take the (then unused) ratio1 volume's place.
    static_assert(sizeof(m3d::SplitPartial) == sizeof(KlPartial), "the split kernels share their partial-sum record");
    const bool split_chain = fused_tail && prune && m3d::fftconv_split_chain_supported(p->fc) &&
                             !(getenv("M3D_RLGC_CHAIN_SPLIT") && atoi(getenv("M3D_RLGC_CHAIN_SPLIT")) == 0) &&
                             (size_t)m3d::fftconv_split_chain_blocks(p->fc, &core) * sizeof(KlPartial) <=
                                 sizeof(float) * (size_t)g.Np;
    if (!split_chain) {  // ratio volumes: pad region stays zero for the whole run (only the core is rewritten)
        M3D_HIP_CHECK(hipMemsetAsync(r1, 0, sizeof(float) * (size_t)g.Np, st));
        M3D_HIP_CHECK(hipMemsetAsync(r2, 0, sizeof(float) * (size_t)g.Np, st));
    }
    auto enqueue_forward = [&](int it) -> int {
        const float *inj =
            (injected && it < n_injected) ? injected + (size_t)it * (size_t)g.N : nullptr;
        if (split_chain) {
            ++elog.convolutions;
            float2 *rows = nullptr, *cbuf2 = (float2 *)w2;
            int rc3 = m3d::fftconv_to_rows(p->fc, recon_cur, otf, 0, cbuf, st, &elog, nullptr, &core, &core, cbuf2,
                                           &rows, hu_x_ready);  // Hu up to its X inverse pass, rlgc.py:624
            hu_x_ready = false;
            if (rc3) return rc3;
            if (rows != cbuf2) {
                set_err("rlgc", "the mirrored forward-model convolution left its rows in an unexpected buffer");
                return M3D_ERR_HIP;
            }
            KlPartial *big = (KlPartial *)r1;
            const m3d::SplitArgs sa = {image, image_dtype == M3D_U16 ? 1 : 0, inj, (uint32_t)it, params->seed, binom,
                                       (m3d::SplitPartial *)big};
            if ((rc3 = m3d::fftconv_split_chain(p->fc, cbuf2, cbuf, &core, &sa, st, &elog))) return rc3;
            hipLaunchKernelGGL(k_kl_reduce, dim3(128), dim3(kBlock), 0, st, big,
                               (int)m3d::fftconv_split_chain_blocks(p->fc, &core), kl_part);
            hipLaunchKernelGGL(k_kl_finalize, dim3(1), dim3(kBlock), 0, st, kl_part, 128, dstat);
            M3D_KERNEL_CHECK();
            return M3D_OK;
        }
        int rc2 = conv(recon_cur, 0, hu, kOutCore | kMirrorIn, nullptr, hu_x_ready);  // Hu, rlgc.py:624 (core only)
        hu_x_ready = false;
        if (rc2) return rc2;
        if (image_dtype == M3D_U16)
            hipLaunchKernelGGL(k_split_ratio<uint16_t>, dim3(g_core), rb, 0, st, hu,
                               (const uint16_t *)image, r1, r2, g, (uint32_t)it, params->seed,
                               binom, inj, kl_part);
        else
            hipLaunchKernelGGL(k_split_ratio<float>, dim3(g_core), rb, 0, st, hu,
                               (const float *)image, r1, r2, g, (uint32_t)it, params->seed, binom,
                               inj, kl_part);
        hipLaunchKernelGGL(k_kl_finalize, dim3(1), dim3(kBlock), 0, st, kl_part, g_core, dstat);
        M3D_KERNEL_CHECK();
        return M3D_OK;
    };
    auto enqueue_update = [&]() -> int {
        int rc2;
        if (!fused_tail && (rc2 = conv(r1, 1, w1, kInCore))) return rc2;  // HTratio1 * update_norm, rlgc.py:654-662
        if (fused_tail) {
            // Neither HTratio (rlgc.py:654-672) reaches memory: both convolutions stop in front of their X
            // inverse pass, which they share - one packed transform per row returns both results and the
            // consensus input and HT1 + HT2 are written from registers; likewise the consensus map feeds
            // the gated update in place.
            float2 *cbuf2 = (float2 *)w2;  // w2 itself is never materialised on this path
            const m3d::ConvBox *in = &core;
            elog.convolutions += 2;
            constexpr bool zpair = true;  // (the two single launches remain for plans without the paired Z pass)
            if (split_chain) {  // cbuf / cbuf2 hold the rows' X transforms of ratio1 / ratio2 already
                if ((rc2 = m3d::fftconv_to_rows_pair(p->fc, nullptr, nullptr, otf, 1, cbuf, cbuf2, st, &elog, in, true)))
                    return rc2;
            } else if (zpair) {
                if ((rc2 = m3d::fftconv_to_rows_pair(p->fc, r1, r2, otf, 1, cbuf, cbuf2, st, &elog, in))) return rc2;
            } else {
                if ((rc2 = m3d::fftconv_to_rows(p->fc, r1, otf, 1, cbuf, st, &elog, in, nullptr, nullptr))) return rc2;
                if ((rc2 = m3d::fftconv_to_rows(p->fc, r2, otf, 1, cbuf2, st, &elog, in, nullptr, nullptr))) return rc2;
            }
            // ... and the consensus input goes straight into the X forward transform of its own convolution
            constexpr bool chain = true;
            const m3d::ConvEpilogue e1 = {m3d::ConvEpilogue::kConsensus, nullptr, un, hu, w1, cons_red, core};
            if ((rc2 = m3d::fftconv_dual_consensus(p->fc, cbuf, cbuf2, st, &elog, &e1, chain))) return rc2;
            m3d::ConvEpilogue e2 = {m3d::ConvEpilogue::kUpdate, recon_cur, w1, recon_prev, nullptr, upd_red, core};
            if (chain_update) e2.chain = cbuf;
            if ((rc2 = conv(chain ? nullptr : hu, 2, w2, kOutCore, &e2))) return rc2;
            hu_x_ready = chain_update;
            hipLaunchKernelGGL(k_fused_finalize, dim3(1), dim3(kBlock), 0, st, cons_red, upd_red, dstat);
            M3D_KERNEL_CHECK();
            return M3D_OK;
        }
        if ((rc2 = conv(r2, 1, w2, kInCore))) return rc2;  // rlgc.py:664-672
        hipLaunchKernelGGL(k_consensus_in, dim3(g_pad), rb, 0, st, w1, w2, un, hu, g, mm_part);
        M3D_KERNEL_CHECK();
        if ((rc2 = conv(hu, 2, w2, kOutCore))) return rc2;  // consensus_map, rlgc.py:677-679 (core only)
        hipLaunchKernelGGL(k_update, dim3(g_core), rb, 0, st, w2, recon_cur, w1, recon_prev, g,
                           up_part);
        if (!virtual_boundary) hipLaunchKernelGGL(k_boundary, dim3(g_pad), rb, 0, st, recon_prev, g);
        hipLaunchKernelGGL(k_update_finalize, dim3(1), dim3(kBlock), 0, st, up_part, g_core,
                           mm_part, g_pad, dstat);
        M3D_KERNEL_CHECK();
        return M3D_OK;
    };
    auto read_status = [&]() -> int {
        M3D_HIP_CHECK(hipMemcpyAsync(p->status_host, dstat, sizeof(Status), hipMemcpyDeviceToHost, st));
        M3D_HIP_CHECK(hipStreamSynchronize(st));
        return M3D_OK;
    };
    auto kl_from = [](double T, double A, double B) { return T / A - std::log(A) + std::log(B); };
    // One loop body = the update of iteration k, the swap of the estimate's two buffers, the forward model + KL sums of
    // iteration k + 1 (enqueued speculatively: a stop costs one wasted convolution).
    auto run_body = [&](int it_next) -> int {
        if (int rc2 = enqueue_update()) return rc2;
        std::swap(recon_cur, recon_prev);  // recon_prev held the updated volume; swap roles
        return enqueue_forward(it_next);   // speculative next Hu + KL
    };

    if (res) {
        memset(res, 0, sizeof(*res));
        res->engine_flags = m3d::fftconv_engine_flags(p->fc) | (split_chain ? M3D_ENGINE_SPLIT_CHAIN : 0) |
                            (chain_update ? M3D_ENGINE_UPDATE_CHAIN : 0);
    }
    double prev1 = INFINITY, prev2 = INFINITY;
    int num_iters = 0, stop = 0;
    float *result = nullptr;

    if (time_loop) {
        for (hipEvent_t &e : p->loop_ev)
            if (!e) M3D_HIP_CHECK(hipEventCreate(&e));
        M3D_HIP_CHECK(hipEventRecord(p->loop_ev[0], st));
    }
    if ((rc = enqueue_forward(0))) return rc;
    if ((rc = read_status())) return rc;
    for (;;) {
        const Status s = *p->status_host;
        const double kim = kl_from(s.Tim, s.A, s.Bim), k1 = kl_from(s.T1, s.A, s.B1),
                     k2 = kl_from(s.T2, s.A, s.B2);
        if (res && num_iters < M3D_RLGC_TRACE_CAP) {
            res->kld_image[num_iters] = kim;
            res->kld_split1[num_iters] = k1;
            res->kld_split2[num_iters] = k2;
            res->n_trace = num_iters + 1;
        }
        const bool restore = params->safe_mode ? ((k1 > prev1) || (k2 > prev2))
                                               : ((k1 > prev1) && (k2 > prev2));
        if (restore) {  // rlgc.py:634-649: previous recon is returned
            result = recon_prev;
            stop = 1;
            break;
        }
        prev1 = k1;
        prev2 = k2;
        if ((rc = run_body(num_iters + 1))) return rc;
        if ((rc = read_status())) return rc;
        const Status u = *p->status_host;
        const double frac = (double)u.updated / (double)g.N;
        const float rmax = fmaxf(u.recon_max, 1e-12f);
        const double mrd = (double)(u.max_abs_delta / rmax);
        if (res && num_iters < M3D_RLGC_TRACE_CAP) {
            res->updated_fraction[num_iters] = frac;
            res->max_relative_delta[num_iters] = mrd;
            res->update_min[num_iters] = u.ht_min;
            res->update_max[num_iters] = u.ht_max;
        }
        ++num_iters;
        if (frac < (double)params->limit) {
            result = recon_cur;
            stop = 2;
            break;
        }
        if (mrd < (double)params->max_delta) {
            result = recon_cur;
            stop = 3;
            break;
        }
        if (params->max_iterations > 0 && num_iters >= params->max_iterations) {
            result = recon_cur;
            stop = 4;
            break;
        }
    }
    if (num_iters == 0 && stop == 1) result = recon_cur;  // cannot happen (prev = inf); safety
    if (time_loop) M3D_HIP_CHECK(hipEventRecord(p->loop_ev[1], st));
    hipLaunchKernelGGL(k_crop, dim3(g_core), rb, 0, st, result, out, g);
    M3D_KERNEL_CHECK();
    M3D_HIP_CHECK(hipStreamSynchronize(st));
    if (res) {
        res->iterations = num_iters;
        res->stop_reason = stop;
        const std::vector<hipEvent_t> &ev = p->events;
        res->conv_calls = (int32_t)elog.convolutions;
        double total = 0.0;
        for (int k = 0; k < 5; ++k) res->kernel_ms[k] = 0.0;
        for (size_t i = 1; i < elog.used; ++i) {
            const int kind = elog.kinds[i];
            if (kind < 0) continue;  // opens a span
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ev[i - 1], ev[i]) != hipSuccess) continue;
            total += ms;
            if (kind < 5) res->kernel_ms[kind] += ms;
        }
        res->conv_ms_total = total;
        res->fused_engine = p->fc ? 1 : 0;
        float loop_ms = 0.0f;
        if (time_loop && hipEventElapsedTime(&loop_ms, p->loop_ev[0], p->loop_ev[1]) != hipSuccess) loop_ms = 0.0f;
        res->loop_ms = loop_ms;
    }
    return M3D_OK;
}

int m3d_rlgc_debug_split(const void *image, int image_dtype, int64_t n, uint64_t seed,
                         int32_t iteration, float *split1, void *stream_v)
{
    M3D_REQUIRE(image && split1 && n >= 0, "null pointer");
    hipStream_t st = (hipStream_t)stream_v;
    if (n == 0) return M3D_OK;
    BinomTables binom;
    if (int rc = device_binom_tables(&binom)) return rc;
    if (image_dtype == M3D_U16)
        hipLaunchKernelGGL(k_debug_split<uint16_t>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                           (const uint16_t *)image, n, seed, (uint32_t)iteration, binom, split1);
    else if (image_dtype == M3D_F32)
        hipLaunchKernelGGL(k_debug_split<float>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                           (const float *)image, n, seed, (uint32_t)iteration, binom, split1);
    else
        M3D_REQUIRE(false, "image dtype must be u16 or f32");
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_fft_debug_pk(const float *a, const float *b, const float *c, int64_t n, float *out, void *stream_v)
{
    M3D_REQUIRE(a && b && c && out && n >= 0, "null pointer");
    return m3d::fft_debug_pk((const float2 *)a, (const float2 *)b, (const float2 *)c, n, (float2 *)out, stream_v);
}

int m3d_binom_debug_table_draw(const uint32_t *q, const uint32_t *hi, const uint32_t *lo, int64_t n, uint32_t *out,
                               void *stream_v)
{
    M3D_REQUIRE(q && hi && lo && out && n >= 0, "null pointer");
    if (n == 0) return M3D_OK;
    BinomTables binom;
    if (int rc = device_binom_tables(&binom)) return rc;
    hipLaunchKernelGGL(k_debug_table_draw, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream_v, q, hi, lo, n,
                       binom, out);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_binom_tables_host(const uint64_t **thresholds, int64_t *n_thresholds, const uint32_t **row_off_len,
                          int32_t *q_max)
{
    const BinomTablesHost &h = binom_tables_host();
    if (!h.certified) {
        set_err("binomial split tables", "a threshold could not be certified");
        return M3D_ERR_HIP;
    }
    static_assert(sizeof(BinomRow) == 2 * sizeof(uint32_t), "rows are (offset, length) pairs");
    if (thresholds) *thresholds = h.thr.data();
    if (n_thresholds) *n_thresholds = (int64_t)h.thr.size();
    if (row_off_len) *row_off_len = (const uint32_t *)h.rows.data();
    if (q_max) *q_max = (int32_t)kBinomQMax;
    return M3D_OK;
}

int m3d_clip_to_u16(const float *in, uint16_t *out, int64_t n, void *stream_v)
{
    M3D_REQUIRE(in && out && n >= 0, "null pointer");
    if (n == 0) return M3D_OK;
    hipLaunchKernelGGL(k_clip_u16, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream_v, in,
                       out, n);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// The solver's building blocks as stand-alone entry points (the reference exposes them as module-level
// functions of utils/rlgc.py; m3d_rlgc_run runs fused versions of the same arithmetic).
// ------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

__global__ void k_filter_update(const float *__restrict__ recon, const float *__restrict__ ht,
                                const float *__restrict__ consensus, float *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = consensus[i] < 0.0f ? recon[i] : recon[i] * ht[i];
}

struct KlSums {
    double A, B, T;
};

// kl_div (rlgc.py:382-412) as three raw sums: with a = (p + eps) m, b = (q + eps) m,
// sum (a/A) (log(a/A) - log(b/B)) = T / A - log A + log B,  T = sum a (log a - log b) over a > 0
// (masked-out terms are 0 * (-inf + inf) = NaN -> 0 in the reference).
__global__ void k_kl_sums(const float *__restrict__ p, const float *__restrict__ q, const float *__restrict__ mask,
                          int64_t n, KlSums *__restrict__ partial)
{
    __shared__ double sm[4];
    const float eps = 1e-4f;
    double A = 0, B = 0, T = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask ? mask[i] : 1.0f;
        const float a = (p[i] + eps) * m, b = (q[i] + eps) * m;
        A += a;
        B += b;
        if (a > 0.0f && b > 0.0f) T += (double)a * ((double)logf(a) - (double)logf(b));
    }
    A = block_sum(A, sm);
    B = block_sum(B, sm);
    T = block_sum(T, sm);
    if (threadIdx.x == 0) {
        KlSums r = {A, B, T};
        partial[blockIdx.x] = r;
    }
}

__global__ void k_kl_total(const KlSums *__restrict__ partial, int nblocks, double *__restrict__ out)
{
    __shared__ double sm[4];
    double A = 0, B = 0, T = 0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) A += partial[i].A, B += partial[i].B, T += partial[i].T;
    A = block_sum(A, sm);
    B = block_sum(B, sm);
    T = block_sum(T, sm);
    if (threadIdx.x == 0) out[0] = (A > 0.0 && B > 0.0) ? T / A - log(A) + log(B) : 0.0;
}

int geo_from(const int32_t padded[3], const int32_t before[3], const int32_t after[3], Geo *g)
{
    M3D_REQUIRE(padded && before && after, "null shape");
    for (int a = 0; a < 3; ++a)
        M3D_REQUIRE(padded[a] > 0 && before[a] >= 0 && after[a] >= 0 && before[a] + after[a] < padded[a],
                    "pad widths must leave an observed region");
    g->Zp = padded[0], g->Yp = padded[1], g->Xp = padded[2];
    g->bz = before[0], g->by = before[1], g->bx = before[2];
    g->Z = padded[0] - before[0] - after[0], g->Y = padded[1] - before[1] - after[1];
    g->X = padded[2] - before[2] - after[2];
    g->Xc = g->Xp / 2 + 1;
    g->N = (int64_t)g->Z * g->Y * g->X;
    g->Np = (int64_t)g->Zp * g->Yp * g->Xp;
    g->Nc = (int64_t)g->Zp * g->Yp * g->Xc;
    return M3D_OK;
}

}  // namespace

struct m3d_fftconv {
    m3d::FftConvPlan *plan = nullptr;
    int32_t shape[3] = {0, 0, 0};
};

extern "C" {

int m3d_filter_update(const float *recon, const float *htratio, const float *consensus_map, float *out, int64_t n,
                      void *stream)
{
    M3D_REQUIRE(recon && htratio && consensus_map && out && n >= 0, "null pointer");
    if (n == 0) return M3D_OK;
    hipLaunchKernelGGL(k_filter_update, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, recon, htratio,
                       consensus_map, out, n);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_enforce_symmetric_boundary(float *volume, const int32_t padded_zyx[3], const int32_t pad_before[3],
                                   const int32_t pad_after[3], void *stream)
{
    M3D_REQUIRE(volume, "null pointer");
    Geo g;
    if (int rc = geo_from(padded_zyx, pad_before, pad_after, &g)) return rc;
    hipLaunchKernelGGL(k_boundary, dim3(row_grid((int64_t)g.Zp * g.Yp)), dim3(64, kRowsPerBlock), 0,
                       (hipStream_t)stream, volume, g);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int m3d_observed_region_mask(float *mask, const int32_t padded_zyx[3], const int32_t pad_before[3],
                             const int32_t pad_after[3], void *stream)
{
    M3D_REQUIRE(mask, "null pointer");
    Geo g;
    if (int rc = geo_from(padded_zyx, pad_before, pad_after, &g)) return rc;
    hipLaunchKernelGGL(k_mask, dim3(row_grid((int64_t)g.Zp * g.Yp)), dim3(64, kRowsPerBlock), 0, (hipStream_t)stream,
                       mask, g);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

size_t m3d_pad_psf_workspace_bytes(void) { return sizeof(Status); }

int m3d_pad_psf(const float *psf, const int32_t psf_zyx[3], float *out, const int32_t padded_zyx[3], int normalize,
                void *work, size_t work_bytes, void *stream)
{
    M3D_REQUIRE(psf && psf_zyx && out && padded_zyx && work, "null pointer");
    if (work_bytes < sizeof(Status)) {
        m3d::set_err("workspace", "smaller than m3d_pad_psf_workspace_bytes()");
        return M3D_ERR_WORKSPACE;
    }
    for (int a = 0; a < 3; ++a)
        M3D_REQUIRE(psf_zyx[a] > 0 && psf_zyx[a] <= padded_zyx[a], "the PSF must fit the padded volume");
    hipStream_t st = (hipStream_t)stream;
    Geo g{};
    g.Zp = padded_zyx[0], g.Yp = padded_zyx[1], g.Xp = padded_zyx[2];
    const int64_t np = (int64_t)g.Zp * g.Yp * g.Xp, npsf = (int64_t)psf_zyx[0] * psf_zyx[1] * psf_zyx[2];
    M3D_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)np * sizeof(float), st));
    hipLaunchKernelGGL(k_psf_sum, dim3(1), dim3(kBlock), 0, st, psf, npsf, (Status *)work);
    hipLaunchKernelGGL(k_psf_embed, dim3(grid_for(npsf)), dim3(kBlock), 0, st, psf, out, psf_zyx[0], psf_zyx[1],
                       psf_zyx[2], g, (const Status *)work, normalize ? 1 : 0);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

size_t m3d_kl_div_workspace_bytes(void) { return sizeof(KlSums) * (size_t)kMaxGrid + sizeof(double); }

int m3d_kl_div(const float *p, const float *q, const float *mask, int64_t n, void *work, size_t work_bytes,
               double *result_host, void *stream)
{
    M3D_REQUIRE(p && q && work && result_host && n > 0, "null pointer or empty input");
    if (work_bytes < m3d_kl_div_workspace_bytes()) {
        m3d::set_err("workspace", "smaller than m3d_kl_div_workspace_bytes()");
        return M3D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    KlSums *partial = (KlSums *)((char *)work + sizeof(double));
    double *total = (double *)work;
    const int grid = grid_for(n, 4);
    hipLaunchKernelGGL(k_kl_sums, dim3(grid), dim3(kBlock), 0, st, p, q, mask, n, partial);
    hipLaunchKernelGGL(k_kl_total, dim3(1), dim3(kBlock), 0, st, partial, grid, total);
    M3D_KERNEL_CHECK();
    M3D_HIP_CHECK(hipMemcpyAsync(result_host, total, sizeof(double), hipMemcpyDeviceToHost, st));
    M3D_HIP_CHECK(hipStreamSynchronize(st));  // the reference returns a Python float (rlgc.py:412)
    return M3D_OK;
}

int m3d_fftconv_create(const int32_t shape_zyx[3], m3d_fftconv **out)
{
    M3D_REQUIRE(shape_zyx && out, "null pointer");
    for (int a = 0; a < 3; ++a) M3D_REQUIRE(shape_zyx[a] > 0, "shape must be positive");
    m3d::FftConvPlan *plan = nullptr;
    const int rc = m3d::fftconv_create(shape_zyx[0], shape_zyx[1], shape_zyx[2], &plan);
    if (rc) {
        if (rc == M3D_ERR_UNSUPPORTED)
            m3d::set_err("unsupported", "fused FFT engine: every axis must be 2^a 3^b and a pass must fit LDS");
        return rc;
    }
    m3d_fftconv *h = new m3d_fftconv();
    h->plan = plan;
    for (int a = 0; a < 3; ++a) h->shape[a] = shape_zyx[a];
    *out = h;
    return M3D_OK;
}

int m3d_fftconv_destroy(m3d_fftconv *h)
{
    if (!h) return M3D_OK;
    m3d::fftconv_destroy(h->plan);
    delete h;
    return M3D_OK;
}

size_t m3d_fftconv_spectrum_bytes(const m3d_fftconv *h)
{
    return h ? (size_t)m3d::fftconv_spectrum_elems(h->plan) * sizeof(float2) : 0;
}

int m3d_fftconv_otf(m3d_fftconv *h, const float *psf_padded, void *otf, void *stream)
{
    M3D_REQUIRE(h && psf_padded && otf, "null pointer");
    return m3d::fftconv_make_otf(h->plan, psf_padded, (float2 *)otf, stream);
}

int m3d_fftconv_apply(m3d_fftconv *h, const float *image, const void *otf, int mode, void *scratch_spectrum, float *out,
                      void *stream)
{
    M3D_REQUIRE(h && image && otf && scratch_spectrum && out, "null pointer");
    M3D_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (H), 1 (conj H) or 2 (|H|^2)");
    return m3d::fftconv_conv(h->plan, image, (const float2 *)otf, mode, (float2 *)scratch_spectrum, out, stream);
}

}  // extern "C"
