// Fused 3-D FFT convolution for RLGC on MI355X: 5 kernels, 44 B of HBM traffic per voxel
// (34 B when the input is zero outside the observed core, or only the core of the output is
// needed - every RLGC convolution is one or the other; see ConvBox).
//
//   k_x_forward   two real rows -> one packed complex FFT (in LDS) -> two Hermitian half rows,
//                 written into the tiled spectrum  S[z][kx/TW][y][kx%TW]      (r 4 B, w 4 B)
//   k_y_pass      FFT along y of one contiguous [Yp][TW] column tile, in place  (r 4 B, w 4 B)
//   k_z_conv_reg  (Zp = 32, 64, 128, 256) FFT along z, multiply by the OTF (or conj / |.|^2), inverse FFT
//                 along z, in place, 32 adjacent spectrum positions per block: 16-point DFTs in registers
//                 around one LDS exchange per direction                          (r 8 B, w 4 B)
//   k_z_conv      the same for every other 2^a 3^b length, as an in-LDS stage chain
//   k_y_pass      inverse                                                        (r 4 B, w 4 B)
//   k_x_inverse   two Hermitian half rows -> packed complex inverse FFT -> two real rows
//                                                                                (r 4 B, w 4 B)
// The spectrum keeps the digit-reversed order of the decimation-in-frequency passes along y (and along z
// in k_z_conv; k_z_conv_reg keeps z natural) - only the OTF, which went through the same passes of the
// same plan, ever meets it.  Further kernels: k_y_forward_mirror (Y pass of a symmetric extension given by
// its core rows), k_x_inverse_dual (two results that meet in the consensus arithmetic share one X inverse
// pass, optionally chained into the next convolution's X forward pass), fused tails (ConvEpilogue).
// Replaces cp.fft.rfftn / "*= H" / cp.fft.irfftn of fft_conv (utils/rlgc.py:316-350).

#include "m3d_fftconv.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "m3d_common.h"
#include "m3d_fft.h"
#include "m3d_index.h"

using namespace m3d;

// loads kept in flight per thread in the load phase of each pass (measured on MI355X)
#ifndef M3D_KU_Z
#ifndef M3D_ZREG_ATTR
#define M3D_ZREG_ATTR
#endif
#ifndef M3D_ZREG_OTF_EARLY
#define M3D_ZREG_OTF_EARLY 0
#endif
#ifndef M3D_KH_Z
#define M3D_KH_Z 16
#endif
#define M3D_KU_Z 8
#endif
#ifndef M3D_KU_Y
#define M3D_KU_Y 3
#endif
#ifndef M3D_KU_XI
#define M3D_KU_XI 9
#endif

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
    int rset_x, rset_y, rset_z;  // 1: lean kernels (no radix 16 / 8 stage on that axis)
    int ythreads, zthreads, xthreads, gx, gy, gz, dual_rows, dual_threads, zmirror;
    int zreg;  // Zp / 16 when the register variant of the Z pass applies (Zp = 32, 64, 128, 256), else 0
    int xreg;  // X passes with register edge stages (radix list 9 ... 9, Xp / 9 <= 254, 256 threads)
    int lch;   // log2 of the rows per y chunk of the spectrum layout
    int xcdpair;
    int ystatic;  // the Y pass runs its static-geometry build
    int xstatic;  // the register-edge X passes run their static-geometry builds (NS = 2187)
};

}  // namespace m3d

// this workgroup's reduction record (ConvEpilogue::red, m3d_fftconv.h)
__device__ __forceinline__ unsigned int *red_slot(unsigned int *red)
{
    return red + (blockIdx.x % (unsigned)m3d::kRedSlots) * m3d::kRedStride;
}

namespace {

struct ConvGeo {
    int Zp, Yp, Xp, Xc, TW, nT, RB, P;
    int lTW, lNB, lP;  // log2 of TW, RB/2, P (all powers of two)
    int GX, GY, GZ;    // transforms one thread carries through a butterfly of each pass
    int lDR;           // log2 of the rows one block of the dual X inverse pass handles
    int lCH;           // log2 of the rows per y chunk of the spectrum layout (see spec_index)
    int xcdpair;       // dual X pass: row pairs that share lines go to one XCD
    int64_t plane;
    m3d::ConvBox in, out;  // support of the input / needed part of the output (full by default)
    m3d::ConvBox mir;      // mirror != 0: the input is the symmetric extension of this box
    int mirror;
    int zmirror;           // the spectrum holds the box's z planes only; the Z pass mirrors them in LDS
};

// Layout of one z plane of the half spectrum (plane = nT * Yp * TW complex elements): rows are grouped in chunks of
// CH = 2^lCH consecutive y; inside a chunk the kx tiles follow each other, each tile holding its rows x TW values:
//     S[z][y / CH][kx / TW][y % CH][kx % TW]          (the last chunk has Yp % CH rows)
// A Y pass reads its column tile as Yp / CH runs of CH * TW * 8 bytes (4 KB at CH = 128, TW = 4); an X pass block (four
// rows, all kx) touches 128-byte lines that are 4 KB apart, and the blocks of neighbouring rows - which run at the same
// time - fill the lines in between.  With CH >= Yp this is the plain tiled layout S[z][kx / TW][y][kx % TW], whose
// lines of one X block lie 70 KB apart: measured 2.3 TB/s for the X forward pass's stores against 3.5 TB/s chunked
// (profiles/r02_x_forward_variants.txt) - DRAM page locality, not line completeness, is what the stores pay for.
// row_base: in-plane offset of (y, kx = 0) and the row count of y's chunk.
__device__ __forceinline__ int row_base(const ConvGeo &g, int y, int &rows_c)
{
    const int y0c = (y >> g.lCH) << g.lCH;
    rows_c = min(1 << g.lCH, g.Yp - y0c);
    return (y0c * g.nT + (y - y0c)) << g.lTW;
}
__device__ __forceinline__ int tile_term(const ConvGeo &g, int k, int rows_c)
{
    return (((k >> g.lTW) * rows_c) << g.lTW) + (k & (g.TW - 1));
}
// z is uniform per block in every caller: keeping its term apart leaves one 64-bit product per block and 32-bit
// arithmetic per element (nT * Yp * TW < 2^31 is checked when the plan is made)
__device__ __forceinline__ int64_t spec_index(const ConvGeo &g, int z, int y, int k)
{
    const int64_t zterm = (int64_t)z * g.plane;  // plane = nT * Yp * TW
    int rows_c;
    const int rb = row_base(g, y, rows_c);
    return zterm + rb + tile_term(g, k, rows_c);
}
// element e = y * TW + kk of column tile kt (the order a Y pass keeps it in LDS) -> in-plane offset
__device__ __forceinline__ int tile_elem(const ConvGeo &g, int kt, int e)
{
    int rows_c;
    const int rb = row_base(g, e >> g.lTW, rows_c);
    return rb + ((kt * rows_c) << g.lTW) + (e & (g.TW - 1));
}

// One tile per block; several blocks share a CU (LDS 16-70 KB each) so the load / transform /
// store phases of different tiles overlap.  (A persistent, register-prefetching variant was
// measured slower: hipcc spills the prefetch arrays and vmcnt is in-order)

// ---------------------------------------------------------------- X forward (R2C)
template <int RSET>
__global__ __launch_bounds__(512) void k_x_forward(const float *__restrict__ src, float2 *__restrict__ spec, ConvGeo g,
                            AxisPlan ax)
{
    extern __shared__ float2 lds_fx[];
    const int tid = threadIdx.x, nt = blockDim.x;
    // row groups start on a multiple of RB so the pairing of rows does not depend on the box
    const int yg0 = (g.in.y0 / g.RB) * g.RB, y_hi = g.in.y0 + g.in.ny, x_hi = g.in.x0 + g.in.nx;
    const int groups = (y_hi - yg0 + g.RB - 1) / g.RB;
    const int z = g.in.z0 + blockIdx.x / groups, y0 = yg0 + (blockIdx.x % groups) * g.RB;
    const int nb = g.RB / 2, N = g.Xp;
    constexpr int kU = 3;   // batch of the table loads below (see k_z_conv)
    constexpr int kUL = 9;  // row loads are 4 bytes per lane: a whole 2187-sample row pair per batch
    for (int bi = 0; bi < nb; ++bi) {
        const int ya = y0 + 2 * bi, yb = ya + 1;
        const bool va = ya >= g.in.y0 && ya < y_hi, vb = yb >= g.in.y0 && yb < y_hi;
        // source rows: themselves, or (mirror == 1) their mirror images inside the box (symmetric extension);
        // mirror == 2 transforms the box's own rows only, extended symmetrically along x
        const int zs = g.mirror == 1 ? sym_src(z, g.mir.z0, g.mir.nz) : z;
        const int yas = g.mirror == 1 ? sym_src(min(ya, g.Yp - 1), g.mir.y0, g.mir.ny) : (va ? ya : g.in.y0);
        const int ybs = g.mirror == 1 ? sym_src(min(yb, g.Yp - 1), g.mir.y0, g.mir.ny) : (vb ? yb : g.in.y0);
        const float *ra = src + ((int64_t)zs * g.Yp + yas) * N;
        const float *rbp = src + ((int64_t)zs * g.Yp + ybs) * N;
        for (int n0 = tid; n0 < N; n0 += nt * kUL) {
            float a[kUL], b[kUL];
#pragma unroll
            for (int u = 0; u < kUL; ++u) {
                int n = min(n0 + u * nt, N - 1);
                if (g.mirror) n = sym_src(n, g.mir.x0, g.mir.nx);
                a[u] = ra[n];
                b[u] = rbp[n];
            }
#pragma unroll
            for (int u = 0; u < kUL; ++u) {
                const int n = n0 + u * nt;
                const bool vx = g.mirror || (n >= g.in.x0 && n < x_hi);
                if (n < N) lds_fx[bi * N + n] = make_float2(va && vx ? a[u] : 0.0f, vb && vx ? b[u] : 0.0f);
            }
        }
    }
    __syncthreads();
    lds_fft<false, RSET>(lds_fx, ax, 1, nb, N, tid, nt, g.GX);
    // Hermitian split: A[k] = (C[k] + conj(C[N-k])) / 2, B[k] = (C[k] - conj(C[N-k])) / (2i)
    const int total = (g.nT << g.lNB) << g.lTW;
    for (int e0 = tid; e0 < total; e0 += nt * kU) {
        int p1[kU], p2[kU];  // digit-reversed positions of k and N-k, fetched as one batch
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = min(e0 + u * nt, total - 1);
            const int k = min(((e >> (g.lTW + g.lNB)) << g.lTW) + (e & (g.TW - 1)), g.Xc - 1);
            p1[u] = ax.pos[k];
            p2[u] = ax.pos[k == 0 ? 0 : N - k];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + u * nt;
            const int kk = e & (g.TW - 1), bi = (e >> g.lTW) & (nb - 1), kt = e >> (g.lTW + g.lNB);
            const int k = (kt << g.lTW) + kk;
            if (e >= total || k >= g.Xc) continue;
            const float2 c1 = lds_fx[bi * N + p1[u]];
            const float2 c2 = lds_fx[bi * N + p2[u]];
            const float2 A = make_float2(0.5f * (c1.x + c2.x), 0.5f * (c1.y - c2.y));
            const float2 B = make_float2(0.5f * (c1.y + c2.y), 0.5f * (c2.x - c1.x));
            const int ya = y0 + 2 * bi;  // rows outside the box are zero-filled by the Y pass
            if (ya >= g.in.y0 && ya < y_hi) spec[spec_index(g, z, ya, k)] = A;
            if (ya + 1 >= g.in.y0 && ya + 1 < y_hi) spec[spec_index(g, z, ya + 1, k)] = B;
        }
    }
}

// ---------------------------------------------------------------- fused tails of the X inverse pass
// Exact elementwise arithmetic (this file is otherwise built with -ffp-contract=fast): the same
// operation order as k_consensus_in / k_update of m3d_rlgc.hip, whose results these must equal.
template <int EPI>
__device__ __forceinline__ void x_inverse_epilogue(const float2 *lds, const ConvGeo &g, const m3d::ConvEpilogue &ep,
                                                   int z, int y0, int nb, int N, int y_hi, int x_hi, int tid, int nt)
{
#pragma clang fp contract(off)
    __shared__ float smf[4];
    __shared__ unsigned long long smu[4];
    constexpr int kU = 3;
    float mn = INFINITY, mx = -INFINITY, dmax = 0.0f;
    unsigned long long updated = 0;
    const bool z_core = z >= ep.core.z0 && z < ep.core.z0 + ep.core.nz;
    for (int bi = 0; bi < nb; ++bi) {
        const int yr[2] = {y0 + 2 * bi, y0 + 2 * bi + 1};
        bool valid[2];
        int64_t row[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            valid[r] = yr[r] >= g.out.y0 && yr[r] < y_hi;
            row[r] = ((int64_t)z * g.Yp + (valid[r] ? yr[r] : g.out.y0)) * N;
        }
        for (int n0 = g.out.x0 + tid; n0 < x_hi; n0 += nt * kU) {
            float av[2][kU], bv[2][kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int n = min(n0 + u * nt, x_hi - 1);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    av[r][u] = ep.a[row[r] + n];
                    bv[r][u] = ep.b[row[r] + n];
                }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int n = n0 + u * nt;
                if (n >= x_hi) continue;
                const float2 v = lds[bi * N + n];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (!valid[r]) continue;
                    const float c = r ? v.y : v.x;
                    if constexpr (EPI == m3d::ConvEpilogue::kConsensus) {
                        const float un = bv[r][u];
                        const float h1 = av[r][u] * un, h2 = c * un;  // un = 1 / update_norm
                        ep.o1[row[r] + n] = (h1 - 1.0f) * (h2 - 1.0f);
                        const float hs = h1 + h2;
                        mn = fminf(mn, hs);
                        mx = fmaxf(mx, hs);
                        const bool core = z_core && yr[r] >= ep.core.y0 && yr[r] < ep.core.y0 + ep.core.ny &&
                                          n >= ep.core.x0 && n < ep.core.x0 + ep.core.nx;
                        if (core) ep.o2[row[r] + n] = hs;
                    } else {
                        const float old = av[r][u];
                        const float nw = c < 0.0f ? old : old * bv[r][u];
                        ep.o1[row[r] + n] = nw;
                        updated += (c >= 0.0f) ? 1ull : 0ull;
                        mx = fmaxf(mx, nw);
                        dmax = fmaxf(dmax, fabsf(nw - old));
                    }
                }
            }
        }
    }
    if constexpr (EPI == m3d::ConvEpilogue::kConsensus) {
        mn = block_min(mn, smf);
        mx = block_max(mx, smf);
        if (tid == 0) {
            unsigned int *red = red_slot(ep.red);
            atomicMin(&red[0], m3d::float_key(mn));
            atomicMax(&red[1], m3d::float_key(mx));
        }
    } else {
        updated = block_sum(updated, smu);
        mx = block_max(mx, smf);
        dmax = block_max(dmax, smf);
        if (tid == 0) {
            unsigned int *red = red_slot(ep.red);
            if (updated) atomicAdd((unsigned long long *)red, updated);
            atomicMax(&red[2], m3d::float_key(mx));
            atomicMax(&red[3], __float_as_uint(dmax));
        }
    }
}

// ---------------------------------------------------------------- X inverse (C2R)
template <int RSET, int EPI>
__global__ __launch_bounds__(512) void k_x_inverse(const float2 *__restrict__ spec, float *__restrict__ dst, ConvGeo g,
                            AxisPlan ax, m3d::ConvEpilogue ep)
{
    extern __shared__ float2 lds_ix[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int yg0 = (g.out.y0 / g.RB) * g.RB, y_hi = g.out.y0 + g.out.ny, x_hi = g.out.x0 + g.out.nx;
    const int groups = (y_hi - yg0 + g.RB - 1) / g.RB;
    const int z = g.out.z0 + blockIdx.x / groups, y0 = yg0 + (blockIdx.x % groups) * g.RB;
    const int nb = g.RB / 2, N = g.Xp;
    const int total = (g.nT << g.lNB) << g.lTW;
    constexpr int kU = M3D_KU_XI;  // independent loads in flight per thread (see k_z_conv)
    for (int e0 = tid; e0 < total; e0 += nt * kU) {
        float2 Av[kU], Bv[kU];
        int p1[kU], p2[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = min(e0 + u * nt, total - 1);
            const int kk = e & (g.TW - 1), bi = (e >> g.lTW) & (nb - 1), kt = e >> (g.lTW + g.lNB);
            const int k = min((kt << g.lTW) + kk, g.Xc - 1);
            p1[u] = ax.pos[k];
            p2[u] = ax.pos[k == 0 ? 0 : N - k];
            const int ya = y0 + 2 * bi;  // rows outside the box were not produced by the Y pass
            const int yac = min(max(ya, g.out.y0), y_hi - 1), ybc = min(max(ya + 1, g.out.y0), y_hi - 1);
            Av[u] = spec[spec_index(g, z, yac, k)];
            Bv[u] = spec[spec_index(g, z, ybc, k)];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + u * nt;
            const int kk = e & (g.TW - 1), bi = (e >> g.lTW) & (nb - 1), kt = e >> (g.lTW + g.lNB);
            const int k = (kt << g.lTW) + kk;
            if (e >= total || k >= g.Xc) continue;
            const int ya = y0 + 2 * bi;
            const float2 A = ya >= g.out.y0 && ya < y_hi ? Av[u] : make_float2(0.f, 0.f);
            const float2 B = ya + 1 >= g.out.y0 && ya + 1 < y_hi ? Bv[u] : make_float2(0.f, 0.f);
            // C[k] = A + iB ; C[N-k] = conj(A) + i conj(B)
            lds_ix[bi * N + p1[u]] = make_float2(A.x - B.y, A.y + B.x);
            if (k != 0 && 2 * k != N) lds_ix[bi * N + p2[u]] = make_float2(A.x + B.y, B.x - A.y);
        }
    }
    __syncthreads();
    lds_fft<true, RSET>(lds_ix, ax, 1, nb, N, tid, nt, g.GX);
    if constexpr (EPI == m3d::ConvEpilogue::kNone) {
        for (int bi = 0; bi < nb; ++bi) {
            const int ya = y0 + 2 * bi, yb = ya + 1;
            float *ra = dst + ((int64_t)z * g.Yp + ya) * N;
            float *rbp = ra + N;
            const bool va = ya >= g.out.y0 && ya < y_hi, vb = yb >= g.out.y0 && yb < y_hi;
            for (int n = g.out.x0 + tid; n < x_hi; n += nt) {
                const float2 v = lds_ix[bi * N + n];
                if (va) ra[n] = v.x;
                if (vb) rbp[n] = v.y;
            }
        }
    } else {
        x_inverse_epilogue<EPI>(lds_ix, g, ep, z, y0, nb, N, y_hi, x_hi, tid, nt);
    }
}

// ---------------------------------------------------------------- dual X inverse (two results, one pass)
// CHAIN: the product is not stored but left in LDS row 0 as (row y0, row y0 + 1) - the packed input of
// the next convolution's X forward transform.
template <bool CHAIN>
__device__ __forceinline__ void dual_consensus_epilogue(float2 *lds, const ConvGeo &g, const m3d::ConvEpilogue &ep,
                                                        int z, int y0, int nb, int N, int tid, int nt)
{
#pragma clang fp contract(off)
    __shared__ float smf[4];
    constexpr int kU = 4;
    float mn = INFINITY, mx = -INFINITY;
    const bool z_core = z >= ep.core.z0 && z < ep.core.z0 + ep.core.nz;
    for (int bi = 0; bi < nb; ++bi) {
        const int y = y0 + bi;
        if (y >= g.Yp) break;
        const int64_t row = ((int64_t)z * g.Yp + y) * N;
        const bool yz_core = z_core && y >= ep.core.y0 && y < ep.core.y0 + ep.core.ny;
        for (int n0 = tid; n0 < N; n0 += nt * kU) {
            float un[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) un[u] = ep.b[row + min(n0 + u * nt, N - 1)];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int n = n0 + u * nt;
                if (n >= N) continue;
                const float2 v = lds[bi * N + n];  // (HTratio1 raw, HTratio2 raw)
                const float h1 = v.x * un[u], h2 = v.y * un[u];  // un = 1 / update_norm
                const float prod = (h1 - 1.0f) * (h2 - 1.0f);
                if constexpr (CHAIN) {
                    if (bi == 0)
                        lds[n] = make_float2(prod, 0.0f);
                    else
                        lds[n].y = prod;
                } else {
                    ep.o1[row + n] = prod;
                }
                const float hs = h1 + h2;
                mn = fminf(mn, hs);
                mx = fmaxf(mx, hs);
                if (yz_core && n >= ep.core.x0 && n < ep.core.x0 + ep.core.nx) ep.o2[row + n] = hs;
            }
        }
    }
    mn = block_min(mn, smf);
    mx = block_max(mx, smf);
    if (tid == 0) {
        unsigned int *red = red_slot(ep.red);
        atomicMin(&red[0], m3d::float_key(mn));
        atomicMax(&red[1], m3d::float_key(mx));
    }
}

// Full output box only (the consensus input is needed on the whole padded volume).
// CHAIN (two rows per block): the consensus input goes straight into the X forward transform of the
// convolution that consumes it, and its two half-spectrum rows overwrite this block's rows of spec1 -
// the product never exists in HBM either.
template <int RSET, bool CHAIN>
__global__ __launch_bounds__(512) void k_x_inverse_dual(float2 *__restrict__ spec1,
                                                        const float2 *__restrict__ spec2, ConvGeo g, AxisPlan ax,
                                                        m3d::ConvEpilogue ep)
{
    extern __shared__ float2 lds_id[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nb = 1 << g.lDR, N = g.Xp;  // nb rows per block, one packed transform each
    const int groups = (g.Yp + nb - 1) / nb;
    const int z = blockIdx.x / groups, y0 = (blockIdx.x % groups) * nb;
    const int total = (g.nT << g.lDR) << g.lTW;
    constexpr int kU = M3D_KU_XI;
    for (int e0 = tid; e0 < total; e0 += nt * kU) {
        float2 Av[kU], Bv[kU];
        int p1[kU], p2[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = min(e0 + u * nt, total - 1);
            const int kk = e & (g.TW - 1), bi = (e >> g.lTW) & (nb - 1), kt = e >> (g.lTW + g.lDR);
            const int k = min((kt << g.lTW) + kk, g.Xc - 1);
            p1[u] = ax.pos[k];
            p2[u] = ax.pos[k == 0 ? 0 : N - k];
            const int64_t idx = spec_index(g, z, min(y0 + bi, g.Yp - 1), k);
            Av[u] = spec1[idx];
            Bv[u] = spec2[idx];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + u * nt;
            const int kk = e & (g.TW - 1), bi = (e >> g.lTW) & (nb - 1), kt = e >> (g.lTW + g.lDR);
            const int k = (kt << g.lTW) + kk;
            if (e >= total || k >= g.Xc) continue;
            const float2 A = Av[u], B = Bv[u];
            lds_id[bi * N + p1[u]] = make_float2(A.x - B.y, A.y + B.x);
            if (k != 0 && 2 * k != N) lds_id[bi * N + p2[u]] = make_float2(A.x + B.y, B.x - A.y);
        }
    }
    __syncthreads();
    lds_fft<true, RSET>(lds_id, ax, 1, nb, N, tid, nt, g.GX);
    dual_consensus_epilogue<CHAIN>(lds_id, g, ep, z, y0, nb, N, tid, nt);
    if constexpr (CHAIN) {
        __syncthreads();
        lds_fft<false, RSET>(lds_id, ax, 1, 1, N, tid, nt, g.GX);
        // Hermitian split of the packed pair, as in k_x_forward
        constexpr int kS = 3;
        const int tot = g.nT << g.lTW;
        for (int e0 = tid; e0 < tot; e0 += nt * kS) {
            int p1[kS], p2[kS];
#pragma unroll
            for (int u = 0; u < kS; ++u) {
                const int k = min(e0 + u * nt, g.Xc - 1);
                p1[u] = ax.pos[k];
                p2[u] = ax.pos[k == 0 ? 0 : N - k];
            }
#pragma unroll
            for (int u = 0; u < kS; ++u) {
                const int k = e0 + u * nt;
                if (k >= g.Xc) continue;
                const float2 c1 = lds_id[p1[u]], c2 = lds_id[p2[u]];
                spec1[spec_index(g, z, y0, k)] = make_float2(0.5f * (c1.x + c2.x), 0.5f * (c1.y - c2.y));
                if (y0 + 1 < g.Yp)
                    spec1[spec_index(g, z, y0 + 1, k)] = make_float2(0.5f * (c1.y + c2.y), 0.5f * (c2.x - c1.x));
            }
        }
    }
}

// ================================================================ X passes with register edge stages
// PMC passes over the kernels above (profiles/r02_pmc_sq_rlgc_before.json): waves parked on s_waitcnt / barriers half
// of their lifetime, 30 % of the LDS cycles lost to bank conflicts, and the gather / split phases - chains of dependent
// `pos` table loads around scattered LDS accesses - cost as much as the four transform stages between them.  For lengths
// N = 9 S whose radix list starts and ends with 9 (2187 = 9 * 9 * 3 * 9, 324 = 9 * 4 * 9) and S <= 256 the two outer
// stages run in registers instead:
//   * the stage of block size N touches x[t + S q], q < 9 - the elements thread t loads from / stores to a real row
//     with coalesced 4-byte accesses (forward: rows -> registers -> butterfly -> LDS; inverse: LDS -> butterfly ->
//     registers -> fused tail -> rows): no LDS round trip, no barrier on that side, every global load issued up front;
//   * the stage of block size 9 works on 9 adjacent LDS positions = the frequencies k' + S d, d < 9.  N - (k' + S d)
//     = (S - k') + S (8 - d), so the blocks of k' and S - k' hold each other's mirror frequencies: one thread owns both
//     and does the Hermitian split / merge in registers, writing or reading the tiled spectrum directly.  The digit-
//     reversed position of a block is pos[k'] / 9 - two table loads per thread instead of two per element.
// LDS passes per transform: 1 write + (stages - 2) read-modify-writes + 1 read (2187: 3 + 3 instead of 5 + 5), three
// barriers instead of five.
template <int R = 9>
__device__ __forceinline__ void edge_twiddles(float2 (&w)[9], const float2 *__restrict__ tw, int t)
{
    // W_N^(t q), q < R: powers 1, 2, 4 from the table (8 t < N), the rest by products - as lds_stage<9>
    w[1] = tw[t];
    w[2] = tw[2 * t];
    w[4] = tw[4 * t];
    w[3] = cmul(w[1], w[2]);
    w[5] = cmul(w[1], w[4]);
    w[6] = cmul(w[2], w[4]);
    w[7] = cmul(w[3], w[4]);
    if constexpr (R == 9) w[8] = cmul(w[4], w[4]);
}
// The edge butterfly of a register-edge X pass: radix 9 (lengths 81 m) or radix 8 (lengths 64 m; round 3).  For R = 8
// the arrays keep nine slots: the pair (k', S - k') of 8-blocks again holds nine stored frequencies at most - five of
// block 0 (S d, d <= 4: the last is the Nyquist frequency), else four of block k' and four of block S - k'.
template <int R, bool INV>
__device__ __forceinline__ void edge_dft(float2 *v)
{
    if constexpr (R == 9) dft9<INV>(v);
    else dft8<INV>(v);
}

// The static X kernels (N = 2187 = 9 * 9 * 3 * 9 at 256 threads) request every twiddle they will need when they start,
// next to their first data loads: a table load in front of a stage is a trip to L2 (the streams evict the table from
// L1) that the whole stage waits for - four such trips per middle pass and one per edge stage in round 2's kernels,
// at four waves per SIMD.  A thread's butterfly indices are the same in every stage of one kind, so nine values serve
// the whole kernel: edge stages W^(t q), q = 1, 2, 4 (t = tid), block-243 stage W^(9 j q) with j = tid % 27, block-27
// stage W^(81 j) with j = (tid + 256 k) / 81, k < 3.  The other powers are the same products as before.
struct XTwiddles {
    float2 e1, e2, e4;  // edge stages
    float2 a1, a2, a4;  // block-243 stage (radix 9)
    float2 b[3];        // block-27 stage (radix 3), items tid + 256 k
};
__device__ __forceinline__ XTwiddles x_twiddles_2187(const float2 *__restrict__ tw, int tid)
{
    XTwiddles x;
    const int t = min(tid, 242), j1 = (tid % 27) * 9;
    x.e1 = tw[t];
    x.e2 = tw[2 * t];
    x.e4 = tw[4 * t];
    x.a1 = tw[j1];
    x.a2 = tw[2 * j1];
    x.a4 = tw[4 * j1];
#pragma unroll
    for (int k = 0; k < 3; ++k) x.b[k] = tw[81 * (min(tid + 256 * k, 728) / 81)];
    return x;
}
__device__ __forceinline__ void powers_124(float2 (&w)[9], float2 w1, float2 w2, float2 w4)
{
    w[1] = w1;
    w[2] = w2;
    w[4] = w4;
    w[3] = cmul(w[1], w[2]);
    w[5] = cmul(w[1], w[4]);
    w[6] = cmul(w[2], w[4]);
    w[7] = cmul(w[3], w[4]);
    w[8] = cmul(w[4], w[4]);
}
// Middle stages of NB rows (row r at lds + r * 2187) with the twiddles in registers; every row's elements are read
// before the first butterfly (NB * 9 independent LDS reads in flight instead of 9, and nothing between them and the
// butterflies but arithmetic).  Same operations in the same order as lds_stage<9> / lds_stage<3>.
template <bool INV, int NB>
__device__ __forceinline__ void x_stage243(float2 *lds, int tid, const XTwiddles &x)
{
    if (tid >= 243) return;
    float2 w[9];
    powers_124(w, x.a1, x.a2, x.a4);
    const int blk = tid / 27, j = tid - blk * 27;
    float2 *p = lds + blk * 243 + j;
    float2 v[NB][9];
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int q = 0; q < 9; ++q) v[r][q] = p[r * 2187 + q * 27];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        if (INV) {
#pragma unroll
            for (int q = 1; q < 9; ++q) v[r][q] = cmul_conj(v[r][q], w[q]);
            dft9<true>(v[r]);
        } else {
            dft9<false>(v[r]);
#pragma unroll
            for (int q = 1; q < 9; ++q) v[r][q] = cmul(v[r][q], w[q]);
        }
    }
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int q = 0; q < 9; ++q) p[r * 2187 + q * 27] = v[r][q];
}
template <bool INV, int NB>
__device__ __forceinline__ void x_stage27(float2 *lds, int tid, const XTwiddles &x)
{
    float2 v[3][NB][3];
    int at[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // item -> (j, block) with the BLOCK running along the lanes: neighbouring lanes are 27 elements apart (odd, so
        // the 16 lanes of an LDS access group fall on 16 different bank pairs); with j along the lanes a group read
        // nine adjacent elements, skipped eighteen and wrapped onto its own banks - every access of this stage took
        // two LDS cycles per group (scripts/lds_banks.py, profiles/r03_lds_banks.txt)
        const int t = min(tid + 256 * k, 728), j = t / 81, blk = t - j * 81;
        at[k] = blk * 27 + j;
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) v[k][r][q] = lds[at[k] + r * 2187 + q * 9];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float2 w1 = x.b[k], w2 = cmul(w1, w1);
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            float2(&u)[3] = v[k][r];
            if (INV) {
                u[1] = cmul_conj(u[1], w1);
                u[2] = cmul_conj(u[2], w2);
                dft3<true>(u[0], u[1], u[2]);
            } else {
                dft3<false>(u[0], u[1], u[2]);
                u[1] = cmul(u[1], w1);
                u[2] = cmul(u[2], w2);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (tid + 256 * k >= 729) continue;  // (the clamped item above only read)
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) lds[at[k] + r * 2187 + q * 9] = v[k][r][q];
    }
}
template <int NB>
__device__ __forceinline__ void middle_forward_2187(float2 *lds, int tid, const XTwiddles &x)
{
    x_stage243<false, NB>(lds, tid, x);
    __syncthreads();
    x_stage27<false, NB>(lds, tid, x);
    __syncthreads();
}
template <int NB>
__device__ __forceinline__ void middle_inverse_2187(float2 *lds, int tid, const XTwiddles &x)
{
    x_stage27<true, NB>(lds, tid, x);
    __syncthreads();
    x_stage243<true, NB>(lds, tid, x);
    __syncthreads();
}

// Static geometry (see k_y_pass_static): NS = 2187 fixes Xp = Yp = 2187, TW = 4, four rows per X block, 128-row
// chunks and the radix list 9 * 9 * 3 * 9, so the index arithmetic of a pass folds into immediates.  NS = 0: run time.
template <int NS>
__device__ __forceinline__ ConvGeo static_geo(const ConvGeo &g)
{
    ConvGeo s = g;
    if constexpr (NS == 2187) {
        s.Xp = 2187;
        s.Yp = 2187;
        s.Xc = 1094;
        s.TW = 4;
        s.lTW = 2;
        s.nT = 274;
        s.RB = 4;
        s.lNB = 1;
        s.lCH = 7;
        s.lDR = 1;
        s.GX = 4;
        s.plane = (int64_t)274 * 2187 * 4;
    }
    return s;
}

// middle stages of the DIF forward transform (block sizes S ... 9 exclusive of the two register stages)
template <int RSET, int NS = 0>
__device__ __forceinline__ void middle_forward(float2 *lds, const AxisPlan &ax, int nb, int N, int tid, int nt, int gmax)
{
    if constexpr (NS == 2187) {
        lds_stage<9, false>(lds, 2187, 243, 1, nb, 2187, ax.tw, tid, 256, 4);
        __syncthreads();
        lds_stage<3, false, true>(lds, 2187, 27, 1, nb, 2187, ax.tw, tid, 256, 4);
        __syncthreads();
        return;
    }
    // The common lengths with radix-8 edge stages get their middle stages with literal geometry (block sizes, strides,
    // trip counts and the division of the item index fold): 2048 = 8 [8 4] 8 at 320 threads, 1728 = 8 [9 3] 8 and
    // 1152 = 8 [9 2] 8 at 256.  Same stages in the same order as the loop below.
    if constexpr (NS == 0 && (RSET == 1 || RSET == 2)) {
#define M3D_MID_F(NN, R1, M1, R2, M2, NTH)                                        \
    if (N == NN && nt == NTH && (nb == 1 || nb == 2)) {                           \
        if (nb == 2) lds_stage<R1, false>(lds, NN, M1, 1, 2, NN, ax.tw, tid, NTH, 4); \
        else lds_stage<R1, false>(lds, NN, M1, 1, 1, NN, ax.tw, tid, NTH, 4);      \
        __syncthreads();                                                          \
        if (nb == 2) lds_stage<R2, false>(lds, NN, M2, 1, 2, NN, ax.tw, tid, NTH, 4); \
        else lds_stage<R2, false>(lds, NN, M2, 1, 1, NN, ax.tw, tid, NTH, 4);      \
        __syncthreads();                                                          \
        return;                                                                   \
    }
        if constexpr (RSET == 2) {
            if (gmax == 4 && N == 576 && nt == 256 && (nb == 1 || nb == 2)) {  // 8 [9] 8
                if (nb == 2) lds_stage<9, false>(lds, 576, 72, 1, 2, 576, ax.tw, tid, 256, 4);
                else lds_stage<9, false>(lds, 576, 72, 1, 1, 576, ax.tw, tid, 256, 4);
                __syncthreads();
                return;
            }
            if (gmax == 4) {
                M3D_MID_F(2048, 8, 256, 4, 32, 320)
                M3D_MID_F(2592, 8, 288, 4, 36, 320)  // 9 [8 4] 9
                M3D_MID_F(1728, 9, 216, 3, 24, 256)
                M3D_MID_F(1152, 9, 144, 2, 16, 256)
            }
        } else if (gmax == 4) {
            M3D_MID_F(1296, 4, 144, 4, 36, 256)  // 9 [4 4] 9
        }
#undef M3D_MID_F
    }
    int M = N / ax.radix[0];
    for (int s = 1; s < ax.nstages - 1; ++s) {
        const int r = ax.radix[s];
        lds_stage_r<false, RSET>(r, lds, N, M, 1, nb, N, ax.tw, tid, nt, gmax);
        M /= r;
        __syncthreads();
    }
}
template <int RSET, int NS = 0>
__device__ __forceinline__ void middle_inverse(float2 *lds, const AxisPlan &ax, int nb, int N, int tid, int nt, int gmax)
{
    if constexpr (NS == 2187) {
        lds_stage<3, true, true>(lds, 2187, 27, 1, nb, 2187, ax.tw, tid, 256, 4);
        __syncthreads();
        lds_stage<9, true>(lds, 2187, 243, 1, nb, 2187, ax.tw, tid, 256, 4);
        __syncthreads();
        return;
    }
    if constexpr (NS == 0 && (RSET == 1 || RSET == 2)) {  // (see middle_forward)
#define M3D_MID_I(NN, R1, M1, R2, M2, NTH)                                       \
    if (N == NN && nt == NTH && (nb == 1 || nb == 2)) {                          \
        if (nb == 2) lds_stage<R2, true>(lds, NN, M2, 1, 2, NN, ax.tw, tid, NTH, 4); \
        else lds_stage<R2, true>(lds, NN, M2, 1, 1, NN, ax.tw, tid, NTH, 4);      \
        __syncthreads();                                                         \
        if (nb == 2) lds_stage<R1, true>(lds, NN, M1, 1, 2, NN, ax.tw, tid, NTH, 4); \
        else lds_stage<R1, true>(lds, NN, M1, 1, 1, NN, ax.tw, tid, NTH, 4);      \
        __syncthreads();                                                         \
        return;                                                                  \
    }
        if constexpr (RSET == 2) {
            if (gmax == 4 && N == 576 && nt == 256 && (nb == 1 || nb == 2)) {
                if (nb == 2) lds_stage<9, true>(lds, 576, 72, 1, 2, 576, ax.tw, tid, 256, 4);
                else lds_stage<9, true>(lds, 576, 72, 1, 1, 576, ax.tw, tid, 256, 4);
                __syncthreads();
                return;
            }
            if (gmax == 4) {
                M3D_MID_I(2048, 8, 256, 4, 32, 320)
                M3D_MID_I(2592, 8, 288, 4, 36, 320)
                M3D_MID_I(1728, 9, 216, 3, 24, 256)
                M3D_MID_I(1152, 9, 144, 2, 16, 256)
            }
        } else if (gmax == 4) {
            M3D_MID_I(1296, 4, 144, 4, 36, 256)
        }
#undef M3D_MID_I
    }
    int M = ax.radix[0];  // (= the last stage's radix: the edge stages share theirs)
    for (int s = ax.nstages - 2; s >= 1; --s) {
        const int r = ax.radix[s];
        M *= r;
        lds_stage_r<true, RSET>(r, lds, N, M, 1, nb, N, ax.tw, tid, nt, gmax);
        __syncthreads();
    }
}

// Last forward stage + Hermitian split for the pair of 9-blocks (k', S - k') of the packed transform at `row`
// (LDS): A (first packed real row) -> dstA[k], B -> dstB[k] for the nine frequencies k < Xc of the pair.
// `sa` / `sb`: element offsets of k = 0 in the tiled spectrum rows (only the tile term depends on k).
// LDS staging of one block's output lines (TW = 4, four rows per block: tile kt of rows y0 .. y0 + 3 is one 128-byte
// line = eight 16-byte chunks, chunk c = 2 r + kk / 2).  Chunks are XOR-swizzled by the line index so that the split's
// 8-byte writes (lanes = consecutive k) and the 16-byte read-back (lanes = consecutive chunks) are both conflict free.
__device__ __forceinline__ int stage_slot(int k, int r)
{
    const int L = k >> 2, kk = k & 3, c = 2 * r + (kk >> 1);
    return ((L * 8 + (c ^ (L & 7))) << 1) + (kk & 1);
}

// Last forward stage for the pair of 9-blocks (k', S - k'): both blocks read from LDS and transformed in registers.
template <int R = 9>
__device__ __forceinline__ void forward_pair_load(const float2 *row, int u1, int u2, float2 (&c1)[9], float2 (&c2)[9])
{
#pragma unroll
    for (int q = 0; q < R; ++q) c1[q] = row[R * u1 + q];
#pragma unroll
    for (int q = 0; q < R; ++q) c2[q] = row[R * u2 + q];
    edge_dft<R, false>(c1);
    edge_dft<R, false>(c2);
}

// Hermitian split of the pair: A (first packed real row) and B for the nine frequencies k < Xc of the pair, written
// to the tiled spectrum rows ya / ya + 1 directly, or (STAGED) into the block's line staging at rows r0 / r0 + 1.
template <bool STAGED = false, int R = 9>
__device__ __forceinline__ void forward_pair_emit(const float2 (&c1)[9], const float2 (&c2)[9], const ConvGeo &g, int S,
                                                  int kp, float2 *__restrict__ spec, int64_t zterm, int ya, bool wa,
                                                  bool wb, float2 *stage = nullptr, int r0 = 0)
{
    const int k2p = kp == 0 ? 0 : S - kp;
    int rows_c = 0;
    const int rbase = STAGED ? 0 : row_base(g, min(ya, g.Yp - 1), rows_c);
    const float2 half = make_float2(0.5f, 0.5f);
    auto put = [&](int k, float2 c, float2 m) {
        // A = (c + conj(m)) / 2, B = -i (c - conj(m)) / 2
        const float2 A = add_conj(c, m) * half, B = split_odd(c, m) * half;
        if constexpr (STAGED) {
            stage[stage_slot(k, r0)] = A;
            stage[stage_slot(k, r0 + 1)] = B;
        } else {
            const int in_plane = rbase + tile_term(g, k, rows_c);  // row ya + 1 (same chunk: ya is even) follows at + TW
            if (wa) spec[zterm + in_plane] = A;
            if (wb) spec[zterm + in_plane + g.TW] = B;
        }
    };
#pragma unroll
    for (int d = 0; d < 5; ++d) {  // selected by value: a conditional on the array elements themselves becomes a
                                   // pointer select and sends both arrays to scratch
        if (R == 8 && d == 4 && kp != 0) break;  // radix 8: k' + 4 S lies beyond the Nyquist frequency unless k' = 0
        const float2 m0 = c1[(R - d) % R], m1 = c2[(R - 1 - d) % R];
        put(kp + S * d, c1[d], make_float2(kp == 0 ? m0.x : m1.x, kp == 0 ? m0.y : m1.y));
    }
    if (k2p != kp) {
#pragma unroll
        for (int d = 0; d < 4; ++d) put(k2p + S * d, c2[d], c1[R - 1 - d]);
    }
}

// Write-out of a block's staged lines: one 16-byte chunk per lane, eight lanes per 128-byte line - every store
// instruction writes whole lines (the direct form writes a quarter of sixteen lines per instruction and runs at
// 2.3 TB/s; profiles/r02_x_forward_variants.txt).  rows_ok: bit r set = row y0 + r is inside the box.
__device__ __forceinline__ void store_staged_lines(const float2 *stage, const ConvGeo &g, float2 *__restrict__ spec,
                                                   int64_t zterm, int y0, unsigned rows_ok, int tid, int nt)
{
    const float4 *st4 = (const float4 *)stage;
    int rows_c;
    const int rbase = row_base(g, y0, rows_c);  // y0 .. y0 + 3 share a chunk (y0 and the chunk size are multiples of 4)
    for (int i = tid; i < g.nT * 8; i += nt) {
        const int L = i >> 3, c = (i & 7) ^ (L & 7), r = c >> 1, k = 4 * L + 2 * (c & 1);
        if (!((rows_ok >> r) & 1u) || k >= g.Xc) continue;
        const float4 v = st4[i];
        float2 *dst = spec + zterm + rbase + ((L * rows_c + r) << 2) + 2 * (c & 1);
        if (k + 1 < g.Xc)
            st_stream((float4 *)dst, v);  // whole lines, read next by another pass: -4 % (update chain), -5 % (forward)
        else
            *dst = make_float2(v.x, v.y);
    }
}

// Hermitian merge + first inverse stage for the pair (k', S - k'): C = A + i B at the nine frequencies k < Xc of the
// pair and conj(A) + i conj(B) at their mirrors, 9-point inverse butterflies, both blocks written to LDS.
template <int R = 9>
__device__ __forceinline__ void inverse_pair_merge(float2 *row, int S, int N, int kp, int u1, int u2,
                                                   const float2 (&Av)[9], const float2 (&Bv)[9])
{
    // Av / Bv: [0..4] = block k' at d = 0..4, [5..8] = block S - k' at d = 0..3
    float2 c1[9], c2[9];
    const int k2p = kp == 0 ? 0 : S - kp;
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        if (R == 8 && d == 4 && kp != 0) break;  // radix 8: block k' > 0 holds four stored frequencies
        const float2 A = Av[d], B = Bv[d];
        const int k = kp + S * d;
        c1[d] = add_i(A, B);
        const float2 mir = conj_add_i_conj(A, B);
        if (kp == 0) {
            if (d != 0 && 2 * d != R) c1[R - d] = mir;  // (radix 8: S * 4 is the Nyquist frequency, its own mirror)
        } else if (2 * k != N) {
            c2[R - 1 - d] = mir;
        }
    }
    if (k2p != kp) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float2 A = Av[5 + d], B = Bv[5 + d];
            c2[d] = add_i(A, B);
            c1[R - 1 - d] = conj_add_i_conj(A, B);
        }
        edge_dft<R, true>(c2);
#pragma unroll
        for (int q = 0; q < R; ++q) row[R * u2 + q] = c2[q];
    } else if (kp != 0) {  // k' = S / 2 (even S): the block mirrors onto itself, d <-> R - 1 - d
#pragma unroll
        for (int d = 0; d < 4; ++d) c1[R - 1 - d] = c2[R - 1 - d];
    }
    edge_dft<R, true>(c1);
#pragma unroll
    for (int q = 0; q < R; ++q) row[R * u1 + q] = c1[q];
}

// spectrum element offsets of the pair's nine frequencies (row term excluded): [0..4] block k', [5..8] block S - k'
__device__ __forceinline__ void pair_offsets(const ConvGeo &g, int S, int kp, int rows_c, int (&off)[9])
{
    const int k2p = kp == 0 ? 0 : S - kp;
#pragma unroll
    for (int d = 0; d < 5; ++d) off[d] = tile_term(g, min(kp + S * d, g.Xc - 1), rows_c);  // (radix 8, k' > 0: the
                                                                                         // fifth is a stand-in)
#pragma unroll
    for (int d = 0; d < 4; ++d) off[5 + d] = tile_term(g, k2p + S * d, rows_c);  // k2p == kp: loaded but unused
}

template <int RSET, int NS = 0, int R = 9>
__global__ __launch_bounds__(NS ? 256 : 320, 4) void k_x_forward_reg(const float *__restrict__ src, float2 *__restrict__ spec,
                                                      ConvGeo g_, AxisPlan ax)
{
    static_assert(R == 9 || (R == 8 && NS == 0), "edge radix 9, or 8 with run-time geometry");
    extern __shared__ float2 lds_fr[];
    const ConvGeo g = static_geo<NS>(g_);
    const int tid = threadIdx.x, nt = NS ? 256 : blockDim.x;
    const int yg0 = (g.in.y0 / g.RB) * g.RB, y_hi = g.in.y0 + g.in.ny, x_hi = g.in.x0 + g.in.nx;
    const int groups = (y_hi - yg0 + g.RB - 1) / g.RB;
    const int z = g.in.z0 + blockIdx.x / groups, y0 = yg0 + (blockIdx.x % groups) * g.RB;
    const int nb = g.RB / 2, N = g.Xp, S = N / R, I = S / 2 + 1;
    // the pair item of the last stage: its two block positions are requested now, used three barriers later
    const int pbi = tid >= I ? 1 : 0, kp = tid - pbi * I;
    const bool pair_live = tid < nb * I;
    int pu1 = 0, pu2 = 0;
    if (pair_live) {
        pu1 = ax.pos[kp];
        pu2 = ax.pos[kp == 0 ? 0 : S - kp];
    }
    XTwiddles xt;
    if constexpr (NS == 2187) xt = x_twiddles_2187(ax.tw, tid);
    if (tid < S) {
        float2 w[9];
        if constexpr (NS == 2187) powers_124(w, xt.e1, xt.e2, xt.e4);
        else edge_twiddles<R>(w, ax.tw, tid);
        const int zs = g.mirror == 1 ? sym_src(z, g.mir.z0, g.mir.nz) : z;
        for (int bi = 0; bi < nb; ++bi) {
            const int ya = y0 + 2 * bi, yb = ya + 1;
            const bool va = ya >= g.in.y0 && ya < y_hi, vb = yb >= g.in.y0 && yb < y_hi;
            const int yas = g.mirror == 1 ? sym_src(min(ya, g.Yp - 1), g.mir.y0, g.mir.ny) : (va ? ya : g.in.y0);
            const int ybs = g.mirror == 1 ? sym_src(min(yb, g.Yp - 1), g.mir.y0, g.mir.ny) : (vb ? yb : g.in.y0);
            const float *ra = src + ((int64_t)zs * g.Yp + yas) * N;
            const float *rbp = src + ((int64_t)zs * g.Yp + ybs) * N;
            float a[9], b[9];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                int n = tid + S * q;
                if (g.mirror) n = sym_src(n, g.mir.x0, g.mir.nx);
                else n = min(max(n, g.in.x0), x_hi - 1);
                a[q] = ra[n];
                b[q] = rbp[n];
            }
            float2 v[9];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int n = tid + S * q;
                const bool vx = g.mirror || (n >= g.in.x0 && n < x_hi);
                v[q] = make_float2(va && vx ? a[q] : 0.0f, vb && vx ? b[q] : 0.0f);
            }
            edge_dft<R, false>(v);
            float2 *row = lds_fr + bi * N + tid;
            row[0] = v[0];
#pragma unroll
            for (int q = 1; q < R; ++q) row[S * q] = cmul(v[q], w[q]);
        }
    }
    __syncthreads();
    if constexpr (NS == 2187) middle_forward_2187<2>(lds_fr, tid, xt);
    else middle_forward<RSET, NS>(lds_fr, ax, nb, N, tid, nt, g.GX);
    {
        float2 c1[9], c2[9];
        if (pair_live) forward_pair_load<R>(lds_fr + pbi * N, pu1 / R, pu2 / R, c1, c2);
        const int ya = y0 + 2 * pbi;
        const int64_t zterm = (int64_t)z * g.plane;
        if (g.TW == 4 && g.RB == 4) {
            __syncthreads();  // every pair has read its blocks: the buffer becomes the line staging
            if (pair_live) forward_pair_emit<true, R>(c1, c2, g, S, kp, spec, zterm, ya, true, true, lds_fr, 2 * pbi);
            __syncthreads();
            unsigned rows_ok = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (y0 + r >= g.in.y0 && y0 + r < y_hi) rows_ok |= 1u << r;
            store_staged_lines(lds_fr, g, spec, zterm, y0, rows_ok, tid, nt);
        } else if (pair_live) {
            forward_pair_emit<false, R>(c1, c2, g, S, kp, spec, zterm, ya, ya >= g.in.y0 && ya < y_hi,
                                        ya + 1 >= g.in.y0 && ya + 1 < y_hi);
        }
    }
}

// Fused tails run on the registers of the last inverse stage: thread t holds x[t + S q] of both packed rows.
// CHAIN (kUpdate, compile-time geometry): the updated estimate's rows go straight into the X forward transform of the
// next iteration's forward-model convolution - symmetric extension along x gathered from the row in LDS, the same
// stages and stores as k_x_forward_reg - written over the spectrum rows this workgroup read (ep.chain == spec).
// The spectrum argument: read-only and restrict-qualified, except in the CHAIN build, which writes the next
// convolution's rows over the rows it has read - there it is a plain pointer the compiler may not assume invariant.
template <bool CHAIN>
struct SpecArg {
    typedef const float2 *__restrict__ type;
};
template <>
struct SpecArg<true> {
    typedef float2 *type;
};
template <int RSET, int EPI, int NS = 0, bool CHAIN = false, int R = 9>
__global__ __launch_bounds__(NS ? 256 : 320, 4) void k_x_inverse_reg(typename SpecArg<CHAIN>::type spec, float *__restrict__ dst,
                                                      ConvGeo g_, AxisPlan ax, m3d::ConvEpilogue ep)
{
#pragma clang fp contract(off)
    static_assert(!CHAIN || (EPI == m3d::ConvEpilogue::kUpdate && NS == 2187), "chain: update epilogue, static geometry");
    static_assert(R == 9 || (R == 8 && NS == 0), "edge radix 9, or 8 with run-time geometry");
    extern __shared__ float2 lds_ir[];
    __shared__ float smf[10];
    __shared__ unsigned long long smu[5];
    const ConvGeo g = static_geo<NS>(g_);
    const int tid = threadIdx.x, nt = NS ? 256 : blockDim.x;
    const int yg0 = (g.out.y0 / g.RB) * g.RB, y_hi = g.out.y0 + g.out.ny, x_hi = g.out.x0 + g.out.nx;
    const int groups = (y_hi - yg0 + g.RB - 1) / g.RB;
    const int z = g.out.z0 + blockIdx.x / groups, y0 = yg0 + (blockIdx.x % groups) * g.RB;
    const int nb = g.RB / 2, N = g.Xp, S = N / R, I = S / 2 + 1;
    // four rows per workgroup: neighbouring lanes take the two row pairs of one frequency pair, so a load instruction
    // touches eight 128-byte lines (64 bytes of each) instead of sixteen (32 bytes of each)
    const int pbi = nb == 2 ? (tid & 1) : 0, kp = nb == 2 ? (tid >> 1) : tid;
    int u1 = 0, u2 = 0;
    XTwiddles xt;
    if constexpr (NS == 2187) xt = x_twiddles_2187(ax.tw, tid);
    if (tid < nb * I) {
        u1 = ax.pos[kp];
        u2 = ax.pos[kp == 0 ? 0 : S - kp];
        const int ya = y0 + 2 * pbi;  // rows outside the box were not produced by the Y pass
        const bool va = ya >= g.out.y0 && ya < y_hi, vb = ya + 1 >= g.out.y0 && ya + 1 < y_hi;
        // an invalid row is read from its valid partner (and zeroed below), so both rows lie in one chunk
        const int yac = va ? ya : (vb ? ya + 1 : g.out.y0), ybc = vb ? ya + 1 : yac;
        int rows_c, off[9];
        const int rba = row_base(g, yac, rows_c), rbb = rba + ((ybc - yac) << g.lTW);
        pair_offsets(g, S, kp, rows_c, off);
        const float2 *sa = spec + (int64_t)z * g.plane + rba;
        const float2 *sb = spec + (int64_t)z * g.plane + rbb;
        float2 Av[9], Bv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Av[i] = Bv[i] = make_float2(0.f, 0.f);
        if (va || vb) {  // a pair wholly outside the box is zero: no load (in the CHAIN build another workgroup may be
                         // rewriting the row this one would otherwise read as a stand-in)
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                Av[i] = sa[off[i]];
                Bv[i] = sb[off[i]];
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (!va) Av[i] = make_float2(0.f, 0.f);
                if (!vb) Bv[i] = make_float2(0.f, 0.f);
            }
        }
        inverse_pair_merge<R>(lds_ir + pbi * N, S, N, kp, u1 / R, u2 / R, Av, Bv);
    }
    __syncthreads();
    if constexpr (NS == 2187) middle_inverse_2187<2>(lds_ir, tid, xt);
    else middle_inverse<RSET, NS>(lds_ir, ax, nb, N, tid, nt, g.GX);
    float mx = -INFINITY, dmax = 0.0f;
    unsigned int updated = 0;
    if (tid < S) {
        float2 w[9];
        // (the chained update pass is at its register limit: its edge twiddles come from the table where they are used)
        if constexpr (NS == 2187 && !CHAIN) powers_124(w, xt.e1, xt.e2, xt.e4);
        else edge_twiddles<R>(w, ax.tw, tid);
        for (int bi = 0; bi < nb; ++bi) {
            const int yr[2] = {y0 + 2 * bi, y0 + 2 * bi + 1};
            bool valid[2];
            int64_t rowo[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                valid[r] = yr[r] >= g.out.y0 && yr[r] < y_hi;
                rowo[r] = ((int64_t)z * g.Yp + (valid[r] ? yr[r] : g.out.y0)) * N;
            }
            // the tail's operands of both rows (36 registers) are requested in front of the last stage, not one row at a
            // time behind it as in round 2, when the kernel sat at its register limit (1.27 -> 1.22 ms)
            float ava[2][9], bva[2][9];
            if constexpr (EPI != m3d::ConvEpilogue::kNone) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        const int n = min(max(tid + S * q, g.out.x0), x_hi - 1);
                        ava[r][q] = ep.a[rowo[r] + n];
                        bva[r][q] = ep.b[rowo[r] + n];
                    }
            }
            float2 v[9];
            const float2 *row = lds_ir + bi * N + tid;
            v[0] = row[0];
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul_conj(row[S * q], w[q]);
            edge_dft<R, true>(v);
            float2 pk[9];  // CHAIN: (new row 2 bi, new row 2 bi + 1) at x = tid + S q; zero outside the core
#pragma unroll
            for (int q = 0; q < 9; ++q) pk[q] = make_float2(0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (!valid[r]) continue;
                float av[9], bv[9];
                if constexpr (EPI != m3d::ConvEpilogue::kNone) {
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        av[q] = ava[r][q];
                        bv[q] = bva[r][q];
                    }
                }
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int n = tid + S * q;
                    if (n < g.out.x0 || n >= x_hi) continue;
                    const float c = r ? v[q].y : v[q].x;
                    if constexpr (EPI == m3d::ConvEpilogue::kNone) {
                        dst[rowo[r] + n] = c;
                    } else if constexpr (EPI == m3d::ConvEpilogue::kUpdate) {
                        const float old = av[q];
                        const float nw = c < 0.0f ? old : old * bv[q];
                        ep.o1[rowo[r] + n] = nw;
                        updated += (c >= 0.0f) ? 1u : 0u;
                        mx = fmaxf(mx, nw);
                        dmax = fmaxf(dmax, fabsf(nw - old));
                        if constexpr (CHAIN) {
                            if (r == 0) pk[q].x = nw;
                            else pk[q].y = nw;
                        }
                    }
                }
            }
            if constexpr (CHAIN) {  // the slots this thread just read
                float2 *rw = lds_ir + bi * N + tid;
#pragma unroll
                for (int q = 0; q < 9; ++q) rw[S * q] = pk[q];
            }
        }
    }
    if constexpr (CHAIN) {
#pragma clang fp contract(fast)
        __syncthreads();
        float2 fv[2][9];
        if (tid < S) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int q = 0; q < 9; ++q)
                    fv[bi][q] = lds_ir[bi * N + sym_src(tid + S * q, g.out.x0, g.out.nx)];
        }
        __syncthreads();  // every gather before the first stage overwrites the rows
        if (tid < S) {
            float2 w[9];
            edge_twiddles(w, ax.tw, tid);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                dft9<false>(fv[bi]);
                float2 *row = lds_ir + bi * N + tid;
                row[0] = fv[bi][0];
#pragma unroll
                for (int q = 1; q < 9; ++q) row[S * q] = cmul(fv[bi][q], w[q]);
            }
        }
        __syncthreads();
        middle_forward_2187<2>(lds_ir, tid, xt);
        {
            float2 c1[9], c2[9];
            const bool pair_live = tid < nb * I;
            if (pair_live) forward_pair_load(lds_ir + pbi * N, u1 / 9, u2 / 9, c1, c2);
            const int64_t zterm = (int64_t)z * g.plane;
            __syncthreads();  // every pair has read its blocks: the buffer becomes the line staging
            float2 *chain;  // ep.chain == spec (checked by the host)
            if constexpr (CHAIN) chain = spec;
            else chain = nullptr;
            if (pair_live) forward_pair_emit<true>(c1, c2, g, S, kp, chain, zterm, y0 + 2 * pbi, true, true, lds_ir, 2 * pbi);
            __syncthreads();
            unsigned rows_ok = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (y0 + r >= g.out.y0 && y0 + r < y_hi) rows_ok |= 1u << r;
            store_staged_lines(lds_ir, g, chain, zterm, y0, rows_ok, tid, nt);
        }
    }
    if constexpr (EPI == m3d::ConvEpilogue::kUpdate) {
        // one barrier for the three block reductions (the count is below 2^24 per wave: a 32-bit shuffle sum)
        const unsigned int upw = m3d::wave_sum(updated);
        mx = m3d::wave_max(mx);
        dmax = m3d::wave_max(dmax);
        if ((tid & 63) == 0) {
            smu[tid >> 6] = upw;
            smf[tid >> 6] = mx;
            smf[5 + (tid >> 6)] = dmax;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned int *red = red_slot(ep.red);
            unsigned long long total = smu[0] + smu[1] + smu[2] + smu[3];
            mx = fmaxf(fmaxf(smf[0], smf[1]), fmaxf(smf[2], smf[3]));
            dmax = fmaxf(fmaxf(smf[5], smf[6]), fmaxf(smf[7], smf[8]));
            if (nt > 256) {  // a fifth wave (run-time geometry, S or the block pairs beyond 256)
                total += smu[4];
                mx = fmaxf(mx, smf[4]);
                dmax = fmaxf(dmax, smf[9]);
            }
            if (total) atomicAdd((unsigned long long *)red, total);
            atomicMax(&red[2], m3d::float_key(mx));
            atomicMax(&red[3], __float_as_uint(dmax));
        }
    }
}

// Dual X inverse pass (two spectra -> HTratio1, HTratio2 of two rows) with the consensus arithmetic in registers and,
// CHAIN, the product's X forward transform started from the same registers: the element the last inverse stage hands
// thread t at x = t + S q is the element the first forward stage takes from it.
template <int RSET, bool CHAIN, int NS = 0, int R = 9>
__global__ __launch_bounds__(NS ? 256 : 320, (NS == 0 && R == 8) ? 5 : 4) void k_x_inverse_dual_reg(float2 *__restrict__ spec1,
                                                           const float2 *__restrict__ spec2, ConvGeo g_, AxisPlan ax,
                                                           m3d::ConvEpilogue ep)
{
#pragma clang fp contract(off)
    static_assert(R == 9 || (R == 8 && NS == 0), "edge radix 9, or 8 with run-time geometry");
    extern __shared__ float2 lds_dr[];
    __shared__ float smf[10];
    const ConvGeo g = static_geo<NS>(g_);
    const int tid = threadIdx.x, nt = NS ? 256 : blockDim.x;
    const int N = g.Xp, S = N / R, I = S / 2 + 1;
    const int groups = (g.Yp + 1) / 2;
    // Two consecutive row pairs share every 128-byte line of the tiled spectra (four rows x 32 bytes).  Workgroups b and
    // b + 8 land on the same XCD, so inside each group of 16 the item order is permuted to give those two workgroups the
    // two halves of the same lines: the second one finds them in that XCD's L2 (reads) and completes them there (writes).
    const int nblk = (int)gridDim.x, b = (int)blockIdx.x;
    const int item = (g.xcdpair && b < (nblk & ~15)) ? ((b & ~15) | ((b & 7) << 1) | ((b >> 3) & 1)) : b;
    const int z = item / groups, y0 = (item % groups) * 2;
    // neighbouring lanes take the two rows of one frequency pair, so a load instruction covers 64 contiguous bytes
    // (both rows) of eight 128-byte lines instead of 32 bytes of sixteen
    const int pbi = tid & 1, kp = tid >> 1;
    const bool pair_live = tid < 2 * I;
    int pu1 = 0, pu2 = 0;
    int fu1 = 0, fu2 = 0;  // the chained forward transform's pairs: thread tid < I takes pair tid
    if (CHAIN && tid < I) {
        fu1 = ax.pos[tid] / R;
        fu2 = ax.pos[tid == 0 ? 0 : S - tid] / R;
    }
    XTwiddles xt;
    if constexpr (NS == 2187) xt = x_twiddles_2187(ax.tw, tid);
    if (pair_live) {
        pu1 = ax.pos[kp] / R;
        pu2 = ax.pos[kp == 0 ? 0 : S - kp] / R;
        int rows_c, off[9];
        const int rb = row_base(g, min(y0 + pbi, g.Yp - 1), rows_c);
        pair_offsets(g, S, kp, rows_c, off);
        const int64_t rbase = (int64_t)z * g.plane + rb;
        float2 Av[9], Bv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            Av[i] = spec1[rbase + off[i]];
            Bv[i] = spec2[rbase + off[i]];
        }
        inverse_pair_merge<R>(lds_dr + pbi * N, S, N, kp, pu1, pu2, Av, Bv);
    }
    __syncthreads();
    if constexpr (NS == 2187) middle_inverse_2187<2>(lds_dr, tid, xt);
    else middle_inverse<RSET, NS>(lds_dr, ax, 2, N, tid, nt, g.GX);
    // 1 / update_norm of both rows.  Requested here, not in front of the middle stages: 18 registers held across them
    // cost more than the exposed latency of these coalesced loads (round 2: spills, 3.23 -> 3.04 ms; round 3, with 24
    // registers to spare: 2.90 vs 2.81 ms)
    const bool row1 = y0 + 1 < g.Yp;
    const int64_t r0 = ((int64_t)z * g.Yp + y0) * N, r1 = r0 + (row1 ? N : 0);
    float un0[9], un1[9];
    if (tid < S) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            un0[q] = ep.b[r0 + tid + S * q];
            un1[q] = ep.b[r1 + tid + S * q];
        }
    }
    float mn = INFINITY, mx = -INFINITY;
    if (tid < S) {
        float2 w[9];
        if constexpr (NS == 2187) powers_124(w, xt.e1, xt.e2, xt.e4);
        else edge_twiddles<R>(w, ax.tw, tid);
        const bool z_core = z >= ep.core.z0 && z < ep.core.z0 + ep.core.nz;
        float2 pk[9];
        // one row after the other: both rows' butterflies in flight at once cost 40 registers more than there are
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float2 v[9];  // (HTratio1 raw, HTratio2 raw) of row y0 + r at x = tid + S q
            const float2 *row = lds_dr + r * N + tid;
            v[0] = row[0];
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul_conj(row[S * q], w[q]);
            edge_dft<R, true>(v);
            const bool live = r == 0 || row1;
            const bool yc = z_core && live && y0 + r >= ep.core.y0 && y0 + r < ep.core.y0 + ep.core.ny;
            const int64_t rr = r ? r1 : r0;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int n = tid + S * q;
                const float un = r ? un1[q] : un0[q];
                const float h1 = v[q].x * un, h2 = v[q].y * un;  // un = 1 / update_norm
                const float prod = (h1 - 1.0f) * (h2 - 1.0f);
                if (live) {
                    const float hs = h1 + h2;
                    mn = fminf(mn, hs);
                    mx = fmaxf(mx, hs);
                    if (yc && n >= ep.core.x0 && n < ep.core.x0 + ep.core.nx) ep.o2[rr + n] = hs;
                    if constexpr (!CHAIN) ep.o1[rr + n] = prod;
                }
                if (r == 0) pk[q] = make_float2(prod, 0.0f);
                else if (row1) pk[q].y = prod;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (CHAIN) {
            // first forward stage of the packed product rows, from the registers that just produced them; the LDS
            // slots written are the ones this thread read above
#pragma clang fp contract(fast)
            edge_dft<R, false>(pk);
            float2 *orow = lds_dr + tid;
            orow[0] = pk[0];
#pragma unroll
            for (int q = 1; q < R; ++q) orow[S * q] = cmul(pk[q], w[q]);
        }
    }
    // per-wave extrema ride the barrier the chained forward needs anyway
    mn = m3d::wave_min(mn);
    mx = m3d::wave_max(mx);
    if ((tid & 63) == 0) {
        smf[tid >> 6] = mn;
        smf[5 + (tid >> 6)] = mx;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned int *red = red_slot(ep.red);
        mn = fminf(fminf(smf[0], smf[1]), fminf(smf[2], smf[3]));
        mx = fmaxf(fmaxf(smf[5], smf[6]), fmaxf(smf[7], smf[8]));
        if (nt > 256) {  // a fifth wave (run-time geometry, S or the block pairs beyond 256)
            mn = fminf(mn, smf[4]);
            mx = fmaxf(mx, smf[9]);
        }
        {
            atomicMin(&red[0], m3d::float_key(mn));
            atomicMax(&red[1], m3d::float_key(mx));
        }
    }
    if constexpr (CHAIN) {
        if constexpr (NS == 2187) middle_forward_2187<1>(lds_dr, tid, xt);
        else middle_forward<RSET, NS>(lds_dr, ax, 1, N, tid, nt, g.GX);
        if (tid < I) {
            float2 c1[9], c2[9];
            forward_pair_load<R>(lds_dr, fu1, fu2, c1, c2);
            forward_pair_emit<false, R>(c1, c2, g, S, tid, spec1, (int64_t)z * g.plane, y0, true, row1);
        }
    }
}

// ---------------------------------------------------------------- X inverse of Hu + split + two X forward transforms
// One row pair (y, y + 1) of one core plane per workgroup; static geometry (2187 = 9 * 9 * 3 * 9) only.  LDS: two
// rows of N complex values - T carries the packed inverse transform of the pair and then ratio1's packed forward
// transform, P ratio2's.  The split arithmetic is k_split_ratio's (m3d_rlgc.hip), operation for operation.
template <typename T>
__device__ __forceinline__ float load_count(const T *p, int64_t i)
{
    return (float)p[i];
}

// NS = 2187: compile-time geometry (the headline size).  NS = 0: run-time geometry for any length the register-edge
// kernels take (edge radix R = 9 or 8, four rows per row group, tiles of four columns); same arithmetic.
template <typename T, int NS = 2187, int R = 9, int RSET = 1>
__global__ __launch_bounds__(NS ? 256 : 320, 4) void k_x_inverse_split_chain(float2 *spec2, float2 *__restrict__ spec1, ConvGeo g_,
                                                              AxisPlan ax, m3d::SplitArgs sa)
{
#pragma clang fp contract(off)
    static_assert(NS == 0 || (NS == 2187 && R == 9), "static geometry: 2187 = 9 * 9 * 3 * 9");
    extern __shared__ float2 lds_sp[];
    __shared__ double smd[5][7];
    // the sampler's guide for the rows of dim voxels (128 to 575 counts), 3.5 KB: four workgroups of 34.2 + 3.8 KB still
    // share a CU's 160 KB.  Filled here, visible after the barrier behind the spectrum merge.
    __shared__ uint16_t lguide[m3d::kBinomGuideEntries];
    const ConvGeo g = static_geo<NS>(g_);
    const int N = NS ? NS : g.Xp, S = N / R, I = S / 2 + 1;
    const int tid = threadIdx.x, nt = NS ? 256 : (int)blockDim.x;
    static_assert(m3d::kBinomGuideEntries % 8 == 0, "staged in 16-byte pieces");
    if (!sa.injected)
        for (int i = tid; i < (int)m3d::kBinomGuideEntries / 8; i += nt)
            ((uint4 *)lguide)[i] = ((const uint4 *)sa.tables.guide)[i];
    const int y_hi = g.out.y0 + g.out.ny, x_hi = g.out.x0 + g.out.nx;
    const int yg0 = (g.out.y0 / 4) * 4;  // the pairing of the other X passes (four rows per group, pairs inside)
    const int pairs = (y_hi - yg0 + 1) / 2;
    const int nblk = (int)gridDim.x, b = (int)blockIdx.x;
    // workgroups b and b + 8 share an XCD: give them the two row pairs that share 128-byte lines (see the dual pass)
    const int item = (g.xcdpair && b < (nblk & ~15)) ? ((b & ~15) | ((b & 7) << 1) | ((b >> 3) & 1)) : b;
    const int z = g.out.z0 + item / pairs, ya = yg0 + (item % pairs) * 2;
    const bool va = ya >= g.out.y0 && ya < y_hi, vb = ya + 1 >= g.out.y0 && ya + 1 < y_hi;
    const int pbi = tid >= I ? 1 : 0, kp = tid - pbi * I;
    const bool pair_live = tid < 2 * I;
    int pu1 = 0, pu2 = 0;
    if (pair_live) {
        pu1 = ax.pos[kp] / R;
        pu2 = ax.pos[kp == 0 ? 0 : S - kp] / R;
    }
    const int64_t zterm = (int64_t)z * g.plane;
    if (tid < I) {
        float2 Av[9], Bv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Av[i] = Bv[i] = make_float2(0.f, 0.f);
        if (va || vb) {  // the first / last item of a plane may lie wholly outside the box (the pairing starts on a
                         // multiple of four rows): nothing is read for it and nothing stored
            const int yac = va ? ya : ya + 1, ybc = vb ? ya + 1 : yac;
            int rows_c, off[9];
            const int rba = row_base(g, yac, rows_c), rbb = rba + ((ybc - yac) << g.lTW);
            pair_offsets(g, S, kp, rows_c, off);
            const float2 *sa_ = spec2 + zterm + rba, *sb_ = spec2 + zterm + rbb;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                Av[i] = sa_[off[i]];
                Bv[i] = sb_[off[i]];
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (!va) Av[i] = make_float2(0.f, 0.f);
                if (!vb) Bv[i] = make_float2(0.f, 0.f);
            }
        }
        inverse_pair_merge<R>(lds_sp, S, N, kp, pu1, pu2, Av, Bv);
    }
    __syncthreads();
    // (table twiddles: this kernel is bound by its vector arithmetic - Philox, the threshold search, the split - and
    // holds 128 registers; the register twiddles of the other static X kernels cost it 6 %)
    middle_inverse<RSET, NS>(lds_sp, ax, 1, N, tid, nt, g.GX);
    const float eps = 1e-4f;
    const uint32_t iteration = sa.iteration;
    float sums[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // A, Bim, B1, B2, Tim, T1, T2 of this thread's 18 voxels
    if (tid < S) {
        const T *image = (const T *)sa.image;
        const int64_t plane0 = (int64_t)(z - g.out.z0) * g.out.ny;
        // counts (and injected splits) of both rows: requested before the last inverse stage; rows and columns outside
        // the box are clamped into it (their values are not used)
        // (a uint16 image's count is the draw's n as it stands: kept as an integer, converted once for the arithmetic)
        using Count = typename std::conditional<sizeof(T) == 2, uint32_t, float>::type;
        Count cnt[2][9];
        float inj[2][9];
        int64_t irows[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            irows[r] = (plane0 + min(max(ya + r - g.out.y0, 0), g.out.ny - 1)) * g.out.nx;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int xi = min(max(tid + S * q - g.out.x0, 0), g.out.nx - 1);
                cnt[r][q] = (Count)image[irows[r] + xi];
                inj[r][q] = sa.injected ? sa.injected[irows[r] + xi] : 0.0f;
            }
        }
        float2 w[9];
        edge_twiddles<R>(w, ax.tw, tid);  // (register limit: from the table, behind the image loads)
        float2 v[9];  // (Hu of row ya, Hu of row ya + 1) at x = tid + S q
        const float2 *row = lds_sp + tid;
        v[0] = row[0];
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmul_conj(row[S * q], w[q]);
        edge_dft<R, true>(v);
        float2 p1[9], p2[9];  // (ratio1 of row ya, of row ya + 1), the same for ratio2; zero outside the core
#pragma unroll
        for (int q = 0; q < 9; ++q) p1[q] = p2[q] = make_float2(0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(r ? vb : va)) continue;
            const int64_t irow = irows[r];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int x = tid + S * q;
                if (x < g.out.x0 || x >= x_hi) continue;
                const float h = r ? v[q].y : v[q].x;
                const float val = (float)cnt[r][q];
                float s1;
                if (sa.injected) {
                    s1 = inj[r][q];
                } else {
                    uint32_t n;
                    if constexpr (sizeof(T) == 2) {
                        n = cnt[r][q];
                    } else {
                        const float vt = val > 0.0f ? val : 0.0f;
                        n = (uint32_t)fminf(vt, 1048576.0f);  // astype(int64) truncation, rlgc.py:619
                    }
                    s1 = (float)m3d::binomial_half<!(sizeof(T) == 2)>(n, (uint64_t)(irow + (x - g.out.x0)), iteration,
                                                                      sa.seed, sa.tables, sizeof(T) == 2 ? lguide : nullptr);
                }
                const float s2 = val - s1;
                const float a = h + eps;  // kl_div, rlgc.py:401-412, as raw sums
                const float bim = val + eps, b1 = s1 + eps, b2 = s2 + eps;
                sums[0] += a;
                sums[1] += bim;
                sums[2] += b1;
                sums[3] += b2;
                if (a > 0.0f) {
                    const float la = __builtin_amdgcn_logf(a);
                    sums[4] += a * (la - __builtin_amdgcn_logf(bim));
                    sums[5] += a * (la - __builtin_amdgcn_logf(b1));
                    sums[6] += a * (la - __builtin_amdgcn_logf(b2));
                }
                // mask * split / (0.5 * (Hu + 1e-12)), rlgc.py:654-668: one reciprocal serves both quotients.  v_rcp_f32 (1 ulp)
                // instead of the correctly rounded quotient's fourteen instructions: the transforms that consume the
                // ratios round coarser, the dual pass multiplies by 1 / update_norm the same way, and this kernel is bound
                // by its vector instructions
                const float inv = __builtin_amdgcn_rcpf(0.5f * (h + 1e-12f));
                if (r == 0) {
                    p1[q].x = s1 * inv;
                    p2[q].x = s2 * inv;
                } else {
                    p1[q].y = s1 * inv;
                    p2[q].y = s2 * inv;
                }
            }
        }
        {   // first forward stage of both packed ratio rows, into the slots this thread read (T) and their twins (P)
#pragma clang fp contract(fast)
            edge_dft<R, false>(p1);
            float2 *o1 = lds_sp + tid;
            o1[0] = p1[0];
#pragma unroll
            for (int q = 1; q < R; ++q) o1[S * q] = cmul(p1[q], w[q]);
            edge_dft<R, false>(p2);
            float2 *o2 = lds_sp + N + tid;
            o2[0] = p2[0];
#pragma unroll
            for (int q = 1; q < R; ++q) o2[S * q] = cmul(p2[q], w[q]);
        }
    }
    // the seven sums: wave shuffles, then four values per sum through LDS; they ride the barrier the transforms need
    {
        double d[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) d[k] = m3d::wave_sum((double)sums[k]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k) smd[tid >> 6][k] = d[k];
        }
    }
    __syncthreads();
    if (tid == 0) {
        const double ln2 = 0.69314718055994530942;
        double t[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            t[k] = ((smd[0][k] + smd[1][k]) + smd[2][k]) + smd[3][k];
            if (nt > 256) t[k] += smd[4][k];  // a fifth wave (run-time geometry, more than 256 edge items or pairs)
        }
        m3d::SplitPartial out = {t[0], t[1], t[2], t[3], t[4] * ln2, t[5] * ln2, t[6] * ln2};
        sa.partial[blockIdx.x] = out;
    }
    {
#pragma clang fp contract(fast)
        middle_forward<RSET, NS>(lds_sp, ax, 2, N, tid, nt, g.GX);
        if (pair_live) {
            float2 c1[9], c2[9];
            forward_pair_load<R>(lds_sp + pbi * N, pu1, pu2, c1, c2);
            forward_pair_emit<false, R>(c1, c2, g, S, kp, pbi ? spec2 : spec1, zterm, ya, va, vb);
        }
    }
}

// ---------------------------------------------------------------- Y pass (in place)
template <bool INV, int RSET>
__global__ __launch_bounds__(1024) void k_y_pass(float2 *__restrict__ spec, ConvGeo g, AxisPlan ay)
{
    extern __shared__ float2 lds_y[];
    const int tid = threadIdx.x, nt = blockDim.x;
    // forward: only the input's z planes hold data, rows outside its y range are zero;
    // inverse: only the output's z planes are transformed and only its y rows are stored
    const m3d::ConvBox &bx = INV ? g.out : g.in;
    const int z = bx.z0 + blockIdx.x / g.nT, kt = blockIdx.x % g.nT;
    float2 *pl = spec + (int64_t)z * g.plane;  // the tile's rows come in runs of one y chunk (tile_elem)
    const int n = g.Yp * g.TW;
    const int lo = INV ? 0 : bx.y0 * g.TW, hi = INV ? n : (bx.y0 + bx.ny) * g.TW;  // load range
    const int slo = INV ? bx.y0 * g.TW : 0, shi = INV ? (bx.y0 + bx.ny) * g.TW : n;  // store range
    if ((g.TW & 1) == 0) {  // 16-byte copies (rows are 16-B aligned when TW is even)
        float4 *l4 = (float4 *)lds_y;
        constexpr int kU = M3D_KU_Y;  // independent 16-byte loads in flight per thread (see k_z_conv)
        for (int i0 = tid; i0 < n / 2; i0 += nt * kU) {
            float4 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u)
                v[u] = ld_stream((const float4 *)(pl + tile_elem(g, kt, 2 * min(max(i0 + u * nt, lo / 2), hi / 2 - 1))));
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i = i0 + u * nt;
                if (i < n / 2) l4[i] = (i >= lo / 2 && i < hi / 2) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    } else {
        for (int i = tid; i < n; i += nt)
            lds_y[i] = (i >= lo && i < hi) ? pl[tile_elem(g, kt, i)] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    lds_fft<INV, RSET>(lds_y, ay, g.TW, g.TW, 1, tid, nt, g.GY);
    if ((g.TW & 1) == 0) {
        float4 *l4 = (float4 *)lds_y;
        for (int i = slo / 2 + tid; i < shi / 2; i += nt) st_stream((float4 *)(pl + tile_elem(g, kt, 2 * i)), l4[i]);
    } else {
        for (int i = slo + tid; i < shi; i += nt) pl[tile_elem(g, kt, i)] = lds_y[i];
    }
}

// ---------------------------------------------------------------- Y pass, static geometry
// Counters (profiles/r02_pmc_sq_rlgc_before.json) put every pass at about one wave instruction of ANY kind per four
// cycles per SIMD: (VALU + SALU + LDS + VMEM instructions) x 4 / 1024 SIMDs reproduces the kernel durations within
// 10 %.  The passes are instruction-issue bound, and in the generic kernels more than half of the instructions are
// index arithmetic on run-time geometry (k_y_pass: 821 scalar + ~250 integer vector instructions around ~350
// floating-point ones).  For the headline geometry the whole pass is compiled with every extent, stride, radix and
// chunk size as a constant: the stage loops unroll, divisions become shifts and immediates.
template <int R, bool INV, int N, int M, int TWc, int NT_, int CPT_ = 1>
__device__ __forceinline__ void y_stage_static(float2 *data, const float2 *__restrict__ tw, int tid)
{
    // four interleaved transforms (TW = 4): element k of transform b at data[k * TW + b]
    constexpr int sub = M / R, per_fft = N / R, twstep = N / M, step = sub * TWc;
    // One (butterfly, column group) per work item.  CPT = 1: every column its own item (four times the items, twiddles
    // recomputed per column; round 2).  CPT = 2: two adjacent columns per item - twiddles and index arithmetic shared,
    // and the two columns of a row are 16 contiguous bytes, one ds_read_b128 / ds_write_b128 instead of two 8-byte
    // accesses; same operations, same order, same twiddle values per element.  Measured (round 3,
    // profiles/r03_y_pass_columns.txt): the inverse pass, whose item fits 62 registers, gains 4 % (731 -> 700 us); the
    // forward passes need 80-88 registers for it (twiddles applied after the butterfly keep both columns live), which
    // costs the second 1024-thread block per CU (spills at 64: 2.7 ms) or, with 768-thread blocks, gives back what the
    // saved instructions bring (647 vs 655 us; the mirror pass 570 vs 483 us) - they stay at one column per item.
    // Requesting the three radix-9 stages' twiddles at kernel start, in front of the tile's loads (18 registers), moved
    // nothing either (643 vs 650 us forward, 500 vs 480 us mirror, inverse 705 vs 704 us with one column per item): the Y
    // pass does not wait for its table loads.
    // Radix 27 (the last forward / first inverse stage of 2187 = 9 * 9 * 27): blocks of 27 elements, every twiddle is 1;
    // 54 registers of data per column, so one column per item.
    static_assert(R != 27 || sub == 1, "radix 27 only as the twiddle-free stage of block size 27");
    constexpr int CPT = (TWc % CPT_ == 0 && R != 27) ? CPT_ : 1, GR = TWc / CPT;
#pragma unroll 1
    for (int t = tid; t < per_fft * GR; t += NT_) {
        const int c = (t % GR) * CPT, it = t / GR;
        const int blk = it / sub, j = it - blk * sub;
        float2 w[R];
        const int j1 = j * twstep;
        constexpr bool kUnit = sub == 1;  // block size = radix: every twiddle is W^0 = 1, the products are skipped
        if constexpr (kUnit) {
        } else if constexpr (R == 9) {
            w[1] = tw[j1];
            w[2] = tw[2 * j1];
            w[4] = tw[4 * j1];
            w[3] = cmul(w[1], w[2]);
            w[5] = cmul(w[1], w[4]);
            w[6] = cmul(w[2], w[4]);
            w[7] = cmul(w[3], w[4]);
            w[8] = cmul(w[4], w[4]);
        } else if constexpr (R == 8) {
            w[1] = tw[j1];
            w[2] = tw[2 * j1];
            w[4] = tw[4 * j1];
            w[3] = cmul(w[1], w[2]);
            w[5] = cmul(w[1], w[4]);
            w[6] = cmul(w[2], w[4]);
            w[7] = cmul(w[3], w[4]);
        } else {
            w[1] = tw[j1];
            if constexpr (R >= 3) w[2] = cmul(w[1], w[1]);
            if constexpr (R >= 4) w[3] = cmul(w[1], w[2]);
        }
        float2 *p = data + (blk * M + j) * TWc + c;
        float2 v[CPT][R];
        if constexpr (CPT == 2) {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const float4 two = *(const float4 *)(p + q * step);
                v[0][q] = make_float2(two.x, two.y);
                v[1][q] = make_float2(two.z, two.w);
            }
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q)
#pragma unroll
                for (int u = 0; u < CPT; ++u) v[u][q] = p[q * step + u];
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            if (INV) {
                if constexpr (!kUnit) {
#pragma unroll
                    for (int q = 1; q < R; ++q) v[u][q] = cmul_conj(v[u][q], w[q]);
                }
                dft_r<R, true>(v[u]);
            } else {
                dft_r<R, false>(v[u]);
                if constexpr (!kUnit) {
#pragma unroll
                    for (int q = 1; q < R; ++q) v[u][q] = cmul(v[u][q], w[q]);
                }
            }
        }
        if constexpr (CPT == 2) {
#pragma unroll
            for (int q = 0; q < R; ++q)
                *(float4 *)(p + q * step) = make_float4(v[0][q].x, v[0][q].y, v[1][q].x, v[1][q].y);
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q)
#pragma unroll
                for (int u = 0; u < CPT; ++u) p[q * step + u] = v[u][q];
        }
    }
}

// YP = R0 * R1 * R2 * R3 (R3 may be 1), TW = TWC (4; 2 for 2592, whose tile of four columns would be 83 KB), 2^LCH rows per
// chunk, NT_ threads.
template <bool INV, int YP, int R0, int R1, int R2, int R3, int LCH, int NT_, int TWC = 4>
__global__ __launch_bounds__(NT_) void k_y_pass_static(float2 *__restrict__ spec, ConvGeo g, const float2 *__restrict__ tw)
{
    extern __shared__ float2 lds_ys[];
    constexpr int TWc = TWC, n = YP * TWc, CH = 1 << LCH, IPR = TWC / 2;  // IPR: 16-byte items per row
    const int tid = threadIdx.x;
    const m3d::ConvBox &bx = INV ? g.out : g.in;
    const int z = bx.z0 + blockIdx.x / g.nT, kt = blockIdx.x % g.nT;
    float2 *pl = spec + (int64_t)z * g.plane;
    const int lo = INV ? 0 : bx.y0 * TWc, hi = INV ? n : (bx.y0 + bx.ny) * TWc;    // load range (elements)
    const int slo = INV ? bx.y0 * TWc : 0, shi = INV ? (bx.y0 + bx.ny) * TWc : n;  // store range
    auto elem = [&](int i) {  // 16-byte item i (half a row at TW = 4, a row at TW = 2) of this tile -> in-plane element offset
        const int y = i / IPR, y0c = y & ~(CH - 1);
        const int rows_c = min(CH, YP - y0c);
        return (y0c * g.nT + (y - y0c)) * TWc + (kt * rows_c) * TWc + ((i % IPR) << 1);
    };
    float4 *l4 = (float4 *)lds_ys;
    constexpr int kU = M3D_KU_Y;
    for (int i0 = tid; i0 < n / 2; i0 += NT_ * kU) {
        float4 v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = ld_stream((const float4 *)(pl + elem(min(max(i0 + u * NT_, lo / 2), hi / 2 - 1))));
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = i0 + u * NT_;
            if (i < n / 2) l4[i] = (i >= lo / 2 && i < hi / 2) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    if (!INV) {
        // two columns per item in the forward radix-9 stages where the registers are there anyway (9 * 9 * 27: the
        // radix-27 stage sets the kernel's 70; forward 592 -> 583 us) - elsewhere they cost the second block per CU
        constexpr int FC = R2 == 27 ? 2 : 1;
        y_stage_static<R0, false, YP, YP, TWc, NT_, FC>(lds_ys, tw, tid);
        __syncthreads();
        y_stage_static<R1, false, YP, YP / R0, TWc, NT_, FC>(lds_ys, tw, tid);
        __syncthreads();
        y_stage_static<R2, false, YP, YP / (R0 * R1), TWc, NT_>(lds_ys, tw, tid);
        __syncthreads();
        if constexpr (R3 > 1) {
            y_stage_static<R3, false, YP, YP / (R0 * R1 * R2), TWc, NT_>(lds_ys, tw, tid);
            __syncthreads();
        }
    } else {
        if constexpr (R3 > 1) {
            y_stage_static<R3, true, YP, R3, TWc, NT_, 2>(lds_ys, tw, tid);
            __syncthreads();
        }
        y_stage_static<R2, true, YP, R3 * R2, TWc, NT_, 2>(lds_ys, tw, tid);
        __syncthreads();
        y_stage_static<R1, true, YP, R3 * R2 * R1, TWc, NT_, 2>(lds_ys, tw, tid);
        __syncthreads();
        y_stage_static<R0, true, YP, YP, TWc, NT_, 2>(lds_ys, tw, tid);
        __syncthreads();
    }
    for (int i = slo / 2 + tid; i < shi / 2; i += NT_) st_stream((float4 *)(pl + elem(i)), l4[i]);
}

// Static-geometry form of k_y_forward_mirror (below): core rows of plane sym_src(z) -> all rows of plane z.
template <int YP, int R0, int R1, int R2, int R3, int LCH, int NT_, int TWC = 4>
__global__ __launch_bounds__(NT_) void k_y_forward_mirror_static(const float2 *__restrict__ src, float2 *__restrict__ dst,
                                                                ConvGeo g, const float2 *__restrict__ tw)
{
    extern __shared__ float2 lds_yms[];
    constexpr int TWc = TWC, n = YP * TWc, CH = 1 << LCH, IPR = TWC / 2;
    const int tid = threadIdx.x;
    const int z = (g.zmirror ? g.mir.z0 : 0) + blockIdx.x / g.nT, kt = blockIdx.x % g.nT;
    const int zs = sym_src(z, g.mir.z0, g.mir.nz);
    const float2 *spl = src + (int64_t)zs * g.plane;
    float2 *dpl = dst + (int64_t)z * g.plane;
    const int lo = g.mir.y0 * TWc, hi = (g.mir.y0 + g.mir.ny) * TWc;
    auto elem = [&](int i) {
        const int y = i / IPR, y0c = y & ~(CH - 1);
        const int rows_c = min(CH, YP - y0c);
        return (y0c * g.nT + (y - y0c)) * TWc + (kt * rows_c) * TWc + ((i % IPR) << 1);
    };
    float4 *l4 = (float4 *)lds_yms;
    constexpr int kU = M3D_KU_Y;
    for (int i0 = lo / 2 + tid; i0 < hi / 2; i0 += NT_ * kU) {
        float4 v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = ld_stream((const float4 *)(spl + elem(min(i0 + u * NT_, hi / 2 - 1))));
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (i0 + u * NT_ < hi / 2) l4[i0 + u * NT_] = v[u];
    }
    __syncthreads();
    // rows outside the core: copies of the core rows they mirror (single reflection when the pad is no deeper than
    // the core, which the host checks before choosing this kernel)
    const int y_hi = g.mir.y0 + g.mir.ny;
    for (int i = tid; i < n / 2; i += NT_) {
        const int y = i / IPR;
        if (y >= g.mir.y0 && y < y_hi) continue;
        const int ys = y < g.mir.y0 ? 2 * g.mir.y0 - 1 - y : 2 * y_hi - 1 - y;
        l4[i] = l4[ys * IPR + (i % IPR)];
    }
    __syncthreads();
    y_stage_static<R0, false, YP, YP, TWc, NT_>(lds_yms, tw, tid);
    __syncthreads();
    y_stage_static<R1, false, YP, YP / R0, TWc, NT_>(lds_yms, tw, tid);
    __syncthreads();
    y_stage_static<R2, false, YP, YP / (R0 * R1), TWc, NT_>(lds_yms, tw, tid);
    __syncthreads();
    if constexpr (R3 > 1) {
        y_stage_static<R3, false, YP, YP / (R0 * R1 * R2), TWc, NT_>(lds_yms, tw, tid);
        __syncthreads();
    }
    for (int i = tid; i < n / 2; i += NT_) st_stream((float4 *)(dpl + elem(i)), l4[i]);
}

// Y forward pass of a volume given by its core rows only (symmetric extension along y and z): plane z is
// built from the rows of core plane sym_src(z) - X transforms of mirrored rows are copies, so the X pass
// only transforms the core.  Out of place (several planes read the same core plane).
template <int RSET>
__global__ __launch_bounds__(1024) void k_y_forward_mirror(const float2 *__restrict__ src, float2 *__restrict__ dst,
                                                          ConvGeo g, AxisPlan ay)
{
    extern __shared__ float2 lds_ym[];
    const int tid = threadIdx.x, nt = blockDim.x;
    // zmirror: only the core planes are built (the Z pass reflects them); else every plane from its source
    const int z = (g.zmirror ? g.mir.z0 : 0) + blockIdx.x / g.nT, kt = blockIdx.x % g.nT;
    const int zs = sym_src(z, g.mir.z0, g.mir.nz);
    const float2 *spl = src + (int64_t)zs * g.plane;
    float2 *dpl = dst + (int64_t)z * g.plane;
    const int n = g.Yp * g.TW;
    const int lo = g.mir.y0 * g.TW, hi = (g.mir.y0 + g.mir.ny) * g.TW;
    {   // TW is even on this path: 16-byte copies
        float4 *l4 = (float4 *)lds_ym;
        constexpr int kU = M3D_KU_Y;
        for (int i0 = lo / 2 + tid; i0 < hi / 2; i0 += nt * kU) {
            float4 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) v[u] = *(const float4 *)(spl + tile_elem(g, kt, 2 * min(i0 + u * nt, hi / 2 - 1)));
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (i0 + u * nt < hi / 2) l4[i0 + u * nt] = v[u];
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        if (i >= lo && i < hi) continue;
        const int y = i >> g.lTW, kk = i & (g.TW - 1);
        lds_ym[i] = lds_ym[(sym_src(y, g.mir.y0, g.mir.ny) << g.lTW) + kk];
    }
    __syncthreads();
    lds_fft<false, RSET>(lds_ym, ay, g.TW, g.TW, 1, tid, nt, g.GY);
    float4 *l4 = (float4 *)lds_ym;
    for (int i = tid; i < n / 2; i += nt) *(float4 *)(dpl + tile_elem(g, kt, 2 * i)) = l4[i];
}

// ---------------------------------------------------------------- Z pass
// mode 0: S*H  1: S*conj(H)  2: S*|H|^2  (then * scale), -1: forward only (OTF construction)
template <int RSET>
__global__ __launch_bounds__(256) void k_z_conv(float2 *__restrict__ spec, const float2 *__restrict__ otf, ConvGeo g,
                         AxisPlan az, int mode, float scale)
{
    extern __shared__ float2 lds_z[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * g.P;
    const int P = g.P;
    const int p = tid & (P - 1), z0 = tid >> g.lP, zstep = nt >> g.lP;  // nt is a multiple of P
    const bool live = p0 + p < g.plane;
    // Loads are issued in batches of kU independent requests per thread (clamped addresses,
    // select afterwards): a load -> LDS-store loop with a run-time trip count keeps one request
    // in flight per wave, and the pass is then bound by HBM latency instead of bandwidth.
    constexpr int kU = M3D_KU_Z;
    const int64_t pc = live ? p0 + p : g.plane - 1;
    float2 *col = spec + pc;
    const int zin_hi = g.in.z0 + g.in.nz;
    // The OTF column is requested before anything else so that its latency hides behind the spectrum
    // loads and the forward transform (kH values per thread; longer columns fall back to loading it late).
    constexpr int kH = M3D_KH_Z;
    const float2 *hcol = otf + pc;
    const bool early = kH > 0 && mode >= 0 && g.Zp <= kH * zstep;
    float2 hreg[kH > 0 ? kH : 1];
    if (early) {
#pragma unroll
        for (int i = 0; i < kH; ++i) hreg[i] = hcol[(int64_t)min(z0 + i * zstep, g.Zp - 1) * g.plane];
    }
    if (!g.zmirror)
        for (int z = z0; z < g.Zp; z += zstep)  // planes outside the input's support are zero
            if (z < g.in.z0 || z >= zin_hi) lds_z[(z << g.lP) + p] = make_float2(0.f, 0.f);
    {
        const int first = g.in.z0 + ((z0 - g.in.z0) % zstep + zstep) % zstep;  // first z >= in.z0, = z0 mod zstep
        for (int zb = first; zb < zin_hi; zb += zstep * kU) {
            float2 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int z = min(zb + u * zstep, zin_hi - 1);
                v[u] = col[(int64_t)z * g.plane];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int z = zb + u * zstep;
                if (z < zin_hi) lds_z[(z << g.lP) + p] = live ? v[u] : make_float2(0.f, 0.f);
            }
        }
    }
    if (g.zmirror) {  // symmetric extension along z: the other planes are copies of the loaded ones
        __syncthreads();
        for (int z = z0; z < g.Zp; z += zstep)
            if (z < g.in.z0 || z >= zin_hi) lds_z[(z << g.lP) + p] = lds_z[(sym_src(z, g.in.z0, g.in.nz) << g.lP) + p];
    }
    __syncthreads();
    lds_fft<false, RSET>(lds_z, az, P, P, 1, tid, nt, g.GZ);
    if (mode < 0) {
        for (int z = z0; z < g.Zp; z += zstep)
            if (live) col[(int64_t)z * g.plane] = lds_z[(z << g.lP) + p];
        return;
    }
    if (early) {
#pragma unroll
        for (int i = 0; i < kH; ++i) {
            const int z = z0 + i * zstep;
            if (z >= g.Zp) continue;
            float2 h = hreg[i];
            if (mode == 1) h.y = -h.y;
            if (mode == 2) h = make_float2(h.x * h.x + h.y * h.y, 0.0f);
            const float2 v = cmul(lds_z[(z << g.lP) + p], h);
            lds_z[(z << g.lP) + p] = make_float2(v.x * scale, v.y * scale);
        }
    } else
    for (int zb = z0; zb < g.Zp; zb += zstep * kU) {
        float2 hv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int z = min(zb + u * zstep, g.Zp - 1);
            hv[u] = hcol[(int64_t)z * g.plane];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int z = zb + u * zstep;
            if (z >= g.Zp) continue;
            float2 h = hv[u];
            if (mode == 1) h.y = -h.y;
            if (mode == 2) h = make_float2(h.x * h.x + h.y * h.y, 0.0f);
            const float2 v = cmul(lds_z[(z << g.lP) + p], h);
            lds_z[(z << g.lP) + p] = make_float2(v.x * scale, v.y * scale);
        }
    }
    __syncthreads();
    lds_fft<true, RSET>(lds_z, az, P, P, 1, tid, nt, g.GZ);
    for (int z = z0; z < g.Zp; z += zstep)  // only the planes the output needs
        if (live && z >= g.out.z0 && z < g.out.z0 + g.out.nz) col[(int64_t)z * g.plane] = lds_z[(z << g.lP) + p];
}

// ---------------------------------------------------------------- Z pass, register variant (Zp = 16 L)
// The 7-barrier LDS transform chain of k_z_conv is what bounds it (all four Z passes of an iteration take the same
// time whatever they move).  For Zp = 16 L, L in {2, 4, 8, 16}: thread (p, l) owns the 16 samples z = l + L m of
// column p and transforms them in registers (16-point DFT), one LDS exchange hands thread (p, l) the L partial
// results of frequencies k1 = l, l + L, ... for an L-point DFT across l, the transfer function is applied in
// registers, and the same steps run backwards: two barriers and four LDS passes per tile.  With
//   z = l + L m, k = k1 + 16 k2:  W_N^(z k) = W_16^(m k1) * W_N^(l k1) * W_L^(l k2).
// Spectra never leave the kernel, so only the OTF's plane order matters: natural (plane k holds frequency k), written
// by this kernel's own mode -1.
// ZM: what lies outside the input's planes - 0 zero, 1 symmetric extension (general), 2 symmetric extension with one
// reflection (pad no deeper than the core).  OUTF: every plane of the output is stored (no per-plane test).
template <int L, int P, int ZM = 1, bool OUTF = false, bool REALH = false>
__global__ __launch_bounds__(P * L) M3D_ZREG_ATTR void k_z_conv_reg(float2 *__restrict__ spec, const float2 *__restrict__ otf,
                                                       ConvGeo g, const float2 *__restrict__ tw, int mode, float scale,
                                                       float2 *__restrict__ spec_b = nullptr)
{
    constexpr int N = 16 * L, LP = P == 64 ? 6 : (P == 32 ? 5 : 4);
    static_assert(P == 16 || P == 32 || P == 64, "16, 32 or 64 columns per block");
    constexpr int KS = L * P + P;  // stride between k1 slabs, padded by one row of columns: conflict-free both ways
    constexpr int NBF = 16 / L;    // (k1, column) groups per thread in the cross-thread step
    __shared__ float2 ex[16 * KS];
    const int tid = threadIdx.x, p = tid & (P - 1), l = tid >> LP;
    // spec_b: two spectra take the same transfer function.  Workgroups b and b + 8 run on the same XCD one after the
    // other, so inside each group of 16 the first eight take eight column tiles of `spec` and the second eight the same
    // tiles of `spec_b`: the second reader of every OTF line finds it in that XCD's L2 instead of in HBM.
    int64_t tile = blockIdx.x;
    if (spec_b) {
        const int b = (int)blockIdx.x;
        tile = (int64_t)(b >> 4) * 8 + (b & 7);
        if (tile * P >= g.plane) return;
        if (b & 8) spec = spec_b;
    }
    const int64_t p0 = tile * P;
    const bool live = p0 + p < g.plane;
    const int64_t pc = live ? p0 + p : g.plane - 1;
    float2 *col = spec + pc;
    const int zin_hi = g.in.z0 + g.in.nz;
    (void)N;

    float2 h[16];  // transfer function at k = (l + b L) + 16 k2
    // mode 3: `otf` is a float volume of the same plane order that holds |H|^2 (half the bytes of mode 2)
    auto load_otf = [&]() {  // REALH: its own instantiation, the other variants keep their code
        if constexpr (REALH) {
            const float *of = reinterpret_cast<const float *>(otf);
#pragma unroll
            for (int b = 0; b < NBF; ++b)
#pragma unroll
                for (int k2 = 0; k2 < L; ++k2)
                    h[b * L + k2] = make_float2(of[pc + (int64_t)((l + b * L) + 16 * k2) * g.plane], 0.0f);
        } else {
#pragma unroll
            for (int b = 0; b < NBF; ++b)
#pragma unroll
                for (int k2 = 0; k2 < L; ++k2) {
                    const float2 *hp = otf + pc + (int64_t)((l + b * L) + 16 * k2) * g.plane;
                    // streamed, except in the all-planes form, where a second workgroup reads the line from L2 (the 4-byte
                    // loads of the real |OTF|^2 volume above got 10 % slower when marked)
                    h[b * L + k2] = OUTF ? *hp : ld_stream(hp);
                }
        }
    };
    if (M3D_ZREG_OTF_EARLY && mode >= 0) load_otf();  // requested before anything else
    float2 v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int z = l + L * m;
        const bool inside = z >= g.in.z0 && z < zin_hi;
        // outside the input's planes: zero, or (zmirror) the plane it mirrors - the same data again, never zero
        // zmirror == 2: the pad on either side is no deeper than the core, one reflection and no modulo
        int zs;
        if constexpr (ZM == 2) zs = z < g.in.z0 ? 2 * g.in.z0 - 1 - z : (z >= zin_hi ? 2 * zin_hi - 1 - z : z);
        else if constexpr (ZM == 1) zs = inside ? z : sym_src(z, g.in.z0, g.in.nz);
        else zs = inside ? z : g.in.z0;
        // streaming loads where every plane is read once (mirrored planes are read twice by the same thread) and the
        // output is cropped; the all-planes form (two ratio spectra per launch) measured 2.6 % slower with them
        v[m] = (ZM == 0 && !OUTF) ? ld_stream(col + (int64_t)zs * g.plane) : col[(int64_t)zs * g.plane];
        if (!live || !(inside || ZM != 0)) v[m] = make_float2(0.f, 0.f);
    }
    float2 w[16];  // W_N^(l k1), l k1 < N: powers 1, 2, 4, 8 from the table, the rest by products
    w[1] = tw[l];
    w[2] = tw[2 * l];
    w[4] = tw[4 * l];
    w[8] = tw[8 * l];
    w[3] = cmul(w[1], w[2]);
    w[5] = cmul(w[1], w[4]);
    w[6] = cmul(w[2], w[4]);
    w[7] = cmul(w[3], w[4]);
    w[9] = cmul(w[1], w[8]);
    w[10] = cmul(w[2], w[8]);
    w[11] = cmul(w[3], w[8]);
    w[12] = cmul(w[4], w[8]);
    w[13] = cmul(w[5], w[8]);
    w[14] = cmul(w[6], w[8]);
    w[15] = cmul(w[7], w[8]);
    dft16<false>(v);
    ex[l * P + p] = v[0];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) ex[k1 * KS + l * P + p] = cmul(v[k1], w[k1]);
    if (!M3D_ZREG_OTF_EARLY && mode >= 0) load_otf();  // requested once the 16 samples are out of the registers
    __syncthreads();
    float2 y[16];
#pragma unroll
    for (int b = 0; b < NBF; ++b) {
        const int k1 = l + b * L;
#pragma unroll
        for (int ll = 0; ll < L; ++ll) y[b * L + ll] = ex[k1 * KS + ll * P + p];
        dft_r<L, false>(&y[b * L]);  // X[k1 + 16 k2] at y[b L + k2]
    }
    if (mode < 0) {
#pragma unroll
        for (int b = 0; b < NBF; ++b)
#pragma unroll
            for (int k2 = 0; k2 < L; ++k2)
                if (live) col[(int64_t)((l + b * L) + 16 * k2) * g.plane] = y[b * L + k2];
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float2 hh = h[i];
        if constexpr (REALH) {  // same two roundings as cmul with a zero imaginary part, then the scale
            y[i] = make_float2(y[i].x * hh.x * scale, y[i].y * hh.x * scale);
            continue;
        }
        if (mode == 1) hh.y = -hh.y;
        if (mode == 2) hh = make_float2(hh.x * hh.x + hh.y * hh.y, 0.0f);
        const float2 t = cmul(y[i], hh);
        y[i] = make_float2(t.x * scale, t.y * scale);
    }
#pragma unroll
    for (int b = 0; b < NBF; ++b) {
        const int k1 = l + b * L;
        dft_r<L, true>(&y[b * L]);  // sum over k2: partial result for lane l' at y[b L + l']
        // each thread rewrites exactly the slots it read: no barrier needed in between
        float2 u[L];  // W_N^(ll k1): powers 1, 2, 4, 8 of W^k1 from the table (ll k1 < N), the rest by products
        u[0] = make_float2(1.0f, 0.0f);
#pragma unroll
        for (int ll = 1; ll < L; ++ll) {
            if ((ll & (ll - 1)) == 0)
                u[ll] = tw[ll * k1];
            else
                u[ll] = cmul(u[ll & (ll - 1)], u[ll & -ll]);  // highest bits x lowest set bit
        }
#pragma unroll
        for (int ll = 0; ll < L; ++ll) ex[k1 * KS + ll * P + p] = cmul_conj(y[b * L + ll], u[ll]);
    }
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = ex[k1 * KS + l * P + p];
    dft16<true>(v);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int z = l + L * m;
        if (live && (OUTF || (z >= g.out.z0 && z < g.out.z0 + g.out.nz))) {
            if constexpr (OUTF) col[(int64_t)z * g.plane] = v[m];
            else st_stream(col + (int64_t)z * g.plane, v[m]);
        }
    }
}


bool factor_23(int n, int *radix, int *nstages, bool z_axis = false, bool x_axis = false)
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
#ifdef M3D_FFT_RADIX27
    if (c3 % 2 == 1 && c3 >= 3) {  // odd power of three: one radix-27 stage, the rest radix 9
        radix[s++] = 27;
        c3 -= 3;
    }
#endif
    // X axis with two or more radix-9 stages: one of them goes last, so that both outer stages are radix 9 and can
    // run in registers (k_x_forward_reg / k_x_inverse_reg)
    const bool nine_last = x_axis && c3 >= 4;
    constexpr bool small_pow2 = true;  // 4,4,4,2 beat 16,8 (VGPRs; measured in round 1)
    // X axis of a length 64 m <= 2048 that has no two radix-9 stages (2048, 1728, 1536, 1152, 1024, 576, ...): a radix-8
    // stage first and last, which the same register-edge kernels run (edge radix 8; M3D_FFT_XREG8=0 disables)
    const bool eight_edges = x_axis && !nine_last && c2 >= 6 && n / 8 <= 256 && small_pow2 &&
                             !(getenv("M3D_FFT_XREG8") && atoi(getenv("M3D_FFT_XREG8")) == 0);
    if (eight_edges) {
        radix[s++] = 8;
        c2 -= 6;
    }
    for (int i = 0; i < c3 / 2 - (nine_last ? 1 : 0); ++i) radix[s++] = 9;
    if (c3 % 2) radix[s++] = 3;
    // Z axis (short power-of-two transforms, k_z_conv<2>): as few radix-8 stages as bring the stage count
    // down to ceil(log2(n) / 3) - 128 = 8 * 4 * 4 has three barriers per transform instead of four
    // (measured 1.93 -> 1.71 ms); M3D_FFT_Z_RADIX4=1 keeps 4, 4, 4, 2.
    // X axis without register edge stages (in-LDS kernels): the same rule - 2048 = 8 * 8 * 8 * 4 is four LDS stages
    // per transform instead of the six of 4 * 4 * 4 * 4 * 4 * 2 (M3D_FFT_X_RADIX4=1 keeps the radix-4 lists)
    const bool x_eights = x_axis;  // (also between two radix-9 edge stages:
                                                                             //  2592 = 9 8 4 9 instead of 9 4 4 2 9)
    const bool y_eights = !x_axis && !z_axis;  // run-time Y kernels alike
    if ((x_eights || y_eights || z_axis) && small_pow2) {
        const int best = (c2 + 2) / 3;
        int n8 = 0;
        while ((c2 - 3 * n8 + 1) / 2 + n8 > best) ++n8;
        for (int i = 0; i < n8; ++i) radix[s++] = 8;
        c2 -= 3 * n8;
    }
    while (c2 >= 4 && !small_pow2) {
        radix[s++] = 16;
        c2 -= 4;
    }
    if (c2 == 3 && !small_pow2) {
        radix[s++] = 8;
        c2 = 0;
    }
    while (c2 >= 2) {
        radix[s++] = 4;
        c2 -= 2;
    }
    if (c2 == 2) radix[s++] = 4;
    if (c2 == 1) radix[s++] = 2;
    if (nine_last) radix[s++] = 9;
    if (eight_edges) radix[s++] = 8;
    *nstages = s;
    return s <= kMaxStages;
}

ConvGeo geo_of(const FftConvPlan *p, const m3d::ConvBox *in = nullptr, const m3d::ConvBox *out = nullptr,
               const m3d::ConvBox *mirror = nullptr)
{
    ConvGeo g;
    const m3d::ConvBox full = {0, p->Zp, 0, p->Yp, 0, p->Xp};
    g.in = in ? *in : full;
    g.out = out ? *out : full;
    g.mir = mirror ? *mirror : full;
    g.mirror = mirror ? 1 : 0;  // callers set 2 for the rows-of-the-box-only variant
    g.zmirror = 0;
    g.Zp = p->Zp;
    g.Yp = p->Yp;
    g.Xp = p->Xp;
    g.Xc = p->Xc;
    g.TW = p->TW;
    g.nT = p->nT;
    g.RB = p->RB;
    g.P = p->P;
    g.lTW = __builtin_ctz((unsigned)p->TW);
    g.lNB = __builtin_ctz((unsigned)(p->RB / 2));
    g.lP = __builtin_ctz((unsigned)p->P);
    g.plane = p->plane;
    g.GX = p->gx;
    g.GY = p->gy;
    g.GZ = p->gz;
    g.lDR = __builtin_ctz((unsigned)p->dual_rows);
    g.lCH = p->lch;
    g.xcdpair = p->xcdpair;
    return g;
}

}  // namespace

namespace m3d {

constexpr size_t kDefaultLdsLimit = 64 * 1024;

int fftconv_create(int Zp, int Yp, int Xp, FftConvPlan **out)
{
    *out = nullptr;
    FftConvPlan *p = new FftConvPlan();
    p->Zp = Zp;
    p->Yp = Yp;
    p->Xp = Xp;
    p->Xc = Xp / 2 + 1;
    if (!factor_23(Xp, p->ax.radix, &p->ax.nstages, false, true) || !factor_23(Yp, p->ay.radix, &p->ay.nstages) ||
        !factor_23(Zp, p->az.radix, &p->az.nstages, true)) {
        delete p;
        return M3D_ERR_UNSUPPORTED;
    }
    p->ax.n = Xp;
    p->ay.n = Yp;
    p->az.n = Zp;
    auto lean = [](const AxisPlan &a) {
        int r = 1;
        for (int s = 0; s < a.nstages; ++s) {
            if (a.radix[s] == 16) return 0;
            if (a.radix[s] == 8) r = 2;
        }
        return r;
    };
    p->rset_x = lean(p->ax);
    p->rset_y = lean(p->ay);
    p->rset_z = lean(p->az);
    // tile width: widest of 4/2/1 whose column tile fits a 72-KB LDS buffer (2 blocks / CU)
    // Tile shapes as measured on MI355X (profiles/).  Rounds 1-3 read them from the environment for tuning sweeps; round 4
    // made them constants.  What is still read: the ROUTE switches the parity tests compare against each other
    // (M3D_FFT_XSTATIC / YSTATIC / XREG8 here, M3D_FFT_PRUNE / M3D_RLGC_CHAIN_* in m3d_rlgc.hip).
    auto knob = [](const char *name, int dflt) {
        const char *v = getenv(name);
        return v ? atoi(v) : dflt;
    };
    p->TW = 4;
    while (p->TW > 1 && (size_t)Yp * p->TW * sizeof(float2) > (size_t)72 * 1024) p->TW /= 2;
    p->lds_y = (size_t)Yp * p->TW * sizeof(float2);
    // rows per X block: 4 real rows (2 packed) when they fit 42 KB (2592: 41.5 KB, three workgroups per CU; measured
    // 33.7 -> 31.8 ms per iteration against one packed row per workgroup), else 2
    p->RB = ((size_t)2 * Xp * sizeof(float2) <= (size_t)42 * 1024) ? 4 : 2;
    p->lds_x = (size_t)(p->RB / 2) * Xp * sizeof(float2);
    // the X forward pass re-uses its buffer as a staging area of whole 128-byte spectrum lines (4 rows x TW = 4)
    if (p->TW == 4 && p->RB == 4) p->lds_x = std::max(p->lds_x, (size_t)((Xp / 2 + 1 + 3) / 4) * 128);
    p->P = 16;
    while (p->P > 1 && (size_t)Zp * p->P * sizeof(float2) > 64 * 1024) p->P /= 2;
    p->lds_z = (size_t)Zp * p->P * sizeof(float2);
    p->xthreads = 256;
    p->gx = 4;
    p->gy = 4;
    p->gz = 4;
    p->zmirror = 1;
    p->zreg = (Zp == 32 || Zp == 64 || Zp == 128 || Zp == 256) ? Zp / 16 : 0;
    p->dual_rows = 2;
    p->dual_threads = 256;
    p->ythreads = 768;
    p->xcdpair = 1;
    {   // rows per y chunk of the spectrum layout: a power of two >= 4 (whole X blocks); >= Yp = plain tiled layout
        const int ch = 128;
        p->lch = 31 - __builtin_clz((unsigned)ch);
        if ((1 << p->lch) >= Yp) p->lch = 30;
    }
    p->zthreads = std::max(p->P, 128);
    p->xreg = (p->ax.nstages >= 2 && p->ax.radix[0] == 9 &&
               p->ax.radix[p->ax.nstages - 1] == 9 && p->xthreads == 256 && p->dual_threads == 256 &&
               p->dual_rows == 2 && (p->RB == 2 || p->RB == 4) && (p->RB / 2) * (Xp / 18 + 1) <= 320 && Xp / 9 <= 320)
                  ? 9 : 0;
    // more than 256 edge items or block pairs (2592 = 9 * 288, the padded length of 2304-pixel sensors): a fifth wave
    if (p->xreg == 9 && (Xp / 9 > 256 || 2 * (Xp / 18 + 1) > 256)) p->xthreads = p->dual_threads = 320;
    // edge radix 8 (factor_23 put an 8 first and last): S = Xp / 8 <= 256 edge items, S / 2 + 1 block pairs per packed
    // row - 2 x 129 at Xp = 2048, so those workgroups take a fifth wave
    if (!p->xreg && knob("M3D_FFT_XREG8", 1) && p->ax.nstages >= 2 && p->ax.radix[0] == 8 &&
        p->ax.radix[p->ax.nstages - 1] == 8 && p->rset_x == 2 && p->xthreads == 256 && p->dual_threads == 256 &&
        p->dual_rows == 2 && (p->RB == 2 || p->RB == 4) && Xp % 64 == 0 && Xp / 8 <= 256) {
        p->xreg = 8;
        if (2 * (Xp / 16 + 1) > 256) p->xthreads = p->dual_threads = 320;
    }
    p->ystatic = (knob("M3D_FFT_YSTATIC", 1) && Yp == 2187 && p->TW == 4 && p->lch == 7 && (p->ythreads == 768 || p->ythreads == 1024) &&
                  p->gy == 4 && p->ay.nstages == 4 && p->ay.radix[0] == 9 && p->ay.radix[1] == 9 &&
                  p->ay.radix[2] == 9 && p->ay.radix[3] == 3) ? 2187 : 0;
    // Other common padded lengths (1024- / 1600- / 1200- / 1920-pixel images + PSF halo) get static Y kernels of
    // their own, with radix-8 stages for the powers of two: the generic kernel runs 2048 as 4 * 4 * 4 * 4 * 4 * 2 - six
    // LDS stages with run-time index arithmetic - and costs 1.55 x per voxel (profiles/r03_sizes.txt).  The Y axis's
    // radix list is private to the Y passes (the spectrum only ever meets the OTF, which the same passes produce).
    if (!p->ystatic && knob("M3D_FFT_YSTATIC", 1) && p->TW == 4 && p->lch == 7 && p->gy == 4) {
        static const int lists[][5] = {{2048, 8, 8, 8, 4}, {1728, 9, 8, 8, 3}, {1296, 9, 9, 4, 4}, {1152, 9, 8, 4, 4},
                                       {576, 9, 8, 8, 1}};
        for (const auto &l : lists)
            if (Yp == l[0]) {
                p->ystatic = Yp;
                p->ay.nstages = l[4] > 1 ? 4 : 3;
                for (int i = 0; i < p->ay.nstages; ++i) p->ay.radix[i] = l[1 + i];
                p->rset_y = lean(p->ay);  // (the run-time mirror pass of a deep pad remains reachable: radix-8 build)
            }
    }
    // the static Y kernels spread (butterfly, column group) items over the threads: 972 (forward) / 486 (inverse) items
    // in the radix-9 stages, one round with 1024 threads (two blocks of 16 waves fill a CU); the generic kernels keep 768
    if (p->ystatic) p->ythreads = 1024;
    // The headline length runs as 9 * 9 * 27 (round 4): the radix-3 stage and the radix-9 stage in front of it become
    // one radix-27 butterfly on blocks of 27 - no twiddles, 54 registers of data - which takes one of the pass's four
    // trips through LDS away: inverse 670 -> 640 us, forward 621 -> 593 us per launch on one box
    // (profiles/r04_workspace_placement.txt section 10; the passes hold the card at its power limit, section 8).  70
    // registers: two blocks of 768 threads per CU.  Static kernels only - the run-time kernels have no radix-27 stage.
    if (p->ystatic == 2187) {
        p->ythreads = 768;
        p->ay.nstages = 3;
        p->ay.radix[0] = 9;
        p->ay.radix[1] = 9;
        p->ay.radix[2] = 27;
    }
    // 2592 (2304-pixel sensors): tiles of two columns (41.5 KB, three workgroups per CU), 9 9 8 4, 640 threads - the 576
    // (butterfly, column) items of a radix-9 stage in one round, 3 x 640 threads per CU
    if (!p->ystatic && knob("M3D_FFT_YSTATIC", 1) && Yp == 2592 && p->TW == 2 && p->lch == 7 && p->gy == 4) {
        static const int l2592[4] = {9, 9, 8, 4};
        p->ystatic = 2592;
        p->ay.nstages = 4;
        for (int i = 0; i < 4; ++i) p->ay.radix[i] = l2592[i];
        p->rset_y = lean(p->ay);
        p->ythreads = 640;
    }
    p->xstatic = (knob("M3D_FFT_XSTATIC", 1) && p->xreg && Xp == 2187 && Yp == 2187 && p->TW == 4 && p->RB == 4 &&
                  p->lch == 7 && p->gx == 4 && p->rset_x && p->ax.nstages == 4 && p->ax.radix[1] == 9 &&
                  p->ax.radix[2] == 3) ? 1 : 0;
    if (p->lds_y > 150 * 1024 || p->lds_x > 150 * 1024 || p->lds_z > 150 * 1024) {
        delete p;
        return M3D_ERR_UNSUPPORTED;
    }
    p->nT = (p->Xc + p->TW - 1) / p->TW;
    p->plane = (int64_t)p->nT * Yp * p->TW;
    if (p->plane >= (int64_t)1 << 31) {  // spec_index keeps in-plane offsets in 32 bits
        delete p;
        return M3D_ERR_UNSUPPORTED;
    }

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
    // Dynamic LDS above the 64 KB every kernel may use by default: the limit is raised on the kernels THIS plan launches
    // and nowhere else (a Y column tile of the 2187-point axis is 70 KB; X, dual and Z buffers above 64 KB are refused).
    // Round 3 set it on 27 instantiations with the status discarded; a library whose host stubs named a kernel that
    // its code object lacked then aborted inside the runtime at this point.
    if (p->lds_x > kDefaultLdsLimit || (size_t)p->dual_rows * Xp * sizeof(float2) > kDefaultLdsLimit ||
        p->lds_z > kDefaultLdsLimit) {
        fftconv_destroy(p);
        return M3D_ERR_UNSUPPORTED;
    }
    int rc = M3D_OK;
    auto allow = [&](const void *fn) {
        if (rc != M3D_OK || p->lds_y <= kDefaultLdsLimit) return;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_y);
        if (e != hipSuccess) {
            set_err("hipFuncSetAttribute(MaxDynamicSharedMemorySize) on a Y pass", hipGetErrorString(e));
            rc = M3D_ERR_HIP;
        }
    };
    if (p->ystatic == 2187) {
        allow((const void *)k_y_forward_mirror_static<2187, 9, 9, 27, 1, 7, 768>);
        allow((const void *)k_y_pass_static<false, 2187, 9, 9, 27, 1, 7, 768>);
        allow((const void *)k_y_pass_static<true, 2187, 9, 9, 27, 1, 7, 768>);
    } else if (p->ystatic && p->lds_y > kDefaultLdsLimit) {  // the other static tiles are 64 KB at most by construction
        fftconv_destroy(p);
        return M3D_ERR_UNSUPPORTED;
    }
    if (!p->ystatic) {
        if (p->rset_y == 2) {
            allow((const void *)k_y_pass<false, 2>);
            allow((const void *)k_y_pass<true, 2>);
        } else if (p->rset_y) {
            allow((const void *)k_y_pass<false, 1>);
            allow((const void *)k_y_pass<true, 1>);
        } else {
            allow((const void *)k_y_pass<false, 0>);
            allow((const void *)k_y_pass<true, 0>);
        }
    }
    // the run-time mirror pass also serves static plans whose pad is deeper than one reflection
    if (p->rset_y == 2) allow((const void *)k_y_forward_mirror<2>);
    else if (p->rset_y) allow((const void *)k_y_forward_mirror<1>);
    else allow((const void *)k_y_forward_mirror<0>);
    if (rc != M3D_OK) {
        fftconv_destroy(p);
        return rc;
    }
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

static int row_groups(const FftConvPlan *p, const m3d::ConvBox &b)
{
    const int yg0 = (b.y0 / p->RB) * p->RB;
    return (b.y0 + b.ny - yg0 + p->RB - 1) / p->RB;
}

// One Z pass (mode -1: forward only, OTF construction) with the kernel family the plan uses throughout.
static int launch_z(const FftConvPlan *p, float2 *spec, const float2 *otf, const ConvGeo &g, int mode, float scale,
                    hipStream_t st, float2 *spec_b = nullptr)
{
    const unsigned nblk = (unsigned)((p->plane + p->P - 1) / p->P);
    // a real |OTF|^2 volume (mode 3) exists for the register Z pass with zero-extended input and cropped output (the
    // consensus convolution); callers check fftconv_real_otf_supported and pass it for that geometry only
    if (mode == 3 && (!p->zreg || g.zmirror || (g.out.z0 == 0 && g.out.nz == p->Zp))) {
        set_err("fftconv", "mode 3 outside the geometry it is built for");
        return M3D_ERR_INVALID_ARG;
    }
    if (!p->zreg && spec_b) {  // the LDS variant has no paired form
        if (int rc = launch_z(p, spec, otf, g, mode, scale, st)) return rc;
        return launch_z(p, spec_b, otf, g, mode, scale, st);
    }
    if (p->zreg) {
        constexpr int zp_env = 0;  // (16 / 32 columns per block were tuning switches of rounds 1-2)
        // columns per block: 512-byte segments (64 columns, 512 threads) are 2-7 % faster than 256-byte ones where the
        // exchange buffer still lets two blocks share a CU (Zp = 128); 16 columns only on request
        const int zp = zp_env == 16 ? 16 : (zp_env == 32 ? 32 : (p->zreg == 8 ? 64 : 32));
        const unsigned ntile = (unsigned)((p->plane + zp - 1) / zp);
        const unsigned nb = spec_b ? ((ntile + 7) / 8) * 16 : ntile;  // pairs of eight tiles (see k_z_conv_reg)
        const bool outf = g.out.z0 == 0 && g.out.nz == p->Zp;
#define M3D_LAUNCH_ZREG(LL, PP)                                                                                       \
    do {                                                                                                              \
        if (g.zmirror == 2 && !outf)                                                                                  \
            hipLaunchKernelGGL((k_z_conv_reg<LL, PP, 2, false>), dim3(nb), dim3(PP * LL), 0, st, spec, otf, g, p->az.tw, mode, scale, spec_b); \
        else if (g.zmirror)                                                                                           \
            hipLaunchKernelGGL((k_z_conv_reg<LL, PP, 1, false>), dim3(nb), dim3(PP * LL), 0, st, spec, otf, g, p->az.tw, mode, scale, spec_b); \
        else if (outf)                                                                                                \
            hipLaunchKernelGGL((k_z_conv_reg<LL, PP, 0, true>), dim3(nb), dim3(PP * LL), 0, st, spec, otf, g, p->az.tw, mode, scale, spec_b);  \
        else if (mode == 3)                                                                                           \
            hipLaunchKernelGGL((k_z_conv_reg<LL, PP, 0, false, true>), dim3(nb), dim3(PP * LL), 0, st, spec, otf, g, p->az.tw, mode, scale, spec_b); \
        else                                                                                                          \
            hipLaunchKernelGGL((k_z_conv_reg<LL, PP, 0, false>), dim3(nb), dim3(PP * LL), 0, st, spec, otf, g, p->az.tw, mode, scale, spec_b); \
    } while (0)
        if (zp == 64 && p->zreg == 8) {
            M3D_LAUNCH_ZREG(8, 64);
        } else if (zp >= 32) {
            switch (p->zreg) {
            case 2: M3D_LAUNCH_ZREG(2, 32); break;
            case 4: M3D_LAUNCH_ZREG(4, 32); break;
            case 8: M3D_LAUNCH_ZREG(8, 32); break;
            default: M3D_LAUNCH_ZREG(16, 32); break;
            }
        } else {
            switch (p->zreg) {
            case 2: M3D_LAUNCH_ZREG(2, 16); break;
            case 4: M3D_LAUNCH_ZREG(4, 16); break;
            case 8: M3D_LAUNCH_ZREG(8, 16); break;
            default: M3D_LAUNCH_ZREG(16, 16); break;
            }
        }
#undef M3D_LAUNCH_ZREG
    } else if (p->rset_z == 2)
        hipLaunchKernelGGL(k_z_conv<2>, dim3(nblk), dim3(p->zthreads), p->lds_z, st, spec, otf, g, p->az, mode, scale);
    else if (p->rset_z)
        hipLaunchKernelGGL(k_z_conv<1>, dim3(nblk), dim3(p->zthreads), p->lds_z, st, spec, otf, g, p->az, mode, scale);
    else
        hipLaunchKernelGGL(k_z_conv<0>, dim3(nblk), dim3(p->zthreads), p->lds_z, st, spec, otf, g, p->az, mode, scale);
    return M3D_OK;
}

// Y pass: the static-geometry build when the plan is the headline one (2187 = 9 * 9 * 27, TW 4, 128-row chunks,
// 768 threads), or one of the other common lengths, else the generic kernel.
template <bool INV>
static void launch_y_pass(const FftConvPlan *p, float2 *spec, const ConvGeo &g, int nz, hipStream_t st)
{
    const dim3 grid((unsigned)(nz * p->nT)), block(p->ythreads);
    if (p->ystatic == 2187) {
        hipLaunchKernelGGL((k_y_pass_static<INV, 2187, 9, 9, 27, 1, 7, 768>), grid, block, p->lds_y, st, spec, g, p->ay.tw);
    } else if (p->ystatic == 2048)
        hipLaunchKernelGGL((k_y_pass_static<INV, 2048, 8, 8, 8, 4, 7, 1024>), grid, block, p->lds_y, st, spec, g, p->ay.tw);
    else if (p->ystatic == 1728)
        hipLaunchKernelGGL((k_y_pass_static<INV, 1728, 9, 8, 8, 3, 7, 1024>), grid, block, p->lds_y, st, spec, g, p->ay.tw);
    else if (p->ystatic == 1296)
        hipLaunchKernelGGL((k_y_pass_static<INV, 1296, 9, 9, 4, 4, 7, 1024>), grid, block, p->lds_y, st, spec, g, p->ay.tw);
    else if (p->ystatic == 1152)
        hipLaunchKernelGGL((k_y_pass_static<INV, 1152, 9, 8, 4, 4, 7, 1024>), grid, block, p->lds_y, st, spec, g, p->ay.tw);
    else if (p->ystatic == 2592)
        hipLaunchKernelGGL((k_y_pass_static<INV, 2592, 9, 9, 8, 4, 7, 640, 2>), grid, block, p->lds_y, st, spec, g, p->ay.tw);
    else if (p->ystatic == 576)
        hipLaunchKernelGGL((k_y_pass_static<INV, 576, 9, 8, 8, 1, 7, 1024>), grid, block, p->lds_y, st, spec, g, p->ay.tw);
    else if (p->rset_y == 2)
        hipLaunchKernelGGL((k_y_pass<INV, 2>), grid, block, p->lds_y, st, spec, g, p->ay);
    else if (p->rset_y)
        hipLaunchKernelGGL((k_y_pass<INV, 1>), grid, block, p->lds_y, st, spec, g, p->ay);
    else
        hipLaunchKernelGGL((k_y_pass<INV, 0>), grid, block, p->lds_y, st, spec, g, p->ay);
}

static void launch_x_forward(const FftConvPlan *p, const float *src, float2 *spec, const ConvGeo &g, int groups,
                             hipStream_t st)
{
    const dim3 grid((unsigned)(g.in.nz * groups)), block(p->xthreads);
    if (p->xstatic) {
        hipLaunchKernelGGL((k_x_forward_reg<1, 2187>), grid, block, p->lds_x, st, src, spec, g, p->ax);
    } else if (p->xreg == 8) {
        hipLaunchKernelGGL((k_x_forward_reg<2, 0, 8>), grid, block, p->lds_x, st, src, spec, g, p->ax);
    } else if (p->xreg && p->rset_x == 2) {
        hipLaunchKernelGGL((k_x_forward_reg<2, 0, 9>), grid, block, p->lds_x, st, src, spec, g, p->ax);
    } else if (p->xreg) {
        if (p->rset_x)
            hipLaunchKernelGGL(k_x_forward_reg<1>, grid, block, p->lds_x, st, src, spec, g, p->ax);
        else
            hipLaunchKernelGGL(k_x_forward_reg<0>, grid, block, p->lds_x, st, src, spec, g, p->ax);
    } else if (p->rset_x == 2)
        hipLaunchKernelGGL(k_x_forward<2>, grid, block, p->lds_x, st, src, spec, g, p->ax);
    else if (p->rset_x)
        hipLaunchKernelGGL(k_x_forward<1>, grid, block, p->lds_x, st, src, spec, g, p->ax);
    else
        hipLaunchKernelGGL(k_x_forward<0>, grid, block, p->lds_x, st, src, spec, g, p->ax);
}

static int launch_forward_xy(const FftConvPlan *p, const float *src, float2 *spec, hipStream_t st,
                             EventLog *log = nullptr, const ConvBox *in_box = nullptr,
                             const ConvBox *mirror_box = nullptr, float2 *spec_alt = nullptr, bool x_done = false)
{
    if (src && mirror_box && spec_alt && p->TW % 2 == 0) {
        // symmetric extension given by its core: X transforms of the core rows only, the Y pass builds
        // every plane from the core plane it mirrors (out of place: spec -> spec_alt)
        ConvGeo g = geo_of(p, mirror_box, nullptr, mirror_box);
        g.mirror = 2;
        g.zmirror = p->zmirror;
        const unsigned yplanes = (unsigned)(p->zmirror ? g.mir.nz : p->Zp);
        const int groups = row_groups(p, g.in);
        if (log && log->mark(EventLog::kStart)) return M3D_ERR_HIP;
        if (!x_done) launch_x_forward(p, src, spec, g, groups, st);  // x_done: chained from the update pass
        if (log && log->mark(EventLog::kXF)) return M3D_ERR_HIP;
        const bool one_reflection = g.mir.y0 <= g.mir.ny && p->Yp - (g.mir.y0 + g.mir.ny) <= g.mir.ny;
#define M3D_LAUNCH_YMS(YP, R0, R1, R2, R3, NT)                                                                          \
    hipLaunchKernelGGL((k_y_forward_mirror_static<YP, R0, R1, R2, R3, 7, NT>), dim3(yplanes * (unsigned)p->nT),           \
                       dim3(p->ythreads), p->lds_y, st, spec, spec_alt, g, p->ay.tw)
        if (p->ystatic == 2187 && !one_reflection) {
            // 9 * 9 * 27 exists as static kernels only (the run-time kernels have no radix-27 stage), and a pad deeper
            // than the core cannot occur at this length: 2187 is chosen for images of 1945 rows and more
            set_err("fftconv", "a symmetric pad deeper than the image at padded length 2187");
            return M3D_ERR_UNSUPPORTED;
        }
        if (p->ystatic == 2187) M3D_LAUNCH_YMS(2187, 9, 9, 27, 1, 768);
        else if (p->ystatic == 2048 && one_reflection) M3D_LAUNCH_YMS(2048, 8, 8, 8, 4, 1024);
        else if (p->ystatic == 1728 && one_reflection) M3D_LAUNCH_YMS(1728, 9, 8, 8, 3, 1024);
        else if (p->ystatic == 1296 && one_reflection) M3D_LAUNCH_YMS(1296, 9, 9, 4, 4, 1024);
        else if (p->ystatic == 1152 && one_reflection) M3D_LAUNCH_YMS(1152, 9, 8, 4, 4, 1024);
        else if (p->ystatic == 576 && one_reflection) M3D_LAUNCH_YMS(576, 9, 8, 8, 1, 1024);
        else if (p->ystatic == 2592 && one_reflection)
            hipLaunchKernelGGL((k_y_forward_mirror_static<2592, 9, 9, 8, 4, 7, 640, 2>), dim3(yplanes * (unsigned)p->nT),
                               dim3(p->ythreads), p->lds_y, st, spec, spec_alt, g, p->ay.tw);
#undef M3D_LAUNCH_YMS
        else if (p->rset_y == 2)
            hipLaunchKernelGGL(k_y_forward_mirror<2>, dim3(yplanes * (unsigned)p->nT), dim3(p->ythreads), p->lds_y, st, spec, spec_alt, g, p->ay);
        else if (p->rset_y)
            hipLaunchKernelGGL(k_y_forward_mirror<1>, dim3(yplanes * (unsigned)p->nT), dim3(p->ythreads), p->lds_y, st, spec, spec_alt, g, p->ay);
        else
            hipLaunchKernelGGL(k_y_forward_mirror<0>, dim3(yplanes * (unsigned)p->nT), dim3(p->ythreads), p->lds_y, st, spec, spec_alt, g, p->ay);
        M3D_KERNEL_CHECK();
        return M3D_OK;
    }
    const ConvGeo g = geo_of(p, in_box, nullptr, mirror_box);
    const int groups = row_groups(p, g.in);
    if (log && log->mark(EventLog::kStart)) return M3D_ERR_HIP;
    if (src) {  // src == nullptr: the half-spectrum rows are already in `spec` (chained from another pass)
        launch_x_forward(p, src, spec, g, groups, st);
        if (log && log->mark(EventLog::kXF)) return M3D_ERR_HIP;
    }
    launch_y_pass<false>(p, spec, g, g.in.nz, st);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int fftconv_make_otf(const FftConvPlan *p, const float *psf_padded, float2 *otf, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_forward_xy(p, psf_padded, otf, st);
    if (rc) return rc;
    const ConvGeo g = geo_of(p);
    if ((rc = launch_z(p, otf, (const float2 *)nullptr, g, -1, 1.0f, st))) return rc;
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

static bool box_ok(const FftConvPlan *p, const ConvBox *b)
{
    return !b || (b->z0 >= 0 && b->nz > 0 && b->z0 + b->nz <= p->Zp && b->y0 >= 0 && b->ny > 0 &&
                  b->y0 + b->ny <= p->Yp && b->x0 >= 0 && b->nx > 0 && b->x0 + b->nx <= p->Xp);
}

int fftconv_to_rows(const FftConvPlan *p, const float *src, const float2 *otf, int mode, float2 *spec,
                    void *stream, EventLog *log, const ConvBox *in_box, const ConvBox *out_box,
                    const ConvBox *mirror_box, float2 *spec_alt, float2 **rows, bool x_forward_done)
{
    hipStream_t st = (hipStream_t)stream;
    if (!box_ok(p, in_box) || !box_ok(p, out_box) || !box_ok(p, mirror_box) || (in_box && mirror_box))
        return M3D_ERR_INVALID_ARG;
    if (x_forward_done && !(src && mirror_box && spec_alt && p->TW % 2 == 0)) return M3D_ERR_INVALID_ARG;
    int rc = launch_forward_xy(p, src, spec, st, log, in_box, mirror_box, spec_alt, x_forward_done);
    if (rc) return rc;
    const bool mirrored = src && mirror_box && spec_alt && p->TW % 2 == 0;
    if (mirrored) spec = spec_alt;  // the mirrored Y pass wrote there
    if (rows) *rows = spec;
    if (log && log->mark(EventLog::kYF)) return M3D_ERR_HIP;
    ConvGeo g = geo_of(p, (mirrored && p->zmirror) ? mirror_box : in_box, out_box);
    g.zmirror = (mirrored && p->zmirror) ? 1 : 0;
    if (g.zmirror && g.in.z0 <= g.in.nz && p->Zp - (g.in.z0 + g.in.nz) <= g.in.nz) g.zmirror = 2;  // single reflection
    const float scale = 1.0f / ((float)p->Zp * (float)p->Yp * (float)p->Xp);
    if ((rc = launch_z(p, spec, otf, g, mode, scale, st))) return rc;
    if (log && log->mark(EventLog::kZ)) return M3D_ERR_HIP;
    launch_y_pass<true>(p, spec, g, g.out.nz, st);
    if (log && log->mark(EventLog::kYI)) return M3D_ERR_HIP;
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

bool fftconv_real_otf_supported(const FftConvPlan *p) { return p && p->zreg != 0; }
bool fftconv_update_chain_supported(const FftConvPlan *p) { return p && p->xstatic && p->TW == 4 && p->RB == 4; }

int fftconv_engine_flags(const FftConvPlan *p)
{
    if (!p) return 0;
    return (p->xstatic ? M3D_ENGINE_X_STATIC : 0) | (p->ystatic ? M3D_ENGINE_Y_STATIC : 0) |
           (p->zreg ? M3D_ENGINE_Z_REGISTER : 0);
}

__global__ void k_otf_abs2(const float2 *__restrict__ otf, float *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float2 h = otf[i];
        out[i] = h.x * h.x + h.y * h.y;  // the arithmetic of mode 2
    }
}
int fftconv_otf_abs2(const FftConvPlan *p, const float2 *otf, float *out, void *stream)
{
    if (!p || !otf || !out) return M3D_ERR_INVALID_ARG;
    const int64_t n = (int64_t)p->Zp * p->plane;
    hipLaunchKernelGGL(k_otf_abs2, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, otf, out, n);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int fftconv_to_rows_pair(const FftConvPlan *p, const float *src1, const float *src2, const float2 *otf, int mode,
                         float2 *spec1, float2 *spec2, void *stream, EventLog *log, const ConvBox *in_box,
                         bool x_forward_done)
{
    hipStream_t st = (hipStream_t)stream;
    if ((!x_forward_done && (!src1 || !src2)) || !spec1 || !spec2 || spec1 == spec2 || !box_ok(p, in_box))
        return M3D_ERR_INVALID_ARG;
    // one span of the event log: both X passes, then both Y passes (two launch_forward_xy calls would open a second
    // span in the middle and drop the first spectrum's Y pass from the attribution)
    ConvGeo g = geo_of(p, in_box, nullptr);
    g.zmirror = 0;
    const int groups = row_groups(p, g.in);
    if (log && log->mark(EventLog::kStart)) return M3D_ERR_HIP;
    if (!x_forward_done) {
        launch_x_forward(p, src1, spec1, g, groups, st);
        launch_x_forward(p, src2, spec2, g, groups, st);
        if (log && log->mark(EventLog::kXF)) return M3D_ERR_HIP;
    }
    launch_y_pass<false>(p, spec1, g, g.in.nz, st);
    launch_y_pass<false>(p, spec2, g, g.in.nz, st);
    M3D_KERNEL_CHECK();
    if (log && log->mark(EventLog::kYF)) return M3D_ERR_HIP;
    const float scale = 1.0f / ((float)p->Zp * (float)p->Yp * (float)p->Xp);
    if (int rc = launch_z(p, spec1, otf, g, mode, scale, st, spec2)) return rc;
    if (log && log->mark(EventLog::kZ)) return M3D_ERR_HIP;
    launch_y_pass<true>(p, spec1, g, g.out.nz, st);
    launch_y_pass<true>(p, spec2, g, g.out.nz, st);
    if (log && log->mark(EventLog::kYI)) return M3D_ERR_HIP;
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

// Static geometry (2187) only.  Round 3 also built the kernel with run-time geometry for every length the register-edge
// kernels take: parity green and no faster than the separate kernels (2048: 18.2 against 17.6 ms per iteration - a fifth
// wave at 121 registers leaves three workgroups per CU; 1728 / 1296 / 1152: -1 to +1 %; profiles/r03_sizes.txt).  Those
// instantiations (and their switch) were taken out in round 4; the kernel keeps its geometry template parameters.
bool fftconv_split_chain_supported(const FftConvPlan *p) { return p && p->xstatic && p->TW == 4 && p->RB == 4; }

int64_t fftconv_split_chain_blocks(const FftConvPlan *p, const ConvBox *core)
{
    if (!p || !core) return 0;
    const int yg0 = (core->y0 / 4) * 4;
    return (int64_t)core->nz * ((core->y0 + core->ny - yg0 + 1) / 2);
}

int fftconv_split_chain(const FftConvPlan *p, float2 *spec_hu, float2 *spec_r1, const ConvBox *core,
                        const SplitArgs *args, void *stream, EventLog *log)
{
    hipStream_t st = (hipStream_t)stream;
    if (!fftconv_split_chain_supported(p)) return M3D_ERR_UNSUPPORTED;
    if (!spec_hu || !spec_r1 || spec_hu == spec_r1 || !core || !box_ok(p, core) || !args || !args->image ||
        !args->partial || !args->tables.thr)
        return M3D_ERR_INVALID_ARG;
    const ConvGeo g = geo_of(p, nullptr, core);
    const dim3 grid((unsigned)fftconv_split_chain_blocks(p, core)), block(256);
    const size_t lds = (size_t)2 * p->Xp * sizeof(float2);
#define M3D_LAUNCH_SC(T, NS, R, RS) \
    hipLaunchKernelGGL((k_x_inverse_split_chain<T, NS, R, RS>), grid, block, lds, st, spec_hu, spec_r1, g, p->ax, *args)
    if (args->image_is_u16) M3D_LAUNCH_SC(uint16_t, 2187, 9, 1);
    else M3D_LAUNCH_SC(float, 2187, 9, 1);
#undef M3D_LAUNCH_SC
    if (log && log->mark(EventLog::kXI)) return M3D_ERR_HIP;
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int fftconv_dual_consensus(const FftConvPlan *p, float2 *spec1, const float2 *spec2, void *stream,
                           EventLog *log, const ConvEpilogue *epilogue, bool chain_forward)
{
    hipStream_t st = (hipStream_t)stream;
    if (!spec1 || !spec2 || !epilogue || epilogue->kind != ConvEpilogue::kConsensus || !epilogue->b ||
        (!chain_forward && !epilogue->o1) || !epilogue->o2 || !epilogue->red ||
        (chain_forward && p->dual_rows != 2))
        return M3D_ERR_INVALID_ARG;
    const ConvGeo g = geo_of(p);
    const int nb = p->dual_rows;
    const size_t lds = (size_t)nb * p->Xp * sizeof(float2);
    const dim3 grid((unsigned)(p->Zp * ((p->Yp + nb - 1) / nb))), block(p->dual_threads);
    if (p->xstatic) {
        if (chain_forward)
            hipLaunchKernelGGL((k_x_inverse_dual_reg<1, true, 2187>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
        else
            hipLaunchKernelGGL((k_x_inverse_dual_reg<1, false, 2187>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
    } else if (p->xreg == 8) {
        if (chain_forward)
            hipLaunchKernelGGL((k_x_inverse_dual_reg<2, true, 0, 8>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
        else
            hipLaunchKernelGGL((k_x_inverse_dual_reg<2, false, 0, 8>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
    } else if (p->xreg && p->rset_x == 2) {
        if (chain_forward)
            hipLaunchKernelGGL((k_x_inverse_dual_reg<2, true, 0, 9>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
        else
            hipLaunchKernelGGL((k_x_inverse_dual_reg<2, false, 0, 9>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
    } else if (p->xreg) {
#define M3D_LAUNCH_XD(RS, CH) \
    hipLaunchKernelGGL((k_x_inverse_dual_reg<RS, CH>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue)
        if (chain_forward) {
            if (p->rset_x) M3D_LAUNCH_XD(1, true);
            else M3D_LAUNCH_XD(0, true);
        } else {
            if (p->rset_x) M3D_LAUNCH_XD(1, false);
            else M3D_LAUNCH_XD(0, false);
        }
#undef M3D_LAUNCH_XD
    } else if (chain_forward) {
        if (p->rset_x == 2)
            hipLaunchKernelGGL((k_x_inverse_dual<2, true>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
        else if (p->rset_x)
            hipLaunchKernelGGL((k_x_inverse_dual<1, true>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
        else
            hipLaunchKernelGGL((k_x_inverse_dual<0, true>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
    } else if (p->rset_x == 2)
        hipLaunchKernelGGL((k_x_inverse_dual<2, false>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
    else if (p->rset_x)
        hipLaunchKernelGGL((k_x_inverse_dual<1, false>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
    else
        hipLaunchKernelGGL((k_x_inverse_dual<0, false>), grid, block, lds, st, spec1, spec2, g, p->ax, *epilogue);
    if (log && log->mark(EventLog::kXI)) return M3D_ERR_HIP;
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

int fftconv_conv(const FftConvPlan *p, const float *src, const float2 *otf, int mode, float2 *spec,
                 float *dst, void *stream, EventLog *log, const ConvBox *in_box, const ConvBox *out_box,
                 const ConvBox *mirror_box, const ConvEpilogue *epilogue, float2 *spec_alt, bool x_forward_done)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = fftconv_to_rows(p, src, otf, mode, spec, stream, log, in_box, out_box, mirror_box, spec_alt, &spec,
                             x_forward_done);
    if (rc) return rc;
    const ConvGeo g = geo_of(p, in_box, out_box);
    const int groups = row_groups(p, g.out);
    const dim3 xg((unsigned)(g.out.nz * groups)), xb(p->xthreads);
    ConvEpilogue ep = {};
    if (epilogue) ep = *epilogue;
    if (ep.kind != ConvEpilogue::kNone && (!ep.a || !ep.b || !ep.o1 || !ep.red ||
                                           (ep.kind == ConvEpilogue::kConsensus && !ep.o2)))
        return M3D_ERR_INVALID_ARG;
#define M3D_LAUNCH_XI(RS, EP) \
    hipLaunchKernelGGL((k_x_inverse<RS, EP>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep)
#define M3D_LAUNCH_XIR(RS, EP) \
    hipLaunchKernelGGL((k_x_inverse_reg<RS, EP>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep)
    if (ep.chain && !(ep.kind == ConvEpilogue::kUpdate && fftconv_update_chain_supported(p) && ep.chain == spec &&
                      out_box && ep.core.z0 == out_box->z0 && ep.core.nz == out_box->nz && ep.core.y0 == out_box->y0 &&
                      ep.core.ny == out_box->ny && ep.core.x0 == out_box->x0 && ep.core.nx == out_box->nx))
        return M3D_ERR_INVALID_ARG;
    if (p->xstatic && ep.kind != ConvEpilogue::kConsensus) {
        if (ep.kind == ConvEpilogue::kUpdate && ep.chain)
            hipLaunchKernelGGL((k_x_inverse_reg<1, 2, 2187, true>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep);
        else if (ep.kind == ConvEpilogue::kUpdate)
            hipLaunchKernelGGL((k_x_inverse_reg<1, 2, 2187>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep);
        else
            hipLaunchKernelGGL((k_x_inverse_reg<1, 0, 2187>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep);
    } else if (p->xreg == 8 && ep.kind != ConvEpilogue::kConsensus) {
        if (ep.kind == ConvEpilogue::kUpdate)
            hipLaunchKernelGGL((k_x_inverse_reg<2, 2, 0, false, 8>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep);
        else
            hipLaunchKernelGGL((k_x_inverse_reg<2, 0, 0, false, 8>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep);
    } else if (p->xreg && p->rset_x == 2 && ep.kind != ConvEpilogue::kConsensus) {
        if (ep.kind == ConvEpilogue::kUpdate)
            hipLaunchKernelGGL((k_x_inverse_reg<2, 2, 0, false, 9>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep);
        else
            hipLaunchKernelGGL((k_x_inverse_reg<2, 0, 0, false, 9>), xg, xb, p->lds_x, st, spec, dst, g, p->ax, ep);
    } else if (p->xreg && ep.kind != ConvEpilogue::kConsensus) {
        if (p->rset_x) {
            if (ep.kind == ConvEpilogue::kUpdate) M3D_LAUNCH_XIR(1, 2);
            else M3D_LAUNCH_XIR(1, 0);
        } else {
            if (ep.kind == ConvEpilogue::kUpdate) M3D_LAUNCH_XIR(0, 2);
            else M3D_LAUNCH_XIR(0, 0);
        }
    } else if (p->rset_x == 2) {
        if (ep.kind == ConvEpilogue::kConsensus) M3D_LAUNCH_XI(2, 1);
        else if (ep.kind == ConvEpilogue::kUpdate) M3D_LAUNCH_XI(2, 2);
        else M3D_LAUNCH_XI(2, 0);
    } else
    if (p->rset_x) {
        if (ep.kind == ConvEpilogue::kConsensus) M3D_LAUNCH_XI(1, 1);
        else if (ep.kind == ConvEpilogue::kUpdate) M3D_LAUNCH_XI(1, 2);
        else M3D_LAUNCH_XI(1, 0);
    } else {
        if (ep.kind == ConvEpilogue::kConsensus) M3D_LAUNCH_XI(0, 1);
        else if (ep.kind == ConvEpilogue::kUpdate) M3D_LAUNCH_XI(0, 2);
        else M3D_LAUNCH_XI(0, 0);
    }
#undef M3D_LAUNCH_XI
#undef M3D_LAUNCH_XIR
    if (log && log->mark(EventLog::kXI)) return M3D_ERR_HIP;
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------- parity hook for the packed complex primitives
// out[i][slot]: the primitives of m3d_fft.h applied to (a[i], b[i], c[i]); slots 10.. hold the 3-, 4-, 8- and
// 9-point butterflies (forward, then inverse) of the sequence a, b, c, a + b, b - c, c * a (element by element), a - c,
// b + b, c - a.
namespace {
__global__ void k_debug_pk(const float2 *__restrict__ a, const float2 *__restrict__ b, const float2 *__restrict__ c,
                           int64_t n, float2 *__restrict__ out)
{
    using namespace m3d;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 A = a[i], B = b[i], C = c[i];
    float2 *o = out + i * kDebugPkSlots;
    const float2 s = make_float2(C.x, C.x);
    o[0] = add_i(A, B);
    o[1] = sub_i(A, B);
    o[2] = conj_add_i_conj(A, B);
    o[3] = add_conj(A, B);
    o[4] = split_odd(A, B);
    o[5] = make_float2(0.f, 0.f);
    o[6] = make_float2(0.f, 0.f);
    o[7] = cmul(A, B);
    o[8] = cmul_conj(A, B);
    o[9] = ctw<false>(A, 0.6f, 0.8f);
    // scalar-register pairs: the value must be wave uniform - taken from element 0
    const float su = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s.x)));
    const float2 s0 = make_float2(su, su);
    o[5] = fma_i(A, s0, B);
    o[6] = fms_i(A, s0, B);
    const float2 seq[9] = {A, B, C, A + B, B - C, C * A, A - C, B + B, C - A};
    int at = 10;
    auto run = [&](auto fn, int R) {
        float2 v[9];
        for (int q = 0; q < 9; ++q) v[q] = seq[q];
        fn(v);
        for (int q = 0; q < R; ++q) o[at++] = v[q];
    };
    run([](float2 *v) { dft3<false>(v[0], v[1], v[2]); }, 3);
    run([](float2 *v) { dft3<true>(v[0], v[1], v[2]); }, 3);
    run([](float2 *v) { dft4<false>(v[0], v[1], v[2], v[3]); }, 4);
    run([](float2 *v) { dft4<true>(v[0], v[1], v[2], v[3]); }, 4);
    run([](float2 *v) { dft8<false>(v); }, 8);
    run([](float2 *v) { dft8<true>(v); }, 8);
    run([](float2 *v) { dft9<false>(v); }, 9);
    run([](float2 *v) { dft9<true>(v); }, 9);
    // radix 27 (the twiddle-free stage of the 9 * 9 * 27 Y passes): the nine values and their products with B and with C
    auto run27 = [&](auto fn) {
        float2 v[27];
        for (int q = 0; q < 9; ++q) {
            v[q] = seq[q];
            v[9 + q] = cmul(seq[q], B);
            v[18 + q] = cmul(seq[q], C);
        }
        fn(v);
        for (int q = 0; q < 27; ++q) o[at++] = v[q];
    };
    run27([](float2 *v) { dft27<false>(v); });
    run27([](float2 *v) { dft27<true>(v); });
}
}  // namespace

int fft_debug_pk(const float2 *a, const float2 *b, const float2 *c, int64_t n, float2 *out, void *stream)
{
    if (n <= 0) return M3D_OK;
    hipLaunchKernelGGL(k_debug_pk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, n, out);
    M3D_KERNEL_CHECK();
    return M3D_OK;
}

}  // namespace m3d
