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
// Radices 2,3,4,8,9,16,27 (128 = 16*8, 2187 = 27*9*9); a thread owns one butterfly index
// for up to four interleaved transforms, so twiddles and index arithmetic are shared.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace m3d {

constexpr int kMaxStages = 16;

// Workgroup barrier that only drains LDS traffic.  __syncthreads() also emits
// s_waitcnt vmcnt(0), which would retire the register prefetch of the next tile (and the
// trailing stores of the previous one) at every FFT stage and serialise HBM behind compute.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef M3D_TW_POWERS
#define M3D_TW_POWERS 1
#endif

struct AxisPlan {
    int n;                 // transform length (2^a 3^b)
    int nstages;
    int radix[kMaxStages];  // DIF order: first stage splits the full length
    const float2 *tw;       // device: exp(-2 pi i k / n), k in [0, n)
    const int *pos;         // device: position of frequency k after the DIF pass
};

// Complex arithmetic on the two-lane vector type: the passes are vector-ALU bound (one wave instruction per four
// cycles per SIMD, whatever its kind), and v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 handle real and imaginary part
// in one issue.  Sums, differences and products with real pairs are written on HIP's float2 operators (the compiler
// selects the packed forms).  What it does NOT select is a swap of the halves combined with a sign on ONE of them -
// every multiplication by +-i and every complex product is that - although the instruction encodes both for free
// (op_sel / op_sel_hi choose the half of each source per result lane, neg_lo / neg_hi negate a source per lane): it
// emits v_xor + v_mov in front of the packed operation instead (27 % of the dual X pass's vector instructions in
// round 2).  Those forms are spelled out below as one instruction each.
// (Inline asm is opaque to the compiler's hazard recogniser: an operand must not be the direct result of a transcendental
// instruction - v_rcp / v_log / v_sqrt ... need one wait state before a vector consumer on gfx94x/95x, which the compiler
// inserts for its own instructions only.  Nothing here is fed that way: operands are loads, LDS reads and other packed
// results.)
//   result.lo = [-]src0[op_sel[0]] * src1[op_sel[1]] + src2[op_sel[2]]          (neg_lo)
//   result.hi = [-]src0[op_sel_hi[0]] * src1[op_sel_hi[1]] + src2[op_sel_hi[2]]  (neg_hi)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Streaming accesses (the `nt` bit): every spectrum, OTF and volume element of a pass is touched once per launch and
// the working set of an iteration is 70 GB, so nothing a pass reads or writes is worth keeping in L2 / the Infinity
// Cache for the next one.  Measured on the Y passes (profiles/r03_streaming_accesses.txt): -4 % with loads AND stores
// marked, nothing with either alone.
__device__ __forceinline__ float4 ld_stream(const float4 *p)
{
    const v4f t = __builtin_nontemporal_load((const v4f *)p);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st_stream(float4 *p, float4 v)
{
    const v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (v4f *)p);
}
__device__ __forceinline__ float2 ld_stream(const float2 *p)
{
    const v2f t = __builtin_nontemporal_load((const v2f *)p);
    return make_float2(t.x, t.y);
}
__device__ __forceinline__ void st_stream(float2 *p, float2 v)
{
    const v2f t = {v.x, v.y};
    __builtin_nontemporal_store(t, (v2f *)p);
}
__device__ __forceinline__ float ld_stream(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ v2f nv(float2 a)
{
    v2f r = {a.x, a.y};
    return r;
}
__device__ __forceinline__ float2 f2(v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 swap_xy(float2 a) { return make_float2(a.y, a.x); }
__device__ __forceinline__ float2 xx(float2 a) { return make_float2(a.x, a.x); }
__device__ __forceinline__ float2 yy(float2 a) { return make_float2(a.y, a.y); }
// b + i x = (b.x - x.y, b.y + x.x)
__device__ __forceinline__ float2 add_i(float2 b, float2 x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(nv(b)), "v"(nv(x)));
    return f2(r);
}
// b - i x = (b.x + x.y, b.y - x.x)
__device__ __forceinline__ float2 sub_i(float2 b, float2 x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(nv(b)), "v"(nv(x)));
    return f2(r);
}
// conj(a) + i conj(b) = (a.x + b.y, b.x - a.y): the mirror element of the Hermitian merge
__device__ __forceinline__ float2 conj_add_i_conj(float2 a, float2 b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(nv(a)), "v"(nv(b)));
    return f2(r);
}
// a + conj(b) = (a.x + b.x, a.y - b.y) and -i (a - conj(b)) = (a.y + b.y, b.x - a.x): the two halves of the Hermitian
// split (before the factor 1/2)
__device__ __forceinline__ float2 add_conj(float2 a, float2 b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(nv(a)), "v"(nv(b)));
    return f2(r);
}
__device__ __forceinline__ float2 split_odd(float2 a, float2 b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(nv(a)), "v"(nv(b)));
    return f2(r);
}
// m + i s x and m - i s x for a real s held in a scalar register pair: s must be uniform over the wave (the butterfly
// constants are; a per-lane value would be read from its first lane only).  Both result lanes
// read the LOW register of the pair (op_sel_hi 0): a packed-f32 instruction cannot take the high half of a scalar
// source for its high lane - the compiler's own packed code never does either (tests/test_gpu_fft_primitives.py
// pins the form used here).
__device__ __forceinline__ float2 fma_i(float2 x, float2 s, float2 m)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]"
        : "=v"(r)
        : "v"(nv(x)), "s"(nv(s)), "v"(nv(m)));
    return f2(r);
}
__device__ __forceinline__ float2 fms_i(float2 x, float2 s, float2 m)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]"
        : "=v"(r)
        : "v"(nv(x)), "s"(nv(s)), "v"(nv(m)));
    return f2(r);
}
// a b = (a.x b.x - a.y b.y, a.y b.x + a.x b.y): (a.x, a.y) b.x, then (-a.y, a.x) b.y added
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(nv(a)), "v"(nv(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r)
        : "v"(nv(a)), "v"(nv(b)), "v"(t));
    return f2(r);
}
// a conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(nv(a)), "v"(nv(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r)
        : "v"(nv(a)), "v"(nv(b)), "v"(t));
    return f2(r);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return a + b; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return a - b; }
// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot90(float2 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// b + W4 x and b - W4 x with W4 = -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 add_rot90(float2 b, float2 x)
{
    return INV ? add_i(b, x) : sub_i(b, x);
}
template <bool INV>
__device__ __forceinline__ float2 sub_rot90(float2 b, float2 x)
{
    return INV ? sub_i(b, x) : add_i(b, x);
}
// multiply by the constant exp(-+ i theta) given (cos, sin): a c -+ i s a
template <bool INV>
__device__ __forceinline__ float2 ctw(float2 a, float c, float s)
{
    const float2 t = a * make_float2(c, c);
    return INV ? fma_i(a, make_float2(s, s), t) : fms_i(a, make_float2(s, s), t);
}

template <bool INV>
__device__ __forceinline__ void dft2(float2 &a, float2 &b)
{
    const float2 t = a;
    a = t + b;
    b = t - b;
}
template <bool INV>
__device__ __forceinline__ void dft3(float2 &v0, float2 &v1, float2 &v2)
{
    const float2 s = make_float2(0.86602540378443864676f, 0.86602540378443864676f);
    const float2 t1 = v1 + v2;
    const float2 d = v1 - v2;
    const float2 t2 = v0 - t1 * make_float2(0.5f, 0.5f);
    v0 = v0 + t1;
    v1 = INV ? fma_i(d, s, t2) : fms_i(d, s, t2);
    v2 = INV ? fms_i(d, s, t2) : fma_i(d, s, t2);
}
template <bool INV>
__device__ __forceinline__ void dft4(float2 &v0, float2 &v1, float2 &v2, float2 &v3)
{
    const float2 a = cadd(v0, v2), b = csub(v0, v2);
    const float2 c = cadd(v1, v3), e = csub(v1, v3);
    v0 = cadd(a, c);
    v2 = csub(a, c);
    v1 = add_rot90<INV>(b, e);
    v3 = sub_rot90<INV>(b, e);
}
template <bool INV>
__device__ __forceinline__ void dft8(float2 *v)
{
    // x[2a + b] -> X[c + 4d]
    const float h = 0.70710678118654752440f;
    dft4<INV>(v[0], v[2], v[4], v[6]);
    dft4<INV>(v[1], v[3], v[5], v[7]);
    v[3] = ctw<INV>(v[3], h, h);
    v[7] = ctw<INV>(v[7], -h, h);
    float2 o[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float2 a = v[2 * c], b = v[2 * c + 1];
        if (c == 2) {  // v[5] carries W8^2 = -+i
            o[c] = add_rot90<INV>(a, b);
            o[c + 4] = sub_rot90<INV>(a, b);
            continue;
        }
        dft2<INV>(a, b);
        o[c] = a;
        o[c + 4] = b;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = o[i];
}
template <bool INV>
__device__ __forceinline__ void dft9(float2 *v)
{
    // x[3a + b] -> X[c + 3d]
    dft3<INV>(v[0], v[3], v[6]);
    dft3<INV>(v[1], v[4], v[7]);
    dft3<INV>(v[2], v[5], v[8]);
    v[4] = ctw<INV>(v[4], 0.76604444311897803520f, 0.64278760968653932632f);   // W9^1
    v[7] = ctw<INV>(v[7], 0.17364817766693034885f, 0.98480775301220805937f);   // W9^2
    v[5] = ctw<INV>(v[5], 0.17364817766693034885f, 0.98480775301220805937f);   // W9^2
    v[8] = ctw<INV>(v[8], -0.93969262078590838405f, 0.34202014332566873304f);  // W9^4
    // now v[3c + b]... values u_b[c] sit at v[3c + b]; transform over b for each c
    float2 o[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float2 a0 = v[3 * c], a1 = v[3 * c + 1], a2 = v[3 * c + 2];
        dft3<INV>(a0, a1, a2);
        o[c] = a0;
        o[c + 3] = a1;
        o[c + 6] = a2;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = o[i];
}
template <bool INV>
__device__ __forceinline__ void dft16(float2 *v)
{
    // x[4a + b] -> X[c + 4d]
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // u_b[c] sits at v[4c + b]; twiddle W16^(b c)
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
    const float h = 0.70710678118654752440f;
    v[4 + 1] = ctw<INV>(v[4 + 1], c1, s1);     // W^1
    v[4 + 2] = ctw<INV>(v[4 + 2], h, h);       // W^2
    v[4 + 3] = ctw<INV>(v[4 + 3], s1, c1);     // W^3
    v[8 + 1] = ctw<INV>(v[8 + 1], h, h);       // W^2
    v[8 + 2] = rot90<INV>(v[8 + 2]);           // W^4
    v[8 + 3] = ctw<INV>(v[8 + 3], -h, h);      // W^6
    v[12 + 1] = ctw<INV>(v[12 + 1], s1, c1);   // W^3
    v[12 + 2] = ctw<INV>(v[12 + 2], -h, h);    // W^6
    v[12 + 3] = ctw<INV>(v[12 + 3], -c1, -s1); // W^9
    float2 o[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float2 a0 = v[4 * c], a1 = v[4 * c + 1], a2 = v[4 * c + 2], a3 = v[4 * c + 3];
        dft4<INV>(a0, a1, a2, a3);
        o[c] = a0;
        o[c + 4] = a1;
        o[c + 8] = a2;
        o[c + 12] = a3;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = o[i];
}
template <bool INV>
__device__ __forceinline__ void dft27(float2 *v)
{
    // x[9a + b] (a < 3, b < 9) -> X[c + 3d] (c < 3, d < 9)
#pragma unroll
    for (int b = 0; b < 9; ++b) dft3<INV>(v[b], v[9 + b], v[18 + b]);
    // u_b[c] sits at v[9c + b]; twiddle W27^(b c)
    const float cs[17][2] = {
        {1.0f, 0.0f},
        {0.97304487057982381f, 0.23061587074244017f},  {0.89363264032341228f, 0.44879918020046217f},
        {0.76604444311897801f, 0.64278760968653925f},  {0.59715859170278618f, 0.80212319275504373f},
        {0.39607976603915690f, 0.91821610688027400f},  {0.17364817766693041f, 0.98480775301220802f},
        {-0.058144828910475774f, 0.99830815827126818f}, {-0.28680323271109021f, 0.95798951231548887f},
        {-0.49999999999999978f, 0.86602540378443871f}, {-0.68624163786873349f, 0.72737364157304885f},
        {-0.83548781141293627f, 0.54950897807080624f}, {-0.93969262078590832f, 0.34202014332566888f},
        {-0.99323835774194302f, 0.11609291412522993f}, {-0.99323835774194302f, -0.11609291412523012f},
        {-0.93969262078590854f, -0.34202014332566821f}, {-0.83548781141293649f, -0.54950897807080601f}};
#pragma unroll
    for (int b = 1; b < 9; ++b) {
        v[9 + b] = ctw<INV>(v[9 + b], cs[b][0], cs[b][1]);
        v[18 + b] = ctw<INV>(v[18 + b], cs[2 * b][0], cs[2 * b][1]);
    }
    float2 o[27];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float2 t[9];
#pragma unroll
        for (int b = 0; b < 9; ++b) t[b] = v[9 * c + b];
        dft9<INV>(t);
#pragma unroll
        for (int d = 0; d < 9; ++d) o[c + 3 * d] = t[d];
    }
#pragma unroll
    for (int i = 0; i < 27; ++i) v[i] = o[i];
}

template <int R, bool INV>
__device__ __forceinline__ void dft_r(float2 *v)
{
    if constexpr (R == 2) dft2<INV>(v[0], v[1]);
    if constexpr (R == 3) dft3<INV>(v[0], v[1], v[2]);
    if constexpr (R == 4) dft4<INV>(v[0], v[1], v[2], v[3]);
    if constexpr (R == 8) dft8<INV>(v);
    if constexpr (R == 9) dft9<INV>(v);
    if constexpr (R == 16) dft16<INV>(v);
    if constexpr (R == 27) dft27<INV>(v);
}

// One in-place stage over `batch` interleaved transforms of length n held in LDS.
// Element k of transform b lives at data[k * estride + b * bstride]; M = current block
// size.  A work item = one butterfly index t for a group of G = min(batch, 4) transforms
// (twiddles and the (block, j) split are computed once per item).  batch / G must be a
// power of two.
// BLOCK_LANES: items run over the blocks first (lane -> block, then j) instead of over j first: neighbouring lanes
// are M elements apart.  For the radix-3 stage of block size 27 that is an odd stride - conflict free - where j-first
// order reads 9 adjacent elements, skips 18 and wraps onto its own banks (two LDS cycles per access group).
template <int R, bool INV, bool BLOCK_LANES = false>
__device__ __forceinline__ void lds_stage(float2 *data, int n, int M, int estride, int batch,
                                          int bstride, const float2 *__restrict__ tw, int tid,
                                          int nthreads, int gmax = 4)
{
    const int sub = M / R;      // butterflies per block
    const int per_fft = n / R;  // butterflies per transform
    const int twstep = n / M;
    const int G = batch < gmax ? batch : gmax;  // gmax in {1, 2, 4}
    const int nbg = batch / G;
    const int bshift = __builtin_ctz((unsigned)nbg);
    const int total = per_fft << bshift;
    const uint32_t magic = sub > 1 ? (uint32_t)(0x100000000ull / (uint32_t)sub) + 1u : 0u;
    const int step = sub * estride;
    for (int i = tid; i < total; i += nthreads) {
        const int bg = i & (nbg - 1);
        const int t = i >> bshift;
        int blk, j;
        if constexpr (BLOCK_LANES) {
            const int nblk = n / M;
            j = t / nblk;
            blk = t - j * nblk;
        } else if (sub > 1) {
            blk = (int)__umulhi((uint32_t)t, magic);
            j = t - blk * sub;
        } else {
            blk = t;
            j = 0;
        }
        float2 w[R];
        // Twiddles W^(j q): a few table loads, the other powers one or two complex products away (<= 2 extra
        // roundings each) - the loads go through an L1 that the streaming traffic of the pass keeps evicting,
        // and a stage waits for them.  M3D_TW_POWERS: 0 = every power from the table, 1 = 1,2,4 from the table
        // (radix 9), 2 = only the first from the table.
        const int j1 = j * twstep;
        if constexpr (R == 9 && M3D_TW_POWERS == 1) {
            w[1] = tw[j1];
            w[2] = tw[2 * j1];
            w[4] = tw[4 * j1];
            w[3] = cmul(w[1], w[2]);
            w[5] = cmul(w[1], w[4]);
            w[6] = cmul(w[2], w[4]);
            w[7] = cmul(w[3], w[4]);
            w[8] = cmul(w[4], w[4]);
        } else if constexpr (R == 9 && M3D_TW_POWERS == 2) {
            w[1] = tw[j1];
            w[2] = cmul(w[1], w[1]);
            w[4] = cmul(w[2], w[2]);
            w[3] = cmul(w[1], w[2]);
            w[5] = cmul(w[1], w[4]);
            w[6] = cmul(w[2], w[4]);
            w[7] = cmul(w[3], w[4]);
            w[8] = cmul(w[4], w[4]);
        } else if constexpr (R == 8 && M3D_TW_POWERS >= 1) {
            w[1] = tw[j1];
            w[2] = tw[2 * j1];
            w[4] = tw[4 * j1];
            w[3] = cmul(w[1], w[2]);
            w[5] = cmul(w[1], w[4]);
            w[6] = cmul(w[2], w[4]);
            w[7] = cmul(w[3], w[4]);
        } else if constexpr ((R == 3 || R == 4) && M3D_TW_POWERS >= 1) {
            w[1] = tw[j1];
            w[2] = cmul(w[1], w[1]);
            if constexpr (R == 4) w[3] = cmul(w[1], w[2]);
        } else {
#pragma unroll
            for (int q = 1; q < R; ++q) w[q] = tw[q * j1];
        }
        float2 *p0 = data + (size_t)(blk * M + j) * estride + (size_t)(bg * G) * bstride;
        // interleaved layouts (bstride == 1): lanes 8 apart sit 256 B apart = same LDS banks;
        // rotating the member each lane starts with makes every access conflict free
        const int rot = (bstride == 1) ? (i >> 3) : 0;
        for (int it = 0; it < G; ++it) {
            const int gi = (it + rot) & (G - 1);
            float2 *p = p0 + (size_t)gi * bstride;
            float2 v[R];
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = p[q * step];
            if (INV) {
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = cmul_conj(v[q], w[q]);
                dft_r<R, true>(v);
            } else {
                dft_r<R, false>(v);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = cmul(v[q], w[q]);
            }
#pragma unroll
            for (int q = 0; q < R; ++q) p[q * step] = v[q];
        }
    }
}

// RSET selects which butterflies a kernel is compiled with: 0 = all, 1 = lean {9,4,3,2}, 2 = lean + radix 8
// (registers: the radix-16 butterfly alone costs ~40 VGPRs more than radix 9).
template <bool INV, int RSET>
__device__ __forceinline__ void lds_stage_r(int r, float2 *data, int n, int M, int estride, int batch,
                                            int bstride, const float2 *__restrict__ tw, int tid,
                                            int nthreads, int gmax = 4)
{
    switch (r) {
#ifdef M3D_FFT_RADIX27  // costs ~180 VGPRs: halves the resident waves of every pass
        case 27: lds_stage<27, INV>(data, n, M, estride, batch, bstride, tw, tid, nthreads, gmax); break;
#endif
        case 16:
            if constexpr (RSET == 0) lds_stage<16, INV>(data, n, M, estride, batch, bstride, tw, tid, nthreads, gmax);
            break;
        case 9: lds_stage<9, INV>(data, n, M, estride, batch, bstride, tw, tid, nthreads, gmax); break;
        case 8:
            if constexpr (RSET == 0 || RSET == 2) lds_stage<8, INV>(data, n, M, estride, batch, bstride, tw, tid, nthreads, gmax);
            break;
        case 4: lds_stage<4, INV>(data, n, M, estride, batch, bstride, tw, tid, nthreads, gmax); break;
        case 3: lds_stage<3, INV>(data, n, M, estride, batch, bstride, tw, tid, nthreads, gmax); break;
        default: lds_stage<2, INV>(data, n, M, estride, batch, bstride, tw, tid, nthreads, gmax); break;
    }
}

// Full transform: DIF forward (natural -> permuted) or DIT inverse (permuted -> natural,
// unnormalised).  Every thread of the block must call it; ends with a barrier.
template <bool INV, int RSET = 0>
__device__ __forceinline__ void lds_fft(float2 *data, const AxisPlan &ap, int estride, int batch,
                                        int bstride, int tid, int nthreads, int gmax = 4)
{
    const int n = ap.n;
    if (!INV) {
        int M = n;
        for (int s = 0; s < ap.nstages; ++s) {
            const int r = ap.radix[s];
            lds_stage_r<false, RSET>(r, data, n, M, estride, batch, bstride, ap.tw, tid, nthreads, gmax);
            M /= r;
            __syncthreads();
        }
    } else {
        int M = 1;
        for (int s = ap.nstages - 1; s >= 0; --s) {
            const int r = ap.radix[s];
            M *= r;
            lds_stage_r<true, RSET>(r, data, n, M, estride, batch, bstride, ap.tw, tid, nthreads, gmax);
            __syncthreads();
        }
    }
}

}  // namespace m3d
