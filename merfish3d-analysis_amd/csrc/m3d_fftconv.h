// Host interface of the fused FFT-convolution engine (m3d_fftconv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "m3d_binom.h"

namespace m3d {

// Optional HIP-event log (bench.py roofline).  mark(kind) drops an event on the stream; the time since
// the previous event is attributed to `kind`: kStart opens a span (nothing attributed), kXF..kXI are the
// five passes of the fused engine, kWhole a convolution timed as one piece (rocFFT path).
struct EventLog {
    enum Kind { kStart = -1, kXF = 0, kYF = 1, kZ = 2, kYI = 3, kXI = 4, kWhole = 5 };
    std::vector<hipEvent_t> *events = nullptr;
    std::vector<int> kinds;
    size_t used = 0;
    int convolutions = 0;  // counted by the callers
    hipStream_t stream = nullptr;
    bool enabled = false;
    int mark(int kind)
    {
        if (!enabled) return 0;
        if (used == events->size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -3;
            events->push_back(e);
        }
        kinds.push_back(kind);
        return hipEventRecord((*events)[used++], stream) == hipSuccess ? 0 : -3;
    }
};

struct FftConvPlan;

// Support box (z, y, x offsets and extents in the padded volume).  As `in_box`: src is zero
// outside it, so rows / planes outside are neither read nor transformed.  As `out_box`: only
// that part of dst is needed, so the inverse passes skip everything else (dst keeps stale
// values outside the box).
struct ConvBox {
    int z0, nz, y0, ny, x0, nx;
};

// Builds radix lists and device tables for a real volume Zp x Yp x Xp (all 2^a 3^b).
// Returns M3D_ERR_UNSUPPORTED when a length has other factors or a pass cannot fit LDS.
int fftconv_create(int Zp, int Yp, int Xp, FftConvPlan **out);
void fftconv_destroy(FftConvPlan *p);
// complex elements of one tiled spectrum (>= Zp*Yp*(Xp/2+1); includes tile padding)
int64_t fftconv_spectrum_elems(const FftConvPlan *p);
// otf <- forward transform of the padded, centred PSF volume
int fftconv_make_otf(const FftConvPlan *p, const float *psf_padded, float2 *otf, void *stream);
// Optional fused tail of the X inverse pass: the convolution result is combined with other
// volumes in the store loop instead of being written to dst and read back by the next kernel.
//   kConsensus (rlgc.py:661-678, full volume): HT1 = a / b, HT2 = result / b;
//       o1 <- (HT1 - 1)(HT2 - 1); o2 <- HT1 + HT2 on the core (o2 may alias a);
//       red[0] / red[1] <- min / max of HT1 + HT2 as ordered keys (float_key below)
//   kUpdate (filter_update rlgc.py:23-31, :682-694, core only): c = result, old = a;
//       o1 <- c < 0 ? old : old * b; red64[0] += (c >= 0); red[2] <- max o1 (key); red[3] <- max |o1 - old| (bits)
// The epilogues' reductions end in atomics.  A single record serialises them at one address (~12 ns each, 280 k
// workgroups: the whole kernel); workgroups spread over kRedSlots records on separate 128-byte lines instead, and
// the reader folds the records.
constexpr int kRedSlots = 128, kRedStride = 32;
struct ConvEpilogue {
    enum Kind { kNone = 0, kConsensus = 1, kUpdate = 2 };
    int kind;
    const float *a, *b;
    float *o1, *o2;
    unsigned int *red;  // kRedSlots records of 4 words, kRedStride words apart (128-byte aligned)
    ConvBox core;
    // kUpdate only, optional: the spectrum the convolution itself runs in.  The X forward transform of the updated
    // estimate (core rows, symmetric extension along x) is written over the rows just read - the first pass of the next
    // iteration's forward-model convolution, which then starts at its mirrored Y pass (x_forward_done).  Honoured
    // where fftconv_update_chain_supported says so; ignored elsewhere.
    float2 *chain = nullptr;
};
bool fftconv_update_chain_supported(const struct FftConvPlan *p);
// M3D_ENGINE_X_STATIC | M3D_ENGINE_Y_STATIC | M3D_ENGINE_Z_REGISTER of this plan's kernel selection
int fftconv_engine_flags(const struct FftConvPlan *p);
// order-preserving map float -> uint32 (atomicMin / atomicMax on floats of either sign)
__host__ __device__ inline unsigned int float_key(float f)
{
    unsigned int b;
    __builtin_memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float key_float(unsigned int k)
{
    const unsigned int b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

// dst <- irfftn(rfftn(src) * H) with H = otf (mode 0), conj(otf) (1), |otf|^2 (2);
// `spec` is a scratch spectrum.  src may equal dst.  src == nullptr: `spec` already holds the X forward
// transform of the (full-volume) input, and the convolution starts at its Y pass.
// With an enabled `log` every kernel's time is attributed to its pass (EventLog::Kind).
int fftconv_conv(const FftConvPlan *p, const float *src, const float2 *otf, int mode, float2 *spec,
                 float *dst, void *stream, EventLog *log = nullptr, const ConvBox *in_box = nullptr,
                 const ConvBox *out_box = nullptr, const ConvBox *mirror_box = nullptr,
                 const ConvEpilogue *epilogue = nullptr, float2 *spec_alt = nullptr,
                 bool x_forward_done = false);  // mirror_box form only: `spec` already holds the core rows' X transforms
// `mirror_box` (exclusive with in_box): src is only valid inside the box and stands for its
// symmetric extension to the whole volume (np.pad mode="symmetric", rlgc.py:203-271) - the
// X pass gathers through the index map instead of reading a materialised pad region.  With a second
// scratch spectrum `spec_alt` only the box's own rows are transformed along x (mirrored rows have the
// same transform) and the Y pass assembles every plane from the core plane it mirrors.

// |OTF|^2 as a float volume (same plane order as the OTF): the consensus convolution (mode 2) then reads half the
// transfer-function bytes.  Pass it as `otf` with mode 3; needs the register Z pass (fftconv_real_otf_supported).
bool fftconv_real_otf_supported(const FftConvPlan *p);
int fftconv_otf_abs2(const FftConvPlan *p, const float2 *otf, float *out, void *stream);

// The split stage of an RLGC iteration fused between two convolutions (rlgc.py:619-672): the X inverse pass of the
// forward-model convolution (Hu on the core) does not store Hu but runs k_split_ratio's arithmetic on the registers
// of its last stage - binomial photon split, the three KL sums, the two ratio volumes - and hands both ratio rows
// straight to the X forward transforms of the two ratio convolutions.  Hu, ratio1 and ratio2 never exist in HBM (38 ->
// 14 bytes per core voxel and three launches less per iteration).  `spec_hu` holds the half-spectrum rows of Hu (after
// its inverse Y pass) and receives the rows of ratio2's X transform in place; `spec_r1` receives ratio1's.
struct SplitPartial {  // one per workgroup; summed in a fixed order by the caller
    double A, Bim, B1, B2, Tim, T1, T2;
};
struct SplitArgs {
    const void *image;      // the observed image [core.nz][core.ny][core.nx], uint16 or float32
    int image_is_u16;
    const float *injected;  // optional split1 volume of the same shape (parity hook), else nullptr
    uint32_t iteration;
    uint64_t seed;
    BinomTables tables;
    SplitPartial *partial;  // fftconv_split_chain_blocks() records
};
bool fftconv_split_chain_supported(const FftConvPlan *p);
int64_t fftconv_split_chain_blocks(const FftConvPlan *p, const ConvBox *core);
int fftconv_split_chain(const FftConvPlan *p, float2 *spec_hu, float2 *spec_r1, const ConvBox *core,
                        const SplitArgs *args, void *stream, EventLog *log);

// fftconv_to_rows for two inputs that take the same transfer function (the two HTratio convolutions of an RLGC
// iteration): the Z pass of both runs as one launch that reads each OTF line from HBM once.
// x_forward_done: the spectra already hold the X transforms of the inputs' rows (fftconv_split_chain); src1 / src2 unused.
int fftconv_to_rows_pair(const FftConvPlan *p, const float *src1, const float *src2, const float2 *otf, int mode,
                         float2 *spec1, float2 *spec2, void *stream, EventLog *log, const ConvBox *in_box,
                         bool x_forward_done = false);

// fft_conv without its X inverse pass: afterwards `spec` holds the half spectrum along x of every
// (z, y) row of the result (rows of `out_box` only).
int fftconv_to_rows(const FftConvPlan *p, const float *src, const float2 *otf, int mode, float2 *spec,
                    void *stream, EventLog *log = nullptr, const ConvBox *in_box = nullptr,
                    const ConvBox *out_box = nullptr, const ConvBox *mirror_box = nullptr,
                    float2 *spec_alt = nullptr, float2 **rows = nullptr,  // *rows: the buffer that holds them
                    bool x_forward_done = false);
// The X inverse pass shared by two such results that meet in the consensus arithmetic: the same row of
// both spectra is one packed complex row (S1 + i S2), one inverse transform returns both real rows and
// ConvEpilogue::kConsensus runs on them in registers (`a` unused: HT1's numerator is the first result).
// Neither convolution result reaches HBM.
// `chain_forward`: the consensus input is not stored (o1 unused) but transformed along x on the spot;
// its half-spectrum rows replace spec1, ready for fftconv_conv(src = nullptr, spec = spec1, ...).
int fftconv_dual_consensus(const FftConvPlan *p, float2 *spec1, const float2 *spec2, void *stream,
                           EventLog *log, const ConvEpilogue *epilogue, bool chain_forward = false);

// Parity hook (tests): the packed complex primitives and small butterflies of m3d_fft.h on n element triples;
// out holds kDebugPkSlots values per element (layout in k_debug_pk).
constexpr int kDebugPkSlots = 10 + 2 * (3 + 4 + 8 + 9 + 27);
int fft_debug_pk(const float2 *a, const float2 *b, const float2 *c, int64_t n, float2 *out, void *stream);

}  // namespace m3d
