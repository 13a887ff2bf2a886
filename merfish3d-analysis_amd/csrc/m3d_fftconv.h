// Host interface of the fused FFT-convolution engine (m3d_fftconv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace m3d {

// Optional HIP-event log (bench.py roofline): record() drops an event on the stream.
struct EventLog {
    std::vector<hipEvent_t> *events = nullptr;
    size_t used = 0;
    hipStream_t stream = nullptr;
    bool enabled = false;
    int record()
    {
        if (!enabled) return 0;
        if (used == events->size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -3;
            events->push_back(e);
        }
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
// dst <- irfftn(rfftn(src) * H) with H = otf (mode 0), conj(otf) (1), |otf|^2 (2);
// `spec` is a scratch spectrum.  src may equal dst.
// With an enabled `log`, six events bracket the five kernels (x fwd, y fwd, z conv, y inv, x inv).
int fftconv_conv(const FftConvPlan *p, const float *src, const float2 *otf, int mode, float2 *spec,
                 float *dst, void *stream, EventLog *log = nullptr, const ConvBox *in_box = nullptr,
                 const ConvBox *out_box = nullptr, const ConvBox *mirror_box = nullptr);
// `mirror_box` (exclusive with in_box): src is only valid inside the box and stands for its
// symmetric extension to the whole volume (np.pad mode="symmetric", rlgc.py:203-271) - the
// X pass gathers through the index map instead of reading a materialised pad region.

}  // namespace m3d
