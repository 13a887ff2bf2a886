/* m3d_hip.h - C ABI of libm3d_hip.so, the MI355X (gfx950) implementation of the
 * per-tile preprocess -> decode hot path of QI2lab/merfish3d-analysis.
 *
 * The reference has no FFI: its boundary is a Python operator API
 * (SURVEY.md section 8b).  Each entry point below names the reference
 * function (file:line under src/merfish3danalysis/) whose device work it
 * replaces; the Python host layer in merfish3d-analysis_amd/ keeps that
 * function's name and signature and calls through here.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host;
 *  - volumes are Z,Y,X C-contiguous; shapes are int32[3] in Z,Y,X order;
 *  - `stream` is a hipStream_t passed as void* (0 = default stream);
 *    entry points are asynchronous on it unless documented otherwise;
 *  - return value: M3D_OK (0) or a negative m3d_status; no exceptions, no
 *    allocation except inside *_plan_create;
 *  - the caller owns every buffer, including the workspace a plan asks for.
 */
#ifndef M3D_HIP_H
#define M3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum m3d_status {
    M3D_OK = 0,
    M3D_ERR_INVALID_ARG = -1,   /* -> ValueError   (rlgc.py:160,844-860; PixelDecoder.py:502) */
    M3D_ERR_OUT_OF_MEMORY = -2, /* -> MemoryError  (rlgc.py:760-780 OOM fallback)            */
    M3D_ERR_HIP = -3,           /* -> RuntimeError                                            */
    M3D_ERR_ROCFFT = -4,        /* -> RuntimeError                                            */
    M3D_ERR_WORKSPACE = -5,     /* -> ValueError: workspace smaller than the plan needs       */
    M3D_ERR_UNSUPPORTED = -6    /* -> NotImplementedError                                     */
} m3d_status;

typedef enum m3d_dtype { M3D_U16 = 0, M3D_F32 = 1, M3D_F16 = 2, M3D_I16 = 3 } m3d_dtype;

int m3d_version(void);
/* SHA-256 (64 hex digits) of the sources this library was compiled from (csrc/build.py links it in); the Python
 * binding compares it with the tree's and refuses a stale library.  No reference counterpart (the reference has no
 * native code): build hygiene of this boundary. */
const char *m3d_source_hash(void);
const char *m3d_status_string(int status);
/* last HIP / rocFFT error text of the calling thread (host string, never NULL) */
const char *m3d_last_error(void);

/* ------------------------------------------------------------------------------------
 * RLGC deconvolution  (utils/rlgc.py)
 * ---------------------------------------------------------------------------------- */

/* next_gpu_fft_size, rlgc.py:72-97 */
int m3d_next_fft_size(int x);
/* _axis_linear_fft_padding + pad_for_linear_fft geometry, rlgc.py:100-173.
 * Outputs: padded shape and the (before, after) pad widths per axis. */
int m3d_rlgc_geometry(const int32_t image_zyx[3], const int32_t psf_zyx[3], int pad_yx,
                      int32_t padded_zyx[3], int32_t pad_before[3], int32_t pad_after[3]);

typedef struct m3d_rlgc_plan m3d_rlgc_plan;

typedef struct m3d_rlgc_params {
    int32_t safe_mode;      /* rlgc.py:630-633: stop when EITHER split KLD rises            */
    float limit;            /* rlgc.py:716: min updated fraction (0.01)                     */
    float max_delta;        /* rlgc.py:727: max relative delta threshold (0.001)            */
    uint64_t seed;          /* rlgc.py:564: split seed (42; +tile_idx when tiled, :983)     */
    int32_t normalize_psf;  /* rlgc.py:310-312                                              */
    int32_t max_iterations; /* 0 = unbounded like the reference; >0 is a test/bench bound   */
    int32_t profile;        /* bit 0: HIP events around every FFT pass; bit 1: only the two events around the
                               iteration loop (loop_ms); 0: none (bench.py)                                  */
    int32_t reserved;
} m3d_rlgc_params;

#define M3D_RLGC_TRACE_CAP 512
/* m3d_rlgc_result.engine_flags */
#define M3D_ENGINE_X_STATIC 1     /* register-edge X passes compiled for the 2187-point axis              */
#define M3D_ENGINE_Y_STATIC 2     /* static-geometry Y passes                                             */
#define M3D_ENGINE_SPLIT_CHAIN 4  /* photon split fused between the Hu and the ratio convolutions' X passes */
#define M3D_ENGINE_UPDATE_CHAIN 8 /* update pass chained into the next forward transform                  */
#define M3D_ENGINE_Z_REGISTER 16  /* register Z pass (padded Z = 32 / 64 / 128 / 256)                      */

typedef struct m3d_rlgc_result {
    int32_t iterations;  /* completed multiplicative updates (num_iters, rlgc.py:698)      */
    int32_t stop_reason; /* 1 restore_previous_recon, 2 limit, 3 max_delta, 4 max_iterations */
    int32_t n_trace;     /* entries valid in the arrays below (<= M3D_RLGC_TRACE_CAP)       */
    int32_t engine_flags; /* which kernel families ran (diagnostic; tests assert on it): M3D_ENGINE_* below */
    double kld_image[M3D_RLGC_TRACE_CAP];          /* rlgc.py:626 */
    double kld_split1[M3D_RLGC_TRACE_CAP];         /* rlgc.py:627 */
    double kld_split2[M3D_RLGC_TRACE_CAP];         /* rlgc.py:628 */
    double updated_fraction[M3D_RLGC_TRACE_CAP];   /* rlgc.py:689 */
    double max_relative_delta[M3D_RLGC_TRACE_CAP]; /* rlgc.py:692 */
    double update_min[M3D_RLGC_TRACE_CAP];         /* rlgc.py:690 */
    double update_max[M3D_RLGC_TRACE_CAP];         /* rlgc.py:691 */
    double conv_ms_total; /* profile: summed device time of the fft_conv calls (rlgc.py:316-350) */
    int32_t conv_calls;   /* profile: number of fft_conv calls timed                          */
    int32_t fused_engine; /* 1: in-LDS engine (kernel_ms valid), 0: rocFFT                    */
    double kernel_ms[5];  /* profile, fused engine: summed time of k_x_forward, k_y_pass fwd,
                             k_z_conv, k_y_pass inv, k_x_inverse over all timed convolutions  */
    double loop_ms;       /* profile: device time of the whole iteration loop (first Hu convolution
                             to the last update), i.e. iterations x everything one iteration runs */
} m3d_rlgc_result;

/* FFT engine for plans created afterwards: 0 = the fused in-LDS convolution engine
 * (m3d_fftconv.hip; rocFFT only if a padded length cannot be handled), 1 = rocFFT. */
int m3d_set_fft_engine(int engine);
int m3d_rlgc_plan_uses_rocfft(const m3d_rlgc_plan *plan);

/* Builds the FFT tables / plans for the padded shape (replaces the cuFFT plan cache +
 * _fft_cache_3d, rlgc.py:36-69, 340-347).  Host-synchronous. */
int m3d_rlgc_plan_create(const int32_t image_zyx[3], const int32_t psf_zyx[3], int pad_yx,
                         m3d_rlgc_plan **plan);
size_t m3d_rlgc_plan_workspace_bytes(const m3d_rlgc_plan *plan);
int m3d_rlgc_plan_destroy(m3d_rlgc_plan *plan);

/* rlgc(), rlgc.py:497-757: pad -> OTF -> iterate -> crop, everything device resident.
 *  image            : u16 or f32 [Z,Y,X] (image_dtype = M3D_U16 | M3D_F32)
 *  psf              : f32 [Zp,Yp,Xp] as given to plan_create
 *  out              : f32 [Z,Y,X]
 *  injected_splits  : optional f32 [n_injected, Z,Y,X]; iteration k < n_injected uses
 *                     row k as split1 instead of the Philox draw (parity hook)
 *  result_host      : optional host struct filled before return
 * Blocks the host once per iteration (one 64-byte read-back), returns when done. */
int m3d_rlgc_run(m3d_rlgc_plan *plan, const void *image, int image_dtype, const float *psf,
                 float *out, void *workspace, size_t workspace_bytes,
                 const m3d_rlgc_params *params, const float *injected_splits, int n_injected,
                 m3d_rlgc_result *result_host, void *stream);

/* ---- the solver's building blocks, stand-alone (module-level functions of utils/rlgc.py;
 *      m3d_rlgc_run runs fused versions of the same arithmetic) ---------------------------- */
/* filter_update ElementwiseKernel, rlgc.py:23-31: out = consensus < 0 ? recon : recon * HTratio */
int m3d_filter_update(const float *recon, const float *htratio, const float *consensus_map, float *out, int64_t n,
                      void *stream);
/* enforce_symmetric_boundary, rlgc.py:230-271 (+ _symmetric_padded_axis_indices :203-227), in place */
int m3d_enforce_symmetric_boundary(float *volume, const int32_t padded_zyx[3], const int32_t pad_before[3],
                                   const int32_t pad_after[3], void *stream);
/* _observed_region_mask, rlgc.py:353-379 */
int m3d_observed_region_mask(float *mask, const int32_t padded_zyx[3], const int32_t pad_before[3],
                             const int32_t pad_after[3], void *stream);
/* pad_psf, rlgc.py:274-313: zero-embed, centre (the 6 rolls + ifftshift), optional unit sum */
size_t m3d_pad_psf_workspace_bytes(void);
int m3d_pad_psf(const float *psf, const int32_t psf_zyx[3], float *out, const int32_t padded_zyx[3], int normalize,
                void *work, size_t work_bytes, void *stream);
/* kl_div, rlgc.py:382-412 (mask may be NULL); synchronises the stream and writes one host double */
size_t m3d_kl_div_workspace_bytes(void);
int m3d_kl_div(const float *p, const float *q, const float *mask, int64_t n, void *work, size_t work_bytes,
               double *result_host, void *stream);
/* fft_conv, rlgc.py:316-350, on the fused FFT engine (axes 2^a 3^b; else M3D_ERR_UNSUPPORTED):
 * the transfer function lives in the engine's tiled spectrum layout, so it is made by m3d_fftconv_otf from the
 * padded PSF (rlgc.py:582 `rfftn(psf)`) and applied as H (mode 0), conj H (1) or |H|^2 (2) (rlgc.py:583-585).
 * `otf` and `scratch_spectrum` are device buffers of m3d_fftconv_spectrum_bytes(); image may equal out. */
typedef struct m3d_fftconv m3d_fftconv;
int m3d_fftconv_create(const int32_t shape_zyx[3], m3d_fftconv **out);
int m3d_fftconv_destroy(m3d_fftconv *handle);
size_t m3d_fftconv_spectrum_bytes(const m3d_fftconv *handle);
int m3d_fftconv_otf(m3d_fftconv *handle, const float *psf_padded, void *otf, void *stream);
int m3d_fftconv_apply(m3d_fftconv *handle, const float *image, const void *otf, int mode, void *scratch_spectrum,
                      float *out, void *stream);

/* Philox split of one iteration, for the parity tests of the sampler itself. */
int m3d_rlgc_debug_split(const void *image, int image_dtype, int64_t n, uint64_t seed,
                         int32_t iteration, float *split1, void *stream);

/* The split sampler's tables as the library built them (host memory, valid for the life of the process; no GPU
   call): thresholds floor(2^63 P(|X - 32 q| <= i)) of X ~ Binomial(64 q, 1/2); row q (2 <= q <= *q_max) is given as an
   (offset, length) pair, rows have one stride (n_thresholds / (q_max + 1)) and sit between sentinels:
   thresholds[offset] = 0, thresholds[offset + 1 + i] = threshold i, 2^64 - 1 from offset + 1 + length to the end of
   the row.  For the parity test against the oracle's exact integers
   (the sampler replaces the binomial draw of rlgc.py:564, 619-622). */
int m3d_binom_tables_host(const uint64_t **thresholds, int64_t *n_thresholds, const uint32_t **row_off_len,
                          int32_t *q_max);

/* Binomial(64 q[i], 1/2) by table inversion of the 64-bit uniforms (hi[i]:lo[i]) on the device, 2 <= q[i] <= q_max:
   lets the parity tests place draws on and next to every threshold. */
int m3d_binom_debug_table_draw(const uint32_t *q, const uint32_t *hi, const uint32_t *lo, int64_t n, uint32_t *out,
                               void *stream);

/* Parity hook for the packed complex arithmetic the FFT passes are built from (csrc/m3d_fft.h: products with +-i,
   complex products, Hermitian split / merge terms, 3- / 4- / 8- / 9-point butterflies): a, b, c are n complex64
   values each, out receives 58 complex64 values per element (layout: k_debug_pk in csrc/m3d_fftconv.hip).  The
   butterflies stand in for numpy.fft / scipy.fft inside rlgc.py:424-436's fft convolutions. */
int m3d_fft_debug_pk(const float *a, const float *b, const float *c, int64_t n, float *out, void *stream);

/* clip(0, 65535).astype(uint16) of RLGC output, DataRegistration.py:435, 993-995 */
int m3d_clip_to_u16(const float *in, uint16_t *out, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------
 * Decode front end  (PixelDecoder.py)
 * ---------------------------------------------------------------------------------- */

/* float32(readout) * float32(predictor), PixelDecoder.py:1879-1881 / 1060-1062 */
int m3d_weight_readout(const uint16_t *readout, const float *predictor, float *out, int64_t n,
                       void *stream);

/* cupyx gaussian_filter(mode="reflect", truncate=4) per bit, PixelDecoder.py:1948-1980.
 * is_3d != 0: separable Z,Y,X passes with sigma_zyx; is_3d == 0: each plane with
 * (sigma_y, sigma_x).  `scratch` is one more f32[Z,Y,X]. in may equal out. */
int m3d_gaussian_lowpass(const float *in, float *out, float *scratch, const int32_t shape_zyx[3],
                         const float sigma_zyx[3], int is_3d, void *stream);

/* _decode_pixels, PixelDecoder.py:2523-2643 (+ _scale/_clip/_normalize_pixel_traces
 * :2354-2468, _calculate_distances :2470-2521, exclusions :863-877) fused in one pass:
 *  stack       f32 [B, N] (N = Z*Y*X voxels, bit-major planes)
 *  background, normalization : f32[B] or NULL for un-normalised decoding
 *  codebook    f32 [C, B] unit-norm codewords
 *  excluded    int32[n_excluded] codeword indices (may be NULL)
 * outputs: decoded int16[N]; magnitude, distance f16[N]; scaled f16[B, N];
 *          optional f32 magnitude/distance before quantisation (may be NULL).
 * flags: M3D_DECODE_PRENORMALIZED - rows of `stack` are already unit-norm traces; only the
 *        distance / argmin / call runs (_calculate_distances, PixelDecoder.py:2470-2521). */
#define M3D_DECODE_PRENORMALIZED 1
int m3d_decode_pixels(const float *stack, int32_t B, int64_t N, const float *background,
                      const float *normalization, const float *codebook, int32_t C,
                      float pixel_threshold, float magnitude_lo, float magnitude_hi,
                      const int32_t *excluded, int32_t n_excluded, int16_t *decoded,
                      uint16_t *magnitude_f16, uint16_t *distance_f16, uint16_t *scaled_f16,
                      float *magnitude_f32, float *distance_f32, int32_t flags, void *stream);

/* ------------------------------------------------------------------------------------
 * Transcript extraction  (PixelDecoder._extract_barcodes, PixelDecoder.py:2908-3201)
 * ---------------------------------------------------------------------------------- */

/* cucim.skimage.measure.label(decoded, background=-1, connectivity=c), PixelDecoder.py:2946-2972.
 * Components are regions of EQUAL codeword id.  per_plane != 0 labels every z plane on its
 * own (the 2-D decode mode).  Outputs:
 *   labels int32[N]: voxel index of the component's raster-first voxel (its root), -1 = background
 *   slot   int32[N]: at root voxels, a dense component number in [0, *n_components)
 *   n_components: device int32 scalar */
int m3d_ccl_label(const int16_t *decoded, const int32_t shape_zyx[3], int connectivity,
                  int per_plane, int32_t *labels, int32_t *slot, int32_t *n_components,
                  void *stream);

/* regionprops_table columns used at PixelDecoder.py:2991-3042 as exact per-component sums.
 *   intensity: f16 or f32 [B, N] (scaled traces, or raw low-passed data while optimising
 *              normalisation, :2935-2941); magnitude/distance: f16 [N]
 * outputs per component slot s:
 *   comp_root[s]  voxel index of the first voxel (label order = ascending root)
 *   comp_code[s]  codeword id of the component (label -> codeword map of :3066-3078)
 *   comp_min_distance[s]  intensity_min of the distance image (:2991-2995)
 *   comp_sums[s][11 + B] = area, sum z, y, x, sum zz, yy, xx, zy, zx, yx, sum magnitude,
 *                          sum intensity[0..B)   (doubles; exact for f16 inputs) */
int m3d_region_stats(const int16_t *decoded, const int32_t *labels, const int32_t *slot,
                     const int32_t shape_zyx[3], int32_t B, const void *intensity,
                     int intensity_dtype, const uint16_t *magnitude_f16,
                     const uint16_t *distance_f16, int32_t n_components, int32_t *comp_root,
                     int32_t *comp_code, float *comp_min_distance, double *comp_sums, void *stream);

/* ------------------------------------------------------------------------------------
 * Global normalisation statistics  (PixelDecoder._global_normalization_vectors, :981-1199)
 * ---------------------------------------------------------------------------------- */

/* One pass of an exact 8-bit radix select over float32 values: adds to hist256 (device
 * int64[256], caller zeroes it) the histogram of key byte `pass` (0 = most significant) of
 * all values whose more significant key bytes equal those of `prefix`.  Optional on-the-fly
 * transform (1: max(x - sub, 0), PixelDecoder.py:1144-1148) and filter (1: x < thr, :1115;
 * 2: x > thr, :1162).  Summing the histograms of several GPUs (RCCL all-reduce) gives the
 * pooled order statistic cp.percentile / cp.median compute at :1113, :1127, :1157, :1177. */
int m3d_radix_hist(const float *values, int64_t n, int transform, float sub, int filter, double thr,
                   int pass, uint32_t prefix, int64_t *hist256, void *stream);

/* image[image > threshold] = replacement, PixelDecoder.py:1072-1074 (hot pixels) */
int m3d_replace_above(float *values, int64_t n, float threshold, float replacement, void *stream);

/* Separable FIR filter with scipy boundary handling (wrap = 0: "reflect", 1: "grid-wrap"):
 * out[i] = sum_k w[k] * in[idx(i + k - radius + offset)] per axis; radius < 0 skips an axis.
 * Weights are host arrays of 2*radius+1 floats (radius <= 32).  Used for the cubic-spline
 * ndi.shift(mode="grid-wrap") inside phase_cross_correlation's disambiguation: prefilter and
 * 4-tap B-spline evaluation of a pure translation compose to one FIR per axis. */
int m3d_separable_filter(const float *in, float *out, float *scratch, const int32_t shape_zyx[3],
                         const float *weights_host[3], const int32_t radius[3],
                         const int32_t offset[3], int wrap, void *stream);

/* ------------------------------------------------------------------------------------
 * Phase correlation  (cucim phase_cross_correlation, utils/multiview_registration.py:265-301)
 * ---------------------------------------------------------------------------------- */
typedef struct m3d_fft_plan m3d_fft_plan;
/* complex64 in-place FFT of any 1..3-D shape (slowest axis first) through rocFFT */
int m3d_fft_plan_create(int ndim, const int32_t *shape, m3d_fft_plan **plan);
size_t m3d_fft_plan_workspace_bytes(const m3d_fft_plan *plan);
int m3d_fft_exec(m3d_fft_plan *plan, void *data_c64, int inverse, void *work, size_t work_bytes,
                 void *stream); /* unnormalised */
int m3d_fft_plan_destroy(m3d_fft_plan *plan);
/* float32(real) -> complex64 */
int m3d_real_to_complex(const void *in, int dtype, int64_t n, void *out_c64, void *stream);
/* image_product = src * conj(target) [/ max(|.|, 100 eps) when normalize] */
int m3d_cross_power(const void *src_freq, const void *target_freq, int64_t n, int normalize,
                    void *out, void *stream);
/* np.argmax(np.abs(x)) (first maximum); scratch >= 32 KiB; outputs are device scalars */
int m3d_argmax_abs(const void *data_c64, int64_t n, void *scratch, int64_t *index_out,
                   float *value_out, void *stream);
/* out[a, r] = sum_x kernel[a, x] * (conj?) data[r, x]  - one tensordot of _upsampled_dft */
int m3d_contract_last(const void *data_c64, int64_t rows, int32_t X, const void *kernel_c64,
                      int32_t Ka, int conj_data, void *out_c64, void *stream);
/* n, sum a, sum b, sum aa, sum bb, sum ab over the box [lo, hi) - np.corrcoef of one
 * overlap tile in _disambiguate_shift */
int m3d_box_corr_sums(const float *a, const float *b, const int32_t shape_zyx[3],
                      const int32_t lo_zyx[3], const int32_t hi_zyx[3], double *sums6, void *stream);

/* ------------------------------------------------------------------------------------
 * Resampling  (utils/multiview_registration.py)
 * ---------------------------------------------------------------------------------- */
#define M3D_MODE_CONSTANT 0 /* scipy mode="constant": cval outside [0, n-1]          */
#define M3D_MODE_NEAREST 1  /* scipy mode="nearest": clamp to the edge sample         */

/* _max_z_projection_gpu, multiview_registration.py:57-77: f32[Y,X] = max over Z */
int m3d_max_z(const void *in, int dtype, const int32_t shape_zyx[3], float *out, void *stream);

/* warp_array_to_reference_gpu, multiview_registration.py:797-902:
 * cupyx.scipy.ndimage.affine_transform(order=1) with matrix_px / offset_px as built at
 * :867-868 (host doubles).  Output dtype = input dtype (u16 | f32). */
int m3d_affine_warp(const void *in, int dtype, const int32_t in_shape_zyx[3],
                    const double matrix_px_host[9], const double offset_px_host[3], void *out,
                    const int32_t out_shape_zyx[3], int mode, double cval, void *stream);

/* warp_array_to_reference_with_affine_and_sofima_flow_gpu, multiview_registration.py:905-1131.
 *  flow_xyz f32 [3, fz, fy, fx] (channels X,Y,Z); stride / box start / spacing / origin in
 *  Z,Y,X order; transform = 4x4 physical matrix, row major (host floats). */
int m3d_flow_warp(const void *in, int dtype, const int32_t in_shape_zyx[3], void *out,
                  const int32_t out_shape_zyx[3], const float *flow_xyz,
                  const int32_t flow_shape_zyx[3], const float stride_zyx_host[3],
                  const float box_start_zyx_host[3], const float spacing_zyx_host[3],
                  const float origin_zyx_host[3], const float transform_zyx_um_host[16], int mode,
                  float cval, void *stream);

/* ------------------------------------------------------------------------------------
 * Residual flow estimation  (utils/sofima_registration.py; the arithmetic the reference
 * delegates to the `sofima` package: flow_field.JAXMaskedXCorrWithStatsCalculator.flow_field
 * at :676-688, mesh.relax_mesh + mesh.elastic_mesh_3d at :488-509)
 * ---------------------------------------------------------------------------------- */
/* like m3d_fft_plan_create with `batch` contiguous transforms per execution */
int m3d_fft_plan_create_batched(int ndim, const int32_t *shape, int32_t batch, m3d_fft_plan **plan);

/* Flow-grid nodes [node_begin, node_begin + node_count) (raster order over grid_zyx): gather the
 * pre/post patches starting at node * step, subtract each patch mean, write pre + i*post
 * zero-padded into complex64 [node_count, fft_z, fft_y, fft_x] (fft >= 2 * patch - 1). */
int m3d_patch_pack(const float *pre, const float *post, const int32_t shape_zyx[3],
                   const int32_t patch_zyx[3], const int32_t step_zyx[3], const int32_t grid_zyx[3],
                   int64_t node_begin, int32_t node_count, const int32_t fft_zyx[3], void *out_c64,
                   void *stream);
/* in place on the forward transforms of m3d_patch_pack's output: spectrum of the linear
 * cross-correlation c[d] = sum_u post(u) pre(u + d), scaled by 1/N for the unnormalised inverse */
int m3d_patch_cross_spectrum(void *buf_c64, int32_t count, const int32_t fft_zyx[3], void *stream);
/* per node: integer peak of the lag volume d in [-(p-1), p-1]^3 (first maximum), sharpness =
 * peak / min within +-peak_radius, ratio = peak / best other local maximum (no larger value
 * within +-min_distance, > threshold_rel * peak, outside the radius; 0 when none).
 * flow_out[c * channel_stride + node_begin + i], c = x, y, z, sharpness, ratio. */
int m3d_xcorr_peaks(const void *buf_c64, int32_t count, const int32_t fft_zyx[3],
                    const int32_t patch_zyx[3], int32_t min_distance, int32_t peak_radius,
                    float threshold_rel, float *flow_out, int64_t channel_stride, int64_t node_begin,
                    void *stream);
/* m3d_patch_pack + forward transform + m3d_patch_cross_spectrum + inverse transform + m3d_xcorr_peaks in one call for
 * patches whose transform box is 24 x 64 x 64 (m3d_patch_xcorr_fused_supported): the transforms run in LDS / registers
 * instead of through rocFFT (same arithmetic up to the rounding of the transforms; replaces the same reference call,
 * sofima.flow_field.JAXMaskedXCorrWithStatsCalculator.flow_field, utils/sofima_registration.py:670-688).
 * `work`: m3d_patch_xcorr_fused_workspace_bytes(patch, count) bytes of device memory. */
int m3d_patch_xcorr_fused_supported(const int32_t patch_zyx[3], const int32_t fft_zyx[3]);
size_t m3d_patch_xcorr_fused_workspace_bytes(const int32_t patch_zyx[3], int32_t count);
int m3d_patch_xcorr_fused(const float *pre, const float *post, const int32_t shape_zyx[3],
                          const int32_t patch_zyx[3], const int32_t step_zyx[3], const int32_t grid_zyx[3],
                          int64_t node_begin, int32_t count, int32_t min_distance, int32_t peak_radius,
                          float threshold_rel, void *work, size_t work_bytes, float *flow_out,
                          int64_t channel_stride, void *stream);
/* _refine_flow_vectors_subpixel, sofima_registration.py:243-417: for n (node zyx, vector xyz)
 * pairs pick the offset triple (from n_offsets <= 5 host values per axis) minimising the mean
 * squared difference of the normalised post patch and the order-1 sampled pre patch. */
int m3d_subpixel_refine(const float *pre, const float *post, const int32_t shape_zyx[3],
                        const int32_t patch_zyx[3], const int32_t step_zyx[3], const int32_t *nodes_zyx,
                        const float *vectors_xyz, int32_t n, const float *offsets_host, int32_t n_offsets,
                        float epsilon, float *out_xyz, void *stream);

typedef struct m3d_mesh_params {
    float dt, k0, k, stop_v_max, dt_max, start_cap, final_cap, cap_upscale;
    int32_t num_iters, max_iters;
} m3d_mesh_params;
typedef struct m3d_mesh_result {
    int32_t iterations, blocks;
    double final_kinetic_energy;
    float final_v_max, final_cap, final_dt, reserved;
} m3d_mesh_result;
size_t m3d_mesh_relax_workspace_bytes(const int32_t grid_zyx[3]);
/* _relax_flow_field, sofima_registration.py:455-521: x [3 (xyz), gz, gy, gx] relaxed in place;
 * prev = measured vectors (NaN = no measurement); velocity-Verlet + FIRE in blocks of num_iters */
int m3d_mesh_relax(float *x_xyz, const float *prev_xyz, const int32_t grid_zyx[3],
                   const float stride_xyz[3], const m3d_mesh_params *params, void *work,
                   size_t work_bytes, m3d_mesh_result *result, void *stream);
/* _compose_flow_fields_same_grid, sofima_registration.py:178-240 */
int m3d_flow_compose(const float *base, const float *residual, const int32_t grid_zyx[3],
                     const float stride_zyx[3], const float box_start_xyz[3], float *out, void *stream);
/* _normalized_mean_squared_error, sofima_registration.py:149-175: out2 = {sum, sum of squares};
 * out1 = sum (((a - mean_a) / scale_a) - ((b - mean_b) / scale_b))^2 */
int m3d_moments(const float *src, int64_t n, double *out2, void *stream);
int m3d_normalized_sq_diff(const float *a, const float *b, int64_t n, float mean_a, float scale_a,
                           float mean_b, float scale_b, double *out1, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* M3D_HIP_H */
