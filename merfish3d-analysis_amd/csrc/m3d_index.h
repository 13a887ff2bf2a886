// Boundary index maps shared by kernels and the host-side unit test (tests/host_emul).
#pragma once

#if defined(__HIPCC__)
#define M3D_IDX_HD __host__ __device__ __forceinline__
#else
#define M3D_IDX_HD static inline
#endif

namespace m3d {

// Source index (padded coordinates) of the symmetric extension of the observed range
// [b, b+L) for any padded index i - identity inside the range.  Equals
// np.pad(arange(b, b+L), (b, after), mode="symmetric")[i]   (utils/rlgc.py:203-227).
M3D_IDX_HD int sym_src(int i, int b, int L)
{
    if (i >= b && i < b + L) return i;
    const int P = 2 * L;
    if (i < b) {
        int m = (b - 1 - i) % P;
        return b + (m < L ? m : P - 1 - m);
    }
    int m = (i - (b + L)) % P;
    return b + (m < L ? L - 1 - m : m - L);
}

// scipy.ndimage mode="reflect" (half-sample symmetric, d c b a | a b c d | d c b a).
M3D_IDX_HD int reflect_idx(int i, int n)
{
    if (i >= 0 && i < n) return i;
    const int P = 2 * n;
    int m = i % P;
    if (m < 0) m += P;
    return m < n ? m : P - 1 - m;
}

// scipy.ndimage mode="nearest": clamp to the edge sample.
M3D_IDX_HD int clamp_idx(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

}  // namespace m3d
