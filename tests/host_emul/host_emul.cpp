// Host build of the header-only index / RNG helpers the HIP kernels use, so their logic
// is unit-tested on the CPU box (tests/test_host_emul.py).  Test infrastructure only.
#include <cstdint>

#include "../../merfish3d-analysis_amd/csrc/m3d_index.h"
#include "../../merfish3d-analysis_amd/csrc/m3d_binom.h"
#include "../../merfish3d-analysis_amd/csrc/m3d_philox.h"

extern "C" {
void emul_binomial_half(const uint32_t *n, const uint64_t *voxel, int64_t count, uint32_t iteration,
                        uint64_t seed, uint32_t *out)
{
    const m3d::BinomTables t = m3d::binom_tables_host().view();
    for (int64_t i = 0; i < count; ++i) out[i] = n[i] < 65536u && (i & 1) ? m3d::binomial_half<false>(n[i], voxel[i], iteration, seed, t)
                                          : m3d::binomial_half<true>(n[i], voxel[i], iteration, seed, t);
}
void emul_table_draw(const uint32_t *q, const uint32_t *hi, const uint32_t *lo, int64_t count, uint32_t *out)
{
    const m3d::BinomTables t = m3d::binom_tables_host().view();
    for (int64_t i = 0; i < count; ++i) out[i] = m3d::binom_table_draw(q[i], hi[i], lo[i], t);
}
void emul_sym_src(int length, int b, int L, int *out)
{
    for (int i = 0; i < length; ++i) out[i] = m3d::sym_src(i, b, L);
}
void emul_reflect(int lo, int hi, int n, int *out)
{
    for (int i = lo; i < hi; ++i) out[i - lo] = m3d::reflect_idx(i, n);
}
}
