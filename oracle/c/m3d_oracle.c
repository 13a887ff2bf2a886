/* Plain-C pieces of the CPU oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).
 *
 * Holds the loops that NumPy cannot express exactly or quickly:
 *   - the pixel -> codeword distance / argmin with a k-ordered float32 fmaf chain
 *     (the published numerics of v_mfma_f32_32x32x2_f32; restates the expanded
 *     euclidean form cuVS uses at PixelDecoder.py:2505-2513),
 *   - equal-value connected components (scikit-image `label` semantics used at
 *     PixelDecoder.py:2946-2972),
 *   - per-label statistics (regionprops_table columns used at PixelDecoder.py:2991-3042).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/build.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* d[p] = min_c sqrt(max(0, |x_p|^2 + |c|^2 - 2 x_p.c)), first minimum wins. */
void m3d_oracle_distance_argmin(const float *x, const float *cb, int64_t P, int B, int C,
                                float *dmin, int32_t *imin)
{
    float *yn = (float *)malloc(sizeof(float) * (size_t)C);
    for (int c = 0; c < C; ++c) {
        float acc = 0.0f;
        for (int b = 0; b < B; ++b)
            acc = fmaf(cb[(size_t)c * B + b], cb[(size_t)c * B + b], acc);
        yn[c] = acc;
    }
    for (int64_t p = 0; p < P; ++p) {
        const float *xp = x + (size_t)p * B;
        float xn = 0.0f;
        for (int b = 0; b < B; ++b)
            xn = fmaf(xp[b], xp[b], xn);
        float best = INFINITY;
        int32_t besti = 0;
        for (int c = 0; c < C; ++c) {
            float dot = 0.0f;
            for (int b = 0; b < B; ++b)
                dot = fmaf(xp[b], cb[(size_t)c * B + b], dot);
            float t = xn + yn[c];
            float d2 = t - 2.0f * dot;
            float d = sqrtf(fmaxf(d2, 0.0f));
            if (d < best) {
                best = d;
                besti = c;
            }
        }
        dmin[p] = best;
        imin[p] = besti;
    }
    free(yn);
}

/* ---- union-find connected components of equal-valued voxels ----------------------
 * img: int16[Z,Y,X]; background value excluded; connectivity 1..3 (faces / +edges /
 * +corners; for Z == 1 this is the 2-D 4-/8-neighbourhood).  Labels are numbered
 * 1..n in raster order of each component's first voxel (scikit-image convention).
 * Returns n. */
static int64_t uf_find(int64_t *parent, int64_t i)
{
    while (parent[i] != i) {
        parent[i] = parent[parent[i]];
        i = parent[i];
    }
    return i;
}

int64_t m3d_oracle_label_equal(const int16_t *img, int64_t Z, int64_t Y, int64_t X,
                               int16_t background, int connectivity, int32_t *labels)
{
    int64_t n = Z * Y * X;
    int64_t *parent = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i)
        parent[i] = i;
    for (int64_t z = 0; z < Z; ++z)
        for (int64_t y = 0; y < Y; ++y)
            for (int64_t x = 0; x < X; ++x) {
                int64_t i = (z * Y + y) * X + x;
                int16_t v = img[i];
                if (v == background)
                    continue;
                /* visit the 13 "earlier" neighbours */
                for (int dz = -1; dz <= 0; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (dz == 0 && (dy > 0 || (dy == 0 && dx >= 0)))
                                continue;
                            int order = (dz != 0) + (dy != 0) + (dx != 0);
                            if (order > connectivity)
                                continue;
                            int64_t zz = z + dz, yy = y + dy, xx = x + dx;
                            if (zz < 0 || yy < 0 || yy >= Y || xx < 0 || xx >= X)
                                continue;
                            int64_t j = (zz * Y + yy) * X + xx;
                            if (img[j] != v)
                                continue;
                            int64_t ri = uf_find(parent, i), rj = uf_find(parent, j);
                            if (ri != rj) {
                                if (ri < rj)
                                    parent[rj] = ri;
                                else
                                    parent[ri] = rj;
                            }
                        }
            }
    int64_t next = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (img[i] == background) {
            labels[i] = 0;
            continue;
        }
        int64_t r = uf_find(parent, i);
        if (r == i)
            labels[i] = (int32_t)(++next); /* root is the raster-first voxel */
        else
            labels[i] = labels[r];
    }
    free(parent);
    return next;
}
