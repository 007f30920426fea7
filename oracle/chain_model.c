/*
 * oracle/chain_model.c — CPU model of the ARITHMETIC ORDER of the HIP signature kernel.
 *
 * TEST INFRASTRUCTURE ONLY (built by oracle/build.py into oracle/_build/, loaded by tests/,
 * __graft_entry__.smoke() and bench.py's checker; never by lshrs_amd/).
 *
 * What it restates.  The reference computes, per band, `projection @ vector`
 * (lshrs/hash/lsh.py:200), thresholds `> 0` (:204) and packs LSB-first (:208).  The
 * HIP kernel evaluates the same dot products on v_mfma_f32_32x32x2_f32, which the
 * gfx950 guide documents as bit-for-bit a k-ordered fmaf chain
 *      D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)).
 * The kernel feeds k in tiles of 32: MFMA step s (0..15) of tile t multiplies
 * k0 = 32t + s (lanes 0..31) and k1 = 32t + 16 + s (lanes 32..63).  So every
 * projection is this exact sequence of single-rounded fused multiply-adds:
 *
 *      acc = +0
 *      for t in tiles:  for s in 0..15:
 *          acc = fmaf(x[32t+s],    p[32t+s],    acc)
 *          acc = fmaf(x[32t+16+s], p[32t+16+s], acc)      (k >= dim contributes x=p=0)
 *
 * This file evaluates that sequence with libm's fmaf, giving the raw (pre tie-break)
 * kernel bits on any CPU, independent of BLAS.  It is the second, BLAS-free anchor of
 * the parity suite: raw kernel output == this model, bit for bit; final output ==
 * the NumPy literal restatement (oracle/lshrs_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define LSHRS_KTILE 32

static inline float chain_dot(const float *x, const float *p, int dim)
{
    float acc = 0.0f;
    const int tiles = (dim + LSHRS_KTILE - 1) / LSHRS_KTILE;
    for (int t = 0; t < tiles; ++t) {
        const int base = t * LSHRS_KTILE;
        for (int s = 0; s < LSHRS_KTILE / 2; ++s) {
            const int k0 = base + s;
            const int k1 = base + LSHRS_KTILE / 2 + s;
            const float x0 = k0 < dim ? x[k0] : 0.0f;
            const float p0 = k0 < dim ? p[k0] : 0.0f;
            const float x1 = k1 < dim ? x[k1] : 0.0f;
            const float p1 = k1 < dim ? p[k1] : 0.0f;
            acc = fmaf(x0, p0, acc);
            acc = fmaf(x1, p1, acc);
        }
    }
    return acc;
}

/* Y[n][num_perm] = chain-ordered projections of X[n][dim] on P[num_perm][dim]. */
void lshrs_chain_project(const float *X, int64_t n, int dim, const float *P, int num_perm, float *Y)
{
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < num_perm; ++j)
            Y[i * (int64_t)num_perm + j] = chain_dot(X + i * (int64_t)dim, P + (int64_t)j * dim, dim);
}

/*
 * keys[n][num_bands][ceil(rows/8)]: bit i of band b = (chain_dot(P[b*rows+i], X[row]) > 0),
 * LSB-first inside each byte, tail bits zero (np.packbits(bitorder="little") layout,
 * lshrs/hash/lsh.py:208).
 */
void lshrs_chain_hash(const float *X, int64_t n, int dim, const float *P, int num_bands, int rows_per_band,
                      uint8_t *keys)
{
    const int bb = (rows_per_band + 7) / 8;
    memset(keys, 0, (size_t)n * num_bands * bb);
    for (int64_t i = 0; i < n; ++i) {
        const float *x = X + i * (int64_t)dim;
        for (int b = 0; b < num_bands; ++b) {
            uint8_t *dst = keys + (i * num_bands + b) * (int64_t)bb;
            for (int r = 0; r < rows_per_band; ++r) {
                const float y = chain_dot(x, P + ((int64_t)b * rows_per_band + r) * dim, dim);
                if (y > 0.0f)
                    dst[r >> 3] |= (uint8_t)(1u << (r & 7));
            }
        }
    }
}

/* Fused-form cosine used by the rerank kernel, evaluated in double as an error yardstick:
 * score = dot(c, q) / (||c|| * ||q||).  (Reference: lshrs/utils/similarity.py:80-90 normalises
 * first; the two forms agree to ~5e-8, tolerance is 1e-5.) */
void lshrs_cosine_f64(const float *q, const float *cands, int64_t c, int dim, double *scores)
{
    double qq = 0.0;
    for (int k = 0; k < dim; ++k)
        qq += (double)q[k] * q[k];
    for (int64_t i = 0; i < c; ++i) {
        const float *v = cands + i * (int64_t)dim;
        double d = 0.0, vv = 0.0;
        for (int k = 0; k < dim; ++k) {
            d += (double)v[k] * q[k];
            vv += (double)v[k] * v[k];
        }
        scores[i] = d / (sqrt(vv) * sqrt(qq));
    }
}
