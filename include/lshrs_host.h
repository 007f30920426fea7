/* lshrs_host.h - C ABI of the host tie-break engine (liblshrs_host.so, plain C++/pthreads, no HIP).
 *
 * The GPU signature pass (lshrs_hip.h) returns, next to the keys, the list of projections whose sign cannot be
 * trusted against the reference's summation order.  The bit-exact contract is restored by evaluating those
 * (row, band) pairs with the very expression the reference uses,
 *
 *     projections = projection_matrix @ vector          lshrs/hash/lsh.py:200
 *     bits        = projections > 0                     lshrs/hash/lsh.py:204
 *     packed      = np.packbits(bits, bitorder="little")  lshrs/hash/lsh.py:208
 *
 * i.e. one cblas_sgemv(RowMajor, NoTrans, rows_per_band, dim) of the BLAS NumPy is linked against, per pair.
 * This engine issues exactly that call, from several threads, each through a private mapping of the same
 * library file (OpenBLAS serialises callers that share one mapping).  It is a host-side accelerator of the
 * reference's own arithmetic, not a re-implementation of it: there is no summation code in it.
 */
#ifndef LSHRS_HOST_H
#define LSHRS_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSHRS_HOST_ABI_VERSION 2
#define LSHRS_HOST_E_BADARG (-1)

int lshrs_host_abi_version(void);

/* Create an engine with up to n_threads workers (1..64), each bound to its own copy of the shared library at
 * blas_path.  sgemv_symbol names that library's cblas_sgemv ("scipy_cblas_sgemv64_" for NumPy 2's bundled
 * OpenBLAS, "cblas_sgemv" for a system BLAS); ilp64 says whether its integer arguments are 64-bit;
 * set_threads_symbol (may be NULL/"") names its openblas_set_num_threads, called with 1 on every copy.
 * Returns NULL when not even one copy can be mapped; fewer copies than asked for is not an error
 * (lshrs_tb_threads tells). */
void* lshrs_tb_create(const char* blas_path, const char* sgemv_symbol, const char* set_threads_symbol, int ilp64,
                      int n_threads);
int lshrs_tb_threads(void* engine);
void lshrs_tb_destroy(void* engine);

/* One dot product in the summation order of the host BLAS that the GPU's tie replay follows (lshrs_hip.h:
 * lshrs_sig_hash_batch_split_replay_f32), for row `row` of a band of `rows_per_band` hyperplanes - what
 * `projection @ vector` (lshrs/hash/lsh.py:200) computes for that row.  model 1 = OpenBLAS's sgemv_t on x86-64: rows in
 * groups of four through the 8-lane kernel (eight interleaved fma chains over k = j (mod 8), reduced as
 * ((p0+p4) + (p1+p5)) + ((p2+p6) + (p3+p7))); of the rows_per_band % 4 rows left over, a pair through the 4x2 kernel (four
 * chains over k = l (mod 4), multiply and add in two roundings, reduced (v0+v1) + (v2+v3)) and a single one through the 4x1
 * kernel (the eight chains, unfused); the vector consumed in blocks of 4096, each block's sum added to y; of 8 m + 4
 * elements the 8-lane kernels take the first four with their low lanes (every block of 4096 its own first four), then eight at
 * a time; n % 4 elements of scalar tail behind all blocks (model 1: as the SkylakeX build contracts it, from 9 elements; model 2:
 * as the Haswell / Zen build leaves it, every length from 1).  A band of ONE row: sdot (both builds' SIMD kernels, the elements
 * behind the last whole 32 in a double).  model 3 = the SkylakeX build's small-matrix kernels: bands of two rows and more over
 * at most 8 elements (rows in blocks of 16 / 8 / 4 / 2 / 1, each with its own tree per length).  NaN on bad arguments.  Used to check, bit for bit against NumPy in the running process, that the replay may stand in for
 * the engine below.  lshrs_tb_model_dot: row 0 of a four-row band. */
float lshrs_tb_model_row_dot(const float* a, const float* x, int64_t n, int32_t model, int32_t row, int32_t rows_per_band);
float lshrs_tb_model_dot(const float* a, const float* x, int64_t n, int32_t model);

/* For pair p in [0, n_pairs): y = planes[band[p]] (rows_per_band x dim, row-major, contiguous) @ xrows[row_index[p]]
 * (row stride ldx floats); out_keys[p*band_bytes ..] = packbits(y > 0, little), band_bytes = ceil(rows_per_band/8).
 * out_y (may be NULL) receives the rows_per_band projections of every pair.  Blocks until done; one call at a
 * time per engine (serialised internally).  Returns 0 or LSHRS_HOST_E_BADARG.
 * Replaces the per-vector expression of lshrs/hash/lsh.py:200-208 for the pairs the GPU pass flagged. */
int lshrs_tb_patch(void* engine, const float* planes, int32_t num_bands, int32_t rows_per_band, int32_t dim,
                   const float* xrows, int64_t ldx, const int32_t* row_index, const int32_t* band, int64_t n_pairs,
                   uint8_t* out_keys, float* out_y);

/* One pipeline chunk in one call: decode the GPU pass's tie list, evaluate the pairs, lay the patches out for
 * lshrs_scatter_band_keys_u8 (lshrs_hip.h).
 *   entries  n_entries x 2 int64, as written by lshrs_sig_hash_batch_f32: (row * 65536 + index of a 32-column word
 *            of the padded key row, bit mask of the flagged columns in that word)
 *   xstage   one vector per ENTRY (row stride ldx floats), as staged by lshrs_gather_tied_rows_f32
 * Output: the unique (row, band) pairs sorted by (band, row) in out_rows / out_bands and their band keys in
 * out_keys (band_bytes each); *n_pairs = their number.  Returns LSHRS_HOST_E_BADARG (with *n_pairs set) when
 * out_cap pairs are not enough. */
int lshrs_tb_resolve(void* engine, const float* planes, int32_t num_bands, int32_t rows_per_band, int32_t dim,
                     const int64_t* entries, int64_t n_entries, const float* xstage, int64_t ldx, int64_t* out_rows,
                     int32_t* out_bands, uint8_t* out_keys, int64_t out_cap, int64_t* n_pairs);

#ifdef __cplusplus
}
#endif
#endif /* LSHRS_HOST_H */
