/*
 * stem_rans.h -- C ABI of libstem_rans.so: the HOST entropy coder of the STEM path
 * (north_star: "the rANS arithmetic coder stays on host").
 *
 * Replaces the reference's pybind11 modules
 *   compressai.ans   (compressai/cpp_exts/rans/rans_interface.cpp:99-372) and
 *   compressai._CXX  (compressai/cpp_exts/ops/ops.cpp:24-90)
 * over third_party/ryg_rans/rans64.h:59-141 (64-bit state, 32-bit renormalisation,
 * 16-bit probabilities, 4-bit bypass escapes).  Streams are byte-identical to the
 * reference's for the same symbols / indexes / tables.
 *
 * Differences in the boundary (not in the bytes): tables are one dense
 * int32 [ncdf][cdf_stride] array instead of vector<vector<int>> (the reference
 * re-converts the 64x3133 table from Python lists on EVERY call,
 * spatiotemporalpriors.py:1046-1048); inputs are validated and errors are returned
 * (the reference only asserts, i.e. UB in release builds); the encoder never
 * overruns its buffer for tiny inputs (rans_interface.cpp:170 does for n < 3).
 * All pointers are host pointers; handles are not thread-safe, distinct handles are.
 */
#ifndef STEM_RANS_H
#define STEM_RANS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char *stem_rans_last_error(void);

/* RansEncoder.encode_with_indexes (rans_interface.cpp:193-204).  Returns the number of bytes
 * written to out, -1 on invalid input, -2 when cap is too small (8*n + 16 always suffices...
 * use stem_rans_encoder_pending_bytes for exact sizing with the handle API).                 */
long stem_rans_encode(const int32_t *symbols, const int32_t *indexes, size_t n,
                      const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets,
                      uint8_t *out, size_t cap);
/* RansDecoder.decode_with_indexes (rans_interface.cpp:206-275) */
int stem_rans_decode(const uint8_t *stream, size_t nbytes, const int32_t *indexes, size_t n,
                     const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets,
                     int32_t *out);

/* BufferedRansEncoder (rans_interface.cpp:99-191): push any number of symbol runs, then flush once. */
void *stem_rans_encoder_create(void);
void stem_rans_encoder_destroy(void *enc);
int stem_rans_encoder_push(void *enc, const int32_t *symbols, const int32_t *indexes, size_t n,
                           const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets);
size_t stem_rans_encoder_pending_bytes(void *enc);   /* upper bound of the flushed size */
long stem_rans_encoder_flush(void *enc, uint8_t *out, size_t cap);

/* RansDecoder.set_stream / decode_stream (rans_interface.cpp:277-350): incremental decoding. */
void *stem_rans_decoder_create(void);
void stem_rans_decoder_destroy(void *dec);
int stem_rans_decoder_set_stream(void *dec, const uint8_t *stream, size_t nbytes);
int stem_rans_decoder_decode(void *dec, const int32_t *indexes, size_t n,
                             const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets,
                             int32_t *out);

/* compressai._CXX.pmf_to_quantized_cdf (ops.cpp:24-81); cdf holds n+1 entries. */
int stem_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf);

#ifdef __cplusplus
}
#endif
#endif
