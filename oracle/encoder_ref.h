/*
 * oracle/encoder_ref.h -- CPU restatement of brotli.Encoder (encoder.go) and
 * of matchfinder.Writer (matchfinder/matchfinder.go:44-103).
 *
 * TEST INFRASTRUCTURE ONLY (see m4_ref.h).  Used to (1) check the GPU
 * histogram kernels against pass 1 of Encoder.Encode (encoder.go:33-105) and
 * (2) turn Match lists into a real brotli stream so the reference's own test
 * criterion -- decode(encode(x)) == x, brotli_test.go:626-649 -- can be run
 * here with the system brotli decoder standing in for brotli.NewReader.
 *
 * The stream is a VALID brotli stream but is not claimed byte-identical to
 * the Go encoder's: chooseBitDepths (brotli_bit_stream.go:558-635) sorts with
 * slices.SortFunc, whose order among equal counts is Go-pdqsort specific; this
 * restatement breaks ties by symbol index.  Histograms and distance codes are
 * tie-free and are checked bit-exactly.
 */
#ifndef ENCODER_REF_H
#define ENCODER_REF_H

#include <stddef.h>
#include <stdint.h>

#include "m4_ref.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint32_t lit[256];
  uint32_t cmd[704];
  uint32_t dist[64];
} encref_histo;

/* encoder.go:163-167 */
typedef struct {
  uint32_t code;
  uint32_t nExtra;
  uint32_t extraBits;
} encref_distcode;

/* Pass 1 of Encoder.Encode, encoder.go:44-105.  dist_cache (nullable) gets
 * one entry per match (entries for matches that take no distance code are
 * zeroed); cmd_out (nullable) gets the command symbol per match. */
void encref_histograms(const uint8_t *src, size_t n, const m4ref_match *matches,
                       size_t nmatches, encref_histo *h,
                       encref_distcode *dist_cache, uint16_t *cmd_out);

/* code helpers (command.go:111-164, encoder.go:169-177) exposed for tests */
uint16_t encref_insert_length_code(uint32_t insertlen);
uint16_t encref_copy_length_code(uint32_t copylen);
uint16_t encref_combine_length_codes(uint16_t inscode, uint16_t copycode,
                                     int use_last_distance);
encref_distcode encref_distance_code(uint32_t distance);

/* test hook, see encoder_ref.c */
uint64_t encref_static_code_length_header(unsigned *nbits);

typedef struct encref_encoder encref_encoder;
encref_encoder *encref_encoder_new(void);
void encref_encoder_free(encref_encoder *e);
/* encoder.go:12-15 */
void encref_encoder_reset(encref_encoder *e);
/* encoder.go:17-161.  Output bytes of this call are appended to an internal
 * buffer; fetch with encref_encoder_take. */
void encref_encode(encref_encoder *e, const uint8_t *src, size_t n,
                   const m4ref_match *matches, size_t nmatches, int last_block);
/* Returns pointer/size of everything produced since the last take. */
const uint8_t *encref_encoder_take(encref_encoder *e, size_t *size);

/* matchfinder.Writer{MatchFinder: M4{...}, Encoder: &brotli.Encoder{},
 * BlockSize: block_size} over one whole input: Write(data); Close().
 * (matchfinder.go:59-93).  Returns malloc'd stream, caller frees with
 * encref_free_buf.  Also reports total match count if nmatches_out != NULL. */
uint8_t *encref_compress_m4(m4ref *mf, const uint8_t *data, size_t n,
                            size_t block_size, size_t *out_size,
                            size_t *nmatches_out);
void encref_free_buf(uint8_t *p);

#ifdef __cplusplus
}
#endif
#endif
