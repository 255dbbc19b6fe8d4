/*
 * oracle/flate_ref.h -- CPU restatement of the reference's flate / gzip
 * encoders (flate/huffman_bit_writer.go, flate/huffman_code.go, flate/token.go,
 * flate/gzip.go) and of flate.NewWriter's level -> match finder map for the
 * levels that use matchfinder.M4 (flate/writer.go:33-46).
 *
 * TEST INFRASTRUCTURE ONLY (see m4_ref.h): nothing under brotli_amd/ includes,
 * links or loads this.  It checks the GPU deflate encoder (m4g_batch_flate_*)
 * byte for byte and lets the reference's own test criterion run here --
 * flate.NewReader(compressed) == data (flate/flate_test.go:14-38, 44-90) --
 * with zlib's inflate standing in for Go's compress/flate reader.
 *
 * Unlike brotli's chooseBitDepths, every sort in this encoder has a total
 * order (byFreq: frequency then literal, huffman_code.go:335-340), so the
 * restatement is meant to be byte-identical to the Go encoder.  That claim is
 * unpinned (no Go toolchain here); what IS pinned: inflate round trips, Kraft
 * equality and cost optimality of the length-limited codes against an
 * independent package-merge, and the RFC 1951 fixed codes.
 */
#ifndef FLATE_REF_H
#define FLATE_REF_H

#include <stddef.h>
#include <stdint.h>

#include "m4_ref.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FLATEREF_NUM_LIT 286 /* maxNumLit, huffman_bit_writer.go:25 */
#define FLATEREF_NUM_OFF 30  /* offsetCodeCount, huffman_bit_writer.go:13 */

/* flate/token.go:59-80; -1 where the reference panics ("match too long" /
 * "match distance too high") or indexes below the tables (len < 3, off < 1) */
int flateref_length_code(uint32_t len);
int flateref_offset_code(uint32_t off);

/* huffmanEncoder.generate, huffman_code.go:271-309: code (bit-reversed, as
 * written LSB first) and length per symbol; length 0 = unused symbol. */
void flateref_generate(const int32_t *freq, int n, int max_bits, uint16_t *code, uint16_t *len);

/* makeStatistics, huffman_bit_writer.go:425-468, frequencies only (before
 * the "at least one offset" fix-up of lines 459-464).  Returns 0, or -1 when a
 * match breaks the token limits (the reference panics there). */
int flateref_statistics(const uint8_t *src, size_t n, const m4ref_match *matches, size_t nmatches,
                        int32_t *lit_freq /*[286]*/, int32_t *off_freq /*[30]*/);

typedef struct flateref_encoder flateref_encoder;
/* NewEncoder() (huffman_bit_writer.go:80-89) or, with gzip != 0,
 * NewGZIPEncoder() (gzip.go:10-14) whose header carries mtime instead of
 * time.Now() (gzip.go:46). */
flateref_encoder *flateref_encoder_new(int gzip, uint32_t mtime);
void flateref_encoder_free(flateref_encoder *e);
void flateref_encoder_reset(flateref_encoder *e);
/* Encode (huffman_bit_writer.go:506-517 / gzip.go:39-65).  src_is_nil: the Go
 * caller passed a nil slice (storedSize's `in == nil`, line 270).  Returns 0,
 * or -1 where the reference panics.  kind_out (nullable): 0 stored, 1 fixed,
 * 2 dynamic. */
int flateref_encode(flateref_encoder *e, const uint8_t *src, size_t n, int src_is_nil,
                    const m4ref_match *matches, size_t nmatches, int last_block, int *kind_out);
const uint8_t *flateref_encoder_take(flateref_encoder *e, size_t *size);

/* matchfinder.Writer{MatchFinder: mf, Encoder: NewEncoder() / NewGZIPEncoder(),
 * BlockSize: block_size}: Write(data); Close() (matchfinder.go:59-93).
 * malloc'd result (free with flateref_free_buf), NULL where the reference
 * panics. */
uint8_t *flateref_compress_m4(m4ref *mf, const uint8_t *data, size_t n, size_t block_size,
                              int gzip, uint32_t mtime, size_t *out_size);
void flateref_free_buf(uint8_t *p);

/* hash/crc32 IEEE, as gzip.go:58 uses it */
uint32_t flateref_crc32(uint32_t crc, const uint8_t *p, size_t n);

#ifdef __cplusplus
}
#endif
#endif
