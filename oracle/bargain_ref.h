/*
 * oracle/bargain_ref.h -- CPU restatement of the reference's shortest-path
 * match finders matchfinder.Bargain1 / Bargain2 / Bargain3
 * (matchfinder/bargain1.go, bargain2.go, bargain3.go; `arrival` and
 * baseMatchCost: matchfinder/pathfinder.go:57-65; extendMatch2:
 * matchfinder/m4.go:326-339) -- what brotli.NewWriterV2 levels 5-9 plug into
 * matchfinder.Writer (writer.go:133-150).
 *
 * TEST INFRASTRUCTURE ONLY (see m4_ref.h): nothing under brotli_amd/ includes,
 * links or loads this.  It checks the GPU path (m4g_bargain_*) Match for Match.
 *
 * The costs are float32 sums, so the order of every addition is the
 * reference's (bargain1.go:96: startCost + baseMatchCost + insertCost +
 * distanceCost, left to right), and the per-byte literal costs come from a
 * restatement of Go's pure-Go math.Log2 / math.Log / math.Frexp (the amd64
 * build has no assembly for them), compiled without FMA contraction.
 *
 * PARITY PIN: the reference ships no golden Match lists, and no Go toolchain
 * exists here: Match-for-Match parity against the Go code is "parity unpinned".
 * What pins this restatement: hand-derived small cases, the properties the
 * algorithm guarantees (the Match list tiles the block; every match copies
 * equal bytes from within MaxDistance), and the reference's own criterion --
 * the stream the restated Encoder makes from these matches decodes to the
 * input (brotli_test.go:918-932 run `test` on Isaac.Newton-Opticks.txt).
 */
#ifndef BARGAIN_REF_H
#define BARGAIN_REF_H

#include <stddef.h>
#include <stdint.h>

#include "m4_ref.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bargainref bargainref;

/* variant 1, 2, 3 = Bargain1 / Bargain2 / Bargain3; max_distance 0 = the
 * default 1 << 16 (bargain1.go:39-41); skip = the Skip field. */
bargainref *bargainref_new(int variant, int64_t max_distance, int skip);
void bargainref_free(bargainref *z);
/* bargain1.go:33-36 */
void bargainref_reset(bargainref *z);
/* bargain1.go:38-208.  Appends to dst, returns 0. */
int bargainref_find_matches(bargainref *z, m4ref_matchvec *dst, const uint8_t *src, size_t n);
/* One FindMatches call with an empty dst (what matchfinder.Writer does,
 * matchfinder.go:81-87).  Returns the count, or -(needed) if out_cap is too
 * small. */
int64_t bargainref_find_matches_flat(bargainref *z, const uint8_t *src, size_t n, m4ref_match *out,
                                     size_t out_cap);
/* The Writer's loop: src cut into blocks of block_size, one call each.
 * block_off[k] = index of block k's first Match, block_off[nblocks] = total. */
int64_t bargainref_find_blocks(bargainref *z, const uint8_t *src, size_t n, size_t block_size,
                               m4ref_match *out, size_t out_cap, uint64_t *block_off);

/* float32(max(math.Log2(float64(total)/float64(count)), 1)), bargain1.go:47-51,
 * as raw bits (so that +Inf and NaN compare too). */
uint32_t bargainref_byte_cost_bits(uint64_t total, uint32_t count);
/* Go's math.Log2 */
double bargainref_log2(double x);

#ifdef __cplusplus
}
#endif
#endif
