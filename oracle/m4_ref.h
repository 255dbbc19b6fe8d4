/*
 * oracle/m4_ref.h -- CPU restatement of the reference's matchfinder.M4.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this code, and only as the checker.  The product path (brotli_amd/,
 * libm4g.so) never links or calls it.
 *
 * The reference (andybalholm/brotli) is pure Go and no Go toolchain exists in
 * the build image, so the reference itself cannot be run here.  This file is a
 * line-faithful restatement of
 *     matchfinder/m4.go:15-339      (M4, Reset, score, FindMatches,
 *                                    extendMatch, extendMatch2)
 *     matchfinder/emitter.go:5-34   (absoluteMatch, matchEmitter.emit)
 *     matchfinder/matchfinder.go:16-20 (Match)
 *
 * PARITY PIN: the reference ships no golden Match lists,
 * so bit-exact Match parity against the Go code is "parity unpinned".  What
 * pins this oracle is (a) the hand-derived known-answer tests of SURVEY.md
 * section 8(c), (b) an independently written second restatement
 * (oracle/m4_py.py) that must agree on random/text/degenerate inputs, and
 * (c) the reference's own test criterion -- the stream produced from these
 * matches round-trips through a brotli decoder (tests/test_roundtrip.py).
 */
#ifndef M4_REF_H
#define M4_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* matchfinder/matchfinder.go:16-20.  Go ints are 64-bit; every value on this
 * path is < 2^32, so the oracle uses the same 3 x u32 record as the C-ABI. */
typedef struct {
  uint32_t unmatched;
  uint32_t length;
  uint32_t distance;
} m4ref_match;

/* Growable []Match (the `dst` slice of FindMatches). */
typedef struct {
  m4ref_match *v;
  size_t len, cap;
} m4ref_matchvec;

/* matchfinder/m4.go:15-49 */
typedef struct {
  int64_t MaxDistance, MinLength, HashLen, TableBits, ChainLength,
      DistanceBitCost;
  uint32_t *table;
  size_t table_len;
  uint32_t *chain;
  size_t chain_len, chain_cap;
  uint8_t *history;
  size_t history_len, history_cap;
} m4ref;

m4ref *m4ref_new(int64_t max_distance, int64_t min_length, int64_t hash_len,
                 int64_t table_bits, int64_t chain_length,
                 int64_t distance_bit_cost);
void m4ref_free(m4ref *q);
/* m4.go:51-57 */
void m4ref_reset(m4ref *q);
/* m4.go:63-283.  Appends to dst and returns 0, or -1 on an input that would
 * make the Go code panic (out-of-range seeded repeat distance). */
int m4ref_find_matches(m4ref *q, m4ref_matchvec *dst, const uint8_t *src,
                       size_t n);

void m4ref_matchvec_free(m4ref_matchvec *v);

/* Convenience for ctypes callers: one FindMatches call on handle q with an
 * empty dst, optionally seeded with one previous Match whose Distance is
 * seed_prev_distance (0 = dst empty).  Writes up to out_cap records, returns
 * the count (or -(needed) if out_cap is too small, or INT64_MIN on error). */
int64_t m4ref_find_matches_flat(m4ref *q, const uint8_t *src, size_t n,
                                uint32_t seed_prev_distance, m4ref_match *out,
                                size_t out_cap);

/* Debug: copy the first `n` entries of table / chain (for K1 parity). */
size_t m4ref_dump_table(const m4ref *q, uint32_t *out, size_t n);
size_t m4ref_dump_chain(const m4ref *q, uint32_t *out, size_t n);
size_t m4ref_history_len(const m4ref *q);

#ifdef __cplusplus
}
#endif
#endif
