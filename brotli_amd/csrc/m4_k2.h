// m4_k2.h -- the candidate search of matchfinder.M4 with ChainLength > 0 (m4.go:156-181) taken OUT of the
// serial parse: what one position's walk along its match chain can ever yield, computed for all positions
// at once (kernel K2, k2_eval.h), and how the parse (m4_parse.h) uses it.
//
// The walk at loop position i visits the positions c_0 > c_1 > ... with the same table index, nearest first
// (c_0 = the table entry, then up to ChainLength chain hops; it ends at position 0, beyond MaxDistance, or at
// the chain's end).  A candidate that passes the 4-byte test becomes extendMatch2(src, i, c, NextEmit): it
// runs forward by fwd = the common prefix of src[c:] and src[i:], and backward by min(bc, i - NextEmit) where
// bc = the common suffix of src[:c] and src[:i] (at most c: `candidate > 0`, m4.go:330).  The walk keeps the
// first candidate of the greatest score among those longer than MinLength (m4.go:161,177: strict `>`).
// Only NextEmit ties this to the state of the parse, and only through L = i - NextEmit, the room for the
// backward extension.  For every L a candidate's length is fwd + min(bc, L), so the walk's winner as a function
// of L is one of the few candidates on the upper envelope of those "ramp, then flat" functions -- usually one.
//
// K2 stores, per position, up to K2_ENTRIES candidates in walk order that contain the winner for EVERY L
// (k2_add drops a candidate only when a nearer kept one is at least as long for all L -- the distance term of
// the score, bitlen(distance) * DistanceBitCost with a cost >= 0, only favours the nearer one -- or when it
// can never be long enough; a kept one is dropped when a farther one beats its best case with its worst).
// The parse then runs the reference's selection over the kept candidates with the true L: removing
// candidates that never win does not change "first of the greatest score".  K2 looks at K2_FW bytes forward
// and K2_BW bytes backward; a candidate that agrees in all of them is kept with a `capped` mark and the
// parse extends it from there with its own reads, as before.  A position that would need more than
// K2_ENTRIES candidates is marked kK2Fallback and the parse walks its chain as before.  Either way the
// result is the reference's by construction; tools/analysis/k2_proto.cpp and tests/test_gpu_k2.py compare
// the two forms.
#pragma once
// (included by m4_parse.h, after the extension helpers and before parse_range)

namespace m4g {

// (K2_FW, K2_BW, K2_ENTRIES, K2Entry, K2Res: m4_parse.h, next to the memory accessors that carry them)

// The two 48-byte windows [p - K2_BW, p + K2_FW) of a position and of a candidate, as 12 dwords each.
// avail_f = min(K2_FW, len - i) bytes of the forward part are inside the buffer, avail_b = min(K2_BW, c) of
// the backward part (c < i).  Returns false when the 4-byte test fails, else the entry without its distance.
M4G_HD bool k2_pair(const uint32_t *a, const uint32_t *b, int32_t avail_f, int32_t avail_b, uint32_t &code) {
  if (a[4] != b[4]) return false;
  // (constant indices only: on the GPU both windows live in registers)
  int32_t f = K2_FW;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 3 + K2_FW / 4; k >= 5; k--) { // the lowest differing dword decides
    const uint32_t x = a[k] ^ b[k];
    if (x) f = (k - 4) * 4 + (__builtin_ctz(x) >> 3);
  }
  if (f > avail_f) f = avail_f; // (avail_f < K2_FW: the buffer ends there, the length is exact)
  int32_t n = K2_BW;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < K2_BW / 4; k++) { // the highest differing dword decides
    const uint32_t x = a[k] ^ b[k];
    if (x) n = (K2_BW / 4 - 1 - k) * 4 + (__builtin_clz(x) >> 3);
  }
  if (n > avail_b) n = avail_b;   // (avail_b < K2_BW: the candidate reaches position 0 there, exact)
  code = ((uint32_t)(f - 4) << kK2DistBits) | ((uint32_t)n << (kK2DistBits + 5));
  return true;
}
M4G_HD int32_t k2_fwd(uint32_t e) { return (int32_t)((e >> kK2DistBits) & 31) + 4; }
M4G_HD int32_t k2_bwd(uint32_t e) { return (int32_t)(e >> (kK2DistBits + 5)); }
M4G_HD bool k2_capped(uint32_t e) { return k2_fwd(e) == K2_FW || k2_bwd(e) == K2_BW; }

// Adds the candidate e (distance and code) -- farther than everything kept -- to the list of n kept ones
// (kept[n..] are 0).  Returns false when the list would overflow (the position becomes a fallback).
// (Loops of constant trip count and constant indices only: on the GPU the list lives in registers.)
M4G_HD bool k2_add(uint32_t *kept, int &n, uint32_t e, int32_t min_length, int32_t cost) {
  const int32_t f = k2_fwd(e), b = k2_bwd(e);
  const bool capped = k2_capped(e);
  const int32_t dterm = bitlen32(e & kK2DistMask) * cost;
  const int32_t s0 = f * 256 - dterm, s1 = (f + b) * 256 - dterm; // score at L = 0 and at L = infinity
  // (scores in units of 1/256 byte: 256 * length - bitlen(distance) * cost, m4.go:59-61; lengths here are <= 48)
  // never longer than MinLength, or never a positive score (m4.go:161,177)
  bool drop = !capped && (f + b <= min_length || s1 <= 0);
  uint32_t beaten = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < K2_ENTRIES; k++) {
    const uint32_t ke = kept[k];
    const int32_t kf = k2_fwd(ke), kb = k2_bwd(ke);
    const int32_t kd = bitlen32(ke & kK2DistMask) * cost; // <= dterm: nearer
    const int32_t k0 = kf * 256 - kd, k1 = (kf + kb) * 256 - kd;
    // The kept one (its kf, kb are lower bounds if it is capped) is nearer, so it wins ties; it is at least as
    // good for every L when it is at L = 0 and at L = infinity (the difference of two "ramp, then flat"
    // functions takes its minimum at one of the two ends).  It must also be long enough whenever this one
    // is: always (kf > MinLength), or because it is at least as long for every L.
    if (k < n && !capped && k0 >= s0 && k1 >= s1 && (kf > min_length || (kf >= f && kf + kb >= f + b))) drop = true;
    // A kept entry (uncapped: kf, kb exact) that this one beats strictly at L = 0 and at L = infinity (f, b are
    // lower bounds if this one is capped) loses for every L; a strictly greater score at a distance term that
    // is no smaller means strictly longer, so this one is long enough whenever the kept one is.
    if (k < n && !k2_capped(ke) && s0 > k0 && s1 > k1) beaten |= 1u << k;
  }
  if (drop) return true;
  // remove the beaten ones, keeping the order (from the top, so that the bits below stay valid)
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = K2_ENTRIES - 1; k >= 0; k--) {
    if (beaten & (1u << k)) {
#if defined(__HIPCC__)
#pragma unroll
#endif
      for (int t = k; t + 1 < K2_ENTRIES; t++) kept[t] = kept[t + 1];
      kept[K2_ENTRIES - 1] = 0;
      n--;
    }
  }
  if (n == K2_ENTRIES) return false;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < K2_ENTRIES; k++)
    if (k == n) kept[k] = e;
  n++;
  return true;
}

// A cheap test in front of k2_add, for the kernel's inner loop: most candidates lose, for every L, against ONE
// earlier candidate -- the "champion", the best one at L = 0 so far.  ch0 / ch1 are the champion's scores at
// L = 0 and at L = infinity (lower bounds if it is capped; INT32_MIN: no champion yet); a champion is longer
// than MinLength at L = 0 already, so it is long enough whenever the candidate is.  Returns true when e has to
// go through k2_add.  Dropping a candidate here rests on the EXISTENCE of a nearer candidate that is at least
// as good for every L (ties go to the nearer one), whether or not that one is still kept: if it is beaten
// later, then by a candidate that is better still.  The host reference applies the same test, so the results
// are the same word for word.
M4G_HD bool k2_filter(int32_t &ch0, int32_t &ch1, uint32_t e, int32_t min_length, int32_t cost) {
  const int32_t f = k2_fwd(e), b = k2_bwd(e);
  const bool capped = k2_capped(e);
  const int32_t dterm = bitlen32(e & kK2DistMask) * cost;
  const int32_t s0 = f * 256 - dterm, s1 = (f + b) * 256 - dterm;
  const bool lost = !capped && ((s0 <= ch0 && s1 <= ch1) || f + b <= min_length || s1 <= 0);
  if (f > min_length && s0 > ch0 && s1 >= ch1) {
    ch0 = s0;
    ch1 = s1;
  }
  return !lost;
}

// The selection of m4.go:156-181 over the kept candidates of position i.  Same flags, same comparisons as the
// walk in parse_range; `len` may be shorter than the buffer K2 saw (speculative lanes), which the walk's own
// extension would respect in the same way.
template <class Mem>
M4G_HD void k2_select(const Mem &M, int32_t len, const ParseParams &P, ParseState &S, int32_t i, const K2Res &r,
                      AbsMatch &cur, int64_t &cur_score) {
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < K2_ENTRIES; k++) {
    const uint32_t e = r.e[k];
    if (e == 0) break;
    const int32_t cand = i - (int32_t)(e & kK2DistMask);
    AbsMatch m;
    m.end = i + k2_fwd(e);
    if (k2_fwd(e) == K2_FW) m.end = extend_fwd(M, len, cand + K2_FW, i + K2_FW);
    if (m.end > len) m.end = len;
    const int32_t maxb = (i - S.next_emit) < cand ? (i - S.next_emit) : cand;
    int32_t nb = k2_bwd(e);
    if (nb > maxb) nb = maxb;
    else if (nb == K2_BW && maxb > K2_BW) nb = extend_bwd(M, i, cand, K2_BW, maxb);
    m.start = i - nb;
    m.match = cand - nb;
    if (!S.have_prev && m.start == S.next_emit) S.error |= kStateHitLo;
    if (m.end >= len) S.error |= kStateTouchedEnd;
    if (m.end - m.start > P.min_length) {
      const int64_t sc = score(m, P.distance_bit_cost);
      if (sc > cur_score) {
        cur = m;
        cur_score = sc;
      }
    }
  }
}

} // namespace m4g
