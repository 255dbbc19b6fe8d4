// m4_fast.h -- the M4 parse (m4.go:112-260) cut into EVENTS, for ChainLength == 0.
//
// parse_range (m4_parse.h) runs the reference's loop one position at a time.  When 64 lanes of a
// wave each run their own copy of it, every wave-iteration executes the union of all its paths
// and waits for a round trip to far memory, although most iterations do nothing (measured on the
// bench text: per KiB of input 382 iterations, of which 86 flush a match, 106 look at a candidate
// and 0.65 find a repeat match; the rest only move on).  Here the same parse is a cycle of three
// short per-lane steps, so that a wave takes one round per EVENT instead of one per position:
//
//   fast_light     no memory beyond the lane's window: skip every position at which nothing can
//                  happen -- up to five literal positions per call (the repeat probe of
//                  m4.go:128-138 is checked for all of them at once on the 8 cached source
//                  bytes; the table entries through a bit mask made when the window was filled) --
//                  and stop at the first position with an event: a flush is due (m4.go:113), the
//                  table entry may match (m4.go:152-159), the probe hits or needs bytes.
//   fast_issue     at the event: flush (state only), the probe at this position if its bytes are
//                  cached, then say which far bytes the rest of the iteration needs: 8 bytes for
//                  the probe, 32 around the candidate.
//   fast_complete  with those bytes: probe, candidate evaluation (extendMatch2 within the 24
//                  bytes), score test, push (m4.go:159-209).
//
// Whatever does not fit -- a probe that hits, a match longer than the bytes at hand, three pending
// matches (m4.go:211-259), the first and last bytes of a buffer -- sets FastLane.rare BEFORE any
// state of that iteration has changed (a flush excepted: it is complete in itself), and the caller
// runs that one iteration with parse_range.  So every position is processed either by code that
// was checked to do nothing there, by these few lines, or by the generic loop: same result.
//
// Host and device compile the same functions; tests/sim/parse_sim.cpp runs them against the oracle.
#pragma once
#include "m4_parse.h"

namespace m4g {

struct FastLane {
  uint64_t pc;      // 8 source bytes of the repeat probe, at position pc_base
  int32_t pc_base;
  int32_t ld_probe; // >= 0: fast_complete wants the 8 bytes at this position
  int32_t ld_cand;  // >= 8: ... and the bytes [ld_cand - 8, ld_cand + 16)
  uint32_t rare;    // the iteration at S.i is for parse_range
};

M4G_HD void fast_init(FastLane &F) {
  F.pc = 0;
  F.pc_base = -0x40000000;
  F.ld_probe = -1;
  F.ld_cand = -1;
  F.rare = 0;
}

constexpr int kFastEvent = 0;  // S.i stands at a position for fast_issue / fast_complete
constexpr int kFastWindow = 1; // S.i has left the window
constexpr int kFastDone = 2;   // S.i >= the end of the range

constexpr int KF_W = 16; // positions per window; the window serves bytes [w0 - 16, w0 + KF_W + 16)

// m4.go:152-159 can get past its tests: an entry within MaxDistance whose check bits agree
M4G_HD bool fast_heavy(uint32_t ce, const ParseParams &P) {
  return (ce & kChainMaybe) != 0 && (ce & kChainDelta) - 1u < (uint32_t)P.max_distance;
}

// 0x80 in every byte of the result whose byte in x is zero (exact: no carries between bytes)
M4G_HD uint32_t zero_bytes(uint32_t x) {
  const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}

// Smallest k in 0..4 such that bytes k..k+3 of x are all zero; 5 if there is none.
M4G_HD uint32_t first_zero_quad(uint64_t x) {
  const uint32_t zl = zero_bytes((uint32_t)x), zh = zero_bytes((uint32_t)(x >> 32));
  const uint64_t z = ((uint64_t)zh << 32) | zl;
  const uint32_t h = zl & (uint32_t)(z >> 8) & (uint32_t)(z >> 16) & (uint32_t)(z >> 24);
  if (h != 0) return (uint32_t)__builtin_ctz(h) >> 3;
  return zh == 0x80808080u ? 4u : 5u;
}

// The functions below are written as straight-line code with selects: on the GPU every `if` that
// only some lanes take costs a handful of scalar instructions for the execution mask, and lanes
// for which a computed value is meaningless simply do not use it.  Positions handed to the
// accessors are clamped into the window for such lanes.
//
// Mem: near_chain(i), near32(p), near64(p) for positions inside the window (bytes from 16 below
// its first position to 16 above its last), heavy_mask(i) = bit k set iff fast_heavy(entry of
// i + k), for the positions of the window from i on (no bits beyond).
template <class Mem>
M4G_HD int fast_light(const Mem &M, const ParseParams &P, ParseState &S, const FastLane &F, int32_t be,
                      int32_t wend) {
  const int32_t i = S.i, e0 = S.m0.end;
  const bool pend = e0 != 0, interior = pend && i < e0;
  const int32_t lim_pos = be < wend ? be : wend;
  const bool here = i < lim_pos && i >= wend - KF_W; // position i is inside the range and the window
  const int32_t ii = here ? i : wend - 1;
  const int st_out = i >= be ? kFastDone : kFastWindow; // when not here
  // the one searched position inside a pending match (m4.go:149-151)
  const bool heavy_i = fast_heavy(M.near_chain(ii), P);
  const int32_t nx = next_pos(S, P, i);
  // literal positions i, i + 1, ...: the first one whose table entry may match ...
  const uint32_t limit = (uint32_t)(lim_pos - ii) & 31u; // 1..KF_W positions ahead (when here)
  uint32_t n = (uint32_t)__builtin_ctz(M.heavy_mask(ii) | (1u << limit));
  // ... or whose repeat probe (m4.go:128-138) hits or runs out of cached bytes: position i + k
  // compares src[i + 1 + k ..] with the cached source bytes from offset off + k on
  const int32_t c = i + 1 - (int32_t)S.prev_dist;
  const uint32_t off = (uint32_t)(c - F.pc_base);
  const bool ev_now = S.have_prev && (c < 0 || off > 4);
  const uint32_t cover = 5 - (off & 7);
  const uint32_t first = first_zero_quad((F.pc >> (8 * (off & 7))) ^ M.near64(ii + 1));
  const uint32_t np = first < cover ? first : cover; // a hit, or the first position not covered
  if (S.have_prev && np < n) n = np;
  if (ev_now) n = 0;
  const int st_lit = n < limit ? kFastEvent : (lim_pos >= be ? kFastDone : kFastWindow);
  int st;
  int32_t ni;
  if (pend) {
    const bool stay = !interior || !here || heavy_i; // else move on to the end of the match
    ni = stay ? i : nx;
    st = interior && !here ? st_out : (ni >= be ? kFastDone : kFastEvent);
  } else {
    ni = here ? i + (int32_t)n : i;
    st = here ? st_lit : st_out;
  }
  S.i = ni;
  return st;
}

// The event at S.i, part 1: flush, the repeat probe on cached bytes, and which far bytes the rest
// of the iteration needs.  in_window: S.i lies in the window (a flush can be due beyond it; then
// only the flush is done here and the position is looked at again once the window is there).
// Returns true when nothing is left to do at this position (no bytes wanted, not rare): the caller
// may go straight on with fast_light.
template <class Mem, class Sink>
M4G_HD bool fast_issue(const Mem &M, int32_t len, const ParseParams &P, ParseState &S, FastLane &F, Sink &sink,
                       bool in_window, int32_t w0) {
  const int32_t i = S.i;
  if (S.m0.end != 0 && i >= S.m0.end) flush(S, P, sink); // m4.go:113-126
  const bool probing = S.m0.end == 0 && S.have_prev;
  const int32_t c = i + 1 - (int32_t)S.prev_dist;
  const uint32_t off = (uint32_t)(c - F.pc_base);
  // c < 0: the Go code would panic on the slice bounds; parse_range records it
  bool rare = probing && c < 0;
  const bool want_probe = probing && c >= 0 && off > 4; // (c + 8 <= len: c <= i <= len - 8)
  // m4.go:128-138 on the cached bytes: a hit is for parse_range
  const int32_t ii = in_window ? i : w0;
  const bool hit = probing && in_window && off <= 4 && (uint32_t)(F.pc >> (8 * (off & 7))) == M.near32(ii + 1);
  rare = rare || hit;
  const uint32_t ce = M.near_chain(ii);
  const bool hv = in_window && fast_heavy(ce, P);
  const int32_t cand = i - (int32_t)(ce & kChainDelta);
  const bool edge = cand < 8 || i + 24 > len; // the ends of the buffer
  rare = rare || (hv && edge);
  F.ld_probe = want_probe && !rare ? c : -1;
  F.ld_cand = hv && !rare ? cand : -1;
  F.rare = rare;
  return in_window && !rare && !want_probe && !hv;
}

// 8 bytes from byte offset o (5..16) of the 24 bytes q0..q5 (little-endian dwords)
M4G_HD uint64_t bytes_at(uint64_t a, uint64_t b, uint64_t c, uint32_t o) {
  const uint32_t q[7] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32),
                         (uint32_t)c, (uint32_t)(c >> 32), 0u};
  const uint32_t j = o >> 2, sh = 8 * (o & 3);
  uint32_t d0 = q[1], d1 = q[2], d2 = q[3];
  if (j == 2) d0 = q[2], d1 = q[3], d2 = q[4];
  if (j == 3) d0 = q[3], d1 = q[4], d2 = q[5];
  if (j == 4) d0 = q[4], d1 = q[5], d2 = q[6];
  const uint64_t lo = ((uint64_t)d1 << 32) | d0, hi = ((uint64_t)d2 << 32) | d1;
  return (uint64_t)(uint32_t)(lo >> sh) | ((uint64_t)(uint32_t)(hi >> sh) << 32);
}

// The event at S.i, part 2.  pl = the 8 bytes at F.ld_probe; cb / cf / cf2 / cf3 = the 8 bytes at
// F.ld_cand - 8 / + 0 / + 8 / + 16 (each only if asked for).  in_window as above: without the
// window only the probe bytes are kept.
template <class Mem>
M4G_HD void fast_complete(const Mem &M, int32_t len, const ParseParams &P, ParseState &S, FastLane &F, bool in_window,
                          uint64_t pl, uint64_t cb, uint64_t cf, uint64_t cf2, uint64_t cf3) {
  const bool fresh = F.ld_probe >= 0;
  if (fresh) {
    F.pc = pl;
    F.pc_base = F.ld_probe;
  }
  if (!in_window) return;
  const int32_t i = S.i;
  // m4.go:128-138 at this position on the bytes that just came: a hit is for parse_range
  if (fresh && (uint32_t)pl == M.near32(i + 1)) {
    F.rare = 1;
    return;
  }
  if (F.ld_cand < 0) return; // nothing else here: fast_light goes on from this position
  // m4.go:159-164: the 4-byte test and extendMatch2, as far as 16 bytes forward and 8 backward
  const int32_t cand = F.ld_cand;
  const uint64_t x = cf ^ M.near64(i), x2 = cf2 ^ M.near64(i + 8), y = cb ^ M.near64(i - 8);
  const bool hit = (uint32_t)x == 0;
  const bool long_fwd = x == 0 && x2 == 0; // longer than the bytes at hand
  const int32_t end = x != 0 ? i + (int32_t)(__builtin_ctzll(x | (1ull << 63)) >> 3)
                             : i + 8 + (int32_t)(__builtin_ctzll(x2 | (1ull << 63)) >> 3);
  int32_t maxb = (i - S.next_emit) < cand ? (i - S.next_emit) : cand;
  if (maxb < 0) maxb = 0;
  int32_t nb = y != 0 ? (int32_t)(__builtin_clzll(y | 1ull) >> 3) : 8;
  if (nb > maxb) nb = maxb;
  const bool long_bwd = nb == 8 && maxb > 8;
  AbsMatch cur;
  cur.start = i - nb;
  cur.match = cand - nb;
  cur.end = end;
  // m4.go:59-61; at most 24 bytes long: 32 bits do
  const int32_t sc = (cur.end - cur.start) * 256 - bitlen32((uint32_t)(cur.start - cur.match)) * P.distance_bit_cost;
  const bool good = hit && cur.end - cur.start > P.min_length && sc > 0; // else m4.go:183-185
  // m4.go:187-199
  const int64_t penalty = S.m0.end != 0 && cur.start > S.m1.end ? 275 : 0;
  const bool better = good && (int64_t)sc > score(S.m0, P.distance_bit_cost) + penalty;
  const bool third = better && S.m1.end != 0; // a third pending match: m4.go:211-259
  if (hit && (long_fwd || long_bwd || third)) {
    F.rare = 1;
    return;
  }
  if (hit) {
    if (!S.have_prev && cur.start == S.next_emit) S.error |= kStateHitLo;
    if (cur.end >= len) S.error |= kStateTouchedEnd;
  }
  if (better) { // m4.go:201-209
    S.m1 = S.m0;
    S.m0 = cur;
    if (S.first_i < 0) S.first_i = i;
    // When this match is flushed, the repeat probe (m4.go:128-138) starts one byte behind its
    // source: those bytes are among the ones just read (the match ends at most 15 bytes above i).
    F.pc_base = cand + (end - i) + 1;
    F.pc = bytes_at(cf, cf2, cf3, (uint32_t)(end - i + 1));
  }
  S.i = next_pos(S, P, i);
}

} // namespace m4g
