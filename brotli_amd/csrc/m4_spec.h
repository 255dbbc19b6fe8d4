// m4_spec.h -- speculative sub-block parse: how one inherently serial M4 parse
// is spread over many GPU lanes and still comes out bit-exact.
//
// The parse state at a loop top is (i, m0, m1, next_emit, prev_dist)
// (m4_parse.h).  Right after a loop-top flush (m4.go:113-126) at position i the
// state is (i, {}, {}, next_emit = i, prev_dist = d): it depends on NOTHING but
// the end position and the distance of the match just emitted.  So two parses
// of the same bytes that both flush a match with the same (end, distance) are
// in the same state from there on, whatever they did before.
//
//   phase A  (parallel, one lane per sub-block k of a job):  parse positions
//            [s_k, s_k+1) from the guessed state "nothing pending, nothing
//            emitted" and keep every record (flush records are flagged) plus
//            the end state E_k.  Block 0 starts from the true initial state.
//   phase B  (per job, in block order): carry the TRUE state T into block k
//            and run the real parse only until it performs a flush that the
//            speculative parse of block k also performed (same end, same
//            distance).  From that point the speculative records ARE the true
//            records and E_k is the true state entering block k+1.
//            If no such flush exists in the block, the true parse has simply
//            been run over the whole block and T continues into block k+1.
//
// The result per block is:  stitch records (true parse up to the sync point)
// followed by the speculative records after the sync point.  Correctness never
// depends on a guess being right -- a wrong guess only costs time.
#pragma once
#include "m4_parse.h"

namespace m4g {

struct BlockDesc {
  uint32_t n_spec;    // records the speculative parse produced for this block
  uint32_t n_stitch;  // true records produced by the stitch before the sync point
  uint32_t keep_from; // speculative records [keep_from, n_spec) are true
  uint32_t out_off;   // offset of this block's records inside the job's list
};

struct RecSink {
  Rec *out;
  uint32_t n, cap;
  M4G_HD void push(const Rec &r) {
    if (n < cap) out[n] = r;
    n++;
  }
};

M4G_HD int32_t rec_end(const Rec &r) {
  return (int32_t)((r.lit_start & ~kRecFlushBit) + r.unmatched + r.length);
}

// Geometry of a job's sub-blocks.  loop_end = len-7 clipped to >= start.
struct SpecGeom {
  int32_t start, loop_end, sub_block, nblocks;
};
M4G_HD SpecGeom spec_geom(uint32_t start, uint32_t len, int32_t sub_block) {
  SpecGeom g;
  g.start = (int32_t)start;
  g.loop_end = (int32_t)len - 7;
  if (g.loop_end < g.start) g.loop_end = g.start;
  g.sub_block = sub_block;
  int32_t span = g.loop_end - g.start;
  g.nblocks = span > 0 ? (span + sub_block - 1) / sub_block : 1;
  return g;
}
M4G_HD int32_t block_begin(const SpecGeom &g, int32_t k) { return g.start + k * g.sub_block; }
M4G_HD int32_t block_end(const SpecGeom &g, int32_t k) {
  return k == g.nblocks - 1 ? g.loop_end : g.start + (k + 1) * g.sub_block;
}

// Phase A for block k.
template <class Mem, class Sink>
M4G_HD void spec_parse_block(const Mem &M, int32_t len, const ParseParams &P,
                             const SpecGeom &g, int32_t k, uint32_t seed_prev, ParseState &E,
                             Sink &sink) {
  state_init(E, block_begin(g, k), k == 0 ? seed_prev : 0u);
  parse_range<false, true>(M, len, P, E, block_end(g, k), sink);
  if (k == g.nblocks - 1) parse_finish(len, P, E, sink);
}

// Phase B for block k >= 1.  T is the true state entering the block and is
// updated to the true state entering block k+1.  Returns true when the block
// synchronised with its speculative parse.
// CHAIN = false: for callers that know ChainLength == 0 (parse_range's switch of the same name).
template <bool CHAIN = true, bool XW = false, class Mem, class Sink>
M4G_HD bool stitch_block(const Mem &M, int32_t len, const ParseParams &P,
                         const SpecGeom &g, int32_t k, const Rec *__restrict__ spec,
                         uint32_t n_spec, const ParseState &E, ParseState &T, Sink &sink,
                         uint32_t &keep_from) {
  const int32_t stop = block_end(g, k);
  uint32_t j = 0;
  bool synced = false;
  while (T.i < stop) {
    if (!parse_range<true, false, CHAIN, true, XW>(M, len, P, T, stop, sink)) break;
    // a flush just happened: the last record ends at T.next_emit (== T.i)
    const int32_t e = T.next_emit;
    while (j < n_spec && rec_end(spec[j]) < e) j++;
    if (j < n_spec && rec_end(spec[j]) == e && (spec[j].lit_start & kRecFlushBit) &&
        spec[j].distance == T.prev_dist) {
      synced = true;
      break;
    }
  }
  if (synced) {
    const uint32_t err = T.error; // keep this parse's own flags, drop the speculative parse's diagnostics
    T = E;
    T.error = (E.error & kStateErrRange) | err;
    keep_from = j + 1;
    return true;
  }
  keep_from = n_spec;
  if (k == g.nblocks - 1) {
    parse_finish(len, P, T, sink);
    return false;
  }
  // No common flush, but the two parses may still have arrived in the same
  // state at the block end (typically: both hold the same pending match).
  // The repeat distance is dead while a match is pending -- the next flush
  // overwrites it before any probe reads it (m4.go:113-138).  Then every
  // speculative record of this block is dropped, the true ones are in the
  // sink, and E is (equivalent to) the true end state.
  const bool same_state =
      !(E.error & kStateAborted) && T.i == E.i && T.next_emit == E.next_emit &&
      T.m0.start == E.m0.start && T.m0.end == E.m0.end && T.m0.match == E.m0.match &&
      T.m1.start == E.m1.start && T.m1.end == E.m1.end && T.m1.match == E.m1.match &&
      (T.m0.end != 0 || (T.have_prev == E.have_prev && (!T.have_prev || T.prev_dist == E.prev_dist)));
  return same_state;
}

// desc.out_off written by the stitch (k3b), read by the validation pass (k3c)
constexpr uint32_t kStitchUnsynced = 0;   // whole block parsed from E[k-1], no common flush: ST + Tout hold the result
constexpr uint32_t kStitchSynced = 1;     // joined the speculative records at keep_from
constexpr uint32_t kStitchIncomplete = 2; // given up (entry state or this parse aborted): k3c parses the block
// bits 8..: number of following blocks the stitch of block k went on into (the chase): the status then
// speaks about blocks k .. k+extra together, all records are in ST[k], keep_from belongs to block k+extra
constexpr uint32_t K3_MAX_CHASE = 63;

// The stitch of block k >= 1 as k3b runs it: true parse from T (= E[k-1], the
// state the block is ASSUMED to be entered in) until it joins the speculative
// parse, following a pending match into the next blocks (the chase).
// E_k / desc_k / rs_k point at block k's entries of the per-job arrays (block
// k + x is entry x; slot records per block).  Fills d (status word in out_off);
// returns true when T must be kept as Tout[k] (the block did not synchronise).
template <bool CHAIN = true, bool XW = false, class Mem, class Sink>
M4G_HD bool stitch_chase(const Mem &M, int32_t len, const ParseParams &P, const SpecGeom &g,
                         uint32_t k, const Rec *rs_k, uint32_t slot, const ParseState *E_k,
                         const BlockDesc *desc_k, int32_t ext_limit, ParseState &T, Sink &sink,
                         BlockDesc &d) {
  uint32_t keep = 0, extra = 0; // extra: blocks after k this stitch went on into
  bool ok = false, gave_up = false;
  for (;;) {
    const uint32_t b = k + extra;
    const int32_t be = block_end(g, (int32_t)b);
    const int32_t len_eff = (int64_t)be + ext_limit < (int64_t)len ? be + ext_limit : len;
    const ParseState &Eb = E_k[extra];
    uint32_t n_spec = 0;
    if (!(Eb.error & kStateAborted)) n_spec = desc_k[extra].n_spec < slot ? desc_k[extra].n_spec : slot;
    ok = stitch_block<CHAIN, XW>(M, len_eff, P, g, (int32_t)b, rs_k + (size_t)extra * slot, n_spec, Eb, T, sink, keep);
    gave_up = len_eff < len && (T.error & kStateTouchedEnd);
    // A pending match that reaches past the block end leaves nothing to compare
    // with yet: go on into the next block (no records so far, so the slot
    // cannot overflow) until the match ends and the parse can join a speculative
    // one.  A block left with nothing pending, or with records, ends the stitch.
    if (ok || gave_up || sink.n != 0 || T.m0.end == 0 || T.i < be || b + 1 >= (uint32_t)g.nblocks ||
        extra >= K3_MAX_CHASE)
      break;
    extra++;
  }
  d.n_stitch = sink.n;
  d.keep_from = keep; // of block k + extra
  d.out_off = (gave_up ? kStitchIncomplete : (ok ? kStitchSynced : kStitchUnsynced)) | (extra << 8);
  return !ok && !gave_up;
}

// ---- k3c fast path ----------------------------------------------------------
// The true state T entering block k has nothing pending and stands at the block
// start; the speculative parse of the block assumed (nothing pending, nothing
// emitted, next_emit = block start).  The two differ only in next_emit and in
// the repeat distance.  next_emit matters only as the lower bound of backward
// extensions before the first emit (kStateHitLo records whether one reached the
// assumed bound) and in the first record's literal count.  The repeat distance
// matters only through the probe of m4.go:128-138, which runs at every loop
// position while nothing is pending: positions block start .. E.first_i, the
// position where the speculative parse found its first match, INCLUSIVE (the
// probe precedes the search).  From there a match stays pending until the first
// emit, and every emit replaces the distance.  (The first emitted record may
// END at or before first_i: an older match trimmed to the start of a newer one,
// m4.go:233-234 -- so the first record's end is NOT the bound.)
M4G_HD bool fast_entry_ok(const ParseState &T, const ParseState &Ek, int32_t bb) {
  return T.m0.end == 0 && T.i == bb && T.next_emit <= bb &&
         !(Ek.error & (kStateAborted | kStateHitLo | kStateErrRange));
}
// positions [bb, limit) must be checked with T's distance
M4G_HD int32_t fast_probe_limit(const ParseState &Ek, int32_t be) {
  return Ek.first_i >= 0 && Ek.first_i + 1 < be ? Ek.first_i + 1 : be;
}
// would the probe at loop position i with distance prev_dist do anything (or fault)?
M4G_HD bool fast_probe_hits(const uint8_t *src, int32_t len, int32_t i, uint32_t prev_dist) {
  const int32_t p = i + 1, c = p - (int32_t)prev_dist;
  return c < 0 || p + 4 > len || ld32(src + c) == ld32(src + p);
}

} // namespace m4g
