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
  parse_range<false>(M, len, P, E, block_end(g, k), sink);
  if (k == g.nblocks - 1) parse_finish(len, P, E, sink);
}

// Phase B for block k >= 1.  T is the true state entering the block and is
// updated to the true state entering block k+1.  Returns true when the block
// synchronised with its speculative parse.
template <class Mem, class Sink>
M4G_HD bool stitch_block(const Mem &M, int32_t len, const ParseParams &P,
                         const SpecGeom &g, int32_t k, const Rec *__restrict__ spec,
                         uint32_t n_spec, const ParseState &E, ParseState &T, Sink &sink,
                         uint32_t &keep_from) {
  const int32_t stop = block_end(g, k);
  uint32_t j = 0;
  bool synced = false;
  while (T.i < stop) {
    if (!parse_range<true>(M, len, P, T, stop, sink)) break;
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

} // namespace m4g
