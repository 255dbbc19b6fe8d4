// tests/sim/parse_sim.cpp -- HOST build of brotli_amd/csrc/m4_parse.h for the
// CPU test-suite.  TEST HARNESS ONLY: it lets `pytest -m "not gpu"` check the
// parse state machine (and the sub-block speculation logic built on it)
// against the oracle without a GPU.  It is not linked into libm4g.so and no
// product code loads it.
//
// The chain array that kernel K1 produces on the GPU is produced here by a
// plain sequential table sweep (m4.go:140-147).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../brotli_amd/csrc/m4_parse.h"
#if __has_include("../../brotli_amd/csrc/m4_spec.h")
#include "../../brotli_amd/csrc/m4_spec.h"
#define HAVE_SPEC 1
#endif

using namespace m4g;

namespace {
struct VecSink {
  std::vector<Rec> v;
  void push(const Rec &r) { v.push_back(r); }
};

void build_chain(const uint8_t *src, int32_t start, int32_t len, int hash_len, int table_bits,
                 std::vector<uint32_t> &table, uint32_t *chain) {
  const uint64_t mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
  for (int32_t i = start; i < len - 7; i++) {
    uint64_t h = ((ld64(src + i) & mask) * 0x1E35A7BD1E35A7BDull) >> (64 - table_bits);
    uint32_t c = table[h];
    table[h] = (uint32_t)i;
    chain[i] = c ? (((uint32_t)i - c) | kChainMaybe) : 0;
  }
}
} // namespace

extern "C" {

// One fresh-state FindMatches over `src` (P0 semantics), serial parse.
// out: 4 u32 per record (unmatched, length, distance, lit_start).
int64_t sim_find_matches(const uint8_t *src, int32_t n, int max_distance, int min_length,
                         int hash_len, int table_bits, int chain_length, int distance_bit_cost,
                         uint32_t seed_prev, uint32_t *out, int64_t out_cap) {
  ParseParams P{max_distance, min_length, hash_len, chain_length, distance_bit_cost};
  std::vector<uint32_t> table((size_t)1 << table_bits, 0), chain((size_t)n + 8, 0);
  build_chain(src, 0, n, hash_len, table_bits, table, chain.data());
  VecSink sink;
  ParseState S;
  state_init(S, 0, seed_prev);
  GlobalMem M{src, chain.data()};
  parse_range(M, n, P, S, n - 7, sink);
  parse_finish(n, P, S, sink);
  if ((int64_t)sink.v.size() > out_cap) return -(int64_t)sink.v.size();
  for (auto &r : sink.v) r.lit_start &= ~kRecFlushBit;
  if (!sink.v.empty()) std::memcpy(out, sink.v.data(), sink.v.size() * sizeof(Rec));
  return (S.error & m4g::kStateErrRange) ? INT64_MIN : (int64_t)sink.v.size();
}

#ifdef HAVE_SPEC
// Same result through the speculative sub-block scheme of m4_spec.h, executed
// sequentially on the host in the order the device kernels use (all phase-A
// blocks first, then phase B in block order).
// stats[0] = sub-blocks, stats[1] = sub-blocks that synchronised,
// stats[2] = positions walked by the stitcher, stats[3] = stitch records.
int64_t sim_find_matches_spec(const uint8_t *src, int32_t n, int max_distance, int min_length,
                              int hash_len, int table_bits, int chain_length,
                              int distance_bit_cost, uint32_t seed_prev, int32_t sub_block,
                              uint32_t *out, int64_t out_cap, int64_t *stats) {
  ParseParams P{max_distance, min_length, hash_len, chain_length, distance_bit_cost};
  std::vector<uint32_t> table((size_t)1 << table_bits, 0), chain((size_t)n + 8, 0);
  build_chain(src, 0, n, hash_len, table_bits, table, chain.data());
  GlobalMem M{src, chain.data()};
  const SpecGeom g = spec_geom(0, (uint32_t)n, sub_block);
  const int K = g.nblocks;
  std::vector<std::vector<Rec>> RS(K);
  std::vector<ParseState> E(K);
  for (int k = 0; k < K; k++) {
    VecSink sink;
    spec_parse_block(M, n, P, g, k, seed_prev, E[k], sink);
    RS[k] = std::move(sink.v);
  }
  std::vector<Rec> result = RS[0];
  ParseState T = E[0];
  int dbg_left = 6;
  int64_t synced = 0, walked = 0, nst = 0;
  for (int k = 1; k < K; k++) {
    VecSink st;
    uint32_t keep = 0;
    int32_t i0 = T.i;
    ParseState T0 = T;
    bool ok = stitch_block(M, n, P, g, k, RS[k].data(), (uint32_t)RS[k].size(),
                           E[k], T, st, keep);
    if (!ok && std::getenv("SIM_DEBUG") && T0.i < block_end(g, k) && dbg_left-- > 0) {
      std::fprintf(stderr, "block %d [%d,%d) unsynced: entry i=%d m0=[%d,%d) d=%d m1=[%d,%d) ne=%d | exit i=%d m0=[%d,%d) d=%d\n", k,
                   block_begin(g, k), block_end(g, k), T0.i, T0.m0.start, T0.m0.end, T0.m0.start - T0.m0.match,
                   T0.m1.start, T0.m1.end, T0.next_emit, T.i, T.m0.start, T.m0.end, T.m0.start - T.m0.match);
      for (auto &r : st.v) std::fprintf(stderr, "   true rec end=%d len=%u dist=%u flush=%d\n", rec_end(r), r.length, r.distance, (int)(r.lit_start >> 31));
      for (auto &r : RS[k]) std::fprintf(stderr, "   spec rec end=%d len=%u dist=%u flush=%d\n", rec_end(r), r.length, r.distance, (int)(r.lit_start >> 31));
      std::fprintf(stderr, "   spec end state i=%d m0=[%d,%d) d=%d\n", E[k].i, E[k].m0.start, E[k].m0.end, E[k].m0.start - E[k].m0.match);
    }
    synced += ok;
    if (ok && !st.v.empty()) {
      int32_t e = rec_end(st.v.back());
      walked += e > i0 ? e - i0 : 0;
    } else {
      walked += block_end(g, k) > i0 ? block_end(g, k) - i0 : 0;
    }
    nst += (int64_t)st.v.size();
    result.insert(result.end(), st.v.begin(), st.v.end());
    result.insert(result.end(), RS[k].begin() + keep, RS[k].end());
  }
  if (stats) {
    stats[0] = K;
    stats[1] = synced;
    stats[2] = walked;
    stats[3] = nst;
  }
  for (auto &r : result) r.lit_start &= ~kRecFlushBit;
  if ((int64_t)result.size() > out_cap) return -(int64_t)result.size();
  if (!result.empty()) std::memcpy(out, result.data(), result.size() * sizeof(Rec));
  return (T.error & m4g::kStateErrRange) ? INT64_MIN : (int64_t)result.size();
}
#endif
}
