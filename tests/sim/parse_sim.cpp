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
#include "../../brotli_amd/csrc/m4_fast.h"

using namespace m4g;

namespace {
struct VecSink {
  std::vector<Rec> v;
  void push(const Rec &r) { v.push_back(r); }
};

// Host stand-in for the device's FastMem (kernels.h): a window of KF_W positions, aligned like
// the device's, over plain memory.  Reads beyond the buffer return zeros (the device window holds
// whatever follows the chunk; m4_fast.h never lets those bytes decide anything).
struct HostFastMem : GlobalMem {
  int32_t n, w0;
  const ParseParams *P;
  uint32_t near_chain(int32_t p) const { return p >= 0 && p < n ? chain[p] : 0u; }
  uint64_t near64(int32_t p) const {
    uint64_t v = 0;
    for (int k = 0; k < 8; k++)
      if (p + k >= 0 && p + k < n) v |= (uint64_t)src[p + k] << (8 * k);
    return v;
  }
  uint32_t near32(int32_t p) const { return (uint32_t)near64(p); }
  uint32_t heavy_mask(int32_t p) const {
    uint32_t m = 0;
    for (int32_t q = p < 0 ? 0 : p; q < w0 + KF_W; q++)
      if (q < n && fast_heavy(chain[q], *P)) m |= 1u << (q - p);
    return m;
  }
};

// k3a_fast (kernels.h) for one lane: the same cycle of fast_light / fast_issue / fast_complete,
// the parked iteration run at once.  Returns false when the parse was given up (len_eff < n).
template <class Sink>
bool fast_lane_parse(const uint8_t *src, const uint32_t *chain, int32_t n, int32_t len_eff, const ParseParams &P,
                     ParseState &S, int32_t be, Sink &sink) {
  HostFastMem M;
  M.src = src, M.chain = chain, M.n = n, M.P = &P;
  M.w0 = -0x40000000;
  GlobalMem G{src, chain};
  FastLane F;
  fast_init(F);
  for (;;) {
    const int st = fast_light(M, P, S, F, be, M.w0 + KF_W);
    if (st == kFastDone) return true;
    if (st == kFastWindow) {
      M.w0 = S.i & ~(KF_W - 1);
      continue;
    }
    const bool inw = (uint32_t)(S.i - M.w0) < (uint32_t)KF_W;
    const bool free_ = fast_issue(M, len_eff, P, S, F, sink, inw, S.i);
    if (free_) continue; // (the device runs fast_light a second time in the same round)
    uint64_t pl = 0, cb = 0, cf = 0, cf2 = 0, cf3 = 0;
    if (F.ld_probe >= 0) pl = ld64(src + F.ld_probe);
    if (F.ld_cand >= 0)
      cb = ld64(src + F.ld_cand - 8), cf = ld64(src + F.ld_cand), cf2 = ld64(src + F.ld_cand + 8), cf3 = ld64(src + F.ld_cand + 16);
    if (!inw) M.w0 = S.i & ~(KF_W - 1); // (the device may serve the window a round later: in_window = false then)
    if (!F.rare) fast_complete(M, len_eff, P, S, F, inw, pl, cb, cf, cf2, cf3);
    if (F.rare) parse_range<false, false, false, false>(G, len_eff, P, S, S.i + 1, sink);
    if (len_eff < n && (S.error & kStateTouchedEnd)) return false;
  }
}

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
    if (P.chain_length == 0 && !std::getenv("SIM_NOFAST")) {
      state_init(E[k], block_begin(g, k), k == 0 ? seed_prev : 0u);
      fast_lane_parse(src, chain.data(), n, n, P, E[k], block_end(g, k), sink);
      if (k == g.nblocks - 1) parse_finish(n, P, E[k], sink);
    } else {
      spec_parse_block(M, n, P, g, k, seed_prev, E[k], sink);
    }
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

// The whole K3 stage as the device runs it -- k3a (speculative parse with the
// give-up rule), k3b (stitch_chase, shared with the kernel), k3c (validation:
// scan of synchronised blocks, covered-block skip, fast path, serial re-stitch)
// -- executed on the host, one "lane" after the other.  k3c's control flow is
// restated here in scalar form (the kernel spreads it over a wave); its
// decisions come from the same helpers (m4_spec.h: fast_entry_ok,
// fast_probe_limit, fast_probe_hits, stitch_block).
// legacy_limit != 0 reproduces the pre-round-2 bound of the fast path (probe
// checks only below the END of the first speculative record) so that a test can
// show an input on which that bound is wrong.
// stats: [0] blocks, [1] fast-path blocks, [2] serially parsed, [3] skipped, [4] aborted spec lanes
int64_t sim_find_matches_k3(const uint8_t *src, int32_t n, int max_distance, int min_length,
                            int hash_len, int table_bits, int chain_length, int distance_bit_cost,
                            uint32_t seed_prev, int32_t sub_block, int32_t ext_limit,
                            int legacy_limit, uint32_t *out, int64_t out_cap, int64_t *stats) {
  ParseParams P{max_distance, min_length, hash_len, chain_length, distance_bit_cost};
  std::vector<uint32_t> table((size_t)1 << table_bits, 0), chain((size_t)n + 8, 0);
  build_chain(src, 0, n, hash_len, table_bits, table, chain.data());
  GlobalMem M{src, chain.data()};
  const SpecGeom g = spec_geom(0, (uint32_t)n, sub_block);
  const uint32_t K = (uint32_t)g.nblocks;
  // SIM_SLOT: a smaller slot than the bound, as a compact batch plan has (m4g.hip spec_alloc): the run must stay
  // inside its buffers and report the overflow (-2)
  const uint32_t slot = std::getenv("SIM_SLOT") ? (uint32_t)std::atoi(std::getenv("SIM_SLOT")) : (uint32_t)(sub_block / min_length + 18);
  std::vector<Rec> rs((size_t)K * slot), st((size_t)K * slot);
  std::vector<ParseState> E(K), Tout(K);
  std::vector<BlockDesc> desc(K);
  int64_t c_fast = 0, c_slow = 0, c_skip = 0, c_abort = 0;
  uint32_t err = 0;
  // ---- k3a ----
  for (uint32_t k = 0; k < K; k++) {
    const int32_t bb = block_begin(g, (int32_t)k), be = block_end(g, (int32_t)k);
    const int32_t len_eff = (int64_t)be + ext_limit < (int64_t)n ? be + ext_limit : n;
    RecSink sink{rs.data() + (size_t)k * slot, 0, slot};
    ParseState S;
    state_init(S, bb, k == 0 ? seed_prev : 0u);
    if (P.chain_length == 0 && !std::getenv("SIM_NOFAST"))
      fast_lane_parse(src, chain.data(), n, len_eff, P, S, be, sink);
    else
      parse_range<false, true>(M, len_eff, P, S, be, sink);
    const bool aborted = len_eff < n && (S.error & kStateTouchedEnd);
    if (aborted) {
      S.error |= kStateAborted;
      sink.n = 0;
      c_abort++;
    } else if (k == K - 1) {
      parse_finish(n, P, S, sink);
    }
    E[k] = S;
    desc[k] = BlockDesc{sink.n, 0, 0, 0};
    if (sink.n > slot) err |= 2u;
    if (k == 0 && !aborted && (S.error & kStateErrRange)) err |= 1u;
  }
  // ---- k3b ----
  for (uint32_t k = 1; k < K; k++) {
    RecSink sink{st.data() + (size_t)k * slot, 0, slot};
    ParseState T = E[k - 1];
    BlockDesc d = desc[k];
    if (T.error & kStateAborted) {
      d.n_stitch = 0;
      d.keep_from = 0;
      d.out_off = kStitchIncomplete;
      desc[k] = d;
      continue;
    }
    T.error &= kStateErrRange;
    // the kernel's shortcut for a block that emitted nothing (kernels.h, k3b_stitch)
    if (d.n_spec == 0 && k + 1 < K && !T.have_prev && fast_entry_ok(T, E[k], block_begin(g, (int32_t)k)) &&
        !std::getenv("SIM_NO_K3B_SHORTCUT")) {
      ParseState R = E[k];
      R.error = (E[k].error & kStateErrRange) | T.error;
      R.next_emit = T.next_emit;
      R.have_prev = T.have_prev;
      R.prev_dist = T.prev_dist;
      Tout[k] = R;
      d.n_stitch = 0;
      d.keep_from = 0;
      d.out_off = kStitchUnsynced;
      desc[k] = d;
      continue;
    }
    if (stitch_chase(M, n, P, g, k, rs.data() + (size_t)k * slot, slot, E.data() + k, desc.data() + k,
                     ext_limit, T, sink, d))
      Tout[k] = T;
    desc[k] = d;
    if (sink.n > slot) err |= 2u;
  }
  // ---- k3c ----
  uint32_t off = 0, k = 1;
  bool valid = true;
  ParseState T;
  state_init(T, g.start, seed_prev);
  if (E[0].error & kStateAborted) {
    valid = false;
    k = 0;
  } else {
    off = desc[0].n_spec;
    desc[0].out_off = 0;
  }
  while (k < K) {
    if (valid) {
      BlockDesc &d = desc[k];
      if (d.out_off == kStitchSynced) { // plain sync
        const uint32_t cnt = d.n_stitch + (d.n_spec - d.keep_from);
        d.out_off = off;
        off += cnt;
        k++;
        continue;
      }
      const uint32_t word = d.out_off, status = word & 0xFFu, extra = word >> 8;
      const uint32_t ns = d.n_stitch, keep = d.keep_from;
      if (status == kStitchIncomplete) {
        valid = false;
        T = E[k - 1];
        T.error &= kStateErrRange;
      } else {
        uint32_t tail = 0;
        for (uint32_t l = 0; l <= extra && k + l < K; l++) {
          BlockDesc nd = desc[k + l];
          const uint32_t nsp = nd.n_spec < slot ? nd.n_spec : slot;
          nd.n_stitch = l == 0 ? ns : 0;
          nd.keep_from = (status == kStitchSynced && l == extra) ? keep : nsp;
          nd.out_off = off + (l == 0 ? 0 : ns);
          if (status == kStitchSynced && l == extra) tail = nsp - (keep < nsp ? keep : nsp);
          desc[k + l] = nd;
        }
        off += ns + tail;
        if (status == kStitchUnsynced) {
          valid = false;
          T = Tout[k];
        }
        k += extra + 1;
      }
    } else {
      const BlockDesc d = desc[k];
      const ParseState Ek = E[k];
      const int32_t bb = block_begin(g, (int32_t)k), be = block_end(g, (int32_t)k);
      const uint32_t n_spec = d.n_spec < slot ? d.n_spec : slot;
      if (T.i >= be && k + 1 < K) {
        uint32_t covered_end = (uint32_t)((T.i - g.start) / g.sub_block);
        if (covered_end > K - 1) covered_end = K - 1;
        for (uint32_t kk = k; kk < covered_end; kk++) {
          desc[kk].n_stitch = 0;
          desc[kk].keep_from = desc[kk].n_spec < slot ? desc[kk].n_spec : slot;
          desc[kk].out_off = off;
        }
        c_skip += covered_end - k;
        k = covered_end;
        continue;
      }
      bool fast = fast_entry_ok(T, Ek, bb);
      if (fast && T.have_prev) {
        int32_t limit = fast_probe_limit(Ek, be);
        if (legacy_limit) {
          limit = be;
          if (n_spec) limit = rec_end(rs[(size_t)k * slot]) < be ? rec_end(rs[(size_t)k * slot]) : be;
        }
        for (int32_t i = bb; i < limit && fast; i++)
          if (fast_probe_hits(src, n, i, T.prev_dist)) fast = false;
      }
      if (fast) {
        if (n_spec) {
          Rec &r = rs[(size_t)k * slot];
          const uint32_t flag = r.lit_start & kRecFlushBit;
          r.unmatched += (r.lit_start & ~kRecFlushBit) - (uint32_t)T.next_emit;
          r.lit_start = (uint32_t)T.next_emit | flag;
        }
        desc[k].n_stitch = 0;
        desc[k].keep_from = 0;
        desc[k].out_off = off;
        off += n_spec;
        c_fast++;
        const ParseState prevT = T;
        T = Ek;
        T.error = (Ek.error & kStateErrRange) | prevT.error;
        if (n_spec == 0) {
          T.next_emit = prevT.next_emit;
          T.have_prev = prevT.have_prev;
          T.prev_dist = prevT.prev_dist;
        } else {
          valid = true;
        }
        k++;
        continue;
      }
      if (T.i >= be) c_skip++; else c_slow++;
      RecSink sink{st.data() + (size_t)k * slot, 0, slot};
      uint32_t keep = 0;
      const bool ok = stitch_block(M, n, P, g, (int32_t)k, rs.data() + (size_t)k * slot, n_spec, Ek, T, sink, keep);
      if (sink.n > slot) err |= 2u;
      desc[k].n_stitch = sink.n;
      desc[k].keep_from = keep;
      desc[k].out_off = off;
      off += sink.n + (n_spec - keep);
      if (ok) valid = true;
      k++;
    }
  }
  if (T.error & kStateErrRange) err |= 1u;
  // ---- k4: stitch part, then the kept speculative part, block by block ----
  std::vector<Rec> result(off);
  for (uint32_t b = 0; b < K; b++) {
    const BlockDesc &d = desc[b];
    const uint32_t nsp = d.n_spec < slot ? d.n_spec : slot;
    const uint32_t ns = d.n_stitch < slot ? d.n_stitch : slot;
    const uint32_t kf = d.keep_from < nsp ? d.keep_from : nsp;
    const uint32_t nk = nsp - kf;
    for (uint32_t r = 0; r < ns + nk; r++) {
      if ((size_t)d.out_off + r >= result.size()) return -3; // layout broken
      result[d.out_off + r] = r < ns ? st[(size_t)b * slot + r] : rs[(size_t)b * slot + d.keep_from + (r - ns)];
    }
  }
  if (stats) {
    stats[0] = K;
    stats[1] = c_fast;
    stats[2] = c_slow;
    stats[3] = c_skip;
    stats[4] = c_abort;
  }
  if (err & 2u) return -2;
  for (auto &r : result) r.lit_start &= ~kRecFlushBit;
  if ((int64_t)result.size() > out_cap) return -(int64_t)result.size() - 16;
  if (!result.empty()) std::memcpy(out, result.data(), result.size() * sizeof(Rec));
  return (err & 1u) ? INT64_MIN : (int64_t)result.size();
}
#endif

// K2 on the host, by walking the chains (the kernels k1d_depth / k2_place / k2_eval get the same candidates from
// the bucket-sorted position list): per position 4 words of K2Res and, in mark[i], whether the position's chain
// entry carries the mark afterwards.  The reference for tests/test_gpu_k2.py; m4_k2.h explains the contents.
void sim_k2_reference(const uint8_t *src, int32_t n, int max_distance, int min_length, int hash_len, int table_bits,
                      int chain_length, int distance_bit_cost, uint32_t *res_out, uint8_t *mark_out) {
  std::vector<uint32_t> table((size_t)1 << table_bits, 0), chain((size_t)n + 8, 0);
  build_chain(src, 0, n, hash_len, table_bits, table, chain.data());
  auto window = [&](int p, uint32_t *w) {
    uint8_t b[K2_BW + K2_FW];
    for (int k = 0; k < K2_BW + K2_FW; k++) {
      const int q = p - K2_BW + k;
      b[k] = (q >= 0 && q < n) ? src[q] : 0xA5; // what lies outside the buffer must not matter
    }
    std::memcpy(w, b, sizeof(b));
  };
  for (int i = 0; i < n; i++) {
    K2Res r{};
    int cnt = 0;
    bool fb = false;
    int32_t ch0 = INT32_MIN, ch1 = INT32_MIN;
    if (i < n - 7) {
      uint32_t d = chain[i] & kChainDelta;
      int cand = i;
      uint32_t wi[12];
      window(i, wi);
      for (int j = 0; j <= chain_length && !fb; j++) { // the table entry, then ChainLength hops (m4.go:159,166)
        if (d == 0 || d >= (uint32_t)cand) break;       // none, or position 0 and below (m4.go:152,172)
        cand -= (int)d;
        if (i - cand > max_distance) break;
        uint32_t wc[12], code;
        window(cand, wc);
        const int af = n - i < K2_FW ? n - i : K2_FW, ab = cand < K2_BW ? cand : K2_BW;
        if (k2_pair(wi, wc, af, ab, code) && k2_filter(ch0, ch1, (uint32_t)(i - cand) | code, min_length, distance_bit_cost) &&
            !k2_add(r.e, cnt, (uint32_t)(i - cand) | code, min_length, distance_bit_cost))
          fb = true;
        d = chain[cand] & kChainDelta;
      }
    }
    if (fb) {
      r = K2Res{};
      r.e[0] = kK2Fallback;
    }
    std::memcpy(res_out + 4 * (size_t)i, r.e, 16);
    mark_out[i] = fb || cnt > 0;
  }
}

// One fresh-state FindMatches whose parse takes its chain walks from K2's results (computed as above).
int64_t sim_find_matches_k2(const uint8_t *src, int32_t n, int max_distance, int min_length, int hash_len,
                            int table_bits, int chain_length, int distance_bit_cost, uint32_t seed_prev,
                            uint32_t *out, int64_t out_cap) {
  ParseParams P{max_distance, min_length, hash_len, chain_length, distance_bit_cost};
  std::vector<uint32_t> table((size_t)1 << table_bits, 0), chain((size_t)n + 8, 0);
  build_chain(src, 0, n, hash_len, table_bits, table, chain.data());
  std::vector<K2Res> res((size_t)n + 1);
  std::vector<uint8_t> mark((size_t)n + 1);
  sim_k2_reference(src, n, max_distance, min_length, hash_len, table_bits, chain_length, distance_bit_cost,
                   reinterpret_cast<uint32_t *>(res.data()), mark.data());
  for (int i = 0; i < n; i++) chain[i] = (chain[i] & kChainDelta) | (mark[i] ? kChainMaybe : 0u);
  VecSink sink;
  ParseState S;
  state_init(S, 0, seed_prev);
  GlobalMem M{src, chain.data(), res.data()};
  parse_range<false, true, true, true, true>(M, n, P, S, n - 7, sink);
  parse_finish(n, P, S, sink);
  if ((int64_t)sink.v.size() > out_cap) return -(int64_t)sink.v.size();
  for (auto &r : sink.v) r.lit_start &= ~kRecFlushBit;
  if (!sink.v.empty()) std::memcpy(out, sink.v.data(), sink.v.size() * sizeof(Rec));
  return (S.error & m4g::kStateErrRange) ? INT64_MIN : (int64_t)sink.v.size();
}
// One fresh-state FindMatches whose parse reads the candidates of its walks from the bucket-sorted position list
// (SortEnt, m4_parse.h), built here the way k1d_depth / k2_scan_sizes / k2_place build it.
int64_t sim_find_matches_sw(const uint8_t *src, int32_t n, int max_distance, int min_length, int hash_len,
                            int table_bits, int chain_length, int distance_bit_cost, uint32_t seed_prev,
                            uint32_t *out, int64_t out_cap) {
  ParseParams P{max_distance, min_length, hash_len, chain_length, distance_bit_cost};
  std::vector<uint32_t> table((size_t)1 << table_bits, 0), chain((size_t)n + 8, 0);
  build_chain(src, 0, n, hash_len, table_bits, table, chain.data());
  const int32_t end = n >= 8 ? n - 7 : 0;
  const uint64_t mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
  std::vector<uint32_t> size((size_t)1 << table_bits, 0), rank((size_t)n + 1, 0), hs((size_t)n + 1, 0);
  for (int32_t i = 0; i < end; i++) {
    hs[i] = (uint32_t)(((ld64(src + i) & mask) * 0x1E35A7BD1E35A7BDull) >> (64 - table_bits));
    rank[i] = size[hs[i]]++; // depth
  }
  uint32_t run = 0;
  for (auto &v : size) {
    const uint32_t c = v;
    v = run;
    run += c;
  }
  std::vector<SortEnt> sorted((size_t)n + 1);
  for (int32_t i = 0; i < end; i++) {
    const uint32_t d = rank[i], idx = size[hs[i]] + d, dc = (d < kSortDepthCap ? d : kSortDepthCap) << kSortPosBits;
    SortEnt e;
    e.word = (uint32_t)i | dc;
    uint8_t b[4];
    for (int k = 0; k < 4; k++) b[k] = i - 4 + k >= 0 ? src[i - 4 + k] : (uint8_t)(0x5A + k); // arbitrary before the buffer
    std::memcpy(&e.before4, b, 4);
    e.at8 = ld64(src + i);
    sorted[idx] = e;
    rank[i] = idx | dc;
  }
  VecSink sink;
  ParseState S;
  state_init(S, 0, seed_prev);
  GlobalMem M{src, chain.data(), nullptr, sorted.data(), rank.data()};
  parse_range<false, false, true, true, true>(M, n, P, S, n - 7, sink);
  parse_finish(n, P, S, sink);
  if ((int64_t)sink.v.size() > out_cap) return -(int64_t)sink.v.size();
  for (auto &r : sink.v) r.lit_start &= ~kRecFlushBit;
  if (!sink.v.empty()) std::memcpy(out, sink.v.data(), sink.v.size() * sizeof(Rec));
  return (S.error & m4g::kStateErrRange) ? INT64_MIN : (int64_t)sink.v.size();
}
}