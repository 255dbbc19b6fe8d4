// m4_parse.h -- the serial part of matchfinder.M4 (m4.go:109-283) as a
// resumable state machine over a *precomputed* chain array.
//
// The reference interleaves hash-table insertion (m4.go:140-147, pure function
// of the bytes) with the parse.  Here kernel K1 (k1_chain.h) has already
// produced chain[i] = i - (latest earlier inserted position with the same
// hash) for every inserted position, so the parse
//   * never touches a hash table, and
//   * JUMPS over the positions m4.go:149-151 would `continue` past (inside a
//     pending match only position End+2-HashLen is searched),
// which is what makes it cheap enough to run one state machine per GPU lane.
//
// The state between two loop iterations is (i, m0, m1, next_emit, prev_dist):
// matches[2] of the reference is always empty at the loop top.  The machine
// can be stopped at any position and resumed, which the speculative
// sub-block parse (K3) relies on.
//
// Compiles for the device (hipcc) and, for host-side unit tests of the
// algorithm only, for the host.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define M4G_HD __host__ __device__ __forceinline__
#else
#define M4G_HD inline
#endif

// Counting hooks for tools/analysis/parse_stats.cpp (host analysis only): empty everywhere else.
#ifndef M4G_STAT
#define M4G_STAT(k) ((void)0)
#define M4G_STATN(k, n) ((void)0)
#endif

namespace m4g {

struct K2Res;
struct SortEnt;
struct ParseParams {
  int32_t max_distance;
  int32_t min_length;
  int32_t hash_len;
  int32_t chain_length;
  int32_t distance_bit_cost;
  // results of kernel K2 for the chain array the parse reads (element j of it speaks about chain[j]), or null:
  // set by the host for plans that run K2 (m4_k2.h); the kernels hand job J's part to their memory accessor
  const K2Res *k2 = nullptr;
  // the bucket-sorted position list and every position's index in it (k2_eval.h: k2_place), or null: with them
  // the walk of m4.go:166-181 reads its candidates from consecutive list entries instead of chasing the chain
  const SortEnt *sw = nullptr;
  const uint32_t *rank = nullptr;
};

// emitter.go:5-16; end == 0 <=> the reference's zero-valued absoluteMatch
// (a real match always has End >= 5, a clipped one End >= 1).
struct AbsMatch {
  int32_t start, end, match;
};

// Internal 16-byte match record: m4g_match + where its literals begin.
struct Rec {
  uint32_t unmatched, length, distance, lit_start;
};

struct ParseState {
  int32_t i;          // next loop position
  AbsMatch m0, m1;    // pending matches, newest first (m4.go:111)
  int32_t next_emit;  // matchEmitter.NextEmit (emitter.go:24)
  uint32_t prev_dist; // e.Dst[len-1].Distance, valid iff have_prev
  uint32_t have_prev; // len(e.Dst) > 0
  uint32_t error;     // kStateErrRange | diagnostic flags of the speculative parse (below)
  int32_t first_i;    // loop position at which m0 first became non-empty since state_init (-1: not yet).
                      // Up to and including it the parse probes with the ENTRY state's repeat distance
                      // (m4.go:128-138 runs while nothing is pending); from there until the first emit a
                      // match is pending, and the first emit replaces the distance (m4_spec.h, fast path)
};
static_assert(sizeof(ParseState) == 48, "ParseState is stored in arrays (E, Tout)");
// ParseState.error bits.  Only kStateErrRange is an error (the Go code would
// panic); the others tell the speculative sub-block parse (m4_spec.h) what a
// result depended on.
constexpr uint32_t kStateErrRange = 1;   // repeat distance reaches before the buffer start
constexpr uint32_t kStateHitLo = 4;      // before anything was emitted, a match started exactly at next_emit:
                                         // its backward extension may have been cut short by that bound
constexpr uint32_t kStateTouchedEnd = 8; // a match reached `len` (the caller may have passed a shortened len)
constexpr uint32_t kStateAborted = 16;   // set by the kernels: the parse of this block was given up

M4G_HD uint32_t ld32(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
M4G_HD uint64_t ld64(const uint8_t *p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

// Memory accessor of the parse: every read of the input bytes and of the chain
// array goes through one of these, so kernels can substitute a staged (LDS)
// view for the positions a lane is currently walking (kernels.h, StagedMem).
// Results of kernel K2 (m4_k2.h explains them)
constexpr int K2_FW = 32;      // bytes compared forward (from the position)
constexpr int K2_BW = 16;      // bytes compared backward (before the position)
constexpr int K2_ENTRIES = 4;  // candidates kept per position
// One kept candidate = one word: bits 0..21 distance, 22..26 forward length - 4 (28: K2_FW bytes agree, the
// match may go on), 27..31 backward length (16: K2_BW bytes agree, it may go on).  0: no entry.
constexpr uint32_t kK2DistBits = 22, kK2DistMask = (1u << kK2DistBits) - 1;
constexpr uint32_t kK2FwdCapped = K2_FW - 4, kK2BwdCapped = K2_BW;
constexpr uint32_t kK2Fallback = 0xFFFFFFFFu; // in e[0]: more than K2_ENTRIES candidates matter, walk the chain
constexpr int32_t kK2MaxLen = 1 << kK2DistBits; // chunks shorter than this can use K2
struct K2Res {
  uint32_t e[K2_ENTRIES];
};
static_assert(sizeof(K2Res) == 16, "one 16-byte word per position");
// One entry of a job's bucket-sorted position list (k2_eval.h): the positions with one table index stand next
// to each other, ascending, so the candidates of the walk from the entry at index r are the entries r-1, r-2, ...
// An entry carries the 4 bytes before and the 8 bytes at its position: most candidates are decided from those
// (m4.go:175-176: the 4-byte test, and every match shorter than 8 forward / 4 backward) without touching the
// candidate's own cache line.
constexpr uint32_t kSortPosBits = 22, kSortPosMask = (1u << kSortPosBits) - 1; // word: position | min(depth, 127) << 22
constexpr uint32_t kSortDepthCap = 127;                                         // depth = earlier positions in the bucket
struct SortEnt {
  uint32_t word;    // position | min(depth, kSortDepthCap) << kSortPosBits
  uint32_t before4; // bytes [p-4, p) (little endian; what lies before the buffer is arbitrary)
  uint64_t at8;     // bytes [p, p+8)
};
static_assert(sizeof(SortEnt) == 16, "one 16-byte word per position");
struct GlobalMem {
  const uint8_t *src;
  const uint32_t *chain;
  const K2Res *k2 = nullptr; // per-position results of K2 (m4_k2.h), or none
  const SortEnt *sw = nullptr; // the job's sorted position list, or none
  const uint32_t *rank = nullptr; // per position: index in sw | min(depth, 127) << 22
  M4G_HD bool has_sw() const { return sw != nullptr; }
  M4G_HD const SortEnt *sw_ptr(int32_t r) const { return sw + r; }
  M4G_HD uint32_t rank_at(int32_t p) const { return rank[p]; }
  M4G_HD uint8_t ld8(int32_t p) const { return src[p]; }
  M4G_HD uint32_t ld32(int32_t p) const { return m4g::ld32(src + p); }
  M4G_HD uint64_t ld64(int32_t p) const { return m4g::ld64(src + p); }
  // read of a match source: far from the positions being walked
  M4G_HD uint64_t far64(int32_t p) const { return m4g::ld64(src + p); }
  M4G_HD uint32_t chain_at(int32_t p) const { return chain[p]; }
  // reads the light filter of parse_range makes: positions the caller guarantees to be "near"
  M4G_HD uint32_t near_chain(int32_t p) const { return chain[p]; }
  M4G_HD uint32_t near32(int32_t p) const { return m4g::ld32(src + p); }
  M4G_HD bool has_k2() const { return k2 != nullptr; }
  M4G_HD const K2Res *k2_ptr(int32_t p) const { return k2 + p; }
};

M4G_HD int32_t bitlen32(uint32_t x) { return x ? 32 - __builtin_clz(x) : 0; }

// m4.go:59-61 (64-bit: lengths reach 2^25 in streaming mode)
M4G_HD int64_t score(const AbsMatch &m, int32_t cost) {
  return (int64_t)(m.end - m.start) * 256 -
         (int64_t)bitlen32((uint32_t)(m.start - m.match)) * cost;
}

// m4.go:293-324: first index >= j where src[i+k] != src[j+k], or len.
template <class Mem>
M4G_HD int32_t extend_fwd(const Mem &M, int32_t len, int32_t i, int32_t j) {
  M4G_STAT(21);
  while (j + 8 < len) {
    M4G_STAT(22);
    uint64_t a = M.ld64(i), b = M.ld64(j);
    if (a != b) return j + (int32_t)(__builtin_ctzll(a ^ b) >> 3);
    i += 8;
    j += 8;
  }
  while (j < len && M.ld8(i) == M.ld8(j)) {
    i++;
    j++;
  }
  return j;
}

// Backward half of m4.go:330-333 beyond the first bytes: the largest n in [nb, maxb] such that the n
// bytes before pos equal the n bytes before cand (nb of them are known to).
template <class Mem>
M4G_HD int32_t extend_bwd(const Mem &M, int32_t pos, int32_t cand, int32_t nb, int32_t maxb) {
  // 8 bytes per step while 8 more fit (maxb <= cand and maxb <= pos: the reads stay inside the buffer);
  // a match source is far memory, so every step is a round trip -- bytewise this loop was what made
  // long back-extensions (repeated paragraphs) crawl
  M4G_STAT(23);
  while (nb + 8 <= maxb) {
    const uint64_t x = M.ld64(pos - nb - 8) ^ M.ld64(cand - nb - 8);
    if (x != 0) return nb + (int32_t)(__builtin_clzll(x) >> 3);
    nb += 8;
  }
  while (nb < maxb && M.ld8(pos - nb - 1) == M.ld8(cand - nb - 1)) nb++;
  return nb;
}

// m4.go:328-339
template <class Mem>
M4G_HD AbsMatch extend2(const Mem &M, int32_t len, int32_t start, int32_t cand, int32_t lo) {
  AbsMatch m;
  m.end = extend_fwd(M, len, cand + 4, start + 4);
  while (start > lo && cand > 0 && M.ld8(start - 1) == M.ld8(cand - 1)) {
    start--;
    cand--;
  }
  m.start = start;
  m.match = cand;
  return m;
}

// 4-byte equality test (m4.go:131,159,175) and extendMatch2 in one step.
// The reference does them as a chain of small dependent reads; on the GPU every
// read of a match SOURCE is a global-memory round trip, so the common case is
// served by two independent 8-byte reads of the source (at cand and cand-8):
// they decide the equality test, the first 8 bytes of the forward extension
// and up to 8 bytes of the backward extension.  The result is the same
// (start, end, match) extendMatch2 computes -- extension is plain byte
// comparison -- only the read pattern differs.
template <class Mem>
M4G_HD bool match_at(const Mem &M, int32_t len, int32_t pos, int32_t cand, int32_t lo,
                     AbsMatch &m) {
  if (cand >= 8 && pos + 8 <= len) {
    const uint64_t cf = M.ld64(cand), cb = M.ld64(cand - 8);
    const uint64_t x = cf ^ M.ld64(pos);
    if ((uint32_t)x != 0) return false;
    m.end = x != 0 ? pos + (int32_t)(__builtin_ctzll(x) >> 3)
                   : extend_fwd(M, len, cand + 8, pos + 8);
    // backward: at most pos-lo steps (start > lo) and at most cand steps (candidate > 0)
    const int32_t maxb = (pos - lo) < cand ? (pos - lo) : cand;
    int32_t nb = 0;
    if (maxb > 0) {
      const uint64_t y = cb ^ M.ld64(pos - 8);
      nb = y != 0 ? (int32_t)(__builtin_clzll(y) >> 3) : 8;
      if (nb > maxb) nb = maxb;
      if (nb == 8) nb = extend_bwd(M, pos, cand, nb, maxb);
    }
    m.start = pos - nb;
    m.match = cand - nb;
    return true;
  }
  if (M.ld32(cand) != M.ld32(pos)) return false;
  m = extend2(M, len, pos, cand, lo);
  return true;
}

// Chain entry layout: bits 0..30 = distance to the previous position with the
// same hash (0: none); bit 31 = "the 4 bytes there MAY equal the 4 bytes here".
// K1 keeps a few check bits of every position's first 4 bytes next to the
// position in its hash table; when they differ the parse can skip the random
// read of the candidate (m4.go:159) -- on text that is ~95 % of the literal
// positions.  A set bit promises nothing, a clear bit is definite.
constexpr uint32_t kChainMaybe = 0x80000000u;
constexpr uint32_t kChainDelta = 0x7FFFFFFFu;

// matchEmitter.emit (emitter.go:27-34) into a caller-provided sink.
// Bit 31 of Rec.lit_start marks a record emitted as the newest pending match
// by a loop-top flush (m4.go:124): right after it the machine state is fully
// determined by (record end, record distance) -- a synchronisation point for
// the speculative sub-block parse (m4_spec.h).
constexpr uint32_t kRecFlushBit = 0x80000000u;

template <class Sink>
M4G_HD void emit(ParseState &S, const AbsMatch &m, Sink &sink, uint32_t flag = 0) {
  Rec r;
  r.unmatched = (uint32_t)(m.start - S.next_emit);
  r.length = (uint32_t)(m.end - m.start);
  r.distance = (uint32_t)(m.start - m.match);
  r.lit_start = (uint32_t)S.next_emit | flag;
  sink.push(r);
  S.next_emit = m.end;
  S.prev_dist = r.distance;
  S.have_prev = 1;
}

// m4.go:113-126 / 262-273: emit the (clipped) older match, then the newest.
template <class Sink>
M4G_HD void flush(ParseState &S, const ParseParams &P, Sink &sink) {
  if (S.m1.end != 0) {
    if (S.m1.end > S.m0.start) S.m1.end = S.m0.start;
    if (S.m1.end - S.m1.start >= P.min_length && score(S.m1, P.distance_bit_cost) > 0)
      emit(S, S.m1, sink);
  }
  emit(S, S.m0, sink, kRecFlushBit);
  S.m0 = AbsMatch{0, 0, 0};
  S.m1 = AbsMatch{0, 0, 0};
}

M4G_HD void state_init(ParseState &S, int32_t start, uint32_t seed_prev_distance) {
  S.i = start;
  S.m0 = AbsMatch{0, 0, 0};
  S.m1 = AbsMatch{0, 0, 0};
  S.next_emit = start;
  S.prev_dist = seed_prev_distance;
  S.have_prev = seed_prev_distance != 0;
  S.error = 0;
  S.first_i = -1;
}

// Candidate evaluation from bytes that are already in registers: cf/cf2 = the
// 16 bytes at cand, cb = the 8 bytes before cand.  Same result as match_at.
template <class Mem>
M4G_HD bool match_regs(const Mem &M, int32_t len, int32_t pos, int32_t cand, int32_t lo,
                       uint64_t cf, uint64_t cf2, bool have2, uint64_t cb, AbsMatch &m) {
  const uint64_t x = cf ^ M.ld64(pos);
  if ((uint32_t)x != 0) return false;
  if (x != 0) {
    m.end = pos + (int32_t)(__builtin_ctzll(x) >> 3);
  } else if (have2) {
    const uint64_t x2 = cf2 ^ M.ld64(pos + 8);
    m.end = x2 != 0 ? pos + 8 + (int32_t)(__builtin_ctzll(x2) >> 3)
                    : extend_fwd(M, len, cand + 16, pos + 16);
  } else {
    m.end = extend_fwd(M, len, cand + 8, pos + 8);
  }
  const int32_t maxb = (pos - lo) < cand ? (pos - lo) : cand;
  int32_t nb = 0;
  if (maxb > 0) {
    const uint64_t y = cb ^ M.ld64(pos - 8);
    nb = y != 0 ? (int32_t)(__builtin_clzll(y) >> 3) : 8;
    if (nb > maxb) nb = maxb;
    if (nb == 8) nb = extend_bwd(M, pos, cand, nb, maxb);
  }
  m.start = pos - nb;
  m.match = cand - nb;
  return true;
}

// Next loop position after position i has been handled: inside a pending
// match the reference only inserts hashes (K1 did that) except at position
// End+2-HashLen, which is searched (m4.go:149-151); so jump there, or to End.
M4G_HD int32_t next_pos(const ParseState &S, const ParseParams &P, int32_t i) {
  if (S.m0.end == 0) return i + 1;
  const int32_t probe = S.m0.end + 2 - P.hash_len;
  return (probe > i && probe < S.m0.end) ? probe : S.m0.end;
}

} // namespace m4g
#include "m4_k2.h"
namespace m4g {

// Runs loop iterations of m4.go:112-260 while S.i < stop (stop <= len-7).
// M reads the input bytes / chain array by buffer position; `len` is the
// buffer length (forward extension may read up to len, never past it).
// With STOP_AT_FLUSH the function returns true right after a loop-top flush
// (state saved at that position, before the repeat probe); it returns false
// when the range is exhausted.
//
// Loop invariant at the top: no match pending, or i == m0.End (flush), or
// i == m0.End+2-HashLen (the one searched position inside a pending match).
//
// GPU note: every lane of a wave runs its own copy of this loop, and divergent
// branches execute one after the other.  A global-memory read inside a
// divergent branch therefore exposes its full latency to the whole wave.  So
// all reads of match SOURCES that an iteration can need (the repeat-probe
// bytes and the 24 bytes around the hash candidate) are issued together,
// unconditionally (lanes that do not need them read position 0), before any
// branch; the branches then work on registers.  Only rare continuations
// (matches longer than 16 bytes, chain hops) read memory inside a branch.
//
// LIGHT: most iterations do nothing but advance i -- a literal position whose table entry
// cannot match (no entry, too far, check bits differ: m4.go:152-163 fall through to :183) and
// whose repeat probe fails on bytes the lane already holds, or the one searched position inside
// a pending match with such an entry.  With LIGHT every lane first runs through those in a loop
// of a few instructions and only enters the full iteration below at a position where something
// can happen (a flush is due, the entry may match, the probe needs bytes or hits).  The lanes of
// a wave then execute the long body once per EVENT instead of once per position.  The skipped
// iterations change no state but i, so the result is the same by construction.  Mem must serve
// near_chain(i) and near32(i + 1) for i < stop without a range check.
//
// CHAIN = false compiles the chain walk (m4.go:166-181) and the re-search of a trimmed match
// (m4.go:235-253) out -- for callers that know ChainLength == 0; FUSED = false reads a candidate's
// bytes where they are used (match_at) instead of all together at the top of the iteration: less
// code and fewer registers for callers that run a single, rare iteration (k3a_fast's parked lanes).
// XW = true (with CHAIN) compiles in the two other sources of a walk's candidates: K2's per-position results
// (m4_k2.h; Mem::has_k2()) and the bucket-sorted position list (SortEnt; Mem::has_sw()), chosen at run time by
// what the accessor carries.  The kernels of the default path are instantiated without them: the extra state
// cost k3a_spec 20 % under ChainLength 64 (160 spilled registers) even when unused.
template <bool STOP_AT_FLUSH = false, bool LIGHT = false, bool CHAIN = true, bool FUSED = true, bool XW = false, class Mem, class Sink>
M4G_HD bool parse_range(const Mem &M, int32_t len, const ParseParams &P, ParseState &S,
                        int32_t stop, Sink &sink, int32_t budget = 0x7FFFFFFF /* rounds of the loop, at most (a walk
                        in progress is finished first); the caller goes on from S.i */) {
  int32_t i = S.i;
  bool k2_on = false; // the walks of m4.go:156-181 come from K2 (m4_k2.h)
  if constexpr (CHAIN && XW) k2_on = P.chain_length > 0 && M.has_k2();
  bool sw_on = false; // the hops of m4.go:166-181 read the sorted position list (SortEnt)
  if constexpr (CHAIN && XW) sw_on = P.chain_length > 0 && !k2_on && M.has_sw();
  // 8 source bytes of the repeat probe, kept across iterations: consecutive
  // literal positions probe consecutive addresses (same prev_dist).
  uint64_t pc_bytes = 0;
  int32_t pc_base = -0x40000000;
  // The walk over the sorted list is SUSPENDED between batches of four candidates: a lane inside a walk takes one
  // batch per round of this loop while the other lanes of its wave go on with their positions.  (Run to the end
  // where it starts, every position of every lane costs the wave the longest walk among its 64 lanes -- measured:
  // 4 % of the lanes active in the hop loop, and no gain over chasing the chain.)
  bool walking = false;
  int32_t w_next = 0, w_left = 0; // the next entry to read (they are read downwards), candidates left
  AbsMatch cur{0, 0, 0};
  while (i < stop) {
   if (!(CHAIN && XW && walking) && budget <= 0) break;
   budget--;
   if (CHAIN && XW && walking) {
    if constexpr (CHAIN && XW) {
      int64_t cur_score = score(cur, P.distance_bit_cost); // (0 for the empty match, as at the start of the walk)
      AbsMatch m;
      const uint64_t my8 = M.ld64(i);
      const uint32_t myb = i >= 4 ? M.ld32(i - 4) : (M.ld32(0) << (8 * (4 - i))); // (i >= 1: position 0 has no candidate)
      // Four entries per round trip; an entry decides its candidate unless the match reaches 8 bytes forward or 4
      // (usable) bytes backward -- then the candidate's bytes are read as before (match_at).  Same candidates in the
      // same order as the chain gives, same comparisons.
      SortEnt ent[4];
#if defined(__HIPCC__)
#pragma unroll
#endif
      for (int t = 0; t < 4; t++) ent[t] = *M.sw_ptr(w_next - (t < w_left ? t : 0));
      const int32_t nb4 = w_left < 4 ? w_left : 4;
      w_next -= 4;
      w_left -= nb4;
#if defined(__HIPCC__)
#pragma unroll
#endif
      for (int t = 0; t < 4; t++) {
        if (t >= nb4) break;
        M4G_STAT(24);
        const int32_t c2 = (int32_t)(ent[t].word & kSortPosMask);
        if (c2 <= 0 || i - c2 > P.max_distance) { // position 0 reads as "none"; beyond MaxDistance the walk ends
          w_left = 0;
          break;
        }
        const uint64_t x = ent[t].at8 ^ my8;
        if ((uint32_t)x != 0) continue; // m4.go:175
        bool hop_hit = true;
        const int32_t maxb = (i - S.next_emit) < c2 ? (i - S.next_emit) : c2;
        int32_t nb = 0;
        bool real = x == 0;
        if (!real && maxb > 0) {
          const uint32_t y = ent[t].before4 ^ myb;
          nb = y != 0 ? (int32_t)(__builtin_clz(y) >> 3) : 4;
          if (nb > maxb) nb = maxb;
          else if (nb == 4 && maxb > 4) real = true;
        }
        if (real) {
          M4G_STAT(31);
          hop_hit = match_at(M, len, i, c2, S.next_emit, m);
        } else {
          m.end = i + (int32_t)(__builtin_ctzll(x) >> 3);
          m.start = i - nb;
          m.match = c2 - nb;
        }
        if (hop_hit) {
          M4G_STAT(25);
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
      if (w_left > 0) continue;
      walking = false; // ... and on to m4.go:183 with `cur`
    }
   } else {
    if (LIGHT) {
      bool event = false;
      do {
        if (S.m0.end != 0 && i >= S.m0.end) { // a flush is due
          event = true;
          break;
        }
        const uint32_t le = M.near_chain(i);
        // (with K2's results the mark of an entry says "some candidate of the walk can be accepted")
        if ((le & kChainDelta) - 1u < (uint32_t)P.max_distance &&
            ((le & kChainMaybe) || (P.chain_length > 0 && !(CHAIN && XW && k2_on)))) {
          event = true; // a candidate to look at
          break;
        }
        if (S.m0.end == 0 && S.have_prev) {
          const int32_t lc = i + 1 - (int32_t)S.prev_dist;
          const uint32_t lo = (uint32_t)(lc - pc_base);
          if (lc < 0 || lo > 4 || (uint32_t)(pc_bytes >> (8 * lo)) == M.near32(i + 1)) {
            event = true; // the probe needs bytes, hits, or is out of range
            break;
          }
        }
        M4G_STAT(29);
        i = next_pos(S, P, i);
      } while (i < stop);
      if (!event) break;
    }
    M4G_STAT(0);
    // A. m4.go:113-126
    if (S.m0.end != 0 && i >= S.m0.end) {
      M4G_STAT(1);
      flush(S, P, sink);
      if (STOP_AT_FLUSH) {
        S.i = i;
        return true;
      }
    }
    const bool was_empty = S.m0.end == 0;
    if (was_empty) M4G_STAT(2); else M4G_STAT(26);

    // ---- what this iteration may need from far memory ----------------------
    const int32_t p = i + 1;
    const int32_t c = p - (int32_t)S.prev_dist;
    const bool do_probe = was_empty && S.have_prev && c >= 0;
    const bool probe_err = was_empty && S.have_prev && c < 0;
    const uint32_t pc_off = (uint32_t)(c - pc_base);
    const bool probe_load = FUSED && do_probe && pc_off > 4 && c + 8 <= len;

    const uint32_t ce = M.chain_at(i);
    const uint32_t d = ce & kChainDelta;
    const bool cand_ok = d != 0 && d <= (uint32_t)P.max_distance;
    int32_t cand = i - (int32_t)d;
    const bool cand_fast = FUSED && cand_ok && (ce & kChainMaybe) && cand >= 8 && !k2_on; // i+8 <= len always
    const bool cand_fast2 = cand_fast && i + 16 <= len;

    uint64_t pl = 0, cf = 0, cb = 0, cf2 = 0;
    if (FUSED) {
      pl = M.far64(probe_load ? c : 0);
      cf = M.far64(cand_fast ? cand : 0);
      cb = M.far64(cand_fast ? cand - 8 : 0);
      cf2 = M.far64(cand_fast2 ? cand + 8 : 0);
    }
    uint32_t rk = 0;
    if constexpr (CHAIN && XW) {
      if (sw_on && cand_ok) rk = M.rank_at(i); // (with the other far words of the iteration)
    }
    K2Res k2r{};
    if constexpr (CHAIN && XW) {
      // (read with the other far words of the iteration; positions without an accepted candidate have no mark)
      if (k2_on && cand_ok && (ce & kChainMaybe)) __builtin_memcpy(&k2r, M.k2_ptr(i), sizeof(K2Res));
    }

    // B. m4.go:128-138 repeat match one byte after the current position
    if (probe_err) S.error |= kStateErrRange; // the Go code would panic on the slice bounds
    if (probe_load) M4G_STAT(4);
    if (cand_fast) M4G_STAT(27);
    if (do_probe) {
      M4G_STAT(3);
      uint32_t cw;
      if (pc_off <= 4) {
        cw = (uint32_t)(pc_bytes >> (8 * pc_off));
      } else if (probe_load) {
        pc_bytes = pl;
        pc_base = c;
        cw = (uint32_t)pl;
      } else {
        cw = M.ld32(c);
      }
      AbsMatch m;
      if (cw == M.ld32(p) && match_at(M, len, p, c, S.next_emit + 1, m)) {
        M4G_STAT(5);
        if (m.end >= len) S.error |= kStateTouchedEnd;
        if (m.end - m.start >= P.min_length) {
          M4G_STAT(6);
          S.m0 = m;
          if (S.first_i < 0) S.first_i = i;
        }
      }
    }

    // D. m4.go:149-151: a match set just now makes i an interior position
    if (was_empty && S.m0.end != 0 && i != S.m0.end + 2 - P.hash_len) {
      M4G_STAT(7);
      i = next_pos(S, P, i);
      continue;
    }
    // m4.go:152-154
    if (!cand_ok) {
      M4G_STAT(8);
      i = next_pos(S, P, i);
      continue;
    }

    // E. m4.go:156-181
    cur = AbsMatch{0, 0, 0};
    int64_t cur_score = 0;
    AbsMatch m;
    bool walk = true;
    if constexpr (CHAIN && XW) {
      if (k2_on) {
        walk = (ce & kChainMaybe) && k2r.e[0] == kK2Fallback;
        if ((ce & kChainMaybe) && !walk) {
          M4G_STAT(30);
          k2_select(M, len, P, S, i, k2r, cur, cur_score);
        }
      }
    }
    if (walk && (ce & kChainMaybe)) {
      M4G_STAT(9);
      const bool hit = cand_fast ? match_regs(M, len, i, cand, S.next_emit, cf, cf2, cand_fast2, cb, m)
                                 : match_at(M, len, i, cand, S.next_emit, m);
      if (hit) {
        M4G_STAT(10);
        if (!S.have_prev && m.start == S.next_emit) S.error |= kStateHitLo;
        if (m.end >= len) S.error |= kStateTouchedEnd;
      }
      if (hit && m.end - m.start > P.min_length) {
        int64_t sc = score(m, P.distance_bit_cost);
        if (sc > 0) {
          M4G_STAT(11);
          cur = m;
          cur_score = sc;
        }
      }
    }
    // m4.go:166-181.  The chain entry of a candidate and the bytes around it are
    // read together (one far round trip per hop instead of two dependent ones).
    if constexpr (CHAIN && XW) {
      if (sw_on) {
        // The hops come from the sorted list, in later rounds of this loop (see its top): candidate j of the walk
        // (0 = the table entry, handled above) is the entry j + 1 below this position's own.
        walk = false;
        const int32_t depth = (int32_t)(rk >> kSortPosBits);
        w_left = (depth < P.chain_length + 1 ? depth : P.chain_length + 1) - 1;
        w_next = (int32_t)(rk & kSortPosMask) - 2;
        if (w_left > 0) {
          walking = true;
          continue;
        }
      }
    }
    uint32_t next_entry = CHAIN && walk && P.chain_length > 0 ? M.chain_at(cand) : 0u;
    for (int32_t j = 0; CHAIN && walk && j < P.chain_length; j++) {
      uint32_t dd = next_entry & kChainDelta;
      if (dd == 0) break;
      M4G_STAT(24);
      // Go: candidate -= int(delta) in 64-bit; dd > cand means candidate <= 0
      if (dd >= (uint32_t)cand) break;
      cand -= (int32_t)dd;
      if (i - cand > P.max_distance) break;
      bool hop_hit;
      if (cand >= 8) { // i + 8 <= len always
        const bool two = i + 16 <= len;
        next_entry = M.chain_at(cand);
        const uint64_t hf = M.far64(cand), hb = M.far64(cand - 8), hf2 = M.far64(two ? cand + 8 : cand);
        hop_hit = match_regs(M, len, i, cand, S.next_emit, hf, hf2, two, hb, m);
      } else {
        next_entry = M.chain_at(cand);
        hop_hit = match_at(M, len, i, cand, S.next_emit, m);
      }
      if (hop_hit) {
        M4G_STAT(25);
        if (!S.have_prev && m.start == S.next_emit) S.error |= kStateHitLo;
        if (m.end >= len) S.error |= kStateTouchedEnd;
        if (m.end - m.start > P.min_length) {
          int64_t sc = score(m, P.distance_bit_cost);
          if (sc > cur_score) {
            cur = m;
            cur_score = sc;
          }
        }
      }
    }
   } // (!walking)
    const int64_t cur_score = score(cur, P.distance_bit_cost); // (= the walk's running score; 0 for the empty match)
    // m4.go:183-185
    if (cur.end - cur.start < P.min_length) {
      M4G_STAT(12);
      i = next_pos(S, P, i);
      continue;
    }

    // F. m4.go:187-199
    M4G_STAT(13);
    int64_t penalty = 0;
    if (S.m0.end != 0) {
      penalty = 275;
      if (cur.start <= S.m1.end) penalty = 0;
    }
    if (cur_score <= score(S.m0, P.distance_bit_cost) + penalty) {
      M4G_STAT(14);
      i = next_pos(S, P, i);
      continue;
    }

    // m4.go:201-209
    M4G_STAT(15);
    AbsMatch m2 = S.m1;
    S.m1 = S.m0;
    S.m0 = cur;
    if (S.first_i < 0) S.first_i = i;
    if (m2.end != 0) {
      // G. m4.go:211-259
      M4G_STAT(16);
      if (S.m0.start < m2.end) {
        M4G_STAT(17);
        S.m1 = m2;
      } else if (S.m0.start < m2.end + P.min_length) {
        M4G_STAT(18);
        emit(S, m2, sink);
        S.m1 = AbsMatch{0, 0, 0};
      } else {
        M4G_STAT(19);
        if (m2.end > S.m1.start) {
          M4G_STAT(20);
          m2.end = S.m1.start;
          int32_t mlen = m2.end - m2.start;
          if (CHAIN && P.chain_length > 0 && mlen >= P.min_length) {
            // m4.go:236-252: look for a closer source for the trimmed match
            int32_t pos = m2.start;
            for (;;) {
              uint32_t dd = M.chain_at(pos) & kChainDelta;
              if (dd == 0) break;
              if ((int64_t)pos - (int64_t)dd <= (int64_t)m2.match) break;
              pos -= (int32_t)dd;
              int32_t k = 0;
              while (k < mlen && M.ld8(m2.start + k) == M.ld8(pos + k)) k++;
              if (k == mlen) {
                m2.match = pos;
                break;
              }
            }
          }
        }
        if (m2.end - m2.start >= P.min_length && score(m2, P.distance_bit_cost) > 0)
          emit(S, m2, sink);
      }
    }
    i = next_pos(S, P, i);
  }
  S.i = i;
  return false;
}

// m4.go:262-281: after the loop, emit what is pending plus the trailing
// literal run.
template <class Sink>
M4G_HD void parse_finish(int32_t len, const ParseParams &P, ParseState &S, Sink &sink) {
  if (S.m1.end != 0) {
    if (S.m1.end > S.m0.start) S.m1.end = S.m0.start;
    if (S.m1.end - S.m1.start >= P.min_length && score(S.m1, P.distance_bit_cost) > 0)
      emit(S, S.m1, sink);
  }
  if (S.m0.end != 0) emit(S, S.m0, sink);
  if (S.next_emit < len) {
    Rec r;
    r.unmatched = (uint32_t)(len - S.next_emit);
    r.length = 0;
    r.distance = 0;
    r.lit_start = (uint32_t)S.next_emit;
    sink.push(r);
  }
}

} // namespace m4g
