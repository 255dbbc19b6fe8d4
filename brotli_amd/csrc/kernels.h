// kernels.h -- gfx950 kernels of libm4g (included by m4g.hip only).
//
//   K1  k1_chain_build   hash every inserted position, build the chain array
//                        (m4.go:140-147) with the hash table held in LDS
//   K3  k3_parse_*       the M4 parse (m4.go:109-283) over the chain array
//   K4  k4_*             prefix sum of per-chunk match counts + compaction of
//                        the per-chunk lists into one m4g_match array
//   K5  k5_histograms    pass 1 of brotli.Encoder.Encode (encoder.go:44-105)
//   KS  ks_*             streaming-mode helpers (history slide, m4.go:82-99)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "m4_parse.h"
#include "m4_spec.h"
#include "m4_fast.h"

namespace m4g {

// One unit of work: a buffer region to match-find.  Batch mode: one job per
// chunk (start = 0).  Streaming mode: one job, the handle's history buffer,
// start = length of the history before this block was appended.
struct Job {
  uint64_t src_off;   // byte offset of buffer position 0 in the src base
  uint64_t chain_off; // element offset of position 0 in the chain base
  uint32_t start;     // first position to insert / parse (e.NextEmit at entry)
  uint32_t len;       // buffer length (positions [start, len-7) are inserted)
  uint32_t seed_prev; // Distance of the caller's last dst element, 0 = none
  uint32_t store_from; // K1 only: positions in [start, store_from) are inserted but get no chain entry
                       // (table priming of a stream segment); 0 otherwise
  uint32_t k1_flags;   // K1 only, with a persistent table: kJobLoadTable | kJobStoreTable
  uint32_t pad;
};
constexpr uint32_t kJobLoadTable = 1, kJobStoreTable = 2;

// Block structure of one stream handled in a single launch ("window primed",
// SURVEY.md section 8f.1): what a sequence of FindMatches calls on consecutive
// blocks does to the hash table, restated for workgroups that each build the
// chain of one segment of the stream.
//   * the last 7 positions of every block are never inserted (m4.go:112);
//   * a history slide (m4.go:82-99) turns every table entry at or below the new
//     buffer origin into "empty"; origins only grow, so it is enough to test a
//     predecessor against the origin in force at the position that reads it.
// Positions are buffer positions at the time of the call (after its first
// slide, if any).
constexpr int kStreamMaxSlides = 4096;
struct StreamInfo {
  uint32_t x0;          // first position of the call's data (= start of block 0)
  uint32_t x_end;       // end of the call's data
  uint32_t block_size;  // >= K1_SUB + 7: at most one block end per tile
  uint32_t nbp;         // slide breakpoints in use (>= 1; entry 0 = {0, 0})
  uint32_t final_org;   // origin after the call: table entries are stored relative to it
  uint32_t bp_x[kStreamMaxSlides];   // first position (block start) governed by bp_org[k]
  uint32_t bp_org[kStreamMaxSlides]; // buffer origin from there on (ascending)
};

constexpr uint64_t kHashMul64 = 0x1E35A7BD1E35A7BDull; // m4.go:285

// ---------------------------------------------------------------------------
// K1: chain build.
//
// chain[i] = i - table[h(i)] (0 when the table entry is empty), table[h(i)] = i,
// for i = start .. len-8 in order.  "table[h] before the insertion of i" is
// the largest already inserted position with the same hash, i.e. a pure
// function of the bytes, so the sequential table of the reference becomes:
//   * the table lives in LDS, split by the top hash bits into `nparts`
//     partitions of 2^part_bits entries (2^15 * 4 B = 128 KiB of the 160 KiB
//     LDS); one workgroup owns one (job, partition) and streams the whole
//     buffer region, so the 4 partition workgroups of a job are placed on one
//     XCD (blockIdx % 8 equal) and share its L2 for the input bytes;
//   * the input is staged through LDS in tiles of K1_SUB positions with
//     coalesced 16-byte loads (the next tile is in flight while the current
//     one is processed); every thread forms its unaligned 8-byte hash windows
//     from LDS;
//   * PRODUCE: every wave hashes 64*K1_PER positions; the ~1/nparts of them
//     that fall into this partition are routed by the next 4 hash bits to one
//     of 16 OWNER waves (one LDS queue each, appended with ds_add_rtn).
//     Positions that continue a run of equal hashes (zero pages) are not
//     queued: their chain entry is 1 and only the run head is queued, with
//     the run length in a side array;
//   * CONSUME: wave c owns table slice c and is the only one that touches it.
//     It sorts its <= 128 queue keys (hash index | position) with a bitonic
//     network across its lanes.  After the sort equal hashes are adjacent and
//     in position order, so a key's predecessor is its left neighbour when
//     the hash is equal and the table entry otherwise; the last key of each
//     hash updates the table.  Any multiplicity costs the same, and the
//     workgroup needs two barriers per tile;
//   * results go back to position order through an LDS array so that the
//     chain stores of a wave fall into a few cache lines;
//   * a tile that sends more than K1_QCAP keys to one owner (short-period
//     data) is redone in slices of 128 positions, which cannot overflow.
// A table value is  V(p) = p << cb | chk(p):  the position in the high bits
// and cb check bits of the 4 bytes at p.
// ---------------------------------------------------------------------------
constexpr int K1_THREADS = 1024;
constexpr int K1_WAVES = K1_THREADS / 64;
constexpr int K1_PER = 3;                              // positions per thread per tile
constexpr int K1_SUB = K1_THREADS * K1_PER;            // positions per tile
constexpr int K1_GROUPS = K1_SUB / 16 + 2;             // 16-byte groups staged per tile (+halo)
constexpr int K1_TILE_BYTES = K1_GROUPS * 16;
constexpr int K1_QCAP = 128;                           // keys per owner queue (2 per lane)
constexpr int K1_MAX_CHECK_BITS = 5;
// key = index within the owner's slice << 21 | position in tile << 9 | run flag | check bits
constexpr int K1_KEY_IDX_SHIFT = 21;
constexpr int K1_KEY_REL_SHIFT = 9;
constexpr uint32_t K1_KEY_REL_MASK = 0xFFFu;
constexpr uint32_t K1_KEY_RUN = 0x20u;
constexpr uint32_t K1_KEY_INVALID = 0xFFFFFFFFu;       // position 4095 >= K1_SUB: never a real key
static_assert(K1_SUB <= 4095, "tile position must fit the key");
static_assert(K1_GROUPS <= K1_THREADS, "one 16-byte group per thread");

// dynamic LDS of k1_chain_build (host and device agree through this)
constexpr size_t k1_lds_bytes(int part_bits) {
  return ((size_t)1 << part_bits) * 4 + (size_t)K1_TILE_BYTES + (size_t)K1_WAVES * K1_QCAP * 4 +
         (size_t)K1_SUB * 4 + (size_t)K1_SUB + 2 * K1_WAVES * 4 + (2 + K1_PER) * 4 + 12;
}

// 8 bytes at byte offset `o` of an LDS byte image (3 aligned dword reads +
// two funnel shifts; consecutive lanes read consecutive bytes, so the dword
// reads broadcast and are conflict free).
__device__ __forceinline__ uint2 lds_ld64(const uint32_t *img, uint32_t o) {
  const uint32_t w = o >> 2, sh = o & 3;
  const uint32_t d0 = img[w], d1 = img[w + 1], d2 = img[w + 2];
  return make_uint2(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh));
}

// 16 bytes of the source buffer at byte offset `g` (relative to src_base),
// zero-filled outside [0, total).
__device__ __forceinline__ uint4 load_group(const uint8_t *src_base, int64_t g, int64_t total) {
  if (g >= 0 && g + 16 <= total) {
    uint4 v;
    __builtin_memcpy(&v, src_base + g, 16);
    return v;
  }
  uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
  for (int k = 0; k < 16; k++) {
    int64_t a = g + k;
    if (a >= 0 && a < total) w[k >> 2] |= (uint32_t)src_base[a] << (8 * (k & 3));
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits
// for vmcnt(0), which would expose the latency of the chain stores and of the
// next tile's prefetch at every barrier.  Nothing in K1 communicates through
// global memory inside a launch, so LDS ordering is all the barrier must give.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- wave-wide bitonic sort, one u32 key per lane ---------------------------
// Compare-exchange steps whose partner lies in the same row of 16 lanes run on
// DPP operands (no LDS traffic); when the lanes that keep the minimum are whole
// banks of 4 lanes, a bank-masked min and a bank-masked max do the step in two
// instructions.  Steps that cross rows use ds_bpermute with byte addresses
// computed once per kernel (K1SortAddr).  All 64 lanes must be active.
struct K1SortAddr {
  int x16, x32, m32, m64; // (lane ^ 16) << 2, (lane ^ 32) << 2, (lane ^ 31) << 2, (lane ^ 63) << 2
};
constexpr int kDppQuadXor1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int kDppQuadMirror = 0x1B; // quad_perm:[3,2,1,0]
constexpr int kDppWaveShl1 = 0x130;  // lane i reads lane i+1
constexpr int kDppWaveShr1 = 0x138;  // lane i reads lane i-1

// lane i reads lane i-1 / i+1; the first / last lane gets `edge`
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t edge) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, kDppWaveShr1, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v, uint32_t edge) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, kDppWaveShl1, 0xF, 0xF, false);
}

template <int CTRL>
__device__ __forceinline__ uint32_t k1_ce_quad(uint32_t K, bool upper) {
  const uint32_t p = (uint32_t)__builtin_amdgcn_update_dpp((int)K, (int)K, CTRL, 0xF, 0xF, true);
  return upper ? max(K, p) : min(K, p);
}
__device__ __forceinline__ uint32_t k1_ce_far(uint32_t K, int addr, bool upper) {
  const uint32_t p = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)K);
  return upper ? max(K, p) : min(K, p);
}
// min on the banks of LO_BANKS (partner through LO_CTRL), max on the others.
// s_nop 1: a DPP operand must not be read within 2 cycles of the VALU write.
#define K1_CE_BANKED(K, LO_CTRL, LO_BANKS, HI_CTRL, HI_BANKS)                                        \
  do {                                                                                               \
    uint32_t n_;                                                                                     \
    asm("s_nop 1\n\t"                                                                                \
        "v_min_u32_dpp %0, %1, %1 " LO_CTRL " row_mask:0xf bank_mask:" LO_BANKS "\n\t"               \
        "v_max_u32_dpp %0, %1, %1 " HI_CTRL " row_mask:0xf bank_mask:" HI_BANKS                      \
        : "=&v"(n_)                                                                                  \
        : "v"(K));                                                                                   \
    K = n_;                                                                                          \
  } while (0)

// Merge steps lane ^ j for j = JSTART .. 1 (input bitonic in blocks of 2*JSTART).
template <int JSTART>
__device__ __forceinline__ uint32_t k1_merge(uint32_t K, uint32_t lane, const K1SortAddr &A) {
  if (JSTART >= 32) K = k1_ce_far(K, A.x32, lane & 32);
  if (JSTART >= 16) K = k1_ce_far(K, A.x16, lane & 16);
  if (JSTART >= 8) K1_CE_BANKED(K, "row_shl:8", "0x3", "row_shr:8", "0xc");
  if (JSTART >= 4) K1_CE_BANKED(K, "row_shl:4", "0x5", "row_shr:4", "0xa");
  if (JSTART >= 2) K = k1_ce_quad<kDppQuadXor2>(K, lane & 2);
  return k1_ce_quad<kDppQuadXor1>(K, lane & 1);
}

// Ascending sort of 64 keys, one per lane (21 compare-exchange steps).  The
// first step of each merge pairs a lane with its mirror inside the block.
__device__ __forceinline__ uint32_t k1_sort64(uint32_t K, uint32_t lane, const K1SortAddr &A) {
  K = k1_ce_quad<kDppQuadXor1>(K, lane & 1);
  K = k1_ce_quad<kDppQuadMirror>(K, lane & 2);
  K = k1_merge<1>(K, lane, A);
  K1_CE_BANKED(K, "row_half_mirror", "0x5", "row_half_mirror", "0xa");
  K = k1_merge<2>(K, lane, A);
  K1_CE_BANKED(K, "row_mirror", "0x3", "row_mirror", "0xc");
  K = k1_merge<4>(K, lane, A);
  K = k1_ce_far(K, A.m32, lane & 16);
  K = k1_merge<8>(K, lane, A);
  K = k1_ce_far(K, A.m64, lane & 32);
  return k1_merge<16>(K, lane, A);
}

// Owner wave: sort NJ*64 keys, link equal hashes, read / update the table
// slice, leave the chain entries in res[] (indexed by tile position).
template <int NJ, bool STREAM>
__device__ __forceinline__ void k1_consume(uint32_t *slice, uint32_t *res, const uint8_t *runlen8,
                                           const uint32_t *myq, uint32_t n, uint32_t sb,
                                           uint32_t lane, uint32_t cb, const K1SortAddr &A,
                                           uint32_t org_cur, uint32_t org_next_x, uint32_t org_next) {
  const uint32_t cmask = (1u << cb) - 1;
  uint32_t K[NJ], V[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const uint32_t e = j * 64 + lane;
    K[j] = e < n ? myq[e] : K1_KEY_INVALID;
  }
  K[0] = k1_sort64(K[0], lane, A);
  if (NJ == 2) {
    K[1] = k1_sort64(K[1], lane, A);
    const uint32_t br = (uint32_t)__builtin_amdgcn_ds_bpermute(A.m64, (int)K[1]); // reversed
    const uint32_t lo = min(K[0], br), hi = max(K[0], br);
    K[0] = k1_merge<32>(lo, lane, A);
    K[1] = k1_merge<32>(hi, lane, A);
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const uint32_t rel = (K[j] >> K1_KEY_REL_SHIFT) & K1_KEY_REL_MASK;
    uint32_t len1 = 0; // run length - 1
    if (K[j] != K1_KEY_INVALID && (K[j] & K1_KEY_RUN)) len1 = runlen8[rel];
    V[j] = ((sb + rel + len1) << cb) | (K[j] & cmask);
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    // sorted neighbours: left (Kp, Vp) and right (Kn), across the two slots
    uint32_t ek = K1_KEY_INVALID, ev = 0, en = K1_KEY_INVALID;
    if (j > 0) {
      ek = (uint32_t)__builtin_amdgcn_readlane((int)K[j > 0 ? j - 1 : 0], 63);
      ev = (uint32_t)__builtin_amdgcn_readlane((int)V[j > 0 ? j - 1 : 0], 63);
    }
    if (j < NJ - 1) en = (uint32_t)__builtin_amdgcn_readlane((int)K[j < NJ - 1 ? j + 1 : j], 0);
    const uint32_t Kp = wave_shr1(K[j], ek), Vp = wave_shr1(V[j], ev);
    const uint32_t Kn = wave_shl1(K[j], en);
    if (K[j] == K1_KEY_INVALID) continue;
    const uint32_t idxl = K[j] >> K1_KEY_IDX_SHIFT;
    const uint32_t rel = (K[j] >> K1_KEY_REL_SHIFT) & K1_KEY_REL_MASK;
    const bool samep = Kp != K1_KEY_INVALID && (Kp >> K1_KEY_IDX_SHIFT) == idxl;
    const bool tail = Kn == K1_KEY_INVALID || (Kn >> K1_KEY_IDX_SHIFT) != idxl;
    const uint32_t P = samep ? Vp : slice[idxl];
    // a predecessor at or below the buffer origin counts as "empty" (m4.go:152, 92-98)
    const uint32_t pp = P >> cb, pos = sb + rel, org = STREAM ? (pos >= org_next_x ? org_next : org_cur) : 0u;
    // check bits 0 = unknown (entry loaded from a persistent table): may be equal
    const uint32_t pck = P & cmask;
    res[rel] = pp > org ? ((pos - pp) | ((pck == 0 || pck == (K[j] & cmask)) ? kChainMaybe : 0u)) : 0u;
    if (tail) slice[idxl] = V[j];
  }
}

// STREAM: the jobs are segments of one stream (StreamInfo *si is set): block
// tails are skipped, origins move, priming positions get no chain entry.  The
// batch instantiation folds all of that away.
template <bool STREAM>
__global__ __launch_bounds__(K1_THREADS) void k1_chain_build(
    const uint8_t *__restrict__ src_base, uint64_t src_total, uint32_t *__restrict__ chain_base,
    const Job *__restrict__ jobs, uint32_t *__restrict__ table_io, int njobs,
    int hash_len, int table_bits, int part_bits, int check_bits, int dbg,
    const StreamInfo *__restrict__ si, uint32_t *__restrict__ table_out) {
  extern __shared__ uint32_t lds[];
  const uint32_t part_size = 1u << part_bits;
  uint32_t *table = lds;
  uint32_t *tile = lds + part_size;                 // K1_TILE_BYTES/4 dwords
  uint32_t *queues = tile + K1_TILE_BYTES / 4;      // K1_WAVES * K1_QCAP keys
  uint32_t *res = queues + K1_WAVES * K1_QCAP;      // K1_SUB chain entries, by tile position
  uint8_t *runlen8 = reinterpret_cast<uint8_t *>(res + K1_SUB); // K1_SUB run lengths - 1
  uint32_t *qn = res + K1_SUB + K1_SUB / 4;         // 2 x K1_WAVES queue fills (by tile parity)
  uint32_t *ovf = qn + 2 * K1_WAVES;                // overflow flags: 2 by tile parity + K1_PER fallback rows

  const int nparts = 1 << (table_bits - part_bits);
  // XCD-aware placement: blocks b and b+8 share an XCD, so give the nparts
  // partitions of one job the same b % 8.
  const uint32_t b = blockIdx.x;
  const uint32_t x = b & 7, r = b >> 3;
  const uint32_t q = r % nparts;
  const uint32_t job = (r / nparts) * 8 + x;
  if (job >= (uint32_t)njobs) return;

  const Job J = jobs[job];
  uint32_t *chain = chain_base + J.chain_off;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const K1SortAddr SA = {(int)((lane ^ 16) << 2), (int)((lane ^ 32) << 2), (int)((lane ^ 31) << 2),
                         (int)((lane ^ 63) << 2)};

  const bool load_table = table_io && (J.k1_flags & kJobLoadTable);
  const bool store_table = table_io && (J.k1_flags & kJobStoreTable);
  for (uint32_t k = tid; k < part_size; k += K1_THREADS)
    table[k] = load_table ? table_io[(q << part_bits) + k] << check_bits : 0u; // plain positions, check bits unknown
  if (tid < 2 * K1_WAVES + 2 + K1_PER) qn[tid] = 0; // fills and overflow flags
  // block structure of a stream handled in one launch (none: one block, origin 0)
  const uint32_t st_x0 = STREAM ? si->x0 : 0u, st_xend = STREAM ? si->x_end : 0xFFFFFFFFu;
  const uint32_t st_block = STREAM ? si->block_size : 0u, st_nbp = STREAM ? si->nbp : 1u;
  uint32_t bpi = 0; // breakpoint in force at the current tile
  const uint32_t store_from = STREAM ? J.store_from : 0u;

  // hash = ((v & mask) * kHashMul64) >> (64 - table_bits), in 32-bit pieces:
  // only the upper dword of the low 64 product bits is needed (table_bits <= 24)
  const uint64_t mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
  const uint32_t mask_lo = (uint32_t)mask, mask_hi = (uint32_t)(mask >> 32);
  const uint32_t mul_lo = (uint32_t)kHashMul64, mul_hi = (uint32_t)(kHashMul64 >> 32);
  const uint32_t shift = 32 - table_bits;
  const uint32_t idx_mask = part_size - 1;
  const uint32_t own_shift = part_bits > 4 ? part_bits - 4 : 0; // slice = 2^own_shift entries
  const uint32_t own_mask = (1u << own_shift) - 1;
  const uint32_t cb = (uint32_t)check_bits; // <= K1_MAX_CHECK_BITS and positions << cb fit 32 bits
  const uint32_t end = J.len >= 8 ? J.len - 7 : 0; // positions < end are inserted

  // staged tile: bytes [A, A + K1_TILE_BYTES) of the source buffer, where A is
  // the 16-byte aligned offset at or below the tile's first position
  auto tile_origin = [&](uint32_t base) -> int64_t {
    return (int64_t)((J.src_off + base) & ~(uint64_t)15);
  };
  uint4 pre = make_uint4(0, 0, 0, 0);
  if (J.start < end && tid < K1_GROUPS) {
    pre = load_group(src_base, tile_origin(J.start) + 16 * (int64_t)tid, (int64_t)src_total);
    reinterpret_cast<uint4 *>(tile)[tid] = pre;
    if (J.start + K1_SUB < end)
      pre = load_group(src_base, tile_origin(J.start + K1_SUB) + 16 * (int64_t)tid, (int64_t)src_total);
  }
  lds_barrier(); // LDS init and the first tile image are in place

  uint32_t par = 0; // tile parity: selects the qn / ovf set
#pragma unroll 1
  for (uint32_t sb = J.start; sb < end; sb += K1_SUB, par ^= 1) {
    const uint32_t lds_off0 = (uint32_t)((J.src_off + sb) & 15);
    uint32_t *myqn = qn + par * K1_WAVES;
    // origin in force at the tile start, and the next change (at most one per
    // tile: slides are more than MaxDistance >= K1_SUB positions apart)
    uint32_t org_cur = 0, org_next_x = 0xFFFFFFFFu, org_next = 0;
    // end of the block the tile starts in: its last 7 positions are not inserted
    uint32_t blk_end = 0xFFFFFFFFu;
    if (STREAM) {
      while (bpi + 1 < st_nbp && si->bp_x[bpi + 1] <= sb) bpi++;
      org_cur = si->bp_org[bpi];
      if (bpi + 1 < st_nbp) {
        org_next_x = si->bp_x[bpi + 1];
        org_next = si->bp_org[bpi + 1];
      }
      if (sb >= st_x0) blk_end = min(st_xend, st_x0 + ((sb - st_x0) / st_block + 1) * st_block);
      else blk_end = st_x0; // history of earlier calls is never re-inserted (segments start at x0)
    }

    // Hash NW windows (64 consecutive positions per wave each, window row
    // kfirst + w) and queue the run heads of this partition at their owners.
    // Written in phases (all loads, all hashes, all queue atomics, all queue
    // writes) so the LDS round trips of the windows overlap.  Returns a bit
    // per window: this lane queued a key.
    auto produce = [&](auto nw_tag, int kfirst, bool store_interior, uint32_t *overflow) -> uint32_t {
      constexpr int NW = decltype(nw_tag)::value;
      uint2 v[NW];
      uint32_t key[NW], own[NW], slot[NW], heads = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) v[w] = lds_ld64(tile, lds_off0 + (kfirst + w) * K1_THREADS + tid);
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const uint32_t rel = (kfirst + w) * K1_THREADS + tid;
        const uint32_t lo = v[w].x & mask_lo, hi = v[w].y & mask_hi;
        const uint32_t h = (__umulhi(lo, mul_lo) + lo * mul_hi + hi * mul_lo) >> shift;
        const uint32_t pos = sb + rel;
        // (blk_end: block the tile starts in; st_xend: a last block shorter than a tile, k1_xchg.h)
        const bool in = pos < end && (h >> part_bits) == q &&
                        !(STREAM && ((pos + 7 >= blk_end && pos < blk_end) || pos + 7 >= st_xend));
        // check bits of the first 4 bytes (top bits of a multiplicative hash)
        // (never 0: that value marks table entries whose bytes are not known)
        const uint32_t ck = cb ? max(1u, (v[w].x * 0x9E3779B1u) >> (32 - cb)) : 0u;
        // run folding: this position continues the previous one (lane - 1) when
        // hash and check bits are equal
        const uint32_t tag = in ? ((h << K1_MAX_CHECK_BITS) | ck) : 0xFFFFFFFFu; // h < 2^24
        const uint32_t ptag = wave_shr1(tag, 0xFFFFFFFFu);
        const bool same = in && tag == ptag;
        const unsigned long long cont = __ballot(same);
        const uint32_t idx = h & idx_mask;
        own[w] = idx >> own_shift;
        key[w] = ((idx & own_mask) << K1_KEY_IDX_SHIFT) | (rel << K1_KEY_REL_SHIFT) | ck;
        if (in && !same) heads |= 1u << w;
        if (cont != 0) { // wave-uniform: some run in this window
          if (same && store_interior && !(dbg & 1) && pos >= store_from) // predecessor at the origin = "empty"
            chain[pos] = pos - 1 > (STREAM && pos >= org_next_x ? org_next : org_cur) ? (1u | kChainMaybe) : 0u;
          uint32_t runlen1 = 0; // run length - 1
          if (lane < 63) runlen1 = (uint32_t)__builtin_ctzll(~(cont >> (lane + 1)));
          if ((heads >> w & 1) && runlen1) {
            runlen8[rel] = (uint8_t)runlen1;
            key[w] |= K1_KEY_RUN;
          }
        }
      }
#pragma unroll
      for (int w = 0; w < NW; w++)
        if (heads >> w & 1) slot[w] = atomicAdd(&myqn[own[w]], 1u);
#pragma unroll
      for (int w = 0; w < NW; w++)
        if (heads >> w & 1) {
          if (slot[w] < (uint32_t)K1_QCAP) queues[own[w] * K1_QCAP + slot[w]] = key[w];
          else *overflow = 1;
        }
      return heads;
    };
    auto consume = [&]() {
      const uint32_t n = myqn[wave];
      if (n == 0) return;
      uint32_t *slice = table + (wave << own_shift);
      if (n <= 64)
        k1_consume<1, STREAM>(slice, res, runlen8, queues + wave * K1_QCAP, n, sb, lane, cb, SA, org_cur, org_next_x, org_next);
      else
        k1_consume<2, STREAM>(slice, res, runlen8, queues + wave * K1_QCAP, n, sb, lane, cb, SA, org_cur, org_next_x, org_next);
    };

    const uint32_t headmask = produce(std::integral_constant<int, K1_PER>{}, 0, true, &ovf[par]);
    lds_barrier(); // queues complete
    if (!(dbg & 2)) {
      if (!ovf[par]) {
        consume();
      } else if (!(dbg & 4)) {
        // Some owner got more than K1_QCAP keys (short-period data).  Redo the
        // tile in spans of windows: a row of 16 windows (1024 positions) first,
        // halving a span that still overflows, down to two windows, which hold
        // <= 128 keys in total and cannot overflow.  ovf[2] is the flag of the
        // current attempt; all decisions are workgroup-uniform.
        if (tid < K1_WAVES) myqn[tid] = 0;
        if (tid == 0) ovf[2] = 0;
        lds_barrier();
        uint32_t u = 0, span = K1_WAVES; // next window (row = u / 16, wave = u % 16), windows per attempt
#pragma unroll 1
        while (u < (uint32_t)(K1_PER * K1_WAVES)) {
          const uint32_t w0 = u % K1_WAVES;
          if (wave >= w0 && wave < w0 + span)
            produce(std::integral_constant<int, 1>{}, (int)(u / K1_WAVES), false, &ovf[2]);
          lds_barrier();
          const bool over = ovf[2] != 0; // cannot happen with span == 2
          if (!over) consume();
          lds_barrier(); // everybody has read the flag and the queues
          if (tid < K1_WAVES) myqn[tid] = 0;
          if (tid == 0) ovf[2] = 0;
          lds_barrier();
          if (over && span > 2) {
            span >>= 1;
          } else { // (over with span == 2 cannot happen; advancing keeps the loop finite regardless)
            u += span;
            if (u % K1_WAVES == 0) span = K1_WAVES; // a new row starts with a full span again
          }
        }
      }
    }
    // stage the next tile (everybody is done reading the current image) and
    // start fetching the one after it
    if (tid < K1_GROUPS && sb + K1_SUB < end) {
      reinterpret_cast<uint4 *>(tile)[tid] = pre;
      if (sb + 2 * K1_SUB < end)
        pre = load_group(src_base, tile_origin(sb + 2 * K1_SUB) + 16 * (int64_t)tid, (int64_t)src_total);
    }
    lds_barrier(); // res[] complete, next image staged
    // this parity's counters are next used two tiles from now
    if (lane == 0) myqn[wave] = 0;
    if (tid == 0) {
      ovf[par] = 0;
    }
    if (dbg & 1) continue; // profiling aid: no chain stores
#pragma unroll
    for (int k = 0; k < K1_PER; k++)
      if ((headmask & (1u << k)) && sb + k * K1_THREADS + tid >= store_from)
        chain[sb + k * K1_THREADS + tid] = res[k * K1_THREADS + tid];
  }

  if (store_table) { // entries relative to the origin after the call, at or below it = empty
    const uint32_t rebase = STREAM ? si->final_org : 0u;
    lds_barrier();
    for (uint32_t k = tid; k < part_size; k += K1_THREADS) {
      const uint32_t v = table[k] >> check_bits; // the persistent table holds plain positions
      (table_out ? table_out : table_io)[(q << part_bits) + k] = v > rebase ? v - rebase : 0u;
    }
  }
}

} // namespace m4g
#include "k1_xchg.h"
#include "k1_split.h"
#include "k2_eval.h"
namespace m4g {

// ---------------------------------------------------------------------------
// K3: the parse, spread over lanes by sub-block speculation (m4_spec.h).
//
//   k3a_spec      one lane per (job, sub-block): speculative parse -> RS slot,
//                 end state E.
//   k3b_stitch    one lane per (job, sub-block >= 1): true parse from E[k-1]
//                 until it synchronises with RS[k] -> ST slot, descriptor,
//                 Tout (true state after the block when it did not sync).
//   k3c_validate  one wave per job: walks the blocks in order; as long as every
//                 block synchronised, the assumption "block k is entered in
//                 state E[k-1]" holds by induction and only offsets are summed
//                 (64 blocks per step).  After a block that did not sync, the
//                 following blocks are re-stitched serially from the true
//                 state until one synchronises again.
// Slot layout: block b = job * kmax + k owns RS/ST records [b*slot, (b+1)*slot).
// ---------------------------------------------------------------------------
struct SpecLayout {
  uint32_t kmax;      // sub-blocks reserved per job
  uint32_t slot;      // Recs per (job, block) slot in RS and in ST
  int32_t sub_block;  // positions per sub-block
};

constexpr int K3_THREADS = 64;

// ---- staged view of the positions a lane is walking ------------------------
// Every lane of k3a walks its own sub-block, so a wave has 64 independent
// sequential streams through src and chain.  Read straight from global memory
// each 4-byte access pulls a whole cache line that is evicted again long
// before the lane comes back to it (measured: 1.4 L2 misses per input byte).
// Instead the wave advances in epochs of K3_W positions: at the start of an
// epoch it cooperatively copies, for each of its 64 sub-blocks, the bytes
// [w0-K3_BACK, w0+K3_W+K3_FWD) and the chain entries [w0, w0+K3_W) into LDS
// with 16-byte loads that use every byte they fetch.  Reads inside the staged
// range are served from LDS (row strides are odd, so lanes at equal offsets
// hit different banks); reads outside it (match sources, long extensions)
// still go to global memory.
#ifndef M4G_K3_W
#define M4G_K3_W 16
#endif
#ifndef M4G_K3_MINWAVES
#define M4G_K3_MINWAVES 4
#endif
constexpr int K3_W = M4G_K3_W;
constexpr int K3_BACK = 16;
#ifndef M4G_K3_FWD
#define M4G_K3_FWD 32
#endif
constexpr int K3_FWD = M4G_K3_FWD;
constexpr int K3_SRC_BYTES = K3_BACK + K3_W + K3_FWD;  // 80 = 5 x 16
constexpr int K3_SRC_PIECES = K3_SRC_BYTES / 16;
constexpr int K3_SRC_DW = K3_SRC_BYTES / 4 + 1;        // +1: the funnel shift reads one dword ahead
constexpr int K3_CH_PIECES = K3_W / 4;
constexpr int K3_CH_DW = K3_W + 1;
static_assert(K3_SRC_DW % 2 == 1 && K3_CH_DW % 2 == 1, "odd row strides keep the banks apart");

struct StagedMem {
  const uint8_t *src;
  const uint32_t *chain;
  const uint32_t *lsrc; // LDS row of this lane: bytes [w0 - K3_BACK, w0 + K3_W + K3_FWD)
  const uint32_t *lch;  // LDS row of this lane: chain[w0 .. w0 + K3_W)
  int32_t w0;
  int32_t give_up_at;   // > 0: a match reaching this position makes the caller drop the parse (k3a)
  const K2Res *k2;      // K2's results for this job's positions, or null (m4_k2.h)
  __device__ __forceinline__ bool has_k2() const { return k2 != nullptr; }
  __device__ __forceinline__ const K2Res *k2_ptr(int32_t p) const { return k2 + p; }
  const SortEnt *sw;    // the job's sorted position list and the positions' indices in it, or null (m4_parse.h)
  const uint32_t *rank;
  __device__ __forceinline__ bool has_sw() const { return sw != nullptr; }
  __device__ __forceinline__ const SortEnt *sw_ptr(int32_t r) const { return sw + r; }
  __device__ __forceinline__ uint32_t rank_at(int32_t p) const { return rank[p]; }
  __device__ __forceinline__ uint8_t ld8(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + K3_BACK);
    if (o < (uint32_t)K3_SRC_BYTES) return (uint8_t)(lsrc[o >> 2] >> (8 * (o & 3)));
    return src[p];
  }
  __device__ __forceinline__ uint32_t ld32(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + K3_BACK);
    if (o <= (uint32_t)(K3_SRC_BYTES - 4)) {
      const uint32_t w = o >> 2;
      return __builtin_amdgcn_alignbyte(lsrc[w + 1], lsrc[w], o & 3);
    }
    return m4g::ld32(src + p);
  }
  __device__ __forceinline__ uint64_t ld64(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + K3_BACK);
    if (o <= (uint32_t)(K3_SRC_BYTES - 8)) {
      const uint32_t w = o >> 2, sh = o & 3;
      const uint32_t d0 = lsrc[w], d1 = lsrc[w + 1], d2 = lsrc[w + 2];
      return ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) |
             __builtin_amdgcn_alignbyte(d1, d0, sh);
    }
    return m4g::ld64(src + p);
  }
  // match sources are (almost) never inside the staged range: always global,
  // no branch, so the loads of one iteration issue back to back
  __device__ __forceinline__ uint64_t far64(int32_t p) const { return m4g::ld64(src + p); }
  __device__ __forceinline__ uint32_t chain_at(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0);
    if (o < (uint32_t)K3_W) return lch[o];
    return chain[p];
  }
  // light filter of parse_range: p in [w0, w0 + K3_W), resp. p + 4 <= w0 + K3_W + K3_FWD
  __device__ __forceinline__ uint32_t near_chain(int32_t p) const { return lch[p - w0]; }
  __device__ __forceinline__ uint32_t near32(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + K3_BACK), w = o >> 2;
    return __builtin_amdgcn_alignbyte(lsrc[w + 1], lsrc[w], o & 3);
  }
};

// Forward extension of a speculative lane.  Same result as the generic one,
// except that a match which has run for 256 bytes and also agrees in the 8
// bytes just below give_up_at is reported as reaching give_up_at at once: the
// caller abandons the block then anyway, and lanes inside long repeats stop
// paying for extensions nobody uses.
__device__ __forceinline__ int32_t extend_fwd(const StagedMem &M, int32_t len, int32_t i, int32_t j) {
  int32_t run = 0;
  while (j + 8 < len) {
    const uint64_t a = M.ld64(i), b = M.ld64(j);
    if (a != b) return j + (int32_t)(__builtin_ctzll(a ^ b) >> 3);
    i += 8;
    j += 8;
    run += 8;
    if (run == 256 && M.give_up_at > 0 && j + 16 < M.give_up_at) {
      const int32_t d = M.give_up_at - 8 - j;
      if (M.far64(i + d) == M.far64(j + d)) return M.give_up_at;
    }
  }
  while (j < len && M.ld8(i) == M.ld8(j)) {
    i++;
    j++;
  }
  return j;
}

__device__ __forceinline__ int64_t shfl64(int64_t v, int srclane) {
  const uint32_t lo = __shfl((uint32_t)v, srclane), hi = __shfl((uint32_t)((uint64_t)v >> 32), srclane);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Wave-cooperative copy of one epoch's rows into LDS.  gsrc / gch are this
// lane's byte / element offsets of window position w0 relative to src_base /
// chain_base; rows of inactive lanes are skipped.
__device__ __forceinline__ void k3_stage(const uint8_t *__restrict__ src_base, int64_t src_total,
                                         const uint32_t *__restrict__ chain_base,
                                         uint32_t *lds_src, uint32_t *lds_ch, int lane,
                                         bool row_on, int64_t gsrc, int64_t gch) {
#pragma unroll
  for (int it = 0; it < K3_SRC_PIECES; it++) {
    const int p = it * 64 + lane;
    const int row = p / K3_SRC_PIECES, part = p - row * K3_SRC_PIECES;
    const int64_t g = shfl64(gsrc, row) - K3_BACK + part * 16;
    const bool on = __shfl((int)row_on, row) != 0;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (on) v = load_group(src_base, g, src_total);
    uint32_t *d = lds_src + row * K3_SRC_DW + part * 4;
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
    d[3] = v.w;
  }
#pragma unroll
  for (int it = 0; it < K3_CH_PIECES; it++) {
    const int p = it * 64 + lane;
    const int row = p / K3_CH_PIECES, part = p - row * K3_CH_PIECES;
    const int64_t g = shfl64(gch, row) + part * 4;
    const bool on = __shfl((int)row_on, row) != 0;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (on) __builtin_memcpy(&v, chain_base + g, 16);
    uint32_t *d = lds_ch + row * K3_CH_DW + part * 4;
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
    d[3] = v.w;
  }
}

// XCD-aware wave placement: workgroups b and b+8 run on the same XCD (observed
// round-robin dispatch; a speed assumption only).  A job's sub-blocks are
// spread over kmax/64 consecutive waves; give all of them the same b % 8 so
// that the job's bytes -- read at random as match sources -- are cached in ONE
// XCD's L2 instead of being pulled into all eight.
__device__ __forceinline__ uint64_t k3_wave_of_block(uint32_t b, uint32_t nblocks, uint32_t kmax) {
  const uint32_t wpj = (kmax + K3_THREADS - 1) / K3_THREADS; // waves per job (when kmax % 64 == 0)
  const uint32_t group = 8 * wpj;
  if (kmax % K3_THREADS != 0 || b >= (nblocks / group) * group) return b;
  const uint32_t g = b / group, r = b % group;
  const uint32_t xcd = r % 8, slot = r / 8;
  return (uint64_t)g * group + xcd * wpj + slot;
}

// A speculative lane gives its block up when a match runs more than this far
// past the block end: inside long repeats (zero pages, short periods) every
// lane would otherwise extend the same match to the end of the buffer.  The
// true parse usually has that match pending from an earlier block and skips
// the block in O(1); where it does not, k3c redoes the block with a forward
// extension that uses the whole wave.  Measured on paragraphs of 2000 / 8192 /
// 40000 bytes repeated at random (256 x 1 MiB): 4096 -> 18.7 / 19.4 / 26.4 GB/s,
// 2048 -> 18.4 / 27.2 / 32.2, 1024 -> 15.9 / 29.3 / 33.0 (k3c, one wave per chunk,
// then redoes too many blocks of the first input); text is indifferent.
// Round 3, with k3a_fast (tools/ext_ab.sh): 1024 -> 16.9 / 31.0 / 36.0, 2048 -> 24.9 / 28.8 / 36.3, 3072 -> 23.8 /
// 25.0 / 34.0, 4096 -> 25.2 / 20.8 / 30.9, 8192 -> 23.8 / 12.3 / 20.5; at 1024 chunks, where k3c's one wave per chunk
// is no longer the bound, 2048 gives 45.4 / 48.6 / 52.3.
#ifndef M4G_K3_EXT_LIMIT
#define M4G_K3_EXT_LIMIT 2048
#endif
constexpr int32_t K3_EXT_LIMIT = M4G_K3_EXT_LIMIT;

template <bool LIGHT, bool XW = false /* parse_range's switch of the same name */>
__global__ __launch_bounds__(K3_THREADS, M4G_K3_MINWAVES) void k3a_spec(
    const uint8_t *__restrict__ src_base, uint64_t src_total,
    const uint32_t *__restrict__ chain_base, const Job *__restrict__ jobs, Rec *__restrict__ rs,
    ParseState *__restrict__ E, BlockDesc *__restrict__ desc, uint32_t *__restrict__ errors,
    int njobs, SpecLayout L, ParseParams P, int budget /* > 0: per-lane windows, see below */) {
  __shared__ uint32_t lds_src[K3_THREADS * K3_SRC_DW];
  __shared__ uint32_t lds_ch[K3_THREADS * K3_CH_DW];
  const int lane = threadIdx.x;
  const uint64_t t = k3_wave_of_block(blockIdx.x, gridDim.x, L.kmax) * K3_THREADS + lane;
  const uint32_t job = (uint32_t)(t / L.kmax), k = (uint32_t)(t % L.kmax);
  // no early return: every lane takes part in the cooperative staging
  Job J{};
  if (job < (uint32_t)njobs) J = jobs[job];
  const SpecGeom g = spec_geom(J.start, J.len, L.sub_block);
  const bool active = job < (uint32_t)njobs && k < (uint32_t)g.nblocks;
  const int32_t bb = block_begin(g, active ? (int32_t)k : 0);
  const int32_t be = block_end(g, active ? (int32_t)k : 0);
  const int32_t len = (int32_t)J.len;
  // matches are followed at most K3_EXT_LIMIT past the block (see above)
  const int32_t len_eff = (int64_t)be + K3_EXT_LIMIT < (int64_t)len ? be + K3_EXT_LIMIT : len;

  RecSink sink{rs + t * L.slot, 0, L.slot};
  ParseState S;
  state_init(S, bb, (active && k == 0) ? J.seed_prev : 0u);
  StagedMem M;
  M.src = src_base + J.src_off;
  M.chain = chain_base + J.chain_off;
  M.lsrc = lds_src + lane * K3_SRC_DW;
  M.lch = lds_ch + lane * K3_CH_DW;
  M.give_up_at = len_eff < len ? len_eff : 0;
  M.k2 = P.k2 ? P.k2 + J.chain_off : nullptr;
  M.sw = P.sw ? P.sw + J.chain_off : nullptr;
  M.rank = P.rank ? P.rank + J.chain_off : nullptr;
  bool aborted = false;

  if (budget > 0) {
    // Chain walks: the work of 16 positions differs a lot from lane to lane (a lane whose positions all lie in
    // crowded buckets walks 65 candidates at each, most lanes a few), and in lockstep epochs every lane waits for
    // the slowest one of its wave, 64 times per sub-block.  Here every lane has its OWN window, staged where the
    // lane stands (a lane that jumped over a match skips the windows in between), and a pass ends after `budget`
    // rounds of the parse loop at the latest: lanes with light windows move on, a lane deep in walks continues
    // in its next window from where it stopped.
    int32_t at = bb;
    while (__any(active && !aborted && at < be)) {
      const bool row_on = active && !aborted && at < be;
      const int32_t w0 = at & ~3; // (chain rows are copied in 16-byte pieces: keep them word-aligned as the epochs do)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      k3_stage(src_base, (int64_t)src_total, chain_base, lds_src, lds_ch, lane, row_on,
               (int64_t)J.src_off + w0, (int64_t)J.chain_off + w0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (row_on) {
        M.w0 = w0;
        parse_range<false, LIGHT, true, true, XW>(M, len_eff, P, S, min(w0 + K3_W, be), sink, budget);
        aborted = len_eff < len && (S.error & kStateTouchedEnd);
        at = S.i;
      }
    }
  } else
  for (int32_t e = 0; e < L.sub_block; e += K3_W) { // uniform trip count
    const int32_t w0 = bb + e;
    const bool row_on = active && !aborted && w0 < be;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    k3_stage(src_base, (int64_t)src_total, chain_base, lds_src, lds_ch, lane, row_on,
             (int64_t)J.src_off + w0, (int64_t)J.chain_off + w0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (row_on) {
      M.w0 = w0;
      // (the LIGHT filter pays only without chain walks: with ChainLength > 0 every entry within MaxDistance is an
      // event, and params D measured 158 ms per GiB with it against 144 without)
      // (two instantiations, chosen at the launch: both in one kernel cost it its fourth wave per SIMD)
      parse_range<false, LIGHT, true, true, XW>(M, len_eff, P, S, min(w0 + K3_W, be), sink);
      aborted = len_eff < len && (S.error & kStateTouchedEnd);
    }
  }
  if (!active) return;
  if (aborted) {
    S.error |= kStateAborted;
    sink.n = 0;
  } else if (k == (uint32_t)g.nblocks - 1) {
    parse_finish(len, P, S, sink);
  }
  E[t] = S;
  BlockDesc d;
  d.n_spec = sink.n;
  d.n_stitch = 0;
  d.keep_from = 0;
  d.out_off = 0;
  desc[t] = d;
  if (sink.n > L.slot) atomicOr(errors, 2u);
  // a range error of a speculative parse may come from its guessed state: only
  // block 0, whose parse is the true one, reports here; k3b / k3c report theirs
  if (k == 0 && !aborted && (S.error & kStateErrRange)) atomicOr(errors, 1u);
}

// ---------------------------------------------------------------------------
// K3a for ChainLength == 0: the same speculative parse, one lane per sub-block, but the wave
// advances by EVENTS (m4_fast.h) and every lane has its own window.
//
// k3a_spec above runs parse_range in lockstep epochs of 16 positions: each wave-iteration executes
// the union of the loop's paths (~475 instructions) and waits for far memory, ~700 times per
// sub-block, although a lane has something to do at only ~200 of its 1024 positions
// (tools/analysis/parse_stats.cpp), and a lane that jumps over a match idles until the epoch ends.
// Here one ROUND of the wave is:
//   1. fast_light   every running lane skips to its next event (straight-line code, LDS only);
//   2. fast_issue   flushes, and names the far bytes the event needs; lanes whose position has
//                   left their window ask for a new one (16 positions, aligned);
//      a lane whose event needs nothing from far memory (a flush whose probe bytes were kept when
//      the match was found) repeats 1 and 2 once in the same round;
//   3. ONE batch of loads: probe / candidate bytes of the event lanes and the windows (up to
//                   KF_MAX_ROWS rows, copied cooperatively in 16-byte pieces, the piece -> row
//                   assignment through a list in LDS); one wait;
//   4. fast_complete  candidate evaluation and the queue logic on registers;
//   5. lanes whose iteration needs the generic code (a probe that hits, a long match, three pending
//      matches: ~6 per KiB) PARK; when KF_RARE_BATCH of them wait, or nobody else can run, the wave
//      runs parse_range for exactly that one iteration of each.
// A lane takes ~2 rounds per match (flush + probe bytes, candidate), the wave ~1/3 of the rounds it
// took iterations, and a round issues fewer instructions than an iteration did.
// ---------------------------------------------------------------------------
constexpr int KF_BACK = 16, KF_FWD = 16;
constexpr int KF_SRC_BYTES = KF_BACK + KF_W + KF_FWD; // 48
constexpr int KF_SRC_PIECES = KF_SRC_BYTES / 16;      // 3
constexpr int KF_CH_PIECES = KF_W / 4;                // 4
constexpr int KF_PIECES = KF_SRC_PIECES + KF_CH_PIECES;
constexpr int KF_SRC_DW = KF_SRC_BYTES / 4 + 1;       // 13: odd row strides keep the banks apart
constexpr int KF_CH_DW = KF_W + 1;                    // 17
#ifndef M4G_KF_ITERS
#define M4G_KF_ITERS 3
#endif
constexpr int KF_ITERS = M4G_KF_ITERS;                // 16-byte pieces per lane and round
constexpr int KF_MAX_ROWS = KF_ITERS * 64 / KF_PIECES; // windows filled per round
#ifndef M4G_KF_MINWAVES
#define M4G_KF_MINWAVES 4
#endif
#ifndef M4G_KF_PASSES
#define M4G_KF_PASSES 2
#endif
constexpr int KF_PASSES = M4G_KF_PASSES; // events without far loads a lane may take per round (+1)
#ifndef M4G_KF_RARE_BATCH
#define M4G_KF_RARE_BATCH 8
#endif
constexpr int KF_RARE_BATCH = M4G_KF_RARE_BATCH;
static_assert(KF_SRC_DW % 2 == 1 && KF_CH_DW % 2 == 1, "odd row strides");

struct FastMem {
  const uint8_t *src;
  const uint32_t *chain;
  const uint32_t *lsrc; // LDS row of this lane: bytes [w0 - KF_BACK, w0 + KF_W + KF_FWD)
  const uint32_t *lch;  // LDS row of this lane: chain[w0 .. w0 + KF_W)
  int32_t w0;
  uint32_t hm;          // bit k: fast_heavy(chain[w0 + k])
  int32_t give_up_at;
  // --- the window, no range checks (m4_fast.h) ---
  __device__ __forceinline__ uint32_t near_chain(int32_t p) const { return lch[p - w0]; }
  __device__ __forceinline__ uint32_t heavy_mask(int32_t p) const { return hm >> (p - w0); }
  __device__ __forceinline__ uint32_t near32(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + KF_BACK), w = o >> 2;
    return __builtin_amdgcn_alignbyte(lsrc[w + 1], lsrc[w], o & 3);
  }
  __device__ __forceinline__ uint64_t near64(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + KF_BACK), w = o >> 2, sh = o & 3;
    const uint32_t d0 = lsrc[w], d1 = lsrc[w + 1], d2 = lsrc[w + 2];
    return ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sh);
  }
  // --- the generic accessor (parse_range for the parked iterations) ---
  __device__ __forceinline__ uint8_t ld8(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + KF_BACK);
    if (o < (uint32_t)KF_SRC_BYTES) return (uint8_t)(lsrc[o >> 2] >> (8 * (o & 3)));
    return src[p];
  }
  __device__ __forceinline__ uint32_t ld32(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + KF_BACK);
    if (o <= (uint32_t)(KF_SRC_BYTES - 4)) return near32(p);
    return m4g::ld32(src + p);
  }
  __device__ __forceinline__ uint64_t ld64(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + KF_BACK);
    if (o <= (uint32_t)(KF_SRC_BYTES - 8)) return near64(p);
    return m4g::ld64(src + p);
  }
  __device__ __forceinline__ uint64_t far64(int32_t p) const { return m4g::ld64(src + p); }
  __device__ __forceinline__ uint32_t chain_at(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0);
    if (o < (uint32_t)KF_W) return lch[o];
    return chain[p];
  }
};

// (the give-up shortcut of the StagedMem overload above)
__device__ __forceinline__ int32_t extend_fwd(const FastMem &M, int32_t len, int32_t i, int32_t j) {
  int32_t run = 0;
  // 32 bytes per step while they fit: the four loads of each stream are in flight together (a parked lane's
  // match is usually a long one -- 48-byte record bodies in log data -- and every step is a far round trip)
  while (j + 32 < len) {
    const uint64_t a0 = M.far64(i), a1 = M.far64(i + 8), a2 = M.far64(i + 16), a3 = M.far64(i + 24);
    const uint64_t b0 = M.far64(j), b1 = M.far64(j + 8), b2 = M.far64(j + 16), b3 = M.far64(j + 24);
    if (a0 != b0) return j + (int32_t)(__builtin_ctzll(a0 ^ b0) >> 3);
    if (a1 != b1) return j + 8 + (int32_t)(__builtin_ctzll(a1 ^ b1) >> 3);
    if (a2 != b2) return j + 16 + (int32_t)(__builtin_ctzll(a2 ^ b2) >> 3);
    if (a3 != b3) return j + 24 + (int32_t)(__builtin_ctzll(a3 ^ b3) >> 3);
    i += 32;
    j += 32;
    run += 32;
    if (run == 256 && M.give_up_at > 0 && j + 16 < M.give_up_at) {
      const int32_t d = M.give_up_at - 8 - j;
      if (M.far64(i + d) == M.far64(j + d)) return M.give_up_at;
    }
  }
  while (j + 8 < len) {
    const uint64_t a = M.ld64(i), b = M.ld64(j);
    if (a != b) return j + (int32_t)(__builtin_ctzll(a ^ b) >> 3);
    i += 8;
    j += 8;
    run += 8;
    if (run == 256 && M.give_up_at > 0 && j + 16 < M.give_up_at) {
      const int32_t d = M.give_up_at - 8 - j;
      if (M.far64(i + d) == M.far64(j + d)) return M.give_up_at;
    }
  }
  while (j < len && M.ld8(i) == M.ld8(j)) {
    i++;
    j++;
  }
  return j;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16 bytes of a window row; pieces that reach past either end of the whole source buffer (the
// first window of the first job, the last of the last) take the byte-wise path.
__device__ __forceinline__ uint4 kf_load_src_piece(const uint8_t *src_base, int64_t g, int64_t total) {
  if (__builtin_expect(g >= 0 && g + 16 <= total, 1)) {
    uint4 v;
    __builtin_memcpy(&v, src_base + g, 16);
    return v;
  }
  return load_group(src_base, g, total);
}

// Requires L.kmax % 64 == 0 (the host rounds it): a wave's 64 sub-blocks belong to ONE job, so the
// job's fields and everything derived from them are wave-uniform (scalar registers).
__global__ __launch_bounds__(K3_THREADS, M4G_KF_MINWAVES) void k3a_fast(
    const uint8_t *__restrict__ src_base, uint64_t src_total,
    const uint32_t *__restrict__ chain_base, const Job *__restrict__ jobs, Rec *__restrict__ rs,
    ParseState *__restrict__ E, BlockDesc *__restrict__ desc, uint32_t *__restrict__ errors,
    int njobs, SpecLayout L, ParseParams P) {
  __shared__ uint32_t lds_src[K3_THREADS * KF_SRC_DW];
  __shared__ uint32_t lds_ch[K3_THREADS * KF_CH_DW];
  __shared__ uint32_t lds_hm[K3_THREADS];
  __shared__ uint2 lds_list[K3_THREADS]; // rows that get a window this round: {lane, first position}
  const int lane = threadIdx.x;
  const uint32_t wv = (uint32_t)k3_wave_of_block(blockIdx.x, gridDim.x, L.kmax);
  const uint32_t wpj = L.kmax / K3_THREADS;
  const uint32_t job = wv / wpj;
  if (job >= (uint32_t)njobs) return; // (wave-uniform)
  const uint32_t k = (wv - job * wpj) * K3_THREADS + lane;
  const uint64_t t = (uint64_t)wv * K3_THREADS + lane;
  const Job J = jobs[job];
  const SpecGeom g = spec_geom(J.start, J.len, L.sub_block);
  const bool active = k < (uint32_t)g.nblocks;
  const int32_t bb = block_begin(g, active ? (int32_t)k : 0);
  const int32_t be = block_end(g, active ? (int32_t)k : 0);
  const int32_t len = (int32_t)J.len;
  const int32_t len_eff = (int64_t)be + K3_EXT_LIMIT < (int64_t)len ? be + K3_EXT_LIMIT : len;
  const uint8_t *jsrc = src_base + J.src_off;
  const uint32_t *jchain = chain_base + J.chain_off;

  RecSink sink{rs + t * L.slot, 0, L.slot};
  ParseState S;
  state_init(S, bb, (active && k == 0) ? J.seed_prev : 0u);
  FastLane F;
  fast_init(F);
  FastMem M;
  M.src = jsrc;
  M.chain = jchain;
  M.lsrc = lds_src + lane * KF_SRC_DW;
  M.lch = lds_ch + lane * KF_CH_DW;
  M.w0 = -0x40000000; // no window yet
  M.hm = 0;
  M.give_up_at = len_eff < len ? len_eff : 0;
  bool done = !active, rare = false, aborted = false;
  uint32_t round = 0;
#ifdef M4G_KF_STATS
  uint32_t n_rare = 0, n_fill = 0; // rounds that ran the parked batch / filled windows
  uint64_t t_rare = 0;
  const uint64_t t_begin = wall_clock64();
#endif

  for (;;) {
    // 1. to the next event; 2. flush, and what the event needs.  A lane whose event needs nothing
    // from far memory (typically a flush whose probe bytes were kept when the match was found)
    // goes on to its next event in the same round.
    bool ev = false, needw = false, go = !done && !rare;
#pragma unroll 1
    for (int pass = 0; pass < KF_PASSES; pass++) {
      if (go) {
        const int st = fast_light(M, P, S, F, be, M.w0 + KF_W);
        done = st == kFastDone;
        needw = st == kFastWindow;
        ev = st == kFastEvent;
        go = false;
        if (ev) {
          const bool inw = (uint32_t)(S.i - M.w0) < (uint32_t)KF_W;
          go = fast_issue(M, len_eff, P, S, F, sink, inw, M.w0);
          needw = !inw;
          rare = F.rare != 0;
          ev = !go;
        }
      }
      if (__ballot(go) == 0) break;
    }
    // (a lane still free after the last pass has nothing outstanding: it simply runs again next round)
    if (__ballot(!done) == 0) break;

    // 3. loads: windows (the rows take turns when more than KF_MAX_ROWS ask) ...
    const bool lc = ev && F.ld_cand >= 0;
    const unsigned long long wmask = __ballot(needw);
    uint4 rv[KF_ITERS];
    int npieces = 0;
    bool served = false;
    const int32_t nw0 = S.i & ~(KF_W - 1);
    if (wmask) {
#ifdef M4G_KF_STATS
      n_fill++;
#endif
      const unsigned rot = (round * KF_MAX_ROWS) & 63u;
      const unsigned long long wrot = (wmask >> rot) | (rot ? wmask << (64 - rot) : 0ull);
      const unsigned rl = ((unsigned)lane - rot) & 63u; // this lane's place in the rotated order
      const int rank = __builtin_popcountll(wrot & ((1ull << rl) - 1ull));
      npieces = min(__builtin_popcountll(wmask), KF_MAX_ROWS) * KF_PIECES;
      served = needw && rank < KF_MAX_ROWS;
      if (served) {
        lds_list[rank] = make_uint2((uint32_t)lane, (uint32_t)nw0);
        lds_hm[lane] = 0;
      }
      wave_lds_sync();
#pragma unroll
      for (int it = 0; it < KF_ITERS; it++) {
        rv[it] = make_uint4(0, 0, 0, 0);
        if (it * 64 < npieces) { // (wave-uniform)
          const int p = it * 64 + lane;
          if (p < npieces) {
            const int r = p / KF_PIECES, part = p - r * KF_PIECES;
            const int32_t rw0 = (int32_t)lds_list[r].y;
            if (part < KF_SRC_PIECES)
              rv[it] = kf_load_src_piece(src_base, (int64_t)J.src_off + rw0 - KF_BACK + part * 16, (int64_t)src_total);
            else
              __builtin_memcpy(&rv[it], jchain + rw0 + (part - KF_SRC_PIECES) * 4, 16);
          }
        }
      }
    }
    // ... and the events' far bytes
    const bool lp = ev && F.ld_probe >= 0;
    const uint64_t pl = M.far64(lp ? F.ld_probe : 0);
    const uint64_t cb = M.far64(lc ? F.ld_cand - 8 : 0);
    const uint64_t cf = M.far64(lc ? F.ld_cand : 0);
    const uint64_t cf2 = M.far64(lc ? F.ld_cand + 8 : 0);
    const uint64_t cf3 = M.far64(lc ? F.ld_cand + 16 : 0);

    if (wmask) {
#pragma unroll
      for (int it = 0; it < KF_ITERS; it++) {
        const int p = it * 64 + lane;
        if (p < npieces) {
          const int r = p / KF_PIECES, part = p - r * KF_PIECES;
          const int row = (int)lds_list[r].x;
          const bool is_ch = part >= KF_SRC_PIECES;
          uint32_t *d = is_ch ? lds_ch + row * KF_CH_DW + (part - KF_SRC_PIECES) * 4 : lds_src + row * KF_SRC_DW + part * 4;
          d[0] = rv[it].x, d[1] = rv[it].y, d[2] = rv[it].z, d[3] = rv[it].w;
          const uint32_t bits = (uint32_t)fast_heavy(rv[it].x, P) | (uint32_t)fast_heavy(rv[it].y, P) << 1 |
                                (uint32_t)fast_heavy(rv[it].z, P) << 2 | (uint32_t)fast_heavy(rv[it].w, P) << 3;
          if (is_ch && bits) atomicOr(&lds_hm[row], bits << (4 * (part - KF_SRC_PIECES)));
        }
      }
      wave_lds_sync();
      if (served) {
        M.w0 = nw0;
        M.hm = lds_hm[lane];
      }
    }
    // 4. the rest of the event's iteration
    if (ev && !rare) {
      fast_complete(M, len_eff, P, S, F, (uint32_t)(S.i - M.w0) < (uint32_t)KF_W, pl, cb, cf, cf2, cf3);
      rare = F.rare != 0;
      if (len_eff < len && (S.error & kStateTouchedEnd)) aborted = done = true, rare = false;
    }
    // 5. parked iterations, a batch at a time
    const unsigned long long rmask = __ballot(rare);
    if (__builtin_expect(rmask && (__builtin_popcountll(rmask) >= KF_RARE_BATCH || __ballot(!done && !rare) == 0), 0)) {
#ifdef M4G_KF_STATS
      n_rare++;
      const uint64_t tr0 = wall_clock64();
#endif
      if (rare) {
        parse_range<false, false, false, false>(M, len_eff, P, S, S.i + 1, sink);
        rare = false;
        if (len_eff < len && (S.error & kStateTouchedEnd)) aborted = done = true;
      }
#ifdef M4G_KF_STATS
      t_rare += wall_clock64() - tr0;
#endif
    }
    round++;
  }
#ifdef M4G_KF_STATS
  if (lane == 0) {
    atomicAdd(errors + 13, round), atomicAdd(errors + 14, n_rare), atomicAdd(errors + 15, n_fill);
    atomicAdd(errors + 9, (uint32_t)((wall_clock64() - t_begin) >> 8)), atomicAdd(errors + 10, (uint32_t)(t_rare >> 8)); // 100 MHz ticks / 256
  }
#endif
  if (!active) return;
  if (aborted) {
    S.error |= kStateAborted;
    sink.n = 0;
  } else if (k == (uint32_t)g.nblocks - 1) {
    parse_finish(len, P, S, sink);
  }
  E[t] = S;
  BlockDesc d;
  d.n_spec = sink.n;
  d.n_stitch = 0;
  d.keep_from = 0;
  d.out_off = 0;
  desc[t] = d;
  if (sink.n > L.slot) atomicOr(errors, 2u);
  if (k == 0 && !aborted && (S.error & kStateErrRange)) atomicOr(errors, 1u);
}

// desc.out_off doubles as the k3b -> k3c status (kStitch*).
// CHAIN = false (ChainLength 0): 6 waves per SIMD (80 registers, 26 spilled) -- the stitch is one lane per block
// waiting for dependent reads, and more resident waves hide more of them (measured on 4096 x 1 MiB of text, K3 stage:
// 4 / 5 / 6 / 8 waves 31.30 / 31.15 / 31.06 / 31.54 ms).
#ifndef M4G_K3B_MINWAVES
#define M4G_K3B_MINWAVES 6
#endif
#ifndef M4G_K3B_CHAIN_WAVES
#define M4G_K3B_CHAIN_WAVES 6 // (ChainLength 64, 128-byte sub-blocks: 4 / 5 / 6 / 8 waves per SIMD 5.31 / 4.90 / 4.71 / 5.62 ms per 256 MiB)
#endif
template <bool CHAIN, bool XW = false>
__global__ __launch_bounds__(K3_THREADS, CHAIN ? M4G_K3B_CHAIN_WAVES : M4G_K3B_MINWAVES) void k3b_stitch(
    const uint8_t *__restrict__ src_base, const uint32_t *__restrict__ chain_base,
    const Job *__restrict__ jobs, const Rec *__restrict__ rs, Rec *__restrict__ st,
    const ParseState *__restrict__ E, ParseState *__restrict__ Tout,
    BlockDesc *__restrict__ desc, uint32_t *__restrict__ errors, int njobs, SpecLayout L,
    ParseParams P) {
  const uint64_t t = k3_wave_of_block(blockIdx.x, gridDim.x, L.kmax) * K3_THREADS + threadIdx.x;
  const uint32_t job = (uint32_t)(t / L.kmax), k = (uint32_t)(t % L.kmax);
  if (job >= (uint32_t)njobs || k == 0) return;
  const Job J = jobs[job];
  const SpecGeom g = spec_geom(J.start, J.len, L.sub_block);
  if (k >= (uint32_t)g.nblocks) return;
  const uint8_t *src = src_base + J.src_off;
  const uint32_t *chain = chain_base + J.chain_off;
  RecSink sink{st + t * L.slot, 0, L.slot};
  ParseState T = E[t - 1];
  BlockDesc d = desc[t];
  if (T.error & kStateAborted) { // no entry state to start from
    d.n_stitch = 0;
    d.keep_from = 0;
    d.out_off = kStitchIncomplete;
    desc[t] = d;
    return;
  }
  T.error &= kStateErrRange;
  // A block whose speculative parse emitted nothing, entered with nothing pending and no repeat distance (the
  // state the speculative parse assumed, up to next_emit): the true parse from T does what the speculative one
  // did -- the case of k3c's fast path (m4_spec.h) without a probe to check -- so its end state is E[k] with T's
  // literals and distance carried over, and there is nothing to parse again.  Incompressible stretches are made
  // of such blocks (measured on random bytes: k3b spent more time re-parsing them than k3a had spent parsing).
  // The block stays "unsynced": k3c takes Tout as the state entering block k + 1.
  {
    const ParseState Ek = E[t];
    if (d.n_spec == 0 && k + 1 < (uint32_t)g.nblocks && !T.have_prev && fast_entry_ok(T, Ek, block_begin(g, (int32_t)k))) {
      ParseState R = Ek;
      R.error = (Ek.error & kStateErrRange) | T.error;
      R.next_emit = T.next_emit;
      R.have_prev = T.have_prev;
      R.prev_dist = T.prev_dist;
      Tout[t] = R;
      d.n_stitch = 0;
      d.keep_from = 0;
      d.out_off = kStitchUnsynced;
      desc[t] = d;
      return;
    }
  }
  GlobalMem M{src, chain, CHAIN && P.k2 ? P.k2 + J.chain_off : nullptr, CHAIN && P.sw ? P.sw + J.chain_off : nullptr,
              CHAIN && P.rank ? P.rank + J.chain_off : nullptr};
  if (stitch_chase<CHAIN, XW>(M, (int32_t)J.len, P, g, k, rs + t * L.slot, L.slot, E + t, desc + t, K3_EXT_LIMIT, T, sink, d))
    Tout[t] = T;
  desc[t] = d;
  if (sink.n > L.slot) atomicOr(errors, 2u);
}

// Memory accessor for code that every lane of a wave runs in lockstep with
// identical arguments (k3c's serial re-stitch of one sub-block).  Read straight
// from global memory, every position the parse visits costs several DEPENDENT
// round trips (chain entry, near bytes, far bytes) with nothing to hide them
// behind -- one wave per chunk -- which made inputs built from long repeats
// crawl (profiles/r01_kind_probe.log: 6-13 GB/s).  So the wave first copies the
// sub-block's bytes [w0 - K3C_BACK, w0 + wn + K3C_FWD) and chain entries
// [w0, w0 + wn) into LDS with full-width loads; reads inside the window are LDS
// broadcasts, only match sources and overlong extensions go to memory.  A long
// forward extension compares 64 x 8 bytes per step instead of 8.
constexpr int K3C_WIN = 1024;  // positions staged (larger sub-blocks: the rest is read from memory)
constexpr int K3C_BACK = 16, K3C_FWD = 48;
constexpr int K3C_SRC_DW = (K3C_BACK + K3C_WIN + K3C_FWD) / 4 + 1;
constexpr int K3C_RECS = 288;  // speculative records of the block staged alongside (more: read from memory)
struct WaveMem : GlobalMem {
  const uint32_t *lsrc; // LDS image of bytes [w0 - K3C_BACK, w0 - K3C_BACK + nbytes)
  const uint32_t *lch;  // LDS image of chain[w0 .. w0 + wn)
  int32_t w0, wn, nbytes;
  __device__ __forceinline__ uint8_t ld8(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + K3C_BACK);
    if (o < (uint32_t)nbytes) return (uint8_t)(lsrc[o >> 2] >> (8 * (o & 3)));
    return src[p];
  }
  __device__ __forceinline__ uint32_t ld32(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + K3C_BACK);
    if (o <= (uint32_t)nbytes - 4) { // (nbytes >= 64; `o + 4 <= nbytes` would wrap for positions just below the window)
      const uint32_t w = o >> 2;
      return __builtin_amdgcn_alignbyte(lsrc[w + 1], lsrc[w], o & 3);
    }
    return m4g::ld32(src + p);
  }
  __device__ __forceinline__ uint64_t ld64(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0 + K3C_BACK);
    if (o <= (uint32_t)nbytes - 8) {
      const uint32_t w = o >> 2, sh = o & 3;
      const uint32_t d0 = lsrc[w], d1 = lsrc[w + 1], d2 = lsrc[w + 2];
      return ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sh);
    }
    return m4g::ld64(src + p);
  }
  __device__ __forceinline__ uint32_t chain_at(int32_t p) const {
    const uint32_t o = (uint32_t)(p - w0);
    if (o < (uint32_t)wn) return lch[o];
    return chain[p];
  }
};
__device__ __forceinline__ int32_t extend_fwd(const WaveMem &M, int32_t len, int32_t i, int32_t j) {
  const int32_t lane = (int32_t)(threadIdx.x & 63);
  while ((int64_t)j + 64 * 8 + 8 < (int64_t)len) {
    const uint64_t a = M.ld64(i + 8 * lane), b = M.ld64(j + 8 * lane);
    const unsigned long long ne = __ballot(a != b);
    if (ne) {
      const int l = __builtin_ctzll(ne);
      const uint64_t x = (uint64_t)shfl64((int64_t)(a ^ b), l);
      return j + 8 * l + (int32_t)(__builtin_ctzll(x) >> 3);
    }
    i += 64 * 8;
    j += 64 * 8;
  }
  while (j + 8 < len) { // (the generic loop, spelled out: the overload set would pick this function again)
    const uint64_t a = M.ld64(i), b = M.ld64(j);
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
// The same for the backward extension: 64 bytes per step (lane l compares the bytes l + 1 below).
__device__ __forceinline__ int32_t extend_bwd(const WaveMem &M, int32_t pos, int32_t cand, int32_t nb, int32_t maxb) {
  const int32_t lane = (int32_t)(threadIdx.x & 63);
  while (nb < maxb) {
    const int32_t k = nb + lane; // this lane's byte: k + 1 below pos / cand
    const bool ne = k < maxb && M.ld8(pos - k - 1) != M.ld8(cand - k - 1);
    const unsigned long long m = __ballot(ne);
    if (m) return nb + (int32_t)__builtin_ctzll(m);
    nb = min(nb + 64, maxb);
  }
  return nb;
}
// Wave-cooperative fill of the window of sub-block [bb, be) of a job whose buffer is src[0, len).
// lsrc has K3C_SRC_DW words (the last one is the funnel shift's read-ahead), lch has K3C_WIN.
__device__ __forceinline__ void k3c_stage(WaveMem &M, uint32_t *lsrc, uint32_t *lch, int32_t len, int32_t bb,
                                          int32_t be, int lane) {
  M.w0 = bb;
  M.wn = min(be - bb, K3C_WIN);
  M.nbytes = K3C_BACK + M.wn + K3C_FWD;
  M.lsrc = lsrc;
  M.lch = lch;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // earlier reads of the previous window are done
  __builtin_amdgcn_wave_barrier();
  for (int g = lane; g * 16 < M.nbytes; g += 64) {
    const uint4 v = load_group(M.src, (int64_t)bb - K3C_BACK + 16 * g, (int64_t)len);
    uint32_t *d = lsrc + 4 * g;
    d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
  }
  if (lane == 0) lsrc[(M.nbytes + 3) / 4] = 0;
  for (int g = lane; g * 4 < M.wn; g += 64) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (4 * g + k < M.wn) lch[4 * g + k] = M.chain[bb + 4 * g + k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool XW = false>
__global__ __launch_bounds__(64) void k3c_validate(
    const uint8_t *__restrict__ src_base, const uint32_t *__restrict__ chain_base,
    const Job *__restrict__ jobs, Rec *__restrict__ rs, Rec *__restrict__ st,
    const ParseState *__restrict__ E, const ParseState *__restrict__ Tout,
    BlockDesc *__restrict__ desc, uint32_t *__restrict__ counts, uint32_t *__restrict__ errors,
    int njobs, SpecLayout L, ParseParams P) {
  __shared__ uint32_t c_src[K3C_SRC_DW + 1];
  __shared__ uint32_t c_ch[K3C_WIN];
  __shared__ Rec c_rec[K3C_RECS];
  const uint32_t job = blockIdx.x;
  if (job >= (uint32_t)njobs) return;
  const int lane = threadIdx.x;
  const Job J = jobs[job];
  const SpecGeom g = spec_geom(J.start, J.len, L.sub_block);
  const uint32_t K = (uint32_t)g.nblocks;
  const uint64_t b0 = (uint64_t)job * L.kmax;
  const uint8_t *src = src_base + J.src_off;
  const uint32_t *chain = chain_base + J.chain_off;
  const int32_t len = (int32_t)J.len;

  uint32_t off = 0, k = 1;
  bool valid = true;
  ParseState T; // the true state entering block k; meaningful only while !valid
  state_init(T, g.start, J.seed_prev);
  uint32_t err = 0;
  uint32_t c_valid = 0, c_fast = 0, c_slow = 0, c_skip = 0; // diagnostics (M4G_TRACE)
#ifdef M4G_K3C_TIMING
  uint64_t tk[4] = {0, 0, 0, 0}, tk0 = wall_clock64();
#define K3C_TICK(b) do { const uint64_t n_ = wall_clock64(); tk[b] += n_ - tk0; tk0 = n_; } while (0)
#else
#define K3C_TICK(b) do {} while (0)
#endif
  if (E[b0].error & kStateAborted) { // even block 0 was given up: parse from the initial state
    valid = false;
    k = 0;
  } else {
    off = desc[b0].n_spec; // block 0: the speculative parse IS the true parse
    if (lane == 0) desc[b0].out_off = 0;
  }
  while (k < K) {
    if (valid) {
      c_valid++;
      const uint32_t kk = k + lane;
      BlockDesc d{0, 0, 0, kStitchSynced};
      if (kk < K) d = desc[b0 + kk];
      const unsigned long long bad = __ballot(kk < K && d.out_off != kStitchSynced); // plain sync only
      const uint32_t room = min(64u, K - k);
      const uint32_t ngood = bad ? min((uint32_t)__builtin_ctzll(bad), room) : room;
      uint32_t cnt = (uint32_t)lane < ngood ? d.n_stitch + (d.n_spec - d.keep_from) : 0;
      uint32_t incl = cnt;
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      if ((uint32_t)lane < ngood) desc[b0 + kk].out_off = off + incl - cnt;
      off += __shfl(incl, 63);
      k += ngood;
      if (bad && k < K) {
        const int bl = (int)ngood;
        const uint32_t word = __shfl(d.out_off, bl), status = word & 0xFFu, extra = word >> 8;
        const uint32_t ns = __shfl(d.n_stitch, bl), keep = __shfl(d.keep_from, bl);
        if (status == kStitchIncomplete) {
          // the stitch of block k was given up: redo it below from the state
          // its predecessor ended in, which is the true one (it synchronised)
          valid = false;
          T = E[b0 + k - 1];
          T.error &= kStateErrRange;
        } else {
          // Block k was entered in the right state, so its stitch is true.  It
          // covers blocks k .. k+extra: all its records sit in ST[k]; the blocks
          // in between have none; if it synchronised, block k+extra keeps its
          // speculative records from `keep` on.
          const uint32_t kl = k + lane; // lane l describes block k + l
          uint32_t tail = 0;
          if ((uint32_t)lane <= extra && kl < K) {
            BlockDesc nd = desc[b0 + kl];
            const uint32_t nsp = min(nd.n_spec, L.slot);
            nd.n_stitch = lane == 0 ? ns : 0;
            nd.keep_from = (status == kStitchSynced && (uint32_t)lane == extra) ? keep : nsp;
            nd.out_off = off + (lane == 0 ? 0 : ns);
            if (status == kStitchSynced && (uint32_t)lane == extra) tail = nsp - min(keep, nsp);
            desc[b0 + kl] = nd;
          }
          off += ns + __shfl(tail, (int)extra);
          if (status == kStitchUnsynced) {
            valid = false;
            T = Tout[b0 + k];
          }
          k += extra + 1;
        }
      }
      K3C_TICK(0);
    } else {
      // block k from the true state T (every lane runs the same code)
      const uint64_t t = b0 + k;
      const BlockDesc d = desc[t];
      const ParseState Ek = E[t];
      const int32_t bb = block_begin(g, (int32_t)k), be = block_end(g, (int32_t)k);
      const uint32_t n_spec = min(d.n_spec, L.slot);
      if (T.i >= be && k + 1 < K) {
        // A pending match covers this block and maybe many more: they contribute
        // no records.  All of them at once, 64 per step (never the last block,
        // whose parse also emits the tail).
        const uint32_t covered_end = min((uint32_t)((T.i - g.start) / g.sub_block), K - 1);
        for (uint32_t base = k; base < covered_end; base += 64) {
          const uint32_t kk = base + lane;
          if (kk < covered_end) {
            BlockDesc nd = desc[b0 + kk];
            nd.n_stitch = 0;
            nd.keep_from = min(nd.n_spec, L.slot);
            nd.out_off = off;
            desc[b0 + kk] = nd;
          }
        }
        c_skip += covered_end - k;
        k = covered_end;
        K3C_TICK(1);
        continue;
      }
      // Fast path.  T has nothing pending and stands at the block start, so it
      // differs from the state the speculative parse assumed only in next_emit
      // and in the repeat distance.  If no backward extension of the speculative
      // parse ran into its assumed next_emit (kStateHitLo) and the repeat probe
      // with T's distance fails (4-byte compare) at every position up to the one
      // where the speculative parse found its first match (m4_spec.h,
      // fast_probe_limit), the true parse IS the speculative one: same records
      // (the first one starts its literals at T.next_emit), same end state.
      bool fast = fast_entry_ok(T, Ek, bb);
      if (fast && T.have_prev) {
        const int32_t limit = fast_probe_limit(Ek, be); // positions below it are probed with T's distance
        // four loop positions per lane and step, from two 8-byte reads (positions i .. i + 3 compare the
        // 4-byte windows at i + 1 .. i + 4: 7 bytes of each stream); two steps in flight.  Lanes at the edges
        // of the buffer take the scalar form.  An incompressible stretch behind a match is ALL such blocks,
        // one wave per chunk walks them in order, and this loop is what it does there.
        bool hit = false;
        const int32_t pd = (int32_t)T.prev_dist;
        auto probe4 = [&](int32_t i) -> bool {
          if (i >= limit) return false;
          const int32_t p = i + 1, c = p - pd;
          if (c >= 0 && p + 8 <= len && i + 4 <= limit) {
            const uint64_t a = ld64(src + c), b = ld64(src + p), x = a ^ b;
            return (uint32_t)x == 0 || (uint32_t)(x >> 8) == 0 || (uint32_t)(x >> 16) == 0 || (uint32_t)(x >> 24) == 0;
          }
          bool h = false;
          for (int32_t q = i; q < min(i + 4, limit); q++) h |= fast_probe_hits(src, len, q, T.prev_dist);
          return h;
        };
        for (int32_t i = bb + 4 * lane; i < limit; i += 512) {
          const bool h0 = probe4(i), h1 = probe4(i + 256);
          hit |= h0 | h1;
        }
        if (__ballot(hit)) fast = false;
      }
      if (fast) {
        if (n_spec && lane == 0) { // the first record's literals begin where T's do
          Rec r = rs[t * L.slot];
          const uint32_t flag = r.lit_start & kRecFlushBit;
          r.unmatched += (r.lit_start & ~kRecFlushBit) - (uint32_t)T.next_emit;
          r.lit_start = (uint32_t)T.next_emit | flag;
          rs[t * L.slot] = r;
        }
        if (lane == 0) {
          BlockDesc nd = d;
          nd.n_stitch = 0;
          nd.keep_from = 0;
          nd.out_off = off;
          desc[t] = nd;
        }
        off += n_spec;
        c_fast++;
        const ParseState prevT = T;
        T = Ek;
        T.error = (Ek.error & kStateErrRange) | prevT.error;
        if (n_spec == 0) { // nothing emitted: literals and repeat distance carry over
          T.next_emit = prevT.next_emit;
          T.have_prev = prevT.have_prev;
          T.prev_dist = prevT.prev_dist;
        } else {
          valid = true; // E[k] is the true end state, so block k+1's stitch stands
        }
        k++;
        K3C_TICK(2);
        continue;
      }
      if (T.i >= be) c_skip++; else c_slow++;
      RecSink sink{st + t * L.slot, 0, L.slot};
      uint32_t keep = 0;
      WaveMem M;
      M.src = src;
      M.chain = chain;
      M.k2 = P.k2 ? P.k2 + J.chain_off : nullptr;
      M.sw = P.sw ? P.sw + J.chain_off : nullptr;
      M.rank = P.rank ? P.rank + J.chain_off : nullptr;
      k3c_stage(M, c_src, c_ch, len, bb, be, lane);
      // the block's speculative records, which the stitch scans for a common flush, staged the same way
      const Rec *spec = rs + t * L.slot;
      if (n_spec <= (uint32_t)K3C_RECS) {
        for (uint32_t r = lane; r < n_spec; r += 64) c_rec[r] = spec[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        spec = c_rec;
      }
      const bool ok = stitch_block<true, XW>(M, len, P, g, (int32_t)k, spec, n_spec, Ek, T, sink, keep);
      if (sink.n > L.slot) err |= 2u;
      if (lane == 0) {
        BlockDesc nd = d;
        nd.n_stitch = sink.n;
        nd.keep_from = keep;
        nd.out_off = off;
        desc[t] = nd;
      }
      off += sink.n + (n_spec - keep);
      if (ok) valid = true;
      k++;
      K3C_TICK(3);
    }
  }
  if (T.error & kStateErrRange) err |= 1u;
  if (lane == 0) {
    counts[job] = off;
    if (err) atomicOr(errors, err);
    atomicAdd(errors + 4, c_valid);
    atomicAdd(errors + 5, c_fast);
    atomicAdd(errors + 6, c_slow);
    atomicAdd(errors + 7, c_skip);
#ifdef M4G_K3C_TIMING
    for (int q = 0; q < 4; q++) atomicAdd(errors + 9 + q, (uint32_t)(tk[q] >> 4)); // 100 MHz ticks / 16
#endif
  }
}

// ---------------------------------------------------------------------------
// K4: exclusive prefix sum of counts (single workgroup) and compaction.
// ---------------------------------------------------------------------------
// carry_in: the jobs continue a list whose length so far an earlier launch left in out_off[0]
// (the groups of a plan in its worst-case layout, m4g.hip run_batch_pipeline).
__global__ __launch_bounds__(1024) void k4_scan_counts(
    const uint32_t *__restrict__ counts, uint64_t *__restrict__ out_off, int njobs, int carry_in) {
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = carry_in ? out_off[0] : 0;
  __syncthreads();
  for (int base = 0; base < njobs; base += 1024) {
    int j = base + tid;
    uint64_t v = j < njobs ? counts[j] : 0, incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    const uint64_t carry = carry_s;
    if (j < njobs) out_off[j] = carry + woff + incl - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) out_off[njobs] = carry_s;
}

// A literal run of K5_HUGE bytes or more (incompressible input: the whole chunk
// can be ONE run) is not counted by the wave of k5 that owns its match: k4_compact
// puts it on a list while it copies the record (the record holds the run's
// start) and k5b_long_literals counts it, 16 KiB per workgroup, which
// adds to the histogram k5 has written.  (A run of a whole 1 MiB chunk took one
// wave 15 us per MiB; measured on random bytes K5 was a third of the total.)
#ifndef M4G_K5_HUGE
#define M4G_K5_HUGE 8192
#endif
constexpr uint32_t K5_HUGE = M4G_K5_HUGE;
constexpr uint32_t K5B_SEG = 16384; // bytes per k5b work item
constexpr int K5B_THREADS = 256;
struct LongRun {
  uint32_t job, start, count, pad;
};

// One wave per (job, sub-block): copy the block's true records (stitch part,
// then the kept speculative part) to their place in the job's Match list.
constexpr int K4_THREADS = 256;
__global__ __launch_bounds__(K4_THREADS) void k4_compact(
    const Rec *__restrict__ rs, const Rec *__restrict__ st, const Job *__restrict__ jobs,
    const BlockDesc *__restrict__ desc, const uint64_t *__restrict__ out_off,
    uint32_t *__restrict__ out /* 3 words per match */, uint64_t out_cap, int njobs, SpecLayout L,
    LongRun *__restrict__ long_runs /* nullable */, uint32_t *__restrict__ n_long_runs, uint32_t long_cap,
    uint32_t job_base /* index of jobs[0] in the batch (LongRun.job counts from the batch's first job) */) {
  const uint64_t t = (uint64_t)blockIdx.x * (K4_THREADS / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const uint32_t job = (uint32_t)(t / L.kmax), k = (uint32_t)(t % L.kmax);
  if (job >= (uint32_t)njobs) return;
  const Job J = jobs[job];
  const SpecGeom g = spec_geom(J.start, J.len, L.sub_block);
  if (k >= (uint32_t)g.nblocks) return;
  const BlockDesc d = desc[t];
  const uint64_t o = out_off[job] + d.out_off;
  const uint32_t ns = min(d.n_stitch, L.slot);
  const uint32_t nk = min(d.n_spec, L.slot) - min(d.keep_from, min(d.n_spec, L.slot));
  const Rec *a = st + t * L.slot;
  const Rec *b = rs + t * L.slot + d.keep_from;
  for (uint32_t r = lane; r < ns + nk; r += 64) {
    if (o + r >= out_cap) break;
    const uint4 v = *reinterpret_cast<const uint4 *>(r < ns ? a + r : b + (r - ns));
    uint32_t *dst = out + (o + r) * 3;
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
    if (v.x >= K5_HUGE && long_runs) {
      const uint32_t s = atomicAdd(n_long_runs, 1u);
      if (s < long_cap) long_runs[s] = LongRun{job_base + job, v.w & ~kRecFlushBit, v.x, 0};
    }
  }
}

// ---------------------------------------------------------------------------
// K5: histograms (encoder.go:44-105) over the compacted Match list of a job.
//
// The only sequential pieces of that pass are (a) the running input position
// (sum of Unmatched+Length) and (b) the ring of the last four distances.
// (a) is a prefix sum.  (b): a distance is pushed exactly when it differs from
// the previous match's distance, so the ring seen by
// match j is the last four "flagged" matches before j.  Inside a wave those
// are the top set bits of the flag ballot below the lane (distances fetched
// with cross-lane reads); what is missing comes from the flagged tails of the
// earlier waves of the iteration and from a carried ring, so one workgroup
// barrier per 256 matches suffices and no match reads another match's record
// from memory.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t log2_floor_nz(uint32_t n) { return 31u - __builtin_clz(n); }

// command.go:111-126
__device__ __forceinline__ uint32_t insert_length_code(uint32_t insertlen) {
  if (insertlen < 6) return insertlen;
  if (insertlen < 130) {
    uint32_t nbits = log2_floor_nz(insertlen - 2) - 1;
    return (nbits << 1) + ((insertlen - 2) >> nbits) + 2;
  }
  if (insertlen < 2114) return log2_floor_nz(insertlen - 66) + 10;
  if (insertlen < 6210) return 21;
  if (insertlen < 22594) return 22;
  return 23;
}
// command.go:128-139
__device__ __forceinline__ uint32_t copy_length_code(uint32_t copylen) {
  if (copylen < 10) return copylen - 2;
  if (copylen < 134) {
    uint32_t nbits = log2_floor_nz(copylen - 6) - 1;
    return (nbits << 1) + ((copylen - 6) >> nbits) + 4;
  }
  if (copylen < 2118) return log2_floor_nz(copylen - 70) + 12;
  return 23;
}
// command.go:141-164
__device__ __forceinline__ uint32_t combine_length_codes(uint32_t inscode, uint32_t copycode,
                                                         bool use_last_distance) {
  uint32_t bits64 = (copycode & 7u) | ((inscode & 7u) << 3);
  if (use_last_distance && inscode < 8 && copycode < 16) return copycode < 8 ? bits64 : (bits64 | 64u);
  uint32_t offset = 2 * ((copycode >> 3) + 3 * (inscode >> 3));
  offset = (offset << 5) + 0x40 + ((0x520D40u >> offset) & 0xC0u);
  return offset | bits64;
}
// encoder.go:169-177 (only the symbol is histogrammed)
__device__ __forceinline__ uint32_t distance_symbol(uint32_t distance) {
  uint32_t d = distance + 3;
  uint32_t nbits = log2_floor_nz(d) - 1;
  uint32_t prefix = (d >> nbits) & 1;
  return 2 * (nbits - 1) + prefix + 16;
}

// One record per match for the encoder's second pass (encoder.go:122-154): the
// command symbol (encoder.go:57-63) and the match's distCache entry
// (encoder.go:42, 66-97, 163-177).  Same layout as m4g_code in include/m4g.h.
struct CodeRec {
  uint16_t command;
  uint8_t dist_code; // 0 with n_extra 0 when the match takes no distance code
  uint8_t n_extra;
  uint32_t extra_bits;
};
static_assert(sizeof(CodeRec) == 8, "m4g_code layout");

#ifndef M4G_K5_THREADS
#define M4G_K5_THREADS 512
#endif
constexpr int K5_THREADS = M4G_K5_THREADS;
constexpr int K5_WAVES = K5_THREADS / 64;

// Up to four most recent flagged distances, most recent first.
struct K5Ring {
  uint32_t d[4];
  uint32_t n;
};
// ring of (newer ++ older), truncated to four
__device__ __forceinline__ K5Ring k5_ring_cat(const K5Ring &newer, const K5Ring &older) {
  K5Ring r;
  r.n = min(4u, newer.n + older.n);
#pragma unroll
  for (int t = 0; t < 4; t++) {
    uint32_t v = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if ((uint32_t)t < newer.n) {
        if (u == t) v = newer.d[u];
      } else if ((uint32_t)(t - (int)newer.n) == (uint32_t)u) {
        v = older.d[u];
      }
    }
    r.d[t] = v;
  }
  return r;
}

// CODES: also write one CodeRec per match (codes[] is indexed like matches[]).
template <bool CODES>
__global__ __launch_bounds__(K5_THREADS) void k5_histograms(
    const uint8_t *__restrict__ src_base, const uint32_t *__restrict__ matches /* 3 words each */,
    const Job *__restrict__ jobs, const uint64_t *__restrict__ out_off, uint64_t out_cap,
    uint32_t *__restrict__ histos /* 1024 u32 per job */, CodeRec *__restrict__ codes,
    const uint32_t *__restrict__ errors /* nullable: the error word of the parse that made the lists */) {
  // A parse that dropped records (error bit 1: a compact plan's slots overflowed) leaves holes in the lists;
  // what is in them must not be used as lengths and positions.  The run is repeated (m4g.hip, batch_settle).
  if (errors && (errors[0] & 2u)) return;
  __shared__ uint32_t lit[K5_WAVES][256];
  __shared__ uint32_t cmd[704];
  __shared__ uint32_t dst[64];
  // per-wave summaries and carries, double buffered by iteration parity so one
  // barrier per iteration orders everything
  __shared__ uint32_t wadv[2][K5_WAVES];
  __shared__ K5Ring wring[2][K5_WAVES];
  __shared__ uint32_t cpos[2];
  __shared__ K5Ring cring[2];

  const uint32_t job = blockIdx.x;
  const Job J = jobs[job];
  const uint64_t o0 = out_off[job], o1 = out_off[job + 1];
  uint32_t *h = histos + (size_t)job * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (o1 > out_cap) { // list truncated by the caller's out_cap: no histogram
    for (int k = tid; k < 1024; k += K5_THREADS) h[k] = 0;
    return;
  }
  const uint32_t n = (uint32_t)(o1 - o0);
  const uint32_t *rec = matches + o0 * 3;
  const uint8_t *src = src_base + J.src_off;

  for (int k = tid; k < K5_WAVES * 256; k += K5_THREADS) (&lit[0][0])[k] = 0;
  for (int k = tid; k < 704; k += K5_THREADS) cmd[k] = 0;
  if (tid < 64) dst[tid] = 0;
  if (tid == 0) {
    cpos[0] = J.start; // input position of the first match's literals
    cring[0].n = 0;
  }
  __syncthreads();

  uint32_t *mylit = lit[wid];
  uint32_t par = 0;
  // this thread's match of the next iteration (and, for lane 0, the distance
  // of the match before it), loaded one iteration ahead: the records are the
  // head of every iteration's dependency chain
  uint32_t nx_un = 0, nx_len = 0, nx_dist = 0, nx_prev = 0xFFFFFFFFu;
  auto load_rec = [&](uint32_t jj) {
    nx_un = nx_len = nx_dist = 0;
    nx_prev = 0xFFFFFFFFu;
    if (jj < n) {
      nx_un = rec[3 * jj];
      nx_len = rec[3 * jj + 1];
      nx_dist = rec[3 * jj + 2];
      if (lane == 0 && jj > 0) nx_prev = rec[3 * jj - 1];
    }
  };
  load_rec((uint32_t)tid);
  for (uint32_t base = 0; base < n; base += K5_THREADS, par ^= 1) {
    const uint32_t j = base + tid;
    const uint32_t un = nx_un, len = nx_len, dist = nx_dist, prev0 = nx_prev;
    load_rec(j + K5_THREADS);
    // distance of the previous match (none before the first)
    uint32_t prevd = __shfl_up(dist, 1);
    if (lane == 0) prevd = prev0;
    const bool flag = j < n && len != 0 && dist != prevd;
    const bool use_last = j < n && dist == prevd;

    // inclusive prefix sum of (un + len) inside the wave
    const uint32_t adv = un + len;
    uint32_t incl = adv;
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    // this wave's flagged distances: the top set bits of the ballot
    const unsigned long long F = __ballot(flag);
    K5Ring mine;
    {
      unsigned long long f = F;
      mine.n = min(4u, (uint32_t)__builtin_popcountll(F));
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int l = f ? 63 - __builtin_clzll(f) : 0;
        mine.d[t] = __shfl(dist, l);
        f &= ~(1ull << l);
      }
    }
    if (lane == 63) wadv[par][wid] = incl;
    if (lane == 0) wring[par][wid] = mine;
    __syncthreads();

    // carries from the earlier iterations and the earlier waves of this one
    uint32_t before = cpos[par];
    K5Ring R = cring[par];
    for (int w = 0; w < wid; w++) {
      before += wadv[par][w];
      R = k5_ring_cat(wring[par][w], R);
    }
    const uint32_t ls = before + incl - adv;
    if (tid == K5_THREADS - 1) {
      cpos[par ^ 1] = before + incl;
      cring[par ^ 1] = k5_ring_cat(mine, R);
    }

    // ring seen by this lane: flagged lanes below it, then R.  -10 = none.
    int64_t ring[4];
    {
      unsigned long long f = F & ((1ull << lane) - 1);
      const uint32_t c = (uint32_t)__builtin_popcountll(f);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int l = f ? 63 - __builtin_clzll(f) : 0;
        const uint32_t dv = __shfl(dist, l);
        f &= ~(1ull << l);
        if ((uint32_t)t < c) {
          ring[t] = (int64_t)dv;
        } else {
          const uint32_t u = (uint32_t)t - c;
          ring[t] = u < R.n ? (int64_t)(u == 0 ? R.d[0] : u == 1 ? R.d[1] : u == 2 ? R.d[2] : R.d[3])
                            : -10;
        }
      }
    }

    if (j < n) {
      uint32_t ic = insert_length_code(un);
      uint32_t cc = len ? copy_length_code(len) : 2u;
      uint32_t command = combine_length_codes(ic, cc, use_last);
      atomicAdd(&cmd[command], 1u);
      CodeRec cr{(uint16_t)command, 0, 0, 0};
      if (command >= 128 && len != 0) {
        const int64_t D = dist, d3 = ring[0];
        uint32_t code;
        if (D == d3) code = 0;
        else if (D == ring[1]) code = 1;
        else if (D == ring[2]) code = 2;
        else if (D == ring[3]) code = 3;
        else if (D == d3 - 1) code = 4;
        else if (D == d3 + 1) code = 5;
        else if (D == d3 - 2) code = 6;
        else if (D == d3 + 2) code = 7;
        else if (D == d3 - 3) code = 8;
        else if (D == d3 + 3) code = 9;
        else code = distance_symbol(dist);
        atomicAdd(&dst[code], 1u);
        if (CODES) {
          cr.dist_code = (uint8_t)code;
          if (code >= 16) { // getDistanceCode, encoder.go:169-177
            const uint32_t dd = dist + 3, nbits = log2_floor_nz(dd) - 1;
            cr.n_extra = (uint8_t)nbits;
            cr.extra_bits = dd - ((2 + ((dd >> nbits) & 1)) << nbits);
          }
        }
      }
      if (CODES) codes[o0 + j] = cr;
    }
    // literals: short runs per lane (8 bytes per load), long runs by the whole wave
    const bool longrun = un >= 64;
    if (!longrun) {
      for (uint32_t k = 0; k < un; k += 8) {
        const uint32_t m = min(un - k, 8u);
        if (ls + k + 8 <= J.len) {
          uint64_t w;
          __builtin_memcpy(&w, src + ls + k, 8);
          for (uint32_t b = 0; b < m; b++, w >>= 8) atomicAdd(&mylit[w & 0xFF], 1u);
        } else {
          for (uint32_t b = 0; b < m; b++) atomicAdd(&mylit[src[ls + k + b]], 1u);
        }
      }
    }
    unsigned long long lm = __ballot(longrun && un < K5_HUGE); // longer ones: k5b_long_literals
    while (lm) {
      int l = __builtin_ctzll(lm);
      lm &= lm - 1;
      uint32_t u = __shfl(un, l), s0 = __shfl(ls, l);
      for (uint32_t k = lane; k < u; k += 64) atomicAdd(&mylit[src[s0 + k]], 1u);
    }
  }
  __syncthreads();

  for (int k = tid; k < 256; k += K5_THREADS) {
    uint32_t s = 0;
    for (int w = 0; w < K5_WAVES; w++) s += lit[w][k];
    h[k] = s;
  }
  for (int k = tid; k < 704; k += K5_THREADS) h[256 + k] = cmd[k];
  if (tid < 64) h[960 + tid] = dst[tid];
}

// Literal runs set aside by k5_histograms.  Work item id = run * segs_per_run +
// segment (segs_per_run covers the longest possible run); a fixed grid strides
// over the items, so no host round trip is needed to size the launch.
__global__ __launch_bounds__(K5B_THREADS) void k5b_long_literals(
    const uint8_t *__restrict__ src_base, const Job *__restrict__ jobs,
    const LongRun *__restrict__ long_runs, const uint32_t *__restrict__ n_long_runs,
    uint32_t long_cap, uint32_t segs_per_run, const uint64_t *__restrict__ out_off, uint64_t out_cap,
    uint32_t *__restrict__ histos, const uint32_t *__restrict__ errors /* nullable, as in k5_histograms */) {
  if (errors && (errors[0] & 2u)) return;
  __shared__ uint32_t lit[K5B_THREADS / 64][256];
  const int tid = threadIdx.x, wid = tid >> 6;
  const uint32_t nruns = min(*n_long_runs, long_cap);
  const uint64_t items = (uint64_t)nruns * segs_per_run;
  for (uint64_t id = blockIdx.x; id < items; id += gridDim.x) {
    const LongRun R = long_runs[id / segs_per_run];
    const uint32_t seg = (uint32_t)(id % segs_per_run);
    if ((uint64_t)seg * K5B_SEG >= R.count) continue; // uniform for the workgroup
    if (out_off[R.job + 1] > out_cap) continue;        // truncated list: k5 reports no histogram
    for (int k = tid; k < (K5B_THREADS / 64) * 256; k += K5B_THREADS) (&lit[0][0])[k] = 0;
    __syncthreads();
    const uint8_t *p = src_base + jobs[R.job].src_off + R.start + (size_t)seg * K5B_SEG;
    const uint32_t n = min(K5B_SEG, R.count - seg * K5B_SEG);
    uint32_t *mylit = lit[wid];
    // the run starts at an arbitrary byte: head up to 16-byte alignment, body, tail
    const uint32_t head = min(n, (uint32_t)((16 - ((uintptr_t)p & 15)) & 15));
    if ((uint32_t)tid < head) atomicAdd(&mylit[p[tid]], 1u);
    const uint32_t body = (n - head) / 16;
    const uint4 *q = reinterpret_cast<const uint4 *>(p + head);
    for (uint32_t k = tid; k < body; k += K5B_THREADS) {
      const uint4 v = q[k];
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) atomicAdd(&mylit[(w[a] >> (8 * b)) & 0xFF], 1u);
    }
    const uint32_t done = head + body * 16;
    if ((uint32_t)tid < n - done) atomicAdd(&mylit[p[done + tid]], 1u);
    __syncthreads();
    uint32_t *h = histos + (size_t)R.job * 1024;
    for (int k = tid; k < 256; k += K5B_THREADS) {
      uint32_t c = 0;
      for (int w = 0; w < K5B_THREADS / 64; w++) c += lit[w][k];
      if (c) atomicAdd(&h[k], c);
    }
    __syncthreads();
  }
}

} // namespace m4g
#include "k6_encode.h"
#include "k6_flate.h"
namespace m4g {

// ---------------------------------------------------------------------------
// KS: streaming helpers.
// ---------------------------------------------------------------------------
// m4.go:92-98: every table entry v -> max(v - delta, 0)
__global__ void ks_rebase_table(uint32_t *__restrict__ table, uint32_t n, uint32_t delta) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    uint32_t v = table[k];
    table[k] = v > delta ? v - delta : 0u;
  }
}

} // namespace m4g
