// k1_xchg.h -- K1 (chain build) with an ordered LDS exchange instead of a sort.
// Included by kernels.h after the K1 helpers.
//
// chain[i] = i - table[h(i)], table[h(i)] = i  (m4.go:140-147) for every
// inserted position in order is one atomic exchange per position, provided the
// exchanges on one table entry happen in position order.  Two hardware facts
// give that order without sorting anything:
//   * the lanes of ONE ds_wrxchg_rtn_b32 instruction that hit the same LDS
//     address are performed in ascending lane order
//     (tools/probes/lds_xchg_order.hip: 134 M exchanges, 78 M of them behind an
//     earlier lane of the same instruction, 0 deviations; libm4g repeats a short
//     form of the probe -- kx_order_selftest below -- once per device when a
//     context or plan is created and uses the sorting kernel k1_chain_build
//     instead if it ever fails);
//   * the LDS instructions of one wave are performed in program order.
// So the keys of a tile only have to reach ONE wave in position order:
//   * 15 PRODUCER waves hash the tile.  Wave w owns KX_WAVE_SPAN consecutive
//     positions (rows of 64) and reads their 8-byte windows straight from
//     global memory (unaligned 8-byte loads, one tile ahead).  It keeps the
//     ~1/nparts of its positions that belong to this workgroup's table
//     partition and compacts their keys, in position order (ballot + mbcnt),
//     into ITS segment of the tile's key queue;
//   * the CONSUMER wave (wave 15) walks the segments of the PREVIOUS tile in
//     wave order, 64 keys per exchange instruction, and overwrites each key
//     with its chain entry;
//   * two tiles later every producer picks its chain entries up from its own
//     segment and stores them, 64 consecutive positions per wave instruction.
// A segment is written and read back by one wave only, so the single LDS-only
// barrier per tile is the hand-over between producers and consumer.
// Any multiplicity of a hash inside a tile (zero pages, short periods, log
// records) is handled by the LDS atomic unit itself: no run folding, no queue
// overflow, no fallback path.
#pragma once

namespace m4g {

constexpr int KX_THREADS = 1024;
constexpr int KX_PROD_WAVES = 15;                       // wave 15 consumes
#ifndef M4G_KX_PER
#define M4G_KX_PER 4
#endif
constexpr int KX_PER = M4G_KX_PER;                      // rows of 64 positions per producer wave and tile
constexpr int KX_WAVE_SPAN = 64 * KX_PER;               // consecutive positions of one producer wave
constexpr int KX_SUB = KX_PROD_WAVES * KX_WAVE_SPAN;    // positions per tile (3840)
// key = index within the partition << 17 | position in tile << cb | check bits (cb of them)
constexpr int KX_KEY_IDX_SHIFT = 17;
constexpr int KX_KEY_REL_SHIFT = 5;
constexpr uint32_t KX_KEY_REL_MASK = 0xFFFu;
static_assert(KX_SUB <= 4096, "tile position must fit the key");
static_assert(K1_MAX_CHECK_BITS <= KX_KEY_REL_SHIFT, "check bits must fit the key");
constexpr int KX_U = 5; // exchange instructions the consumer keeps in flight

// dynamic LDS of k1x_chain_build: table | 2 key queues | 2 x 16 wave counts
constexpr size_t k1x_lds_bytes(int part_bits) {
  return ((size_t)1 << part_bits) * 4 + 2 * (size_t)KX_SUB * 4 + 2 * 16 * 4;
}
static_assert(k1x_lds_bytes(15) <= 160 * 1024, "LDS of one CU");

// Self-test of the ordering k1x_chain_build relies on (run once per device when
// a context or plan is created): 4 waves x 64 exchange instructions whose lanes
// collide on purpose; a lane must get the value of the nearest lower lane of the
// same instruction with the same address, or the entry's earlier content, and
// the entry must end up with the value of the highest such lane.
__global__ __launch_bounds__(256) void kx_order_selftest(uint32_t *__restrict__ mismatches) {
  __shared__ uint32_t tab[4][64];
  __shared__ uint32_t tadd[4][64]; // the same check for add-with-return (k1_split.h, counting sort)
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *t = tab[wave], *ta = tadd[wave];
  t[lane] = 0;
  ta[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t bad = 0;
  for (uint32_t it = 0; it < 64; it++) {
    const uint32_t range = 1u + (it * 7u + wave * 3u) % 64u;       // 1 .. 64 distinct addresses
    const uint32_t idx = ((lane + it) * 2654435761u >> 7) % range; // many lanes per address
    const uint32_t val = (it << 8 | lane) + 1;
    const uint32_t before = t[idx], before_a = ta[idx];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t got = __hip_atomic_exchange(&t[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t got_a = __hip_atomic_fetch_add(&ta[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t after = t[idx];
    uint32_t expect = before, expect_after = val, below = 0;
    for (uint32_t l = 0; l < 64; l++) {
      const uint32_t li = __shfl(idx, l), lv = __shfl(val, l);
      if (li == idx && l < lane) expect = lv, below++;
      if (li == idx && l > lane) expect_after = lv;
    }
    if (got != expect || after != expect_after || got_a != before_a + below) bad++;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (bad) atomicAdd(mismatches, bad);
}

// XH (batches for the Bargain match finders, bargain.h): the hash multiplier is
// an argument -- ((u << s) * prime) >> (64 - bits) of bargain1.go:210-212 is
// ((u & mask) * (prime << s)) >> (64 - bits) -- and every job is a stream cut into
// blocks of `batch_block` bytes (>= KX_SUB + 11) whose last xtail (7, or 11 for
// Bargain3) positions are not inserted (bargain1.go:136-139, bargain3.go:153),
// as in the STREAM form; a job may be a SEGMENT of a long stream that first inserts the
// positions [start, store_from) without chain entries (table priming).  xmul2 != 0: Bargain3's 12-byte hash
// (u * mul + uint64(e) * (xmul2 << 32), bargain3.go:276-278; e = the 4 bytes at i + 8).
template <bool STREAM, bool XH = false>
__global__ __launch_bounds__(KX_THREADS) void k1x_chain_build(
    const uint8_t *__restrict__ src_base, uint64_t src_total, uint32_t *__restrict__ chain_base,
    const Job *__restrict__ jobs, uint32_t *__restrict__ table_io, int njobs,
    int hash_len, int table_bits, int part_bits, int check_bits, int dbg,
    const StreamInfo *__restrict__ si, uint32_t *__restrict__ table_out,
    uint32_t xmul_lo, uint32_t xmul_hi, uint32_t batch_block, uint32_t xmul2, uint32_t xtail) {
  extern __shared__ uint32_t lds[];
  const uint32_t part_size = 1u << part_bits;
  uint32_t *table = lds;
  uint32_t *buf = lds + part_size;               // 2 x KX_SUB keys / chain entries, by tile parity
  uint32_t *cnt = buf + 2 * KX_SUB;              // 2 x 16 keys per producer wave, by tile parity

  const int nparts = 1 << (table_bits - part_bits);
  // XCD-aware placement: blocks b and b+8 share an XCD, so give the nparts
  // partitions of one job the same b % 8 (they read the same input bytes).
  const uint32_t b = blockIdx.x;
  const uint32_t x = b & 7, r = b >> 3;
  const uint32_t q = r % nparts;
  const uint32_t job = (r / nparts) * 8 + x;
  if (job >= (uint32_t)njobs) return;

  const Job J = jobs[job];
  uint32_t *chain = chain_base + J.chain_off;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const bool consumer = wave == (uint32_t)KX_PROD_WAVES;
  if (consumer) __builtin_amdgcn_s_setprio(3); // the one wave everybody waits for

  const bool load_table = table_io && (J.k1_flags & kJobLoadTable);
  const bool store_table = table_io && (J.k1_flags & kJobStoreTable);
  for (uint32_t k = tid; k < part_size; k += KX_THREADS)
    table[k] = load_table ? table_io[(q << part_bits) + k] << check_bits : 0u; // plain positions, check bits unknown
  if (tid < 32) cnt[tid] = 0;
  const uint32_t st_x0 = STREAM ? si->x0 : 0u, st_xend = STREAM ? si->x_end : 0xFFFFFFFFu;
  const uint32_t st_block = STREAM ? si->block_size : 0u, st_nbp = STREAM ? si->nbp : 1u;
  uint32_t bpi = 0; // breakpoint in force at the current tile
  const uint32_t store_from = (STREAM || XH) ? J.store_from : 0u;

  // hash = ((v & mask) * kHashMul64) >> (64 - table_bits), in 32-bit pieces
  const uint64_t mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
  const uint32_t mask_lo = (uint32_t)mask, mask_hi = (uint32_t)(mask >> 32);
  const uint32_t mul_lo = XH ? xmul_lo : (uint32_t)kHashMul64, mul_hi = XH ? xmul_hi : (uint32_t)(kHashMul64 >> 32);
  const uint32_t shift = 32 - table_bits;
  const uint32_t idx_mask = part_size - 1;
  const uint32_t cb = (uint32_t)check_bits; // <= K1_MAX_CHECK_BITS and positions << cb fit 32 bits
  const uint32_t cmask = (1u << cb) - 1;
  const uint32_t tsk = XH ? xtail : 7u;
  const uint32_t end = J.len > tsk ? J.len - tsk : 0; // positions < end are inserted
  const uint32_t ntiles = J.start < end ? (end - J.start + KX_SUB - 1) / KX_SUB : 0;

  // the 8 bytes at this lane's KX_PER positions of the tile at `base`
  // (positions past the buffer read its last 8 bytes; they are never used)
  const uint32_t my_rel = wave * KX_WAVE_SPAN + lane; // row 0
  auto load_windows = [&](uint32_t base, uint2 *v) {
    const uint64_t a0 = J.src_off + base; // wave-uniform
    if (a0 + KX_SUB + 8 <= src_total) { // whole tile inside the buffer: scalar base + lane offset
      const uint8_t *tp = src_base + a0;
#pragma unroll
      for (int j = 0; j < KX_PER; j++) __builtin_memcpy(&v[j], tp + (my_rel + j * 64), 8);
    } else {
#pragma unroll
      for (int j = 0; j < KX_PER; j++) {
        uint64_t a = a0 + my_rel + j * 64;
        if (a + 8 > src_total) a = src_total - 8;
        __builtin_memcpy(&v[j], src_base + a, 8);
      }
    }
  };
  uint2 vn[KX_PER]; // windows of the next tile
#pragma unroll
  for (int j = 0; j < KX_PER; j++) vn[j] = make_uint2(0, 0);
  if (!consumer && ntiles > 0) load_windows(J.start, vn);
  lds_barrier(); // table in place

  // where this lane's keys went in the two previous tiles (queue slots, rows as bits)
  uint32_t slot1[KX_PER], slot2[KX_PER], in1 = 0, in2 = 0;
#pragma unroll
  for (int j = 0; j < KX_PER; j++) slot1[j] = slot2[j] = 0;
  // origin in force at the tile start and the next change (at most one per
  // tile: slides are more than MaxDistance >= KX_SUB positions apart); p_*: of
  // the previous tile, which is the one the consumer works on
  uint32_t org_cur = 0, org_next_x = 0xFFFFFFFFu, org_next = 0;
  uint32_t p_org_cur = 0, p_org_next_x = 0xFFFFFFFFu, p_org_next = 0;

#pragma unroll 1
  for (uint32_t t = 0; t < ntiles + 2; t++) {
    const uint32_t sb = J.start + t * KX_SUB; // (may run past `end` in the two drain iterations)
    const uint32_t par = t & 1;
    uint32_t *bufc = buf + par * KX_SUB; // holds the chain entries of tile t-2, then receives the keys of tile t
    uint32_t blk_end = 0xFFFFFFFFu;      // end of the block the tile starts in: its last 7 positions are not inserted
    if (STREAM) {
      p_org_cur = org_cur, p_org_next_x = org_next_x, p_org_next = org_next;
      while (bpi + 1 < st_nbp && si->bp_x[bpi + 1] <= sb) bpi++;
      org_cur = si->bp_org[bpi];
      org_next_x = 0xFFFFFFFFu;
      if (bpi + 1 < st_nbp) {
        org_next_x = si->bp_x[bpi + 1];
        org_next = si->bp_org[bpi + 1];
      }
      if (sb >= st_x0) blk_end = min(st_xend, st_x0 + ((sb - st_x0) / st_block + 1) * st_block);
      else blk_end = st_x0; // history of earlier calls is never re-inserted (segments start at x0)
    }
    if (XH && batch_block) blk_end = min(J.len, (sb / batch_block + 1) * batch_block);

    uint32_t key[KX_PER], in0 = 0;
    unsigned long long bal[KX_PER];
    bool inr[KX_PER]; // row j of this lane has a key for this partition
#pragma unroll
    for (int j = 0; j < KX_PER; j++) inr[j] = false;
    if (!consumer) {
      uint2 v[KX_PER];
#pragma unroll
      for (int j = 0; j < KX_PER; j++) v[j] = vn[j];
      if (t + 1 < ntiles) load_windows(sb + KX_SUB, vn);
      // chain entries of tile t-2 (the consumer finished it one iteration ago):
      // read now, stored after the hashing below has covered the LDS latency
      // (slots of rows without a key are <= KX_SUB: inside the LDS image)
      const bool pickup = t >= 2 && !(dbg & 1);
      uint32_t ce[KX_PER];
      if (pickup) {
#pragma unroll
        for (int j = 0; j < KX_PER; j++) ce[j] = bufc[slot2[j]];
      }
      // hash tile t
      uint32_t total = 0;
      const uint32_t lim = end > sb ? end - sb : 0u; // positions of the tile below this are inserted
      if (t < ntiles && !(dbg & 4)) {
#pragma unroll
        for (int j = 0; j < KX_PER; j++) {
          const uint32_t rel = my_rel + j * 64;
          const uint32_t lo = v[j].x & mask_lo, hi = v[j].y & mask_hi;
          uint32_t hh = __umulhi(lo, mul_lo) + lo * mul_hi + hi * mul_lo;
          const uint32_t pos = sb + rel;
          if (XH && xmul2) { // the high word of uint64(e) * (xmul2 << 32) is e * xmul2 mod 2^32
            const uint64_t ea = J.src_off + pos + 8;
            uint32_t e = 0;
            if (ea + 4 <= src_total) __builtin_memcpy(&e, src_base + ea, 4);
            hh += e * xmul2;
          }
          const uint32_t h = hh >> shift;
          // the last 7 positions of a block are never inserted (m4.go:112).  blk_end is the end of
          // the block the tile starts in; a tile can also hold the end of the call's LAST block when
          // that one is shorter than a tile (every other block is longer than a tile)
          const bool in = rel < lim && (h >> part_bits) == q &&
                          !((STREAM || XH) && ((pos + tsk >= blk_end && pos < blk_end) || pos + 7 >= st_xend));
          // check bits of the first 4 bytes (never 0: that value marks table
          // entries whose bytes are not known); cb >= 1 (launch_k1)
          const uint32_t ck = max(1u, (v[j].x * 0x9E3779B1u) >> (32 - cb));
          // the low 17 key bits + (tile start << cb) are the table value of the position
          key[j] = ((h & idx_mask) << KX_KEY_IDX_SHIFT) | (rel << cb) | ck;
          inr[j] = in;
          bal[j] = __builtin_amdgcn_ballot_w64(in);
          if (in) in0 |= 1u << j;
          total += (uint32_t)__builtin_popcountll(bal[j]);
        }
      }
      if (pickup) {
        uint32_t *cp = chain + (sb - 2 * KX_SUB); // wave-uniform
#pragma unroll
        for (int j = 0; j < KX_PER; j++)
          if ((in2 >> j & 1) && (!(STREAM || XH) || sb - 2 * KX_SUB + my_rel + j * 64 >= store_from))
            cp[my_rel + j * 64] = ce[j];
      }
      if (lane == 0) cnt[par * 16 + wave] = total;
    } else if (t >= 1 && t - 1 < ntiles && !(dbg & 2)) {
      // CONSUME tile t-1: one exchange per key, in queue (= position) order.
      // This wave's instruction stream is the critical path of the workgroup:
      // everything a key needs besides the exchange itself is 4 VALU
      // instructions before it and the conversion of the returned table value
      // into a chain entry after it.
      uint32_t *bufp = buf + (par ^ 1) * KX_SUB + lane;
      uint32_t n = lane < (uint32_t)KX_PROD_WAVES ? cnt[(par ^ 1) * 16 + lane] : 0u;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) n += __shfl_xor(n, o);
      n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
      const uint32_t vbase = (sb - KX_SUB) << cb; // table value = vbase + low key bits
      const uint32_t low_mask = (1u << KX_KEY_IDX_SHIFT) - 1;
      auto convert = [&](uint32_t k, uint32_t old) -> uint32_t {
        const uint32_t val = vbase + (k & low_mask);
        // a predecessor at or below the buffer origin counts as "empty" (m4.go:152, 92-98)
        const uint32_t pos = val >> cb, pp = old >> cb, pck = old & cmask;
        const uint32_t org = STREAM ? (pos >= p_org_next_x ? p_org_next : p_org_cur) : 0u;
        // check bits 0 = unknown (entry loaded from a persistent table): may be equal
        return pp > org ? ((pos - pp) | ((pck == 0 || pck == (val & cmask)) ? kChainMaybe : 0u)) : 0u;
      };
      uint32_t e0 = 0;
#pragma unroll 1
      for (; e0 + 64 * KX_U <= n; e0 += 64 * KX_U) { // full batches: no masking
        uint32_t k[KX_U], old[KX_U];
#pragma unroll
        for (int u = 0; u < KX_U; u++) k[u] = bufp[e0 + u * 64];
#pragma unroll
        for (int u = 0; u < KX_U; u++)
          old[u] = __hip_atomic_exchange(&table[k[u] >> KX_KEY_IDX_SHIFT], vbase + (k[u] & low_mask),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int u = 0; u < KX_U; u++) bufp[e0 + u * 64] = convert(k[u], old[u]);
      }
#pragma unroll 1
      for (; e0 < n; e0 += 64) { // the rest, one batch at a time
        const bool act = e0 + lane < n;
        const uint32_t k = act ? bufp[e0] : 0u;
        uint32_t old = 0;
        if (act)
          old = __hip_atomic_exchange(&table[k >> KX_KEY_IDX_SHIFT], vbase + (k & low_mask),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (act) bufp[e0] = convert(k, old);
      }
    }
    lds_barrier(); // wave counts of tile t complete; chain entries of tile t-1 in place
    if (!consumer && t < ntiles) {
      // ordered compaction: queue slot = keys of earlier waves + of earlier rows + of earlier lanes
      // keys of the waves before this one: sum of cnt[0 .. wave) over lanes 0..15 of row 0
      uint32_t c = lane < wave ? cnt[par * 16 + (lane & 15)] : 0u;
      c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x111, 0xf, 0xf, true); // row_shr:1
      c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x112, 0xf, 0xf, true); // row_shr:2
      c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x114, 0xf, 0xf, true); // row_shr:4
      c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x118, 0xf, 0xf, true); // row_shr:8
      uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)c, 15);
#pragma unroll
      for (int j = 0; j < KX_PER; j++) {
        slot2[j] = slot1[j];
        // keys of earlier waves and rows + keys of the lanes below
        slot1[j] = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[j] >> 32),
                                             __builtin_amdgcn_mbcnt_lo((uint32_t)bal[j], base));
        if (inr[j]) bufc[slot1[j]] = key[j];
        base += (uint32_t)__builtin_popcountll(bal[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < KX_PER; j++) slot2[j] = slot1[j];
    }
    in2 = in1;
    in1 = in0;
    lds_barrier(); // keys of tile t queued
  }

  if (store_table) { // entries relative to the origin after the call, at or below it = empty
    const uint32_t rebase = STREAM ? si->final_org : 0u;
    for (uint32_t k = tid; k < part_size; k += KX_THREADS) {
      const uint32_t v = table[k] >> check_bits; // the persistent table holds plain positions
      (table_out ? table_out : table_io)[(q << part_bits) + k] = v > rebase ? v - rebase : 0u;
    }
  }
}

} // namespace m4g
