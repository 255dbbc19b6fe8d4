// k1_split.h -- K1 (chain build) for batches of independent chunks, split so
// that every position is hashed ONCE.  Included by kernels.h after k1_xchg.h.
//
// k1x_chain_build (k1_xchg.h) runs one workgroup per (chunk, table partition):
// each of the 4 partition workgroups of a chunk streams the whole chunk and
// hashes every position, only to keep the quarter that falls into its partition
// (measured: instruction-issue bound, 34 ms per 4 GiB).  Here the two halves of
// that kernel become two kernels:
//
//   k1p_partition  one workgroup per few tiles of KP_TILE positions, any chunk.
//                  Hashes each position once and sorts the tile's keys by BIN =
//                  top bits of the hash = (table partition, owner wave inside
//                  it): a stable 64-way counting sort in LDS (per-wave counters
//                  bumped with ds_add_rtn in position order, a 16 x 64 prefix
//                  sum, a staged coalesced copy-out).  Output per tile: its keys
//                  grouped by bin, in position order inside a bin, plus 64
//                  {offset, count} words.  The keys live where the chunk's RS
//                  record slots will be (dead until K3).
//   k1c_consume    one workgroup per (chunk, table partition) as before, but all
//                  16 waves are consumers: wave w owns 1/16 of the partition's
//                  table entries and walks ITS bin of every tile: <= 64 keys per
//                  ordered LDS exchange (k1_xchg.h explains why the exchange
//                  yields the predecessor in position order), chain entries
//                  scattered into an LDS tile image, and after ONE barrier per
//                  tile the image's valid words are stored, 64 consecutive
//                  positions per wave instruction, as k1x does.
// No wave waits for another one's queue, nothing is hashed twice, and a
// multiplicity of one hash (zero pages) only lengthens one wave's bin.
//
// Both kernels rest on the LDS ordering the exchange kernel rests on (same-address
// lanes of one LDS atomic are performed in ascending lane order; LDS operations of
// one wave in program order); kx_order_selftest checks the add-with-return form
// used by the counting sort as well.  Results are identical to k1x_chain_build's
// (tests: test_gpu_scale.py::test_k1_variants_agree_on_1gib_of_mixed_input).
#pragma once

namespace m4g {

constexpr int KP_THREADS = 1024;
constexpr int KP_ROWS = 3;                          // rows of 64 positions per wave and tile
constexpr int KP_WAVE_SPAN = 64 * KP_ROWS;          // consecutive positions of one wave
constexpr int KP_TILE = KP_THREADS * KP_ROWS;       // 3072 positions: ~48 keys per bin on average
constexpr int KP_BINS = 64;                         // <= 4 partitions x 16 owner waves
constexpr int KP_CSTRIDE = KP_BINS + 1;             // odd stride: a wave's counters spread over the banks
#ifndef M4G_KP_TPW
#define M4G_KP_TPW 8 // (1024 x 1 MiB of text, k1p per launch: 2 / 4 / 8 / 16 / 32 tiles: 2.64 / 2.64 / 2.55 / 2.57 / 2.53 ms)
#endif
constexpr int KP_TPW = M4G_KP_TPW;                         // tiles per k1p workgroup (consecutive, one job)
constexpr int KP_KEY_IDX_SHIFT = 17;                // key = index in the owner's slice << 17 | rel << cb | check bits
constexpr uint32_t KP_KEY_SET_ONLY = 0x80000000u;   // key flag: exchange, but the chain entry is already stored
constexpr uint32_t KP_EMPTY = 0xFFFFFFFFu;          // tile image word without a chain entry (no entry has this value)
static_assert(KP_TILE <= 4096, "tile position must fit the key");

// words of the per-tile header: bin b of tile t of job j -> hdr[((j * kt_max) + t) * KP_BINS + b]
// = first key of the bin inside the tile's key block << 16 | number of keys
__host__ __device__ constexpr uint32_t kp_tiles(uint32_t len) {
  return len >= 8 ? (len - 7 + KP_TILE - 1) / KP_TILE : 0;
}
constexpr size_t k1c_lds_bytes(int part_bits) { return ((size_t)1 << part_bits) * 4 + 2 * (size_t)KP_TILE * 4 + KP_THREADS * 4; }
static_assert(k1c_lds_bytes(15) <= 160 * 1024, "LDS of one CU");

__global__ __launch_bounds__(KP_THREADS, 8) void k1p_partition(
    const uint8_t *__restrict__ src_base, uint64_t src_total, uint32_t *__restrict__ keys_base,
    uint32_t *__restrict__ hdr_base, uint32_t *__restrict__ chain_base, const Job *__restrict__ jobs, int njobs,
    uint32_t kt_max, uint32_t groups_per_job, int hash_len, int table_bits, int bin_bits, int check_bits,
    int fold /* 0: every inserted position gets a key (k1d_depth, k2_eval.h, counts them) */) {
  __shared__ uint32_t cnt[16 * KP_CSTRIDE]; // [wave][bin]: keys of the wave in the bin, then their first slot
  __shared__ uint32_t tot[KP_BINS];
  __shared__ uint32_t stage[KP_TILE];
  const uint32_t job = blockIdx.x / groups_per_job, grp = blockIdx.x % groups_per_job;
  if (job >= (uint32_t)njobs) return;
  const Job J = jobs[job];
  const uint32_t end = J.len >= 8 ? J.len - 7 : 0; // positions < end are inserted (m4.go:112)
  const uint32_t ntiles = kp_tiles(J.len);
  const uint32_t t0 = grp * KP_TPW;
  if (t0 >= ntiles) return;
  const uint32_t t1 = min(t0 + (uint32_t)KP_TPW, ntiles);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  // hash = ((v & mask) * kHashMul64) >> (64 - table_bits), in 32-bit pieces (k1_xchg.h)
  const uint64_t mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
  const uint32_t mask_lo = (uint32_t)mask, mask_hi = (uint32_t)(mask >> 32);
  const uint32_t mul_lo = (uint32_t)kHashMul64, mul_hi = (uint32_t)(kHashMul64 >> 32);
  const uint32_t shift = 32 - table_bits;
  const uint32_t slice_bits = (uint32_t)(table_bits - bin_bits);
  const uint32_t slice_mask = (1u << slice_bits) - 1;
  const uint32_t cb = (uint32_t)check_bits;
  uint32_t *mycnt = cnt + wave * KP_CSTRIDE;
  const uint32_t my_rel = wave * KP_WAVE_SPAN + lane; // row 0

  auto load_windows = [&](uint32_t base, uint2 *v) {
    const uint64_t a0 = J.src_off + base; // wave-uniform
    if (a0 + KP_TILE + 8 <= src_total) {  // whole tile inside the buffer: scalar base + lane offset
      const uint8_t *tp = src_base + a0;
#pragma unroll
      for (int j = 0; j < KP_ROWS; j++) __builtin_memcpy(&v[j], tp + (my_rel + j * 64), 8);
    } else {
#pragma unroll
      for (int j = 0; j < KP_ROWS; j++) {
        uint64_t a = a0 + my_rel + j * 64;
        if (a + 8 > src_total) a = src_total - 8; // positions past the buffer are never used
        __builtin_memcpy(&v[j], src_base + a, 8);
      }
    }
  };
  uint2 vn[KP_ROWS];
  load_windows(t0 * KP_TILE, vn);

#pragma unroll 1
  for (uint32_t t = t0; t < t1; t++) {
    const uint32_t sb = t * KP_TILE;
    for (uint32_t k = tid; k < 16 * KP_CSTRIDE; k += KP_THREADS) cnt[k] = 0;
    uint2 v[KP_ROWS];
#pragma unroll
    for (int j = 0; j < KP_ROWS; j++) v[j] = vn[j];
    if (t + 1 < t1) load_windows(sb + KP_TILE, vn);
    lds_barrier(); // counters cleared; the previous tile's stage has been copied out

    uint32_t key[KP_ROWS], bin[KP_ROWS], rank[KP_ROWS];
    const uint32_t lim = end > sb ? end - sb : 0u;
#pragma unroll
    for (int j = 0; j < KP_ROWS; j++) {
      const uint32_t rel = my_rel + j * 64;
      const uint32_t lo = v[j].x & mask_lo, hi = v[j].y & mask_hi;
      const uint32_t h = (__umulhi(lo, mul_lo) + lo * mul_hi + hi * mul_lo) >> shift;
      // check bits of the first 4 bytes (never 0: k1_xchg.h)
      const uint32_t ck = max(1u, (v[j].x * 0x9E3779B1u) >> (32 - cb));
      key[j] = ((h & slice_mask) << KP_KEY_IDX_SHIFT) | (rel << cb) | ck;
      bin[j] = rel < lim ? (h >> slice_bits) : KP_BINS; // KP_BINS: no key
      rank[j] = 0;
      // Runs of neighbouring positions with one hash (zero padding, constant fills) inside this row of 64:
      // every member but the first has its left neighbour as predecessor, so its chain entry is known here
      // and is stored here; the table only has to see the first member (a normal key: it looks the entry
      // up) and the last one (a KP_KEY_SET_ONLY key: it leaves its position there).  A page of zeros
      // becomes 2 keys per row instead of 64, all of them in one bin.
      {
        const uint32_t hv = rel < lim ? h : 0x80000000u | lane; // no key: equal to nobody
        const uint32_t ph = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)hv, 0x138, 0xf, 0xf, false); // wave_shr:1
        const uint32_t pck = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ck, 0x138, 0xf, 0xf, false);
        const uint32_t nh = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)hv, 0x130, 0xf, 0xf, false); // wave_shl:1
        if (fold && hv == ph) { // (position sb + rel >= 1 here; a predecessor at position 0 counts as "empty", m4.go:152)
          chain_base[J.chain_off + sb + rel] = sb + rel > 1 ? (1u | (pck == ck ? kChainMaybe : 0u)) : 0u;
          if (hv == nh) bin[j] = KP_BINS;
          else key[j] |= KP_KEY_SET_ONLY;
        }
      }
      // rows in program order, lanes of a row in ascending order: rank = position order inside (wave, bin)
      if (bin[j] < KP_BINS)
        rank[j] = __hip_atomic_fetch_add(&mycnt[bin[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    lds_barrier(); // all counts in

    // thread (bin, w') = (tid >> 4, tid & 15): exclusive prefix over the waves of a bin, inside a DPP row
    {
      const uint32_t b = tid >> 4, w = tid & 15;
      const uint32_t c0 = cnt[w * KP_CSTRIDE + b];
      uint32_t c = c0;
      c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x111, 0xf, 0xf, true); // row_shr:1
      c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x112, 0xf, 0xf, true); // row_shr:2
      c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x114, 0xf, 0xf, true); // row_shr:4
      c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x118, 0xf, 0xf, true); // row_shr:8
      cnt[w * KP_CSTRIDE + b] = c - c0;
      if (w == 15) tot[b] = c;
    }
    lds_barrier(); // per-wave offsets and bin totals in

    // every wave: exclusive prefix of the 64 bin totals (lane = bin)
    const uint32_t tv = tot[lane];
    uint32_t inc = tv;
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, true);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15);
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 47);
    inc += (lane >= 16 ? r0 : 0u) + (lane >= 32 ? r1 : 0u) + (lane >= 48 ? r2 : 0u);
    const uint32_t binbase = inc - tv;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (wave == 0) hdr_base[((uint64_t)job * kt_max + t) * KP_BINS + lane] = (binbase << 16) | tv;
#pragma unroll
    for (int j = 0; j < KP_ROWS; j++) {
      const uint32_t b = bin[j] < KP_BINS ? bin[j] : 0u;
      const uint32_t bb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b << 2), (int)binbase);
      const uint32_t wo = mycnt[b];
      if (bin[j] < KP_BINS) stage[bb + wo + rank[j]] = key[j];
    }
    lds_barrier(); // the tile's keys, grouped by bin, staged

    uint32_t *kp = keys_base + J.chain_off + sb; // wave-uniform
#pragma unroll
    for (int j = 0; j < KP_ROWS; j++) {
      const uint32_t s = j * KP_THREADS + tid;
      if (s < total) kp[s] = stage[s];
    }
  }
}

// k1c_scatter: the same consumer without the tile image: every wave stores its
// chain entries straight to chain[pos] (64 scattered words per instruction) and
// never meets another wave: no barrier.  A wave works in steps of KS_G tiles:
// the KS_G headers are requested together, then the first 64 keys of its bin in
// each tile, then the exchanges run in tile order.  (Memory latency is paid twice
// per step, not per tile; the loads are pinned in place with empty asm statements
// because the compiler otherwise sinks each one into the branch that uses it and
// waits for it there, one round trip per tile.)
constexpr uint32_t KS_G = 16; // tiles per step
constexpr uint32_t KS_X = 8;  // batches per group in a bin with more than 64 keys
constexpr size_t k1s_lds_bytes(int part_bits) { return ((size_t)1 << part_bits) * 4 + KP_THREADS * 4; }
static_assert(k1s_lds_bytes(15) <= 160 * 1024, "LDS of one CU");

__global__ __launch_bounds__(KP_THREADS) void k1c_scatter(
    const uint32_t *__restrict__ keys_base, const uint32_t *__restrict__ hdr_base,
    uint32_t *__restrict__ chain_base, const Job *__restrict__ jobs, int njobs, uint32_t kt_max,
    int table_bits, int part_bits, int check_bits) {
  extern __shared__ uint32_t lds[];
  const uint32_t part_size = 1u << part_bits;
  uint32_t *table = lds;
  const int nparts = 1 << (table_bits - part_bits);
  const uint32_t b = blockIdx.x;
  const uint32_t x = b & 7, r = b >> 3;
  const uint32_t q = r % nparts;
  const uint32_t job = (r / nparts) * 8 + x;
  if (job >= (uint32_t)njobs) return;
  const Job J = jobs[job];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t ntiles = kp_tiles(J.len);
  if (ntiles == 0) return;
  for (uint32_t k = tid; k < part_size; k += KP_THREADS) table[k] = 0u;
  const uint32_t cb = (uint32_t)check_bits, cmask = (1u << cb) - 1;
  const uint32_t slice_bits = (uint32_t)part_bits - 4;
  uint32_t *slice = table + (wave << slice_bits);
  uint32_t *dummy = table + part_size + tid; // where the exchange of a lane without a key goes
  const uint32_t *hp = hdr_base + (uint64_t)job * kt_max * KP_BINS + q * 16 + wave; // + t * KP_BINS
  const uint32_t *keys = keys_base + J.chain_off;
  uint32_t *chain = chain_base + J.chain_off;
  const uint32_t low_mask = (1u << KP_KEY_IDX_SHIFT) - 1;
  lds_barrier(); // table cleared (cooperatively; from here on a wave only touches its own slice)

  auto entry = [&](uint32_t val, uint32_t old) -> uint32_t {
    // a predecessor at position 0 (the buffer origin) or none counts as "empty" (m4.go:152)
    const uint32_t pos = val >> cb, pp = old >> cb;
    return pp > 0 ? ((pos - pp) | ((old & cmask) == (val & cmask) ? kChainMaybe : 0u)) : 0u;
  };

#pragma unroll 1
  for (uint32_t tb = 0; tb < ntiles; tb += KS_G) {
    uint32_t h[KS_G], k[KS_G];
#pragma unroll
    for (uint32_t u = 0; u < KS_G; u++) { // (tiles past the end re-read the last one and are masked below)
      const uint32_t t = min(tb + u, ntiles - 1);
      h[u] = __builtin_nontemporal_load(hp + (uint64_t)t * KP_BINS + (lane & 0)); // a vector load on purpose
    }
#pragma unroll
    for (uint32_t u = 0; u < KS_G; u++) asm volatile("" : "+v"(h[u]));
#pragma unroll
    for (uint32_t u = 0; u < KS_G; u++) {
      const uint32_t t = min(tb + u, ntiles - 1);
      k[u] = keys[(uint64_t)t * KP_TILE + (h[u] >> 16) + lane]; // lanes past the bin read a few words on: unused
    }
#pragma unroll
    for (uint32_t u = 0; u < KS_G; u++) asm volatile("" : "+v"(k[u]));
#pragma unroll
    for (uint32_t u = 0; u < KS_G; u++) {
      const uint32_t t = tb + u;
      const uint32_t sb = t * KP_TILE;
      const uint32_t n = t < ntiles ? (h[u] & 0xFFFFu) : 0u;
      const uint32_t vbase = sb << cb; // table value = vbase + low key bits = position << cb | check bits
      {
        const bool act = lane < n;
        const uint32_t val = vbase + (k[u] & low_mask);
        uint32_t *p = act ? slice + ((k[u] & ~KP_KEY_SET_ONLY) >> KP_KEY_IDX_SHIFT) : dummy;
        const uint32_t old = __hip_atomic_exchange(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (act && !(k[u] & KP_KEY_SET_ONLY)) chain[val >> cb] = entry(val, old);
      }
      if (n > 64) { // a crowded bin (zero pages, short periods): the rest in groups of KS_X batches
        const uint32_t *kp = keys + sb + (h[u] >> 16);
#pragma unroll 1
        for (uint32_t e0 = 64; e0 < n; e0 += 64 * KS_X) {
          uint32_t kk[KS_X];
#pragma unroll
          for (uint32_t v = 0; v < KS_X; v++) kk[v] = kp[min(e0 + v * 64 + lane, n - 1)];
#pragma unroll
          for (uint32_t v = 0; v < KS_X; v++) asm volatile("" : "+v"(kk[v]));
#pragma unroll
          for (uint32_t v = 0; v < KS_X; v++) {
            const bool act = e0 + v * 64 + lane < n;
            const uint32_t val = vbase + (kk[v] & low_mask);
            uint32_t *p = act ? slice + ((kk[v] & ~KP_KEY_SET_ONLY) >> KP_KEY_IDX_SHIFT) : dummy;
            const uint32_t old = __hip_atomic_exchange(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (act && !(kk[v] & KP_KEY_SET_ONLY)) chain[val >> cb] = entry(val, old);
          }
        }
      }
    }
  }
}

// Loads of a wave, all software-pipelined (a wave that waits for memory holds up the 15 others at the tile's
// barrier):
//   * headers: ONE load per 64 tiles, lane = tile; a tile's header is then a v_readlane (scalar), two blocks
//     of 64 tiles ahead;
//   * the first 128 keys of the wave's bin in every tile, in steps of KC_G tiles, one step ahead (a bin of
//     text holds 48 keys on average and more than 64 in one tile out of six for some wave of the workgroup);
//   * keys 128 .. 128 + 64 * KC_X of a crowded bin (log records: ~900 of a tile's 3072 positions share one
//     hash; zero pages, short periods), one tile ahead into one of two register buffers; whatever lies
//     beyond that is loaded on the spot.
// Loads are issued back to back and pinned with empty asm statements -- the compiler otherwise sinks each load
// into the branch that uses it and waits for it there, one memory round trip per tile.
#ifndef M4G_KC_G
#define M4G_KC_G 8
#endif
constexpr uint32_t KC_G = M4G_KC_G; // tiles per load step
constexpr uint32_t KC_X = 16;       // batches of 64 keys per extras buffer
static_assert(64 % KC_G == 0 && KC_G % 2 == 0, "steps tile a header block; the extras buffers alternate by tile parity");
__global__ __launch_bounds__(KP_THREADS) void k1c_consume(
    const uint32_t *__restrict__ keys_base, const uint32_t *__restrict__ hdr_base,
    uint32_t *__restrict__ chain_base, const Job *__restrict__ jobs, int njobs, uint32_t kt_max,
    int table_bits, int part_bits, int check_bits) {
  extern __shared__ uint32_t lds[];
  const uint32_t part_size = 1u << part_bits;
  uint32_t *table = lds;
  uint32_t *res = lds + part_size; // 2 tile images of chain entries, by tile parity

  const int nparts = 1 << (table_bits - part_bits);
  // XCD-aware placement (k1_xchg.h): the partitions of one job on one XCD -- they write
  // interleaved words of the same chain lines, which that XCD's L2 merges
  const uint32_t b = blockIdx.x;
  const uint32_t x = b & 7, r = b >> 3;
  const uint32_t q = r % nparts;
  const uint32_t job = (r / nparts) * 8 + x;
  if (job >= (uint32_t)njobs) return;

  const Job J = jobs[job];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t ntiles = kp_tiles(J.len);
  if (ntiles == 0) return;

  for (uint32_t k = tid; k < part_size; k += KP_THREADS) table[k] = 0u;
  for (uint32_t k = tid; k < 2 * KP_TILE; k += KP_THREADS) res[k] = KP_EMPTY;

  const uint32_t cb = (uint32_t)check_bits, cmask = (1u << cb) - 1;
  const uint32_t slice_bits = (uint32_t)part_bits - 4;
  uint32_t *slice = table + (wave << slice_bits);
  uint32_t *dummy = res + 2 * KP_TILE + tid; // where the exchange of a lane without a key goes
  const uint32_t *hp = hdr_base + (uint64_t)job * kt_max * KP_BINS + q * 16 + wave; // + t * KP_BINS
  const uint32_t *keys = keys_base + J.chain_off;
  uint32_t *chain = chain_base + J.chain_off;
  const uint32_t low_mask = (1u << KP_KEY_IDX_SHIFT) - 1;

  auto entry = [&](uint32_t val, uint32_t old) -> uint32_t {
    // a predecessor at position 0 (the buffer origin) or none counts as "empty" (m4.go:152)
    const uint32_t pos = val >> cb, pp = old >> cb;
    return pp > 0 ? ((pos - pp) | ((old & cmask) == (val & cmask) ? kChainMaybe : 0u)) : 0u;
  };
  // headers of the 64 tiles from T0 on, lane = tile (tiles past the end repeat the last one; masked at use)
  auto load_hv = [&](uint32_t T0) -> uint32_t { return hp[(uint64_t)min(T0 + lane, ntiles - 1) * KP_BINS]; };
  uint32_t hv = load_hv(0), hvn = load_hv(64), hvnn = load_hv(128);
  // header of tile (first tile of the current block) + i, i < 128, as a scalar
  auto hdr_at = [&](uint32_t i) -> uint32_t {
    return i < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)hv, (int)i)
                  : (uint32_t)__builtin_amdgcn_readlane((int)hvn, (int)(i - 64));
  };
  // keys 0 .. 127 of the wave's bin in the KC_G tiles from tb on (= block tile i0 on); lanes past the bin
  // read on into the next bins (the key buffer ends with a margin): unused
  auto load_keys = [&](uint32_t tb, uint32_t i0, uint32_t *k, uint32_t *kb) {
#pragma unroll
    for (uint32_t u = 0; u < KC_G; u++) {
      const uint32_t *kp = keys + (uint64_t)min(tb + u, ntiles - 1) * KP_TILE + (hdr_at(i0 + u) >> 16);
      k[u] = kp[lane];
      kb[u] = kp[64 + lane];
    }
  };
  auto load_extra = [&](uint32_t t, uint32_t h, uint32_t e0, uint32_t *kx) {
    const uint32_t n = h & 0xFFFFu;
    const uint32_t *kp = keys + (uint64_t)t * KP_TILE + (h >> 16);
#pragma unroll
    for (uint32_t v = 0; v < KC_X; v++) kx[v] = kp[min(e0 + v * 64 + lane, n - 1)];
  };
  auto pin = [&](uint32_t *v, uint32_t cnt) {
#pragma unroll
    for (uint32_t u = 0; u < cnt; u++) asm volatile("" : "+v"(v[u]));
  };

  uint32_t k0[KC_G], k0b[KC_G], k1[KC_G], k1b[KC_G], xa[KC_X], xb[KC_X];
  load_keys(0, 0, k0, k0b);
#pragma unroll
  for (uint32_t v = 0; v < KC_X; v++) xa[v] = xb[v] = 0;
  {
    const uint32_t h = hdr_at(0);
    if ((h & 0xFFFFu) > 128) load_extra(0, h, 128, xa);
  }
  lds_barrier(); // table and images initialised

  // The store phase of tile t-1 runs inside iteration t, around the exchanges of tile t: one barrier per
  // tile, and the LDS round trips of the two halves overlap within a wave.
  // Barrier at the top of iteration t: image (t-1) & 1 is complete (every wave has finished tile t-1), and
  // image t & 1 has been emptied again (its words were stored and reset during iteration t-1).
  // NEVER remove that barrier, not even in a timing experiment: without it a wave stores stale image words
  // with a later tile's base, beyond the end of the chain array.
  auto store_tile = [&](const uint32_t *ent, uint32_t *img, uint32_t sb) {
    // every wave stores 64 consecutive positions per instruction; only this partition's words are set
#pragma unroll
    for (int j = 0; j < KP_ROWS; j++) {
      const uint32_t rel = j * KP_THREADS + tid;
      if (ent[j] != KP_EMPTY) {
        chain[sb + rel] = ent[j];
        img[rel] = KP_EMPTY;
      }
    }
  };
#pragma unroll 1
  for (uint32_t T0 = 0; T0 < ntiles; T0 += 64) {
#pragma unroll 1
    for (uint32_t s = 0; s < 64 / KC_G; s++) {
      const uint32_t tb = T0 + s * KC_G;
      if (tb >= ntiles) break;
      pin(k0, KC_G); // requested one step ago
      pin(k0b, KC_G);
      load_keys(tb + KC_G, s * KC_G + KC_G, k1, k1b);
#pragma unroll
      for (uint32_t u = 0; u < KC_G; u++) {
        const uint32_t t = tb + u;
        if (t >= ntiles) break; // workgroup-uniform
        const uint32_t sb = t * KP_TILE;
        uint32_t *img = res + (t & 1) * KP_TILE, *imgp = res + ((t & 1) ^ 1) * KP_TILE;
        uint32_t *cur = (u & 1) ? xb : xa, *nxt = (u & 1) ? xa : xb; // extras of this tile / of the next one
        const uint32_t h = hdr_at(s * KC_G + u), n = h & 0xFFFFu;
        const uint32_t *kp = keys + sb + (h >> 16);
        if (n > 128) pin(cur, KC_X); // requested one tile ago; waited for BEFORE the next request goes out
        {
          const uint32_t hn = hdr_at(s * KC_G + u + 1);
          if (t + 1 < ntiles && (hn & 0xFFFFu) > 128) load_extra(t + 1, hn, 128, nxt);
        }
        uint32_t ent[KP_ROWS];
        if (t > 0) lds_barrier();
#pragma unroll
        for (int j = 0; j < KP_ROWS; j++) ent[j] = t > 0 ? imgp[j * KP_THREADS + tid] : KP_EMPTY;
        const uint32_t vbase = sb << cb; // table value = vbase + low key bits = position << cb | check bits
        {
          const bool act = lane < n;
          const uint32_t low = k0[u] & low_mask, val = vbase + low;
          uint32_t *p = act ? slice + ((k0[u] & ~KP_KEY_SET_ONLY) >> KP_KEY_IDX_SHIFT) : dummy;
          const uint32_t old = __hip_atomic_exchange(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          uint32_t *ip = act && !(k0[u] & KP_KEY_SET_ONLY) ? img + (low >> cb) : dummy;
          *ip = entry(val, old);
        }
        if (n > 64) {
          // Runs of neighbouring keys with the same table entry are folded: 64 lanes exchanging on ONE
          // address are performed one after the other.  Inside a run every key's predecessor is its left
          // neighbour; the run's first lane alone goes to the table, leaves the run's LAST value there and
          // takes the old one.
          auto batch = [&](uint32_t key, uint32_t e0) {
            const bool act = e0 + lane < n;
            const uint32_t low = key & low_mask, val = vbase + low;
            const uint32_t idx = act ? (key & ~KP_KEY_SET_ONLY) >> KP_KEY_IDX_SHIFT : 0x10000u | lane; // no key: equal to nobody
            const uint32_t pidx = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)idx, 0x138, 0xf, 0xf, false); // wave_shr:1
            const uint32_t pval = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)val, 0x138, 0xf, 0xf, false);
            const uint32_t nidx = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)idx, 0x130, 0xf, 0xf, false); // wave_shl:1
            const bool cont = idx == pidx;
            const unsigned long long tails = __ballot(idx != nidx); // lane 63 is always one
            const uint32_t tl = lane + (uint32_t)__builtin_ctzll(tails >> lane);
            const uint32_t tval = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tl << 2), (int)val);
            uint32_t *p = act && !cont ? slice + idx : dummy;
            const uint32_t old = __hip_atomic_exchange(p, tval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t *ip = act && !(key & KP_KEY_SET_ONLY) ? img + (low >> cb) : dummy;
            *ip = entry(val, cont ? pval : old);
          };
          batch(k0b[u], 64);
          uint32_t e = 128;
          while (e < n) {
#pragma unroll 1
            for (uint32_t v = 0; v < KC_X && e < n; v++, e += 64) {
              batch(cur[0], e);
#pragma unroll
              for (uint32_t w = 0; w + 1 < KC_X; w++) cur[w] = cur[w + 1];
            }
            if (e < n) { // more than 128 + 64 * KC_X keys (zero pages): on the spot
              load_extra(t, h, e, cur);
              pin(cur, KC_X);
            }
          }
        }
        store_tile(ent, imgp, sb - KP_TILE); // (nothing in iteration 0: every word is KP_EMPTY)
        (void)kp;
      }
#pragma unroll
      for (uint32_t u = 0; u < KC_G; u++) {
        k0[u] = k1[u];
        k0b[u] = k1b[u];
      }
    }
    hv = hvn;
    asm volatile("" : "+v"(hvnn)); // requested a block ago
    hvn = hvnn;
    hvnn = load_hv(T0 + 192);
  }
  lds_barrier(); // the image of the last tile is complete
  {
    uint32_t *img = res + ((ntiles - 1) & 1) * KP_TILE;
    uint32_t ent[KP_ROWS];
#pragma unroll
    for (int j = 0; j < KP_ROWS; j++) ent[j] = img[j * KP_THREADS + tid];
    store_tile(ent, img, (ntiles - 1) * KP_TILE);
  }
}

} // namespace m4g
