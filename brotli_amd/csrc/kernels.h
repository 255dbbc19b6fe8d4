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

#include "m4_parse.h"

namespace m4g {

// One unit of work: a buffer region to match-find.  Batch mode: one job per
// chunk (start = 0).  Streaming mode: one job, the handle's history buffer,
// start = length of the history before this block was appended.
struct Job {
  uint64_t src_off;   // byte offset of buffer position 0 in the src base
  uint64_t chain_off; // element offset of position 0 in the chain base
  uint64_t slot_off;  // first Rec of this job's output slot
  uint32_t start;     // first position to insert / parse (e.NextEmit at entry)
  uint32_t len;       // buffer length (positions [start, len-7) are inserted)
  uint32_t seed_prev; // Distance of the caller's last dst element, 0 = none
  uint32_t slot_cap;  // Recs available in the slot
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
//   * positions are processed in tiles of blockDim.x in increasing order;
//     inside a tile  old = table[h]  ->  barrier  ->  ds_max(table[h], i)  ->
//     barrier  ->  a position that does not read itself back has a later
//     duplicate in the same tile; those rare "losers" go to a short LDS list
//     that same-hash positions scan to find an in-tile predecessor.
// ---------------------------------------------------------------------------
constexpr int K1_THREADS = 1024;

__device__ __forceinline__ uint32_t hash_at(const uint8_t *p, uint64_t mask, int shift) {
  return (uint32_t)(((ld64(p) & mask) * kHashMul64) >> shift);
}

__global__ __launch_bounds__(K1_THREADS) void k1_chain_build(
    const uint8_t *__restrict__ src_base, uint32_t *__restrict__ chain_base,
    const Job *__restrict__ jobs, uint32_t *__restrict__ table_io, int njobs,
    int hash_len, int table_bits, int part_bits) {
  extern __shared__ uint32_t lds[];
  const uint32_t part_size = 1u << part_bits;
  uint32_t *table = lds;
  uint2 *dlist = reinterpret_cast<uint2 *>(lds + part_size);
  uint32_t *dcnt = reinterpret_cast<uint32_t *>(dlist + K1_THREADS);

  const int nparts = 1 << (table_bits - part_bits);
  // XCD-aware placement: blocks b and b+8 share an XCD, so give the nparts
  // partitions of one job the same b % 8.
  const uint32_t b = blockIdx.x;
  const uint32_t x = b & 7, r = b >> 3;
  const uint32_t q = r % nparts;
  const uint32_t job = (r / nparts) * 8 + x;
  if (job >= (uint32_t)njobs) return;

  const Job J = jobs[job];
  const uint8_t *src = src_base + J.src_off;
  uint32_t *chain = chain_base + J.chain_off;
  const uint32_t tid = threadIdx.x;

  for (uint32_t k = tid; k < part_size; k += K1_THREADS)
    table[k] = table_io ? table_io[(q << part_bits) + k] : 0u;
  if (tid == 0) *dcnt = 0;
  __syncthreads();

  const uint64_t mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
  const int shift = 64 - table_bits;
  const uint32_t idx_mask = part_size - 1;
  const uint32_t end = J.len >= 8 ? J.len - 7 : 0; // positions < end are inserted

  for (uint32_t base = J.start; base < end; base += K1_THREADS) {
    const uint32_t pos = base + tid;
    bool in = false;
    uint32_t idx = 0, old = 0;
    if (pos < end) {
      uint32_t h = hash_at(src + pos, mask, shift);
      in = (h >> part_bits) == q;
      idx = h & idx_mask;
    }
    if (in) old = table[idx];
    __syncthreads();
    if (in) atomicMax(&table[idx], pos);
    __syncthreads();
    const bool loser = in && table[idx] != pos;
    if (__syncthreads_or(loser)) {
      if (loser) {
        uint32_t k = atomicAdd(dcnt, 1u);
        dlist[k] = make_uint2(idx, pos);
      }
      __syncthreads();
      const uint32_t cnt = *dcnt;
      if (in) {
        for (uint32_t k = 0; k < cnt; k++) {
          uint2 e = dlist[k];
          if (e.x == idx && e.y < pos && e.y > old) old = e.y;
        }
      }
      __syncthreads();
      if (tid == 0) *dcnt = 0;
    }
    if (in) chain[pos] = old ? pos - old : 0u;
  }

  if (table_io) {
    __syncthreads();
    for (uint32_t k = tid; k < part_size; k += K1_THREADS)
      table_io[(q << part_bits) + k] = table[k];
  }
}

// ---------------------------------------------------------------------------
// K3 (serial form): one lane walks one job from start to end.
// ---------------------------------------------------------------------------
struct SlotSink {
  Rec *out;
  uint32_t n, cap;
  __device__ __forceinline__ void push(const Rec &r) {
    if (n < cap) {
      uint4 v = make_uint4(r.unmatched, r.length, r.distance, r.lit_start);
      *reinterpret_cast<uint4 *>(out + n) = v;
    }
    n++;
  }
};

__global__ __launch_bounds__(64) void k3_parse_serial(
    const uint8_t *__restrict__ src_base, const uint32_t *__restrict__ chain_base,
    const Job *__restrict__ jobs, Rec *__restrict__ slots,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ errors, int njobs,
    ParseParams P) {
  // one wave per job, lane 0 runs the state machine
  const uint32_t job = blockIdx.x;
  if (job >= (uint32_t)njobs || threadIdx.x != 0) return;
  const Job J = jobs[job];
  const uint8_t *src = src_base + J.src_off;
  const uint32_t *chain = chain_base + J.chain_off;
  SlotSink sink{slots + J.slot_off, 0, J.slot_cap};
  ParseState S;
  state_init(S, (int32_t)J.start, J.seed_prev);
  const int32_t len = (int32_t)J.len;
  parse_range(src, len, chain, P, S, len - 7, sink);
  parse_finish(len, P, S, sink);
  counts[job] = sink.n;
  if (S.error) atomicOr(errors, 1u);
}

// ---------------------------------------------------------------------------
// K4: exclusive prefix sum of counts (single workgroup) and compaction.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k4_scan_counts(
    const uint32_t *__restrict__ counts, uint64_t *__restrict__ out_off, int njobs) {
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
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

// grid = (njobs, K4_SPLIT): each workgroup copies a strided share of one slot.
constexpr int K4_SPLIT = 4;
__global__ __launch_bounds__(256) void k4_compact(
    const Rec *__restrict__ slots, const Job *__restrict__ jobs,
    const uint32_t *__restrict__ counts, const uint64_t *__restrict__ out_off,
    uint32_t *__restrict__ out /* 3 words per match */, uint64_t out_cap) {
  const uint32_t job = blockIdx.x;
  const Job J = jobs[job];
  const uint32_t n = min(counts[job], J.slot_cap);
  const uint64_t o = out_off[job];
  const Rec *s = slots + J.slot_off;
  for (uint32_t k = blockIdx.y * blockDim.x + threadIdx.x; k < n; k += blockDim.x * K4_SPLIT) {
    if (o + k >= out_cap) break;
    uint4 v = *reinterpret_cast<const uint4 *>(s + k);
    uint32_t *d = out + (o + k) * 3;
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
  }
}

// ---------------------------------------------------------------------------
// K5: histograms (encoder.go:44-105).
//
// The only sequential piece of that pass is the ring of the last four
// distances.  A distance is pushed exactly when it differs from the previous
// match's distance, so the ring seen by match j is the
// last four "flagged" matches before j.  A max-scan gives every match the
// index of the latest flagged match before it (stored in pf[]); four hops
// along pf[] rebuild the ring with no serial dependence between matches.
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

constexpr int K5_THREADS = 256;
constexpr int K5_WAVES = K5_THREADS / 64;

__global__ __launch_bounds__(K5_THREADS) void k5_histograms(
    const uint8_t *__restrict__ src_base, const Rec *__restrict__ slots,
    const Job *__restrict__ jobs, const uint32_t *__restrict__ counts,
    uint32_t *__restrict__ pf_base, uint32_t *__restrict__ histos /* 1024 u32 per job */) {
  __shared__ uint32_t lit[K5_WAVES][256];
  __shared__ uint32_t cmd[704];
  __shared__ uint32_t dst[64];
  __shared__ uint32_t wmax[K5_WAVES];
  __shared__ uint32_t carry_s;

  const uint32_t job = blockIdx.x;
  const Job J = jobs[job];
  const uint32_t n = min(counts[job], J.slot_cap);
  const Rec *rec = slots + J.slot_off;
  uint32_t *pf = pf_base + J.slot_off;
  const uint8_t *src = src_base + J.src_off + 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  for (int k = tid; k < K5_WAVES * 256; k += K5_THREADS) (&lit[0][0])[k] = 0;
  for (int k = tid; k < 704; k += K5_THREADS) cmd[k] = 0;
  if (tid < 64) dst[tid] = 0;
  if (tid == 0) carry_s = 0;
  __syncthreads();

  // pass 1: pf[j] = 1 + index of the latest flagged match before j (0: none)
  for (uint32_t base = 0; base < n; base += K5_THREADS) {
    const uint32_t j = base + tid;
    uint32_t flag = 0;
    if (j < n) {
      uint4 v = *reinterpret_cast<const uint4 *>(rec + j);
      if (v.y != 0) {
        uint32_t prevd = j ? rec[j - 1].distance : 0xFFFFFFFFu;
        flag = (v.z != prevd) ? j + 1 : 0;
      }
    }
    uint32_t incl = flag;
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl = max(incl, t);
    }
    if (lane == 63) wmax[wid] = incl;
    __syncthreads();
    uint32_t before = carry_s;
    for (int w = 0; w < wid; w++) before = max(before, wmax[w]);
    uint32_t excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0;
    excl = max(excl, before);
    if (j < n) pf[j] = excl;
    __syncthreads();
    if (tid == K5_THREADS - 1) carry_s = max(before, incl);
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();

  // pass 2: command + distance symbols, literal histogram
  uint32_t *mylit = lit[wid];
  for (uint32_t base = 0; base < n; base += K5_THREADS) {
    const uint32_t j = base + tid;
    uint32_t un = 0, ls = 0;
    if (j < n) {
      uint4 v = *reinterpret_cast<const uint4 *>(rec + j);
      un = v.x;
      ls = v.w;
      const uint32_t len = v.y, dist = v.z;
      uint32_t ic = insert_length_code(un);
      uint32_t cc = len ? copy_length_code(len) : 2u;
      bool use_last = j > 0 && dist == rec[j - 1].distance;
      uint32_t command = combine_length_codes(ic, cc, use_last);
      atomicAdd(&cmd[command], 1u);
      if (command >= 128 && len != 0) {
        // ring d[3], d[2], d[1], d[0] = last four pushed distances (-10: none)
        int64_t ring[4];
        uint32_t p = pf[j];
        for (int t = 0; t < 4; t++) {
          if (p) {
            ring[t] = (int64_t)rec[p - 1].distance;
            p = pf[p - 1];
          } else {
            ring[t] = -10;
          }
        }
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
      }
    }
    // literals: short runs per lane, long runs by the whole wave
    const bool longrun = un >= 64;
    if (!longrun)
      for (uint32_t k = 0; k < un; k++) atomicAdd(&mylit[src[ls + k]], 1u);
    unsigned long long lm = __ballot(longrun);
    while (lm) {
      int l = __builtin_ctzll(lm);
      lm &= lm - 1;
      uint32_t u = __shfl(un, l), s0 = __shfl(ls, l);
      for (uint32_t k = lane; k < u; k += 64) atomicAdd(&mylit[src[s0 + k]], 1u);
    }
  }
  __syncthreads();

  uint32_t *h = histos + (size_t)job * 1024;
  for (int k = tid; k < 256; k += K5_THREADS) {
    uint32_t s = 0;
    for (int w = 0; w < K5_WAVES; w++) s += lit[w][k];
    h[k] = s;
  }
  for (int k = tid; k < 704; k += K5_THREADS) h[256 + k] = cmd[k];
  if (tid < 64) h[960 + tid] = dst[tid];
}

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
