// k2_eval.h -- kernel K2: the chain walks of matchfinder.M4 (m4.go:156-181) for ALL positions of a batch at
// once, by hash bucket instead of by pointer chase.  Included by kernels.h after k1_split.h.
//
// A walk from position i visits the earlier positions with i's table index, nearest first.  The serial parse
// (one lane per sub-block) pays one dependent memory round trip per hop and reads ~2 cache lines for it; with
// ChainLength 64 on text that is 2.5 hops per input byte and 719 fetched bytes per input byte (round 2), all
// of it latency.  But the positions of one table index are known before the parse starts (K1 has hashed
// them all), and every position's bytes are compared with the SAME few bytes of its bucket neighbours.  So:
//
//   k1d_depth   second consumer of the split K1's bin-sorted keys (k1_split.h): the table entry of a hash is
//               a COUNTER here; a key's fetch-and-add is the number of earlier positions with its hash
//               (depth), and the final counters are the bucket sizes.  Same ordering argument as k1c.
//   k2_scan_sizes  exclusive prefix sum of a job's bucket sizes -> bucket bases.
//   k2_place    sorted[base[hash(i)] + depth(i)] = i's entry (SortEnt, m4_parse.h: position, depth, the 4 bytes
//               before and the 8 bytes at it): the job's positions grouped by bucket, ascending; rank[i] = that
//               index.  The walk of i is then sorted[r-1], sorted[r-2], ... with r = rank[i]: no pointer to chase,
//               and the parse (parse_range's sorted walk, the default for chain-walking batch plans) decides
//               most candidates from the entries alone.
//   k2_eval     one lane per sorted entry: 48 bytes around the position are fetched ONCE into LDS, and the
//               lane compares its own with those of the up to ChainLength + 1 entries before it in its
//               bucket (ds_read, no memory), keeping the candidates that can win for some state of the parse
//               (m4_k2.h: k2_pair, k2_add).  Output: K2Res per position, and the "may match" mark of the
//               position's chain entry rewritten to "has a result".
// The parse (k3a/k3b/k3c through parse_range) then replaces each walk by one 16-byte read.
#pragma once

namespace m4g {

static_assert(kSortPosBits == kK2DistBits, "positions and distances share their width");
constexpr uint32_t kSortDepthShift = kSortPosBits;

// One workgroup per (job, table partition), wave w owns 1/16 of the partition's counters and walks ITS bin of
// every tile (k1c_scatter's structure: no barrier inside the loop).  Requires keys WITHOUT run folding
// (k1p_partition's fold = 0): every inserted position has a key.
__global__ __launch_bounds__(KP_THREADS) void k1d_depth(
    const uint32_t *__restrict__ keys_base, const uint32_t *__restrict__ hdr_base,
    uint32_t *__restrict__ depth_base, uint32_t depth_stride, uint32_t *__restrict__ sizes_base,
    const Job *__restrict__ jobs, int njobs, uint32_t kt_max, int table_bits, int part_bits, int check_bits) {
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
  for (uint32_t k = tid; k < part_size; k += KP_THREADS) table[k] = 0u;
  const uint32_t cb = (uint32_t)check_bits;
  const uint32_t slice_bits = (uint32_t)part_bits - 4;
  uint32_t *slice = table + (wave << slice_bits);
  uint32_t *dummy = table + part_size + tid; // where the add of a lane without a key goes
  const uint32_t *hp = hdr_base + (uint64_t)job * kt_max * KP_BINS + q * 16 + wave; // + t * KP_BINS
  const uint32_t *keys = keys_base + J.chain_off;
  uint32_t *depth = depth_base + J.chain_off * (uint64_t)depth_stride;
  const uint32_t low_mask = (1u << KP_KEY_IDX_SHIFT) - 1;
  lds_barrier(); // counters cleared (cooperatively; from here on a wave only touches its own slice)

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
      {
        const bool act = lane < n;
        uint32_t *p = act ? slice + ((k[u] & ~KP_KEY_SET_ONLY) >> KP_KEY_IDX_SHIFT) : dummy;
        // lanes on one counter are served in ascending lane order = position order inside the bin
        const uint32_t old = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (act) depth[(uint64_t)(sb + ((k[u] & low_mask) >> cb)) * depth_stride] = old;
      }
      if (n > 64) { // a crowded bin: the rest, 64 keys at a time
        const uint32_t *kp = keys + sb + (h[u] >> 16);
#pragma unroll 1
        for (uint32_t e0 = 64; e0 < n; e0 += 64) {
          const bool act = e0 + lane < n;
          const uint32_t kk = kp[min(e0 + lane, n - 1)];
          uint32_t *p = act ? slice + ((kk & ~KP_KEY_SET_ONLY) >> KP_KEY_IDX_SHIFT) : dummy;
          const uint32_t old = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (act) depth[(uint64_t)(sb + ((kk & low_mask) >> cb)) * depth_stride] = old;
        }
      }
    }
  }
  lds_barrier(); // every wave's counters are final
  uint32_t *sizes = sizes_base + ((uint64_t)job << table_bits) + ((uint64_t)q << part_bits);
  for (uint32_t k = tid; k < part_size; k += KP_THREADS) sizes[k] = table[k];
}

// Exclusive prefix sum over the 2^table_bits bucket sizes of a job, in place.  One workgroup per job.
__global__ __launch_bounds__(1024) void k2_scan_sizes(uint32_t *__restrict__ sizes_base, int njobs, int table_bits) {
  __shared__ uint32_t part[1024];
  const uint32_t job = blockIdx.x, tid = threadIdx.x;
  if (job >= (uint32_t)njobs) return;
  uint32_t *s = sizes_base + ((uint64_t)job << table_bits);
  const uint32_t T = 1u << table_bits;
  const uint32_t per = (T + 1023) / 1024;
  const uint32_t k0 = tid * per, k1 = min(k0 + per, T);
  uint32_t sum = 0;
  for (uint32_t k = k0; k < k1; k++) sum += s[k];
  part[tid] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) { // inclusive scan of the 1024 partial sums
    const uint32_t v = tid >= d ? part[tid - d] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;
  for (uint32_t k = k0; k < k1; k++) {
    const uint32_t v = s[k];
    s[k] = run;
    run += v;
  }
}

constexpr int K2P_THREADS = 256, K2P_PER = 16; // positions per workgroup of k2_place: 4096
// XCD-aware placement as in K1 / K3: the workgroups of one job on one XCD (they scatter words into the same
// 4 MiB of `sorted`, which that XCD's L2 merges into whole lines).
__device__ __forceinline__ bool k2_job_of_block(uint32_t b, uint32_t wpj, int njobs, uint32_t &job, uint32_t &part) {
  const uint32_t x = b & 7, r = b >> 3;
  part = r % wpj;
  job = (r / wpj) * 8 + x;
  return job < (uint32_t)njobs;
}

// sorted[base[hash(p)] + depth(p)] = the entry of p; rank[p] (in place of depth[p]) = that index | depth.
__global__ __launch_bounds__(K2P_THREADS) void k2_place(
    const uint8_t *__restrict__ src_base, uint32_t *__restrict__ rank_base, const uint32_t *__restrict__ base_base,
    SortEnt *__restrict__ sorted_base, const Job *__restrict__ jobs, int njobs, uint32_t wpj, int hash_len,
    int table_bits) {
  uint32_t job, part;
  if (!k2_job_of_block(blockIdx.x, wpj, njobs, job, part)) return;
  const Job J = jobs[job];
  const uint32_t end = J.len >= 8 ? J.len - 7 : 0; // positions < end are inserted (m4.go:112)
  const uint64_t mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
  const uint8_t *src = src_base + J.src_off;
  uint32_t *rank = rank_base + J.chain_off;
  const uint32_t *base = base_base + ((uint64_t)job << table_bits);
  SortEnt *sorted = sorted_base + J.chain_off;
#pragma unroll 4
  for (int it = 0; it < K2P_PER; it++) {
    const uint32_t p = part * (K2P_THREADS * K2P_PER) + it * K2P_THREADS + threadIdx.x;
    if (p >= end) continue;
    uint64_t v;
    __builtin_memcpy(&v, src + p, 8); // p + 8 <= len
    uint32_t before;
    if (J.src_off + p >= 4) {
      __builtin_memcpy(&before, src + p - 4, 4); // (p < 4: partly the previous chunk's bytes, which are never used)
    } else { // the first bytes of the whole source buffer: bytes [0, p) of the job in the top lanes
      uint32_t v0;
      __builtin_memcpy(&v0, src, 4);
      before = p ? v0 << (8 * (4 - p)) : 0u;
    }
    const uint32_t h = (uint32_t)(((v & mask) * kHashMul64) >> (64 - table_bits));
    const uint32_t d = rank[p]; // depth, left by k1d_depth
    const uint32_t idx = base[h] + d;
    const uint32_t dc = min(d, kSortDepthCap) << kSortDepthShift;
    // (idx < end by construction: every inserted position has a key, the bases sum the counts; the test keeps
    // a violated assumption from becoming a store outside the array)
    if (idx < end) {
      SortEnt e;
      e.word = p | dc;
      e.before4 = before;
      e.at8 = v;
      sorted[idx] = e;
    }
    rank[p] = (idx < end ? idx : 0u) | dc;
  }
}

constexpr int K2E_THREADS = 256;
constexpr int K2E_HALO = 80; // >= the longest walk (ChainLength + 1 <= 65 ... see the host's check), multiple of 16
constexpr int K2E_ROWS = K2E_THREADS + K2E_HALO;
constexpr int K2E_DW = (K2_BW + K2_FW) / 4; // 12 dwords per window
constexpr int K2_MAX_CHAIN = K2E_HALO - 1;  // ChainLength K2 can serve


constexpr int K2E_QUEUE = 4; // candidates a lane holds back before the whole wave runs k2_add

// The inner loop is bound by instruction issue (every wave runs as many rounds as its deepest bucket has
// entries, 65 at most), so a round does the least it can: one LDS word for the 4-byte test, the candidate's
// window as three 16-byte LDS reads, k2_pair, and the champion test (k2_filter).  The 1.7 candidates per
// position that pass it wait in a small per-lane queue; the full update of the kept list (k2_add, ~150
// instructions) runs for the whole wave only when some lane's queue is full, and once at the end.
__global__ __launch_bounds__(K2E_THREADS) void k2_eval(
    const uint8_t *__restrict__ src_base, uint64_t src_total, const SortEnt *__restrict__ sorted_base,
    uint32_t *__restrict__ chain_base, K2Res *__restrict__ res_base, const Job *__restrict__ jobs, int njobs,
    uint32_t wpj, ParseParams P) {
  __shared__ uint4 win[K2E_ROWS][3];       // [entry]: the 48 bytes around its position
  __shared__ uint32_t first4[K2E_ROWS];    // [entry]: the 4 bytes at its position (consecutive lanes, no conflicts)
  __shared__ uint32_t meta[K2E_ROWS];
  uint32_t job, part;
  if (!k2_job_of_block(blockIdx.x, wpj, njobs, job, part)) return;
  const Job J = jobs[job];
  const int32_t end = J.len >= 8 ? (int32_t)J.len - 7 : 0;
  const int32_t s0 = (int32_t)part * K2E_THREADS;
  if (s0 >= end) return;
  const SortEnt *sorted = sorted_base + J.chain_off;
  const int tid = threadIdx.x;
  // ---- the windows of this workgroup's entries and of the K2E_HALO entries before them ----
  for (int e = tid; e < K2E_ROWS; e += K2E_THREADS) {
    const int32_t s = s0 - K2E_HALO + e;
    uint32_t word = 0;
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0;
    if (s >= 0 && s < end) {
      word = sorted[s].word;
      const int64_t g = (int64_t)J.src_off + (int64_t)(word & kSortPosMask) - K2_BW;
      v0 = load_group(src_base, g, (int64_t)src_total);
      v1 = load_group(src_base, g + 16, (int64_t)src_total);
      v2 = load_group(src_base, g + 32, (int64_t)src_total);
    }
    meta[e] = word;
    first4[e] = v1.x;
    win[e][0] = v0, win[e][1] = v1, win[e][2] = v2;
  }
  __syncthreads();
  const int me = K2E_HALO + tid;
  const uint32_t myword = meta[me];
  const int32_t pos = (int32_t)(myword & kSortPosMask);
  const bool live = s0 + tid < end && pos < end; // (pos < end always; see k2_place)
  uint32_t *cw_ptr = chain_base + J.chain_off + pos;
  uint32_t cw = 0;
  if (live) cw = *cw_ptr;
  uint32_t a[K2E_DW];
  {
    const uint4 a0 = win[me][0], a1 = win[me][1], a2 = win[me][2];
    a[0] = a0.x, a[1] = a0.y, a[2] = a0.z, a[3] = a0.w, a[4] = a1.x, a[5] = a1.y, a[6] = a1.z, a[7] = a1.w;
    a[8] = a2.x, a[9] = a2.y, a[10] = a2.z, a[11] = a2.w;
  }
  const int32_t avail_f = min(K2_FW, (int32_t)J.len - pos);
  // the walk: the table entry, then up to ChainLength hops (m4.go:159, 166)
  int32_t nwalk = live ? min((int32_t)(myword >> kSortDepthShift), P.chain_length + 1) : 0;
  uint32_t kept[K2_ENTRIES], q[K2E_QUEUE];
#pragma unroll
  for (int k = 0; k < K2_ENTRIES; k++) kept[k] = 0;
#pragma unroll
  for (int k = 0; k < K2E_QUEUE; k++) q[k] = 0;
  int n = 0, qn = 0;
  bool fb = false;
  int32_t ch0 = INT32_MIN, ch1 = INT32_MIN;
  auto drain = [&]() {
#pragma unroll
    for (int t = 0; t < K2E_QUEUE; t++) {
      if (!__any(t < qn)) break;
      if (t < qn && !fb && !k2_add(kept, n, q[t], P.min_length, P.distance_bit_cost)) {
        fb = true;
        nwalk = 0;
      }
    }
    qn = 0;
  };
#pragma unroll 1
  for (int32_t k = 1; __any(k <= nwalk); k++) {
    bool on = k <= nwalk;
    const int ce = on ? me - k : me;
    const int32_t cpos = (int32_t)(meta[ce] & kSortPosMask);
    const int32_t dist = pos - cpos;
    // position 0 reads as "no entry" (m4.go:152, 172), and nothing beyond MaxDistance is visited: the walk ends
    if (on && (cpos <= 0 || dist > P.max_distance)) {
      on = false;
      nwalk = 0;
    }
    const bool hit = on && first4[ce] == a[4];
    if (__any(hit)) {
      const uint4 b0 = win[ce][0], b1 = win[ce][1], b2 = win[ce][2];
      const uint32_t bw[K2E_DW] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
      uint32_t code = 0;
      bool pass = hit && k2_pair(a, bw, avail_f, min(K2_BW, cpos), code);
      const uint32_t e = (uint32_t)dist | code;
      pass = pass && k2_filter(ch0, ch1, e, P.min_length, P.distance_bit_cost);
      if (pass) {
#pragma unroll
        for (int t = 0; t < K2E_QUEUE; t++)
          if (t == qn) q[t] = e;
        qn++;
      }
      if (__any(qn == K2E_QUEUE)) drain();
    }
  }
  drain();
  if (!live) return;
  const bool mark = fb || n > 0;
  if (mark) {
    K2Res r;
#pragma unroll
    for (int k = 0; k < K2_ENTRIES; k++) r.e[k] = fb ? (k == 0 ? kK2Fallback : 0u) : kept[k];
    res_base[J.chain_off + pos] = r;
  }
  const uint32_t nw = (cw & kChainDelta) | (mark ? kChainMaybe : 0u);
  if (nw != cw) *cw_ptr = nw;
}

} // namespace m4g
