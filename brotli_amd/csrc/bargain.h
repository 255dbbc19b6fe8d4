// bargain.h -- matchfinder.Bargain1 / Bargain2 on the GPU.
//
// What the reference does (matchfinder/bargain1.go:38-208, bargain2.go): per
// block handed to FindMatches, a shortest-path search over "arrivals" (cheapest
// known way to reach each position, float32 bit costs), fed by one hash-table
// candidate per table and a repeat-distance probe at every position, then a
// backward walk that turns the path into Matches.
//
// How it is spread over the GPU:
//   * the hash tables are inserted unconditionally and in position order
//     (bargain1.go:141-146), so "the table entry before position i" is the chain
//     entry K1 already computes (k1_xchg.h, XH form: this finder's multiplier);
//     history slides (bargain1.go:66-80) only clear entries the distance test of
//     :166 rejects anyway;
//   * every block's search is independent of the other blocks' (fresh arrivals,
//     bargain1.go:55-64): ONE WAVE per block walks the positions in order --
//     the costs are sums of float32 roundings, there is no way to cut that chain
//     -- and its 64 lanes do the parts that are parallel: candidate checks and
//     forward / backward extension for 64 positions ahead of the walk, the
//     arrival range a match updates (bargain1.go:97-107), the probe of one
//     repeat distance at 64 positions, long extensions 512 bytes per step;
//   * arrivals live in an LDS ring around the walk (KB_WIN entries); what falls
//     out goes to a per-wave array in HBM, which the backward walk reads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace m4g {

constexpr int KB_WIN = 256;  // arrivals in LDS per wave (ring; positions [wlo, wlo + KB_WIN))
constexpr int KB_BACK = 64;  // of which this many behind the current batch of 64 (the batch and the next 64 are in registers)
constexpr int KB_FCAP = 32;  // bytes beyond the first 4 compared ahead of the walk
constexpr int KB_MAX_TABLES = 3;

struct BBlock {
  uint32_t job;   // chunk (stream) the block belongs to
  uint32_t start; // first position of the block in its chunk = len(history) at the call
  uint32_t len;   // > 0
  uint32_t pad;
};

struct KbTable {
  uint32_t mul_lo, mul_hi; // hash = ((u & mask) * mul) >> (64 - bits)
  uint32_t mask_lo, mask_hi;
  uint32_t bits;
  uint32_t mul2; // != 0: + uint64(e) * (mul2 << 32), e = the 4 bytes after the 8 (bargain3.go:276-278)
  uint64_t chain_base_off; // element offset of this table's chain array in the chain base
};

struct KbParams {
  uint32_t max_distance, skip, overlap, block_size;
  uint32_t tail, pad; // positions i > len(src) - tail are not hashed: 8 (bargain1.go:136), 12 (bargain3.go:153)
  KbTable t[KB_MAX_TABLES];
};

// ---- Go's math.Log2 (pure Go on amd64: log.go, log10.go), operation for operation.
// The literal costs are float32 roundings of it (bargain1.go:47-51).
__device__ inline double kb_go_log(double x) {
#pragma clang fp contract(off)
  const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10,
               L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
               L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
               L7 = 1.479819860511658591e-01;
  int ki;
  double f1 = frexp(x, &ki);
  if (f1 < 0.70710678118654752440) {
    f1 *= 2;
    ki--;
  }
  const double f = f1 - 1;
  const double k = (double)ki;
  const double s = f / (2 + f);
  const double s2 = s * s;
  const double s4 = s2 * s2;
  const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
  const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
  const double R = t1 + t2;
  const double hfsq = 0.5 * f * f;
  return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

__device__ inline float kb_byte_cost_of(uint32_t total, uint32_t count) {
#pragma clang fp contract(off)
  if (count == 0) return __builtin_inff(); // Log2(+Inf); a byte that does not occur is never priced
  const double x = (double)total / (double)count; // >= 1
  int e;
  const double frac = frexp(x, &e);
  double c;
  if (frac == 0.5) c = (double)(e - 1);
  else c = kb_go_log(frac) * (1 / 0.693147180559945309417232121458176568) + (double)e;
  if (c < 1) c = 1;
  return (float)c;
}

// byteCost of every block (bargain1.go:43-51): one workgroup per block.
__global__ __launch_bounds__(256) void kb_byte_cost(const uint8_t *__restrict__ src_base, const Job *__restrict__ jobs,
                                                    const BBlock *__restrict__ blocks, uint32_t nblocks,
                                                    float *__restrict__ out) {
  __shared__ uint32_t h[4][256]; // one histogram per wave
  const uint32_t b = blockIdx.x;
  if (b >= nblocks) return;
  const BBlock B = blocks[b];
  const uint8_t *src = src_base + jobs[B.job].src_off + B.start;
  const uint32_t tid = threadIdx.x, w = tid >> 6;
  for (int k = 0; k < 4; k++) h[k][tid] = 0;
  __syncthreads();
  // 16 bytes per thread and step once the address is 16-byte aligned
  const uint32_t head = min(B.len, (uint32_t)((16 - ((uintptr_t)src & 15)) & 15));
  if (tid < head) atomicAdd(&h[w][src[tid]], 1u);
  const uint32_t nvec = (B.len - head) / 16;
  const uint4 *v = reinterpret_cast<const uint4 *>(src + head);
  for (uint32_t k = tid; k < nvec; k += 256) {
    const uint4 q = v[k];
    const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      atomicAdd(&h[w][ws[j] & 255], 1u);
      atomicAdd(&h[w][(ws[j] >> 8) & 255], 1u);
      atomicAdd(&h[w][(ws[j] >> 16) & 255], 1u);
      atomicAdd(&h[w][ws[j] >> 24], 1u);
    }
  }
  for (uint32_t k = head + nvec * 16 + tid; k < B.len; k += 256) atomicAdd(&h[w][src[k]], 1u);
  __syncthreads();
  const uint32_t c = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
  out[(size_t)b * 256 + tid] = kb_byte_cost_of(B.len, c);
}

__device__ inline uint64_t kb_ld64(const uint8_t *p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ inline uint32_t kb_ld32(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
// 8 bytes at src + off where the buffer ends at lim (bytes past it read as 0)
__device__ inline uint64_t kb_ld64s(const uint8_t *src, uint64_t off, uint64_t lim) {
  if (off + 8 <= lim) return kb_ld64(src + off);
  uint64_t v = 0;
  for (int t = 0; t < 8; t++)
    if (off + t < lim) v |= (uint64_t)src[off + t] << (8 * t);
  return v;
}
__device__ inline uint32_t kb_bitlen(uint32_t x) { return x ? 32u - (uint32_t)__builtin_clz(x) : 0u; }

// Number of equal bytes of src[a..] and src[b..] (a < b), at most KB_FCAP and at most maxf.
__device__ inline uint32_t kb_fwd32(const uint8_t *src, uint64_t a, uint64_t b, uint32_t maxf, uint64_t lim) {
  uint32_t f = 0;
#pragma unroll
  for (int q = 0; q < KB_FCAP / 8; q++) {
    const uint64_t x = kb_ld64s(src, a + 8 * q, lim) ^ kb_ld64s(src, b + 8 * q, lim);
    if (x) {
      f += (uint32_t)__builtin_ctzll(x) >> 3;
      break;
    }
    f += 8;
  }
  return min(f, maxf);
}

// Everything the walk needs to know about the candidate c of position i (c < i, i + 8 <= he), from
// loads that are all issued together (one memory round trip): are the first 4 bytes equal
// (bargain1.go:167), how many of the next KB_FCAP bytes are (at most maxf), and how many of the 8 bytes
// below (at most maxb; extendMatch2, m4.go:326-339).  cv = the 8 bytes at i.
struct KbEval {
  bool eq4, f_long, b_long;
  uint32_t f, bl;
};
__device__ inline KbEval kb_eval(const uint8_t *src, uint64_t lim, uint32_t c, uint32_t i, uint64_t cv, uint32_t maxf,
                                 uint32_t maxb) {
  KbEval e;
  e.bl = 0, e.b_long = false;
  if ((uint64_t)i + 40 <= lim) {
    uint64_t a[5], o[5], pa = 0, pi = 0;
    const bool back = maxb > 0 && c >= 8;
#pragma unroll
    for (int q = 0; q < 5; q++) a[q] = kb_ld64(src + c + 8 * q);
#pragma unroll
    for (int q = 1; q < 5; q++) o[q] = kb_ld64(src + i + 8 * q);
    if (back) pa = kb_ld64(src + c - 8), pi = kb_ld64(src + i - 8);
    o[0] = cv;
    e.eq4 = (uint32_t)a[0] == (uint32_t)cv;
    const uint32_t x0 = (uint32_t)((a[0] ^ cv) >> 32), x4 = (uint32_t)(a[4] ^ o[4]);
    const uint64_t x1 = a[1] ^ o[1], x2 = a[2] ^ o[2], x3 = a[3] ^ o[3];
    uint32_t f;
    if (x0) f = (uint32_t)__builtin_ctz(x0) >> 3;
    else if (x1) f = 4 + ((uint32_t)__builtin_ctzll(x1) >> 3);
    else if (x2) f = 12 + ((uint32_t)__builtin_ctzll(x2) >> 3);
    else if (x3) f = 20 + ((uint32_t)__builtin_ctzll(x3) >> 3);
    else f = x4 ? 28 + ((uint32_t)__builtin_ctz(x4) >> 3) : 32u;
    e.f = min(f, maxf);
    if (back) {
      const uint64_t x = pa ^ pi;
      const uint32_t eq = x ? (uint32_t)__builtin_clzll(x) >> 3 : 8u;
      e.bl = min(eq, maxb);
      e.b_long = eq == 8 && maxb > 8;
    } else if (maxb > 0) {
      e.b_long = true; // the first 8 bytes of the stream: the walk compares byte by byte
    }
  } else { // the last bytes of the source buffer
    e.eq4 = kb_ld32(src + c) == (uint32_t)cv;
    e.f = kb_fwd32(src, (uint64_t)c + 4, (uint64_t)i + 4, maxf, lim);
    e.b_long = maxb > 0;
  }
  e.f_long = e.f == (uint32_t)KB_FCAP && maxf > (uint32_t)KB_FCAP;
  return e;
}

#ifdef KB_STATS
__device__ unsigned long long kb_stats[8]; // rep evals, add_match calls, far calls, wave_extend, wave_back, pd lookups out of window
#define KB_COUNT(k) (st_##k++)
#else
#define KB_COUNT(k) ((void)0)
#endif

#define KB_LIKELY(x) __builtin_expect(!!(x), 1)
#define KB_UNLIKELY(x) __builtin_expect(!!(x), 0)

template <int NT>
__global__ __launch_bounds__(64, 8) void kb_parse(
    const uint8_t *__restrict__ src_base, uint64_t src_total, const uint32_t *__restrict__ chain_base,
    const Job *__restrict__ jobs, const BBlock *__restrict__ blocks, uint32_t nblocks,
    const float *__restrict__ byte_cost, uint2 *__restrict__ ws_ld, float *__restrict__ ws_cost, uint32_t ws_stride,
    m4g_match *__restrict__ stage, uint32_t stage_cap, uint32_t *__restrict__ counts, KbParams P) {
  __shared__ uint32_t wl[KB_WIN], wd[KB_WIN];
  __shared__ float wc[KB_WIN];
  __shared__ float bc[256];
  constexpr uint32_t M = KB_WIN - 1;
  const uint32_t lane = threadIdx.x;
  uint2 *gLD = ws_ld + (size_t)blockIdx.x * ws_stride;
  float *gC = ws_cost + (size_t)blockIdx.x * ws_stride;

  // (the grid is what the GPU holds at once, bargain_host.h; blocks are equally long but for the last of a stream)
#ifdef M4G_KB_HANDOUT
  // Variant: blocks handed out through a counter (the word behind the counts, zeroed by the host on this stream
  // before the launch).  ONE lane fetches the index and it is broadcast through a scalar register, so the loop
  // test below is wave-uniform (DESIGN 3.6 has the ISA and what a per-lane index does to this loop).
  uint32_t *next_block = counts + nblocks;
#pragma unroll 1
  for (;;) {
#ifdef M4G_KB_HANDOUT_BRANCHY /* the form that never returns: DESIGN 3.6 */
    uint32_t bv = 0;
    if (lane == 0) bv = atomicAdd(next_block, 1u);
#else
    // every lane takes part in the fetch (lane 0 adds 1, the others 0): no `if (lane == 0)` here for the
    // compiler to merge with the one that ends the body
    const uint32_t bv = atomicAdd(next_block, lane == 0 ? 1u : 0u);
#endif
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv);
    if (b >= nblocks) break;
#else
#pragma unroll 1
  for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
#endif
    const BBlock B = blocks[b];
    const Job J = jobs[B.job];
    const uint8_t *src = src_base + J.src_off;
    const uint64_t lim = src_total - J.src_off;
    const uint32_t hs = B.start, n = B.len, he = hs + n;
    const uint32_t *chain[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) chain[t] = chain_base + P.t[t].chain_base_off + J.chain_off;

    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) bc[lane + 64 * k] = byte_cost[(size_t)b * 256 + lane + 64 * k];
#pragma unroll
    for (int k = 0; k < KB_WIN / 64; k++) wl[lane + 64 * k] = 0, wd[lane + 64 * k] = 0, wc[lane + 64 * k] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // position 0 of the stream: the table slot of its hash holds {0, its bytes}, every other
    // slot that nobody has written holds {0, 0} (zfast.go:8-11); K1 reports both as "no predecessor"
    const bool pos0_inserted = min(P.block_size, J.len) >= P.tail;
    const uint32_t src0 = J.len >= 4 ? kb_ld32(src) : 0u;
    uint32_t h0[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      h0[t] = 0;
      if (pos0_inserted) {
        const uint64_t u = kb_ld64(src);
        const uint64_t mul = (uint64_t)P.t[t].mul_hi << 32 | P.t[t].mul_lo, mask = (uint64_t)P.t[t].mask_hi << 32 | P.t[t].mask_lo;
        const uint64_t x = P.t[t].mul2 ? (uint64_t)(kb_ld32(src + 8) * P.t[t].mul2) << 32 : 0ull;
        h0[t] = (uint32_t)(((u & mask) * mul + x) >> (64 - P.t[t].bits));
      }
    }

#ifdef KB_STATS
    uint32_t st_rep = 0, st_add = 0, st_far = 0, st_ext = 0, st_back = 0, st_pdfar = 0, st_cand = 0, st_repm = 0;
#endif
    uint32_t wlo = 0; // ring = arrivals [wlo, wlo + KB_WIN); arrival k = after k bytes of the block
    uint32_t ghw = 0; // highest arrival index ever written beyond the ring
    int32_t nos = 0;  // nextOverlapSearch (chunk position)
    uint32_t rep_pd = 0, rep_f = 0;
    unsigned long long rep_mask = 0, rep_long = 0;
    // Arrivals [b0, b0 + 128) of the batch being walked live in registers (lane l: b0 + l and
    // b0 + 64 + l); their ring slots are stale until the batch ends.
    uint32_t b0 = 0;
    uint32_t L0 = 0, L1 = 0, D0 = 0, D1 = 0;
    float C0 = 0.f, C1 = 0.f;

    auto in_ring = [&](uint32_t k) { return k >= wlo && k - wlo < (uint32_t)KB_WIN; };
    // (the three below: never for k in [b0, b0 + 128))
    auto read_c = [&](uint32_t k) -> float {
      if (in_ring(k)) return wc[k & M];
      return (k < wlo || k <= ghw) ? gC[k] : 0.f;
    };
    auto read_d = [&](uint32_t k) -> uint32_t {
      if (in_ring(k)) return wd[k & M];
      return (k < wlo || k <= ghw) ? gLD[k].y : 0u;
    };
    auto write_a = [&](uint32_t k, uint32_t len, uint32_t dist, float cost) {
      if (in_ring(k)) {
        wl[k & M] = len, wd[k & M] = dist, wc[k & M] = cost;
      } else {
        gLD[k] = make_uint2(len, dist);
        gC[k] = cost;
      }
    };
    // wave-uniform reads of the register window: idx = k - b0 < 128
    auto rget = [&](uint32_t v0, uint32_t v1, uint32_t idx) -> uint32_t { // (both rows read: no branch)
      const uint32_t x0 = (uint32_t)__builtin_amdgcn_readlane((int)v0, (int)(idx & 63));
      const uint32_t x1 = (uint32_t)__builtin_amdgcn_readlane((int)v1, (int)(idx & 63));
      return idx < 64 ? x0 : x1;
    };
    auto uread_d = [&](uint32_t k) -> uint32_t { // k wave-uniform
      if (KB_LIKELY(k - b0 < 128u)) return rget(D0, D1, k - b0);
      KB_COUNT(pdfar);
      return (uint32_t)__builtin_amdgcn_readfirstlane((int)read_d(k));
    };
    auto uread_c = [&](uint32_t k) -> float {
      if (KB_LIKELY(k - b0 < 128u)) return __uint_as_float(rget(__float_as_uint(C0), __float_as_uint(C1), k - b0));
      return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(read_c(k))));
    };

    // addMatch (bargain1.go:86-108) in block coordinates; all arguments wave-uniform: the match starts at
    // arrival ks (= m.Start - historyLen), ends at arrival k_top, dcost = float32(bits.Len(distance)) or 0
    // for a repeat.  The reference's loop runs from m.End down and stops at the first arrival that is at least
    // as cheap: per 64 arrivals, one compare and one ballot find it.
    // start_cost = the cost of arrival ks at the time of the call (0 for ks = 0); top_first: look at arrival
    // k_top alone before the rest (the loop stops there when a match found at the previous position already
    // covers this one at a lower cost, which is the rule for the backward-extended form and common for the plain one).
    auto add_match = [&](uint32_t ks, float start_cost, int32_t k_top, uint32_t dist, uint32_t unmatched, float dcost,
                         bool top_first) {
      KB_COUNT(add);
      float cost = start_cost + 4.0f; // baseMatchCost, pathfinder.go:64
      cost = cost + (float)kb_bitlen(unmatched);
      cost = cost + dcost;
      const int32_t k_lo = (int32_t)ks + 3;
      const int32_t r_lo = (int32_t)b0, r_hi = (int32_t)b0 + 127;
      if (top_first && k_top >= k_lo && (uint32_t)(k_top - r_lo) < 128u) {
        const float ct = __uint_as_float(rget(__float_as_uint(C0), __float_as_uint(C1), (uint32_t)(k_top - r_lo)));
        if (ct > 0.f && ct <= cost) return;
      }
      if (KB_LIKELY(k_lo >= r_lo && k_top <= r_lo + 63)) { // the whole range is in the first register row
        const int32_t k0 = r_lo + (int32_t)lane;
        const bool in0 = k0 >= k_lo && k0 <= k_top;
        const unsigned long long sm0 = __builtin_amdgcn_ballot_w64(in0 && C0 > 0.f && C0 <= cost);
        const int32_t hi0 = sm0 ? 63 - (int32_t)__builtin_clzll(sm0) : -1;
        const bool w0 = in0 && (int32_t)lane > hi0;
        L0 = w0 ? (uint32_t)k0 - ks : L0, D0 = w0 ? dist : D0, C0 = w0 ? cost : C0;
        return;
      }
      if (KB_LIKELY(k_lo >= r_lo && k_top <= r_hi)) { // the whole range is in registers: no branch below
        const int32_t k0 = r_lo + (int32_t)lane, k1 = k0 + 64;
        const bool in1 = k1 >= k_lo && k1 <= k_top, in0 = k0 >= k_lo && k0 <= k_top;
        const unsigned long long sm1 = __builtin_amdgcn_ballot_w64(in1 && C1 > 0.f && C1 <= cost);
        const unsigned long long sm0 = __builtin_amdgcn_ballot_w64(in0 && C0 > 0.f && C0 <= cost);
        const int32_t hi1 = sm1 ? 63 - (int32_t)__builtin_clzll(sm1) : -1, hi0 = sm0 ? 63 - (int32_t)__builtin_clzll(sm0) : -1;
        const bool w1 = in1 && (int32_t)lane > hi1;
        const bool w0 = in0 && sm1 == 0 && (int32_t)lane > hi0;
        L1 = w1 ? (uint32_t)k1 - ks : L1, D1 = w1 ? dist : D1, C1 = w1 ? cost : C1;
        L0 = w0 ? (uint32_t)k0 - ks : L0, D0 = w0 ? dist : D0, C0 = w0 ? cost : C0;
        return;
      }
      bool stopped = false;
      // arrivals [lo, top] outside the register window, 64 per step
      auto far = [&](int32_t top_k, int32_t lo_k) {
        KB_COUNT(far);
#pragma unroll 1
        for (int32_t top = top_k; top >= lo_k; top -= 64) {
          const int32_t k = top - (int32_t)lane;
          const bool act = k >= lo_k;
          const float c = act ? read_c((uint32_t)k) : 0.f;
          const bool stop = act && c > 0.f && c <= cost;
          const unsigned long long sm = __builtin_amdgcn_ballot_w64(stop);
          const uint32_t first = sm ? (uint32_t)__builtin_ctzll(sm) : 64u;
          if (act && lane < first) write_a((uint32_t)k, (uint32_t)k - ks, dist, cost);
          if (sm) {
            stopped = true;
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      };
      if (k_top > r_hi) {
        far(k_top, max(k_lo, r_hi + 1));
        if ((uint32_t)k_top >= wlo + KB_WIN) ghw = max(ghw, (uint32_t)k_top);
      }
      if (!stopped && k_top >= r_lo + 64 && k_lo <= r_hi) {
        const int32_t k = r_lo + 64 + (int32_t)lane;
        const bool in = k >= k_lo && k <= k_top;
        const unsigned long long sm = __builtin_amdgcn_ballot_w64(in && C1 > 0.f && C1 <= cost);
        const int32_t hi = sm ? 63 - (int32_t)__builtin_clzll(sm) : -1;
        if (in && (int32_t)lane > hi) L1 = (uint32_t)k - ks, D1 = dist, C1 = cost;
        stopped = sm != 0;
      }
      if (!stopped && k_top >= r_lo && k_lo <= r_lo + 63) {
        const int32_t k = r_lo + (int32_t)lane;
        const bool in = k >= k_lo && k <= k_top;
        const unsigned long long sm = __builtin_amdgcn_ballot_w64(in && C0 > 0.f && C0 <= cost);
        const int32_t hi = sm ? 63 - (int32_t)__builtin_clzll(sm) : -1;
        if (in && (int32_t)lane > hi) L0 = (uint32_t)k - ks, D0 = dist, C0 = cost;
        stopped = sm != 0;
      }
      if (!stopped && k_lo < r_lo) far(min(k_top, r_lo - 1), k_lo);
    };

    // first position >= b0 where src[a0 + t] != src[b0 + t], at most he (extendMatch, m4.go:287-324)
    auto wave_extend = [&](uint64_t a0, uint64_t p0) -> uint32_t {
      KB_COUNT(ext);
#pragma unroll 1
      for (;;) {
        const uint64_t pb = p0 + 8ull * lane;
        const uint32_t nb = pb < he ? (uint32_t)min((uint64_t)8, he - pb) : 0u;
        uint64_t x = 0;
        if (nb) x = kb_ld64s(src, a0 + 8ull * lane, lim) ^ kb_ld64s(src, pb, lim);
        if (nb && nb < 8) x &= (1ull << (8 * nb)) - 1;
        const bool term = nb < 8 || x != 0;
        const uint32_t tp = nb == 0 ? he : (uint32_t)pb + (x ? (uint32_t)__builtin_ctzll(x) >> 3 : nb);
        const unsigned long long tm = __builtin_amdgcn_ballot_w64(term);
        if (tm) return (uint32_t)__builtin_amdgcn_readlane((int)tp, (int)__builtin_ctzll(tm));
        a0 += 512, p0 += 512;
      }
    };
    // equal bytes below (i, cand), at most maxb (extendMatch2's downward loop, m4.go:330-333)
    auto wave_back = [&](uint32_t i, uint32_t cand, uint32_t maxb) -> uint32_t {
      KB_COUNT(back);
      uint32_t base = 0;
#pragma unroll 1
      for (;;) {
        const uint32_t o = base + lane + 1;
        const bool act = o <= maxb;
        const bool ne = !act || src[i - o] != src[cand - o];
        const unsigned long long nm = __builtin_amdgcn_ballot_w64(ne);
        if (nm) return base + (uint32_t)__builtin_ctzll(nm);
        base += 64;
      }
    };

#pragma unroll 1
    for (b0 = 0; b0 < n; b0 += 64) {
      // ---- slide the ring: [b0 - KB_BACK, b0 - KB_BACK + KB_WIN)
      while (wlo + KB_BACK < b0) {
        const uint32_t k = wlo + lane, slot = k & M;
        gLD[k] = make_uint2(wl[slot], wd[slot]);
        gC[k] = wc[slot];
        const uint32_t k2 = k + KB_WIN;
        uint2 v = make_uint2(0, 0);
        float cst = 0.f;
        if (k2 <= ghw) v = gLD[k2], cst = gC[k2];
        wl[slot] = v.x, wd[slot] = v.y, wc[slot] = cst;
        wlo += 64;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }

      // ---- phase A: lane = position b0 + lane
      const uint32_t r_l = b0 + lane, i_l = hs + r_l;
      const bool inb = r_l < n;
      const float lit_l = inb ? bc[src[i_l]] : 0.f; // byteCost[src[i]], bargain1.go:127
      const bool valid = inb && i_l + P.tail <= he;   // bargain1.go:136, bargain3.go:153
      uint64_t cv = 0;
      if (valid) cv = kb_ld64(src + i_l);
      const uint32_t cv32 = (uint32_t)cv;
      uint32_t cdist[NT], cktop[NT], cb[NT]; // distance, arrival index of the match end, backward extension
      float cdc[NT];                         // float32(bits.Len(distance)), bargain1.go:94
      unsigned long long okm[NT], flm[NT], blm[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        bool ok = false, f_long = false, b_long = false;
        uint32_t c = 0, f = 0, bl = 0;
        if (valid) {
          const uint32_t ce = chain[t][i_l];
          const uint32_t d = ce & ~kChainMaybe;
          bool pre;
          if (d > 0) {
            c = i_l - d;
            pre = d < P.max_distance && (ce & kChainMaybe); // bargain1.go:166
          } else { // no predecessor: table entry {0, 0}, or {0, bytes of position 0} in position 0's slot
            const uint64_t mul = (uint64_t)P.t[t].mul_hi << 32 | P.t[t].mul_lo,
                           mask = (uint64_t)P.t[t].mask_hi << 32 | P.t[t].mask_lo;
            const uint64_t x = P.t[t].mul2 ? (uint64_t)(kb_ld32(src + i_l + 8) * P.t[t].mul2) << 32 : 0ull;
            const uint32_t h = (uint32_t)(((cv & mask) * mul + x) >> (64 - P.t[t].bits));
            pre = i_l > 0 && i_l < P.max_distance && (cv32 == 0 || (pos0_inserted && h == h0[t]));
          }
#if defined(KB_EXP) && KB_EXP >= 3
          pre = false;
#endif
          if (pre) {
            const KbEval e = kb_eval(src, lim, c, i_l, cv, he - (i_l + 4), min(r_l, c));
            ok = e.eq4; // bargain1.go:167
            f = e.f, f_long = e.f_long, bl = e.bl, b_long = e.b_long;
          }
        }
        cdist[t] = i_l - c, cktop[t] = r_l + 4 + f, cb[t] = bl;
        cdc[t] = (float)kb_bitlen(i_l - c);
        okm[t] = __builtin_amdgcn_ballot_w64(ok);
        flm[t] = __builtin_amdgcn_ballot_w64(f_long);
        blm[t] = __builtin_amdgcn_ballot_w64(b_long);
      }
      const unsigned long long vmask = __builtin_amdgcn_ballot_w64(valid);
      rep_pd = 0; // the repeat-probe cache is per batch

      // ---- phase B: the walk, wave-uniform
      const uint32_t steps = min(64u, n - b0);
#pragma unroll 1
      for (uint32_t t = 0; t < steps; t++) {
        const uint32_t r = b0 + t, i = hs + r;
        const uint32_t a_len = (uint32_t)__builtin_amdgcn_readlane((int)L0, (int)t);
        const uint32_t a_dist = (uint32_t)__builtin_amdgcn_readlane((int)D0, (int)t);
        const float a_cost = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(C0), (int)t));
        const float n_cost = __uint_as_float(rget(__float_as_uint(C0), __float_as_uint(C1), t + 1));
        const uint32_t unmatched = a_dist == 0 ? a_len : 0u;
        uint32_t pd = 0;
        if (unmatched != 0 && r > unmatched) { // arrival r - unmatched is below r: first register row, or older
          const uint32_t kb = r - unmatched;
          pd = KB_LIKELY(kb >= b0) ? (uint32_t)__builtin_amdgcn_readlane((int)D0, (int)(kb - b0)) : uread_d(kb);
        }
#if defined(KB_EXP) && KB_EXP >= 2
        pd = 0;
#endif
        const float c1 = a_cost + __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(lit_l), (int)t));
        bool nbu;
        {
          const bool relax = n_cost == 0.f || c1 < n_cost;
          const bool w0 = relax && lane == t + 1, w1 = relax && t == 63 && lane == 0;
          L0 = w0 ? unmatched + 1 : L0, D0 = w0 ? 0u : D0, C0 = w0 ? c1 : C0;
          L1 = w1 ? unmatched + 1 : L1, D1 = w1 ? 0u : D1, C1 = w1 ? c1 : C1;
          nbu = relax || rget(D0, D1, t + 1) == 0;
        }
        if (KB_UNLIKELY(!(vmask >> t & 1))) continue;

        // repeat match (bargain1.go:148-157)
        if (pd != 0 && pd != rget(D0, D1, t + 4)) {
          if (KB_UNLIKELY(pd != rep_pd)) { // the 4-byte probe at all 64 positions of the batch: one load
            KB_COUNT(rep);
            const bool eq = valid && i_l >= pd && kb_ld32(src + (i_l - pd)) == cv32;
            rep_mask = __builtin_amdgcn_ballot_w64(eq);
            rep_pd = pd;
            rep_long = 0;
            if (KB_UNLIKELY(rep_mask != 0)) { // (rare: 6 probes in 10000 hit) how far the matches run
              bool lng = false;
              uint32_t f = 0;
              if (eq) {
                const KbEval e = kb_eval(src, lim, i_l - pd, i_l, cv, he - (i_l + 4), 0u);
                f = e.f, lng = e.f_long;
              }
              rep_f = f;
              rep_long = __builtin_amdgcn_ballot_w64(lng);
            }
          }
          if (KB_UNLIKELY(rep_mask >> t & 1)) {
            KB_COUNT(repm);
            uint32_t end = i + 4 + (uint32_t)__builtin_amdgcn_readlane((int)rep_f, (int)t);
            if (KB_UNLIKELY(rep_long >> t & 1)) end = wave_extend((uint64_t)(i - pd) + 4 + KB_FCAP, (uint64_t)i + 4 + KB_FCAP);
            add_match(r, a_cost, (int32_t)(end - hs), pd, unmatched, 0.f, true);
          }
        }

        if (P.skip && (i & 1)) continue; // bargain1.go:159 (no slide moves the origin inside a batch chunk)

#if defined(KB_EXP) && KB_EXP >= 1
        if (false) {
#else
        if (unmatched > 0 || (int32_t)i >= nos || nbu) {
#endif
#pragma unroll
          for (int q = 0; q < NT; q++) {
            if (!(okm[q] >> t & 1)) continue;
            KB_COUNT(cand);
            const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)cdist[q], (int)t);
            int32_t k_top = __builtin_amdgcn_readlane((int)cktop[q], (int)t);
            uint32_t delta = (uint32_t)__builtin_amdgcn_readlane((int)cb[q], (int)t);
            const float dcost = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cdc[q]), (int)t));
            if (KB_UNLIKELY((flm[q] | blm[q]) >> t & 1)) { // extensions beyond what phase A compared
              const uint32_t c = i - dist;
              if (flm[q] >> t & 1)
                k_top = (int32_t)(wave_extend((uint64_t)c + 4 + KB_FCAP, (uint64_t)i + 4 + KB_FCAP) - hs);
              if (blm[q] >> t & 1) delta = wave_back(i, c, min(r, c));
            }
            // extended backwards: with and without the extra (bargain1.go:172-178)
            if (delta != 0)
              add_match(r - delta, r > delta ? uread_c(r - delta) : 0.f, k_top, dist, unmatched > delta ? unmatched - delta : 0u,
                        dcost, true);
            // (the extended form may have rewritten arrival r when delta >= 3: read its cost again)
            add_match(r, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(C0), (int)t)), k_top, dist,
                      unmatched, dcost, true);
            const uint32_t end = hs + (uint32_t)k_top;
            nos = max(nos, max((int32_t)i + 1, (int32_t)end - (int32_t)P.overlap));
          }
        }
      }

      // ---- the batch's arrivals go to the ring; the next 64 come in from it
      {
        const uint32_t s0 = (b0 + lane) & M;
        wl[s0] = L0, wd[s0] = D0, wc[s0] = C0;
        L0 = L1, D0 = D1, C0 = C1;
        const uint32_t s2 = (b0 + 128 + lane) & M;
        L1 = wl[s2], D1 = wd[s2], C1 = wc[s2];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    // (b0 is now the first multiple of 64 at or above n: registers hold [b0, b0 + 128), arrival n may be entry 0)
    {
      const uint32_t s0 = (b0 + lane) & M;
      wl[s0] = L0, wd[s0] = D0, wc[s0] = C0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    // ---- the ring goes to HBM, then the backward walk (bargain1.go:184-207)
    for (uint32_t k = wlo + lane; k <= n && k < wlo + KB_WIN; k += 64) {
      gLD[k] = make_uint2(wl[k & M], wd[k & M]);
      gC[k] = wc[k & M];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    m4g_match *st = stage + (size_t)b * stage_cap;
    uint32_t top = stage_cap; // records are written downwards: the list comes out in stream order
    bool have = false;
    uint32_t cu = 0, cl = 0, cd = 0; // the record being built (matches[len(matches)-1])
    uint32_t k = n;
#pragma unroll 1
    while (k >= 1) {
      // arrivals k, k-1, ..., k-63 in lanes 0..63
      uint2 v = make_uint2(1, 0);
      if (k >= lane + 1) v = gLD[k - lane];
      uint32_t o = 0;
#pragma unroll 1
      while (o < 64 && o < k) {
        const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)v.x, (int)o);
        const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)v.y, (int)o);
        if (dist > 0) {
          if (have && top > 0) { // (top > 0: stage_cap holds the longest list a block can have)
            top--;
            if (lane == 0) st[top].unmatched = cu, st[top].length = cl, st[top].distance = cd;
          }
          have = true, cu = 0, cl = len, cd = dist;
        } else {
          if (!have) have = true, cu = 0, cl = 0, cd = 0;
          cu = len;
        }
        o += max(len, 1u); // (every arrival 1..n has been written with length >= 1)
      }
      k -= min(o, k);
    }
    if (have && top > 0) {
      top--;
      if (lane == 0) st[top].unmatched = cu, st[top].length = cl, st[top].distance = cd;
    }
    if (lane == 0) counts[b] = stage_cap - top;
#ifdef KB_STATS
    if (lane == 0) {
      atomicAdd(&kb_stats[0], (unsigned long long)st_rep), atomicAdd(&kb_stats[1], (unsigned long long)st_add);
      atomicAdd(&kb_stats[2], (unsigned long long)st_far), atomicAdd(&kb_stats[3], (unsigned long long)st_ext);
      atomicAdd(&kb_stats[4], (unsigned long long)st_back), atomicAdd(&kb_stats[5], (unsigned long long)st_pdfar);
      atomicAdd(&kb_stats[6], (unsigned long long)st_cand), atomicAdd(&kb_stats[7], (unsigned long long)st_repm);
    }
#endif
  }
}

// Staged records (at the end of every block's stage slots) -> one list in block order.
__global__ __launch_bounds__(256) void kb_compact(const m4g_match *__restrict__ stage, uint32_t stage_cap,
                                                  const uint32_t *__restrict__ counts,
                                                  const uint64_t *__restrict__ block_off, uint32_t nblocks,
                                                  m4g_match *__restrict__ out) {
  const uint32_t b = blockIdx.x;
  if (b >= nblocks) return;
  const uint32_t c = counts[b];
  const uint32_t *s = reinterpret_cast<const uint32_t *>(stage + (size_t)b * stage_cap + (stage_cap - c));
  uint32_t *d = reinterpret_cast<uint32_t *>(out + block_off[b]);
  for (uint32_t k = threadIdx.x; k < c * 3; k += 256) d[k] = s[k];
}

// Literal runs of K5_HUGE bytes or more, per block Match list: the list k5b_long_literals counts (in the batch
// pipeline k4_compact builds it).  One thread per block.
__global__ __launch_bounds__(64) void kb_long_runs(const m4g_match *__restrict__ m, const uint64_t *__restrict__ off,
                                                   uint32_t njobs, LongRun *__restrict__ runs, uint32_t *__restrict__ nruns,
                                                   uint32_t cap) {
  const uint32_t j = blockIdx.x * 64 + threadIdx.x;
  if (j >= njobs) return;
  uint32_t pos = 0;
  for (uint64_t k = off[j]; k < off[j + 1]; k++) {
    const uint32_t u = m[k].unmatched, l = m[k].length;
    if (u >= K5_HUGE) {
      const uint32_t idx = atomicAdd(nruns, 1u);
      if (idx < cap) runs[idx] = LongRun{j, pos, u, 0};
    }
    pos += u + l;
  }
}

} // namespace m4g
