// k6_flate.h -- the reference's flate encoder (flate/huffman_bit_writer.go) on
// the GPU, for the chunks of a batch.  Included by kernels.h after k6_encode.h.
//
// huffmanBitWriter.writeBlock (huffman_bit_writer.go:361-419) makes two passes
// over a block's Match list, like brotli.Encoder.Encode: makeStatistics
// (lines 425-468) counts literal / length and offset symbols, writeTokens
// (lines 472-504) writes them with the codes built in between.  Both passes are
// data parallel once every match knows its input position (a prefix sum of
// Unmatched + Length) and its output bit position (a prefix sum of bit counts):
//   k5f_flate_stats   one workgroup per chunk: the 286 + 30 symbol counts
//                     (without the end-of-block symbol, which the host adds) and
//                     the number of matches outside the token limits of
//                     flate/token.go:59-71 (Length 3..258, Distance 1..32768),
//                     where the reference panics;
//   (host)            code lengths and codes per chunk, choice of stored / fixed /
//                     dynamic block, header bits, absolute bit positions
//                     (flate_host.h);
//   k6f_flate_bodies  one workgroup per chunk: every thread writes its match's
//                     literals, length code + extra bits, offset code + extra
//                     bits at their final bit positions of the one deflate
//                     stream (atomic OR into the zeroed stream, as K6); a stored
//                     block's bytes are copied to their byte-aligned place.
// Literal runs of 64 bytes or more are counted / written by the whole wave.
#pragma once
#include <type_traits>

namespace m4g {

constexpr int KF_THREADS = 512;
constexpr int KF_WAVES = KF_THREADS / 64;
constexpr int KF_LIT = 286, KF_OFF = 30; // maxNumLit, offsetCodeCount (huffman_bit_writer.go:13,25)
constexpr int KF_TAB = 320;              // words per chunk: literal/length[286] | offset[30] | bad matches | pad
constexpr int KF_BAD = KF_LIT + KF_OFF;  // slot of the count of matches outside the token limits

// RFC 1951 3.2.5 (= lengthBase / lengthExtraBits / offsetBase / offsetExtraBits of
// huffman_bit_writer.go:31-60 and the two lookup tables of token.go:11-57, which are
// "the last code whose base does not exceed the value")
struct KFTables {
  uint8_t len_extra[29], off_extra[30];
  uint16_t len_base[29], off_base[30]; // of length - 3 and distance - 1
  uint8_t length_code[256], offset_code[256];
};
constexpr KFTables kf_tables() {
  KFTables t{};
  uint32_t b = 0;
  for (int c = 0; c < 28; c++) { // 8 codes without extra bits, then 4 codes per extra-bit count
    const int e = c < 8 ? 0 : (c - 4) / 4;
    t.len_extra[c] = (uint8_t)e;
    t.len_base[c] = (uint16_t)b;
    b += 1u << e;
  }
  t.len_extra[28] = 0; // length 258 has a code of its own
  t.len_base[28] = 255;
  b = 0;
  for (int c = 0; c < 30; c++) { // 4 codes without extra bits, then 2 codes per extra-bit count
    const int e = c < 4 ? 0 : (c - 2) / 2;
    t.off_extra[c] = (uint8_t)e;
    t.off_base[c] = (uint16_t)b;
    b += 1u << e;
  }
  for (int v = 0; v < 256; v++) {
    int c = 0;
    while (c + 1 < 28 && t.len_base[c + 1] <= v) c++;
    t.length_code[v] = (uint8_t)(v == 255 ? 28 : c);
    int o = 0;
    while (o + 1 < 16 && t.off_base[o + 1] <= v) o++;
    t.offset_code[v] = (uint8_t)o;
  }
  return t;
}
__constant__ const KFTables kKF = kf_tables();

// flate/token.go:59-80 (callers check the limits first)
__device__ __forceinline__ uint32_t kf_length_code(uint32_t len) { return kKF.length_code[len - 3]; }
__device__ __forceinline__ uint32_t kf_offset_code(uint32_t dist) {
  const uint32_t off = dist - 1;
  if (off < 256) return kKF.offset_code[off];
  if ((off >> 7) < 256) return kKF.offset_code[off >> 7] + 14u;
  return kKF.offset_code[off >> 14] + 28u;
}
__device__ __forceinline__ bool kf_token_ok(uint32_t len, uint32_t dist) {
  return len >= 3 && len <= 258 && dist >= 1 && dist <= 32768;
}

// makeStatistics, huffman_bit_writer.go:425-446
__global__ __launch_bounds__(KF_THREADS) void k5f_flate_stats(
    const uint8_t *__restrict__ src_base, const uint32_t *__restrict__ matches /* 3 words each */,
    const Job *__restrict__ jobs, const uint64_t *__restrict__ out_off, uint32_t *__restrict__ stats /* KF_TAB per job */) {
  __shared__ uint32_t hist[KF_WAVES][KF_TAB]; // per wave: hot literals would otherwise serialise the whole workgroup
  __shared__ uint32_t wadv[2][KF_WAVES];
  __shared__ uint32_t cpos[2];
  const uint32_t job = blockIdx.x;
  const Job J = jobs[job];
  const uint64_t o0 = out_off[job];
  const uint32_t n = (uint32_t)(out_off[job + 1] - o0);
  const uint32_t *rec = matches + o0 * 3;
  const uint8_t *src = src_base + J.src_off;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int k = tid; k < KF_WAVES * KF_TAB; k += KF_THREADS) (&hist[0][0])[k] = 0;
  if (tid == 0) cpos[0] = J.start;
  __syncthreads();
  uint32_t *h = hist[wid];

  uint32_t par = 0;
  for (uint32_t base = 0; base < n; base += KF_THREADS, par ^= 1) {
    const uint32_t j = base + tid;
    uint32_t un = 0, len = 0, dist = 0;
    if (j < n) {
      un = rec[3 * j];
      len = rec[3 * j + 1];
      dist = rec[3 * j + 2];
    }
    const uint32_t adv = un + len;
    uint32_t incl = adv;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wadv[par][wid] = incl;
    __syncthreads();
    uint32_t before = cpos[par];
    for (int w = 0; w < wid; w++) before += wadv[par][w];
    const uint32_t ls = before + incl - adv;
    if (tid == KF_THREADS - 1) cpos[par ^ 1] = before + incl;

    const bool longrun = un >= 64;
    if (!longrun)
      for (uint32_t k = 0; k < un; k++) atomicAdd(&h[src[ls + k]], 1u);
    for (unsigned long long lm = __ballot(longrun); lm; lm &= lm - 1) {
      const int l = __builtin_ctzll(lm);
      const uint32_t u = __shfl(un, l), s0 = __shfl(ls, l);
      for (uint32_t k = lane; k < u; k += 64) atomicAdd(&h[src[s0 + k]], 1u);
    }
    if (j < n && len != 0) {
      if (kf_token_ok(len, dist)) {
        atomicAdd(&h[257 + kf_length_code(len)], 1u);
        atomicAdd(&h[KF_LIT + kf_offset_code(dist)], 1u);
      } else {
        atomicAdd(&h[KF_BAD], 1u);
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < KF_TAB; k += KF_THREADS) {
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < KF_WAVES; w++) s += hist[w][k];
    stats[(size_t)job * KF_TAB + k] = s;
  }
}

// writeTokens, huffman_bit_writer.go:472-504 (mode != 0), or the bytes of a stored
// block (writeBytes, line 405; mode == 0).  tables: KF_TAB words per job,
// word = code bits (already reversed for LSB-first writing) | length << 16.
__global__ __launch_bounds__(KF_THREADS) void k6f_flate_bodies(
    const uint8_t *__restrict__ src_base, const uint32_t *__restrict__ matches, const Job *__restrict__ jobs,
    const uint64_t *__restrict__ out_off, const uint32_t *__restrict__ tables, const uint64_t *__restrict__ body_pos,
    const uint32_t *__restrict__ mode, uint32_t *__restrict__ words, uint64_t *__restrict__ body_bits) {
  __shared__ uint32_t tab[KF_TAB];
  __shared__ uint32_t win[K6_WIN_WORDS + 2]; // the bits of one iteration, assembled here (k6_encode.h)
  __shared__ uint32_t wadv[2][KF_WAVES], wbits[2][KF_WAVES];
  __shared__ uint32_t cpos[2];
  __shared__ uint64_t cbit[2];
  const uint32_t job = blockIdx.x;
  const Job J = jobs[job];
  const uint8_t *src = src_base + J.src_off;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint64_t pos0 = body_pos[job];
  if (mode[job] == 0) { // stored: bytes [start, len) at the byte-aligned position pos0
    for (uint32_t k = J.start + 4 * (uint32_t)tid; k < J.len; k += 4 * KF_THREADS) {
      uint32_t v = 0;
      const uint32_t m = min(4u, J.len - k);
      for (uint32_t q = 0; q < m; q++) v |= (uint32_t)src[k + q] << (8 * q);
      k6_or_at(words, pos0 + 8ull * (k - J.start), v);
    }
    if (tid == 0) body_bits[job] = 8ull * (J.len - J.start);
    return;
  }
  const uint64_t o0 = out_off[job];
  const uint32_t n = (uint32_t)(out_off[job + 1] - o0);
  const uint32_t *rec = matches + o0 * 3;
  for (int k = tid; k < KF_TAB; k += KF_THREADS) tab[k] = tables[(size_t)job * KF_TAB + k];
  for (int k = tid; k < K6_WIN_WORDS + 2; k += KF_THREADS) win[k] = 0;
  if (tid == 0) {
    cpos[0] = J.start;
    cbit[0] = pos0;
  }
  __syncthreads();

  uint32_t par = 0;
  for (uint32_t base = 0; base < n; base += KF_THREADS, par ^= 1) {
    const uint32_t j = base + tid;
    uint32_t un = 0, len = 0, dist = 0;
    if (j < n) {
      un = rec[3 * j];
      len = rec[3 * j + 1];
      dist = rec[3 * j + 2];
    }
    const uint32_t adv = un + len;
    uint32_t incl = adv;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wadv[par][wid] = incl;
    __syncthreads();
    uint32_t before = cpos[par];
    for (int w = 0; w < wid; w++) before += wadv[par][w];
    const uint32_t ls = before + incl - adv;
    if (tid == KF_THREADS - 1) cpos[par ^ 1] = before + incl;

    // ---- bit count of this match ----
    const bool has = j < n && len != 0 && kf_token_ok(len, dist); // (the host refuses a batch with a bad match)
    const uint32_t lc = has ? kf_length_code(len) : 0u, oc = has ? kf_offset_code(dist) : 0u;
    const uint32_t tl = tab[257 + lc], to = tab[KF_LIT + oc];
    const uint32_t le = kKF.len_extra[lc], oe = kKF.off_extra[oc];
    const uint32_t tail_bits = has ? (tl >> 16) + le + (to >> 16) + oe : 0u;
    uint32_t lit_bits = 0;
    const bool longrun = un >= 64;
    if (!longrun)
      for (uint32_t k = 0; k < un; k++) lit_bits += tab[src[ls + k]] >> 16;
    for (unsigned long long lm = __ballot(longrun); lm; lm &= lm - 1) {
      const int l = __builtin_ctzll(lm);
      const uint32_t u = __shfl(un, l), s0 = __shfl(ls, l);
      uint32_t s = 0;
      for (uint32_t k = lane; k < u; k += 64) s += tab[src[s0 + k]] >> 16;
      for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
      if (lane == l) lit_bits = s;
    }
    const uint32_t total = lit_bits + tail_bits;
    uint32_t binc = total;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(binc, o);
      if (lane >= o) binc += t;
    }
    if (lane == 63) wbits[par][wid] = binc;
    __syncthreads();
    const uint64_t s0 = cbit[par];
    uint64_t bpos = s0, s1 = s0;
    for (int w = 0; w < KF_WAVES; w++) {
      if (w < wid) bpos += wbits[par][w];
      s1 += wbits[par][w];
    }
    if (tid == KF_THREADS - 1) cbit[par ^ 1] = bpos + binc;
    bpos += binc - total; // where this match's bits begin
    const bool windowed = k6_span_fits(s0, s1); // workgroup-uniform
    // (two instantiations, so that the window's atomics are LDS instructions and the stream's are global ones)
    auto emit = [&](auto to_window) {
    const K6Sink sink = decltype(to_window)::value ? K6Sink{win, s0 & ~31ull} : K6Sink{words, 0};

    // ---- write: literals, length code + extra, offset code + extra ----
    if (j < n) {
      K6Out o;
      if (!longrun) {
        o.start(sink, bpos);
        for (uint32_t k = 0; k < un; k++) {
          const uint32_t t = tab[src[ls + k]];
          o.put(t >> 16, t & 0xFFFFu);
        }
      } else {
        o.start(sink, bpos + lit_bits); // the wave writes the run below
      }
      if (has) {
        o.put(tl >> 16, tl & 0xFFFFu);
        if (le) o.put(le, len - 3 - kKF.len_base[lc]);
        o.put(to >> 16, to & 0xFFFFu);
        if (oe) o.put(oe, dist - 1 - kKF.off_base[oc]);
      }
      o.finish();
    }
    for (unsigned long long lm = __ballot(longrun); lm; lm &= lm - 1) { // long runs: 64 literals per step
      const int l = __builtin_ctzll(lm);
      const uint32_t u = __shfl(un, l), s0 = __shfl(ls, l);
      const uint32_t plo = __shfl((uint32_t)bpos, l), phi = __shfl((uint32_t)(bpos >> 32), l);
      uint64_t p = ((uint64_t)phi << 32) | plo;
      for (uint32_t k0 = 0; k0 < u; k0 += 64) {
        const uint32_t t = k0 + lane < u ? tab[src[s0 + k0 + lane]] : 0u;
        uint32_t pin = t >> 16;
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_up(pin, o);
          if (lane >= o) pin += v;
        }
        if (t >> 16) k6_or_at(sink, p + pin - (t >> 16), t & 0xFFFFu);
        p += __shfl(pin, 63);
      }
    }
    };
    if (windowed) emit(std::true_type{});
    else emit(std::false_type{});
    if (windowed) { // (the next iteration's two barriers come before its first write into the window)
      __syncthreads();
      k6_store_window(win, words, s0, s1, tid, KF_THREADS);
    }
  }
  __syncthreads();
  if (tid == 0) body_bits[job] = cbit[par] - pos0;
}

// CRC-32 (IEEE, hash/crc32 as gzip.go:58 uses it) of every KC_SEG-byte segment of
// every chunk, each from a zero register: the host joins them (crc32_combine).
constexpr uint32_t KC_SEG = 4096;
__global__ __launch_bounds__(256) void k7_crc32_segments(const uint8_t *__restrict__ src_base,
                                                         const Job *__restrict__ jobs, int njobs, uint32_t segs_per_job,
                                                         uint32_t *__restrict__ out /* njobs x segs_per_job */) {
  __shared__ uint32_t table[256];
  {
    uint32_t c = threadIdx.x;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    table[threadIdx.x] = c;
  }
  __syncthreads();
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t job = (uint32_t)(t / segs_per_job), seg = (uint32_t)(t % segs_per_job);
  if (job >= (uint32_t)njobs) return;
  const Job J = jobs[job];
  const uint64_t b0 = (uint64_t)J.start + (uint64_t)seg * KC_SEG;
  if (b0 >= J.len) return;
  const uint32_t nb = (uint32_t)min((uint64_t)KC_SEG, (uint64_t)J.len - b0);
  const uint8_t *p = src_base + J.src_off + b0;
  uint32_t crc = 0xFFFFFFFFu;
  uint32_t k = 0;
  for (; k + 8 <= nb; k += 8) {
    uint64_t v;
    __builtin_memcpy(&v, p + k, 8);
#pragma unroll
    for (int q = 0; q < 8; q++) crc = table[(crc ^ (uint32_t)(v >> (8 * q))) & 0xFF] ^ (crc >> 8);
  }
  for (; k < nb; k++) crc = table[(crc ^ p[k]) & 0xFF] ^ (crc >> 8);
  out[t] = ~crc;
}

} // namespace m4g
