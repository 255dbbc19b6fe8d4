// k6_encode.h -- K6: pass 2 of brotli.Encoder.Encode (encoder.go:122-154) on the
// GPU: every chunk's commands, extra bits, literals and distance codes are
// written, LSB first (bitwriter.go:19-33), at their final bit positions of ONE
// output stream.  Included by kernels.h.
//
// The reference's pass 2 is a serial loop over the matches of a block that
// appends to a bit writer whose state carries over from block to block
// (meta-blocks are not byte aligned), which is why Encode runs in stream order on
// one core and bounds a V2 pipeline.  It parallelises once
// bit LENGTHS are known before bits are written:
//   * the host builds the three prefix codes of every chunk from the histograms
//     K5 produced (encoder.go:110-120 stays on the CPU: O(alphabet) work) and gets
//     every chunk's body length from histogram x depth sums, hence the bit
//     position where each chunk's body starts (enc_host.h);
//   * here one workgroup per chunk walks the match list 512 matches at a time: a
//     prefix sum of Unmatched+Length gives each match its input position (as in
//     K5), a prefix sum of per-match bit counts gives it its output bit position,
//     and every thread then writes its own match: command code + insert / copy
//     extra bits, its literals, distance code + extra bits (codes from K5's
//     m4g_code records, so no distance ring here either);
//   * words are merged into the zero-initialised stream with atomic OR: matches
//     of neighbouring threads, chunks and the host-written meta-block headers
//     share words.
// Literal runs of 64 bytes or more are written by the whole wave (64 literals
// per step, positions from a wave prefix sum of their code lengths).
#pragma once
#include <type_traits>

namespace m4g {

constexpr int K6_THREADS = 512;
constexpr int K6_WAVES = K6_THREADS / 64;

// command.go:3-109 (kInsExtra / kCopyExtra; the bases are their running sums, built at compile time)
struct K6LenTables {
  uint8_t ins_extra[24], copy_extra[24];
  uint32_t ins_base[24], copy_base[24];
};
constexpr K6LenTables k6_len_tables() {
  K6LenTables t{};
  const uint8_t ie[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
  const uint8_t ce[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
  uint32_t ib = 0, cbase = 2;
  for (int i = 0; i < 24; i++) {
    t.ins_extra[i] = ie[i];
    t.copy_extra[i] = ce[i];
    t.ins_base[i] = ib;
    t.copy_base[i] = cbase;
    ib += 1u << ie[i];
    cbase += 1u << ce[i];
  }
  return t;
}
__constant__ const K6LenTables kK6Len = k6_len_tables();

// Prefix-code table of a chunk as the host uploads it: 1024 words, symbol order
// literal[256] | command[704] | distance[64] (the order of m4g_histo),
// word = code bits (already bit-reversed for LSB-first writing) | depth << 16.
constexpr int K6_TAB = 1024;

// Where the bits of one iteration (K6_THREADS matches) go.  The span of an iteration is contiguous in the stream
// (bit positions are prefix sums), so it is assembled in an LDS window first and then stored with plain, coalesced
// word stores; only the first and the last word of the span, which neighbours share, are merged into the zeroed
// stream with atomic OR.  A span larger than the window (long literal runs) goes to the stream word by word with
// atomic OR, as every write did before (measured: 2.9 GB read + 4.0 GB written per GiB of input at 1 TB/s).
constexpr int K6_WIN_WORDS = 2048; // 64 Kbit: ~4x the span of 512 matches of text
struct K6Sink {
  uint32_t *base;  // window (LDS) or stream (global)
  uint64_t bit0;   // stream bit position of base's bit 0 (a multiple of 32)
};

// LSB-first bit appender of one thread.
struct K6Out {
  uint32_t *w;   // next word
  uint64_t acc;  // pending bits (the low n)
  uint32_t n;
  __device__ __forceinline__ void start(const K6Sink &s, uint64_t bitpos) {
    w = s.base + ((bitpos - s.bit0) >> 5);
    acc = 0;
    n = (uint32_t)(bitpos & 31); // the bits below belong to somebody else: zero here, harmless to OR
  }
  __device__ __forceinline__ void put(uint32_t nb, uint32_t v) { // nb <= 32, v < 2^nb
    acc |= (uint64_t)v << n;
    n += nb;
    if (n >= 32) {
      atomicOr(w++, (uint32_t)acc);
      acc >>= 32;
      n -= 32;
    }
  }
  __device__ __forceinline__ void finish() {
    if (acc) atomicOr(w, (uint32_t)acc);
  }
};
// one code of up to 32 bits at an arbitrary bit position
__device__ __forceinline__ void k6_or_at(const K6Sink &s, uint64_t bitpos, uint32_t v) {
  const uint64_t x = (uint64_t)v << (bitpos & 31);
  uint32_t *w = s.base + ((bitpos - s.bit0) >> 5);
  if ((uint32_t)x) atomicOr(w, (uint32_t)x);
  if ((uint32_t)(x >> 32)) atomicOr(w + 1, (uint32_t)(x >> 32));
}
__device__ __forceinline__ void k6_or_at(uint32_t *words, uint64_t bitpos, uint32_t v) {
  k6_or_at(K6Sink{words, 0}, bitpos, v);
}

// The window of an iteration whose bits are [s0, s1): chosen before the writes ...
__device__ __forceinline__ bool k6_span_fits(uint64_t s0, uint64_t s1) {
  return s1 - (s0 & ~31ull) <= (uint64_t)K6_WIN_WORDS * 32;
}
// ... and stored after them (all threads; a barrier separates the writes from this).  Leaves the window zero.
__device__ __forceinline__ void k6_store_window(uint32_t *win, uint32_t *words, uint64_t s0, uint64_t s1, int tid, int nthreads) {
  const uint64_t w0 = s0 >> 5, w1 = (s1 + 31) >> 5; // words [w0, w1) hold bits of the span
  for (uint64_t k = w0 + tid; k < w1; k += nthreads) {
    const uint32_t v = win[k - w0];
    win[k - w0] = 0;
    const bool shared_word = (k == w0 && (s0 & 31)) || (k == w1 - 1 && (s1 & 31));
    if (shared_word) {
      if (v) atomicOr(words + k, v);
    } else {
      words[k] = v;
    }
  }
}

__global__ __launch_bounds__(K6_THREADS) void k6_emit_bodies(
    const uint8_t *__restrict__ src_base, const uint32_t *__restrict__ matches /* 3 words each */,
    const CodeRec *__restrict__ codes, const Job *__restrict__ jobs, const uint64_t *__restrict__ out_off,
    const uint32_t *__restrict__ tables /* K6_TAB words per job */, const uint64_t *__restrict__ body_pos /* per job */,
    uint32_t *__restrict__ words, uint64_t *__restrict__ body_bits /* per job: bits written (for the host's check) */) {
  __shared__ uint32_t tab[K6_TAB];
  __shared__ uint32_t win[K6_WIN_WORDS + 2];
  __shared__ uint32_t wadv[2][K6_WAVES], wbits[2][K6_WAVES];
  __shared__ uint32_t cpos[2];
  __shared__ uint64_t cbit[2];
  const uint32_t job = blockIdx.x;
  const Job J = jobs[job];
  const uint64_t o0 = out_off[job], o1 = out_off[job + 1];
  const uint32_t n = (uint32_t)(o1 - o0);
  const uint32_t *rec = matches + o0 * 3;
  const CodeRec *cr = codes + o0;
  const uint8_t *src = src_base + J.src_off;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int k = tid; k < K6_TAB; k += K6_THREADS) tab[k] = tables[(size_t)job * K6_TAB + k];
  for (int k = tid; k < K6_WIN_WORDS + 2; k += K6_THREADS) win[k] = 0;
  if (tid == 0) {
    cpos[0] = J.start;
    cbit[0] = body_pos[job];
  }
  __syncthreads();

  uint32_t par = 0;
  for (uint32_t base = 0; base < n; base += K6_THREADS, par ^= 1) {
    const uint32_t j = base + tid;
    uint32_t un = 0, len = 0;
    CodeRec c{0, 0, 0, 0};
    if (j < n) {
      un = rec[3 * j];
      len = rec[3 * j + 1];
      c = cr[j];
    }
    // input position of this match's literals: prefix sum of Unmatched + Length (encoder.go:153)
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
    if (tid == K6_THREADS - 1) cpos[par ^ 1] = before + incl;

    // ---- bit count of this match (encoder.go:131-151) ----
    const bool has_dist = j < n && c.command >= 128 && len != 0;
    const uint32_t ic = insert_length_code(un), cc = len ? copy_length_code(len) : 2u;
    const uint32_t tc = tab[256 + c.command], td = tab[960 + c.dist_code];
    const uint32_t ien = kK6Len.ins_extra[ic], cen = kK6Len.copy_extra[cc];
    const uint32_t head_bits = j < n ? (tc >> 16) + ien + cen : 0u;
    const uint32_t dist_bits = has_dist ? (td >> 16) + c.n_extra : 0u;
    uint32_t lit_bits = 0;
    const bool longrun = un >= 64;
    if (!longrun)
      for (uint32_t k = 0; k < un; k++) lit_bits += tab[src[ls + k]] >> 16;
    for (unsigned long long lm = __ballot(longrun); lm; lm &= lm - 1) { // long runs: the whole wave adds up
      const int l = __builtin_ctzll(lm);
      const uint32_t u = __shfl(un, l), s0 = __shfl(ls, l);
      uint32_t s = 0;
      for (uint32_t k = lane; k < u; k += 64) s += tab[src[s0 + k]] >> 16;
      for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
      if (lane == l) lit_bits = s;
    }
    const uint32_t total = head_bits + lit_bits + dist_bits;
    uint32_t binc = total;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(binc, o);
      if (lane >= o) binc += t;
    }
    if (lane == 63) wbits[par][wid] = binc;
    __syncthreads();
    const uint64_t s0 = cbit[par];
    uint64_t bpos = s0, s1 = s0;
    for (int w = 0; w < K6_WAVES; w++) {
      if (w < wid) bpos += wbits[par][w];
      s1 += wbits[par][w];
    }
    if (tid == K6_THREADS - 1) cbit[par ^ 1] = bpos + binc;
    bpos += binc - total; // where this match's bits begin
    const bool windowed = k6_span_fits(s0, s1); // workgroup-uniform
    // (two instantiations, so that the window's atomics are LDS instructions and the stream's are global ones)
    auto emit = [&](auto to_window) {
    const K6Sink sink = decltype(to_window)::value ? K6Sink{win, s0 & ~31ull} : K6Sink{words, 0};

    // ---- write (encoder.go:131-151) ----
    if (j < n) {
      K6Out o;
      o.start(sink, bpos);
      o.put(tc >> 16, tc & 0xFFFFu);
      if (ien) o.put(ien, un - kK6Len.ins_base[ic]);
      if (cen) o.put(cen, len - kK6Len.copy_base[cc]);
      if (!longrun)
        for (uint32_t k = 0; k < un; k++) {
          const uint32_t t = tab[src[ls + k]];
          o.put(t >> 16, t & 0xFFFFu);
        }
      o.finish();
      if (has_dist) {
        o.start(sink, bpos + head_bits + lit_bits);
        o.put(td >> 16, td & 0xFFFFu);
        if (c.n_extra) o.put(c.n_extra, c.extra_bits);
        o.finish();
      }
    }
    for (unsigned long long lm = __ballot(longrun); lm; lm &= lm - 1) { // long runs: 64 literals per step
      const int l = __builtin_ctzll(lm);
      const uint32_t u = __shfl(un, l), s0 = __shfl(ls, l);
      const uint32_t hb = __shfl(head_bits, l);
      const uint32_t plo = __shfl((uint32_t)bpos, l), phi = __shfl((uint32_t)(bpos >> 32), l);
      uint64_t p = (((uint64_t)phi << 32) | plo) + hb;
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
      k6_store_window(win, words, s0, s1, tid, K6_THREADS);
    }
  }
  __syncthreads();
  if (tid == 0) body_bits[job] = cbit[par] - body_pos[job];
}

} // namespace m4g
