// flate_host.h -- host side of the GPU flate encoder (m4g_batch_flate_encode):
// what huffmanBitWriter.writeBlock (flate/huffman_bit_writer.go:361-419) does
// between its two passes over the Match list.  Included by m4g.hip only.
//
// Per chunk, from the symbol counts k5f_flate_stats produced:
//   * the end-of-block count, numLiterals / numOffsets and the "at least one
//     offset" rule (huffman_bit_writer.go:447-464);
//   * the literal/length and offset codes (huffmanEncoder.generate,
//     huffman_code.go:271-309: length-limited code lengths by the boundary
//     package-merge of bitCounts, lines 131-241; canonical codes assigned per
//     length in literal order, lines 245-265, written bit-reversed);
//   * the run-length coded code lengths and their code (generateCodegen,
//     huffman_bit_writer.go:152-237; generate(.., 7));
//   * the sizes of the three encodings and the choice (lines 364-407);
//   * the header bits (lines 302-358) and the exact number of token bits, so that
//     every chunk's place in the one bit stream is known before a bit is written.
// Every sort here has a total order (frequency, then literal), so the result
// does not depend on a sort implementation.
#pragma once

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstring>
#include <vector>

#include "enc_host.h"

namespace m4g_flate {

constexpr int kNumLit = 286, kNumOff = 30, kNumCodegen = 19; // huffman_bit_writer.go:13,22,25
constexpr int kEndBlock = 256, kLengthStart = 257;           // lines 16,19
constexpr int kTab = 320;                                    // = KF_TAB (k6_flate.h)
constexpr size_t kMaxStored = 65535;                         // maxStoreBlockSize, line 26

struct Code {
  uint16_t code = 0, len = 0;
};

inline uint16_t reverse_bits(uint16_t v, unsigned nbits) { // huffman_code.go:344-346
  uint16_t r = 0;
  for (unsigned i = 0; i < nbits; i++) r = (uint16_t)(r | (((v >> i) & 1u) << (nbits - 1 - i)));
  return r;
}

// RFC 1951 3.2.5 extra bits (huffman_bit_writer.go:32-37, 47-51), by the same rule as k6_flate.h
inline int length_extra(int c) { return c < 8 || c == 28 ? 0 : (c - 4) / 4; }
inline int offset_extra(int c) { return c < 4 ? 0 : (c - 2) / 2; }

// A prefix code over up to 286 symbols (huffmanEncoder)
struct Coder {
  Code codes[kNumLit];

  int bit_length(const int32_t *freq, int n) const { // huffman_code.go:106-114
    int total = 0;
    for (int i = 0; i < n; i++) total += freq[i] * (int)codes[i].len;
    return total;
  }

  // huffman_code.go:271-309
  void generate(const int32_t *freq, int n, int max_bits) {
    struct Node {
      uint16_t literal;
      int32_t freq;
    };
    Node list[kNumLit + 1];
    int count = 0;
    for (int i = 0; i < n; i++) {
      if (freq[i] != 0) list[count++] = Node{(uint16_t)i, freq[i]};
      else codes[i].len = 0;
    }
    if (count <= 2) { // both symbols get one bit, in literal order (lines 294-302)
      for (int i = 0; i < count; i++) codes[list[i].literal] = Code{(uint16_t)i, 1};
      return;
    }
    std::sort(list, list + count, [](const Node &a, const Node &b) { // byFreq, lines 335-340
      return a.freq != b.freq ? a.freq < b.freq : a.literal < b.literal;
    });
    list[count] = Node{0xFFFF, INT32_MAX}; // the sentinel of bitCounts (line 137)

    // ---- bitCounts, lines 131-241: symbols per code length under the limit ----
    constexpr int kLevels = 17;
    const int n_sym = count;
    const int top = std::min(max_bits, n_sym - 1);
    struct Level {
      int32_t last_freq, next_char_freq, next_pair_freq, needed;
    } lv[kLevels + 1] = {};
    int32_t leaf[kLevels + 1][kLevels + 1] = {}; // leaf[l][j]: leaves left of the level-j ancestor of level l's last node
    for (int l = 1; l <= top; l++) {
      lv[l] = Level{list[1].freq, list[2].freq, list[0].freq + list[1].freq, 0};
      leaf[l][l] = 2;
    }
    lv[1].next_pair_freq = INT32_MAX;
    lv[top].needed = 2 * n_sym - 4;
    for (int l = top;;) {
      Level &L = lv[l];
      if (L.next_pair_freq == INT32_MAX && L.next_char_freq == INT32_MAX) { // this row is exhausted
        L.needed = 0;
        lv[l + 1].next_pair_freq = INT32_MAX;
        l++;
        continue;
      }
      const int32_t prev = L.last_freq;
      if (L.next_char_freq < L.next_pair_freq) { // next item of the row: a leaf
        const int32_t m = ++leaf[l][l];
        L.last_freq = L.next_char_freq;
        L.next_char_freq = list[m].freq;
      } else { // a pair from the row below
        L.last_freq = L.next_pair_freq;
        std::copy(leaf[l - 1], leaf[l - 1] + l, leaf[l]);
        lv[l - 1].needed = 2;
      }
      if (--L.needed == 0) {
        if (l == top) break;
        lv[l + 1].next_pair_freq = prev + L.last_freq;
        l++;
      } else {
        while (lv[l - 1].needed > 0) l--;
      }
    }
    // ---- assignEncodingAndSize, lines 245-265: bits = 1 takes the most frequent symbols ----
    int remaining = n_sym;
    uint16_t code = 0;
    for (int bits = 1, level = top; level > 0; bits++, level--) {
      code = (uint16_t)(code << 1);
      const int cnt = leaf[top][level] - leaf[top][level - 1];
      if (cnt == 0) continue;
      Node *chunk = list + (remaining - cnt);
      std::sort(chunk, chunk + cnt, [](const Node &a, const Node &b) { return a.literal < b.literal; });
      for (int k = 0; k < cnt; k++) codes[chunk[k].literal] = Code{reverse_bits(code++, (unsigned)bits), (uint16_t)bits};
      remaining -= cnt;
    }
  }
};

// huffman_code.go:63-101
struct FixedCodes {
  Coder lit, off;
  FixedCodes() {
    for (int ch = 0; ch < kNumLit; ch++) {
      unsigned bits, size;
      if (ch < 144) bits = (unsigned)ch + 48, size = 8;
      else if (ch < 256) bits = (unsigned)ch + 400 - 144, size = 9;
      else if (ch < 280) bits = (unsigned)ch - 256, size = 7;
      else bits = (unsigned)ch + 192 - 280, size = 8;
      lit.codes[ch] = Code{reverse_bits((uint16_t)bits, size), (uint16_t)size};
    }
    for (int ch = 0; ch < kNumOff; ch++) off.codes[ch] = Code{reverse_bits((uint16_t)ch, 5), 5};
  }
};

enum Kind { kStored = 0, kFixed = 1, kDynamic = 2 };

struct ChunkPlan {
  int kind = kFixed;
  m4g_enc::BitWriter header; // the 3 block bits (+ the dynamic header); a stored block's LEN / NLEN follow the padding
  uint64_t token_bits = 0;   // what k6f_flate_bodies writes: tokens (Huffman) or 8 * len (stored)
  Code eob;                  // written by the host right after the tokens
};

// stats: KF_TAB counts of k5f_flate_stats.  table: KF_TAB words for k6f (code | len << 16).
inline void plan_chunk(const uint32_t *stats, size_t len, bool eof, uint32_t *table, ChunkPlan &P, const FixedCodes &fx) {
  static const uint8_t kCodegenOrder[kNumCodegen] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int32_t lit_freq[kNumLit], off_freq[kNumOff];
  for (int i = 0; i < kNumLit; i++) lit_freq[i] = (int32_t)stats[i];
  for (int i = 0; i < kNumOff; i++) off_freq[i] = (int32_t)stats[kNumLit + i];
  // real bits of the length / offset extra fields (lines 371-378 count them only for a storable block)
  int64_t extra_true = 0;
  for (int c = 8; c < 29; c++) extra_true += (int64_t)lit_freq[kLengthStart + c] * length_extra(c);
  for (int c = 4; c < kNumOff; c++) extra_true += (int64_t)off_freq[c] * offset_extra(c);
  int64_t n_off_true = 0;
  for (int c = 0; c < kNumOff; c++) n_off_true += off_freq[c];

  // makeStatistics, lines 447-466
  lit_freq[kEndBlock]++;
  int num_lit = kNumLit;
  while (lit_freq[num_lit - 1] == 0) num_lit--;
  int num_off = kNumOff;
  while (num_off > 0 && off_freq[num_off - 1] == 0) num_off--;
  if (num_off == 0) {
    off_freq[0] = 1; // so that the offset tree of a dynamic block can be written
    num_off = 1;
  }
  Coder lit, off, cg;
  lit.generate(lit_freq, kNumLit, 15);
  off.generate(off_freq, kNumOff, 15);

  // writeBlock, lines 364-400
  const bool storable = len <= kMaxStored; // (the library never sees Go's nil slice: an empty chunk is an empty slice)
  const int64_t stored_size = storable ? (int64_t)(len + 5) * 8 : 0;
  const int extra = storable ? (int)extra_true : 0;
  int size = 3 + fx.lit.bit_length(lit_freq, kNumLit) + fx.off.bit_length(off_freq, kNumOff) + extra; // fixedSize
  const Coder *le = &fx.lit, *oe = &fx.off;

  // generateCodegen, lines 152-237
  uint8_t sizes[kNumLit + kNumOff + 1];
  for (int i = 0; i < num_lit; i++) sizes[i] = (uint8_t)lit.codes[i].len;
  for (int i = 0; i < num_off; i++) sizes[num_lit + i] = (uint8_t)off.codes[i].len;
  const int nsizes = num_lit + num_off;
  std::vector<uint8_t> codegen; // symbols 0..18, each of 16 / 17 / 18 followed by its repeat field
  int32_t cg_freq[kNumCodegen] = {};
  auto put = [&](int sym) {
    codegen.push_back((uint8_t)sym);
    cg_freq[sym]++;
  };
  for (int i = 0; i < nsizes;) {
    const int sz = sizes[i];
    int run = 1;
    while (i + run < nsizes && sizes[i + run] == sz) run++;
    i += run;
    if (sz != 0) {
      put(sz);
      run--;
      while (run >= 3) { // "repeat the previous length 3..6 times"
        const int k = std::min(run, 6);
        put(16);
        codegen.push_back((uint8_t)(k - 3));
        run -= k;
      }
    } else {
      while (run >= 11) { // "11..138 zeros"
        const int k = std::min(run, 138);
        put(18);
        codegen.push_back((uint8_t)(k - 11));
        run -= k;
      }
      if (run >= 3) { // "3..10 zeros"
        put(17);
        codegen.push_back((uint8_t)(run - 3));
        run = 0;
      }
    }
    for (; run > 0; run--) put(sz);
  }
  cg.generate(cg_freq, kNumCodegen, 7);
  // dynamicSize, lines 240-256
  int num_cg = kNumCodegen;
  while (num_cg > 4 && cg_freq[kCodegenOrder[num_cg - 1]] == 0) num_cg--;
  const int dyn_header = 3 + 5 + 5 + 4 + 3 * num_cg + cg.bit_length(cg_freq, kNumCodegen) + cg_freq[16] * 2 +
                         cg_freq[17] * 3 + cg_freq[18] * 7;
  const int dyn_size = dyn_header + lit.bit_length(lit_freq, kNumLit) + off.bit_length(off_freq, kNumOff) + extra;
  if (dyn_size < size) {
    size = dyn_size;
    le = &lit;
    oe = &off;
  }

  P.header = m4g_enc::BitWriter();
  if (storable && stored_size < size) { // lines 403-407, writeStoredHeader lines 340-349
    P.kind = kStored;
    P.header.write(3, eof ? 1 : 0);
    P.token_bits = 8ull * len;
    P.eob = Code{};
    std::memset(table, 0, kTab * sizeof(uint32_t));
    return;
  }
  if (le == &fx.lit) { // writeFixedHeader, lines 351-358
    P.kind = kFixed;
    P.header.write(3, eof ? 3 : 2);
  } else { // writeDynamicHeader, lines 302-338
    P.kind = kDynamic;
    P.header.write(3, eof ? 5 : 4);
    P.header.write(5, (uint64_t)(num_lit - 257));
    P.header.write(5, (uint64_t)(num_off - 1));
    P.header.write(4, (uint64_t)(num_cg - 4));
    for (int i = 0; i < num_cg; i++) P.header.write(3, cg.codes[kCodegenOrder[i]].len);
    for (size_t i = 0; i < codegen.size(); i++) {
      const int sym = codegen[i];
      P.header.write(cg.codes[sym].len, cg.codes[sym].code);
      if (sym == 16) P.header.write(2, codegen[++i]);
      else if (sym == 17) P.header.write(3, codegen[++i]);
      else if (sym == 18) P.header.write(7, codegen[++i]);
    }
  }
  // the tokens' bits: every counted symbol but the end-of-block one and the stand-in offset
  lit_freq[kEndBlock]--;
  if (n_off_true == 0) off_freq[0] = 0;
  P.token_bits = (uint64_t)(le->bit_length(lit_freq, kNumLit) + oe->bit_length(off_freq, kNumOff) + extra_true);
  P.eob = le->codes[kEndBlock];
  for (int i = 0; i < kNumLit; i++) table[i] = le->codes[i].code | ((uint32_t)le->codes[i].len << 16);
  for (int i = 0; i < kNumOff; i++) table[kNumLit + i] = oe->codes[i].code | ((uint32_t)oe->codes[i].len << 16);
  for (int i = kNumLit + kNumOff; i < kTab; i++) table[i] = 0;
}

// hash/crc32 (IEEE): crc(A || B) from crc(A), crc(B) and len(B), in the form zlib's
// crc32_combine has had since 1.2.12: multiplication by x^(8 len(B)) modulo the polynomial.
struct Crc32Join {
  static constexpr uint32_t kPoly = 0xEDB88320u;
  uint32_t x2n[32];
  static uint32_t mul(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
      if (a & m) {
        p ^= b;
        if ((a & (m - 1)) == 0) break;
      }
      m >>= 1;
      b = (b & 1) ? (b >> 1) ^ kPoly : b >> 1;
    }
    return p;
  }
  Crc32Join() {
    uint32_t p = 1u << 30; // x^1
    x2n[0] = p;
    for (int n = 1; n < 32; n++) x2n[n] = p = mul(p, p);
  }
  uint32_t x_pow_8n(uint64_t nbytes) const { // x^(8 n)
    uint32_t p = 1u << 31;
    for (unsigned k = 3; nbytes; nbytes >>= 1, k++)
      if (nbytes & 1) p = mul(x2n[k & 31], p);
    return p;
  }
  uint32_t join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) const { return mul(x_pow_8n(len_b), crc_a) ^ crc_b; }
};

} // namespace m4g_flate
