// enc_host.h -- host side of the GPU encoder (K6): what brotli.Encoder.Encode does
// BETWEEN its two passes, per chunk, on the CPU (encoder.go:107-120): the
// meta-block header, the three prefix codes built from the histograms and their
// serialisation.  O(alphabet) work per chunk, spread over host threads.
// Included by m4g.hip only (host code; no HIP types).
//
// Follows, with the reference lines cited at each function:
//   compress_fragment_two_pass.go:222-241  storeMetaBlockHeaderBW
//   brotli_bit_stream.go:558-635           chooseBitDepths
//   brotli_bit_stream.go:637-741           buildAndStoreHuffmanTreeFastBW
//   entropy_encode.go:550-592              convertBitDepthsToSymbols
//   entropy_encode_static.go:3,779-788     static code-length code (re-derived, not copied:
//                                          the tables of repeat codes are generated from RFC 7932 3.5)
// chooseBitDepths sorts with an unstable sort whose order among equal counts is
// specific to Go's pdqsort; ties are broken by ascending symbol here (any order
// gives a valid code).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

namespace m4g_enc {

// bitwriter.go:11-56, into a growable byte vector; bit 0 of the vector is the first bit written
struct BitWriter {
  std::vector<uint8_t> dst;
  uint64_t bits = 0;
  unsigned nbits = 0;
  uint64_t total = 0; // bits written so far
  void write(unsigned nb, uint64_t b) {
    bits |= b << nbits;
    nbits += nb;
    total += nb;
    while (nbits >= 8) {
      dst.push_back((uint8_t)bits);
      bits >>= 8;
      nbits -= 8;
    }
  }
  void flush() { // pad to a byte boundary with zeros (jumpToByteBoundary)
    if (nbits) {
      dst.push_back((uint8_t)bits);
      bits = 0;
      nbits = 0;
    }
  }
};

// command.go:3-109 kInsExtra / kCopyExtra
static const uint8_t kInsExtra[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
static const uint8_t kCopyExtra[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};

// command.go:141-164 combineLengthCodes
inline uint16_t combine_length_codes(uint16_t inscode, uint16_t copycode, bool use_last_distance) {
  const uint16_t bits64 = (uint16_t)((copycode & 7u) | ((inscode & 7u) << 3));
  if (use_last_distance && inscode < 8 && copycode < 16) return copycode < 8 ? bits64 : (uint16_t)(bits64 | 64);
  uint32_t offset = 2u * ((copycode >> 3) + 3u * (inscode >> 3));
  offset = (offset << 5) + 0x40 + ((0x520D40u >> offset) & 0xC0u);
  return (uint16_t)(offset | bits64);
}
// extra bits (insert + copy) a command symbol carries: the inverse of combineLengthCodes, by enumeration
struct CmdExtra {
  uint8_t n[704];
  CmdExtra() {
    std::memset(n, 0, sizeof(n));
    for (int ic = 0; ic < 24; ic++)
      for (int cc = 0; cc < 24; cc++)
        for (int last = 0; last < 2; last++)
          n[combine_length_codes((uint16_t)ic, (uint16_t)cc, last != 0)] = (uint8_t)(kInsExtra[ic] + kCopyExtra[cc]);
  }
};
// encoder.go:169-177: extra bits of a distance symbol (ring codes 0..15 have none)
inline unsigned dist_extra_bits(unsigned code) { return code < 16 ? 0u : ((code - 16) >> 1) + 1; }

// compress_fragment_two_pass.go:222-241
inline void store_meta_block_header(size_t len, bool is_uncompressed, BitWriter &bw) {
  unsigned nibbles = 6;
  bw.write(1, 0); // ISLAST
  if (len <= (1u << 16)) nibbles = 4;
  else if (len <= (1u << 20)) nibbles = 5;
  bw.write(2, nibbles - 4);
  bw.write(nibbles * 4, (uint64_t)len - 1);
  bw.write(1, is_uncompressed ? 1 : 0);
}

// entropy_encode.go:550-592
inline uint16_t reverse_bits(unsigned num_bits, uint16_t bits) {
  uint16_t r = 0;
  for (unsigned i = 0; i < num_bits; i++, bits >>= 1) r = (uint16_t)((r << 1) | (bits & 1));
  return r;
}
inline void convert_bit_depths_to_symbols(const uint8_t *depth, size_t len, uint16_t *bits) {
  uint16_t bl_count[16] = {0}, next_code[16];
  for (size_t i = 0; i < len; i++) bl_count[depth[i]]++;
  bl_count[0] = 0;
  next_code[0] = 0;
  int code = 0;
  for (int i = 1; i < 16; i++) {
    code = (code + bl_count[i - 1]) << 1;
    next_code[i] = (uint16_t)code;
  }
  for (size_t i = 0; i < len; i++)
    if (depth[i]) bits[i] = reverse_bits(depth[i], next_code[depth[i]]++);
}

// brotli_bit_stream.go:558-635
inline void choose_bit_depths(const uint32_t *histogram, size_t hlen, uint8_t *depth, int max_bits) {
  struct SC {
    uint32_t symbol, count;
  };
  const int total_space = 1 << max_bits;
  SC symbols[704];
  int nsym = 0;
  uint32_t total_count = 0;
  for (size_t i = 0; i < hlen; i++)
    if (histogram[i]) {
      symbols[nsym++] = SC{(uint32_t)i, histogram[i]};
      total_count += histogram[i];
    }
  std::sort(symbols, symbols + nsym, [](const SC &a, const SC &b) {
    return a.count != b.count ? a.count > b.count : a.symbol < b.symbol;
  });
  int boundaries[21] = {0};
  int used = 0;
  const double total_ratio = (double)total_count / (double)total_space;
  int cur = 1;
  for (int i = 0; i < nsym; i++) {
    while (cur < max_bits && (double)symbols[i].count / (double)(1 << (max_bits - cur)) < total_ratio) {
      cur++;
      boundaries[cur] = i;
    }
    used += 1 << (max_bits - cur);
  }
  for (int i = cur + 1; i <= max_bits; i++) boundaries[i] = nsym;
  while (used < total_space) { // move the boundary that buys the most symbol count per code space
    const int available = total_space - used;
    double best_ratio = 0.0;
    int best = 0;
    for (int i = 2; i <= max_bits; i++) {
      const int cost = 1 << (max_bits - i);
      if (cost > available || boundaries[i] == nsym || (i < max_bits && boundaries[i] == boundaries[i + 1])) continue;
      const double ratio = (double)symbols[boundaries[i]].count / (double)cost;
      if (ratio > best_ratio) {
        best_ratio = ratio;
        best = i;
      }
    }
    boundaries[best]++;
    used += 1 << (max_bits - best);
  }
  std::memset(depth, 0, hlen);
  for (int i = 1; i < max_bits; i++)
    for (int j = boundaries[i]; j < boundaries[i + 1]; j++) depth[symbols[j].symbol] = (uint8_t)i;
  for (int j = boundaries[max_bits]; j < nsym; j++) depth[symbols[j].symbol] = (uint8_t)max_bits;
}

// The static code-length code (entropy_encode_static.go:3, 779-788): depths of the 18
// code-length symbols and their canonical LSB-first codes.
struct CodeLengthCode {
  uint8_t depth[18] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 0, 4, 4};
  uint16_t bits[18] = {};
  CodeLengthCode() { convert_bit_depths_to_symbols(depth, 18, bits); }
  // RFC 7932 section 3.5 serialisation of the depths above (HSKIP 0); 40 bits, the
  // reference writes them as the constants 0x55555554 and 0xFF
  void store(BitWriter &bw) const {
    static const uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    static const uint8_t vl_n[6] = {2, 4, 3, 2, 2, 4}, vl_v[6] = {0, 7, 3, 2, 1, 15};
    bw.write(2, 0);
    int space = 0;
    for (int k = 0; k < 18 && space < 32; k++) {
      const uint8_t l = depth[order[k]];
      bw.write(vl_n[l], vl_v[l]);
      if (l) space += 32 >> l;
    }
  }
  // a run of `reps` zero depths: symbol 17 chains (3 extra bits each, count -> 8 (count - 2) + extra + 3)
  void zero_reps(BitWriter &bw, uint32_t reps) const {
    if (reps < 3) {
      for (uint32_t k = 0; k < reps; k++) bw.write(depth[0], bits[0]);
      return;
    }
    uint8_t extra[8];
    int n = 0;
    while (reps > 10) {
      extra[n++] = (uint8_t)((reps - 3) & 7);
      reps = ((reps - 3) >> 3) + 2;
    }
    extra[n++] = (uint8_t)(reps - 3);
    while (n--) {
      bw.write(depth[17], bits[17]);
      bw.write(3, extra[n]);
    }
  }
  // reps >= 3 further copies of the previous non-zero depth: symbol 16 chains (2 extra bits each)
  void nonzero_reps(BitWriter &bw, uint32_t reps) const {
    uint8_t extra[12];
    int n = 0;
    while (reps > 6) {
      extra[n++] = (uint8_t)((reps - 3) & 3);
      reps = ((reps - 3) >> 2) + 2;
    }
    extra[n++] = (uint8_t)(reps - 3);
    while (n--) {
      bw.write(depth[16], bits[16]);
      bw.write(2, extra[n]);
    }
  }
};

// brotli_bit_stream.go:637-741
inline void build_and_store_tree(const uint32_t *histogram, size_t histogram_total, unsigned max_bits, uint8_t *depth,
                                 uint16_t *bits, BitWriter &bw, const CodeLengthCode &cl) {
  size_t count = 0, symbols[4] = {0, 0, 0, 0}, length = 0;
  for (size_t total = histogram_total; total != 0; length++)
    if (histogram[length]) {
      if (count < 4) symbols[count] = length;
      count++;
      total -= histogram[length];
    }
  if (count <= 1) {
    bw.write(4, 1);
    bw.write(max_bits, symbols[0]);
    depth[symbols[0]] = 0;
    bits[symbols[0]] = 0;
    return;
  }
  choose_bit_depths(histogram, length, depth, 14);
  convert_bit_depths_to_symbols(depth, length, bits);
  if (count <= 4) {
    bw.write(2, 1);         // simple prefix code
    bw.write(2, count - 1); // NSYM - 1
    for (size_t i = 0; i < count; i++)
      for (size_t j = i + 1; j < count; j++)
        if (depth[symbols[j]] < depth[symbols[i]]) std::swap(symbols[i], symbols[j]);
    for (size_t i = 0; i < count; i++) bw.write(max_bits, symbols[i]);
    if (count == 4) bw.write(1, depth[symbols[0]] == 1); // tree-select
    return;
  }
  uint8_t previous = 8;
  cl.store(bw);
  for (size_t i = 0; i < length;) {
    const uint8_t value = depth[i];
    size_t reps = 1;
    for (size_t k = i + 1; k < length && depth[k] == value; k++) reps++;
    i += reps;
    if (value == 0) {
      cl.zero_reps(bw, (uint32_t)reps);
    } else {
      if (previous != value) {
        bw.write(cl.depth[value], cl.bits[value]);
        reps--;
      }
      if (reps < 3) {
        for (; reps; reps--) bw.write(cl.depth[value], cl.bits[value]);
      } else {
        cl.nonzero_reps(bw, (uint32_t)reps);
      }
      previous = value;
    }
  }
}

// Everything Encode writes for one chunk before its second pass (encoder.go:107-120), plus what the
// second pass needs: `table` in the layout of kernels K6 (code bits | depth << 16, symbol order of
// m4g_histo) and the exact number of bits the second pass will write.
struct ChunkPlan {
  BitWriter prologue; // meta-block header, 13 zero bits, the three serialised prefix codes (not byte aligned)
  uint64_t body_bits = 0;
};
inline void plan_chunk(const uint32_t *histo /* 1024 words: lit, cmd, dist */, size_t chunk_len, uint32_t *table,
                       ChunkPlan &out, const CodeLengthCode &cl, const CmdExtra &cx) {
  const uint32_t *lit = histo, *cmd = histo + 256, *dist = histo + 960;
  size_t nlit = 0, ncmd = 0, ndist = 0;
  for (int i = 0; i < 256; i++) nlit += lit[i];
  for (int i = 0; i < 704; i++) ncmd += cmd[i];
  for (int i = 0; i < 64; i++) ndist += dist[i];
  uint8_t depth[1024] = {0};
  uint16_t bits[1024] = {0};
  BitWriter &bw = out.prologue;
  store_meta_block_header(chunk_len, false, bw);
  bw.write(13, 0);
  build_and_store_tree(lit, nlit, 8, depth, bits, bw, cl);
  build_and_store_tree(cmd, ncmd, 10, depth + 256, bits + 256, bw, cl);
  build_and_store_tree(dist, ndist, 6, depth + 960, bits + 960, bw, cl);
  uint64_t body = 0;
  for (int i = 0; i < 256; i++) body += (uint64_t)lit[i] * depth[i];
  for (int i = 0; i < 704; i++) body += (uint64_t)cmd[i] * (depth[256 + i] + cx.n[i]);
  for (int i = 0; i < 64; i++) body += (uint64_t)dist[i] * (depth[960 + i] + dist_extra_bits((unsigned)i));
  out.body_bits = body;
  for (int i = 0; i < 1024; i++) table[i] = (uint32_t)bits[i] | ((uint32_t)depth[i] << 16);
}

// OR `nbits` bits of src (bit 0 first) into dst at bit position pos (dst is zero or holds other fields)
inline void or_bits(uint8_t *dst, uint64_t pos, const uint8_t *src, uint64_t nbits) {
  const unsigned sh = (unsigned)(pos & 7);
  uint8_t *d = dst + (pos >> 3);
  const uint64_t nbytes = (nbits + 7) / 8;
  for (uint64_t k = 0; k < nbytes; k++) {
    uint32_t v = src[k];
    if (k == nbytes - 1 && (nbits & 7)) v &= (1u << (nbits & 7)) - 1;
    v <<= sh;
    d[k] |= (uint8_t)v;
    if (v >> 8) d[k + 1] |= (uint8_t)(v >> 8);
  }
}

} // namespace m4g_enc
