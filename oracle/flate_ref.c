/*
 * oracle/flate_ref.c -- CPU restatement of the reference's flate and gzip
 * encoders.  TEST INFRASTRUCTURE ONLY; see flate_ref.h.
 *
 * Follows, function by function:
 *   flate/token.go:9-80                      lengthCodes, offsetCodes, lengthCode, offsetCode
 *   flate/huffman_code.go:63-104             fixed literal / offset encodings
 *   flate/huffman_code.go:106-114            bitLength
 *   flate/huffman_code.go:131-241            bitCounts (length-limited code lengths)
 *   flate/huffman_code.go:245-265            assignEncodingAndSize
 *   flate/huffman_code.go:271-309            generate
 *   flate/huffman_code.go:335-346            byFreq order, reverseBits
 *   flate/huffman_bit_writer.go:31-63        extra-bit / base tables, codegenOrder
 *   flate/huffman_bit_writer.go:95-138       flush, writeBits, writeBytes
 *   flate/huffman_bit_writer.go:152-237      generateCodegen
 *   flate/huffman_bit_writer.go:240-277      dynamicSize, fixedSize, storedSize
 *   flate/huffman_bit_writer.go:302-358      writeDynamicHeader, writeStoredHeader, writeFixedHeader
 *   flate/huffman_bit_writer.go:361-419      writeBlock
 *   flate/huffman_bit_writer.go:425-468      makeStatistics
 *   flate/huffman_bit_writer.go:472-517      writeTokens, Encode
 *   flate/gzip.go:39-65                      gzipEncoder.Encode
 *   matchfinder/matchfinder.go:59-93         Writer.Write / writeBlock / Close
 * The tables of token.go are generated from the RFC 1951 base values instead
 * of being listed.
 */
#include "flate_ref.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

enum {
  kOffsetCodeCount = FLATEREF_NUM_OFF,
  kEndBlockMarker = 256,
  kLengthCodesStart = 257,
  kCodegenCodeCount = 19,
  kBadCode = 255,
  kMaxNumLit = FLATEREF_NUM_LIT,
  kMaxStoreBlockSize = 65535,
  kBaseMatchLength = 3,
  kBaseMatchOffset = 1,
  kMaxBitsLimit = 16
};

/* huffman_bit_writer.go:31-60 */
static const int8_t kLengthExtraBits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                            2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const int kLengthBase[29] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  10,  12,  14,  16,  20, 24,
                                    28, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 255};
static const int8_t kOffsetExtraBits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3,  3,  4,  4,  5,  5,  6,
                                            6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const int kOffsetBase[30] = {0x000000, 0x000001, 0x000002, 0x000003, 0x000004, 0x000006, 0x000008, 0x00000c,
                                    0x000010, 0x000018, 0x000020, 0x000030, 0x000040, 0x000060, 0x000080, 0x0000c0,
                                    0x000100, 0x000180, 0x000200, 0x000300, 0x000400, 0x000600, 0x000800, 0x000c00,
                                    0x001000, 0x001800, 0x002000, 0x003000, 0x004000, 0x006000};
/* huffman_bit_writer.go:63 */
static const uint32_t kCodegenOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* token.go:11-57 as functions of the tables above: the code of a length is the
 * last code whose base does not exceed it (length 258 has its own code 28);
 * offsetCodes[] covers offsets 0..255 the same way. */
static uint8_t g_length_codes[256], g_offset_codes[256];
static int g_tables_ready;
static void build_tables(void) {
  if (g_tables_ready) return;
  for (int l = 0; l < 256; l++) {
    int c = 0;
    while (c + 1 < 28 && kLengthBase[c + 1] <= l) c++;
    g_length_codes[l] = (uint8_t)(l == 255 ? 28 : c);
  }
  for (int o = 0; o < 256; o++) {
    int c = 0;
    while (c + 1 < 16 && kOffsetBase[c + 1] <= o) c++;
    g_offset_codes[o] = (uint8_t)c;
  }
  g_tables_ready = 1;
}

/* token.go:59-64 */
int flateref_length_code(uint32_t len) {
  build_tables();
  if (len > 258 || len < kBaseMatchLength) return -1;
  return g_length_codes[len - kBaseMatchLength];
}
/* token.go:67-80 */
int flateref_offset_code(uint32_t off_in) {
  build_tables();
  if (off_in > 32768 || off_in < kBaseMatchOffset) return -1;
  uint32_t off = off_in - kBaseMatchOffset;
  if (off < 256) return g_offset_codes[off];
  if ((off >> 7) < 256) return g_offset_codes[off >> 7] + 14;
  return g_offset_codes[off >> 14] + 28;
}

/* ---------- huffman_code.go ---------- */
typedef struct {
  uint16_t code, len;
} hcode;
typedef struct {
  uint16_t literal;
  int32_t freq;
} literal_node;
typedef struct {
  hcode codes[kMaxNumLit];
  int size;
} huffman_encoder;

/* huffman_code.go:344-346 */
static uint16_t reverse_bits(uint16_t number, unsigned bit_length) {
  uint16_t v = (uint16_t)(number << (16 - bit_length)), r = 0;
  for (int i = 0; i < 16; i++) r = (uint16_t)(r | (((v >> i) & 1u) << (15 - i)));
  return r;
}

/* huffman_code.go:63-101 */
static huffman_encoder g_fixed_lit, g_fixed_off;
static int g_fixed_ready;
static void build_fixed(void) {
  if (g_fixed_ready) return;
  g_fixed_lit.size = kMaxNumLit;
  for (unsigned ch = 0; ch < kMaxNumLit; ch++) {
    unsigned bits, size;
    if (ch < 144) {
      bits = ch + 48;
      size = 8;
    } else if (ch < 256) {
      bits = ch + 400 - 144;
      size = 9;
    } else if (ch < 280) {
      bits = ch - 256;
      size = 7;
    } else {
      bits = ch + 192 - 280;
      size = 8;
    }
    g_fixed_lit.codes[ch].code = reverse_bits((uint16_t)bits, size);
    g_fixed_lit.codes[ch].len = (uint16_t)size;
  }
  g_fixed_off.size = 30;
  for (unsigned ch = 0; ch < 30; ch++) {
    g_fixed_off.codes[ch].code = reverse_bits((uint16_t)ch, 5);
    g_fixed_off.codes[ch].len = 5;
  }
  g_fixed_ready = 1;
}

/* huffman_code.go:106-114 */
static int bit_length(const huffman_encoder *h, const int32_t *freq, int n) {
  int total = 0;
  for (int i = 0; i < n; i++)
    if (freq[i] != 0) total += (int)freq[i] * (int)h->codes[i].len;
  return total;
}

typedef struct {
  int32_t level, lastFreq, nextCharFreq, nextPairFreq, needed;
} level_info;

/* huffman_code.go:131-241.  list[0..n) ascending by (freq, literal), room for
 * one more node.  bit_count[1..max_bits_out] = symbols per code length. */
static void bit_counts(literal_node *list, int32_t n, int32_t maxBits, int32_t *bit_count, int32_t *max_bits_out) {
  list[n].literal = 0xFFFF; /* maxNode(), line 57 */
  list[n].freq = INT32_MAX;
  if (maxBits > n - 1) maxBits = n - 1; /* lines 141-143 */

  level_info levels[kMaxBitsLimit + 1];
  int32_t leafCounts[kMaxBitsLimit + 1][kMaxBitsLimit + 1];
  memset(levels, 0, sizeof(levels));
  memset(leafCounts, 0, sizeof(leafCounts));
  for (int32_t level = 1; level <= maxBits; level++) { /* lines 156-169 */
    levels[level].level = level;
    levels[level].lastFreq = list[1].freq;
    levels[level].nextCharFreq = list[2].freq;
    levels[level].nextPairFreq = list[0].freq + list[1].freq;
    leafCounts[level][level] = 2;
    if (level == 1) levels[level].nextPairFreq = INT32_MAX;
  }
  levels[maxBits].needed = 2 * n - 4; /* line 172 */

  int32_t level = maxBits;
  for (;;) { /* lines 175-223 */
    level_info *l = &levels[level];
    if (l->nextPairFreq == INT32_MAX && l->nextCharFreq == INT32_MAX) {
      l->needed = 0;
      levels[level + 1].nextPairFreq = INT32_MAX;
      level++;
      continue;
    }
    const int32_t prevFreq = l->lastFreq;
    if (l->nextCharFreq < l->nextPairFreq) { /* a leaf */
      const int32_t m = leafCounts[level][level] + 1;
      l->lastFreq = l->nextCharFreq;
      leafCounts[level][level] = m;
      l->nextCharFreq = list[m].freq;
    } else { /* a pair from the row below */
      l->lastFreq = l->nextPairFreq;
      for (int32_t k = 0; k < level; k++) leafCounts[level][k] = leafCounts[level - 1][k];
      levels[l->level - 1].needed = 2;
    }
    if (--l->needed == 0) {
      if (l->level == maxBits) break;
      levels[l->level + 1].nextPairFreq = prevFreq + l->lastFreq;
      level++;
    } else {
      while (levels[level - 1].needed > 0) level--;
    }
  }
  /* lines 231-240 */
  int bits = 1;
  const int32_t *counts = leafCounts[maxBits];
  bit_count[0] = 0;
  for (int32_t lv = maxBits; lv > 0; lv--) bit_count[bits++] = counts[lv] - counts[lv - 1];
  *max_bits_out = maxBits;
}

static int by_literal(const void *a, const void *b) {
  const literal_node *x = (const literal_node *)a, *y = (const literal_node *)b;
  return (int)x->literal - (int)y->literal;
}
/* huffman_code.go:335-340 */
static int by_freq(const void *a, const void *b) {
  const literal_node *x = (const literal_node *)a, *y = (const literal_node *)b;
  if (x->freq == y->freq) return (int)x->literal - (int)y->literal;
  return x->freq < y->freq ? -1 : 1;
}

/* huffman_code.go:245-265 */
static void assign_encoding_and_size(huffman_encoder *h, const int32_t *bit_count, int32_t max_bits,
                                     literal_node *list, int32_t list_len) {
  uint16_t code = 0;
  for (int32_t n = 0; n <= max_bits; n++) {
    const int32_t bits = bit_count[n];
    code = (uint16_t)(code << 1);
    if (n == 0 || bits == 0) continue;
    literal_node *chunk = list + (list_len - bits);
    qsort(chunk, (size_t)bits, sizeof(literal_node), by_literal);
    for (int32_t k = 0; k < bits; k++) {
      h->codes[chunk[k].literal].code = reverse_bits(code, (unsigned)n);
      h->codes[chunk[k].literal].len = (uint16_t)n;
      code++;
    }
    list_len -= bits;
  }
}

/* huffman_code.go:271-309 */
static void generate(huffman_encoder *h, const int32_t *freq, int nfreq, int32_t maxBits) {
  literal_node list[kMaxNumLit + 2];
  int32_t count = 0;
  h->size = nfreq;
  for (int i = 0; i < nfreq; i++) {
    if (freq[i] != 0) {
      list[count].literal = (uint16_t)i;
      list[count].freq = freq[i];
      count++;
    } else {
      h->codes[i].len = 0;
    }
  }
  if (count <= 2) { /* lines 294-302: list is in order of increasing literal value */
    for (int32_t i = 0; i < count; i++) {
      h->codes[list[i].literal].code = (uint16_t)i;
      h->codes[list[i].literal].len = 1;
    }
    return;
  }
  qsort(list, (size_t)count, sizeof(literal_node), by_freq);
  int32_t bit_count[kMaxBitsLimit + 1], mb;
  bit_counts(list, count, maxBits, bit_count, &mb);
  assign_encoding_and_size(h, bit_count, mb, list, count);
}

void flateref_generate(const int32_t *freq, int n, int max_bits, uint16_t *code, uint16_t *len) {
  huffman_encoder h;
  memset(&h, 0, sizeof(h));
  generate(&h, freq, n, max_bits);
  for (int i = 0; i < n; i++) {
    len[i] = h.codes[i].len;
    code[i] = h.codes[i].len ? h.codes[i].code : 0;
  }
}

/* ---------- huffman_bit_writer.go ---------- */
typedef struct {
  uint8_t *p;
  size_t len, cap;
} bytebuf;
static void bb_push(bytebuf *b, uint8_t v) {
  if (b->len == b->cap) {
    b->cap = b->cap ? b->cap * 2 : 4096;
    b->p = (uint8_t *)realloc(b->p, b->cap);
  }
  b->p[b->len++] = v;
}
static void bb_append(bytebuf *b, const uint8_t *src, size_t n) {
  if (b->len + n > b->cap) {
    while (b->len + n > b->cap) b->cap = b->cap ? b->cap * 2 : 4096;
    b->p = (uint8_t *)realloc(b->p, b->cap);
  }
  if (n) memcpy(b->p + b->len, src, n);
  b->len += n;
}

typedef struct {
  bytebuf dst;
  uint64_t bits; /* data waiting to be written: the low nbits */
  unsigned nbits;
  int32_t codegenFreq[kCodegenCodeCount];
  int32_t literalFreq[kMaxNumLit];
  int32_t offsetFreq[kOffsetCodeCount];
  uint8_t codegen[kMaxNumLit + kOffsetCodeCount + 1];
  huffman_encoder literalEncoding, offsetEncoding, codegenEncoding;
} bit_writer;

/* lines 95-108 */
static void hbw_flush(bit_writer *w) {
  while (w->nbits != 0) {
    bb_push(&w->dst, (uint8_t)w->bits);
    w->bits >>= 8;
    w->nbits = w->nbits > 8 ? w->nbits - 8 : 0;
  }
  w->bits = 0;
}
/* lines 110-126 */
static void hbw_write_bits(bit_writer *w, int32_t b, unsigned nb) {
  w->bits |= (uint64_t)(uint32_t)b << w->nbits;
  w->nbits += nb;
  if (w->nbits >= 48) {
    const uint64_t bits = w->bits;
    w->bits >>= 48;
    w->nbits -= 48;
    for (int k = 0; k < 6; k++) bb_push(&w->dst, (uint8_t)(bits >> (8 * k)));
  }
}
/* lines 279-295 */
static void hbw_write_code(bit_writer *w, hcode c) { hbw_write_bits(w, c.code, c.len); }
/* lines 128-138 (the caller guarantees nbits % 8 == 0) */
static void hbw_write_bytes(bit_writer *w, const uint8_t *bytes, size_t n) {
  while (w->nbits != 0) {
    bb_push(&w->dst, (uint8_t)w->bits);
    w->bits >>= 8;
    w->nbits -= 8;
  }
  bb_append(&w->dst, bytes, n);
}

/* lines 152-237 */
static void generate_codegen(bit_writer *w, int numLiterals, int numOffsets, const huffman_encoder *litEnc,
                             const huffman_encoder *offEnc) {
  memset(w->codegenFreq, 0, sizeof(w->codegenFreq));
  uint8_t *codegen = w->codegen;
  for (int i = 0; i < numLiterals; i++) codegen[i] = (uint8_t)litEnc->codes[i].len;
  for (int i = 0; i < numOffsets; i++) codegen[numLiterals + i] = (uint8_t)offEnc->codes[i].len;
  codegen[numLiterals + numOffsets] = kBadCode;

  uint8_t size = codegen[0];
  int count = 1, outIndex = 0;
  for (int inIndex = 1; size != kBadCode; inIndex++) {
    const uint8_t nextSize = codegen[inIndex];
    if (nextSize == size) {
      count++;
      continue;
    }
    if (size != 0) {
      codegen[outIndex++] = size;
      w->codegenFreq[size]++;
      count--;
      while (count >= 3) {
        const int n = count < 6 ? count : 6;
        codegen[outIndex++] = 16;
        codegen[outIndex++] = (uint8_t)(n - 3);
        w->codegenFreq[16]++;
        count -= n;
      }
    } else {
      while (count >= 11) {
        const int n = count < 138 ? count : 138;
        codegen[outIndex++] = 18;
        codegen[outIndex++] = (uint8_t)(n - 11);
        w->codegenFreq[18]++;
        count -= n;
      }
      if (count >= 3) {
        codegen[outIndex++] = 17;
        codegen[outIndex++] = (uint8_t)(count - 3);
        w->codegenFreq[17]++;
        count = 0;
      }
    }
    count--;
    for (; count >= 0; count--) {
      codegen[outIndex++] = size;
      w->codegenFreq[size]++;
    }
    size = nextSize;
    count = 1;
  }
  codegen[outIndex] = kBadCode;
}

/* lines 240-256 */
static int dynamic_size(bit_writer *w, const huffman_encoder *litEnc, const huffman_encoder *offEnc, int extraBits,
                        int *numCodegens_out) {
  int numCodegens = kCodegenCodeCount;
  while (numCodegens > 4 && w->codegenFreq[kCodegenOrder[numCodegens - 1]] == 0) numCodegens--;
  const int header = 3 + 5 + 5 + 4 + 3 * numCodegens +
                     bit_length(&w->codegenEncoding, w->codegenFreq, kCodegenCodeCount) + (int)w->codegenFreq[16] * 2 +
                     (int)w->codegenFreq[17] * 3 + (int)w->codegenFreq[18] * 7;
  *numCodegens_out = numCodegens;
  return header + bit_length(litEnc, w->literalFreq, kMaxNumLit) + bit_length(offEnc, w->offsetFreq, kOffsetCodeCount) +
         extraBits;
}
/* lines 259-264 */
static int fixed_size(bit_writer *w, int extraBits) {
  return 3 + bit_length(&g_fixed_lit, w->literalFreq, kMaxNumLit) +
         bit_length(&g_fixed_off, w->offsetFreq, kOffsetCodeCount) + extraBits;
}

/* lines 302-338 */
static void write_dynamic_header(bit_writer *w, int numLiterals, int numOffsets, int numCodegens, int isEof) {
  hbw_write_bits(w, isEof ? 5 : 4, 3);
  hbw_write_bits(w, numLiterals - 257, 5);
  hbw_write_bits(w, numOffsets - 1, 5);
  hbw_write_bits(w, numCodegens - 4, 4);
  for (int i = 0; i < numCodegens; i++) hbw_write_bits(w, w->codegenEncoding.codes[kCodegenOrder[i]].len, 3);
  int i = 0;
  for (;;) {
    const int codeWord = w->codegen[i++];
    if (codeWord == kBadCode) break;
    hbw_write_code(w, w->codegenEncoding.codes[codeWord]);
    switch (codeWord) {
    case 16: hbw_write_bits(w, w->codegen[i++], 2); break;
    case 17: hbw_write_bits(w, w->codegen[i++], 3); break;
    case 18: hbw_write_bits(w, w->codegen[i++], 7); break;
    }
  }
}

/* lines 425-468; -1 where lengthCode / offsetCode panic */
static int make_statistics(bit_writer *w, const m4ref_match *matches, size_t nmatches, const uint8_t *input,
                           int *numLiterals_out, int *numOffsets_out) {
  memset(w->literalFreq, 0, sizeof(w->literalFreq));
  memset(w->offsetFreq, 0, sizeof(w->offsetFreq));
  size_t pos = 0;
  for (size_t k = 0; k < nmatches; k++) {
    const m4ref_match m = matches[k];
    for (uint32_t j = 0; j < m.unmatched; j++) w->literalFreq[input[pos + j]]++;
    pos += m.unmatched;
    if (m.length == 0) continue;
    const int lc = flateref_length_code(m.length), oc = flateref_offset_code(m.distance);
    if (lc < 0 || oc < 0) return -1;
    w->literalFreq[kLengthCodesStart + lc]++;
    w->offsetFreq[oc]++;
    pos += m.length;
  }
  if (numLiterals_out == NULL) return 0; /* flateref_statistics: frequencies only */
  w->literalFreq[kEndBlockMarker]++;
  int numLiterals = kMaxNumLit;
  while (w->literalFreq[numLiterals - 1] == 0) numLiterals--;
  int numOffsets = kOffsetCodeCount;
  while (numOffsets > 0 && w->offsetFreq[numOffsets - 1] == 0) numOffsets--;
  if (numOffsets == 0) {
    w->offsetFreq[0] = 1;
    numOffsets = 1;
  }
  generate(&w->literalEncoding, w->literalFreq, kMaxNumLit, 15);
  generate(&w->offsetEncoding, w->offsetFreq, kOffsetCodeCount, 15);
  *numLiterals_out = numLiterals;
  *numOffsets_out = numOffsets;
  return 0;
}

int flateref_statistics(const uint8_t *src, size_t n, const m4ref_match *matches, size_t nmatches, int32_t *lit_freq,
                        int32_t *off_freq) {
  (void)n;
  bit_writer *w = (bit_writer *)calloc(1, sizeof(bit_writer));
  const int rc = make_statistics(w, matches, nmatches, src, NULL, NULL);
  memcpy(lit_freq, w->literalFreq, sizeof(w->literalFreq));
  memcpy(off_freq, w->offsetFreq, sizeof(w->offsetFreq));
  free(w);
  return rc;
}

/* lines 472-504 */
static void write_tokens(bit_writer *w, const m4ref_match *matches, size_t nmatches, const uint8_t *input,
                         const hcode *leCodes, const hcode *oeCodes) {
  size_t pos = 0;
  for (size_t k = 0; k < nmatches; k++) {
    const m4ref_match m = matches[k];
    for (uint32_t j = 0; j < m.unmatched; j++) hbw_write_code(w, leCodes[input[pos + j]]);
    pos += m.unmatched;
    if (m.length == 0) continue;
    const int lc = flateref_length_code(m.length);
    hbw_write_code(w, leCodes[lc + kLengthCodesStart]);
    if (kLengthExtraBits[lc] > 0)
      hbw_write_bits(w, (int32_t)(m.length - kBaseMatchLength - kLengthBase[lc]), (unsigned)kLengthExtraBits[lc]);
    const int oc = flateref_offset_code(m.distance);
    hbw_write_code(w, oeCodes[oc]);
    if (kOffsetExtraBits[oc] > 0)
      hbw_write_bits(w, (int32_t)(m.distance - kBaseMatchOffset - kOffsetBase[oc]), (unsigned)kOffsetExtraBits[oc]);
    pos += m.length;
  }
}

/* lines 361-419 */
static int write_block(bit_writer *w, const m4ref_match *matches, size_t nmatches, int eof, const uint8_t *input,
                       size_t input_len, int input_is_nil, int *kind_out) {
  int numLiterals, numOffsets;
  if (make_statistics(w, matches, nmatches, input, &numLiterals, &numOffsets)) return -1;

  int extraBits = 0;
  /* storedSize, lines 269-277 */
  const int storable = !input_is_nil && input_len <= kMaxStoreBlockSize;
  const int storedSize = storable ? (int)(input_len + 5) * 8 : 0;
  if (storable) {
    for (int lengthCode = kLengthCodesStart + 8; lengthCode < numLiterals; lengthCode++)
      extraBits += (int)w->literalFreq[lengthCode] * (int)kLengthExtraBits[lengthCode - kLengthCodesStart];
    for (int offsetCode = 4; offsetCode < numOffsets; offsetCode++)
      extraBits += (int)w->offsetFreq[offsetCode] * (int)kOffsetExtraBits[offsetCode];
  }

  const huffman_encoder *literalEncoding = &g_fixed_lit, *offsetEncoding = &g_fixed_off;
  int size = fixed_size(w, extraBits);

  int numCodegens;
  generate_codegen(w, numLiterals, numOffsets, &w->literalEncoding, &w->offsetEncoding);
  generate(&w->codegenEncoding, w->codegenFreq, kCodegenCodeCount, 7);
  const int dynamicSize = dynamic_size(w, &w->literalEncoding, &w->offsetEncoding, extraBits, &numCodegens);
  if (dynamicSize < size) {
    size = dynamicSize;
    literalEncoding = &w->literalEncoding;
    offsetEncoding = &w->offsetEncoding;
  }

  if (storable && storedSize < size) {
    /* writeStoredHeader, lines 340-349 */
    hbw_write_bits(w, eof ? 1 : 0, 3);
    hbw_flush(w);
    hbw_write_bits(w, (int32_t)input_len, 16);
    hbw_write_bits(w, (int32_t)(uint16_t)~(uint16_t)input_len, 16);
    hbw_write_bytes(w, input, input_len);
    if (kind_out) *kind_out = 0;
    return 0;
  }
  if (literalEncoding == &g_fixed_lit) {
    hbw_write_bits(w, eof ? 3 : 2, 3); /* writeFixedHeader, lines 351-358 */
    if (kind_out) *kind_out = 1;
  } else {
    write_dynamic_header(w, numLiterals, numOffsets, numCodegens, eof);
    if (kind_out) *kind_out = 2;
  }
  write_tokens(w, matches, nmatches, input, literalEncoding->codes, offsetEncoding->codes);
  hbw_write_code(w, literalEncoding->codes[kEndBlockMarker]);
  return 0;
}

/* ---------- hash/crc32 (IEEE), used by gzip.go:58 ---------- */
static uint32_t g_crc_table[256];
static int g_crc_ready;
uint32_t flateref_crc32(uint32_t crc, const uint8_t *p, size_t n) {
  if (!g_crc_ready) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      g_crc_table[i] = c;
    }
    g_crc_ready = 1;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; i++) crc = g_crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

/* ---------- the two Encoders ---------- */
struct flateref_encoder {
  bit_writer w;
  int gzip;
  uint32_t mtime;
  uint32_t length, crc;
  int wroteHeader;
};

flateref_encoder *flateref_encoder_new(int gzip, uint32_t mtime) {
  build_tables();
  build_fixed();
  flateref_encoder *e = (flateref_encoder *)calloc(1, sizeof(*e));
  e->gzip = gzip;
  e->mtime = mtime;
  e->w.literalEncoding.size = kMaxNumLit;
  e->w.codegenEncoding.size = kCodegenCodeCount;
  e->w.offsetEncoding.size = kOffsetCodeCount;
  return e;
}
void flateref_encoder_free(flateref_encoder *e) {
  if (!e) return;
  free(e->w.dst.p);
  free(e);
}
/* huffman_bit_writer.go:91-93, gzip.go:23-28 */
void flateref_encoder_reset(flateref_encoder *e) {
  e->w.bits = 0;
  e->w.nbits = 0;
  e->length = 0;
  e->crc = 0;
  e->wroteHeader = 0;
}

static void append_u32(bytebuf *b, uint32_t v) {
  for (int k = 0; k < 4; k++) bb_push(b, (uint8_t)(v >> (8 * k)));
}

int flateref_encode(flateref_encoder *e, const uint8_t *src, size_t n, int src_is_nil, const m4ref_match *matches,
                    size_t nmatches, int last_block, int *kind_out) {
  if (e->gzip && !e->wroteHeader) { /* gzip.go:40-53 */
    bb_push(&e->w.dst, 0x1f);
    bb_push(&e->w.dst, 0x8b);
    bb_push(&e->w.dst, 8);
    bb_push(&e->w.dst, 0);
    append_u32(&e->w.dst, e->mtime);
    bb_push(&e->w.dst, 0);
    bb_push(&e->w.dst, 255);
    e->wroteHeader = 1;
  }
  /* huffman_bit_writer.go:506-517 */
  if (write_block(&e->w, matches, nmatches, last_block, src, n, src_is_nil, kind_out)) return -1;
  if (last_block) hbw_flush(&e->w);
  if (e->gzip) { /* gzip.go:57-63 */
    e->length += (uint32_t)n;
    e->crc = flateref_crc32(e->crc, src, n);
    if (last_block) {
      append_u32(&e->w.dst, e->crc);
      append_u32(&e->w.dst, e->length);
    }
  }
  return 0;
}

const uint8_t *flateref_encoder_take(flateref_encoder *e, size_t *size) {
  *size = e->w.dst.len;
  e->w.dst.len = 0;
  return e->w.dst.p;
}

/* matchfinder.go:59-93 */
uint8_t *flateref_compress_m4(m4ref *mf, const uint8_t *data, size_t n, size_t block_size, int gzip, uint32_t mtime,
                              size_t *out_size) {
  flateref_encoder *enc = flateref_encoder_new(gzip, mtime);
  m4ref_matchvec mv = {0, 0, 0};
  bytebuf dest = {0, 0, 0};
  size_t pos = 0;
  int bad = 0;
  for (int pass = 0; pass < 2 && !bad; pass++) {
    size_t len;
    int last;
    if (pass == 0) {
      if (block_size == 0) { /* each Write is one block (lines 64-66) */
        len = n;
        last = 0;
      } else {
        for (; pos + block_size <= n && !bad; pos += block_size) {
          mv.len = 0;
          m4ref_find_matches(mf, &mv, data + pos, block_size);
          bad = flateref_encode(enc, data + pos, block_size, 0, mv.v, mv.len, 0, NULL);
          size_t sz;
          const uint8_t *o = flateref_encoder_take(enc, &sz);
          bb_append(&dest, o, sz);
        }
        continue;
      }
    } else { /* Close (lines 89-93): the remainder, possibly empty, as the last block */
      len = n - pos;
      last = 1;
    }
    mv.len = 0;
    m4ref_find_matches(mf, &mv, data + pos, len);
    /* Close hands over w.inBuf: nil when BlockSize is 0 or nothing was ever written */
    bad = flateref_encode(enc, data + pos, len, pass == 1 && (n == 0 || block_size == 0), mv.v, mv.len, last, NULL);
    size_t sz;
    const uint8_t *o = flateref_encoder_take(enc, &sz);
    bb_append(&dest, o, sz);
    pos += len;
  }
  flateref_encoder_free(enc);
  m4ref_matchvec_free(&mv);
  if (bad) {
    free(dest.p);
    return NULL;
  }
  *out_size = dest.len;
  if (!dest.p) dest.p = (uint8_t *)malloc(1);
  return dest.p;
}
void flateref_free_buf(uint8_t *p) { free(p); }
