/*
 * oracle/encoder_ref.c -- CPU restatement of brotli.Encoder and of
 * matchfinder.Writer.  TEST INFRASTRUCTURE ONLY; see encoder_ref.h.
 *
 * Reference lines followed (paths relative to /root/reference/):
 *   encoder.go:12-177                      Encoder.Reset/Encode, getDistanceCode
 *   command.go:111-164                     insert/copy length codes, combine
 *   fast_log.go:16-18                      log2FloorNonZero
 *   bitwriter.go:11-56                     bitWriter
 *   compress_fragment_two_pass.go:222-241  storeMetaBlockHeaderBW
 *   brotli_bit_stream.go:553-741           chooseBitDepths,
 *                                          buildAndStoreHuffmanTreeFastBW
 *   entropy_encode.go:531-592              reverseBits, convertBitDepthsToSymbols
 *   matchfinder/matchfinder.go:59-93       Writer.Write / writeBlock / Close
 * The insert/copy base+extra tables and the code-length-code constants are
 * re-derived from RFC 7932 (sections 3.5 and 5), not copied.
 */
#include "encoder_ref.h"

#include <stdlib.h>
#include <string.h>

/* ---------- fast_log.go:16-18 ---------- */
static uint32_t log2_floor_nonzero(uint32_t n) {
  return 31u ^ (uint32_t)__builtin_clz(n);
}

/* ---------- RFC 7932 section 5: insert / copy length code tables ---------- */
static const uint8_t kInsExtraBits[24] = {0, 0, 0, 0, 0, 0, 1, 1,  2,  2,  3,  3,
                                          4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
static const uint8_t kCopyExtraBits[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1,  2,  2,
                                           3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
static uint32_t kInsBase[24], kCopyBase[24];
static int tables_ready = 0;
static void init_tables(void) {
  if (tables_ready) return;
  uint32_t b = 0;
  for (int i = 0; i < 24; i++) {
    kInsBase[i] = b;
    b += 1u << kInsExtraBits[i];
  }
  b = 2;
  for (int i = 0; i < 24; i++) {
    kCopyBase[i] = b;
    b += 1u << kCopyExtraBits[i];
  }
  tables_ready = 1;
}

/* ---------- command.go:111-126 ---------- */
uint16_t encref_insert_length_code(uint32_t insertlen) {
  if (insertlen < 6) {
    return (uint16_t)insertlen;
  } else if (insertlen < 130) {
    uint32_t nbits = log2_floor_nonzero(insertlen - 2) - 1;
    return (uint16_t)((nbits << 1) + ((insertlen - 2) >> nbits) + 2);
  } else if (insertlen < 2114) {
    return (uint16_t)(log2_floor_nonzero(insertlen - 66) + 10);
  } else if (insertlen < 6210) {
    return 21;
  } else if (insertlen < 22594) {
    return 22;
  }
  return 23;
}

/* ---------- command.go:128-139 ---------- */
uint16_t encref_copy_length_code(uint32_t copylen) {
  if (copylen < 10) {
    return (uint16_t)(copylen - 2);
  } else if (copylen < 134) {
    uint32_t nbits = log2_floor_nonzero(copylen - 6) - 1;
    return (uint16_t)((nbits << 1) + ((copylen - 6) >> nbits) + 4);
  } else if (copylen < 2118) {
    return (uint16_t)(log2_floor_nonzero(copylen - 70) + 12);
  }
  return 23;
}

/* ---------- command.go:141-164 ---------- */
uint16_t encref_combine_length_codes(uint16_t inscode, uint16_t copycode,
                                     int use_last_distance) {
  uint16_t bits64 = (uint16_t)((copycode & 0x7) | ((inscode & 0x7) << 3));
  if (use_last_distance && inscode < 8 && copycode < 16) {
    return copycode < 8 ? bits64 : (uint16_t)(bits64 | 64);
  }
  uint32_t offset = 2 * (((uint32_t)copycode >> 3) + 3 * ((uint32_t)inscode >> 3));
  offset = (offset << 5) + 0x40 + ((0x520D40u >> offset) & 0xC0);
  return (uint16_t)(offset | bits64);
}

/* ---------- encoder.go:169-177 ---------- */
encref_distcode encref_distance_code(uint32_t distance) {
  encref_distcode r;
  uint32_t d = distance + 3;
  uint32_t nbits = log2_floor_nonzero(d) - 1;
  uint32_t prefix = (d >> nbits) & 1;
  uint32_t offset = (2 + prefix) << nbits;
  r.code = 2 * (nbits - 1) + prefix + 16;
  r.nExtra = nbits;
  r.extraBits = d - offset;
  return r;
}

/* command symbol of match i (shared by both passes: encoder.go:57-63,124-130) */
static uint16_t command_of(const m4ref_match *matches, size_t i,
                           uint16_t *ins_out, uint16_t *copy_out) {
  const m4ref_match *m = &matches[i];
  uint16_t insertCode = encref_insert_length_code(m->unmatched);
  uint16_t copyCode = encref_copy_length_code(m->length);
  if (m->length == 0) copyCode = 2; /* dummy copy length at end of stream */
  if (ins_out) *ins_out = insertCode;
  if (copy_out) *copy_out = copyCode;
  return encref_combine_length_codes(
      insertCode, copyCode, i > 0 && m->distance == matches[i - 1].distance);
}

/* ---------- encoder.go:44-105 (pass 1) ---------- */
void encref_histograms(const uint8_t *src, size_t n, const m4ref_match *matches,
                       size_t nmatches, encref_histo *h,
                       encref_distcode *dist_cache, uint16_t *cmd_out) {
  (void)n;
  memset(h, 0, sizeof(*h));
  size_t pos = 0;
  int64_t d[4] = {-10, -10, -10, -10};
  for (size_t i = 0; i < nmatches; i++) {
    const m4ref_match *m = &matches[i];
    for (size_t k = 0; k < m->unmatched; k++) h->lit[src[pos + k]]++;
    uint16_t command = command_of(matches, i, NULL, NULL);
    h->cmd[command]++;
    if (cmd_out) cmd_out[i] = command;
    if (dist_cache) memset(&dist_cache[i], 0, sizeof(dist_cache[i]));
    if (command >= 128 && m->length != 0) {
      encref_distcode dc = {0, 0, 0};
      int64_t D = (int64_t)m->distance;
      if (D == d[3]) dc.code = 0;
      else if (D == d[2]) dc.code = 1;
      else if (D == d[1]) dc.code = 2;
      else if (D == d[0]) dc.code = 3;
      else if (D == d[3] - 1) dc.code = 4;
      else if (D == d[3] + 1) dc.code = 5;
      else if (D == d[3] - 2) dc.code = 6;
      else if (D == d[3] + 2) dc.code = 7;
      else if (D == d[3] - 3) dc.code = 8;
      else if (D == d[3] + 3) dc.code = 9;
      else dc = encref_distance_code(m->distance);
      if (dist_cache) dist_cache[i] = dc;
      h->dist[dc.code]++;
      if (dc.code != 0) {
        d[0] = d[1];
        d[1] = d[2];
        d[2] = d[3];
        d[3] = D;
      }
    }
    pos += (size_t)m->unmatched + m->length;
  }
}

/* ---------- bitwriter.go:11-56 ---------- */
typedef struct {
  uint8_t *dst;
  size_t len, cap;
  uint64_t bits;
  unsigned nbits;
} bitwriter;

static void bw_append(bitwriter *w, uint8_t b) {
  if (w->len == w->cap) {
    w->cap = w->cap ? w->cap * 2 : 4096;
    w->dst = (uint8_t *)realloc(w->dst, w->cap);
  }
  w->dst[w->len++] = b;
}
static void write_bits(bitwriter *w, unsigned nb, uint64_t b) {
  w->bits |= b << w->nbits;
  w->nbits += nb;
  if (w->nbits >= 32) {
    uint64_t bits = w->bits;
    w->bits >>= 32;
    w->nbits -= 32;
    bw_append(w, (uint8_t)bits);
    bw_append(w, (uint8_t)(bits >> 8));
    bw_append(w, (uint8_t)(bits >> 16));
    bw_append(w, (uint8_t)(bits >> 24));
  }
}
static void jump_to_byte_boundary(bitwriter *w) {
  while (w->nbits != 0) {
    bw_append(w, (uint8_t)w->bits);
    w->bits >>= 8;
    if (w->nbits > 8) w->nbits -= 8;
    else w->nbits = 0;
  }
  w->bits = 0;
}

/* ---------- compress_fragment_two_pass.go:222-241 ---------- */
static void store_meta_block_header(size_t len, int is_uncompressed, bitwriter *bw) {
  unsigned nibbles = 6;
  write_bits(bw, 1, 0); /* ISLAST */
  if (len <= (1u << 16)) nibbles = 4;
  else if (len <= (1u << 20)) nibbles = 5;
  else if (len > (1u << 24)) abort(); /* Go: panic("metablock too long") */
  write_bits(bw, 2, nibbles - 4);
  write_bits(bw, nibbles * 4, (uint64_t)len - 1);
  write_bits(bw, 1, is_uncompressed ? 1 : 0);
}

/* ---------- entropy_encode.go:550-592 ---------- */
static uint16_t reverse_bits(unsigned num_bits, uint16_t bits) {
  uint16_t r = 0;
  for (unsigned i = 0; i < num_bits; i++) {
    r = (uint16_t)((r << 1) | (bits & 1));
    bits >>= 1;
  }
  return r;
}
static void convert_bit_depths_to_symbols(const uint8_t *depth, size_t len,
                                          uint16_t *bits) {
  uint16_t bl_count[16] = {0}, next_code[16];
  int code = 0;
  for (size_t i = 0; i < len; i++) bl_count[depth[i]]++;
  bl_count[0] = 0;
  next_code[0] = 0;
  for (int i = 1; i < 16; i++) {
    code = (code + bl_count[i - 1]) << 1;
    next_code[i] = (uint16_t)code;
  }
  for (size_t i = 0; i < len; i++)
    if (depth[i] != 0) bits[i] = reverse_bits(depth[i], next_code[depth[i]]++);
}

/* ---------- brotli_bit_stream.go:553-635 ---------- */
typedef struct {
  uint32_t symbol, count;
} sym_count;
static int cmp_sym_count(const void *a, const void *b) {
  const sym_count *x = (const sym_count *)a, *y = (const sym_count *)b;
  if (x->count != y->count) return x->count > y->count ? -1 : 1;
  /* tie order: Go's slices.SortFunc is unstable (pdqsort); any order yields
   * a valid code.  We use ascending symbol. */
  return x->symbol < y->symbol ? -1 : (x->symbol > y->symbol);
}
static void choose_bit_depths(const uint32_t *histogram, size_t hlen,
                              uint8_t *depth, int maxBits) {
  int totalCodeSpace = 1 << maxBits;
  sym_count symbols[704];
  int nsym = 0;
  uint32_t totalCount = 0;
  for (size_t i = 0; i < hlen; i++)
    if (histogram[i] != 0) {
      symbols[nsym].symbol = (uint32_t)i;
      symbols[nsym].count = histogram[i];
      nsym++;
      totalCount += histogram[i];
    }
  qsort(symbols, (size_t)nsym, sizeof(sym_count), cmp_sym_count);

  int boundaries[21] = {0};
  int codeSpaceUsed = 0;
  double totalRatio = (double)totalCount / (double)totalCodeSpace;
  int currentDepth = 1;
  for (int i = 0; i < nsym; i++) {
    while (currentDepth < maxBits &&
           (double)symbols[i].count / (double)(1 << (maxBits - currentDepth)) <
               totalRatio) {
      currentDepth++;
      boundaries[currentDepth] = i;
    }
    codeSpaceUsed += 1 << (maxBits - currentDepth);
  }
  for (int i = currentDepth + 1; i <= maxBits; i++) boundaries[i] = nsym;

  while (codeSpaceUsed < totalCodeSpace) {
    int available = totalCodeSpace - codeSpaceUsed;
    double bestRatio = 0.0;
    int bestBoundary = 0;
    for (int i = 2; i <= maxBits; i++) {
      int cost = 1 << (maxBits - i);
      if (cost > available) continue;
      if (boundaries[i] == nsym) continue;
      if (i < maxBits && boundaries[i] == boundaries[i + 1]) continue;
      double ratio = (double)symbols[boundaries[i]].count / (double)cost;
      if (ratio > bestRatio) {
        bestRatio = ratio;
        bestBoundary = i;
      }
    }
    boundaries[bestBoundary]++;
    codeSpaceUsed += 1 << (maxBits - bestBoundary);
  }

  memset(depth, 0, hlen);
  for (int i = 1; i < maxBits; i++)
    for (int j = boundaries[i]; j < boundaries[i + 1]; j++)
      depth[symbols[j].symbol] = (uint8_t)i;
  for (int j = boundaries[maxBits]; j < nsym; j++)
    depth[symbols[j].symbol] = (uint8_t)maxBits;
}

/* ---------- static code-length code (entropy_encode_static.go:3,779-788) ----
 * Depths of the 18 code-length symbols: 0..12 -> 4, 13,14 -> 5, 15 -> 0,
 * 16,17 -> 4.  Codes are the canonical ones, bit-reversed for LSB-first
 * writing; the header is the RFC 7932 section 3.5 serialisation of those
 * depths (HSKIP = 0, order 1,2,3,4,0,5,17,6,16,7,...).  Tests assert that it
 * equals the reference's constant 0xFF55555554 (40 bits). */
static const uint8_t kCLDepth[18] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 0, 4, 4};
static uint16_t kCLBits[18];
static int cl_ready = 0;
static void init_cl(void) {
  if (cl_ready) return;
  convert_bit_depths_to_symbols(kCLDepth, 18, kCLBits);
  cl_ready = 1;
}
static void store_static_code_length_code(bitwriter *bw) {
  static const uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  /* variable-length code for code-length-code lengths 0..5 (nbits, value) */
  static const uint8_t vl_n[6] = {2, 4, 3, 2, 2, 4};
  static const uint8_t vl_v[6] = {0, 7, 3, 2, 1, 15};
  write_bits(bw, 2, 0); /* HSKIP */
  int space = 0;
  for (int k = 0; k < 18 && space < 32; k++) {
    uint8_t l = kCLDepth[order[k]];
    write_bits(bw, vl_n[l], vl_v[l]);
    if (l) space += 32 >> l;
  }
}

/* Test hook: the serialised static code-length code as (bits, nbits); the
 * reference writes the constant 0x55555554 (32 bits) then 0xFF (8 bits),
 * entropy_encode_static.go:785-788. */
uint64_t encref_static_code_length_header(unsigned *nbits) {
  bitwriter bw;
  memset(&bw, 0, sizeof(bw));
  init_cl();
  /* 40 bits: the first 32 are flushed to dst, 8 stay pending */
  store_static_code_length_code(&bw);
  uint64_t v = 0;
  for (size_t i = 0; i < bw.len; i++) v |= (uint64_t)bw.dst[i] << (8 * i);
  v |= bw.bits << (8 * bw.len);
  *nbits = (unsigned)(8 * bw.len) + bw.nbits;
  free(bw.dst);
  return v;
}

/* RFC 7932 section 3.5 repeat codes.  A run of `reps` zeros as a chain of
 * symbol-17 codes (3 extra bits): each further 17 turns the running count c
 * into 8*(c-2) + extra + 3. */
static void write_zero_reps(bitwriter *bw, uint32_t reps) {
  if (reps < 3) {
    for (uint32_t k = 0; k < reps; k++) write_bits(bw, kCLDepth[0], kCLBits[0]);
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
    write_bits(bw, kCLDepth[17], kCLBits[17]);
    write_bits(bw, 3, extra[n]);
  }
}
/* Same for symbol 16 (2 extra bits): c -> 4*(c-2) + extra + 3; reps >= 3. */
static void write_nonzero_reps(bitwriter *bw, uint32_t reps) {
  uint8_t extra[12];
  int n = 0;
  while (reps > 6) {
    extra[n++] = (uint8_t)((reps - 3) & 3);
    reps = ((reps - 3) >> 2) + 2;
  }
  extra[n++] = (uint8_t)(reps - 3);
  while (n--) {
    write_bits(bw, kCLDepth[16], kCLBits[16]);
    write_bits(bw, 2, extra[n]);
  }
}

/* ---------- brotli_bit_stream.go:637-741 ---------- */
static void build_and_store_huffman_tree_fast(const uint32_t *histogram,
                                              size_t histogram_total,
                                              unsigned max_bits, uint8_t *depth,
                                              uint16_t *bits, bitwriter *bw) {
  size_t count = 0, symbols[4] = {0, 0, 0, 0}, length = 0;
  size_t total = histogram_total;
  while (total != 0) {
    if (histogram[length] != 0) {
      if (count < 4) symbols[count] = length;
      count++;
      total -= histogram[length];
    }
    length++;
  }

  if (count <= 1) {
    write_bits(bw, 4, 1);
    write_bits(bw, max_bits, symbols[0]);
    depth[symbols[0]] = 0;
    bits[symbols[0]] = 0;
    return;
  }

  choose_bit_depths(histogram, length, depth, 14);
  convert_bit_depths_to_symbols(depth, length, bits);

  if (count <= 4) {
    write_bits(bw, 2, 1);         /* simple prefix code */
    write_bits(bw, 2, count - 1); /* NSYM - 1 */
    for (size_t i = 0; i < count; i++)
      for (size_t j = i + 1; j < count; j++)
        if (depth[symbols[j]] < depth[symbols[i]]) {
          size_t tmp = symbols[j];
          symbols[j] = symbols[i];
          symbols[i] = tmp;
        }
    for (size_t i = 0; i < count; i++) write_bits(bw, max_bits, symbols[i]);
    if (count == 4) write_bits(bw, 1, depth[symbols[0]] == 1); /* tree-select */
  } else {
    uint8_t previous_value = 8;
    store_static_code_length_code(bw);
    for (size_t i = 0; i < length;) {
      uint8_t value = depth[i];
      size_t reps = 1;
      for (size_t k = i + 1; k < length && depth[k] == value; k++) reps++;
      i += reps;
      if (value == 0) {
        write_zero_reps(bw, (uint32_t)reps);
      } else {
        if (previous_value != value) {
          write_bits(bw, kCLDepth[value], kCLBits[value]);
          reps--;
        }
        if (reps < 3) {
          while (reps != 0) {
            reps--;
            write_bits(bw, kCLDepth[value], kCLBits[value]);
          }
        } else {
          write_nonzero_reps(bw, (uint32_t)reps);
        }
        previous_value = value;
      }
    }
  }
}

/* ---------- encoder.go:6-15 ---------- */
struct encref_encoder {
  int wroteHeader;
  bitwriter bw;
  encref_distcode *distCache;
  size_t distCacheCap;
};

encref_encoder *encref_encoder_new(void) {
  init_tables();
  init_cl();
  return (encref_encoder *)calloc(1, sizeof(encref_encoder));
}
void encref_encoder_free(encref_encoder *e) {
  if (!e) return;
  free(e->bw.dst);
  free(e->distCache);
  free(e);
}
void encref_encoder_reset(encref_encoder *e) {
  e->wroteHeader = 0;
  e->bw.bits = 0;
  e->bw.nbits = 0;
  e->bw.len = 0;
}
const uint8_t *encref_encoder_take(encref_encoder *e, size_t *size) {
  *size = e->bw.len;
  e->bw.len = 0; /* next Encode call starts a fresh dst; pending bits persist */
  return e->bw.dst;
}

/* ---------- encoder.go:17-161 ---------- */
void encref_encode(encref_encoder *e, const uint8_t *src, size_t n,
                   const m4ref_match *matches, size_t nmatches, int last_block) {
  bitwriter *bw = &e->bw;
  if (!e->wroteHeader) {
    write_bits(bw, 4, 15);
    e->wroteHeader = 1;
  }
  if (n == 0) {
    if (last_block) {
      write_bits(bw, 2, 3); /* islast + isempty */
      jump_to_byte_boundary(bw);
    }
    return;
  }

  if (e->distCacheCap < nmatches) {
    free(e->distCache);
    e->distCache = (encref_distcode *)malloc(nmatches * sizeof(encref_distcode));
    e->distCacheCap = nmatches;
  }

  encref_histo h;
  encref_histograms(src, n, matches, nmatches, &h, e->distCache, NULL);
  size_t literalCount = 0, commandCount = 0, distanceCount = 0;
  for (int i = 0; i < 256; i++) literalCount += h.lit[i];
  for (int i = 0; i < 704; i++) commandCount += h.cmd[i];
  for (int i = 0; i < 64; i++) distanceCount += h.dist[i];

  store_meta_block_header(n, 0, bw);
  write_bits(bw, 13, 0);

  uint8_t literalDepths[256] = {0}, commandDepths[704] = {0}, distanceDepths[64] = {0};
  uint16_t literalBits[256] = {0}, commandBits[704] = {0}, distanceBits[64] = {0};
  build_and_store_huffman_tree_fast(h.lit, literalCount, 8, literalDepths, literalBits, bw);
  build_and_store_huffman_tree_fast(h.cmd, commandCount, 10, commandDepths, commandBits, bw);
  build_and_store_huffman_tree_fast(h.dist, distanceCount, 6, distanceDepths, distanceBits, bw);

  size_t pos = 0;
  for (size_t i = 0; i < nmatches; i++) {
    const m4ref_match *m = &matches[i];
    uint16_t insertCode, copyCode;
    uint16_t command = command_of(matches, i, &insertCode, &copyCode);
    write_bits(bw, commandDepths[command], commandBits[command]);
    if (kInsExtraBits[insertCode] > 0)
      write_bits(bw, kInsExtraBits[insertCode],
                 (uint64_t)m->unmatched - kInsBase[insertCode]);
    if (kCopyExtraBits[copyCode] > 0)
      write_bits(bw, kCopyExtraBits[copyCode],
                 (uint64_t)m->length - kCopyBase[copyCode]);
    for (size_t k = 0; k < m->unmatched; k++) {
      uint8_t c = src[pos + k];
      write_bits(bw, literalDepths[c], literalBits[c]);
    }
    if (command >= 128 && m->length != 0) {
      encref_distcode dc = e->distCache[i];
      write_bits(bw, distanceDepths[dc.code], distanceBits[dc.code]);
      if (dc.nExtra > 0) write_bits(bw, dc.nExtra, dc.extraBits);
    }
    pos += (size_t)m->unmatched + m->length;
  }

  if (last_block) {
    write_bits(bw, 2, 3); /* islast + isempty */
    jump_to_byte_boundary(bw);
  }
}

/* ---------- matchfinder/matchfinder.go:59-93 ---------- */
typedef struct {
  uint8_t *p;
  size_t len, cap;
} bytebuf;
static void bb_append(bytebuf *b, const uint8_t *d, size_t n) {
  if (b->len + n > b->cap) {
    size_t nc = b->cap ? b->cap : 65536;
    while (nc < b->len + n) nc *= 2;
    b->p = (uint8_t *)realloc(b->p, nc);
    b->cap = nc;
  }
  if (n) memcpy(b->p + b->len, d, n);
  b->len += n;
}

static void write_block(m4ref *mf, encref_encoder *enc, m4ref_matchvec *mv,
                        const uint8_t *p, size_t n, int last, bytebuf *dest,
                        size_t *nm_total) {
  mv->len = 0; /* w.matches[:0] */
  m4ref_find_matches(mf, mv, p, n);
  if (nm_total) *nm_total += mv->len;
  encref_encode(enc, p, n, mv->v, mv->len, last);
  size_t sz;
  const uint8_t *o = encref_encoder_take(enc, &sz);
  bb_append(dest, o, sz);
}

uint8_t *encref_compress_m4(m4ref *mf, const uint8_t *data, size_t n,
                            size_t block_size, size_t *out_size,
                            size_t *nmatches_out) {
  encref_encoder *enc = encref_encoder_new();
  m4ref_matchvec mv = {0, 0, 0};
  bytebuf dest = {0, 0, 0};
  size_t nm = 0, pos = 0;
  if (block_size == 0) {
    /* matchfinder.go:64-66: each Write is one block */
    write_block(mf, enc, &mv, data, n, 0, &dest, &nm);
    pos = n;
  } else {
    for (; pos + block_size <= n; pos += block_size)
      write_block(mf, enc, &mv, data + pos, block_size, 0, &dest, &nm);
  }
  /* Close (matchfinder.go:89-93): remainder, possibly empty, as last block */
  write_block(mf, enc, &mv, data + pos, n - pos, 1, &dest, &nm);
  encref_encoder_free(enc);
  m4ref_matchvec_free(&mv);
  *out_size = dest.len;
  if (nmatches_out) *nmatches_out = nm;
  if (!dest.p) dest.p = (uint8_t *)malloc(1);
  return dest.p;
}
void encref_free_buf(uint8_t *p) { free(p); }
