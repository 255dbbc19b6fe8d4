/*
 * oracle/bargain_ref.c -- see bargain_ref.h.  TEST INFRASTRUCTURE ONLY.
 *
 * One body serves the three variants; what differs between bargain1.go,
 * bargain2.go and bargain3.go is listed in `variant_of` below.
 */
#include "bargain_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- Go's math.Frexp / math.Log / math.Log2 (pure Go on amd64) ------------- */

/* math.Log (log.go): the FreeBSD e_log.c algorithm, same operation order */
static double go_log(double x) {
  static const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10,
                      L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                      L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                      L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                      L7 = 1.479819860511658591e-01;
  if (isnan(x) || (isinf(x) && x > 0)) return x;
  if (x < 0) return NAN;
  if (x == 0) return -INFINITY;
  int ki;
  double f1 = frexp(x, &ki); /* math.Frexp: frac in [1/2, 1) */
  if (f1 < 0.70710678118654752440 /* Sqrt2/2 */) {
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

/* math.Log2 (log10.go:21-37) */
double bargainref_log2(double x) {
  if (isnan(x) || isinf(x) || x <= 0) return go_log(x); /* Frexp passes these through; Log classifies them */
  int e;
  const double frac = frexp(x, &e);
  if (frac == 0.5) return (double)(e - 1);
  return go_log(frac) * (1 / 0.693147180559945309417232121458176568) + (double)e;
}

uint32_t bargainref_byte_cost_bits(uint64_t total, uint32_t count) {
  /* bargain1.go:49-50: cost := max(math.Log2(float64(len(src))/float64(n)), 1) */
  double c = bargainref_log2((double)total / (double)count);
  if (!isnan(c) && c < 1) c = 1; /* builtin max: NaN if either operand is NaN */
  const float f = (float)c;
  uint32_t b;
  memcpy(&b, &f, 4);
  return b;
}

/* ---- the match finders ---------------------------------------------------- */

typedef struct {
  uint32_t val;
  int32_t offset;
} table_entry; /* zfast.go:8-11 */

typedef struct {
  uint32_t length, distance;
  float cost;
} arrival; /* pathfinder.go:57-61 */

#define BASE_MATCH_COST 4.0f /* pathfinder.go:64 */

typedef struct {
  int ntables;
  int bits[3];
  int tail;    /* positions i > len(src)-tail are not hashed: 8 (bargain1.go:136), 12 (bargain3.go:153) */
  int overlap; /* nextOverlapSearch = max(.., m.End-overlap): 4 (bargain1.go:179), 6 (bargain2.go:195) */
} variant_desc;

static variant_desc variant_of(int v) {
  variant_desc d;
  memset(&d, 0, sizeof d);
  if (v == 1) {
    d.ntables = 1, d.bits[0] = 18, d.tail = 8, d.overlap = 4; /* bargain1.go:10-13 */
  } else if (v == 2) {
    d.ntables = 2, d.bits[0] = 18, d.bits[1] = 18, d.tail = 8, d.overlap = 6; /* bargain2.go:10-13 */
  } else {
    d.ntables = 3, d.bits[0] = 17, d.bits[1] = 18, d.bits[2] = 19, d.tail = 12, d.overlap = 6; /* bargain3.go:20-22 */
  }
  return d;
}

struct bargainref {
  int variant;
  variant_desc d;
  int64_t MaxDistance;
  int Skip;
  uint8_t *history;
  size_t hlen, hcap;
  table_entry *table[3];
  arrival *arrivals;
  size_t acap;
};

static uint64_t load64(const uint8_t *p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
static uint32_t load32(const uint8_t *p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}

/* bargain1.go:210-212, bargain2.go:242-248, bargain3.go:268-278 */
static uint32_t hash_of(const bargainref *z, int t, uint64_t u, uint32_t e) {
  const int sh = 64 - z->d.bits[t];
  if (z->variant == 1) return (uint32_t)(((u << 16) * 227718039650203ull) >> sh);
  if (t == 0) return (uint32_t)(((u << 24) * 889523592379ull) >> sh);
  if (t == 1) return (uint32_t)((u * 0xcf1bbcdcb7a56463ull) >> sh);
  return (uint32_t)((u * 0xcf1bbcdcb7a56463ull + (uint64_t)e * (2654435761ull << 32)) >> sh);
}

bargainref *bargainref_new(int variant, int64_t max_distance, int skip) {
  if (variant < 1 || variant > 3) return NULL;
  bargainref *z = calloc(1, sizeof *z);
  if (!z) return NULL;
  z->variant = variant;
  z->d = variant_of(variant);
  z->MaxDistance = max_distance;
  z->Skip = skip;
  for (int t = 0; t < z->d.ntables; t++) {
    z->table[t] = calloc((size_t)1 << z->d.bits[t], sizeof(table_entry));
    if (!z->table[t]) {
      bargainref_free(z);
      return NULL;
    }
  }
  return z;
}

void bargainref_free(bargainref *z) {
  if (!z) return;
  for (int t = 0; t < 3; t++) free(z->table[t]);
  free(z->history);
  free(z->arrivals);
  free(z);
}

void bargainref_reset(bargainref *z) {
  for (int t = 0; t < z->d.ntables; t++) memset(z->table[t], 0, sizeof(table_entry) << z->d.bits[t]);
  z->hlen = 0;
}

static int bits_len(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; } /* bits.Len */

typedef struct {
  int64_t Start, End, Match;
} abs_match; /* emitter.go:5-13 */

/* m4.go:287-324 (the byte loop; the 8-byte steps give the same result) */
static int64_t extend_match(const uint8_t *src, int64_t len, int64_t i, int64_t j) {
  while (j < len && src[i] == src[j]) i++, j++;
  return j;
}

/* m4.go:326-339 */
static abs_match extend_match2(const uint8_t *src, int64_t len, int64_t start, int64_t candidate, int64_t min) {
  abs_match m;
  m.End = extend_match(src, len, candidate + 4, start + 4);
  while (start > min && candidate > 0 && src[start - 1] == src[candidate - 1]) start--, candidate--;
  m.Start = start;
  m.Match = candidate;
  return m;
}

/* bargain1.go:86-108 */
static void add_match(arrival *arrivals, int64_t historyLen, abs_match m, int64_t unmatched, int repeat) {
  float startCost = 0;
  if (m.Start > historyLen) startCost = arrivals[m.Start - historyLen - 1].cost;
  const float insertCost = (float)bits_len((uint64_t)unmatched);
  float distanceCost = 0;
  if (!repeat) distanceCost = (float)bits_len((uint64_t)(m.Start - m.Match));
  float cost = startCost + BASE_MATCH_COST;
  cost = cost + insertCost;
  cost = cost + distanceCost;
  for (int64_t j = m.End; j >= m.Start + 3; j--) {
    arrival *a = &arrivals[j - historyLen - 1];
    if (a->cost > 0 && a->cost <= cost) break;
    a->length = (uint32_t)(j - m.Start);
    a->distance = (uint32_t)(m.Start - m.Match);
    a->cost = cost;
  }
}

static int vec_push(m4ref_matchvec *v, m4ref_match m) {
  if (v->len == v->cap) {
    size_t nc = v->cap ? v->cap * 2 : 64;
    m4ref_match *nv = realloc(v->v, nc * sizeof *nv);
    if (!nv) return -1;
    v->v = nv;
    v->cap = nc;
  }
  v->v[v->len++] = m;
  return 0;
}

int bargainref_find_matches(bargainref *z, m4ref_matchvec *dst, const uint8_t *in, size_t n) {
  if (z->MaxDistance == 0) z->MaxDistance = 1 << 16;

  /* bargain1.go:43-51 */
  uint32_t histogram[256] = {0};
  for (size_t k = 0; k < n; k++) histogram[in[k]]++;
  float byteCost[256];
  for (int b = 0; b < 256; b++) {
    const uint32_t bitsv = bargainref_byte_cost_bits(n, histogram[b]);
    memcpy(&byteCost[b], &bitsv, 4);
  }

  /* :55-64 */
  if (z->acap < n) {
    free(z->arrivals);
    z->arrivals = malloc((n ? n : 1) * sizeof(arrival));
    if (!z->arrivals) return -1;
    z->acap = n;
  }
  arrival *arrivals = z->arrivals;
  memset(arrivals, 0, n * sizeof(arrival));

  /* :66-80 */
  if ((int64_t)z->hlen > z->MaxDistance * 2) {
    const int64_t delta = (int64_t)z->hlen - z->MaxDistance;
    memmove(z->history, z->history + delta, (size_t)z->MaxDistance);
    z->hlen = (size_t)z->MaxDistance;
    for (int t = 0; t < z->d.ntables; t++) {
      const size_t tn = (size_t)1 << z->d.bits[t];
      for (size_t k = 0; k < tn; k++) {
        const int32_t v = z->table[t][k].offset - (int32_t)delta;
        if (v < 0) z->table[t][k].val = 0, z->table[t][k].offset = 0;
        else z->table[t][k].offset = v;
      }
    }
  }

  /* :82-84 */
  const int64_t historyLen = (int64_t)z->hlen;
  if (z->hlen + n > z->hcap) {
    size_t nc = (z->hlen + n) * 2 + 64;
    uint8_t *nh = realloc(z->history, nc);
    if (!nh) return -1;
    z->history = nh;
    z->hcap = nc;
  }
  memcpy(z->history + z->hlen, in, n);
  z->hlen += n;
  const uint8_t *src = z->history;
  const int64_t len = (int64_t)z->hlen;

  int64_t nextOverlapSearch = 0;

  for (int64_t i = historyLen; i < len; i++) {
    arrival arrivedHere = {0, 0, 0};
    if (i > historyLen) arrivedHere = arrivals[i - historyLen - 1];

    int64_t unmatched = 0;
    if (arrivedHere.distance == 0) unmatched = arrivedHere.length;
    int64_t prevDistance = 0;
    if (unmatched != 0 && i - unmatched > historyLen) prevDistance = arrivals[i - historyLen - 1 - unmatched].distance;

    const float literalCost = byteCost[src[i]];
    arrival *nextArrival = &arrivals[i - historyLen];
    if (nextArrival->cost == 0 || arrivedHere.cost + literalCost < nextArrival->cost) {
      nextArrival->cost = arrivedHere.cost + literalCost;
      nextArrival->length = (uint32_t)(unmatched + 1);
      nextArrival->distance = 0;
    }

    if (i > len - z->d.tail) continue; /* no room to check hashes */

    const uint64_t cv = load64(src + i);
    const uint32_t extra = z->variant == 3 ? load32(src + i + 8) : 0;
    table_entry cand[3];
    for (int t = 0; t < z->d.ntables; t++) {
      const uint32_t h = hash_of(z, t, cv, extra);
      cand[t] = z->table[t][h];
      z->table[t][h].offset = (int32_t)i;
      z->table[t][h].val = (uint32_t)cv;
    }

    /* :148-157 */
    if (prevDistance != 0 && prevDistance != (int64_t)arrivals[i - historyLen - 1 + 4].distance) {
      const int64_t repIndex = i - prevDistance;
      if (repIndex >= 0 && load32(src + repIndex) == (uint32_t)cv) {
        const abs_match m = extend_match2(src, len, i, repIndex, i);
        add_match(arrivals, historyLen, m, unmatched, 1);
      }
    }

    if (z->Skip && i % 2 != 0) continue;

    const int nextByteIsUnmatched = arrivals[i - historyLen - 1 + 1].distance == 0;

    if (unmatched > 0 || i >= nextOverlapSearch || nextByteIsUnmatched) {
      for (int t = 0; t < z->d.ntables; t++) {
        const table_entry c = cand[t];
        if ((int64_t)c.offset < i && i - (int64_t)c.offset < z->MaxDistance && (uint32_t)cv == c.val &&
            load32(src + c.offset) == (uint32_t)cv) {
          abs_match m = extend_match2(src, len, i, c.offset, historyLen);
          const int64_t delta = i - m.Start;
          if (delta == 0) {
            add_match(arrivals, historyLen, m, unmatched, 0);
          } else {
            /* extended backwards: add it with and without the extra */
            add_match(arrivals, historyLen, m, unmatched - delta > 0 ? unmatched - delta : 0, 0);
            m.Start += delta;
            m.Match += delta;
            add_match(arrivals, historyLen, m, unmatched, 0);
          }
          int64_t nos = nextOverlapSearch;
          if (m.Start + 1 > nos) nos = m.Start + 1;
          if (m.End - z->d.overlap > nos) nos = m.End - z->d.overlap;
          nextOverlapSearch = nos;
        }
      }
    }
  }

  /* :184-207: walk the shortest path backward; the list is built reversed */
  const size_t first = dst->len;
  int64_t i = (int64_t)n - 1;
  while (i >= 0) {
    const arrival a = arrivals[i];
    if (a.distance > 0) {
      const m4ref_match m = {0, a.length, a.distance};
      if (vec_push(dst, m)) return -1;
    } else {
      if (dst->len == first) {
        const m4ref_match m = {0, 0, 0};
        if (vec_push(dst, m)) return -1;
      }
      dst->v[dst->len - 1].unmatched = a.length;
    }
    i -= (int64_t)a.length;
  }
  for (size_t lo = first, hi = dst->len; lo + 1 < hi; lo++, hi--) {
    const m4ref_match t = dst->v[lo];
    dst->v[lo] = dst->v[hi - 1];
    dst->v[hi - 1] = t;
  }
  return 0;
}

int64_t bargainref_find_matches_flat(bargainref *z, const uint8_t *src, size_t n, m4ref_match *out, size_t out_cap) {
  m4ref_matchvec v = {0, 0, 0};
  if (bargainref_find_matches(z, &v, src, n)) {
    m4ref_matchvec_free(&v);
    return INT64_MIN;
  }
  int64_t r = (int64_t)v.len;
  if (v.len > out_cap) r = -r;
  else if (v.len) memcpy(out, v.v, v.len * sizeof *out);
  m4ref_matchvec_free(&v);
  return r;
}

int64_t bargainref_find_blocks(bargainref *z, const uint8_t *src, size_t n, size_t block_size, m4ref_match *out,
                               size_t out_cap, uint64_t *block_off) {
  m4ref_matchvec v = {0, 0, 0};
  size_t k = 0;
  for (size_t p = 0; p < n; p += block_size, k++) {
    block_off[k] = v.len;
    const size_t m = n - p < block_size ? n - p : block_size;
    if (bargainref_find_matches(z, &v, src + p, m)) {
      m4ref_matchvec_free(&v);
      return INT64_MIN;
    }
  }
  block_off[k] = v.len;
  int64_t r = (int64_t)v.len;
  if (v.len > out_cap) r = -r;
  else if (v.len) memcpy(out, v.v, v.len * sizeof *out);
  m4ref_matchvec_free(&v);
  return r;
}
