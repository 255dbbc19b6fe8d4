/*
 * oracle/m4_ref.c -- CPU restatement of matchfinder.M4 (TEST INFRASTRUCTURE;
 * see m4_ref.h for scope and the "parity unpinned" note).
 *
 * Every block below cites the reference lines it follows
 * (paths relative to /root/reference/matchfinder/).
 */
#include "m4_ref.h"

#include <stdlib.h>
#include <string.h>

/* emitter.go:5-16 */
typedef struct {
  int64_t Start, End, Match;
} abs_match;

static int am_is_zero(abs_match m) {
  return m.Start == 0 && m.End == 0 && m.Match == 0;
}

/* emitter.go:19-25 (Dst is the caller's vector) */
typedef struct {
  m4ref_matchvec *Dst;
  int64_t NextEmit;
} emitter;

static void vec_push(m4ref_matchvec *v, m4ref_match m) {
  if (v->len == v->cap) {
    size_t nc = v->cap ? v->cap * 2 : 256;
    v->v = (m4ref_match *)realloc(v->v, nc * sizeof(m4ref_match));
    v->cap = nc;
  }
  v->v[v->len++] = m;
}

/* emitter.go:27-34 */
static void emit(emitter *e, abs_match m) {
  m4ref_match r;
  r.unmatched = (uint32_t)(m.Start - e->NextEmit);
  r.length = (uint32_t)(m.End - m.Start);
  r.distance = (uint32_t)(m.Start - m.Match);
  vec_push(e->Dst, r);
  e->NextEmit = m.End;
}

static uint32_t le32(const uint8_t *p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v; /* little-endian host (x86-64) */
}
static uint64_t le64(const uint8_t *p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}

static int leading_zeros32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

/* m4.go:59-61 */
static int64_t score(const m4ref *q, abs_match m) {
  return (m.End - m.Start) * 256 +
         (int64_t)(leading_zeros32((uint32_t)(m.Start - m.Match)) - 32) *
             q->DistanceBitCost;
}

/* m4.go:285 */
#define HASH_MUL64 0x1E35A7BD1E35A7BDULL

/* m4.go:293-324 (amd64 branch: 8 bytes at a time, then bytewise tail) */
static int64_t extend_match(const uint8_t *src, int64_t len, int64_t i,
                            int64_t j) {
  while (j + 8 < len) {
    uint64_t a = le64(src + i), b = le64(src + j);
    if (a != b) return j + (__builtin_ctzll(a ^ b) >> 3);
    i += 8;
    j += 8;
  }
  for (; j < len && src[i] == src[j]; i++, j++) {
  }
  return j;
}

/* m4.go:328-339 */
static abs_match extend_match2(const uint8_t *src, int64_t len, int64_t start,
                               int64_t candidate, int64_t min) {
  abs_match m;
  int64_t end = extend_match(src, len, candidate + 4, start + 4);
  while (start > min && candidate > 0 && src[start - 1] == src[candidate - 1]) {
    start--;
    candidate--;
  }
  m.Start = start;
  m.End = end;
  m.Match = candidate;
  return m;
}

m4ref *m4ref_new(int64_t max_distance, int64_t min_length, int64_t hash_len,
                 int64_t table_bits, int64_t chain_length,
                 int64_t distance_bit_cost) {
  m4ref *q = (m4ref *)calloc(1, sizeof(m4ref));
  q->MaxDistance = max_distance;
  q->MinLength = min_length;
  q->HashLen = hash_len;
  q->TableBits = table_bits;
  q->ChainLength = chain_length;
  q->DistanceBitCost = distance_bit_cost;
  return q;
}

void m4ref_free(m4ref *q) {
  if (!q) return;
  free(q->table);
  free(q->chain);
  free(q->history);
  free(q);
}

/* m4.go:51-57 */
void m4ref_reset(m4ref *q) {
  if (q->table) memset(q->table, 0, q->table_len * sizeof(uint32_t));
  q->history_len = 0;
  q->chain_len = 0;
}

void m4ref_matchvec_free(m4ref_matchvec *v) {
  free(v->v);
  v->v = NULL;
  v->len = v->cap = 0;
}

static void history_reserve(m4ref *q, size_t need) {
  if (need > q->history_cap) {
    size_t nc = q->history_cap ? q->history_cap : 4096;
    while (nc < need) nc *= 2;
    q->history = (uint8_t *)realloc(q->history, nc);
    q->history_cap = nc;
  }
}
static void chain_reserve(m4ref *q, size_t need) {
  if (need > q->chain_cap) {
    size_t nc = q->chain_cap ? q->chain_cap : 4096;
    while (nc < need) nc *= 2;
    q->chain = (uint32_t *)realloc(q->chain, nc * sizeof(uint32_t));
    q->chain_cap = nc;
  }
}

/* m4.go:63-283 */
int m4ref_find_matches(m4ref *q, m4ref_matchvec *dst, const uint8_t *in,
                       size_t n) {
  /* m4.go:64-78 defaults */
  if (q->MaxDistance == 0) q->MaxDistance = 65535;
  if (q->MinLength == 0) q->MinLength = 4;
  if (q->HashLen == 0) q->HashLen = 6;
  if (q->TableBits == 0) q->TableBits = 17;
  if (q->table_len < ((size_t)1 << q->TableBits)) {
    free(q->table);
    q->table_len = (size_t)1 << q->TableBits;
    q->table = (uint32_t *)calloc(q->table_len, sizeof(uint32_t));
  }

  emitter e;
  e.Dst = dst;
  e.NextEmit = 0;

  /* m4.go:82-99 trim the history buffer */
  if ((int64_t)q->history_len > q->MaxDistance * 2) {
    int64_t delta = (int64_t)q->history_len - q->MaxDistance;
    memmove(q->history, q->history + delta, (size_t)q->MaxDistance);
    q->history_len = (size_t)q->MaxDistance;
    if (q->ChainLength > 0) {
      memmove(q->chain, q->chain + delta,
              (size_t)q->MaxDistance * sizeof(uint32_t));
      q->chain_len = (size_t)q->MaxDistance;
    }
    for (size_t t = 0; t < q->table_len; t++) {
      int64_t newV = (int64_t)q->table[t] - delta;
      if (newV < 0) newV = 0;
      q->table[t] = (uint32_t)newV;
    }
  }

  /* m4.go:101-107 append src to the history buffer */
  e.NextEmit = (int64_t)q->history_len;
  history_reserve(q, q->history_len + n);
  if (n) memcpy(q->history + q->history_len, in, n);
  q->history_len += n;
  if (q->ChainLength > 0) {
    chain_reserve(q, q->chain_len + n);
    memset(q->chain + q->chain_len, 0, n * sizeof(uint32_t));
    q->chain_len += n;
  }
  const uint8_t *src = q->history;
  const int64_t len = (int64_t)q->history_len;

  /* Go evaluates `1<<(8*HashLen) - 1` in uint64; a shift by 64 yields 0, so
   * HashLen 8 gives an all-ones mask (m4.go:141). */
  const uint64_t hash_mask =
      (q->HashLen >= 8) ? ~0ULL : ((1ULL << (8 * q->HashLen)) - 1);
  const int hash_shift = (int)(64 - q->TableBits);

  /* m4.go:111 */
  abs_match matches[3];
  memset(matches, 0, sizeof(matches));
  const abs_match zero = {0, 0, 0};

  for (int64_t i = e.NextEmit; i < len - 7; i++) {
    /* m4.go:113-126 */
    if (!am_is_zero(matches[0]) && i >= matches[0].End) {
      if (!am_is_zero(matches[1])) {
        if (matches[1].End > matches[0].Start)
          matches[1].End = matches[0].Start;
        if (matches[1].End - matches[1].Start >= q->MinLength &&
            score(q, matches[1]) > 0)
          emit(&e, matches[1]);
      }
      emit(&e, matches[0]);
      matches[0] = matches[1] = matches[2] = zero;
    }

    /* m4.go:128-138 repeat match one byte after the current position */
    if (am_is_zero(matches[0]) && e.Dst->len > 0) {
      int64_t prevDistance = (int64_t)e.Dst->v[e.Dst->len - 1].distance;
      if (i + 1 - prevDistance < 0) return -1; /* Go: slice bounds panic */
      if (le32(src + i + 1) == le32(src + i + 1 - prevDistance)) {
        abs_match m =
            extend_match2(src, len, i + 1, i + 1 - prevDistance, e.NextEmit + 1);
        if (m.End - m.Start >= q->MinLength) matches[0] = m;
      }
    }

    /* m4.go:140-147 calculate and store the hash */
    uint64_t h = ((le64(src + i) & hash_mask) * HASH_MUL64) >> hash_shift;
    int64_t candidate = (int64_t)q->table[h];
    q->table[h] = (uint32_t)i;
    if (q->ChainLength > 0 && candidate != 0) {
      int64_t delta = i - candidate;
      q->chain[i] = (uint32_t)delta;
    }

    /* m4.go:149-154 */
    if (i < matches[0].End && i != matches[0].End + 2 - q->HashLen) continue;
    if (candidate == 0 || i - candidate > q->MaxDistance) continue;

    /* m4.go:156-164 */
    abs_match currentMatch = zero;
    if (le32(src + candidate) == le32(src + i)) {
      abs_match m = extend_match2(src, len, i, candidate, e.NextEmit);
      if (m.End - m.Start > q->MinLength && score(q, m) > 0) currentMatch = m;
    }

    /* m4.go:166-181 */
    for (int64_t j = 0; j < q->ChainLength; j++) {
      uint32_t delta = q->chain[candidate];
      if (delta == 0) break;
      candidate -= (int64_t)delta;
      if (candidate <= 0 || i - candidate > q->MaxDistance) break;
      if (le32(src + candidate) == le32(src + i)) {
        abs_match m = extend_match2(src, len, i, candidate, e.NextEmit);
        if (m.End - m.Start > q->MinLength &&
            score(q, m) > score(q, currentMatch))
          currentMatch = m;
      }
    }

    /* m4.go:183-185 */
    if (currentMatch.End - currentMatch.Start < q->MinLength) continue;

    /* m4.go:187-199 */
    int64_t overlapPenalty = 0;
    if (!am_is_zero(matches[0])) {
      overlapPenalty = 275;
      if (currentMatch.Start <= matches[1].End) overlapPenalty = 0;
    }
    if (score(q, currentMatch) <= score(q, matches[0]) + overlapPenalty)
      continue;

    /* m4.go:201-209 */
    matches[2] = matches[1];
    matches[1] = matches[0];
    matches[0] = currentMatch;
    if (am_is_zero(matches[2])) continue;

    /* m4.go:211-259 three matches: emit one and/or eliminate one */
    if (matches[0].Start < matches[2].End) {
      /* m4.go:213-219 */
      matches[1] = matches[2];
      matches[2] = zero;
    } else if (matches[0].Start < matches[2].End + q->MinLength) {
      /* m4.go:221-229 */
      emit(&e, matches[2]);
      matches[1] = zero;
      matches[2] = zero;
    } else {
      /* m4.go:231-258 */
      if (matches[2].End > matches[1].Start) {
        matches[2].End = matches[1].Start;
        if (q->ChainLength > 0 &&
            matches[2].End - matches[2].Start >= q->MinLength) {
          int64_t pos = matches[2].Start;
          int64_t mlen = matches[2].End - matches[2].Start;
          for (;;) {
            int64_t delta = (int64_t)q->chain[pos];
            if (delta == 0) break;
            pos -= delta;
            if (pos <= matches[2].Match) break;
            if (memcmp(src + matches[2].Start, src + pos, (size_t)mlen) == 0) {
              matches[2].Match = pos;
              break;
            }
          }
        }
      }
      if (matches[2].End - matches[2].Start >= q->MinLength &&
          score(q, matches[2]) > 0)
        emit(&e, matches[2]);
      matches[2] = zero;
    }
  }

  /* m4.go:262-273 emit the remaining matches */
  if (!am_is_zero(matches[1])) {
    if (matches[1].End > matches[0].Start) matches[1].End = matches[0].Start;
    if (matches[1].End - matches[1].Start >= q->MinLength &&
        score(q, matches[1]) > 0)
      emit(&e, matches[1]);
  }
  if (!am_is_zero(matches[0])) emit(&e, matches[0]);

  /* m4.go:275-281 */
  if (e.NextEmit < len) {
    m4ref_match r;
    r.unmatched = (uint32_t)(len - e.NextEmit);
    r.length = 0;
    r.distance = 0;
    vec_push(dst, r);
  }
  return 0;
}

int64_t m4ref_find_matches_flat(m4ref *q, const uint8_t *src, size_t n,
                                uint32_t seed_prev_distance, m4ref_match *out,
                                size_t out_cap) {
  m4ref_matchvec v = {0, 0, 0};
  size_t skip = 0;
  if (seed_prev_distance) {
    /* Model a non-empty caller dst whose last element has this Distance
     * (m4.go:129-130 reads only e.Dst[len-1].Distance). */
    m4ref_match s = {0, 0, seed_prev_distance};
    vec_push(&v, s);
    skip = 1;
  }
  if (m4ref_find_matches(q, &v, src, n) != 0) {
    m4ref_matchvec_free(&v);
    return INT64_MIN;
  }
  size_t cnt = v.len - skip;
  int64_t ret;
  if (cnt > out_cap) {
    ret = -(int64_t)cnt;
  } else {
    if (cnt) memcpy(out, v.v + skip, cnt * sizeof(m4ref_match));
    ret = (int64_t)cnt;
  }
  m4ref_matchvec_free(&v);
  return ret;
}

size_t m4ref_dump_table(const m4ref *q, uint32_t *out, size_t n) {
  size_t k = n < q->table_len ? n : q->table_len;
  if (k) memcpy(out, q->table, k * sizeof(uint32_t));
  return k;
}
size_t m4ref_dump_chain(const m4ref *q, uint32_t *out, size_t n) {
  size_t k = n < q->chain_len ? n : q->chain_len;
  if (k) memcpy(out, q->chain, k * sizeof(uint32_t));
  return k;
}
size_t m4ref_history_len(const m4ref *q) { return q->history_len; }
