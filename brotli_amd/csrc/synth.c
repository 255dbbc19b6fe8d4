/*
 * synth.c -- deterministic synthetic workloads for bench.py and the tests.
 *
 * BASELINE.json quotes its metric on "4 GiB synthetic Markov text, 1 MiB
 * independent chunks" and on a "low-entropy binary log"; the reference ships
 * no such generator (its only corpus, testdata/Isaac.Newton-Opticks.txt, does
 * not travel to the GPU box), so the generators are defined here, integer-only
 * and seeded per chunk so that any worker can produce chunk c independently
 * (definition: SURVEY.md section 8(d)).
 *
 * Plain C, host only, multi-threaded with pthreads.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* ---- vocabulary: 8192 words, length 2..10, quadratically skewed letters ---- */
#define VOCAB 8192
static uint8_t vocab_len[VOCAB];
static char vocab[VOCAB][10];
static pthread_once_t vocab_once = PTHREAD_ONCE_INIT;

static void build_vocab(void) {
  static const char alphabet[] = "etaoinshrdlcumwfgypbvkjxqz";
  uint64_t s = 0x9E3779B97F4A7C15ULL;
  for (int w = 0; w < VOCAB; w++) {
    uint64_t r = splitmix64(&s);
    int len = 2 + (int)(r % 9);
    vocab_len[w] = (uint8_t)len;
    for (int k = 0; k < len; k++) {
      uint64_t r32 = splitmix64(&s) >> 32;
      uint64_t idx = (((r32 * r32) >> 32) * 26) >> 32;
      vocab[w][k] = alphabet[idx];
    }
  }
}

/* One chunk of Markov-ish text: half the time the next word is one of four
 * fixed successors of the previous word, otherwise a cubic-skewed fresh draw. */
void m4g_synth_text_chunk(uint8_t *out, size_t n, uint64_t chunk_index) {
  pthread_once(&vocab_once, build_vocab);
  uint64_t seed = 0xB5AD4ECEDA1CE2A9ULL + chunk_index;
  uint64_t s = splitmix64(&seed);
  size_t pos = 0;
  uint32_t prev = 0;
  int line = 0, cap = 1;
  while (pos < n) {
    uint64_t r = splitmix64(&s);
    uint32_t idx;
    if (r & 1) {
      idx = (uint32_t)((prev * 2654435761u + (uint32_t)((r >> 1) & 3)) & (VOCAB - 1));
    } else {
      uint64_t r32 = r >> 32;
      idx = (uint32_t)((((((r32 * r32) >> 32) * r32) >> 32) * VOCAB) >> 32);
    }
    prev = idx;
    int len = vocab_len[idx];
    for (int k = 0; k < len && pos < n; k++) {
      char c = vocab[idx][k];
      if (k == 0 && cap) c = (char)(c - 32);
      out[pos++] = (uint8_t)c;
    }
    cap = 0;
    line += len;
    if (((r >> 8) & 15) == 0) {
      if (pos < n) out[pos++] = '.';
      line++;
      cap = 1;
    }
    if (pos < n) {
      if (line > 72) {
        out[pos++] = '\n';
        line = 0;
      } else {
        out[pos++] = ' ';
        line++;
      }
    }
  }
}

/* One chunk of a low-entropy binary log: 64-byte records
 *   u32le ts (monotone), u16le level, u16le source, u64le counter,
 *   48-byte body = one of 16 ASCII templates, zero padded. */
void m4g_synth_log_chunk(uint8_t *out, size_t n, uint64_t chunk_index) {
  static const char *tmpl[16] = {
      "connection accepted from peer",
      "connection closed by peer",
      "request completed status=200",
      "request completed status=404",
      "cache miss key lookup fell through",
      "cache hit",
      "flushing write-ahead log segment",
      "compaction started level=2",
      "compaction finished",
      "heartbeat ok",
      "retrying rpc after timeout backoff=250ms",
      "tls handshake complete cipher=aes128gcm",
      "worker idle, parking thread",
      "queue depth above high watermark",
      "checkpoint written",
      "gc pause",
  };
  uint64_t seed = 0xD1B54A32D192ED03ULL + chunk_index;
  uint64_t s = splitmix64(&seed);
  uint32_t ts = (uint32_t)(chunk_index * 1000003u);
  uint64_t counter = chunk_index << 20;
  size_t pos = 0;
  while (pos < n) {
    uint8_t rec[64];
    uint64_t r = splitmix64(&s);
    ts += 1 + (uint32_t)(r % 16);
    uint32_t lv = (uint32_t)((r >> 8) & 7);
    uint16_t level = lv < 4 ? 0 : (lv < 6 ? 1 : (lv < 7 ? 2 : 3));
    uint16_t source = (uint16_t)((r >> 16) % 32);
    counter += 1;
    memset(rec, 0, sizeof(rec));
    memcpy(rec + 0, &ts, 4);
    memcpy(rec + 4, &level, 2);
    memcpy(rec + 6, &source, 2);
    memcpy(rec + 8, &counter, 8);
    const char *t = tmpl[(r >> 24) % 16];
    size_t tl = strlen(t);
    memcpy(rec + 16, t, tl > 48 ? 48 : tl);
    size_t k = n - pos < 64 ? n - pos : 64;
    memcpy(out + pos, rec, k);
    pos += k;
  }
}

typedef struct {
  uint8_t *out;
  size_t chunk_bytes, nchunks;
  uint64_t first_chunk;
  int kind, tid, nthreads;
} job;

static void *worker(void *p) {
  job *j = (job *)p;
  for (size_t c = (size_t)j->tid; c < j->nchunks; c += (size_t)j->nthreads) {
    uint8_t *o = j->out + c * j->chunk_bytes;
    if (j->kind == 0) m4g_synth_text_chunk(o, j->chunk_bytes, j->first_chunk + c);
    else m4g_synth_log_chunk(o, j->chunk_bytes, j->first_chunk + c);
  }
  return NULL;
}

/* kind 0 = text, 1 = log.  Fills nchunks * chunk_bytes bytes. */
void m4g_synth_fill(uint8_t *out, size_t chunk_bytes, uint64_t first_chunk,
                    size_t nchunks, int kind, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  pthread_t th[64];
  job jobs[64];
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (job){out, chunk_bytes, nchunks, first_chunk, kind, t, nthreads};
    pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}
