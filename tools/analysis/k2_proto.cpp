// tools/analysis/k2_proto.cpp -- host check of the K2 form of the chain walk (m4_k2.h): for every chunk, the
// per-position results are computed by walking the chains on the CPU, the parse runs once with them and once
// with its own walks, and the two record lists must be equal.  Prints how many positions keep 0 / 1 / 2
// candidates or fall back.  Analysis tool, not product, not a test (tests/test_gpu_k2.py checks the kernels).
//   g++ -O2 -std=c++17 -o /tmp/k2_proto tools/analysis/k2_proto.cpp -ldl && /tmp/k2_proto [chunks] [kind] [chain] [hashlen] [minlen] [cost] [maxdist]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <vector>
static uint64_t g_cnt[64];
#define M4G_STAT(k) (g_cnt[k]++)
#define M4G_STATN(k, n) (g_cnt[k] += (n))
#include "../../brotli_amd/csrc/m4_parse.h"
using namespace m4g;
struct VecSink { std::vector<Rec> v; void push(const Rec &r) { v.push_back(r); } };

static void window(const uint8_t *src, int n, int p, uint32_t *w) {
  uint8_t b[48];
  for (int k = 0; k < 48; k++) {
    const int q = p - K2_BW + k;
    b[k] = (q >= 0 && q < n) ? src[q] : 0xA5; // what lies outside must not matter
  }
  memcpy(w, b, 48);
}

int main(int argc, char **argv) {
  const int chunks = argc > 1 ? atoi(argv[1]) : 4;
  const int kind = argc > 2 ? atoi(argv[2]) : 0;
  const int chain_len = argc > 3 ? atoi(argv[3]) : 64;
  const int hash_len = argc > 4 ? atoi(argv[4]) : 5;
  const int min_len = argc > 5 ? atoi(argv[5]) : 4;
  const int cost = argc > 6 ? atoi(argv[6]) : 66;
  const int max_dist = argc > 7 ? atoi(argv[7]) : 16777200;
  const int n = argc > 8 ? atoi(argv[8]) : 1 << 20;
  void *h = dlopen("brotli_amd/libm4gsynth.so", RTLD_NOW);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  auto fill = (void (*)(void *, size_t, uint64_t, size_t, int, int))dlsym(h, "m4g_synth_fill");
  std::vector<uint8_t> buf((size_t)n * chunks + 64);
  fill(buf.data(), n, 0, chunks, kind, 4);
  uint64_t passed = 0, hist[8] = {0}, marked = 0, capped = 0, bad = 0, recs = 0;
  for (int c = 0; c < chunks; c++) {
    const uint8_t *src = buf.data() + (size_t)c * n;
    std::vector<uint32_t> table(1 << 17, 0), chain(n + 8, 0);
    const uint64_t mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
    for (int i = 0; i < n - 7; i++) {
      uint64_t hh = ((ld64(src + i) & mask) * 0x1E35A7BD1E35A7BDull) >> (64 - 17);
      uint32_t ck = (ld32(src + i) * 0x9E3779B1u) >> 27; if (!ck) ck = 1;
      uint32_t old = table[hh];
      table[hh] = ((uint32_t)i << 5) | ck;
      uint32_t pp = old >> 5;
      chain[i] = pp ? (((uint32_t)i - pp) | (((old & 31) == ck) ? kChainMaybe : 0)) : 0;
    }
    ParseParams P{max_dist, min_len, hash_len, chain_len, cost};
    // plain parse
    VecSink a;
    {
      ParseState S; state_init(S, 0, 0);
      GlobalMem M{src, chain.data()};
      parse_range(M, n, P, S, n - 7, a);
      parse_finish(n, P, S, a);
    }
    // K2 on the CPU: walk order = chain order
    std::vector<K2Res> k2(n);
    std::vector<uint32_t> chain2(chain);
    for (int i = 0; i < n - 7; i++) {
      K2Res r{};
      int cnt = 0;
      bool fb = false;
      int32_t ch0 = INT32_MIN, ch1 = INT32_MIN;
      uint32_t d = chain[i] & kChainDelta;
      int cand = i;
      uint32_t wi[12];
      window(src, n, i, wi);
      for (int j = 0; j <= chain_len && !fb; j++) {
        if (d == 0 || d >= (uint32_t)cand) break; // none, or position 0 and below
        cand -= (int)d;
        if (i - cand > max_dist) break;
        uint32_t wc[12], info;
        window(src, n, cand, wc);
        const int af = n - i < K2_FW ? n - i : K2_FW, ab = cand < K2_BW ? cand : K2_BW;
        if (k2_pair(wi, wc, af, ab, info) && k2_filter(ch0, ch1, (uint32_t)(i - cand) | info, min_len, cost)) {
          passed++;
          if (!k2_add(r.e, cnt, (uint32_t)(i - cand) | info, min_len, cost)) fb = true;
        }
        d = chain[cand] & kChainDelta;
      }
      if (fb) { r = K2Res{}; r.e[0] = kK2Fallback; hist[7]++; }
      else {
        hist[cnt]++;
        for (int k = 0; k < cnt; k++) if (k2_capped(r.e[k])) { capped++; break; }
        for (int k = cnt; k < K2_ENTRIES; k++) r.e[k] = 0;
      }
      k2[i] = r;
      const bool mark = fb || cnt > 0;
      marked += mark;
      chain2[i] = (chain[i] & kChainDelta) | (mark ? kChainMaybe : 0u);
    }
    for (int light = 0; light < 2; light++) {
      VecSink b;
      ParseState S; state_init(S, 0, 0);
      GlobalMem M{src, chain2.data(), k2.data()};
      if (light) parse_range<false, true, true, true, true>(M, n, P, S, n - 7, b);
      else parse_range<false, false, true, true, true>(M, n, P, S, n - 7, b);
      parse_finish(n, P, S, b);
      if (a.v.size() != b.v.size() || memcmp(a.v.data(), b.v.data(), a.v.size() * sizeof(Rec))) {
        bad++;
        size_t k = 0;
        while (k < a.v.size() && k < b.v.size() && !memcmp(&a.v[k], &b.v[k], sizeof(Rec))) k++;
        fprintf(stderr, "chunk %d light %d: DIFFERENT at record %zu of %zu / %zu\n", c, light, k, a.v.size(), b.v.size());
      }
    }
    recs += a.v.size();
  }
  const double tot = (double)n * chunks;
  printf("bytes %.0f records %llu; different chunks %llu\n", tot, (unsigned long long)recs, (unsigned long long)bad);
  printf("positions with 0 / 1 / 2 kept / fallback: %.4f %.4f %.4f %.4f; marked %.4f; with a capped entry %.4f\n",
         hist[0] / tot, hist[1] / tot, (hist[2]+hist[3]+hist[4]) / tot, hist[7] / tot, marked / tot, capped / tot);
  printf("chain hops per KiB (walks left in the parse) %.1f, k2 selections per KiB %.1f, iterations per KiB %.1f\n",
         g_cnt[24] / tot * 1024 / 3, g_cnt[30] / tot * 1024 / 2, g_cnt[0] / tot * 1024 / 3);
  printf("candidates through the champion test into k2_add: %.3f per position\n", passed / tot);
  return bad != 0;
}
