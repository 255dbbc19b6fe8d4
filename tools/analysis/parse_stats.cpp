// tools/analysis/parse_stats.cpp -- iteration mix of the serial parse (m4_parse.h) on the bench's
// synthetic text: what a K3a lane does per loop iteration.  Analysis tool, not product, not a test.
//   g++ -O2 -std=c++17 -o /tmp/parse_stats tools/analysis/parse_stats.cpp -ldl && /tmp/parse_stats
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
struct NullSink { uint64_t n = 0; void push(const Rec &) { n++; } };
int main(int argc, char **argv) {
  const int n = 1 << 20;
  const int chunks = argc > 1 ? atoi(argv[1]) : 4;
  const int kind = argc > 2 ? atoi(argv[2]) : 0;
  const int chain_len = argc > 3 ? atoi(argv[3]) : 0;
  const int hash_len = argc > 4 ? atoi(argv[4]) : 6;
  void *h = dlopen("brotli_amd/libm4gsynth.so", RTLD_NOW);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  auto fill = (void (*)(void *, size_t, uint64_t, size_t, int, int))dlsym(h, "m4g_synth_fill");
  std::vector<uint8_t> buf((size_t)n * chunks + 16);
  fill(buf.data(), n, 0, chunks, kind, 4);
  uint64_t recs = 0;
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
    ParseParams P{1 << 20, 4, hash_len, chain_len, 66};
    if (chain_len) P.max_distance = 16777200;
    ParseState S; state_init(S, 0, 0);
    GlobalMem M{src, chain.data()};
    NullSink sink;
    parse_range(M, n, P, S, n - 7, sink);
    parse_finish(n, P, S, sink);
    recs += sink.n;
  }
  const char *names[] = {"iterations", "flush", "was_empty", "do_probe", "probe_load", "probe_4hit", "probe_accept",
    "exit_D_interior", "exit_no_cand", "maybe", "cand_hit", "cand_cur", "exit_short", "reach_F", "exit_F", "push",
    "resolve_G", "G_drop_mid", "G_emit_old", "G_default", "G_clip", "ext_fwd_calls", "ext_fwd_steps", "ext_bwd_calls",
    "chain_hops", "hop_hit", "interior_iter", "far_reads", "maybe_nocandok", "light_skips"};
  const double tot = (double)n * chunks;
  printf("bytes %.0f records %llu (%.1f B/rec)\n", tot, (unsigned long long)recs, tot / recs);
  for (unsigned k = 0; k < sizeof(names) / sizeof(*names); k++)
    printf("%-18s %12llu  per-iter %.4f  per-KiB %.2f\n", names[k], (unsigned long long)g_cnt[k],
           (double)g_cnt[k] / g_cnt[0], g_cnt[k] / tot * 1024);
}
