// Compile-and-link check of include/m4g_matchfinder.hpp (C++ host mirror).
// Run on a GPU box it also does one FindMatches through the mirror.
#include <cstdio>
#include <cstring>

#include "../../include/m4g_matchfinder.hpp"

int main() {
  if (m4g_device_count() == 0) {
    std::puts("no gfx950 device: link check only");
    return 0;
  }
  matchfinder::M4 mf;
  std::vector<matchfinder::Match> dst;
  const char *s = "0123456789abcdef0123456789abcdef";
  mf.FindMatches(dst, reinterpret_cast<const uint8_t *>(s), std::strlen(s));
  if (dst.size() != 1 || dst[0].Unmatched != 16 || dst[0].Length != 16 || dst[0].Distance != 16) {
    std::puts("FAIL");
    return 1;
  }
  // block-parallel call: 4 blocks of 8 KiB of a periodic text; every block after
  // the first finds its content one block back
  std::vector<uint8_t> data(4 * 8192);
  for (size_t i = 0; i < data.size(); i++) data[i] = (uint8_t)("the quick brown fox "[i % 20] + (i % 8192) / 1024);
  matchfinder::M4 mb;
  std::vector<matchfinder::Match> all;
  std::vector<size_t> off;
  mb.FindMatchesBlocks(all, data.data(), data.size(), 8192, off);
  matchfinder::M4 ms;
  for (size_t b = 0; b < 4; b++) {
    std::vector<matchfinder::Match> one;
    ms.FindMatches(one, data.data() + b * 8192, 8192);
    if (off[b + 1] - off[b] != one.size()) {
      std::puts("FAIL blocks (count)");
      return 1;
    }
    for (size_t k = 0; k < one.size(); k++) {
      const matchfinder::Match &x = all[off[b] + k];
      if (x.Unmatched != one[k].Unmatched || x.Length != one[k].Length || x.Distance != one[k].Distance) {
        std::puts("FAIL blocks (match)");
        return 1;
      }
    }
  }
  // the Writer with both plug-ins on the GPU: a brotli stream of those bytes written in ragged pieces and, after
  // Reset, in one piece -- the same bytes either way (the Python tests compare such streams with the restated
  // Writer); the flate encoder refuses this input as the reference does (matches longer than 258 bytes)
  struct Sink {
    std::vector<uint8_t> v;
    void operator()(const uint8_t *p, size_t n) { v.insert(v.end(), p, p + n); }
  } a, b, f;
  m4g_params fields{};
  fields.max_distance = 1 << 16;
  matchfinder::GPUWriter<Sink> w(&a, fields, M4G_FORMAT_BROTLI, 8192);
  w.Write(data.data(), 5);
  w.Write(data.data() + 5, 20000);
  w.Write(data.data() + 20005, data.size() - 20005);
  w.Close();
  w.Reset(&b);
  w.Write(data.data(), data.size());
  w.Close();
  if (a.v.size() < 8 || (a.v[0] & 15) != 15 || a.v != b.v || a.v.size() > data.size() / 4) {
    std::puts("FAIL writer");
    return 1;
  }
  bool refused = false;
  try {
    matchfinder::GPUWriter<Sink> wf(&f, matchfinder::GPUWriter<Sink>::flateFields(6), M4G_FORMAT_GZIP, 8192, 42);
    wf.Write(data.data(), data.size());
    wf.Close();
  } catch (const std::runtime_error &) {
    refused = true;
  }
  if (!refused) {
    std::puts("FAIL writer (token limits)");
    return 1;
  }
  // Bargain1 over the same bytes in 8192-byte blocks: every block's list tiles its block, copies equal bytes
  // (the Python tests compare the lists with the restated finder)
  {
    matchfinder::Bargain bg;
    bg.MaxDistance = 1 << 16;
    std::vector<matchfinder::Match> bm;
    std::vector<size_t> boff;
    bg.FindMatchesBlocks(bm, data.data(), data.size(), 8192, boff);
    const size_t nb = (data.size() + 8191) / 8192;
    if (boff.size() != nb + 1) {
      std::puts("FAIL bargain (blocks)");
      return 1;
    }
    for (size_t b = 0; b < nb; b++) {
      size_t pos = b * 8192;
      for (size_t k = boff[b]; k < boff[b + 1]; k++) {
        pos += bm[k].Unmatched;
        for (int t = 0; t < bm[k].Length; t++)
          if (bm[k].Distance <= 0 || (size_t)bm[k].Distance > pos + t || data[pos + t] != data[pos + t - bm[k].Distance]) {
            std::puts("FAIL bargain (copy)");
            return 1;
          }
        pos += bm[k].Length;
      }
      if (pos != std::min(data.size(), (b + 1) * 8192)) {
        std::puts("FAIL bargain (tiling)");
        return 1;
      }
    }
  }
  // Bargain1 as a MatchFinder behind the Writer template: block-by-block FindMatches equals the whole-stream lists
  {
    matchfinder::Bargain bg;
    bg.MaxDistance = 1 << 16;
    bg.BlockSize = 8192;
    std::vector<matchfinder::Match> whole, calls;
    std::vector<size_t> boff;
    bg.FindMatchesBlocks(whole, data.data(), data.size(), 8192, boff);
    matchfinder::MatchFinder *mf = &bg;
    for (size_t p = 0; p < data.size(); p += 8192) mf->FindMatches(calls, data.data() + p, std::min<size_t>(8192, data.size() - p));
    bool same = whole.size() == calls.size();
    for (size_t k = 0; same && k < whole.size(); k++)
      same = whole[k].Unmatched == calls[k].Unmatched && whole[k].Length == calls[k].Length && whole[k].Distance == calls[k].Distance;
    if (!same) {
      std::puts("FAIL bargain (MatchFinder calls)");
      return 1;
    }
    mf->Reset();
  }
  // NewWriterV2(dst, 5) as a handle: ragged Writes and one Write give the same stream
  {
    Sink v1, v2;
    matchfinder::GPUWriter<Sink> wa(&v1, matchfinder::GPUWriter<Sink>::V2{5});
    wa.Write(data.data(), 20005);
    wa.Write(data.data() + 20005, data.size() - 20005);
    wa.Close();
    wa.Reset(&v2);
    wa.Write(data.data(), data.size());
    wa.Close();
    if (v1.v.size() < 8 || v1.v != v2.v || v1.v.size() > data.size() / 4) {
      std::puts("FAIL writer v2");
      return 1;
    }
  }
  std::puts("ok");
  return 0;
}
