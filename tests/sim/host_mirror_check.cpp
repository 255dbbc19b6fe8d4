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
  std::puts("ok");
  return 0;
}
