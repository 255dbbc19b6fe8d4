// Probe: does one ds_wrxchg_rtn_b32 instruction whose lanes hit the same LDS
// address perform its lanes in ascending lane order?  (K1's exchange-based
// chain build relies on it; libm4g repeats a short form of this at create time.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(1024) void probe(const uint32_t *idx, uint32_t *ret, uint32_t *fin, int iters, int tsize) {
  extern __shared__ uint32_t lds[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t *table = lds + wave * tsize; // one table per wave
  for (int k = lane; k < tsize; k += 64) table[k] = 0;
  __syncthreads();
  const size_t base = ((size_t)blockIdx.x * 16 + wave) * iters * 64;
  for (int it = 0; it < iters; it++) {
    const uint32_t a = idx[base + it * 64 + lane];
    const uint32_t v = it * 64 + lane + 1;
    uint32_t old = 0;
    if (a != 0xFFFFFFFFu) old = __hip_atomic_exchange(&table[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ret[base + it * 64 + lane] = old;
  }
  __syncthreads();
  for (int k = lane; k < tsize; k += 64) fin[((size_t)blockIdx.x * 16 + wave) * tsize + k] = table[k];
}

int main() {
  const int blocks = 512, iters = 256, tsize = 2048;
  const size_t n = (size_t)blocks * 16 * iters * 64;
  std::vector<uint32_t> idx(n), ret(n), fin((size_t)blocks * 16 * tsize);
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
  for (size_t w = 0; w < n / 64; w++) {
    const int mode = rnd() % 8;
    const uint32_t range = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 7 : mode == 3 ? 33 : mode == 4 ? 64 : mode == 5 ? 300 : tsize;
    for (int l = 0; l < 64; l++) {
      uint32_t a = rnd() % range;
      if (mode == 7 && rnd() % 4 == 0) a = 0xFFFFFFFFu; // inactive lanes
      if (mode == 6) a = (rnd() % 3) * 64 + (l % 5);     // same bank, different address
      idx[w * 64 + l] = a;
    }
  }
  uint32_t *d_idx, *d_ret, *d_fin;
  hipMalloc(&d_idx, n * 4); hipMalloc(&d_ret, n * 4); hipMalloc(&d_fin, fin.size() * 4);
  hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * tsize * 4);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 16 * tsize * 4, 0, d_idx, d_ret, d_fin, iters, tsize);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
  hipMemcpy(ret.data(), d_ret, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(fin.data(), d_fin, fin.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0, dupl = 0;
  std::vector<uint32_t> t(tsize);
  for (size_t wv = 0; wv < (size_t)blocks * 16; wv++) {
    std::fill(t.begin(), t.end(), 0u);
    for (int it = 0; it < iters; it++)
      for (int l = 0; l < 64; l++) {
        const size_t e = (wv * iters + it) * 64 + l;
        if (idx[e] == 0xFFFFFFFFu) continue;
        const uint32_t exp = t[idx[e]];
        if (exp > (uint32_t)it * 64) dupl++;
        if (ret[e] != exp) { if (bad < 10) printf("mismatch wave %zu it %d lane %d idx %u: got %u expected %u\n", wv, it, l, idx[e], ret[e], exp); bad++; }
        t[idx[e]] = it * 64 + l + 1;
      }
    for (int k = 0; k < tsize; k++) if (fin[wv * tsize + k] != t[k]) bad++;
  }
  printf("lds_xchg_order: %zu exchanges, %zu with an earlier lane of the same instruction on the same address, %zu mismatches -> %s\n",
         n, dupl, bad, bad ? "NOT lane ordered" : "lane ordered");
  return bad ? 1 : 0;
}
