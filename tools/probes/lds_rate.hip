// Probe: cost of one 64-lane LDS exchange with return on a 128 KiB table (what K1's
// consumer wave does per 64 keys) against plain read + write + read-back.
//   hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip && ./lds_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int TBL = 32768;

template <int MODE, int U>
__global__ __launch_bounds__(1024) void probe(uint32_t *out, uint64_t *ticks, int iters, int busy_waves, int runlen) {
  extern __shared__ uint32_t table[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < TBL; k += blockDim.x) table[k] = k;
  __syncthreads();
  if (wave != 0) { // optional VALU/LDS noise from the other waves
    if ((int)wave <= busy_waves) {
      uint32_t acc = tid;
      for (int it = 0; it < iters * U; it++) {
        acc = acc * 1664525u + 1013904223u;
        acc ^= table[(acc >> 12) & (TBL - 1)];
      }
      if (acc == 0x12345) out[tid] = acc;
    }
    return;
  }
  // lanes in groups of `runlen` share one table entry (runlen 1: all different)
  uint32_t s = (lane / (uint32_t)runlen) * 2654435761u + blockIdx.x, acc = 0;
  const uint64_t t0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    uint32_t idx[U], val[U], old[U], chk[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      s = s * 1664525u + 1013904223u;
      idx[u] = (s >> 13) & (TBL - 1);
      val[u] = s | 1u;
    }
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < U; u++)
        old[u] = __hip_atomic_exchange(&table[idx[u]], val[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int u = 0; u < U; u++) acc += old[u];
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        old[u] = __hip_atomic_load(&table[idx[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&table[idx[u]], val[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (MODE == 1) chk[u] = __hip_atomic_load(&table[idx[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        acc += old[u];
        if (MODE == 1) acc += __builtin_popcountll(__ballot(chk[u] != val[u]));
      }
    }
  }
  const uint64_t t1 = wall_clock64();
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE, int U>
void run(const char *name, int busy, int runlen = 1) {
  const int blocks = 256, iters = 4096;
  uint32_t *d_out; uint64_t *d_t;
  hipMalloc(&d_out, blocks * 1024 * 4); hipMalloc(&d_t, blocks * 8);
  hipFuncSetAttribute((const void *)probe<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, TBL * 4);
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL((probe<MODE, U>), dim3(blocks), dim3(1024), TBL * 4, 0, d_out, d_t, iters, busy, runlen);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  uint64_t t[256]; hipMemcpy(t, d_t, sizeof(t), hipMemcpyDeviceToHost);
  double sum = 0; for (int b = 0; b < blocks; b++) sum += (double)t[b];
  const double ns = sum / blocks * 10.0 / ((double)iters * U); // wall_clock64: 100 MHz
  printf("%-22s U=%d busy=%2d same-address run %2d  %7.1f ns per 64-lane row  (~%5.0f cycles at 2.4 GHz)\n", name, U, busy, runlen, ns, ns * 2.4);
  hipFree(d_out); hipFree(d_t);
}

int main() {
  for (int busy : {0, 15}) {
    run<0, 1>("xchg_rtn", busy);
    run<0, 5>("xchg_rtn", busy);
    run<0, 10>("xchg_rtn", busy);
    run<1, 5>("read+write+readback", busy);
    run<1, 10>("read+write+readback", busy);
    run<2, 5>("read+write", busy);
  }
  for (int runlen : {2, 4, 16, 64}) run<0, 5>("xchg_rtn", 0, runlen);
  return 0;
}
