// bargain_host.h -- host side of the m4g_bargain_* entry points (included by
// m4g.hip: same translation unit, same helpers).  Kernels: bargain.h.
#pragma once

namespace {

constexpr int kKbWavesPerCu = 32; // 4 KiB of LDS and <= 64 VGPRs per wave (__launch_bounds__(64, 8))

struct BargainDesc {
  int nt = 0;
  int hash_len[KB_MAX_TABLES] = {0, 0, 0};
  KbParams P{};
};

// bargain1.go:10-13, 210-212; bargain2.go:10-13, 242-248; writer.go:142-150
int bargain_desc(const m4g_bargain_params *in, BargainDesc *d) {
  if (!in) return fail(M4G_ERR_BAD_PARAM, "NULL params");
  BargainDesc D;
  D.P.max_distance = in->max_distance ? (uint32_t)in->max_distance : 1u << 16; // bargain1.go:39-41
  D.P.skip = in->skip ? 1u : 0u;
  D.P.block_size = in->block_size ? (uint32_t)in->block_size : 1u << 16; // writer.go:156
  if (in->max_distance < 0 || in->block_size < 0) return fail(M4G_ERR_BAD_PARAM, "negative field");
  if (D.P.block_size < (uint32_t)KX_SUB + 11 || D.P.block_size > (1u << 24))
    return fail(M4G_ERR_BAD_PARAM, "block_size must be 3851 .. 16 MiB");
  auto table = [](KbTable &t, uint64_t prime, int shift, int hash_len, int bits) {
    const uint64_t mul = prime << shift, mask = hash_len >= 8 ? ~0ull : ((1ull << (8 * hash_len)) - 1);
    t.mul_lo = (uint32_t)mul, t.mul_hi = (uint32_t)(mul >> 32);
    t.mask_lo = (uint32_t)mask, t.mask_hi = (uint32_t)(mask >> 32);
    t.bits = (uint32_t)bits;
  };
  D.P.tail = 8;
  if (in->variant == 1) {
    D.nt = 1, D.P.overlap = 4;
    D.hash_len[0] = 6;
    table(D.P.t[0], 227718039650203ull, 16, 6, 18);
  } else if (in->variant == 2) {
    D.nt = 2, D.P.overlap = 6;
    D.hash_len[0] = 5, D.hash_len[1] = 8;
    table(D.P.t[0], 889523592379ull, 24, 5, 18);
    table(D.P.t[1], 0xcf1bbcdcb7a56463ull, 0, 8, 18);
  } else if (in->variant == 3) { // bargain3.go:20-22, 153, 268-278
    D.nt = 3, D.P.overlap = 6, D.P.tail = 12;
    D.hash_len[0] = 5, D.hash_len[1] = 8, D.hash_len[2] = 8;
    table(D.P.t[0], 889523592379ull, 24, 5, 17);
    table(D.P.t[1], 0xcf1bbcdcb7a56463ull, 0, 8, 18);
    table(D.P.t[2], 0xcf1bbcdcb7a56463ull, 0, 8, 19);
    D.P.t[2].mul2 = 2654435761u;
  } else {
    return fail(M4G_ERR_BAD_PARAM, "variant must be 1, 2 or 3 (Bargain1 / Bargain2 / Bargain3)");
  }
  *d = D;
  return M4G_OK;
}

} // namespace

struct m4g_bargain {
  BargainDesc d;
  int device = 0;
  size_t max_chunks = 0, max_total = 0, max_chunk_bytes = 0, max_blocks = 0;
  size_t chain_words = 0; // per table
  uint32_t nwaves = 0, ws_stride = 0, stage_cap = 0;
  Job *d_jobs = nullptr, *h_jobs = nullptr;
  Job *d_k1jobs = nullptr, *h_k1jobs = nullptr; // K1 works on segments of long streams (table priming)
  size_t max_k1jobs = 0, seg_bytes = 0;
  BBlock *d_blocks = nullptr, *h_blocks = nullptr;
  uint32_t *d_chain = nullptr, *d_counts = nullptr;
  float *d_bc = nullptr, *d_wc = nullptr;
  uint2 *d_wld = nullptr;
  m4g_match *d_stage = nullptr;
  uint64_t *d_off = nullptr, *h_total = nullptr;
  hipEvent_t ev[5] = {};
  float ms[4] = {0, 0, 0, 0};
  size_t last_blocks = 0;
};

extern "C" void m4g_bargain_destroy(m4g_bargain *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  (void)hipFree(g->d_jobs);
  (void)hipFree(g->d_k1jobs);
  (void)hipHostFree(g->h_k1jobs);
  (void)hipFree(g->d_blocks);
  (void)hipFree(g->d_chain);
  (void)hipFree(g->d_counts);
  (void)hipFree(g->d_bc);
  (void)hipFree(g->d_wc);
  (void)hipFree(g->d_wld);
  (void)hipFree(g->d_stage);
  (void)hipFree(g->d_off);
  (void)hipHostFree(g->h_jobs);
  (void)hipHostFree(g->h_blocks);
  (void)hipHostFree(g->h_total);
  for (auto &e : g->ev)
    if (e) (void)hipEventDestroy(e);
  delete g;
}

extern "C" int m4g_bargain_create(const m4g_bargain_params *params, int device, size_t max_chunks, size_t max_total_bytes,
                                  size_t max_chunk_bytes, m4g_bargain **out) {
  if (!out) return fail(M4G_ERR_BAD_PARAM, "NULL out");
  *out = nullptr;
  BargainDesc d;
  if (int rc = bargain_desc(params, &d)) return rc;
  if (!max_chunks || !max_total_bytes || !max_chunk_bytes) return fail(M4G_ERR_BAD_PARAM, "zero capacity");
  if (max_chunk_bytes >= (1ull << 31)) return fail(M4G_ERR_TOO_LARGE, "chunks of 2 GiB or more");
  if (d.P.skip && ((d.P.block_size | d.P.max_distance) & 1) && max_chunk_bytes > 2ull * d.P.max_distance)
    // bargain1.go:159 tests the parity of the HISTORY position, which a slide of odd length flips: the first slide
    // drops k * block_size - MaxDistance bytes, every later one a multiple of block_size (bargain1.go:66-70)
    return fail(M4G_ERR_BAD_PARAM, "Skip with an odd block_size or MaxDistance needs chunks of at most 2 * max_distance bytes");
  if (int rc = check_device(device)) return rc;
  HIP_TRY(hipSetDevice(device));
  if (int rc = k1_prepare()) return rc;
  if (k1_use_sort()) return fail(M4G_ERR_NO_DEVICE, "the Bargain path needs the exchange form of K1");
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1x_chain_build<false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)k1x_lds_bytes(kMaxPartBits)));
  m4g_bargain *g = new (std::nothrow) m4g_bargain();
  if (!g) return fail(M4G_ERR_NO_MEMORY, "host allocation failed");
  g->d = d;
  g->device = device;
  g->max_chunks = max_chunks, g->max_total = max_total_bytes, g->max_chunk_bytes = max_chunk_bytes;
  g->max_blocks = max_total_bytes / d.P.block_size + max_chunks;
  g->chain_words = max_total_bytes + 16 * max_chunks;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    m4g_bargain_destroy(g);
    return fail(M4G_ERR_HIP, "hipGetDeviceProperties failed");
  }
  // the grid is what the GPU holds at once: a workgroup that had to wait for a slot would run its blocks alone at the end
  int per_cu = 0;
  const void *fn = d.nt == 1 ? reinterpret_cast<const void *>(kb_parse<1>)
                             : d.nt == 2 ? reinterpret_cast<const void *>(kb_parse<2>) : reinterpret_cast<const void *>(kb_parse<3>);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    per_cu = kKbWavesPerCu;
  }
  g->nwaves = (uint32_t)std::min<size_t>(g->max_blocks, (size_t)prop.multiProcessorCount * std::min(per_cu, kKbWavesPerCu));
  g->ws_stride = (uint32_t)((std::min<size_t>(d.P.block_size, max_chunk_bytes) + 1 + 127) & ~(size_t)63);
  g->stage_cap = d.P.block_size / 3 + 2;
  bool ok = true;
  auto dev = [&](auto **p, size_t bytes) { ok = ok && hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(bytes, 256)) == hipSuccess; };
  // a stream longer than 1.5 segments is cut into segments of whole blocks for K1; each segment first inserts the
  // max_distance bytes before it (a predecessor further back is rejected by bargain1.go:166 whether found or not)
  g->seg_bytes = std::max<size_t>(4 * (size_t)d.P.max_distance, (size_t)4 << 20);
  g->seg_bytes = (g->seg_bytes + d.P.block_size - 1) / d.P.block_size * d.P.block_size;
  g->max_k1jobs = max_chunks + max_total_bytes / g->seg_bytes + 1;
  dev(&g->d_jobs, max_chunks * sizeof(Job));
  dev(&g->d_k1jobs, g->max_k1jobs * sizeof(Job));
  dev(&g->d_blocks, g->max_blocks * sizeof(BBlock));
  dev(&g->d_chain, g->chain_words * 4 * (size_t)d.nt);
  dev(&g->d_counts, (g->max_blocks + 1) * 4); // (+1: the hand-out counter of the M4G_KB_HANDOUT variant)
  dev(&g->d_bc, g->max_blocks * 256 * sizeof(float));
  dev(&g->d_wc, (size_t)g->nwaves * g->ws_stride * sizeof(float));
  dev(&g->d_wld, (size_t)g->nwaves * g->ws_stride * sizeof(uint2));
  dev(&g->d_stage, g->max_blocks * (size_t)g->stage_cap * sizeof(m4g_match));
  dev(&g->d_off, (g->max_blocks + 1) * 8);
  ok = ok && hipHostMalloc(reinterpret_cast<void **>(&g->h_jobs), max_chunks * sizeof(Job)) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void **>(&g->h_k1jobs), g->max_k1jobs * sizeof(Job)) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void **>(&g->h_blocks), g->max_blocks * sizeof(BBlock)) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void **>(&g->h_total), 64) == hipSuccess;
  for (auto &e : g->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    m4g_bargain_destroy(g);
    return fail(M4G_ERR_NO_MEMORY, "device allocation failed (m4g_bargain_create)");
  }
  *out = g;
  return M4G_OK;
}

extern "C" size_t m4g_bargain_block_count(const m4g_bargain *g, const uint64_t *chunk_off, size_t nchunks) {
  if (!g || !chunk_off) return 0;
  size_t nb = 0;
  for (size_t c = 0; c < nchunks; c++) {
    const uint64_t n = chunk_off[c + 1] - chunk_off[c];
    nb += (size_t)((n + g->d.P.block_size - 1) / g->d.P.block_size);
  }
  return nb;
}

namespace {
int64_t bargain_run_impl(m4g_bargain *g, const uint8_t *d_src, const uint64_t *chunk_off, size_t nchunks, m4g_match *d_out,
                         size_t out_cap, uint64_t *d_block_off, void *hip_stream, uint32_t first_pos);
}
extern "C" int64_t m4g_bargain_run_device(m4g_bargain *g, const uint8_t *d_src, const uint64_t *chunk_off, size_t nchunks,
                                          m4g_match *d_out, size_t out_cap, uint64_t *d_block_off, void *hip_stream) {
  return bargain_run_impl(g, d_src, chunk_off, nchunks, d_out, out_cap, d_block_off, hip_stream, 0);
}
namespace {
// first_pos (one chunk only, a multiple of block_size): the chunk's first first_pos bytes are history of earlier
// calls -- inserted into the tables, not searched; the blocks start there.
int64_t bargain_run_impl(m4g_bargain *g, const uint8_t *d_src, const uint64_t *chunk_off, size_t nchunks, m4g_match *d_out,
                         size_t out_cap, uint64_t *d_block_off, void *hip_stream, uint32_t first_pos) {
  if (!g || !d_src || !chunk_off || (!d_out && out_cap) || !d_block_off) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (first_pos && (nchunks != 1 || first_pos % g->d.P.block_size)) return fail(M4G_ERR_INTERNAL, "first_pos");
  if (nchunks > g->max_chunks) return fail(M4G_ERR_TOO_LARGE, "more chunks than the plan holds");
  HIP_TRY(hipSetDevice(g->device));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (nchunks == 0) {
    HIP_TRY(hipMemsetAsync(d_block_off, 0, 8, st));
    HIP_TRY(hipStreamSynchronize(st));
    g->last_blocks = 0;
    return 0;
  }
  const uint64_t total = chunk_off[nchunks] - chunk_off[0];
  if (total > g->max_total) return fail(M4G_ERR_TOO_LARGE, "more bytes than the plan holds");
  if (int rc = fill_batch_jobs(g->h_jobs, chunk_off, nchunks, g->max_chunk_bytes)) return rc;
  const uint32_t bs = g->d.P.block_size;
  size_t nb = 0;
  uint32_t max_len = 0;
  for (size_t c = 0; c < nchunks; c++) {
    const uint32_t n = g->h_jobs[c].len;
    max_len = std::max(max_len, n);
    for (uint32_t p = first_pos; p < n; p += bs) {
      BBlock B{};
      B.job = (uint32_t)c, B.start = p, B.len = std::min(bs, n - p);
      g->h_blocks[nb++] = B;
    }
  }
  g->last_blocks = nb;
  if (nb == 0) {
    HIP_TRY(hipMemsetAsync(d_block_off, 0, 8, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
  }
  size_t nk1 = 0;
  for (size_t c = 0; c < nchunks; c++) {
    const Job &J = g->h_jobs[c];
    if ((size_t)J.len - first_pos <= g->seg_bytes + g->seg_bytes / 2) {
      Job S = J;
      S.store_from = first_pos;
      g->h_k1jobs[nk1++] = S;
      continue;
    }
    for (size_t p = first_pos; p < J.len; p += g->seg_bytes) { // segment boundaries are block boundaries
      Job S = J;
      S.start = (uint32_t)(p > g->d.P.max_distance ? p - g->d.P.max_distance : 0);
      S.store_from = (uint32_t)p;
      S.len = (uint32_t)std::min<size_t>(J.len, p + g->seg_bytes);
      if (nk1 >= g->max_k1jobs) return fail(M4G_ERR_INTERNAL, "segment list overflow");
      g->h_k1jobs[nk1++] = S;
    }
  }
  HIP_TRY(hipMemcpyAsync(g->d_jobs, g->h_jobs, nchunks * sizeof(Job), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(g->d_k1jobs, g->h_k1jobs, nk1 * sizeof(Job), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(g->d_blocks, g->h_blocks, nb * sizeof(BBlock), hipMemcpyHostToDevice, st));
  const uint64_t src_total = chunk_off[nchunks];
  HIP_TRY(hipEventRecord(g->ev[0], st));
  // K1 per table: chain entry = distance to the latest earlier position with the same hash
  int posbits = 1;
  while (posbits < 32 && (max_len >> posbits) != 0) posbits++;
  const int cb = std::max(0, std::min(K1_MAX_CHECK_BITS, 32 - posbits));
  if (cb < 1) return fail(M4G_ERR_TOO_LARGE, "buffer positions need all 32 bits");
  const int njobs = (int)nk1;
  for (int t = 0; t < g->d.nt; t++) {
    const KbTable &T = g->d.P.t[t];
    const int pb = std::min((int)T.bits, kMaxPartBits), nparts = 1 << ((int)T.bits - pb);
    g->d.P.t[t].chain_base_off = (uint64_t)t * g->chain_words;
    hipLaunchKernelGGL((k1x_chain_build<false, true>), dim3((unsigned)(((njobs + 7) / 8) * 8 * nparts)), dim3(KX_THREADS),
                       k1x_lds_bytes(pb), st, d_src, src_total, g->d_chain + (size_t)t * g->chain_words, g->d_k1jobs,
                       (uint32_t *)nullptr, njobs, g->d.hash_len[t], (int)T.bits, pb, cb, 0, (const StreamInfo *)nullptr,
                       (uint32_t *)nullptr, T.mul_lo, T.mul_hi, bs, T.mul2, g->d.P.tail - 1);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(g->ev[1], st));
  hipLaunchKernelGGL(kb_byte_cost, dim3((unsigned)nb), dim3(256), 0, st, d_src, g->d_jobs, g->d_blocks, (uint32_t)nb, g->d_bc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(g->ev[2], st));
  const unsigned grid = (unsigned)std::min<size_t>(nb, g->nwaves);
#ifdef M4G_KB_HANDOUT
  HIP_TRY(hipMemsetAsync(g->d_counts + nb, 0, 4, st)); // the hand-out counter, on the kernels' stream
#endif
  if (g->d.nt == 1)
    hipLaunchKernelGGL(kb_parse<1>, dim3(grid), dim3(64), 0, st, d_src, src_total, g->d_chain, g->d_jobs, g->d_blocks,
                       (uint32_t)nb, g->d_bc, g->d_wld, g->d_wc, g->ws_stride, g->d_stage, g->stage_cap, g->d_counts, g->d.P);
  else if (g->d.nt == 2)
    hipLaunchKernelGGL(kb_parse<2>, dim3(grid), dim3(64), 0, st, d_src, src_total, g->d_chain, g->d_jobs, g->d_blocks,
                       (uint32_t)nb, g->d_bc, g->d_wld, g->d_wc, g->ws_stride, g->d_stage, g->stage_cap, g->d_counts, g->d.P);
  else
    hipLaunchKernelGGL(kb_parse<3>, dim3(grid), dim3(64), 0, st, d_src, src_total, g->d_chain, g->d_jobs, g->d_blocks,
                       (uint32_t)nb, g->d_bc, g->d_wld, g->d_wc, g->ws_stride, g->d_stage, g->stage_cap, g->d_counts, g->d.P);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(g->ev[3], st));
  hipLaunchKernelGGL(k4_scan_counts, dim3(1), dim3(1024), 0, st, g->d_counts, g->d_off, (int)nb, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(g->h_total, g->d_off + nb, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const uint64_t nm = *g->h_total;
  HIP_TRY(hipMemcpyAsync(d_block_off, g->d_off, (nb + 1) * 8, hipMemcpyDeviceToDevice, st));
  if (nm <= out_cap && nm > 0) {
    hipLaunchKernelGGL(kb_compact, dim3((unsigned)nb), dim3(256), 0, st, g->d_stage, g->stage_cap, g->d_counts, g->d_off,
                       (uint32_t)nb, d_out);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(g->ev[4], st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int k = 0; k < 4; k++) HIP_TRY(hipEventElapsedTime(&g->ms[k], g->ev[k], g->ev[k + 1]));
  if (nm > out_cap) return -(int64_t)(nm + M4G_STREAM_NEEDED_BIAS);
  return (int64_t)nm;
}
} // namespace

extern "C" int m4g_bargain_last_timings(const m4g_bargain *g, float *ms, int cap) {
  if (!g || !ms) return 0;
  const int n = std::min(cap, 4);
  for (int k = 0; k < n; k++) ms[k] = g->ms[k];
  return n;
}

// One stream in host memory, cut into blocks as matchfinder.Writer does (matchfinder.go:59-93):
// the Match lists of its blocks, block_off[k] = index of block k's first Match.
extern "C" int64_t m4g_bargain_find_blocks(const m4g_bargain_params *params, int device, const uint8_t *src, size_t n,
                                           m4g_match *out, size_t out_cap, uint64_t *block_off) {
  if ((!src && n) || (!out && out_cap) || !block_off) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  BargainDesc d;
  if (int rc = bargain_desc(params, &d)) return rc;
  if (n == 0) {
    block_off[0] = 0;
    return 0;
  }
  m4g_bargain *g = nullptr;
  if (int rc = m4g_bargain_create(params, device, 1, n, n, &g)) return rc;
  const size_t nb = (n + d.P.block_size - 1) / d.P.block_size;
  const size_t cap = n / 3 + 2 * nb + 8; // a Match is at least 3 bytes long, plus one literal-only record per block
  uint8_t *d_src = nullptr;
  m4g_match *d_out = nullptr;
  uint64_t *d_off = nullptr;
  int64_t r = M4G_OK;
  auto cleanup = [&]() {
    (void)hipFree(d_src);
    (void)hipFree(d_out);
    (void)hipFree(d_off);
    m4g_bargain_destroy(g);
  };
  if (hipMalloc(reinterpret_cast<void **>(&d_src), n + 64) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&d_out), cap * sizeof(m4g_match)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&d_off), (nb + 1) * 8) != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    return fail(M4G_ERR_NO_MEMORY, "device allocation failed (m4g_bargain_find_blocks)");
  }
  const uint64_t chunk_off[2] = {0, n};
  static const bool timing = std::getenv("M4G_V2_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  if (hipMemcpy(d_src, src, n, hipMemcpyHostToDevice) != hipSuccess) r = fail(M4G_ERR_HIP, "upload failed");
  const auto t1 = std::chrono::steady_clock::now();
  if (r == M4G_OK) r = m4g_bargain_run_device(g, d_src, chunk_off, 1, d_out, cap, d_off, nullptr);
  if (timing)
    std::fprintf(stderr, "m4g_bargain_find_blocks: upload %.1f ms, run %.1f ms (links %.1f costs %.1f search %.1f compact %.1f)\n",
                 std::chrono::duration<double, std::milli>(t1 - t0).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(), g->ms[0], g->ms[1],
                 g->ms[2], g->ms[3]);
  if (r >= 0) {
    if ((uint64_t)r > out_cap) {
      r = -(int64_t)((uint64_t)r + M4G_STREAM_NEEDED_BIAS);
    } else if (hipMemcpy(block_off, d_off, (nb + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess ||
               (r > 0 && hipMemcpy(out, d_out, (size_t)r * sizeof(m4g_match), hipMemcpyDeviceToHost) != hipSuccess)) {
      r = fail(M4G_ERR_HIP, "download failed");
    }
  }
  cleanup();
  return r;
}

namespace {
template <class T> int grow_dev(T **p, size_t *cap, size_t need) {
  if (need <= *cap) return M4G_OK;
  (void)hipFree(*p);
  *p = nullptr, *cap = 0;
  const size_t want = need + need / 2 + 64;
  if (hipMalloc(reinterpret_cast<void **>(p), want * sizeof(T)) != hipSuccess) {
    (void)hipGetLastError();
    return fail(M4G_ERR_NO_MEMORY, "device allocation failed (Bargain handle)");
  }
  *cap = want;
  return M4G_OK;
}
} // namespace

// ---------------------------------------------------------------------------
// Bargain1 / Bargain2 / Bargain3 as a matchfinder.MatchFinder: a handle with FindMatches / Reset
// (bargain1.go:33-38, 66-84; writer.go:142-150 plugs it into matchfinder.Writer).  include/m4g.h.
// ---------------------------------------------------------------------------
struct m4g_bargain_ctx {
  m4g_bargain_params bp{};
  BargainDesc d;
  int device = 0;
  std::vector<uint8_t> hist;  // the stream's last whole blocks, MaxDistance bytes or more (host copy)
  std::vector<uint8_t> chunk; // [hist | new blocks], staged for the upload
  bool tail_seen = false;     // a block shorter than block_size was found: the stream's last one
  m4g_bargain *plan = nullptr;
  size_t plan_bytes = 0;
  uint8_t *d_src = nullptr;
  m4g_match *d_m = nullptr;
  uint64_t *d_off = nullptr;
  size_t d_src_cap = 0, d_m_cap = 0, d_off_cap = 0;
};

extern "C" void m4g_bargain_ctx_destroy(m4g_bargain_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  m4g_bargain_destroy(c->plan);
  (void)hipFree(c->d_src);
  (void)hipFree(c->d_m);
  (void)hipFree(c->d_off);
  delete c;
}

extern "C" int m4g_bargain_ctx_create(const m4g_bargain_params *params, int device, m4g_bargain_ctx **out) {
  if (!out) return fail(M4G_ERR_BAD_PARAM, "out is NULL");
  *out = nullptr;
  BargainDesc d;
  if (int rc = bargain_desc(params, &d)) return rc;
  if (int rc = check_device(device)) return rc;
  if (d.P.skip && ((d.P.block_size | d.P.max_distance) & 1))
    // bargain1.go:159 tests the parity of the HISTORY position; the handle cuts the history at block boundaries
    return fail(M4G_ERR_BAD_PARAM, "Skip needs an even block_size and MaxDistance on this handle");
  m4g_bargain_ctx *c = new (std::nothrow) m4g_bargain_ctx();
  if (!c) return fail(M4G_ERR_NO_MEMORY, "host allocation failed");
  c->bp = *params;
  c->d = d;
  c->device = device;
  *out = c;
  return M4G_OK;
}

// bargain1.go:33-36
extern "C" int m4g_bargain_ctx_reset(m4g_bargain_ctx *c) {
  if (!c) return fail(M4G_ERR_BAD_PARAM, "ctx is NULL");
  c->hist.clear();
  c->tail_seen = false;
  return M4G_OK;
}

// FindMatches on every consecutive block of block_size bytes of src, continuing the handle's stream
// (n = one block: FindMatches itself).  block_off: nullable.
extern "C" int64_t m4g_bargain_ctx_find_blocks(m4g_bargain_ctx *c, const uint8_t *src, size_t n, m4g_match *out,
                                               size_t out_cap, uint64_t *block_off) {
  if (!c || (!src && n) || (!out && out_cap)) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (block_off) block_off[0] = 0;
  if (n == 0) return 0; // bargain1.go:38-64 on an empty block: nothing inserted, nothing emitted
  const size_t bs = c->d.P.block_size;
  if (c->tail_seen)
    return fail(M4G_ERR_BAD_PARAM, "m4g_bargain_ctx: a block shorter than block_size ends the stream (matchfinder.Writer's "
                                   "shape); m4g_bargain_ctx_reset starts the next one");
  const size_t hist = c->hist.size(), total = hist + n, nb = (n + bs - 1) / bs;
  if (total >= ((uint64_t)1 << 31)) return fail(M4G_ERR_TOO_LARGE, "at most 2 GiB of history + data per call");
  const size_t need = n / 3 + 2 * nb + 8; // a Match is at least 3 bytes long, plus one literal-only record per block
  if (out_cap < need) {
    g_last_error = "out_cap below n / 3 + 2 * blocks + 8";
    return -(int64_t)(need + M4G_STREAM_NEEDED_BIAS);
  }
  HIP_TRY(hipSetDevice(c->device));
  c->chunk.resize(total);
  if (hist) std::memcpy(c->chunk.data(), c->hist.data(), hist);
  std::memcpy(c->chunk.data() + hist, src, n);
  if (int rc = grow_dev(&c->d_src, &c->d_src_cap, total + 64)) return rc;
  if (int rc = grow_dev(&c->d_m, &c->d_m_cap, need)) return rc;
  if (int rc = grow_dev(&c->d_off, &c->d_off_cap, nb + 2)) return rc;
  if (!c->plan || total > c->plan_bytes) {
    m4g_bargain_destroy(c->plan);
    c->plan = nullptr;
    const size_t cap = std::max<size_t>(total + total / 2, (size_t)4 << 20);
    if (int rc = m4g_bargain_create(&c->bp, c->device, 1, cap, cap, &c->plan)) return rc;
    c->plan_bytes = cap;
  }
  HIP_TRY(hipMemcpy(c->d_src, c->chunk.data(), total, hipMemcpyHostToDevice));
  const uint64_t chunk_off[2] = {0, total};
  const int64_t r = bargain_run_impl(c->plan, c->d_src, chunk_off, 1, c->d_m, c->d_m_cap, c->d_off, nullptr, (uint32_t)hist);
  if (r < 0) return r < -(int64_t)M4G_STREAM_NEEDED_BIAS ? fail(M4G_ERR_INTERNAL, "Match list larger than its bound") : r;
  if (r) HIP_TRY(hipMemcpy(out, c->d_m, (size_t)r * sizeof(m4g_match), hipMemcpyDeviceToHost));
  if (block_off) HIP_TRY(hipMemcpy(block_off, c->d_off, (nb + 1) * 8, hipMemcpyDeviceToHost));
  // what the next call needs of the stream: its last whole blocks, MaxDistance bytes or more
  if (n % bs) {
    c->tail_seen = true;
  } else {
    const size_t keep = ((size_t)c->d.P.max_distance + bs - 1) / bs * bs;
    if (total > keep) c->hist.assign(c->chunk.begin() + (ptrdiff_t)(total - keep), c->chunk.end());
    else c->hist = c->chunk;
  }
  return r;
}

// One stream over several devices: the Match lists of m4g_bargain_find_blocks, every device with a contiguous range
// of blocks and the whole blocks (MaxDistance bytes or more) before it as history -- the same cut the handle above
// makes between calls.
extern "C" int64_t m4g_bargain_find_blocks_multi(const m4g_bargain_params *params, const int *devices, int ndevices,
                                                 const uint8_t *src, size_t n, m4g_match *out, size_t out_cap,
                                                 uint64_t *block_off) {
  if (!devices || ndevices <= 0 || (!src && n) || (!out && out_cap) || !block_off)
    return fail(M4G_ERR_BAD_PARAM, "NULL argument or no device");
  BargainDesc d;
  if (int rc = bargain_desc(params, &d)) return rc;
  block_off[0] = 0;
  if (n == 0) return 0;
  const size_t bs = d.P.block_size, nblocks = (n + bs - 1) / bs;
  const size_t keep = ((size_t)d.P.max_distance + bs - 1) / bs * bs;
  const size_t per = (nblocks + (size_t)ndevices - 1) / (size_t)ndevices;
  struct Part {
    std::vector<m4g_match> m;
    std::vector<uint64_t> off;
    int64_t r = 0;
    std::string err;
  };
  std::vector<Part> parts((size_t)ndevices);
  auto work = [&](int dv) {
    Part &P = parts[(size_t)dv];
    const size_t b0 = std::min(nblocks, (size_t)dv * per), b1 = std::min(nblocks, b0 + per);
    if (b0 >= b1) return;
    const size_t x0 = b0 * bs, x1 = std::min(n, b1 * bs), lo = x0 > keep ? x0 - keep : 0;
    m4g_bargain_ctx *c = nullptr;
    int64_t rc = m4g_bargain_ctx_create(params, devices[dv], &c);
    if (rc == 0) {
      c->hist.assign(src + lo, src + x0); // (whole blocks: lo and x0 are block boundaries)
      const size_t nb = b1 - b0, len = x1 - x0;
      P.m.resize(len / 3 + 2 * nb + 8);
      P.off.resize(nb + 1);
      rc = m4g_bargain_ctx_find_blocks(c, src + x0, len, P.m.data(), P.m.size(), P.off.data());
    }
    if (rc < 0) P.err = g_last_error;
    P.r = rc;
    m4g_bargain_ctx_destroy(c);
  };
  {
    std::vector<std::thread> th;
    for (int dv = 1; dv < ndevices; dv++) th.emplace_back(work, dv);
    work(0);
    for (auto &t : th) t.join();
  }
  uint64_t total = 0;
  for (int dv = 0; dv < ndevices; dv++) {
    Part &P = parts[(size_t)dv];
    if (P.r < 0) {
      g_last_error = P.err;
      return P.r;
    }
    const size_t b0 = std::min(nblocks, (size_t)dv * per), b1 = std::min(nblocks, b0 + per);
    if (b0 >= b1) continue;
    if (total + (uint64_t)P.r > out_cap) return -(int64_t)(n / 3 + 2 * nblocks + 8 + M4G_STREAM_NEEDED_BIAS);
    if (P.r) std::memcpy(out + total, P.m.data(), (size_t)P.r * sizeof(m4g_match));
    for (size_t b = 1; b <= b1 - b0; b++) block_off[b0 + b] = total + P.off[b];
    total += (uint64_t)P.r;
  }
  return (int64_t)total;
}

// w := brotli.NewWriterV2(dst, level); w.Write(src); w.Close() for the levels whose MatchFinder is one of the
// Bargain finders (writer.go:126-161): level 5, 6 Bargain1, 7 Bargain2 with Skip, 8 Bargain2, 9 (and above) Bargain3,
// MaxDistance 1 << 20, BlockSize 1 << 16, Encoder &brotli.Encoder{}.  Host memory in, the brotli stream out.
extern "C" int64_t m4g_brotli_compress_v2(int level, int device, const uint8_t *src, size_t n, uint8_t *out, size_t out_cap) {
  if ((!src && n) || (!out && out_cap)) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (level > 9) level = 9; // writer.go:129-131
  if (level < 5)
    return fail(M4G_ERR_BAD_PARAM, "NewWriterV2 levels 0-4 use ZFast / ZDFast / ZM / Trio, which are not built (levels 5-9 are)");
  m4g_bargain_params p{};
  p.variant = level <= 6 ? 1 : (level <= 8 ? 2 : 3);
  p.max_distance = 1 << 20;
  p.skip = level == 7 ? 1 : 0;
  p.block_size = 1 << 16;
  const size_t bs = (size_t)p.block_size, nb = (n + bs - 1) / bs;
  static const bool timing = std::getenv("M4G_V2_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  if (n == 0) { // Close() alone: one empty last block
    const uint64_t off0[1] = {0};
    return m4g_brotli_encode_blocks(device, src, 0, bs, nullptr, off0, out, out_cap);
  }
  if (n >= ((uint64_t)1 << 31)) return fail(M4G_ERR_TOO_LARGE, "m4g_brotli_compress_v2: at most 2 GiB per call");
  // the bytes go up once; the Match lists stay on the device between the finder and the encoder
  struct Dev {
    m4g_bargain *g = nullptr;
    uint8_t *src = nullptr;
    m4g_match *m = nullptr;
    uint64_t *off = nullptr;
    ~Dev() {
      (void)hipFree(src);
      (void)hipFree(m);
      (void)hipFree(off);
      m4g_bargain_destroy(g);
    }
  } D;
  if (int rc = m4g_bargain_create(&p, device, 1, n, n, &D.g)) return rc;
  const size_t mcap = n / 3 + 2 * nb + 8;
  if (hipMalloc(reinterpret_cast<void **>(&D.src), n + 64) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&D.m), mcap * sizeof(m4g_match)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&D.off), (nb + 1) * 8) != hipSuccess) {
    (void)hipGetLastError();
    return fail(M4G_ERR_NO_MEMORY, "device allocation failed (m4g_brotli_compress_v2)");
  }
  HIP_TRY(hipMemcpy(D.src, src, n, hipMemcpyHostToDevice));
  const uint64_t chunk_off[2] = {0, n};
  const int64_t nm = m4g_bargain_run_device(D.g, D.src, chunk_off, 1, D.m, mcap, D.off, nullptr);
  if (nm < 0) return nm < -(int64_t)M4G_STREAM_NEEDED_BIAS ? fail(M4G_ERR_INTERNAL, "Match list larger than its bound") : nm;
  std::vector<uint64_t> off(nb + 1, 0);
  HIP_TRY(hipMemcpy(off.data(), D.off, (nb + 1) * 8, hipMemcpyDeviceToHost));
  const auto t1 = std::chrono::steady_clock::now();
  BlockUpload U;
  if (int rc = U.init(device, nullptr, n, bs, nullptr, off.data(), true, D.src, D.m)) return rc;
  const int64_t r = brotli_encode_uploaded(U, nullptr, n, out, out_cap, "m4g_brotli_compress_v2");
  if (timing) {
    const auto t2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "m4g_brotli_compress_v2: set-up + upload + finder %.1f ms (links %.1f costs %.1f search %.1f compact %.1f), "
                 "encoder %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), D.g->ms[0], D.g->ms[1], D.g->ms[2],
                 D.g->ms[3], std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
  return r;
}

// ---- matchfinder.Writer with a Bargain finder: brotli.NewWriterV2(dst, 5..9) as a streaming handle -------------
namespace {

// The blocks of `span` (whole blocks of bs bytes; final: Close()'s remainder, shorter than a block, possibly empty).
// The finder sees the stream so far through the bytes kept in w->bhist: whole blocks, at least MaxDistance of them once
// the stream is that long -- a table entry further back is rejected by bargain1.go:166 whether it survives the
// reference's slides (bargain1.go:66-80) or not, and the blocks keep their alignment, so K1's rule for the last
// positions of a block (bargain1.go:136) and Skip's parity (bargain1.go:159; block sizes are even) see what the
// reference sees.
int writer_process_bargain(m4g_writer *w, const uint8_t *span, size_t len, size_t bs, bool final) {
  HIP_TRY(hipSetDevice(w->device));
  if (bs == 0 || (!final && len % bs) || (final && len >= bs)) return fail(M4G_ERR_INTERNAL, "m4g_writer (Bargain): block structure");
  const size_t hist = w->bhist.size(), total = hist + len;
  const size_t nlists = (len + bs - 1) / bs;              // blocks the finder sees
  const size_t njobs = final ? std::max<size_t>(nlists, 1) : nlists; // blocks the encoder sees (Close: one, possibly empty)
  if (njobs == 0) return M4G_OK;
  if (total >= ((uint64_t)1 << 31)) return fail(M4G_ERR_TOO_LARGE, "at most 2 GiB of history + data per Write");
  w->bchunk.resize(total);
  if (hist) std::memcpy(w->bchunk.data(), w->bhist.data(), hist);
  if (len) std::memcpy(w->bchunk.data() + hist, span, len);
  if (int rc = grow_dev(&w->d_bsrc, &w->d_bsrc_cap, total + 64)) return rc;
  if (int rc = grow_dev(&w->d_bm, &w->d_bm_cap, len / 3 + 2 * nlists + 8)) return rc;
  if (int rc = grow_dev(&w->d_boff, &w->d_boff_cap, njobs + 2)) return rc;
  if (int rc = grow_dev(&w->d_bjobs, &w->d_bjobs_cap, njobs + 1)) return rc;
  if (total) HIP_TRY(hipMemcpy(w->d_bsrc, w->bchunk.data(), total, hipMemcpyHostToDevice));
  uint64_t nm = 0;
  std::vector<uint64_t> off(njobs + 1, 0);
  if (nlists) {
    if (!w->bplan || total > w->bplan_bytes) {
      m4g_bargain_destroy(w->bplan);
      w->bplan = nullptr;
      const size_t cap = std::max<size_t>(total + total / 2, (size_t)8 << 20);
      if (int rc = m4g_bargain_create(&w->bp, w->device, 1, cap, cap, &w->bplan)) return rc;
      w->bplan_bytes = cap;
    }
    const uint64_t chunk_off[2] = {0, total};
    const int64_t r = bargain_run_impl(w->bplan, w->d_bsrc, chunk_off, 1, w->d_bm, w->d_bm_cap, w->d_boff, nullptr, (uint32_t)hist);
    if (r < 0) return r < -(int64_t)M4G_STREAM_NEEDED_BIAS ? fail(M4G_ERR_INTERNAL, "Match list larger than its bound") : (int)r;
    nm = (uint64_t)r;
    HIP_TRY(hipMemcpy(off.data(), w->d_boff, (nlists + 1) * 8, hipMemcpyDeviceToHost));
  } else {
    HIP_TRY(hipMemcpy(w->d_boff, off.data(), (njobs + 1) * 8, hipMemcpyHostToDevice));
  }
  // the encoder's jobs: block k = bytes [hist + k * bs, ...) of the device copy
  std::vector<Job> jobs(njobs);
  std::vector<uint64_t> lens(njobs);
  for (size_t k = 0; k < njobs; k++) {
    Job J{};
    J.src_off = hist + k * bs;
    J.len = (uint32_t)std::min<size_t>(bs, total - J.src_off);
    jobs[k] = J;
    lens[k] = J.len;
  }
  HIP_TRY(hipMemcpy(w->d_bjobs, jobs.data(), njobs * sizeof(Job), hipMemcpyHostToDevice));
  const size_t run_cap = len / K5_HUGE + njobs + 64;
  if (int rc = writer_reserve_brotli(w, njobs, nm, run_cap)) return rc;
  HIP_TRY(hipMemsetAsync(w->d_nruns, 0, 4, nullptr));
  if (nm) {
    hipLaunchKernelGGL(kb_long_runs, dim3((unsigned)((njobs + 63) / 64)), dim3(64), 0, nullptr, w->d_bm, w->d_boff, (uint32_t)njobs,
                       w->d_runs, w->d_nruns, (uint32_t)(w->runs_cap - 1));
    HIP_TRY(hipGetLastError());
  }
  if (int rc = writer_encode(w, w->d_bsrc, w->d_bjobs, lens, w->d_bm, w->d_boff, nm, nullptr, final)) return rc;
  // what the next call needs of the stream: its last whole blocks, MaxDistance bytes or more
  const size_t md = w->bp.max_distance ? (size_t)w->bp.max_distance : (size_t)1 << 16;
  const size_t keep = (md + bs - 1) / bs * bs;
  if (!final) {
    if (total > keep) w->bhist.assign(w->bchunk.begin() + (total - keep), w->bchunk.end());
    else w->bhist = w->bchunk;
  }
  return M4G_OK;
}
} // namespace

// w := brotli.NewWriterV2(dst, level) for levels 5-9 (writer.go:126-161) as an m4g_writer: m4g_writer_write /
// _close / _reset / _destroy are its Write / Close / Reset.
extern "C" int m4g_writer_create_v2(int level, int device, m4g_writer **out) {
  if (!out) return fail(M4G_ERR_BAD_PARAM, "out is NULL");
  *out = nullptr;
  if (level > 9) level = 9; // writer.go:129-131
  if (level < 5)
    return fail(M4G_ERR_BAD_PARAM, "NewWriterV2 levels 0-4 use ZFast / ZDFast / ZM / Trio, which are not built (levels 5-9 are)");
  if (int rc = check_device(device)) return rc;
  HIP_TRY(hipSetDevice(device));
  m4g_writer *w = new (std::nothrow) m4g_writer();
  if (!w) return fail(M4G_ERR_NO_MEMORY, "host allocation failed");
  w->v2 = true;
  w->bp.variant = level <= 6 ? 1 : (level <= 8 ? 2 : 3);
  w->bp.max_distance = 1 << 20;
  w->bp.skip = level == 7 ? 1 : 0;
  w->bp.block_size = 1 << 16;
  w->device = device;
  w->format = M4G_FORMAT_BROTLI;
  w->bs = (size_t)1 << 16;
  if (hipEventCreate(&w->e0) != hipSuccess || hipEventCreate(&w->e1) != hipSuccess) {
    m4g_writer_destroy(w);
    return fail(M4G_ERR_HIP, "hipEventCreate failed");
  }
  *out = w;
  return M4G_OK;
}

#ifdef KB_STATS
// Debug build only (tools/bargain_stats.sh): event counters of the search kernel since the last call.
extern "C" int m4g_debug_bargain_stats(unsigned long long *out) {
  unsigned long long z[8] = {0};
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(kb_stats), sizeof z));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kb_stats), z, sizeof z));
  return M4G_OK;
}
#endif
