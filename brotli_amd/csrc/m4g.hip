// m4g.hip -- host side of libm4g.so: the C-ABI of include/m4g.h over the
// gfx950 kernels in kernels.h.  No CPU compute path exists in this library.
#include "../../include/m4g.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"
#include "bargain.h"
#include "enc_host.h"
#include "flate_host.h"

#include <thread>

using namespace m4g;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(M4G_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// m4.go:64-75 defaults
struct Params {
  int32_t max_distance, min_length, hash_len, table_bits, chain_length, distance_bit_cost;
};

int resolve_params(const m4g_params *in, Params *p) {
  if (!in) return fail(M4G_ERR_BAD_PARAM, "params is NULL");
  p->max_distance = in->max_distance ? in->max_distance : 65535;
  p->min_length = in->min_length ? in->min_length : 4;
  p->hash_len = in->hash_len ? in->hash_len : 6;
  p->table_bits = in->table_bits ? in->table_bits : 17;
  p->chain_length = in->chain_length;
  p->distance_bit_cost = in->distance_bit_cost;
  if (p->max_distance < 1 || p->min_length < 1 || p->hash_len < 1 || p->hash_len > 8 ||
      p->table_bits < 1 || p->table_bits > 24 || p->chain_length < 0 || p->distance_bit_cost < 0)
    return fail(M4G_ERR_BAD_PARAM, "parameter out of range");
  return M4G_OK;
}

ParseParams parse_params(const Params &p) {
  ParseParams P;
  P.max_distance = p.max_distance;
  P.min_length = p.min_length;
  P.hash_len = p.hash_len;
  P.chain_length = p.chain_length;
  P.distance_bit_cost = p.distance_bit_cost;
  return P;
}

constexpr int kMaxPartBits = 15; // 2^15 * 4 B = 128 KiB of LDS
constexpr int kK1MaxTile = K1_SUB > KX_SUB ? K1_SUB : KX_SUB; // positions per K1 tile, either kernel

int part_bits_of(const Params &p) { return std::min(p.table_bits, kMaxPartBits); }

int check_device(int device) {
  if (device < 0)
    return fail(M4G_ERR_NO_DEVICE,
                "device < 0: libm4g has no CPU backend (the CPU oracle is test infrastructure)");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(M4G_ERR_NO_DEVICE, "no HIP device available");
  if (device >= n) return fail(M4G_ERR_NO_DEVICE, "device index out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return fail(M4G_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(M4G_ERR_NO_DEVICE,
                std::string("device is ") + prop.gcnArchName + ", libm4g is built for gfx950 only");
  return M4G_OK;
}

int k1_selftest(uint32_t *d_word);
// d_word: a device word for the self-test's result; NULL = allocate one for the
// duration (callers that must not allocate pass their own).
int k1_prepare(uint32_t *d_word = nullptr) {
  if (int rc = k1_selftest(d_word)) return rc;
  // > 64 KiB of dynamic LDS needs the attribute once per device.
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1_chain_build<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)k1_lds_bytes(kMaxPartBits)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1_chain_build<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)k1_lds_bytes(kMaxPartBits)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1x_chain_build<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)k1x_lds_bytes(kMaxPartBits)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1x_chain_build<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)k1x_lds_bytes(kMaxPartBits)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1c_consume),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)k1c_lds_bytes(kMaxPartBits)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1c_scatter),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)k1c_lds_bytes(kMaxPartBits)));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1d_depth),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)k1c_lds_bytes(kMaxPartBits)));
  return M4G_OK;
}

// K1 variant: the exchange kernel (k1_xchg.h), unless M4G_K1_SORT is set or the
// device failed the ordering self-test (then the sorting kernel, kernels.h).
std::atomic<int> g_xchg_order[64]; // per device: 0 = not tested, 1 = ordered, 2 = not ordered
std::atomic<int> g_k1_force{-1};   // test hook (m4g_debug_set_k1_variant): -1 = automatic, 0 = sort, 1 = exchange, 2 = split
bool k1_use_sort() {
  const int f = g_k1_force.load();
  if (f >= 0) return f == 0;
  static const bool forced = std::getenv("M4G_K1_SORT") != nullptr;
  if (forced) return true;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  return g_xchg_order[dev].load() == 2;
}
int k1_selftest(uint32_t *d_word) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || g_xchg_order[dev].load() != 0) return M4G_OK;
  uint32_t *d = d_word, h = 0;
  if (!d) HIP_TRY(hipMalloc(&d, 4));
  hipError_t e = hipMemset(d, 0, 4);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(kx_order_selftest, dim3(8), dim3(256), 0, nullptr, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  if (!d_word) (void)hipFree(d);
  HIP_TRY(e);
  g_xchg_order[dev].store(h == 0 ? 1 : 2);
  if (h != 0)
    std::fprintf(stderr, "[m4g] device %d: LDS exchange is not lane ordered (%u deviations); K1 uses the sorting kernel\n",
                 dev, h);
  return M4G_OK;
}

// Device buffers of the speculative parse (kernels.h, K3) for up to `blocks`
// (job, sub-block) pairs.
struct SpecWork {
  SpecLayout L{};
  size_t blocks = 0;
  Rec *rs = nullptr, *st = nullptr;
  ParseState *E = nullptr, *Tout = nullptr;
  BlockDesc *desc = nullptr;
  // Compact plans (spec_alloc's `compact`): RS and ST are two halves of ONE allocation, sized for the record
  // density inputs have in practice instead of the densest a parse can emit; a run that overflows a slot is
  // repeated in the worst-case layout (spec_use_worst), which holds group_jobs jobs at a time.
  bool arena = false;      // st points into the allocation of rs
  size_t arena_recs = 0;   // records that allocation holds
  uint32_t slot_worst = 0; // records a parse can emit per sub-block at most
  size_t group_jobs = 0;   // K3 + K4 run over this many jobs at a time (0: all at once)
  size_t bytes = 0;        // device memory held
  bool compact() const { return arena && L.slot < slot_worst; }
};

void spec_free(SpecWork &w) {
  (void)hipFree(w.rs);
  if (!w.arena) (void)hipFree(w.st);
  (void)hipFree(w.E);
  (void)hipFree(w.Tout);
  (void)hipFree(w.desc);
  w = SpecWork{};
}

// The sticky switch of a compact plan to slots no input can overflow: the same allocation then holds the
// records of group_jobs jobs, and run_batch_pipeline parses and compacts the batch group after group.
void spec_use_worst(SpecWork &w) {
  w.L.slot = w.slot_worst;
  w.group_jobs = std::max<size_t>(1, w.arena_recs / (2 * (size_t)w.L.kmax * w.slot_worst));
  w.st = w.rs + w.group_jobs * w.L.kmax * w.slot_worst;
}

// K2 serves fresh-state jobs built by the split K1, whose positions and distances fit the packed entries.
bool k2_usable(const Params &p, size_t max_chunk_bytes) {
  // Opt-in (M4G_K2=1, read at every plan creation): on text K2 evaluates 6 times the candidates the parse would
  // visit and pays for a sort of the positions on top, which costs more than the parse saves.
  const char *e = std::getenv("M4G_K2");
  return e && std::atoi(e) != 0 && p.chain_length > 0 && p.chain_length <= K2_MAX_CHAIN && p.distance_bit_cost >= 0 &&
         p.min_length >= 1 && max_chunk_bytes < (size_t)kK2MaxLen;
}
// The sorted position list serves fresh-state jobs built by the split K1 whose positions fit its entries.
bool sw_usable(const Params &p, size_t max_chunk_bytes) {
  // Opt-in (M4G_SW=1, read at every plan creation; K2 needs it and switches it on): the list costs a sort of the
  // positions (k1d_depth + k2_place: 10 ms per 256 MiB) and the parse that reads it is no faster than the one that
  // chases the chain (DESIGN.md 3.2c has the measurements and what they say about the bound).
  const char *e = std::getenv("M4G_SW"), *e2 = std::getenv("M4G_K2");
  return ((e && std::atoi(e) != 0) || (e2 && std::atoi(e2) != 0)) && p.chain_length > 0 && p.chain_length + 1 <= (int)kSortDepthCap &&
         max_chunk_bytes < ((size_t)1 << kSortPosBits);
}
constexpr int kDefaultSubBlock = 1024;

// Lanes a single streaming call is spread over at least (spec_alloc's min_lanes).
// One call is one job, so K3a's time is the serial walk of ONE sub-block per lane with nothing to
// hide its memory round trips behind: the smaller the sub-block the better, down to 64 bytes
// (measured per call, 64 KiB / 256 KiB / 1 MiB blocks: 512 lanes 344 / 1000 / 2285 us,
// 16384 lanes 272 / 456 / 1222 us; 32-byte sub-blocks lose again: 412 us at 64 KiB).
#ifndef M4G_STREAM_LANES
#define M4G_STREAM_LANES 16384
#endif
constexpr size_t kStreamLanes = M4G_STREAM_LANES;

int sub_block_size() {
  const char *e = std::getenv("M4G_SUB_BLOCK");
  int v = e ? std::atoi(e) : kDefaultSubBlock;
  return v >= 64 && v <= (1 << 24) ? v : kDefaultSubBlock;
}

// Layout for jobs of at most max_bytes buffer positions to parse.
// min_lanes > 0 (one job per launch: the streaming handle): sub-blocks small enough that the job
// spreads over at least that many lanes -- a 64 KiB block parsed by 64 lanes of 1 KiB each is one
// wave walking 64 epochs in a row (1.4 ms per call); in 128-byte sub-blocks it takes 0.37 ms, in 64-byte ones 0.27.
int spec_alloc(SpecWork &w, const Params &p, size_t max_jobs, size_t max_bytes, size_t min_lanes = 0,
               bool compact = false, size_t min_arena_bytes = 0) {
  spec_free(w);
  w.L.sub_block = sub_block_size();
  // Plans of a GiB and more have lanes to spare (2^19 sub-blocks of 2 KiB): longer sub-blocks mean fewer stitches
  // and less compaction (measured, 4096 x 1 MiB of text: K3 32.5 -> 31.6 ms, K4 2.6 -> 2.0; at 256 chunks the
  // inputs that defeat the speculation lose up to 25 %, so smaller plans keep 1 KiB).
  if (!min_lanes && !std::getenv("M4G_SUB_BLOCK") && p.chain_length == 0 && max_jobs * max_bytes >= ((size_t)1 << 30) &&
      max_bytes >= ((size_t)1 << 17)) // (chain walks are latency bound and want every lane they can get: params D 158 -> 184 ms per GiB)
    w.L.sub_block = 2 * kDefaultSubBlock;
  // Chain walks: the parse is a string of dependent far reads per lane, and what shortens the stage is more lanes
  // walking at once (measured, 256 x 1 MiB of text under ChainLength 64, tools/d_sub_ab.sh: sub-blocks of 1024 / 512 /
  // 256 / 128 / 64 bytes: K3 40.8 / 35.3 / 30.9 / 28.5 / 30.7 ms, K4 0.16 / 0.22 / 0.37 / 0.69 / 1.28).  Plans that
  // run K2 (k2_eval.h) walk (almost) no chains in the parse and keep the default.
  if (!min_lanes && !std::getenv("M4G_SUB_BLOCK") && p.chain_length > 0 && !(k2_usable(p, max_bytes) && sw_usable(p, max_bytes)) &&
      max_jobs * max_bytes >= ((size_t)1 << 26) && max_bytes >= ((size_t)1 << 14))
    // (by chain length and window, tools/chain_sub_probe.py, K3 + K4 per 256 MiB at 1024 / 512 / 256 / 128 bytes --
    // windows of 1 MiB: ChainLength 1: text 7.9 / 6.8 / 6.5 / 6.4 ms, log records 3.3 / 3.3 / 3.3 / 4.3; ChainLength 8: 17.9 /
    // 14.9 / 13.6 / 12.4 and 7.0 / 6.5 / 6.1 / 7.7; windows of 32 KiB (the flate levels): ChainLength 16: 10.4 / 10.4 / 10.8 /
    // 11.2 and 8.0 / 7.8 / 7.6 / 9.9, ChainLength 64: 12.2 / 12.0 / 12.4 / 12.7 and 17.6 / 17.1 / 17.0 / 21.3: far walks
    // want lanes, near ones do not care, and 128 bytes pay only for the longest chains over the widest windows:
    // ChainLength 64 over 16 MiB, the deep-search configuration: text 39.8 / 34.8 / 30.9 / 29.0, log records 30.0 / 27.9 /
    // 25.3 / 30.8 -- 128 is chosen for the text the configuration is measured on, at 20 % on log records)
    w.L.sub_block = p.max_distance <= (1 << 16) ? kDefaultSubBlock / 2
                    : p.chain_length >= 48      ? kDefaultSubBlock / 8
                                                : kDefaultSubBlock / 4;
  if (min_lanes && !std::getenv("M4G_SUB_BLOCK")) {
    int sub = 64;
    while (sub < w.L.sub_block && (size_t)sub * min_lanes < max_bytes) sub *= 2;
    w.L.sub_block = sub;
  }
  w.L.kmax = (uint32_t)std::max<size_t>(1, (max_bytes + (size_t)w.L.sub_block - 1) / (size_t)w.L.sub_block);
  // whole waves per job, so that the 64 sub-blocks of a wave belong to ONE job (k3a_fast needs it)
  if (w.L.kmax > K3_THREADS) w.L.kmax = (w.L.kmax + K3_THREADS - 1) / K3_THREADS * K3_THREADS;
  // records a parse can emit while walking one sub-block: the matches are
  // disjoint and at least min_length long, plus the pending ones at entry and
  // the tail records of the last block.
  w.slot_worst = w.L.slot = (uint32_t)((size_t)w.L.sub_block / (size_t)p.min_length + 18);
  w.blocks = max_jobs * w.L.kmax;
  size_t nrec = w.blocks * w.L.slot;
  hipError_t e = hipSuccess;
  if (compact) {
    // One record per 8 positions: what the slots have to hold in practice (records per KiB, measured with the
    // oracle on 1 MiB: text 66, binary log 32, random bytes of 2 / 3 / 4 symbols 118 / 109 / 90, 4-byte words
    // without separators 124; the bound above is 256 + 18 per KiB).  Inputs beyond that exist (ChainLength 8 with
    // HashLen 5 on 4 to 10 symbols: 150 - 180 per KiB): their first run overflows, is detected (error bit 1) and
    // repeated in the worst-case layout, which the plan then keeps.  The allocation always holds one job in that
    // layout, and the keys of the split K1 (min_arena_bytes), which are dead before K3 starts.
    // (rounded up to an ODD number of 128-byte lines: measured on 4096 x 1 MiB of text, slots of 258 / 256 / 264 /
    // 288 records and the worst case: 68.3 / 68.4 / 69.3 / 69.2 / 69.1 GB/s -- unaligned slots cost K4 0.3 ms and K3
    // 0.6, a power-of-two stride costs K3 0.9)
    uint32_t slot_c = ((uint32_t)w.L.sub_block / 8 + 2 + 7) / 8;
    slot_c = std::min<uint32_t>(w.slot_worst, (slot_c | 1u) * 8);
    if (const char *e = std::getenv("M4G_SLOT_RECS")) // tuning aid
      slot_c = std::min<uint32_t>(w.slot_worst, (uint32_t)std::max(8, std::atoi(e)));
    size_t recs = std::max(2 * w.blocks * slot_c, 2 * (size_t)w.L.kmax * w.slot_worst);
    recs = std::max(recs, (min_arena_bytes + sizeof(Rec) - 1) / sizeof(Rec));
    if (recs < 2 * nrec) {
      w.arena = true;
      w.arena_recs = recs;
      w.L.slot = slot_c;
      nrec = w.blocks * slot_c;
      if ((e = hipMalloc(&w.rs, recs * sizeof(Rec))) == hipSuccess) w.st = w.rs + nrec;
      w.bytes += recs * sizeof(Rec);
    }
  }
  if (!w.arena) {
    if ((e = hipMalloc(&w.rs, nrec * sizeof(Rec))) == hipSuccess) e = hipMalloc(&w.st, nrec * sizeof(Rec));
    w.bytes += 2 * nrec * sizeof(Rec);
  }
  if (e != hipSuccess || hipMalloc(&w.E, w.blocks * sizeof(ParseState)) != hipSuccess ||
      hipMalloc(&w.Tout, w.blocks * sizeof(ParseState)) != hipSuccess ||
      hipMalloc(&w.desc, w.blocks * sizeof(BlockDesc)) != hipSuccess) {
    (void)hipGetLastError();
    spec_free(w);
    return fail(M4G_ERR_NO_MEMORY, "hipMalloc of the parse workspace failed");
  }
  w.bytes += w.blocks * (2 * sizeof(ParseState) + sizeof(BlockDesc));
  return M4G_OK;
}

// Device buffers of the sorted position list (k2_eval.h) for a batch plan, and of K2 when that is asked for.
struct K2Work {
  bool sw = false;            // the plan builds the list and its parse walks it (the default for ChainLength > 0)
  bool on = false;            // ... and K2 evaluates every position's walk ahead of the parse (M4G_K2=1)
  SortEnt *sorted = nullptr;  // the positions of every job grouped by table index (indexed like the chain array)
  uint32_t *base = nullptr;   // njobs << table_bits: bucket sizes, then bucket bases
  uint32_t *rank = nullptr;   // per chain entry: depth in its bucket (k1d_depth), then index in `sorted` | depth
  K2Res *res = nullptr;       // K2 only: one per chain entry
  size_t bytes = 0;
};
// Buffers of the split K1 (k1_split.h): the bin-sorted keys of every tile and the
// per-tile bin headers.  keys == nullptr: the caller has none (streaming handle,
// the entry point that allocates nothing) and K1 is the one-kernel form.
struct K1Split {
  uint32_t *keys = nullptr; // one word per chain entry (same indexing as the chain array)
  uint32_t *hdr = nullptr;  // njobs x kt_max x KP_BINS words
  uint32_t kt_max = 0;      // tiles reserved per job
};
// The split form handles fresh-state jobs (batch chunks) whose table splits into at
// most 4 LDS partitions x 16 owner waves = KP_BINS bins.
bool k1_split_usable(const Params &p) {
  const int f = g_k1_force.load();
  if (f >= 0 && f != 2) return false;
  static const bool off = std::getenv("M4G_K1_NOSPLIT") != nullptr;
  return !off && p.table_bits >= 4 && p.table_bits <= kMaxPartBits + 2;
}

// Launches K1 + K3 for `njobs` jobs whose descriptors are already in d_jobs.
// On return (stream order) counts[job] holds the job's match count and the
// records sit in w.rs / w.st as described by w.desc.
// K1 for the jobs in d_jobs.  d_si: block structure when the jobs are segments
// of one stream (null otherwise).
int launch_k1(const Params &p, const uint8_t *d_src, uint64_t src_total, uint32_t max_len,
              uint32_t *d_chain, const Job *d_jobs, uint32_t *d_table_io, int njobs, hipStream_t st,
              const StreamInfo *d_si, uint32_t *d_table_out = nullptr, const K1Split *split = nullptr,
              bool fold = true /* false: a key for every position (K2 counts them) */, bool *split_ran = nullptr) {
  if (split_ran) *split_ran = false;
  const int pb = part_bits_of(p);
  const int nparts = 1 << (p.table_bits - pb);
  const int groups = (njobs + 7) / 8;
  dim3 g1((unsigned)(groups * 8 * nparts));
  // check bits kept next to a position in the LDS table (kernels.h, K1): as
  // many as fit above the largest position.  The persistent table of a
  // streaming handle stores plain positions; entries loaded from it carry
  // "unknown" check bits.
  int posbits = 1;
  while (posbits < 32 && (max_len >> posbits) != 0) posbits++;
  const int cb = std::max(0, std::min(K1_MAX_CHECK_BITS, 32 - posbits));
  if (cb < 1 && !k1_use_sort()) return fail(M4G_ERR_TOO_LARGE, "buffer positions need all 32 bits");
  if (split && split->keys && !d_si && !d_table_io && cb >= 1 && k1_split_usable(p) &&
      (g_k1_force.load() == 2 || !k1_use_sort()) && kp_tiles(max_len) <= split->kt_max) {
    const uint32_t gpj = (split->kt_max + KP_TPW - 1) / KP_TPW;
    const int bin_bits = (p.table_bits - pb) + 4;
    hipLaunchKernelGGL(k1p_partition, dim3((unsigned)njobs * gpj), dim3(KP_THREADS), 0, st, d_src, src_total,
                       split->keys, split->hdr, d_chain, d_jobs, njobs, split->kt_max, gpj, p.hash_len, p.table_bits, bin_bits, cb,
                       fold ? 1 : 0);
    HIP_TRY(hipGetLastError());
    if (split_ran) *split_ran = true;
    static const bool image = std::getenv("M4G_K1C_SCATTER") == nullptr;
    if (image)
      hipLaunchKernelGGL(k1c_consume, g1, dim3(KP_THREADS), k1c_lds_bytes(pb), st, split->keys, split->hdr, d_chain,
                         d_jobs, njobs, split->kt_max, p.table_bits, pb, cb);
    else
      hipLaunchKernelGGL(k1c_scatter, g1, dim3(KP_THREADS), k1s_lds_bytes(pb), st, split->keys, split->hdr, d_chain,
                         d_jobs, njobs, split->kt_max, p.table_bits, pb, cb);
    HIP_TRY(hipGetLastError());
    return M4G_OK;
  }
  if (!k1_use_sort()) {
    if (d_si)
      hipLaunchKernelGGL(k1x_chain_build<true>, g1, dim3(KX_THREADS), k1x_lds_bytes(pb), st, d_src, src_total, d_chain,
                         d_jobs, d_table_io, njobs, p.hash_len, p.table_bits, pb, cb, 0, d_si, d_table_out, 0u, 0u, 0u, 0u, 7u);
    else
      hipLaunchKernelGGL(k1x_chain_build<false>, g1, dim3(KX_THREADS), k1x_lds_bytes(pb), st, d_src, src_total, d_chain,
                         d_jobs, d_table_io, njobs, p.hash_len, p.table_bits, pb, cb, 0, d_si, d_table_out, 0u, 0u, 0u, 0u, 7u);
  } else if (d_si)
    hipLaunchKernelGGL(k1_chain_build<true>, g1, dim3(K1_THREADS), k1_lds_bytes(pb), st, d_src, src_total, d_chain,
                       d_jobs, d_table_io, njobs, p.hash_len, p.table_bits, pb, cb, 0, d_si, d_table_out);
  else
    hipLaunchKernelGGL(k1_chain_build<false>, g1, dim3(K1_THREADS), k1_lds_bytes(pb), st, d_src, src_total, d_chain,
                       d_jobs, d_table_io, njobs, p.hash_len, p.table_bits, pb, cb, 0, d_si, d_table_out);
  HIP_TRY(hipGetLastError());
  return M4G_OK;
}

// K3 for the jobs in d_jobs, whose chain entries K1 has produced.  On return
// (stream order) counts[job] holds the job's match count and the records sit
// in w.rs / w.st as described by w.desc.
int launch_parse(const Params &p, const uint8_t *d_src, uint64_t src_total, const uint32_t *d_chain,
                 const Job *d_jobs, int njobs, const SpecWork &w, uint32_t *d_counts, uint32_t *d_errors,
                 hipStream_t st, const K2Work *kw = nullptr /* what launch_k2 has built for d_chain's positions */) {
  ParseParams P = parse_params(p);
  const K2Res *k2 = kw && kw->on ? kw->res : nullptr;
  P.k2 = k2;
  if (kw && kw->sw && !k2) {
    P.sw = kw->sorted;
    P.rank = kw->rank;
  }
  const uint64_t lanes = (uint64_t)njobs * w.L.kmax;
  const unsigned g3 = (unsigned)((lanes + K3_THREADS - 1) / K3_THREADS);
  // tuning aid: unused dynamic LDS per workgroup = fewer waves per CU = fewer chunks in flight per XCD
  static const unsigned k3a_pad = std::getenv("M4G_K3A_LDS_PAD") ? (unsigned)std::atoi(std::getenv("M4G_K3A_LDS_PAD")) : 0u;
  // ChainLength 0: the event-driven form of the same speculative parse (k3a_fast, m4_fast.h)
  static const bool k3a_slow = std::getenv("M4G_K3A_EPOCHS") != nullptr;
  if (P.chain_length == 0 && w.L.kmax % K3_THREADS == 0 && !k3a_slow)
    hipLaunchKernelGGL(k3a_fast, dim3(g3), dim3(K3_THREADS), k3a_pad, st, d_src, src_total, d_chain, d_jobs,
                       w.rs, w.E, w.desc, d_errors, njobs, w.L, P);
  else if (P.chain_length == 0)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k3a_spec<true, false>), dim3(g3), dim3(K3_THREADS), k3a_pad, st, d_src, src_total, d_chain,
                       d_jobs, w.rs, w.E, w.desc, d_errors, njobs, w.L, P, 0);
  else if (k2) // (with K2's results only marked entries are events, as without chains)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k3a_spec<true, true>), dim3(g3), dim3(K3_THREADS), k3a_pad, st, d_src, src_total, d_chain,
                       d_jobs, w.rs, w.E, w.desc, d_errors, njobs, w.L, P, 0);
  else {
    // chain walks inside the parse: per-lane windows with a round budget (kernels.h, k3a_spec)
    static const int budget = std::getenv("M4G_K3_BUDGET") ? std::atoi(std::getenv("M4G_K3_BUDGET")) : 24;
    if (P.sw)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k3a_spec<false, true>), dim3(g3), dim3(K3_THREADS), k3a_pad, st, d_src, src_total,
                         d_chain, d_jobs, w.rs, w.E, w.desc, d_errors, njobs, w.L, P, budget);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k3a_spec<false, false>), dim3(g3), dim3(K3_THREADS), k3a_pad, st, d_src, src_total,
                         d_chain, d_jobs, w.rs, w.E, w.desc, d_errors, njobs, w.L, P, budget);
  }
  HIP_TRY(hipGetLastError());
  const bool xw = k2 || P.sw; // the kernels that read K2's results / the sorted list (parse_range's XW)
  if (w.L.kmax > 1) {
    if (P.chain_length == 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k3b_stitch<false, false>), dim3(g3), dim3(K3_THREADS), 0, st, d_src, d_chain, d_jobs, w.rs,
                         w.st, w.E, w.Tout, w.desc, d_errors, njobs, w.L, P);
    else if (xw)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k3b_stitch<true, true>), dim3(g3), dim3(K3_THREADS), 0, st, d_src, d_chain, d_jobs, w.rs,
                         w.st, w.E, w.Tout, w.desc, d_errors, njobs, w.L, P);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k3b_stitch<true, false>), dim3(g3), dim3(K3_THREADS), 0, st, d_src, d_chain, d_jobs, w.rs,
                         w.st, w.E, w.Tout, w.desc, d_errors, njobs, w.L, P);
    HIP_TRY(hipGetLastError());
  }
  if (xw)
    hipLaunchKernelGGL(k3c_validate<true>, dim3((unsigned)njobs), dim3(64), 0, st, d_src, d_chain, d_jobs,
                       w.rs, w.st, w.E, w.Tout, w.desc, d_counts, d_errors, njobs, w.L, P);
  else
    hipLaunchKernelGGL(k3c_validate<false>, dim3((unsigned)njobs), dim3(64), 0, st, d_src, d_chain, d_jobs,
                       w.rs, w.st, w.E, w.Tout, w.desc, d_counts, d_errors, njobs, w.L, P);
  HIP_TRY(hipGetLastError());
  return M4G_OK;
}

// The sorted position list (and K2, when the plan runs it) for the jobs in d_jobs, whose un-folded keys the split
// K1 has just left in split.keys / split.hdr.
int launch_k2(const Params &p, const K2Work &k, const K1Split &split, const uint8_t *d_src, uint64_t src_total,
              uint32_t max_len, uint32_t *d_chain, const Job *d_jobs, int njobs, hipStream_t st) {
  const int pb = part_bits_of(p);
  const int nparts = 1 << (p.table_bits - pb);
  const int groups = (njobs + 7) / 8;
  int posbits = 1;
  while (posbits < 32 && (max_len >> posbits) != 0) posbits++;
  const int cb = std::max(0, std::min(K1_MAX_CHECK_BITS, 32 - posbits)); // as launch_k1
  hipLaunchKernelGGL(k1d_depth, dim3((unsigned)(groups * 8 * nparts)), dim3(KP_THREADS), k1s_lds_bytes(pb), st, split.keys,
                     split.hdr, k.rank, 1u, k.base, d_jobs, njobs, split.kt_max, p.table_bits, pb, cb);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k2_scan_sizes, dim3((unsigned)njobs), dim3(1024), 0, st, k.base, njobs, p.table_bits);
  HIP_TRY(hipGetLastError());
  const uint32_t wpj_p = (max_len + K2P_THREADS * K2P_PER - 1) / (K2P_THREADS * K2P_PER);
  hipLaunchKernelGGL(k2_place, dim3((unsigned)groups * 8 * wpj_p), dim3(K2P_THREADS), 0, st, d_src, k.rank, k.base,
                     k.sorted, d_jobs, njobs, wpj_p, p.hash_len, p.table_bits);
  HIP_TRY(hipGetLastError());
  if (!k.on) return M4G_OK;
  const uint32_t wpj_e = (max_len + K2E_THREADS - 1) / K2E_THREADS;
  ParseParams PP = parse_params(p);
  if (const char *dbg = std::getenv("M4G_K2_DEBUG_WALK")) PP.chain_length = std::atoi(dbg) - 1; // timing aid: WRONG results
  hipLaunchKernelGGL(k2_eval, dim3((unsigned)groups * 8 * wpj_e), dim3(K2E_THREADS), 0, st, d_src, src_total, k.sorted,
                     d_chain, k.res, d_jobs, njobs, wpj_e, PP);
  HIP_TRY(hipGetLastError());
  return M4G_OK;
}

// K1 + K3 for `njobs` jobs whose descriptors are already in d_jobs.
int launch_find(const Params &p, const uint8_t *d_src, uint64_t src_total, uint32_t max_len,
                uint32_t *d_chain, const Job *d_jobs, uint32_t *d_table_io, int njobs, const SpecWork &w, uint32_t *d_counts,
                uint32_t *d_errors, hipStream_t st, hipEvent_t ev_mid, const K1Split *split = nullptr) {
  int rc = launch_k1(p, d_src, src_total, max_len, d_chain, d_jobs, d_table_io, njobs, st, nullptr, nullptr, split);
  if (rc) return rc;
  if (ev_mid) HIP_TRY(hipEventRecord(ev_mid, st));
  return launch_parse(p, d_src, src_total, d_chain, d_jobs, njobs, w, d_counts, d_errors, st);
}

// K4 for jobs already through launch_find: offsets, then the records.
int launch_compact(const SpecWork &w, const Job *d_jobs, const uint32_t *d_counts, int njobs,
                   uint64_t *d_out_off, m4g_match *d_out, size_t out_cap, hipStream_t st,
                   LongRun *d_long_runs = nullptr, uint32_t *d_n_long = nullptr, uint32_t long_cap = 0,
                   uint32_t job_base = 0 /* > 0: the jobs continue the list whose length is in d_out_off[0] */) {
  hipLaunchKernelGGL(k4_scan_counts, dim3(1), dim3(1024), 0, st, d_counts, d_out_off, njobs, job_base ? 1 : 0);
  HIP_TRY(hipGetLastError());
  const uint64_t waves = (uint64_t)njobs * w.L.kmax;
  const unsigned g4 = (unsigned)((waves + (K4_THREADS / 64) - 1) / (K4_THREADS / 64));
  hipLaunchKernelGGL(k4_compact, dim3(g4), dim3(K4_THREADS), 0, st, w.rs, w.st, d_jobs, w.desc,
                     d_out_off, reinterpret_cast<uint32_t *>(d_out), (uint64_t)out_cap, njobs, w.L,
                     d_long_runs, d_n_long, long_cap, job_base);
  HIP_TRY(hipGetLastError());
  return M4G_OK;
}

} // namespace

namespace {
struct Trace {
  bool on = std::getenv("M4G_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char *what) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[m4g] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
} // namespace

// ===========================================================================
// misc
// ===========================================================================
extern "C" const char *m4g_strerror(int code) {
  switch (code) {
  case M4G_OK: return "ok";
  case M4G_ERR_NO_DEVICE: return "no usable gfx950 device (libm4g has no CPU backend)";
  case M4G_ERR_BAD_PARAM: return "bad parameter";
  case M4G_ERR_HIP: return "HIP runtime error";
  case M4G_ERR_TOO_LARGE: return "input too large";
  case M4G_ERR_NO_MEMORY: return "out of memory";
  case M4G_ERR_REPEAT_RANGE: return "seeded repeat distance reaches before the history start";
  case M4G_ERR_INTERNAL: return "internal error (kernel-side overflow); the output of the call is not valid";
  case M4G_ERR_TOKEN_RANGE: return "a match is outside the flate token limits (Length 3..258, Distance 1..32768)";
  default: return code < 0 ? "output buffer too small (-(needed) returned)" : "ok";
  }
}
extern "C" const char *m4g_last_error(void) { return g_last_error.c_str(); }

extern "C" int m4g_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; d++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess &&
        std::strncmp(prop.gcnArchName, "gfx950", 6) == 0)
      ok++;
  }
  return ok;
}

extern "C" void *m4g_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
extern "C" void m4g_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

// ===========================================================================
// batch plan
// ===========================================================================
// Device memory one pass of the batch pipeline (K1 -> K3 -> K4 -> K5) works in.
// m4g_batch owns one (hipMalloc); the entry point that must not allocate points
// one at its file-scope scratch array.
struct BatchWork {
  uint32_t *d_chain = nullptr;
  SpecWork w;                     // RS / ST record slots, states, descriptors
  uint32_t *d_counts = nullptr;   // matches per job
  uint32_t *d_errors = nullptr;   // 16 words: [0] error bits, [4..7] k3c diagnostics, [8] long-run count
  Job *d_jobs = nullptr;
  LongRun *d_long_runs = nullptr; // literal runs >= K5_HUGE set aside by K4 for k5b_long_literals
  uint32_t long_cap = 0;
  K1Split split;                  // keys alias w.rs when that is large enough (dead until K3), else keys_own
  uint32_t *keys_own = nullptr;
  K2Work k2;                      // ChainLength > 0: the chain walks as a kernel of their own (k2_eval.h)
  // two halves of a large batch on two streams (run_batch_overlapped); plans of m4g_batch_create only
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_scan = nullptr, ev_join = nullptr;
  const Job *h_jobs = nullptr;    // host image of d_jobs (chain offsets)
};

// Device buffers of the brotli encoder (K6): owned by a plan (grown on demand) or by one call.
struct EncBuffers {
  uint32_t *tables = nullptr; // K6_TAB words per job
  uint64_t *pos = nullptr, *bits = nullptr; // per job: body bit position / bits written
  size_t jobs_cap = 0;
  uint32_t *words = nullptr;  // the stream
  size_t words_cap = 0;
  void release() {
    (void)hipFree(tables);
    (void)hipFree(pos);
    (void)hipFree(bits);
    (void)hipFree(words);
    *this = EncBuffers();
  }
};

struct GzipCarry {
  uint32_t crc = 0;   // CRC-32 of the bytes encoded so far
  uint64_t total = 0; // their number
};

// Device buffers of the flate encoder: owned by a plan (grown on demand) or by one call.
struct FlateBuffers {
  uint32_t *tables = nullptr;  // KF_TAB words per job: symbol counts, then code tables
  uint64_t *pos = nullptr, *bits = nullptr;
  uint32_t *mode = nullptr;
  size_t jobs_cap = 0;
  uint32_t *words = nullptr;
  size_t words_cap = 0;
  uint32_t *crc = nullptr;
  size_t crc_cap = 0;
  void release() {
    (void)hipFree(tables);
    (void)hipFree(pos);
    (void)hipFree(bits);
    (void)hipFree(mode);
    (void)hipFree(words);
    (void)hipFree(crc);
    *this = FlateBuffers();
  }
};

struct m4g_batch {
  int device = 0;
  Params p{};
  size_t max_chunks = 0, max_chunk_bytes = 0;
  BatchWork W;
  Job *h_jobs = nullptr;          // page-locked image of d_jobs (the upload is asynchronous)
  hipEvent_t ev_jobs = nullptr;   // the last upload of h_jobs has completed
  std::vector<uint64_t> last_off; // chunk_off of the jobs currently in d_jobs
  bool timing = false, ran = false;
  hipEvent_t ev[5] = {};
  hipStream_t last_stream = nullptr;
  EncBuffers enc;                  // m4g_batch_encode: its device buffers, grown on demand
  FlateBuffers flate;              // m4g_batch_flate_encode: its device buffers, grown on demand
  size_t bytes = 0;                // device memory of the plan (without enc / flate)
  struct LastRun {                 // the arguments of the most recent run: batch_settle repeats it after a slot overflow
    const uint8_t *d_src = nullptr;
    uint64_t src_total = 0;
    int njobs = 0;
    m4g_match *d_out = nullptr;
    uint64_t *d_out_off = nullptr;
    size_t out_cap = 0;
    m4g_histo *d_histos = nullptr;
    m4g_code *d_codes = nullptr;
  } last;
  uint32_t reruns = 0;             // runs repeated because a slot overflowed (at most one per plan: the switch is sticky)
};

extern "C" void m4g_batch_destroy(m4g_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->ran) (void)hipStreamSynchronize(b->last_stream);
  (void)hipFree(b->W.d_chain);
  spec_free(b->W.w);
  (void)hipFree(b->W.d_counts);
  (void)hipFree(b->W.d_errors);
  (void)hipFree(b->W.d_jobs);
  (void)hipFree(b->W.d_long_runs);
  (void)hipFree(b->W.split.hdr);
  (void)hipFree(b->W.keys_own);
  (void)hipFree(b->W.k2.sorted);
  (void)hipFree(b->W.k2.base);
  (void)hipFree(b->W.k2.res);
  (void)hipFree(b->W.k2.rank);
  if (b->W.side) (void)hipStreamSynchronize(b->W.side), (void)hipStreamDestroy(b->W.side);
  for (hipEvent_t e : {b->W.ev_fork, b->W.ev_scan, b->W.ev_join})
    if (e) (void)hipEventDestroy(e);
  b->enc.release();
  b->flate.release();
  if (b->h_jobs) (void)hipHostFree(b->h_jobs);
  if (b->ev_jobs) (void)hipEventDestroy(b->ev_jobs);
  for (auto &e : b->ev)
    if (e) (void)hipEventDestroy(e);
  delete b;
}

namespace {
int batch_create(const m4g_params *params, int device, size_t max_chunks, size_t max_chunk_bytes, bool compact,
                 m4g_batch **out);
}
// The plan a caller creates keeps its records in compact slots (spec_alloc; M4G_SLOTS=worst in the environment:
// the worst-case layout from the start); the host-memory entry points below create theirs (one per slice, small)
// in the worst-case layout, since they read the error word of several runs in flight themselves.
extern "C" int m4g_batch_create(const m4g_params *params, int device, size_t max_chunks,
                                size_t max_chunk_bytes, m4g_batch **out) {
  const char *e = std::getenv("M4G_SLOTS");
  return batch_create(params, device, max_chunks, max_chunk_bytes, !(e && std::strcmp(e, "worst") == 0), out);
}

namespace {
int batch_create(const m4g_params *params, int device, size_t max_chunks, size_t max_chunk_bytes, bool compact,
                 m4g_batch **out) {
  if (!out) return fail(M4G_ERR_BAD_PARAM, "out is NULL");
  *out = nullptr;
  Params p;
  int rc = resolve_params(params, &p);
  if (rc) return rc;
  if ((rc = check_device(device))) return rc;
  if (max_chunks == 0 || max_chunks > (1u << 24))
    return fail(M4G_ERR_BAD_PARAM, "max_chunks out of range");
  if (max_chunk_bytes == 0 || max_chunk_bytes > ((size_t)1 << 24))
    return fail(M4G_ERR_TOO_LARGE, "max_chunk_bytes above 2^24 (largest brotli meta-block)");
  HIP_TRY(hipSetDevice(device));
  if ((rc = k1_prepare())) return rc;

  m4g_batch *b = new m4g_batch();
  b->device = device;
  b->p = p;
  b->max_chunks = max_chunks;
  b->max_chunk_bytes = max_chunk_bytes;
  const size_t nbytes = max_chunks * max_chunk_bytes;
  const size_t chain_bytes = (nbytes + 16 * max_chunks) * 4;
  b->W.split.kt_max = kp_tiles((uint32_t)max_chunk_bytes);
  const bool split = p.table_bits >= 4 && p.table_bits <= kMaxPartBits + 2 && b->W.split.kt_max > 0;
  // (the keys of the split K1 live in the record slots: k1c reads up to 127 words past a bin)
  if ((rc = spec_alloc(b->W.w, p, max_chunks, max_chunk_bytes, 0, compact, split ? chain_bytes + 4096 : 0))) {
    m4g_batch_destroy(b);
    return rc;
  }
  hipError_t e;
  b->W.long_cap = (uint32_t)std::min<size_t>(max_chunks * (max_chunk_bytes / K5_HUGE + 1), 1u << 28);
  if ((e = hipMalloc(&b->W.d_chain, chain_bytes + 4096)) != hipSuccess ||
      (e = hipMalloc(&b->W.d_counts, max_chunks * 4)) != hipSuccess ||
      (e = hipMalloc(&b->W.d_errors, 64)) != hipSuccess ||
      (e = hipMalloc(&b->W.d_jobs, max_chunks * sizeof(Job))) != hipSuccess ||
      (e = hipHostMalloc(reinterpret_cast<void **>(&b->h_jobs), max_chunks * sizeof(Job), hipHostMallocDefault)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&b->ev_jobs, hipEventDisableTiming)) != hipSuccess ||
      (e = hipMalloc(&b->W.d_long_runs, b->W.long_cap * sizeof(LongRun))) != hipSuccess) {
    std::string msg = std::string("hipMalloc workspace: ") + hipGetErrorString(e);
    m4g_batch_destroy(b);
    return fail(M4G_ERR_NO_MEMORY, msg);
  }
  b->bytes = b->W.w.bytes + chain_bytes + 4096 + max_chunks * (4 + sizeof(Job)) + 64 + b->W.long_cap * sizeof(LongRun);
  if (split) { // buffers of the split K1 (the test hook may switch variants later)
    const SpecWork &w = b->W.w;
    const size_t rs_bytes = (w.arena ? w.arena_recs : w.blocks * w.L.slot) * sizeof(Rec);
    if (rs_bytes >= chain_bytes + 4096) {
      b->W.split.keys = reinterpret_cast<uint32_t *>(w.rs);
    } else if ((e = hipMalloc(&b->W.keys_own, chain_bytes + 4096)) == hipSuccess) {
      b->W.split.keys = b->W.keys_own;
      b->bytes += chain_bytes + 4096;
    }
    const size_t hdr_bytes = max_chunks * (size_t)b->W.split.kt_max * KP_BINS * 4;
    if (e == hipSuccess) e = hipMalloc(&b->W.split.hdr, hdr_bytes);
    if (e != hipSuccess) {
      std::string msg = std::string("hipMalloc K1 key workspace: ") + hipGetErrorString(e);
      m4g_batch_destroy(b);
      return fail(M4G_ERR_NO_MEMORY, msg);
    }
    b->bytes += hdr_bytes;
  }
  if (split && b->W.split.keys && sw_usable(p, max_chunk_bytes)) {
    K2Work &k = b->W.k2;
    const size_t entries = chain_bytes / 4 + 1024;
    const size_t base_bytes = (max_chunks << p.table_bits) * 4;
    const bool want_k2 = k2_usable(p, max_chunk_bytes);
    if (hipMalloc(&k.sorted, entries * sizeof(SortEnt)) != hipSuccess || hipMalloc(&k.base, base_bytes) != hipSuccess ||
        hipMalloc(&k.rank, entries * 4) != hipSuccess ||
        (want_k2 && hipMalloc(&k.res, entries * sizeof(K2Res)) != hipSuccess)) {
      // (not an error: the plan runs without the list, chasing the chains inside the parse)
      (void)hipGetLastError();
      (void)hipFree(k.sorted), (void)hipFree(k.base), (void)hipFree(k.rank), (void)hipFree(k.res);
      k = K2Work{};
    } else {
      k.sw = true;
      k.on = want_k2;
      k.bytes = entries * (sizeof(SortEnt) + 4 + (want_k2 ? sizeof(K2Res) : 0)) + base_bytes;
      b->bytes += k.bytes;
    }
  }
  for (auto &ev : b->ev) {
    if (hipEventCreate(&ev) != hipSuccess) {
      m4g_batch_destroy(b);
      return fail(M4G_ERR_HIP, "hipEventCreate failed");
    }
  }
  b->W.h_jobs = b->h_jobs;
  if (compact && nbytes >= ((size_t)1 << 28)) { // large plans of the C-ABI: the side stream of run_batch_overlapped
    if (hipStreamCreateWithFlags(&b->W.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&b->W.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->W.ev_scan, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->W.ev_join, hipEventDisableTiming) != hipSuccess) {
      m4g_batch_destroy(b);
      return fail(M4G_ERR_HIP, "creating the plan's side stream failed");
    }
  }
  *out = b;
  return M4G_OK;
}
} // namespace

extern "C" void m4g_batch_set_timing(m4g_batch *b, int enabled) {
  if (b) b->timing = enabled != 0;
}

extern "C" int m4g_batch_run_device(m4g_batch *b, const uint8_t *d_src, const uint64_t *chunk_off,
                                    size_t nchunks, m4g_match *d_out, uint64_t *d_out_off,
                                    size_t out_cap, m4g_histo *d_histos, void *hip_stream) {
  return m4g_batch_run_device_codes(b, d_src, chunk_off, nchunks, d_out, d_out_off, out_cap, d_histos,
                                    nullptr, hip_stream);
}

namespace {
// Job descriptors of a batch of independent chunks (fresh state each): chunk c
// is buffer [chunk_off[c], chunk_off[c+1]) of the source, its chain entries
// start at a 16-entry aligned offset.
int fill_batch_jobs(Job *jobs, const uint64_t *chunk_off, size_t nchunks, size_t max_chunk_bytes) {
  uint64_t chain_off = 0;
  for (size_t c = 0; c < nchunks; c++) {
    if (chunk_off[c + 1] < chunk_off[c]) return fail(M4G_ERR_BAD_PARAM, "chunk_off not ascending");
    const uint64_t n = chunk_off[c + 1] - chunk_off[c];
    if (n > max_chunk_bytes) return fail(M4G_ERR_TOO_LARGE, "chunk larger than the plan's max_chunk_bytes");
    Job J{};
    J.src_off = chunk_off[c];
    J.chain_off = chain_off;
    J.len = (uint32_t)n;
    jobs[c] = J;
    chain_off += (n + 15) & ~(uint64_t)15;
  }
  return M4G_OK;
}

// The same pipeline for a large batch as TWO halves on two streams: half A on the caller's stream, half B on the
// plan's side stream.  K1 is bound by instruction issue and K3a by memory round trips, and every launch ends in a
// tail in which the chip drains: with the halves side by side B's chain build runs next to A's parse and the tails
// are filled (4 GiB of text: 60.7 ms against 61.7 on one stream).  Both halves start together: holding B back until
// A's chain build is done (so that B's K1 meets A's K3) gave 62.2 ms, a lower or higher priority for the side stream
// 62.5 / 62.4 -- the two kernels do not share a CU (K3a fills the registers, k1c the LDS), so what is gained is the
// tails, not a mix of an issue-bound and a latency-bound kernel.
// What has to be kept apart: the record slots double as the key scratch of the split K1.  Half A owns the first
// 2 * jm * kmax * slot records of the allocation (its RS, then its ST), half B the rest; A's keys start at the
// front (as always), B's keys END at the end of the allocation, so that neither half's keys touch the other half's
// records (checked below from the jobs' chain offsets; the caller falls back to one stream when it does not fit).
// B's compaction continues A's offsets (k4_scan_counts' carry), so it waits for A's; the long literal runs are
// counted after both halves' histograms exist.  Returns 1 when the batch does not qualify.
int run_batch_overlapped(const Params &p, const BatchWork &W, uint32_t max_len, const uint8_t *d_src,
                         uint64_t src_total, int njobs, m4g_match *d_out, uint64_t *d_out_off, size_t out_cap,
                         m4g_histo *d_histos, m4g_code *d_codes, hipStream_t st) {
  static const bool off = std::getenv("M4G_NO_OVERLAP") != nullptr;
  const SpecWork &w = W.w;
  // (tools/overlap_ab.sh, text in 1 MiB chunks, overlapped against one stream: 256 chunks 5.10 / 5.10 ms, 512: 8.66 /
  // 8.86, 1024: 15.96 / 16.01, 2048: 30.93 / 31.32, 4096: 60.8 / 61.9)
  static const int min_jobs = std::getenv("M4G_OVERLAP_MIN_CHUNKS") ? std::atoi(std::getenv("M4G_OVERLAP_MIN_CHUNKS")) : 512;
  if (off || W.k2.sw || !W.side || !W.h_jobs || !w.arena || w.group_jobs || njobs < min_jobs || src_total < (uint64_t)min_jobs << 20 ||
      W.split.keys != reinterpret_cast<uint32_t *>(w.rs))
    return 1;
  const int jm = njobs / 2, nb = njobs - jm;
  const uint64_t words_total = (uint64_t)w.arena_recs * (sizeof(Rec) / 4);
  const uint64_t recs_a = 2ull * (uint64_t)jm * w.L.kmax * w.L.slot, recs_b = 2ull * (uint64_t)nb * w.L.kmax * w.L.slot;
  if (recs_a + recs_b > w.arena_recs) return 1;
  const Job &last = W.h_jobs[njobs - 1];
  const uint64_t chain_end = last.chain_off + (((uint64_t)last.len + 15) & ~15ull); // words
  const uint64_t chain_mid = W.h_jobs[jm].chain_off;
  const uint64_t margin = 1024; // words k1c may read past a bin
  const uint64_t a_words = recs_a * (sizeof(Rec) / 4);
  if (chain_mid + margin > a_words) return 1;                       // A's keys inside A's records
  if (words_total < chain_end + margin) return 1;
  const uint64_t b_shift = words_total - margin - chain_end;        // B's keys: keys + b_shift + chain_off
  if (b_shift + chain_mid < a_words) return 1;                      // ... and clear of A's records
  SpecWork va = w, vb = w;
  va.st = w.rs + (uint64_t)jm * w.L.kmax * w.L.slot;
  vb.rs = w.rs + recs_a;
  vb.st = vb.rs + (uint64_t)nb * w.L.kmax * w.L.slot;
  vb.E = w.E + (uint64_t)jm * w.L.kmax;
  vb.Tout = w.Tout + (uint64_t)jm * w.L.kmax;
  vb.desc = w.desc + (uint64_t)jm * w.L.kmax;
  K1Split sb = W.split;
  sb.keys = W.split.keys + b_shift;
  sb.hdr = W.split.hdr + (uint64_t)jm * W.split.kt_max * KP_BINS;
  hipStream_t s2 = W.side;
  uint32_t *n_long = W.d_errors + 8;
  LongRun *runs = d_histos ? W.d_long_runs : nullptr;

  HIP_TRY(hipMemsetAsync(W.d_errors, 0, 64, st));
  HIP_TRY(hipEventRecord(W.ev_fork, st)); // inputs, job descriptors and the cleared error word are in place
  HIP_TRY(hipStreamWaitEvent(s2, W.ev_fork, 0));
  int rc;
  // half A: K1, K3, K4
  if ((rc = launch_k1(p, d_src, src_total, max_len, W.d_chain, W.d_jobs, nullptr, jm, st, nullptr, nullptr, &W.split))) return rc;
  if ((rc = launch_parse(p, d_src, src_total, W.d_chain, W.d_jobs, jm, va, W.d_counts, W.d_errors, st))) return rc;
  if ((rc = launch_compact(va, W.d_jobs, W.d_counts, jm, d_out_off, d_out, out_cap, st, runs, n_long, W.long_cap, 0))) return rc;
  HIP_TRY(hipEventRecord(W.ev_scan, st));
  // half B: K1, K3, then K4 behind A's
  if ((rc = launch_k1(p, d_src, src_total, max_len, W.d_chain, W.d_jobs + jm, nullptr, nb, s2, nullptr, nullptr, &sb))) return rc;
  if ((rc = launch_parse(p, d_src, src_total, W.d_chain, W.d_jobs + jm, nb, vb, W.d_counts + jm, W.d_errors, s2))) return rc;
  HIP_TRY(hipStreamWaitEvent(s2, W.ev_scan, 0));
  if ((rc = launch_compact(vb, W.d_jobs + jm, W.d_counts + jm, nb, d_out_off + jm, d_out, out_cap, s2, runs, n_long, W.long_cap,
                           (uint32_t)jm)))
    return rc;
  if (d_histos) {
    auto k5 = [&](int j0, int nj, hipStream_t s) {
      uint32_t *h = reinterpret_cast<uint32_t *>(d_histos) + (size_t)j0 * 1024;
      if (d_codes)
        hipLaunchKernelGGL(k5_histograms<true>, dim3((unsigned)nj), dim3(K5_THREADS), 0, s, d_src,
                           reinterpret_cast<const uint32_t *>(d_out), W.d_jobs + j0, d_out_off + j0, (uint64_t)out_cap, h,
                           reinterpret_cast<CodeRec *>(d_codes), W.d_errors);
      else
        hipLaunchKernelGGL(k5_histograms<false>, dim3((unsigned)nj), dim3(K5_THREADS), 0, s, d_src,
                           reinterpret_cast<const uint32_t *>(d_out), W.d_jobs + j0, d_out_off + j0, (uint64_t)out_cap, h,
                           static_cast<CodeRec *>(nullptr), W.d_errors);
    };
    k5(0, jm, st);
    HIP_TRY(hipGetLastError());
    k5(jm, nb, s2);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(W.ev_join, s2));
  HIP_TRY(hipStreamWaitEvent(st, W.ev_join, 0));
  if (d_histos) {
    const uint32_t segs = (uint32_t)(((size_t)max_len + K5B_SEG - 1) / K5B_SEG);
    hipLaunchKernelGGL(k5b_long_literals, dim3(2048), dim3(K5B_THREADS), 0, st, d_src, W.d_jobs,
                       W.d_long_runs, n_long, W.long_cap, segs, d_out_off, (uint64_t)out_cap,
                       reinterpret_cast<uint32_t *>(d_histos), W.d_errors);
    HIP_TRY(hipGetLastError());
  }
  return M4G_OK;
}

// K1 -> K3 -> K4 -> K5 for the njobs jobs already in W.d_jobs.  Everything is
// enqueued on st; nothing here allocates or synchronises.  ev: five events
// (stage boundaries) or null.
int run_batch_pipeline(const Params &p, const BatchWork &W, uint32_t max_len, const uint8_t *d_src,
                       uint64_t src_total, int njobs, m4g_match *d_out, uint64_t *d_out_off, size_t out_cap,
                       m4g_histo *d_histos, m4g_code *d_codes, hipStream_t st, hipEvent_t *ev) {
  if (!ev) { // (the stage events belong to the one-stream form)
    const int r2 = run_batch_overlapped(p, W, max_len, d_src, src_total, njobs, d_out, d_out_off, out_cap, d_histos, d_codes, st);
    if (r2 != 1) return r2;
  }
  HIP_TRY(hipMemsetAsync(W.d_errors, 0, 64, st));
  if (ev) HIP_TRY(hipEventRecord(ev[0], st));
  bool split_ran = false;
  int rc = launch_k1(p, d_src, src_total, max_len, W.d_chain, W.d_jobs, nullptr, njobs, st, nullptr, nullptr, &W.split,
                     !W.k2.sw, &split_ran);
  if (rc) return rc;
  const K2Work *k2 = nullptr;
  if (W.k2.sw && split_ran) { // (stage K1 of the timing events includes the sort, and K2 when it runs)
    if ((rc = launch_k2(p, W.k2, W.split, d_src, src_total, max_len, W.d_chain, W.d_jobs, njobs, st))) return rc;
    k2 = &W.k2;
  }
  if (ev) HIP_TRY(hipEventRecord(ev[1], st));
  uint32_t *n_long = W.d_errors + 8; // zeroed with d_errors above
  // K3 + K4, for all jobs at once -- or, for a compact plan after its switch to the worst-case layout, group after
  // group in the same record slots (a group's records are compacted before the next group's parse overwrites them;
  // ev[2] then stands after the last parse, so the earlier groups' compaction counts as parse time)
  const int group = W.w.group_jobs ? (int)std::min<size_t>(W.w.group_jobs, (size_t)njobs) : njobs;
  for (int j0 = 0; j0 < njobs; j0 += group) {
    const int nj = std::min(group, njobs - j0);
    if ((rc = launch_parse(p, d_src, src_total, W.d_chain, W.d_jobs + j0, nj, W.w, W.d_counts + j0, W.d_errors, st, k2)))
      return rc;
    if (ev && j0 + nj >= njobs) HIP_TRY(hipEventRecord(ev[2], st));
    if ((rc = launch_compact(W.w, W.d_jobs + j0, W.d_counts + j0, nj, d_out_off + j0, d_out, out_cap, st,
                             d_histos ? W.d_long_runs : nullptr, n_long, W.long_cap, (uint32_t)j0)))
      return rc;
  }
  if (ev) HIP_TRY(hipEventRecord(ev[3], st));

  if (d_histos) {
    static_assert(sizeof(m4g_histo) == 1024 * 4, "m4g_histo layout");
    static_assert(sizeof(m4g_code) == sizeof(CodeRec), "m4g_code layout");
    if (d_codes)
      hipLaunchKernelGGL(k5_histograms<true>, dim3((unsigned)njobs), dim3(K5_THREADS), 0, st, d_src,
                         reinterpret_cast<const uint32_t *>(d_out), W.d_jobs, d_out_off,
                         (uint64_t)out_cap, reinterpret_cast<uint32_t *>(d_histos),
                         reinterpret_cast<CodeRec *>(d_codes), W.d_errors);
    else
      hipLaunchKernelGGL(k5_histograms<false>, dim3((unsigned)njobs), dim3(K5_THREADS), 0, st, d_src,
                         reinterpret_cast<const uint32_t *>(d_out), W.d_jobs, d_out_off,
                         (uint64_t)out_cap, reinterpret_cast<uint32_t *>(d_histos),
                         static_cast<CodeRec *>(nullptr), W.d_errors);
    HIP_TRY(hipGetLastError());
    // the literal runs K4 set aside (none on compressible input: the grid finds an empty list)
    const uint32_t segs = (uint32_t)(((size_t)max_len + K5B_SEG - 1) / K5B_SEG);
    hipLaunchKernelGGL(k5b_long_literals, dim3(2048), dim3(K5B_THREADS), 0, st, d_src, W.d_jobs,
                       W.d_long_runs, n_long, W.long_cap, segs, d_out_off, (uint64_t)out_cap,
                       reinterpret_cast<uint32_t *>(d_histos), W.d_errors);
    HIP_TRY(hipGetLastError());
  }
  if (ev) HIP_TRY(hipEventRecord(ev[4], st));
  return M4G_OK;
}
} // namespace

extern "C" int m4g_batch_run_device_codes(m4g_batch *b, const uint8_t *d_src, const uint64_t *chunk_off,
                                          size_t nchunks, m4g_match *d_out, uint64_t *d_out_off,
                                          size_t out_cap, m4g_histo *d_histos, m4g_code *d_codes,
                                          void *hip_stream) {
  if (!b || !chunk_off || !d_out_off || (!d_out && out_cap))
    return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (d_codes && !d_histos)
    return fail(M4G_ERR_BAD_PARAM, "d_codes needs d_histos (both come from the same pass)");
  if (d_histos && b->p.min_length < 2) // a trimmed match could have Length 1: no copy length code (command.go:128)
    return fail(M4G_ERR_BAD_PARAM, "histograms / codes need MinLength >= 2 (brotli has no copy length code for 1)");
  if (nchunks > b->max_chunks) return fail(M4G_ERR_TOO_LARGE, "nchunks above the plan's max_chunks");
  if (nchunks && !d_src) return fail(M4G_ERR_BAD_PARAM, "d_src is NULL");
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int njobs = (int)nchunks;
  b->ran = false;

  if (njobs == 0) {
    HIP_TRY(hipMemsetAsync(d_out_off, 0, sizeof(uint64_t), st));
    return M4G_OK;
  }

  // job descriptors (re-uploaded only when the chunk layout changed)
  bool same = b->last_off.size() == nchunks + 1 &&
              std::memcmp(b->last_off.data(), chunk_off, (nchunks + 1) * 8) == 0;
  if (!same) {
    b->last_off.clear();
    HIP_TRY(hipEventSynchronize(b->ev_jobs)); // an earlier upload may still be reading h_jobs
    if (int rc = fill_batch_jobs(b->h_jobs, chunk_off, nchunks, b->max_chunk_bytes)) return rc;
    HIP_TRY(hipMemcpyAsync(b->W.d_jobs, b->h_jobs, nchunks * sizeof(Job), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(b->ev_jobs, st));
    b->last_off.assign(chunk_off, chunk_off + nchunks + 1);
  }

  int rc = run_batch_pipeline(b->p, b->W, (uint32_t)b->max_chunk_bytes, d_src, chunk_off[nchunks], njobs, d_out,
                              d_out_off, out_cap, d_histos, d_codes, st, b->timing ? b->ev : nullptr);
  if (rc) return rc;
  if (std::getenv("M4G_TRACE_K3C")) { // diagnostics of the validation pass (forces a sync)
    uint32_t w[16] = {};
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(w, b->W.d_errors, sizeof(w), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "[m4g] k3c: %u scan steps, %u blocks accepted by the probe check, %u parsed serially, %u skipped (covered)\n",
                 w[4], w[5], w[6], w[7]);
    if (w[13]) // builds with M4G_KF_STATS: k3a_fast's rounds, summed over its waves
      std::fprintf(stderr, "[m4g] k3a_fast: %u wave-rounds, %u of them ran the parked batch, %u filled windows\n", w[13], w[14], w[15]);
    if (w[13] && w[9]) std::fprintf(stderr, "[m4g] k3a_fast wave-ms: total %.2f, parked batches %.2f\n", w[9] * 256 / 1e5, w[10] * 256 / 1e5);
    else if (w[9] | w[10] | w[11] | w[12]) // builds with M4G_K3C_TIMING: wave time per path, summed over the chunks
      std::fprintf(stderr, "[m4g] k3c wave-ms: scan %.2f, covered skip %.2f, probe check %.2f, serial %.2f\n",
                   w[9] * 16 / 1e5, w[10] * 16 / 1e5, w[11] * 16 / 1e5, w[12] * 16 / 1e5);
  }
  b->ran = true;
  b->last_stream = st;
  b->last.d_src = d_src;
  b->last.src_total = chunk_off[nchunks];
  b->last.njobs = njobs;
  b->last.d_out = d_out;
  b->last.d_out_off = d_out_off;
  b->last.out_cap = out_cap;
  b->last.d_histos = d_histos;
  b->last.d_codes = d_codes;
  return M4G_OK;
}

namespace {
int status_from_error_word(uint32_t err) {
  if (err & 2u) return fail(M4G_ERR_INTERNAL, "parse slot overflow: records were dropped, the output is not valid");
  if (err & 1u) return fail(M4G_ERR_REPEAT_RANGE, "repeat distance out of range");
  return M4G_OK;
}
} // namespace

namespace {
// Waits for the plan's most recent run and returns its kernel-side outcome.  A compact plan whose record slots
// overflowed switches to the worst-case layout here (for good) and runs the batch again, on the same stream into
// the same output buffers: its chain entries, job descriptors and arguments are all still in place.
int batch_settle(m4g_batch *b) {
  if (!b->ran) return M4G_OK;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  uint32_t err = 0;
  HIP_TRY(hipMemcpy(&err, b->W.d_errors, 4, hipMemcpyDeviceToHost));
  if ((err & 2u) && b->W.w.compact()) {
    spec_use_worst(b->W.w);
    b->reruns++;
    if (std::getenv("M4G_TRACE"))
      std::fprintf(stderr, "[m4g] record slots overflowed: the plan now parses %zu jobs at a time in worst-case slots; running the batch again\n",
                   b->W.w.group_jobs);
    const m4g_batch::LastRun &r = b->last;
    if (int rc = run_batch_pipeline(b->p, b->W, (uint32_t)b->max_chunk_bytes, r.d_src, r.src_total, r.njobs, r.d_out,
                                    r.d_out_off, r.out_cap, r.d_histos, r.d_codes, b->last_stream, b->timing ? b->ev : nullptr))
      return rc;
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    HIP_TRY(hipMemcpy(&err, b->W.d_errors, 4, hipMemcpyDeviceToHost));
  }
  return status_from_error_word(err);
}
} // namespace

extern "C" int m4g_batch_status(m4g_batch *b) {
  if (!b) return fail(M4G_ERR_BAD_PARAM, "plan is NULL");
  return batch_settle(b);
}

extern "C" int m4g_batch_info(m4g_batch *b, uint64_t *out, int cap) {
  if (!b || !out) return 0;
  const uint64_t v[8] = {b->bytes, b->W.w.L.slot, b->W.w.slot_worst, b->W.w.group_jobs, b->reruns,
                         b->W.k2.on ? 1u : 0u, b->W.k2.bytes, b->W.k2.sw ? 1u : 0u};
  const int n = std::min(cap, 8);
  for (int k = 0; k < n; k++) out[k] = v[k];
  return n;
}

// Test hooks for K2 (k2_eval.h).  _set_k2: a plan that holds K2's buffers runs with (1) or without (0) it from
// the next run on; returns whether the plan holds them.  _dump_k2: after a run (synchronises its stream), the
// chain entries (mark bit included) and K2's results of chunk c of that run, n = the chunk's length each.
extern "C" int m4g_debug_batch_set_k2(m4g_batch *b, int on) {
  if (!b || !b->W.k2.res) return 0;
  b->W.k2.on = on != 0;
  return 1;
}
// The same switch for the sorted position list (the parse chases the chains when it is off).
extern "C" int m4g_debug_batch_set_sorted_walk(m4g_batch *b, int on) {
  if (!b || !b->W.k2.sorted) return 0;
  b->W.k2.sw = on != 0;
  if (!on) b->W.k2.on = false;
  return 1;
}
extern "C" int m4g_debug_batch_dump_k2(m4g_batch *b, size_t c, uint32_t *chain_out, uint32_t *res_out, size_t n) {
  if (!b || !b->ran || c >= (size_t)b->last.njobs) return fail(M4G_ERR_BAD_PARAM, "no run, or no such chunk");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  const Job &J = b->h_jobs[c];
  if (n > J.len) return fail(M4G_ERR_BAD_PARAM, "n above the chunk's length");
  if (chain_out) HIP_TRY(hipMemcpy(chain_out, b->W.d_chain + J.chain_off, n * 4, hipMemcpyDeviceToHost));
  if (res_out) {
    if (!b->W.k2.res) return fail(M4G_ERR_BAD_PARAM, "the plan has no K2 buffers");
    HIP_TRY(hipMemcpy(res_out, b->W.k2.res + J.chain_off, n * sizeof(K2Res), hipMemcpyDeviceToHost));
  }
  return M4G_OK;
}

// ===========================================================================
// GPU encoder: brotli.Encoder.Encode for every chunk of a batch (include/m4g.h,
// m4g_batch_encode).  Pass 1 ran on the GPU already (K5: histograms + codes);
// the prefix codes are built here on host threads (enc_host.h), pass 2 is K6.
// ===========================================================================
namespace {
// Jobs = meta-blocks in stream order; d_histos / d_codes from k5_histograms<true> over the same jobs.
int64_t brotli_encode_jobs(EncBuffers &B, hipEvent_t ev0, hipEvent_t ev1, const uint8_t *d_src, const Job *d_jobs,
                           const std::vector<uint64_t> &lens, const m4g_match *d_out, const uint64_t *d_out_off,
                           const m4g_histo *d_histos, const m4g_code *d_codes, int flags, uint8_t *stream_out,
                           size_t stream_cap, hipStream_t st, float *ms_out, const char *what,
                           unsigned start_bits = 0, uint64_t *end_bits_out = nullptr) {
  // start_bits (< 8): the stream continues one whose last byte holds that many bits; they belong to the caller,
  // who ORs them into byte 0 of stream_out.  end_bits_out: the bit position where this call stopped.
  using namespace m4g_enc;
  const size_t nchunks = lens.size();
  const auto t_begin = std::chrono::steady_clock::now();
  // the histograms of the run (its kernels must have finished)
  std::vector<uint32_t> histos(nchunks * 1024);
  HIP_TRY(hipStreamSynchronize(st));
  if (nchunks) HIP_TRY(hipMemcpy(histos.data(), d_histos, nchunks * sizeof(m4g_histo), hipMemcpyDeviceToHost));
  // per chunk: meta-block header + prefix codes, code tables, body length -- on host threads
  static const CodeLengthCode cl;
  static const CmdExtra cx;
  std::vector<ChunkPlan> plans(nchunks);
  std::vector<uint32_t> tables(nchunks * (size_t)K6_TAB);
  {
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned nt = (unsigned)std::min<size_t>(hw, (nchunks + 31) / 32 + 1);
    auto work = [&](unsigned t) {
      for (size_t c = t; c < nchunks; c += nt) {
        if (lens[c] == 0) continue; // Encode writes nothing for an empty block (encoder.go:24-31)
        plan_chunk(histos.data() + c * 1024, (size_t)lens[c], tables.data() + c * K6_TAB, plans[c], cl, cx);
        plans[c].prologue.flush();
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  const auto t_planned = std::chrono::steady_clock::now();
  // bit positions (one bit writer for the whole stream: bitwriter.go state carries over between blocks)
  std::vector<uint64_t> body_pos(nchunks + 1), pro_pos(nchunks);
  uint64_t pos = start_bits + ((flags & 1) ? 4 : 0); // stream header: WBITS = 24 (encoder.go:19-22)
  for (size_t c = 0; c < nchunks; c++) {
    pro_pos[c] = pos;
    body_pos[c] = pos + plans[c].prologue.total;
    pos = body_pos[c] + plans[c].body_bits;
  }
  const uint64_t end_pos = pos;
  if (flags & 2) pos += 2; // ISLAST + ISEMPTY (encoder.go:156-159)
  if (end_bits_out) *end_bits_out = pos;
  const uint64_t nbytes = (pos + 7) / 8;
  if (nbytes > stream_cap) return -(int64_t)(nbytes + M4G_STREAM_NEEDED_BIAS);
  if (!stream_out && nbytes) return fail(M4G_ERR_BAD_PARAM, "stream_out is NULL");
  const size_t nwords = (size_t)(nbytes / 4 + 2);
  if (nchunks > 0) {
    if (nchunks > B.jobs_cap) {
      (void)hipFree(B.tables);
      (void)hipFree(B.pos);
      (void)hipFree(B.bits);
      B.tables = nullptr;
      B.pos = B.bits = nullptr;
      B.jobs_cap = 0;
      if (hipMalloc(&B.tables, nchunks * (size_t)K6_TAB * 4) != hipSuccess || hipMalloc(&B.pos, nchunks * 8) != hipSuccess ||
          hipMalloc(&B.bits, nchunks * 8) != hipSuccess)
        return fail(M4G_ERR_NO_MEMORY, "hipMalloc of the encoder tables failed");
      B.jobs_cap = nchunks;
    }
    if (nwords > B.words_cap) {
      (void)hipFree(B.words);
      B.words = nullptr;
      B.words_cap = 0;
      if (hipMalloc(&B.words, nwords * 4) != hipSuccess) return fail(M4G_ERR_NO_MEMORY, "hipMalloc of the stream failed");
      B.words_cap = nwords;
    }
    HIP_TRY(hipMemsetAsync(B.words, 0, nwords * 4, st));
    HIP_TRY(hipMemcpyAsync(B.tables, tables.data(), nchunks * (size_t)K6_TAB * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(B.pos, body_pos.data(), nchunks * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(k6_emit_bodies, dim3((unsigned)nchunks), dim3(K6_THREADS), 0, st, d_src,
                       reinterpret_cast<const uint32_t *>(d_out), reinterpret_cast<const CodeRec *>(d_codes), d_jobs,
                       d_out_off, B.tables, B.pos, B.words, B.bits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, st));
    std::vector<uint64_t> written(nchunks);
    HIP_TRY(hipMemcpyAsync(written.data(), B.bits, nchunks * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(stream_out, B.words, nbytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t c = 0; c < nchunks; c++)
      if (written[c] != plans[c].body_bits)
        return fail(M4G_ERR_INTERNAL, std::string(what) + ": a chunk's body length differs from the histogram prediction "
                                                          "(d_out / d_histos / d_codes are not from the same run?)");
    if (ms_out) HIP_TRY(hipEventElapsedTime(&ms_out[1], ev0, ev1));
  } else if (nbytes) {
    std::memset(stream_out, 0, nbytes);
  }
  // the host-written fields share bytes with the bodies: OR them in
  if (flags & 1) {
    const uint8_t hdr = 15;
    or_bits(stream_out, start_bits, &hdr, 4);
  }
  for (size_t c = 0; c < nchunks; c++)
    if (plans[c].prologue.total) or_bits(stream_out, pro_pos[c], plans[c].prologue.dst.data(), plans[c].prologue.total);
  if (flags & 2) {
    const uint8_t fin = 3;
    or_bits(stream_out, end_pos, &fin, 2);
  }
  if (ms_out) {
    ms_out[0] = std::chrono::duration<float, std::milli>(t_planned - t_begin).count();
    ms_out[2] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (nchunks == 0) ms_out[1] = 0.f;
  }
  return (int64_t)nbytes;
}

// Device copies of one stream's bytes, Match lists and block jobs (the *_encode_blocks entry points):
// Writer.Write encodes every full block, Writer.Close the remainder -- possibly empty -- as the last one.
struct BlockUpload {
  uint8_t *d_src = nullptr;
  m4g_match *d_m = nullptr;
  uint64_t *d_off = nullptr;
  Job *d_jobs = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<Job> jobs;
  std::vector<uint64_t> lens, off;
  uint64_t nmatches = 0;
  bool borrowed = false; // d_src / d_m belong to the caller (bytes and Match lists already on the device)
  ~BlockUpload() {
    if (!borrowed) (void)hipFree(d_src);
    if (!borrowed) (void)hipFree(d_m);
    (void)hipFree(d_off);
    (void)hipFree(d_jobs);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
  }
  // whole_stream: jobs = every full block + the remainder, possibly empty (Write; Close).  Otherwise exactly the
  // ceil(n / block_size) blocks of the lists.
  // dev_src / dev_matches non-null: the bytes and the Match lists are on the device already (src, matches unused)
  int init(int device, const uint8_t *src, size_t n, size_t block_size, const m4g_match *matches, const uint64_t *block_off,
           bool whole_stream = true, uint8_t *dev_src = nullptr, m4g_match *dev_matches = nullptr) {
    borrowed = dev_src != nullptr;
    if ((!src && n && !borrowed) || block_size == 0 || !block_off) return fail(M4G_ERR_BAD_PARAM, "NULL argument or block_size 0");
    if (n >= ((uint64_t)1 << 32)) return fail(M4G_ERR_TOO_LARGE, "at most 4 GiB per call");
    if (block_size >= ((uint64_t)1 << 31)) return fail(M4G_ERR_TOO_LARGE, "block_size must stay below 2^31");
    if (int rc = check_device(device)) return rc;
    HIP_TRY(hipSetDevice(device));
    const size_t nlists = (n + block_size - 1) / block_size; // block_off has nlists + 1 entries
    const size_t nfull = n / block_size, njobs = whole_stream ? nfull + 1 : nlists;
    if (njobs == 0) return fail(M4G_ERR_BAD_PARAM, "no block");
    jobs.resize(njobs);
    lens.resize(njobs);
    off.resize(njobs + 1);
    for (size_t j = 0; j < njobs; j++) {
      Job J{};
      J.src_off = (uint64_t)j * block_size;
      J.len = (uint32_t)std::min<uint64_t>(block_size, n - J.src_off);
      jobs[j] = J;
      lens[j] = J.len;
      off[j] = block_off[std::min(j, nlists)];
    }
    off[njobs] = block_off[nlists];
    nmatches = block_off[nlists];
    if (nmatches && !matches && !borrowed) return fail(M4G_ERR_BAD_PARAM, "matches is NULL");
    // A caller's lists must stay inside their blocks: the encoders read src[pos .. pos + Unmatched) for every match
    // (the reference slices the same way and panics on a list that runs past the block, encoder.go:45-52).
    for (size_t j = 0; j < njobs; j++)
      if (off[j + 1] < off[j]) return fail(M4G_ERR_BAD_PARAM, "block_off not ascending");
    if (matches && !borrowed)
      for (size_t j = 0; j < njobs; j++) {
        uint64_t pos = 0;
        for (uint64_t k = off[j]; k < off[j + 1]; k++) {
          pos += (uint64_t)matches[k].unmatched + matches[k].length;
          if (pos > lens[j]) return fail(M4G_ERR_BAD_PARAM, "a Match list runs past the end of its block");
        }
      }
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    if (borrowed) {
      d_src = dev_src, d_m = dev_matches;
    } else if (hipMalloc(&d_src, n + 64) != hipSuccess || hipMalloc(&d_m, (nmatches + 1) * sizeof(m4g_match)) != hipSuccess) {
      return fail(M4G_ERR_NO_MEMORY, "hipMalloc failed");
    }
    if (hipMalloc(&d_off, (njobs + 1) * 8) != hipSuccess || hipMalloc(&d_jobs, njobs * sizeof(Job)) != hipSuccess)
      return fail(M4G_ERR_NO_MEMORY, "hipMalloc failed");
    if (n && !borrowed) HIP_TRY(hipMemcpyAsync(d_src, src, n, hipMemcpyHostToDevice, st));
    if (nmatches && !borrowed) HIP_TRY(hipMemcpyAsync(d_m, matches, nmatches * sizeof(m4g_match), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_off, off.data(), (njobs + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), njobs * sizeof(Job), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st)); // (the host buffers above may be pageable)
    return M4G_OK;
  }
};
} // namespace

extern "C" int64_t m4g_batch_encode(m4g_batch *b, const uint8_t *d_src, size_t nchunks, const m4g_match *d_out,
                                    const uint64_t *d_out_off, const m4g_histo *d_histos, const m4g_code *d_codes,
                                    int flags, uint8_t *stream_out, size_t stream_cap, void *hip_stream, float *ms_out) {
  if (!b || !d_out_off || !d_histos || !d_codes || (nchunks && (!d_src || !d_out)))
    return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (b->last_off.size() != nchunks + 1)
    return fail(M4G_ERR_BAD_PARAM, "m4g_batch_encode: run the plan on these chunks first (m4g_batch_run_device_codes)");
  if (int rc = batch_settle(b)) return rc;
  std::vector<uint64_t> lens(nchunks);
  for (size_t c = 0; c < nchunks; c++) lens[c] = b->last_off[c + 1] - b->last_off[c];
  return brotli_encode_jobs(b->enc, b->ev[0], b->ev[1], d_src, b->W.d_jobs, lens, d_out, d_out_off, d_histos, d_codes,
                            flags, stream_out, stream_cap, static_cast<hipStream_t>(hip_stream), ms_out, "m4g_batch_encode");
}

// matchfinder.Writer{MatchFinder: M4, Encoder: &brotli.Encoder{}, BlockSize} over one whole input, given the Match
// lists of its blocks (m4g_find_matches_blocks): include/m4g.h.
namespace {
// Both passes of Encoder.Encode over the blocks of U.  matches: the host copy of the Match lists, or NULL when they
// exist on the device only (then the long literal runs are listed by a kernel).
int64_t brotli_encode_uploaded(BlockUpload &U, const m4g_match *matches, size_t n, uint8_t *stream_out, size_t stream_cap,
                               const char *what) {
  const size_t njobs = U.jobs.size();
  // pass 1 of Encoder.Encode for every block (K5); literal runs of K5_HUGE bytes or more go to the side kernel,
  // whose list K4 builds in the batch pipeline and the host (or kb_long_runs) builds here
  std::vector<LongRun> runs;
  uint32_t max_len = 0;
  for (size_t j = 0; j < njobs; j++) {
    uint32_t pos = 0;
    max_len = std::max(max_len, U.jobs[j].len);
    if (!matches) continue;
    for (uint64_t k = U.off[j]; k < U.off[j + 1]; k++) {
      if (matches[k].length == 1) // no copy length code for 1 (command.go:128); M4 never returns it with MinLength >= 2
        return fail(M4G_ERR_BAD_PARAM, std::string(what) + ": a match of Length 1");
      // the Encoder's distance histogram has 64 symbols (encoder.go:37): symbol 16 + 2 (nbits - 1) + prefix < 64 means
      // Distance + 3 < 2^26 (the reference indexes past its array beyond that and panics)
      if (matches[k].length != 0 && (((uint64_t)matches[k].distance + 3) >> 26) != 0)
        return fail(M4G_ERR_BAD_PARAM, std::string(what) + ": a Distance beyond the Encoder's 64 distance symbols");
      if (matches[k].unmatched >= K5_HUGE) runs.push_back(LongRun{(uint32_t)j, pos, matches[k].unmatched, 0});
      pos += matches[k].unmatched + matches[k].length;
    }
  }
  struct Dev {
    uint32_t *histos = nullptr;
    CodeRec *codes = nullptr;
    LongRun *runs = nullptr;
    uint32_t *nruns = nullptr;
    EncBuffers B;
    ~Dev() {
      (void)hipFree(histos);
      (void)hipFree(codes);
      (void)hipFree(runs);
      (void)hipFree(nruns);
      B.release();
    }
  } D;
  const uint32_t run_cap = matches ? (uint32_t)runs.size() : (uint32_t)(n / K5_HUGE + njobs + 1);
  const uint32_t nruns = (uint32_t)runs.size();
  if (hipMalloc(&D.histos, njobs * sizeof(m4g_histo)) != hipSuccess ||
      hipMalloc(&D.codes, (U.nmatches + 1) * sizeof(CodeRec)) != hipSuccess ||
      hipMalloc(&D.runs, ((size_t)run_cap + 1) * sizeof(LongRun)) != hipSuccess || hipMalloc(&D.nruns, 4) != hipSuccess)
    return fail(M4G_ERR_NO_MEMORY, "hipMalloc failed");
  if (nruns) HIP_TRY(hipMemcpyAsync(D.runs, runs.data(), runs.size() * sizeof(LongRun), hipMemcpyHostToDevice, U.st));
  HIP_TRY(hipMemcpyAsync(D.nruns, &nruns, 4, hipMemcpyHostToDevice, U.st));
  if (!matches && U.nmatches) {
    hipLaunchKernelGGL(kb_long_runs, dim3((unsigned)((njobs + 63) / 64)), dim3(64), 0, U.st, U.d_m, U.d_off, (uint32_t)njobs,
                       D.runs, D.nruns, run_cap);
    HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(k5_histograms<true>, dim3((unsigned)njobs), dim3(K5_THREADS), 0, U.st, U.d_src,
                     reinterpret_cast<const uint32_t *>(U.d_m), U.d_jobs, U.d_off, (uint64_t)U.nmatches, D.histos, D.codes,
                     static_cast<const uint32_t *>(nullptr));
  HIP_TRY(hipGetLastError());
  const uint32_t segs = (uint32_t)(((size_t)max_len + K5B_SEG - 1) / K5B_SEG);
  if (run_cap) {
    hipLaunchKernelGGL(k5b_long_literals, dim3(2048), dim3(K5B_THREADS), 0, U.st, U.d_src, U.d_jobs, D.runs, D.nruns,
                       run_cap, segs, U.d_off, (uint64_t)U.nmatches, D.histos, static_cast<const uint32_t *>(nullptr));
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(U.st)); // (`nruns` and `runs` are stack / pageable)
  return brotli_encode_jobs(D.B, U.e0, U.e1, U.d_src, U.d_jobs, U.lens, U.d_m, U.d_off,
                            reinterpret_cast<const m4g_histo *>(D.histos), reinterpret_cast<const m4g_code *>(D.codes), 3,
                            stream_out, stream_cap, U.st, nullptr, what);
}
} // namespace

extern "C" int64_t m4g_brotli_encode_blocks(int device, const uint8_t *src, size_t n, size_t block_size,
                                            const m4g_match *matches, const uint64_t *block_off, uint8_t *stream_out,
                                            size_t stream_cap) {
  BlockUpload U;
  if (int rc = U.init(device, src, n, block_size, matches, block_off)) return rc;
  return brotli_encode_uploaded(U, matches, n, stream_out, stream_cap, "m4g_brotli_encode_blocks");
}

// ===========================================================================
// GPU flate / gzip encoder: huffmanBitWriter.Encode (flate/huffman_bit_writer.go:
// 506-517) or gzipEncoder.Encode (flate/gzip.go:39-65) for every chunk of a batch
// (include/m4g.h, m4g_batch_flate_encode).  k5f counts the symbols, the host
// builds the codes and lays the stream out (flate_host.h), k6f writes the tokens.
// ===========================================================================
namespace {
// Jobs = deflate blocks in stream order (Job.src_off / start / len say where a block's bytes are; d_out_off[j]
// .. d_out_off[j + 1] its matches).  last_is_final: the last job is the stream's last block (eof set, then
// flush and, for gzip, the trailer); otherwise flags bit 1 appends the empty last block of Writer.Close().
int64_t flate_encode_jobs(FlateBuffers &B, hipEvent_t ev0, hipEvent_t ev1, const uint8_t *d_src, const Job *d_jobs,
                          const std::vector<uint64_t> &lens, const m4g_match *d_out, const uint64_t *d_out_off, int flags,
                          bool last_is_final, uint32_t gzip_mtime, uint8_t *stream_out, size_t stream_cap,
                          hipStream_t st, float *ms_out, uint8_t *kinds_out, const char *what,
                          unsigned start_bits = 0, uint64_t *end_bits_out = nullptr, GzipCarry *carry = nullptr) {
  // start_bits / end_bits_out: as brotli_encode_jobs.  carry: CRC-32 and length of the stream's earlier bytes;
  // this call's bytes are joined onto it (gzip only).
  using namespace m4g_flate;
  static_assert(kTab == KF_TAB, "table strides");
  const size_t njobs = lens.size();
  const auto t_begin = std::chrono::steady_clock::now();
  const bool gz = (flags & 4) != 0;
  static const FixedCodes fx;
  static const Crc32Join cj;
  if (njobs > B.jobs_cap) {
    (void)hipFree(B.tables);
    (void)hipFree(B.pos);
    (void)hipFree(B.bits);
    (void)hipFree(B.mode);
    B.tables = nullptr;
    B.pos = B.bits = nullptr;
    B.mode = nullptr;
    B.jobs_cap = 0;
    if (hipMalloc(&B.tables, njobs * (size_t)KF_TAB * 4) != hipSuccess || hipMalloc(&B.pos, njobs * 8) != hipSuccess ||
        hipMalloc(&B.bits, njobs * 8) != hipSuccess || hipMalloc(&B.mode, njobs * 4) != hipSuccess)
      return fail(M4G_ERR_NO_MEMORY, "hipMalloc of the encoder tables failed");
    B.jobs_cap = njobs;
  }
  // pass 1 (makeStatistics) on the GPU; with gzip, the CRC-32 of every 4 KiB segment beside it
  std::vector<uint32_t> stats(njobs * (size_t)KF_TAB);
  std::vector<uint32_t> seg_crc;
  uint64_t max_len = 0;
  for (size_t c = 0; c < njobs; c++) max_len = std::max(max_len, lens[c]);
  const uint32_t segs = (uint32_t)((max_len + KC_SEG - 1) / KC_SEG);
  if (njobs > 0) {
    hipLaunchKernelGGL(k5f_flate_stats, dim3((unsigned)njobs), dim3(KF_THREADS), 0, st, d_src,
                       reinterpret_cast<const uint32_t *>(d_out), d_jobs, d_out_off, B.tables);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(stats.data(), B.tables, stats.size() * 4, hipMemcpyDeviceToHost, st));
    if (gz && segs > 0) {
      const size_t nseg = njobs * (size_t)segs;
      if (nseg > B.crc_cap) {
        (void)hipFree(B.crc);
        B.crc = nullptr;
        B.crc_cap = 0;
        if (hipMalloc(&B.crc, nseg * 4) != hipSuccess) return fail(M4G_ERR_NO_MEMORY, "hipMalloc of the CRC buffer failed");
        B.crc_cap = nseg;
      }
      hipLaunchKernelGGL(k7_crc32_segments, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, st, d_src, d_jobs,
                         (int)njobs, segs, B.crc);
      HIP_TRY(hipGetLastError());
      seg_crc.resize(nseg);
      HIP_TRY(hipMemcpyAsync(seg_crc.data(), B.crc, nseg * 4, hipMemcpyDeviceToHost, st));
    }
  }
  HIP_TRY(hipStreamSynchronize(st));
  for (size_t c = 0; c < njobs; c++)
    if (stats[c * KF_TAB + KF_BAD])
      return fail(M4G_ERR_TOKEN_RANGE, std::string(what) + ": chunk " + std::to_string(c) + " holds " +
                                           std::to_string(stats[c * KF_TAB + KF_BAD]) +
                                           " match(es) outside Length 3..258 / Distance 1..32768; the reference's "
                                           "lengthCode / offsetCode panic there (flate/token.go:59-71)");
  // per block: codes, block kind, header bits, token bits -- on host threads
  std::vector<ChunkPlan> plans(njobs);
  std::vector<uint32_t> tables(njobs * (size_t)KF_TAB);
  {
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned nt = (unsigned)std::min<size_t>(hw, (njobs + 31) / 32 + 1);
    auto work = [&](unsigned t) {
      for (size_t c = t; c < njobs; c += nt) {
        plan_chunk(stats.data() + c * KF_TAB, (size_t)lens[c], last_is_final && c + 1 == njobs, tables.data() + c * KF_TAB,
                   plans[c], fx);
        plans[c].header.flush();
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  const auto t_planned = std::chrono::steady_clock::now();
  // bit positions: one bit writer for the whole stream (its pending bits carry over between Encode calls)
  std::vector<uint64_t> hdr_pos(njobs), body_pos(njobs);
  std::vector<uint32_t> mode(njobs);
  uint64_t pos = start_bits + (((flags & 1) && gz) ? 80 : 0); // gzip header (gzip.go:40-53)
  for (size_t c = 0; c < njobs; c++) {
    hdr_pos[c] = pos;
    pos += plans[c].header.total;
    mode[c] = plans[c].kind == kStored ? 0u : 1u;
    if (plans[c].kind == kStored) pos = ((pos + 7) & ~7ull) + 32; // flush, LEN, NLEN (huffman_bit_writer.go:340-349)
    body_pos[c] = pos;
    pos += plans[c].token_bits + plans[c].eob.len;
    if (kinds_out) kinds_out[c] = (uint8_t)plans[c].kind;
  }
  const uint64_t end_pos = pos;
  const bool ends = last_is_final || (flags & 2);
  if (ends) {
    if (!last_is_final) pos += 3 + 7; // Writer.Close() with nothing buffered: an empty last block, which comes out
                                      // as a fixed block (3 header bits + the 7-bit end-of-block code)
    pos = (pos + 7) & ~7ull;          // flush() (huffman_bit_writer.go:510-512)
    if (gz) pos += 64;                // CRC-32 and length (gzip.go:60-63)
  }
  if (end_bits_out) *end_bits_out = pos;
  const uint64_t nbytes = (pos + 7) / 8;
  if (nbytes > stream_cap) return -(int64_t)(nbytes + M4G_STREAM_NEEDED_BIAS);
  if (!stream_out && nbytes) return fail(M4G_ERR_BAD_PARAM, "stream_out is NULL");
  const size_t nwords = (size_t)(nbytes / 4 + 2);
  if (njobs > 0) {
    if (nwords > B.words_cap) {
      (void)hipFree(B.words);
      B.words = nullptr;
      B.words_cap = 0;
      if (hipMalloc(&B.words, nwords * 4) != hipSuccess) return fail(M4G_ERR_NO_MEMORY, "hipMalloc of the stream failed");
      B.words_cap = nwords;
    }
    HIP_TRY(hipMemsetAsync(B.words, 0, nwords * 4, st));
    HIP_TRY(hipMemcpyAsync(B.tables, tables.data(), tables.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(B.pos, body_pos.data(), njobs * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(B.mode, mode.data(), njobs * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(k6f_flate_bodies, dim3((unsigned)njobs), dim3(KF_THREADS), 0, st, d_src,
                       reinterpret_cast<const uint32_t *>(d_out), d_jobs, d_out_off, B.tables, B.pos, B.mode, B.words, B.bits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, st));
    std::vector<uint64_t> written(njobs);
    HIP_TRY(hipMemcpyAsync(written.data(), B.bits, njobs * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(stream_out, B.words, nbytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t c = 0; c < njobs; c++)
      if (written[c] != plans[c].token_bits)
        return fail(M4G_ERR_INTERNAL, std::string(what) + ": a block's token bits differ from the prediction of its "
                                                          "symbol counts (the match lists do not belong to these bytes?)");
    if (ms_out) HIP_TRY(hipEventElapsedTime(&ms_out[1], ev0, ev1));
  } else if (nbytes) {
    std::memset(stream_out, 0, nbytes);
  }
  // the host-written fields share bytes with the tokens: OR them in
  if ((flags & 1) && gz) {
    const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, (uint8_t)gzip_mtime, (uint8_t)(gzip_mtime >> 8), (uint8_t)(gzip_mtime >> 16),
                             (uint8_t)(gzip_mtime >> 24), 0, 255};
    m4g_enc::or_bits(stream_out, start_bits, hdr, 80);
  }
  for (size_t c = 0; c < njobs; c++) {
    const ChunkPlan &P = plans[c];
    m4g_enc::or_bits(stream_out, hdr_pos[c], P.header.dst.data(), P.header.total);
    if (P.kind == kStored) {
      const uint32_t len = (uint32_t)lens[c];
      const uint8_t f[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)~len, (uint8_t)(~len >> 8)};
      m4g_enc::or_bits(stream_out, body_pos[c] - 32, f, 32);
    } else {
      const uint8_t e[2] = {(uint8_t)P.eob.code, (uint8_t)(P.eob.code >> 8)};
      m4g_enc::or_bits(stream_out, body_pos[c] + P.token_bits, e, P.eob.len);
    }
  }
  GzipCarry run = carry ? *carry : GzipCarry{};
  if (gz) { // CRC-32 of the stream so far (gzip.go:57-58)
    const uint32_t xp_seg = cj.x_pow_8n(KC_SEG);
    for (size_t c = 0; c < njobs; c++) {
      const uint64_t len = lens[c];
      for (uint64_t s0 = 0, k = 0; s0 < len; s0 += KC_SEG, k++) {
        const uint64_t sl = std::min<uint64_t>(KC_SEG, len - s0);
        const uint32_t sc = seg_crc[c * (size_t)segs + k];
        run.crc = sl == KC_SEG ? (Crc32Join::mul(xp_seg, run.crc) ^ sc) : cj.join(run.crc, sc, sl);
      }
      run.total += len;
    }
    if (carry) *carry = run;
  }
  if (ends) {
    if (!last_is_final) {
      const uint8_t fin[2] = {3, 0}; // BFINAL = 1, BTYPE = 01; the fixed end-of-block code is 7 zero bits
      m4g_enc::or_bits(stream_out, end_pos, fin, 10);
    }
    if (gz) {
      const uint32_t crc = run.crc;
      const uint32_t len32 = (uint32_t)run.total; // gzip.go:57: a uint32 that wraps
      const uint8_t tr[8] = {(uint8_t)crc,   (uint8_t)(crc >> 8),   (uint8_t)(crc >> 16),   (uint8_t)(crc >> 24),
                             (uint8_t)len32, (uint8_t)(len32 >> 8), (uint8_t)(len32 >> 16), (uint8_t)(len32 >> 24)};
      std::memcpy(stream_out + nbytes - 8, tr, 8);
    }
  }
  if (ms_out) {
    ms_out[0] = std::chrono::duration<float, std::milli>(t_planned - t_begin).count();
    ms_out[2] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (njobs == 0) ms_out[1] = 0.f;
  }
  return (int64_t)nbytes;
}
} // namespace

extern "C" int64_t m4g_batch_flate_encode(m4g_batch *b, const uint8_t *d_src, size_t nchunks, const m4g_match *d_out,
                                          const uint64_t *d_out_off, int flags, uint32_t gzip_mtime,
                                          uint8_t *stream_out, size_t stream_cap, void *hip_stream, float *ms_out,
                                          uint8_t *kinds_out) {
  if (!b || !d_out_off || (nchunks && (!d_src || !d_out))) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (b->last_off.size() != nchunks + 1)
    return fail(M4G_ERR_BAD_PARAM, "m4g_batch_flate_encode: run the plan on these chunks first (m4g_batch_run_device)");
  if (int rc = batch_settle(b)) return rc;
  std::vector<uint64_t> lens(nchunks);
  for (size_t c = 0; c < nchunks; c++) lens[c] = b->last_off[c + 1] - b->last_off[c];
  return flate_encode_jobs(b->flate, b->ev[0], b->ev[1], d_src, b->W.d_jobs, lens, d_out, d_out_off, flags, false,
                           gzip_mtime, stream_out, stream_cap, static_cast<hipStream_t>(hip_stream), ms_out, kinds_out,
                           "m4g_batch_flate_encode");
}

// flate.NewWriter(dst, level) / NewGZIPWriter over one whole input, given the Match lists of its blocks
// (m4g_find_matches_blocks): include/m4g.h.
extern "C" int64_t m4g_flate_encode_blocks(int device, const uint8_t *src, size_t n, size_t block_size,
                                           const m4g_match *matches, const uint64_t *block_off, int gzip,
                                           uint32_t gzip_mtime, uint8_t *stream_out, size_t stream_cap,
                                           uint8_t *kinds_out) {
  BlockUpload U;
  if (int rc = U.init(device, src, n, block_size, matches, block_off)) return rc;
  struct Dev {
    FlateBuffers B;
    ~Dev() { B.release(); }
  } D;
  return flate_encode_jobs(D.B, U.e0, U.e1, U.d_src, U.d_jobs, U.lens, U.d_m, U.d_off, 1 | (gzip ? 4 : 0), true, gzip_mtime,
                           stream_out, stream_cap, U.st, nullptr, kinds_out, "m4g_flate_encode_blocks");
}

// ===========================================================================
// The entry point that allocates nothing (include/m4g.h,
// m4g_batch_run_device_noalloc): the batch pipeline with its workspace in a
// file-scope device array instead of a plan's hipMalloc'ed buffers.
// ===========================================================================
namespace {
constexpr size_t NA_MAX_CHUNKS = M4G_NOALLOC_MAX_CHUNKS;
constexpr size_t NA_MAX_BYTES = M4G_NOALLOC_MAX_BYTES;
constexpr size_t NA_MAX_BLOCKS = M4G_NOALLOC_MAX_BLOCKS;           // (chunk, sub-block) pairs
constexpr size_t NA_MAX_RECS = NA_MAX_BLOCKS * (1024 / 4 + 18);    // records per RS / ST array
constexpr size_t NA_LONG_CAP = NA_MAX_CHUNKS + NA_MAX_BYTES / K5_HUGE + 64;
constexpr size_t na_align(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr size_t NA_OFF_CHAIN = 0;
constexpr size_t NA_OFF_RS = NA_OFF_CHAIN + na_align((NA_MAX_BYTES + 16 * NA_MAX_CHUNKS + 1024) * 4);
constexpr size_t NA_OFF_ST = NA_OFF_RS + na_align(NA_MAX_RECS * sizeof(Rec));
constexpr size_t NA_OFF_E = NA_OFF_ST + na_align(NA_MAX_RECS * sizeof(Rec));
constexpr size_t NA_OFF_TOUT = NA_OFF_E + na_align(NA_MAX_BLOCKS * sizeof(ParseState));
constexpr size_t NA_OFF_DESC = NA_OFF_TOUT + na_align(NA_MAX_BLOCKS * sizeof(ParseState));
constexpr size_t NA_OFF_COUNTS = NA_OFF_DESC + na_align(NA_MAX_BLOCKS * sizeof(BlockDesc));
constexpr size_t NA_OFF_ERRORS = NA_OFF_COUNTS + na_align(NA_MAX_CHUNKS * 4);
constexpr size_t NA_OFF_JOBS = NA_OFF_ERRORS + na_align(64);
constexpr size_t NA_OFF_LONG = NA_OFF_JOBS + na_align(NA_MAX_CHUNKS * sizeof(Job));
constexpr size_t NA_SCRATCH_BYTES = NA_OFF_LONG + na_align(NA_LONG_CAP * sizeof(LongRun));
static_assert(NA_SCRATCH_BYTES <= (size_t)16 << 20, "the scratch array is meant to stay modest");
} // namespace

// Every GPU the library runs on holds its own copy of this array, at its own
// address, until the process ends.  In host code the NAME below is only a
// stand-in for the symbol: the device address comes from hipGetSymbolAddress
// (na_scratch_ptr), and host code uses that pointer and nothing else.
static __device__ __align__(256) unsigned char scratch[NA_SCRATCH_BYTES];

namespace {
struct NaOffsets {
  uint64_t off[NA_MAX_CHUNKS + 1];
};
// Job descriptors from chunk offsets passed by value (no host-to-device copy, so
// the entry point also works under stream capture): same layout rule as
// fill_batch_jobs.
__global__ __launch_bounds__(64) void kna_make_jobs(NaOffsets o, int n, Job *__restrict__ jobs) {
  const int c = threadIdx.x;
  if (c >= n) return;
  uint64_t chain_off = 0;
  for (int k = 0; k < c; k++) chain_off += (o.off[k + 1] - o.off[k] + 15) & ~(uint64_t)15;
  Job J{};
  J.src_off = o.off[c];
  J.chain_off = chain_off;
  J.len = (uint32_t)(o.off[c + 1] - o.off[c]);
  jobs[c] = J;
}

std::atomic<unsigned char *> g_na_ptr[64]; // per device: address of its copy of `scratch`
std::mutex g_na_mutex;

// One-time set-up per device (the device is current): the array's address on
// this device, the K1 ordering self-test (its result word lives in the scratch
// too) and the kernels' LDS attributes.  Synchronises; later calls do not.
int na_scratch_ptr(int device, unsigned char **out) {
  if (device < 0 || device >= 64) return fail(M4G_ERR_NO_DEVICE, "device index out of range");
  unsigned char *p = g_na_ptr[device].load();
  if (!p) {
    std::lock_guard<std::mutex> lock(g_na_mutex);
    p = g_na_ptr[device].load();
    if (!p) {
      if (int rc = check_device(device)) return rc;
      void *sym = nullptr;
      HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));
      if (!sym) return fail(M4G_ERR_HIP, "hipGetSymbolAddress returned a null address for the scratch array");
      if (int rc = k1_prepare(reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(sym) + NA_OFF_ERRORS))) return rc;
      p = static_cast<unsigned char *>(sym);
      g_na_ptr[device].store(p);
    }
  }
  *out = p;
  return M4G_OK;
}
} // namespace

// Test hook: fill this device's copy of the scratch array with `value` (no allocation: a memset on the symbol's
// address), so that a test can show that a call reads nothing an earlier call left behind.
extern "C" int m4g_debug_noalloc_poison(int device, int value, void *hip_stream) {
  HIP_TRY(hipSetDevice(device));
  unsigned char *base = nullptr;
  if (int rc = na_scratch_ptr(device, &base)) return rc;
  HIP_TRY(hipMemsetAsync(base, value, NA_SCRATCH_BYTES, static_cast<hipStream_t>(hip_stream)));
  return M4G_OK;
}

extern "C" int m4g_batch_run_device_noalloc(const m4g_params *params, int device, const uint8_t *d_src,
                                            const uint64_t *chunk_off, size_t nchunks, m4g_match *d_out,
                                            uint64_t *d_out_off, size_t out_cap, m4g_histo *d_histos,
                                            void *hip_stream) {
  Params p;
  if (int rc = resolve_params(params, &p)) return rc;
  if (!chunk_off || !d_out_off || (!d_out && out_cap)) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (nchunks && !d_src) return fail(M4G_ERR_BAD_PARAM, "d_src is NULL");
  if (d_histos && p.min_length < 2)
    return fail(M4G_ERR_BAD_PARAM, "histograms need MinLength >= 2 (brotli has no copy length code for 1)");
  if (nchunks > NA_MAX_CHUNKS)
    return fail(M4G_ERR_TOO_LARGE, "m4g_batch_run_device_noalloc: more than M4G_NOALLOC_MAX_CHUNKS chunks");
  size_t max_len = 1;
  for (size_t c = 0; c < nchunks; c++) {
    if (chunk_off[c + 1] < chunk_off[c]) return fail(M4G_ERR_BAD_PARAM, "chunk_off not ascending");
    max_len = std::max<size_t>(max_len, chunk_off[c + 1] - chunk_off[c]);
  }
  if (nchunks && chunk_off[nchunks] - chunk_off[0] > NA_MAX_BYTES)
    return fail(M4G_ERR_TOO_LARGE, "m4g_batch_run_device_noalloc: more than M4G_NOALLOC_MAX_BYTES bytes in one call");
  // layout of the record slots for THIS call (the plan fixes it at create time)
  SpecLayout L{};
  L.sub_block = kDefaultSubBlock;
  L.kmax = (uint32_t)((max_len + (size_t)L.sub_block - 1) / (size_t)L.sub_block);
  L.slot = (uint32_t)((size_t)L.sub_block / (size_t)p.min_length + 18);
  const size_t blocks = nchunks * L.kmax;
  if (blocks > NA_MAX_BLOCKS || blocks * L.slot > NA_MAX_RECS)
    return fail(M4G_ERR_TOO_LARGE,
                "m4g_batch_run_device_noalloc: nchunks x ceil(largest chunk / 1024) exceeds M4G_NOALLOC_MAX_BLOCKS "
                "(or MinLength < 4 needs more record slots than the scratch array holds)");
  HIP_TRY(hipSetDevice(device));
  unsigned char *base = nullptr;
  if (int rc = na_scratch_ptr(device, &base)) return rc;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (nchunks == 0) {
    HIP_TRY(hipMemsetAsync(d_out_off, 0, sizeof(uint64_t), st));
    return M4G_OK;
  }
  BatchWork W;
  W.d_chain = reinterpret_cast<uint32_t *>(base + NA_OFF_CHAIN);
  W.w.L = L;
  W.w.blocks = blocks;
  W.w.rs = reinterpret_cast<Rec *>(base + NA_OFF_RS);
  W.w.st = reinterpret_cast<Rec *>(base + NA_OFF_ST);
  W.w.E = reinterpret_cast<ParseState *>(base + NA_OFF_E);
  W.w.Tout = reinterpret_cast<ParseState *>(base + NA_OFF_TOUT);
  W.w.desc = reinterpret_cast<BlockDesc *>(base + NA_OFF_DESC);
  W.d_counts = reinterpret_cast<uint32_t *>(base + NA_OFF_COUNTS);
  W.d_errors = reinterpret_cast<uint32_t *>(base + NA_OFF_ERRORS);
  W.d_jobs = reinterpret_cast<Job *>(base + NA_OFF_JOBS);
  W.d_long_runs = reinterpret_cast<LongRun *>(base + NA_OFF_LONG);
  W.long_cap = (uint32_t)NA_LONG_CAP;
  // What must start at zero in the shared array: the error / counter words (the
  // pipeline clears them itself) and nothing else -- every other word is written
  // by the call before it is read (K1 writes the chain entry of every position
  // the parse can reach: chunks start at position 0, and entries of the seven
  // uninserted positions at a chunk's end are never read).
  NaOffsets o;
  for (size_t c = 0; c <= nchunks; c++) o.off[c] = chunk_off[c];
  for (size_t c = nchunks + 1; c <= NA_MAX_CHUNKS; c++) o.off[c] = chunk_off[nchunks];
  hipLaunchKernelGGL(kna_make_jobs, dim3(1), dim3(64), 0, st, o, (int)nchunks, W.d_jobs);
  HIP_TRY(hipGetLastError());
  return run_batch_pipeline(p, W, (uint32_t)max_len, d_src, chunk_off[nchunks], (int)nchunks, d_out, d_out_off, out_cap,
                            d_histos, nullptr, st, nullptr);
}

// Profiling aid (not part of include/m4g.h): runs K1 alone `reps` times on the
// job list of the last m4g_batch_run_device call and returns the average ms.
// dbg bit 0: skip the chain stores; bit 1: hashing + compaction only.
extern "C" int m4g_debug_k1_only(m4g_batch *b, const uint8_t *d_src, uint64_t src_total, int njobs,
                                 int dbg, int reps, float *ms_out) {
  if (!b || !ms_out || njobs <= 0 || (size_t)njobs > b->max_chunks) return fail(M4G_ERR_BAD_PARAM, "bad argument");
  HIP_TRY(hipSetDevice(b->device));
  const int pb = part_bits_of(b->p);
  const int nparts = 1 << (b->p.table_bits - pb);
  dim3 g1((unsigned)(((njobs + 7) / 8) * 8 * nparts));
  HIP_TRY(hipEventRecord(b->ev[0], nullptr));
  for (int r = 0; r < reps; r++)
    if (dbg < 0) { // the variant a plan run would use (split K1 when applicable)
      if (int rc = launch_k1(b->p, d_src, src_total, (uint32_t)b->max_chunk_bytes, b->W.d_chain, b->W.d_jobs, nullptr,
                             njobs, nullptr, nullptr, nullptr, &b->W.split))
        return rc;
    } else if (k1_use_sort())
      hipLaunchKernelGGL(k1_chain_build<false>, g1, dim3(K1_THREADS), k1_lds_bytes(pb), nullptr, d_src, src_total,
                         b->W.d_chain, b->W.d_jobs, nullptr, njobs, b->p.hash_len, b->p.table_bits, pb, 5, dbg, nullptr, nullptr);
    else
      hipLaunchKernelGGL(k1x_chain_build<false>, g1, dim3(KX_THREADS), k1x_lds_bytes(pb), nullptr, d_src, src_total,
                         b->W.d_chain, b->W.d_jobs, nullptr, njobs, b->p.hash_len, b->p.table_bits, pb, 5, dbg, nullptr, nullptr, 0u, 0u, 0u, 0u, 7u);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b->ev[1], nullptr));
  HIP_TRY(hipEventSynchronize(b->ev[1]));
  HIP_TRY(hipEventElapsedTime(ms_out, b->ev[0], b->ev[1]));
  *ms_out /= (float)reps;
  return M4G_OK;
}

// Test hooks (not part of include/m4g.h).  m4g_debug_set_k1_variant: -1 = automatic
// choice, 0 = sorting kernel, 1 = exchange kernel, for every later launch of the
// process.  m4g_debug_k1_run: K1 alone over the job list of the plan's last run,
// chain entries written to the caller's buffer of m4g_debug_chain_words(b) words
// (so a test can compare the two kernels' chain arrays on the device).
extern "C" void m4g_debug_set_k1_variant(int v) { g_k1_force.store(v < 0 ? -1 : (v > 2 ? 1 : v)); }
extern "C" uint64_t m4g_debug_chain_words(m4g_batch *b) {
  return b ? (uint64_t)(b->max_chunks * b->max_chunk_bytes + 16 * b->max_chunks) + 1024 : 0;
}
extern "C" int m4g_debug_k1_run(m4g_batch *b, const uint8_t *d_src, uint64_t src_total, int njobs,
                                uint32_t *d_chain_out, void *hip_stream) {
  if (!b || !d_chain_out || njobs <= 0 || (size_t)njobs > b->max_chunks || b->last_off.size() != (size_t)njobs + 1)
    return fail(M4G_ERR_BAD_PARAM, "bad argument (run the plan on these chunks first)");
  HIP_TRY(hipSetDevice(b->device));
  return launch_k1(b->p, d_src, src_total, (uint32_t)b->max_chunk_bytes, d_chain_out, b->W.d_jobs, nullptr, njobs,
                   static_cast<hipStream_t>(hip_stream), nullptr, nullptr, &b->W.split);
}

// Diagnostic (not part of include/m4g.h): which K1 kernel the device gets --
// 1 = ordered LDS exchange (k1_xchg.h), 0 = sorting kernel.
extern "C" int m4g_debug_k1_variant(int device) {
  int rc = check_device(device);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  if ((rc = k1_prepare())) return rc;
  return k1_use_sort() ? 0 : 1;
}

extern "C" int m4g_batch_last_timings(m4g_batch *b, float *ms, int cap) {
  if (!b || !ms || !b->timing || !b->ran) return 0;
  if (hipSetDevice(b->device) != hipSuccess) return 0;
  if (hipEventSynchronize(b->ev[4]) != hipSuccess) return 0;
  int n = std::min(cap, 4);
  for (int k = 0; k < n; k++)
    if (hipEventElapsedTime(&ms[k], b->ev[k], b->ev[k + 1]) != hipSuccess) return k;
  return n;
}

// ===========================================================================
// host-memory batch over several devices
// ===========================================================================
namespace {
struct DevPart {
  int device = 0;
  std::vector<size_t> chunks; // global chunk indices, ascending
  std::vector<uint64_t> off;  // local offsets (chunks.size()+1)
  m4g_batch *plan = nullptr;
  uint8_t *d_src = nullptr;
  m4g_match *d_out = nullptr;
  uint64_t *d_out_off = nullptr;
  m4g_histo *d_histos = nullptr;
  hipStream_t st = nullptr;
  std::vector<uint64_t> h_out_off;
  size_t out_cap = 0;
};

void free_part(DevPart &P) {
  (void)hipSetDevice(P.device);
  if (P.st) (void)hipStreamSynchronize(P.st);
  m4g_batch_destroy(P.plan);
  (void)hipFree(P.d_src);
  (void)hipFree(P.d_out);
  (void)hipFree(P.d_out_off);
  (void)hipFree(P.d_histos);
  if (P.st) (void)hipStreamDestroy(P.st);
}
} // namespace


namespace {
// One device, many chunks: the batch is cut into slices and the three stages of
// a slice -- upload, kernels, download of its Match lists -- run on three
// streams, so that the copies of one slice overlap the kernels of its
// neighbours.  Results are identical to the unsliced call (chunks are
// independent); only the device-side workspace is smaller.
// Measured on 1 MiB text chunks from page-locked buffers (tools/e2e_probe.py): 512 chunks in slices of 256:
// 24.7 ms against 33.5 ms unsliced; 2048 chunks in slices of 512: 70 ms (30 GB/s); smaller slices lose more in
// the kernels (a slice must still fill the GPU) than the overlap gains.
constexpr size_t kPipeMinChunks = 384;
constexpr size_t kPipeSliceMin = 256;

int64_t batch_one_device_pipelined(const m4g_params *params, const Params &p, const uint8_t *base,
                                   const uint64_t *chunk_off, size_t nchunks, int device, size_t max_bytes,
                                   m4g_match *out, uint64_t *out_off, size_t out_cap, m4g_histo *histos) {
  Trace tr;
  size_t slice = std::max(kPipeSliceMin, (nchunks + 3) / 4);
  if (const char *e = std::getenv("M4G_PIPE_SLICE")) // tuning aid: chunks per slice
    slice = std::max<size_t>(1, (size_t)std::atoll(e));
  const size_t nslices = (nchunks + slice - 1) / slice;
  const uint64_t origin = chunk_off[0], total = chunk_off[nchunks] - origin;

  struct Pipe {
    m4g_batch *plan = nullptr;
    uint8_t *d_src = nullptr;
    m4g_match *d_out = nullptr;
    uint64_t *d_out_off = nullptr;
    m4g_histo *d_histos = nullptr;
    uint64_t *h_out_off = nullptr; // pinned: (slice + 1) per slice
    uint32_t *h_err = nullptr;     // pinned: one word per slice
    hipStream_t st_in = nullptr, st_run = nullptr, st_out = nullptr;
    std::vector<hipEvent_t> ev_in, ev_run;
  } P;
  auto cleanup = [&]() {
    (void)hipSetDevice(device);
    for (hipStream_t st : {P.st_in, P.st_run, P.st_out})
      if (st) (void)hipStreamSynchronize(st);
    m4g_batch_destroy(P.plan);
    (void)hipFree(P.d_src);
    (void)hipFree(P.d_out);
    (void)hipFree(P.d_out_off);
    (void)hipFree(P.d_histos);
    if (P.h_out_off) (void)hipHostFree(P.h_out_off);
    if (P.h_err) (void)hipHostFree(P.h_err);
    for (auto &e : P.ev_in) if (e) (void)hipEventDestroy(e);
    for (auto &e : P.ev_run) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {P.st_in, P.st_run, P.st_out})
      if (st) (void)hipStreamDestroy(st);
  };
  int64_t result = M4G_OK;
#define PIPE_TRY(expr)                                                                 \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      result = fail(M4G_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
      cleanup();                                                                       \
      return result;                                                                   \
    }                                                                                  \
  } while (0)

  PIPE_TRY(hipSetDevice(device));
  PIPE_TRY(hipStreamCreateWithFlags(&P.st_in, hipStreamNonBlocking));
  PIPE_TRY(hipStreamCreateWithFlags(&P.st_run, hipStreamNonBlocking));
  PIPE_TRY(hipStreamCreateWithFlags(&P.st_out, hipStreamNonBlocking));
  int rc = batch_create(params, device, slice, max_bytes, false, &P.plan);
  if (rc) {
    cleanup();
    return rc;
  }
  // offsets relative to the device copy of the input; worst-case output slots per chunk
  std::vector<uint64_t> off(nchunks + 1), cap_off(nchunks + 1);
  cap_off[0] = 0;
  for (size_t c = 0; c <= nchunks; c++) off[c] = chunk_off[c] - origin;
  for (size_t c = 0; c < nchunks; c++) cap_off[c + 1] = cap_off[c] + (off[c + 1] - off[c]) / (size_t)p.min_length + 2;
  PIPE_TRY(hipMalloc(&P.d_src, total + 64));
  PIPE_TRY(hipMalloc(&P.d_out, cap_off[nchunks] * sizeof(m4g_match)));
  PIPE_TRY(hipMalloc(&P.d_out_off, nslices * (slice + 1) * 8));
  if (histos) PIPE_TRY(hipMalloc(&P.d_histos, nchunks * sizeof(m4g_histo)));
  PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(&P.h_out_off), nslices * (slice + 1) * 8, hipHostMallocDefault));
  PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(&P.h_err), nslices * 4, hipHostMallocDefault));
  P.ev_in.assign(nslices, nullptr);
  P.ev_run.assign(nslices, nullptr);
  for (size_t k = 0; k < nslices; k++) {
    PIPE_TRY(hipEventCreateWithFlags(&P.ev_in[k], hipEventDisableTiming));
    PIPE_TRY(hipEventCreateWithFlags(&P.ev_run[k], hipEventDisableTiming));
  }
  tr.mark("pipeline set-up");

  // 1. enqueue every slice: upload on st_in, kernels on st_run behind the upload
  for (size_t k = 0; k < nslices; k++) {
    const size_t c0 = k * slice, c1 = std::min(nchunks, c0 + slice), nc = c1 - c0;
    if (off[c1] > off[c0])
      PIPE_TRY(hipMemcpyAsync(P.d_src + off[c0], base + origin + off[c0], off[c1] - off[c0], hipMemcpyHostToDevice, P.st_in));
    PIPE_TRY(hipEventRecord(P.ev_in[k], P.st_in));
    PIPE_TRY(hipStreamWaitEvent(P.st_run, P.ev_in[k], 0));
    rc = m4g_batch_run_device(P.plan, P.d_src, off.data() + c0, nc, P.d_out + cap_off[c0], P.d_out_off + k * (slice + 1),
                              cap_off[c1] - cap_off[c0], histos ? P.d_histos + c0 : nullptr, P.st_run);
    if (rc) {
      cleanup();
      return rc;
    }
    PIPE_TRY(hipMemcpyAsync(P.h_out_off + k * (slice + 1), P.d_out_off + k * (slice + 1), (nc + 1) * 8,
                            hipMemcpyDeviceToHost, P.st_run));
    PIPE_TRY(hipMemcpyAsync(P.h_err + k, P.plan->W.d_errors, 4, hipMemcpyDeviceToHost, P.st_run));
    PIPE_TRY(hipEventRecord(P.ev_run[k], P.st_run));
  }
  tr.mark("enqueue");

  // 2. as each slice finishes: its chunks' global offsets, then its Match lists straight to their place
  uint64_t run = 0;
  uint32_t err = 0;
  for (size_t k = 0; k < nslices; k++) {
    const size_t c0 = k * slice, c1 = std::min(nchunks, c0 + slice), nc = c1 - c0;
    PIPE_TRY(hipEventSynchronize(P.ev_run[k]));
    err |= P.h_err[k];
    const uint64_t *ho = P.h_out_off + k * (slice + 1);
    for (size_t c = 0; c < nc; c++) out_off[c0 + c] = run + (ho[c] - ho[0]);
    const uint64_t cnt = ho[nc] - ho[0];
    if (!err && cnt && run + cnt <= out_cap)
      PIPE_TRY(hipMemcpyAsync(out + run, P.d_out + cap_off[c0] + ho[0], cnt * sizeof(m4g_match), hipMemcpyDeviceToHost, P.st_out));
    run += cnt;
  }
  out_off[nchunks] = run;
  if (err)
    result = status_from_error_word(err);
  else if (run > out_cap)
    result = -(int64_t)(run + M4G_STREAM_NEEDED_BIAS);
  else if (histos)
    PIPE_TRY(hipMemcpyAsync(histos, P.d_histos, nchunks * sizeof(m4g_histo), hipMemcpyDeviceToHost, P.st_out));
  PIPE_TRY(hipStreamSynchronize(P.st_out));
  tr.mark("kernels + D2H");
  cleanup();
  tr.mark("cleanup");
  return result;
#undef PIPE_TRY
}
} // namespace

extern "C" int64_t m4g_find_matches_batch(const m4g_params *params, const uint8_t *base,
                                          const uint64_t *chunk_off, size_t nchunks,
                                          const int *devices, int ndevices, m4g_match *out,
                                          uint64_t *out_off, size_t out_cap, m4g_histo *histos) {
  Params p;
  int rc = resolve_params(params, &p);
  if (rc) return rc;
  if (!chunk_off || !out_off || (!base && nchunks) || !devices || ndevices < 1)
    return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  out_off[0] = 0;
  if (nchunks == 0) return M4G_OK;
  size_t max_bytes = 1;
  for (size_t c = 0; c < nchunks; c++) {
    if (chunk_off[c + 1] < chunk_off[c]) return fail(M4G_ERR_BAD_PARAM, "chunk_off not ascending");
    max_bytes = std::max<size_t>(max_bytes, chunk_off[c + 1] - chunk_off[c]);
  }
  if (max_bytes > ((size_t)1 << 24)) return fail(M4G_ERR_TOO_LARGE, "chunk above 2^24 bytes");
  for (int g = 0; g < ndevices; g++)
    if ((rc = check_device(devices[g]))) return rc;
  if (ndevices == 1 && (nchunks >= kPipeMinChunks || std::getenv("M4G_PIPE_SLICE")) && !std::getenv("M4G_NO_PIPELINE"))
    return batch_one_device_pipelined(params, p, base, chunk_off, nchunks, devices[0], max_bytes, out, out_off,
                                      out_cap, histos);

  Trace tr;
  // pin the caller's buffers for the duration when possible (faster copies)
  const size_t total = chunk_off[nchunks] - chunk_off[0];
  bool pinned_in = total && !std::getenv("M4G_NO_PIN") && hipHostRegister(const_cast<uint8_t *>(base) + chunk_off[0], total,
                                            hipHostRegisterDefault) == hipSuccess;
  (void)hipGetLastError();

  tr.mark("pin input");
  std::vector<DevPart> parts(ndevices);
  int64_t result = M4G_OK;
  auto cleanup = [&]() {
    for (auto &P : parts) free_part(P);
    if (pinned_in) (void)hipHostUnregister(const_cast<uint8_t *>(base) + chunk_off[0]);
  };
#define BATCH_TRY(expr)                                                                \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      result = fail(M4G_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
      cleanup();                                                                       \
      return result;                                                                   \
    }                                                                                  \
  } while (0)

  // 1. shard: chunk c -> devices[c % ndevices]; upload + run, all devices async
  for (int g = 0; g < ndevices; g++) {
    DevPart &P = parts[g];
    P.device = devices[g];
    P.off.push_back(0);
    for (size_t c = (size_t)g; c < nchunks; c += (size_t)ndevices) {
      P.chunks.push_back(c);
      uint64_t n = chunk_off[c + 1] - chunk_off[c];
      P.off.push_back(P.off.back() + n);
      P.out_cap += n / (size_t)p.min_length + 2;
    }
    if (P.chunks.empty()) continue;
    BATCH_TRY(hipSetDevice(P.device));
    BATCH_TRY(hipStreamCreateWithFlags(&P.st, hipStreamNonBlocking));
    rc = batch_create(params, P.device, P.chunks.size(), max_bytes, false, &P.plan);
    if (rc) {
      cleanup();
      return rc;
    }
    tr.mark("plan create");
    BATCH_TRY(hipMalloc(&P.d_src, P.off.back() + 64));
    BATCH_TRY(hipMalloc(&P.d_out, P.out_cap * sizeof(m4g_match)));
    BATCH_TRY(hipMalloc(&P.d_out_off, (P.chunks.size() + 1) * 8));
    if (histos) BATCH_TRY(hipMalloc(&P.d_histos, P.chunks.size() * sizeof(m4g_histo)));
    if (ndevices == 1) { // the device's chunks are the caller's buffer: one copy
      if (total)
        BATCH_TRY(hipMemcpyAsync(P.d_src, base + chunk_off[0], total, hipMemcpyHostToDevice, P.st));
    } else {
      for (size_t k = 0; k < P.chunks.size(); k++) {
        size_t c = P.chunks[k];
        uint64_t n = chunk_off[c + 1] - chunk_off[c];
        if (n)
          BATCH_TRY(hipMemcpyAsync(P.d_src + P.off[k], base + chunk_off[c], n, hipMemcpyHostToDevice, P.st));
      }
    }
    tr.mark("alloc + H2D enqueue");
    P.h_out_off.resize(P.chunks.size() + 1);
    rc = m4g_batch_run_device(P.plan, P.d_src, P.off.data(), P.chunks.size(), P.d_out, P.d_out_off,
                              P.out_cap, P.d_histos, P.st);
    if (rc) {
      cleanup();
      return rc;
    }
    BATCH_TRY(hipMemcpyAsync(P.h_out_off.data(), P.d_out_off, (P.chunks.size() + 1) * 8,
                             hipMemcpyDeviceToHost, P.st));
  }

  tr.mark("launch");
  // 2. global offsets in chunk order
  for (int g = 0; g < ndevices; g++) {
    DevPart &P = parts[g];
    if (P.chunks.empty()) continue;
    BATCH_TRY(hipSetDevice(P.device));
    BATCH_TRY(hipStreamSynchronize(P.st));
    uint32_t err = 0;
    BATCH_TRY(hipMemcpy(&err, P.plan->W.d_errors, 4, hipMemcpyDeviceToHost));
    if (err) {
      cleanup();
      return status_from_error_word(err);
    }
  }
  tr.mark("kernels done");
  uint64_t run = 0;
  for (size_t c = 0; c < nchunks; c++) {
    DevPart &P = parts[c % (size_t)ndevices];
    size_t k = c / (size_t)ndevices;
    out_off[c] = run;
    run += P.h_out_off[k + 1] - P.h_out_off[k];
  }
  out_off[nchunks] = run;
  if (run > out_cap) {
    cleanup();
    return -(int64_t)(run + M4G_STREAM_NEEDED_BIAS); // (counts above 10^6 are routine here: out of the M4G_ERR_* range)
  }

  // 3. ordered concatenation on the host: each chunk's list lands at out_off[c]
  bool pinned_out = run && !std::getenv("M4G_NO_PIN") &&
                    hipHostRegister(out, run * sizeof(m4g_match), hipHostRegisterDefault) == hipSuccess;
  tr.mark("pin output");
  (void)hipGetLastError();
  for (size_t c = 0; c < nchunks; c++) {
    DevPart &P = parts[c % (size_t)ndevices];
    size_t k = c / (size_t)ndevices;
    uint64_t cnt = P.h_out_off[k + 1] - P.h_out_off[k];
    if (ndevices == 1) { // already in chunk order on the device: one copy
      if (c != 0) break;
      cnt = run;
    }
    if (!cnt) continue;
    hipError_t e = hipSetDevice(P.device);
    if (e == hipSuccess)
      e = hipMemcpyAsync(out + out_off[c], P.d_out + P.h_out_off[k], cnt * sizeof(m4g_match),
                         hipMemcpyDeviceToHost, P.st);
    if (e != hipSuccess) {
      result = fail(M4G_ERR_HIP, std::string("D2H matches: ") + hipGetErrorString(e));
      break;
    }
  }
  if (result == M4G_OK && histos) {
    for (int g = 0; g < ndevices && result == M4G_OK; g++) {
      DevPart &P = parts[g];
      if (P.chunks.empty()) continue;
      std::vector<m4g_histo> tmp(P.chunks.size());
      hipError_t e = hipSetDevice(P.device);
      if (e == hipSuccess) e = hipStreamSynchronize(P.st);
      if (e == hipSuccess)
        e = hipMemcpy(tmp.data(), P.d_histos, tmp.size() * sizeof(m4g_histo), hipMemcpyDeviceToHost);
      if (e != hipSuccess) {
        result = fail(M4G_ERR_HIP, std::string("D2H histograms: ") + hipGetErrorString(e));
        break;
      }
      for (size_t k = 0; k < P.chunks.size(); k++) histos[P.chunks[k]] = tmp[k];
    }
  }
  for (auto &P : parts)
    if (P.st) {
      (void)hipSetDevice(P.device);
      (void)hipStreamSynchronize(P.st);
    }
  tr.mark("D2H");
  if (pinned_out) (void)hipHostUnregister(out);
  cleanup();
  tr.mark("cleanup");
  return result;
#undef BATCH_TRY
}

// ===========================================================================
// persistent host-memory batch (include/m4g.h, m4g_hostbatch_*): the pipeline
// of m4g_find_matches_batch with everything that call sets up and tears down
// every time -- plans, device buffers, page-locked staging, streams, events --
// created once, and one host thread per device.
// ===========================================================================
namespace {
struct HostLane {
  int device = 0;
  size_t share = 0, slice = 0, nslices = 0; // capacity in chunks; chunks per slice; slices reserved
  m4g_batch *plan = nullptr;
  uint8_t *d_src = nullptr;
  m4g_match *d_out = nullptr;
  uint64_t *d_out_off = nullptr;
  m4g_histo *d_histos = nullptr;
  uint64_t *h_out_off = nullptr; // pinned, (slice + 1) per slice
  uint32_t *h_err = nullptr;     // pinned, one word per slice
  m4g_histo *h_histos = nullptr; // pinned staging (several devices: chunks interleave in the caller's array)
  hipStream_t st_in = nullptr, st_run = nullptr, st_out = nullptr;
  std::vector<hipEvent_t> ev_in, ev_run;
  // per run
  // m4g_hostbatch_compress (allocated at its first call)
  CodeRec *d_codes = nullptr;         // K5's per-match records, indexed like d_out
  Job *d_cjobs = nullptr;             // one job per chunk for the encoder kernels (share entries)
  Job *h_cjobs = nullptr;             // pinned image of d_cjobs
  EncBuffers eb;
  FlateBuffers fb;
  hipEvent_t ev_enc[2] = {nullptr, nullptr};
  std::vector<size_t> chunks;         // global indices of this lane's chunks
  std::vector<uint64_t> off, cap_off; // local input offsets / worst-case output slots
  std::vector<uint64_t> count_off;    // local exclusive prefix of match counts (after phase 1)
  int64_t result = M4G_OK;
};
} // namespace

struct m4g_hostbatch {
  Params p{};
  m4g_params raw{};
  size_t max_chunks = 0, max_chunk_bytes = 0;
  std::vector<HostLane> lanes;
};

namespace {
void hostlane_free(HostLane &L) {
  (void)hipSetDevice(L.device);
  for (hipStream_t st : {L.st_in, L.st_run, L.st_out})
    if (st) (void)hipStreamSynchronize(st);
  m4g_batch_destroy(L.plan);
  (void)hipFree(L.d_src);
  (void)hipFree(L.d_out);
  (void)hipFree(L.d_out_off);
  (void)hipFree(L.d_histos);
  if (L.h_out_off) (void)hipHostFree(L.h_out_off);
  if (L.h_err) (void)hipHostFree(L.h_err);
  if (L.h_histos) (void)hipHostFree(L.h_histos);
  (void)hipFree(L.d_codes);
  (void)hipFree(L.d_cjobs);
  if (L.h_cjobs) (void)hipHostFree(L.h_cjobs);
  L.eb.release();
  L.fb.release();
  for (auto &e : L.ev_enc) if (e) (void)hipEventDestroy(e);
  for (auto &e : L.ev_in) if (e) (void)hipEventDestroy(e);
  for (auto &e : L.ev_run) if (e) (void)hipEventDestroy(e);
  for (hipStream_t st : {L.st_in, L.st_run, L.st_out})
    if (st) (void)hipStreamDestroy(st);
  L = HostLane{};
}

#define LANE_TRY(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      L.result = fail(M4G_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
      return;                                                                            \
    }                                                                                    \
  } while (0)

// Phase 1 of a run on one lane: every slice's upload (st_in) and kernels (st_run) are enqueued, then the
// per-chunk match counts of all slices are awaited.  With one device the Match lists of a finished slice
// start their way home (st_out) while later slices still compute: their place in `out` is already known.
void hostlane_phase1(m4g_hostbatch *h, HostLane &L, size_t lane, const uint8_t *base, const uint64_t *chunk_off,
                     size_t nchunks, m4g_match *out, uint64_t *out_off, size_t out_cap, bool want_histos) {
  const size_t ndev = h->lanes.size();
  L.result = M4G_OK;
  L.chunks.clear();
  for (size_t c = lane; c < nchunks; c += ndev) L.chunks.push_back(c);
  const size_t n = L.chunks.size();
  L.off.assign(n + 1, 0);
  L.cap_off.assign(n + 1, 0);
  L.count_off.assign(n + 1, 0);
  for (size_t k = 0; k < n; k++) {
    const uint64_t len = chunk_off[L.chunks[k] + 1] - chunk_off[L.chunks[k]];
    L.off[k + 1] = L.off[k] + len;
    L.cap_off[k + 1] = L.cap_off[k] + len / (size_t)h->p.min_length + 2;
  }
  if (n == 0) return;
  LANE_TRY(hipSetDevice(L.device));
  const size_t nsl = (n + L.slice - 1) / L.slice;
  for (size_t s = 0; s < nsl; s++) {
    const size_t k0 = s * L.slice, k1 = std::min(n, k0 + L.slice), nc = k1 - k0;
    if (ndev == 1) { // the lane's chunks are one piece of the caller's buffer
      if (L.off[k1] > L.off[k0])
        LANE_TRY(hipMemcpyAsync(L.d_src + L.off[k0], base + chunk_off[k0], L.off[k1] - L.off[k0], hipMemcpyHostToDevice, L.st_in));
    } else {
      for (size_t k = k0; k < k1; k++)
        if (L.off[k + 1] > L.off[k])
          LANE_TRY(hipMemcpyAsync(L.d_src + L.off[k], base + chunk_off[L.chunks[k]], L.off[k + 1] - L.off[k],
                                  hipMemcpyHostToDevice, L.st_in));
    }
    LANE_TRY(hipEventRecord(L.ev_in[s], L.st_in));
    LANE_TRY(hipStreamWaitEvent(L.st_run, L.ev_in[s], 0));
    const int rc = m4g_batch_run_device(L.plan, L.d_src, L.off.data() + k0, nc, L.d_out + L.cap_off[k0],
                                        L.d_out_off + s * (L.slice + 1), L.cap_off[k1] - L.cap_off[k0],
                                        want_histos ? L.d_histos + k0 : nullptr, L.st_run);
    if (rc) {
      L.result = rc;
      return;
    }
    LANE_TRY(hipMemcpyAsync(L.h_out_off + s * (L.slice + 1), L.d_out_off + s * (L.slice + 1), (nc + 1) * 8,
                            hipMemcpyDeviceToHost, L.st_run));
    LANE_TRY(hipMemcpyAsync(L.h_err + s, L.plan->W.d_errors, 4, hipMemcpyDeviceToHost, L.st_run));
    LANE_TRY(hipEventRecord(L.ev_run[s], L.st_run));
  }
  uint32_t err = 0;
  uint64_t run = 0;
  for (size_t s = 0; s < nsl; s++) {
    const size_t k0 = s * L.slice, k1 = std::min(n, k0 + L.slice), nc = k1 - k0;
    LANE_TRY(hipEventSynchronize(L.ev_run[s]));
    err |= L.h_err[s];
    const uint64_t *ho = L.h_out_off + s * (L.slice + 1);
    for (size_t k = 0; k < nc; k++) L.count_off[k0 + k] = run + (ho[k] - ho[0]);
    const uint64_t cnt = ho[nc] - ho[0];
    if (ndev == 1) { // chunk order == local order: the slice's lists go to out[run ..) at once
      for (size_t k = 0; k < nc; k++) out_off[k0 + k] = L.count_off[k0 + k];
      if (!err && cnt && run + cnt <= out_cap)
        LANE_TRY(hipMemcpyAsync(out + run, L.d_out + L.cap_off[k0] + ho[0], cnt * sizeof(m4g_match), hipMemcpyDeviceToHost, L.st_out));
    }
    run += cnt;
  }
  L.count_off[n] = run;
  if (err) L.result = status_from_error_word(err);
}

// Phase 2 (several devices): the global offsets are known; every chunk's list goes to its place.
void hostlane_phase2(m4g_hostbatch *h, HostLane &L, const uint64_t *out_off, m4g_match *out, m4g_histo *histos) {
  const size_t n = L.chunks.size();
  if (n == 0 || L.result != M4G_OK) return;
  LANE_TRY(hipSetDevice(L.device));
  const size_t nsl = (n + L.slice - 1) / L.slice;
  for (size_t s = 0; s < nsl; s++) {
    const size_t k0 = s * L.slice, k1 = std::min(n, k0 + L.slice);
    const uint64_t *ho = L.h_out_off + s * (L.slice + 1);
    for (size_t k = k0; k < k1; k++) {
      const uint64_t cnt = ho[k - k0 + 1] - ho[k - k0];
      if (cnt)
        LANE_TRY(hipMemcpyAsync(out + out_off[L.chunks[k]], L.d_out + L.cap_off[k0] + ho[k - k0], cnt * sizeof(m4g_match),
                                hipMemcpyDeviceToHost, L.st_out));
    }
  }
  if (histos) {
    LANE_TRY(hipMemcpyAsync(L.h_histos, L.d_histos, n * sizeof(m4g_histo), hipMemcpyDeviceToHost, L.st_out));
    LANE_TRY(hipStreamSynchronize(L.st_out));
    for (size_t k = 0; k < n; k++) histos[L.chunks[k]] = L.h_histos[k];
  }
  LANE_TRY(hipStreamSynchronize(L.st_out));
  (void)h;
}
#undef LANE_TRY
} // namespace

extern "C" void m4g_hostbatch_destroy(m4g_hostbatch *h) {
  if (!h) return;
  for (auto &L : h->lanes) hostlane_free(L);
  delete h;
}

extern "C" int m4g_hostbatch_create(const m4g_params *params, const int *devices, int ndevices, size_t max_chunks,
                                    size_t max_chunk_bytes, m4g_hostbatch **out) {
  if (!out) return fail(M4G_ERR_BAD_PARAM, "out is NULL");
  *out = nullptr;
  Params p;
  int rc = resolve_params(params, &p);
  if (rc) return rc;
  if (!devices || ndevices < 1 || ndevices > 64) return fail(M4G_ERR_BAD_PARAM, "bad device list");
  if (max_chunks == 0 || max_chunks > (1u << 24)) return fail(M4G_ERR_BAD_PARAM, "max_chunks out of range");
  if (max_chunk_bytes == 0 || max_chunk_bytes > ((size_t)1 << 24))
    return fail(M4G_ERR_TOO_LARGE, "max_chunk_bytes above 2^24 (largest brotli meta-block)");
  for (int g = 0; g < ndevices; g++)
    if ((rc = check_device(devices[g]))) return rc;
  m4g_hostbatch *h = new m4g_hostbatch();
  h->p = p;
  h->raw = *params;
  h->max_chunks = max_chunks;
  h->max_chunk_bytes = max_chunk_bytes;
  h->lanes.resize((size_t)ndevices);
  for (int g = 0; g < ndevices; g++) {
    HostLane &L = h->lanes[(size_t)g];
    L.device = devices[g];
    L.share = (max_chunks + (size_t)ndevices - 1) / (size_t)ndevices;
    // slices: large enough to fill the GPU (>= 256 chunks), at most ~4 per run so that the copies of one
    // overlap the kernels of the next (batch_one_device_pipelined has the measurements)
    L.slice = std::max(kPipeSliceMin, (L.share + 3) / 4);
    if (const char *e = std::getenv("M4G_PIPE_SLICE")) L.slice = std::max<size_t>(1, (size_t)std::atoll(e));
    L.slice = std::min(L.slice, L.share);
    L.nslices = (L.share + L.slice - 1) / L.slice;
    const size_t cap = L.share * (max_chunk_bytes / (size_t)p.min_length + 2);
    hipError_t e = hipSetDevice(L.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.st_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.st_run, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.st_out, hipStreamNonBlocking);
    if (e == hipSuccess && (rc = batch_create(params, L.device, L.slice, max_chunk_bytes, false, &L.plan))) {
      m4g_hostbatch_destroy(h);
      return rc;
    }
    if (e == hipSuccess) e = hipMalloc(&L.d_src, L.share * max_chunk_bytes + 64);
    if (e == hipSuccess) e = hipMalloc(&L.d_out, cap * sizeof(m4g_match));
    if (e == hipSuccess) e = hipMalloc(&L.d_out_off, L.nslices * (L.slice + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&L.d_histos, L.share * sizeof(m4g_histo));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&L.h_out_off), L.nslices * (L.slice + 1) * 8, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&L.h_err), L.nslices * 4, hipHostMallocDefault);
    if (e == hipSuccess && ndevices > 1)
      e = hipHostMalloc(reinterpret_cast<void **>(&L.h_histos), L.share * sizeof(m4g_histo), hipHostMallocDefault);
    L.ev_in.assign(L.nslices, nullptr);
    L.ev_run.assign(L.nslices, nullptr);
    for (size_t s = 0; s < L.nslices && e == hipSuccess; s++) {
      e = hipEventCreateWithFlags(&L.ev_in[s], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&L.ev_run[s], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
      const std::string msg = std::string("m4g_hostbatch_create: ") + hipGetErrorString(e);
      m4g_hostbatch_destroy(h);
      return fail(e == hipErrorOutOfMemory ? M4G_ERR_NO_MEMORY : M4G_ERR_HIP, msg);
    }
  }
  *out = h;
  return M4G_OK;
}

extern "C" int64_t m4g_hostbatch_run(m4g_hostbatch *h, const uint8_t *base, const uint64_t *chunk_off, size_t nchunks,
                                     m4g_match *out, uint64_t *out_off, size_t out_cap, m4g_histo *histos) {
  if (!h || !chunk_off || !out_off || (!base && nchunks) || (!out && out_cap)) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (nchunks > h->max_chunks) return fail(M4G_ERR_TOO_LARGE, "nchunks above the handle's max_chunks");
  out_off[0] = 0;
  if (nchunks == 0) return M4G_OK;
  for (size_t c = 0; c < nchunks; c++) {
    if (chunk_off[c + 1] < chunk_off[c]) return fail(M4G_ERR_BAD_PARAM, "chunk_off not ascending");
    if (chunk_off[c + 1] - chunk_off[c] > h->max_chunk_bytes)
      return fail(M4G_ERR_TOO_LARGE, "chunk larger than the handle's max_chunk_bytes");
  }
  if (histos && h->p.min_length < 2)
    return fail(M4G_ERR_BAD_PARAM, "histograms need MinLength >= 2 (brotli has no copy length code for 1)");
  const size_t ndev = h->lanes.size();
  // phase 1: one host thread per device
  {
    std::vector<std::thread> th;
    for (size_t g = 1; g < ndev; g++)
      th.emplace_back(hostlane_phase1, h, std::ref(h->lanes[g]), g, base, chunk_off, nchunks, out, out_off, out_cap, histos != nullptr);
    hostlane_phase1(h, h->lanes[0], 0, base, chunk_off, nchunks, out, out_off, out_cap, histos != nullptr);
    for (auto &t : th) t.join();
  }
  for (auto &L : h->lanes)
    if (L.result != M4G_OK) { // (fail() text was set on the lane's thread: repeat it on the caller's)
      for (auto &M : h->lanes) {
        (void)hipSetDevice(M.device);
        for (hipStream_t st : {M.st_in, M.st_run, M.st_out}) (void)hipStreamSynchronize(st);
      }
      return fail((int)L.result, "m4g_hostbatch_run: a device lane failed (kernel-side error or HIP error)");
    }
  if (ndev == 1) {
    HostLane &L = h->lanes[0];
    const uint64_t run = L.count_off[nchunks];
    out_off[nchunks] = run;
    HIP_TRY(hipSetDevice(L.device));
    if (run <= out_cap && histos)
      HIP_TRY(hipMemcpyAsync(histos, L.d_histos, nchunks * sizeof(m4g_histo), hipMemcpyDeviceToHost, L.st_out));
    HIP_TRY(hipStreamSynchronize(L.st_out));
    return run > out_cap ? -(int64_t)(run + M4G_STREAM_NEEDED_BIAS) : (int64_t)M4G_OK;
  }
  // several devices: global offsets in chunk order, then every lane sends its lists to their places
  uint64_t run = 0;
  for (size_t c = 0; c < nchunks; c++) {
    const HostLane &L = h->lanes[c % ndev];
    const size_t k = c / ndev;
    out_off[c] = run;
    run += L.count_off[k + 1] - L.count_off[k];
  }
  out_off[nchunks] = run;
  if (run > out_cap) return -(int64_t)(run + M4G_STREAM_NEEDED_BIAS);
  {
    std::vector<std::thread> th;
    for (size_t g = 1; g < ndev; g++) th.emplace_back(hostlane_phase2, h, std::ref(h->lanes[g]), out_off, out, histos);
    hostlane_phase2(h, h->lanes[0], out_off, out, histos);
    for (auto &t : th) t.join();
  }
  for (auto &L : h->lanes)
    if (L.result != M4G_OK) return fail((int)L.result, "m4g_hostbatch_run: copying the Match lists back failed");
  return M4G_OK;
}

// Chunks in host memory -> ONE compressed stream in host memory (include/m4g.h).  Slices as in m4g_hostbatch_run:
// uploads on st_in, K1..K5 of every slice back to back on st_run; the encoder of a finished slice (host code
// building, K6 / k5f + k6f, D2H of its piece of the stream) runs on st_out while later slices still compute.  The
// Match lists stay on the device: the link carries 1 byte in and ~0.35 bytes out per input byte instead of 1 + 1.
extern "C" int64_t m4g_hostbatch_compress(m4g_hostbatch *h, int format, uint32_t gzip_mtime, const uint8_t *base,
                                          const uint64_t *chunk_off, size_t nchunks, uint8_t *stream_out, size_t stream_cap) {
  if (!h || !chunk_off || (!base && nchunks) || (!stream_out && stream_cap)) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (format != M4G_FORMAT_BROTLI && format != M4G_FORMAT_FLATE && format != M4G_FORMAT_GZIP)
    return fail(M4G_ERR_BAD_PARAM, "unknown format");
  if (nchunks > h->max_chunks) return fail(M4G_ERR_TOO_LARGE, "nchunks above the handle's max_chunks");
  if (format == M4G_FORMAT_BROTLI && h->p.min_length < 2)
    return fail(M4G_ERR_BAD_PARAM, "the brotli encoder needs MinLength >= 2 (no copy length code for 1)");
  for (size_t c = 0; c < nchunks; c++) {
    if (chunk_off[c + 1] < chunk_off[c]) return fail(M4G_ERR_BAD_PARAM, "chunk_off not ascending");
    if (chunk_off[c + 1] - chunk_off[c] > h->max_chunk_bytes)
      return fail(M4G_ERR_TOO_LARGE, "chunk larger than the handle's max_chunk_bytes");
  }
  const bool brotli = format == M4G_FORMAT_BROTLI;
  // Several devices: device g takes the g-th CONTIGUOUS range of chunks (stream order), all devices find their
  // matches at the same time, and the one bit writer of the stream passes over the slices of device 0, then those of
  // device 1, ... -- each slice is encoded on the device that holds its Match lists, at the bit position where the
  // previous one stopped, so nothing is moved or shifted afterwards.  The encoder chain (host code building + K6,
  // about a third of the match finding time of a slice) stays serial; the match finding is what the devices share.
  const size_t ndev = h->lanes.size();
  const size_t per = (nchunks + ndev - 1) / std::max<size_t>(ndev, 1);
  auto drain = [&]() { // (copies from / to the caller's buffers must not outlive the call)
    for (HostLane &M : h->lanes) {
      (void)hipSetDevice(M.device);
      for (hipStream_t st : {M.st_in, M.st_run, M.st_out}) (void)hipStreamSynchronize(st);
    }
  };
  struct Range {
    size_t c0 = 0, n = 0, nsl = 0;
  };
  std::vector<Range> ranges(ndev);
  // enqueue, device by device: job list, then per slice its upload and its kernels (nothing here waits)
  for (size_t g = 0; g < ndev; g++) {
    HostLane &L = h->lanes[g];
    Range &R = ranges[g];
    R.c0 = std::min(nchunks, g * per);
    R.n = std::min(nchunks, R.c0 + per) - R.c0;
    R.nsl = (R.n + L.slice - 1) / L.slice;
    if (R.n == 0 && g > 0) continue;
    HIP_TRY(hipSetDevice(L.device));
    const size_t cap = L.share * (h->max_chunk_bytes / (size_t)h->p.min_length + 2);
    if (brotli && !L.d_codes && hipMalloc(&L.d_codes, cap * sizeof(CodeRec)) != hipSuccess) {
      drain();
      return fail(M4G_ERR_NO_MEMORY, "hipMalloc of the per-match code records failed");
    }
    if (!L.d_cjobs) {
      if (hipMalloc(&L.d_cjobs, L.share * sizeof(Job)) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void **>(&L.h_cjobs), L.share * sizeof(Job), hipHostMallocDefault) != hipSuccess ||
          hipEventCreate(&L.ev_enc[0]) != hipSuccess || hipEventCreate(&L.ev_enc[1]) != hipSuccess) {
        drain();
        return fail(M4G_ERR_NO_MEMORY, "allocating the encoder's job list failed");
      }
    }
    const size_t n = R.n;
    const uint64_t *coff = chunk_off + R.c0;
    L.off.assign(n + 1, 0);
    L.cap_off.assign(n + 1, 0);
    for (size_t k = 0; k < n; k++) {
      const uint64_t len = coff[k + 1] - coff[k];
      L.off[k + 1] = L.off[k] + len;
      L.cap_off[k + 1] = L.cap_off[k] + len / (size_t)h->p.min_length + 2;
      Job J{};
      J.src_off = L.off[k];
      J.len = (uint32_t)len;
      L.h_cjobs[k] = J;
    }
    if (n) HIP_TRY(hipMemcpyAsync(L.d_cjobs, L.h_cjobs, n * sizeof(Job), hipMemcpyHostToDevice, L.st_in));
    for (size_t s = 0; s < R.nsl; s++) {
      const size_t k0 = s * L.slice, k1 = std::min(n, k0 + L.slice), nc = k1 - k0;
      if (L.off[k1] > L.off[k0])
        HIP_TRY(hipMemcpyAsync(L.d_src + L.off[k0], base + coff[k0], L.off[k1] - L.off[k0], hipMemcpyHostToDevice, L.st_in));
      HIP_TRY(hipEventRecord(L.ev_in[s], L.st_in));
      HIP_TRY(hipStreamWaitEvent(L.st_run, L.ev_in[s], 0));
      const int rc = m4g_batch_run_device_codes(
          L.plan, L.d_src, L.off.data() + k0, nc, L.d_out + L.cap_off[k0], L.d_out_off + s * (L.slice + 1),
          L.cap_off[k1] - L.cap_off[k0], brotli ? L.d_histos + k0 : nullptr,
          brotli ? reinterpret_cast<m4g_code *>(L.d_codes + L.cap_off[k0]) : nullptr, L.st_run);
      if (rc) {
        drain();
        return rc;
      }
      HIP_TRY(hipMemcpyAsync(L.h_err + s, L.plan->W.d_errors, 4, hipMemcpyDeviceToHost, L.st_run));
      HIP_TRY(hipEventRecord(L.ev_run[s], L.st_run));
    }
  }
  // the slices of the stream in order: (device, slice) pairs; an empty input is one empty piece on device 0
  std::vector<std::pair<size_t, size_t>> pieces;
  for (size_t g = 0; g < ndev; g++)
    for (size_t s = 0; s < ranges[g].nsl; s++) pieces.emplace_back(g, s);
  const bool empty = pieces.empty();
  if (empty) pieces.emplace_back(0, 0);
  // encode piece after piece; the stream continues at bit granularity
  uint64_t bytes = 0; // whole bytes of the stream so far
  unsigned pend_bits = 0;
  uint8_t pend_byte = 0;
  GzipCarry carry;
  for (size_t q = 0; q < pieces.size(); q++) {
    const size_t g = pieces[q].first, s = pieces[q].second;
    HostLane &L = h->lanes[g];
    const size_t n = ranges[g].n;
    const size_t k0 = s * L.slice, k1 = std::min(n, k0 + L.slice);
    const bool first = q == 0, last = q + 1 >= pieces.size();
    std::vector<uint64_t> lens;
    for (size_t k = k0; k < k1; k++) lens.push_back(L.off[k + 1] - L.off[k]);
    HIP_TRY(hipSetDevice(L.device));
    if (!empty) {
      HIP_TRY(hipEventSynchronize(L.ev_run[s]));
      if (L.h_err[s]) {
        drain();
        return status_from_error_word(L.h_err[s]);
      }
    }
    uint64_t end_bits = 0;
    int64_t r;
    uint8_t *dst = stream_out + bytes;
    const size_t room = stream_cap - (size_t)bytes;
    const int flags = (first ? 1 : 0) | (last ? 2 : 0);
    if (brotli)
      r = brotli_encode_jobs(L.eb, L.ev_enc[0], L.ev_enc[1], L.d_src, L.d_cjobs + k0, lens, L.d_out + L.cap_off[k0],
                             L.d_out_off + s * (L.slice + 1), L.d_histos + k0,
                             reinterpret_cast<const m4g_code *>(L.d_codes + L.cap_off[k0]), flags, dst, room, L.st_out,
                             nullptr, "m4g_hostbatch_compress", pend_bits, &end_bits);
    else
      r = flate_encode_jobs(L.fb, L.ev_enc[0], L.ev_enc[1], L.d_src, L.d_cjobs + k0, lens, L.d_out + L.cap_off[k0],
                            L.d_out_off + s * (L.slice + 1), flags | (format == M4G_FORMAT_GZIP ? 4 : 0), false, gzip_mtime,
                            dst, room, L.st_out, nullptr, nullptr, "m4g_hostbatch_compress", pend_bits, &end_bits, &carry);
    if (r < 0) {
      drain();
      if (r < -(int64_t)M4G_STREAM_NEEDED_BIAS) { // this piece did not fit: a bound for the whole stream
        const uint64_t total = chunk_off[nchunks] - chunk_off[0];
        return -(int64_t)(total + total / 4 + nchunks * 1024 + 128 + M4G_STREAM_NEEDED_BIAS);
      }
      return r;
    }
    if (r > 0) dst[0] |= pend_byte;
    const uint64_t whole = last ? (uint64_t)r : end_bits / 8;
    pend_bits = last ? 0u : (unsigned)(end_bits & 7);
    pend_byte = pend_bits ? dst[whole] : 0;
    bytes += whole;
  }
  drain();
  return (int64_t)bytes;
}

// ===========================================================================
// streaming handle
// ===========================================================================
struct m4g_ctx {
  int device = 0;
  Params p{};
  hipStream_t st = nullptr;
  uint8_t *d_hist = nullptr;
  uint32_t *d_chain = nullptr;
  size_t cap = 0;      // capacity (bytes / chain entries)
  int64_t hist_len = 0;
  uint32_t *d_table = nullptr;
  SpecWork w;
  size_t w_bytes = 0;           // block size the parse workspace was sized for
  m4g_match *d_out = nullptr;   // compacted matches of the last block
  uint64_t *d_out_off = nullptr;
  size_t out_cap = 0;
  uint32_t *d_counts = nullptr; // [0] count, [1] errors
  Job *d_job = nullptr;
  m4g_match *h_recs = nullptr; // pinned staging
  size_t guess_recs = 0;       // records fetched together with the count (previous call's count + 25 %)
  // block-parallel mode (m4g_find_matches_blocks)
  SpecWork wb;                  // parse workspace for wb_jobs blocks of wb_bytes
  size_t wb_jobs = 0, wb_bytes = 0;
  Job *d_bjobs = nullptr;       // segment jobs (K1) followed by block jobs (K3..)
  size_t bjobs_cap = 0;
  StreamInfo *d_si = nullptr;
  uint32_t *d_table2 = nullptr; // the segments of one call read d_table and write this one; then swapped
  uint32_t *d_bcounts = nullptr;
  uint64_t *d_boff = nullptr;
  uint8_t *d_tmp = nullptr;     // scratch for the history move after a call
  size_t tmp_cap = 0;
  uint64_t *h_off = nullptr;   // pinned, 2 words
  uint32_t *h_counts = nullptr; // pinned, 2 words
};

extern "C" void m4g_destroy(m4g_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->st) (void)hipStreamSynchronize(c->st);
  (void)hipFree(c->d_hist);
  (void)hipFree(c->d_chain);
  (void)hipFree(c->d_table);
  spec_free(c->w);
  spec_free(c->wb);
  (void)hipFree(c->d_bjobs);
  (void)hipFree(c->d_si);
  (void)hipFree(c->d_table2);
  (void)hipFree(c->d_bcounts);
  (void)hipFree(c->d_boff);
  (void)hipFree(c->d_tmp);
  (void)hipFree(c->d_out);
  (void)hipFree(c->d_out_off);
  (void)hipFree(c->d_counts);
  (void)hipFree(c->d_job);
  if (c->h_recs) (void)hipHostFree(c->h_recs);
  if (c->h_off) (void)hipHostFree(c->h_off);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

extern "C" int m4g_create(const m4g_params *params, int device, m4g_ctx **out) {
  if (!out) return fail(M4G_ERR_BAD_PARAM, "out is NULL");
  *out = nullptr;
  Params p;
  int rc = resolve_params(params, &p);
  if (rc) return rc;
  if ((rc = check_device(device))) return rc;
  HIP_TRY(hipSetDevice(device));
  if ((rc = k1_prepare())) return rc;
  m4g_ctx *c = new m4g_ctx();
  c->device = device;
  c->p = p;
  const size_t tbytes = ((size_t)1 << p.table_bits) * 4;
  if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&c->d_table, tbytes) != hipSuccess || hipMalloc(&c->d_counts, 64) != hipSuccess ||
      hipMalloc(&c->d_job, sizeof(Job)) != hipSuccess ||
      hipHostMalloc(&c->h_counts, 64, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(&c->h_off, 64, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(&c->d_out_off, 64) != hipSuccess ||
      hipMemsetAsync(c->d_table, 0, tbytes, c->st) != hipSuccess) {
    m4g_destroy(c);
    return fail(M4G_ERR_NO_MEMORY, "allocating the streaming handle failed");
  }
  *out = c;
  return M4G_OK;
}

extern "C" int m4g_reset(m4g_ctx *c) {
  if (!c) return fail(M4G_ERR_BAD_PARAM, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  // m4.go:51-57
  HIP_TRY(hipMemsetAsync(c->d_table, 0, ((size_t)1 << c->p.table_bits) * 4, c->st));
  c->hist_len = 0;
  return M4G_OK;
}

extern "C" int64_t m4g_history_len(m4g_ctx *c) { return c ? c->hist_len : -1; }

namespace {
int ctx_reserve(m4g_ctx *c, size_t need) {
  if (need <= c->cap) return M4G_OK;
  size_t nc = c->cap ? c->cap : ((size_t)1 << 20);
  while (nc < need) nc *= 2;
  uint8_t *nh = nullptr;
  uint32_t *nch = nullptr;
  if (hipMalloc(&nh, nc + 64) != hipSuccess || hipMalloc(&nch, nc * 4 + 4096) != hipSuccess) {
    (void)hipFree(nh);
    (void)hipFree(nch);
    return fail(M4G_ERR_NO_MEMORY, "growing the history buffer failed");
  }
  if (c->hist_len) {
    HIP_TRY(hipMemcpyAsync(nh, c->d_hist, (size_t)c->hist_len, hipMemcpyDeviceToDevice, c->st));
    HIP_TRY(hipMemcpyAsync(nch, c->d_chain, (size_t)c->hist_len * 4, hipMemcpyDeviceToDevice, c->st));
  }
  HIP_TRY(hipStreamSynchronize(c->st));
  (void)hipFree(c->d_hist);
  (void)hipFree(c->d_chain);
  c->d_hist = nh;
  c->d_chain = nch;
  c->cap = nc;
  return M4G_OK;
}
} // namespace

namespace {
// m4.go:82-99: slide.  delta > MaxDistance here, so source and destination of
// the copies do not overlap.
int ctx_slide(m4g_ctx *c) {
  const Params &p = c->p;
  if (c->hist_len > 2 * (int64_t)p.max_distance) {
    const int64_t delta = c->hist_len - p.max_distance;
    HIP_TRY(hipMemcpyAsync(c->d_hist, c->d_hist + delta, (size_t)p.max_distance,
                           hipMemcpyDeviceToDevice, c->st));
    HIP_TRY(hipMemcpyAsync(c->d_chain, c->d_chain + delta, (size_t)p.max_distance * 4,
                           hipMemcpyDeviceToDevice, c->st));
    const uint32_t tn = 1u << p.table_bits;
    hipLaunchKernelGGL(ks_rebase_table, dim3((tn + 255) / 256), dim3(256), 0, c->st, c->d_table, tn,
                       (uint32_t)delta);
    HIP_TRY(hipGetLastError());
    c->hist_len = p.max_distance;
  }
  return M4G_OK;
}
} // namespace

extern "C" int64_t m4g_find_matches(m4g_ctx *c, const uint8_t *src, size_t n,
                                    uint32_t seed_prev_distance, m4g_match *out, size_t out_cap) {
  if (!c || (!src && n) || (!out && out_cap)) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  const Params &p = c->p;
  if ((uint64_t)c->hist_len + n >= ((uint64_t)1 << 31))
    return fail(M4G_ERR_TOO_LARGE, "history + block must stay below 2^31 bytes");
  const size_t need = n / (size_t)p.min_length + 2;
  if (out_cap < need && n != 0) {
    // refuse before touching any state, so the handle is unchanged
    g_last_error = "out_cap below n/min_length + 2";
    return -(int64_t)need;
  }
  HIP_TRY(hipSetDevice(c->device));

  { int rc0 = ctx_slide(c); if (rc0) return rc0; }
  if (n == 0) return 0; // m4.go:112,276: nothing inserted, nothing emitted

  int rc = ctx_reserve(c, (size_t)c->hist_len + n);
  if (rc) return rc;
  if (c->w_bytes < n) {
    size_t nb = std::max(n, (size_t)1 << 16);
    if ((rc = spec_alloc(c->w, p, 1, nb, kStreamLanes))) return rc;
    c->w_bytes = nb;
  }
  if (c->out_cap < need) {
    (void)hipFree(c->d_out);
    c->d_out = nullptr;
    if (c->h_recs) (void)hipHostFree(c->h_recs);
    c->h_recs = nullptr;
    size_t nc = std::max(need, (size_t)4096);
    if (hipMalloc(&c->d_out, nc * sizeof(m4g_match)) != hipSuccess ||
        hipHostMalloc(&c->h_recs, nc * sizeof(m4g_match), hipHostMallocDefault) != hipSuccess)
      return fail(M4G_ERR_NO_MEMORY, "allocating the match buffers failed");
    c->out_cap = nc;
  }

  // m4.go:101-107: append src
  HIP_TRY(hipMemcpyAsync(c->d_hist + c->hist_len, src, n, hipMemcpyHostToDevice, c->st));
  Job J{};
  J.src_off = 0;
  J.chain_off = 0;
  J.start = (uint32_t)c->hist_len;
  J.len = (uint32_t)(c->hist_len + (int64_t)n);
  J.seed_prev = seed_prev_distance;
  J.k1_flags = kJobLoadTable | kJobStoreTable;
  HIP_TRY(hipMemcpyAsync(c->d_job, &J, sizeof(Job), hipMemcpyHostToDevice, c->st));
  HIP_TRY(hipMemsetAsync(c->d_counts, 0, 8, c->st));
  // positions in [start, len-7) that K1 does not visit keep chain = 0 only if
  // cleared: the last 7 of every block are never inserted (m4.go:112).
  HIP_TRY(hipMemsetAsync(c->d_chain + c->hist_len, 0, n * 4, c->st));
  rc = launch_find(p, c->d_hist, (uint64_t)c->hist_len + n, (uint32_t)(c->hist_len + (int64_t)n), c->d_chain, c->d_job, c->d_table, 1, c->w, c->d_counts,
                   c->d_counts + 1, c->st, nullptr);
  if (rc) return rc;
  if ((rc = launch_compact(c->w, c->d_job, c->d_counts, 1, c->d_out_off, c->d_out, c->out_cap, c->st)))
    return rc;
  // The match count is only known on the device.  To get count and records in ONE round trip the
  // records are fetched on a guess (the previous call's count + 25 %: consecutive blocks of a stream
  // are alike); only a call that produced more pays a second copy.
  const size_t guess = std::min(std::min(c->out_cap, out_cap), c->guess_recs);
  HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, 8, hipMemcpyDeviceToHost, c->st));
  if (guess)
    HIP_TRY(hipMemcpyAsync(c->h_recs, c->d_out, guess * sizeof(m4g_match), hipMemcpyDeviceToHost, c->st));
  HIP_TRY(hipStreamSynchronize(c->st));
  c->hist_len += (int64_t)n;
  if (int rce = status_from_error_word(c->h_counts[1])) return rce;
  const uint32_t cnt = c->h_counts[0];
  if (cnt > c->out_cap || cnt > out_cap) return fail(M4G_ERR_HIP, "internal: match count above bound");
  if (cnt > guess) {
    HIP_TRY(hipMemcpyAsync(c->h_recs + guess, c->d_out + guess, ((size_t)cnt - guess) * sizeof(m4g_match),
                           hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
  }
  if (cnt) std::memcpy(out, c->h_recs, (size_t)cnt * sizeof(m4g_match));
  c->guess_recs = (size_t)cnt + cnt / 4 + 64;
  return (int64_t)cnt;
}

// ---------------------------------------------------------------------------
// Block-parallel streaming: the result of calling m4g_find_matches on
// consecutive blocks of `block_size` bytes (dst empty each time, as
// matchfinder.Writer does, matchfinder.go:81-87), computed in one launch.
// Each block's parse starts from a fresh matchEmitter, so given the chain
// array the blocks are independent jobs; K1 builds the chain in segments, each
// primed with the MaxDistance bytes before it (kernels.h, StreamInfo).
// ---------------------------------------------------------------------------
namespace {
// What m4g_writer hangs into a block-parallel launch: its encoder runs on the device-resident results (bytes,
// block jobs, Match lists, offsets) before the handle's history is compacted, so neither the Match lists nor
// the bytes cross PCIe a second time.  Only the one-launch path calls it (blocks_one_launch_ok).
struct BlocksHook {
  LongRun *d_long_runs = nullptr; // K4 fills these for k5b when set (brotli)
  uint32_t *d_n_long = nullptr;
  uint32_t long_cap = 0;
  std::function<int(const uint8_t *d_src, const Job *d_blocks, const std::vector<uint64_t> &lens, const m4g_match *d_out,
                    const uint64_t *d_off, uint64_t nmatches, hipStream_t st)> fn;
};
bool blocks_one_launch_ok(const Params &p, size_t n, size_t block_size) {
  const size_t nblocks = (n + block_size - 1) / block_size;
  return nblocks >= 2 && block_size >= (size_t)kK1MaxTile + 7 && (uint64_t)p.max_distance >= (uint64_t)kK1MaxTile;
}
// Span mode (one stream over several devices, m4g_find_matches_blocks_multi): `src` holds the stream from a block
// boundary at or below the origin in force for the first wanted block; the blocks from byte `skip` on are wanted,
// the bytes before only stand for the history a sequential run would hold.  org[b]: the origin of wanted block b
// (m4.go:82-99 replayed over the WHOLE stream by the caller), in coordinates of `src`.  The handle must be fresh
// (no history) and stays so.
struct SpanMode {
  size_t skip;
  const uint64_t *org;
};
int64_t find_blocks_impl(m4g_ctx *c, const uint8_t *src, size_t n, size_t block_size, m4g_match *out, size_t out_cap,
                         uint64_t *block_off, BlocksHook *hook, const SpanMode *sm = nullptr);
} // namespace

extern "C" int64_t m4g_find_matches_blocks(m4g_ctx *c, const uint8_t *src, size_t n, size_t block_size,
                                           m4g_match *out, size_t out_cap, uint64_t *block_off) {
  return find_blocks_impl(c, src, n, block_size, out, out_cap, block_off, nullptr);
}

namespace {
int64_t find_blocks_impl(m4g_ctx *c, const uint8_t *src, size_t n, size_t block_size, m4g_match *out, size_t out_cap,
                         uint64_t *block_off, BlocksHook *hook, const SpanMode *sm) {
  if (!c || (!src && n) || (!out && out_cap) || block_size == 0)
    return fail(M4G_ERR_BAD_PARAM, "NULL argument or block_size 0");
  const Params &p = c->p;
  if ((uint64_t)c->hist_len + n >= ((uint64_t)1 << 31))
    return fail(M4G_ERR_TOO_LARGE, "history + data must stay below 2^31 bytes");
  const size_t skip = sm ? sm->skip : 0; // bytes of `src` that are history only
  if (sm && (c->hist_len != 0 || hook || skip % block_size != 0 || skip >= n))
    return fail(M4G_ERR_INTERNAL, "span mode: fresh handle, whole blocks, no hook");
  const size_t nblocks = (n - skip + block_size - 1) / block_size;
  size_t need = 0;
  for (size_t b = 0; b < nblocks; b++)
    need += std::min(block_size, n - skip - b * block_size) / (size_t)p.min_length + 2;
  if (out_cap < need && n != 0 && !hook) {
    g_last_error = "out_cap below the sum of len/min_length + 2 over the blocks";
    return -(int64_t)need;
  }
  if (block_off) block_off[0] = 0;
  if (n == 0) return 0;
  if (hook && !blocks_one_launch_ok(p, n, block_size)) return fail(M4G_ERR_INTERNAL, "hook outside the one-launch path");

  // The one-launch path needs: more than one block, at most one block end and
  // one slide per K1 tile, and a bounded slide list.  Otherwise: block by block.
  const uint64_t W = (uint64_t)p.max_distance;
  const bool shape_ok = (nblocks >= 2 || sm) && block_size >= (size_t)kK1MaxTile + 7 && W >= (uint64_t)kK1MaxTile;
  if (sm && !shape_ok) return fail(M4G_ERR_INTERNAL, "span mode needs the one-launch shape");
  if (!sm && shape_ok && (n / W + 2 >= (size_t)kStreamMaxSlides || nblocks >= ((size_t)1 << 24))) {
    // more history slides than one launch's list holds (a small MaxDistance: flate's 32 KiB window slides at
    // every block): spans of whole blocks, each one launch; the handle carries the state from span to span
    const size_t span_blocks = std::max<size_t>(2, std::min<size_t>(((size_t)kStreamMaxSlides - 3) * W / block_size,
                                                                   ((size_t)1 << 24) - 1));
    int64_t total = 0;
    for (size_t b0 = 0; b0 < nblocks; b0 += span_blocks) {
      const size_t nb = std::min(span_blocks, nblocks - b0);
      const size_t len = std::min(n - b0 * block_size, nb * block_size);
      std::vector<uint64_t> off(nb + 1);
      const int64_t r = find_blocks_impl(c, src + b0 * block_size, len, block_size, hook ? nullptr : out + total,
                                         hook ? 0 : out_cap - (size_t)total, off.data(), hook);
      if (r < 0) return r;
      if (block_off)
        for (size_t k = 1; k <= nb; k++) block_off[b0 + k] = (uint64_t)total + off[k];
      total += r;
    }
    return total;
  }
  const bool fast = shape_ok;
  if (!fast) {
    int64_t total = 0;
    for (size_t b = 0; b < nblocks; b++) {
      const size_t len = std::min(block_size, n - b * block_size);
      int64_t r = m4g_find_matches(c, src + b * block_size, len, 0, out + total, out_cap - (size_t)total);
      if (r < 0) return r;
      total += r;
      if (block_off) block_off[b + 1] = (uint64_t)total;
    }
    return total;
  }

  HIP_TRY(hipSetDevice(c->device));
  int rc = ctx_slide(c);
  if (rc) return rc;
  const uint64_t x0 = sm ? (uint64_t)skip : (uint64_t)c->hist_len, x_end = sm ? (uint64_t)n : x0 + n;
  if ((rc = ctx_reserve(c, (size_t)x_end))) return rc;

  // slide schedule of the equivalent call sequence (m4.go:82-99)
  StreamInfo si{};
  si.x0 = sm ? 0u : (uint32_t)x0; // (span mode: the buffer begins at a block boundary of the stream)
  si.x_end = (uint32_t)x_end;
  si.block_size = (uint32_t)block_size;
  si.nbp = 1;
  si.bp_x[0] = 0;
  si.bp_org[0] = sm ? (uint32_t)sm->org[0] : 0u;
  uint64_t org = sm ? sm->org[0] : 0;
  std::vector<Job> jobs; // segments first, then blocks
  std::vector<uint64_t> block_org(nblocks);
  for (size_t b = 0; b < nblocks; b++) {
    const uint64_t xb = x0 + b * block_size;
    const bool moved = sm ? sm->org[b] != org : (b > 0 && xb - org > 2 * W); // len(history) > 2*MaxDistance at the start of this call
    if (moved) {
      org = sm ? sm->org[b] : xb - W;
      if (si.nbp >= (uint32_t)kStreamMaxSlides) return fail(M4G_ERR_INTERNAL, "more history slides than one launch's list holds");
      si.bp_x[si.nbp] = (uint32_t)xb;
      si.bp_org[si.nbp] = (uint32_t)org;
      si.nbp++;
    }
    block_org[b] = org;
  }
  si.final_org = (uint32_t)org;

  // K1 segments: the first one continues the handle's table and covers at
  // least MaxDistance bytes, so every later segment primes from this call's data
  const uint64_t S = std::max<uint64_t>(256u << 10, std::min<uint64_t>(4u << 20, (n + 63) / 64));
  // Span mode: no table to continue.  The first segment starts MaxDistance below the first wanted block with an
  // empty table and STORES from there: chain entries of the history are read by chain walks (m4.go:166-181,
  // 239-252), and only those of positions within MaxDistance of a wanted position can be reached; a predecessor
  // further back than that is rejected by the distance test wherever it is read, so "none" stands for it.
  const uint64_t first = sm ? (x0 > W ? x0 - W : 0) : x0;
  uint64_t a = first;
  while (a < x_end) {
    uint64_t e = a == first ? std::min(x_end, std::max(x0, first) + std::max(W, S)) : std::min(x_end, a + S);
    if (x_end - e < S / 4) e = x_end; // no tiny tail segment
    Job J{};
    J.src_off = 0;
    J.chain_off = 0;
    J.start = (uint32_t)(a == first ? first : a - W);
    J.len = (uint32_t)(e + 7); // K1 inserts positions below len - 7 (block ends are skipped by rule)
    J.store_from = (uint32_t)a;
    J.k1_flags = (a == first && !sm ? kJobLoadTable : 0u) | (e == x_end ? kJobStoreTable : 0u);
    jobs.push_back(J);
    a = e;
  }
  const size_t nseg = jobs.size();
  if (std::getenv("M4G_TRACE"))
    std::fprintf(stderr, "[m4g] blocks: one launch, %zu K1 segments, %zu blocks, %u slides\n", nseg, nblocks,
                 si.nbp - 1);
  for (size_t b = 0; b < nblocks; b++) {
    const uint64_t xb = x0 + b * block_size, xe = std::min(x_end, xb + block_size);
    Job J{};
    J.src_off = block_org[b];
    J.chain_off = block_org[b];
    J.start = (uint32_t)(xb - block_org[b]);
    J.len = (uint32_t)(xe - block_org[b]);
    jobs.push_back(J);
  }

  // workspace
  if (c->wb_jobs < nblocks || c->wb_bytes < block_size) {
    const size_t nj = std::max(nblocks, c->wb_jobs), nb = std::max(block_size, c->wb_bytes);
    c->wb_jobs = c->wb_bytes = 0;
    if ((rc = spec_alloc(c->wb, p, nj, nb))) return rc;
    (void)hipFree(c->d_bcounts);
    (void)hipFree(c->d_boff);
    c->d_bcounts = nullptr;
    c->d_boff = nullptr;
    if (hipMalloc(&c->d_bcounts, nj * 4) != hipSuccess || hipMalloc(&c->d_boff, (nj + 1) * 8) != hipSuccess)
      return fail(M4G_ERR_NO_MEMORY, "allocating the per-block counters failed");
    c->wb_jobs = nj;
    c->wb_bytes = nb;
  }
  if (c->bjobs_cap < jobs.size()) {
    (void)hipFree(c->d_bjobs);
    c->d_bjobs = nullptr;
    c->bjobs_cap = 0;
    if (hipMalloc(&c->d_bjobs, jobs.size() * sizeof(Job)) != hipSuccess)
      return fail(M4G_ERR_NO_MEMORY, "allocating the job list failed");
    c->bjobs_cap = jobs.size();
  }
  if (!c->d_si && hipMalloc(&c->d_si, sizeof(StreamInfo)) != hipSuccess)
    return fail(M4G_ERR_NO_MEMORY, "allocating the stream descriptor failed");
  if (!c->d_table2 && hipMalloc(&c->d_table2, ((size_t)1 << p.table_bits) * 4) != hipSuccess)
    return fail(M4G_ERR_NO_MEMORY, "allocating the second table failed");
  if (c->out_cap < need) {
    (void)hipFree(c->d_out);
    c->d_out = nullptr;
    if (c->h_recs) (void)hipHostFree(c->h_recs);
    c->h_recs = nullptr;
    c->out_cap = 0;
    if (hipMalloc(&c->d_out, need * sizeof(m4g_match)) != hipSuccess ||
        hipHostMalloc(&c->h_recs, need * sizeof(m4g_match), hipHostMallocDefault) != hipSuccess)
      return fail(M4G_ERR_NO_MEMORY, "allocating the match buffers failed");
    c->out_cap = need;
  }

  Trace tr;
  // m4.go:101-107: append src; chain entries of never-inserted positions stay 0
  HIP_TRY(hipMemcpyAsync(c->d_hist + (sm ? 0 : x0), src, n, hipMemcpyHostToDevice, c->st));
  HIP_TRY(hipMemsetAsync(c->d_chain + first, 0, (x_end - first) * 4, c->st));
  HIP_TRY(hipMemcpyAsync(c->d_bjobs, jobs.data(), jobs.size() * sizeof(Job), hipMemcpyHostToDevice, c->st));
  HIP_TRY(hipMemcpyAsync(c->d_si, &si, sizeof(si), hipMemcpyHostToDevice, c->st));
  HIP_TRY(hipMemsetAsync(c->d_counts, 0, 8, c->st));
  HIP_TRY(hipStreamSynchronize(c->st)); // jobs / si are stack and vector memory
  tr.mark("blocks: H2D");

  if ((rc = launch_k1(p, c->d_hist, x_end, (uint32_t)x_end, c->d_chain, c->d_bjobs, c->d_table, (int)nseg,
                      c->st, c->d_si, c->d_table2)))
    return rc;
  std::swap(c->d_table, c->d_table2);
  const Job *d_blocks = c->d_bjobs + nseg;
  if ((rc = launch_parse(p, c->d_hist, x_end, c->d_chain, d_blocks, (int)nblocks, c->wb, c->d_bcounts,
                         c->d_counts + 1, c->st)))
    return rc;
  if (hook && hook->d_n_long) HIP_TRY(hipMemsetAsync(hook->d_n_long, 0, 4, c->st));
  if ((rc = launch_compact(c->wb, d_blocks, c->d_bcounts, (int)nblocks, c->d_boff, c->d_out, c->out_cap, c->st,
                           hook ? hook->d_long_runs : nullptr, hook ? hook->d_n_long : nullptr, hook ? hook->long_cap : 0u)))
    return rc;

  std::vector<uint64_t> off(nblocks + 1);
  HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, 8, hipMemcpyDeviceToHost, c->st));
  HIP_TRY(hipMemcpyAsync(off.data(), c->d_boff, (nblocks + 1) * 8, hipMemcpyDeviceToHost, c->st));
  HIP_TRY(hipStreamSynchronize(c->st));
  tr.mark("blocks: kernels");
  if (hook) { // the encoder, on the device-resident results, before the history moves
    if (int rce = status_from_error_word(c->h_counts[1])) return rce;
    if (off[nblocks] > c->out_cap) return fail(M4G_ERR_HIP, "internal: match count above bound");
    std::vector<uint64_t> lens(nblocks);
    for (size_t b = 0; b < nblocks; b++) lens[b] = std::min<uint64_t>(block_size, n - skip - b * block_size);
    if ((rc = hook->fn(c->d_hist, d_blocks, lens, c->d_out, c->d_boff, off[nblocks], c->st))) return rc;
    tr.mark("blocks: encoder hook");
  }

  // the handle continues as after the last block's call: history = buffer from the final origin on
  // (span mode: the handle stays fresh)
  const uint64_t new_len = sm ? 0 : x_end - org;
  if (org > 0 && !sm) {
    if (org >= new_len) {
      HIP_TRY(hipMemcpyAsync(c->d_hist, c->d_hist + org, new_len, hipMemcpyDeviceToDevice, c->st));
      HIP_TRY(hipMemcpyAsync(c->d_chain, c->d_chain + org, new_len * 4, hipMemcpyDeviceToDevice, c->st));
    } else { // overlapping ranges: through a scratch buffer
      if (c->tmp_cap < new_len * 4) {
        (void)hipFree(c->d_tmp);
        c->d_tmp = nullptr;
        c->tmp_cap = 0;
        if (hipMalloc(&c->d_tmp, new_len * 4) != hipSuccess)
          return fail(M4G_ERR_NO_MEMORY, "allocating the history scratch failed");
        c->tmp_cap = new_len * 4;
      }
      HIP_TRY(hipMemcpyAsync(c->d_tmp, c->d_hist + org, new_len, hipMemcpyDeviceToDevice, c->st));
      HIP_TRY(hipMemcpyAsync(c->d_hist, c->d_tmp, new_len, hipMemcpyDeviceToDevice, c->st));
      HIP_TRY(hipMemcpyAsync(c->d_tmp, c->d_chain + org, new_len * 4, hipMemcpyDeviceToDevice, c->st));
      HIP_TRY(hipMemcpyAsync(c->d_chain, c->d_tmp, new_len * 4, hipMemcpyDeviceToDevice, c->st));
    }
  }
  c->hist_len = (int64_t)new_len;

  if (int rce = status_from_error_word(c->h_counts[1])) return rce;
  const uint64_t cnt = off[nblocks];
  if (cnt > c->out_cap || (!hook && cnt > out_cap)) return fail(M4G_ERR_HIP, "internal: match count above bound");
  if (cnt && !hook) {
    HIP_TRY(hipMemcpyAsync(c->h_recs, c->d_out, (size_t)cnt * sizeof(m4g_match), hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    std::memcpy(out, c->h_recs, (size_t)cnt * sizeof(m4g_match));
  } else {
    HIP_TRY(hipStreamSynchronize(c->st));
  }
  tr.mark("blocks: D2H + copy");
  if (block_off)
    for (size_t b = 0; b <= nblocks; b++) block_off[b] = off[b];
  return (int64_t)cnt;
}
} // namespace

// ===========================================================================
// One stream over several devices (protocol P1): include/m4g.h,
// m4g_find_matches_blocks_multi.
// ===========================================================================
namespace {
// One piece of the plan: blocks [b0, b1) of the stream go to devices[dev]; the device gets the bytes from `lo`
// (a block boundary at or below the origin in force for block b0) to the end of block b1 - 1.
struct SpanPiece {
  uint32_t dev;
  uint64_t b0, b1, lo;
};
// m4.go:82-99 replayed over the block lengths of the whole stream: the buffer origin (absolute stream offset of
// history[0]) in force while block b is parsed.
void stream_origins(uint64_t n, uint64_t bs, uint64_t W, std::vector<uint64_t> &org) {
  const uint64_t nblocks = (n + bs - 1) / bs;
  org.resize(nblocks);
  uint64_t origin = 0, hlen = 0;
  for (uint64_t b = 0; b < nblocks; b++) {
    if (hlen > 2 * W) { // the slide happens BEFORE the append
      origin += hlen - W;
      hlen = W;
    }
    org[b] = origin;
    hlen += std::min(bs, n - b * bs);
  }
}
// Contiguous ranges of whole blocks, one per device, each cut into pieces that one launch can hold: at most
// kStreamMaxSlides - 3 origin changes, fewer than 2^24 blocks, and (history + piece) below max_piece_bytes.
void plan_spans(uint64_t n, uint64_t bs, uint64_t W, int ndev, uint64_t max_piece_bytes, const std::vector<uint64_t> &org,
                std::vector<SpanPiece> &plan) {
  const uint64_t nblocks = org.size();
  const uint64_t per = (nblocks + (uint64_t)ndev - 1) / (uint64_t)ndev;
  for (int d = 0; d < ndev; d++) {
    const uint64_t d0 = std::min(nblocks, (uint64_t)d * per), d1 = std::min(nblocks, d0 + per);
    uint64_t b0 = d0;
    while (b0 < d1) {
      const uint64_t lo = org[b0] / bs * bs;
      uint64_t b1 = b0 + 1, moves = 0;
      while (b1 < d1) {
        const uint64_t m = moves + (org[b1] != org[b1 - 1] ? 1 : 0);
        const uint64_t end = std::min(n, (b1 + 1) * bs);
        if (m + 3 >= (uint64_t)kStreamMaxSlides || b1 + 1 - b0 >= ((uint64_t)1 << 24) || end - lo > max_piece_bytes) break;
        moves = m;
        b1++;
      }
      plan.push_back(SpanPiece{(uint32_t)d, b0, b1, lo});
      b0 = b1;
    }
  }
}
constexpr uint64_t kSpanPieceBytes = (uint64_t)768 << 20; // bytes a device holds per launch (history included)
} // namespace

extern "C" int64_t m4g_debug_span_plan(const m4g_params *params, size_t n, size_t block_size, int ndevices,
                                       uint64_t *pieces /* 4 per piece: dev, b0, b1, lo */, size_t cap,
                                       uint64_t *origins /* nullable, one per block */) {
  Params p;
  if (int rc = resolve_params(params, &p)) return rc;
  if (block_size == 0 || ndevices <= 0) return fail(M4G_ERR_BAD_PARAM, "block_size 0 or no device");
  std::vector<uint64_t> org;
  stream_origins(n, block_size, (uint64_t)p.max_distance, org);
  std::vector<SpanPiece> plan;
  plan_spans(n, block_size, (uint64_t)p.max_distance, ndevices, kSpanPieceBytes, org, plan);
  if (origins) std::copy(org.begin(), org.end(), origins);
  for (size_t k = 0; k < plan.size() && k < cap; k++) {
    pieces[4 * k] = plan[k].dev, pieces[4 * k + 1] = plan[k].b0, pieces[4 * k + 2] = plan[k].b1, pieces[4 * k + 3] = plan[k].lo;
  }
  return (int64_t)plan.size();
}

extern "C" int64_t m4g_find_matches_blocks_multi(const m4g_params *params, const int *devices, int ndevices,
                                                 const uint8_t *src, size_t n, size_t block_size, m4g_match *out,
                                                 size_t out_cap, uint64_t *block_off) {
  Params p;
  if (int rc = resolve_params(params, &p)) return rc;
  if (!devices || ndevices <= 0 || (!src && n) || (!out && out_cap) || block_size == 0)
    return fail(M4G_ERR_BAD_PARAM, "NULL argument, no device or block_size 0");
  if (n >= ((uint64_t)1 << 40)) return fail(M4G_ERR_TOO_LARGE, "at most 1 TiB per stream");
  for (int d = 0; d < ndevices; d++)
    if (int rc = check_device(devices[d])) return rc;
  const size_t nblocks = (n + block_size - 1) / block_size;
  size_t need = 0;
  for (size_t b = 0; b < nblocks; b++) need += std::min(block_size, n - b * block_size) / (size_t)p.min_length + 2;
  if (out_cap < need && n != 0) {
    g_last_error = "out_cap below the sum of len/min_length + 2 over the blocks";
    return -(int64_t)(need + M4G_STREAM_NEEDED_BIAS);
  }
  if (block_off) block_off[0] = 0;
  if (n == 0) return 0;
  const uint64_t W = (uint64_t)p.max_distance;
  const bool shape_ok = block_size >= (size_t)kK1MaxTile + 7 && W >= (uint64_t)kK1MaxTile;
  if (!shape_ok || nblocks < 2) {
    // tiny blocks or a tiny window: the block-by-block path of one handle (nothing to spread)
    if (n >= ((uint64_t)1 << 31)) return fail(M4G_ERR_TOO_LARGE, "this block shape runs on one handle: below 2^31 bytes");
    m4g_ctx *c = nullptr;
    if (int rc = m4g_create(params, devices[0], &c)) return rc;
    const int64_t r = find_blocks_impl(c, src, n, block_size, out, out_cap, block_off, nullptr);
    m4g_destroy(c);
    return r;
  }
  std::vector<uint64_t> org;
  stream_origins(n, block_size, W, org);
  std::vector<SpanPiece> plan;
  plan_spans(n, block_size, W, ndevices, kSpanPieceBytes, org, plan);

  // one host thread per device; every piece into its own list, joined in stream order afterwards
  struct PieceOut {
    std::vector<m4g_match> m;
    std::vector<uint64_t> off;
    int64_t count = 0;
  };
  std::vector<PieceOut> res(plan.size());
  std::vector<int64_t> rcs((size_t)ndevices, 0);
  std::vector<std::string> errs((size_t)ndevices);
  auto work = [&](int d) {
    m4g_ctx *c = nullptr;
    int64_t rc = m4g_create(params, devices[d], &c);
    for (size_t k = 0; k < plan.size() && rc == 0; k++) {
      const SpanPiece &S = plan[k];
      if ((int)S.dev != d) continue;
      const uint64_t x0 = S.b0 * block_size, x1 = std::min<uint64_t>(n, S.b1 * block_size);
      const size_t nb = (size_t)(S.b1 - S.b0);
      std::vector<uint64_t> o(nb);
      for (size_t b = 0; b < nb; b++) o[b] = org[S.b0 + b] - S.lo;
      size_t cap = 0;
      for (size_t b = 0; b < nb; b++) cap += (size_t)(std::min<uint64_t>(n, (S.b0 + b + 1) * block_size) - (S.b0 + b) * block_size) / (size_t)p.min_length + 2;
      res[k].m.resize(cap);
      res[k].off.resize(nb + 1);
      SpanMode sm{(size_t)(x0 - S.lo), o.data()};
      const int64_t r = find_blocks_impl(c, src + S.lo, (size_t)(x1 - S.lo), block_size, res[k].m.data(), cap, res[k].off.data(),
                                         nullptr, &sm);
      if (r < 0) rc = r;
      else res[k].count = r;
    }
    if (rc) errs[(size_t)d] = g_last_error; // (thread-local: carried to the caller's thread below)
    if (c) m4g_destroy(c);
    rcs[(size_t)d] = rc;
  };
  {
    std::vector<std::thread> th;
    for (int d = 1; d < ndevices; d++) th.emplace_back(work, d);
    work(0);
    for (auto &t : th) t.join();
  }
  for (int d = 0; d < ndevices; d++)
    if (rcs[(size_t)d]) {
      if (d) g_last_error = errs[(size_t)d];
      return rcs[(size_t)d];
    }
  // ordered concatenation (pieces are in stream order: devices take ascending ranges)
  uint64_t total = 0;
  for (size_t k = 0; k < plan.size(); k++) {
    const size_t nb = (size_t)(plan[k].b1 - plan[k].b0);
    if (total + (uint64_t)res[k].count > out_cap) return fail(M4G_ERR_INTERNAL, "match count above bound");
    if (res[k].count) std::memcpy(out + total, res[k].m.data(), (size_t)res[k].count * sizeof(m4g_match));
    if (block_off)
      for (size_t b = 1; b <= nb; b++) block_off[plan[k].b0 + b] = total + res[k].off[b];
    total += (uint64_t)res[k].count;
  }
  return (int64_t)total;
}

extern "C" int m4g_debug_dump_tables(m4g_ctx *c, uint32_t *table, uint32_t *chain, size_t chain_cap) {
  if (!c) return fail(M4G_ERR_BAD_PARAM, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->st));
  if (table)
    HIP_TRY(hipMemcpy(table, c->d_table, ((size_t)1 << c->p.table_bits) * 4, hipMemcpyDeviceToHost));
  if (chain && c->hist_len) {
    size_t k = std::min(chain_cap, (size_t)c->hist_len);
    HIP_TRY(hipMemcpy(chain, c->d_chain, k * 4, hipMemcpyDeviceToHost));
  }
  return M4G_OK;
}

// ===========================================================================
// m4g_writer: matchfinder.Writer (matchfinder/matchfinder.go:44-103) with the
// match finder AND the encoder on the GPU (include/m4g.h).
// ===========================================================================
struct m4g_writer {
  m4g_ctx *ctx = nullptr;
  int device = 0, format = 0;
  size_t bs = 0;
  uint32_t mtime = 0;
  std::vector<uint8_t> inbuf; // Writer.inBuf: bytes that do not make up a block yet
  std::vector<uint8_t> outq;  // produced, not yet handed to the caller
  bool wrote_header = false;
  uint8_t pend_byte = 0;      // the bit writer's pending bits (bitwriter.go:11-17, huffman_bit_writer.go:68-70)
  unsigned pend_bits = 0;
  GzipCarry carry;
  EncBuffers eb;
  FlateBuffers fb;
  uint32_t *d_histos = nullptr;
  size_t histos_cap = 0;
  CodeRec *d_codes = nullptr;
  size_t codes_cap = 0;
  LongRun *d_runs = nullptr;
  uint32_t *d_nruns = nullptr;
  size_t runs_cap = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<uint8_t> stage;
  bool closed_pending = false; // Close() has encoded the last block; its bytes wait for a large enough `out`
  bool failed = false;         // a call failed half way: the handle's state is undefined until m4g_writer_reset
  // MatchFinder = Bargain1 / 2 / 3 instead of M4 (m4g_writer_create_v2; bargain_host.h): ctx stays NULL
  bool v2 = false;
  m4g_bargain_params bp{};
  m4g_bargain *bplan = nullptr;
  size_t bplan_bytes = 0;       // what the plan holds
  std::vector<uint8_t> bhist;   // the stream's last bytes: whole blocks, at least MaxDistance bytes once it is that long
  std::vector<uint8_t> bchunk;  // bhist + the call's blocks, contiguous (what goes to the device)
  uint8_t *d_bsrc = nullptr;
  m4g_match *d_bm = nullptr;
  uint64_t *d_boff = nullptr;
  Job *d_bjobs = nullptr;
  size_t d_bsrc_cap = 0, d_bm_cap = 0, d_boff_cap = 0, d_bjobs_cap = 0;
};

namespace {
int writer_reserve_brotli(m4g_writer *w, size_t njobs, uint64_t nmatches, size_t nruns) {
  if (njobs > w->histos_cap) {
    (void)hipFree(w->d_histos);
    w->d_histos = nullptr;
    w->histos_cap = 0;
    if (hipMalloc(&w->d_histos, njobs * sizeof(m4g_histo)) != hipSuccess) return fail(M4G_ERR_NO_MEMORY, "hipMalloc failed");
    w->histos_cap = njobs;
  }
  if (nmatches + 1 > w->codes_cap) {
    (void)hipFree(w->d_codes);
    w->d_codes = nullptr;
    w->codes_cap = 0;
    if (hipMalloc(&w->d_codes, (nmatches + 1) * sizeof(CodeRec)) != hipSuccess) return fail(M4G_ERR_NO_MEMORY, "hipMalloc failed");
    w->codes_cap = nmatches + 1;
  }
  if (nruns + 1 > w->runs_cap) {
    (void)hipFree(w->d_runs);
    w->d_runs = nullptr;
    w->runs_cap = 0;
    if (hipMalloc(&w->d_runs, (nruns + 1) * sizeof(LongRun)) != hipSuccess) return fail(M4G_ERR_NO_MEMORY, "hipMalloc failed");
    w->runs_cap = nruns + 1;
  }
  if (!w->d_nruns && hipMalloc(&w->d_nruns, 4) != hipSuccess) return fail(M4G_ERR_NO_MEMORY, "hipMalloc failed");
  return M4G_OK;
}

// The encoder over device-resident blocks: what Writer.writeBlock does after FindMatches, for all of them.
// brotli: d_runs / d_nruns already hold the literal runs of K5_HUGE bytes or more.
int writer_encode(m4g_writer *w, const uint8_t *d_src, const Job *d_jobs, const std::vector<uint64_t> &lens,
                  const m4g_match *d_out, const uint64_t *d_off, uint64_t nmatches, hipStream_t st, bool final) {
  uint64_t total = 0, max_len = 0;
  for (uint64_t l : lens) total += l, max_len = std::max(max_len, l);
  w->stage.resize((size_t)(total + total / 4) + lens.size() * 1024 + 128); // (fixed codes: up to 9 bits per literal)
  uint64_t end_bits = 0;
  int64_t r;
  const int first = w->wrote_header ? 0 : 1;
  if (w->format == M4G_FORMAT_BROTLI) {
    const size_t njobs = lens.size();
    hipLaunchKernelGGL(k5_histograms<true>, dim3((unsigned)njobs), dim3(K5_THREADS), 0, st, d_src,
                       reinterpret_cast<const uint32_t *>(d_out), d_jobs, d_off, (uint64_t)nmatches, w->d_histos, w->d_codes,
                       static_cast<const uint32_t *>(nullptr));
    HIP_TRY(hipGetLastError());
    const uint32_t segs = (uint32_t)(((size_t)max_len + K5B_SEG - 1) / K5B_SEG);
    hipLaunchKernelGGL(k5b_long_literals, dim3(2048), dim3(K5B_THREADS), 0, st, d_src, d_jobs, w->d_runs, w->d_nruns,
                       (uint32_t)(w->runs_cap - 1), segs, d_off, (uint64_t)nmatches, w->d_histos,
                       static_cast<const uint32_t *>(nullptr));
    HIP_TRY(hipGetLastError());
    r = brotli_encode_jobs(w->eb, w->e0, w->e1, d_src, d_jobs, lens, d_out, d_off,
                           reinterpret_cast<const m4g_histo *>(w->d_histos), reinterpret_cast<const m4g_code *>(w->d_codes),
                           first | (final ? 2 : 0), w->stage.data(), w->stage.size(), st, nullptr, "m4g_writer",
                           w->pend_bits, &end_bits);
  } else {
    r = flate_encode_jobs(w->fb, w->e0, w->e1, d_src, d_jobs, lens, d_out, d_off,
                          first | (w->format == M4G_FORMAT_GZIP ? 4 : 0), final, w->mtime, w->stage.data(), w->stage.size(),
                          st, nullptr, nullptr, "m4g_writer", w->pend_bits, &end_bits, &w->carry);
  }
  if (r < 0) return r < -M4G_STREAM_NEEDED_BIAS ? fail(M4G_ERR_INTERNAL, "m4g_writer: staging buffer too small") : (int)r;
  if (r > 0) w->stage[0] |= w->pend_byte;
  w->wrote_header = true;
  const uint64_t whole = final ? (uint64_t)r : end_bits / 8;
  w->outq.insert(w->outq.end(), w->stage.begin(), w->stage.begin() + (size_t)whole);
  w->pend_bits = final ? 0u : (unsigned)(end_bits & 7);
  w->pend_byte = w->pend_bits ? w->stage[(size_t)whole] : 0;
  return M4G_OK;
}

// blocks of `bs` bytes of span (the last one may be shorter) -> outq.  final: the span is Close()'s remainder.
int writer_process_impl(m4g_writer *w, const uint8_t *span, size_t len, size_t bs, bool final);
int writer_process(m4g_writer *w, const uint8_t *span, size_t len, size_t bs, bool final) {
  const int rc = writer_process_impl(w, span, len, bs, final);
  if (rc) w->failed = true;
  return rc;
}
int writer_process_bargain(m4g_writer *w, const uint8_t *span, size_t len, size_t bs, bool final); // bargain_host.h
int writer_process_impl(m4g_writer *w, const uint8_t *span, size_t len, size_t bs, bool final) {
  if (w->v2) return writer_process_bargain(w, span, len, bs, final);
  m4g_ctx *c = w->ctx;
  const Params &p = c->p;
  HIP_TRY(hipSetDevice(w->device));
  if (!final && bs != 0 && blocks_one_launch_ok(p, len, bs)) {
    // match finding and encoding in one pass over device memory
    BlocksHook hook;
    const size_t nblocks = (len + bs - 1) / bs;
    uint64_t bound = 0;
    for (size_t b = 0; b < nblocks; b++) bound += std::min(bs, len - b * bs) / (size_t)p.min_length + 2;
    if (w->format == M4G_FORMAT_BROTLI) {
      const size_t nruns = len / K5_HUGE + nblocks + 64;
      if (int rc = writer_reserve_brotli(w, nblocks, bound, nruns)) return rc;
      hook.d_long_runs = w->d_runs;
      hook.d_n_long = w->d_nruns;
      hook.long_cap = (uint32_t)(w->runs_cap - 1);
    }
    hook.fn = [w](const uint8_t *d_src, const Job *d_blocks, const std::vector<uint64_t> &lens, const m4g_match *d_out,
                  const uint64_t *d_off, uint64_t nmatches, hipStream_t st) -> int {
      if (w->format == M4G_FORMAT_BROTLI)
        if (int rc = writer_reserve_brotli(w, lens.size(), nmatches, 0)) return rc; // (a span of a longer call)
      return writer_encode(w, d_src, d_blocks, lens, d_out, d_off, nmatches, st, false);
    };
    const int64_t r = find_blocks_impl(c, span, len, bs, nullptr, 0, nullptr, &hook);
    return r < 0 ? (int)r : M4G_OK;
  }
  // one block, tiny blocks, or the remainder: Match lists over the host, then the same encoder
  const size_t bs_eff = bs ? bs : std::max<size_t>(len, 1);
  const size_t nlists = (len + bs_eff - 1) / bs_eff;
  size_t need = 1;
  for (size_t b = 0; b < nlists; b++) need += std::min(bs_eff, len - b * bs_eff) / (size_t)p.min_length + 2;
  std::vector<m4g_match> ms(need);
  std::vector<uint64_t> off(nlists + 1, 0);
  if (len) {
    const int64_t r = find_blocks_impl(c, span, len, bs_eff, ms.data(), ms.size(), off.data(), nullptr);
    if (r < 0) return (int)r;
  }
  BlockUpload U;
  // (Close's remainder and the empty Write of BlockSize 0 are ONE block even when it is empty)
  const bool one = final || (bs == 0 && len == 0);
  if (int rc = U.init(w->device, span, len, one ? std::max<size_t>(len, 1) + 1 : bs_eff, ms.data(), off.data(), one))
    return rc;
  if (w->format == M4G_FORMAT_BROTLI) {
    std::vector<LongRun> runs;
    for (size_t j = 0; j < U.jobs.size(); j++) {
      uint32_t pos = 0;
      for (uint64_t k = U.off[j]; k < U.off[j + 1]; k++) {
        if (ms[k].unmatched >= K5_HUGE) runs.push_back(LongRun{(uint32_t)j, pos, ms[k].unmatched, 0});
        pos += ms[k].unmatched + ms[k].length;
      }
    }
    if (int rc = writer_reserve_brotli(w, U.jobs.size(), U.nmatches, runs.size())) return rc;
    const uint32_t nruns = (uint32_t)runs.size();
    if (nruns) HIP_TRY(hipMemcpyAsync(w->d_runs, runs.data(), runs.size() * sizeof(LongRun), hipMemcpyHostToDevice, U.st));
    HIP_TRY(hipMemcpyAsync(w->d_nruns, &nruns, 4, hipMemcpyHostToDevice, U.st));
    HIP_TRY(hipStreamSynchronize(U.st));
  }
  return writer_encode(w, U.d_src, U.d_jobs, U.lens, U.d_m, U.d_off, U.nmatches, U.st, final);
}

int64_t writer_drain(m4g_writer *w, uint8_t *out, size_t out_cap) {
  const size_t k = w->outq.size();
  if (k > out_cap) return -(int64_t)(k + M4G_STREAM_NEEDED_BIAS);
  if (k) std::memcpy(out, w->outq.data(), k);
  w->outq.clear();
  return (int64_t)k;
}
} // namespace

extern "C" int m4g_writer_create(const m4g_params *params, int device, int format, size_t block_size,
                                 uint32_t gzip_mtime, m4g_writer **out) {
  if (!out) return fail(M4G_ERR_BAD_PARAM, "out is NULL");
  *out = nullptr;
  if (format != M4G_FORMAT_BROTLI && format != M4G_FORMAT_FLATE && format != M4G_FORMAT_GZIP)
    return fail(M4G_ERR_BAD_PARAM, "unknown format");
  if (block_size >= ((size_t)1 << 30)) return fail(M4G_ERR_TOO_LARGE, "block_size must stay below 2^30");
  m4g_ctx *ctx = nullptr;
  if (int rc = m4g_create(params, device, &ctx)) return rc;
  if (format == M4G_FORMAT_BROTLI && ctx->p.min_length < 2) {
    m4g_destroy(ctx);
    return fail(M4G_ERR_BAD_PARAM, "the brotli encoder needs MinLength >= 2 (no copy length code for 1)");
  }
  m4g_writer *w = new m4g_writer();
  w->ctx = ctx;
  w->device = device;
  w->format = format;
  w->bs = block_size;
  w->mtime = gzip_mtime;
  if (hipEventCreate(&w->e0) != hipSuccess || hipEventCreate(&w->e1) != hipSuccess) {
    m4g_writer_destroy(w);
    return fail(M4G_ERR_HIP, "hipEventCreate failed");
  }
  *out = w;
  return M4G_OK;
}

extern "C" void m4g_writer_destroy(m4g_writer *w) {
  if (!w) return;
  (void)hipSetDevice(w->device);
  w->eb.release();
  w->fb.release();
  (void)hipFree(w->d_histos);
  (void)hipFree(w->d_codes);
  (void)hipFree(w->d_runs);
  (void)hipFree(w->d_nruns);
  if (w->e0) (void)hipEventDestroy(w->e0);
  if (w->e1) (void)hipEventDestroy(w->e1);
  m4g_destroy(w->ctx);
  m4g_bargain_destroy(w->bplan);
  (void)hipFree(w->d_bsrc);
  (void)hipFree(w->d_bm);
  (void)hipFree(w->d_boff);
  (void)hipFree(w->d_bjobs);
  delete w;
}

extern "C" int m4g_writer_reset(m4g_writer *w) {
  if (!w) return fail(M4G_ERR_BAD_PARAM, "writer is NULL");
  w->inbuf.clear();
  w->outq.clear();
  w->wrote_header = false;
  w->pend_byte = 0;
  w->pend_bits = 0;
  w->carry = GzipCarry();
  w->closed_pending = false;
  w->failed = false;
  w->bhist.clear();
  return w->v2 ? M4G_OK : m4g_reset(w->ctx);
}

extern "C" int64_t m4g_writer_write(m4g_writer *w, const uint8_t *src, size_t n, uint8_t *out, size_t out_cap) {
  if (!w || (!src && n) || (!out && out_cap)) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (n >= ((uint64_t)1 << 31)) return fail(M4G_ERR_TOO_LARGE, "at most 2 GiB per Write");
  if (w->failed) return fail(M4G_ERR_BAD_PARAM, "m4g_writer: an earlier call failed; m4g_writer_reset first");
  if (w->bs == 0) { // each Write is one block (matchfinder.go:64-66)
    // ... an empty Write too (brotli: nothing but the stream header; flate: an empty block).  The one exception is
    // the retry of include/m4g.h -- n = 0 while output of the previous call is still kept -- which only fetches it.
    if (n == 0 && !w->outq.empty()) return writer_drain(w, out, out_cap);
    if (int rc = writer_process(w, src, n, 0, false)) return rc;
    return writer_drain(w, out, out_cap);
  }
  // matchfinder.go:68-78: append, encode every full block, keep the rest
  const uint8_t *span = src;
  size_t avail = n;
  if (!w->inbuf.empty()) {
    const size_t blocks = (w->inbuf.size() + n) / w->bs;
    if (blocks == 0) { // still no full block
      w->inbuf.insert(w->inbuf.end(), src, src + n);
      return writer_drain(w, out, out_cap);
    }
    // the pending bytes and the new ones up to the last block boundary, contiguous
    const size_t take = blocks * w->bs - w->inbuf.size();
    w->inbuf.insert(w->inbuf.end(), src, src + take);
    if (int rc = writer_process(w, w->inbuf.data(), w->inbuf.size(), w->bs, false)) return rc;
    w->inbuf.clear();
    span = src + take;
    avail = n - take;
  }
  const size_t full = (avail / w->bs) * w->bs;
  if (full)
    if (int rc = writer_process(w, span, full, w->bs, false)) return rc;
  w->inbuf.insert(w->inbuf.end(), span + full, span + avail);
  return writer_drain(w, out, out_cap);
}

extern "C" int64_t m4g_writer_close(m4g_writer *w, uint8_t *out, size_t out_cap) {
  if (!w || (!out && out_cap)) return fail(M4G_ERR_BAD_PARAM, "NULL argument");
  if (w->failed) return fail(M4G_ERR_BAD_PARAM, "m4g_writer: an earlier call failed; m4g_writer_reset first");
  if (!w->closed_pending) {
    if (int rc = writer_process(w, w->inbuf.data(), w->inbuf.size(), w->bs, true)) return rc; // matchfinder.go:89-93
    w->inbuf.clear();
    w->closed_pending = true;
  }
  const int64_t r = writer_drain(w, out, out_cap);
  if (r >= 0) w->closed_pending = false;
  return r;
}

#include "bargain_host.h"
