/*
 * m4g.h -- C-ABI of libm4g.so: an MI355X (gfx950) implementation of the
 * andybalholm/brotli `matchfinder.M4` match finder and of the histogram pass
 * of `brotli.Encoder`, for use behind the reference's own plugin interface
 *
 *     type MatchFinder interface {                 // matchfinder/matchfinder.go:23-30
 *         FindMatches(dst []Match, src []byte) []Match
 *         Reset()
 *     }
 *
 * Plain C types only (no torch / HIP types in any signature).  The reference
 * is pure Go with no FFI today; these are the entry points a cgo shim binds
 * (INTEGRATION.md shows that shim).  Each declaration cites the reference
 * code it replaces (paths relative to the reference checkout).
 *
 * Threading: a handle (m4g_ctx / m4g_batch) may be used by one thread at a
 * time; distinct handles are independent.  Every entry point selects its HIP
 * device itself, so callers whose OS thread changes between calls (Go
 * goroutines) need no device bookkeeping.
 *
 * There is NO CPU backend: every entry point runs hand-written HIP kernels
 * and returns M4G_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef M4G_H
#define M4G_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* matchfinder.Match (matchfinder/matchfinder.go:16-20).  Go's three ints are
 * 64-bit; all values on this path are < 2^32, the shim widens them. */
typedef struct {
  uint32_t unmatched; /* Match.Unmatched */
  uint32_t length;    /* Match.Length (0 only for a trailing literal run) */
  uint32_t distance;  /* Match.Distance */
} m4g_match;

/* The exported fields of matchfinder.M4 (matchfinder/m4.go:15-43).
 * 0 means "reference default", exactly as in m4.go:64-75:
 * MaxDistance 65535, MinLength 4, HashLen 6, TableBits 17. */
typedef struct {
  int32_t max_distance;
  int32_t min_length;
  int32_t hash_len;
  int32_t table_bits;
  int32_t chain_length;
  int32_t distance_bit_cost;
} m4g_params;

/* The three histograms built by pass 1 of brotli.Encoder.Encode
 * (encoder.go:33-35: literalHisto, commandHisto, distanceHisto). */
typedef struct {
  uint32_t lit[256];
  uint32_t cmd[704];
  uint32_t dist[64];
} m4g_histo;

/* What pass 2 of brotli.Encoder.Encode (encoder.go:122-154) needs per match
 * besides the Match itself: the command symbol (encoder.go:57-63, recomputed
 * by the reference at :124-130) and the match's `distCache` entry
 * (`distanceCode{code, nExtra, extraBits}`, encoder.go:42, 66-97, 163-177).
 * A match that takes no distance code (command < 128, or the trailing literal
 * run with Length 0; encoder.go:67) has dist_code = n_extra = extra_bits = 0 --
 * the reference leaves a stale entry there and never reads it. */
typedef struct {
  uint16_t command;    /* 0..703 */
  uint8_t dist_code;   /* distanceCode.code: 0..9 ring codes, >= 16 direct */
  uint8_t n_extra;     /* distanceCode.nExtra */
  uint32_t extra_bits; /* distanceCode.extraBits */
} m4g_code;

typedef struct m4g_ctx m4g_ctx;     /* one streaming M4 (one Go `*M4`) */
typedef struct m4g_batch m4g_batch; /* reusable plan for independent chunks */

enum {
  M4G_OK = 0,
  M4G_ERR_NO_DEVICE = -1000001, /* no usable HIP device / extension kernels */
  M4G_ERR_BAD_PARAM = -1000002,
  M4G_ERR_HIP = -1000003,       /* a HIP call failed; see m4g_last_error() */
  M4G_ERR_TOO_LARGE = -1000004, /* input beyond the limits stated below */
  M4G_ERR_NO_MEMORY = -1000005,
  M4G_ERR_REPEAT_RANGE = -1000006, /* seeded repeat distance reaches before the
                                      history start (the Go code would panic) */
  M4G_ERR_TOKEN_RANGE = -1000008,  /* m4g_batch_flate_encode: a match outside
                                    * Length 3..258 / Distance 1..32768 */
  M4G_ERR_INTERNAL = -1000007      /* a kernel-side capacity check failed; the
                                      output of that call is not valid */
};
/* Count-returning entry points of ONE block use values >= -1000000 for
 * "-(needed)"; the entry points whose outputs can be far larger -- the stream
 * encoders and the batch calls over host memory (m4g_find_matches_batch,
 * m4g_hostbatch_run) -- return -(needed + M4G_STREAM_NEEDED_BIAS) instead,
 * which is below every M4G_ERR_* value. */
#define M4G_STREAM_NEEDED_BIAS 2000000

/* Human-readable text for an error code. */
const char *m4g_strerror(int code);
/* Detail of the last failure on the calling thread (HIP error text etc.). */
const char *m4g_last_error(void);
/* Number of usable gfx950 devices (0 when none / no driver). */
int m4g_device_count(void);

/* ---- streaming: one call == one M4.FindMatches ------------------------- */

/* Replaces: constructing `&matchfinder.M4{...}` (m4.go:15-49).
 * device >= 0 selects the HIP device.  device < 0 is rejected
 * (M4G_ERR_NO_DEVICE): there is deliberately no CPU fallback. */
int m4g_create(const m4g_params *params, int device, m4g_ctx **out);
void m4g_destroy(m4g_ctx *ctx);

/* Replaces: M4.Reset (m4.go:51-57). */
int m4g_reset(m4g_ctx *ctx);

/* Replaces: M4.FindMatches(dst, src) (m4.go:63-283) with the same streaming
 * semantics: the handle keeps history / hash table / chain between calls,
 * including the history slide of m4.go:82-99.  `src` (host memory, borrowed
 * for the call) is appended to the history; the matches of this block are
 * written to `out`.
 *   seed_prev_distance: Distance of the last element of the caller's `dst`
 *       when `dst` is non-empty (m4.go:129-130 reads it for the repeat
 *       probe); 0 when dst is empty -- matchfinder.Writer always passes
 *       w.matches[:0] (matchfinder.go:83).
 *   returns the number of matches (>= 0), or an M4G_ERR_*.  out_cap must be
 *       at least n/min_length + 2 (the worst case); a smaller out_cap returns
 *       -(that bound) before anything is touched, so the handle is unchanged.
 * Limit: n plus retained history must stay below 2^31 bytes. */
int64_t m4g_find_matches(m4g_ctx *ctx, const uint8_t *src, size_t n,
                         uint32_t seed_prev_distance, m4g_match *out,
                         size_t out_cap);

/* Block-parallel streaming -- SURVEY.md section 8(f).1.  Same result, and
 * same handle state afterwards, as calling m4g_find_matches(ctx, block_b, len_b,
 * 0, ...) for the consecutive blocks of `block_size` bytes of src (the last
 * one may be shorter): what matchfinder.Writer.writeBlock does block after
 * block with an empty dst (matchfinder/matchfinder.go:81-87, 59-79), including
 * the history slide schedule (m4.go:82-99) and the seven uninserted positions at
 * every block end (m4.go:112).  All blocks are processed in one launch.
 * out receives the concatenated Match lists, block_off (nullable, nblocks+1
 * entries) the start of each block's list.  Returns the total count, or
 * -(needed capacity) when out_cap is below the sum of len_b/min_length + 2
 * (handle unchanged), or an M4G_ERR_* code.  After this call
 * m4g_debug_dump_tables may show empty table entries where the sequential
 * calls would hold positions further back than MaxDistance; no result
 * depends on those. */
int64_t m4g_find_matches_blocks(m4g_ctx *ctx, const uint8_t *src, size_t n,
                                size_t block_size, m4g_match *out,
                                size_t out_cap, uint64_t *block_off);

/* ONE stream over several devices (protocol P1).  Replaces:
 *   m := &matchfinder.M4{params}
 *   for each block of block_size bytes of src: m.FindMatches(nil, block)
 * i.e. what matchfinder.Writer (matchfinder.go:59-93) asks of a fresh M4 over a
 * whole input -- with the blocks spread over `devices`: device d parses a
 * contiguous range of whole blocks and is handed, besides them, the bytes from
 * the buffer origin in force for its first block (m4.go:82-99 replayed on the
 * host over the whole stream), which lies at least MaxDistance below it once
 * the stream has slid.  No device talks to another; one host thread per
 * device; the per-block Match lists are joined in stream order on the host.
 * Arguments and results as m4g_find_matches_blocks on a fresh handle; returns
 * the number of Matches, -(needed + M4G_STREAM_NEEDED_BIAS) when out_cap is too
 * small, or an M4G_ERR_*.  A device index may be listed more than once (its
 * ranges then run side by side on that device). */
int64_t m4g_find_matches_blocks_multi(const m4g_params *params, const int *devices, int ndevices,
                                      const uint8_t *src, size_t n, size_t block_size, m4g_match *out,
                                      size_t out_cap, uint64_t *block_off);
/* Debug / tests: the plan of the call above -- per piece {device slot, first
 * block, end block, first byte handed over}, and per block the buffer origin
 * (absolute stream offset) in force while it is parsed.  Returns the number
 * of pieces (fills at most cap of them).  No device is touched. */
int64_t m4g_debug_span_plan(const m4g_params *params, size_t n, size_t block_size, int ndevices,
                            uint64_t *pieces, size_t cap, uint64_t *origins);

/* Debug / parity: copy the handle's hash table (1<<table_bits entries) and
 * chain deltas (one per history byte; reference keeps them only when
 * ChainLength > 0, this implementation always) to host memory.  Either
 * pointer may be NULL.  (m4.go:45-46 `table`, `chain`.) */
int m4g_debug_dump_tables(m4g_ctx *ctx, uint32_t *table, uint32_t *chain,
                          size_t chain_cap);
/* Current history length (len(q.history), m4.go:48). */
int64_t m4g_history_len(m4g_ctx *ctx);

/* ---- batch of independent chunks (throughput path) ---------------------
 * Semantics per chunk: `m.Reset(); m.FindMatches(nil, chunk)` == the
 * reference's matchfinder.AutoReset wrapper (matchfinder/textencoder.go:46-53),
 * followed (optionally) by pass 1 of brotli.Encoder.Encode (encoder.go:44-105)
 * on that chunk's matches.
 */

/* Plan + device workspace for up to max_chunks chunks of up to
 * max_chunk_bytes bytes each (max_chunk_bytes <= 2^24, the largest
 * meta-block brotli.Encoder accepts, compress_fragment_two_pass.go:232). */
int m4g_batch_create(const m4g_params *params, int device, size_t max_chunks,
                     size_t max_chunk_bytes, m4g_batch **out);
void m4g_batch_destroy(m4g_batch *b);

/* Device-resident form: every pointer prefixed d_ is device memory on the
 * plan's device; chunk_off is HOST memory (nchunks+1 ascending byte offsets
 * into d_src).  Work is enqueued on `hip_stream` (a hipStream_t, NULL = the
 * default stream) and the call returns without synchronising.
 *   d_out      : out_cap records; chunk c's matches land at
 *                d_out[d_out_off[c] .. d_out_off[c+1]) in chunk order.
 *   d_out_off  : nchunks+1 uint64 (exclusive prefix sum of match counts).
 *   d_histos   : nullable; one m4g_histo per chunk.
 * If the total match count exceeds out_cap nothing is written past out_cap
 * and d_out_off still holds the true offsets (the caller can detect it); a
 * chunk whose list was cut by out_cap gets an ALL-ZERO histogram (and no codes).
 * d_histos needs MinLength >= 2 (M4G_ERR_BAD_PARAM otherwise): a trimmed match
 * of Length 1 has no brotli copy length code (command.go:128-139; the Go code
 * would index out of range). */
int m4g_batch_run_device(m4g_batch *b, const uint8_t *d_src,
                         const uint64_t *chunk_off, size_t nchunks,
                         m4g_match *d_out, uint64_t *d_out_off, size_t out_cap,
                         m4g_histo *d_histos, void *hip_stream);

/* Same as m4g_batch_run_device, and additionally writes one m4g_code per match
 * to d_codes (device memory, out_cap records, indexed like d_out): the
 * per-match results of Encoder.Encode's first pass (encoder.go:44-105) that
 * its second pass consumes, so a host encoder fed from d_out + d_histos +
 * d_codes only has to build the three prefix codes and write bits.
 * d_codes == NULL behaves exactly like m4g_batch_run_device; d_codes without
 * d_histos is M4G_ERR_BAD_PARAM. */
int m4g_batch_run_device_codes(m4g_batch *b, const uint8_t *d_src,
                               const uint64_t *chunk_off, size_t nchunks,
                               m4g_match *d_out, uint64_t *d_out_off,
                               size_t out_cap, m4g_histo *d_histos,
                               m4g_code *d_codes, void *hip_stream);

/* Host-memory form over one or more devices: chunk c goes to
 * devices[c % ndevices]; per-chunk Match lists and histograms come back over
 * pinned async copies and are concatenated in chunk order on the host.
 * No inter-GPU communication.  Returns M4G_OK, or -(needed +
 * M4G_STREAM_NEEDED_BIAS) when out_cap is too small (out_off[nchunks] holds
 * the count as well), or an M4G_ERR_*. */
int64_t m4g_find_matches_batch(const m4g_params *params, const uint8_t *base,
                               const uint64_t *chunk_off, size_t nchunks,
                               const int *devices, int ndevices, m4g_match *out,
                               uint64_t *out_off, size_t out_cap,
                               m4g_histo *histos);

/* The same call on a persistent handle: plans, device buffers, page-locked
 * staging, streams and events are created ONCE for up to max_chunks chunks of
 * up to max_chunk_bytes bytes over the listed devices (chunk c on
 * devices[c % ndevices], one host thread per device), so a run pays for copies
 * and kernels only -- m4g_find_matches_batch allocates and frees all of it on
 * every call (tens of GiB of workspace for a few thousand chunks).  One
 * device: the batch is cut into slices whose upload, kernels and download
 * overlap.  Semantics, results and return values as m4g_find_matches_batch.
 * One thread at a time per handle.  No counterpart in the reference beyond the
 * per-chunk semantics cited above (AutoReset + Encoder pass 1). */
typedef struct m4g_hostbatch m4g_hostbatch;
int m4g_hostbatch_create(const m4g_params *params, const int *devices, int ndevices,
                         size_t max_chunks, size_t max_chunk_bytes, m4g_hostbatch **out);
int64_t m4g_hostbatch_run(m4g_hostbatch *h, const uint8_t *base, const uint64_t *chunk_off,
                          size_t nchunks, m4g_match *out, uint64_t *out_off, size_t out_cap,
                          m4g_histo *histos);
void m4g_hostbatch_destroy(m4g_hostbatch *h);

/* Chunks in host memory -> ONE compressed stream in host memory, on the
 * persistent handle: for every chunk c, in order,
 * what matchfinder.Writer.writeBlock does (matchfinder.go:81-87) with
 * MatchFinder = AutoReset{&M4{params}} and Encoder = &brotli.Encoder{}
 * (M4G_FORMAT_BROTLI), flate.NewEncoder() (M4G_FORMAT_FLATE) or
 * flate.NewGZIPEncoder() (M4G_FORMAT_GZIP; gzip_mtime for time.Now()), then
 * Close() with nothing buffered: one meta-block / deflate block per chunk,
 * byte-identical to m4g_batch_encode / m4g_batch_flate_encode over the same
 * chunks.  The Match lists never leave the device, so the link carries the
 * input one way and the compressed stream the other (the host-memory batch
 * call is bound by the 12-byte Match records it brings back).  A handle made
 * with several devices gives device g the g-th contiguous range of chunks: all
 * devices find matches at once, and the stream's one bit writer passes over
 * the slices of device 0, then device 1, ...; each slice is encoded on the
 * device that holds its Match lists at the bit position where the previous
 * one stopped.  The bytes do not depend on the device list.  Returns the
 * stream length, -(bound + M4G_STREAM_NEEDED_BIAS) when stream_cap is too small
 * (bound: a capacity that always suffices), or an M4G_ERR_*. */
int64_t m4g_hostbatch_compress(m4g_hostbatch *h, int format, uint32_t gzip_mtime,
                               const uint8_t *base, const uint64_t *chunk_off, size_t nchunks,
                               uint8_t *stream_out, size_t stream_cap);

/* Replaces: brotli.Encoder.Encode(dst, chunk, matches, false) called for every
 * chunk of the batch in chunk order on ONE Encoder (encoder.go:17-161), i.e.
 * what matchfinder.Writer does after FindMatches (matchfinder.go:84) -- the
 * serial stage that bounds a V2 pipeline.  Input: the device-resident results
 * of the plan's last m4g_batch_run_device_codes call on these chunks (d_out,
 * d_out_off, d_histos, d_codes; d_src as in that call).  Pass 1 (encoder.go:
 * 44-105) was done by that call; here the host builds the three prefix codes
 * of every chunk from its histograms (encoder.go:107-120, on host threads) and
 * the GPU writes every chunk's commands, literals and distances (encoder.go:
 * 122-154) at their final bit positions of one stream (bitwriter.go:19-33:
 * meta-blocks are not byte aligned).
 *   flags bit 0: this call starts a stream: write the stream header
 *                (encoder.go:19-22); bit 1: it also ends it: ISLAST + ISEMPTY
 *                and padding to a byte (encoder.go:156-159).  Both set = one
 *                complete brotli stream holding the chunks as consecutive
 *                meta-blocks (Writer.Write(all chunks); Writer.Close()).
 *                Without bit 1 the last byte may be partial (zero padded): a
 *                caller that continues the stream must carry those bits over.
 *   stream_out : host memory, stream_cap bytes.  Returns the stream length in
 *                bytes, -(needed + M4G_STREAM_NEEDED_BIAS) when stream_cap is too
 *                small, or an M4G_ERR_*.
 *   ms_out     : nullable, 3 floats: host code building, K6 kernel, whole call (ms).
 * Synchronises hip_stream.  The bytes equal Encoder.Encode's up to the order in
 * which chooseBitDepths (brotli_bit_stream.go:571) sorts symbols of equal count
 * (Go's unstable sort; ascending symbol here): always a valid stream. */
int64_t m4g_batch_encode(m4g_batch *b, const uint8_t *d_src, size_t nchunks,
                         const m4g_match *d_out, const uint64_t *d_out_off,
                         const m4g_histo *d_histos, const m4g_code *d_codes,
                         int flags, uint8_t *stream_out, size_t stream_cap,
                         void *hip_stream, float *ms_out);

/* Replaces: flate.NewEncoder().Encode(dst, chunk, matches, false) -- or
 * flate.NewGZIPEncoder()'s, flags bit 2 -- called for every chunk of the batch
 * in chunk order on ONE encoder (flate/huffman_bit_writer.go:506-517,
 * flate/gzip.go:39-65), i.e. what matchfinder.Writer does after FindMatches when
 * it was made by flate.NewWriter / NewGZIPWriter (flate/writer.go:14-66;
 * levels 5-7 use matchfinder.M4 with MaxDistance 1<<15, HashLen 5,
 * ChainLength 8 / 16 / 32, DistanceBitCost 66).  Input: the device-resident
 * Match lists of the plan's last m4g_batch_run_device[_codes] call on these
 * chunks.  makeStatistics (huffman_bit_writer.go:425-446) and writeTokens
 * (472-504) run on the GPU; the code construction (huffman_code.go), the choice
 * between a stored, a fixed and a dynamic block and the block headers
 * (huffman_bit_writer.go:361-415) on host threads.  Every chunk becomes one
 * deflate block; blocks follow each other at bit granularity as in the
 * reference (the writer keeps its pending bits between Encode calls).
 *   flags bit 0: this call starts a stream (gzip: the 10-byte header with
 *                gzip_mtime where the reference writes time.Now(), gzip.go:46);
 *         bit 1: it also ends it, as Writer.Close() with nothing buffered does
 *                (matchfinder.go:89-93): an empty last block (which comes out as
 *                a fixed block), flush to a byte and, for gzip, CRC-32 and
 *                length of all chunks (computed on the GPU, joined on the host);
 *         bit 2: gzip framing (flate/gzip.go) instead of raw deflate.
 *                Without bit 1 the last byte may be partial (zero padded).
 *   kinds_out  : nullable, one byte per chunk: 0 stored, 1 fixed, 2 dynamic.
 *   ms_out     : nullable, 3 floats: host code building, token kernel, whole call.
 * A match with Length outside 3..258 or Distance outside 1..32768 makes the
 * reference panic (flate/token.go:59-71: nothing limits M4's match lengths);
 * here the call fails with M4G_ERR_TOKEN_RANGE and names the chunk.
 * Returns the stream length in bytes, -(needed + M4G_STREAM_NEEDED_BIAS) when
 * stream_cap is too small, or an M4G_ERR_*.  Synchronises hip_stream.  The
 * encoder has no sort without a total order, so the bytes are meant to equal
 * the Go encoder's (unpinned here: no Go toolchain; pinned against the
 * restated encoder and by inflating). */
int64_t m4g_batch_flate_encode(m4g_batch *b, const uint8_t *d_src, size_t nchunks,
                               const m4g_match *d_out, const uint64_t *d_out_off,
                               int flags, uint32_t gzip_mtime, uint8_t *stream_out,
                               size_t stream_cap, void *hip_stream, float *ms_out,
                               uint8_t *kinds_out);

/* Replaces: the encoder half of matchfinder.Writer{MatchFinder: &M4{...},
 * Encoder: &brotli.Encoder{}, BlockSize: block_size} over one whole input
 * (the writer the reference's own tests build, brotli_test.go:626-649,
 * 730-744): Write(src) encodes every full block as one meta-block, Close() the
 * remainder -- possibly empty -- followed by ISLAST + ISEMPTY
 * (matchfinder.go:59-93, encoder.go:17-161).  matches / block_off are what
 * m4g_find_matches_blocks(ctx, src, n, block_size, ...) returned for the same
 * bytes.  Host memory in, host memory out; both passes of Encoder.Encode run
 * on `device` (K5 with the per-match codes, then K6), the prefix codes on host
 * threads.  Returns the stream length, -(needed + M4G_STREAM_NEEDED_BIAS), or
 * an M4G_ERR_*.  Byte identity with Go: as m4g_batch_encode. */
int64_t m4g_brotli_encode_blocks(int device, const uint8_t *src, size_t n, size_t block_size,
                                 const m4g_match *matches, const uint64_t *block_off,
                                 uint8_t *stream_out, size_t stream_cap);

/* Replaces: the encoder half of flate.NewWriter(dst, level) / NewGZIPWriter
 * (flate/writer.go:14-66) over one whole input: Writer.Write(src) encodes every
 * full block of block_size bytes as a non-final deflate block, Writer.Close()
 * the remainder -- possibly empty -- as the final one (matchfinder.go:59-93).
 * matches / block_off are what m4g_find_matches_blocks(ctx, src, n, block_size,
 * ...) returned for the same bytes (block_off has ceil(n / block_size) + 1
 * entries), i.e. the Match lists of the block-after-block FindMatches calls the
 * Writer makes.  Host memory in, host memory out; the symbol counts, tokens and
 * (gzip != 0) CRC-32 are computed on `device`.  With the level's M4 fields
 * (see m4g_batch_flate_encode) the result is the stream the reference's writer
 * produces for that level, gzip_mtime standing in for time.Now().  Returns the
 * stream length, -(needed + M4G_STREAM_NEEDED_BIAS), M4G_ERR_TOKEN_RANGE where
 * the reference panics, or another M4G_ERR_*.  kinds_out: nullable, one byte
 * per block (n / block_size + 1 of them). */
int64_t m4g_flate_encode_blocks(int device, const uint8_t *src, size_t n, size_t block_size,
                                const m4g_match *matches, const uint64_t *block_off,
                                int gzip, uint32_t gzip_mtime, uint8_t *stream_out,
                                size_t stream_cap, uint8_t *kinds_out);

/* Replaces: matchfinder.Writer (matchfinder/matchfinder.go:44-103) -- the type
 * brotli.NewWriterV2, flate.NewWriter and flate.NewGZIPWriter return and the
 * reference's tests build by hand -- with BOTH of its plug-ins on the GPU:
 *   MatchFinder = &matchfinder.M4{params}                       (m4.go)
 *   Encoder     = &brotli.Encoder{}            M4G_FORMAT_BROTLI (encoder.go)
 *                 flate.NewEncoder()           M4G_FORMAT_FLATE  (flate/huffman_bit_writer.go)
 *                 flate.NewGZIPEncoder()       M4G_FORMAT_GZIP   (flate/gzip.go; gzip_mtime for time.Now())
 *   BlockSize   = block_size (0: every Write is one block, matchfinder.go:64-66)
 * m4g_writer_write = Writer.Write: appends to the pending input, encodes every
 * full block, keeps the rest (lines 59-79); m4g_writer_close = Writer.Close: the
 * remainder, possibly empty, as the last block (lines 89-93); m4g_writer_reset
 * = Writer.Reset (lines 95-103).  Where the reference hands bytes to Dest.Write,
 * these calls return them in `out`.  The CONCATENATION of what the calls return,
 * Close included, is the reference Writer's output for the same sequence of
 * calls.  brotli: every call also returns exactly what the reference has written
 * by then (whole bytes; fewer than 8 pending bits stay inside, bitwriter.go).
 * flate / gzip: the split between calls can differ by up to 5 bytes -- the
 * reference's huffmanBitWriter hands bytes on in groups of 6 once 48 bits are
 * pending (flate/huffman_bit_writer.go:110-126), this one keeps fewer than 8
 * bits.  With BlockSize 0 an empty Write is a block too (flate: an empty stored
 * or fixed block), EXCEPT when output of the previous call is still kept: then
 * n = 0 only fetches it (the retry below).  A call whose blocks qualify for the
 * one-launch path of m4g_find_matches_blocks (two or more blocks of >= 3847
 * bytes) finds the matches and encodes without moving Match lists or bytes over
 * PCIe a second time.  If out_cap is too small the call returns -(needed +
 * M4G_STREAM_NEEDED_BIAS) with the input consumed and the output kept: call
 * again with n = 0 (write) or simply again (close) and a larger buffer.
 * flate formats: M4G_ERR_TOKEN_RANGE where the reference panics (a match longer
 * than 258 bytes, flate/token.go:59-71); the writer is then unusable until
 * m4g_writer_reset.  One thread at a time per writer. */
#define M4G_FORMAT_BROTLI 0
#define M4G_FORMAT_FLATE 1
#define M4G_FORMAT_GZIP 2
typedef struct m4g_writer m4g_writer;
int m4g_writer_create(const m4g_params *params, int device, int format, size_t block_size,
                      uint32_t gzip_mtime, m4g_writer **out);
int64_t m4g_writer_write(m4g_writer *w, const uint8_t *src, size_t n, uint8_t *out, size_t out_cap);
int64_t m4g_writer_close(m4g_writer *w, uint8_t *out, size_t out_cap);
int m4g_writer_reset(m4g_writer *w);
void m4g_writer_destroy(m4g_writer *w);

/* Kernel-side outcome of the plan's most recent m4g_batch_run_device[_codes]
 * call.  Those calls only enqueue work and return before it runs, so conditions
 * the kernels detect cannot come back through their return value: this call
 * synchronises the stream of that run and returns M4G_OK, M4G_ERR_REPEAT_RANGE
 * or M4G_ERR_INTERNAL (records were dropped; d_out / d_out_off of that run must
 * not be used).  It is also where a run that overflowed the plan's compact
 * record slots is repeated (m4g_batch_info).  Has no counterpart in the reference (FindMatches cannot fail,
 * m4.go:63); callers of the device-resident form check it once per batch, the
 * host-memory forms check it themselves. */
int m4g_batch_status(m4g_batch *b);

/* Device memory of a plan and the layout of its record slots; no counterpart in
 * the reference.  A plan from m4g_batch_create holds about 8.3 bytes of
 * workspace per input byte it is sized for (4 for the chain entries, 4 for the
 * record slots, which also serve the chain build as scratch): the slots take
 * one Match per 8 positions of a sub-block, which text, logs and random bytes
 * of 2 to 4 symbols stay below.  An input with more (the
 * bound is one per MinLength positions) overflows them; m4g_batch_status -- and
 * m4g_batch_encode / m4g_batch_flate_encode, which settle the run first -- then
 * switch the plan for good to slots of the full bound, which the same memory
 * holds for out[3] chunks at a time, and run the batch again on the same stream
 * into the same buffers before returning.  So the outputs of a run are final
 * once one of these calls has returned M4G_OK, not before.  M4G_SLOTS=worst in
 * the environment: plans start in the full layout (about 12.4 bytes per byte).
 * out[0] = bytes of device memory, out[1] = Match records per sub-block slot,
 * out[2] = the full bound, out[3] = chunks parsed at a time (0: all), out[4] =
 * runs repeated so far (0 or 1).  Returns the number of values written.
 * out[5] = 1 when the plan evaluates the chain walks in a kernel of their own
 * (K2; ChainLength > 0, chunks below 4 MiB, DistanceBitCost >= 0), out[6] =
 * the device memory K2 and the sorted position list take (included in out[0]),
 * out[7] = 1 when the plan builds that list and its parse reads the candidates
 * of a walk from it (ChainLength > 0, chunks below 4 MiB; k2_eval.h). */
int m4g_batch_info(m4g_batch *b, uint64_t *out, int cap);
/* Test hooks for K2, the data-parallel form of the chain walks of m4.go:156-181
 * (no counterpart in the reference: results are the same with and without).
 * _set_k2: a plan that holds K2's buffers runs with (on = 1) or without it from
 * the next run on; returns 1 when the plan holds them, else 0.  _dump_k2: after
 * a run, chunk c's chain entries and per-position K2 results (4 words each),
 * n <= the chunk's length of each, to host memory (either may be NULL). */
int m4g_debug_batch_set_k2(m4g_batch *b, int on);
/* The same switch for the sorted position list a chain-walking plan builds
 * (out[7] of m4g_batch_info; the parse chases the chains when it is off). */
int m4g_debug_batch_set_sorted_walk(m4g_batch *b, int on);
int m4g_debug_batch_dump_k2(m4g_batch *b, size_t c, uint32_t *chain_out, uint32_t *res_out, size_t n);

/* Per-stage kernel time of the plan's most recent run, measured with HIP
 * events on the stream the kernels ran on.  Synchronises that stream.
 * ms[0] = chain build (K1), ms[1] = parse (K3), ms[2] = compaction (K4),
 * ms[3] = histograms (K5).  Returns the number of stages written. */
int m4g_batch_last_timings(m4g_batch *b, float *ms, int cap);
/* Enable (1) / disable (0) the event recording used by the call above. */
void m4g_batch_set_timing(m4g_batch *b, int enabled);

/* ---- the entry point that allocates nothing ------------------------------
 * No counterpart in the reference.  It exists for callers that cannot call
 * hipMalloc: code under HIP stream / graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same semantics and results as
 * m4g_batch_run_device on a plan (fresh M4 state per chunk, optional
 * histograms), but the workspace is a fixed array inside the library's code
 * object (about 14 MiB, held by every GPU the library is used on until the
 * process ends) instead of a plan, so the call makes no hipMalloc / hipFree and
 * no host-to-device copy: it only enqueues kernels and one 64-byte memset on
 * hip_stream and returns without synchronising.
 * Limits (M4G_ERR_TOO_LARGE beyond them): */
#define M4G_NOALLOC_MAX_CHUNKS 64        /* chunks per call */
#define M4G_NOALLOC_MAX_BYTES (1u << 20) /* bytes per call, all chunks together */
#define M4G_NOALLOC_MAX_BLOCKS 1088      /* nchunks x ceil(largest chunk / 1024) */
/* (with MinLength < 4 a sub-block needs more record slots: the product above
 * must then stay below 1088 x 274 / (1024 / MinLength + 18)).
 * The FIRST call on a device does one-time set-up there (address of the array,
 * a kernel self-test) and synchronises: make it before capturing.
 * All calls on one GPU share that GPU's array, which keeps the last call's
 * contents: two calls whose work overlaps in time on ONE GPU (two streams, two
 * threads) race -- serialise them; calls on different GPUs are independent.
 * Within the limits no kernel-side error can occur (no seeded repeat distance,
 * record slots sized for the worst case), so there is no status call. */
int m4g_batch_run_device_noalloc(const m4g_params *params, int device, const uint8_t *d_src,
                                 const uint64_t *chunk_off, size_t nchunks,
                                 m4g_match *d_out, uint64_t *d_out_off, size_t out_cap,
                                 m4g_histo *d_histos, void *hip_stream);
/* Test hook for the entry point above: fills this device's copy of the scratch
 * array with `value` (a memset on the symbol's address, no allocation). */
int m4g_debug_noalloc_poison(int device, int value, void *hip_stream);

/* ---- matchfinder.Bargain1 / Bargain2 / Bargain3 -------------
 * The shortest-path match finders that brotli.NewWriterV2 plugs into
 * matchfinder.Writer for levels 5-9 (writer.go:142-150; Bargain1:
 * matchfinder/bargain1.go:15-31, FindMatches :38-208; Bargain2: bargain2.go;
 * Bargain3: bargain3.go).
 * The Writer hands FindMatches one block of BlockSize bytes at a time
 * (matchfinder.go:59-93); what connects the calls is the hash table and the
 * history.  Here a CHUNK is one such stream (fresh state: Reset(),
 * bargain1.go:33-36) and a launch does every block of every chunk: K1 builds
 * the tables' predecessor links for all positions, then one wavefront per block
 * runs that block's search (float32 costs in the reference's operation order).
 * Output: the Match lists of all blocks in (chunk, block) order and
 * block_off[k] = index of block k's first Match (nblocks + 1 entries; blocks of
 * chunk c: ceil(len_c / block_size), m4g_bargain_block_count() sums them).
 */
typedef struct m4g_bargain_params {
  int32_t variant;      /* 1 = Bargain1, 2 = Bargain2, 3 = Bargain3 */
  int32_t max_distance; /* MaxDistance; 0 = 1 << 16 (bargain1.go:39-41); NewWriterV2 uses 1 << 20 */
  int32_t skip;         /* Skip (bargain1.go:20-22, 159); with an odd block_size or max_distance only for
                         * chunks of at most 2 x max_distance bytes (a slide of odd length would flip the
                         * parity the reference tests): M4G_ERR_BAD_PARAM otherwise */
  int32_t block_size;   /* Writer.BlockSize; 0 = 1 << 16 (writer.go:156); 3851 .. 1 << 24 */
} m4g_bargain_params;
typedef struct m4g_bargain m4g_bargain;

/* Plan: device workspace for launches of up to max_chunks chunks, max_total_bytes
 * bytes together, no chunk longer than max_chunk_bytes (< 2 GiB).  Workspace:
 * 4 B per byte and table (links) + 4 B per byte (staged Matches) + 1 KiB per
 * block + 12 B x block_size per resident wavefront (16 per CU). */
int m4g_bargain_create(const m4g_bargain_params *params, int device, size_t max_chunks,
                       size_t max_total_bytes, size_t max_chunk_bytes, m4g_bargain **out);
void m4g_bargain_destroy(m4g_bargain *g);
size_t m4g_bargain_block_count(const m4g_bargain *g, const uint64_t *chunk_off, size_t nchunks);
/* d_src / d_out / d_block_off: device memory; chunk_off: host, nchunks + 1
 * entries.  Returns the number of Matches, -(needed + M4G_STREAM_NEEDED_BIAS)
 * when out_cap is too small (d_block_off is valid then, d_out is not), or an
 * M4G_ERR_*.  A Match is at least 3 bytes long and a block adds at most one
 * literal-only record: out_cap = total / 3 + 2 x nblocks always suffices.
 * Synchronises the stream before it returns. */
int64_t m4g_bargain_run_device(m4g_bargain *g, const uint8_t *d_src, const uint64_t *chunk_off,
                               size_t nchunks, m4g_match *d_out, size_t out_cap,
                               uint64_t *d_block_off, void *hip_stream);
/* Stage durations of the last run in ms: links (K1), byte costs, search, compaction. */
int m4g_bargain_last_timings(const m4g_bargain *g, float *ms, int cap);
/* One stream from host memory = the FindMatches calls of a Writer with this
 * match finder over `src` (one Write, then Close).  block_off: host,
 * ceil(n / block_size) + 1 entries. */
int64_t m4g_bargain_find_blocks(const m4g_bargain_params *params, int device, const uint8_t *src,
                                size_t n, m4g_match *out, size_t out_cap, uint64_t *block_off);

/* Replaces: matchfinder.Bargain1 / Bargain2 / Bargain3 AS A matchfinder.MatchFinder
 * (bargain1.go:33-38 Reset, :38-208 FindMatches; writer.go:142-150 plugs
 * &Bargain1{MaxDistance: 1 << 20} ... into matchfinder.Writer, which then calls
 * FindMatches block after block, matchfinder.go:81-87).  A handle carries the
 * stream from call to call: m4g_bargain_ctx_find_blocks(ctx, block, n, ...)
 * with n <= block_size is FindMatches(dst[:0], block); with a longer src it is
 * that call for every consecutive block of params->block_size bytes (the
 * throughput form: one wavefront per block, all blocks of the call in one
 * launch), block b's Matches in out[block_off[b] .. block_off[b + 1]).
 * m4g_bargain_ctx_reset = Reset.  The blocks of a stream must have the shape
 * matchfinder.Writer gives them: block_size bytes each, a shorter one only as
 * the last before Reset (M4G_ERR_BAD_PARAM otherwise: the finder leaves the last
 * 7 / 11 positions of every block out of its tables, bargain1.go:136-139, and
 * the handle keeps no list of block ends).  Between calls the handle keeps
 * the stream's last whole blocks, MaxDistance bytes or more, on the host and
 * re-inserts them (without searching them) in front of the new blocks: a
 * history slide (bargain1.go:66-80) leaves no other trace.  Safe out_cap:
 * n / 3 + 2 * blocks + 8.  Returns the number of Matches, -(needed +
 * M4G_STREAM_NEEDED_BIAS), or an M4G_ERR_*.  One thread at a time per handle. */
typedef struct m4g_bargain_ctx m4g_bargain_ctx;
int m4g_bargain_ctx_create(const m4g_bargain_params *params, int device, m4g_bargain_ctx **out);
void m4g_bargain_ctx_destroy(m4g_bargain_ctx *ctx);
int m4g_bargain_ctx_reset(m4g_bargain_ctx *ctx);
int64_t m4g_bargain_ctx_find_blocks(m4g_bargain_ctx *ctx, const uint8_t *src, size_t n, m4g_match *out,
                                    size_t out_cap, uint64_t *block_off /* nullable */);

/* m4g_bargain_find_blocks with the stream's blocks spread over `devices`
 * (protocol P1): every device gets a contiguous range of blocks
 * and the whole blocks before it (MaxDistance bytes or more) as history; no
 * device talks to another, the lists are joined in stream order on the host. */
int64_t m4g_bargain_find_blocks_multi(const m4g_bargain_params *params, const int *devices, int ndevices,
                                      const uint8_t *src, size_t n, m4g_match *out, size_t out_cap,
                                      uint64_t *block_off);

/* Replaces: brotli.NewWriterV2(dst, level) for levels 5-9 (writer.go:126-161) as a streaming handle: the returned
 * m4g_writer's m4g_writer_write / _close / _reset / _destroy are the Write / Close / Reset of the
 * matchfinder.Writer{Bargain1 | Bargain2{Skip} | Bargain2 | Bargain3, &Encoder{}, BlockSize 1 << 16} the reference
 * builds, with the same bytes after the same calls.  A Write that completes many blocks does them in one launch
 * (one wavefront per block); the handle keeps the last MaxDistance bytes of the stream (whole blocks) for the next
 * call.  Levels above 9 mean 9; levels 0-4 are M4G_ERR_BAD_PARAM (other match finders, not built). */
int m4g_writer_create_v2(int level, int device, m4g_writer **out);

/* Replaces: w := brotli.NewWriterV2(dst, level); w.Write(src); w.Close() for levels 5-9 (writer.go:126-161:
 * matchfinder.Writer{MatchFinder: Bargain1 | Bargain2{Skip} | Bargain2 | Bargain3, MaxDistance 1 << 20,
 * Encoder: &Encoder{}, BlockSize: 1 << 16}): the bytes go to the device once, the Match lists never leave it.
 * Host memory in (at most 2 GiB), the complete brotli stream out.  Levels above 9 mean 9 (writer.go:129-131); levels 0-4 use
 * other match finders (ZFast, ZDFast, ZM, Trio), which are not built: M4G_ERR_BAD_PARAM.  Returns the stream
 * length, -(needed + M4G_STREAM_NEEDED_BIAS) when out_cap is too small, or an M4G_ERR_*. */
int64_t m4g_brotli_compress_v2(int level, int device, const uint8_t *src, size_t n, uint8_t *out,
                               size_t out_cap);

/* Host staging helpers for callers without a HIP binding (bench.py uses
 * torch tensors instead; the Go shim uses these): page-locked host memory.
 * Buffers handed to m4g_find_matches_batch / _blocks should come from here
 * when throughput matters: from pageable memory the HIP runtime stages every
 * copy (measured 512 x 1 MiB chunks: 6 GB/s end to end, 21 GB/s from these
 * buffers; hipHostRegister of a caller's buffer is refused on some hosts, so
 * the library cannot pin for the caller). */
void *m4g_host_alloc(size_t bytes);
void m4g_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* M4G_H */
