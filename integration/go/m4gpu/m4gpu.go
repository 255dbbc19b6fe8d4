// Package m4gpu is matchfinder.M4 on an AMD MI355X: a cgo binding of libm4g
// (include/m4g.h) that implements matchfinder.MatchFinder
// (matchfinder/matchfinder.go:23-30), so matchfinder.Writer, brotli.Encoder and
// the flate encoders of andybalholm/brotli keep working unchanged.
//
// NOT COMPILED in the build image of this repository (it has no Go toolchain):
// the same call sequences are exercised from C++ (include/m4g_matchfinder.hpp,
// tests/sim/host_mirror_check.cpp) and from Python (brotli_amd/m4.py).
//
// Place this directory at matchfinder/m4gpu inside a checkout of the reference
// and point the two cgo paths below at this repository's include/ and
// brotli_amd/ (or install libm4g.so and m4g.h system-wide).
package m4gpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/m4g/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/m4g/lib -lm4g -Wl,-rpath,${SRCDIR}/../../third_party/m4g/lib
#include <stdlib.h>
#include "m4g.h"
*/
import "C"

import (
	"fmt"
	"io"
	"runtime"
	"unsafe"

	"github.com/andybalholm/brotli/matchfinder"
)

// M4 has the exported fields of matchfinder.M4 (matchfinder/m4.go:15-43) plus
// the HIP device to run on.  Zero fields take the reference's defaults
// (m4.go:64-75).  Like matchfinder.M4 it must not be used by two goroutines at
// once; distinct values are independent.
type M4 struct {
	MaxDistance, MinLength, HashLen, TableBits, ChainLength, DistanceBitCost int
	Device                                                                 int

	ctx *C.m4g_ctx
	buf []C.m4g_match
}

var _ matchfinder.MatchFinder = (*M4)(nil)

func fail(where string, rc C.int) {
	panic("m4gpu: " + where + ": " + C.GoString(C.m4g_strerror(rc)) + ": " + C.GoString(C.m4g_last_error()))
}

func (q *M4) params() C.m4g_params {
	return C.m4g_params{
		max_distance: C.int32_t(q.MaxDistance), min_length: C.int32_t(q.MinLength),
		hash_len: C.int32_t(q.HashLen), table_bits: C.int32_t(q.TableBits),
		chain_length: C.int32_t(q.ChainLength), distance_bit_cost: C.int32_t(q.DistanceBitCost),
	}
}

func (q *M4) minLength() int {
	if q.MinLength == 0 {
		return 4 // m4.go:67-69
	}
	return q.MinLength
}

func (q *M4) init() {
	if q.ctx != nil {
		return
	}
	p := q.params()
	if rc := C.m4g_create(&p, C.int(q.Device), &q.ctx); rc != 0 {
		fail("m4g_create", rc)
	}
	runtime.SetFinalizer(q, func(q *M4) { C.m4g_destroy(q.ctx) })
}

// Reset implements matchfinder.MatchFinder (m4.go:51-57).
func (q *M4) Reset() {
	if q.ctx != nil {
		if rc := C.m4g_reset(q.ctx); rc != 0 {
			fail("m4g_reset", rc)
		}
	}
}

// FindMatches implements matchfinder.MatchFinder (m4.go:63-283): same
// streaming semantics (history, hash table and chain live on the device and
// slide as in m4.go:82-99), same Match list.
func (q *M4) FindMatches(dst []matchfinder.Match, src []byte) []matchfinder.Match {
	q.init()
	need := len(src)/q.minLength() + 2
	if len(q.buf) < need {
		q.buf = make([]C.m4g_match, need)
	}
	var seed C.uint32_t
	if len(dst) > 0 { // m4.go:129-130 reads the caller's last Distance
		seed = C.uint32_t(dst[len(dst)-1].Distance)
	}
	var p *C.uint8_t
	if len(src) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&src[0])) // borrowed for the call only; the library copies it
	}
	n := C.m4g_find_matches(q.ctx, p, C.size_t(len(src)), seed, &q.buf[0], C.size_t(len(q.buf)))
	if n < 0 {
		fail("m4g_find_matches", C.int(n))
	}
	for _, m := range q.buf[:n] { // widen 3 x u32 to matchfinder.Match (3 x int)
		dst = append(dst, matchfinder.Match{
			Unmatched: int(m.unmatched), Length: int(m.length), Distance: int(m.distance)})
	}
	return dst
}

// FindMatchesBlocks returns what FindMatches(nil, block) returns for each of
// the consecutive blockSize-byte blocks of src (the last may be shorter), all
// computed in one launch; blockOff[b]:blockOff[b+1] delimits block b's list.
// The handle ends in the state the block-by-block calls would leave.  This is
// what a buffering variant of matchfinder.Writer.Write (matchfinder.go:59-79)
// calls instead of one FindMatches per block (a 64 KiB call is latency bound).
func (q *M4) FindMatchesBlocks(src []byte, blockSize int) (matches []matchfinder.Match, blockOff []uint64) {
	q.init()
	nblocks := (len(src) + blockSize - 1) / blockSize
	need := 0
	for b := 0; b < nblocks; b++ {
		need += min(blockSize, len(src)-b*blockSize)/q.minLength() + 2
	}
	if len(q.buf) < need+1 {
		q.buf = make([]C.m4g_match, need+1)
	}
	blockOff = make([]uint64, nblocks+1)
	var p *C.uint8_t
	if len(src) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&src[0]))
	}
	n := C.m4g_find_matches_blocks(q.ctx, p, C.size_t(len(src)), C.size_t(blockSize),
		&q.buf[0], C.size_t(len(q.buf)), (*C.uint64_t)(unsafe.Pointer(&blockOff[0])))
	if n < 0 {
		fail("m4g_find_matches_blocks", C.int(n))
	}
	matches = make([]matchfinder.Match, n)
	for i, m := range q.buf[:n] {
		matches[i] = matchfinder.Match{Unmatched: int(m.unmatched), Length: int(m.length), Distance: int(m.distance)}
	}
	return
}

// Histograms are the three histograms pass 1 of brotli.Encoder.Encode builds
// (encoder.go:33-35), as the GPU returns them per chunk.
type Histograms struct {
	Literal  [256]uint32
	Command  [704]uint32
	Distance [64]uint32
}

// Batch runs `AutoReset{&M4{...}}.FindMatches(nil, chunk)` (textencoder.go:46-53)
// for every chunk data[chunkOff[c]:chunkOff[c+1]] on the given devices (chunk c
// on devices[c % len(devices)], no inter-GPU traffic) and returns the Match
// lists concatenated in chunk order with their offsets, plus (optionally) each
// chunk's Encoder histograms.  data and the returned slices are ordinary Go
// memory; for the 20+ GB/s path allocate them with m4g_host_alloc (see
// INTEGRATION.md section 2).
func (q *M4) Batch(data []byte, chunkOff []uint64, devices []int, wantHistos bool) (
	matches []matchfinder.Match, outOff []uint64, histos []Histograms) {
	nchunks := len(chunkOff) - 1
	if nchunks <= 0 {
		return nil, []uint64{0}, nil
	}
	need := 1
	for c := 0; c < nchunks; c++ {
		need += int(chunkOff[c+1]-chunkOff[c])/q.minLength() + 2
	}
	out := make([]C.m4g_match, need)
	outOff = make([]uint64, nchunks+1)
	devs := make([]C.int, len(devices))
	for i, d := range devices {
		devs[i] = C.int(d)
	}
	var hp *C.m4g_histo
	if wantHistos {
		histos = make([]Histograms, nchunks)
		hp = (*C.m4g_histo)(unsafe.Pointer(&histos[0]))
	}
	p := q.params()
	var dp *C.uint8_t
	if len(data) > 0 {
		dp = (*C.uint8_t)(unsafe.Pointer(&data[0]))
	}
	rc := C.m4g_find_matches_batch(&p, dp, (*C.uint64_t)(unsafe.Pointer(&chunkOff[0])), C.size_t(nchunks),
		&devs[0], C.int(len(devs)), &out[0], (*C.uint64_t)(unsafe.Pointer(&outOff[0])), C.size_t(len(out)), hp)
	if rc < -C.M4G_STREAM_NEEDED_BIAS { // cannot happen with the worst-case capacity above; the count is -(rc + bias)
		panic(fmt.Sprintf("m4gpu: m4g_find_matches_batch: out too small, %d matches", -int64(rc)-C.M4G_STREAM_NEEDED_BIAS))
	}
	if rc != 0 {
		fail("m4g_find_matches_batch", C.int(rc))
	}
	matches = make([]matchfinder.Match, outOff[nchunks])
	for i := range matches {
		m := out[i]
		matches[i] = matchfinder.Match{Unmatched: int(m.unmatched), Length: int(m.length), Distance: int(m.distance)}
	}
	return
}

// FlateCompress is flate.NewWriter(&buf, level) -- gz: flate.NewGZIPWriter -- followed by w.Write(data);
// w.Close(), for the levels whose match finder is matchfinder.M4 (5, 6, 7: flate/writer.go:33-46).  Match finding
// for all 64 KiB blocks in one launch (history carried over, as the Writer's block-after-block FindMatches calls),
// then the flate encoder on the GPU (m4g_flate_encode_blocks).  mtime stands in for the time.Now() of the gzip
// header (flate/gzip.go:46).  Panics where the reference panics: a match longer than 258 bytes (flate/token.go:60).
func FlateCompress(data []byte, level int, gz bool, mtime uint32, device int) []byte {
	chain, ok := map[int]int{5: 8, 6: 16, 7: 32}[level]
	if !ok {
		panic("m4gpu: flate levels 5..7 use matchfinder.M4")
	}
	q := &M4{MaxDistance: 1 << 15, HashLen: 5, ChainLength: chain, DistanceBitCost: 66, Device: device}
	const bs = 1 << 16 // flate/writer.go:65
	q.init()
	nb := (len(data) + bs - 1) / bs
	ms := make([]C.m4g_match, len(data)/4+2*nb+2)
	off := make([]C.uint64_t, nb+1)
	var p *C.uint8_t
	if len(data) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&data[0]))
	}
	if n := C.m4g_find_matches_blocks(q.ctx, p, C.size_t(len(data)), bs, &ms[0], C.size_t(len(ms)), &off[0]); n < 0 {
		fail("m4g_find_matches_blocks", C.int(n))
	}
	out := make([]byte, len(data)+16*(nb+2)+64)
	g := C.int(0)
	if gz {
		g = 1
	}
	n := C.m4g_flate_encode_blocks(C.int(device), p, C.size_t(len(data)), bs, &ms[0], &off[0], g, C.uint32_t(mtime),
		(*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(len(out)), nil)
	if n < 0 {
		fail("m4g_flate_encode_blocks", C.int(n))
	}
	return out[:n]
}

// BrotliCompress is matchfinder.Writer{MatchFinder: q's parameters, Encoder: &brotli.Encoder{}, BlockSize: blockSize}
// followed by w.Write(data); w.Close() (the writer of brotli_test.go:626-649): match finding for all blocks in one
// launch, both passes of Encoder.Encode on the GPU (m4g_brotli_encode_blocks).
func (q *M4) BrotliCompress(data []byte, blockSize int) []byte {
	q.init()
	nb := (len(data) + blockSize - 1) / blockSize
	ms := make([]C.m4g_match, len(data)/q.minLength()+2*nb+2)
	off := make([]C.uint64_t, nb+1)
	var p *C.uint8_t
	if len(data) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&data[0]))
	}
	if n := C.m4g_find_matches_blocks(q.ctx, p, C.size_t(len(data)), C.size_t(blockSize), &ms[0], C.size_t(len(ms)), &off[0]); n < 0 {
		fail("m4g_find_matches_blocks", C.int(n))
	}
	out := make([]byte, len(data)+1024*(nb+2)+64)
	n := C.m4g_brotli_encode_blocks(C.int(q.Device), p, C.size_t(len(data)), C.size_t(blockSize), &ms[0], &off[0],
		(*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(len(out)))
	if n < 0 {
		fail("m4g_brotli_encode_blocks", C.int(n))
	}
	return out[:n]
}

// Writer is matchfinder.Writer (matchfinder/matchfinder.go:44-103) with the match finder and the encoder on the
// GPU (m4g_writer_*): an io.WriteCloser with Reset.
type Writer struct {
	Dest io.Writer
	w    *C.m4g_writer
	buf  []byte
	err  error
}

const (
	FormatBrotli = 0 // Encoder: &brotli.Encoder{}
	FormatFlate  = 1 // Encoder: flate.NewEncoder()
	FormatGZIP   = 2 // Encoder: flate.NewGZIPEncoder()
)

func newWriter(dst io.Writer, q M4, format, blockSize int, mtime uint32) *Writer {
	p := q.params()
	var h *C.m4g_writer
	if rc := C.m4g_writer_create(&p, C.int(q.Device), C.int(format), C.size_t(blockSize), C.uint32_t(mtime), &h); rc != 0 {
		fail("m4g_writer_create", rc)
	}
	w := &Writer{Dest: dst, w: h, buf: make([]byte, 1<<20)}
	runtime.SetFinalizer(w, func(w *Writer) { C.m4g_writer_destroy(w.w) })
	return w
}

// NewBrotliWriter is &matchfinder.Writer{Dest: dst, MatchFinder: &matchfinder.M4{q's fields}, Encoder: &brotli.Encoder{},
// BlockSize: blockSize} (brotli_test.go:626-649).
func NewBrotliWriter(dst io.Writer, q M4, blockSize int) *Writer { return newWriter(dst, q, FormatBrotli, blockSize, 0) }

// NewWriterV2 is brotli.NewWriterV2(dst, level) for levels 5-9 (writer.go:126-161): Bargain1 / Bargain2 / Bargain3 and
// brotli.Encoder on the GPU behind matchfinder.Writer's Write / Close / Reset (m4g_writer_create_v2).
func NewWriterV2(dst io.Writer, level int) *Writer {
	var h *C.m4g_writer
	if rc := C.m4g_writer_create_v2(C.int(level), 0, &h); rc != 0 {
		fail("m4g_writer_create_v2", rc)
	}
	w := &Writer{Dest: dst, w: h, buf: make([]byte, 1<<20)}
	runtime.SetFinalizer(w, func(w *Writer) { C.m4g_writer_destroy(w.w) })
	return w
}

func flateM4(level int) M4 {
	chain, ok := map[int]int{5: 8, 6: 16, 7: 32}[level] // flate/writer.go:33-46
	if !ok {
		panic("m4gpu: flate levels 5..7 use matchfinder.M4")
	}
	return M4{MaxDistance: 1 << 15, HashLen: 5, ChainLength: chain, DistanceBitCost: 66}
}

// NewFlateWriter is flate.NewWriter(dst, level) for level 5, 6 or 7 (flate/writer.go:14-16).
func NewFlateWriter(dst io.Writer, level int) *Writer {
	return newWriter(dst, flateM4(level), FormatFlate, 1<<16, 0)
}

// NewGZIPWriter is flate.NewGZIPWriter(dst, level) for level 5, 6 or 7; mtime replaces time.Now() (flate/gzip.go:46).
func NewGZIPWriter(dst io.Writer, level int, mtime uint32) *Writer {
	return newWriter(dst, flateM4(level), FormatGZIP, 1<<16, mtime)
}

// emit hands what the library produced to Dest; output that did not fit is kept inside and fetched again.
func (w *Writer) emit(call func(out *C.uint8_t, cap C.size_t, again bool) C.int64_t) {
	n := call((*C.uint8_t)(unsafe.Pointer(&w.buf[0])), C.size_t(len(w.buf)), false)
	if n < -C.M4G_STREAM_NEEDED_BIAS {
		w.buf = make([]byte, int(-n)-C.M4G_STREAM_NEEDED_BIAS+4096)
		n = call((*C.uint8_t)(unsafe.Pointer(&w.buf[0])), C.size_t(len(w.buf)), true)
	}
	if n < 0 {
		fail("m4g_writer", C.int(n))
	}
	if n > 0 && w.err == nil {
		_, w.err = w.Dest.Write(w.buf[:n])
	}
}

// Write implements io.Writer (matchfinder.go:59-79).
func (w *Writer) Write(p []byte) (int, error) {
	if w.err != nil {
		return 0, w.err
	}
	w.emit(func(out *C.uint8_t, cap C.size_t, again bool) C.int64_t {
		if again || len(p) == 0 {
			return C.m4g_writer_write(w.w, nil, 0, out, cap)
		}
		return C.m4g_writer_write(w.w, (*C.uint8_t)(unsafe.Pointer(&p[0])), C.size_t(len(p)), out, cap)
	})
	return len(p), w.err
}

// Close implements io.Closer (matchfinder.go:89-93).
func (w *Writer) Close() error {
	w.emit(func(out *C.uint8_t, cap C.size_t, again bool) C.int64_t { return C.m4g_writer_close(w.w, out, cap) })
	return w.err
}

// Reset is matchfinder.Writer.Reset (matchfinder.go:95-103).
func (w *Writer) Reset(newDest io.Writer) {
	if rc := C.m4g_writer_reset(w.w); rc != 0 {
		fail("m4g_writer_reset", rc)
	}
	w.err = nil
	w.Dest = newDest
}

// BargainBlocks returns the Match lists that the FindMatches calls of
// matchfinder.Writer{MatchFinder: &matchfinder.BargainN{MaxDistance: maxDistance, Skip: skip}, BlockSize: blockSize}
// produce for data (variant N = 1, 2, 3; bargain1.go:38-208, bargain2.go, bargain3.go; matchfinder.go:59-93):
// matches[off[k]:off[k+1]] is block k.  brotli.NewWriterV2 levels 5-9 use these finders with MaxDistance 1 << 20 and
// BlockSize 1 << 16 (writer.go:142-156); encode block k with Encoder.Encode(dst, block, matches[off[k]:off[k+1]], last).
func BargainBlocks(data []byte, variant, maxDistance int, skip bool, blockSize int) (matches []matchfinder.Match, off []uint64) {
	if blockSize == 0 {
		blockSize = 1 << 16
	}
	nb := (len(data) + blockSize - 1) / blockSize
	off = make([]uint64, nb+1)
	if len(data) == 0 {
		return nil, off
	}
	p := C.m4g_bargain_params{variant: C.int32_t(variant), max_distance: C.int32_t(maxDistance), block_size: C.int32_t(blockSize)}
	if skip {
		p.skip = 1
	}
	raw := make([]C.m4g_match, len(data)/3+2*nb+8)
	n := C.m4g_bargain_find_blocks(&p, 0, (*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data)),
		&raw[0], C.size_t(len(raw)), (*C.uint64_t)(unsafe.Pointer(&off[0])))
	if n < 0 {
		fail("m4g_bargain_find_blocks", C.int(n))
	}
	matches = make([]matchfinder.Match, n)
	for i := range matches {
		matches[i] = matchfinder.Match{Unmatched: int(raw[i].unmatched), Length: int(raw[i].length), Distance: int(raw[i].distance)}
	}
	return matches, off
}

// Bargain1, Bargain2 and Bargain3 have the exported fields of matchfinder.Bargain1 / Bargain2 / Bargain3
// (bargain1.go:15-31: MaxDistance, Skip) and implement matchfinder.MatchFinder on the GPU (m4g_bargain_ctx_*), so
//   &matchfinder.Writer{Dest: dst, MatchFinder: &m4gpu.Bargain1{MaxDistance: 1 << 20}, Encoder: &brotli.Encoder{}, BlockSize: 1 << 16}
// is brotli.NewWriterV2(dst, 5) with its match finder replaced and nothing else changed (writer.go:142-156).
// BlockSize must be the Writer's (0 = 1 << 16): the finder leaves the last bytes of every block out of its tables
// (bargain1.go:136-139) and the handle learns the block ends from it.  A block shorter than BlockSize ends the
// stream; Reset starts the next one.  One call is one wavefront's work: fast use is FindMatchesBlocks.
type Bargain1 struct {
	MaxDistance int
	Skip        bool
	BlockSize   int
	Device      int
	b           bargain
}
type Bargain2 struct {
	MaxDistance int
	Skip        bool
	BlockSize   int
	Device      int
	b           bargain
}
type Bargain3 struct {
	MaxDistance int
	Skip        bool
	BlockSize   int
	Device      int
	b           bargain
}

var (
	_ matchfinder.MatchFinder = (*Bargain1)(nil)
	_ matchfinder.MatchFinder = (*Bargain2)(nil)
	_ matchfinder.MatchFinder = (*Bargain3)(nil)
)

type bargain struct {
	ctx *C.m4g_bargain_ctx
	raw []C.m4g_match
}

func (b *bargain) find(variant, maxDistance int, skip bool, blockSize, device int, dst []matchfinder.Match, src []byte) []matchfinder.Match {
	if blockSize == 0 {
		blockSize = 1 << 16
	}
	if b.ctx == nil {
		p := C.m4g_bargain_params{variant: C.int32_t(variant), max_distance: C.int32_t(maxDistance), block_size: C.int32_t(blockSize)}
		if skip {
			p.skip = 1
		}
		if rc := C.m4g_bargain_ctx_create(&p, C.int(device), &b.ctx); rc != 0 {
			fail("m4g_bargain_ctx_create", rc)
		}
		runtime.SetFinalizer(b, func(b *bargain) { C.m4g_bargain_ctx_destroy(b.ctx) })
	}
	if len(src) == 0 {
		return dst
	}
	nb := (len(src) + blockSize - 1) / blockSize
	if need := len(src)/3 + 2*nb + 8; len(b.raw) < need {
		b.raw = make([]C.m4g_match, need)
	}
	n := C.m4g_bargain_ctx_find_blocks(b.ctx, (*C.uint8_t)(unsafe.Pointer(&src[0])), C.size_t(len(src)),
		&b.raw[0], C.size_t(len(b.raw)), nil)
	if n < 0 {
		fail("m4g_bargain_ctx_find_blocks", C.int(n))
	}
	for i := 0; i < int(n); i++ {
		m := b.raw[i]
		dst = append(dst, matchfinder.Match{Unmatched: int(m.unmatched), Length: int(m.length), Distance: int(m.distance)})
	}
	return dst
}

func (b *bargain) reset() {
	if b.ctx != nil {
		C.m4g_bargain_ctx_reset(b.ctx)
	}
}

// FindMatches implements matchfinder.MatchFinder (bargain1.go:38).
func (q *Bargain1) FindMatches(dst []matchfinder.Match, src []byte) []matchfinder.Match {
	return q.b.find(1, q.MaxDistance, q.Skip, q.BlockSize, q.Device, dst, src)
}
func (q *Bargain1) Reset() { q.b.reset() } // bargain1.go:33-36
func (q *Bargain2) FindMatches(dst []matchfinder.Match, src []byte) []matchfinder.Match {
	return q.b.find(2, q.MaxDistance, q.Skip, q.BlockSize, q.Device, dst, src)
}
func (q *Bargain2) Reset() { q.b.reset() }
func (q *Bargain3) FindMatches(dst []matchfinder.Match, src []byte) []matchfinder.Match {
	return q.b.find(3, q.MaxDistance, q.Skip, q.BlockSize, q.Device, dst, src)
}
func (q *Bargain3) Reset() { q.b.reset() }

// CompressV2 is w := brotli.NewWriterV2(dst, level); w.Write(data); w.Close() for levels 5-9 (writer.go:126-161)
// with match finder and encoder on the GPU; it returns the brotli stream.
func CompressV2(data []byte, level int) []byte {
	out := make([]byte, len(data)+(len(data)>>16+2)*1024+64)
	var src *C.uint8_t
	if len(data) > 0 {
		src = (*C.uint8_t)(unsafe.Pointer(&data[0]))
	}
	n := C.m4g_brotli_compress_v2(C.int(level), 0, src, C.size_t(len(data)), (*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(len(out)))
	if n < 0 {
		fail("m4g_brotli_compress_v2", C.int(n))
	}
	return out[:n]
}
