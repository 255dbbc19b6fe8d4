// m4g_matchfinder.hpp -- C++ mirror of the reference's plugin interface over
// the C-ABI of m4g.h (header only).  The reference's host language is Go and
// no Go toolchain exists in the build image, so the host side above the C-ABI
// is written in C++ with the reference's names and semantics; the Go shim a
// maintainer would add is in INTEGRATION.md.
//
//   matchfinder.Match        matchfinder/matchfinder.go:16-20
//   matchfinder.MatchFinder  matchfinder/matchfinder.go:23-30
//   matchfinder.Encoder      matchfinder/matchfinder.go:33-41
//   matchfinder.M4           matchfinder/m4.go:15-57 (fields, Reset, FindMatches)
//   matchfinder.AutoReset    matchfinder/textencoder.go:46-53
//   matchfinder.Writer       matchfinder/matchfinder.go:44-103
#pragma once
#include <cstdint>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "m4g.h"

namespace matchfinder {

struct Match {
  int64_t Unmatched = 0, Length = 0, Distance = 0;
};

struct MatchFinder {
  virtual ~MatchFinder() = default;
  // appends to dst and returns it
  virtual std::vector<Match> &FindMatches(std::vector<Match> &dst, const uint8_t *src, size_t n) = 0;
  virtual void Reset() = 0;
};

struct Encoder {
  virtual ~Encoder() = default;
  virtual std::vector<uint8_t> &Encode(std::vector<uint8_t> &dst, const uint8_t *src, size_t n,
                                       const std::vector<Match> &matches, bool lastBlock) = 0;
  virtual void Reset() = 0;
};

// matchfinder.M4 running on an MI355X.  Broken invariants panic in the
// reference (m0.go:50-52); here they throw.
class M4 : public MatchFinder {
public:
  int MaxDistance = 0, MinLength = 0, HashLen = 0, TableBits = 0, ChainLength = 0,
      DistanceBitCost = 0;
  int Device = 0;

  ~M4() override { m4g_destroy(ctx_); }

  std::vector<Match> &FindMatches(std::vector<Match> &dst, const uint8_t *src, size_t n) override {
    if (!ctx_) {
      m4g_params p{MaxDistance, MinLength, HashLen, TableBits, ChainLength, DistanceBitCost};
      check(m4g_create(&p, Device, &ctx_), "m4g_create");
    }
    const size_t cap = n / (size_t)(MinLength ? MinLength : 4) + 2;
    if (buf_.size() < cap) buf_.resize(cap);
    const uint32_t seed = dst.empty() ? 0u : (uint32_t)dst.back().Distance;
    const int64_t r = m4g_find_matches(ctx_, src, n, seed, buf_.data(), buf_.size());
    check(r, "m4g_find_matches");
    dst.reserve(dst.size() + (size_t)r);
    for (int64_t k = 0; k < r; k++)
      dst.push_back(Match{buf_[k].unmatched, buf_[k].length, buf_[k].distance});
    return dst;
  }
  void Reset() override {
    if (ctx_) check(m4g_reset(ctx_), "m4g_reset");
  }

  // Not in the reference: FindMatches(nil, block) for every consecutive block of
  // `blockSize` bytes of src in one GPU launch (m4g_find_matches_blocks).  The
  // Match list of block b is dst[blockOff[b] .. blockOff[b+1]) (appended to dst).
  std::vector<Match> &FindMatchesBlocks(std::vector<Match> &dst, const uint8_t *src, size_t n,
                                        size_t blockSize, std::vector<size_t> &blockOff) {
    if (!ctx_) {
      m4g_params p{MaxDistance, MinLength, HashLen, TableBits, ChainLength, DistanceBitCost};
      check(m4g_create(&p, Device, &ctx_), "m4g_create");
    }
    const size_t minLen = (size_t)(MinLength ? MinLength : 4);
    const size_t nblocks = blockSize ? (n + blockSize - 1) / blockSize : 0;
    size_t cap = 0;
    for (size_t b = 0; b < nblocks; b++) cap += std::min(blockSize, n - b * blockSize) / minLen + 2;
    if (buf_.size() < cap) buf_.resize(cap);
    std::vector<uint64_t> off(nblocks + 1, 0);
    const int64_t r = m4g_find_matches_blocks(ctx_, src, n, blockSize, buf_.data(), buf_.size(), off.data());
    check(r, "m4g_find_matches_blocks");
    const size_t base = dst.size();
    dst.reserve(base + (size_t)r);
    for (int64_t k = 0; k < r; k++)
      dst.push_back(Match{buf_[k].unmatched, buf_[k].length, buf_[k].distance});
    blockOff.assign(nblocks + 1, base);
    for (size_t b = 0; b <= nblocks; b++) blockOff[b] = base + (size_t)off[b];
    return dst;
  }

private:
  static void check(int64_t rc, const char *what) {
    if (rc < 0)
      throw std::runtime_error(std::string(what) + ": " + m4g_strerror((int)rc) + " (" +
                               m4g_last_error() + ")");
  }
  m4g_ctx *ctx_ = nullptr;
  std::vector<m4g_match> buf_;
};

// matchfinder.Bargain1 / Bargain2 / Bargain3 (bargain1.go:15-31, bargain2.go, bargain3.go) on an MI355X.
// The reference's FindMatches takes one block per call and its state is the hash table and history; one block
// is one wavefront's work, so the GPU form takes the whole stream: FindMatchesBlocks returns what the
// FindMatches calls of matchfinder.Writer{MatchFinder: &BargainN{MaxDistance, Skip}, BlockSize} produce for
// src (matchfinder.go:59-93), the Match list of block b in dst[blockOff[b] .. blockOff[b+1]).
//
// Bargain IS a MatchFinder (bargain1.go:33-38): FindMatches takes one block of the stream per call -- BlockSize
// bytes each, a shorter one only as the last before Reset, which is what matchfinder.Writer{BlockSize} hands over --
// and Reset starts a new stream (m4g_bargain_ctx_*: the handle keeps the stream's last MaxDistance bytes).
class Bargain : public MatchFinder {
public:
  int Variant = 1; // 1, 2, 3 = Bargain1, Bargain2, Bargain3
  int MaxDistance = 0;
  bool Skip = false;
  size_t BlockSize = (size_t)1 << 16; // of the Writer this finder is plugged into
  int Device = 0;

  ~Bargain() override { m4g_bargain_ctx_destroy(ctx_); }

  std::vector<Match> &FindMatches(std::vector<Match> &dst, const uint8_t *src, size_t n) override {
    std::vector<size_t> off;
    return FindMatchesStream(dst, src, n, off);
  }
  void Reset() override {
    if (ctx_) m4g_bargain_ctx_reset(ctx_);
  }
  // FindMatches for every consecutive block of BlockSize bytes of src, continuing the stream, in one launch
  std::vector<Match> &FindMatchesStream(std::vector<Match> &dst, const uint8_t *src, size_t n, std::vector<size_t> &blockOff) {
    if (!ctx_) {
      m4g_bargain_params p{Variant, MaxDistance, Skip ? 1 : 0, (int32_t)BlockSize};
      const int rc = m4g_bargain_ctx_create(&p, Device, &ctx_);
      if (rc) throw std::runtime_error(std::string("m4g_bargain_ctx_create: ") + m4g_strerror(rc) + " (" + m4g_last_error() + ")");
    }
    const size_t nblocks = (n + BlockSize - 1) / BlockSize;
    std::vector<m4g_match> buf(n / 3 + 2 * nblocks + 8);
    std::vector<uint64_t> off(nblocks + 1, 0);
    const int64_t r = m4g_bargain_ctx_find_blocks(ctx_, src, n, buf.data(), buf.size(), off.data());
    if (r < 0)
      throw std::runtime_error(std::string("m4g_bargain_ctx_find_blocks: ") + m4g_strerror((int)r) + " (" + m4g_last_error() + ")");
    const size_t base = dst.size();
    dst.reserve(base + (size_t)r);
    for (int64_t k = 0; k < r; k++) dst.push_back(Match{buf[k].unmatched, buf[k].length, buf[k].distance});
    blockOff.assign(nblocks + 1, base);
    for (size_t b = 0; b <= nblocks; b++) blockOff[b] = base + (size_t)off[b];
    return dst;
  }

  std::vector<Match> &FindMatchesBlocks(std::vector<Match> &dst, const uint8_t *src, size_t n, size_t blockSize,
                                        std::vector<size_t> &blockOff) {
    m4g_bargain_params p{Variant, MaxDistance, Skip ? 1 : 0, (int32_t)blockSize};
    const size_t bs = blockSize ? blockSize : (size_t)1 << 16;
    const size_t nblocks = (n + bs - 1) / bs;
    std::vector<m4g_match> buf(n / 3 + 2 * nblocks + 8);
    std::vector<uint64_t> off(nblocks + 1, 0);
    const int64_t r = m4g_bargain_find_blocks(&p, Device, src, n, buf.data(), buf.size(), off.data());
    if (r < 0)
      throw std::runtime_error(std::string("m4g_bargain_find_blocks: ") + m4g_strerror((int)r) + " (" +
                               m4g_last_error() + ")");
    const size_t base = dst.size();
    dst.reserve(base + (size_t)r);
    for (int64_t k = 0; k < r; k++) dst.push_back(Match{buf[k].unmatched, buf[k].length, buf[k].distance});
    blockOff.assign(nblocks + 1, base);
    for (size_t b = 0; b <= nblocks; b++) blockOff[b] = base + (size_t)off[b];
    return dst;
  }

private:
  m4g_bargain_ctx *ctx_ = nullptr;
};

struct AutoReset : MatchFinder {
  MatchFinder *Inner = nullptr;
  std::vector<Match> &FindMatches(std::vector<Match> &dst, const uint8_t *src, size_t n) override {
    Inner->Reset();
    return Inner->FindMatches(dst, src, n);
  }
  void Reset() override { Inner->Reset(); }
};

// Dest is any callable taking (const uint8_t*, size_t).
template <class Dest> class Writer {
public:
  Dest *dest = nullptr;
  MatchFinder *matchFinder = nullptr;
  Encoder *encoder = nullptr;
  size_t BlockSize = 0;

  size_t Write(const uint8_t *p, size_t n) {
    if (BlockSize == 0) {
      writeBlock(p, n, false);
      return n;
    }
    inBuf_.insert(inBuf_.end(), p, p + n);
    size_t pos = 0;
    for (; pos + BlockSize <= inBuf_.size(); pos += BlockSize)
      writeBlock(inBuf_.data() + pos, BlockSize, false);
    if (pos) inBuf_.erase(inBuf_.begin(), inBuf_.begin() + (ptrdiff_t)pos);
    return n;
  }
  void Close() {
    writeBlock(inBuf_.data(), inBuf_.size(), true);
    inBuf_.clear();
  }
  void Reset(Dest *newDest) {
    matchFinder->Reset();
    encoder->Reset();
    inBuf_.clear();
    dest = newDest;
  }

private:
  void writeBlock(const uint8_t *p, size_t n, bool last) {
    outBuf_.clear();
    matches_.clear();
    matchFinder->FindMatches(matches_, p, n);
    encoder->Encode(outBuf_, p, n, matches_, last);
    (*dest)(outBuf_.data(), outBuf_.size());
  }
  std::vector<uint8_t> inBuf_, outBuf_;
  std::vector<Match> matches_;
};

// The same Writer with BOTH plug-ins on the GPU (m4g_writer_*): MatchFinder = M4{fields}, Encoder = brotli.Encoder
// (M4G_FORMAT_BROTLI), flate.NewEncoder() (M4G_FORMAT_FLATE) or flate.NewGZIPEncoder() (M4G_FORMAT_GZIP).  Dest
// receives the bytes the reference's Writer passes to Dest.Write after the same calls.
template <class Dest> class GPUWriter {
public:
  Dest *dest = nullptr;

  GPUWriter(Dest *d, const m4g_params &fields, int format, size_t blockSize, uint32_t gzipMtime = 0, int device = 0)
      : dest(d), buf_(1 << 20) {
    const int rc = m4g_writer_create(&fields, device, format, blockSize, gzipMtime, &w_);
    if (rc != M4G_OK)
      throw std::runtime_error(std::string("m4g_writer_create: ") + m4g_strerror(rc) + " (" + m4g_last_error() + ")");
  }
  // brotli.NewWriterV2(dst, level) for levels 5-9 (writer.go:126-161): Bargain1 / 2 / 3 + brotli.Encoder
  struct V2 {
    int level;
  };
  GPUWriter(Dest *d, V2 v2, int device = 0) : dest(d), buf_(1 << 20) {
    const int rc = m4g_writer_create_v2(v2.level, device, &w_);
    if (rc != M4G_OK)
      throw std::runtime_error(std::string("m4g_writer_create_v2: ") + m4g_strerror(rc) + " (" + m4g_last_error() + ")");
  }
  ~GPUWriter() { m4g_writer_destroy(w_); }
  GPUWriter(const GPUWriter &) = delete;
  GPUWriter &operator=(const GPUWriter &) = delete;

  // flate.NewWriter(dst, level) / flate.NewGZIPWriter(dst, level), level 5..7 (flate/writer.go:33-66)
  static m4g_params flateFields(int level) {
    m4g_params p{};
    p.max_distance = 1 << 15;
    p.hash_len = 5;
    p.chain_length = level == 5 ? 8 : level == 6 ? 16 : 32;
    p.distance_bit_cost = 66;
    return p;
  }

  size_t Write(const uint8_t *p, size_t n) {
    emit(m4g_writer_write(w_, p, n, buf_.data(), buf_.size()), true);
    return n;
  }
  void Close() { emit(m4g_writer_close(w_, buf_.data(), buf_.size()), false); }
  void Reset(Dest *newDest) {
    const int rc = m4g_writer_reset(w_);
    if (rc != M4G_OK) throw std::runtime_error(std::string("m4g_writer_reset: ") + m4g_strerror(rc));
    dest = newDest;
  }

private:
  void emit(int64_t r, bool writing) {
    if (r < -(int64_t)M4G_STREAM_NEEDED_BIAS) { // kept inside: fetch it with a buffer that is large enough
      buf_.resize((size_t)(-r - M4G_STREAM_NEEDED_BIAS) + 4096);
      r = writing ? m4g_writer_write(w_, nullptr, 0, buf_.data(), buf_.size()) : m4g_writer_close(w_, buf_.data(), buf_.size());
    }
    if (r < 0)
      throw std::runtime_error(std::string("m4g_writer: ") + m4g_strerror((int)r) + " (" + m4g_last_error() + ")");
    if (r > 0) (*dest)(buf_.data(), (size_t)r);
  }
  m4g_writer *w_ = nullptr;
  std::vector<uint8_t> buf_;
};

} // namespace matchfinder
