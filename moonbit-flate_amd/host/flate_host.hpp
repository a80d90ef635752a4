// flate_host.hpp -- header-only C++ mirror of the reference's host API over the C ABI
// (include/flate_hip.h).  Names, argument meaning and error behaviour follow the reference:
//   Writer::new / write / close          writer.mbt:10,45,53
//   Compressor sticky errors             deflate.mbt:74,154-183,280-294
//   writer_closed_error                  deflate.mbt:154
//   &Reader::new / read / close, ioeof    inflate.mbt:305,382,410,19
//   corrupt_input_error                   inflate.mbt:38
// The reference is single-stream and synchronous: Compressor::write compresses every full
// 65535-byte window as soon as it is there and the sink sees the bytes (deflate.mbt:280-294,
// huffman-bit-writer.mbt:193-196).  Writer does the same through flate_hip_stream_write: write()
// calls are staged (the output is a function of the concatenated bytes only, deflate.mbt:222-229),
// and whenever `emit_windows` full windows are pending they are compressed and their finished bytes
// go to the sink -- memory stays bounded and output leaves before close().  BatchWriter (below) closes
// many Writers with one kernel pipeline -- the intended way to use the engine -- and a Writer built
// with chunk_bytes > 0 cuts ONE large stream into independent chunks that the GPU compresses in
// parallel and splices into one legal DEFLATE stream (different bytes than a single
// Writer would produce, same inflated result).
#pragma once

#include <algorithm>
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "flate_hip.h"

namespace flate_host {

struct IOError {  // @io.IOError
  std::string msg;
  bool operator==(const IOError &o) const { return msg == o.msg; }
};
using Err = std::optional<IOError>;

inline const IOError &writer_closed_error() {  // deflate.mbt:154
  static const IOError e{"writer closed"};
  return e;
}

// &@io.Writer
struct ByteSink {
  virtual ~ByteSink() = default;
  virtual std::pair<int, Err> write(const uint8_t *p, size_t n) = 0;
};

// @io.Buffer used as in-memory sink
struct Buffer : ByteSink {
  std::vector<uint8_t> bytes;
  std::pair<int, Err> write(const uint8_t *p, size_t n) override {
    bytes.insert(bytes.end(), p, p + n);
    return {(int)n, std::nullopt};
  }
};

// One GPU context (flate_hip_ctx).  No GPU => construction reports the error; there is no CPU path.
class Engine {
 public:
  explicit Engine(int device = 0) { rc_ = flate_hip_init(device, &ctx_); }
  ~Engine() { flate_hip_destroy(ctx_); }
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;
  bool ok() const { return rc_ == 0; }
  int status() const { return rc_; }
  flate_hip_ctx *ctx() const { return ctx_; }

 private:
  flate_hip_ctx *ctx_ = nullptr;
  int rc_ = 0;
};

inline Err make_error(const Engine &e, int rc) {
  std::string m = std::string("flate_hip: ") + flate_hip_strerror(rc);
  const char *h = e.ctx() ? flate_hip_last_hip_error(e.ctx()) : "";
  if (h && *h) m += std::string(" [") + h + "]";
  return IOError{m};
}

// Compress `streams` (each with fresh-Writer semantics) in one batch; out[i] = stream i's bytes.
inline Err compress_batch(Engine &e, const std::vector<std::vector<uint8_t>> &streams,
                          std::vector<std::vector<uint8_t>> &out, uint32_t flags = 0) {
  if (!e.ok()) return make_error(e, e.status());
  const uint32_t n = (uint32_t)streams.size();
  std::vector<uint64_t> in_off(n + 1, 0), out_off(n + 1, 0);
  uint64_t cap = 16;
  for (uint32_t i = 0; i < n; ++i) {
    in_off[i + 1] = in_off[i] + streams[i].size();
    cap += flate_hip_deflate_bound(streams[i].size());
  }
  std::vector<uint8_t> in(in_off[n] + 1), buf(cap);
  for (uint32_t i = 0; i < n; ++i)
    std::copy(streams[i].begin(), streams[i].end(), in.begin() + in_off[i]);
  const int rc = flate_hip_deflate_fast_batch(e.ctx(), in.data(), in_off.data(), n, buf.data(), cap,
                                              out_off.data(), flags);
  if (rc != 0) return make_error(e, rc);
  out.resize(n);
  for (uint32_t i = 0; i < n; ++i) out[i].assign(buf.begin() + out_off[i], buf.begin() + out_off[i + 1]);
  return std::nullopt;
}

// The container formats around a raw stream (the reference has neither): zlib (RFC 1950: CMF, FLG,
// data, Adler-32 big endian) and gzip (RFC 1952: ten header bytes, data, CRC-32 and length little endian).
// The checksums come from the GPU (flate_hip_checksum_batch), the framing is these few bytes.
enum class Wrap { Raw, Zlib, Gzip };

inline Err checksum_batch(Engine &e, const std::vector<std::vector<uint8_t>> &streams, uint32_t kind,
                          std::vector<uint32_t> &sums) {
  if (!e.ok()) return make_error(e, e.status());
  const uint32_t n = (uint32_t)streams.size();
  std::vector<uint64_t> in_off(n + 1, 0);
  for (uint32_t i = 0; i < n; ++i) in_off[i + 1] = in_off[i] + streams[i].size();
  std::vector<uint8_t> in(in_off[n] + 1);
  for (uint32_t i = 0; i < n; ++i) std::copy(streams[i].begin(), streams[i].end(), in.begin() + in_off[i]);
  sums.assign(n, 0);
  const int rc = flate_hip_checksum_batch(e.ctx(), in.data(), in_off.data(), n, kind, sums.data(), 0);
  if (rc != 0) return make_error(e, rc);
  return std::nullopt;
}

// compress_batch with every stream inside its container
inline Err compress_batch(Engine &e, const std::vector<std::vector<uint8_t>> &streams,
                          std::vector<std::vector<uint8_t>> &out, Wrap wrap, uint32_t flags = 0) {
  std::vector<std::vector<uint8_t>> raw;
  if (Err er = compress_batch(e, streams, raw, flags)) return er;
  if (wrap == Wrap::Raw) {
    out = std::move(raw);
    return std::nullopt;
  }
  std::vector<uint32_t> sums;
  if (Err er = checksum_batch(e, streams, wrap == Wrap::Zlib ? FLATE_HIP_CHECKSUM_ADLER32 : FLATE_HIP_CHECKSUM_CRC32, sums))
    return er;
  static const uint8_t zhead[2] = {0x78, 0x01};  // CM = 8, 32 KiB window; FLEVEL = 0 (fastest), FCHECK
  static const uint8_t ghead[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4 /* XFL: fastest */, 255 /* OS: unknown */};
  out.resize(raw.size());
  for (size_t i = 0; i < raw.size(); ++i) {
    std::vector<uint8_t> &m = out[i];
    m.clear();
    if (wrap == Wrap::Zlib) {
      m.insert(m.end(), zhead, zhead + 2);
      m.insert(m.end(), raw[i].begin(), raw[i].end());
      for (int k = 3; k >= 0; --k) m.push_back((uint8_t)(sums[i] >> (8 * k)));
    } else {
      m.insert(m.end(), ghead, ghead + 10);
      m.insert(m.end(), raw[i].begin(), raw[i].end());
      for (int k = 0; k < 4; ++k) m.push_back((uint8_t)(sums[i] >> (8 * k)));
      for (int k = 0; k < 4; ++k) m.push_back((uint8_t)((uint32_t)streams[i].size() >> (8 * k)));
    }
  }
  return std::nullopt;
}

inline Err compress_spliced(Engine &e, const std::vector<std::vector<uint8_t>> &streams,
                            std::vector<uint8_t> &out, std::vector<uint64_t> *bit_off, uint32_t flags);

// Writer (writer.mbt:2-15): write() appends to the stream, close() emits the DEFLATE bytes to the
// sink given at construction.  Sticky error rules of Compressor (deflate.mbt:157-183,280-294).
// chunk_bytes == 0 (default): the bytes a reference Writer produces, bit for bit.
// chunk_bytes  > 0 (opt-in): the stream is cut into independent chunks of that many
// bytes, each compressed as a fresh Writer would (no matches across chunk borders), all chunks in
// one GPU batch, and emitted as ONE spliced DEFLATE stream.  Any inflater reads it back to the
// same bytes, but it is NOT the byte sequence of a single Writer -- a one-wavefront stream gets no
// parallelism on the GPU, this mode does.
class Writer {
 public:
  static constexpr size_t kWindow = 65535;  // max_store_block_size, deflate-fast.mbt:46
  // emit_windows: full windows staged before they are compressed and handed to the sink (16 = 1 MiB)
  Writer(ByteSink &w, Engine &e, uint32_t flags = 0, size_t chunk_bytes = 0, size_t emit_windows = 16)
      : w_(w), e_(e), flags_(flags), chunk_(chunk_bytes), emit_(emit_windows ? emit_windows : 1) {}
  // Writer::new_dict(w, dict) (writer.mbt:25-31) AS THE REFERENCE BEHAVES: fill_window
  // (deflate.mbt:108-151) copies the last window_size = 32768 bytes of the dictionary into the input
  // window and sets window_end, i.e. they are unprocessed DATA -- the stream is the one Writer::new
  // produces when those bytes are written first (the reference's own test asserts exactly that
  // equality, deflate_test.mbt:12-35).  Go ignores the dictionary at this level.
  static std::unique_ptr<Writer> new_dict(ByteSink &w, Engine &e, const uint8_t *dict, size_t n, uint32_t flags = 0) {
    std::unique_ptr<Writer> wr(new Writer(w, e, flags));
    constexpr size_t kDictWindow = 32768;  // window_size, deflate.mbt:12
    if (n > kDictWindow) {
      dict += n - kDictWindow;
      n = kDictWindow;
    }
    wr->pending_.assign(dict, dict + n);
    return wr;
  }
  ~Writer() { flate_hip_stream_free(st_); }
  Writer(const Writer &) = delete;
  Writer &operator=(const Writer &) = delete;

  std::pair<int, Err> write(const uint8_t *p, size_t n) {  // deflate.mbt:280-294
    if (err_) return {0, err_};
    pending_.insert(pending_.end(), p, p + n);
    if (chunk_ == 0 && !batch_member_ && pending_.size() >= emit_ * kWindow) {
      const size_t k = pending_.size() / kWindow * kWindow;  // every full window that is there
      if (Err er = feed(k, false)) {
        err_ = er;
        return {0, err_};
      }
    }
    return {(int)n, std::nullopt};
  }
  std::pair<int, Err> write(const std::vector<uint8_t> &b) { return write(b.data(), b.size()); }

  Err close() {  // deflate.mbt:157-183
    if (err_ && *err_ == writer_closed_error()) return std::nullopt;
    if (err_) return err_;
    Err er;
    if (chunk_ > 0 && pending_.size() > chunk_) {
      std::vector<std::vector<uint8_t>> parts;
      std::vector<uint8_t> one;
      for (size_t o = 0; o < pending_.size(); o += chunk_)
        parts.emplace_back(pending_.begin() + o, pending_.begin() + std::min(pending_.size(), o + chunk_));
      er = compress_spliced(e_, parts, one, nullptr, flags_);
      if (!er) {
        auto r = w_.write(one.data(), one.size());
        er = r.second;
      }
    } else {
      er = feed(pending_.size(), true);  // the rest (any length) + Writer::close's block
    }
    if (er) {
      err_ = er;
      return err_;
    }
    err_ = writer_closed_error();
    return std::nullopt;
  }
  // compressed bytes handed to the sink so far (before close(): the pieces already emitted)
  uint64_t emitted() const { return emitted_; }

 private:
  // compress the first n staged bytes as the stream's next piece and forward what is finished
  Err feed(size_t n, bool final) {
    if (!e_.ok()) return make_error(e_, e_.status());
    if (!st_) {
      const int rc = flate_hip_stream_open(e_.ctx(), flags_, &st_);
      if (rc != 0) return make_error(e_, rc);
    }
    std::vector<uint8_t> out(flate_hip_stream_bound(n));
    uint64_t len = 0;
    const int rc = flate_hip_stream_write(st_, pending_.data(), n, final ? 1 : 0, out.data(), out.size(), &len);
    if (rc != 0) return make_error(e_, rc);
    pending_.erase(pending_.begin(), pending_.begin() + n);
    emitted_ += len;
    auto r = w_.write(out.data(), len);
    return r.second;
  }
  ByteSink &w_;
  Engine &e_;
  uint32_t flags_;
  size_t chunk_, emit_;
  bool batch_member_ = false;  // closed by a BatchWriter: staged until close_all
  flate_hip_stream *st_ = nullptr;
  uint64_t emitted_ = 0;
  std::vector<uint8_t> pending_;
  Err err_;
  friend class BatchWriter;
};

// Many Writers closed by ONE kernel pipeline: every stream still gets the bytes of its own
// Writer::new / write / close (writer.mbt:10,45,53), but the GPU sees them as one batch.
//   BatchWriter bw(engine);  Writer &a = bw.add(sink_a);  Writer &b = bw.add(sink_b);
//   a.write(...); b.write(...);  bw.close_all();
class BatchWriter {
 public:
  explicit BatchWriter(Engine &e, uint32_t flags = 0) : e_(e), flags_(flags) {}
  Writer &add(ByteSink &sink) {
    ws_.emplace_back(new Writer(sink, e_, flags_));
    ws_.back()->batch_member_ = true;
    return *ws_.back();
  }
  size_t size() const { return ws_.size(); }
  // close() of every Writer that is still open; returns the first error (each Writer keeps its
  // own sticky state exactly as a lone close() would leave it)
  Err close_all() {
    std::vector<std::vector<uint8_t>> in, out;
    std::vector<Writer *> open;
    for (auto &w : ws_)
      if (!w->err_) {
        open.push_back(w.get());
        in.push_back(std::move(w->pending_));
      }
    if (open.empty()) return std::nullopt;
    Err er = compress_batch(e_, in, out, flags_);
    Err first;
    for (size_t i = 0; i < open.size(); ++i) {
      Writer *w = open[i];
      if (er) {
        w->err_ = er;
      } else {
        auto r = w->w_.write(out[i].data(), out[i].size());
        w->err_ = r.second ? r.second : Err(writer_closed_error());
      }
      if (!first && w->err_ && !(*w->err_ == writer_closed_error())) first = w->err_;
    }
    return first;
  }

 private:
  Engine &e_;
  uint32_t flags_;
  std::vector<std::unique_ptr<Writer>> ws_;
};

// ---- spliced form (no reference counterpart): one DEFLATE stream for the batch ----
inline Err compress_spliced(Engine &e, const std::vector<std::vector<uint8_t>> &streams,
                            std::vector<uint8_t> &out, std::vector<uint64_t> *bit_off = nullptr,
                            uint32_t flags = 0);
inline Err compress_spliced(Engine &e, const std::vector<std::vector<uint8_t>> &streams,
                            std::vector<uint8_t> &out, std::vector<uint64_t> *bit_off, uint32_t flags) {
  if (!e.ok()) return make_error(e, e.status());
  const uint32_t n = (uint32_t)streams.size();
  std::vector<uint64_t> in_off(n + 1, 0), bo(n + 1, 0);
  uint64_t cap = 16;
  for (uint32_t i = 0; i < n; ++i) {
    in_off[i + 1] = in_off[i] + streams[i].size();
    cap += flate_hip_deflate_bound(streams[i].size());
  }
  std::vector<uint8_t> in(in_off[n] + 1), buf(cap);
  for (uint32_t i = 0; i < n; ++i)
    std::copy(streams[i].begin(), streams[i].end(), in.begin() + in_off[i]);
  uint64_t len = 0;
  const int rc = flate_hip_deflate_fast_spliced(e.ctx(), in.data(), in_off.data(), n, buf.data(), cap, &len,
                                                bo.data(), flags);
  if (rc != 0) return make_error(e, rc);
  out.assign(buf.begin(), buf.begin() + len);
  if (bit_off) *bit_off = bo;
  return std::nullopt;
}

// ---- decode side ---------------------------------------------------------------------------
// pub let ioeof (inflate.mbt:19) and the errors of the Decompressor (inflate.mbt:38,780-785)
inline const IOError &ioeof() {
  static const IOError e{"EOF"};
  return e;
}
inline IOError corrupt_input_error(long long off) {  // inflate.mbt:38-40
  return IOError{"flate: corrupt input before offset " + std::to_string(off)};
}
inline const IOError &err_unexpected_eof() {  // @io.err_unexpected_eof via no_eof, inflate.mbt:780-785
  static const IOError e{"unexpected EOF"};
  return e;
}

struct Inflated {
  std::vector<uint8_t> bytes;  // what was decoded (also in front of an error, as read() flushes it)
  Err err;                     // nullopt, corrupt_input_error(offset) or err_unexpected_eof
  int status = 0;              // the FLATE_HIP_E_* code behind err (0 = none)
};

// The size every stream inflates to, without storing anything (FLATE_HIP_SIZE_ONLY): what a Reader
// of a stream of unknown size runs first.  sizes[i] counts the bytes in front of an error, too.
inline Err inflate_sizes(Engine &e, const std::vector<std::vector<uint8_t>> &streams, std::vector<uint64_t> &sizes) {
  if (!e.ok()) return make_error(e, e.status());
  const uint32_t n = (uint32_t)streams.size();
  std::vector<uint64_t> in_off(n + 1, 0);
  std::vector<int32_t> status(n + 1, 0);
  std::vector<int64_t> err_off(n + 1, -1);
  for (uint32_t i = 0; i < n; ++i) in_off[i + 1] = in_off[i] + streams[i].size();
  std::vector<uint8_t> in(in_off[n] + 8);
  for (uint32_t i = 0; i < n; ++i) std::copy(streams[i].begin(), streams[i].end(), in.begin() + in_off[i]);
  sizes.assign(n + 1, 0);
  const int rc = flate_hip_inflate_batch(e.ctx(), in.data(), in_off.data(), n, nullptr, nullptr, sizes.data(),
                                         status.data(), err_off.data(), FLATE_HIP_SIZE_ONLY);
  sizes.resize(n);
  if (rc != 0 && rc != FLATE_HIP_E_CORRUPT && rc != FLATE_HIP_E_UNEXPECTED_EOF) return make_error(e, rc);
  return std::nullopt;
}

// Decode independent streams in one batch.  sizes[i] = capacity for stream i's output.
inline Err decompress_batch(Engine &e, const std::vector<std::vector<uint8_t>> &streams,
                            const std::vector<uint64_t> &sizes, std::vector<Inflated> &out) {
  if (!e.ok()) return make_error(e, e.status());
  const uint32_t n = (uint32_t)streams.size();
  std::vector<uint64_t> in_off(n + 1, 0), out_off(n + 1, 0), out_len(n + 1, 0);
  std::vector<int32_t> status(n + 1, 0);
  std::vector<int64_t> err_off(n + 1, -1);
  for (uint32_t i = 0; i < n; ++i) {
    in_off[i + 1] = in_off[i] + streams[i].size();
    out_off[i + 1] = out_off[i] + sizes[i];
  }
  std::vector<uint8_t> in(in_off[n] + 8), buf(out_off[n] + 8);
  for (uint32_t i = 0; i < n; ++i)
    std::copy(streams[i].begin(), streams[i].end(), in.begin() + in_off[i]);
  const int rc = flate_hip_inflate_batch(e.ctx(), in.data(), in_off.data(), n, buf.data(), out_off.data(),
                                         out_len.data(), status.data(), err_off.data(), 0);
  if (rc != 0 && rc != FLATE_HIP_E_CORRUPT && rc != FLATE_HIP_E_UNEXPECTED_EOF && rc != FLATE_HIP_E_OUT_TOO_SMALL)
    return make_error(e, rc);
  out.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    out[i].bytes.assign(buf.begin() + out_off[i], buf.begin() + out_off[i] + out_len[i]);
    out[i].err = std::nullopt;
    out[i].status = status[i];
    if (status[i] == FLATE_HIP_E_CORRUPT) out[i].err = corrupt_input_error(err_off[i]);
    else if (status[i] == FLATE_HIP_E_UNEXPECTED_EOF) out[i].err = err_unexpected_eof();
    else if (status[i] != 0) out[i].err = make_error(e, status[i]);
  }
  return std::nullopt;
}

// header / trailer bytes of one zlib or gzip member, {-1, 0} if its header is not one (RFC 1950 2.2: CM = 8,
// window <= 32 KiB, FCHECK, no preset dictionary; RFC 1952 2.3: magic, CM = 8, reserved flag bits zero, the
// optional fields skipped)
inline std::pair<int, int> container_header(const std::vector<uint8_t> &m, Wrap wrap) {
  if (wrap == Wrap::Zlib) {
    if (m.size() < 2 || (m[0] & 15) != 8 || (m[0] >> 4) > 7 || ((m[0] << 8) | m[1]) % 31 || (m[1] & 0x20)) return {-1, 0};
    return {2, 4};
  }
  if (m.size() < 10 || m[0] != 0x1f || m[1] != 0x8b || m[2] != 8 || (m[3] & 0xe0)) return {-1, 0};
  const uint8_t flg = m[3];
  size_t p = 10;
  if (flg & 4) {  // FEXTRA
    if (m.size() < p + 2) return {-1, 0};
    p += 2 + (size_t)(m[p] | (m[p + 1] << 8));
  }
  for (int bit : {8, 16})  // FNAME, FCOMMENT: zero-terminated
    if (flg & bit) {
      while (p < m.size() && m[p]) ++p;
      if (p >= m.size()) return {-1, 0};
      ++p;
    }
  if (flg & 2) p += 2;  // FHCRC
  return p <= m.size() ? std::pair<int, int>{(int)p, 8} : std::pair<int, int>{-1, 0};
}

// decompress_batch for zlib / gzip members: the raw streams decoded on the GPU, the trailers checked against the
// checksums of what came out (flate_hip_checksum_batch); a bad header, checksum or (gzip) length is
// corrupt_input_error at the member's end.  sizes[i] = capacity for member i's output (gzip: 0 = its ISIZE).
inline Err decompress_batch(Engine &e, const std::vector<std::vector<uint8_t>> &members, std::vector<uint64_t> sizes,
                            std::vector<Inflated> &out, Wrap wrap) {
  if (wrap == Wrap::Raw) return decompress_batch(e, members, sizes, out);
  const size_t n = members.size();
  std::vector<std::vector<uint8_t>> raw(n);
  std::vector<uint32_t> want(n, 0), isize(n, 0);
  std::vector<bool> bad(n, false);
  sizes.resize(n, 0);
  for (size_t i = 0; i < n; ++i) {
    const auto ht = container_header(members[i], wrap);
    if (ht.first < 0 || members[i].size() < (size_t)(ht.first + ht.second)) {
      bad[i] = true;
      continue;
    }
    raw[i].assign(members[i].begin() + ht.first, members[i].end() - ht.second);
    const uint8_t *t = members[i].data() + members[i].size() - ht.second;
    if (wrap == Wrap::Zlib) {
      want[i] = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    } else {
      want[i] = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
      isize[i] = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
      if (sizes[i] == 0) sizes[i] = isize[i];
    }
  }
  if (Err er = decompress_batch(e, raw, sizes, out)) return er;
  std::vector<std::vector<uint8_t>> plain(n);
  for (size_t i = 0; i < n; ++i) plain[i] = out[i].bytes;
  std::vector<uint32_t> sums;
  if (Err er = checksum_batch(e, plain, wrap == Wrap::Zlib ? FLATE_HIP_CHECKSUM_ADLER32 : FLATE_HIP_CHECKSUM_CRC32, sums)) return er;
  for (size_t i = 0; i < n; ++i) {
    const bool mismatch = !bad[i] && out[i].status == 0 &&
                          (sums[i] != want[i] || (wrap == Wrap::Gzip && (uint32_t)out[i].bytes.size() != isize[i]));
    if (bad[i] || mismatch) {
      out[i].status = FLATE_HIP_E_CORRUPT;
      out[i].err = corrupt_input_error(bad[i] ? 0 : (int64_t)members[i].size());
    }
  }
  return std::nullopt;
}

// &Reader (inflate.mbt:227-232): where a Decompressor pulls its input from
struct ByteSource {
  virtual ~ByteSource() = default;
  virtual std::pair<int, Err> read(uint8_t *p, size_t n) = 0;  // (0, ioeof) at the end
};
struct BytesReader : ByteSource {  // @io.Buffer used as source
  std::vector<uint8_t> bytes;
  size_t pos = 0;
  explicit BytesReader(std::vector<uint8_t> b) : bytes(std::move(b)) {}
  std::pair<int, Err> read(uint8_t *p, size_t n) override {
    if (pos == bytes.size()) return {0, ioeof()};
    const size_t k = std::min(n, bytes.size() - pos);
    std::copy(bytes.begin() + pos, bytes.begin() + pos + k, p);
    pos += k;
    return {(int)k, std::nullopt};
  }
};

// &Reader::new(r) -> Decompressor (inflate.mbt:305) with read (:382-405) and close (:410-415).
// As the reference's, this Reader holds a PIECE of the compressed stream and a piece of the output,
// never the whole of either (flate_hip_inflate_stream_*: the decoder's state -- window, tables of the
// block in progress, bit position, a copy that did not fit -- rests on the device between calls): a
// 10 GB stream needs the two pieces, not 10 GB.  The bytes and errors come out as the reference hands
// them out -- data first, the error (ioeof at a clean end) together with the last bytes, nothing but
// the error afterwards.  (size_hint is accepted for source compatibility and ignored: nothing has to
// be sized any more.)
class Reader {
 public:
  Reader(ByteSource &r, Engine &e, uint64_t size_hint = 0, size_t in_piece = 1u << 20, size_t out_piece = 1u << 20)
      : r_(&r), e_(e), in_piece_(in_piece < 4096 ? 4096 : in_piece), out_piece_(out_piece < 1 ? 1 : out_piece) {
    (void)size_hint;
  }
  // &Reader::new_dict(r, dict) (inflate.mbt:315-317): the stream decodes as if its output started with
  // `dict`, which has already been read (its last 32768 bytes are the history, dict-decoder.mbt:40-60)
  static std::unique_ptr<Reader> new_dict(ByteSource &r, Engine &e, std::vector<uint8_t> dict, size_t in_piece = 1u << 20,
                                          size_t out_piece = 1u << 20) {
    std::unique_ptr<Reader> rd(new Reader(r, e, 0, in_piece, out_piece));
    rd->dict_ = std::move(dict);
    return rd;
  }
  ~Reader() { flate_hip_inflate_stream_free(st_); }
  Reader(const Reader &) = delete;
  Reader &operator=(const Reader &) = delete;

  // Decompressor::reset(r, dict) (inflate.mbt:862-884): forget everything and decode the stream `r`
  // delivers next, with `dict` (may be empty) as its preset dictionary.
  void reset(ByteSource &r, std::vector<uint8_t> dict = {}) {
    r_ = &r;
    dict_ = std::move(dict);
    fresh_ = true;  // (the handle, if any, is reset in front of the next decode)
    in_.clear();
    src_end_ = false;
    data_.clear();
    pos_ = 0;
    err_ = std::nullopt;
  }

  // Decompressor::make_reader (inflate.mbt:857-860): the stream goes on from another source; nothing
  // else changes (bytes already pulled from the old source are used first).
  void make_reader(ByteSource &r) {
    r_ = &r;
    src_end_ = false;
  }
  // The reference pulls its source a byte at a time and stops at the end of the final block; this
  // Reader pulls a piece ahead.  What it pulled and did not use -- the bytes BEHIND the stream when the
  // source holds more than one payload -- is here (valid once read() has returned the stream's end).
  const std::vector<uint8_t> &unread() const { return in_; }

  std::pair<int, Err> read(uint8_t *p, size_t n) {
    for (;;) {
      if (pos_ < data_.size()) {  // :384-396
        const size_t k = std::min(n, data_.size() - pos_);
        std::copy(data_.begin() + pos_, data_.begin() + pos_ + k, p);
        pos_ += k;
        if (pos_ == data_.size()) return {(int)k, err_};
        return {(int)k, std::nullopt};
      }
      if (err_) return {0, err_};  // :397-400
      step();
    }
  }
  Err close() {  // :410-415
    if (err_ && *err_ == ioeof()) return std::nullopt;
    return err_;
  }
  // what this Reader holds at most: the two pieces (+ the decoder's 40 KiB on the device)
  size_t resident_bytes() const { return in_.capacity() + data_.capacity(); }

 private:
  // one call of the decoder: top the input piece up from the source, decode into the output piece
  void step() {
    if (!e_.ok()) {
      err_ = make_error(e_, e_.status());
      return;
    }
    if (!st_) {
      const int rc = flate_hip_inflate_stream_open(e_.ctx(), &st_);
      if (rc != 0) {
        err_ = make_error(e_, rc);
        return;
      }
      fresh_ = !dict_.empty();
    }
    if (fresh_) {
      const int rc = flate_hip_inflate_stream_reset(st_, dict_.data(), dict_.size());
      if (rc != 0) {
        err_ = make_error(e_, rc);
        return;
      }
      fresh_ = false;
    }
    while (!src_end_ && in_.size() < in_piece_) {
      const size_t at = in_.size();
      in_.resize(in_piece_);
      auto r = r_->read(in_.data() + at, in_piece_ - at);
      in_.resize(at + (size_t)r.first);
      if (r.second) {
        if (!(*r.second == ioeof())) {  // the source's own error ends the stream (more_bits, :771-787)
          err_ = r.second;
          return;
        }
        src_end_ = true;
      }
    }
    data_.resize(out_piece_);
    pos_ = 0;
    uint64_t used = 0, got = 0;
    int64_t eoff = -1;
    const int rc = flate_hip_inflate_stream_read(st_, in_.data(), in_.size(), src_end_ ? 1 : 0, data_.data(),
                                                 out_piece_, &used, &got, &eoff);
    data_.resize((size_t)got);
    in_.erase(in_.begin(), in_.begin() + (size_t)used);
    if (rc == FLATE_HIP_STREAM_END) err_ = ioeof();
    else if (rc == FLATE_HIP_E_CORRUPT) err_ = corrupt_input_error(eoff);
    else if (rc == FLATE_HIP_E_UNEXPECTED_EOF) err_ = err_unexpected_eof();
    else if (rc != 0) err_ = make_error(e_, rc);
    else if (got == 0 && used == 0 && (src_end_ || in_.size() >= in_piece_))
      err_ = make_error(e_, FLATE_HIP_E_INTERNAL);  // (no progress although the decoder has what it may ask for)
  }
  ByteSource *r_;
  Engine &e_;
  size_t in_piece_, out_piece_;
  flate_hip_inflate_stream *st_ = nullptr;
  std::vector<uint8_t> dict_;  // preset dictionary of the stream being decoded (kept for nothing else)
  bool fresh_ = false;         // the handle must be reset before it decodes
  std::vector<uint8_t> in_;   // the bytes of the stream the decoder has not used yet
  bool src_end_ = false;
  std::vector<uint8_t> data_;  // decoded, not handed out yet (to_read, inflate.mbt:286)
  size_t pos_ = 0;
  Err err_;
};

// One spliced stream decoded in parallel from its index (flate_hip_inflate_spliced).
inline Err decompress_spliced(Engine &e, const std::vector<uint8_t> &stream, const std::vector<uint64_t> &bit_off,
                              const std::vector<uint64_t> &sizes, std::vector<uint8_t> &out) {
  if (!e.ok()) return make_error(e, e.status());
  const uint32_t n = (uint32_t)sizes.size();
  std::vector<uint64_t> out_off(n + 1, 0), out_len(n + 1, 0);
  std::vector<int32_t> status(n + 1, 0);
  std::vector<int64_t> err_off(n + 1, -1);
  for (uint32_t i = 0; i < n; ++i) out_off[i + 1] = out_off[i] + sizes[i];
  std::vector<uint8_t> in(stream);
  in.resize(in.size() + 8);
  out.assign(out_off[n] + 8, 0);
  const int rc = flate_hip_inflate_spliced(e.ctx(), in.data(), stream.size(), bit_off.data(), n, out.data(),
                                           out_off.data(), out_len.data(), status.data(), err_off.data(), 0);
  if (rc != 0) {
    for (uint32_t i = 0; i < n; ++i)
      if (status[i] == FLATE_HIP_E_CORRUPT) return corrupt_input_error(err_off[i]);
    return make_error(e, rc);
  }
  out.resize(out_off[n]);
  return std::nullopt;
}

}  // namespace flate_host
