// flate_host.hpp -- header-only C++ mirror of the reference's host API over the C ABI
// (include/flate_hip.h).  Names, argument meaning and error behaviour follow the reference:
//   Writer::new / write / close          writer.mbt:10,45,53
//   Compressor sticky errors             deflate.mbt:74,154-183,280-294
//   writer_closed_error                  deflate.mbt:154
// The reference is single-stream and synchronous; the GPU engine is a batch engine, so Writer
// buffers write() calls (the 65535-byte staging window of deflate.mbt:222-229 makes the output a
// function of the concatenated bytes only) and compresses in close().  BatchWriter closes many
// streams with one kernel pipeline -- the intended way to use the engine.
#pragma once

#include <cstdint>
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

// Writer (writer.mbt:2-15): write() appends to the stream, close() emits the DEFLATE bytes to the
// sink given at construction.  Sticky error rules of Compressor (deflate.mbt:157-183,280-294).
class Writer {
 public:
  Writer(ByteSink &w, Engine &e, uint32_t flags = 0) : w_(w), e_(e), flags_(flags) {}

  std::pair<int, Err> write(const uint8_t *p, size_t n) {  // deflate.mbt:280-294
    if (err_) return {0, err_};
    pending_.insert(pending_.end(), p, p + n);
    return {(int)n, std::nullopt};
  }
  std::pair<int, Err> write(const std::vector<uint8_t> &b) { return write(b.data(), b.size()); }

  Err close() {  // deflate.mbt:157-183
    if (err_ && *err_ == writer_closed_error()) return std::nullopt;
    if (err_) return err_;
    std::vector<std::vector<uint8_t>> out;
    Err er = compress_batch(e_, {pending_}, out, flags_);
    if (er) {
      err_ = er;
      return err_;
    }
    auto r = w_.write(out[0].data(), out[0].size());
    if (r.second) {
      err_ = r.second;
      return err_;
    }
    err_ = writer_closed_error();
    return std::nullopt;
  }

 private:
  ByteSink &w_;
  Engine &e_;
  uint32_t flags_;
  std::vector<uint8_t> pending_;
  Err err_;
};

}  // namespace flate_host
