// Test driver for the C++ host mirror (moonbit-flate_amd/host/flate_host.hpp); run on a GPU box.
#include <cstdio>
#include <cstring>

#include "flate_host.hpp"

using namespace flate_host;

static void hex(const char *tag, const std::vector<uint8_t> &b) {
  printf("%s ", tag);
  for (uint8_t c : b) printf("%02x", c);
  printf("\n");
}

int main() {
  Engine eng(0);
  if (!eng.ok()) {
    printf("noengine %d\n", eng.status());
    return 3;
  }
  // deflate_test.mbt:12-23: write("hello world") + write("hello again world") + close
  Buffer b;
  Writer w(b, eng);
  const char *d1 = "hello world", *d2 = "hello again world";
  auto r1 = w.write((const uint8_t *)d1, strlen(d1));
  auto r2 = w.write((const uint8_t *)d2, strlen(d2));
  printf("write %d %d\n", r1.first, r2.first);
  Err e = w.close();
  printf("close %s\n", e ? e->msg.c_str() : "none");
  hex("hello", b.bytes);
  Err e2 = w.close();  // deflate.mbt:158-160
  printf("close2 %s\n", e2 ? e2->msg.c_str() : "none");
  auto r3 = w.write((const uint8_t *)d1, 3);  // deflate.mbt:281-283
  printf("write_after_close %d %s\n", r3.first, r3.second ? r3.second->msg.c_str() : "none");
  // a batch: empty stream, 65536-byte ramp, 100 zero bytes
  std::vector<std::vector<uint8_t>> in(3), out;
  in[1].resize(65536);
  for (size_t i = 0; i < in[1].size(); ++i) in[1][i] = (uint8_t)(i & 127);
  in[2].assign(100, 0);
  Err be = compress_batch(eng, in, out);
  printf("batch %s\n", be ? be->msg.c_str() : "none");
  if (!be)
    for (size_t i = 0; i < out.size(); ++i) hex("stream", out[i]);
  return 0;
}
