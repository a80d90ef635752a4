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

static std::vector<uint8_t> read_all(Reader &r, Err &last) {
  std::vector<uint8_t> got, piece(5000);
  for (int k = 0; k < 100000; ++k) {
    auto rr = r.read(piece.data(), piece.size());
    got.insert(got.end(), piece.begin(), piece.begin() + rr.first);
    if (rr.second) {
      last = rr.second;
      break;
    }
  }
  return got;
}

int main(int argc, char **argv) {
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
  // deflate_test.mbt:25-35: new_dict("hello world") + write("hello again world") + close gives the SAME bytes
  {
    Buffer b1;
    auto wd = Writer::new_dict(b1, eng, (const uint8_t *)d1, strlen(d1));
    auto rd = wd->write((const uint8_t *)d2, strlen(d2));
    Err ed = wd->close();
    printf("dictwrite %d %s %d\n", rd.first, ed ? ed->msg.c_str() : "none", b1.bytes == b.bytes ? 1 : 0);
    hex("dicthello", b1.bytes);
  }
  // a batch: empty stream, 65536-byte ramp, 100 zero bytes
  std::vector<std::vector<uint8_t>> in(3), out;
  in[1].resize(65536);
  for (size_t i = 0; i < in[1].size(); ++i) in[1][i] = (uint8_t)(i & 127);
  in[2].assign(100, 0);
  Err be = compress_batch(eng, in, out);
  printf("batch %s\n", be ? be->msg.c_str() : "none");
  if (!be)
    for (size_t i = 0; i < out.size(); ++i) hex("stream", out[i]);

  // the same three streams as zlib and as gzip members (checksums from the GPU)
  {
    std::vector<std::vector<uint8_t>> z, g;
    Err ze = compress_batch(eng, in, z, Wrap::Zlib);
    Err ge = compress_batch(eng, in, g, Wrap::Gzip);
    printf("framed %s %s\n", ze ? ze->msg.c_str() : "none", ge ? ge->msg.c_str() : "none");
    for (size_t i = 0; i < z.size(); ++i) hex("zlibm", z[i]);
    for (size_t i = 0; i < g.size(); ++i) hex("gzipm", g[i]);
    // and back; member 1 of the gzip batch with a damaged CRC, member 2 of the zlib batch with a bad header
    if (g.size() == 3 && z.size() == 3) {
      g[1][g[1].size() - 6] ^= 0x10;
      z[2][0] = 0x79;
      std::vector<Inflated> zb, gb;
      Err e1 = decompress_batch(eng, z, {0, 65536, 100}, zb, Wrap::Zlib);
      Err e2 = decompress_batch(eng, g, {}, gb, Wrap::Gzip);
      printf("unframed %s %s", e1 ? e1->msg.c_str() : "none", e2 ? e2->msg.c_str() : "none");
      for (auto *b : {&zb, &gb})
        for (size_t i = 0; i < b->size(); ++i) printf(" %d:%zu", (*b)[i].status, (*b)[i].bytes.size());
      printf("\n");
    }
  }

  // &Reader::new + read in 7-byte pieces + close (inflate.mbt:305,382-405,410-415)
  {
    BytesReader src(b.bytes);
    Reader r(src, eng);
    uint8_t piece[7];
    std::vector<uint8_t> got;
    for (int k = 0; k < 7; ++k) {
      auto rr = r.read(piece, sizeof piece);
      got.insert(got.end(), piece, piece + rr.first);
      printf("read %d %s\n", rr.first, rr.second ? rr.second->msg.c_str() : "none");
    }
    Err ce = r.close();
    printf("rclose %s\n", ce ? ce->msg.c_str() : "none");
    hex("plain", got);
    // Decompressor::reset (inflate.mbt:862): the same object decodes another stream afterwards
    BytesReader src2(out[2]);
    r.reset(src2);
    std::vector<uint8_t> big(200);
    auto r2 = r.read(big.data(), big.size());
    size_t nz = 0;
    for (int k = 0; k < r2.first; ++k) nz += big[k] != 0;
    printf("reset %d %s %zu\n", r2.first, r2.second ? r2.second->msg.c_str() : "none", nz);
  }
  // a source that holds more than the stream: the reader stops at the end of the final block, what it
  // pulled ahead is handed back; and make_reader (inflate.mbt:857-860) before the first read
  {
    std::vector<uint8_t> two = b.bytes;
    const char *behind = "NEXT PAYLOAD";
    two.insert(two.end(), behind, behind + strlen(behind));
    BytesReader other(std::vector<uint8_t>{0x07}), src(two);
    Reader r(other, eng);
    r.make_reader(src);
    Err last;
    std::vector<uint8_t> got = read_all(r, last);
    printf("trailing %zu %s %.*s\n", got.size(), last ? last->msg.c_str() : "none", (int)r.unread().size(),
           (const char *)r.unread().data());
  }
  // a corrupt stream: reserved block type in the first header (bits 1,1,1)
  {
    std::vector<uint8_t> bad = b.bytes;
    bad[0] = 0x07;
    BytesReader src(bad);
    Reader r(src, eng);
    uint8_t piece[64];
    auto rr = r.read(piece, sizeof piece);
    printf("badread %d %s\n", rr.first, rr.second ? rr.second->msg.c_str() : "none");
    Err ce = r.close();
    printf("badclose %s\n", ce ? ce->msg.c_str() : "none");
  }
  // truncated: unexpected EOF after the decodable part has been handed out
  {
    std::vector<uint8_t> cut(out[1].begin(), out[1].end() - 9);
    BytesReader src(cut);
    Reader r(src, eng, 70000);
    std::vector<uint8_t> piece(70000);
    auto rr = r.read(piece.data(), piece.size());
    printf("cutread %d %s\n", rr.first, rr.second ? rr.second->msg.c_str() : "none");
  }
  // spliced pair: the same three streams as ONE DEFLATE stream and back
  {
    std::vector<uint8_t> one;
    std::vector<uint64_t> bit_off;
    Err se = compress_spliced(eng, in, one, &bit_off);
    printf("spliced %s\n", se ? se->msg.c_str() : "none");
    hex("one", one);
    std::vector<uint8_t> back;
    Err de = decompress_spliced(eng, one, bit_off, {0, 65536, 100}, back);
    size_t diff = 0;
    for (size_t i = 0; i < back.size(); ++i) diff += back[i] != (i < 65536 ? in[1][i] : in[2][i - 65536]);
    printf("unspliced %s %zu %zu\n", de ? de->msg.c_str() : "none", back.size(), diff);
  }
  // BatchWriter: three Writers closed by one kernel pipeline, each with its own sink and state
  {
    BatchWriter bw(eng);
    Buffer s0, s1, s2;
    Writer &w0 = bw.add(s0), &w1 = bw.add(s1), &w2 = bw.add(s2);
    w0.write((const uint8_t *)d1, strlen(d1));
    w0.write((const uint8_t *)d2, strlen(d2));
    w1.write(in[1]);
    (void)w2;  // nothing written: the empty stream
    Err ce = bw.close_all();
    printf("bw_close %s\n", ce ? ce->msg.c_str() : "none");
    hex("bw0", s0.bytes);
    hex("bw1", s1.bytes);
    hex("bw2", s2.bytes);
    auto ra = w0.write((const uint8_t *)d1, 3);
    printf("bw_write_after_close %d %s\n", ra.first, ra.second ? ra.second->msg.c_str() : "none");
    Err c2 = bw.close_all();
    printf("bw_close2 %s\n", c2 ? c2->msg.c_str() : "none");
  }
  // opt-in chunked Writer: one 300000-byte stream cut into 65536-byte independent
  // chunks, compressed in one batch, emitted as ONE spliced stream
  {
    std::vector<uint8_t> big(300000);
    uint32_t x = 12345;
    for (size_t i = 0; i < big.size(); ++i) {
      x = x * 1664525u + 1013904223u;
      big[i] = (uint8_t)("etaoin shrdlu"[(x >> 24) % 13]);
    }
    Buffer sink;
    Writer cw(sink, eng, 0, 65536);
    cw.write(big.data(), 100000);
    cw.write(big.data() + 100000, 200000);
    Err ce = cw.close();
    printf("chunked %s %zu\n", ce ? ce->msg.c_str() : "none", sink.bytes.size());
    hex("chunked_bytes", sink.bytes);
    BytesReader src(sink.bytes);
    Reader r(src, eng, 300000);
    std::vector<uint8_t> back(300001);
    auto rr = r.read(back.data(), back.size());
    size_t diff = 0;
    for (int k = 0; k < rr.first && k < 300000; ++k) diff += back[k] != big[k];
    printf("chunked_back %d %s %zu\n", rr.first, rr.second ? rr.second->msg.c_str() : "none", diff);
  }
  // Writer on ONE long stream (the default, bit-exact mode): output reaches the sink while input is
  // still arriving (Compressor::write, deflate.mbt:280-294), not only at close
  {
    std::vector<uint8_t> big(5 * 65535 + 4321);
    uint32_t x = 777;
    for (size_t i = 0; i < big.size(); ++i) {
      x = x * 1664525u + 1013904223u;
      big[i] = (uint8_t)("the quick brown fox "[(x >> 24) % 20]);
    }
    Buffer sink;
    Writer sw(sink, eng, 0, 0, 2);  // hand over every two full windows
    size_t at = 0, seen_before_close = 0;
    while (at < big.size()) {
      const size_t k = std::min<size_t>(50000, big.size() - at);
      sw.write(big.data() + at, k);
      at += k;
      if (at < big.size()) seen_before_close = sink.bytes.size();
    }
    Err ce = sw.close();
    printf("streamed %s %zu %zu\n", ce ? ce->msg.c_str() : "none", seen_before_close, sink.bytes.size());
    hex("streamed_bytes", sink.bytes);
    // Reader without a size hint: one size pass + one decode, no retries
    BytesReader src(sink.bytes);
    Reader r(src, eng);
    std::vector<uint8_t> back(big.size() + 1);
    auto rr = r.read(back.data(), back.size());
    size_t diff = 0;
    for (int k = 0; k < rr.first && (size_t)k < big.size(); ++k) diff += back[k] != big[k];
    printf("streamed_back %d %s %zu\n", rr.first, rr.second ? rr.second->msg.c_str() : "none", diff);
    // and a WRONG hint is corrected from the status code, not from a message
    BytesReader src2(sink.bytes);
    Reader r2(src2, eng, 1000);
    auto r2r = r2.read(back.data(), back.size());
    printf("wrong_hint %d %s\n", r2r.first, r2r.second ? r2r.second->msg.c_str() : "none");
  }
  // Reader on a long stream with SMALL pieces: it never holds more than its two pieces (the reference's
  // Decompressor holds a 32 KiB window and a 4-byte buffer, inflate.mbt:252-290)
  {
    std::vector<uint8_t> big(6 * 1000 * 1000);
    uint32_t x = 4242;
    for (size_t i = 0; i < big.size(); ++i) {
      x = x * 1664525u + 1013904223u;
      big[i] = (uint8_t)("a stream long enough not to fit "[(x >> 24) % 32]);
    }
    std::vector<std::vector<uint8_t>> cin{big}, cout;
    Err ce = compress_batch(eng, cin, cout);
    BytesReader src(cout.empty() ? std::vector<uint8_t>{} : cout[0]);
    Reader r(src, eng, 0, 65536, 100000);
    std::vector<uint8_t> piece(33333);
    size_t total = 0, diff = 0, reads = 0, most = 0;
    Err last;
    for (;;) {
      auto rr = r.read(piece.data(), piece.size());
      for (int k = 0; k < rr.first; ++k) diff += total + (size_t)k >= big.size() || piece[k] != big[total + (size_t)k];
      total += (size_t)rr.first;
      ++reads;
      most = std::max(most, r.resident_bytes());
      if (rr.second) {
        last = rr.second;
        break;
      }
      if (reads > 100000) break;
    }
    printf("long_reader %s %zu %s %zu %d\n", ce ? ce->msg.c_str() : "none", total, last ? last->msg.c_str() : "none", diff,
           most <= 65536 + 100000 + 8192 ? 1 : 0);
  }
  // &Reader::new_dict / Decompressor::reset(r, dict) (inflate.mbt:315-317,862-884).  argv[1]: a file
  // the test wrote -- u32 dictionary length, dictionary, u32 stream length, a stream compressed against it
  if (argc > 1) {
    FILE *f = fopen(argv[1], "rb");
    uint32_t dl = 0, cl = 0;
    std::vector<uint8_t> dict, comp;
    if (f && fread(&dl, 4, 1, f) == 1) {
      dict.resize(dl);
      if (dl && fread(dict.data(), 1, dl, f) != dl) dict.clear();
      if (fread(&cl, 4, 1, f) == 1) {
        comp.resize(cl);
        if (cl && fread(comp.data(), 1, cl, f) != cl) comp.clear();
      }
    }
    if (f) fclose(f);
    BytesReader src(comp);
    auto r = Reader::new_dict(src, eng, dict, 4096, 7000);
    Err last;
    std::vector<uint8_t> got = read_all(*r, last);
    printf("dict_read %zu %s\n", got.size(), last ? last->msg.c_str() : "none");
    hex("dict_plain", got);
    BytesReader src2(comp);  // the same handle without the dictionary: the first far copy is corrupt
    r->reset(src2);
    Err last2;
    std::vector<uint8_t> got2 = read_all(*r, last2);
    printf("nodict_read %zu %s\n", got2.size(), last2 ? last2->msg.c_str() : "none");
    BytesReader src3(comp);  // and with it again
    r->reset(src3, dict);
    Err last3;
    std::vector<uint8_t> got3 = read_all(*r, last3);
    printf("redict_read %zu %s %d\n", got3.size(), last3 ? last3->msg.c_str() : "none", got3 == got ? 1 : 0);
  }
  return 0;
}
