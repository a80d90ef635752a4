/*
 * checksum.c -- TEST INFRASTRUCTURE (oracle).  The two checksums and the two container formats that wrap a
 * raw DEFLATE stream ("optional gzip/zlib wrappers (absent from reference)").  The reference
 * has neither; these follow the formats' own specifications and are pinned against zlib's crc32() /
 * adler32() / decompress() in tests/test_checksum.py:
 *   Adler-32   RFC 1950 section 8.2 (and its sample code, section 9)
 *   CRC-32     RFC 1952 section 8 (sample code: reflected polynomial 0xedb88320, pre- and post-inverted)
 *   zlib frame RFC 1950 section 2.2: CMF, FLG, data, ADLER32 (big endian)
 *   gzip frame RFC 1952 section 2.3: ID1 ID2 CM FLG MTIME XFL OS, data, CRC32, ISIZE (little endian)
 */
#include "flate_oracle.h"

#include <string.h>

uint32_t orc_adler32(const uint8_t *p, size_t n) {
  uint32_t s1 = 1, s2 = 0; /* RFC 1950 section 9: update_adler32 */
  for (size_t i = 0; i < n; i++) {
    s1 = (s1 + p[i]) % 65521u;
    s2 = (s2 + s1) % 65521u;
  }
  return (s2 << 16) | s1;
}

uint32_t orc_crc32(const uint8_t *p, size_t n) {
  static uint32_t table[256];
  static int ready;
  if (!ready) { /* RFC 1952 section 8: make_crc_table */
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = 1;
  }
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

size_t orc_frame_overhead(int kind) { return kind == ORC_FRAME_ZLIB ? 6 : (kind == ORC_FRAME_GZIP ? 18 : 0); }

/* raw[0, raw_len) is the DEFLATE stream of data[0, n).  Returns the framed length (out needs raw_len + overhead). */
size_t orc_frame(int kind, const uint8_t *raw, size_t raw_len, const uint8_t *data, size_t n, uint8_t *out) {
  size_t o = 0;
  if (kind == ORC_FRAME_ZLIB) {
    out[o++] = 0x78; /* CM = 8, CINFO = 7: a 32 KiB window */
    out[o++] = 0x01; /* FLEVEL = 0 (fastest), no FDICT, FCHECK makes 0x7801 a multiple of 31 */
    memcpy(out + o, raw, raw_len);
    o += raw_len;
    const uint32_t a = orc_adler32(data, n);
    out[o++] = (uint8_t)(a >> 24);
    out[o++] = (uint8_t)(a >> 16);
    out[o++] = (uint8_t)(a >> 8);
    out[o++] = (uint8_t)a;
  } else if (kind == ORC_FRAME_GZIP) {
    static const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4 /* XFL: fastest */, 255 /* OS: unknown */};
    memcpy(out, hdr, 10);
    o = 10;
    memcpy(out + o, raw, raw_len);
    o += raw_len;
    const uint32_t c = orc_crc32(data, n);
    for (int k = 0; k < 4; k++) out[o++] = (uint8_t)(c >> (8 * k));
    for (int k = 0; k < 4; k++) out[o++] = (uint8_t)((uint32_t)n >> (8 * k)); /* ISIZE = n mod 2^32 */
  } else {
    memcpy(out, raw, raw_len);
    o = raw_len;
  }
  return o;
}
