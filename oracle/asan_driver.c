/* TEST INFRASTRUCTURE: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: the pool
 * has no GPU sanitizer).  Round trips of generated inputs in both compat modes, decoding of corrupted and
 * truncated streams (the decoder must refuse them without touching memory it does not own), and the
 * preset-dictionary decoder.  Prints "ASAN_DRIVER_OK <cases>" and exits 0. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flate_oracle.h"

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}

static void fill(uint8_t *p, size_t n, int kind) {
  static const char *words[] = {"the ", "quick ", "brown ", "fox ", "jumps ", "over ", "lazy ", "dog ", "and ", "again "};
  size_t i = 0;
  switch (kind) {
    case 0: /* text */
      while (i < n) {
        const char *w = words[rnd() % 10];
        for (; *w && i < n; ++w) p[i++] = (uint8_t)*w;
      }
      break;
    case 1: /* random */
      for (; i < n; ++i) p[i] = (uint8_t)rnd();
      break;
    case 2: /* ramp, deflate-fast_test.mbt:15-24 */
      for (; i < n; ++i) p[i] = (uint8_t)(i & 127);
      break;
    default: /* zeros */
      memset(p, 0, n);
  }
}

int main(void) {
  static const size_t sizes[] = {0, 1, 15, 16, 17, 127, 128, 129, 4000, 65534, 65535, 65536, 65537, 70000, 131071, 200000};
  size_t cases = 0;
  for (int kind = 0; kind < 4; ++kind)
    for (size_t si = 0; si < sizeof sizes / sizeof sizes[0]; ++si)
      for (int compat = 0; compat < 2; ++compat) {
        const size_t n = sizes[si];
        /* exact-size allocations: an access one byte past either end is reported */
        uint8_t *in = malloc(n ? n : 1);
        fill(in, n, kind);
        const size_t cap = orc_deflate_bound(n);
        uint8_t *comp = malloc(cap);
        size_t clen = 0;
        if (orc_deflate_stream(in, n, NULL, 0, comp, cap, &clen, compat) != 0) {
          fprintf(stderr, "deflate failed kind %d size %zu\n", kind, n);
          return 1;
        }
        uint8_t *tight = malloc(clen ? clen : 1);
        memcpy(tight, comp, clen);
        uint8_t *back = malloc(n ? n : 1);
        size_t blen = 0, used = 0;
        long long eo = -1;
        if (orc_inflate_stream(tight, clen, back, n, &blen, &used, &eo) != 0 || blen != n || memcmp(back, in, n) != 0) {
          fprintf(stderr, "round trip failed kind %d size %zu compat %d\n", kind, n, compat);
          return 1;
        }
        ++cases;
        /* corrupted and truncated copies: any status, no crash, never more than `cap` bytes written */
        for (int t = 0; t < 12 && clen > 0; ++t) {
          size_t m = (t & 1) ? 1 + rnd() % clen : clen;
          uint8_t *bad = malloc(m);
          memcpy(bad, tight, m);
          if (!(t & 1)) bad[rnd() % m] ^= (uint8_t)(1 + rnd() % 255);
          const size_t ocap = (t % 3 == 0) ? n / 2 : n + 100;
          uint8_t *o = malloc(ocap ? ocap : 1);
          (void)orc_inflate_stream(bad, m, o, ocap, &blen, &used, &eo);
          if (blen > ocap || used > m) {
            fprintf(stderr, "decoder overran its buffers\n");
            return 1;
          }
          free(o);
          free(bad);
          ++cases;
        }
        /* a dictionary changes nothing for a stream that never refers to one */
        if (n > 0 && n < 70000) {
          uint8_t dict[300];
          fill(dict, sizeof dict, 0);
          if (orc_inflate_stream_dict(tight, clen, dict, sizeof dict, back, n, &blen, &used, &eo) != 0 || blen != n ||
              memcmp(back, in, n) != 0) {
            fprintf(stderr, "dict round trip failed\n");
            return 1;
          }
          ++cases;
        }
        free(back);
        free(tight);
        free(comp);
        free(in);
      }
  /* a copy that reaches into a preset dictionary: fixed Huffman block, "abc" + match(len 4, dist 6) with the
   * dictionary "xyzuvw" in front: BFINAL=1 BTYPE=01, literals 'a' 'b' 'c' (8-bit codes 0x91 0x92 0x93),
   * length 4 = code 258 (7 bits 0000010), distance 6 = code 4 + 1 extra bit (5 bits 00100, extra 1), EOB */
  {
    /* bits, LSB-first per byte: 1 10 | 10010001 (a, MSB-first code) ... assembled by hand below */
    uint8_t bits[64];
    int nb = 0;
    memset(bits, 0, sizeof bits);
#define PUT(v, n_, msb)                                                           \
  for (int k_ = 0; k_ < (n_); ++k_) {                                             \
    int b_ = (msb) ? (((v) >> ((n_)-1 - k_)) & 1) : (((v) >> k_) & 1);            \
    bits[nb >> 3] |= (uint8_t)(b_ << (nb & 7));                                   \
    ++nb;                                                                         \
  }
    PUT(1, 1, 0) PUT(1, 2, 0)                       /* BFINAL, BTYPE=01 */
    PUT(0x30 + 'a', 8, 1) PUT(0x30 + 'b', 8, 1) PUT(0x30 + 'c', 8, 1)
    PUT(258 - 256, 7, 1)                            /* length code 258: 7-bit code 0000010 */
    PUT(4, 5, 1) PUT(1, 1, 0)                       /* distance code 4 (base 5) + extra 1 -> 6 */
    PUT(0, 7, 1)                                    /* end of block */
#undef PUT
    const size_t m = (size_t)(nb + 7) / 8;
    uint8_t *s = malloc(m);
    memcpy(s, bits, m);
    uint8_t out7[7];
    size_t blen = 0, used = 0;
    long long eo = -1;
    const uint8_t dict[6] = {'x', 'y', 'z', 'u', 'v', 'w'};
    int rc = orc_inflate_stream_dict(s, m, dict, 6, out7, 7, &blen, &used, &eo);
    if (rc != 0 || blen != 7 || memcmp(out7, "abcuvwa", 7) != 0) {
      fprintf(stderr, "dictionary copy: rc %d len %zu\n", rc, blen);
      return 1;
    }
    rc = orc_inflate_stream(s, m, out7, 7, &blen, &used, &eo); /* without it: distance 6 > 3 bytes of history */
    if (rc != ORC_E_CORRUPT || blen != 3) {
      fprintf(stderr, "missing dictionary not refused: rc %d len %zu\n", rc, blen);
      return 1;
    }
    free(s);
    cases += 2;
  }
  printf("ASAN_DRIVER_OK %zu\n", cases);
  return 0;
}
