/*
 * inflate.c -- TEST INFRASTRUCTURE (oracle).  CPU restatement of
 * /root/reference/inflate.mbt and dict-decoder.mbt as a whole-buffer decoder.
 *
 * The reference is a resumable step machine over a 32 KiB circular DictDecoder
 * that hands slices to the caller (inflate.mbt:382-407, dict-decoder.mbt).
 * With the whole output in one flat buffer the history IS the output, so
 * hist_size() (dict-decoder.mbt:63-68) == min(bytes produced, 32768) and
 * write_copy/try_write_copy (dict-decoder.mbt:114-185) reduce to the forward
 * byte copy below.  Bit reading, table construction, header validation, symbol
 * decoding and every error condition (with its roffset) follow the reference.
 */
#include "flate_oracle.h"

#include <string.h>

#define MAX_CODE_LEN 16        /* inflate.mbt:22 */
#define MAX_NUM_LIT 286        /* inflate.mbt:28 */
#define MAX_NUM_DIST 30        /* inflate.mbt:31 */
#define NUM_CODES 19           /* inflate.mbt:34 */
#define HUFFMAN_CHUNK_BITS 9   /* inflate.mbt:69 */
#define HUFFMAN_NUM_CHUNKS 512 /* inflate.mbt:72 */
#define HUFFMAN_COUNT_MASK 15u /* inflate.mbt:75 */
#define HUFFMAN_VALUE_SHIFT 4  /* inflate.mbt:78 */
#define END_BLOCK_MARKER 256
#define HIST_SIZE 32768 /* max_match_offset, inflate.mbt:330 */

/* inflate.mbt:81-86.  links: at most 512 link tables of at most 2^(15-9)=64. */
typedef struct {
  int min;
  uint32_t chunks[HUFFMAN_NUM_CHUNKS];
  uint32_t links[HUFFMAN_NUM_CHUNKS][64];
  int nlinks;
  int link_len;
  uint32_t link_mask;
} huffman_decoder;

/* inflate.mbt:100-223 */
static int hd_initialize(huffman_decoder *h, const int *lengths, int nlen) {
  if (h->min != 0) {
    h->min = 0;
    memset(h->chunks, 0, sizeof(h->chunks));
    h->nlinks = 0;
    h->link_len = 0;
    h->link_mask = 0;
  }

  int count[MAX_CODE_LEN];
  memset(count, 0, sizeof(count));
  int min = 0, max = 0;
  for (int k = 0; k < nlen; k++) {
    int n = lengths[k];
    if (n == 0) continue;
    if (min == 0 || n < min) min = n;
    if (n > max) max = n;
    count[n]++;
  }
  if (max == 0) return 1; /* empty tree (:143-145) */

  int code = 0;
  int nextcode[MAX_CODE_LEN];
  memset(nextcode, 0, sizeof(nextcode));
  for (int i = min; i <= max; i++) {
    code <<= 1;
    nextcode[i] = code;
    code += count[i];
  }
  if (code != (1 << max) && !(code == 1 && max == 1)) return 0; /* :161 */

  h->min = min;
  if (max > HUFFMAN_CHUNK_BITS) { /* :167-188 */
    uint32_t num_links = 1u << (max - HUFFMAN_CHUNK_BITS);
    h->link_mask = num_links - 1;
    int link = nextcode[HUFFMAN_CHUNK_BITS + 1] >> 1;
    h->nlinks = HUFFMAN_NUM_CHUNKS - link;
    h->link_len = (int)num_links;
    for (uint32_t j = (uint32_t)link; j < HUFFMAN_NUM_CHUNKS; j++) {
      int reverse = (int)orc_reverse16(j & 0xffff);
      reverse >>= (16 - HUFFMAN_CHUNK_BITS);
      uint32_t off = j - (uint32_t)link;
      h->chunks[reverse] = (off << HUFFMAN_VALUE_SHIFT) | (HUFFMAN_CHUNK_BITS + 1);
      memset(h->links[off], 0, sizeof(uint32_t) * num_links);
    }
  }

  for (int i = 0; i < nlen; i++) { /* :191-221 */
    int n = lengths[i];
    if (n == 0) continue;
    int c = nextcode[n];
    nextcode[n]++;
    uint32_t chunk = ((uint32_t)i << HUFFMAN_VALUE_SHIFT) | (uint32_t)n;
    int reverse = (int)orc_reverse16((uint32_t)c & 0xffff);
    reverse >>= (16 - n);
    if (n <= HUFFMAN_CHUNK_BITS) {
      for (int off = reverse; off < HUFFMAN_NUM_CHUNKS; off += (1 << n))
        h->chunks[off] = chunk;
    } else {
      int j = reverse & (HUFFMAN_NUM_CHUNKS - 1);
      uint32_t value = h->chunks[j] >> HUFFMAN_VALUE_SHIFT;
      uint32_t *linktab = h->links[value];
      reverse >>= HUFFMAN_CHUNK_BITS;
      for (int off = reverse; off < h->link_len; off += (1 << (n - HUFFMAN_CHUNK_BITS)))
        linktab[off] = chunk;
    }
  }
  return 1;
}

/* inflate.mbt:257-291, flattened */
typedef struct {
  const uint8_t *in;
  size_t in_len;
  long long roffset;
  uint32_t b;
  uint32_t nb;
  huffman_decoder h1, h2;
  int bits[MAX_NUM_LIT + MAX_NUM_DIST];
  int codebits[NUM_CODES];
  uint8_t *out;
  size_t out_len, out_cap;
  /* preset dictionary (&Reader::new_dict, inflate.mbt:315-317; Decompressor::reset :862-884):
   * DictDecoder::new (dict-decoder.mbt:40-60) keeps its last HIST_SIZE bytes as history that
   * "has already been read": the output behaves as if it started with them. */
  const uint8_t *dict;
  size_t dict_len; /* <= HIST_SIZE */
  int final_flag;
  int err;
  long long err_off;
} decompressor;

static int corrupt(decompressor *f) { /* inflate.mbt:38 */
  f->err = ORC_E_CORRUPT;
  f->err_off = f->roffset;
  return f->err;
}

/* inflate.mbt:789-799; returns nonzero at end of input (ioeof) */
static int more_bits(decompressor *f) {
  if ((size_t)f->roffset >= f->in_len) return 1;
  uint8_t c = f->in[f->roffset];
  f->roffset++;
  f->b |= (uint32_t)c << f->nb;
  f->nb += 8;
  return 0;
}

/* inflate.mbt:803-854.  Returns the symbol or -1 with f->err set. */
static int huff_sym(decompressor *f, const huffman_decoder *h) {
  uint32_t n = (uint32_t)h->min;
  uint32_t nb = f->nb, b = f->b;
  for (;;) {
    while (nb < n) {
      if ((size_t)f->roffset >= f->in_len) {
        f->b = b;
        f->nb = nb;
        f->err = ORC_E_UNEXPECTED_EOF; /* no_eof(), :781 */
        return -1;
      }
      uint8_t c = f->in[f->roffset];
      f->roffset++;
      b |= (uint32_t)c << (nb & 31);
      nb += 8;
    }
    uint32_t chunk = h->chunks[b & (HUFFMAN_NUM_CHUNKS - 1)];
    n = chunk & HUFFMAN_COUNT_MASK;
    if (n > HUFFMAN_CHUNK_BITS) {
      chunk = h->links[chunk >> HUFFMAN_VALUE_SHIFT][(b >> HUFFMAN_CHUNK_BITS) & h->link_mask];
      n = chunk & HUFFMAN_COUNT_MASK;
    }
    if (n <= nb) {
      if (n == 0) {
        f->b = b;
        f->nb = nb;
        corrupt(f);
        return -1;
      }
      f->b = b >> (n & 31);
      f->nb = nb - n;
      return (int)(chunk >> HUFFMAN_VALUE_SHIFT);
    }
  }
}

/* inflate.mbt:424-426 */
static const int code_order[NUM_CODES] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                          11, 4,  12, 3, 13, 2, 14, 1, 15};

/* inflate.mbt:429-548 */
static int read_huffman(decompressor *f) {
  while (f->nb < 5 + 5 + 4)
    if (more_bits(f)) return f->err = ORC_E_UNEXPECTED_EOF; /* see note below */
  int nlit = (int)(f->b & 0x1F) + 257;
  if (nlit > MAX_NUM_LIT) return corrupt(f);
  f->b >>= 5;
  int ndist = (int)(f->b & 0x1F) + 1;
  if (ndist > MAX_NUM_DIST) return corrupt(f);
  f->b >>= 5;
  int nclen = (int)(f->b & 0xF) + 4;
  f->b >>= 4;
  f->nb -= 5 + 5 + 4;

  for (int i = 0; i < nclen; i++) {
    while (f->nb < 3)
      if (more_bits(f)) return f->err = ORC_E_UNEXPECTED_EOF;
    f->codebits[code_order[i]] = (int)(f->b & 0x7);
    f->b >>= 3;
    f->nb -= 3;
  }
  for (int i = nclen; i < NUM_CODES; i++) f->codebits[code_order[i]] = 0;
  if (!hd_initialize(&f->h1, f->codebits, NUM_CODES)) return corrupt(f);

  int i = 0;
  int n = nlit + ndist;
  while (i < n) {
    int x = huff_sym(f, &f->h1);
    if (x < 0) return f->err;
    if (x < 16) {
      f->bits[i] = x;
      i++;
      continue;
    }
    int rep = 0;
    uint32_t nb = 0;
    int b = 0;
    switch (x) {
      case 16:
        rep = 3;
        nb = 2;
        if (i == 0) return corrupt(f);
        b = f->bits[i - 1];
        break;
      case 17:
        rep = 3;
        nb = 3;
        b = 0;
        break;
      case 18:
        rep = 11;
        nb = 7;
        b = 0;
        break;
      default:
        return f->err = ORC_E_INTERNAL;
    }
    while (f->nb < nb)
      if (more_bits(f)) return f->err = ORC_E_UNEXPECTED_EOF;
    rep += (int)(f->b & ((1u << nb) - 1));
    f->b >>= nb;
    f->nb -= nb;
    if (i + rep > n) return corrupt(f);
    for (int j = 0; j < rep; j++) {
      f->bits[i] = b;
      i++;
    }
  }

  if (!hd_initialize(&f->h1, f->bits, nlit) || !hd_initialize(&f->h2, f->bits + nlit, ndist))
    return corrupt(f);

  if (f->h1.min < f->bits[END_BLOCK_MARKER]) f->h1.min = f->bits[END_BLOCK_MARKER];
  return 0;
}
/* Note on end-of-input inside a header: the reference returns the raw ioeof
 * from more_bits there (inflate.mbt:431-436,454-458,512-516) whereas huff_sym
 * maps it to err_unexpected_eof (:824).  Both are "stream ended early"; the
 * oracle reports ORC_E_UNEXPECTED_EOF for either. */

static int put_byte(decompressor *f, uint8_t c) {
  if (f->out_len >= f->out_cap) return f->err = ORC_E_OUT_TOO_SMALL;
  f->out[f->out_len++] = c;
  return 0;
}

/* inflate.mbt:565-684 (read_literal) + 689-704 (copy_history) for one block.
 * hd == NULL means the fixed distance coding of fixed-Huffman blocks. */
static int huffman_block(decompressor *f, const huffman_decoder *hl,
                         const huffman_decoder *hd) {
  for (;;) {
    int v = huff_sym(f, hl);
    if (v < 0) return f->err;
    uint32_t n = 0;
    int length = 0;
    if (v < 256) {
      if (put_byte(f, (uint8_t)v)) return f->err;
      continue;
    }
    if (v == 256) return 0; /* finish_block */
    if (v < 265) {
      length = v - (257 - 3);
      n = 0;
    } else if (v < 269) {
      length = v * 2 - (265 * 2 - 11);
      n = 1;
    } else if (v < 273) {
      length = v * 4 - (269 * 4 - 19);
      n = 2;
    } else if (v < 277) {
      length = v * 8 - (273 * 8 - 35);
      n = 3;
    } else if (v < 281) {
      length = v * 16 - (277 * 16 - 67);
      n = 4;
    } else if (v < 285) {
      length = v * 32 - (281 * 32 - 131);
      n = 5;
    } else if (v < MAX_NUM_LIT) {
      length = 258;
      n = 0;
    } else {
      return corrupt(f);
    }
    if (n > 0) {
      while (f->nb < n)
        if (more_bits(f)) return f->err = ORC_E_UNEXPECTED_EOF;
      length += (int)(f->b & ((1u << n) - 1));
      f->b >>= n;
      f->nb -= n;
    }

    int dist = 0;
    if (!hd) { /* :632-641 */
      while (f->nb < 5)
        if (more_bits(f)) return f->err = ORC_E_UNEXPECTED_EOF;
      uint32_t to_rev = ((f->b & 0x1F) << 3) & 0xff;
      dist = (int)(orc_reverse16(to_rev << 8) & 0xff); /* reverse8 */
      f->b >>= 5;
      f->nb -= 5;
    } else {
      dist = huff_sym(f, hd);
      if (dist < 0) return f->err;
    }

    if (dist < 4) { /* :656-674 */
      dist++;
    } else if (dist < MAX_NUM_DIST) {
      uint32_t nb = (uint32_t)(dist - 2) >> 1;
      int extra = (dist & 1) << nb;
      while (f->nb < nb)
        if (more_bits(f)) return f->err = ORC_E_UNEXPECTED_EOF;
      extra |= (int)(f->b & ((1u << nb) - 1));
      f->b >>= nb;
      f->nb -= nb;
      dist = (1 << (nb + 1)) + 1 + extra;
    } else {
      return corrupt(f);
    }

    /* hist_size (dict-decoder.mbt:63-69): the window is full, or wr_pos = dictionary + output */
    size_t hist = f->out_len + f->dict_len < HIST_SIZE ? f->out_len + f->dict_len : HIST_SIZE;
    if ((size_t)dist > hist) return corrupt(f); /* :677-680 */

    if (f->out_len + (size_t)length > f->out_cap) return f->err = ORC_E_OUT_TOO_SMALL;
    uint8_t *dst = f->out + f->out_len;
    if ((size_t)dist <= f->out_len) {
      const uint8_t *src = dst - dist;
      for (int i = 0; i < length; i++) dst[i] = src[i]; /* forward (overlapping) copy */
    } else { /* the copy starts inside the preset dictionary and may run on into the output */
      long long p = (long long)f->out_len - dist; /* < 0: dictionary byte dict_len + p */
      for (int i = 0; i < length; i++, p++) dst[i] = p < 0 ? f->dict[(long long)f->dict_len + p] : f->out[p];
    }
    f->out_len += (size_t)length;
  }
}

/* inflate.mbt:708-766 (data_block + copy_data) */
static int data_block(decompressor *f) {
  f->nb = 0;
  f->b = 0;
  size_t avail = f->in_len - (size_t)f->roffset;
  size_t nr = avail < 4 ? avail : 4;
  const uint8_t *buf = f->in + f->roffset;
  f->roffset += (long long)nr;
  if (nr < 4) return f->err = ORC_E_UNEXPECTED_EOF;
  int n = buf[0] | (buf[1] << 8);
  int nn = buf[2] | (buf[3] << 8);
  if ((nn & 0xffff) != ((~n) & 0xffff)) return corrupt(f);
  if (n == 0) return 0;
  avail = f->in_len - (size_t)f->roffset;
  size_t cnt = avail < (size_t)n ? avail : (size_t)n;
  if (f->out_len + cnt > f->out_cap) return f->err = ORC_E_OUT_TOO_SMALL;
  memcpy(f->out + f->out_len, f->in + f->roffset, cnt);
  f->out_len += cnt;
  f->roffset += (long long)cnt;
  if (cnt < (size_t)n) return f->err = ORC_E_UNEXPECTED_EOF;
  return 0;
}

/* fixed_huffman_decoder, inflate.mbt:886-939: the literal table there equals
 * initialize() over the RFC 1951 3.2.6 lengths with min = 7. */
static huffman_decoder fixed_decoder;
static int fixed_ready;
static const huffman_decoder *get_fixed(void) {
  if (!fixed_ready) {
    int bits[288];
    for (int i = 0; i < 144; i++) bits[i] = 8;
    for (int i = 144; i < 256; i++) bits[i] = 9;
    for (int i = 256; i < 280; i++) bits[i] = 7;
    for (int i = 280; i < 288; i++) bits[i] = 8;
    memset(&fixed_decoder, 0, sizeof(fixed_decoder));
    hd_initialize(&fixed_decoder, bits, 288);
    fixed_ready = 1;
  }
  return &fixed_decoder;
}

/* inflate.mbt:345-379 (next_block) driven until final_flag (finish_block :769) */
int orc_inflate_stream(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                       size_t *out_len, size_t *consumed, long long *err_off) {
  return orc_inflate_stream_dict(in, n, NULL, 0, out, cap, out_len, consumed, err_off);
}

int orc_inflate_stream_dict(const uint8_t *in, size_t n, const uint8_t *dict, size_t dict_len, uint8_t *out,
                            size_t cap, size_t *out_len, size_t *consumed, long long *err_off) {
  static __thread decompressor fs;
  decompressor *f = &fs;
  if (dict_len > HIST_SIZE) { /* dict-decoder.mbt:48-50: only the last `size` bytes are kept */
    dict += dict_len - HIST_SIZE;
    dict_len = HIST_SIZE;
  }
  f->dict = dict;
  f->dict_len = dict_len;
  f->in = in;
  f->in_len = n;
  f->roffset = 0;
  f->b = 0;
  f->nb = 0;
  f->h1.min = 1; /* force re-initialisation of reused decoders */
  f->h2.min = 1;
  f->out = out;
  f->out_len = 0;
  f->out_cap = cap;
  f->final_flag = 0;
  f->err = 0;
  f->err_off = -1;
  const huffman_decoder *fixed = get_fixed();

  while (!f->final_flag && !f->err) {
    int eof = 0;
    while (f->nb < 1 + 2) {
      if (more_bits(f)) {
        eof = 1;
        break;
      }
    }
    if (eof) {
      f->err = ORC_E_UNEXPECTED_EOF;
      break;
    }
    f->final_flag = (f->b & 1) == 1;
    f->b >>= 1;
    uint32_t typ = f->b & 3;
    f->b >>= 2;
    f->nb -= 1 + 2;
    switch (typ) {
      case 0:
        data_block(f);
        break;
      case 1:
        huffman_block(f, fixed, NULL);
        break;
      case 2:
        if (read_huffman(f) == 0) huffman_block(f, &f->h1, &f->h2);
        break;
      default:
        corrupt(f);
        break;
    }
  }
  *out_len = f->out_len;
  if (consumed) *consumed = (size_t)f->roffset;
  if (err_off) *err_off = f->err_off;
  return f->err;
}

/* ---- batch driver: independent streams over a few host threads (bench.py's cpu_baseline leg
 * and the full-size property tests; the decoder itself is orc_inflate_stream above) ---- */
#include <pthread.h>
#include <stdlib.h>

typedef struct {
  const uint8_t *in;
  const uint64_t *in_off;
  uint8_t *out;
  const uint64_t *out_off;
  uint64_t *out_len;
  int32_t *status;
  uint32_t lo, hi;
} inf_job;

static void *inf_worker(void *arg) {
  inf_job *j = (inf_job *)arg;
  for (uint32_t i = j->lo; i < j->hi; i++) {
    size_t n = 0, used = 0;
    long long eo = -1;
    const int rc = orc_inflate_stream(j->in + j->in_off[i], (size_t)(j->in_off[i + 1] - j->in_off[i]),
                                      j->out + j->out_off[i], (size_t)(j->out_off[i + 1] - j->out_off[i]),
                                      &n, &used, &eo);
    j->out_len[i] = n;
    j->status[i] = rc;
  }
  return NULL;
}

int orc_inflate_batch(const uint8_t *in, const uint64_t *in_off, uint32_t n_streams, uint8_t *out,
                      const uint64_t *out_off, uint64_t *out_len, int32_t *status, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n_streams && n_streams > 0) nthreads = (int)n_streams;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  inf_job *jobs = (inf_job *)calloc((size_t)nthreads, sizeof(inf_job));
  for (int t = 0; t < nthreads; t++) {
    inf_job *j = &jobs[t];
    j->in = in;
    j->in_off = in_off;
    j->out = out;
    j->out_off = out_off;
    j->out_len = out_len;
    j->status = status;
    j->lo = (uint32_t)((uint64_t)n_streams * (uint64_t)t / (uint64_t)nthreads);
    j->hi = (uint32_t)((uint64_t)n_streams * (uint64_t)(t + 1) / (uint64_t)nthreads);
    if (nthreads > 1)
      pthread_create(&th[t], NULL, inf_worker, j);
    else
      inf_worker(j);
  }
  int rc = 0;
  for (int t = 0; t < nthreads; t++)
    if (nthreads > 1) pthread_join(th[t], NULL);
  for (uint32_t i = 0; i < n_streams && !rc; i++) rc = status[i];
  free(th);
  free(jobs);
  return rc;
}
