/*
 * compressor.c -- TEST INFRASTRUCTURE (oracle).  CPU restatement of
 * /root/reference/deflate.mbt:46-100,157-196,222-294 and writer.mbt (the
 * Writer::new / write / close stream driver).  The preset-dictionary path
 * (new_dict / fill_window / bulk_hash4) is out of scope.
 */
#include "orc_internal.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* per-thread block trace (test aid, not in the reference) */
#define TRACE_MAX 4096
static __thread orc_block_info trace_blocks[TRACE_MAX];
static __thread int trace_n;
void orc_trace_reset(void) { trace_n = 0; }
void orc_trace_block(int kind, int in_len, int ntokens, long long bit_start) {
  if (trace_n < TRACE_MAX) {
    trace_blocks[trace_n].kind = kind;
    trace_blocks[trace_n].in_len = in_len;
    trace_blocks[trace_n].ntokens = ntokens;
    trace_blocks[trace_n].bit_start = bit_start;
    trace_n++;
  }
}
int orc_last_blocks(orc_block_info *dst, int max) {
  int n = trace_n < max ? trace_n : max;
  if (dst) memcpy(dst, trace_blocks, (size_t)n * sizeof(*dst));
  return trace_n;
}

/* deflate.mbt:46-78 (hash-chain fields omitted: only used by new_dict) */
typedef struct {
  orc_bit_writer w;
  orc_deflate_fast *best_speed;
  uint8_t window[ORC_MAX_STORE_BLOCK_SIZE];
  int window_end;
  int sync;
  uint32_t *tokens; /* capacity max_store_block_size + 1 */
  int ntokens;
  int err;
} compressor;

/* deflate.mbt:81-100 */
static int compressor_init(compressor *d, orc_sink *sink, int compat) {
  orc_bw_init(&d->w, sink, compat);
  d->best_speed = orc_df_new(compat);
  d->tokens = (uint32_t *)malloc(sizeof(uint32_t) * (ORC_MAX_STORE_BLOCK_SIZE + 1));
  d->window_end = 0;
  d->sync = 0;
  d->ntokens = 0;
  d->err = 0;
  return (d->best_speed && d->tokens) ? 0 : ORC_E_INTERNAL;
}
static void compressor_free(compressor *d) {
  orc_df_free(d->best_speed);
  free(d->tokens);
}

/* deflate.mbt:186-196 */
static int write_stored_block(compressor *d, const uint8_t *buf, int n) {
  orc_trace_block(0, n, 0, orc_bw_bitpos(&d->w));
  orc_bw_write_stored_header(&d->w, n, 0);
  if (d->w.err) return d->w.err;
  orc_bw_write_bytes(&d->w, buf, n);
  return d->w.err;
}

/* deflate.mbt:222-229 */
static int fill_store(compressor *d, const uint8_t *b, int blen) {
  int n = ORC_MAX_STORE_BLOCK_SIZE - d->window_end;
  if (n > blen) n = blen;
  for (int i = 0; i < n; i++) d->window[d->window_end + i] = b[i];
  d->window_end += n;
  return n;
}

/* deflate.mbt:236-277 */
static void enc_speed(compressor *d) {
  if (d->window_end < ORC_MAX_STORE_BLOCK_SIZE) {
    if (!d->sync) return;
    if (d->window_end < 128) { /* small sizes */
      if (d->window_end == 0) return;
      if (d->window_end <= 16) {
        d->err = write_stored_block(d, d->window, d->window_end);
      } else {
        long long bp = orc_bw_bitpos(&d->w);
        int kind = orc_bw_write_block_huff(&d->w, 0, d->window, d->window_end);
        orc_trace_block(kind, d->window_end, 0, bp);
        d->err = d->w.err;
      }
      d->window_end = 0;
      orc_df_reset(d->best_speed);
      return;
    }
  }
  /* encode the block: fresh token array each time (:259-262) */
  d->ntokens = orc_df_encode(d->best_speed, d->tokens, 0, d->window, d->window_end);

  long long bp = orc_bw_bitpos(&d->w);
  int kind;
  int ntok = d->ntokens;
  if (d->ntokens > d->window_end - (d->window_end >> 4)) { /* :266 */
    kind = orc_bw_write_block_huff(&d->w, 0, d->window, d->window_end);
  } else {
    kind = orc_bw_write_block_dynamic(&d->w, d->tokens, d->ntokens, 0, d->window,
                                      d->window_end);
  }
  orc_trace_block(kind, d->window_end, ntok, bp);
  d->err = d->w.err;
  d->window_end = 0;
}

/* deflate.mbt:280-294 */
static int compressor_write(compressor *d, const uint8_t *b, size_t blen, size_t *nw) {
  *nw = 0;
  if (d->err) return d->err;
  size_t n = blen;
  while (blen > 0) {
    enc_speed(d);
    int chunk = blen > (size_t)ORC_MAX_STORE_BLOCK_SIZE ? ORC_MAX_STORE_BLOCK_SIZE
                                                        : (int)blen;
    int k = fill_store(d, b, chunk);
    b += k;
    blen -= (size_t)k;
    if (d->err) return d->err;
  }
  *nw = n;
  return 0;
}

/* deflate.mbt:157-183 */
static int compressor_close(compressor *d) {
  if (d->err == ORC_E_CLOSED) return 0;
  if (d->err) return d->err;
  d->sync = 1;
  enc_speed(d);
  if (d->err) return d->err;
  orc_trace_block(0, 0, 0, orc_bw_bitpos(&d->w));
  orc_bw_write_stored_header(&d->w, 0, 1);
  if (d->w.err) return d->w.err;
  orc_bw_flush(&d->w);
  if (d->w.err) return d->w.err;
  d->err = ORC_E_CLOSED;
  return 0;
}

/* writer.mbt:10,45,53: Writer::new + write* + close over an in-memory sink */
int orc_deflate_stream(const uint8_t *in, size_t n, const size_t *sizes, int nwrites,
                       uint8_t *out, size_t cap, size_t *out_len, int compat) {
  orc_sink sink = {out, 0, cap, 0};
  compressor *d = (compressor *)malloc(sizeof(compressor));
  if (!d) return ORC_E_INTERNAL;
  orc_trace_reset();
  int rc = compressor_init(d, &sink, compat);
  if (rc == 0) {
    size_t one = n;
    if (!sizes) {
      sizes = &one;
      nwrites = 1;
    }
    size_t pos = 0;
    for (int i = 0; i < nwrites && rc == 0; i++) {
      size_t nw;
      if (pos + sizes[i] > n) {
        rc = ORC_E_INTERNAL;
        break;
      }
      rc = compressor_write(d, in + pos, sizes[i], &nw);
      pos += sizes[i];
    }
    if (rc == 0) rc = compressor_close(d);
  }
  *out_len = sink.len;
  compressor_free(d);
  free(d);
  return rc;
}

/* SURVEY 8(f)-3 -- no counterpart in the reference, whose Writer makes one stream per Writer:
 * the n streams of a batch as ONE legal DEFLATE stream.  Stream i is compressed exactly as a fresh
 * Writer would (fresh DeflateFast: deflate.mbt:92, deflate-fast.mbt:111-117; a sync'ed enc_speed
 * at its end as in close, deflate.mbt:163-166) but all streams share one HuffmanBitWriter, so a
 * block starts at the bit where the previous stream's last block ended and a stored block is
 * padded (write_stored_header -> flush, huffman-bit-writer.mbt:474-487,139-158) relative to the
 * spliced stream; the closing empty stored block with BFINAL=1 (deflate.mbt:171-176) is written
 * once, after the last stream.  Every other block has BFINAL=0 (deflate.mbt:251,267,269), which
 * is what makes the concatenation legal.  bit_off (n+1 entries, optional) = bit position of each
 * stream's first block ("stream index"). */
int orc_deflate_spliced(const uint8_t *in, const uint64_t *in_off, uint32_t n_streams, uint8_t *out,
                        size_t cap, size_t *out_len, uint64_t *bit_off, int compat) {
  orc_sink sink = {out, 0, cap, 0};
  compressor *d = (compressor *)malloc(sizeof(compressor));
  if (!d) return ORC_E_INTERNAL;
  orc_trace_reset();
  int rc = compressor_init(d, &sink, compat);
  for (uint32_t i = 0; i < n_streams && rc == 0; i++) {
    if (bit_off) bit_off[i] = (uint64_t)orc_bw_bitpos(&d->w);
    if (i > 0) { /* fresh Writer state for this stream; the bit writer goes on */
      orc_df_free(d->best_speed);
      d->best_speed = orc_df_new(compat);
      if (!d->best_speed) rc = ORC_E_INTERNAL;
    }
    size_t nw;
    if (rc == 0) rc = compressor_write(d, in + in_off[i], (size_t)(in_off[i + 1] - in_off[i]), &nw);
    if (rc == 0) {
      d->sync = 1;
      enc_speed(d);
      d->sync = 0;
      rc = d->err;
    }
  }
  if (rc == 0) {
    if (bit_off) bit_off[n_streams] = (uint64_t)orc_bw_bitpos(&d->w);
    rc = compressor_close(d);
  }
  *out_len = sink.len;
  compressor_free(d);
  free(d);
  return rc;
}

/* Upper bound on the stream size for n input bytes: every window could be
 * emitted Huffman-coded at <= 15 bits/byte plus a <= 300-byte header; stored
 * blocks cost 5 bytes per 65535; plus the final empty stored block. */
size_t orc_deflate_bound(size_t n) {
  size_t windows = n / ORC_MAX_STORE_BLOCK_SIZE + 1;
  return n * 2 + windows * 320 + 16;
}

typedef struct {
  const uint8_t *in;
  const uint64_t *in_off;
  uint8_t *out;
  const uint64_t *out_off;
  uint64_t *out_len;
  uint32_t lo, hi;
  int compat;
  int rc;
} batch_job;

static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  for (uint32_t i = j->lo; i < j->hi; i++) {
    size_t olen = 0;
    int rc = orc_deflate_stream(j->in + j->in_off[i], (size_t)(j->in_off[i + 1] - j->in_off[i]),
                                NULL, 0, j->out + j->out_off[i],
                                (size_t)(j->out_off[i + 1] - j->out_off[i]), &olen, j->compat);
    j->out_len[i] = olen;
    if (rc && !j->rc) j->rc = rc;
  }
  return NULL;
}

int orc_deflate_batch(const uint8_t *in, const uint64_t *in_off, uint32_t n_streams,
                      uint8_t *out, const uint64_t *out_off, uint64_t *out_len, int compat,
                      int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n_streams && n_streams > 0) nthreads = (int)n_streams;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  batch_job *jobs = (batch_job *)calloc((size_t)nthreads, sizeof(batch_job));
  for (int t = 0; t < nthreads; t++) {
    jobs[t].in = in;
    jobs[t].in_off = in_off;
    jobs[t].out = out;
    jobs[t].out_off = out_off;
    jobs[t].out_len = out_len;
    jobs[t].lo = (uint32_t)((uint64_t)n_streams * (uint64_t)t / (uint64_t)nthreads);
    jobs[t].hi = (uint32_t)((uint64_t)n_streams * (uint64_t)(t + 1) / (uint64_t)nthreads);
    jobs[t].compat = compat;
    if (nthreads > 1)
      pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    else
      batch_worker(&jobs[t]);
  }
  int rc = 0;
  for (int t = 0; t < nthreads; t++) {
    if (nthreads > 1) pthread_join(th[t], NULL);
    if (jobs[t].rc && !rc) rc = jobs[t].rc;
  }
  free(th);
  free(jobs);
  return rc;
}
