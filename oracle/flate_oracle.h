/*
 * flate_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the deflate-fast encode path and the inflater of
 * gmlewis/moonbit-flate (reference mounted at /root/reference; MoonBit, cannot be
 * compiled here: no moon/moonc/go toolchain).  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (moonbit-flate_amd/)
 * never links or calls it.
 *
 * Parity pinning: checked against every known-answer the reference's own tests
 * hold for this path (token.mbt:95, bits.mbt:24, huffman-code.mbt:289,
 * deflate_test.mbt:12-35 [28 B -> 38 B], deflate-fast_test.mbt:14-100 [96 round
 * trips]) plus independent inflaters (zlib).  The reference has no golden
 * compressed bytes, so the compressed bit stream itself is pinned only by
 * faithful restatement ("bitstream parity pinned by restatement + reference KATs").
 */
#ifndef FLATE_ORACLE_H
#define FLATE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* compat modes */
#define ORC_COMPAT_MOONBIT 0 /* what /root/reference does (judged parity)      */
#define ORC_COMPAT_GO 1      /* Go 1.23.1 compress/flate semantics (D1+D2 fixed) */

/* error codes */
#define ORC_OK 0
#define ORC_E_OUT_TOO_SMALL (-1)
#define ORC_E_CORRUPT (-2)        /* corrupt_input_error, offset in *err_off */
#define ORC_E_UNEXPECTED_EOF (-3) /* @io.err_unexpected_eof                  */
#define ORC_E_INTERNAL (-4)
#define ORC_E_CLOSED (-5) /* writer_closed_error */

/* ---- constants (deflate-fast.mbt:12-55,89-92) ---- */
#define ORC_TABLE_BITS 14
#define ORC_TABLE_SIZE (1 << ORC_TABLE_BITS)
#define ORC_TABLE_MASK (ORC_TABLE_SIZE - 1)
#define ORC_TABLE_SHIFT (32 - ORC_TABLE_BITS)
#define ORC_BASE_MATCH_LENGTH 3
#define ORC_MAX_MATCH_LENGTH 258
#define ORC_BASE_MATCH_OFFSET 1
#define ORC_MAX_MATCH_OFFSET (1 << 15)
#define ORC_MAX_STORE_BLOCK_SIZE 65535
#define ORC_BUFFER_RESET (2147483647 - ORC_MAX_STORE_BLOCK_SIZE * 2)
#define ORC_INPUT_MARGIN (16 - 1)
#define ORC_MIN_NON_LITERAL_BLOCK_SIZE (1 + 1 + ORC_INPUT_MARGIN)

#define ORC_MAX_NUM_LIT 286
#define ORC_OFFSET_CODE_COUNT 30
#define ORC_CODEGEN_CODE_COUNT 19

/* ---- token.mbt ---- */
uint32_t orc_literal_token(uint32_t literal);
uint32_t orc_match_token(uint32_t xlength, uint32_t xoffset);
uint32_t orc_token_literal(uint32_t t);
uint32_t orc_token_offset(uint32_t t);
uint32_t orc_token_length(uint32_t t);
int orc_length_code(uint32_t len);
int orc_offset_code(uint32_t off);

/* ---- bits.mbt / huffman-code.mbt ---- */
uint32_t orc_reverse16(uint32_t x);
uint32_t orc_reverse_bits(uint32_t number, int bit_length);
uint32_t orc_hash(uint32_t u);

/* Build a length-limited canonical Huffman code exactly as
 * HuffmanEncoder::generate (huffman-code.mbt:295-343).  codes/lens have n
 * entries and are treated as a fresh encoder (all zero) before the call. */
void orc_huffman_generate(const int32_t *freq, int n, int max_bits,
                          uint32_t *codes, uint32_t *lens);

/* ---- deflate-fast.mbt: stateful match finder ---- */
typedef struct orc_deflate_fast orc_deflate_fast;
orc_deflate_fast *orc_df_new(int compat);
void orc_df_free(orc_deflate_fast *e);
/* DeflateFast::encode: appends tokens for src[0..n) to dst, returns the new
 * token count.  dst must have room for ntok + n entries. */
int orc_df_encode(orc_deflate_fast *e, uint32_t *dst, int ntok,
                  const uint8_t *src, int n);
void orc_df_reset(orc_deflate_fast *e);
int32_t orc_df_cur(const orc_deflate_fast *e);
/* Test hook, not in the reference: lowers buffer_reset (deflate-fast.mbt:55, reached after
 * 32 766 windows of one Writer) for every encoder of this process; 0 restores the real value. */
void orc_test_set_buffer_reset(int32_t v);

/* ---- whole-stream encode: Writer::new + write(...)* + close ---- */
/* sizes[0..nwrites) are the byte counts of successive Writer::write calls
 * (they must sum to n); sizes == NULL means one write of n bytes.  Output is
 * written to out (cap bytes); *out_len receives the stream length. */
int orc_deflate_stream(const uint8_t *in, size_t n, const size_t *sizes,
                       int nwrites, uint8_t *out, size_t cap, size_t *out_len,
                       int compat);

/* Per-block trace of the last orc_deflate_stream call on this thread (test aid):
 * kind 0 = stored, 1 = huff-only (write_block_huff), 2 = dynamic. */
typedef struct {
  int kind;
  int in_len;
  int ntokens;
  long long bit_start; /* bit offset of the block header in the stream */
} orc_block_info;
int orc_last_blocks(orc_block_info *dst, int max);

/* Batch of independent streams (fresh Writer each), static partition over
 * nthreads pthreads.  in_off/out_off have n_streams+1 entries; out_off[i] is
 * an INPUT (slot start), out_len[i] the produced length.  Returns ORC_OK or
 * the first error.  Used by bench.py's cpu_baseline leg. */
int orc_deflate_batch(const uint8_t *in, const uint64_t *in_off,
                      uint32_t n_streams, uint8_t *out, const uint64_t *out_off,
                      uint64_t *out_len, int compat, int nthreads);

size_t orc_deflate_bound(size_t n);

/* ---- inflate.mbt + dict-decoder.mbt: whole-stream decode ---- */
/* Decodes one DEFLATE stream (through the BFINAL block).  *consumed = input
 * bytes read (roffset), *err_off = offset reported by corrupt_input_error. */
int orc_inflate_stream(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                       size_t *out_len, size_t *consumed, long long *err_off);
/* &Reader::new_dict (inflate.mbt:315-317): the same with a preset dictionary, whose last 32768 bytes
 * are history that has already been read (DictDecoder::new, dict-decoder.mbt:40-60). */
int orc_inflate_stream_dict(const uint8_t *in, size_t n, const uint8_t *dict, size_t dict_len, uint8_t *out,
                            size_t cap, size_t *out_len, size_t *consumed, long long *err_off);
/* The same for n independent streams over nthreads host threads; stream i is
 * in[in_off[i]..in_off[i+1]) -> out[out_off[i]..out_off[i+1]) (capacity).  Returns the first
 * non-zero status. */
int orc_inflate_batch(const uint8_t *in, const uint64_t *in_off, uint32_t n_streams, uint8_t *out,
                      const uint64_t *out_off, uint64_t *out_len, int32_t *status, int nthreads);

/* -- splice: the n streams of a batch compressed into ONE legal DEFLATE stream
 * (compressor.c); out needs sum of orc_deflate_bound() bytes. */
int orc_deflate_spliced(const uint8_t *in, const uint64_t *in_off, uint32_t n_streams, uint8_t *out,
                        size_t cap, size_t *out_len, uint64_t *bit_off, int compat);

#ifdef __cplusplus
}
#endif
/* -- checksum.c: the checksums and container formats around a raw stream (RFC 1950 / RFC 1952) -- */
#define ORC_FRAME_RAW 0
#define ORC_FRAME_ZLIB 1
#define ORC_FRAME_GZIP 2
uint32_t orc_adler32(const uint8_t *p, size_t n);
uint32_t orc_crc32(const uint8_t *p, size_t n);
size_t orc_frame_overhead(int kind);
size_t orc_frame(int kind, const uint8_t *raw, size_t raw_len, const uint8_t *data, size_t n, uint8_t *out);

#endif
