/* orc_internal.h -- TEST INFRASTRUCTURE (oracle); shared private types. */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "flate_oracle.h"

/* huffman-code.mbt:37-40 */
typedef struct {
  uint32_t code;
  uint32_t len;
} orc_hcode;

/* huffman-code.mbt:29-32 */
typedef struct {
  uint32_t literal;
  int32_t freq;
} orc_literal_node;

/* huffman-code.mbt:9-13 */
typedef struct {
  int size;
  orc_hcode codes[ORC_MAX_NUM_LIT];
  orc_literal_node freqcache[ORC_MAX_NUM_LIT + 1];
  int bit_count[17];
} orc_huffman_encoder;

void orc_henc_init(orc_huffman_encoder *h, int size);
void orc_henc_generate(orc_huffman_encoder *h, const int32_t *freq, int nfreq,
                       int max_bits);
int orc_henc_bit_length(const orc_huffman_encoder *h, const int32_t *freq, int n);

/* Byte sink standing in for the &@io.Writer trait object
 * (huffman-bit-writer.mbt:92,165).  Fixed capacity; overflow sets err. */
typedef struct {
  uint8_t *p;
  size_t len, cap;
  int err;
} orc_sink;

/* huffman-bit-writer.mbt:88-109 */
typedef struct {
  orc_sink *writer;
  uint64_t bits;
  uint32_t nbits;
  uint8_t bytes[248];
  int32_t codegen_freq[ORC_CODEGEN_CODE_COUNT];
  int nbytes;
  int32_t literal_freq[ORC_MAX_NUM_LIT];
  int32_t offset_freq[ORC_OFFSET_CODE_COUNT];
  uint8_t codegen[ORC_MAX_NUM_LIT + ORC_OFFSET_CODE_COUNT + 1];
  orc_huffman_encoder literal_encoding;
  orc_huffman_encoder offset_encoding;
  orc_huffman_encoder codegen_encoding;
  int err;
  int compat;
  long long bits_out; /* total bits handed to the stream so far (trace aid) */
} orc_bit_writer;

void orc_bw_init(orc_bit_writer *w, orc_sink *sink, int compat);
void orc_bw_flush(orc_bit_writer *w);
void orc_bw_write_stored_header(orc_bit_writer *w, int length, int is_eof);
void orc_bw_write_bytes(orc_bit_writer *w, const uint8_t *b, int n);
/* tokens must have room for one more entry (EOB is pushed, :507). Returns the
 * block kind actually written: 0 stored, 2 dynamic. */
int orc_bw_write_block_dynamic(orc_bit_writer *w, uint32_t *tokens, int ntok, int eof,
                               const uint8_t *input, int input_len);
/* Returns 0 stored, 1 huffman-only. */
int orc_bw_write_block_huff(orc_bit_writer *w, int eof, const uint8_t *input,
                            int input_len);
long long orc_bw_bitpos(const orc_bit_writer *w);

void orc_trace_reset(void);
void orc_trace_block(int kind, int in_len, int ntokens, long long bit_start);

#endif
