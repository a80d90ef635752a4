/*
 * bit_writer.c -- TEST INFRASTRUCTURE (oracle).  CPU restatement of
 * /root/reference/huffman-bit-writer.mbt.  See flate_oracle.h.
 */
#include "orc_internal.h"

#include <string.h>

/* huffman-bit-writer.mbt:11-44 */
#define END_BLOCK_MARKER 256
#define LENGTH_CODES_START 257
#define BAD_CODE 0xff
#define BUFFER_FLUSH_SIZE 240

/* huffman-bit-writer.mbt:49-54 */
static const int length_extra_bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                          2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
/* huffman-bit-writer.mbt:59-62 */
static const uint32_t length_base[29] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  10,
                                         12, 14, 16, 20, 24, 28, 32, 40, 48, 56,
                                         64, 80, 96, 112, 128, 160, 192, 224, 255};
/* huffman-bit-writer.mbt:67-70 */
static const int offset_extra_bits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3,  3,  4,  4,  5,  5,  6,
                                          6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
/* huffman-bit-writer.mbt:73-78 */
static const uint32_t offset_base[30] = {
    0x000000, 0x000001, 0x000002, 0x000003, 0x000004, 0x000006, 0x000008, 0x00000c,
    0x000010, 0x000018, 0x000020, 0x000030, 0x000040, 0x000060, 0x000080, 0x0000c0,
    0x000100, 0x000180, 0x000200, 0x000300, 0x000400, 0x000600, 0x000800, 0x000c00,
    0x001000, 0x001800, 0x002000, 0x003000, 0x004000, 0x006000};
/* huffman-bit-writer.mbt:83-85 */
static const int codegen_order[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                      11, 4,  12, 3, 13, 2, 14, 1, 15};

/* huffman-code.mbt:691-726: literal-only blocks use a fixed offset encoder
 * whose code 0 has length 1 and every other code length 0. */
static orc_huffman_encoder huff_offset;
static int huff_offset_ready;
static const orc_huffman_encoder *get_huff_offset(void) {
  if (!huff_offset_ready) {
    orc_henc_init(&huff_offset, ORC_OFFSET_CODE_COUNT);
    huff_offset.codes[0].code = 0;
    huff_offset.codes[0].len = 1;
    huff_offset_ready = 1;
  }
  return &huff_offset;
}

/* huffman-bit-writer.mbt:112-136 */
void orc_bw_init(orc_bit_writer *w, orc_sink *sink, int compat) {
  memset(w, 0, sizeof(*w));
  w->writer = sink;
  w->compat = compat;
  orc_henc_init(&w->literal_encoding, ORC_MAX_NUM_LIT);
  orc_henc_init(&w->codegen_encoding, ORC_CODEGEN_CODE_COUNT);
  orc_henc_init(&w->offset_encoding, ORC_OFFSET_CODE_COUNT);
  get_huff_offset();
}

long long orc_bw_bitpos(const orc_bit_writer *w) {
  return (long long)w->writer->len * 8 + (long long)w->nbytes * 8 + w->nbits;
}

/* huffman-bit-writer.mbt:161-167 (sticky error) */
static void bw_write(orc_bit_writer *w, const uint8_t *b, int n) {
  if (w->err) return;
  orc_sink *s = w->writer;
  if (s->len + (size_t)n > s->cap) {
    s->err = ORC_E_OUT_TOO_SMALL;
    w->err = ORC_E_OUT_TOO_SMALL;
    return;
  }
  memcpy(s->p + s->len, b, (size_t)n);
  s->len += (size_t)n;
}

/* huffman-bit-writer.mbt:139-158 */
void orc_bw_flush(orc_bit_writer *w) {
  if (w->err) {
    w->nbits = 0;
    return;
  }
  int n = w->nbytes;
  while (w->nbits != 0) {
    w->bytes[n] = (uint8_t)w->bits;
    w->bits >>= 8;
    if (w->nbits > 8)
      w->nbits -= 8;
    else
      w->nbits = 0;
    n++;
  }
  w->bits = 0;
  bw_write(w, w->bytes, n);
  w->nbytes = 0;
}

/* the 6-byte spill shared by write_bits/write_code/write_tokens
 * (huffman-bit-writer.mbt:181-198, 394-411, 620-639) */
static void spill48(orc_bit_writer *w) {
  uint64_t bits = w->bits;
  w->bits >>= 48;
  w->nbits -= 48;
  int n = w->nbytes;
  for (int k = 0; k < 6; k++) w->bytes[n + k] = (uint8_t)(bits >> (8 * k));
  n += 6;
  if (n >= BUFFER_FLUSH_SIZE) {
    bw_write(w, w->bytes, n);
    n = 0;
  }
  w->nbytes = n;
}

/* huffman-bit-writer.mbt:170-199 */
static void write_bits(orc_bit_writer *w, int b, uint32_t nb) {
  if (w->err) return;
  w->bits |= (uint64_t)(uint32_t)b << w->nbits;
  w->nbits += nb;
  if (w->nbits >= 48) spill48(w);
}

/* huffman-bit-writer.mbt:387-412 */
static void write_code(orc_bit_writer *w, orc_hcode c) {
  if (w->err) return;
  w->bits |= (uint64_t)c.code << w->nbits;
  w->nbits += c.len;
  if (w->nbits >= 48) spill48(w);
}

/* huffman-bit-writer.mbt:202-225 */
void orc_bw_write_bytes(orc_bit_writer *w, const uint8_t *b, int nb) {
  if (w->err) return;
  int n = w->nbytes;
  if ((w->nbits & 7u) != 0) {
    w->err = ORC_E_INTERNAL;
    return;
  }
  while (w->nbits != 0) {
    w->bytes[n] = (uint8_t)w->bits;
    w->bits >>= 8;
    w->nbits -= 8;
    n++;
  }
  if (n != 0) bw_write(w, w->bytes, n);
  w->nbytes = 0;
  bw_write(w, b, nb);
}

/* huffman-bit-writer.mbt:241-330 */
static void generate_codegen(orc_bit_writer *w, int num_literals, int num_offsets,
                             const orc_huffman_encoder *lit_enc,
                             const orc_huffman_encoder *off_enc) {
  for (int i = 0; i < ORC_CODEGEN_CODE_COUNT; i++) w->codegen_freq[i] = 0;
  uint8_t *codegen = w->codegen;
  for (int i = 0; i < num_literals; i++) codegen[i] = (uint8_t)lit_enc->codes[i].len;
  for (int i = 0; i < num_offsets; i++)
    codegen[num_literals + i] = (uint8_t)off_enc->codes[i].len;
  codegen[num_literals + num_offsets] = BAD_CODE;

  uint8_t size = codegen[0];
  int count = 1;
  int out_index = 0;
  for (int in_index = 1; size != BAD_CODE; in_index++) {
    uint8_t next_size = codegen[in_index];
    if (next_size == size) {
      count++;
      continue;
    }
    if (size != 0) {
      codegen[out_index++] = size;
      w->codegen_freq[size]++;
      count--;
      while (count >= 3) {
        int n = 6;
        if (n > count) n = count;
        codegen[out_index++] = 16;
        codegen[out_index++] = (uint8_t)(n - 3);
        w->codegen_freq[16]++;
        count -= n;
      }
    } else {
      while (count >= 11) {
        int n = 138;
        if (n > count) n = count;
        codegen[out_index++] = 18;
        codegen[out_index++] = (uint8_t)(n - 11);
        w->codegen_freq[18]++;
        count -= n;
      }
      if (count >= 3) {
        codegen[out_index++] = 17;
        codegen[out_index++] = (uint8_t)(count - 3);
        w->codegen_freq[17]++;
        count = 0;
      }
    }
    count--;
    for (; count >= 0; count--) {
      codegen[out_index++] = size;
      w->codegen_freq[size]++;
    }
    size = next_size;
    count = 1;
  }
  codegen[out_index] = BAD_CODE;
}

/* huffman-bit-writer.mbt:335-360 */
static int dynamic_size(orc_bit_writer *w, const orc_huffman_encoder *lit_enc,
                        const orc_huffman_encoder *off_enc, int extra_bits,
                        int *num_codegens_out) {
  int num_codegens = ORC_CODEGEN_CODE_COUNT;
  while (num_codegens > 4 && w->codegen_freq[codegen_order[num_codegens - 1]] == 0)
    num_codegens--;
  int header = 3 + 5 + 5 + 4 + 3 * num_codegens +
               orc_henc_bit_length(&w->codegen_encoding, w->codegen_freq,
                                   ORC_CODEGEN_CODE_COUNT) +
               w->codegen_freq[16] * 2 + w->codegen_freq[17] * 3 +
               w->codegen_freq[18] * 7;
  int size = header + orc_henc_bit_length(lit_enc, w->literal_freq, ORC_MAX_NUM_LIT) +
             orc_henc_bit_length(off_enc, w->offset_freq, ORC_OFFSET_CODE_COUNT) +
             extra_bits;
  *num_codegens_out = num_codegens;
  return size;
}

/* huffman-bit-writer.mbt:375-384 */
static int stored_size(int inp_length, int *storable) {
  if (inp_length == 0) {
    *storable = 0;
    return 0;
  }
  if (inp_length <= ORC_MAX_STORE_BLOCK_SIZE) {
    *storable = 1;
    return (inp_length + 5) * 8;
  }
  *storable = 0;
  return 0;
}

/* The stored-vs-Huffman decision.  MoonBit (huffman-bit-writer.mbt:527,780):
 * ssize < (size + size) >> 4.  Go 1.23.1: ssize < size + size>>4. */
static int prefer_stored(const orc_bit_writer *w, int ssize, int storable, int size) {
  if (!storable) return 0;
  if (w->compat == ORC_COMPAT_GO) return ssize < size + (size >> 4);
  return ssize < ((size + size) >> 4);
}

/* huffman-bit-writer.mbt:421-471 */
static void write_dynamic_header(orc_bit_writer *w, int num_literals, int num_offsets,
                                 int num_codegens, int is_eof) {
  if (w->err) return;
  int first_bits = 4;
  if (is_eof) first_bits = 5;
  write_bits(w, first_bits, 3);
  write_bits(w, num_literals - 257, 5);
  write_bits(w, num_offsets - 1, 5);
  write_bits(w, num_codegens - 4, 4);

  for (int i = 0; i < num_codegens; i++) {
    int value = (int)w->codegen_encoding.codes[codegen_order[i]].len;
    write_bits(w, value, 3);
  }

  int i = 0;
  for (;;) {
    int code_word = w->codegen[i];
    i++;
    if (code_word == BAD_CODE) break;
    write_code(w, w->codegen_encoding.codes[code_word]);
    switch (code_word) {
      case 16:
        write_bits(w, w->codegen[i], 2);
        i++;
        break;
      case 17:
        write_bits(w, w->codegen[i], 3);
        i++;
        break;
      case 18:
        write_bits(w, w->codegen[i], 7);
        i++;
        break;
      default:
        break;
    }
  }
}

/* huffman-bit-writer.mbt:474-487 */
void orc_bw_write_stored_header(orc_bit_writer *w, int length, int is_eof) {
  if (w->err) return;
  int flag = is_eof ? 1 : 0;
  write_bits(w, flag, 3);
  orc_bw_flush(w);
  write_bits(w, length, 16);
  write_bits(w, (~length) & 0xffff, 16);
}

/* huffman-bit-writer.mbt:550-593 */
static void index_tokens(orc_bit_writer *w, const uint32_t *tokens, int ntok,
                         int *num_literals_out, int *num_offsets_out) {
  for (int i = 0; i < ORC_MAX_NUM_LIT; i++) w->literal_freq[i] = 0;
  for (int i = 0; i < ORC_OFFSET_CODE_COUNT; i++) w->offset_freq[i] = 0;

  for (int k = 0; k < ntok; k++) {
    uint32_t t = tokens[k];
    if (t < (1u << 30)) {
      w->literal_freq[orc_token_literal(t)]++;
      continue;
    }
    uint32_t length = orc_token_length(t);
    uint32_t offset = orc_token_offset(t);
    w->literal_freq[LENGTH_CODES_START + orc_length_code(length)]++;
    w->offset_freq[orc_offset_code(offset)]++;
  }

  int num_literals = ORC_MAX_NUM_LIT;
  while (w->literal_freq[num_literals - 1] == 0) num_literals--;
  int num_offsets = ORC_OFFSET_CODE_COUNT;
  while (num_offsets > 0 && w->offset_freq[num_offsets - 1] == 0) num_offsets--;
  if (num_offsets == 0) {
    w->offset_freq[0] = 1;
    num_offsets = 1;
  }
  orc_henc_generate(&w->literal_encoding, w->literal_freq, ORC_MAX_NUM_LIT, 15);
  orc_henc_generate(&w->offset_encoding, w->offset_freq, ORC_OFFSET_CODE_COUNT, 15);
  *num_literals_out = num_literals;
  *num_offsets_out = num_offsets;
}

/* huffman-bit-writer.mbt:596-731.  The reference keeps bits/nbits/nbytes in
 * locals and writes them back; the observable byte stream is identical. */
static void write_tokens(orc_bit_writer *w, const uint32_t *tokens, int ntok,
                         const orc_hcode *le_codes, const orc_hcode *oe_codes) {
  if (w->err) return;
  for (int k = 0; k < ntok; k++) {
    uint32_t t = tokens[k];
    if (t < (1u << 30)) {
      orc_hcode c = le_codes[orc_token_literal(t)];
      w->bits |= (uint64_t)c.code << w->nbits;
      w->nbits += c.len;
    } else {
      uint32_t length = orc_token_length(t);
      int lc = orc_length_code(length);
      orc_hcode c = le_codes[lc + LENGTH_CODES_START];
      w->bits |= (uint64_t)c.code << w->nbits;
      w->nbits += c.len;
      if (w->nbits >= 48) {
        spill48(w);
        if (w->err) return;
      }
      uint32_t extra_length_bits = (uint32_t)length_extra_bits[lc];
      if (extra_length_bits > 0) {
        int extra_length = (int)(length - length_base[lc]);
        w->bits |= (uint64_t)(uint32_t)extra_length << w->nbits;
        w->nbits += extra_length_bits;
      }
      if (w->nbits >= 48) {
        spill48(w);
        if (w->err) return;
      }
      uint32_t offset = orc_token_offset(t);
      int oc = orc_offset_code(offset);
      c = oe_codes[oc];
      w->bits |= (uint64_t)c.code << w->nbits;
      w->nbits += c.len;
      if (w->nbits >= 48) {
        spill48(w);
        if (w->err) return;
      }
      uint32_t extra_offset_bits = (uint32_t)offset_extra_bits[oc];
      if (extra_offset_bits > 0) {
        int extra_offset = (int)(offset - offset_base[oc]);
        w->bits |= (uint64_t)(uint32_t)extra_offset << w->nbits;
        w->nbits += extra_offset_bits;
      }
    }
    if (w->nbits >= 48) {
      spill48(w);
      if (w->err) return;
    }
  }
}

/* huffman-bit-writer.mbt:496-542 */
int orc_bw_write_block_dynamic(orc_bit_writer *w, uint32_t *tokens, int ntok, int eof,
                               const uint8_t *input, int input_len) {
  if (w->err) return -1;
  tokens[ntok++] = END_BLOCK_MARKER; /* :507 */
  int num_literals, num_offsets;
  index_tokens(w, tokens, ntok, &num_literals, &num_offsets);

  generate_codegen(w, num_literals, num_offsets, &w->literal_encoding,
                   &w->offset_encoding);
  orc_henc_generate(&w->codegen_encoding, w->codegen_freq, ORC_CODEGEN_CODE_COUNT, 7);
  int num_codegens;
  int size = dynamic_size(w, &w->literal_encoding, &w->offset_encoding, 0, &num_codegens);

  int storable;
  int ssize = stored_size(input_len, &storable);
  if (prefer_stored(w, ssize, storable, size)) {
    orc_bw_write_stored_header(w, input_len, eof);
    orc_bw_write_bytes(w, input, input_len);
    return 0;
  }

  write_dynamic_header(w, num_literals, num_offsets, num_codegens, eof);
  write_tokens(w, tokens, ntok, w->literal_encoding.codes, w->offset_encoding.codes);
  return 2;
}

/* huffman-bit-writer.mbt:738-824 (+ histogram :831) */
int orc_bw_write_block_huff(orc_bit_writer *w, int eof, const uint8_t *input,
                            int input_len) {
  if (w->err) return -1;
  for (int i = 0; i < ORC_MAX_NUM_LIT; i++) w->literal_freq[i] = 0;
  for (int i = 0; i < input_len; i++) w->literal_freq[input[i]]++;
  w->literal_freq[END_BLOCK_MARKER] = 1;
  int num_literals = END_BLOCK_MARKER + 1;
  w->offset_freq[0] = 1; /* NB: the other offset_freq entries keep stale values */
  int num_offsets = 1;
  orc_henc_generate(&w->literal_encoding, w->literal_freq, ORC_MAX_NUM_LIT, 15);

  const orc_huffman_encoder *ho = get_huff_offset();
  generate_codegen(w, num_literals, num_offsets, &w->literal_encoding, ho);
  orc_henc_generate(&w->codegen_encoding, w->codegen_freq, ORC_CODEGEN_CODE_COUNT, 7);
  int num_codegens;
  int size = dynamic_size(w, &w->literal_encoding, ho, 0, &num_codegens);

  int storable;
  int ssize = stored_size(input_len, &storable);
  if (prefer_stored(w, ssize, storable, size)) {
    orc_bw_write_stored_header(w, input_len, eof);
    orc_bw_write_bytes(w, input, input_len);
    return 0;
  }

  write_dynamic_header(w, num_literals, num_offsets, num_codegens, eof);
  const orc_hcode *encoding = w->literal_encoding.codes;
  for (int i = 0; i < input_len; i++) {
    orc_hcode c = encoding[input[i]];
    w->bits |= (uint64_t)c.code << w->nbits;
    w->nbits += c.len;
    if (w->nbits < 48) continue;
    spill48(w);
    if (w->err) return -1;
  }
  write_code(w, encoding[END_BLOCK_MARKER]);
  return 1;
}
