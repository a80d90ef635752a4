/*
 * deflate_fast.c -- TEST INFRASTRUCTURE (oracle).  CPU restatement of
 * /root/reference/deflate-fast.mbt and token.mbt.  See flate_oracle.h.
 */
#include "flate_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ token.mbt */

/* token.mbt:13-24 */
#define LENGTH_SHIFT 22
#define OFFSET_MASK ((1u << LENGTH_SHIFT) - 1)
#define LITERAL_TYPE (0u << 30)
#define MATCH_TYPE (1u << 30)

/* token.mbt:30-44: length code for (length - 3).  Generated from RFC 1951
 * 3.2.5 rather than typed in: codes 0..7 are single lengths, then groups of 4
 * codes share 1,2,3,4,5 extra bits, and length 258 (index 255) is code 28. */
static uint8_t length_codes[256];
/* token.mbt:47-61: offset code for small (offset - 1) values < 256. */
static uint8_t offset_codes[256];
static int tables_ready;

static void init_tables(void) {
  if (tables_ready) return;
  int code = 0, idx = 0;
  for (; code < 8; code++) length_codes[idx++] = (uint8_t)code;
  for (int extra = 1; extra <= 5; extra++)
    for (int k = 0; k < 4; k++, code++)
      for (int j = 0; j < (1 << extra); j++)
        if (idx < 256) length_codes[idx++] = (uint8_t)code;
  length_codes[255] = 28;
  /* offsets: codes 0..3 single, then pairs of codes with 1,2,.. extra bits */
  idx = 0;
  for (code = 0; code < 4; code++) offset_codes[idx++] = (uint8_t)code;
  for (int extra = 1; idx < 256; extra++)
    for (int k = 0; k < 2 && idx < 256; k++, code++)
      for (int j = 0; j < (1 << extra) && idx < 256; j++)
        offset_codes[idx++] = (uint8_t)code;
  tables_ready = 1;
}

/* token.mbt:69 */
uint32_t orc_literal_token(uint32_t literal) { return LITERAL_TYPE + literal; }
/* token.mbt:76 */
uint32_t orc_match_token(uint32_t xlength, uint32_t xoffset) {
  return MATCH_TYPE + (xlength << LENGTH_SHIFT) + xoffset;
}
/* token.mbt:83 */
uint32_t orc_token_literal(uint32_t t) { return t - LITERAL_TYPE; }
/* token.mbt:90 */
uint32_t orc_token_offset(uint32_t t) { return t & OFFSET_MASK; }
/* token.mbt:102 */
uint32_t orc_token_length(uint32_t t) { return (t - MATCH_TYPE) >> LENGTH_SHIFT; }
/* token.mbt:107 */
int orc_length_code(uint32_t len) {
  init_tables();
  return length_codes[len];
}
/* token.mbt:112-123 */
int orc_offset_code(uint32_t off) {
  init_tables();
  if (off < 256) return offset_codes[off];
  if ((off >> 7) < 256) return offset_codes[off >> 7] + 14;
  return offset_codes[off >> 14] + 28;
}

/* ----------------------------------------------------------- deflate-fast.mbt */

/* deflate-fast.mbt:58 */
static uint32_t load32(const uint8_t *b, int i) {
  return (uint32_t)b[i] | ((uint32_t)b[i + 1] << 8) | ((uint32_t)b[i + 2] << 16) |
         ((uint32_t)b[i + 3] << 24);
}
/* deflate-fast.mbt:66 */
static uint64_t load64(const uint8_t *b, int i) {
  return (uint64_t)load32(b, i) | ((uint64_t)load32(b, i + 4) << 32);
}
/* deflate-fast.mbt:78 */
uint32_t orc_hash(uint32_t u) { return (u * 0x1e35a7bdu) >> ORC_TABLE_SHIFT; }

/* deflate-fast.mbt:95 */
typedef struct {
  uint32_t val;
  int32_t offset;
} table_entry;

/* deflate-fast.mbt:104 */
struct orc_deflate_fast {
  table_entry table[ORC_TABLE_SIZE];
  uint8_t prev[ORC_MAX_STORE_BLOCK_SIZE]; /* previous block */
  int prev_len;                           /* "zero length if unknown" */
  int32_t cur;
  int compat;
};

/* deflate-fast.mbt:111-117 */
orc_deflate_fast *orc_df_new(int compat) {
  init_tables();
  orc_deflate_fast *e = (orc_deflate_fast *)calloc(1, sizeof(*e));
  if (!e) return NULL;
  e->prev_len = 0;
  e->cur = ORC_MAX_STORE_BLOCK_SIZE;
  e->compat = compat;
  return e;
}
void orc_df_free(orc_deflate_fast *e) { free(e); }
int32_t orc_df_cur(const orc_deflate_fast *e) { return e->cur; }

/* Test hook (not in the reference): buffer_reset (deflate-fast.mbt:55) is reached after 32 766
 * windows of one Writer, about 2.1 GB; the tests lower it so that the shift_offsets branches run
 * within a few windows.  0 = the reference's value.  Process-wide; set before the encoders run. */
static volatile int32_t test_buffer_reset = 0;
void orc_test_set_buffer_reset(int32_t v) { test_buffer_reset = v; }
static int32_t buffer_reset(void) { return test_buffer_reset ? test_buffer_reset : ORC_BUFFER_RESET; }

/* deflate-fast.mbt:366-389 */
static void shift_offsets(orc_deflate_fast *e) {
  if (e->prev_len == 0) {
    for (int i = 0; i < ORC_TABLE_SIZE; i++) {
      e->table[i].val = 0;
      e->table[i].offset = 0;
    }
    e->cur = ORC_MAX_MATCH_OFFSET + 1;
    return;
  }
  for (int i = 0; i < ORC_TABLE_SIZE; i++) {
    int32_t v = e->table[i].offset - e->cur + ORC_MAX_MATCH_OFFSET + 1;
    if (v < 0) v = 0;
    e->table[i].offset = v;
  }
  e->cur = ORC_MAX_MATCH_OFFSET + 1;
}

/* deflate-fast.mbt:348-358 */
void orc_df_reset(orc_deflate_fast *e) {
  e->prev_len = 0;
  e->cur += ORC_MAX_MATCH_OFFSET;
  if (e->cur >= buffer_reset()) shift_offsets(e);
}

/* deflate-fast.mbt:273-279 */
static int emit_literal(uint32_t *dst, int ntok, const uint8_t *lit, int n) {
  for (int i = 0; i < n; i++) dst[ntok++] = orc_literal_token(lit[i]);
  return ntok;
}

/* deflate-fast.mbt:286-342 */
static int match_len(const orc_deflate_fast *e, int s, int t, const uint8_t *src,
                     int src_len) {
  int s1 = s + ORC_MAX_MATCH_LENGTH - 4;
  if (s1 > src_len) s1 = src_len;

  if (t >= 0) { /* inside the current block (:298-307) */
    int a_length = s1 - s;
    for (int i = 0; i < a_length; i++)
      if (src[s + i] != src[t + i]) return i;
    return a_length;
  }

  /* match in the previous block (:310-313).  In the reference prev is always
   * empty, so tp = t < 0 and this returns 0. */
  int tp = e->prev_len + t;
  if (tp < 0) return 0;

  int a_length = s1 - s;
  int b_length = e->prev_len - tp;
  if (b_length > a_length) b_length = a_length;
  for (int i = 0; i < b_length; i++)
    if (src[s + i] != e->prev[tp + i]) return i;

  int n = b_length;
  if (s + n == s1) return n;

  a_length = s1 - (s + n);
  for (int i = 0; i < a_length; i++)
    if (src[s + n + i] != src[i]) return i + n;
  return a_length + n;
}

/* emit_remainder closure, deflate-fast.mbt:152-159.
 * MoonBit: slice_copy(self.prev, src) copies min(len(prev)=0, len(src)) = 0
 * bytes (dict-decoder.mbt:188-194) and leaves prev empty; Go does
 * e.prev = e.prev[:len(src)]; copy(e.prev, src). */
static int emit_remainder(orc_deflate_fast *e, uint32_t *dst, int ntok,
                          const uint8_t *src, int n, int next_emit) {
  if (next_emit < n) ntok = emit_literal(dst, ntok, src + next_emit, n - next_emit);
  e->cur += n;
  if (e->compat == ORC_COMPAT_GO) {
    memcpy(e->prev, src, (size_t)n);
    e->prev_len = n;
  }
  return ntok;
}

/* deflate-fast.mbt:123-270 */
int orc_df_encode(orc_deflate_fast *e, uint32_t *dst, int ntok, const uint8_t *src,
                  int n) {
  if (e->cur >= buffer_reset()) shift_offsets(e); /* :130 */

  if (n < ORC_MIN_NON_LITERAL_BLOCK_SIZE) { /* :136-140 */
    e->cur += ORC_MAX_STORE_BLOCK_SIZE;
    e->prev_len = 0;
    return emit_literal(dst, ntok, src, n);
  }

  int s_limit = n - ORC_INPUT_MARGIN; /* :145 */
  int next_emit = 0;
  int s = 0;
  uint32_t cv = load32(src, s);
  int next_hash = (int)orc_hash(cv);

  for (;;) {
    int skip = 32; /* :178 */
    int next_s = s;
    table_entry candidate = {0, 0};
    for (;;) { /* :183-202 */
      s = next_s;
      int bytes_between_hash_lookups = skip >> 5;
      next_s = s + bytes_between_hash_lookups;
      skip += bytes_between_hash_lookups;
      if (next_s > s_limit) return emit_remainder(e, dst, ntok, src, n, next_emit);
      candidate = e->table[next_hash & ORC_TABLE_MASK];
      uint32_t now = load32(src, next_s);
      e->table[next_hash & ORC_TABLE_MASK].offset = s + e->cur;
      e->table[next_hash & ORC_TABLE_MASK].val = cv;
      next_hash = (int)orc_hash(now);
      int offset = s - (candidate.offset - e->cur);
      if (offset > ORC_MAX_MATCH_OFFSET || cv != candidate.val) {
        cv = now;
        continue;
      }
      break;
    }

    /* :207 */
    ntok = emit_literal(dst, ntok, src + next_emit, s - next_emit);

    for (;;) { /* :217-266 */
      s += 4;
      int t = candidate.offset - e->cur + 4;
      int l = match_len(e, s, t, src, n);
      dst[ntok++] = orc_match_token((uint32_t)(l + 4 - ORC_BASE_MATCH_LENGTH),
                                    (uint32_t)(s - t - ORC_BASE_MATCH_OFFSET));
      s += l;
      next_emit = s;
      if (s >= s_limit) return emit_remainder(e, dst, ntok, src, n, next_emit);

      uint64_t x = load64(src, s - 1); /* :246 */
      int prev_hash = (int)orc_hash((uint32_t)x);
      e->table[prev_hash & ORC_TABLE_MASK].offset = e->cur + s - 1;
      e->table[prev_hash & ORC_TABLE_MASK].val = (uint32_t)x;
      x >>= 8;
      int curr_hash = (int)orc_hash((uint32_t)x);
      candidate = e->table[curr_hash & ORC_TABLE_MASK];
      e->table[curr_hash & ORC_TABLE_MASK].offset = e->cur + s;
      e->table[curr_hash & ORC_TABLE_MASK].val = (uint32_t)x;

      int offset = s - (candidate.offset - e->cur);
      if (offset > ORC_MAX_MATCH_OFFSET || (uint32_t)x != candidate.val) {
        cv = (uint32_t)(x >> 8);
        next_hash = (int)orc_hash(cv);
        s += 1;
        break;
      }
    }
  }
}
