/*
 * huffman.c -- TEST INFRASTRUCTURE (oracle).  CPU restatement of
 * /root/reference/huffman-code.mbt (1-356, 690-726), bits.mbt and the sort
 * semantics of simple-quicksort.mbt.  See flate_oracle.h.
 */
#include "orc_internal.h"

#include <limits.h>
#include <string.h>

/* bits.mbt:18-22 (reverse16 via byte table :30-46; computed, not tabulated) */
static uint32_t reverse8(uint32_t x) {
  uint32_t r = 0;
  for (int i = 0; i < 8; i++)
    if (x & (1u << i)) r |= 0x80u >> i;
  return r;
}
uint32_t orc_reverse16(uint32_t x) {
  return reverse8((x & 0xff00) >> 8) | (reverse8(x & 0xff) << 8);
}
/* huffman-code.mbt:283-286 */
uint32_t orc_reverse_bits(uint32_t number, int bit_length) {
  return orc_reverse16(number << (16 - bit_length));
}

/* huffman-code.mbt:16-26 */
void orc_henc_init(orc_huffman_encoder *h, int size) {
  memset(h, 0, sizeof(*h));
  h->size = size;
}

/* huffman-code.mbt:83-91 */
int orc_henc_bit_length(const orc_huffman_encoder *h, const int32_t *freq, int n) {
  int total = 0;
  for (int i = 0; i < n; i++)
    if (freq[i] != 0) total += freq[i] * (int)h->codes[i].len;
  return total;
}

/* huffman-code.mbt:346-351 */
static int by_frequency(const orc_literal_node *a, const orc_literal_node *b) {
  if (a->freq == b->freq) return (int32_t)a->literal < (int32_t)b->literal;
  return a->freq < b->freq;
}
/* huffman-code.mbt:354-356 */
static int by_literal(const orc_literal_node *a, const orc_literal_node *b) {
  return (int32_t)a->literal < (int32_t)b->literal;
}

/* simple-quicksort.mbt:45: both comparators are total orders on distinct keys,
 * so any correct sort yields the same permutation; insertion sort (n <= 286). */
static void sort_nodes(orc_literal_node *a, int n,
                       int (*less)(const orc_literal_node *, const orc_literal_node *)) {
  for (int i = 1; i < n; i++) {
    orc_literal_node x = a[i];
    int j = i - 1;
    while (j >= 0 && less(&x, &a[j])) {
      a[j + 1] = a[j];
      j--;
    }
    a[j + 1] = x;
  }
}

#define MAX_BITS_LIMIT 16 /* huffman-code.mbt:94 */

/* huffman-code.mbt:45-62 */
typedef struct {
  int level;
  int last_freq;
  int next_char_freq;
  int next_pair_freq;
  int needed;
} level_info;

/* huffman-code.mbt:112-244.  list has n entries plus room for the sentinel.
 * Returns max_bits actually used; bit_count[1..max_bits] filled. */
static int bit_counts(orc_huffman_encoder *h, orc_literal_node *list, int n,
                      int max_bits) {
  list[n].literal = 0xffffffffu; /* max_node(), :78-80 */
  list[n].freq = INT_MAX;

  if (max_bits > n - 1) max_bits = n - 1; /* :126-129 */

  level_info levels[MAX_BITS_LIMIT + 1];
  int leaf_counts[MAX_BITS_LIMIT][MAX_BITS_LIMIT];
  memset(levels, 0, sizeof(levels));
  memset(leaf_counts, 0, sizeof(leaf_counts));

  for (int level = 1; level <= max_bits; level++) { /* :151-165 */
    levels[level].level = level;
    levels[level].last_freq = list[1].freq;
    levels[level].next_char_freq = list[2].freq;
    levels[level].next_pair_freq = list[0].freq + list[1].freq;
    levels[level].needed = 0;
    leaf_counts[level][level] = 2;
    if (level == 1) levels[level].next_pair_freq = INT_MAX;
  }

  levels[max_bits].needed = 2 * n - 4; /* :168 */

  int level = max_bits;
  for (;;) { /* :172-227 */
    level_info *l = &levels[level];
    if (l->next_pair_freq == INT_MAX && l->next_char_freq == INT_MAX) {
      l->needed = 0;
      levels[level + 1].next_pair_freq = INT_MAX;
      level++;
      continue;
    }

    int prev_freq = l->last_freq;
    if (l->next_char_freq < l->next_pair_freq) { /* strict: ties take the pair */
      int nn = leaf_counts[level][level] + 1;
      l->last_freq = l->next_char_freq;
      leaf_counts[level][level] = nn;
      l->next_char_freq = list[nn].freq;
    } else {
      l->last_freq = l->next_pair_freq;
      for (int i = 0; i < level; i++) leaf_counts[level][i] = leaf_counts[level - 1][i];
      levels[l->level - 1].needed = 2;
    }

    l->needed--;
    if (l->needed == 0) {
      if (l->level == max_bits) break;
      levels[l->level + 1].next_pair_freq = prev_freq + l->last_freq;
      level++;
    } else {
      while (levels[level - 1].needed > 0) level--;
    }
  }

  /* :231-233 is an abort() on an internal invariant. */
  int bits = 1;
  const int *counts = leaf_counts[max_bits];
  for (int lv = max_bits; lv > 0; lv--) { /* :237-242 */
    h->bit_count[bits] = counts[lv] - counts[lv - 1];
    bits++;
  }
  h->bit_count[0] = 0; /* never read with n == 0 (:259) */
  return max_bits;
}

/* huffman-code.mbt:250-280 */
static void assign_encoding_and_size(orc_huffman_encoder *h, int max_bits,
                                     orc_literal_node *list, int list_len) {
  int code = 0;
  for (int n = 0; n <= max_bits; n++) {
    int bits = h->bit_count[n];
    code <<= 1;
    if (n == 0 || bits == 0) continue;
    orc_literal_node *chunk = list + (list_len - bits);
    sort_nodes(chunk, bits, by_literal);
    for (int k = 0; k < bits; k++) {
      int key = (int)chunk[k].literal;
      h->codes[key].code = orc_reverse_bits((uint32_t)code & 0xffff, n);
      h->codes[key].len = (uint32_t)n;
      code++;
    }
    list_len -= bits;
  }
}

/* huffman-code.mbt:295-343 */
void orc_henc_generate(orc_huffman_encoder *h, const int32_t *freq, int nfreq,
                       int max_bits) {
  orc_literal_node *list = h->freqcache;
  int count = 0;
  for (int i = 0; i < nfreq; i++) {
    if (freq[i] != 0) {
      list[count].literal = (uint32_t)(i & 0xffff);
      list[count].freq = freq[i];
      count++;
    } else {
      h->codes[i].len = 0; /* code keeps its stale value (:320) */
    }
  }

  if (count <= 2) { /* :326-336 */
    for (int i = 0; i < count; i++) {
      int key = (int)(list[i].literal & 0xffff);
      h->codes[key].code = (uint32_t)(i & 0xffff);
      h->codes[key].len = 1;
    }
    return;
  }
  sort_nodes(list, count, by_frequency); /* :337 */
  int mb = bit_counts(h, list, count, max_bits);
  assign_encoding_and_size(h, mb, list, count);
}

void orc_huffman_generate(const int32_t *freq, int n, int max_bits, uint32_t *codes,
                          uint32_t *lens) {
  orc_huffman_encoder h;
  orc_henc_init(&h, n);
  orc_henc_generate(&h, freq, n, max_bits);
  for (int i = 0; i < n; i++) {
    codes[i] = h.codes[i].code;
    lens[i] = h.codes[i].len;
  }
}
