/*
 * merge_stats_oracle.c -- an instrumented COPY of oracle/lva_oracle.c for scripts/merge_stats.py (kernel design input:
 * pops per target, duplicate lineage, twin words).  Not the oracle: the file that defines "correct" is oracle/lva_oracle.c,
 * which carries none of the LVA_ORACLE_STATS bookkeeping below.  Built into /tmp by merge_stats.py only.
 *
 * lva_oracle.c -- plain-C CPU oracle of the reference list-Viterbi decode path.
 *
 * TEST INFRASTRUCTURE ONLY (see lva_oracle.h).  Never linked into the product.
 *
 * Every function cites the lines of /root/reference/viterbi/viterbi_convolutional_code.cpp
 * (written ":NNN") whose behaviour it restates.  Data structures differ from the
 * reference on purpose (flat arrays, W-word messages, precomputed per-position
 * tables); the observable results are the same, which tests/golden pins against
 * the unmodified reference binary.
 */
#define _GNU_SOURCE
#include "lva_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NCRF 8u        /* flip A,C,G,T then flop A,C,G,T                      (:19-21) */
#define MAX_POS 256u   /* st_pos2msg_pos[BITSET_SIZE]                          (:28,75) */
#define MAX_PRED 32    /* stay + up to 7 source crf states x up to 4 conv preds */

typedef struct {
  uint32_t conv;
  uint8_t crf, row, col, shift, newbits;
} pred_t;

struct lva_oracle_code {
  int m, rate, rc;
  uint32_t msg_len, nconv, g[2], init, final;
  int plen;
  uint8_t pattern[16];
  uint32_t npos;
  uint32_t pos2msg[MAX_POS + 1];
  uint32_t sync_len, sync_period;
  uint8_t sync[MAX_POS];
};

/* ------------------------------------------------------------------ bits */

/* reverse_integer_bits (:417-424) */
static uint32_t bitrev(uint32_t v, uint32_t nbits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < nbits; i++) r = (r << 1) | ((v >> i) & 1u);
  return r;
}

/* conv_prev_state (:433-438): undo one shift, the lost LSB was `bit` */
static uint32_t step_back(const lva_oracle_code *c, uint32_t st, uint32_t bit) {
  return ((st << 1) | (bit & 1u)) & (c->nconv - 1u);
}

/* conv_next_state (:426-431): the new bit enters at position m-1 */
static uint32_t step_fwd(const lva_oracle_code *c, uint32_t st, uint32_t bit) {
  return (st | (bit ? c->nconv : 0u)) >> 1;
}

/* conv_output (:440-448): parity of the (m+1)-bit register under generator k,
 * complemented when decoding a reverse-complement read */
static uint32_t out_bit(const lva_oracle_code *c, int k, uint32_t st, uint32_t bit) {
  uint32_t reg = st | (bit ? c->nconv : 0u);
  return (uint32_t)(__builtin_parity(reg & c->g[k])) ^ (uint32_t)(c->rc ? 1 : 0);
}

/* ------------------------------------------------------------------ code setup */

/* set_conv_params (:264-415) */
lva_oracle_code *lva_oracle_code_new(int mem_conv, int rate, uint32_t msg_len, int rc,
                                     const char *sync_marker, uint32_t sync_period, int *status) {
  int st_local;
  if (!status) status = &st_local;
  *status = LVA_ORACLE_OK;
  lva_oracle_code *c = (lva_oracle_code *)calloc(1, sizeof(*c));
  if (!c) { *status = LVA_ORACLE_NOMEM; return NULL; }
  c->m = mem_conv; c->rate = rate; c->rc = rc ? 1 : 0; c->msg_len = msg_len;
  /* generator pairs (octal) and start states (:269-293) */
  switch (mem_conv) {
    case 6:  c->g[0] = 0171;    c->g[1] = 0133;    c->init = 0x25;   break;   /* 0b100101 */
    case 8:  c->g[0] = 0515;    c->g[1] = 0677;    c->init = 0x96;   break;   /* 0b10010110 */
    case 11: c->g[0] = 05537;   c->g[1] = 06131;   c->init = 0x4B1;  break;   /* 0b10010110001 */
    case 14: c->g[0] = 075063;  c->g[1] = 056711;  c->init = 0x258D; break;   /* 0b10010110001101 */
    default: *status = LVA_ORACLE_BAD_MEM; free(c); return NULL;
  }
  c->nconv = 1u << mem_conv;
  c->final = bitrev(c->init, (uint32_t)mem_conv);                      /* :294 */
  /* puncturing block types per rate (:296-339):
   * 0 keeps both output bits of one input bit; 1,2,3 keep two of the four
   * output bits of two input bits (which two: see emit loop in encode) */
  static const uint8_t P1[] = {0}, P2[] = {0, 2, 0}, P3[] = {0, 1}, P4[] = {0, 3, 0, 2, 1},
                       P5[] = {0, 1, 2}, P7[] = {0, 3, 1, 1};
  const uint8_t *pp; int pl;
  switch (rate) {
    case 1: pp = P1; pl = 1; break;
    case 2: pp = P2; pl = 3; break;
    case 3: pp = P3; pl = 2; break;
    case 4: pp = P4; pl = 5; break;
    case 5: pp = P5; pl = 3; break;
    case 7: pp = P7; pl = 4; break;
    default: *status = LVA_ORACLE_BAD_RATE; free(c); return NULL;
  }
  memcpy(c->pattern, pp, (size_t)pl); c->plen = pl;
  /* trellis positions: position p is reached after consuming pos2msg[p] message bits (:344-357) */
  uint32_t total = msg_len + (uint32_t)mem_conv, bits = 0;
  c->npos = 1; c->pos2msg[0] = 0;
  while (bits < total) {
    bits += (c->pattern[(c->npos - 1) % (uint32_t)pl] == 0) ? 1u : 2u;
    if (c->npos >= MAX_POS) { *status = LVA_ORACLE_MSG_TOO_LONG; free(c); return NULL; }
    c->pos2msg[c->npos++] = bits;
  }
  if (bits != total) { *status = LVA_ORACLE_BAD_LENGTH; free(c); return NULL; }

  if (c->rc) {                                                         /* :359-386 */
    c->g[0] = bitrev(c->g[0], (uint32_t)mem_conv + 1);
    c->g[1] = bitrev(c->g[1], (uint32_t)mem_conv + 1);
    uint32_t old_init_rev = bitrev(c->init, (uint32_t)mem_conv);
    c->init = bitrev(c->final, (uint32_t)mem_conv);
    c->final = old_init_rev;
    uint8_t fwd[16]; memcpy(fwd, c->pattern, 16);
    static const uint8_t mirror[4] = {0, 2, 1, 3};   /* block types 1 and 2 swap when read backwards */
    uint32_t last = (c->npos - 2) % (uint32_t)pl;    /* block type of the last forward step */
    for (uint32_t i = 0; i < (uint32_t)pl; i++)
      c->pattern[i] = mirror[fwd[((uint32_t)pl - i + last) % (uint32_t)pl]];
    for (uint32_t i = 0, j = c->npos - 1; i < j; i++, j--) {
      uint32_t tmp = c->pos2msg[i]; c->pos2msg[i] = c->pos2msg[j]; c->pos2msg[j] = tmp;
    }
    for (uint32_t i = 0; i < c->npos; i++) c->pos2msg[i] = total - c->pos2msg[i];
  }

  if (sync_marker && sync_marker[0]) {                                 /* :388-414 */
    size_t n = strlen(sync_marker);
    if (n >= MAX_POS) { *status = LVA_ORACLE_BAD_SYNC; free(c); return NULL; }
    if (sync_period < n) { *status = LVA_ORACLE_BAD_SYNC; free(c); return NULL; }
    for (size_t i = 0; i < n; i++) {
      if (sync_marker[i] != '0' && sync_marker[i] != '1') { *status = LVA_ORACLE_BAD_SYNC; free(c); return NULL; }
      c->sync[i] = (uint8_t)(sync_marker[i] - '0');
    }
    c->sync_len = (uint32_t)n; c->sync_period = sync_period;
  }
  return c;
}

void lva_oracle_code_free(lva_oracle_code *c) { free(c); }
uint32_t lva_oracle_nstate_pos(const lva_oracle_code *c) { return c->npos; }
uint32_t lva_oracle_nstate_conv(const lva_oracle_code *c) { return c->nconv; }
uint32_t lva_oracle_initial_state(const lva_oracle_code *c) { return c->init; }
uint32_t lva_oracle_final_state(const lva_oracle_code *c) { return c->final; }
void lva_oracle_pos2msg(const lva_oracle_code *c, uint32_t *out) { memcpy(out, c->pos2msg, c->npos * sizeof(uint32_t)); }

/* the block type used by the step into position pos (:693-696) */
int lva_oracle_pattern_at(const lva_oracle_code *c, uint32_t pos) {
  return pos == 0 ? 0 : c->pattern[(pos - 1) % (uint32_t)c->plen];
}

/* is_valid_state (:944-978) evaluated at trellis position pos (:630 passes st_pos2msg_pos[pos]) */
int lva_oracle_is_valid_state(const lva_oracle_code *c, uint32_t pos, uint32_t st_conv) {
  int64_t consumed = (int64_t)c->pos2msg[pos];
  for (int64_t age = 0; age < c->m; age++) {
    int64_t mp = consumed - 1 - age;                 /* message index held in register bit m-1-age */
    int64_t mp_fwd = c->rc ? (int64_t)c->msg_len - 1 - mp : mp;
    uint32_t have = (st_conv >> (c->m - 1 - age)) & 1u;
    if (mp < 0) {
      if (have != ((c->init >> (c->m + mp)) & 1u)) return 0;
    } else if (mp >= (int64_t)c->msg_len) {
      if (have != ((c->final >> (mp - (int64_t)c->msg_len)) & 1u)) return 0;
    } else if (c->sync_len > 0 && (mp_fwd % (int64_t)c->sync_period) < (int64_t)c->sync_len) {
      if (have != c->sync[mp_fwd % (int64_t)c->sync_period]) return 0;
    }
  }
  return 1;
}

/* find_prev_states (:860-942) for target (st_conv, st_crf) under block type `pattern` */
static int list_preds(const lva_oracle_code *c, uint32_t st_conv, uint32_t st_crf, int pattern, pred_t *out) {
  int n = 0;
  uint8_t row = (uint8_t)(st_crf >= 4 ? 4 : st_crf);        /* to_idx_crf_in_post (:582-587) */
  uint32_t newest = st_conv >> (c->m - 1);                  /* last message bit shifted in */
  uint32_t second = (st_conv >> (c->m - 2)) & 1u;           /* the one before it */
  /* stay in the same (position, conv, crf) state: always entry 0 (:867-876) */
  out[n++] = (pred_t){st_conv, (uint8_t)st_crf, row, (uint8_t)st_crf, 0, 0};
  uint32_t want = st_crf & 3u;
  for (uint32_t src = 0; src < NCRF; src++) {
    if (src == st_crf) continue;
    if (st_crf >= 4 && src != st_crf - 4) continue;         /* flop X only from flip X (:879-881) */
    if (pattern == 0) {                                     /* one message bit per base (:890-904) */
      for (uint32_t lost = 0; lost < 2; lost++) {
        uint32_t from = step_back(c, st_conv, lost);
        uint32_t base = 2 * out_bit(c, 0, from, newest) + out_bit(c, 1, from, newest);
        if (base == want) out[n++] = (pred_t){from, (uint8_t)src, row, (uint8_t)src, 1, (uint8_t)newest};
      }
    } else {                                                /* two message bits per base (:906-937) */
      for (uint32_t lost_a = 0; lost_a < 2; lost_a++)
        for (uint32_t lost_b = 0; lost_b < 2; lost_b++) {
          uint32_t mid = step_back(c, st_conv, lost_a);
          uint32_t from = step_back(c, mid, lost_b);
          uint32_t o0 = out_bit(c, 0, from, second), o1 = out_bit(c, 1, from, second);
          uint32_t o2 = out_bit(c, 0, mid, newest), o3 = out_bit(c, 1, mid, newest);
          uint32_t hi, lo;                                  /* which two of the four survive */
          if (pattern == 1) { hi = o1; lo = o2; }
          else if (pattern == 2) { hi = o0; lo = o3; }
          else { hi = o1; lo = o3; }
          uint32_t base = c->rc ? (2 * lo + hi) : (2 * hi + lo);   /* :918-925 */
          if (base == want)
            out[n++] = (pred_t){from, (uint8_t)src, row, (uint8_t)src, 2, (uint8_t)(2 * second + newest)};
        }
    }
  }
  return n;
}

int lva_oracle_prev_states(const lva_oracle_code *c, uint32_t st_conv, uint32_t st_crf, int pattern,
                           int32_t *out, int cap) {
  pred_t tmp[MAX_PRED];
  int n = list_preds(c, st_conv, st_crf, pattern, tmp);
  for (int i = 0; i < n && i < cap; i++) {
    out[6 * i + 0] = (int32_t)tmp[i].conv; out[6 * i + 1] = tmp[i].crf; out[6 * i + 2] = tmp[i].row;
    out[6 * i + 3] = tmp[i].col; out[6 * i + 4] = tmp[i].shift; out[6 * i + 5] = tmp[i].newbits;
  }
  return n;
}

/* ------------------------------------------------------------------ encoder */

/* conv_encode (:450-499) + 2-bits-per-base packing of write_bit_array_in_bases (:540-551) */
int lva_oracle_encode(const lva_oracle_code *c, const uint8_t *msg, uint8_t *out_bases) {
  if (c->rc) return LVA_ORACLE_BAD_RATE;
  uint32_t total = c->msg_len + (uint32_t)c->m;
  uint8_t *coded = (uint8_t *)malloc(2u * total + 4);
  if (!coded) return LVA_ORACLE_NOMEM;
  uint32_t st = c->init;
  for (uint32_t i = 0; i < total; i++) {
    /* message bits, then the m terminating bits taken LSB-first from the final state (:458-464) */
    uint32_t b = i < c->msg_len ? (msg[i] & 1u) : ((c->final >> (i - c->msg_len)) & 1u);
    coded[2 * i] = (uint8_t)out_bit(c, 0, st, b);
    coded[2 * i + 1] = (uint8_t)out_bit(c, 1, st, b);
    st = step_fwd(c, st, b);
  }
  if (st != c->final) { free(coded); return LVA_ORACLE_BAD_LENGTH; }     /* :465-467 */
  uint32_t k = 0;
  for (uint32_t pos = 0; pos + 1 < c->npos; pos++) {                     /* :471-495 */
    uint32_t hi, lo;
    switch (c->pattern[pos % (uint32_t)c->plen]) {
      case 0: hi = coded[k]; lo = coded[k + 1]; k += 2; break;
      case 1: hi = coded[k + 1]; lo = coded[k + 2]; k += 4; break;
      case 2: hi = coded[k]; lo = coded[k + 3]; k += 4; break;
      default: hi = coded[k + 1]; lo = coded[k + 3]; k += 4; break;
    }
    out_bases[pos] = (uint8_t)(2 * hi + lo);
  }
  free(coded);
  return k == 2 * total ? LVA_ORACLE_OK : LVA_ORACLE_BAD_LENGTH;         /* :496-497 */
}

/* ------------------------------------------------------------------ libstdc++ heap, restated */

typedef struct { float score; uint32_t ps; uint32_t j; } hnode;   /* heap_elem_t (:49-55), key = score only */

/* GCC 11 bits/stl_heap.h __push_heap: bubble `v` up from `hole` while parent < v */
static void heap_sift_up(hnode *h, int hole, int top, hnode v) {
  int parent = (hole - 1) / 2;
  while (hole > top && h[parent].score < v.score) {
    h[hole] = h[parent]; hole = parent; parent = (hole - 1) / 2;
  }
  h[hole] = v;
}

/* GCC 11 bits/stl_heap.h __adjust_heap: walk the hole down taking the right child
 * unless right < left, handle a lone left child when len is even, then sift v up */
static void heap_adjust(hnode *h, int hole, int len, hnode v) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (h[child].score < h[child - 1].score) child--;
    h[hole] = h[child]; hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    h[hole] = h[child - 1]; hole = child - 1;
  }
  heap_sift_up(h, hole, top, v);
}

/* std::make_heap */
static void heap_build(hnode *h, int len) {
  if (len < 2) return;
  for (int parent = (len - 2) / 2;; parent--) {
    hnode v = h[parent];
    heap_adjust(h, parent, len, v);
    if (parent == 0) return;
  }
}

/* std::pop_heap followed by back()/pop_back(): returns the popped top, len shrinks by one */
static hnode heap_pop(hnode *h, int *len) {
  int n = *len;
  hnode top = h[0];
  if (n > 1) {
    hnode v = h[n - 1];
    h[n - 1] = h[0];
    heap_adjust(h, 0, n - 1, v);
  }
  *len = n - 1;
  return top;
}

/* push_back followed by std::push_heap */
static void heap_push(hnode *h, int *len, hnode v) {
  int n = *len;
  h[n] = v;
  heap_sift_up(h, n, 0, v);
  *len = n + 1;
}

/* ------------------------------------------------------------------ libstdc++ std::sort, restated
 * Elements: (score, original index); comparator "a before b" = a.score > b.score (:818-821).
 * GCC 11 bits/stl_algo.h: introsort (median-of-3 to first, unguarded partition, threshold 16,
 * depth limit 2*floor(log2 n), heapsort fallback) + final insertion sort. */

typedef struct { float score; uint32_t idx; } sitem;
#define SBEFORE(a, b) ((a).score > (b).score)

static void s_swap(sitem *a, sitem *b) { sitem t = *a; *a = *b; *b = t; }

static void s_unguarded_linear_insert(sitem *last) {
  sitem v = *last; sitem *next = last - 1;
  while (SBEFORE(v, *next)) { *last = *next; last = next; --next; }
  *last = v;
}
static void s_insertion_sort(sitem *first, sitem *last) {
  if (first == last) return;
  for (sitem *i = first + 1; i != last; ++i) {
    if (SBEFORE(*i, *first)) { sitem v = *i; memmove(first + 1, first, (size_t)(i - first) * sizeof(sitem)); *first = v; }
    else s_unguarded_linear_insert(i);
  }
}
/* heap primitives under the sort comparator (used only by the depth-limit fallback) */
static void s_push_heap(sitem *f, long hole, long top, sitem v) {
  long parent = (hole - 1) / 2;
  while (hole > top && SBEFORE(f[parent], v)) { f[hole] = f[parent]; hole = parent; parent = (hole - 1) / 2; }
  f[hole] = v;
}
static void s_adjust_heap(sitem *f, long hole, long len, sitem v) {
  const long top = hole; long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (SBEFORE(f[child], f[child - 1])) child--;
    f[hole] = f[child]; hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); f[hole] = f[child - 1]; hole = child - 1; }
  s_push_heap(f, hole, top, v);
}
static void s_heapsort(sitem *first, sitem *last) {   /* __partial_sort(first,last,last) */
  long len = last - first;
  if (len >= 2) for (long parent = (len - 2) / 2;; parent--) { sitem v = first[parent]; s_adjust_heap(first, parent, len, v); if (parent == 0) break; }
  while (last - first > 1) { --last; sitem v = *last; *last = *first; s_adjust_heap(first, 0, last - first, v); }
}
static void s_median_to_first(sitem *result, sitem *a, sitem *b, sitem *c) {
  if (SBEFORE(*a, *b)) {
    if (SBEFORE(*b, *c)) s_swap(result, b);
    else if (SBEFORE(*a, *c)) s_swap(result, c);
    else s_swap(result, a);
  } else if (SBEFORE(*a, *c)) s_swap(result, a);
  else if (SBEFORE(*b, *c)) s_swap(result, c);
  else s_swap(result, b);
}
static sitem *s_unguarded_partition(sitem *first, sitem *last, sitem *pivot) {
  for (;;) {
    while (SBEFORE(*first, *pivot)) ++first;
    --last;
    while (SBEFORE(*pivot, *last)) --last;
    if (!(first < last)) return first;
    s_swap(first, last); ++first;
  }
}
static void s_introsort_loop(sitem *first, sitem *last, long depth) {
  while (last - first > 16) {
    if (depth == 0) { s_heapsort(first, last); return; }
    --depth;
    sitem *mid = first + (last - first) / 2;
    s_median_to_first(first, first + 1, mid, last - 1);
    sitem *cut = s_unguarded_partition(first + 1, last, first);
    s_introsort_loop(cut, last, depth);
    last = cut;
  }
}
static void std_sort_desc(sitem *first, sitem *last) {
  if (first == last) return;
  long n = last - first, lg = 0;
  while ((n >> (lg + 1)) > 0) lg++;
  s_introsort_loop(first, last, 2 * lg);
  if (last - first > 16) {
    s_insertion_sort(first, first + 16);
    for (sitem *i = first + 16; i != last; ++i) s_unguarded_linear_insert(i);
  } else s_insertion_sort(first, last);
}

/* ------------------------------------------------------------------ decoder */

/* time-step band (:677-679).  (:673-675 are dead stores.) */
void lva_oracle_band(const lva_oracle_code *c, uint32_t t, uint32_t nblk, uint32_t max_deviation,
                     int band_fma, uint32_t *start, uint32_t *end) {
  double q = (double)t / (double)nblk, centre;
  if (band_fma) centre = fma(q, (double)c->npos, -(double)max_deviation);
  else { volatile double prod = q * (double)c->npos; centre = prod - (double)max_deviation; }
  int64_t s64 = (int64_t)centre;
  if (s64 < 0) s64 = 0;
  uint32_t s = (uint32_t)s64;                       /* assignment to uint32_t st_pos_start */
  uint32_t e = s + 2u * max_deviation;              /* uint32_t arithmetic, wraps like the reference */
  if (e > c->npos) e = c->npos;
  *start = s; *end = e;
}

typedef struct {
  float *score;      /* [nstate][L] */
  uint32_t *msg;     /* [nstate][L][W] */
} plane;

static inline void msg_push(uint32_t *dst, const uint32_t *src, uint32_t W, uint32_t shift, uint32_t newbits) {
  /* (msg << shift) | newbits over W little-endian words, bits beyond 32W dropped (:738-741,774) */
  if (shift == 0) { for (uint32_t w = 0; w < W; w++) dst[w] = src[w]; dst[0] |= newbits; return; }
  uint32_t carry = 0;
  for (uint32_t w = 0; w < W; w++) {
    uint32_t v = src[w];
    dst[w] = (v << shift) | carry;
    carry = v >> (32 - shift);
  }
  dst[0] |= newbits;
}

#ifdef LVA_ORACLE_STATS
/* merge statistics for kernel design (scripts/merge_stats.py builds a copy of this file with
 * -DLVA_ORACLE_STATS): how many heap pops a target needs, how deep each candidate list is consumed */
#define ST_MAXL 65
uint64_t lva_stats_targets, lva_stats_pops[8 * ST_MAXL + 1], lva_stats_accepted[ST_MAXL + 1];
uint64_t lva_stats_stay_depth[ST_MAXL + 1], lva_stats_src_depth[ST_MAXL + 1], lva_stats_src_rank_depth[8][ST_MAXL + 1];
uint64_t lva_stats_src_pops_below[ST_MAXL + 1];     /* [K]: source-list pops with index < K */
uint64_t lva_stats_src_pops_total, lva_stats_stay_pops_total, lva_stats_targets_src_within[ST_MAXL + 1];
uint64_t lva_stats_dup_kind[3];                     /* duplicate pops: source vs source, popped from stay, matched a stay entry */
uint64_t lva_stats_pops_noss[8 * ST_MAXL + 1];      /* pops per target if source-vs-source duplicates were skipped, not popped */
/* lineage: is a duplicate the SAME path seen twice (the stay entry was created from exactly that source entry, which has stayed in
 * its own list since), or two different paths that spell the same message? */
uint64_t lva_stats_dup_lineage[3];                  /* [0] same path, [1] different paths, [2] identities equal but messages differ */
uint64_t lva_stats_pops_nolin[8 * ST_MAXL + 1];     /* pops per target if same-path duplicates were dropped without a pop */
/* the same with what a GPU lane would have: one twin word per entry (predecessor list, index) written at the merge, one
 * forward map per list and step (old index -> new index), a step stamp per list (stale rows at the band edge give no twins) */
uint64_t lva_stats_tw[3];                           /* [0] duplicate pops named by the twin words, [1] named although not a duplicate (must be 0), [2] all duplicate pops */
uint64_t lva_stats_pops_notw[8 * ST_MAXL + 1];      /* pops per target without the duplicate pops the twin words name */
/* ties: targets in which a popped candidate had an equal score still in the heap (the GPU's fast paths leave such a target to the exact
 * path): [0] targets with a tie, [1] of them: every tie was between exactly two candidates carrying the SAME message (whichever pops
 * first, the list comes out the same), [2] tie events, [3] benign tie events */
uint64_t lva_stats_ties[4];
void lva_oracle_stats_reset(void) {
  memset(lva_stats_ties, 0, sizeof lva_stats_ties);
  memset(lva_stats_dup_kind, 0, sizeof lva_stats_dup_kind); memset(lva_stats_pops_noss, 0, sizeof lva_stats_pops_noss);
  memset(lva_stats_dup_lineage, 0, sizeof lva_stats_dup_lineage); memset(lva_stats_pops_nolin, 0, sizeof lva_stats_pops_nolin);
  memset(lva_stats_tw, 0, sizeof lva_stats_tw); memset(lva_stats_pops_notw, 0, sizeof lva_stats_pops_notw);
  lva_stats_targets = lva_stats_src_pops_total = lva_stats_stay_pops_total = 0;
  memset(lva_stats_pops, 0, sizeof lva_stats_pops); memset(lva_stats_accepted, 0, sizeof lva_stats_accepted);
  memset(lva_stats_stay_depth, 0, sizeof lva_stats_stay_depth); memset(lva_stats_src_depth, 0, sizeof lva_stats_src_depth);
  memset(lva_stats_src_rank_depth, 0, sizeof lva_stats_src_rank_depth);
  memset(lva_stats_src_pops_below, 0, sizeof lva_stats_src_pops_below);
  memset(lva_stats_targets_src_within, 0, sizeof lva_stats_targets_src_within);
}
static void lva_oracle_stats_record(uint32_t L, int np, const uint32_t *depth, uint32_t pops, uint32_t accepted, const uint32_t *dupk) {
  if (L >= ST_MAXL) return;
#pragma omp critical(lva_stats)
  {
    lva_stats_targets++; lva_stats_pops[pops]++; lva_stats_accepted[accepted]++;
    for (int q = 0; q < 3; q++) lva_stats_dup_kind[q] += dupk[q];
    lva_stats_pops_noss[pops - dupk[0]]++;
    lva_stats_dup_lineage[0] += dupk[3]; lva_stats_dup_lineage[1] += dupk[4]; lva_stats_dup_lineage[2] += dupk[5];
    lva_stats_pops_nolin[pops - dupk[3]]++;
    lva_stats_tw[0] += dupk[6]; lva_stats_tw[1] += dupk[7]; lva_stats_tw[2] += dupk[0] + dupk[1] + dupk[2];
    lva_stats_pops_notw[pops - dupk[6]]++;
    lva_stats_stay_depth[depth[0]]++; lva_stats_stay_pops_total += depth[0];
    uint32_t d[8]; int n = 0; uint32_t mx = 0;
    for (int i = 1; i < np; i++) {
      d[n++] = depth[i]; lva_stats_src_depth[depth[i]]++; lva_stats_src_pops_total += depth[i];
      if (depth[i] > mx) mx = depth[i];
      for (uint32_t K = 0; K <= L; K++) lva_stats_src_pops_below[K] += depth[i] < K ? depth[i] : K;
    }
    for (uint32_t K = mx; K <= L; K++) lva_stats_targets_src_within[K]++;
    for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) if (d[b] > d[a]) { uint32_t t = d[a]; d[a] = d[b]; d[b] = t; }
    for (int a = 0; a < n; a++) lva_stats_src_rank_depth[a][d[a]]++;
  }
}
#endif

int lva_oracle_decode(const lva_oracle_code *c, const float *post, uint32_t nblk, uint32_t L,
                      uint32_t max_deviation, int num_threads, uint32_t max_steps, int band_fma,
                      uint8_t *out_msgs, float *out_scores, uint32_t *out_count) {
  const uint32_t npos = c->npos, nconv = c->nconv, m = (uint32_t)c->m;
  *out_count = 0;
  uint64_t nstate64 = (uint64_t)npos * NCRF * nconv;
  if (nstate64 >= ((uint64_t)1 << 32)) return LVA_ORACLE_TOO_MANY_STATES;        /* :595-597 */
  if (nblk < npos + 1) return LVA_ORACLE_POST_TOO_SHORT;                           /* :600-601 */
  if (c->msg_len > 255 || c->msg_len + m > 256) return LVA_ORACLE_MSG_TOO_LONG;   /* :604-605, :831 */
  if (L == 0) return LVA_ORACLE_OK;
  const size_t nstate = (size_t)nstate64;
  const uint32_t W = (c->msg_len + m + 31) / 32;
  const float NEG = -INFINITY;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);                           /* :593 */
#else
  (void)num_threads;
#endif

  plane buf[2];
  for (int b = 0; b < 2; b++) {
    buf[b].score = (float *)malloc(nstate * L * sizeof(float));
    buf[b].msg = (uint32_t *)calloc(nstate * L * W, sizeof(uint32_t));
    if (!buf[b].score || !buf[b].msg) return LVA_ORACLE_NOMEM;
    for (size_t i = 0; i < nstate * L; i++) buf[b].score[i] = NEG;                /* :616-619 */
  }
#ifdef LVA_ORACLE_STATS
  uint64_t *st_lin[2], *st_par[2]; uint32_t *st_pst[2];
  for (int b = 0; b < 2; b++) {
    st_lin[b] = (uint64_t *)calloc(nstate * L, sizeof(uint64_t)); st_par[b] = (uint64_t *)calloc(nstate * L, sizeof(uint64_t));
    st_pst[b] = (uint32_t *)calloc(nstate * L, sizeof(uint32_t));
    for (size_t i = 0; i < nstate * L; i++) st_lin[b][i] = (uint64_t)(i + 1);        /* time 0: every slot its own path */
  }
  uint16_t *tw_w[2]; uint8_t *tw_f[2]; uint32_t *tw_stamp[2];
  for (int b = 0; b < 2; b++) {
    tw_w[b] = (uint16_t *)calloc(nstate * L, sizeof(uint16_t)); tw_f[b] = (uint8_t *)calloc(nstate * L, 1);
    tw_stamp[b] = (uint32_t *)calloc(nstate, sizeof(uint32_t));
  }
#endif
  /* valid-state mask (:624-630) */
  uint8_t *valid = (uint8_t *)malloc((size_t)npos * nconv);
  for (uint32_t p = 0; p < npos; p++)
    for (uint32_t s = 0; s < nconv; s++) valid[(size_t)p * nconv + s] = (uint8_t)lva_oracle_is_valid_state(c, p, s);
  /* predecessor tables for the block types this rate uses (:636-650) */
  pred_t *ptab[4] = {0, 0, 0, 0}; uint8_t *pcnt[4] = {0, 0, 0, 0};
  for (int pt = 0; pt < 4; pt++) {
    int used = 0;
    for (int i = 0; i < c->plen; i++) used |= (c->pattern[i] == pt);
    if (!used) continue;
    ptab[pt] = (pred_t *)malloc((size_t)nconv * NCRF * MAX_PRED * sizeof(pred_t));
    pcnt[pt] = (uint8_t *)malloc((size_t)nconv * NCRF);
    for (uint32_t s = 0; s < nconv; s++)
      for (uint32_t k = 0; k < NCRF; k++)
        pcnt[pt][s * NCRF + k] = (uint8_t)list_preds(c, s, k, pt, ptab[pt] + ((size_t)s * NCRF + k) * MAX_PRED);
  }

  plane *cur = &buf[0], *prev = &buf[1];
  for (uint32_t k = 0; k < NCRF; k++)                                              /* :657-663 */
    cur->score[((size_t)0 * nconv * NCRF + (size_t)c->init * NCRF + k) * L + 0] = 0.0f;

  uint32_t steps = (max_steps && max_steps < nblk) ? max_steps : nblk;
  for (uint32_t t = 0; t < steps; t++) {                                           /* :667 */
    plane *tmp = cur; cur = prev; prev = tmp;                                      /* :669-670 */
    uint32_t lo, hi;
    lva_oracle_band(c, t, nblk, max_deviation, band_fma, &lo, &hi);
    const float *pt_row = post + (size_t)t * 40;   /* [to_row 0..4][from 0..7], read_crf_post (:553-575) */
#ifdef LVA_ORACLE_STATS
    const int st_cb = cur == &buf[0] ? 0 : 1, st_pb = 1 - st_cb;
#endif
    long p;
#pragma omp parallel for schedule(dynamic)
    for (p = (long)lo; p < (long)hi; p++) {                                        /* :685-687 */
      const uint32_t pos = (uint32_t)p;
      const int ptype = lva_oracle_pattern_at(c, pos);
      hnode heap[MAX_PRED + 1];
      uint32_t cand[8];
      for (uint32_t s = 0; s < nconv; s++) {
        if (!valid[(size_t)pos * nconv + s]) continue;                             /* :700 */
        for (uint32_t k = 0; k < NCRF; k++) {
          const size_t st = ((size_t)pos * nconv + s) * NCRF + k;                  /* get_state_idx (:577-580) */
          const pred_t *pl = ptab[ptype] + ((size_t)s * NCRF + k) * MAX_PRED;
          const int np = pcnt[ptype][s * NCRF + k];
          float *cs = cur->score + st * L;
          uint32_t *cm = cur->msg + st * L * W;
          if (pos == 0) {                                                          /* :706-713 */
            memcpy(cm, prev->msg + st * L * W, W * sizeof(uint32_t));
#ifdef LVA_ORACLE_STATS
            st_lin[st_cb][st * L] = st_lin[st_pb][st * L]; st_par[st_cb][st * L] = st_par[st_pb][st * L]; st_pst[st_cb][st * L] = st_pst[st_pb][st * L];
#endif
            cs[0] = prev->score[st * L] + pt_row[pl[0].row * 8 + pl[0].col];
            for (uint32_t l = 1; l < L; l++) cs[l] = NEG;
            continue;
          }
          if (L == 1) {                                                            /* :715-742 */
            float best = NEG; int bi = 0; size_t bst = 0;
            for (int i = 0; i < np; i++) {
              size_t from = ((size_t)(pos - (i == 0 ? 0u : 1u)) * nconv + pl[i].conv) * NCRF + pl[i].crf;
              float sc = prev->score[from] + pt_row[pl[i].row * 8 + pl[i].col];
              if (sc > best) { best = sc; bi = i; bst = from; }
            }
            if (best == NEG) cs[0] = NEG;
            else { cs[0] = best; msg_push(cm, prev->msg + bst * W, W, pl[bi].shift, pl[bi].newbits); }
            continue;
          }
          /* list merge (:743-800) */
          int hn = 0;
          size_t from_of[MAX_PRED];
          for (int i = 0; i < np; i++) {
            size_t from = ((size_t)(pos - (i == 0 ? 0u : 1u)) * nconv + pl[i].conv) * NCRF + pl[i].crf;
            from_of[i] = from;
            float head = prev->score[from * L];
            if (head != NEG) heap[hn++] = (hnode){head + pt_row[pl[i].row * 8 + pl[i].col], (uint32_t)i, 0};
          }
          heap_build(heap, hn);
          uint32_t l = 0;
#ifdef LVA_ORACLE_STATS
          uint32_t st_tie_events = 0, st_tie_benign = 0;
          uint32_t st_depth[MAX_PRED] = {0}, st_pops = 0, st_dupk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_acc_ps[ST_MAXL]; const int st_heads = hn;
          /* twin words of the own list, translated through the forward maps of the (fresh) source lists */
          uint16_t tw_now[ST_MAXL], tw_new[ST_MAXL]; uint8_t tw_fnew[ST_MAXL], tw_dead_stay[ST_MAXL], tw_dead_src[MAX_PRED][ST_MAXL];
          if (L < ST_MAXL) {
            memset(tw_now, 0, sizeof tw_now); memset(tw_new, 0, sizeof tw_new); memset(tw_fnew, 0xFF, sizeof tw_fnew);
            memset(tw_dead_stay, 0, sizeof tw_dead_stay); memset(tw_dead_src, 0, sizeof tw_dead_src);
            if (tw_stamp[st_pb][from_of[0]] == t)
              for (uint32_t j = 0; j < L; j++) {
                const uint16_t v = tw_w[st_pb][from_of[0] * L + j];
                if (!(v & 0x8000u)) continue;
                const uint32_t ti = (v >> 8) & 0x7Fu, tj = v & 0xFFu;
                if ((int)ti >= np || tw_stamp[st_pb][from_of[ti]] != t) continue;
                const uint8_t f = tw_f[st_pb][from_of[ti] * L + tj];
                if (f != 0xFF) tw_now[j] = (uint16_t)(0x8000u | ti << 8 | f);
              }
          }
#endif
          while (hn > 0 && l < L) {
            hnode top = heap_pop(heap, &hn);
#ifdef LVA_ORACLE_STATS
            st_depth[top.ps]++; st_pops++;
#endif
            const pred_t *pi = &pl[top.ps];
            const size_t from = from_of[top.ps];
            msg_push(cand, prev->msg + (from * L + top.j) * W, W, pi->shift, pi->newbits);
#ifdef LVA_ORACLE_STATS
            {
              int neq = 0, same = 0;
              for (int e = 0; e < hn; e++)
                if (heap[e].score == top.score) {
                  uint32_t other[8];
                  msg_push(other, prev->msg + (from_of[heap[e].ps] * L + heap[e].j) * W, W, pl[heap[e].ps].shift, pl[heap[e].ps].newbits);
                  neq++; same += memcmp(other, cand, W * sizeof(uint32_t)) == 0;
                }
              if (neq) { st_tie_events++; if (neq == 1 && same == 1) st_tie_benign++; }
            }
#endif
            int dup = 0;
            for (uint32_t a = 0; a < l && !dup; a++) {
              dup = (memcmp(cm + a * W, cand, W * sizeof(uint32_t)) == 0);
#ifdef LVA_ORACLE_STATS
              if (dup && L < ST_MAXL) {    /* twin words: the forward map and the refreshed twin of the accepted entry */
                if (top.ps == 0) tw_fnew[top.j] = (uint8_t)a;
                else tw_new[a] = (uint16_t)(0x8000u | top.ps << 8 | top.j);
              }
              if (!dup && L < ST_MAXL) {   /* the converse: identities that say "twin" for two different messages (must never happen) */
                const size_t ce = from * L + top.j, ae = st * L + a;
                int idm = 0;
                if (top.ps != 0 && st_acc_ps[a] == 0) idm = st_par[st_cb][ae] == st_lin[st_pb][ce] && st_pst[st_cb][ae] == (uint32_t)from;
                else if (top.ps == 0 && st_acc_ps[a] != 0) idm = st_par[st_pb][ce] == st_par[st_cb][ae] && st_pst[st_pb][ce] == st_pst[st_cb][ae];
                if (idm) st_dupk[5]++;
              }
              if (dup && L < ST_MAXL) {
                st_dupk[top.ps == 0 ? 1 : (st_acc_ps[a] == 0 ? 2 : 0)]++;
                /* the stay-side entry's parent (path and state it was created from) against the source-side entry */
                const size_t ce = from * L + top.j, ae = st * L + a;
                int same = 0;
                if (top.ps != 0 && st_acc_ps[a] == 0) same = st_par[st_cb][ae] == st_lin[st_pb][ce] && st_pst[st_cb][ae] == (uint32_t)from;
                else if (top.ps == 0 && st_acc_ps[a] != 0) same = st_par[st_pb][ce] == st_par[st_cb][ae] && st_pst[st_pb][ce] == st_pst[st_cb][ae];
                st_dupk[same ? 3 : 4]++;
                /* keep identities alive across the merge: a rejected stay entry hands its identity to the copy that beat it
                 * (targets downstream know the message under that identity); a rejected source entry is the accepted
                 * entry's twin from now on */
                if (top.ps == 0) st_lin[st_cb][ae] = st_lin[st_pb][ce];
                else { st_par[st_cb][ae] = st_lin[st_pb][ce]; st_pst[st_cb][ae] = (uint32_t)from; }
              }
#endif
            }
#ifdef LVA_ORACLE_STATS
            if (L < ST_MAXL && (top.ps == 0 ? tw_dead_stay[top.j] : tw_dead_src[top.ps][top.j])) st_dupk[dup ? 6 : 7]++;
            if (!dup && L < ST_MAXL) {
              if (top.ps == 0) {
                tw_fnew[top.j] = (uint8_t)l; tw_new[l] = tw_now[top.j];
                if (tw_now[top.j] & 0x8000u) tw_dead_src[(tw_now[top.j] >> 8) & 0x7Fu][tw_now[top.j] & 0xFFu] = 1;
              } else {
                tw_new[l] = (uint16_t)(0x8000u | top.ps << 8 | top.j);
                for (uint32_t j0 = 0; j0 < L; j0++) if (tw_now[j0] == tw_new[l]) tw_dead_stay[j0] = 1;
              }
              st_acc_ps[l] = top.ps;
              const size_t ce = from * L + top.j, ae = st * L + l;
              if (top.ps == 0) { st_lin[st_cb][ae] = st_lin[st_pb][ce]; st_par[st_cb][ae] = st_par[st_pb][ce]; st_pst[st_cb][ae] = st_pst[st_pb][ce]; }
              else { st_lin[st_cb][ae] = ((uint64_t)(t + 1) << 36) | (uint64_t)(ae + 1); st_par[st_cb][ae] = st_lin[st_pb][ce]; st_pst[st_cb][ae] = (uint32_t)from; }
            }
#endif
            if (!dup) { memcpy(cm + l * W, cand, W * sizeof(uint32_t)); cs[l] = top.score; l++; }
            if (top.j == L - 1) continue;                                          /* :788 */
            float nxt = prev->score[from * L + top.j + 1];
            if (nxt != NEG)
              heap_push(heap, &hn, (hnode){nxt + pt_row[pi->row * 8 + pi->col], top.ps, top.j + 1});
          }
#ifdef LVA_ORACLE_STATS
          if (st_heads > 0) lva_oracle_stats_record(L, np, st_depth, st_pops, l, st_dupk);
          if (st_tie_events) {
#pragma omp critical(lva_stats_ties)
            { lva_stats_ties[0]++; lva_stats_ties[1] += st_tie_events == st_tie_benign; lva_stats_ties[2] += st_tie_events; lva_stats_ties[3] += st_tie_benign; }
          }
          if (L < ST_MAXL) {
            for (uint32_t j = 0; j < L; j++) { tw_w[st_cb][st * L + j] = j < l ? tw_new[j] : 0; tw_f[st_cb][st * L + j] = tw_fnew[j]; }
            tw_stamp[st_cb][st] = t + 1;
          }
#endif
          for (; l < L; l++) cs[l] = NEG;                                          /* :799 */
        }
      }
    }
  }

  /* final selection (:806-824) */
  sitem *fin = (sitem *)malloc((size_t)NCRF * L * sizeof(sitem));
  uint32_t nf = 0;
  for (uint32_t k = 0; k < NCRF; k++) {
    size_t st = ((size_t)(npos - 1) * nconv + c->final) * NCRF + k;
    for (uint32_t l = 0; l < L; l++)
      if (cur->score[st * L + l] != NEG) fin[nf++] = (sitem){cur->score[st * L + l], (uint32_t)(k * L + l)};
  }
  std_sort_desc(fin, fin + nf);
  if (nf > L) nf = L;
  /* bitset -> bits, oldest message bit first; reverse for rc (:826-836) */
  for (uint32_t i = 0; i < nf; i++) {
    uint32_t k = fin[i].idx / L, l = fin[i].idx % L;
    size_t st = ((size_t)(npos - 1) * nconv + c->final) * NCRF + k;
    const uint32_t *mw = cur->msg + (st * L + l) * W;
    uint8_t *o = out_msgs + (size_t)i * c->msg_len;
    for (uint32_t b = 0; b < c->msg_len; b++) {
      uint32_t bit = c->msg_len + m - 1 - b;
      uint8_t v = (uint8_t)((mw[bit >> 5] >> (bit & 31)) & 1u);
      if (c->rc) o[c->msg_len - 1 - b] = v; else o[b] = v;
    }
    out_scores[i] = fin[i].score;
  }
  *out_count = nf;

  free(fin);
  for (int pt = 0; pt < 4; pt++) { free(ptab[pt]); free(pcnt[pt]); }
  free(valid);
  for (int b = 0; b < 2; b++) { free(buf[b].score); free(buf[b].msg); }
  return LVA_ORACLE_OK;
}
