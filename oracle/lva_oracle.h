/*
 * lva_oracle.h -- CPU oracle for the list-Viterbi decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
 * nanopore_dna_storage_amd/ or its C-ABI library) may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker.
 *
 * It restates, in plain C over flat arrays, what the reference does in
 *   viterbi/viterbi_convolutional_code.cpp   (cited per function in the .c)
 * including the two behaviours that decide bit-exactness (SURVEY.md TL;DR 5):
 *   (a) the two score/message buffers cover ALL trellis positions and are never
 *       cleared, so out-of-band entries keep stale values (":667-687");
 *   (b) ties are broken exactly as libstdc++'s make_heap/pop_heap/push_heap and
 *       std::sort do (GCC 11 bits/stl_heap.h, bits/stl_algo.h), restated here.
 *   (c) the band start is evaluated with a fused multiply-subtract, which is
 *       what `g++ -O3 -march=native` (install.sh:9) emits for ":678" on any
 *       FMA-capable host (vfmsub213sd in the reference binary).
 *
 * Parity pinned: this file is checked against the unmodified reference binary
 * (oracle/_ref, built by oracle/Makefile from /root/reference) by
 * tests/golden/make_golden.py, whose outputs are committed under tests/golden/.
 */
#ifndef LVA_ORACLE_H
#define LVA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct lva_oracle_code lva_oracle_code;

/* status codes */
enum {
  LVA_ORACLE_OK = 0,
  LVA_ORACLE_BAD_MEM = -1,       /* reference: "Invalid mem_conv"              (:290-292) */
  LVA_ORACLE_BAD_RATE = -2,      /* reference: "Invalid rate parameter"        (:336-338) */
  LVA_ORACLE_BAD_LENGTH = -3,    /* reference: "Output length not even"        (:353-357) */
  LVA_ORACLE_BAD_SYNC = -4,      /* reference: sync marker checks              (:390-413) */
  LVA_ORACLE_TOO_MANY_STATES = -5, /* reference: runtime_error                 (:595-597) */
  LVA_ORACLE_POST_TOO_SHORT = -6,  /* reference: runtime_error                 (:600-601) */
  LVA_ORACLE_MSG_TOO_LONG = -7,    /* reference: runtime_error (:604-605) + uint8_t loop (:831) */
  LVA_ORACLE_NOMEM = -8
};

/* set_conv_params (:264-415).  sync_marker may be NULL or "" for none. */
lva_oracle_code *lva_oracle_code_new(int mem_conv, int rate, uint32_t msg_len, int rc,
                                     const char *sync_marker, uint32_t sync_period, int *status);
void lva_oracle_code_free(lva_oracle_code *c);

uint32_t lva_oracle_nstate_pos(const lva_oracle_code *c);
uint32_t lva_oracle_nstate_conv(const lva_oracle_code *c);
uint32_t lva_oracle_initial_state(const lva_oracle_code *c);
uint32_t lva_oracle_final_state(const lva_oracle_code *c);
/* copies nstate_pos entries */
void lva_oracle_pos2msg(const lva_oracle_code *c, uint32_t *out);
/* pattern (0..3) that governs the step INTO position pos (pos>=1), 0 for pos 0 */
int lva_oracle_pattern_at(const lva_oracle_code *c, uint32_t pos);
int lva_oracle_is_valid_state(const lva_oracle_code *c, uint32_t pos, uint32_t st_conv);
/* find_prev_states (:860-942): fills up to cap entries of 6 ints each
 * (st_conv, st_crf, post_row, post_col, msg_shift, msg_newbits); returns count */
int lva_oracle_prev_states(const lva_oracle_code *c, uint32_t st_conv, uint32_t st_crf, int pattern,
                           int32_t *out, int cap);

/* conv_encode (:450-499) for a forward code (rc must be 0).  msg: msg_len bytes of 0/1.
 * out_bases: (nstate_pos-1) bytes, values 0..3 (A,C,G,T).  returns status */
int lva_oracle_encode(const lva_oracle_code *c, const uint8_t *msg, uint8_t *out_bases);

/* decode_post_conv_parallel_LVA (:589-858).
 * post: nblk*40 floats in the .post file order.  out_msgs: list_size*msg_len bytes (0/1),
 * out_scores: list_size floats, *out_count = number of list entries produced (<= list_size).
 * max_steps: 0 = all nblk steps; >0 stops the forward pass early (timing samples only:
 * the list returned is then meaningless).  band_fma: 1 = fused band start (reference as
 * built by install.sh on FMA hosts), 0 = separately rounded multiply and subtract. */
int lva_oracle_decode(const lva_oracle_code *c, const float *post, uint32_t nblk, uint32_t list_size,
                      uint32_t max_deviation, int num_threads, uint32_t max_steps, int band_fma,
                      uint8_t *out_msgs, float *out_scores, uint32_t *out_count);

/* the band of step t (:677-679): writes start/end */
void lva_oracle_band(const lva_oracle_code *c, uint32_t t, uint32_t nblk, uint32_t max_deviation,
                     int band_fma, uint32_t *start, uint32_t *end);

#ifdef __cplusplus
}
#endif
#endif
