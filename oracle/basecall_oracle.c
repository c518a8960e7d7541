/*
 * basecall_oracle.c -- CPU oracle for row N3 of SURVEY.md section 8(f): flappie's flip-flop
 * basecall Viterbi on the posterior matrix, the step in front of the list decoder on real data.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as lva_oracle.h).
 *
 * PARITY UNPINNED: flappie's decode path cannot be built in this image -- decode.c links against
 * flappie_matrix.c, which includes <cblas.h> (absent), and the reference holds no golden vectors
 * for it.  This file restates the published algorithm line by line and is checked on
 * hand-computable cases and properties only (tests/test_basecall_oracle.py).
 *
 * Restated:
 *   decode_crf_flipflop   flappie/src/decode.c:119-204  (combine_stays = false, flappie.c:273)
 *   change_positions      flappie/src/decode.c:66-79
 *   basecall / trans file flappie/src/flappie.c:274-285 (base_lookup "ACGT", decode.h:16)
 *   argmaxf               flappie/src/util.c:17-31
 * The matrix is the .post file: float32[nblk][40], row = block (stride 40: flappie pads rows to a
 * multiple of 4, flappie_matrix.c:24-31, and 40 already is one).
 */
#include <stdint.h>
#include <stdlib.h>

#define NBASE 4
#define NSTATE 8

/* returns the number of called bases, or -1 (allocation failure / bad arguments).
 * path: nblk+1 states (0-3 flip, 4-7 flop); bases / trans: capacity nblk. */
int bc_oracle_basecall(const float* post, uint32_t nblk, int32_t* path, char* bases, uint32_t* trans, float* score_out) {
  static const char base_lookup[4] = {'A', 'C', 'G', 'T'};
  if (!post || !path || !bases || !trans || nblk == 0) return -1;
  uint8_t* tb = (uint8_t*)malloc((size_t)nblk * NSTATE);
  if (!tb) return -1;
  float mem[2 * NSTATE] = {0};                                   /* calloc: scores start at 0 (:131) */
  float* curr = mem;
  float* prev = mem + NSTATE;
  for (uint32_t blk = 0; blk < nblk; ++blk) {                    /* forwards pass (:145-183) */
    const float* t = post + (size_t)blk * 40;
    const float* tflop = t + NSTATE * NBASE;
    uint8_t* tbrow = tb + (size_t)blk * NSTATE;
    { float* tmp = curr; curr = prev; prev = tmp; }
    for (int b2 = NBASE; b2 < NSTATE; ++b2) {
      curr[b2] = prev[b2] + tflop[b2];                           /* stay in flop (:157-158) */
      tbrow[b2] = (uint8_t)b2;
      const int from_base = b2 - NBASE;                          /* flip -> flop (:160-165) */
      const float score = prev[from_base] + tflop[from_base];
      if (score > curr[b2]) { curr[b2] = score; tbrow[b2] = (uint8_t)from_base; }
    }
    for (int b1 = 0; b1 < NBASE; ++b1) {                         /* flip states (:169-182) */
      const float* ts = t + b1 * NSTATE;
      curr[b1] = ts[0] + prev[0];
      tbrow[b1] = 0;
      for (int from = 1; from < NSTATE; ++from) {
        const float score = ts[from] + prev[from];
        if (score > curr[b1]) { curr[b1] = score; tbrow[b1] = (uint8_t)from; }
      }
    }
  }
  int imax = 0;                                                  /* traceback (:186-193), argmaxf: first maximum */
  float vmax = curr[0];
  for (int i = 1; i < NSTATE; ++i) if (curr[i] > vmax) { vmax = curr[i]; imax = i; }
  if (score_out) *score_out = vmax;
  path[nblk] = imax;
  for (uint32_t blk = nblk; blk > 0; --blk) path[blk - 1] = tb[(size_t)(blk - 1) * NSTATE + path[blk]];
  free(tb);
  int nch = 0;                                                   /* change_positions(path, nblock, ..) + flappie.c:276-285 */
  for (uint32_t pos = 1; pos < nblk; ++pos) {
    if (path[pos] == path[pos - 1]) continue;
    trans[nch] = pos;
    bases[nch] = base_lookup[path[pos] % NBASE];
    ++nch;
  }
  return nch;
}
