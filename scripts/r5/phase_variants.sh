#!/bin/bash
# usage (repo root, no GPU needed): bash scripts/r5/phase_variants.sh -- builds variants/out2.so and variants/merge2.so: lva_step_big_rec with
# the output phase / the merge executed twice (idempotent: results identical), for scripts/r5/phase_counters.sh and the phase-doubling timings.
bash scripts/r4/variant_from_patch.sh out2 '
rep("""  if (npd == 1) rounds(std::integral_constant<uint32_t, 1>{});
  else if (npd == 2) rounds(std::integral_constant<uint32_t, 2>{});
  else rounds(std::integral_constant<uint32_t, 3>{});""", """  for (int rep2 = 0; rep2 < 2; ++rep2) {
  if (npd == 1) rounds(std::integral_constant<uint32_t, 1>{});
  else if (npd == 2) rounds(std::integral_constant<uint32_t, 2>{});
  else rounds(std::integral_constant<uint32_t, 3>{});
  asm volatile("" ::: "memory");
  }""")
'
bash scripts/r4/variant_from_patch.sh merge2 '
rep("""  if (valid) {
    float h[NL]; uint32_t hf[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {                                       // list heads (:750-761)""", """  if (valid) for (int rep2 = 0; rep2 < 2; ++rep2) {
    why = 0; lc = 0; acc_hi = 0; rv0 = 0; rh0 = 0;
    asm volatile("" ::: "memory");
    float h[NL]; uint32_t hf[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {                                       // list heads (:750-761)""")
'
