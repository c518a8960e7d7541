// lva_device.h -- structures shared by the host driver (lva_api.cpp) and the HIP kernels.
//
// Trellis memory (HBM), per read slot:
//     word[parity 2][ring R][crf 8][list L][field F][conv N]        (uint32 words, conv fastest)
// field 0 = path score (fp32 bits), field 1 = 32-bit message fingerprint, fields 2..2+W-1 =
// the message so far as W little-endian words.  F = W + 2.
//   * conv fastest: a wavefront of 64 consecutive conv states reads/writes 256 contiguous
//     bytes per (list entry, field) -- every store of the step kernels is fully coalesced.
//   * ring: only positions [band_lo-1, band_hi) of the two parity buffers are live
//     (reference :677-679 band; the extra position below the band carries the reference's
//     stale-score behaviour), so R = min(nstate_pos, 2*max_deviation+1)
//     positions are stored instead of the reference's nstate_pos.
//   * states the reference leaves at -inf forever (invalid conv states :700, and (conv,crf)
//     pairs whose only predecessor is "stay") are neither stored nor read: readers test the
//     same predicates instead.
#pragma once
#include <cstdint>

namespace lva {

constexpr int kMaxSlots = 64;

struct DevCode {                 // one per orientation (0 = forward, 1 = reverse complement)
  uint32_t m, nconv, npos, init, fin;
  uint8_t ptype[256];            // block type of the step into pos
  uint32_t vmask[256], vval[256];
  const uint16_t* predtab[4];    // device pointers, [nconv] each (nullptr when unused)
};

struct SlotStep {                // what one read slot does in one trellis-step launch
  const float* post_row;         // 40 log-posteriors of block t (device)
  uint32_t slot;                 // trellis buffer index
  uint32_t t;
  uint32_t lo, hi;               // band of step t
  uint32_t prev_hi;              // band end of step t-1 (1 at t = 0: only position 0 is initialised)
  uint32_t orient;
};

struct StepArgs {
  uint32_t nslots;
  uint32_t L, W, R;
  uint32_t band_max;             // max over slots of hi-lo
  uint32_t pad;
  SlotStep s[kMaxSlots];
};

struct Geometry {                // strides in words
  uint32_t N, L, W, F, R;
  uint64_t sF, sL, sCrf, sRing, sPar, sSlot;
};

inline Geometry make_geometry(uint32_t N, uint32_t L, uint32_t W, uint32_t R) {
  Geometry g;
  g.N = N; g.L = L; g.W = W; g.F = W + 2; g.R = R;
  g.sF = N; g.sL = (uint64_t)g.F * N; g.sCrf = g.sL * L; g.sRing = g.sCrf * 8;
  g.sPar = g.sRing * R; g.sSlot = g.sPar * 2;
  return g;
}

// final-state gather: result record per read = [crf 8][list L][field F] words
struct GatherArgs {
  uint32_t slot, parity, orient, read;
};

}  // namespace lva
