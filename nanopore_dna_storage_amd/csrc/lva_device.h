// lva_device.h -- structures shared by the host driver (lva_api.cpp) and the HIP kernels.
//
// Trellis memory (HBM), per read slot:
//     block[parity 2][ring R][crf 8][list L]   each block = N conv states x F words:
//         words [0, 2N)        (score fp32 bits, message fingerprint) pairs, conv-major   "SH"
//         words [2N, 2N + W*N) message-so-far, W little-endian words per conv state       "MSG"
//   F = 2 + W, W = message words rounded up to even (8-byte aligned per-lane accesses).
//   * conv fastest: a wavefront of 64 consecutive conv states reads/writes 512 contiguous
//     bytes of SH and 64*4W contiguous bytes of MSG per list entry -- every store of the step
//     kernels is fully coalesced, and the list-head/score traffic (SH) is separated from the
//     message traffic (MSG), which is only touched for entries that survive the merge.
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
  uint32_t band_max;             // max over slots of hi-lo
  uint32_t step_parity;          // launch counter & 1: selects the work-list counter
  uint32_t pad;
  SlotStep s[kMaxSlots];
};

struct Geometry {                // strides in 32-bit words
  uint32_t N, L, W, F, R;
  uint32_t sBlk;                 // one (ring, crf, list entry) block = N*F
  uint64_t sCrf, sRing, sPar, sSlot;
};

inline Geometry make_geometry(uint32_t N, uint32_t L, uint32_t msg_words, uint32_t R) {
  Geometry g;
  g.N = N; g.L = L; g.W = (msg_words + 1u) & ~1u; g.F = g.W + 2; g.R = R;
  g.sBlk = N * g.F;
  g.sCrf = (uint64_t)g.sBlk * L; g.sRing = g.sCrf * 8;
  g.sPar = g.sRing * R; g.sSlot = g.sPar * 2;
  return g;
}

// Targets the fast kernel could not decide exactly (score ties, non-finite posteriors,
// fingerprint collisions): redone by the exact kernel right after, in the same stream.
struct WorkHdr {
  uint32_t count[2];             // per step parity
  uint32_t overflow[2];          // list full: the exact kernel redoes the whole step
  unsigned long long total;      // states redone since the last reset (profile)
  uint32_t cap;
  uint32_t pad;
  unsigned long long reason[4];  // why: 0 tie on top, 1 non-finite arithmetic, 2 too many fingerprint matches, 3 fingerprint collision
};

// final-state gather: result record per read = [crf 8][list L][field F] words
struct GatherArgs {
  uint32_t slot, parity, orient, read;
};

}  // namespace lva
