// lva_code.h -- host-side description of one (mem_conv, rate, msg_len, orientation) code.
//
// Replaces the globals + set_conv_params() of the reference
// (viterbi/viterbi_convolutional_code.cpp:57-75, :264-415) with a value type, and the
// per-state vectors built inside decode_post_conv_parallel_LVA (:624-650) with two
// compact per-position / per-conv-state tables that the HIP kernels index directly:
//
//   * valid states (is_valid_state :944-978) are a bit test:
//         valid(pos, c)  <=>  (c & vmask[pos]) == vval[pos]
//     because every constraint the reference checks pins one register bit.
//   * predecessors (find_prev_states :860-942): for a target conv state c entered by
//     emitting base b there is AT MOST ONE source conv state ((c << sh) | y) mod 2^m,
//     shared by the flip target b and the flop target b+4 (all four generator
//     polynomials have both end taps set).  predtab[type][c] packs, per base b, a
//     nibble (has << 3) | y.  build() verifies that uniqueness and refuses codes
//     for which it does not hold.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lva {

constexpr uint32_t kMaxPos = 256;   // st_pos2msg_pos[BITSET_SIZE] (:75)
constexpr uint32_t kCrf = 8;        // flip A,C,G,T + flop A,C,G,T (:19-21)

struct Code {
  int mem_conv = 0, rate = 0, rc = 0;
  uint32_t msg_len = 0;
  uint32_t nconv = 0, g[2] = {0, 0}, init = 0, fin = 0;
  int plen = 0;
  uint8_t pattern[16] = {0};
  uint32_t npos = 0;
  uint32_t pos2msg[kMaxPos + 1] = {0};
  uint32_t sync_len = 0, sync_period = 0;
  uint8_t sync[kMaxPos] = {0};

  // derived tables
  uint8_t ptype[kMaxPos] = {0};          // block type of the step INTO pos (0 for pos 0)
  uint32_t vmask[kMaxPos] = {0}, vval[kMaxPos] = {0};
  uint32_t nbits[kMaxPos] = {0};         // message bits consumed on arrival at pos (either orientation)
  uint32_t fpc[kMaxPos][4] = {{0}};      // fingerprint delta of the step into pos, by new bits (lva_device.h)
  std::vector<uint16_t> predtab[4];      // [type][nconv], empty when the rate never uses the type
  std::vector<uint32_t> reach_per_pos;   // structurally reachable (conv,crf) states per position

  uint32_t msg_words() const { return (msg_len + (uint32_t)mem_conv + 31) / 32; }
  uint32_t msg_bits() const { return msg_len + (uint32_t)mem_conv; }
  uint32_t oligo_len() const { return npos - 1; }
  uint32_t shift_of(int type) const { return type == 0 ? 1u : 2u; }
  bool valid(uint32_t pos, uint32_t c) const { return (c & vmask[pos]) == vval[pos]; }

  // conv_output (:440-448)
  uint32_t out_bit(int k, uint32_t st, uint32_t bit) const;
  // conv_encode (:450-499); msg: msg_len values 0/1; bases_out: oligo_len() values 0..3
  int encode(const uint8_t* msg, uint8_t* bases_out) const;
  // band of time step t (:677-679) as the reference binary evaluates it (fused multiply-subtract)
  void band(uint32_t t, uint32_t nblk, uint32_t max_dev, uint32_t* lo, uint32_t* hi) const;
  // algorithmic bytes of one read: sum_t [2*R(t)*L*(4+4W) + 160]
  double algorithmic_bytes(uint32_t nblk, uint32_t list_size, uint32_t max_dev) const;
  // the same sum over the band the kernels work on (the reference's without the positions a path cannot have reached yet and
  // the positions that cannot reach the final one any more): what bench.py calls `frac_moved`
  double working_bytes(uint32_t nblk, uint32_t list_size, uint32_t max_dev) const;
  void working_band(uint32_t t, uint32_t nblk, uint32_t max_dev, uint32_t* lo, uint32_t* hi) const;
};

// set_conv_params (:264-415).  Returns 0 or a negative LVA_ERR_* code (include/lva_decoder.h).
int build_code(Code* c, int mem_conv, int rate, uint32_t msg_len, int rc, const char* sync_marker,
               uint32_t sync_period);

}  // namespace lva
