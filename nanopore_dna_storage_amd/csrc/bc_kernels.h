// bc_kernels.h -- launchers of the kernels in bc_kernels.hip: flappie's flip-flop basecall of the
// posterior matrix and the barcode search on it.
#pragma once
#include <cstdint>

namespace lva {

constexpr int kMaxBarcode = 64;
constexpr uint32_t kBcNone = 0xFFFFFFFFu;      // no window searched
constexpr int32_t kBcInf = 1 << 28;            // "np.inf" of helper.py:179,208 in integer arithmetic

struct BcPatterns {        // 0: start barcode, 1: end barcode, 2: start barcode of the rc orientation, 3: its end barcode
  uint8_t len[4];
  char pat[4][kMaxBarcode];
};

struct BcResult {          // layout of lva_payload_pos (include/lva_decoder.h)
  int32_t start_pos, end_pos, dist_start, dist_end, rc, ok;
};

// 8 lanes per read: Viterbi forward pass, traceback, base / transition-position list.  tb: 8 bytes per block
int launch_bc_basecall(const float* post, const int64_t* row_off, int32_t n_reads, uint32_t* tb, uint8_t* path,
                       char* bases, uint32_t* trans, int32_t* nbases, void* stream);
// best edit-distance match of every pattern; n_orient = 1 (patterns 0,1) or 2 (all four)
int launch_bc_search(const char* bases, const int64_t* base_off, const int32_t* nbases, int32_t n_reads,
                     const BcPatterns& pat, int n_orient, uint32_t* best, void* stream);
// positions in the posterior matrix, orientation choice and length check
int launch_bc_finalize(const uint32_t* trans, const int64_t* base_off, const int32_t* nbases, int32_t n_reads,
                       const BcPatterns& pat, int n_orient, uint32_t min_len, const uint32_t* best, BcResult* out,
                       void* stream);

}  // namespace lva
