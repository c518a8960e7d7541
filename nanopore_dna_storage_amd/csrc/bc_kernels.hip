// bc_kernels.hip -- gfx950 kernels of SURVEY.md section 8(f) row N3: the step in front of the list
// decoder on real data.  From the same posterior matrix the decoder consumes:
//   bc_basecall   flappie's flip-flop Viterbi basecall (flappie/src/decode.c:119-204 with
//                 combine_stays = false as flappie.c:273 calls it), change_positions (decode.c:66-79)
//                 and the base / position lists flappie writes to the fastq and --trans-output-file
//                 (flappie.c:274-285).  An 8-state recurrence with a dependent chain of nblk steps:
//                 one thread per read, scores in registers, the 8 back-pointers of a block packed into
//                 one 32-bit word.
//   bc_search     helper.find_barcode_pos_in_post's search loops (helper.py:181-191): unit-cost edit
//                 distance of the barcode against every window of the basecall in the allowed half, one
//                 thread per window, first minimum wins.
//   bc_finalize   positions in the posterior matrix (helper.py:192-210), orientation choice and length
//                 check of generate_decoded_lists.py:68-79.
// Integer and fp32-add/compare work only: results are identical to the CPU restatement in oracle/.
#include <hip/hip_runtime.h>

#include "bc_kernels.h"

namespace lva {

__global__ __launch_bounds__(64) void bc_basecall(const float* __restrict__ post, const int64_t* __restrict__ row_off,
                                                  int32_t n_reads, uint32_t* __restrict__ tb, uint8_t* __restrict__ path,
                                                  char* __restrict__ bases, uint32_t* __restrict__ trans,
                                                  int32_t* __restrict__ nbases) {
  const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const int64_t off = row_off[r];
  const uint32_t nblk = (uint32_t)(row_off[r + 1] - off);
  const float4* p = reinterpret_cast<const float4*>(post + off * 40);
  uint32_t* tbr = tb + off;
  uint8_t* pr = path + off + r;              // nblk + 1 states per read
  float prev[8], curr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) prev[s] = 0.0f;                            // calloc (:131)
  for (uint32_t blk = 0; blk < nblk; ++blk) {                             // forwards pass (:145-183)
    float t[40];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const float4 v = p[(size_t)blk * 10 + q];
      t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
    }
    uint32_t back = 0;
#pragma unroll
    for (int b2 = 4; b2 < 8; ++b2) {
      float best = prev[b2] + t[32 + b2];                                 // stay in flop (:157-158)
      uint32_t from = b2;
      const float score = prev[b2 - 4] + t[32 + b2 - 4];                  // flip -> flop (:160-165)
      if (score > best) { best = score; from = b2 - 4; }
      curr[b2] = best; back |= from << (3 * b2);
    }
#pragma unroll
    for (int b1 = 0; b1 < 4; ++b1) {                                      // flip states (:169-182)
      float best = t[b1 * 8] + prev[0];
      uint32_t from = 0;
#pragma unroll
      for (int f = 1; f < 8; ++f) {
        const float score = t[b1 * 8 + f] + prev[f];
        if (score > best) { best = score; from = f; }
      }
      curr[b1] = best; back |= from << (3 * b1);
    }
    tbr[blk] = back;
#pragma unroll
    for (int s = 0; s < 8; ++s) prev[s] = curr[s];
  }
  uint32_t st = 0;                                                        // argmaxf: first maximum (util.c:17-31)
  float vmax = prev[0];
#pragma unroll
  for (int s = 1; s < 8; ++s) if (prev[s] > vmax) { vmax = prev[s]; st = s; }
  if (nblk == 0) { nbases[r] = 0; pr[0] = (uint8_t)st; return; }
  pr[nblk] = (uint8_t)st;                                                 // traceback (:186-193)
  for (uint32_t blk = nblk; blk > 0; --blk) {
    st = (tbr[blk - 1] >> (3 * st)) & 7u;
    pr[blk - 1] = (uint8_t)st;
  }
  int32_t nch = 0;                                                        // change_positions(path, nblock, ..) + flappie.c:276-285
  uint32_t last = pr[0];
  for (uint32_t pos = 1; pos < nblk; ++pos) {
    const uint32_t s = pr[pos];
    if (s != last) {
      trans[off + nch] = pos;
      bases[off + nch] = "ACGT"[s & 3u];
      ++nch;
    }
    last = s;
  }
  nbases[r] = nch;
}

// grid: x = read, y = pattern (0 start, 1 end, 2 start-rc, 3 end-rc); 256 threads = 256 windows at a time
__global__ __launch_bounds__(256) void bc_search(const char* __restrict__ bases, const int64_t* __restrict__ base_off,
                                                 const int32_t* __restrict__ nbases, BcPatterns pat,
                                                 uint32_t* __restrict__ best) {
  __shared__ uint8_t rows[(kMaxBarcode + 1) * 256];      // DP row of thread t: rows[j * 256 + t]
  __shared__ uint32_t red[4];
  const uint32_t r = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
  const int32_t n = nbases[r];
  const int32_t ls = pat.len[w & 2u], le = pat.len[(w & 2u) + 1];
  const int32_t m = pat.len[w];
  const char* txt = bases + base_off[r];
  uint32_t key = kBcNone;
  int32_t lo = 0, hi = 0;                                // windows [lo, hi)
  if (ls + le <= n) {                                    // helper.py:177-179
    if ((w & 1u) == 0) { lo = 0; hi = n / 2 + 1 - ls; }  // :181 start barcode: first half
    else { lo = n / 2; hi = n - le; }                    // :185 end barcode: second half
  }
  for (int32_t i = lo + (int32_t)tid; i < hi; i += 256) {
    // unit-cost edit distance of pat[w] and txt[i .. i+m)   (distance.levenshtein, helper.py:183,187)
    for (int32_t j = 0; j <= m; ++j) rows[j * 256 + tid] = (uint8_t)j;
    for (int32_t a = 1; a <= m; ++a) {
      const char ca = pat.pat[w][a - 1];
      uint32_t diag = rows[tid];                         // D[a-1][0]
      rows[tid] = (uint8_t)a;
      uint32_t left = a;
      for (int32_t j = 1; j <= m; ++j) {
        const uint32_t up = rows[j * 256 + tid];
        const uint32_t sub = diag + (ca != txt[i + j - 1] ? 1u : 0u);
        uint32_t v = up + 1 < left + 1 ? up + 1 : left + 1;
        v = sub < v ? sub : v;
        rows[j * 256 + tid] = (uint8_t)v;
        diag = up; left = v;
      }
    }
    const uint32_t k = ((uint32_t)rows[m * 256 + tid] << 20) | (uint32_t)(i - lo);
    key = k < key ? k : key;                             // smallest distance, then first index (:190-191)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = __shfl_down(key, o);
    key = other < key ? other : key;
  }
  if ((tid & 63u) == 0) red[tid >> 6] = key;
  __syncthreads();
  if (tid == 0) {
    uint32_t k = red[0];
    for (int q = 1; q < 4; ++q) k = red[q] < k ? red[q] : k;
    best[r * 4 + w] = k;
  }
}

__global__ void bc_finalize(const uint32_t* __restrict__ trans, const int64_t* __restrict__ base_off,
                            const int32_t* __restrict__ nbases, int32_t n_reads, BcPatterns pat, int n_orient,
                            uint32_t min_len, const uint32_t* __restrict__ best, BcResult* __restrict__ out) {
  const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const int32_t n = nbases[r];
  const uint32_t* tr = trans + base_off[r];
  int32_t sp[2], ep[2], ds[2], de[2];
  for (int o = 0; o < 2; ++o) {
    sp[o] = -1; ep[o] = -1; ds[o] = kBcInf; de[o] = kBcInf;
    if (o >= n_orient) continue;
    const uint32_t ks = best[r * 4 + 2 * o], ke = best[r * 4 + 2 * o + 1];
    if (ks == kBcNone || ke == kBcNone) continue;        // too short a read (:177-179) or an empty search range
    const int32_t s_first = (int32_t)(ks & 0xFFFFFu);
    const int32_t e_first = n / 2 + (int32_t)(ke & 0xFFFFFu);
    const int32_t s_last = s_first + pat.len[2 * o] - 1;
    const int32_t a = (int32_t)tr[s_last + 1] - 1;       // :193
    const int32_t b = (int32_t)tr[e_first - 1] - 1;      // :194
    if (b < a) continue;                                 // "Barcode removal failure" :206-208
    sp[o] = a; ep[o] = b; ds[o] = (int32_t)(ks >> 20); de[o] = (int32_t)(ke >> 20);
  }
  const int rc = (n_orient == 2 && ds[0] + de[0] > ds[1] + de[1]) ? 1 : 0;    // generate_decoded_lists.py:71
  BcResult res;
  res.start_pos = sp[rc]; res.end_pos = ep[rc];
  res.dist_start = ds[rc] == kBcInf ? 0x7FFFFFFF : ds[rc];
  res.dist_end = de[rc] == kBcInf ? 0x7FFFFFFF : de[rc];
  res.rc = rc;
  res.ok = !(sp[rc] == -1 || (uint32_t)(ep[rc] - sp[rc] + 1) < min_len) ? 1 : 0;   // :76
  out[r] = res;
}

int launch_bc_basecall(const float* post, const int64_t* row_off, int32_t n_reads, uint32_t* tb, uint8_t* path,
                       char* bases, uint32_t* trans, int32_t* nbases, void* stream) {
  if (n_reads <= 0) return 0;
  hipLaunchKernelGGL(bc_basecall, dim3((n_reads + 63) / 64), dim3(64), 0, (hipStream_t)stream, post, row_off, n_reads, tb,
                     path, bases, trans, nbases);
  return (int)hipGetLastError();
}

int launch_bc_search(const char* bases, const int64_t* base_off, const int32_t* nbases, int32_t n_reads,
                     const BcPatterns& pat, int n_orient, uint32_t* best, void* stream) {
  if (n_reads <= 0) return 0;
  hipLaunchKernelGGL(bc_search, dim3(n_reads, 2 * n_orient), dim3(256), 0, (hipStream_t)stream, bases, base_off, nbases,
                     pat, best);
  return (int)hipGetLastError();
}

int launch_bc_finalize(const uint32_t* trans, const int64_t* base_off, const int32_t* nbases, int32_t n_reads,
                       const BcPatterns& pat, int n_orient, uint32_t min_len, const uint32_t* best, BcResult* out,
                       void* stream) {
  if (n_reads <= 0) return 0;
  hipLaunchKernelGGL(bc_finalize, dim3((n_reads + 63) / 64), dim3(64), 0, (hipStream_t)stream, trans, base_off, nbases,
                     n_reads, pat, n_orient, min_len, best, out);
  return (int)hipGetLastError();
}

}  // namespace lva
