// bc_kernels.hip -- gfx950 kernels of SURVEY.md section 8(f) row N3: the step in front of the list
// decoder on real data.  From the same posterior matrix the decoder consumes:
//   bc_basecall   flappie's flip-flop Viterbi basecall (flappie/src/decode.c:119-204 with
//                 combine_stays = false as flappie.c:273 calls it), change_positions (decode.c:66-79)
//                 and the base / position lists flappie writes to the fastq and --trans-output-file
//                 (flappie.c:274-285).  An 8-state recurrence with a dependent chain of nblk steps:
//                 8 lanes per read (one per state), 8 reads per wavefront.
//   bc_search     helper.find_barcode_pos_in_post's search loops (helper.py:181-191): unit-cost edit
//                 distance of the barcode against every window of the basecall in the allowed half, one
//                 thread per window (bit-vector recurrence, 64-bit words), first minimum wins.
//   bc_finalize   positions in the posterior matrix (helper.py:192-210), orientation choice and length
//                 check of generate_decoded_lists.py:68-79.
// Integer and fp32-add/compare work only: results are identical to the CPU restatement in oracle/.
#include <hip/hip_runtime.h>

#include "bc_kernels.h"

namespace lva {

// One wavefront = 8 reads x 8 lanes; lane s of a group owns crf state s of its read (0-3 flip, 4-7 flop).
// Per block the 8 scores of a read are exchanged inside the group (ds_bpermute), lane s adds the
// transition weights into its state in flappie's candidate order and keeps the first maximum; its
// back-pointer goes to byte s of the block's 8-byte traceback row.  The posterior rows of the next four
// blocks are in flight while the current four are consumed.
__global__ __launch_bounds__(64) void bc_basecall(const float* __restrict__ post, const int64_t* __restrict__ row_off,
                                                  int32_t n_reads, uint8_t* __restrict__ tb, uint8_t* __restrict__ path,
                                                  char* __restrict__ bases, uint32_t* __restrict__ trans,
                                                  int32_t* __restrict__ nbases) {
  const uint32_t lane = threadIdx.x, s = lane & 7u, grp = lane & ~7u;
  const int32_t r = blockIdx.x * 8 + (int32_t)(lane >> 3);
  const bool live = r < n_reads;
  const int64_t off = live ? row_off[r] : 0;
  const uint32_t nblk = live ? (uint32_t)(row_off[r + 1] - off) : 0u;
  const float* p = post + off * 40;
  uint8_t* tbr = tb + off * 8;               // 8 back-pointers per block
  uint8_t* pr = path + off + (live ? r : 0); // nblk + 1 states per read
  // what lane s reads of a block: a flip state the 8 weights into it (32 contiguous bytes), a flop state
  // the weights of "stay in flop" and "flip -> flop" (decode.c:157-165)
  const bool flip = s < 4;
  auto load_block = [&](uint32_t blk, float (&t)[8]) {
    const float* q = p + (size_t)blk * 40;
    if (flip) {
      const float4 a = *reinterpret_cast<const float4*>(q + s * 8), b = *reinterpret_cast<const float4*>(q + s * 8 + 4);
      t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
    } else {
      t[0] = q[32 + s]; t[1] = q[32 + s - 4];
    }
  };
  float mine = 0.0f;                                                      // calloc (:131)
  constexpr uint32_t D = 4;                  // blocks per chunk: the next chunk's rows are in flight while this one is consumed
  float tn[D][8];
#pragma unroll
  for (uint32_t u = 0; u < D; ++u) {
#pragma unroll
    for (int f = 0; f < 8; ++f) tn[u][f] = 0.0f;
    if (u < nblk) load_block(u, tn[u]);
  }
  for (uint32_t blk0 = 0; __any(blk0 < nblk); blk0 += D) {                // forwards pass (:145-183)
    float tc[D][8];
#pragma unroll
    for (uint32_t u = 0; u < D; ++u) {
#pragma unroll
      for (int f = 0; f < 8; ++f) tc[u][f] = tn[u][f];
      if (blk0 + D + u < nblk) load_block(blk0 + D + u, tn[u]);
    }
#pragma unroll
    for (uint32_t u = 0; u < D; ++u) {
      const uint32_t blk = blk0 + u;
      const bool on = blk < nblk;
      float pv[8];
#pragma unroll
      for (int f = 0; f < 8; ++f) pv[f] = __shfl(mine, (int)(grp | (uint32_t)f));
      float best; uint32_t from;
      if (flip) {                                                         // flip states (:169-182)
        best = tc[u][0] + pv[0]; from = 0;
#pragma unroll
        for (int f = 1; f < 8; ++f) {
          const float score = tc[u][f] + pv[f];
          if (score > best) { best = score; from = f; }
        }
      } else {
        float own = pv[4], below = pv[0];                                 // prev[b2], prev[b2 - nbase]
#pragma unroll
        for (int f = 1; f < 4; ++f) { own = s == 4u + f ? pv[4 + f] : own; below = s == 4u + f ? pv[f] : below; }
        best = own + tc[u][0]; from = s;                                  // stay in flop (:157-158)
        const float score = below + tc[u][1];                             // flip -> flop (:160-165)
        if (score > best) { best = score; from = s - 4; }
      }
      if (on) { mine = best; tbr[(size_t)blk * 8 + s] = (uint8_t)from; }
    }
  }
  // argmaxf: first maximum (util.c:17-31)
  float pv[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) pv[f] = __shfl(mine, (int)(grp | (uint32_t)f));
  uint32_t st = 0;
  float vmax = pv[0];
#pragma unroll
  for (int f = 1; f < 8; ++f) if (pv[f] > vmax) { vmax = pv[f]; st = f; }
  __threadfence_block();
  __syncthreads();                                                        // traceback rows written by the other lanes
  // traceback (:186-193): every lane of the group walks the same chain, lane 0 records it
  if (live && s == 0) pr[nblk] = (uint8_t)st;
  for (uint32_t top = nblk; top > 0;) {                                   // 8 rows in flight, then 8 dependent look-ups
    const uint32_t cnt = top < 8u ? top : 8u;
    uint2 rows[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u)
      rows[u] = u < cnt ? *reinterpret_cast<const uint2*>(tbr + (size_t)(top - 1 - u) * 8) : make_uint2(0u, 0u);
    uint32_t packed = 0;                                                  // states of blocks top-1 .. top-8, 4 bits each
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t w = st < 4 ? rows[u].x : rows[u].y;
      st = u < cnt ? (w >> (8 * (st & 3u))) & 7u : st;
      packed |= st << (4 * u);
    }
    if (s < cnt) pr[top - 1 - s] = (uint8_t)((packed >> (4 * s)) & 7u);   // lane u of the group stores block top-1-u
    top -= cnt;
  }
  __threadfence_block();
  __syncthreads();                                                        // the path written by lane 0 of the group
  // change_positions(path, nblock, ..) + flappie.c:276-285: 8 positions of a read at a time
  const uint32_t gshift = lane & 56u;
  uint32_t nch = 0;
  for (uint32_t base = 1; __any(base < nblk); base += 8) {
    const uint32_t pos = base + s;
    uint32_t cur_s = 0;
    bool flag = false;
    if (pos < nblk) { cur_s = pr[pos]; flag = cur_s != pr[pos - 1]; }
    const uint32_t m8 = (uint32_t)(__ballot(flag) >> gshift) & 0xFFu;
    if (flag) {
      const uint32_t idx = nch + __popc(m8 & ((1u << s) - 1u));
      trans[off + idx] = pos;
      bases[off + idx] = "ACGT"[cur_s & 3u];
    }
    nch += __popc(m8);
  }
  if (live && s == 0) nbases[r] = (int32_t)nch;
}

// grid: x = read, y = pattern (0 start, 1 end, 2 start-rc, 3 end-rc); 256 threads = 256 windows at a time.
// Edit distance of the barcode (m <= 64 characters) against a window: the bit-vector recurrence of Myers in
// Hyyro's global-distance form -- column j of the DP matrix is kept as vertical +1 / -1 delta bit masks, the
// horizontal delta entering row 0 is always +1 (D[0][j] = j), and the score follows bit m-1.  64-bit integer
// arithmetic only: the value is the same D[m][m] the row-by-row DP of distance.levenshtein produces.
__global__ __launch_bounds__(256) void bc_search(const char* __restrict__ bases, const int64_t* __restrict__ base_off,
                                                 const int32_t* __restrict__ nbases, BcPatterns pat,
                                                 uint32_t* __restrict__ best) {
  __shared__ unsigned long long s_peq[4];        // bit i of s_peq[b]: pattern character i is base b
  __shared__ uint32_t red[4];
  const uint32_t r = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
  const int32_t n = nbases[r];
  const int32_t ls = pat.len[w & 2u], le = pat.len[(w & 2u) + 1];
  const int32_t m = pat.len[w];
  const char* txt = bases + base_off[r];
  if (tid < 4) {
    unsigned long long q = 0;
    for (int32_t i = 0; i < m; ++i) q |= (unsigned long long)(pat.pat[w][i] == "ACGT"[tid]) << i;
    s_peq[tid] = q;
  }
  __syncthreads();
  const unsigned long long peq0 = s_peq[0], peq1 = s_peq[1], peq2 = s_peq[2], peq3 = s_peq[3];
  const unsigned long long top = 1ull << (m - 1);
  uint32_t key = kBcNone;
  int32_t lo = 0, hi = 0;                                // windows [lo, hi)
  if (ls + le <= n) {                                    // helper.py:177-179
    if ((w & 1u) == 0) { lo = 0; hi = n / 2 + 1 - ls; }  // :181 start barcode: first half
    else { lo = n / 2; hi = n - le; }                    // :185 end barcode: second half
  }
  for (int32_t i = lo + (int32_t)tid; i < hi; i += 256) {
    // unit-cost edit distance of pat[w] and txt[i .. i+m)   (distance.levenshtein, helper.py:183,187)
    unsigned long long vp = ~0ull, vn = 0ull;
    uint32_t score = (uint32_t)m;
    for (int32_t j = 0; j < m; ++j) {
      const char ch = txt[i + j];
      const unsigned long long eq = ch == 'A' ? peq0 : ch == 'C' ? peq1 : ch == 'G' ? peq2 : ch == 'T' ? peq3 : 0ull;
      const unsigned long long x = eq | vn;
      const unsigned long long d0 = ((vp + (x & vp)) ^ vp) | x;
      const unsigned long long hn = vp & d0;
      const unsigned long long hp = vn | ~(vp | d0);
      score += (hp & top) ? 1u : 0u;
      score -= (hn & top) ? 1u : 0u;
      const unsigned long long xs = (hp << 1) | 1ull;
      vn = xs & d0;
      vp = (hn << 1) | ~(xs | d0);
    }
    const uint32_t k = (score << 20) | (uint32_t)(i - lo);
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
  hipLaunchKernelGGL(bc_basecall, dim3((n_reads + 7) / 8), dim3(64), 0, (hipStream_t)stream, post, row_off, n_reads,
                     reinterpret_cast<uint8_t*>(tb), path, bases, trans, nbases);
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
