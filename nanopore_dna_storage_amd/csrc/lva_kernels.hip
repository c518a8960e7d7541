// lva_kernels.hip -- gfx950 kernels of the list-Viterbi trellis step.
//
// What one launch computes: for every active read slot, every trellis position p in the
// band of that read's time step t, every valid conv state c and every reachable crf state k,
// the new list of (score, message) entries of state (p, c, k) from the previous step's
// lists -- the body of the reference's time loop
// (viterbi/viterbi_convolutional_code.cpp:685-804), with identical results.
//
// Kernels
//   lva_step_exact   one thread per target state, reproduces the reference's list merge
//                    literally (libstdc++ binary heap order, :743-800) -- correct for any
//                    input including exact score ties, -inf posteriors and any list size.
//                    Also used as the fix-up pass behind the fast kernel (worklist mode).
//   lva_init_slot    initial scores (:657-663)
//   lva_gather_final final state's lists -> result record (:806-815)
#include <hip/hip_runtime.h>

#include "lva_device.h"
#include "lva_kernels.h"

namespace lva {

namespace {

constexpr uint32_t kNegInfBits = 0xFF800000u;
constexpr uint32_t kFpPoly = 0x04C11DB7u;   // fingerprint: message as a GF(2) polynomial mod this

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// fingerprint of (msg << 1 | bit) from the fingerprint of msg
__device__ __forceinline__ uint32_t fp_shift1(uint32_t h) { return (h << 1) ^ ((h >> 31) ? kFpPoly : 0u); }
__device__ __forceinline__ uint32_t fp_advance(uint32_t h, uint32_t shift, uint32_t newbits) {
  if (shift >= 1) h = fp_shift1(h);
  if (shift >= 2) h = fp_shift1(h);
  return h ^ newbits;    // shift 0 (stay) has newbits 0
}

struct Target {          // everything the merge of one target state needs
  uint32_t own;          // word offset of (ring(pos), k, l=0, f=0, c) inside a parity buffer
  uint32_t src;          // word offset of (ring(pos-1), crf 0, l=0, f=0, c') inside a parity buffer
  uint32_t k, row;       // target crf state and its row in the posterior block
  uint32_t shift, newbits;
  uint32_t okmask;       // bit i: list i exists (bit 0 = stay)
  uint32_t nlists;       // 8 for a flip target, 2 for a flop target
};

// crf state feeding list i (i >= 1) of target k: flip targets take every other crf state in
// ascending order, flop target X- only takes X+ (:878-885)
__device__ __forceinline__ uint32_t list_crf(uint32_t k, uint32_t i) {
  return k < 4 ? (i - 1) + ((i - 1) >= k ? 1u : 0u) : k - 4;
}

// Resolve the predecessor structure of target (pos, c, k).  Returns false when the target
// is not stored (invalid conv state, or no non-stay predecessor).
__device__ __forceinline__ bool resolve_target(const DevCode& cd, const Geometry& g, const SlotStep& ss,
                                               uint32_t pos, uint32_t c, uint32_t k, Target* tg) {
  if ((c & cd.vmask[pos]) != cd.vval[pos]) return false;               // :700
  tg->k = k;
  tg->row = k >= 4 ? 4u : k;                                           // :582-587
  tg->own = (uint32_t)((pos % g.R) * g.sRing + k * g.sCrf + c);
  const bool stay_ok = pos < ss.prev_hi;   // written at step t-1 (or initialised, t = 0)
  if (pos == 0) {
    tg->src = 0; tg->shift = 0; tg->newbits = 0; tg->nlists = 1; tg->okmask = stay_ok ? 1u : 0u;
    return true;
  }
  const uint32_t T = cd.ptype[pos];
  const uint32_t nib = (cd.predtab[T][c] >> (4 * (k & 3))) & 0xFu;
  if (!(nib & 8)) return false;            // only "stay" leads here: -inf forever, not stored
  const uint32_t sh = T == 0 ? 1u : 2u;
  const uint32_t cp = ((c << sh) | (nib & 7u)) & (cd.nconv - 1);
  const uint32_t newest = c >> (cd.m - 1), second = (c >> (cd.m - 2)) & 1u;
  tg->shift = sh;
  tg->newbits = sh == 1 ? newest : (2 * second + newest);             // :901, :933
  tg->src = (uint32_t)(((pos - 1) % g.R) * g.sRing + cp);
  // which crf states of the source (pos-1, cp) hold data
  uint32_t reach = 0;
  if (((cp & cd.vmask[pos - 1]) == cd.vval[pos - 1]) && (pos - 1 < ss.prev_hi)) {
    if (pos - 1 == 0) reach = 0xFFu;
    else {
      const uint32_t pk = cd.predtab[cd.ptype[pos - 1]][cp];
#pragma unroll
      for (int b = 0; b < 4; ++b) if ((pk >> (4 * b)) & 8u) reach |= (0x11u << b);
    }
  }
  uint32_t ok = stay_ok ? 1u : 0u;
  if (k < 4) {
    tg->nlists = 8;
#pragma unroll
    for (uint32_t i = 1; i < 8; ++i) ok |= ((reach >> list_crf(k, i)) & 1u) << i;
  } else {
    tg->nlists = 2;
    ok |= ((reach >> (k - 4)) & 1u) << 1;
  }
  tg->okmask = ok;
  return true;
}

// ---------------------------------------------------------------------------------------
// Exact merge of one target state, the reference's algorithm verbatim (:706-800).
// ---------------------------------------------------------------------------------------
__device__ void exact_state(const Geometry& g, const SlotStep& ss, const uint32_t* __restrict__ prev,
                            uint32_t* __restrict__ cur, const Target& tg, uint32_t pos) {
  const uint32_t L = g.L, W = g.W;
  const uint32_t sF = (uint32_t)g.sF, sL = (uint32_t)g.sL, sCrf = (uint32_t)g.sCrf;
  const float* post = ss.post_row;
  const float NEG = -INFINITY;

  if (pos == 0) {                                                      // :706-713
    const float s = u2f(prev[tg.own]) + post[tg.row * 8 + tg.k];
    cur[tg.own] = f2u(s);
    for (uint32_t f = 1; f < g.F; ++f) cur[tg.own + f * sF] = prev[tg.own + f * sF];
    for (uint32_t l = 1; l < L; ++l) cur[tg.own + l * sL] = kNegInfBits;
    return;
  }

  // list i: base offset (entry 0, field 0) and additive transition score
  auto list_off = [&](uint32_t i) -> uint32_t { return i == 0 ? tg.own : tg.src + list_crf(tg.k, i) * sCrf; };
  auto list_add = [&](uint32_t i) -> float { return post[tg.row * 8 + (i == 0 ? tg.k : list_crf(tg.k, i))]; };

  if (L == 1) {                                                        // :715-742
    float best = NEG; uint32_t bi = 0;
    for (uint32_t i = 0; i < tg.nlists; ++i) {
      if (!((tg.okmask >> i) & 1u)) continue;
      const float sc = u2f(prev[list_off(i)]) + list_add(i);
      if (sc > best) { best = sc; bi = i; }
    }
    cur[tg.own] = f2u(best);
    if (best != NEG) {
      const uint32_t e = list_off(bi);
      const uint32_t sh = bi == 0 ? 0u : tg.shift, nb = bi == 0 ? 0u : tg.newbits;
      cur[tg.own + sF] = fp_advance(prev[e + sF], sh, nb);
      uint32_t carry = nb;
      for (uint32_t w = 0; w < W; ++w) {
        const uint32_t v = prev[e + (2 + w) * sF];
        cur[tg.own + (2 + w) * sF] = sh ? ((v << sh) | carry) : v;
        carry = sh ? (v >> (32 - sh)) : 0u;
      }
    }
    return;
  }

  // ---- L > 1: k-way merge through a binary heap in libstdc++'s exact order (:743-800) ----
  float hs[8]; uint32_t hx[8];      // heap: score, (list << 16 | index in list)
  int hn = 0;
  for (uint32_t i = 0; i < tg.nlists; ++i) {                           // :750-761
    if (!((tg.okmask >> i) & 1u)) continue;
    const float head = u2f(prev[list_off(i)]);
    if (head != NEG) { hs[hn] = head + list_add(i); hx[hn] = i << 16; ++hn; }
  }
  // GCC 11 bits/stl_heap.h, restated: __push_heap / __adjust_heap / make_heap / pop_heap
  auto sift_up = [&](int hole, int top, float vs, uint32_t vx) {
    int parent = (hole - 1) / 2;
    while (hole > top && hs[parent] < vs) {
      hs[hole] = hs[parent]; hx[hole] = hx[parent];
      hole = parent; parent = (hole - 1) / 2;
    }
    hs[hole] = vs; hx[hole] = vx;
  };
  auto adjust = [&](int hole, int len, float vs, uint32_t vx) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (hs[child] < hs[child - 1]) --child;
      hs[hole] = hs[child]; hx[hole] = hx[child];
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      hs[hole] = hs[child - 1]; hx[hole] = hx[child - 1];
      hole = child - 1;
    }
    sift_up(hole, top, vs, vx);
  };
  if (hn >= 2)                                                         // std::make_heap :762
    for (int parent = (hn - 2) / 2;; --parent) {
      adjust(parent, hn, hs[parent], hx[parent]);
      if (parent == 0) break;
    }

  uint32_t l = 0;
  uint32_t cand[8];
  while (hn > 0 && l < L) {                                            // :764
    // std::pop_heap + back + pop_back (:766-768)
    const float ts = hs[0]; const uint32_t tx = hx[0];
    if (hn > 1) {
      const float vs = hs[hn - 1]; const uint32_t vx = hx[hn - 1];
      adjust(0, hn - 1, vs, vx);
    }
    --hn;
    const uint32_t i = tx >> 16, j = tx & 0xFFFFu;
    const uint32_t e = list_off(i) + j * sL;
    const uint32_t sh = i == 0 ? 0u : tg.shift, nb = i == 0 ? 0u : tg.newbits;
    const uint32_t ch = fp_advance(prev[e + sF], sh, nb);
    uint32_t carry = nb;
    for (uint32_t w = 0; w < W; ++w) {                                 // :771-774
      const uint32_t v = prev[e + (2 + w) * sF];
      cand[w] = sh ? ((v << sh) | carry) : v;
      carry = sh ? (v >> (32 - sh)) : 0u;
    }
    bool dup = false;                                                  // :778-779
    for (uint32_t a = 0; a < l && !dup; ++a) {
      if (cur[tg.own + a * sL + sF] != ch) continue;   // different fingerprint => different message
      bool same = true;
      for (uint32_t w = 0; w < W; ++w) same &= (cur[tg.own + a * sL + (2 + w) * sF] == cand[w]);
      dup = same;
    }
    if (!dup) {                                                        // :780-783
      const uint32_t o = tg.own + l * sL;
      cur[o] = f2u(ts);
      cur[o + sF] = ch;
      for (uint32_t w = 0; w < W; ++w) cur[o + (2 + w) * sF] = cand[w];
      ++l;
    }
    if (j == L - 1) continue;                                          // :788
    const float nxt = u2f(prev[e + sL]);                               // :789
    if (nxt != NEG) {                                                  // :790-796
      hs[hn] = nxt + list_add(i); hx[hn] = (i << 16) | (j + 1);
      sift_up(hn, 0, hs[hn], hx[hn]);
      ++hn;
    }
  }
  for (; l < L; ++l) cur[tg.own + l * sL] = kNegInfBits;               // :799
}

}  // namespace

// grid: x = conv chunks of 64, y = band position index, z = slot index.  block = 256 threads:
// wavefront w handles base w of 64 consecutive conv states: first the flip target w (8-way
// merge), then the flop target w+4 (2-way merge) -- the two share their source conv state.
__global__ __launch_bounds__(256) void lva_step_exact(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                      uint32_t* __restrict__ trellis) {
  const SlotStep& ss = args.s[blockIdx.z];
  const uint32_t pos = ss.lo + blockIdx.y;
  if (pos >= ss.hi) return;
  const uint32_t c = blockIdx.x * 64 + (threadIdx.x & 63);
  const DevCode& cd = codes[ss.orient];
  if (c >= cd.nconv) return;
  const uint32_t b = threadIdx.x >> 6;
  uint32_t* base = trellis + (uint64_t)ss.slot * g.sSlot;
  const uint32_t* prev = base + (uint64_t)(ss.t & 1u) * g.sPar;        // :669-670 swap
  uint32_t* cur = base + (uint64_t)((ss.t + 1) & 1u) * g.sPar;
  Target tg;
  if (resolve_target(cd, g, ss, pos, c, b, &tg)) exact_state(g, ss, prev, cur, tg, pos);
  if (resolve_target(cd, g, ss, pos, c, b + 4, &tg)) exact_state(g, ss, prev, cur, tg, pos);
}

// (:657-663) score 0 at (pos 0, initial conv state, every crf state, list entry 0), empty message
__global__ void lva_init_slot(Geometry g, const DevCode* __restrict__ codes, uint32_t* __restrict__ trellis,
                              uint32_t slot, uint32_t orient) {
  const DevCode& cd = codes[orient];
  uint32_t* par0 = trellis + (uint64_t)slot * g.sSlot;   // parity 0 is "prev" at t = 0
  const uint32_t n = 8 * g.L * g.F;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t f = i % g.F, l = (i / g.F) % g.L, k = i / (g.F * g.L);
    const uint64_t off = (uint64_t)k * g.sCrf + (uint64_t)l * g.sL + (uint64_t)f * g.sF + cd.init;
    par0[off] = (f == 0 && l > 0) ? kNegInfBits : 0u;
  }
}

// copy the lists of (last position, final conv state, crf 0..7) into the result record,
// writing -inf scores for crf states that are not stored (:806-815 reads them as -inf)
__global__ void lva_gather_final(Geometry g, const DevCode* __restrict__ codes, const uint32_t* __restrict__ trellis,
                                 GatherArgs a, uint32_t* __restrict__ results) {
  const DevCode& cd = codes[a.orient];
  const uint32_t pos = cd.npos - 1, c = cd.fin;
  const uint32_t* buf = trellis + (uint64_t)a.slot * g.sSlot + (uint64_t)a.parity * g.sPar;
  uint32_t reach = 0xFFu;
  if (pos > 0) {
    reach = 0;
    const uint32_t pk = cd.predtab[cd.ptype[pos]][c];
    for (int b = 0; b < 4; ++b) if ((pk >> (4 * b)) & 8u) reach |= (0x11u << b);
  }
  if ((c & cd.vmask[pos]) != cd.vval[pos]) reach = 0;
  const uint32_t n = 8 * g.L * g.F;
  uint32_t* out = results + (uint64_t)a.read * n;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t f = i % g.F, l = (i / g.F) % g.L, k = i / (g.F * g.L);
    uint32_t v;
    if ((reach >> k) & 1u)
      v = buf[(uint64_t)(pos % g.R) * g.sRing + (uint64_t)k * g.sCrf + (uint64_t)l * g.sL + (uint64_t)f * g.sF + c];
    else
      v = f == 0 ? kNegInfBits : 0u;
    out[i] = v;
  }
}

// ---------------------------------------------------------------------------------------
// host-callable launchers (no HIP types in the signatures seen by lva_api.cpp's callers)
// ---------------------------------------------------------------------------------------
int launch_step_exact(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, void* stream) {
  if (a.nslots == 0 || a.band_max == 0) return 0;
  dim3 grid((g.N + 63) / 64, a.band_max, a.nslots), block(256);
  hipLaunchKernelGGL(lva_step_exact, grid, block, 0, (hipStream_t)stream, a, g, codes, trellis);
  return (int)hipGetLastError();
}

int launch_init_slot(const Geometry& g, const DevCode* codes, uint32_t* trellis, uint32_t slot, uint32_t orient,
                     void* stream) {
  hipLaunchKernelGGL(lva_init_slot, dim3(1), dim3(64), 0, (hipStream_t)stream, g, codes, trellis, slot, orient);
  return (int)hipGetLastError();
}

int launch_gather_final(const Geometry& g, const DevCode* codes, const uint32_t* trellis, const GatherArgs& a,
                        uint32_t* results, void* stream) {
  hipLaunchKernelGGL(lva_gather_final, dim3(1), dim3(64), 0, (hipStream_t)stream, g, codes, trellis, a, results);
  return (int)hipGetLastError();
}

}  // namespace lva
