// lva_kernels.hip -- gfx950 kernels of the list-Viterbi trellis step.
//
// What one step computes: for every active read slot, every trellis position p in the band
// of that read's time step t, every valid conv state c and every reachable crf state k, the
// new list of (score, message) entries of state (p, c, k) from the previous step's lists --
// the body of the reference's time loop (viterbi/viterbi_convolutional_code.cpp:685-804),
// with identical results.
//
// Kernels
//   lva_step_fast<L,W>  butterfly-tiled fast path.  One workgroup = one (slot, position, tile
//                    of 64 source conv states); the (score, fingerprint) pairs of all source
//                    lists are staged once into LDS with fully coalesced 16-byte loads and
//                    shared by the 64 target conv states x up to 4 bases that consume them.
//                    Each thread merges one flip target (8 lists) and one flop target
//                    (2 lists) with a register tournament over the list heads, de-duplicates
//                    on message fingerprints, then gathers only the surviving messages from
//                    HBM, verifies every fingerprint match on the full message, and writes
//                    coalesced.  Whenever the result could depend on libstdc++'s heap order
//                    (equal scores at the top), on non-finite arithmetic or on a fingerprint
//                    collision, the target is queued for the exact kernel instead.
//   lva_step_exact   one thread per target state, reproduces the reference's list merge
//                    literally (libstdc++ binary heap order, :743-800) -- correct for any
//                    input and any list size.  Grid mode (whole step) or work-list mode
//                    (fix-up pass behind the fast kernel).
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
  uint32_t own;          // word offset of block (ring(pos), k, l=0) inside a parity buffer
  uint32_t src;          // word offset of block (ring(pos-1), crf 0, l=0) inside a parity buffer
  uint32_t c, cp;        // target conv state, source conv state
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

// which crf states of source (pos-1, cp) hold data at this step
__device__ __forceinline__ uint32_t source_reach(const DevCode& cd, const SlotStep& ss, uint32_t pos, uint32_t cp) {
  uint32_t reach = 0;
  if (((cp & cd.vmask[pos - 1]) == cd.vval[pos - 1]) && (pos - 1 < ss.prev_hi)) {
    if (pos - 1 == 0) reach = 0xFFu;
    else {
      const uint32_t pk = cd.predtab[cd.ptype[pos - 1]][cp];
#pragma unroll
      for (int b = 0; b < 4; ++b) if ((pk >> (4 * b)) & 8u) reach |= (0x11u << b);
    }
  }
  return reach;
}

// Resolve the predecessor structure of target (pos, c, k).  Returns false when the target
// is not stored (invalid conv state, or no non-stay predecessor).
__device__ __forceinline__ bool resolve_target(const DevCode& cd, const Geometry& g, const SlotStep& ss,
                                               uint32_t pos, uint32_t c, uint32_t k, Target* tg) {
  if ((c & cd.vmask[pos]) != cd.vval[pos]) return false;               // :700
  tg->k = k; tg->c = c;
  tg->row = k >= 4 ? 4u : k;                                           // :582-587
  tg->own = (uint32_t)(((uint64_t)(pos % g.R) * 8 + k) * g.sCrf);
  const bool stay_ok = pos < ss.prev_hi;   // written at step t-1 (or initialised, t = 0)
  if (pos == 0) {
    tg->src = 0; tg->cp = 0; tg->shift = 0; tg->newbits = 0; tg->nlists = 1; tg->okmask = stay_ok ? 1u : 0u;
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
  tg->cp = cp;
  tg->src = (uint32_t)((uint64_t)((pos - 1) % g.R) * 8 * g.sCrf);
  const uint32_t reach = source_reach(cd, ss, pos, cp);
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
__device__ __noinline__ void exact_state(const Geometry& g, const SlotStep& ss, const uint32_t* __restrict__ prev,
                                         uint32_t* __restrict__ cur, const Target& tg, uint32_t pos) {
  const uint32_t L = g.L, W = g.W, N = g.N, sBlk = g.sBlk, sCrf = (uint32_t)g.sCrf;
  const float* post = ss.post_row;
  const float NEG = -INFINITY;
  // word offsets of entry (list i, index j): SH pair and message
  auto blk_of = [&](uint32_t i, uint32_t j) -> uint32_t {
    return (i == 0 ? tg.own : tg.src + list_crf(tg.k, i) * sCrf) + j * sBlk;
  };
  auto conv_of = [&](uint32_t i) -> uint32_t { return i == 0 ? tg.c : tg.cp; };
  auto list_add = [&](uint32_t i) -> float { return post[tg.row * 8 + (i == 0 ? tg.k : list_crf(tg.k, i))]; };
  const uint32_t own_sh = tg.own + 2 * tg.c, own_msg = tg.own + 2 * N + W * tg.c;

  if (pos == 0) {                                                      // :706-713
    const float s = u2f(prev[own_sh]) + post[tg.row * 8 + tg.k];
    cur[own_sh] = f2u(s);
    cur[own_sh + 1] = prev[own_sh + 1];
    for (uint32_t w = 0; w < W; ++w) cur[own_msg + w] = prev[own_msg + w];
    for (uint32_t l = 1; l < L; ++l) cur[own_sh + l * sBlk] = kNegInfBits;
    return;
  }

  if (L == 1) {                                                        // :715-742
    float best = NEG; uint32_t bi = 0;
    for (uint32_t i = 0; i < tg.nlists; ++i) {
      if (!((tg.okmask >> i) & 1u)) continue;
      const float sc = u2f(prev[blk_of(i, 0) + 2 * conv_of(i)]) + list_add(i);
      if (sc > best) { best = sc; bi = i; }
    }
    cur[own_sh] = f2u(best);
    if (best != NEG) {
      const uint32_t b = blk_of(bi, 0), cv = conv_of(bi);
      const uint32_t sh = bi == 0 ? 0u : tg.shift, nb = bi == 0 ? 0u : tg.newbits;
      cur[own_sh + 1] = fp_advance(prev[b + 2 * cv + 1], sh, nb);
      uint32_t carry = nb;
      for (uint32_t w = 0; w < W; ++w) {
        const uint32_t v = prev[b + 2 * N + W * cv + w];
        cur[own_msg + w] = sh ? ((v << sh) | carry) : v;
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
    const float head = u2f(prev[blk_of(i, 0) + 2 * conv_of(i)]);
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
    const uint32_t b = blk_of(i, j), cv = conv_of(i);
    const uint32_t sh = i == 0 ? 0u : tg.shift, nb = i == 0 ? 0u : tg.newbits;
    const uint32_t ch = fp_advance(prev[b + 2 * cv + 1], sh, nb);
    uint32_t carry = nb;
    for (uint32_t w = 0; w < W; ++w) {                                 // :771-774
      const uint32_t v = prev[b + 2 * N + W * cv + w];
      cand[w] = sh ? ((v << sh) | carry) : v;
      carry = sh ? (v >> (32 - sh)) : 0u;
    }
    bool dup = false;                                                  // :778-779
    for (uint32_t a = 0; a < l && !dup; ++a) {
      if (cur[own_sh + a * sBlk + 1] != ch) continue;   // different fingerprint => different message
      bool same = true;
      for (uint32_t w = 0; w < W; ++w) same &= (cur[own_msg + a * sBlk + w] == cand[w]);
      dup = same;
    }
    if (!dup) {                                                        // :780-783
      cur[own_sh + l * sBlk] = f2u(ts);
      cur[own_sh + l * sBlk + 1] = ch;
      for (uint32_t w = 0; w < W; ++w) cur[own_msg + l * sBlk + w] = cand[w];
      ++l;
    }
    if (j == L - 1) continue;                                          // :788
    const float nxt = u2f(prev[b + sBlk + 2 * cv]);                    // :789
    if (nxt != NEG) {                                                  // :790-796
      hs[hn] = nxt + list_add(i); hx[hn] = (i << 16) | (j + 1);
      sift_up(hn, 0, hs[hn], hx[hn]);
      ++hn;
    }
  }
  for (; l < L; ++l) cur[own_sh + l * sBlk] = kNegInfBits;             // :799
}

__device__ __forceinline__ void slot_buffers(const SlotStep& ss, const Geometry& g, uint32_t* trellis,
                                             const uint32_t** prev, uint32_t** cur) {
  uint32_t* base = trellis + (uint64_t)ss.slot * g.sSlot;
  *prev = base + (uint64_t)(ss.t & 1u) * g.sPar;                       // :669-670 swap
  *cur = base + (uint64_t)((ss.t + 1) & 1u) * g.sPar;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// exact kernel, grid mode.  grid: x = conv chunks of 64, y = band position index, z = slot.
// block = 256 threads: wavefront w handles base w of 64 consecutive conv states: the flip
// target w (8-way merge), then the flop target w+4 (2-way merge).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lva_step_exact(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                      uint32_t* __restrict__ trellis) {
  const SlotStep& ss = args.s[blockIdx.z];
  const uint32_t pos = ss.lo + blockIdx.y;
  if (pos >= ss.hi) return;
  const uint32_t c = blockIdx.x * 64 + (threadIdx.x & 63);
  const DevCode& cd = codes[ss.orient];
  if (c >= cd.nconv) return;
  const uint32_t b = threadIdx.x >> 6;
  const uint32_t* prev; uint32_t* cur;
  slot_buffers(ss, g, trellis, &prev, &cur);
  Target tg;
  if (resolve_target(cd, g, ss, pos, c, b, &tg)) exact_state(g, ss, prev, cur, tg, pos);
  if (resolve_target(cd, g, ss, pos, c, b + 4, &tg)) exact_state(g, ss, prev, cur, tg, pos);
}

// exact kernel, work-list mode (fix-up pass).  item = slotidx<<25 | posidx<<17 | crf<<14 | conv
__global__ __launch_bounds__(256) void lva_step_fixup(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                      uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                      const uint32_t* __restrict__ items) {
  const uint32_t par = args.step_parity;
  const uint32_t n = hdr->count[par];
  const bool all = hdr->overflow[par] != 0;
  if (n == 0 && !all) return;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (gid == 0) atomicAdd(&hdr->total, (unsigned long long)(all ? 0xFFFFFFFFu : n));
  Target tg;
  if (!all) {
    for (uint32_t idx = gid; idx < n; idx += stride) {
      const uint32_t it = items[idx];
      const SlotStep& ss = args.s[it >> 25];
      const uint32_t pos = ss.lo + ((it >> 17) & 0xFFu), k = (it >> 14) & 7u, c = it & 0x3FFFu;
      const uint32_t* prev; uint32_t* cur;
      slot_buffers(ss, g, trellis, &prev, &cur);
      if (resolve_target(codes[ss.orient], g, ss, pos, c, k, &tg)) exact_state(g, ss, prev, cur, tg, pos);
    }
  } else {
    const uint64_t per_slot = (uint64_t)args.band_max * g.N * 8, total = per_slot * args.nslots;
    for (uint64_t idx = gid; idx < total; idx += stride) {
      const uint32_t si = (uint32_t)(idx / per_slot);
      const uint32_t rem = (uint32_t)(idx % per_slot);
      const SlotStep& ss = args.s[si];
      const uint32_t c = rem % g.N, k = (rem / g.N) & 7u, pos = ss.lo + rem / (g.N * 8);
      if (pos >= ss.hi) continue;
      const uint32_t* prev; uint32_t* cur;
      slot_buffers(ss, g, trellis, &prev, &cur);
      if (resolve_target(codes[ss.orient], g, ss, pos, c, k, &tg)) exact_state(g, ss, prev, cur, tg, pos);
    }
  }
}

// ---------------------------------------------------------------------------------------
// fast kernel
// ---------------------------------------------------------------------------------------
namespace {

// Runtime-indexed read of a small register array as a compare/select chain.  The empty asm
// makes each element an opaque value: without it LLVM folds select(load a[i], load a[j]) into a
// load from a selected ADDRESS, which pins the whole array in scratch memory.
template <int N> __device__ __forceinline__ float pick(const float (&a)[N], uint32_t idx) {
  float r = a[0];
  asm("" : "+v"(r));
#pragma unroll
  for (int i = 1; i < N; ++i) {
    float ai = a[i];
    asm("" : "+v"(ai));
    r = idx == (uint32_t)i ? ai : r;
  }
  return r;
}
template <int N> __device__ __forceinline__ uint32_t pick(const uint32_t (&a)[N], uint32_t idx) {
  uint32_t r = a[0];
  asm("" : "+v"(r));
#pragma unroll
  for (int i = 1; i < N; ++i) {
    uint32_t ai = a[i];
    asm("" : "+v"(ai));
    r = idx == (uint32_t)i ? ai : r;
  }
  return r;
}
template <int N, typename T> __device__ __forceinline__ void put(T (&a)[N], uint32_t idx, T v) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = idx == (uint32_t)i ? v : a[i];
}

template <int W> __device__ __forceinline__ void load_msg(const uint32_t* __restrict__ p, uint32_t (&m)[W]) {
#pragma unroll
  for (int w = 0; w < W; w += 2) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + w);
    m[w] = v.x; m[w + 1] = v.y;
  }
}
template <int W> __device__ __forceinline__ void store_msg(uint32_t* __restrict__ p, const uint32_t (&m)[W]) {
#pragma unroll
  for (int w = 0; w < W; w += 2) *reinterpret_cast<uint2*>(p + w) = make_uint2(m[w], m[w + 1]);
}
// m = (m << sh) | nb, sh in {0,1,2}
template <int W> __device__ __forceinline__ void push_bits(uint32_t (&m)[W], uint32_t sh, uint32_t nb) {
  if (sh == 0) return;
  uint32_t carry = nb;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const uint32_t v = m[w];
    m[w] = (v << sh) | carry;
    carry = v >> (32 - sh);
  }
}

// One target state on the fast path.  NL = 8 (flip target) or 2 (flop target).
// s_src: LDS image [crf 8][LL][64] of (score, fingerprint) pairs of the source conv states.
// Returns 0, or the reason (1 tie, 2 non-finite, 3 too many matches, 4 collision) why the
// target must be redone by the exact kernel.
template <int LL, int W, int NL>
__device__ __forceinline__ int fast_merge(const Geometry& g, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur,
                                           const uint2* s_src, const float* s_post, uint32_t k, uint32_t c, uint32_t cp,
                                           uint32_t sc, uint32_t own, uint32_t src, uint32_t okmask, uint32_t sh,
                                           uint32_t nb) {
  const float NEG = -INFINITY;
  const uint32_t sBlk = g.sBlk, sCrf = (uint32_t)g.sCrf, N = g.N;
  const uint32_t row = k >= 4 ? 4u : k;
  const uint32_t own_sh = own + 2 * c, own_msg = own + 2 * N + W * c;
  bool good = true;

  // the target's own ("stay") list lives in registers
  float st_s[LL]; uint32_t st_h[LL];
  if (okmask & 1u) {
#pragma unroll
    for (int l = 0; l < LL; ++l) {
      const uint2 v = *reinterpret_cast<const uint2*>(prev + own_sh + l * sBlk);
      st_s[l] = u2f(v.x); st_h[l] = v.y;
    }
  } else {
#pragma unroll
    for (int l = 0; l < LL; ++l) { st_s[l] = NEG; st_h[l] = 0; }
  }
  // additive transition scores; any non-finite one sends the target to the exact kernel
  float add[NL];
  add[0] = s_post[row * 8 + k];
#pragma unroll
  for (int i = 1; i < NL; ++i) add[i] = s_post[row * 8 + list_crf(k, i)];
#pragma unroll
  for (int i = 0; i < NL; ++i) good &= (fabsf(add[i]) < INFINITY);

  // list heads (:750-761)
  float h[NL]; uint32_t hh[NL];
  h[0] = st_s[0] != NEG ? st_s[0] + add[0] : NEG; hh[0] = st_h[0];
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    const uint2 v = s_src[(list_crf(k, i) * LL + 0) * 64 + sc];
    const bool ok = ((okmask >> i) & 1u) && u2f(v.x) != NEG;
    h[i] = ok ? u2f(v.x) + add[i] : NEG;
    good &= !ok || (h[i] > NEG);           // a finite sum is required of every live head
    hh[i] = fp_advance(v.y, sh, nb);
  }
  good &= !(st_s[0] != NEG) || (h[0] > NEG);

  float as[LL]; uint32_t ah[LL]; uint32_t asrc[LL];
#pragma unroll
  for (int l = 0; l < LL; ++l) { as[l] = NEG; ah[l] = 0; asrc[l] = 0; }
  // fingerprint matches waiting for verification, filed under the accepted entry they matched:
  // two slots of 7 bits (valid, list, index).  A message can sit in at most three lists (stay,
  // flip X, flop X of the base it ends in), so an accepted entry collects at most two.
  uint32_t rejs[LL];
#pragma unroll
  for (int l = 0; l < LL; ++l) rejs[l] = 0;
  uint32_t ptr = 0, lc = 0;

  int why = good ? 0 : 2;
  while (why == 0 && lc < (uint32_t)LL) {                              // :764
    float M = h[0];
#pragma unroll
    for (int i = 1; i < NL; ++i) M = fmaxf(M, h[i]);
    if (!(M > NEG)) break;                 // every list exhausted (heap empty)
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) mask |= (h[i] == M ? 1u : 0u) << i;
    if (mask & (mask - 1)) { why = 1; break; }        // equal scores on top: heap order decides
    const uint32_t sel = __builtin_ctz(mask);
    const uint32_t j = (ptr >> (4 * sel)) & 15u;
    const uint32_t ch = pick<NL>(hh, sel);
    int dup = -1;                                                      // :778-779 on fingerprints
#pragma unroll
    for (int a = 0; a < LL; ++a) dup = ((uint32_t)a < lc && ah[a] == ch) ? a : dup;
    if (dup >= 0) {
      const uint32_t cur_slots = pick<LL>(rejs, (uint32_t)dup);
      if (cur_slots & 0x2000u) { why = 3; break; }   // third match on one entry: cannot all be real
      const uint32_t rec = 0x40u | (sel << 3) | j;
      put<LL>(rejs, (uint32_t)dup, (cur_slots & 0x40u) ? (cur_slots | (rec << 7)) : rec);
    } else {                                                           // :780-783
      put<LL>(as, lc, M); put<LL>(ah, lc, ch); put<LL>(asrc, lc, (sel << 4) | j);
      ++lc;
    }
    // next element of the popped list (:788-796)
    float ns = NEG; uint32_t nh = 0;
    if (j + 1 < (uint32_t)LL) {
      float raw; float a;
      if (sel == 0) {
        // the stay list is consumed front to back: slide it so that its next element is [1]
        // (static register indices only -- a runtime index would send the array to scratch)
        raw = st_s[1]; nh = st_h[1]; a = add[0];
#pragma unroll
        for (int l = 1; l + 1 < LL; ++l) { st_s[l] = st_s[l + 1]; st_h[l] = st_h[l + 1]; }
      } else {
        const uint32_t kk = list_crf(k, sel);
        const uint2 v = s_src[(kk * LL + j + 1) * 64 + sc];
        raw = u2f(v.x); nh = fp_advance(v.y, sh, nb); a = s_post[row * 8 + kk];
      }
      if (raw != NEG) {
        ns = raw + a;
        if (!(ns > NEG)) why = 2;          // overflowed to -inf: the reference would still queue it
      }
    }
    put<NL>(h, sel, ns); put<NL>(hh, sel, nh);
    ptr += 1u << (4 * sel);
  }
  if (why) return why;

  // scores + fingerprints, coalesced per list entry (:781, :799)
#pragma unroll
  for (int l = 0; l < LL; ++l)
    *reinterpret_cast<uint2*>(cur + own_sh + l * sBlk) =
        (uint32_t)l < lc ? make_uint2(f2u(as[l]), ah[l]) : make_uint2(kNegInfBits, 0u);

  // surviving messages: gather from HBM, shift in the new bits, store coalesced (:771-774, :780);
  // every fingerprint match filed under the entry must be the same message, else the exact
  // kernel redoes the target
#pragma unroll
  for (int l = 0; l < LL; ++l) {
    if ((uint32_t)l < lc) {
      const uint32_t i = asrc[l] >> 4, j = asrc[l] & 15u;
      const uint32_t from = i == 0 ? own_msg + j * sBlk
                                   : src + list_crf(k, i) * sCrf + j * sBlk + 2 * N + W * cp;
      uint32_t m[W];
      load_msg<W>(prev + from, m);
      push_bits<W>(m, i == 0 ? 0u : sh, nb);
      store_msg<W>(cur + own_msg + l * sBlk, m);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint32_t rec = (rejs[l] >> (7 * s)) & 0x7Fu;
        if (rec & 0x40u) {
          const uint32_t ri = (rec >> 3) & 7u, rj = rec & 7u;
          const uint32_t rfrom = ri == 0 ? own_msg + rj * sBlk
                                         : src + list_crf(k, ri) * sCrf + rj * sBlk + 2 * N + W * cp;
          uint32_t q[W];
          load_msg<W>(prev + rfrom, q);
          push_bits<W>(q, ri == 0 ? 0u : sh, nb);
#pragma unroll
          for (int w = 0; w < W; ++w) good &= (q[w] == m[w]);
        }
      }
    }
  }
  return good ? 0 : 4;
}

// L == 1: plain add-compare-select, first maximum wins (:715-742).  No heap, no ties issue.
template <int W, int NL>
__device__ __forceinline__ void fast_acs(const Geometry& g, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur,
                                         const uint2* s_src, const float* s_post, uint32_t k, uint32_t c, uint32_t cp,
                                         uint32_t sc, uint32_t own, uint32_t src, uint32_t okmask, uint32_t sh,
                                         uint32_t nb) {
  const float NEG = -INFINITY;
  const uint32_t sCrf = (uint32_t)g.sCrf, N = g.N;
  const uint32_t row = k >= 4 ? 4u : k;
  const uint32_t own_sh = own + 2 * c, own_msg = own + 2 * N + W * c;
  float best = NEG; uint32_t bi = 0, bh = 0;
  if (okmask & 1u) {
    const uint2 v = *reinterpret_cast<const uint2*>(prev + own_sh);
    const float s = u2f(v.x) + s_post[row * 8 + k];
    if (s > best) { best = s; bi = 0; bh = v.y; }
  }
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    if ((okmask >> i) & 1u) {
      const uint2 v = s_src[list_crf(k, i) * 64 + sc];
      const float s = u2f(v.x) + s_post[row * 8 + list_crf(k, i)];
      if (s > best) { best = s; bi = i; bh = fp_advance(v.y, sh, nb); }
    }
  }
  *reinterpret_cast<uint2*>(cur + own_sh) = make_uint2(f2u(best), bh);
  if (best != NEG) {
    const uint32_t from = bi == 0 ? own_msg : src + list_crf(k, bi) * sCrf + 2 * N + W * cp;
    uint32_t m[W];
    load_msg<W>(prev + from, m);
    push_bits<W>(m, bi == 0 ? 0u : sh, nb);
    store_msg<W>(cur + own_msg, m);
  }
}

}  // namespace

// grid: x = tiles of 64 source conv states, y = band position index, z = slot index.
// block = 256 threads = 64 target conv states x up to 4 bases; each thread does the flip
// target of its (conv, base) and then the flop target, which share the source conv state.
template <int LL, int W>
__global__ __launch_bounds__(256) void lva_step_fast(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                     uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                     uint32_t* __restrict__ items) {
  __shared__ uint2 s_src[8 * LL * 64];
  __shared__ float s_post[40];
  const SlotStep& ss = args.s[blockIdx.z];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    hdr->count[args.step_parity ^ 1u] = 0;     // the other parity's list was consumed by the last fix-up
    hdr->overflow[args.step_parity ^ 1u] = 0;
  }
  const uint32_t pos = ss.lo + blockIdx.y;
  if (pos >= ss.hi) return;
  const DevCode& cd = codes[ss.orient];
  const uint32_t N = cd.nconv, tid = threadIdx.x, tile = blockIdx.x;
  const uint32_t* prev; uint32_t* cur;
  slot_buffers(ss, g, trellis, &prev, &cur);

  if (pos == 0) {                          // stay-only update of the 8 start states (:706-713)
    if (tile == cd.init / 64 && tid < 8) {
      const uint32_t k = tid, c = cd.init;
      const uint32_t own = (uint32_t)((uint64_t)k * g.sCrf), own_sh = own + 2 * c, own_msg = own + 2 * N + W * c;
      const float s = u2f(prev[own_sh]) + ss.post_row[(k >= 4 ? 4u : k) * 8 + k];
      cur[own_sh] = f2u(s);
      cur[own_sh + 1] = prev[own_sh + 1];
      for (int w = 0; w < W; ++w) cur[own_msg + w] = prev[own_msg + w];
      for (int l = 1; l < LL; ++l) cur[own_sh + l * g.sBlk] = kNegInfBits;
    }
    return;
  }

  // ---- stage the (score, fingerprint) pairs of 64 source conv states: 8 crf x LL rows of 512 B ----
  const uint32_t src = (uint32_t)((uint64_t)((pos - 1) % g.R) * 8 * g.sCrf);
  {
    const uint32_t lane32 = tid & 31u;
#pragma unroll
    for (int pass = 0; pass < LL; ++pass) {
      const uint32_t rowi = pass * 8 + (tid >> 5);      // = crf * LL + l
      const uint4 v = *reinterpret_cast<const uint4*>(prev + src + (uint64_t)rowi * g.sBlk + 2 * (tile * 64) + 4 * lane32);
      *reinterpret_cast<uint4*>(&s_src[rowi * 64 + 2 * lane32]) = v;
    }
    if (tid < 40) s_post[tid] = ss.post_row[tid];
  }
  __syncthreads();

  // ---- this thread's (target conv, base) ----
  const uint32_t T = cd.ptype[pos], sh = T == 0 ? 1u : 2u;
  const uint32_t Tn = 64u >> sh;                         // target conv states per butterfly leg
  const uint32_t tcl = tid & 63u, r = tid >> 6;
  const uint32_t c = tile * Tn + (tcl & (Tn - 1)) + (tcl >> (6 - sh)) * (N >> sh);
  if ((c & cd.vmask[pos]) != cd.vval[pos]) return;       // :700
  const uint32_t pk = cd.predtab[T][c];
  uint32_t base = r;
  if (T == 0) {                                          // only two bases are reachable: r-th of them
    if (r >= 2) return;
    const uint32_t has = ((pk >> 3) & 1u) | (((pk >> 7) & 1u) << 1) | (((pk >> 11) & 1u) << 2) | (((pk >> 15) & 1u) << 3);
    if (__builtin_popcount(has) <= r) return;
    const uint32_t first = __builtin_ctz(has);
    base = r == 0 ? first : __builtin_ctz(has & ~(1u << first));
  }
  const uint32_t nib = (pk >> (4 * base)) & 0xFu;
  if (!(nib & 8u)) return;
  const uint32_t cp = ((c << sh) | (nib & 7u)) & (N - 1);
  const uint32_t sc = cp - tile * 64;
  const uint32_t newest = c >> (cd.m - 1), second = (c >> (cd.m - 2)) & 1u;
  const uint32_t nb = sh == 1 ? newest : (2 * second + newest);
  const uint32_t reach = source_reach(cd, ss, pos, cp);
  const uint32_t stay_ok = pos < ss.prev_hi ? 1u : 0u;

#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const uint32_t k = base + 4 * half;
    const uint32_t own = (uint32_t)(((uint64_t)(pos % g.R) * 8 + k) * g.sCrf);
    uint32_t ok = stay_ok;
    int why;
    if (half == 0) {
#pragma unroll
      for (uint32_t i = 1; i < 8; ++i) ok |= ((reach >> list_crf(k, i)) & 1u) << i;
      if constexpr (LL == 1) { fast_acs<W, 8>(g, prev, cur, s_src, s_post, k, c, cp, sc, own, src, ok, sh, nb); why = 0; }
      else why = fast_merge<LL, W, 8>(g, prev, cur, s_src, s_post, k, c, cp, sc, own, src, ok, sh, nb);
    } else {
      ok |= ((reach >> base) & 1u) << 1;
      if constexpr (LL == 1) { fast_acs<W, 2>(g, prev, cur, s_src, s_post, k, c, cp, sc, own, src, ok, sh, nb); why = 0; }
      else why = fast_merge<LL, W, 2>(g, prev, cur, s_src, s_post, k, c, cp, sc, own, src, ok, sh, nb);
    }
    if (why) {
      atomicAdd(&hdr->reason[why - 1], 1ull);
      const uint32_t idx = atomicAdd(&hdr->count[args.step_parity], 1u);
      if (idx < hdr->cap) items[idx] = (blockIdx.z << 25) | (blockIdx.y << 17) | (k << 14) | c;
      else hdr->overflow[args.step_parity] = 1u;
    }
  }
}

// (:657-663) score 0 at (pos 0, initial conv state, every crf state, list entry 0), empty message
__global__ void lva_init_slot(Geometry g, const DevCode* __restrict__ codes, uint32_t* __restrict__ trellis,
                              uint32_t slot, uint32_t orient) {
  const DevCode& cd = codes[orient];
  uint32_t* par0 = trellis + (uint64_t)slot * g.sSlot;   // parity 0 is "prev" at t = 0
  const uint32_t n = 8 * g.L * g.F;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t f = i % g.F, l = (i / g.F) % g.L, k = i / (g.F * g.L);
    const uint64_t blk = ((uint64_t)k * g.L + l) * g.sBlk;     // ring slot 0 = position 0
    const uint64_t off = f < 2 ? blk + 2 * cd.init + f : blk + 2 * g.N + (uint64_t)g.W * cd.init + (f - 2);
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
    if ((reach >> k) & 1u) {
      const uint64_t blk = (((uint64_t)(pos % g.R) * 8 + k) * g.L + l) * g.sBlk;
      v = buf[f < 2 ? blk + 2 * c + f : blk + 2 * g.N + (uint64_t)g.W * c + (f - 2)];
    } else {
      v = f == 0 ? kNegInfBits : 0u;
    }
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

bool fast_kernel_available(const Geometry& g) {
  const bool l_ok = g.L == 1 || g.L == 2 || g.L == 4 || g.L == 8;
  const bool w_ok = g.W == 2 || g.W == 4 || g.W == 6 || g.W == 8;
  return l_ok && w_ok && g.N >= 64;
}

template <int LL>
static int launch_fast_w(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, WorkHdr* hdr,
                         uint32_t* items, hipStream_t st) {
  dim3 grid(g.N / 64, a.band_max, a.nslots), block(256);
  switch (g.W) {
    case 2: hipLaunchKernelGGL((lva_step_fast<LL, 2>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 4: hipLaunchKernelGGL((lva_step_fast<LL, 4>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 6: hipLaunchKernelGGL((lva_step_fast<LL, 6>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 8: hipLaunchKernelGGL((lva_step_fast<LL, 8>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    default: return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

int launch_step_fast(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, WorkHdr* hdr,
                     uint32_t* items, void* stream) {
  if (a.nslots == 0 || a.band_max == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int e;
  switch (g.L) {
    case 1: e = launch_fast_w<1>(a, g, codes, trellis, hdr, items, st); break;
    case 2: e = launch_fast_w<2>(a, g, codes, trellis, hdr, items, st); break;
    case 4: e = launch_fast_w<4>(a, g, codes, trellis, hdr, items, st); break;
    case 8: e = launch_fast_w<8>(a, g, codes, trellis, hdr, items, st); break;
    default: return (int)hipErrorInvalidValue;
  }
  if (e) return e;
  if (g.L > 1) {   // fix-up pass: exits at once when the work list is empty
    hipLaunchKernelGGL(lva_step_fixup, dim3(512), dim3(256), 0, st, a, g, codes, trellis, hdr, items);
    e = (int)hipGetLastError();
  }
  return e;
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
