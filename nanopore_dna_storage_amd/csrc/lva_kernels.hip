// lva_kernels.hip -- gfx950 kernels of the list-Viterbi trellis step.
//
// What one step computes: for every active read slot, every trellis position p in the band
// of that read's time step t, every valid conv state c and every reachable crf state k, the
// new list of (score, message) entries of state (p, c, k) from the previous step's lists --
// the body of the reference's time loop (viterbi/viterbi_convolutional_code.cpp:685-804),
// with identical results.
//
// Kernels (which one runs: lva_api.cpp lva_decoder_create)
//   lva_step_lazy<L,P,ANCHOR>  list sizes 2 / 4 / 8 (the default there, kernel mode 4).  Butterfly tile: one workgroup =
//                    one (slot, position, tile of 64 source conv states); the (score, fingerprint) pairs of all
//                    source lists are staged once into LDS with coalesced 16-byte loads and shared by the target
//                    conv states x up to 4 bases that consume them.  Wavefronts 0-3 merge the flip targets (8 lists),
//                    4-7 the flop targets (2 lists), one target per thread: a register tournament over the list
//                    heads, de-duplication on message fingerprints (every match verified on the full message).
//                    Messages are materialised on even time steps only ("anchor" instance: two hops to the stored
//                    message); odd steps store one back-pointer byte per entry.  A launch runs one instance: the host
//                    keeps all slots at the same step parity.  Equal scores on top, non-finite arithmetic and
//                    fingerprint collisions queue the target for the exact path.
//   lva_step_fixup_lazy   that exact path: one wavefront per queued target, the reference's heap merge (:743-800)
//                    replayed literally on lane-resident values.
//   lva_step_fast<L,P> + lva_step_fixup   the same tile with messages moved on every step (kernel mode 2 at L = 2/4/8).
//   lva_step_acs<P>  L = 1: plain add-compare-select on the same tile, 256 threads.
//   lva_step_big<LL,P> / lva_step_big_rec<LL> + lva_step_fixup_wave   other list sizes up to 64: list heads read on
//                    demand; plane layout / record layout (Geometry::rec: records of 16, 24 or 32 bytes by trellis position).
//   lva_step_wave    the literal merge with one wavefront per target over the whole step (kernel mode 3).
//   lva_step_exact   one thread per target state, the same literal merge straight from HBM -- any list size; kernel
//                    mode 1, the default above 64 entries, and the overflow path of lva_step_fixup.
//   lva_prepare_step per launch: slot descriptor -> this launch's SlotStep record of every slot
//   lva_init_slot    initial scores (:657-663)
//   lva_gather_final final state's lists -> result record (:806-815)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "lva_device.h"
#include "lva_kernels.h"

namespace lva {

namespace {

constexpr uint32_t kNegInfBits = 0xFF800000u;

// A pointer that went through registers as integers (opqs), or came out of a structure in memory, is a GENERIC pointer to
// the compiler: loads through it are flat_load -- counted in vmcnt AND lgkmcnt, returned out of order, so every use waits
// for all outstanding memory and LDS operations.  Everything these kernels read lives in device memory: say so.
#define LVA_GLOBAL(T, p) ((const __attribute__((address_space(1))) T*)(p))
typedef unsigned int lva_u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a class: copying one out of an
typedef unsigned int lva_u32x2 __attribute__((ext_vector_type(2)));   //  address-space pointer goes through a generic reference)

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }

// An opaque copy of a register value.  Selecting between two elements of a local array,
// `c ? a[i] : a[j]`, is folded by LLVM into a load from a selected ADDRESS, which pins the whole
// array in scratch memory; routing the operands through an empty asm keeps them register values.
template <typename T> __device__ __forceinline__ T opq(T x) { asm("" : "+v"(x)); return x; }
// the same for a value that is uniform over the workgroup: stays in scalar registers
__device__ __forceinline__ uint32_t opqs(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ const uint32_t* opqs(const uint32_t* p) {
  const unsigned long long v = (unsigned long long)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (const uint32_t*)(((unsigned long long)hi << 32) | lo);
}
// c ? a : b on register values (both operands made opaque BEFORE the select: no control flow)
template <typename T> __device__ __forceinline__ T selv(bool c, T a, T b) { a = opq(a); b = opq(b); return c ? a : b; }
// c ? a : b where a and b are already plain values (scalars or registers): nothing to protect
template <typename T> __device__ __forceinline__ T sel(bool c, T a, T b) { return c ? a : b; }

constexpr uint32_t TS_TILE = 64;   // source conv states per tile of the butterfly kernels (lva_step_lazy / fast / acs: TS below)

struct Target {          // everything the merge of one target state needs
  uint32_t own;          // word offset of block (ring(pos), k, l=0) inside a parity buffer
  uint32_t src;          // word offset of block (ring(pos-1), crf 0, l=0) inside a parity buffer
  uint32_t c, cp;        // target conv state, source conv state
  uint32_t k, row;       // target crf state and its row in the posterior block
  uint32_t shift, newbits;
  uint32_t fpc;          // fingerprint delta of a non-stay transition into this target
  uint32_t np_dst, np_src; // message planes in use at pos and at pos-1
  uint32_t okmask;       // bit i: list i exists (bit 0 = stay)
  uint32_t nlists;       // 8 for a flip target, 2 for a flop target
  uint32_t csrc;         // 1: the source position stores compact lists (Geometry::cmp): crf state kk's list is list kk >> 1 there
};

// Compact lists (Geometry::cmp, lazy mode): is `pos` a one-bit position >= 1 (four lists per conv state: crf k -> list k >> 1)?
__device__ __forceinline__ uint32_t compact_pos(const DevCode& cd, const Geometry& g, uint32_t pos) {
  return (g.cmp && pos >= 1 && cd.ptype[pos] == 0) ? 1u : 0u;
}

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
      const uint32_t pk = LVA_GLOBAL(uint16_t, cd.predtab[cd.ptype[pos - 1]])[cp];
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
  tg->own = (uint32_t)(((uint64_t)(pos % g.R) * 8 + (k >> compact_pos(cd, g, pos))) * g.sCrf);
  tg->np_dst = cd.npair[pos];
  tg->csrc = 0;
  const bool stay_ok = pos < ss.prev_hi;   // written at step t-1 (or initialised, t = 0)
  if (pos == 0) {
    tg->src = 0; tg->cp = 0; tg->shift = 0; tg->newbits = 0; tg->fpc = 0; tg->np_src = 1;
    tg->nlists = 1; tg->okmask = stay_ok ? 1u : 0u;
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
  tg->fpc = cd.fpc[pos][tg->newbits];
  tg->np_src = cd.npair[pos - 1];
  tg->cp = cp;
  tg->src = (uint32_t)((uint64_t)((pos - 1) % g.R) * 8 * g.sCrf);
  tg->csrc = compact_pos(cd, g, pos - 1);
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

__device__ __forceinline__ void slot_buffers(const SlotStep& ss, const Geometry& g, uint32_t* trellis,
                                             const uint32_t** prev, uint32_t** cur) {
  uint32_t* base = trellis + (uint64_t)ss.slot * g.sSlot;
  *prev = base + (uint64_t)(ss.t & 1u) * g.sPar;                       // :669-670 swap
  *cur = base + (uint64_t)((ss.t + 1) & 1u) * g.sPar;
}

// This launch's time step of read slot z, or false when the slot takes no part (no read yet, or its read
// has finished).  Uniform per workgroup: one scalar load of the record lva_prepare_step wrote.
__device__ __forceinline__ bool load_slot(const StepArgs& a, uint32_t z, SlotStep* ss) {
  *ss = a.steps[z];
  return ss->t != 0xFFFFFFFFu;
}

// The same with the whole record requested at once: the compiler otherwise sinks the loads of the fields behind the
// early exits that test t and hi, and a workgroup then waits for three scalar loads one after the other.
__device__ __forceinline__ bool load_slot_whole(const StepArgs& a, uint32_t z, SlotStep* ss) {
  static_assert(sizeof(SlotStep) == 48, "twelve words");
  const uint32_t* p = reinterpret_cast<const uint32_t*>(a.steps + z);
  uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4], w5 = p[5], w6 = p[6], w7 = p[7], w8 = p[8], w10 = p[10], w11 = p[11];
  asm volatile("" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3), "+s"(w4), "+s"(w5), "+s"(w6), "+s"(w7), "+s"(w8), "+s"(w10), "+s"(w11));
  ss->post_row = reinterpret_cast<const float*>(((unsigned long long)w1 << 32) | w0);
  ss->slot = w2; ss->t = w3; ss->lo = w4; ss->hi = w5; ss->prev_hi = w6; ss->orient = w7; ss->flags = w8; ss->pad = 0;
  ss->srccmp[0] = w10; ss->srccmp[1] = w11;
  return w3 != 0xFFFFFFFFu;
}

// Does the SOURCE position pos - 1 of band position index y store compact lists (Geometry::cmp)?  From the slot record for the
// first 64 band positions (no further load in front of the staging requests), from the position record beyond (unbanded decodes).
__device__ __forceinline__ uint32_t source_compact(const Geometry& g, const DevCode& cd, const SlotStep& ss, uint32_t y, uint32_t pos) {
  if (!g.cmp) return 0u;
  if (y < 64u) {
    const unsigned long long mask = ((unsigned long long)ss.srccmp[1] << 32) | ss.srccmp[0];   // (no indexing: the record stays in registers)
    return (uint32_t)(mask >> y) & 1u;
  }
  return (cd.rec[pos].cmp3 >> 1) & 1u;
}

// Does tile `tile` of 64 source conv states feed ANY valid target conv state at position pos (:700)?  Only asked at the first and
// last few positions (outside StepArgs::full_lo .. full_hi), where it costs a load of the position record in front of the staging
// requests and saves staging 16-32 KB for nothing: 6.9 % of the benchmark's (position, tile) pairs.
__device__ __forceinline__ bool tile_has_target(const StepArgs& a, const DevCode& cd, uint32_t pos, uint32_t tile) {
  if (pos >= a.full_lo && pos <= a.full_hi) return true;
  const uint32_t info = cd.rec[pos].info, vmask = cd.rec[pos].vmask, vval = cd.rec[pos].vval;
  const uint32_t sh = (info & 0xFFu) == 0 ? 1u : 2u, Tn = TS_TILE >> sh, N = cd.nconv;
  const uint32_t mid = (N - 1u) & ~(Tn - 1u) & ~(((1u << sh) - 1u) << (cd.m - sh));
  return ((tile * Tn) & vmask & mid) == (vval & mid);
}

// XCD-aware tile order (the L = 1 kernel, lva_step_acs).  Workgroups are dealt round-robin over the chip's 8 XCDs in linear block order
// (observed, MI355X_MICROARCH.md; nothing but speed rests on it), the grid's x extent -- the tiles -- is a multiple of 8 wherever
// this is used, so blockIdx.x & 7 labels the XCD.  The tile a workgroup takes carries that label in bits xs .. xs+2 (a bijection
// of the tiles; xs = 0: tile = blockIdx.x), xs per position from the host (PosRec::xs, lva_api.cpp upload_codes): the rows read as
// stay entries by the workgroups at pos and as source entries by those at pos + 1 then meet in one XCD's L2.  Measured (round 6,
// one box, same library): m=11 r=5/6 L=1 443 -> 455 reads/s; the list kernels gain nothing from it (their workgroups reach their
// stay lists 6-10 us after their neighbours staged the same rows: gone from L2 by then; m=14 L=8 -1 %) and keep the plain order.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t x, uint32_t xs) {
  const uint32_t rest = x >> 3, lab = x & 7u;
  return ((rest >> xs) << (xs + 3u)) | (lab << xs) | (rest & ((1u << xs) - 1u));
}

// work-list item: (((slot << 8 | band position index) << 3 | crf) << m) | conv
__device__ __forceinline__ uint32_t make_item(uint32_t m, uint32_t z, uint32_t y, uint32_t k, uint32_t c) {
  return ((((z << 8) | y) << 3 | k) << m) | c;
}

// ---- entry addressing: block base -> plane q of conv state c (2 words) ----
__device__ __forceinline__ uint32_t plane_off(const Geometry& g, uint32_t q, uint32_t c) { return 2 * g.N * q + 2 * c; }

// Message storage of one entry = the 2N*P words after its SH plane.  How the words of conv
// state c are laid out depends on np, the number of 64-bit planes in use at the entry's trellis
// position (uniform over a workgroup), so that a thread moves its message with the widest loads:
//   np == 1          words 0-1 at [2c]                                  (one 8-byte access)
//   np >= 2          words 0-3 at [4c]                                  (one 16-byte access)
//   np == 3          words 4-5 at [4N + 2c]                             (+ one 8-byte access)
//   np == 4          words 4-7 at [4N + 4c]                             (+ one 16-byte access)
__device__ __forceinline__ uint32_t msg_word_off(uint32_t N, uint32_t c, uint32_t w, uint32_t np) {
  if (np == 1) return 2 * c + w;
  if (w < 4) return 4 * c + w;
  return 4 * N + (np == 3 ? 2 * c : 4 * c) + (w - 4);
}
// read message word w (0 = least significant) of an entry whose block base is `blk`; words in
// planes that are not in use at the entry's position are zero
__device__ __forceinline__ uint32_t msg_word(const Geometry& g, const uint32_t* __restrict__ buf, uint32_t blk, uint32_t c,
                                             uint32_t w, uint32_t np) {
  return (w >> 1) < np ? buf[blk + 2 * g.N + msg_word_off(g.N, c, w, np)] : 0u;
}

// Either layout (Geometry::rec, lva_device.h): entry l of conv state c of the list that starts at word `list`.
//   rec_sh    word offset of its (score, fingerprint) pair
//   rec_word  word offset of its message word w
// np = message planes in use at the list's trellis position.  Record layout: a list is [conv][entry] records of 2 + 2 np words --
// score, fingerprint, the 2 np message words in use there, least significant first: 16, 24 or 32 bytes per entry, so that a
// position in the first third of a read moves half the bytes of one in the last third (as the plane layout does with its planes).
__device__ __forceinline__ uint32_t rec_sh(const Geometry& g, uint32_t list, uint32_t c, uint32_t l, uint32_t np) {
  return g.rec ? list + (c * g.L + l) * (2u + 2u * np) : list + l * g.sBlk + 2 * c;
}
__device__ __forceinline__ uint32_t rec_word(const Geometry& g, uint32_t list, uint32_t c, uint32_t l, uint32_t w, uint32_t np) {
  return g.rec ? list + (c * g.L + l) * (2u + 2u * np) + 2u + w : list + l * g.sBlk + 2 * g.N + msg_word_off(g.N, c, w, np);
}
// message word w of that entry; words in planes that are not in use at the entry's position are zero
__device__ __forceinline__ uint32_t entry_word(const Geometry& g, const uint32_t* __restrict__ buf, uint32_t list, uint32_t c, uint32_t l,
                                               uint32_t w, uint32_t np) {
  return (w >> 1) < np ? buf[rec_word(g, list, c, l, w, np)] : 0u;
}

// message of an entry (layout: msg_word_off above).  `ent` = the entry's block base + 2N (start of
// its message region); words in planes >= np are zero and are not read
template <int P> __device__ __forceinline__ void load_msg(const uint32_t* __restrict__ ent, uint32_t N, uint32_t c, uint32_t np,
                                                          uint32_t (&m)[2 * P]) {
#pragma unroll
  for (int w = 0; w < 2 * P; ++w) m[w] = 0;
  if (np == 1) {
    const lva_u32x2 v = *LVA_GLOBAL(lva_u32x2, ent + 2 * c);
    m[0] = v.x; m[1] = v.y;
    return;
  }
  if constexpr (P >= 2) {
    const lva_u32x4 v = *LVA_GLOBAL(lva_u32x4, ent + 4 * c);
    m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w;
    if constexpr (P >= 3) {
      if (np == 3) {
        const lva_u32x2 u = *LVA_GLOBAL(lva_u32x2, ent + 4 * N + 2 * c);
        m[4] = u.x; m[5] = u.y;
      }
    }
    if constexpr (P >= 4) {
      if (np == 4) {
        const lva_u32x4 u = *LVA_GLOBAL(lva_u32x4, ent + 4 * N + 4 * c);
        m[4] = u.x; m[5] = u.y; m[6] = u.z; m[7] = u.w;
      }
    }
  }
}
// the same with the plane count a compile-time constant (callers that branch on a uniform np once, for several entries)
template <int P, int NP> __device__ __forceinline__ void load_msg_np(const uint32_t* __restrict__ ent, uint32_t N, uint32_t c,
                                                                    uint32_t (&m)[2 * P]) {
#pragma unroll
  for (int w = 0; w < 2 * P; ++w) m[w] = 0;
  if constexpr (NP == 1) {
    const lva_u32x2 v = *LVA_GLOBAL(lva_u32x2, ent + 2 * c);
    m[0] = v.x; m[1] = v.y;
  } else if constexpr (P >= 2) {
    const lva_u32x4 v = *LVA_GLOBAL(lva_u32x4, ent + 4 * c);
    m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w;
    if constexpr (NP == 3 && P >= 3) {
      const lva_u32x2 u = *LVA_GLOBAL(lva_u32x2, ent + 4 * N + 2 * c);
      m[4] = u.x; m[5] = u.y;
    }
    if constexpr (NP == 4 && P >= 4) {
      const lva_u32x4 u = *LVA_GLOBAL(lva_u32x4, ent + 4 * N + 4 * c);
      m[4] = u.x; m[5] = u.y; m[6] = u.z; m[7] = u.w;
    }
  }
}
// non-temporal stores: written once, next read by another CU a step later (+4 % measured)
template <int P> __device__ __forceinline__ void store_msg(uint32_t* __restrict__ ent, uint32_t N, uint32_t c, uint32_t np,
                                                           const uint32_t (&m)[2 * P]) {
  if (np == 1) {
    __builtin_nontemporal_store(m[0], ent + 2 * c); __builtin_nontemporal_store(m[1], ent + 2 * c + 1);
    return;
  }
  if constexpr (P >= 2) {
#pragma unroll
    for (int w = 0; w < 4; ++w) __builtin_nontemporal_store(m[w], ent + 4 * c + w);
    if constexpr (P >= 3) {
      if (np == 3) { __builtin_nontemporal_store(m[4], ent + 4 * N + 2 * c); __builtin_nontemporal_store(m[5], ent + 4 * N + 2 * c + 1); }
    }
    if constexpr (P >= 4) {
      if (np == 4) {
#pragma unroll
        for (int w = 0; w < 4; ++w) __builtin_nontemporal_store(m[4 + w], ent + 4 * N + 4 * c + w);
      }
    }
  }
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

// m = (m << sh) | bits with a per-lane sh in 0..31: one funnel shift per word
template <int W> __device__ __forceinline__ void push_var(uint32_t (&m)[W], uint32_t sh, uint32_t bits) {
  const uint32_t back = 32u - sh;
#pragma unroll
  for (int w = W - 1; w >= 1; --w) m[w] = sh ? __builtin_amdgcn_alignbit(m[w], m[w - 1], back) : m[w];
  m[0] = (m[0] << sh) | bits;
}

// ---------------------------------------------------------------------------------------
// Exact merge of one target state by ONE thread, the reference's algorithm verbatim (:706-800).
// ---------------------------------------------------------------------------------------
__device__ __noinline__ void exact_state(const Geometry& g, const SlotStep& ss, const uint32_t* __restrict__ prev,
                                         uint32_t* __restrict__ cur, const Target& tg, uint32_t pos) {
  const uint32_t L = g.L, sBlk = g.sBlk, sCrf = (uint32_t)g.sCrf;
  const uint32_t Wd = 2 * tg.np_dst;      // message words in use at the target
  const float* post = ss.post_row;
  const float NEG = -INFINITY;
  auto blk_of = [&](uint32_t i, uint32_t j) -> uint32_t {
    return (i == 0 ? tg.own : tg.src + list_crf(tg.k, i) * sCrf) + j * sBlk;
  };
  auto conv_of = [&](uint32_t i) -> uint32_t { return i == 0 ? tg.c : tg.cp; };
  auto np_of = [&](uint32_t i) -> uint32_t { return i == 0 ? tg.np_dst : tg.np_src; };
  auto list_add = [&](uint32_t i) -> float { return post[tg.row * 8 + (i == 0 ? tg.k : list_crf(tg.k, i))]; };
  auto own_sh = [&](uint32_t l) -> uint32_t { return tg.own + l * sBlk + 2 * tg.c; };
  auto own_word = [&](uint32_t l, uint32_t w) -> uint32_t {
    return tg.own + l * sBlk + 2 * g.N + msg_word_off(g.N, tg.c, w, tg.np_dst);
  };
  // candidate message of entry (i, j): (msg << shift) | newbits, over Wd words
  auto build = [&](uint32_t i, uint32_t j, uint32_t* out) {
    const uint32_t b = blk_of(i, j), cv = conv_of(i), np = np_of(i);
    const uint32_t sh = i == 0 ? 0u : tg.shift;
    uint32_t carry = i == 0 ? 0u : tg.newbits;
    for (uint32_t w = 0; w < Wd; ++w) {
      const uint32_t v = msg_word(g, prev, b, cv, w, np);
      out[w] = sh ? ((v << sh) | carry) : v;
      carry = sh ? (v >> (32 - sh)) : 0u;
    }
  };

  if (pos == 0) {                                                      // :706-713
    const float s = u2f(prev[own_sh(0)]) + post[tg.row * 8 + tg.k];
    cur[own_sh(0)] = f2u(s);
    cur[own_sh(0) + 1] = prev[own_sh(0) + 1];
    for (uint32_t w = 0; w < Wd; ++w) cur[own_word(0, w)] = prev[own_word(0, w)];
    for (uint32_t l = 1; l < L; ++l) cur[own_sh(l)] = kNegInfBits;
    return;
  }

  uint32_t cand[8];
  if (L == 1) {                                                        // :715-742
    float best = NEG; uint32_t bi = 0;
    for (uint32_t i = 0; i < tg.nlists; ++i) {
      if (!((tg.okmask >> i) & 1u)) continue;
      const float sc = u2f(prev[blk_of(i, 0) + 2 * conv_of(i)]) + list_add(i);
      if (sc > best) { best = sc; bi = i; }
    }
    cur[own_sh(0)] = f2u(best);
    if (best != NEG) {
      cur[own_sh(0) + 1] = prev[blk_of(bi, 0) + 2 * conv_of(bi) + 1] ^ (bi == 0 ? 0u : tg.fpc);
      build(bi, 0, cand);
      for (uint32_t w = 0; w < Wd; ++w) cur[own_word(0, w)] = cand[w];
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
    const uint32_t ch = prev[b + 2 * cv + 1] ^ (i == 0 ? 0u : tg.fpc);
    build(i, j, cand);                                                 // :771-774
    bool dup = false;                                                  // :778-779
    for (uint32_t a = 0; a < l && !dup; ++a) {
      if (cur[own_sh(a) + 1] != ch) continue;   // different fingerprint => different message
      bool same = true;
      for (uint32_t w = 0; w < Wd; ++w) same &= (cur[own_word(a, w)] == cand[w]);
      dup = same;
    }
    if (!dup) {                                                        // :780-783
      cur[own_sh(l)] = f2u(ts);
      cur[own_sh(l) + 1] = ch;
      for (uint32_t w = 0; w < Wd; ++w) cur[own_word(l, w)] = cand[w];
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
  for (; l < L; ++l) cur[own_sh(l)] = kNegInfBits;                     // :799
}

}  // namespace

// ---------------------------------------------------------------------------------------
// exact kernel, grid mode.  grid: x = conv chunks of 64, y = band position index, z = slot.
// block = 256 threads: wavefront w handles base w of 64 consecutive conv states: the flip
// target w (8-way merge), then the flop target w+4 (2-way merge).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lva_step_exact(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                      uint32_t* __restrict__ trellis) {
  SlotStep ss;
  if (!load_slot(args, blockIdx.z, &ss)) return;
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

// The exact merge of ONE target by ONE WAVEFRONT for list sizes 2..8 (the reference's heap merge :743-800 on
// lane-resident values).  tg and pos are wavefront-uniform; all 64 lanes must be active.
__device__ __forceinline__ void fixup_small(const Geometry& g, const SlotStep& ss, const uint32_t* __restrict__ prev,
                                            uint32_t* __restrict__ cur, const Target& tg, uint32_t lane) {
  const uint32_t L = g.L, sBlk = g.sBlk, sCrf = (uint32_t)g.sCrf, k = tg.k;
  const float NEG = -INFINITY;
  // Everything the merge decides on is wavefront-uniform and lives in registers spread over the
  // lanes (candidate (list i, index j) in lane i*8+j, heap element e in lane e, accepted entry a
  // in lane a), read with v_readlane and written with a lane-select: a few cycles per access.
  auto rdf = [](float v, uint32_t ln) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)ln)); };
  auto rdu = [](uint32_t v, uint32_t ln) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)ln); };
  auto wrf = [lane](float& v, uint32_t ln, float x) { v = lane == ln ? x : v; };          // (clang 22 has no writelane builtin)
  auto wru = [lane](uint32_t& v, uint32_t ln, uint32_t x) { v = lane == ln ? x : v; };
  // 1. candidates: lane = list*8 + index; transition score of list i in lane i
  float cs = NEG; uint32_t cy = 0;
  {
    const uint32_t i = lane >> 3, j = lane & 7u;
    if (i < tg.nlists && ((tg.okmask >> i) & 1u) && j < L) {
      const uint32_t b = (i == 0 ? tg.own : tg.src + list_crf(k, i) * sCrf) + j * sBlk + 2 * (i == 0 ? tg.c : tg.cp);
      const uint2 v = *reinterpret_cast<const uint2*>(prev + b);
      cs = u2f(v.x); cy = i != 0 ? v.y ^ tg.fpc : v.y;
    }
  }
  float addv = 0.0f;
  if (lane < tg.nlists) addv = ss.post_row[tg.row * 8 + (lane == 0 ? k : list_crf(k, lane))];
  // word w of the candidate message built from entry (li, lj) -- per lane, for full compares and output
  auto word_of = [&](uint32_t li, uint32_t lj, uint32_t w) -> uint32_t {
    const uint32_t b = (li == 0 ? tg.own : tg.src + list_crf(k, li) * sCrf) + lj * sBlk;
    const uint32_t cv = li == 0 ? tg.c : tg.cp, np = li == 0 ? tg.np_dst : tg.np_src;
    const uint32_t v = msg_word(g, prev, b, cv, w, np);
    if (li == 0) return v;
    const uint32_t lowpart = w == 0 ? tg.newbits : (msg_word(g, prev, b, cv, w - 1, np) >> (32 - tg.shift));
    return (v << tg.shift) | lowpart;
  };
  // 2. the reference merge (:743-800): GCC 11 bits/stl_heap.h restated on lane-resident arrays
  float hs = NEG; uint32_t hx = 0;                       // heap element e in lane e
  float as = NEG; uint32_t ay = 0, ax = 0;               // accepted entry a in lane a: score, fingerprint, source
  auto sift_up = [&](uint32_t hole, uint32_t top, float vs, uint32_t vx) {
    while (hole > top) {
      const uint32_t parent = (hole - 1) / 2;
      const float ps = rdf(hs, parent);
      if (!(ps < vs)) break;
      wrf(hs, hole, ps); wru(hx, hole, rdu(hx, parent));
      hole = parent;
    }
    wrf(hs, hole, vs); wru(hx, hole, vx);
  };
  auto adjust = [&](uint32_t hole, uint32_t len, float vs, uint32_t vx) {
    const uint32_t top = hole;
    uint32_t child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (rdf(hs, child) < rdf(hs, child - 1)) --child;
      wrf(hs, hole, rdf(hs, child)); wru(hx, hole, rdu(hx, child));
      hole = child;
    }
    if ((len & 1u) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      wrf(hs, hole, rdf(hs, child - 1)); wru(hx, hole, rdu(hx, child - 1));
      hole = child - 1;
    }
    sift_up(hole, top, vs, vx);
  };
  uint32_t hn = 0;
  for (uint32_t i = 0; i < tg.nlists; ++i) {                         // :750-761
    const float head = rdf(cs, i * 8);
    if (head != NEG) { wrf(hs, hn, head + rdf(addv, i)); wru(hx, hn, i << 16); ++hn; }
  }
  if (hn >= 2)                                                       // std::make_heap :762
    for (uint32_t parent = (hn - 2) / 2;; --parent) {
      adjust(parent, hn, rdf(hs, parent), rdu(hx, parent));
      if (parent == 0) break;
    }
  uint32_t l = 0;
  const uint32_t Wd = 2 * tg.np_dst;
  while (hn > 0 && l < L) {                                          // :764
    const float ts = rdf(hs, 0); const uint32_t tx = rdu(hx, 0);     // pop_heap + back + pop_back :766-768
    if (hn > 1) adjust(0, hn - 1, rdf(hs, hn - 1), rdu(hx, hn - 1));
    --hn;
    const uint32_t i = tx >> 16, j = tx & 0xFFFFu;
    const uint32_t ch = rdu(cy, i * 8 + j);
    bool dup = false;                                                // :778-779
    unsigned long long match = __ballot(lane < l && ay == ch);      // different fingerprint => different message
    while (match && !dup) {
      const uint32_t a = (uint32_t)__builtin_ctzll(match);
      match &= match - 1;
      const uint32_t asrc = rdu(ax, a);
      uint32_t diff = 0;
      if (lane < Wd) diff = word_of(i, j, lane) ^ word_of(asrc >> 16, asrc & 0xFFFFu, lane);
      dup = __ballot(diff != 0) == 0ull;
    }
    if (!dup) { wrf(as, l, ts); wru(ay, l, ch); wru(ax, l, tx); ++l; }   // :780-783
    if (j == L - 1) continue;                                        // :788
    const float nxt = rdf(cs, i * 8 + j + 1);
    if (nxt != NEG) {                                                // :790-796
      sift_up(hn, 0, nxt + rdf(addv, i), (i << 16) | (j + 1));
      ++hn;
    }
  }
  // 3. outputs: lane = entry*8 + word
  {
    const uint32_t e = lane >> 3, w = lane & 7u;
    const float es = __shfl(as, (int)e);
    const uint32_t ey = __shfl(ay, (int)e), ex = __shfl(ax, (int)e);
    if (e < L && w == 0)
      *reinterpret_cast<uint2*>(cur + tg.own + e * sBlk + 2 * tg.c) = e < l ? make_uint2(f2u(es), ey) : make_uint2(kNegInfBits, 0u);
    if (e < l && w < Wd)
      cur[tg.own + e * sBlk + 2 * g.N + msg_word_off(g.N, tg.c, w, tg.np_dst)] = word_of(ex >> 16, ex & 0xFFFFu, w);
  }
}

// ---------------------------------------------------------------------------------------
// fix-up kernel: exact path over the fast kernel's work list, ONE WAVEFRONT PER TARGET.
// item: make_item().   Requires 2 <= L <= 8.
// All 64 lanes run the same (uniform) merge on register-resident candidate heads (one per lane,
// v_readlane access); loads and stores of the 8x8 candidate entries are spread over the lanes, so
// one target costs a few memory round trips instead of a few hundred.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lva_step_fixup(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                      uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                      const uint32_t* __restrict__ items) {
  const uint32_t par = args.step_parity;
  const uint32_t n = hdr->count[par] < hdr->cap ? hdr->count[par] : hdr->cap;
  const bool all = hdr->overflow[par] != 0;
  if (n == 0 && !all) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (all) atomicAdd(&hdr->overflow_steps, 1u); else atomicAdd(&hdr->total, (unsigned long long)n);
  }
  Target tg;
  if (all) {   // work list overflowed: redo the whole step, one thread per target
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    const uint64_t per_slot = (uint64_t)args.band_max * g.N * 8, total = per_slot * args.nslots;
    for (uint64_t idx = gid; idx < total; idx += stride) {
      const uint32_t si = (uint32_t)(idx / per_slot);
      const uint32_t rem = (uint32_t)(idx % per_slot);
      SlotStep ss;
      if (!load_slot(args, si, &ss)) continue;
      const uint32_t c = rem % g.N, k = (rem / g.N) & 7u, pos = ss.lo + rem / (g.N * 8);
      if (pos >= ss.hi) continue;
      const uint32_t* prev; uint32_t* cur;
      slot_buffers(ss, g, trellis, &prev, &cur);
      if (resolve_target(codes[ss.orient], g, ss, pos, c, k, &tg)) exact_state(g, ss, prev, cur, tg, pos);
    }
    return;
  }

  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t idx = blockIdx.x * 4 + wv; idx < n; idx += nwaves) {
    const uint32_t it = items[idx];
    const uint32_t mm = codes[0].m;
    SlotStep ss;
    if (!load_slot(args, it >> (mm + 11), &ss)) continue;
    const uint32_t pos = ss.lo + ((it >> (mm + 3)) & 0xFFu), k = (it >> mm) & 7u, c = it & ((1u << mm) - 1u);
    const uint32_t* prev; uint32_t* cur;
    slot_buffers(ss, g, trellis, &prev, &cur);
    if (!resolve_target(codes[ss.orient], g, ss, pos, c, k, &tg)) continue;   // (uniform per wavefront)
    if (pos == 0) continue;              // position 0 never reaches the work list
    fixup_small(g, ss, prev, cur, tg, lane);
  }
}

// ---------------------------------------------------------------------------------------
// The literal reference merge of ONE target by ONE WAVEFRONT, list sizes 2 <= L <= 64.  Lane j
// holds entry j of each of the target's <= 8 candidate lists (8 registers), heap element e lives
// in lane e, accepted entry a in lane a; the merge itself is wavefront-uniform (v_readlane), the
// de-duplication scan over the accepted fingerprints is ONE ballot instead of a loop over L
// entries, and all loads/stores of list entries are spread over the lanes.
// ---------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void wave_target(const Geometry& g, const SlotStep& ss, const uint32_t* __restrict__ prev,
                                            uint32_t* __restrict__ cur, const Target& tg, uint32_t pos, uint32_t lane) {
  const uint32_t k = tg.k;
  const uint32_t L = g.L, sBlk = g.sBlk, sCrf = (uint32_t)g.sCrf;
  const uint32_t Wd = 2 * tg.np_dst;
  const float NEG = -INFINITY;
  const uint32_t own_sh = rec_sh(g, tg.own, tg.c, 0, tg.np_dst);
  if (pos == 0) {                                                      // :706-713
    if (lane == 0) {
      cur[own_sh] = f2u(u2f(prev[own_sh]) + ss.post_row[tg.row * 8 + k]);
      cur[own_sh + 1] = prev[own_sh + 1];
    }
    if (lane < Wd) cur[rec_word(g, tg.own, tg.c, 0, lane, tg.np_dst)] = prev[rec_word(g, tg.own, tg.c, 0, lane, tg.np_dst)];
    if (lane >= 1 && lane < L) cur[rec_sh(g, tg.own, tg.c, lane, tg.np_dst)] = kNegInfBits;
    return;
  }
  auto rdf = [](float v, uint32_t ln) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)ln)); };
  auto rdu = [](uint32_t v, uint32_t ln) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)ln); };
  auto wrf = [lane](float& v, uint32_t ln, float x) { v = lane == ln ? x : v; };
  auto wru = [lane](uint32_t& v, uint32_t ln, uint32_t x) { v = lane == ln ? x : v; };
  // 1. candidates: register i of lane j = entry j of list i (score, fingerprint with the step's delta applied)
  float cs[8]; uint32_t cy[8];
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) {
    cs[i] = NEG; cy[i] = 0;
    if (i < tg.nlists && ((tg.okmask >> i) & 1u) && lane < L) {
      const uint32_t b = rec_sh(g, i == 0 ? tg.own : tg.src + (list_crf(k, i) >> tg.csrc) * sCrf, i == 0 ? tg.c : tg.cp, lane, i == 0 ? tg.np_dst : tg.np_src);
      const uint2 v = *reinterpret_cast<const uint2*>(prev + b);
      cs[i] = u2f(v.x); cy[i] = i != 0 ? v.y ^ tg.fpc : v.y;
    }
  }
  auto cand_s = [&](uint32_t i, uint32_t j) -> float {      // i, j wavefront-uniform
    float v = opq(cs[0]);                                   // (opaque copies: a plain select chain is folded into an indexed
#pragma unroll                                              //  load and the array then lives in scratch memory)
    for (uint32_t u = 1; u < 8; ++u) v = i == u ? opq(cs[u]) : v;
    return rdf(v, j);
  };
  auto cand_y = [&](uint32_t i, uint32_t j) -> uint32_t {
    uint32_t v = opq(cy[0]);
#pragma unroll
    for (uint32_t u = 1; u < 8; ++u) v = i == u ? opq(cy[u]) : v;
    return rdu(v, j);
  };
  float addv = 0.0f;                                        // transition score of list i in lane i
  if (lane < tg.nlists) addv = ss.post_row[tg.row * 8 + (lane == 0 ? k : list_crf(k, lane))];
  // word w of the candidate message built from entry (li, lj): per-lane arguments allowed
  auto word_of = [&](uint32_t li, uint32_t lj, uint32_t w) -> uint32_t {
    const uint32_t lst = li == 0 ? tg.own : tg.src + (list_crf(k, li) >> tg.csrc) * sCrf;
    const uint32_t cv = li == 0 ? tg.c : tg.cp, np = li == 0 ? tg.np_dst : tg.np_src;
    const uint32_t v = entry_word(g, prev, lst, cv, lj, w, np);
    if (li == 0) return v;
    const uint32_t lowpart = w == 0 ? tg.newbits : (entry_word(g, prev, lst, cv, lj, w - 1, np) >> (32 - tg.shift));
    return (v << tg.shift) | lowpart;
  };
  // 2. the reference merge (:743-800): GCC 11 bits/stl_heap.h restated on lane-resident arrays
  float hs = NEG; uint32_t hx = 0;                          // heap element e in lane e
  float as = NEG; uint32_t ay = 0, ax = 0;                  // accepted entry a in lane a
  auto sift_up = [&](uint32_t hole, uint32_t top, float vs, uint32_t vx) {
    while (hole > top) {
      const uint32_t parent = (hole - 1) / 2;
      const float ps = rdf(hs, parent);
      if (!(ps < vs)) break;
      wrf(hs, hole, ps); wru(hx, hole, rdu(hx, parent));
      hole = parent;
    }
    wrf(hs, hole, vs); wru(hx, hole, vx);
  };
  auto adjust = [&](uint32_t hole, uint32_t len, float vs, uint32_t vx) {
    const uint32_t top = hole;
    uint32_t child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (rdf(hs, child) < rdf(hs, child - 1)) --child;
      wrf(hs, hole, rdf(hs, child)); wru(hx, hole, rdu(hx, child));
      hole = child;
    }
    if ((len & 1u) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      wrf(hs, hole, rdf(hs, child - 1)); wru(hx, hole, rdu(hx, child - 1));
      hole = child - 1;
    }
    sift_up(hole, top, vs, vx);
  };
  uint32_t hn = 0;
  for (uint32_t i = 0; i < tg.nlists; ++i) {                           // :750-761
    const float head = cand_s(i, 0);
    if (head != NEG) { wrf(hs, hn, head + rdf(addv, i)); wru(hx, hn, i << 16); ++hn; }
  }
  if (hn >= 2)                                                         // std::make_heap :762
    for (uint32_t parent = (hn - 2) / 2;; --parent) {
      adjust(parent, hn, rdf(hs, parent), rdu(hx, parent));
      if (parent == 0) break;
    }
  uint32_t l = 0;
  while (hn > 0 && l < L) {                                            // :764
    const float ts = rdf(hs, 0); const uint32_t tx = rdu(hx, 0);       // pop_heap + back + pop_back :766-768
    if (hn > 1) adjust(0, hn - 1, rdf(hs, hn - 1), rdu(hx, hn - 1));
    --hn;
    const uint32_t i = tx >> 16, j = tx & 0xFFFFu;
    const uint32_t ch = cand_y(i, j);
    bool dup = false;                                                  // :778-779
    unsigned long long match = __ballot(lane < l && ay == ch);        // different fingerprint => different message
    while (match && !dup) {
      const uint32_t a = (uint32_t)__builtin_ctzll(match);
      match &= match - 1;
      const uint32_t asrc = rdu(ax, a);
      uint32_t diff = 0;
      if (lane < Wd) diff = word_of(i, j, lane) ^ word_of(asrc >> 16, asrc & 0xFFFFu, lane);
      dup = __ballot(diff != 0) == 0ull;
    }
    if (!dup) { wrf(as, l, ts); wru(ay, l, ch); wru(ax, l, tx); ++l; }   // :780-783
    if (j == L - 1) continue;                                          // :788
    const float nxt = cand_s(i, j + 1);
    if (nxt != NEG) {                                                  // :790-796
      sift_up(hn, 0, nxt + rdf(addv, i), (i << 16) | (j + 1));
      ++hn;
    }
  }
  // 3. outputs: lane a writes list entry a (:781, :799) and, if accepted, its message
  if (lane < L) {
    *reinterpret_cast<uint2*>(cur + rec_sh(g, tg.own, tg.c, lane, tg.np_dst)) = lane < l ? make_uint2(f2u(as), ay) : make_uint2(kNegInfBits, 0u);
    if (lane < l) {          // the whole message in the widest pieces the layout has (not word by word: every access of a lane is its own line)
      const uint32_t li = ax >> 16, lj = ax & 0xFFFFu;
      const uint32_t lst = li == 0 ? tg.own : tg.src + (list_crf(k, li) >> tg.csrc) * sCrf, cv = li == 0 ? tg.c : tg.cp;
      uint32_t m[8];
      if (g.rec) {           // record layout: the words in use at the entry's position follow its (score, fingerprint) pair
        const uint32_t npi = li == 0 ? tg.np_dst : tg.np_src;
        const uint32_t* rp = prev + rec_sh(g, lst, cv, lj, npi) + 2;
#pragma unroll
        for (uint32_t q2 = 0; q2 < 4; ++q2) {
          lva_u32x2 v = {0u, 0u};
          if (q2 < npi) v = *LVA_GLOBAL(lva_u32x2, rp + 2 * q2);
          m[2 * q2] = v.x; m[2 * q2 + 1] = v.y;
        }
        push_bits<8>(m, li == 0 ? 0u : tg.shift, tg.newbits);
        uint32_t* wp = cur + rec_sh(g, tg.own, tg.c, lane, tg.np_dst) + 2;
#pragma unroll
        for (uint32_t q2 = 0; q2 < 4; ++q2)
          if (q2 < tg.np_dst) *reinterpret_cast<lva_u32x2*>(wp + 2 * q2) = lva_u32x2{m[2 * q2], m[2 * q2 + 1]};
      } else {
        load_msg<4>(prev + lst + lj * sBlk + 2 * g.N, g.N, cv, li == 0 ? tg.np_dst : tg.np_src, m);
        push_bits<8>(m, li == 0 ? 0u : tg.shift, tg.newbits);
        store_msg<4>(cur + tg.own + lane * sBlk + 2 * g.N, g.N, tg.c, tg.np_dst, m);
      }
    }
  }
}
}  // namespace

// wave kernel: wave_target over the whole step -- kernel mode 3.
// grid: x = groups of 4 conv states, y = band position index * 8 + crf state, z = slot.
__global__ __launch_bounds__(256) void lva_step_wave(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                     uint32_t* __restrict__ trellis) {
  SlotStep ss;
  if (!load_slot(args, blockIdx.z, &ss)) return;
  const uint32_t pos = ss.lo + (blockIdx.y >> 3), k = blockIdx.y & 7u;
  if (pos >= ss.hi) return;
  const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const DevCode& cd = codes[ss.orient];
  if (c >= cd.nconv) return;
  const uint32_t* prev; uint32_t* cur;
  slot_buffers(ss, g, trellis, &prev, &cur);
  Target tg;
  if (!resolve_target(cd, g, ss, pos, c, k, &tg)) return;              // (uniform per wavefront)
  wave_target(g, ss, prev, cur, tg, pos, threadIdx.x & 63u);
}

// fix-up pass behind the big-list fast kernel (8 < L <= 64, and list sizes that are not a power of
// two): wave_target over the work list; same item format and overflow behaviour as lva_step_fixup.
__global__ __launch_bounds__(256, 8) void lva_step_fixup_wave(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                           uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                           const uint32_t* __restrict__ items) {
  const uint32_t par = args.step_parity;
  const uint32_t n = hdr->count[par] < hdr->cap ? hdr->count[par] : hdr->cap;
  const bool all = hdr->overflow[par] != 0;
  if (n == 0 && !all) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (all) atomicAdd(&hdr->overflow_steps, 1u); else atomicAdd(&hdr->total, (unsigned long long)n);
  }
  // work list overflowed: the whole step once more on this exact path (every target of every slot; no other code in this
  // kernel -- a call of the thread-per-target routine would cost it 114 registers and its candidate registers a place in scratch memory)
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u, nwaves = gridDim.x * 4;
  const uint32_t mm = codes[0].m;
  const uint32_t nouter = all ? args.nslots : 1u, ninner = all ? (args.band_max << (mm + 3)) : n;   // (N = 2^m conv states, 8 crf states)
  Target tg;
  for (uint32_t si = 0; si < nouter; ++si)
  for (uint32_t idx = blockIdx.x * 4 + wv; idx < ninner; idx += nwaves) {
    const uint32_t it = all ? ((si << (mm + 11)) | idx) : items[idx];      // (uniform per wavefront; make_item's format)
    SlotStep ss;
    if (!load_slot(args, it >> (mm + 11), &ss)) continue;
    const uint32_t pos = ss.lo + ((it >> (mm + 3)) & 0xFFu), k = (it >> mm) & 7u, c = it & ((1u << mm) - 1u);
    if (pos >= ss.hi) continue;            // (whole-step pass: band positions beyond this slot's band)
    const uint32_t* prev; uint32_t* cur;
    slot_buffers(ss, g, trellis, &prev, &cur);
    if (!resolve_target(codes[ss.orient], g, ss, pos, c, k, &tg)) continue;   // (uniform per wavefront)
    wave_target(g, ss, prev, cur, tg, pos, lane);
  }
}

// ---------------------------------------------------------------------------------------
// fast kernel
// ---------------------------------------------------------------------------------------
namespace {

// Tuning constants, each the measured best of its experiment series (DESIGN_HISTORY.md; the rejected variants are kept as
// diffs under scripts/experiments/, not as switches in this file).
constexpr uint32_t TS = TS_TILE;         // source conv states per workgroup tile (workgroup = 8*TS threads); 32 measured 10 % slower
constexpr int kLazyMinWaves = 8;         // both lazy instances held to 64 registers: four 512-thread workgroups per CU (the anchor
                                         // instance's own count is 66: one 8-byte spill outside the merge loop, +1.6 % at m=11)
constexpr uint32_t kFixupLazyGrid = 4096;   // workgroups (of four wavefronts) of lva_step_fixup_lazy
constexpr int kLazyInFlight = 2;         // anchor instance: entries whose message loads are in flight together (3: -3 %, 4: -12 %)



// maximum of three / two floats that are known not to be NaN (the list heads: a NaN sends the target to the exact path before it
// gets here) -- fmaxf would first quiet each operand that comes straight from memory (one more instruction per operand)
__device__ __forceinline__ float max3_nn(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max2_nn(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
constexpr int kRecBits = 8;              // bits per fingerprint-match record in rej0 (valid 0x40 | list << 3 | index), one per accepted entry

// Output phase of fast_merge: gather the surviving messages from HBM, shift in the new bits, store coalesced
// (:771-774, :780); the fingerprint match filed under an entry must be the same message -- false = a collision,
// the exact path redoes the target.
template <int LL, int P>
__device__ __forceinline__ bool fast_output(const Geometry& g, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur, uint32_t k,
                                            uint32_t c, uint32_t cp, uint32_t own, uint32_t src, uint32_t sh, uint32_t nb,
                                            uint32_t np_dst, uint32_t np_src, unsigned long long asrc, unsigned long long rej0,
                                            uint32_t lc) {
  const uint32_t N = g.N, sBlk = g.sBlk, sCrf = mul24(sBlk, LL), pw = 2 * g.N;
  bool good = true;
  constexpr int GB = LL >= 4 ? 4 : LL;       // entries whose loads are in flight together (8: 86 VGPRs, slower)
#pragma unroll
  for (int l0 = 0; l0 < LL; l0 += GB) {
    uint32_t m[GB][2 * P];
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int l = l0 + u;
      if ((uint32_t)l < lc) {
        const uint32_t a8 = (uint32_t)(asrc >> (8 * l)) & 0xFFu;
        const uint32_t i = a8 >> 3, j = a8 & 7u;
        const uint32_t from = i == 0 ? own + mul24(j, sBlk) : src + mul24(list_crf(k, i), sCrf) + mul24(j, sBlk);
        load_msg<P>(prev + from + pw, N, i == 0 ? c : cp, i == 0 ? np_dst : np_src, m[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int l = l0 + u;
      if ((uint32_t)l < lc) {
        const uint32_t a8 = (uint32_t)(asrc >> (8 * l)) & 0xFFu;
        push_bits<2 * P>(m[u], (a8 >> 3) == 0 ? 0u : sh, nb);      // (push_var here: 76 instead of 68 registers with four planes)
        store_msg<P>(cur + own + l * sBlk + pw, N, c, np_dst, m[u]);
        const uint32_t rec = (uint32_t)(rej0 >> (kRecBits * l)) & 0x7Fu;
        if (rec & 0x40u) {
          const uint32_t ri = (rec >> 3) & 7u, rj = rec & 7u;
          const uint32_t rfrom = ri == 0 ? own + mul24(rj, sBlk) : src + mul24(list_crf(k, ri), sCrf) + mul24(rj, sBlk);
          uint32_t qm[2 * P];
          load_msg<P>(prev + rfrom + pw, N, ri == 0 ? c : cp, ri == 0 ? np_dst : np_src, qm);
          push_bits<2 * P>(qm, ri == 0 ? 0u : sh, nb);
#pragma unroll
          for (int w = 0; w < 2 * P; ++w) good &= (qm[w] == m[u][w]);
        }
      }
    }
  }
  return good;
}

// One target state on the fast path.  NL = 8 (flip target) or 2 (flop target).
// s_src: LDS image [crf 8][LL][64] of (score, fingerprint) pairs of the source conv states.
// Returns 0, or the reason (1 tie, 2 non-finite, 3 too many matches, 4 collision) why the
// target must be redone by the exact path.
// The merge proper: decides the new list (scores and fingerprints are stored as it goes) and reports where every accepted
// entry came from (asrc) and which fingerprint matches still have to be verified on the full message (rej0 / rej1).
// crow (uniform): 1 = the staged image has the 4 x LL rows of a compact source position (row of crf state kk = kk >> 1).
template <int LL, int NL>
__device__ __forceinline__ int fast_merge_core(const Geometry& g, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur,
                                                const uint2* s_src, const float* s_post, uint32_t k, uint32_t c,
                                                uint32_t sc, uint32_t own, uint32_t okmask, uint32_t fpc,
                                                unsigned long long* o_asrc, unsigned long long* o_rej0, uint32_t* o_lc,
                                                uint32_t crow = 0) {
  const float NEG = -INFINITY;
  const uint32_t sBlk = g.sBlk;
  const uint32_t row = k >= 4 ? 4u : k;
  const uint32_t own_c = own + 2 * c;                    // + l*sBlk: SH of own entry l
  int why = 0;

  // the target's own ("stay") list lives in registers, transition score already added
  float st_s[LL]; uint32_t st_h[LL];
  const float add0 = s_post[row * 8 + k];
  if (okmask & 1u) {
#pragma unroll
    for (int l = 0; l < LL; ++l) {
      const uint2 v = *reinterpret_cast<const uint2*>(prev + own_c + l * sBlk);
      const float raw = u2f(v.x);
      st_s[l] = raw != NEG ? raw + add0 : NEG;
      if (raw != NEG && !(st_s[l] > NEG)) why = 2;       // non-finite sum: the exact path decides
      st_h[l] = v.y;
    }
  } else {
#pragma unroll
    for (int l = 0; l < LL; ++l) { st_s[l] = NEG; st_h[l] = 0; }
  }

  // list heads (:750-761)
  float h[NL];
  h[0] = st_s[0];
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    const uint32_t kk = list_crf(k, i);
    const uint2 v = s_src[((kk >> crow) * LL) * TS + sc];
    const bool ok = ((okmask >> i) & 1u) && u2f(v.x) != NEG;
    h[i] = ok ? u2f(v.x) + s_post[row * 8 + kk] : NEG;
    if (ok && !(h[i] > NEG)) why = 2;
  }

  // Accepted entries.  ah: their fingerprints, NEWEST FIRST (a shift register: static register indices only) -- bits 31..3 of the
  // fingerprint with the entry's index l in the low three bits, so that ONE unsigned minimum over (ah[a] ^ candidate) says whether
  // an accepted entry has the candidate's fingerprint (minimum < 8: the upper 29 bits agree) and which (the minimum itself).  A
  // match on 29 bits that is none on 32 is confirmed on the full messages like every other and fails there (reason 4).  Empty
  // slots hold all ones: a candidate whose upper 29 bits are all ones would match them and goes to the exact path (reason 3;
  // 2^-29 of the pops -- NOT zero, the fingerprint of every all-zero message).
  // asrc: 8 bits (list << 3 | index) per entry in acceptance order.  rej0: the fingerprint match waiting for verification, filed
  // under the accepted entry it matched, kRecBits bits (valid 0x40, list, index) per entry -- ONE per entry (see below).
  constexpr uint32_t kBlank = 0xFFFFFFF8u;
  uint32_t ah[LL];
#pragma unroll
  for (int l = 0; l < LL; ++l) ah[l] = kBlank;
  unsigned long long asrc = 0, rej0 = 0;
  uint32_t ptr = 0, lc = 0;

  // The loop body is written branch-free (selects) except for the store of an accepted entry:
  // lanes that are done keep running harmless iterations until the wavefront's last lane exits.
  bool go = why == 0;
  while (go) {                                                         // :764
    float M;
    if constexpr (NL == 8) M = max2_nn(max3_nn(max3_nn(max3_nn(h[0], h[1], h[2]), h[3], h[4]), h[5], h[6]), h[7]);
    else M = max2_nn(h[0], h[1]);
    bool eq[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) eq[i] = h[i] == M;
    // first and last head equal to the maximum; they differ when equal scores sit on top, and
    // then libstdc++'s heap order decides
    uint32_t sel = NL - 1;
#pragma unroll
    for (int i = NL - 2; i >= 0; --i) sel = eq[i] ? (uint32_t)i : sel;
    uint32_t last = 0;                     // (the same test as lane-mask logic on the scalar unit: 0.4 % slower)
#pragma unroll
    for (int i = 1; i < NL; ++i) last = eq[i] ? (uint32_t)i : last;
    const bool two = sel != last;
    const bool alive = M > NEG;            // false: every list exhausted (heap empty)
    const bool proceed = alive && !two;
    const uint32_t j = (ptr >> (4 * sel)) & 15u;
    // source side, computed for every lane (a stay pop reads list 1's slot harmlessly)
    const uint32_t kk = list_crf(k, sel == 0 ? 1u : sel);
    const uint32_t at = (mul24(kk >> crow, LL) + j) * TS + sc;
    const bool has_next = j + 1 < (uint32_t)LL;
    const uint32_t fp_src = s_src[at].y ^ fpc;
    const float raw1 = u2f(s_src[has_next ? at + TS : at].x);
    const bool nxt_ok = has_next && raw1 != NEG;
    const float addk = s_post[row * 8 + kk];                           // unconditional: no branch around one LDS read
    const float ns_src = nxt_ok ? raw1 + addk : NEG;                   // :788-796
    const bool is_stay = sel == 0;
    const bool bad = !is_stay && nxt_ok && !(ns_src > NEG);   // overflowed to -inf: the reference would still queue it
    const uint32_t ch = selv(is_stay, st_h[0], fp_src);
    const float ns = selv(is_stay, LL > 1 ? st_s[LL > 1 ? 1 : 0] : NEG, ns_src);
    // the stay list is consumed front to back: slide it when it was popped.  A real branch on purpose (predicated
    // v_mov): a run of VOP2 v_cndmask on one condition issues at ~23 cycles each on gfx950 (scripts/ubench/valu_rates.hip)
    if (is_stay) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int l = 0; l + 1 < LL; ++l) { st_s[l] = opq(st_s[l + 1]); st_h[l] = opq(st_h[l + 1]); }
      st_s[LL - 1] = NEG;
    }
    // de-duplicate on fingerprints (:778-779)
    const uint32_t chk = ch & ~7u;
    uint32_t mn = ah[0] ^ chk;
#pragma unroll
    for (int a = 1; a < LL; ++a) mn = min(mn, ah[a] ^ chk);            // (v_min3_u32)
    const bool blank = chk == kBlank;
    const bool isdup = mn < 8u;                                        // mn = index of the accepted entry with this fingerprint
    const bool accept = proceed && !isdup && !blank, reject = proceed && isdup && !blank;
    // Practically every duplicate pairs an entry of the stay list with an entry of ONE source list: the source lists of a
    // target hardly ever share a message (instrumented oracle, scripts/merge_stats.py: 4-6 in 100 000 duplicate pops at
    // m = 6 / 8 / 11, clean and noisy), so an accepted entry has one match at most.  A second one -- a message in three
    // lists, or a fingerprint collision -- is reason 3: the exact path decides.  Two records filed under one entry merge their
    // valid bits; the count of valid bits is compared with the count of rejected candidates behind the loop.
    const uint32_t rec = reject ? (0x40u | (sel << 3) | j) : 0u;
    rej0 |= (unsigned long long)rec << ((mn << 3) & 56u);
    if (accept) {
      *reinterpret_cast<uint2*>(cur + own_c + mul24(lc, sBlk)) = make_uint2(f2u(M), ch);   // :780-783
#pragma unroll
      for (int a = LL - 1; a >= 1; --a) ah[a] = opq(ah[a - 1]);
      ah[0] = chk | lc;
      asrc |= (unsigned long long)((sel << 3) | j) << (8 * lc);
      lc += 1u;
    }
    // advance the popped list
#pragma unroll
    for (int i = 0; i < NL; ++i) h[i] = selv(eq[i], ns, h[i]);
    ptr += 1u << (4 * sel);
    why = (alive && two) ? 1 : ((proceed && bad) ? 2 : ((proceed && blank) ? 3 : 0));
    go = proceed && why == 0 && lc < (uint32_t)LL;
  }
  if (why) return why;
  {
    // candidates popped = the list pointers' sum: accepted + rejected (+ the one pop that found every list exhausted, when the
    // list is not full); every rejected candidate must have its own record
    const uint32_t t4 = (ptr & 0x0F0F0F0Fu) + ((ptr >> 4) & 0x0F0F0F0Fu);
    const uint32_t pops = (t4 * 0x01010101u) >> 24;
    const uint32_t nrej = pops - lc - (lc < (uint32_t)LL ? 1u : 0u);
    if ((uint32_t)__builtin_popcountll(rej0 & 0x4040404040404040ull) != nrej) return 3;
  }

  // unused tail of the list (:799)
#pragma unroll
  for (int l = 0; l < LL; ++l)
    if ((uint32_t)l >= lc) *reinterpret_cast<uint2*>(cur + own_c + l * sBlk) = make_uint2(kNegInfBits, 0u);
  *o_asrc = asrc; *o_rej0 = rej0; *o_lc = lc;
  return 0;
}

template <int LL, int P, int NL>
__device__ __forceinline__ int fast_merge(const Geometry& g, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur,
                                           const uint2* s_src, const float* s_post, uint32_t k, uint32_t c, uint32_t cp,
                                           uint32_t sc, uint32_t own, uint32_t src, uint32_t okmask, uint32_t sh,
                                           uint32_t nb, uint32_t fpc, uint32_t np_dst, uint32_t np_src) {
  unsigned long long asrc, rej0;
  uint32_t lc;
  const int why = fast_merge_core<LL, NL>(g, prev, cur, s_src, s_post, k, c, sc, own, okmask, fpc, &asrc, &rej0, &lc);
  if (why) return why;
  return fast_output<LL, P>(g, prev, cur, k, c, cp, own, src, sh, nb, np_dst, np_src, asrc, rej0, lc) ? 0 : 4;
}

// The butterfly: thread tid of the workgroup that owns source tile `tile` (TSx consecutive source
// conv states at pos-1) -> its target state at pos.  Thread = (role, base r, target conv): role 0
// merges the flip target of its (conv, base), role 1 the flop target.  False: nothing to do.
struct TileTarget {
  uint32_t c, cp, sc;            // target conv, source conv, source conv relative to the tile
  uint32_t k, sh, nb, fpc;       // target crf state; message shift, new bits and fingerprint delta of the step
  uint32_t np_dst, np_src;
  uint32_t ok;                   // bit i: list i exists (bit 0 = stay)
  uint32_t own;                  // word offset of block (ring(pos), k, l=0)
  uint32_t base, reach;          // the base the target ends in; crf states of the source conv state that are stored
  uint32_t pk1;                  // predecessor-table word of the source conv state at pos-1 (0 where reach did not need it)
  uint32_t c0;                   // 1: pos stores compact lists (crf k -> list k >> 1); `own` already names the target's list
};
// compact-list flags of a position record, masked by the geometry (bit 0: pos, bit 1: pos-1, bit 2: pos-2)
__device__ __forceinline__ uint32_t cmp_bits(const Geometry& g, const PosRec& pr) { return g.cmp ? pr.cmp3 : 0u; }
template <uint32_t TSx>
__device__ __forceinline__ bool tile_target(const DevCode& cd, const Geometry& g, const SlotStep& ss, uint32_t pos,
                                            uint32_t tile, uint32_t tid, TileTarget* t) {
  const uint32_t N = cd.nconv;
  const PosRec pr = cd.rec[pos];                         // (uniform: scalar registers)
  const uint32_t T = pr.info & 0xFFu, sh = T == 0 ? 1u : 2u;
  const uint32_t Tn = TSx >> sh;                         // target conv states per butterfly leg
  const uint32_t role = tid / (4 * TSx), r = (tid / TSx) & 3u, tcl = tid % TSx;
  const uint32_t c = tile * Tn + (tcl & (Tn - 1)) + (tcl / Tn) * (N >> sh);
  if ((c & pr.vmask) != pr.vval) return false;           // :700
  const uint32_t pk = LVA_GLOBAL(uint16_t, pr.pred)[c];
  uint32_t base = r;
  if (T == 0) {                                          // only two bases are reachable: r-th of them
    if (r >= 2) return false;
    const uint32_t has = ((pk >> 3) & 1u) | (((pk >> 7) & 1u) << 1) | (((pk >> 11) & 1u) << 2) | (((pk >> 15) & 1u) << 3);
    if ((uint32_t)__builtin_popcount(has) <= r) return false;
    const uint32_t first = __builtin_ctz(has);
    base = r == 0 ? first : __builtin_ctz(has & ~(1u << first));
  }
  const uint32_t nib = (pk >> (4 * base)) & 0xFu;
  if (!(nib & 8u)) return false;
  const uint32_t cp = ((c << sh) | (nib & 7u)) & (N - 1);
  const uint32_t newest = c >> (cd.m - 1), second = (c >> (cd.m - 2)) & 1u;
  t->c = c; t->cp = cp; t->sc = cp - tile * TSx; t->sh = sh;
  t->nb = sh == 1 ? newest : (2 * second + newest);
  t->fpc = t->nb & 2u ? (t->nb & 1u ? pr.fpc[3] : pr.fpc[2]) : (t->nb & 1u ? pr.fpc[1] : pr.fpc[0]);
  t->np_dst = (pr.info >> 16) & 0xFFu; t->np_src = pr.info >> 24;
  uint32_t reach = 0;                                    // crf states of the source conv state that are stored (source_reach)
  t->pk1 = 0;
  if (((cp & pr.vmask1) == pr.vval1) && (pos - 1 < ss.prev_hi)) {
    if (pos - 1 == 0) reach = 0xFFu;
    else {
      const uint32_t pk1 = LVA_GLOBAL(uint16_t, pr.pred1)[cp];
      t->pk1 = pk1;
#pragma unroll
      for (int b = 0; b < 4; ++b) if ((pk1 >> (4 * b)) & 8u) reach |= (0x11u << b);
    }
  }
  const uint32_t k = base + 4 * role;
  t->k = k; t->base = base; t->reach = reach;
  t->c0 = cmp_bits(g, pr) & 1u;
  t->own = (uint32_t)(((uint64_t)(pos % g.R) * 8 + (k >> t->c0)) * g.sCrf);
  uint32_t ok = pos < ss.prev_hi ? 1u : 0u;
  if (role == 0) {
#pragma unroll
    for (uint32_t i = 1; i < 8; ++i) ok |= ((reach >> list_crf(k, i)) & 1u) << i;
  } else {
    ok |= ((reach >> base) & 1u) << 1;
  }
  t->ok = ok;
  return true;
}

// L == 1, both targets of a thread (role 0: the flip target of its (base, conv); role 1: the flop target) side by side:
// an add-compare-select is a chain of dependent round trips (stay entry -> winner -> its message -> store), and two chains
// walked one after the other cost a workgroup twice the time of two chains walked together -- both stay entries are requested
// first, then both winners' messages (:715-742; first maximum wins, stay before the source lists).
template <int P>
__device__ __forceinline__ void acs_pair(const Geometry& g, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur,
                                         const uint2* s_src, const float* s_post, uint32_t src, bool v0, const TileTarget& t0,
                                         bool v1, const TileTarget& t1, lva_u32x2 s0, lva_u32x2 s1, uint32_t crow) {
  const float NEG = -INFINITY;
  const uint32_t sCrf = g.sBlk, pw = 2 * g.N;
  const uint32_t own0 = t0.own + 2 * t0.c, own1 = t1.own + 2 * t1.c;
  float best0 = NEG, best1 = NEG; uint32_t bi0 = 0, bh0 = 0, bi1 = 0, bh1 = 0;
  if (v0) {
    const uint32_t k = t0.k;                                       // k < 4: row = k
    const float s = u2f(s0.x) + s_post[k * 8 + k];
    if ((t0.ok & 1u) && s > best0) { best0 = s; bh0 = s0.y; }
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      if ((t0.ok >> i) & 1u) {
        const uint2 v = s_src[(list_crf(k, i) >> crow) * TS + t0.sc];
        const float c = u2f(v.x) + s_post[k * 8 + list_crf(k, i)];
        if (c > best0) { best0 = c; bi0 = i; bh0 = v.y ^ t0.fpc; }
      }
    }
  }
  if (v1) {
    const uint32_t k = t1.k;                                       // k >= 4: row 4
    const float s = u2f(s1.x) + s_post[4 * 8 + k];
    if ((t1.ok & 1u) && s > best1) { best1 = s; bh1 = s1.y; }
    if ((t1.ok >> 1) & 1u) {
      const uint2 v = s_src[(list_crf(k, 1) >> crow) * TS + t1.sc];
      const float c = u2f(v.x) + s_post[4 * 8 + list_crf(k, 1)];
      if (c > best1) { best1 = c; bi1 = 1; bh1 = v.y ^ t1.fpc; }
    }
  }
  const bool w0 = v0 && best0 != NEG, w1 = v1 && best1 != NEG;
  uint32_t m0[2 * P], m1[2 * P];
  const uint32_t* e0 = prev + (bi0 == 0 ? t0.own : src + mul24(list_crf(t0.k, bi0) >> crow, sCrf)) + pw;
  const uint32_t* e1 = prev + (bi1 == 0 ? t1.own : src + mul24(list_crf(t1.k, bi1) >> crow, sCrf)) + pw;
  const uint32_t c0 = bi0 == 0 ? t0.c : t0.cp, c1 = bi1 == 0 ? t1.c : t1.cp;
  const uint32_t npd = opqs(t0.np_dst), nps = opqs(t0.np_src);     // (uniform over the workgroup)
  if (npd == nps) {     // the plane count does not change at this position (all but two or three positions): ONE uniform
                        // branch on it, and both messages are requested before either is waited for
    // (ONE test for both: straight-line code is what lets the second request leave before the first is waited for; a target
    //  without a winner beside one that has a winner has bi = 0 and reads its own, allocated, stay message for nothing)
    if (w0 || w1) {
      if (npd == 1) { load_msg_np<P, 1>(e0, g.N, c0, m0); load_msg_np<P, 1>(e1, g.N, c1, m1); }
      else if (npd == 2) { load_msg_np<P, 2>(e0, g.N, c0, m0); load_msg_np<P, 2>(e1, g.N, c1, m1); }
      else if (npd == 3) { load_msg_np<P, 3>(e0, g.N, c0, m0); load_msg_np<P, 3>(e1, g.N, c1, m1); }
      else { load_msg_np<P, 4>(e0, g.N, c0, m0); load_msg_np<P, 4>(e1, g.N, c1, m1); }
    }
  } else {
    if (w0) load_msg<P>(e0, g.N, c0, bi0 == 0 ? npd : nps, m0);
    if (w1) load_msg<P>(e1, g.N, c1, bi1 == 0 ? npd : nps, m1);
  }
  if (v0) *reinterpret_cast<uint2*>(cur + own0) = make_uint2(f2u(best0), bh0);
  if (v1) *reinterpret_cast<uint2*>(cur + own1) = make_uint2(f2u(best1), bh1);
  if (w0) {
    push_bits<2 * P>(m0, bi0 == 0 ? 0u : t0.sh, t0.nb);
    store_msg<P>(cur + t0.own + pw, g.N, t0.c, t0.np_dst, m0);
  }
  if (w1) {
    push_bits<2 * P>(m1, bi1 == 0 ? 0u : t1.sh, t1.nb);
    store_msg<P>(cur + t1.own + pw, g.N, t1.c, t1.np_dst, m1);
  }
}

}  // namespace

// grid: x = tiles of 64 source conv states, y = band position index, z = slot index.
// block = 512 threads: thread (role, base r, target conv) -- role 0 (wavefronts 0-3) merges
// the flip target of its (conv, base), role 1 (wavefronts 4-7) the flop target.
template <int LL, int P>
__global__ __launch_bounds__(8 * TS) void lva_step_fast(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                     uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                     uint32_t* __restrict__ items) {
  __shared__ uint2 s_src[8 * LL * TS];
  __shared__ float s_post[40];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    hdr->count[args.step_parity ^ 1u] = 0;     // the other parity's list was consumed by the last fix-up
    hdr->overflow[args.step_parity ^ 1u] = 0;
  }
  SlotStep ss;
  if (!load_slot(args, blockIdx.z, &ss)) return;
  const uint32_t pos = ss.lo + blockIdx.y;
  if (pos >= ss.hi) return;
  const DevCode& cd = codes[ss.orient];
  const uint32_t N = cd.nconv, tid = threadIdx.x, tile = blockIdx.x;
  const uint32_t* prev; uint32_t* cur;
  slot_buffers(ss, g, trellis, &prev, &cur);

  if (pos == 0) {                          // stay-only update of the 8 start states (:706-713)
    if (tile == cd.init / TS && tid < 8) {
      const uint32_t k = tid, c = cd.init;
      const uint32_t own_c = (uint32_t)((uint64_t)k * g.sCrf) + 2 * c;
      const float s = u2f(prev[own_c]) + ss.post_row[(k >= 4 ? 4u : k) * 8 + k];
      cur[own_c] = f2u(s);
      cur[own_c + 1] = prev[own_c + 1];
      cur[own_c + 2 * N] = prev[own_c + 2 * N];          // plane 1 (the only one in use at position 0)
      cur[own_c + 2 * N + 1] = prev[own_c + 2 * N + 1];
      for (int l = 1; l < LL; ++l) cur[own_c + l * g.sBlk] = kNegInfBits;
    }
    return;
  }

  // ---- stage the (score, fingerprint) pairs of 64 source conv states: 8 crf x LL rows of 512 B ----
  const uint32_t src = (uint32_t)((uint64_t)((pos - 1) % g.R) * 8 * g.sCrf);
  for (uint32_t chunk = tid; chunk < 8u * LL * (TS / 2); chunk += 8u * TS) {
    const uint32_t rowi = chunk / (TS / 2), lane2 = chunk % (TS / 2);       // rowi = crf * LL + l
    const uint4 v = *reinterpret_cast<const uint4*>(prev + src + (uint64_t)rowi * g.sBlk + 2 * (tile * TS) + 4 * lane2);
    *reinterpret_cast<uint4*>(&s_src[rowi * TS + 2 * lane2]) = v;
  }
  if (tid < 40) s_post[tid] = LVA_GLOBAL(float, ss.post_row)[tid];
  __syncthreads();

  // ---- this thread's (role, target conv, base) ----
  TileTarget t;
  if (!tile_target<TS>(cd, g, ss, pos, tile, tid, &t)) return;
  const int why = t.k < 4 ? fast_merge<LL, P, 8>(g, prev, cur, s_src, s_post, t.k, t.c, t.cp, t.sc, t.own, src, t.ok, t.sh, t.nb, t.fpc, t.np_dst, t.np_src)
                          : fast_merge<LL, P, 2>(g, prev, cur, s_src, s_post, t.k, t.c, t.cp, t.sc, t.own, src, t.ok, t.sh, t.nb, t.fpc, t.np_dst, t.np_src);
  const uint32_t k = t.k, c = t.c;
  if (why) {
    atomicAdd(&hdr->reason[why - 1], 1ull);
    const uint32_t idx = atomicAdd(&hdr->count[args.step_parity], 1u);
    if (idx < hdr->cap) items[idx] = make_item(cd.m, blockIdx.z, blockIdx.y, k, c);
    else hdr->overflow[args.step_parity] = 1u;
  }
}

// ---------------------------------------------------------------------------------------
// Lazy messages (kernel mode 4, list sizes 2 / 4 / 8).
//
// The reference carries the whole message with every list entry through every time step (:771-774).  Scores and
// fingerprints decide everything the merge does; the message itself is only needed to confirm a fingerprint match and
// at the very end.  So here the message is MATERIALISED EVERY SECOND STEP: a step with even t ("anchor") stores messages
// as lva_step_fast does; a step with odd t stores one byte per accepted entry instead -- where the entry came from in
// step t-1's lists (list i, index j) and which message buffer holds that entry's message.  The next anchor step follows
// two hops (its candidate -> step t-1's entry -> that entry's byte -> step t-2's entry), gathers the message there and
// shifts in the bits of both moves.  Odd steps move no messages at all (except to confirm fingerprint matches): about a
// quarter of the bytes of a step pair disappears.
//   * two message buffers (the message planes of the two parity buffers, otherwise unused at odd steps): anchor step t
//     writes buffer (t >> 1) & 1 and reads, through the bytes, whichever buffer the byte names -- never rows it writes;
//   * the stale row (position lo-1 read with contents older than t-1) keeps working: its entries' messages
//     are where they were written (rows below the band are never overwritten); the host tells odd steps which buffer
//     that is (SlotStep.flags), anchor steps read it from the byte; two hops reach position lo-2, so the ring has one
//     more position (R = 2 max_deviation + 2);
//   * the exact path (lva_step_fixup_lazy, one wavefront per target) resolves messages the same way.
// ---------------------------------------------------------------------------------------
namespace {

struct LazyCtx {
  const uint32_t* M0; const uint32_t* M1;   // the two parity buffers of the slot (message planes = message buffers 0 and 1)
  const uint8_t* bp_prev;        // back-pointer bytes of the previous step's buffer (anchor steps)
  uint32_t N, sBlk, sCrf, pw, m;
  uint32_t c, cp, k, own, src, src2;   // target conv, source conv, target crf; word offsets of (ring(pos),k), ring(pos-1), ring(pos-2)
  uint32_t sh_p, nb_p;           // the move into (pos, c)
  uint32_t sh_q, nb_q, pk1;      // the move into (pos-1, cp); predecessors of cp there (predtab nibbles)
  uint32_t np_p, np_p1, np_p2;   // message planes in use at pos, pos-1, pos-2
  uint32_t t, fb, stale_pos1, stale_mb;  // time step; message buffer of fresh step t-1 entries; sources at pos-1 stale? its buffer
  uint32_t c1, c2;               // compact lists at pos-1 / at pos-2 (crf kk -> list kk >> 1); `own` already names the target's list
};

// byte index, inside a parity buffer, of the back-pointer byte of entry j of conv state `conv` of the list that starts at word
// `list`: the bytes of a conv state's L entries are adjacent (one L-byte store per thread and odd step)
__device__ __forceinline__ uint32_t bp_byte_index(const Geometry& g, uint32_t list, uint32_t j, uint32_t conv) {
  return (list + g.L * g.sBlk) * 4u + conv * g.L + j;
}

// message buffer that holds the message of step t-1's entry of list i (odd steps)
__device__ __forceinline__ uint32_t lazy_mbuf(const LazyCtx& x, uint32_t i) {
  return sel(i != 0 && opqs(x.stale_pos1) != 0, opqs(x.stale_mb), opqs(x.fb));
}

// Where the stored message behind candidate (list i, index j) of the previous step lives, and the moves to apply to it:
// *ent = message region, *conv / *np = conv index and planes in use there, (s1, n1) then (s2, nb_p) = shifts and new bits.
// bp1 = the candidate's own back-pointer byte (anchor steps; ignored at odd steps).  false: the message is empty (t = 0).
// (Selections between fields of the context go through selv: a select between two loads would pin the struct in scratch.)
__device__ __forceinline__ bool lazy_locate(const LazyCtx& x, uint32_t i, uint32_t j, uint32_t bp1, const uint32_t** ent, uint32_t* conv,
                                            uint32_t* np, uint32_t* s1, uint32_t* n1, uint32_t* s2) {
  const uint32_t kk = i == 0 ? x.k : list_crf(x.k, i);
  const uint32_t own = opq(x.own), c = opq(x.c), cp = opq(x.cp);
  const uint32_t src = opqs(x.src), src2 = opqs(x.src2);                         // (uniform over the workgroup)
  const uint32_t np_p = opqs(x.np_p), np_p1 = opqs(x.np_p1), np_p2 = opqs(x.np_p2), sh_p = opqs(x.sh_p);
  const uint32_t* M0 = opqs(x.M0); const uint32_t* M1 = opqs(x.M1);
  *s1 = 0; *n1 = 0;
  const bool stay = i == 0;
  if (x.t & 1u) {                                            // odd step: step t-1's entries carry their messages
    const uint32_t lst = sel(stay, own, src + mul24(kk >> opqs(x.c1), x.sCrf));
    *ent = sel(lazy_mbuf(x, i) != 0, M1, M0) + lst + mul24(j, x.sBlk) + x.pw;
    *conv = sel(stay, c, cp); *np = sel(stay, np_p, np_p1); *s2 = sel(stay, 0u, sh_p);
    return true;
  }
  *s2 = sel(stay, 0u, sh_p);
  if (x.t == 0) { *ent = M0; *conv = 0; *np = 1; return false; }   // the initial entries: empty message
  const uint32_t i1 = (bp1 >> 3) & 7u, j1 = bp1 & 7u;
  const bool stay1 = i1 == 0;
  // four cases: (stay, stay) own state; (stay, move) and (move, stay) the source state at pos-1; (move, move) pos-2
  const uint32_t y1 = (x.pk1 >> (4 * (kk & 3u))) & 7u;
  const uint32_t cpp = ((cp << x.sh_q) | y1) & (x.N - 1u);
  const uint32_t crf1 = list_crf(kk, stay1 ? 1u : i1);       // crf of the second hop's source when it is a move
  const uint32_t c1 = opqs(x.c1), c2 = opqs(x.c2);
  const uint32_t l_ss = own, l_sm = src + mul24(crf1 >> c1, x.sCrf), l_ms = src + mul24(kk >> c1, x.sCrf), l_mm = src2 + mul24(crf1 >> c2, x.sCrf);
  const uint32_t lst = sel(stay, sel(stay1, l_ss, l_sm), sel(stay1, l_ms, l_mm));
  *conv = sel(stay, sel(stay1, c, cp), sel(stay1, cp, cpp));
  *np = sel(stay, sel(stay1, np_p, np_p1), sel(stay1, np_p1, np_p2));
  const bool mm = !stay && !stay1;
  *s1 = sel(mm, opqs(x.sh_q), 0u); *n1 = sel(mm, opq(x.nb_q), 0u);
  *s2 = sel(stay && stay1, 0u, sh_p);
  *ent = sel(((bp1 >> 6) & 1u) != 0, M1, M0) + lst + mul24(j1, x.sBlk) + x.pw;
  return true;
}

// Message of candidate (list i, index j) of the previous step AS IT WOULD STAND IN THE TARGET (all moves applied).
template <int P>
__device__ __forceinline__ void lazy_message(const LazyCtx& x, uint32_t i, uint32_t j, uint32_t bp1, uint32_t (&mw)[2 * P]) {
  const uint32_t* ent; uint32_t conv, np, s1, n1, s2;
  if (lazy_locate(x, i, j, bp1, &ent, &conv, &np, &s1, &n1, &s2)) load_msg<P>(ent, x.N, conv, np, mw);
  else {
#pragma unroll
    for (int w = 0; w < 2 * P; ++w) mw[w] = 0;
  }
  push_var<2 * P>(mw, s1 + s2, (n1 << s2) | (s2 ? x.nb_p : 0u));     // both moves in one funnel shift per word
}

__device__ __forceinline__ void lazy_ctx(const DevCode& cd, const Geometry& g, const SlotStep& ss, const uint32_t* slot_base, uint32_t pos,
                                         uint32_t c, uint32_t cp, uint32_t k, uint32_t own, LazyCtx* x, int pk1_known = -1) {
  x->M0 = slot_base; x->M1 = slot_base + g.sPar;
  x->bp_prev = reinterpret_cast<const uint8_t*>(slot_base + (uint64_t)(ss.t & 1u) * g.sPar);
  x->N = g.N; x->sBlk = g.sBlk; x->sCrf = (uint32_t)g.sCrf; x->pw = 2 * g.N; x->m = cd.m;
  x->c = c; x->cp = cp; x->k = k; x->own = own;
  x->src = (uint32_t)((uint64_t)((pos + g.R - 1) % g.R) * 8 * g.sCrf);
  x->src2 = (uint32_t)((uint64_t)((pos + g.R - 2) % g.R) * 8 * g.sCrf);
  const PosRec pr = cd.rec[pos];           // (one scalar load; pos >= 1 here)
  const uint32_t Tp = pr.info & 0xFFu;
  x->sh_p = Tp == 0 ? 1u : 2u;
  x->nb_p = x->sh_p == 1 ? (c >> (cd.m - 1)) : (2 * ((c >> (cd.m - 2)) & 1u) + (c >> (cd.m - 1)));
  const uint32_t Tq = (pr.info >> 8) & 0xFFu;
  x->sh_q = Tq == 0 ? 1u : 2u;
  x->nb_q = x->sh_q == 1 ? (cp >> (cd.m - 1)) : (2 * ((cp >> (cd.m - 2)) & 1u) + (cp >> (cd.m - 1)));
  // (pk1_known >= 0: the caller's tile_target has read this word already -- wherever the target has a source list at all)
  x->pk1 = pk1_known >= 0 ? (uint32_t)pk1_known : (pos >= 2 && !(ss.t & 1u)) ? (uint32_t)LVA_GLOBAL(uint16_t, pr.pred1)[cp] : 0u;
  x->np_p = (pr.info >> 16) & 0xFFu; x->np_p1 = pr.info >> 24; x->np_p2 = pr.np2;
  x->c1 = (cmp_bits(g, pr) >> 1) & 1u; x->c2 = (cmp_bits(g, pr) >> 2) & 1u;
  x->t = ss.t; x->fb = ((ss.t - 1u) >> 1) & 1u;
  x->stale_pos1 = (pos == ss.lo) && (ss.flags & 1u);
  x->stale_mb = (ss.flags >> 1) & 1u;
}

// Output phase of one target on the lazy path.  false = a fingerprint match did not survive the comparison of the
// full messages (collision): the exact path redoes the target.
// own_bp: the back-pointer bytes of the target's own (stay) list in the previous buffer, entry j in byte j (anchor steps).
template <int LL, int P, bool ANCHOR>
__device__ __forceinline__ bool lazy_output(const Geometry& g, const LazyCtx& x, uint32_t* __restrict__ cur, uint32_t* __restrict__ mout,
                                            const uint8_t* s_bp, uint32_t sc, unsigned long long own_bp, unsigned long long asrc,
                                            unsigned long long rej0, uint32_t lc) {
  bool good = true;
  // the candidate's own back-pointer byte: staged in LDS for source lists, prefetched for the stay list
  auto bp_of = [&](uint32_t i, uint32_t j) __attribute__((always_inline)) -> uint32_t {
    if (i == 0) return (uint32_t)(own_bp >> (8 * j)) & 0xFFu;
    return s_bp[((list_crf(x.k, i) >> opqs(x.c1)) * TS + sc) * LL + j];
  };
  // the fingerprint match filed under entry l must be the same message as the entry's (mw): anchor steps, where the
  // entry's message is in registers anyway
  auto verify = [&](int l, const uint32_t (&mw)[2 * P]) __attribute__((always_inline)) {
    const uint32_t rec = (uint32_t)(rej0 >> (kRecBits * l)) & 0x7Fu;
    if (rec & 0x40u) {
      const uint32_t ri = (rec >> 3) & 7u, rj = rec & 7u;
      uint32_t qm[2 * P];
      lazy_message<P>(x, ri, rj, bp_of(ri, rj), qm);
#pragma unroll
      for (int w = 0; w < 2 * P; ++w) good &= (qm[w] == mw[w]);
    }
  };
  // Odd steps confirm their fingerprint matches in a loop over the entries that HAVE one (about one per target, rarely more
  // than three): inside an unrolled entry loop each of the eight `if (match filed under entry l)` blocks runs for the whole
  // wavefront as soon as one lane has a match there -- always -- so a wavefront executed eight confirmations for one per lane.
  // (The same loop in the anchor instance, which has every entry's message in registers at some point anyway: -3 %.)
  auto verify_loop = [&]() __attribute__((always_inline)) {
    uint32_t todo = 0;
#pragma unroll
    for (int l = 0; l < LL; ++l) todo |= ((uint32_t)(rej0 >> (kRecBits * l + 6)) & 1u) << l;
    while (todo) {
      const uint32_t l = (uint32_t)__builtin_ctz(todo);
      todo &= todo - 1u;
      const uint32_t a8 = (uint32_t)(asrc >> (8 * l)) & 0x3Fu, rec = (uint32_t)(rej0 >> (kRecBits * l)) & 0x3Fu;
      uint32_t ma[2 * P], mb[2 * P];
      {
        // odd step: the pair is a stay entry (the target's own list, message where the last anchor step put it) and an entry of one
        // source list (two source lists: 5 in 100 000, the general path below).  Which is which differs per lane, where they live
        // does not: planes in use, message buffers and the move are uniform over the workgroup -- both messages are requested
        // together behind uniform branches only, one round trip per confirmation instead of one per piece.
        const bool a_stay = (a8 >> 3) == 0, r_stay = (rec >> 3) == 0;
        if (a_stay != r_stay) {
          const uint32_t s = a_stay ? (a8 & 7u) : (rec & 7u), f6 = a_stay ? rec : a8;
          const uint32_t* Mf = opqs(x.fb) ? opqs(x.M1) : opqs(x.M0);
          const uint32_t* Ms = opqs(x.stale_pos1) ? (opqs(x.stale_mb) ? opqs(x.M1) : opqs(x.M0)) : Mf;
          const uint32_t np_a = opqs(x.np_p), np_b = opqs(x.np_p1);
          load_msg<P>(Mf + x.own + mul24(s, x.sBlk) + x.pw, x.N, x.c, np_a, ma);
          load_msg<P>(Ms + opqs(x.src) + mul24(list_crf(x.k, f6 >> 3) >> opqs(x.c1), x.sCrf) + mul24(f6 & 7u, x.sBlk) + x.pw, x.N, x.cp, np_b, mb);
          push_var<2 * P>(mb, opqs(x.sh_p), x.nb_p);
#pragma unroll
          for (int w = 0; w < 2 * P; ++w) good &= (ma[w] == mb[w]);
          continue;
        }
      }
      lazy_message<P>(x, a8 >> 3, a8 & 7u, 0u, ma);
      lazy_message<P>(x, rec >> 3, rec & 7u, 0u, mb);
#pragma unroll
      for (int w = 0; w < 2 * P; ++w) good &= (ma[w] == mb[w]);
    }
  };
  if constexpr (!ANCHOR) {
    // ---- odd step: one byte per accepted entry; messages are touched only to confirm fingerprint matches ----
    unsigned long long packed = 0;
#pragma unroll
    for (int l = 0; l < LL; ++l) {
      if ((uint32_t)l < lc) {
        const uint32_t a8 = (uint32_t)(asrc >> (8 * l)) & 0xFFu;
        packed |= (unsigned long long)(a8 | (lazy_mbuf(x, a8 >> 3) << 6)) << (8 * l);
      }
    }
    uint8_t* dst = reinterpret_cast<uint8_t*>(cur) + bp_byte_index(g, x.own, 0, x.c);
    if constexpr (LL == 8) *reinterpret_cast<unsigned long long*>(dst) = packed;
    else if constexpr (LL == 4) *reinterpret_cast<uint32_t*>(dst) = (uint32_t)packed;
    else *reinterpret_cast<uint16_t*>(dst) = (uint16_t)packed;
    verify_loop();
    return good;
  }
  // ---- anchor step: two hops to the stored message, both moves applied, stored coalesced; kLazyInFlight entries in flight ----
  // (four message planes: one entry in flight -- 8 more message registers would cost the anchor instance a wavefront per SIMD;
  //  measured at m=14: 4.91 against 4.69 reads/s)
  constexpr int GBW = P >= 4 ? 1 : kLazyInFlight;
  constexpr int GB = LL >= GBW ? GBW : LL;
#pragma unroll
  for (int l0 = 0; l0 < LL; l0 += GB) {
    uint32_t m[GB][2 * P], mv[GB];                         // mv: the moves to apply, packed (s1 | n1 << 2 | s2 << 4)
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int l = l0 + u;
      mv[u] = 0;
#pragma unroll
      for (int w = 0; w < 2 * P; ++w) m[u][w] = 0;
      if ((uint32_t)l < lc) {
        const uint32_t a8 = (uint32_t)(asrc >> (8 * l)) & 0xFFu;
        const uint32_t i = a8 >> 3, j = a8 & 7u;
        const uint32_t* ent; uint32_t conv, np, s1, n1, s2;
        if (lazy_locate(x, i, j, bp_of(i, j), &ent, &conv, &np, &s1, &n1, &s2)) load_msg<P>(ent, x.N, conv, np, m[u]);
        mv[u] = s1 | (n1 << 2) | (s2 << 4);
      }
    }
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int l = l0 + u;
      if ((uint32_t)l < lc) {
        push_var<2 * P>(m[u], (mv[u] & 3u) + (mv[u] >> 4), (((mv[u] >> 2) & 3u) << (mv[u] >> 4)) | ((mv[u] >> 4) ? x.nb_p : 0u));
        store_msg<P>(mout + x.own + l * x.sBlk + x.pw, x.N, x.c, x.np_p, m[u]);
        if ((uint32_t)(rej0 >> (kRecBits * l)) & 0x40u) verify(l, m[u]);
      }
    }
  }
  return good;
}

}  // namespace

// grid / block as lva_step_fast.  Two instances per step: ANCHOR = true serves the slots whose time step is even,
// ANCHOR = false those at an odd step (workgroups of the other kind leave at once) -- the odd-step path keeps the
// merge's small register footprint (no message in flight), the anchor path is the only one that pays for two hops.
template <int LL, int P, bool ANCHOR>
__global__ __launch_bounds__(8 * TS, kLazyMinWaves) void lva_step_lazy(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                     uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                     uint32_t* __restrict__ items) {
  __shared__ uint2 s_src[8 * LL * TS];
  __shared__ uint8_t s_bp[ANCHOR ? 8 * LL * TS : 4];
  __shared__ float s_post[40];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    hdr->count[args.step_parity ^ 1u] = 0;     // the other parity's list was consumed by the last fix-up
    hdr->overflow[args.step_parity ^ 1u] = 0;
  }
  SlotStep ss;
  if (!load_slot(args, blockIdx.z, &ss)) return;
  if (!(ss.t & 1u) != ANCHOR) return;
  const uint32_t pos = ss.lo + blockIdx.y;
  if (pos >= ss.hi) return;
  const DevCode& cd = codes[ss.orient];
  const uint32_t N = cd.nconv, tid = threadIdx.x, tile = blockIdx.x;
  uint32_t* slot_base = trellis + (uint64_t)ss.slot * g.sSlot;
  const uint32_t* prev; uint32_t* cur;
  slot_buffers(ss, g, trellis, &prev, &cur);
  constexpr bool anchor = ANCHOR;
  uint32_t* mout = slot_base + (uint64_t)((ss.t >> 1) & 1u) * g.sPar;      // message buffer an anchor step writes

  if (pos == 0) {                          // stay-only update of the 8 start states (:706-713); their message stays empty
    if (tile == cd.init / TS && tid < 8) {
      const uint32_t k = tid, c = cd.init;
      const uint32_t own = (uint32_t)((uint64_t)k * g.sCrf), own_c = own + 2 * c;
      const float s = u2f(prev[own_c]) + ss.post_row[(k >= 4 ? 4u : k) * 8 + k];
      cur[own_c] = f2u(s);
      cur[own_c + 1] = prev[own_c + 1];
      if (anchor) { mout[own + 2 * N + 2 * c] = 0u; mout[own + 2 * N + 2 * c + 1] = 0u; }
      else reinterpret_cast<uint8_t*>(cur)[bp_byte_index(g, own, 0, c)] = (uint8_t)((((ss.t - 1u) >> 1) & 1u) << 6);   // stay, entry 0
      for (int l = 1; l < LL; ++l) cur[own_c + l * g.sBlk] = kNegInfBits;
    }
    return;
  }

  if (!tile_has_target(args, cd, pos, tile)) return;     // (first / last positions: most tiles have no valid target)
  // ---- stage the (score, fingerprint) pairs of 64 source conv states (and, for an anchor step, their back-pointer bytes):
  //      8 crf lists, or the 4 compact lists of a one-bit source position (all of them data) ----
  const uint32_t src = (uint32_t)((uint64_t)((pos - 1) % g.R) * 8 * g.sCrf);
  const uint32_t crow = source_compact(g, cd, ss, blockIdx.y, pos);
  const uint32_t nrow = 8u >> crow;
  // Every request of the workgroup's first phase is in flight before the first LDS write waits: the staging loads (NCH chunks of
  // 16 bytes per thread: LL / 2 for 8 rows, half of that for the 4 rows of a compact source position -- uniform), at an anchor
  // step the NBP words of back-pointer bytes, and the posteriors.  Until round 6 these were loops over run-time counts, and such
  // a loop waits for each load before it asks for the next: four + two + one round trips in a row in front of the barrier, one
  // now (benchmark 48.3 -> 49.2 reads/s; DESIGN_HISTORY R6).
  TileTarget t;
  auto stage = [&](auto nch, auto nbp) __attribute__((always_inline)) {
    constexpr uint32_t NCH = decltype(nch)::value, NBP = decltype(nbp)::value;
    static_assert(NCH >= 1 && NCH <= 4 && NBP >= 1 && NBP <= 2, "chunks per thread");
    constexpr uint32_t kW = TS * LL / 4;   // back-pointer words per crf: the L bytes of 64 conv states = TS*LL contiguous bytes
    const uint32_t nchunk = nrow * LL * (TS / 2), nbw = nrow * kW;
    // (named registers, not arrays: the compiler leaves an array it cannot fully scalarise in scratch memory or in LDS)
    uint4 v0, v1, v2, v3;
    uint32_t bw0, bw1;
    float pv;
    auto src_of = [&](uint32_t u) __attribute__((always_inline)) -> const uint4* {
      const uint32_t chunk = tid + u * 8u * TS, rowi = chunk / (TS / 2), lane2 = chunk % (TS / 2);       // rowi = crf * LL + l
      return reinterpret_cast<const uint4*>(prev + src + (uint64_t)(rowi / LL) * g.sCrf + (uint64_t)(rowi % LL) * g.sBlk +
                                            2 * (tile * TS) + 4 * lane2);
    };
    auto dst_of = [&](uint32_t u) __attribute__((always_inline)) -> uint4* {
      const uint32_t chunk = tid + u * 8u * TS, rowi = chunk / (TS / 2), lane2 = chunk % (TS / 2);
      return reinterpret_cast<uint4*>(&s_src[rowi * TS + 2 * lane2]);
    };
    // (only at LL = 2 can a row block be smaller than the workgroup: no per-lane guard elsewhere)
    const bool in0 = LL != 2 || tid < nchunk;
    if (in0) v0 = *src_of(0);
    if constexpr (NCH > 1) v1 = *src_of(1);
    if constexpr (NCH > 2) v2 = *src_of(2);
    if constexpr (NCH > 3) v3 = *src_of(3);
    auto bp_of = [&](uint32_t u) __attribute__((always_inline)) -> const uint32_t* {
      const uint32_t chunk = tid + u * 8u * TS;
      return prev + src + (uint64_t)(chunk / kW) * g.sCrf + (uint64_t)LL * g.sBlk + (tile * TS * LL) / 4 + chunk % kW;
    };
    const bool b0 = LL == 8 || tid < nbw;
    if (anchor && ss.t != 0) {
      if (b0) bw0 = *bp_of(0);
      if constexpr (NBP > 1) bw1 = *bp_of(1);
    }
    if (tid < 40) pv = LVA_GLOBAL(float, ss.post_row)[tid];
    if (in0) *dst_of(0) = v0;
    if constexpr (NCH > 1) *dst_of(1) = v1;
    if constexpr (NCH > 2) *dst_of(2) = v2;
    if constexpr (NCH > 3) *dst_of(3) = v3;
    if (anchor && ss.t != 0) {
      if (b0) *reinterpret_cast<uint32_t*>(&s_bp[(tid / kW) * TS * LL + 4 * (tid % kW)]) = bw0;
      if constexpr (NBP > 1) *reinterpret_cast<uint32_t*>(&s_bp[((tid + 8u * TS) / kW) * TS * LL + 4 * ((tid + 8u * TS) % kW)]) = bw1;
    }
    if (tid < 40) s_post[tid] = pv;
  };
  {
    constexpr uint32_t kCh = (8u * LL * (TS / 2) + 8u * TS - 1u) / (8u * TS);          // LL = 8: 4, LL = 4: 2, LL = 2: 1
    constexpr uint32_t kBp = (8u * (TS * LL / 4) + 8u * TS - 1u) / (8u * TS);          // LL = 8: 2, else 1
    if (crow) stage(std::integral_constant<uint32_t, (kCh + 1) / 2>{}, std::integral_constant<uint32_t, (kBp + 1) / 2>{});
    else stage(std::integral_constant<uint32_t, kCh>{}, std::integral_constant<uint32_t, kBp>{});
  }
  __syncthreads();

  if (!tile_target<TS>(cd, g, ss, pos, tile, tid, &t)) return;
  // an anchor step needs the back-pointer bytes of its own (stay) list: requested now, used after the merge
  unsigned long long own_bp = 0;
  if (anchor && ss.t != 0 && (t.ok & 1u)) {
    const uint8_t* bpp = reinterpret_cast<const uint8_t*>(prev) + bp_byte_index(g, t.own, 0, t.c);
    if constexpr (LL == 8) own_bp = *reinterpret_cast<const unsigned long long*>(bpp);
    else if constexpr (LL == 4) own_bp = *reinterpret_cast<const uint32_t*>(bpp);
    else own_bp = *reinterpret_cast<const uint16_t*>(bpp);
  }
  unsigned long long asrc = 0, rej0 = 0;
  uint32_t lc = 0;
  int why = t.k < 4 ? fast_merge_core<LL, 8>(g, prev, cur, s_src, s_post, t.k, t.c, t.sc, t.own, t.ok, t.fpc, &asrc, &rej0, &lc, crow)
                    : fast_merge_core<LL, 2>(g, prev, cur, s_src, s_post, t.k, t.c, t.sc, t.own, t.ok, t.fpc, &asrc, &rej0, &lc, crow);
  if (!why) {
    LazyCtx x;
    lazy_ctx(cd, g, ss, slot_base, pos, t.c, t.cp, t.k, t.own, &x, ANCHOR ? (int)t.pk1 : 0);
    if (!lazy_output<LL, P, ANCHOR>(g, x, cur, mout, s_bp, t.sc, own_bp, asrc, rej0, lc)) why = 4;
  }
  if (why) {
    atomicAdd(&hdr->reason[why - 1], 1ull);
    const uint32_t idx = atomicAdd(&hdr->count[args.step_parity], 1u);
    if (idx < hdr->cap) items[idx] = make_item(cd.m, blockIdx.z, blockIdx.y, t.k, t.c);
    else hdr->overflow[args.step_parity] = 1u;
  }
}

// exact path behind lva_step_lazy: one wavefront per queued target, the reference merge (:743-800) on lane-resident values
// as fixup_small, messages resolved through lazy_message.
template <int P>
__global__ __launch_bounds__(256) void lva_step_fixup_lazy(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                           uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                           const uint32_t* __restrict__ items) {
  const uint32_t par = args.step_parity;
  const uint32_t n = hdr->count[par] < hdr->cap ? hdr->count[par] : hdr->cap;
  // work list overflowed (tie-dense posteriors: quantised or constant matrices): the whole step is redone on the exact path,
  // still one wavefront per target -- every target of every active slot, whatever the fast kernel wrote for it (the
  // exact path reads only the previous step's rows and message rows no step is writing, so redoing a target is idempotent)
  const bool all = hdr->overflow[par] != 0;
  if (n == 0 && !all) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (all) atomicAdd(&hdr->overflow_steps, 1u); else atomicAdd(&hdr->total, (unsigned long long)n);
  }
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u, nwaves = gridDim.x * 4;
  const uint32_t mm = codes[0].m;
  const uint64_t per_slot = (uint64_t)args.band_max * g.N * 8;
  const uint64_t ntargets = all ? per_slot * args.nslots : (uint64_t)n;
  const uint32_t L = g.L, sBlk = g.sBlk, sCrf = (uint32_t)g.sCrf;
  const float NEG = -INFINITY;
  auto rdf = [](float v, uint32_t ln) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)ln)); };
  auto rdu = [](uint32_t v, uint32_t ln) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)ln); };
  auto wrf = [lane](float& v, uint32_t ln, float x) { v = lane == ln ? x : v; };
  auto wru = [lane](uint32_t& v, uint32_t ln, uint32_t x) { v = lane == ln ? x : v; };
  for (uint64_t idx = blockIdx.x * 4 + wv; idx < ntargets; idx += nwaves) {
    uint32_t it;                           // (uniform per wavefront)
    if (all) {
      const uint32_t si = (uint32_t)(idx / per_slot), rem = (uint32_t)(idx % per_slot);
      it = make_item(mm, si, rem / (g.N * 8), (rem / g.N) & 7u, rem % g.N);
    } else {
      it = items[idx];
    }
    SlotStep ss;
    if (!load_slot(args, it >> (mm + 11), &ss)) continue;
    const uint32_t pos = ss.lo + ((it >> (mm + 3)) & 0xFFu), k = (it >> mm) & 7u, c = it & ((1u << mm) - 1u);
    if (pos >= ss.hi) continue;            // (whole-step pass: band positions beyond this slot's band)
    const DevCode& cd = codes[ss.orient];
    const uint32_t* prev; uint32_t* cur;
    slot_buffers(ss, g, trellis, &prev, &cur);
    uint32_t* slot_base = trellis + (uint64_t)ss.slot * g.sSlot;
    uint32_t* mout = slot_base + (uint64_t)((ss.t >> 1) & 1u) * g.sPar;
    Target tg;
    if (!resolve_target(cd, g, ss, pos, c, k, &tg)) continue;   // (uniform per wavefront)
    if (pos == 0) continue;                // the fast kernel's stay-only update of position 0 is exact
    LazyCtx x;
    lazy_ctx(cd, g, ss, slot_base, pos, tg.c, tg.cp, k, tg.own, &x);
    const bool anchor = !(ss.t & 1u);
    // 1. candidates: lane = list*8 + index
    float cs = NEG; uint32_t cy = 0, cb = 0;
    {
      const uint32_t i = lane >> 3, j = lane & 7u;
      if (i < tg.nlists && ((tg.okmask >> i) & 1u) && j < L) {
        const uint32_t lst = i == 0 ? tg.own : tg.src + (list_crf(k, i) >> tg.csrc) * sCrf;
        const uint2 v = *reinterpret_cast<const uint2*>(prev + lst + j * sBlk + 2 * (i == 0 ? tg.c : tg.cp));
        cs = u2f(v.x); cy = i != 0 ? v.y ^ tg.fpc : v.y;
        if (anchor && ss.t != 0) cb = x.bp_prev[bp_byte_index(g, lst, j, i == 0 ? tg.c : tg.cp)];
      }
    }
    float addv = 0.0f;
    if (lane < tg.nlists) addv = ss.post_row[tg.row * 8 + (lane == 0 ? k : list_crf(k, lane))];
    // word w of the candidate message of entry (li, lj): every lane computes the whole message and picks its word
    auto word_of = [&](uint32_t li, uint32_t lj, uint32_t w) -> uint32_t {
      uint32_t mw[2 * P];
      lazy_message<P>(x, li, lj, rdu(cb, li * 8 + lj), mw);
      uint32_t v = 0;
#pragma unroll
      for (int u = 0; u < 2 * P; ++u) v = w == (uint32_t)u ? mw[u] : v;
      return v;
    };
    float hs = NEG; uint32_t hx = 0;
    float as = NEG; uint32_t ay = 0, ax = 0;
    auto sift_up = [&](uint32_t hole, uint32_t top, float vs, uint32_t vx) {
      while (hole > top) {
        const uint32_t parent = (hole - 1) / 2;
        const float ps = rdf(hs, parent);
        if (!(ps < vs)) break;
        wrf(hs, hole, ps); wru(hx, hole, rdu(hx, parent));
        hole = parent;
      }
      wrf(hs, hole, vs); wru(hx, hole, vx);
    };
    auto adjust = [&](uint32_t hole, uint32_t len, float vs, uint32_t vx) {
      const uint32_t top = hole;
      uint32_t child = hole;
      while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (rdf(hs, child) < rdf(hs, child - 1)) --child;
        wrf(hs, hole, rdf(hs, child)); wru(hx, hole, rdu(hx, child));
        hole = child;
      }
      if ((len & 1u) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        wrf(hs, hole, rdf(hs, child - 1)); wru(hx, hole, rdu(hx, child - 1));
        hole = child - 1;
      }
      sift_up(hole, top, vs, vx);
    };
    uint32_t hn = 0;
    for (uint32_t i = 0; i < tg.nlists; ++i) {                         // :750-761
      const float head = rdf(cs, i * 8);
      if (head != NEG) { wrf(hs, hn, head + rdf(addv, i)); wru(hx, hn, i << 16); ++hn; }
    }
    if (hn >= 2)                                                       // std::make_heap :762
      for (uint32_t parent = (hn - 2) / 2;; --parent) {
        adjust(parent, hn, rdf(hs, parent), rdu(hx, parent));
        if (parent == 0) break;
      }
    uint32_t l = 0;
    const uint32_t Wd = 2 * tg.np_dst;
    while (hn > 0 && l < L) {                                          // :764
      const float ts = rdf(hs, 0); const uint32_t tx = rdu(hx, 0);
      if (hn > 1) adjust(0, hn - 1, rdf(hs, hn - 1), rdu(hx, hn - 1));
      --hn;
      const uint32_t i = tx >> 16, j = tx & 0xFFFFu;
      const uint32_t ch = rdu(cy, i * 8 + j);
      bool dup = false;                                                // :778-779
      unsigned long long match = __ballot(lane < l && ay == ch);
      while (match && !dup) {
        const uint32_t a = (uint32_t)__builtin_ctzll(match);
        match &= match - 1;
        const uint32_t asrc = rdu(ax, a);
        uint32_t diff = 0;
        const uint32_t wa = word_of(i, j, lane < Wd ? lane : 0u), wb2 = word_of(asrc >> 16, asrc & 0xFFFFu, lane < Wd ? lane : 0u);
        if (lane < Wd) diff = wa ^ wb2;
        dup = __ballot(diff != 0) == 0ull;
      }
      if (!dup) { wrf(as, l, ts); wru(ay, l, ch); wru(ax, l, tx); ++l; }
      if (j == L - 1) continue;
      const float nxt = rdf(cs, i * 8 + j + 1);
      if (nxt != NEG) {
        sift_up(hn, 0, nxt + rdf(addv, i), (i << 16) | (j + 1));
        ++hn;
      }
    }
    // 3. outputs: lane = entry*8 + word
    {
      const uint32_t e = lane >> 3, w = lane & 7u;
      const float es = __shfl(as, (int)e);
      const uint32_t ey = __shfl(ay, (int)e), ex = __shfl(ax, (int)e);
      if (e < L && w == 0)
        *reinterpret_cast<uint2*>(cur + tg.own + e * sBlk + 2 * tg.c) = e < l ? make_uint2(f2u(es), ey) : make_uint2(kNegInfBits, 0u);
      const uint32_t li = (ex >> 16) & 7u, lj = ex & 7u;
      const uint32_t b1 = (uint32_t)__shfl((int)cb, (int)(li * 8 + lj));   // the entry's own back-pointer byte (per lane: no readlane here)
      if (anchor) {
        uint32_t wv2 = 0;
        if (e < l) {
          uint32_t mw[2 * P];
          lazy_message<P>(x, li, lj, b1, mw);
#pragma unroll
          for (int u = 0; u < 2 * P; ++u) wv2 = w == (uint32_t)u ? mw[u] : wv2;
        }
        if (e < l && w < Wd) mout[tg.own + e * sBlk + 2 * g.N + msg_word_off(g.N, tg.c, w, tg.np_dst)] = wv2;
      } else if (e < l && w == 0) {
        reinterpret_cast<uint8_t*>(cur)[bp_byte_index(g, tg.own, e, tg.c)] = (uint8_t)((li << 3) | lj | (lazy_mbuf(x, li) << 6));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// L == 1 (plain Viterbi, :715-742): the same butterfly tile with HALF the threads -- thread (base r, target conv)
// does the flip target of its (conv, base) AND the flop target, side by side (acs_pair).  An add-compare-select is a
// handful of instructions behind a chain of dependent round trips, so what matters is how many tiles a CU has in
// flight (256-thread workgroups: 8 instead of 4 per CU) and how short the chain is: slot record (one load) ->
// staging | position record -> tables -> stay entries, all requested before the first wait -> barrier -> winners ->
// their messages (both together) -> stores.  No ties to resolve (first maximum wins), no work list.
// grid: x = tiles of 64 source conv states, y = band position index, z = slot index.
// ---------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(4 * TS) void lva_step_acs(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                       uint32_t* __restrict__ trellis) {
  __shared__ uint2 s_src[8 * TS];
  __shared__ float s_post[40];
  SlotStep ss;
  if (!load_slot_whole(args, blockIdx.z, &ss)) return;
  const uint32_t pos = ss.lo + blockIdx.y;
  if (pos >= ss.hi) return;
  const DevCode& cd = codes[ss.orient];
  const uint32_t N = cd.nconv, tid = threadIdx.x, tile = xcd_tile(blockIdx.x, cd.rec[pos].xs & 0xFFu);
  const uint32_t* prev; uint32_t* cur;
  slot_buffers(ss, g, trellis, &prev, &cur);
  if (pos == 0) {                          // stay-only update of the 8 start states (:706-713)
    if (tile == cd.init / TS && tid < 8) {
      const uint32_t k = tid, c = cd.init;
      const uint32_t own_c = (uint32_t)((uint64_t)k * g.sCrf) + 2 * c;
      const float s = u2f(prev[own_c]) + ss.post_row[(k >= 4 ? 4u : k) * 8 + k];
      cur[own_c] = f2u(s);
      cur[own_c + 1] = prev[own_c + 1];
      cur[own_c + 2 * N] = prev[own_c + 2 * N];          // plane 1 (the only one in use at position 0)
      cur[own_c + 2 * N + 1] = prev[own_c + 2 * N + 1];
    }
    return;
  }
  if (!tile_has_target(args, cd, pos, tile)) return;     // (first / last positions: most tiles have no valid target)
  // stage the (score, fingerprint) pairs of 64 source conv states: 8 crf rows of 512 B, one 16-byte piece per thread
  const uint32_t src = (uint32_t)((uint64_t)((pos - 1) % g.R) * 8 * g.sCrf);
  // Every request that does not depend on another goes out before anything is waited for: the staging piece, the posteriors,
  // the tables of this thread's targets and -- behind those -- their stay entries.  What is left of the chain of round trips:
  // (slot record) -> (staging | tables -> stay entries) -> barrier -> winners -> their messages -> stores.
  const uint32_t rowi = tid / (TS / 2), lane2 = tid % (TS / 2);
  const uint32_t crow = source_compact(g, cd, ss, blockIdx.y, pos);      // 4 rows instead of 8: a one-bit source position
  lva_u32x4 sv = {kNegInfBits, 0u, kNegInfBits, 0u};
  if (rowi < (8u >> crow)) sv = *LVA_GLOBAL(lva_u32x4, prev + src + (uint64_t)rowi * g.sBlk + 2 * (tile * TS) + 4 * lane2);
  float pv = 0.0f;
  if (tid < 40) pv = LVA_GLOBAL(float, ss.post_row)[tid];
  TileTarget t0;
  const bool valid = tile_target<TS>(cd, g, ss, pos, tile, tid, &t0);   // role 0: the flip target of (base, conv)
  TileTarget t1 = t0;                                                    // role 1: the flop target of the same (base, conv)
  t1.k = t0.base + 4; t1.own = t0.own + (4u >> t0.c0) * (uint32_t)g.sCrf;     // (compact lists: flop list = flip list + 2)
  t1.ok = (t0.ok & 1u) | (((t0.reach >> t0.base) & 1u) << 1);
  lva_u32x2 s0 = {kNegInfBits, 0u}, s1 = {kNegInfBits, 0u};
  if (valid && (t0.ok & 1u)) {           // (the stay bit is the same for both)
    s0 = *LVA_GLOBAL(lva_u32x2, prev + t0.own + 2 * t0.c);
    s1 = *LVA_GLOBAL(lva_u32x2, prev + t1.own + 2 * t1.c);
  }
  *reinterpret_cast<lva_u32x4*>(&s_src[rowi * TS + 2 * lane2]) = sv;
  if (tid < 40) s_post[tid] = pv;
  __syncthreads();
  if (!valid) return;
  acs_pair<P>(g, prev, cur, s_src, s_post, src, true, t0, true, t1, s0, s1, crow);
}

// ---------------------------------------------------------------------------------------
// big-list fast kernel: list sizes 2 <= L <= 64 that lva_step_fast has no instance for
// (LL = 16, 32 or 64 >= L is the compile-time capacity).  Same butterfly tiling and the same
// one-target-per-thread tournament, but a list of 64 entries per state does not fit LDS or
// registers, so:
//   * list heads are read from HBM/L2 on demand -- one 8-byte (score, fingerprint) load per pop,
//     issued before the de-duplication scan that hides most of its latency;
//   * the only per-entry register state is the accepted fingerprints (LL registers); where each
//     accepted entry came from (list, index: 9 bits) and the fingerprint matches that still have to
//     be verified against it (at most two: a message can sit in at most three lists -- stay, flip X
//     and flop X of the base it ends in) go to LDS, one byte each, plus one bit each in registers;
//   * nothing is written inside the merge loop: afterwards all lanes walk the output rows in
//     lockstep -- entry l of 64 neighbouring conv states at a time -- re-read the source entry,
//     recompute score + transition exactly as the merge did, and store full 512-byte rows.
// Ties, non-finite sums, a third fingerprint match on one entry and fingerprint collisions send the
// target to lva_step_fixup_wave.
// ---------------------------------------------------------------------------------------
namespace {

constexpr uint32_t TSB = 32;        // source conv states per workgroup tile (workgroup = 8*TSB threads)
constexpr uint32_t kBigInFlight = 6;   // output phase at L > 32: entries whose loads are in flight together

template <int LL, int P, int NL>
__device__ __forceinline__ int big_merge(const Geometry& g, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur,
                                          const float* s_post, uint8_t* s_acc, uint8_t* s_rej0, uint8_t* s_rej1,
                                          const TileTarget& t, uint32_t src, uint32_t crow) {
  constexpr uint32_t NT = 8 * TSB;
  const float NEG = -INFINITY;
  const uint32_t L = g.L, N = g.N, sBlk = g.sBlk, sCrf = (uint32_t)g.sCrf, pw = 2 * g.N;
  const uint32_t k = t.k, row = k >= 4 ? 4u : k;
  const uint32_t own_c = t.own + 2 * t.c, src_c = src + 2 * t.cp;
  // word offset of the (score, fingerprint) pair of entry 0 of list i; its transition score
  // (crow: the source position stores compact lists -- crf state kk's list is list kk >> 1 there, Geometry::cmp)
  auto lbase = [&](uint32_t i) -> uint32_t { return i == 0 ? own_c : src_c + mul24(list_crf(k, i) >> crow, sCrf); };
  auto ladd = [&](uint32_t i) -> float { return s_post[row * 8 + (i == 0 ? k : list_crf(k, i))]; };
  int why = 0;

  // list heads (:750-761)
  float h[NL]; uint32_t hf[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    h[i] = NEG; hf[i] = 0;
    if ((t.ok >> i) & 1u) {
      const uint2 v = *reinterpret_cast<const uint2*>(prev + lbase(i));
      const bool ok = u2f(v.x) != NEG;
      h[i] = ok ? u2f(v.x) + ladd(i) : NEG;
      if (ok && !(h[i] > NEG)) why = 2;                  // non-finite sum: the exact path decides
      hf[i] = i ? v.y ^ t.fpc : v.y;
    }
  }

  uint32_t ah[LL];                 // accepted fingerprints, NEWEST FIRST (shift register: static indices only)
#pragma unroll
  for (int l = 0; l < LL; ++l) ah[l] = 0;
  unsigned long long ptr = 0;      // 7 bits per list: entries consumed
  // (list << 6 | index) of accepted entry a: low byte in s_acc[a], bit 8 in bit a of acc_hi.  The same
  // for the <= 2 fingerprint matches filed under entry a: s_rej0/1[a], rh0/1, and rv0/1 = slot in use.
  unsigned long long acc_hi = 0, rv0 = 0, rv1 = 0, rh0 = 0, rh1 = 0;
  uint32_t lc = 0;

  bool go = why == 0;
  while (go) {                                                         // :764
    float M = h[0];
#pragma unroll
    for (int i = 1; i < NL; ++i) M = fmaxf(M, h[i]);
    bool eq[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) eq[i] = h[i] == M;
    uint32_t sel = NL - 1, last = 0;   // first / last head equal to the maximum: they differ on a tie
#pragma unroll
    for (int i = NL - 2; i >= 0; --i) sel = eq[i] ? (uint32_t)i : sel;
#pragma unroll
    for (int i = 1; i < NL; ++i) last = eq[i] ? (uint32_t)i : last;
    const bool two = sel != last;
    const bool alive = M > NEG;            // false: every list exhausted (heap empty)
    const bool proceed = alive && !two;
    const uint32_t j = (uint32_t)(ptr >> (7 * sel)) & 127u;
    // next entry of the popped list: issue the load now, use it after the scan below (:788-796)
    const bool has_next = j + 1 < L;
    const uint32_t lb = lbase(sel);
    const uint2 nv = *reinterpret_cast<const uint2*>(prev + lb + mul24(has_next ? j + 1 : j, sBlk));
    uint32_t ch = hf[NL - 1];
#pragma unroll
    for (int i = NL - 2; i >= 0; --i) ch = selv(eq[i], hf[i], ch);
    // de-duplicate on fingerprints (:778-779): position q in ah <-> accepted entry lc-1-q
    int q = -1;
#pragma unroll
    for (int a = LL - 1; a >= 0; --a) q = ah[a] == ch ? a : q;
    const bool isdup = q >= 0 && (uint32_t)q < lc;
    const bool accept = proceed && !isdup, reject = proceed && isdup;
    const uint32_t from9 = (sel << 6) | j;
    const unsigned long long hi9 = (unsigned long long)(sel >> 2);
    if (accept) {
      s_acc[lc * NT] = (uint8_t)from9;
    }
    acc_hi |= accept ? hi9 << lc : 0ull;
    const uint32_t ra = lc - 1u - (uint32_t)q;                 // the entry it matched (only meaningful when reject)
    const unsigned long long rbit = 1ull << (ra & 63u);
    const bool use0 = reject && !(rv0 & rbit), use1 = reject && !use0 && !(rv1 & rbit), rej_full = reject && !use0 && !use1;
    if (use0) s_rej0[ra * NT] = (uint8_t)from9;
    if (use1) s_rej1[ra * NT] = (uint8_t)from9;
    rv0 |= use0 ? rbit : 0ull; rh0 |= (use0 && hi9) ? rbit : 0ull;
    rv1 |= use1 ? rbit : 0ull; rh1 |= (use1 && hi9) ? rbit : 0ull;
#pragma unroll
    for (int a = LL - 1; a >= 1; --a) ah[a] = selv(accept, ah[a - 1], ah[a]);
    ah[0] = selv(accept, ch, ah[0]);
    lc += accept ? 1u : 0u;
    // advance the popped list
    const float addsel = ladd(sel);
    const bool nxt_ok = has_next && u2f(nv.x) != NEG;
    const float ns = nxt_ok ? u2f(nv.x) + addsel : NEG;
    const bool bad = nxt_ok && !(ns > NEG);   // overflowed to -inf: the reference would still queue it
    const uint32_t nf = sel ? nv.y ^ t.fpc : nv.y;
#pragma unroll
    for (int i = 0; i < NL; ++i) { h[i] = selv(eq[i], ns, h[i]); hf[i] = selv(eq[i], nf, hf[i]); }
    ptr += 1ull << (7 * sel);
    why = (alive && two) ? 1 : ((proceed && bad) ? 2 : (rej_full ? 3 : 0));
    go = proceed && why == 0 && lc < L;
  }
  if (why) return why;

  // where entry `from9` of the previous step lives: message region base, conv state, planes in use
  auto locate = [&](uint32_t from9, uint32_t* i_out) -> uint32_t {
    const uint32_t i = from9 >> 6, j = from9 & 63u;
    *i_out = i;
    return (i == 0 ? t.own : src + mul24(list_crf(k, i) >> crow, sCrf)) + mul24(j, sBlk);
  };
  // outputs, entry l of the whole wavefront at a time (:771-774, :780-783, :799).  Every fingerprint
  // match filed under an entry must be the same message (else: collision, the exact path decides).
  bool good = true;
  constexpr uint32_t GB = LL >= 64 ? kBigInFlight : 4;       // entries whose loads are in flight together (VGPRs: the fingerprints are dead by now)
  for (uint32_t l0 = 0; l0 < L; l0 += GB) {
    uint32_t m[GB][2 * P], q0[GB][2 * P]; uint2 sh2[GB]; uint32_t iu[GB], ir[GB];
#pragma unroll
    for (uint32_t u = 0; u < GB; ++u) {
      const uint32_t l = l0 + u;
      iu[u] = 0; ir[u] = 0; sh2[u] = make_uint2(kNegInfBits, 0u);
      if (l < lc) {
        const uint32_t f = locate((uint32_t)s_acc[l * NT] | ((uint32_t)(acc_hi >> l) & 1u) << 8, &iu[u]);
        const uint32_t cv = iu[u] == 0 ? t.c : t.cp;
        sh2[u] = *reinterpret_cast<const uint2*>(prev + f + 2 * cv);
        load_msg<P>(prev + f + pw, N, cv, iu[u] == 0 ? t.np_dst : t.np_src, m[u]);
        if ((rv0 >> l) & 1ull) {          // the first match filed under this entry: its load travels with the others
          const uint32_t fr = locate((uint32_t)s_rej0[l * NT] | ((uint32_t)(rh0 >> l) & 1u) << 8, &ir[u]);
          load_msg<P>(prev + fr + pw, N, ir[u] == 0 ? t.c : t.cp, ir[u] == 0 ? t.np_dst : t.np_src, q0[u]);
        }
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < GB; ++u) {
      const uint32_t l = l0 + u;
      if (l < lc) {
        const float sc = u2f(sh2[u].x) + ladd(iu[u]);
        *reinterpret_cast<uint2*>(cur + own_c + mul24(l, sBlk)) = make_uint2(f2u(sc), iu[u] ? sh2[u].y ^ t.fpc : sh2[u].y);
        push_var<2 * P>(m[u], iu[u] == 0 ? 0u : t.sh, iu[u] == 0 ? 0u : t.nb);
        store_msg<P>(cur + t.own + mul24(l, sBlk) + pw, N, t.c, t.np_dst, m[u]);
        if ((rv0 >> l) & 1ull) {
          push_bits<2 * P>(q0[u], ir[u] == 0 ? 0u : t.sh, t.nb);
#pragma unroll
          for (int w = 0; w < 2 * P; ++w) good &= (q0[u][w] == m[u][w]);
          if ((rv1 >> l) & 1ull) {        // a second match on the same entry is rare: loaded here
            uint32_t i1;
            const uint32_t fr = locate((uint32_t)s_rej1[l * NT] | ((uint32_t)(rh1 >> l) & 1u) << 8, &i1);
            uint32_t qm[2 * P];
            load_msg<P>(prev + fr + pw, N, i1 == 0 ? t.c : t.cp, i1 == 0 ? t.np_dst : t.np_src, qm);
            push_bits<2 * P>(qm, i1 == 0 ? 0u : t.sh, t.nb);
#pragma unroll
            for (int w = 0; w < 2 * P; ++w) good &= (qm[w] == m[u][w]);
          }
        }
      } else if (l < L) {
        *reinterpret_cast<uint2*>(cur + own_c + mul24(l, sBlk)) = make_uint2(kNegInfBits, 0u);
      }
    }
  }
  return good ? 0 : 4;
}

}  // namespace

// grid: x = tiles of TSB source conv states, y = band position index, z = slot index; 256 threads.
template <int LL, int P>
__global__ __launch_bounds__(8 * TSB) void lva_step_big(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                      uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                      uint32_t* __restrict__ items) {
  __shared__ uint8_t s_acc[LL * 8 * TSB], s_rej0[LL * 8 * TSB], s_rej1[LL * 8 * TSB];
  __shared__ float s_post[40];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    hdr->count[args.step_parity ^ 1u] = 0;     // the other parity's list was consumed by the last fix-up
    hdr->overflow[args.step_parity ^ 1u] = 0;
  }
  SlotStep ss;
  if (!load_slot(args, blockIdx.z, &ss)) return;
  const uint32_t pos = ss.lo + blockIdx.y;
  if (pos >= ss.hi) return;
  const DevCode& cd = codes[ss.orient];
  const uint32_t N = cd.nconv, tid = threadIdx.x, tile = blockIdx.x;
  const uint32_t* prev; uint32_t* cur;
  slot_buffers(ss, g, trellis, &prev, &cur);

  if (pos == 0) {                          // stay-only update of the 8 start states (:706-713)
    if (tile == cd.init / TSB && tid < 8) {
      const uint32_t k = tid, c = cd.init;
      const uint32_t own_c = (uint32_t)((uint64_t)k * g.sCrf) + 2 * c;
      const float s = u2f(prev[own_c]) + ss.post_row[(k >= 4 ? 4u : k) * 8 + k];
      cur[own_c] = f2u(s);
      cur[own_c + 1] = prev[own_c + 1];
      cur[own_c + 2 * N] = prev[own_c + 2 * N];          // plane 1 (the only one in use at position 0)
      cur[own_c + 2 * N + 1] = prev[own_c + 2 * N + 1];
      for (uint32_t l = 1; l < g.L; ++l) cur[own_c + l * g.sBlk] = kNegInfBits;
    }
    return;
  }
  if (tid < 40) s_post[tid] = LVA_GLOBAL(float, ss.post_row)[tid];
  __syncthreads();

  const uint32_t src = (uint32_t)((uint64_t)((pos - 1) % g.R) * 8 * g.sCrf);
  TileTarget t;
  if (!tile_target<TSB>(cd, g, ss, pos, tile, tid, &t)) return;
  const uint32_t crow = source_compact(g, cd, ss, blockIdx.y, pos);
  const int why = t.k < 4 ? big_merge<LL, P, 8>(g, prev, cur, s_post, s_acc + tid, s_rej0 + tid, s_rej1 + tid, t, src, crow)
                          : big_merge<LL, P, 2>(g, prev, cur, s_post, s_acc + tid, s_rej0 + tid, s_rej1 + tid, t, src, crow);
  if (why) {
    atomicAdd(&hdr->reason[why - 1], 1ull);
    const uint32_t idx = atomicAdd(&hdr->count[args.step_parity], 1u);
    if (idx < hdr->cap) items[idx] = make_item(cd.m, blockIdx.z, blockIdx.y, t.k, t.c);
    else hdr->overflow[args.step_parity] = 1u;
  }
}


// ---------------------------------------------------------------------------------------
// Big-list kernel on the RECORD layout (Geometry::rec: three message planes, 32 <= L <= 64, L a multiple of 4).
//
// A (ring, crf) list is [conv][entry] records of 2 + 2 np words -- score, fingerprint, the 2 np message words in use at the
// list's trellis position (np = 1, 2, 3: 16, 24, 32 bytes; a position in the first third of a read moves half the bytes of
// one in the last third) -- so the entries of ONE conv state's list are adjacent:
//   * a thread that walks a list (the merge pops ~46 of a target's 64 entries from one source list) pulls a line per
//     four to eight entries instead of one per entry, and the lines it pulls are the ones the output phase needs (a packed copy
//     of the pairs for the walk made the output's gathers cold and the kernel 22 % slower: round 5);
//   * an accepted entry is ONE line (one or two loads: score and fingerprint come with the message) instead of three;
//   * a target's output is contiguous.  Threads cannot store it themselves (64 partial lines per instruction: measured 1.6x
//     slower in round 2; an 8-byte store per lane and round inside the merge loop wrote 12.5 GB instead of 7.4: round 5): each
//     wavefront passes four entries of its 64 targets through LDS and stores runs of 64 / 96 / 128 bytes per target, 16 bytes
//     per lane.  For that the wavefront stays whole: threads without a target, and threads whose target goes to the work list,
//     keep running as store helpers.
//   * The output phase requests the four records of a round with ALL lanes, in straight-line code, before it waits for any: a
//     load behind a per-lane branch has an unknown place in the memory queue and the compiler then waits for everything
//     (s_waitcnt vmcnt(0)) in front of the next write to its register -- four round trips per round instead of one (round 5).
// One fingerprint-match slot per accepted entry (a second match is reason 3: the exact path decides).
// Same decisions as big_merge otherwise; ties, non-finite sums and collisions go to lva_step_fixup_wave, which reads and
// writes the same layout (rec_sh / rec_word).
// ---------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kTrRow = 36;      // words per thread in the transpose buffer: 4 records + 4 words of padding (conflict-free b128)
typedef unsigned int lva_u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));   // 16 bytes at an 8-byte aligned address (24-byte records)

template <int LL, int NL>
__device__ __forceinline__ int big_merge_rec(const Geometry& g, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur,
                                              const float* s_post, uint8_t* s_acc, uint8_t* s_rej0, uint32_t* s_tr, uint32_t* s_base,
                                              bool valid, const TileTarget& t, uint32_t src, uint32_t npd, uint32_t nps) {
  constexpr uint32_t NT = 8 * TSB;
  const float NEG = -INFINITY;
  const uint32_t L = g.L, sCrf = (uint32_t)g.sCrf;
  const uint32_t lane = threadIdx.x & 63u, wrow = threadIdx.x & ~63u;     // first thread of this wavefront
  const uint32_t k = t.k, row = k >= 4 ? 4u : k;
  // records of 2 + 2 np words: np = message planes in use at the list's position -- the own (stay) list lives at pos, the source
  // lists at pos - 1 (one plane less at the two or three positions where the message crosses a multiple of 64 bits)
  const uint32_t rws = 2u + 2u * nps, rwd = 2u + 2u * npd;     // (npd, nps: from the position record, uniform over the workgroup)
  const uint32_t own_r = t.own + mul24(t.c, L) * rwd, src_r = src + mul24(t.cp, L) * rws;   // record 0 of the own / of a source list 0
  auto lrec = [&](uint32_t i) -> uint32_t { return i == 0 ? own_r : src_r + mul24(list_crf(k, i), sCrf); };
  auto lrw = [&](uint32_t i) -> uint32_t { return i == 0 ? rwd : rws; };
  auto ladd = [&](uint32_t i) -> float { return s_post[row * 8 + (i == 0 ? k : list_crf(k, i))]; };
  int why = 0;
  uint32_t lc = 0;
  unsigned long long acc_hi = 0, rv0 = 0, rh0 = 0;

  if (valid) {
    float h[NL]; uint32_t hf[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {                                       // list heads (:750-761)
      h[i] = NEG; hf[i] = 0;
      if ((t.ok >> i) & 1u) {
        const lva_u32x2 v = *LVA_GLOBAL(lva_u32x2, prev + lrec(i));
        const bool ok = u2f(v.x) != NEG;
        h[i] = ok ? u2f(v.x) + ladd(i) : NEG;
        if (ok && !(h[i] > NEG)) why = 2;                  // non-finite sum: the exact path decides
        hf[i] = i ? v.y ^ t.fpc : v.y;
      }
    }
    uint32_t ah[LL];                 // accepted fingerprints, NEWEST FIRST (shift register: static indices only)
#pragma unroll
    for (int l = 0; l < LL; ++l) ah[l] = 0;
    unsigned long long ptr = 0;      // 7 bits per list: entries consumed
    bool rej_full = false;
    // one pop (:766-785) of entry (sel, jj) with fingerprint fp, for the threads in `pred`: de-duplicate on fingerprints
    // (position q in ah <-> accepted entry lc-1-q), file the entry or the match.  Entries at positions >= lc are unused (zero,
    // never a valid match), so the scan and the shift only touch the 16-entry segments some thread of the wavefront has reached.
    auto pop = [&](bool pred, uint32_t sel, uint32_t jj, uint32_t fp) __attribute__((always_inline)) {
      bool seg[LL / 16];
#pragma unroll
      for (int sg = 1; sg < LL / 16; ++sg) seg[sg] = __ballot(lc >= 16u * sg) != 0ull;
      int q = -1;
#pragma unroll
      for (int sg = LL / 16 - 1; sg >= 1; --sg)
        if (seg[sg]) {
#pragma unroll
          for (int a = 16 * sg + 15; a >= 16 * sg; --a) q = ah[a] == fp ? a : q;
        }
#pragma unroll
      for (int a = 15; a >= 0; --a) q = ah[a] == fp ? a : q;
      const bool isdup = q >= 0 && (uint32_t)q < lc;
      const bool accept = pred && !isdup, reject = pred && isdup;
      const uint32_t from9 = (sel << 6) | jj;
      const unsigned long long hi9 = (unsigned long long)(sel >> 2);
      if (accept) s_acc[lc * NT] = (uint8_t)from9;
      acc_hi |= accept ? hi9 << lc : 0ull;
      const uint32_t ra = lc - 1u - (uint32_t)q;                 // the entry it matched (only meaningful when reject)
      const unsigned long long rbit = 1ull << (ra & 63u);
      const bool use0 = reject && !(rv0 & rbit);
      rej_full = rej_full || (reject && !use0);
      if (use0) s_rej0[ra * NT] = (uint8_t)from9;
      rv0 |= use0 ? rbit : 0ull; rh0 |= (use0 && hi9) ? rbit : 0ull;
#pragma unroll
      for (int sg = LL / 16 - 1; sg >= 1; --sg)
        if (seg[sg]) {
#pragma unroll
          for (int a = 16 * sg + 15; a >= 16 * sg; --a) ah[a] = selv(accept, ah[a - 1], ah[a]);
        }
#pragma unroll
      for (int a = 15; a >= 1; --a) ah[a] = selv(accept, ah[a - 1], ah[a]);
      ah[0] = selv(accept, fp, ah[0]);
      lc += accept ? 1u : 0u;
    };
    bool go = why == 0;
    while (go) {                                                         // :764
      float M = h[0];
#pragma unroll
      for (int i = 1; i < NL; ++i) M = fmaxf(M, h[i]);
      bool eq[NL];
#pragma unroll
      for (int i = 0; i < NL; ++i) eq[i] = h[i] == M;
      uint32_t sel = NL - 1, last = 0;   // first / last head equal to the maximum: they differ on a tie
#pragma unroll
      for (int i = NL - 2; i >= 0; --i) sel = eq[i] ? (uint32_t)i : sel;
#pragma unroll
      for (int i = 1; i < NL; ++i) last = eq[i] ? (uint32_t)i : last;
      const bool two = sel != last;
      const bool alive = M > NEG;            // false: every list exhausted (heap empty)
      const bool proceed = alive && !two;
      const uint32_t j = (uint32_t)(ptr >> (7 * sel)) & 127u;
      // next entry of the popped list -- the neighbouring record: requested now, used after the scan (:788-796)
      const bool has_next = j + 1 < L;
      const lva_u32x2 nv = *LVA_GLOBAL(lva_u32x2, prev + lrec(sel) + mul24(has_next ? j + 1 : j, lrw(sel)));
      uint32_t ch = hf[NL - 1];
#pragma unroll
      for (int i = NL - 2; i >= 0; --i) ch = selv(eq[i], hf[i], ch);
      pop(proceed, sel, j, ch);
      const float addsel = ladd(sel);
      const bool nxt_ok = has_next && u2f(nv.x) != NEG;
      const float ns = nxt_ok ? u2f(nv.x) + addsel : NEG;
      const bool bad = nxt_ok && !(ns > NEG);   // overflowed to -inf: the reference would still queue it
      const uint32_t nf = sel ? nv.y ^ t.fpc : nv.y;
#pragma unroll
      for (int i = 0; i < NL; ++i) { h[i] = selv(eq[i], ns, h[i]); hf[i] = selv(eq[i], nf, hf[i]); }
      ptr += 1ull << (7 * sel);
      why = (alive && two) ? 1 : ((proceed && bad) ? 2 : (rej_full ? 3 : 0));
      go = proceed && why == 0 && lc < L;
    }
  }

  // ---- outputs: the whole wavefront, four entries of every target at a time (:771-774, :780-783, :799) ----
  const bool act = valid && why == 0;                     // this thread's target is written here (else: not stored / the exact path)
  s_base[threadIdx.x] = act ? own_r : 0xFFFFFFFFu;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  auto locate = [&](uint32_t from9, uint32_t* i_out) -> uint32_t {   // record of entry `from9` of the previous step
    const uint32_t i = from9 >> 6, j = from9 & 63u;
    *i_out = i;
    return lrec(i) + mul24(j, lrw(i));
  };
  bool good = true;
  // the transpose buffer holds the four records of HALF a wavefront's targets (32 rows per wavefront): 51 KB of LDS per
  // workgroup instead of 70 KB -- three workgroups per CU, which is what this latency-bound kernel needs (12 wavefronts per CU)
  const uint32_t half = lane >> 5;
  uint32_t* trow = s_tr + ((wrow >> 1) + (lane & 31u)) * kTrRow;
  // One round = four entries of every target of the wavefront.  NP = the plane count at pos (uniform; compile-time inside the
  // instance).  At the two or three positions where pos - 1 has one plane less (`mixed`), an entry of a source list is read as if
  // it had NP planes -- every lane the same straight-line loads -- and its top plane (the first words of the next record) is
  // replaced by the zeros it stands for.
  const bool mixed = nps != npd;
  auto rounds = [&](auto npc) __attribute__((always_inline)) {
    constexpr uint32_t NP = decltype(npc)::value;
    constexpr uint32_t RW = 2u + 2u * NP;                 // words per record of the target's list
    for (uint32_t l0 = 0; l0 < L; l0 += 4) {
      // Every request of the round goes out before anything is waited for, and the four entries' records are requested and
      // consumed by ALL lanes (a lane without an entry reads the first record of the buffer and ignores it): a load that is issued
      // on some paths only has an unknown place in the queue, the compiler then waits for EVERYTHING in front of the next write to
      // its register -- the four entries of a round used to be four round trips one after the other.
      lva_u32x4_a8 ra4[4], qa[4]; lva_u32x2 rb2[4], rc2[4], qb[4], qc[4]; uint32_t iu[4], ir[4]; bool on[4], chk[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t l = l0 + u;
        on[u] = act && l < lc;
        uint32_t i0 = 0;
        const uint32_t f = locate((uint32_t)s_acc[l * NT] | ((uint32_t)(acc_hi >> l) & 1u) << 8, &i0);
        iu[u] = on[u] ? i0 : 0u;
        const uint32_t* rp = prev + (on[u] ? f : 0u);
        ra4[u] = *LVA_GLOBAL(lva_u32x4_a8, rp);           // score, fingerprint, words 0-1
        rb2[u] = lva_u32x2{0u, 0u}; rc2[u] = lva_u32x2{0u, 0u};
        if constexpr (NP == 2) rb2[u] = *LVA_GLOBAL(lva_u32x2, rp + 4);       // words 2-3
        if constexpr (NP == 3) {                                               // words 2-5: one load (32-byte records are 16-byte aligned)
          const lva_u32x4_a8 v = *LVA_GLOBAL(lva_u32x4_a8, rp + 4);         // (8-byte aligned where pos - 1 has 24-byte records)
          rb2[u] = lva_u32x2{v.x, v.y}; rc2[u] = lva_u32x2{v.z, v.w};
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {                  // the match filed under an entry (one in ten): its loads travel with the others
        const uint32_t l = l0 + u;
        chk[u] = on[u] && ((rv0 >> l) & 1ull);
        ir[u] = 0; qa[u] = lva_u32x4_a8{0u, 0u, 0u, 0u}; qb[u] = lva_u32x2{0u, 0u}; qc[u] = lva_u32x2{0u, 0u};
        if (chk[u]) {
          const uint32_t fr = locate((uint32_t)s_rej0[l * NT] | ((uint32_t)(rh0 >> l) & 1u) << 8, &ir[u]);
          qa[u] = *LVA_GLOBAL(lva_u32x4_a8, prev + fr);
          if constexpr (NP == 2) qb[u] = *LVA_GLOBAL(lva_u32x2, prev + fr + 4);
          if constexpr (NP == 3) {
            const lva_u32x4_a8 v = *LVA_GLOBAL(lva_u32x4_a8, prev + fr + 4);
            qb[u] = lva_u32x2{v.x, v.y}; qc[u] = lva_u32x2{v.z, v.w};
          }
        }
      }
      uint32_t ow[4][8];                                  // the four output records (2 + 2 np words each)
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        uint32_t m[6] = {ra4[u].z, ra4[u].w, rb2[u].x, rb2[u].y, rc2[u].x, rc2[u].y};
        if (mixed && iu[u] != 0) { m[2 * NP - 2] = 0; m[2 * NP - 1] = 0; }
        const float sc = u2f(ra4[u].x) + ladd(iu[u]);
        push_var<6>(m, iu[u] == 0 ? 0u : t.sh, iu[u] == 0 ? 0u : t.nb);
        // (:799) the unused tail of the list: -inf, empty message
        ow[u][0] = on[u] ? f2u(sc) : kNegInfBits; ow[u][1] = on[u] ? (iu[u] ? ra4[u].y ^ t.fpc : ra4[u].y) : 0u;
#pragma unroll
        for (int w = 0; w < 6; ++w) ow[u][2 + w] = on[u] ? m[w] : 0u;
        uint32_t q[6] = {qa[u].z, qa[u].w, qb[u].x, qb[u].y, qc[u].x, qc[u].y};
        if (mixed && ir[u] != 0) { q[2 * NP - 2] = 0; q[2 * NP - 1] = 0; }
        push_var<6>(q, ir[u] == 0 ? 0u : t.sh, ir[u] == 0 ? 0u : t.nb);
        uint32_t diff = 0;
#pragma unroll
        for (int w = 0; w < 6; ++w) diff |= q[w] ^ m[w];
        good &= !(chk[u] && diff != 0);
      }
      // out in runs of 4 RW words per target, half a wavefront's targets per round: 32 targets x RW pieces of 16 bytes
#pragma unroll
      for (uint32_t h2 = 0; h2 < 2; ++h2) {
        if (half == h2) {
#pragma unroll
          for (uint32_t u = 0; u < 4; ++u)
#pragma unroll
            for (uint32_t w = 0; w < 8; w += 2)
              if (w < RW) *reinterpret_cast<lva_u32x2*>(trow + u * RW + w) = lva_u32x2{ow[u][w], ow[u][w + 1]};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t tb[4], td[4]; lva_u32x4 tv[4];           // (all LDS reads requested before the first store waits for one)
#pragma unroll
        for (uint32_t s8 = 0; s8 < 4; ++s8) {
          if (s8 * 2 >= RW) break;                        // (32 targets x RW pieces: RW / 2 instructions)
          const uint32_t p2 = 64 * s8 + lane;             // 16-byte piece p2 of the half's 32 RW pieces
          const uint32_t Tl = NP == 1 ? p2 >> 2 : NP == 2 ? (p2 * 171u) >> 10 : p2 >> 3, piece = p2 - Tl * RW;
          const bool in = p2 < 32u * RW;
          tb[s8] = in ? s_base[wrow + 32 * h2 + (Tl & 31u)] : 0xFFFFFFFFu;
          td[s8] = l0 * RW + 4 * piece;
          tv[s8] = *reinterpret_cast<const lva_u32x4*>(s_tr + ((wrow >> 1) + (Tl & 31u)) * kTrRow + 4 * (in ? piece : 0u));
        }
#pragma unroll
        for (uint32_t s8 = 0; s8 < 4; ++s8) {
          if (s8 * 2 >= RW) break;
          if (tb[s8] != 0xFFFFFFFFu) {
            uint32_t* dst = cur + tb[s8] + td[s8];
            __builtin_nontemporal_store(tv[s8].x, dst); __builtin_nontemporal_store(tv[s8].y, dst + 1);
            __builtin_nontemporal_store(tv[s8].z, dst + 2); __builtin_nontemporal_store(tv[s8].w, dst + 3);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  };
  if (npd == 1) rounds(std::integral_constant<uint32_t, 1>{});
  else if (npd == 2) rounds(std::integral_constant<uint32_t, 2>{});
  else rounds(std::integral_constant<uint32_t, 3>{});
  if (!valid) return 0;
  if (why) return why;
  return good ? 0 : 4;
}

}  // namespace

// grid: x = tiles of TSB source conv states, y = band position index, z = slot index; 256 threads (as lva_step_big).
template <int LL>
__global__ __launch_bounds__(8 * TSB) void lva_step_big_rec(StepArgs args, Geometry g, const DevCode* __restrict__ codes,
                                                          uint32_t* __restrict__ trellis, WorkHdr* __restrict__ hdr,
                                                          uint32_t* __restrict__ items) {
  __shared__ uint8_t s_acc[LL * 8 * TSB], s_rej0[LL * 8 * TSB];
  __shared__ __attribute__((aligned(16))) uint32_t s_tr[4 * TSB * kTrRow];        // 32 rows per wavefront
  __shared__ uint32_t s_base[8 * TSB];
  __shared__ float s_post[40];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    hdr->count[args.step_parity ^ 1u] = 0;     // the other parity's list was consumed by the last fix-up
    hdr->overflow[args.step_parity ^ 1u] = 0;
  }
  SlotStep ss;
  if (!load_slot(args, blockIdx.z, &ss)) return;
  const uint32_t pos = ss.lo + blockIdx.y;
  if (pos >= ss.hi) return;
  const DevCode& cd = codes[ss.orient];
  const uint32_t tid = threadIdx.x, tile = blockIdx.x;
  const uint32_t* prev; uint32_t* cur;
  slot_buffers(ss, g, trellis, &prev, &cur);

  if (pos == 0) {                          // stay-only update of the 8 start states (:706-713)
    if (tile == cd.init / TSB && tid < 8) {
      const uint32_t k = tid, c = cd.init;
      const uint32_t np0 = cd.npair[0];
      const uint32_t lst = (uint32_t)((uint64_t)k * g.sCrf), r0 = rec_sh(g, lst, c, 0, np0);
      const float s = u2f(prev[r0]) + ss.post_row[(k >= 4 ? 4u : k) * 8 + k];
      cur[r0] = f2u(s);
      for (uint32_t w = 1; w < 2 + 2 * np0; ++w) cur[r0 + w] = prev[r0 + w];      // fingerprint and the (empty) message
      for (uint32_t l = 1; l < g.L; ++l) cur[rec_sh(g, lst, c, l, np0)] = kNegInfBits;
    }
    return;
  }
  if (tid < 40) s_post[tid] = LVA_GLOBAL(float, ss.post_row)[tid];
  __syncthreads();

  const uint32_t src = (uint32_t)((uint64_t)((pos - 1) % g.R) * 8 * g.sCrf);
  TileTarget t;
  t.c = 0; t.cp = 0; t.sc = 0; t.k = (tid / (4 * TSB)) * 4; t.sh = 1; t.nb = 0; t.fpc = 0; t.np_dst = 1; t.np_src = 1; t.ok = 0; t.own = 0;
  t.base = 0; t.reach = 0; t.pk1 = 0;
  const bool valid = tile_target<TSB>(cd, g, ss, pos, tile, tid, &t);
  // (wavefronts 0-1 hold the flip targets, 2-3 the flop targets: the merge width is uniform per wavefront)
  const uint32_t pinfo = cd.rec[pos].info, npd = (pinfo >> 16) & 0xFFu, nps = pinfo >> 24;      // message planes in use at pos / at pos - 1
  const int why = tid < 4 * TSB ? big_merge_rec<LL, 8>(g, prev, cur, s_post, s_acc + tid, s_rej0 + tid, s_tr, s_base, valid, t, src, npd, nps)
                                : big_merge_rec<LL, 2>(g, prev, cur, s_post, s_acc + tid, s_rej0 + tid, s_tr, s_base, valid, t, src, npd, nps);
  if (why) {
    atomicAdd(&hdr->reason[why - 1], 1ull);
    const uint32_t idx = atomicAdd(&hdr->count[args.step_parity], 1u);
    if (idx < hdr->cap) items[idx] = make_item(cd.m, blockIdx.z, blockIdx.y, t.k, t.c);
    else hdr->overflow[args.step_parity] = 1u;
  }
}

// One thread per read slot: resolve descriptor -> time step -> band of this launch into the SlotStep record the
// step kernels read (small trellises run hundreds of slots per launch, and every workgroup of a slot would otherwise
// walk the same chain of dependent loads, cold, by itself).
__global__ void lva_prepare_step(StepArgs a, const DevCode* __restrict__ codes, SlotStep* __restrict__ steps) {
  const uint32_t z = blockIdx.x * blockDim.x + threadIdx.x;
  if (z >= a.nslots) return;
  const SlotDesc d = a.slots[z];
  const uint32_t t = a.launch_no - d.start;
  SlotStep ss;
  ss.post_row = nullptr; ss.slot = z; ss.t = 0xFFFFFFFFu; ss.lo = 0; ss.hi = 0; ss.prev_hi = 0; ss.orient = 0; ss.flags = 0; ss.pad = 0;
  ss.srccmp[0] = 0; ss.srccmp[1] = 0;
  if (t < d.nblk) {
    const uint32_t b = d.band[t];
    ss.post_row = d.post + (size_t)t * 40;
    ss.t = t; ss.lo = b & 0xFFFFu; ss.hi = (b >> 16) & 0x3FFFu; ss.flags = b >> 30;
    ss.prev_hi = t ? (d.band[t - 1] >> 16) & 0x3FFFu : 1u;       // what step t-1 wrote (only position 0 is initialised at t = 0)
    ss.orient = d.orient;
    const DevCode& cd = codes[d.orient];          // (the masks are only read where the geometry has compact lists)
    for (uint32_t p = ss.lo < 2u ? 2u : ss.lo; p < ss.hi && p - ss.lo < 64u; ++p)
      if (cd.ptype[p - 1] == 0) ss.srccmp[(p - ss.lo) >> 5] |= 1u << ((p - ss.lo) & 31u);
  }
  steps[z] = ss;
}

// (:657-663) score 0 at (pos 0, initial conv state, every crf state, list entry 0), empty message
__global__ void lva_init_slot(Geometry g, const DevCode* __restrict__ codes, uint32_t* __restrict__ trellis,
                              InitBatch batch, SlotDesc* __restrict__ slots) {
  if (blockIdx.x >= batch.n) return;
  const uint32_t slot = batch.slot[blockIdx.x];
  const SlotDesc desc = batch.desc[blockIdx.x];
  const uint32_t orient = desc.orient;
  if (threadIdx.x == 0) slots[slot] = desc;             // the read enters the slot: later step launches see it
  const DevCode& cd = codes[orient];
  uint32_t* par0 = trellis + (uint64_t)slot * g.sSlot;   // parity 0 is "prev" at t = 0
  const uint32_t n = 8 * g.L * g.F;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t f = i % g.F, l = (i / g.F) % g.L, k = i / (g.F * g.L);
    const uint64_t blk = (uint64_t)k * g.sCrf + (uint64_t)l * g.sBlk;     // ring slot 0 = position 0
    if (g.rec && f >= 2 + 2 * cd.npair[0]) continue;          // (record layout: only the words in use at position 0 exist)
    const uint64_t at = g.rec ? (f < 2 ? rec_sh(g, (uint32_t)((uint64_t)k * g.sCrf), cd.init, l, cd.npair[0]) + f
                                       : rec_word(g, (uint32_t)((uint64_t)k * g.sCrf), cd.init, l, f - 2, cd.npair[0]))
                              : blk + plane_off(g, f >> 1, cd.init) + (f & 1u);
    par0[at] = (f == 0 && l > 0) ? kNegInfBits : 0u;
  }
}

// copy the lists of (last position, final conv state, crf 0..7) into the result record
// [crf][l][score, fingerprint, message words], writing -inf scores for crf states that are
// not stored (:806-815 reads them as -inf)
__global__ void lva_gather_final(Geometry g, const DevCode* __restrict__ codes, const uint32_t* __restrict__ trellis,
                                 GatherBatch batch, uint32_t* __restrict__ results) {
  if (blockIdx.x >= batch.n) return;
  const GatherArgs a = batch.a[blockIdx.x];
  const DevCode& cd = codes[a.orient];
  const uint32_t pos = cd.npos - 1, c = cd.fin;
  const uint32_t* buf = trellis + (uint64_t)a.slot * g.sSlot + (uint64_t)a.parity * g.sPar;
  if (g.lazy) {      // kernel mode 4: the last step's entries hold messages (even last step) or back-pointer bytes (odd)
    const uint32_t* base = trellis + (uint64_t)a.slot * g.sSlot;
    const uint32_t tl = a.nblk - 1, n = 8 * g.L * g.F;
    uint32_t reach = 0;
    const uint32_t T = cd.ptype[pos], pk = cd.predtab[T][c], sh = T == 0 ? 1u : 2u;
    for (int b = 0; b < 4; ++b) if ((pk >> (4 * b)) & 8u) reach |= (0x11u << b);
    if ((c & cd.vmask[pos]) != cd.vval[pos]) reach = 0;
    const uint32_t nbp = sh == 1 ? (c >> (cd.m - 1)) : (2 * ((c >> (cd.m - 2)) & 1u) + (c >> (cd.m - 1)));
    uint32_t* out = results + (uint64_t)a.read * n;
    for (uint32_t e = threadIdx.x; e < 8 * g.L; e += blockDim.x) {
      const uint32_t k = e / g.L, l = e % g.L;
      uint32_t w[2 + 8];
      w[0] = kNegInfBits;
      for (uint32_t f = 1; f < g.F; ++f) w[f] = 0;
      const uint32_t own = (uint32_t)(((uint64_t)(pos % g.R) * 8 + (k >> compact_pos(cd, g, pos))) * g.sCrf);
      if ((reach >> k) & 1u) {
        w[0] = buf[own + l * g.sBlk + 2 * c]; w[1] = buf[own + l * g.sBlk + 2 * c + 1];
        if (w[0] != kNegInfBits) {
          if (!(tl & 1u)) {                      // messages stored by the last (anchor) step in buffer (tl >> 1) & 1
            const uint32_t* mb = base + (uint64_t)((tl >> 1) & 1u) * g.sPar;
            for (uint32_t f = 2; f < g.F; ++f) w[f] = msg_word(g, mb, own + l * g.sBlk, c, f - 2, cd.npair[pos]);
          } else {                               // one hop back: the entry's source in step tl-1
            const uint32_t bp = reinterpret_cast<const uint8_t*>(buf)[bp_byte_index(g, own, l, c)];
            const uint32_t i = (bp >> 3) & 7u, j = bp & 7u;
            const uint32_t* mb = base + (uint64_t)((bp >> 6) & 1u) * g.sPar;
            uint32_t lst = own, conv = c, np = cd.npair[pos], s1 = 0;
            if (i != 0) {
              const uint32_t y = (pk >> (4 * (k & 3u))) & 7u;
              conv = ((c << sh) | y) & (cd.nconv - 1);
              lst = (uint32_t)(((uint64_t)((pos - 1) % g.R) * 8 + (list_crf(k, i) >> compact_pos(cd, g, pos - 1))) * g.sCrf);
              np = cd.npair[pos - 1]; s1 = sh;
            }
            uint32_t carry = s1 ? nbp : 0u;
            for (uint32_t f = 2; f < g.F; ++f) {
              const uint32_t v = msg_word(g, mb, lst + j * g.sBlk, conv, f - 2, np);
              w[f] = s1 ? ((v << s1) | carry) : v;
              carry = s1 ? (v >> (32 - s1)) : 0u;
            }
          }
        }
      }
      for (uint32_t f = 0; f < g.F; ++f) out[e * g.F + f] = w[f];
    }
    return;
  }
  uint32_t reach = 0xFFu;
  if (pos > 0) {
    reach = 0;
    const uint32_t pk = cd.predtab[cd.ptype[pos]][c];
    for (int b = 0; b < 4; ++b) if ((pk >> (4 * b)) & 8u) reach |= (0x11u << b);
  }
  if ((c & cd.vmask[pos]) != cd.vval[pos]) reach = 0;
  const uint32_t n = 8 * g.L * g.F, np = cd.npair[pos];
  uint32_t* out = results + (uint64_t)a.read * n;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t f = i % g.F, l = (i / g.F) % g.L, k = i / (g.F * g.L);
    uint32_t v = f == 0 ? kNegInfBits : 0u;
    if (((reach >> k) & 1u) && (f < 2 || ((f - 2) >> 1) < np)) {
      const uint32_t lst = (uint32_t)(((uint64_t)(pos % g.R) * 8 + (k >> compact_pos(cd, g, pos))) * g.sCrf);
      v = f < 2 ? buf[rec_sh(g, lst, c, l, np) + f] : entry_word(g, buf, lst, c, l, f - 2, np);
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

int launch_step_wave(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, void* stream) {
  if (a.nslots == 0 || a.band_max == 0) return 0;
  dim3 grid((g.N + 3) / 4, a.band_max * 8, a.nslots), block(256);
  hipLaunchKernelGGL(lva_step_wave, grid, block, 0, (hipStream_t)stream, a, g, codes, trellis);
  return (int)hipGetLastError();
}

bool wave_kernel_available(const Geometry& g) { return g.L >= 2 && g.L <= 64; }

static bool small_list(const Geometry& g) { return g.L == 1 || g.L == 2 || g.L == 4 || g.L == 8; }

bool fast_kernel_available(const Geometry& g) {
  const bool l_ok = small_list(g) || (g.L >= 2 && g.L <= 64);
  return l_ok && g.P >= 1 && g.P <= 4 && g.N >= 64;
}

template <int LL>
static int launch_fast_p(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, WorkHdr* hdr,
                         uint32_t* items, hipStream_t st) {
  dim3 grid(g.N / TS, a.band_max, a.nslots), block(8 * TS);
  switch (g.P) {
    case 1: hipLaunchKernelGGL((lva_step_fast<LL, 1>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 2: hipLaunchKernelGGL((lva_step_fast<LL, 2>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 3: hipLaunchKernelGGL((lva_step_fast<LL, 3>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 4: hipLaunchKernelGGL((lva_step_fast<LL, 4>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    default: return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

template <int LL>
static int launch_big_p(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, WorkHdr* hdr,
                        uint32_t* items, hipStream_t st) {
  dim3 grid(g.N / TSB, a.band_max, a.nslots), block(8 * TSB);
  switch (g.P) {
    case 1: hipLaunchKernelGGL((lva_step_big<LL, 1>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 2: hipLaunchKernelGGL((lva_step_big<LL, 2>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 3: hipLaunchKernelGGL((lva_step_big<LL, 3>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    case 4: hipLaunchKernelGGL((lva_step_big<LL, 4>), grid, block, 0, st, a, g, codes, trellis, hdr, items); break;
    default: return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

int launch_step_fast(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, WorkHdr* hdr,
                     uint32_t* items, void* stream, void* ev_mid) {
  if (a.nslots == 0 || a.band_max == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int e;
  if (!small_list(g)) {   // big-list kernel + wavefront-per-target fix-up
    if (g.rec) {          // record layout (three message planes, 32 <= L <= 64, L a multiple of 4)
      dim3 grid(g.N / TSB, a.band_max, a.nslots), block(8 * TSB);
      if (g.L <= 32) hipLaunchKernelGGL((lva_step_big_rec<32>), grid, block, 0, st, a, g, codes, trellis, hdr, items);
      else hipLaunchKernelGGL((lva_step_big_rec<64>), grid, block, 0, st, a, g, codes, trellis, hdr, items);
      e = (int)hipGetLastError();
    } else
    e = g.L <= 16 ? launch_big_p<16>(a, g, codes, trellis, hdr, items, st)
      : g.L <= 32 ? launch_big_p<32>(a, g, codes, trellis, hdr, items, st)
                  : launch_big_p<64>(a, g, codes, trellis, hdr, items, st);
    if (e) return e;
    if (ev_mid && (e = (int)hipEventRecord((hipEvent_t)ev_mid, st))) return e;
    hipLaunchKernelGGL(lva_step_fixup_wave, dim3(4096), dim3(256), 0, st, a, g, codes, trellis, hdr, items);
    return (int)hipGetLastError();
  }
  if (g.lazy) {
    dim3 grid(g.N / TS, a.band_max, a.nslots), block(8 * TS);
    // phase-aligned slots (the host starts every read on an even launch): all slots are at an even time step on even launches
    // and at an odd one on odd launches -- one instance per launch, no workgroups of the wrong kind
    const bool run_anchor = !a.phase_aligned || !(a.launch_no & 1u), run_odd = !a.phase_aligned || (a.launch_no & 1u);
#define LVA_LAZY_CASE(LLv, Pv) { if (run_anchor) hipLaunchKernelGGL((lva_step_lazy<LLv, Pv, true>), grid, block, 0, st, a, g, codes, trellis, hdr, items); \
                                 if (run_odd) hipLaunchKernelGGL((lva_step_lazy<LLv, Pv, false>), grid, block, 0, st, a, g, codes, trellis, hdr, items); }
#define LVA_LAZY_L(LLv) switch (g.P) { case 1: LVA_LAZY_CASE(LLv, 1); break; case 2: LVA_LAZY_CASE(LLv, 2); break; \
                                      case 3: LVA_LAZY_CASE(LLv, 3); break; case 4: LVA_LAZY_CASE(LLv, 4); break; default: return (int)hipErrorInvalidValue; }
    switch (g.L) {
      case 2: LVA_LAZY_L(2); break;
      case 4: LVA_LAZY_L(4); break;
      case 8: LVA_LAZY_L(8); break;
      default: return (int)hipErrorInvalidValue;
    }
    e = (int)hipGetLastError();
    if (e) return e;
    if (ev_mid && (e = (int)hipEventRecord((hipEvent_t)ev_mid, st))) return e;
    // one target per wavefront and pass: the pass is a chain of dependent round trips, so more (mostly idle) wavefronts, not fewer
    constexpr uint32_t kFixGrid = kFixupLazyGrid;
    switch (g.P) {
      case 1: hipLaunchKernelGGL((lva_step_fixup_lazy<1>), dim3(kFixGrid), dim3(256), 0, st, a, g, codes, trellis, hdr, items); break;
      case 2: hipLaunchKernelGGL((lva_step_fixup_lazy<2>), dim3(kFixGrid), dim3(256), 0, st, a, g, codes, trellis, hdr, items); break;
      case 3: hipLaunchKernelGGL((lva_step_fixup_lazy<3>), dim3(kFixGrid), dim3(256), 0, st, a, g, codes, trellis, hdr, items); break;
      default: hipLaunchKernelGGL((lva_step_fixup_lazy<4>), dim3(kFixGrid), dim3(256), 0, st, a, g, codes, trellis, hdr, items); break;
    }
    return (int)hipGetLastError();
  }
  switch (g.L) {
    case 1: {
      dim3 grid(g.N / TS, a.band_max, a.nslots), block(4 * TS);
      switch (g.P) {
        case 1: hipLaunchKernelGGL((lva_step_acs<1>), grid, block, 0, st, a, g, codes, trellis); break;
        case 2: hipLaunchKernelGGL((lva_step_acs<2>), grid, block, 0, st, a, g, codes, trellis); break;
        case 3: hipLaunchKernelGGL((lva_step_acs<3>), grid, block, 0, st, a, g, codes, trellis); break;
        case 4: hipLaunchKernelGGL((lva_step_acs<4>), grid, block, 0, st, a, g, codes, trellis); break;
        default: return (int)hipErrorInvalidValue;
      }
      e = (int)hipGetLastError();
      break;
    }
    case 2: e = launch_fast_p<2>(a, g, codes, trellis, hdr, items, st); break;
    case 4: e = launch_fast_p<4>(a, g, codes, trellis, hdr, items, st); break;
    case 8: e = launch_fast_p<8>(a, g, codes, trellis, hdr, items, st); break;
    default: return (int)hipErrorInvalidValue;
  }
  if (e) return e;
  if (ev_mid && (e = (int)hipEventRecord((hipEvent_t)ev_mid, st))) return e;
  if (g.L > 1) {   // fix-up pass: exits at once when the work list is empty
    hipLaunchKernelGGL(lva_step_fixup, dim3(256), dim3(256), 0, st, a, g, codes, trellis, hdr, items);
    e = (int)hipGetLastError();
  }
  return e;
}

int launch_prepare_step(const StepArgs& a, const DevCode* codes, SlotStep* steps, void* stream) {
  if (a.nslots == 0) return 0;
  hipLaunchKernelGGL(lva_prepare_step, dim3((a.nslots + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, codes, steps);
  return (int)hipGetLastError();
}

int launch_init_slots(const Geometry& g, const DevCode* codes, uint32_t* trellis, const InitBatch& batch, SlotDesc* slots, void* stream) {
  if (batch.n == 0) return 0;
  hipLaunchKernelGGL(lva_init_slot, dim3(batch.n), dim3(64), 0, (hipStream_t)stream, g, codes, trellis, batch, slots);
  return (int)hipGetLastError();
}

int launch_gather_finals(const Geometry& g, const DevCode* codes, const uint32_t* trellis, const GatherBatch& batch,
                         uint32_t* results, void* stream) {
  if (batch.n == 0) return 0;
  hipLaunchKernelGGL(lva_gather_final, dim3(batch.n), dim3(64), 0, (hipStream_t)stream, g, codes, trellis, batch, results);
  return (int)hipGetLastError();
}

}  // namespace lva
