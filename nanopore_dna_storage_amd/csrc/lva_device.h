// lva_device.h -- structures shared by the host driver (lva_api.cpp) and the HIP kernels.
//
// Trellis memory (HBM), per read slot:
//     block[parity 2][ring R][crf 8][list L]   each block = 1+P planes of N conv states x 8 bytes:
//         plane 0      (score fp32 bits, message fingerprint) pairs                        "SH"
//         plane 1..P   the message so far, 64 bits per plane, least significant pair first  "MSG"
//   P = ceil((msg_len + mem_conv) / 64) pairs; W = 2P words; F = 2 + W words per entry.
//   * conv fastest: a wavefront of 64 consecutive conv states reads/writes 512 contiguous
//     bytes per plane -- every store of the step kernels is fully coalesced, and the list-head
//     traffic (SH) is separated from the message traffic (MSG), which is only touched for
//     entries that survive the merge.
//   * message planes are position dependent: a state at trellis position p has consumed
//     nbits[p] message bits, so only np = ceil(nbits[p]/64) planes' worth of words are ever
//     written or read there (higher bits are zero by construction): -33 % message bytes.  Inside
//     an entry's message region the words of one conv state are laid out for the widest
//     per-lane access the position allows (lva_kernels.hip msg_word_off): 8 bytes when np = 1,
//     16 bytes (words 0-3 adjacent) when np >= 2, plus 8 or 16 bytes for words 4-7.
//   * ring: only positions [band_lo-1, band_hi) of the two parity buffers are live
//     (reference :677-679 band; the extra position below the band carries the reference's
//     stale-score behaviour), so R = min(nstate_pos, 2*max_deviation+1)
//     positions are stored instead of the reference's nstate_pos.
//   * states the reference leaves at -inf forever (invalid conv states :700, and (conv,crf)
//     pairs whose only predecessor is "stay") are neither stored nor read: readers test the
//     same predicates instead.  The same holds for positions a path cannot have reached yet
//     (> t + 1 after step t) and for positions that cannot reach the final one any more
//     (< npos - nblk + t): the host clips the band table of every step to that range
//     (lva_api.cpp decode_impl), and "outside the band" already reads as -inf / is never used.
//   * variants of this layout, selected per decoder (Geometry below): compact lists at one-bit
//     positions (cmp: 4 lists per ring position there), back-pointer bytes behind every list
//     (lazy), and [conv][entry] records instead of planes for long lists (rec).
//
// Message fingerprint: XOR over the set message bits of a fixed pseudo-random 32-bit word per
// bit index (index = order of consumption).  Appending bits at a step XORs a constant that
// depends only on (position, new bits): fpc[pos][newbits].  Equal messages have equal
// fingerprints; the converse is verified on the full message wherever it decides anything.
#pragma once
#include <cstdint>

namespace lva {

constexpr int kMaxSlotsLimit = 4096;   // reads in flight (the work-list item format allows 2^(21-m))

struct PosRec {                  // everything a workgroup needs to know about the step INTO pos, uniform over the workgroup:
                                 // one 64-byte scalar load instead of a chain of byte loads and pointer look-ups
  uint32_t info;                 // ptype[pos] | ptype[pos-1] << 8 | npair[pos] << 16 | npair[pos-1] << 24   (pos = 0: its own values twice)
  uint32_t vmask, vval;          // of pos
  uint32_t vmask1, vval1;        // of pos-1
  uint32_t fpc[4];               // fingerprint delta of the step into pos, by new bits
  uint32_t np2;                  // npair[pos-2] (1 when pos < 2)
  const uint16_t* pred;          // predtab[ptype[pos]]
  const uint16_t* pred1;         // predtab[ptype[pos-1]]
  uint32_t cmp3;                 // compact-list flags (Geometry::cmp): bit 0 = position pos, bit 1 = pos-1, bit 2 = pos-2 is a one-bit
                                 // position >= 1 (its lists are stored four per conv state)
  uint32_t xs;                   // XCD-aware tile order of the butterfly kernels (lva_kernels.hip xcd_tile): the workgroups (tile, pos) whose
                                 // blockIdx.x has the same low three bits -- one XCD's share under round-robin dispatch -- are those with
                                 // equal bits xs .. xs+2 of `tile`; chosen per position so that the rows a workgroup at pos reads as its
                                 // targets' own (stay) lists are the rows the same XCD's workgroups at pos + 1 stage as source lists
};
static_assert(sizeof(PosRec) == 64, "PosRec is one 64-byte scalar load");

struct DevCode {                 // one per orientation (0 = forward, 1 = reverse complement)
  uint32_t m, nconv, npos, init, fin;
  uint8_t ptype[256];            // block type of the step into pos
  uint8_t npair[256];            // message planes in use at pos: max(1, ceil(nbits[pos]/64))
  uint32_t vmask[256], vval[256];
  uint32_t fpc[256][4];          // fingerprint delta of the step into pos, by new bits
  const uint16_t* predtab[4];    // device pointers, [nconv] each (nullptr when unused)
  PosRec rec[256] __attribute__((aligned(64)));
};

// One per read slot, resident in device memory: written (by lva_init_slot) when a read enters the
// slot, read by every trellis-step launch.  A launch carries only its number; the slot's time step
// is t = launch_no - start, and the slot takes part while t < nblk -- so the host enqueues the
// whole schedule of a batch without per-launch argument tables, for any number of slots.
struct SlotDesc {
  const float* post;             // block 0 of the read's posterior matrix (device)
  const uint32_t* band;          // [nblk] lo | hi << 16 (| lazy-mode flags << 30): the band of every time step (:677-679),
                                 // evaluated on the host exactly as the reference binary does (Code::band)
  uint32_t nblk, orient;
  uint32_t start;                // launch number of the read's time step 0
  uint32_t pad;
};

struct SlotStep {                // what one read slot does in one trellis-step launch (built in registers
                                 // from the SlotDesc by load_slot, lva_kernels.hip)
  const float* post_row;         // 40 log-posteriors of block t (device)
  uint32_t slot;                 // trellis buffer index
  uint32_t t;
  uint32_t lo, hi;               // band of step t
  uint32_t prev_hi;              // band end of step t-1 (1 at t = 0: only position 0 is initialised)
  uint32_t orient;
  uint32_t flags;                // lazy mode (kernel 4): bit 0 = the row of position lo-1 in the previous buffer is stale (written
                                 // before step t-1); bit 1 = which message buffer the messages of its entries live in
  uint32_t pad;
  uint32_t srccmp[2];            // Geometry::cmp: bit y = the SOURCE position lo + y - 1 of band position lo + y stores compact lists
                                 // (a workgroup then stages 4 x L rows instead of 8 x L) -- known from the slot record alone, before
                                 // anything else is loaded; band positions beyond 64 read the flag from the position record
};
static_assert(sizeof(SlotStep) == 48, "twelve words");

struct StepArgs {
  const SlotDesc* slots;         // device
  const SlotStep* steps;         // device, [nslots]: this launch's SlotStep of every slot, written by lva_prepare_step
                                 // right before the launch (t = 0xFFFFFFFF: the slot takes no part) -- one 40-byte record per
                                 // workgroup instead of the dependent chain  descriptor -> band[t], band[t-1]
  uint32_t nslots;               // slots in use by this batch (grid z)
  uint32_t band_max;             // positions per band at most (grid y)
  uint32_t step_parity;          // launch_no & 1: selects the work-list counter
  uint32_t launch_no;
  uint32_t phase_aligned;        // lazy mode: every slot's time step has the parity of launch_no (the host starts reads on even
                                 // launches only), so only the lva_step_lazy instance of that parity is launched
  uint32_t full_lo, full_hi;     // positions full_lo .. full_hi (both orientations): every source tile feeds at least one VALID target
                                 // conv state (:700).  Outside -- the first and last few positions, where most of the register is pinned
                                 // to the initial / final state -- a workgroup tests its tile before it stages anything (tile_has_target)
  uint32_t pad;
};

struct Geometry {                // strides in 32-bit words
  uint32_t N, L, W, F, R, P;
  uint32_t sBlk;                 // one (ring, crf, list entry) block = N*F
  uint32_t lazy;                 // kernel mode 4: messages are materialised every second time step (lva_kernels.hip, "lazy")
  uint32_t cmp;                  // compact lists (every plane-layout fast kernel: lazy mode, the L = 1 kernel, the big-list kernel): at a one-bit position >= 1 a conv state has two COMPLEMENTARY bases
                                 // ({A,T} or {C,G}: its two predecessors differ in the register bit the step shifts out, which both generators tap), so 4 of its 8 crf lists exist -- the list of
                                 // crf state k is stored as list k >> 1 of the ring position (flip {A|C}, flip {T|G}, flop {A|C}, flop {T|G}).
                                 // Readers stage 4 x L rows of such a source position, all of them data, instead of 8 x L rows of which
                                 // half is never-written memory interleaved at conv-state granularity
  uint32_t rec;                  // big-list kernel, three message planes, L >= 32 and a multiple of 4: RECORD layout.  A (ring, crf) list is
                                 //   [conv N][entry L][2 + 2 np words: score, fingerprint, the 2 np message words in use at the position]
                                 // instead of L blocks of conv-fastest planes: the entries of one conv state's list are adjacent
                                 // (4 to 8 per 128-byte line), so a thread that walks a list pulls each line once instead of one line
                                 // per entry, and an entry is one line instead of three (allocated for np = 3: sCrf = N*L*8 words)
  uint64_t sCrf, sRing, sPar, sSlot;
};

inline Geometry make_geometry(uint32_t N, uint32_t L, uint32_t msg_bits, uint32_t R, uint32_t lazy = 0, uint32_t rec = 0, uint32_t cmp = 0) {
  Geometry g;
  g.N = N; g.L = L; g.P = (msg_bits + 63) / 64; if (g.P == 0) g.P = 1;
  g.W = 2 * g.P; g.F = g.W + 2; g.R = R;
  g.lazy = lazy; g.rec = (rec && !lazy && g.P == 3 && L >= 32 && L % 4 == 0) ? 1u : 0u;   // (below 32 entries the plane layout is faster: measured)
  g.cmp = ((lazy || cmp) && !g.rec) ? 1u : 0u;       // (the record layout keeps a list's entries together: a list that does not exist is a hole of whole lines)
  g.sBlk = N * g.F;
  // lazy mode: behind the L entry blocks of a (ring, crf) list, L back-pointer bytes per conv state ([conv][entry])
  g.sCrf = (uint64_t)g.sBlk * L + (lazy ? (uint64_t)N * L / 4 : 0); g.sRing = g.sCrf * 8;
  g.sPar = g.sRing * R; g.sSlot = g.sPar * 2;
  return g;
}

// Targets the fast kernel could not decide exactly (score ties, non-finite posteriors,
// fingerprint collisions): redone by the exact kernel right after, in the same stream.
struct WorkHdr {
  uint32_t count[2];             // per step parity
  uint32_t overflow[2];          // list full: the exact kernel redoes the whole step
  unsigned long long total;      // queued states redone since the last reset (profile)
  uint32_t cap;
  uint32_t overflow_steps;       // launches whose work list overflowed: the fix-up pass redid the WHOLE step on its exact path
                                 // (a performance cliff on tie-dense input that callers should be able to see)
  unsigned long long reason[4];  // why: 0 tie on top, 1 non-finite arithmetic, 2 too many fingerprint matches, 3 fingerprint collision
};

// final-state gather: result record per read = [crf 8][list L][field F] words
struct GatherArgs {
  uint32_t slot, parity, orient, read;
  uint32_t nblk;                 // blocks of the read (lazy mode: whether the last step stored messages or back-pointers)
};

// Reads enter and leave their slots in batches: one launch serves every slot that turns over at this step (small trellises turn
// several over per step, and a launch per slot -- two per read -- was 8 % of the device time at m = 6, L = 1), one workgroup each.
constexpr int kTurnoverBatch = 16;
struct InitBatch {
  uint32_t n, pad;
  uint32_t slot[kTurnoverBatch];
  SlotDesc desc[kTurnoverBatch];
};
struct GatherBatch {
  uint32_t n;
  GatherArgs a[kTurnoverBatch];
};

}  // namespace lva
