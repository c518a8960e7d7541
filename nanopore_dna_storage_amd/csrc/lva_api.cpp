// lva_api.cpp -- C ABI (include/lva_decoder.h) and the host driver of the trellis kernels.
//
// Scheduling: reads are independent.  The decoder keeps S read slots resident in
// HBM; one trellis-step launch advances every active slot by one time step of its own read,
// so reads of different lengths overlap freely: a slot whose read finishes is gathered and
// re-initialised for the next read while the others continue ("continuous batching" of the
// reference's sequential time loop :667).  Everything is enqueued on one HIP stream without
// host synchronisation until the results are copied back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/lva_decoder.h"
#include "lva_code.h"
#include "lva_device.h"
#include "lva_kernels.h"
#include "bc_kernels.h"

using namespace lva;

namespace {

thread_local std::string g_hip_error;

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess) {                                                                   \
      g_hip_error = std::string(#expr) + ": " + hipGetErrorString(e__);                        \
      return LVA_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

}  // namespace

struct lva_decoder {
  lva_config cfg{};
  std::string sync_marker;
  Code code[2];                // forward, reverse complement
  uint32_t max_dev = 0;
  Geometry g{};
  int slots = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_total0 = nullptr, ev_total1 = nullptr, ev_step0 = nullptr, ev_step1 = nullptr, ev_h2d = nullptr;
  DevCode* d_codes = nullptr;
  uint16_t* d_predtab = nullptr;
  uint32_t* d_trellis = nullptr;
  uint32_t* d_results = nullptr;
  size_t results_cap = 0;      // reads
  SlotDesc* d_slots = nullptr; // [slots]
  SlotStep* d_steps = nullptr; // [slots] this launch's time step of every slot (lva_prepare_step)
  uint32_t* d_band = nullptr;  // band tables of the batch in flight: lo | hi << 16 per (read, time step)
  size_t band_cap = 0;         // words
  WorkHdr* d_work = nullptr;   // header followed by the item array
  uint32_t work_cap = 1u << 20;
  int kernel = 1;              // 1 = exact, 2 = fast + exact fix-up
  uint32_t launch_no = 0;      // trellis-step launches since creation (the slots' clock)
  uint32_t full_lo = 1, full_hi = 0;   // positions at which every 64-source tile has a valid target (StepArgs::full_lo/hi)
  int launch_events = 0;       // lva_decoder_set_launch_events
  std::vector<hipEvent_t> ev_pool;
  lva_profile prof{};
};

namespace {
// every error exit of a call that has enqueued asynchronous work drains the stream first: pending
// device->host copies target buffers that die with the call's frame
struct StreamDrain {
  hipStream_t s;
  bool armed = true;
  explicit StreamDrain(hipStream_t st) : s(st) {}
  ~StreamDrain() { if (armed && s) (void)hipStreamSynchronize(s); }
};
}  // namespace

extern "C" {

#ifndef LVA_BUILD_ID
#define LVA_BUILD_ID "unknown"
#endif
const char* lva_version(void) { return "lva_hip 5.0 (gfx950) build " LVA_BUILD_ID; }

int lva_abi_version(void) { return LVA_ABI_VERSION; }

const char* lva_last_hip_error(void) { return g_hip_error.c_str(); }

const char* lva_strerror(int code) {
  switch (code) {
    case LVA_OK: return "ok";
    case LVA_ERR_MEM_CONV: return "Invalid mem_conv (allowed: 6, 8, 11, 14)";
    case LVA_ERR_RATE: return "Invalid rate parameter (allowed: 1, 2, 3, 4, 5, 7)";
    case LVA_ERR_MSG_LEN: return "Output length not even. Try padding with a single 0 at end.";
    case LVA_ERR_SYNC: return "Invalid sync marker / sync period";
    case LVA_ERR_TOO_MANY_STATES: return "Too many states, can't fit in 32 bits";
    case LVA_ERR_POST_TOO_SHORT: return "Too small post matrix";
    case LVA_ERR_MSG_TOO_LONG: return "msg_len too large (message + memory must fit 256 bits, msg_len <= 255)";
    case LVA_ERR_NOMEM: return "out of memory";
    case LVA_ERR_HIP: return "HIP runtime error";
    case LVA_ERR_ARG: return "invalid argument";
    case LVA_ERR_NO_DEVICE: return "no usable HIP device (the decoder has no CPU fallback)";
    case LVA_ERR_UNSUPPORTED: return "unsupported code structure";
    default: return "unknown error";
  }
}

int lva_code_describe(int32_t mem_conv, int32_t rate, uint32_t msg_len, int32_t rc, const char* sync_marker,
                      uint32_t sync_period, lva_code_info* out) {
  Code c;
  const int st = build_code(&c, mem_conv, rate, msg_len, rc, sync_marker, sync_period);
  if (st != LVA_OK) return st;
  if (!out) return LVA_OK;
  std::memset(out, 0, sizeof *out);
  out->nstate_pos = c.npos; out->nstate_conv = c.nconv; out->oligo_len = c.oligo_len();
  out->msg_words = c.msg_words(); out->initial_state = c.init; out->final_state = c.fin;
  out->g0 = c.g[0]; out->g1 = c.g[1]; out->pattern_len = c.plen;
  std::memcpy(out->pattern, c.pattern, 16);
  return LVA_OK;
}

int lva_code_tables(int32_t mem_conv, int32_t rate, uint32_t msg_len, int32_t rc, const char* sync_marker,
                    uint32_t sync_period, uint32_t* pos2msg, uint8_t* ptype, uint32_t* vmask, uint32_t* vval,
                    uint16_t* predtab) {
  Code c;
  const int st = build_code(&c, mem_conv, rate, msg_len, rc, sync_marker, sync_period);
  if (st != LVA_OK) return st;
  if (pos2msg) std::memcpy(pos2msg, c.pos2msg, c.npos * sizeof(uint32_t));
  if (ptype) std::memcpy(ptype, c.ptype, c.npos);
  if (vmask) std::memcpy(vmask, c.vmask, c.npos * sizeof(uint32_t));
  if (vval) std::memcpy(vval, c.vval, c.npos * sizeof(uint32_t));
  if (predtab)
    for (int T = 0; T < 4; ++T) {
      uint16_t* dst = predtab + (size_t)T * c.nconv;
      if (c.predtab[T].empty()) std::memset(dst, 0, c.nconv * sizeof(uint16_t));
      else std::memcpy(dst, c.predtab[T].data(), c.nconv * sizeof(uint16_t));
    }
  return LVA_OK;
}

int lva_encode(int32_t mem_conv, int32_t rate, uint32_t msg_len, const uint8_t* msgs, int32_t n_msgs,
               uint8_t* out_bases) {
  if (n_msgs < 0 || (n_msgs > 0 && (!msgs || !out_bases))) return LVA_ERR_ARG;
  Code c;
  const int st = build_code(&c, mem_conv, rate, msg_len, 0, nullptr, 0);
  if (st != LVA_OK) return st;
  for (int32_t i = 0; i < n_msgs; ++i) {
    const int e = c.encode(msgs + (size_t)i * msg_len, out_bases + (size_t)i * c.oligo_len());
    if (e != LVA_OK) return e;
  }
  return LVA_OK;
}

// the band of time step t the kernels work on: Code::working_band (lva_code.cpp)
static void working_band(const Code& c, uint32_t t, uint32_t nblk, uint32_t max_dev, uint32_t* lo_out, uint32_t* hi_out) {
  c.working_band(t, nblk, max_dev, lo_out, hi_out);
}

int lva_band_table(int32_t mem_conv, int32_t rate, uint32_t msg_len, int32_t rc, const char* sync_marker, uint32_t sync_period,
                   uint32_t nblk, uint32_t max_deviation, uint32_t* reference_lo_hi, uint32_t* working_lo_hi) {
  Code c;
  const int st = build_code(&c, mem_conv, rate, msg_len, rc, sync_marker, sync_period);
  if (st != LVA_OK) return st;
  if (max_deviation == LVA_MAX_DEVIATION_DEFAULT) max_deviation = msg_len + (uint32_t)mem_conv + 1;
  for (uint32_t t = 0; t < nblk; ++t) {
    uint32_t lo, hi;
    if (reference_lo_hi) { c.band(t, nblk, max_deviation, &lo, &hi); reference_lo_hi[2 * t] = lo; reference_lo_hi[2 * t + 1] = hi; }
    if (working_lo_hi) { working_band(c, t, nblk, max_deviation, &lo, &hi); working_lo_hi[2 * t] = lo; working_lo_hi[2 * t + 1] = hi; }
  }
  return LVA_OK;
}

int lva_algorithmic_bytes(int32_t mem_conv, int32_t rate, uint32_t msg_len, int32_t rc, const char* sync_marker,
                          uint32_t sync_period, uint32_t nblk, uint32_t list_size, uint32_t max_deviation,
                          double* out) {
  Code c;
  const int st = build_code(&c, mem_conv, rate, msg_len, rc, sync_marker, sync_period);
  if (st != LVA_OK) return st;
  if (max_deviation == LVA_MAX_DEVIATION_DEFAULT) max_deviation = msg_len + (uint32_t)mem_conv + 1;
  if (out) *out = c.algorithmic_bytes(nblk, list_size, max_deviation);
  return LVA_OK;
}

// ------------------------------------------------------------------------------------------

static int upload_codes(lva_decoder* d) {
  // predecessor tables of both orientations in one allocation: [orient 2][type 4][nconv]
  const uint32_t N = d->code[0].nconv;
  std::vector<uint16_t> host((size_t)2 * 4 * N, 0);
  for (int o = 0; o < 2; ++o)
    for (int T = 0; T < 4; ++T)
      if (!d->code[o].predtab[T].empty())
        std::memcpy(host.data() + ((size_t)o * 4 + T) * N, d->code[o].predtab[T].data(), N * sizeof(uint16_t));
  HIP_TRY(hipMalloc(&d->d_predtab, host.size() * sizeof(uint16_t)));
  HIP_TRY(hipMemcpy(d->d_predtab, host.data(), host.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  DevCode dc[2];
  std::memset(dc, 0, sizeof dc);
  for (int o = 0; o < 2; ++o) {
    const Code& c = d->code[o];
    dc[o].m = (uint32_t)c.mem_conv; dc[o].nconv = c.nconv; dc[o].npos = c.npos; dc[o].init = c.init; dc[o].fin = c.fin;
    std::memcpy(dc[o].ptype, c.ptype, sizeof dc[o].ptype);
    std::memcpy(dc[o].vmask, c.vmask, sizeof dc[o].vmask);
    std::memcpy(dc[o].vval, c.vval, sizeof dc[o].vval);
    std::memcpy(dc[o].fpc, c.fpc, sizeof dc[o].fpc);
    for (uint32_t p = 0; p < c.npos; ++p) dc[o].npair[p] = (uint8_t)std::max<uint32_t>(1, (c.nbits[p] + 63) / 64);
    for (int T = 0; T < 4; ++T)
      dc[o].predtab[T] = c.predtab[T].empty() ? nullptr : d->d_predtab + ((size_t)o * 4 + T) * N;
    for (uint32_t p = 0; p < c.npos; ++p) {
      const uint32_t q = p ? p - 1 : 0;
      PosRec& r = dc[o].rec[p];
      r.info = (uint32_t)dc[o].ptype[p] | (uint32_t)dc[o].ptype[q] << 8 | (uint32_t)dc[o].npair[p] << 16 | (uint32_t)dc[o].npair[q] << 24;
      r.vmask = c.vmask[p]; r.vval = c.vval[p]; r.vmask1 = c.vmask[q]; r.vval1 = c.vval[q];
      for (int nb = 0; nb < 4; ++nb) r.fpc[nb] = c.fpc[p][nb];
      r.np2 = p >= 2 ? dc[o].npair[p - 2] : 1u;
      auto one_bit = [&](int64_t at) -> uint32_t { return at >= 1 && dc[o].ptype[at] == 0 ? 1u : 0u; };   // compact lists there (Geometry::cmp)
      r.cmp3 = one_bit(p) | one_bit((int64_t)p - 1) << 1 | one_bit((int64_t)p - 2) << 2;
      r.xs = 0;
      r.pred = dc[o].predtab[dc[o].ptype[p] & 3]; r.pred1 = dc[o].predtab[dc[o].ptype[q] & 3];
    }
    // XCD-aware tile order (PosRec::xs; lva_kernels.hip xcd_tile).  A 128-byte line of position p's lists -- 16 consecutive conv
    // states, line index q = conv / 16 -- is read twice in the launch after the one that wrote it: as the own (stay) list of its
    // targets by the workgroup at p whose butterfly ends there (tile = q without the bits the step into p shifted in, sh = 1 or 2 of
    // them), and as a source list by the workgroup (tile q >> 2, p + 1).  With workgroups dealt round-robin over the 8 XCDs
    // (MI355X_MICROARCH.md, workgroup dispatch: observed, a speed matter only) a workgroup (tile, p) runs on the XCD labelled by
    // bits xs[p] .. xs[p]+2 of its tile, and both readers share an XCD -- and its L2 -- when xs[p] = xs[p+1] + sh.  Greedy chain
    // from the largest shift the tile count allows; 0 = the plain order.  Used by the L = 1 kernel only: the list kernels
    // reach their stay lists too late for the staged rows to be still in L2 (measured: nothing at m=11, -1 % at m=14).
    auto chain = [&](int tile_bits) {
      if (tile_bits < 4) return;
      const uint32_t smax = std::min<uint32_t>((uint32_t)tile_bits - 3u, 2u);   // (larger shifts spread a launch's neighbours over the rows: m=14 -2 %)
      uint32_t s = smax;
      for (uint32_t p = 1; p < c.npos; ++p) {
        dc[o].rec[p].xs = s;
        const uint32_t sh = c.shift_of(c.ptype[p]);
        s = s >= sh ? s - sh : smax;
      }
    };
    // (measurements: LVA_NO_XCD_ORDER=1 together with LVA_TESTING=1 keeps the plain order)
    const char* plain = std::getenv("LVA_NO_XCD_ORDER"); const char* testing = std::getenv("LVA_TESTING");
    if (!(plain && plain[0] == '1' && testing && testing[0] == '1')) chain(c.mem_conv - 6);
  }
  // Compact lists (Geometry::cmp) store crf state k's list as list k >> 1 at a one-bit position: that needs the reachable bases
  // of every valid conv state there to be a complementary pair ({A,T} or {C,G}).  True of the four built-in generator pairs
  // (tests/test_code_tables.py); checked here so that a code added later cannot alias two lists silently.
  if (d->g.cmp)
    for (int o = 0; o < 2; ++o) {
      const Code& c = d->code[o];
      for (uint32_t p = 1; p < c.npos; ++p) {
        if (c.ptype[p] != 0 || c.predtab[0].empty()) continue;
        for (uint32_t cv = 0; cv < N; ++cv) {
          if ((cv & c.vmask[p]) != c.vval[p]) continue;
          const uint32_t pk = c.predtab[0][cv];
          const uint32_t has = ((pk >> 3) & 1u) | (((pk >> 7) & 1u) << 1) | (((pk >> 11) & 1u) << 2) | (((pk >> 15) & 1u) << 3);
          if (has != 0x9u && has != 0x6u && has != 0u) return LVA_ERR_UNSUPPORTED;
        }
      }
    }
  HIP_TRY(hipMalloc(&d->d_codes, sizeof dc));
  HIP_TRY(hipMemcpy(d->d_codes, dc, sizeof dc, hipMemcpyHostToDevice));
  return LVA_OK;
}

int lva_decoder_create(const lva_config* cfg, lva_decoder** out) {
  if (!cfg || !out) return LVA_ERR_ARG;
  *out = nullptr;
  if (cfg->list_size == 0 || cfg->list_size > 65535) return LVA_ERR_ARG;
  lva_decoder* d = new (std::nothrow) lva_decoder();
  if (!d) return LVA_ERR_NOMEM;
  d->cfg = *cfg;
  d->sync_marker = cfg->sync_marker ? cfg->sync_marker : "";
  d->cfg.sync_marker = nullptr;
  for (int o = 0; o < 2; ++o) {
    const int st = build_code(&d->code[o], cfg->mem_conv, cfg->rate, cfg->msg_len, o, d->sync_marker.c_str(),
                              cfg->sync_period);
    if (st != LVA_OK) { delete d; return st; }
  }
  const Code& c = d->code[0];
  if (c.msg_len > 255 || c.msg_len + (uint32_t)c.mem_conv > 256) { delete d; return LVA_ERR_MSG_TOO_LONG; }
  if ((uint64_t)c.npos * kCrf * c.nconv >= ((uint64_t)1 << 32)) { delete d; return LVA_ERR_TOO_MANY_STATES; }
  d->max_dev = cfg->max_deviation == LVA_MAX_DEVIATION_DEFAULT ? c.msg_len + (uint32_t)c.mem_conv + 1 : cfg->max_deviation;
  {
    // Positions at which every tile of 64 source conv states feeds at least one valid target conv state, in both orientations:
    // the longest run [full_lo, full_hi].  A target conv state of tile x at position p is  x*Tn + low + leg*(N >> sh)  (low < Tn =
    // 64 >> sh, leg < 2^sh): its middle bits are the tile's, so the tile has a valid target iff those bits agree with the mask.
    const uint32_t N = c.nconv;
    std::vector<uint8_t> full(c.npos, 0);
    for (uint32_t p = 1; p < c.npos; ++p) {
      bool ok = N >= 64;
      for (int o = 0; o < 2 && ok; ++o) {
        const Code& co = d->code[o];
        const uint32_t sh = co.ptype[p] == 0 ? 1u : 2u, Tn = 64u >> sh;
        const uint32_t mid = (N - 1) & ~(Tn - 1) & ~(((1u << sh) - 1u) << ((uint32_t)co.mem_conv - sh));
        for (uint32_t x = 0; x < N / 64 && ok; ++x) ok = ((x * Tn) & co.vmask[p] & mid) == (co.vval[p] & mid);
      }
      full[p] = ok ? 1 : 0;
    }
    uint32_t best = 0, run = 0;
    for (uint32_t p = 1; p < c.npos; ++p) {
      run = full[p] ? run + 1 : 0;
      if (run > best) { best = run; d->full_hi = p; d->full_lo = p + 1 - run; }
    }
  }
  // kernel mode 4 ("lazy", list sizes 2, 4, 8): messages are materialised every second time step and carried as one-byte
  // back-pointers in between; two-hop chains reach one position further below the band, hence one more ring position
  // Default (kernel 0) for list sizes 2 / 4 / 8 (m = 11 L = 8: +5 % over kernel 2, m = 8: +9 %; m = 14 with four message planes: +8 %
  // since the anchor instance keeps one entry in flight there).
  const bool lazy_ok = (cfg->list_size == 2 || cfg->list_size == 4 || cfg->list_size == 8) && c.nconv >= 64;
  if (cfg->kernel == 4 && !lazy_ok) { delete d; return LVA_ERR_UNSUPPORTED; }
  const bool lazy = cfg->kernel == 4 || (cfg->kernel == 0 && lazy_ok);
  const uint64_t ring = std::min<uint64_t>(c.npos, 2ull * d->max_dev + (lazy ? 2 : 1));
  // the big-list kernel (kernel mode 2 at list sizes without a small-list instance) keeps its lists in the record layout
  // where the message has three planes and L is a multiple of 4 (Geometry::rec)
  const bool small = cfg->list_size == 1 || cfg->list_size == 2 || cfg->list_size == 4 || cfg->list_size == 8;
  const bool big = !lazy && !small && cfg->list_size <= 64 && (cfg->kernel == 0 || cfg->kernel == 2) && c.nconv >= 64;
  // compact lists at one-bit positions (Geometry::cmp): the lazy kernels, the L = 1 kernel (lva_step_acs) and the big-list kernel on
  // the plane layout (make_geometry drops the flag where the record layout applies)
  const bool acs = cfg->list_size == 1 && (cfg->kernel == 0 || cfg->kernel == 2) && c.nconv >= 64;
  d->g = make_geometry(c.nconv, cfg->list_size, c.msg_bits(), (uint32_t)std::max<uint64_t>(ring, 1), lazy ? 1u : 0u, big ? 1u : 0u,
                       (acs || big) ? 1u : 0u);
  if (d->g.sPar >= ((uint64_t)1 << 32)) { delete d; return LVA_ERR_TOO_MANY_STATES; }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    delete d;
    return LVA_ERR_NO_DEVICE;
  }
  d->device = cfg->device;
  auto fail = [&](int code) { lva_decoder_destroy(d); return code; };
  if (hipSetDevice(d->device) != hipSuccess) return fail(LVA_ERR_NO_DEVICE);
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) return fail(LVA_ERR_HIP);
  for (hipEvent_t* ev : {&d->ev_total0, &d->ev_total1, &d->ev_step0, &d->ev_step1, &d->ev_h2d})
    if (hipEventCreate(ev) != hipSuccess) return fail(LVA_ERR_HIP);
  {
    const int st = upload_codes(d);
    if (st != LVA_OK) return fail(st);
  }
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return fail(LVA_ERR_HIP);
  const uint64_t slot_bytes = d->g.sSlot * sizeof(uint32_t);
  uint64_t budget = cfg->mem_budget_bytes ? cfg->mem_budget_bytes : (uint64_t)(free_b * 0.6);
  // reads in flight: enough of them that one trellis-step launch fills the chip -- a launch has
  // (2^m / 64 tiles) x (<= 2 max_deviation positions) x slots workgroups, so small trellises take
  // proportionally more slots (128 at m >= 11, 256 at m = 8, 1024 at m = 6: measured), and four times as many at
  // L = 1, whose launches are that much shorter (m = 6: 0.11 ms at 1024 slots; 4096 slots +15 %, m = 11: 128 slots +5 %: round 5,
  // scripts/r5/slot_sweep_small.sh); bounded by the memory budget and by the work-list item format (21 - m bits of slot index)
  const uint64_t hard = std::min<uint64_t>(kMaxSlotsLimit, 1ull << (21 - std::min(c.mem_conv, 20)));
  int slots = (int)std::min<uint64_t>(budget / slot_bytes, hard);
  if (cfg->max_slots > 0) slots = std::min(slots, (int)cfg->max_slots);
  else {
    const uint64_t per_l = cfg->list_size == 1 ? 4 : 1;
    // (never fewer than 128: the tail of a launch -- 74 k workgroups over 1024 resident ones at 64 slots of m = 11 -- costs 1.8 %)
    slots = std::min<int>(slots, (int)std::max<uint64_t>(128, std::min<uint64_t>(1024 * per_l, per_l * 32ull * 2048 / c.nconv)));
  }
  if (slots < 1) return fail(LVA_ERR_NOMEM);
  d->slots = slots;
  if (hipMalloc(&d->d_trellis, (size_t)slots * slot_bytes) != hipSuccess) return fail(LVA_ERR_NOMEM);
  if (hipMalloc(&d->d_slots, (size_t)slots * sizeof(SlotDesc)) != hipSuccess) return fail(LVA_ERR_NOMEM);
  if (hipMemset(d->d_slots, 0, (size_t)slots * sizeof(SlotDesc)) != hipSuccess) return fail(LVA_ERR_HIP);
  if (hipMalloc(&d->d_steps, (size_t)slots * sizeof(SlotStep)) != hipSuccess) return fail(LVA_ERR_NOMEM);
  d->prof.slots = slots;
  const bool fast_ok = fast_kernel_available(d->g);
  if (cfg->kernel == 2 && !fast_ok) return fail(LVA_ERR_UNSUPPORTED);
  // 1 = exact (one thread per target), 2 = fast + fix-up, 3 = wavefront per target (lists of 9..64 entries)
  if (cfg->kernel == 3 && !wave_kernel_available(d->g)) return fail(LVA_ERR_UNSUPPORTED);
  d->kernel = cfg->kernel == 1 ? 1 : cfg->kernel == 3 ? 3 : lazy ? 4 : (fast_ok ? 2 : (wave_kernel_available(d->g) ? 3 : 1));
  if (d->kernel == 4 && (!fast_ok || c.nconv < 64)) return fail(LVA_ERR_UNSUPPORTED);
  d->prof.kernel = d->kernel;
  // tests: force the work-list overflow path.  Honoured only together with LVA_TESTING=1 -- a stray LVA_WORK_CAP in a user's
  // environment would cost an order of magnitude silently (visible through lva_profile.overflow_steps only)
  if (const char* cap = std::getenv("LVA_WORK_CAP")) {
    const char* testing = std::getenv("LVA_TESTING");
    const long v = std::atol(cap);
    if (testing && testing[0] == '1' && v >= 1 && v <= (1l << 24)) d->work_cap = (uint32_t)v;
  }
  if (hipMalloc(&d->d_work, sizeof(WorkHdr) + (size_t)d->work_cap * sizeof(uint32_t)) != hipSuccess) return fail(LVA_ERR_NOMEM);
  *out = d;
  return LVA_OK;
}

void lva_decoder_destroy(lva_decoder* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->d_trellis) (void)hipFree(d->d_trellis);
  if (d->d_results) (void)hipFree(d->d_results);
  if (d->d_work) (void)hipFree(d->d_work);
  if (d->d_slots) (void)hipFree(d->d_slots);
  if (d->d_steps) (void)hipFree(d->d_steps);
  if (d->d_band) (void)hipFree(d->d_band);
  for (hipEvent_t e : d->ev_pool) (void)hipEventDestroy(e);
  if (d->ev_h2d) (void)hipEventDestroy(d->ev_h2d);
  if (d->d_codes) (void)hipFree(d->d_codes);
  if (d->d_predtab) (void)hipFree(d->d_predtab);
  if (d->ev_total0) (void)hipEventDestroy(d->ev_total0);
  if (d->ev_total1) (void)hipEventDestroy(d->ev_total1);
  if (d->ev_step0) (void)hipEventDestroy(d->ev_step0);
  if (d->ev_step1) (void)hipEventDestroy(d->ev_step1);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int lva_decoder_set_launch_events(lva_decoder* d, int32_t on) {
  if (!d) return LVA_ERR_ARG;
  d->launch_events = on ? 1 : 0;
  return LVA_OK;
}

int lva_decoder_profile(const lva_decoder* d, lva_profile* out) {
  if (!d || !out) return LVA_ERR_ARG;
  *out = d->prof;
  return LVA_OK;
}

// final selection (:806-844) on the host: gather finite entries crf-major, std::sort by score
// descending (the same libstdc++ algorithm the reference calls), keep list_size, unpack bits
static void finish_read(const lva_decoder* d, int orient, const uint32_t* rec, uint8_t* out_msgs, float* out_scores,
                        int32_t* out_count) {
  struct Path { float score; const uint32_t* words; };
  const uint32_t L = d->g.L, F = d->g.F;
  const Code& c = d->code[orient];
  std::vector<Path> paths;
  paths.reserve((size_t)8 * L);
  const float NEG = -std::numeric_limits<float>::infinity();
  for (uint32_t k = 0; k < 8; ++k)
    for (uint32_t l = 0; l < L; ++l) {
      const uint32_t* e = rec + ((size_t)k * L + l) * F;
      float s;
      std::memcpy(&s, e, sizeof s);
      if (s != NEG) paths.push_back({s, e + 2});
    }
  std::sort(paths.begin(), paths.end(), [](const Path& a, const Path& b) -> bool { return a.score > b.score; });
  if (paths.size() > L) paths.resize(L);
  const uint32_t total = c.msg_len + (uint32_t)c.mem_conv;
  for (size_t i = 0; i < paths.size(); ++i) {
    uint8_t* o = out_msgs + i * c.msg_len;
    for (uint32_t b = 0; b < c.msg_len; ++b) {
      const uint32_t bit = total - 1 - b;                              // :831-834
      const uint8_t v = (uint8_t)((paths[i].words[bit >> 5] >> (bit & 31)) & 1u);
      o[c.rc ? c.msg_len - 1 - b : b] = v;                             // :835
    }
    if (out_scores) out_scores[i] = paths[i].score;
  }
  *out_count = (int32_t)paths.size();
}

static int decode_impl(lva_decoder* d, const float* post_dev, const int64_t* beg, const int64_t* len, int32_t n, const uint8_t* rc_flags,
                       uint8_t* out_msgs, float* out_scores, int32_t* out_counts, bool timed_total_started) {
  const Geometry& g = d->g;
  const uint32_t npos = d->code[0].npos, L = g.L;
  const size_t rec_words = (size_t)8 * L * g.F;
  // host buffers that asynchronous copies read or write: declared BEFORE the drain, so that on every error exit the
  // drain's hipStreamSynchronize runs first and these die afterwards (locals are destroyed in reverse order)
  std::vector<uint32_t> band, host;
  std::vector<uint8_t> gathered;
  WorkHdr h0, h1;
  StreamDrain drain(d->stream);
  // reads the reference would refuse (:600-601)
  std::vector<int32_t> order;
  size_t band_words = 0;
  for (int32_t i = 0; i < n; ++i) {
    const int64_t nb = len[i];
    if (nb < 0 || nb > 0xFFFFFFFFll || beg[i] < 0) return LVA_ERR_ARG;
    if ((uint64_t)nb < (uint64_t)npos + 1) out_counts[i] = LVA_ERR_POST_TOO_SHORT;
    else { out_counts[i] = 0; order.push_back(i); band_words += (size_t)nb; }
  }
  // longest first: the tail of the schedule is then made of short reads
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return len[a] > len[b]; });

  if ((size_t)n > d->results_cap) {
    if (d->d_results) (void)hipFree(d->d_results);
    d->d_results = nullptr; d->results_cap = 0;
    HIP_TRY(hipMalloc(&d->d_results, (size_t)n * rec_words * sizeof(uint32_t)));
    d->results_cap = (size_t)n;
  }
  if (!timed_total_started) HIP_TRY(hipEventRecord(d->ev_total0, d->stream));

  // band of every time step of every read (:677-679), evaluated here as the reference binary does
  band.assign(std::max<size_t>(band_words, 1), 0u);
  std::vector<size_t> band_at((size_t)n, 0);
  {
    size_t at = 0;
    for (int32_t r : order) {
      band_at[(size_t)r] = at;
      const Code& c = d->code[rc_flags && rc_flags[r] ? 1 : 0];
      const uint32_t nb = (uint32_t)len[r];
      // lazy mode: when was each position's row of either parity buffer last written?  Step t reads the buffer written
      // by steps of t-1's parity; only the row of position lo-1 can be older than t-1 ("stale"), and at odd t
      // its entries' messages live in the message buffer of the (even) step that wrote it
      std::vector<int64_t> last_w[2];
      if (d->g.lazy) { last_w[0].assign(c.npos + 1, -1); last_w[1].assign(c.npos + 1, -1); if (c.npos) last_w[1][0] = -1; }
      for (uint32_t t = 0; t < nb; ++t) {
        uint32_t lo, hi;
        working_band(c, t, nb, d->max_dev, &lo, &hi);
        uint32_t w = lo | (hi << 16);
        if (d->g.lazy) {
          const int pc = (int)((t + 1) & 1u);                // parity class of the steps that wrote step t's "prev" buffer: t-1
          if (t >= 1 && lo >= 1) {
            const int64_t lw = last_w[pc][lo - 1];
            if (lw >= 0 && lw != (int64_t)t - 1) w |= 1u << 30 | (uint32_t)((lw >> 1) & 1) << 31;
          }
          for (uint32_t p = lo; p < hi; ++p) last_w[t & 1u][p] = t;
        }
        band[at + t] = w;
      }
      at += nb;
    }
  }
  if (band.size() > d->band_cap) {
    if (d->d_band) (void)hipFree(d->d_band);
    d->d_band = nullptr; d->band_cap = 0;
    HIP_TRY(hipMalloc(&d->d_band, band.size() * sizeof(uint32_t)));
    d->band_cap = band.size();
  }
  HIP_TRY(hipMemcpyAsync(d->d_band, band.data(), band.size() * sizeof(uint32_t), hipMemcpyHostToDevice, d->stream));
  std::memset(&h0, 0, sizeof h0);
  h0.cap = d->work_cap;
  HIP_TRY(hipMemcpyAsync(d->d_work, &h0, sizeof h0, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(hipMemsetAsync(d->d_slots, 0, (size_t)d->slots * sizeof(SlotDesc), d->stream));   // nblk = 0: no slot takes part yet

  struct Slot { int32_t read = -1; uint32_t end = 0; };     // end: launch number after the read's last step
  std::vector<Slot> slot((size_t)std::min<size_t>((size_t)d->slots, std::max<size_t>(order.size(), 1)));
  size_t next = 0;
  gathered.assign((size_t)n, 0);
  d->prof.step_launches = 0; d->prof.read_steps = 0; d->prof.algorithmic_bytes = 0; d->prof.working_bytes = 0; d->prof.fixup_states = 0; d->prof.overflow_steps = 0;
  d->prof.dominant_kernel_ms = 0; d->prof.step_pair_ms = 0; d->prof.timed_launches = 0;
  const uint32_t band_max = std::min<uint32_t>(npos, 2 * d->max_dev);
  bool first_step = true;
  size_t active = 0, ev_used = 0;
  auto next_event = [&](hipEvent_t* out) -> int {
    if (ev_used == d->ev_pool.size()) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      d->ev_pool.push_back(e);
    }
    *out = d->ev_pool[ev_used++];
    return LVA_OK;
  };
  InitBatch ib; GatherBatch gb;
  ib.n = 0; ib.pad = 0; gb.n = 0;
  auto flush_inits = [&]() -> int {
    const int e = launch_init_slots(g, d->d_codes, d->d_trellis, ib, d->d_slots, d->stream);
    ib.n = 0;
    if (e) { g_hip_error = hipGetErrorString((hipError_t)e); return LVA_ERR_HIP; }
    return LVA_OK;
  };
  auto flush_gathers = [&]() -> int {
    const int e = launch_gather_finals(g, d->d_codes, d->d_trellis, gb, d->d_results, d->stream);
    gb.n = 0;
    if (e) { g_hip_error = hipGetErrorString((hipError_t)e); return LVA_ERR_HIP; }
    return LVA_OK;
  };
  for (;;) {
    size_t waiting = 0;              // slots whose read starts with the next launch (lazy mode's phase alignment)
    // (re)fill idle slots: the reads' descriptors and initial scores (:657-663) go in stream order, a batch of slots per launch
    // (behind the gathers of the reads that left them: the retire loop below runs first)
    for (size_t s = 0; s < slot.size(); ++s) {
      if (slot[s].read >= 0 || next >= order.size()) continue;
      const int32_t r = order[next++];
      SlotDesc sd;
      sd.post = post_dev + (size_t)beg[r] * 40;
      sd.band = d->d_band + band_at[(size_t)r];
      sd.nblk = (uint32_t)len[r]; sd.orient = rc_flags && rc_flags[r] ? 1u : 0u;
      // Lazy mode: every read starts on an EVEN launch (a read that arrives on an odd one idles for one launch: 1 in ~500),
      // so all slots are at an even time step on even launches and at an odd one on odd launches -- a launch then runs ONE
      // instance of lva_step_lazy over a grid without workgroups of the wrong kind (launch_step_fast, phase_aligned)
      sd.start = d->launch_no + (d->kernel == 4 ? (d->launch_no & 1u) : 0u); sd.pad = 0;
      slot[s].read = r; slot[s].end = sd.start + sd.nblk;
      if (sd.start != d->launch_no) ++waiting;
      ib.slot[ib.n] = (uint32_t)s; ib.desc[ib.n] = sd;
      if (++ib.n == (uint32_t)kTurnoverBatch) { const int st = flush_inits(); if (st) return st; }
      d->prof.algorithmic_bytes += d->code[sd.orient].algorithmic_bytes(sd.nblk, L, d->max_dev);
      d->prof.working_bytes += d->code[sd.orient].working_bytes(sd.nblk, L, d->max_dev);
      ++active;
    }
    { const int st = flush_inits(); if (st) return st; }
    if (active == 0) break;
    StepArgs a;
    a.slots = d->d_slots; a.steps = d->d_steps; a.nslots = (uint32_t)slot.size(); a.band_max = band_max;
    a.launch_no = d->launch_no; a.step_parity = d->launch_no & 1u;
    a.phase_aligned = d->kernel == 4 ? 1u : 0u;
    a.full_lo = d->full_lo; a.full_hi = d->full_hi; a.pad = 0;
    {
      const int e = launch_prepare_step(a, d->d_codes, d->d_steps, d->stream);
      if (e) { g_hip_error = hipGetErrorString((hipError_t)e); return LVA_ERR_HIP; }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (d->launch_events) {
      int st;
      if ((st = next_event(&e0)) || (st = next_event(&e1)) || (st = next_event(&e2))) return st;
      HIP_TRY(hipEventRecord(e0, d->stream));
      if (first_step) HIP_TRY(hipEventRecord(d->ev_step0, d->stream));
    } else if (first_step) {
      HIP_TRY(hipEventRecord(d->ev_step0, d->stream));
    }
    first_step = false;
    {
      const int e = d->kernel == 2 || d->kernel == 4
                        ? launch_step_fast(a, g, d->d_codes, d->d_trellis, d->d_work, reinterpret_cast<uint32_t*>(d->d_work + 1), d->stream, e1)
                        : d->kernel == 3 ? launch_step_wave(a, g, d->d_codes, d->d_trellis, d->stream)
                                         : launch_step_exact(a, g, d->d_codes, d->d_trellis, d->stream);
      if (e) { g_hip_error = hipGetErrorString((hipError_t)e); return LVA_ERR_HIP; }
    }
    if (e2) {
      if (d->kernel != 2 && d->kernel != 4) HIP_TRY(hipEventRecord(e1, d->stream));
      HIP_TRY(hipEventRecord(e2, d->stream));
    }
    ++d->launch_no;
    d->prof.step_launches += 1;
    d->prof.read_steps += active - waiting;
    // retire finished reads
    for (size_t s = 0; s < slot.size(); ++s) {
      if (slot[s].read < 0 || slot[s].end != d->launch_no) continue;
      const int32_t r = slot[s].read;
      const uint32_t nb = (uint32_t)len[r], orient = rc_flags && rc_flags[r] ? 1u : 0u;
      const uint32_t last = band[band_at[(size_t)r] + nb - 1];
      if ((last & 0xFFFFu) <= npos - 1 && npos - 1 < ((last >> 16) & 0x3FFFu)) {   // otherwise the final state was never written: empty list
        gb.a[gb.n] = GatherArgs{(uint32_t)s, (uint32_t)(nb & 1u), orient, (uint32_t)r, nb};
        if (++gb.n == (uint32_t)kTurnoverBatch) { const int st = flush_gathers(); if (st) return st; }
        gathered[(size_t)r] = 1;
      }
      slot[s].read = -1;
      --active;
    }
    { const int st = flush_gathers(); if (st) return st; }
  }
  if (!first_step) HIP_TRY(hipEventRecord(d->ev_step1, d->stream));
  host.assign((size_t)n * rec_words, 0u);
  if (n > 0) HIP_TRY(hipMemcpyAsync(host.data(), d->d_results, host.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipMemcpyAsync(&h1, d->d_work, sizeof h1, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipEventRecord(d->ev_total1, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  drain.armed = false;
  d->prof.fixup_states = h1.total;
  d->prof.overflow_steps = h1.overflow_steps;
  for (int i = 0; i < 4; ++i) d->prof.fixup_reason[i] = h1.reason[i];
  float ms = 0;
  if (!first_step) { HIP_TRY(hipEventElapsedTime(&ms, d->ev_step0, d->ev_step1)); }
  d->prof.step_kernel_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, d->ev_total0, d->ev_total1));
  d->prof.total_ms = ms;
  for (size_t i = 0; i + 3 <= ev_used; i += 3) {
    float a_ms = 0, b_ms = 0;
    HIP_TRY(hipEventElapsedTime(&a_ms, d->ev_pool[i], d->ev_pool[i + 1]));
    HIP_TRY(hipEventElapsedTime(&b_ms, d->ev_pool[i], d->ev_pool[i + 2]));
    d->prof.dominant_kernel_ms += a_ms;
    d->prof.step_pair_ms += b_ms;
    d->prof.timed_launches += 1;
  }

  for (int32_t i = 0; i < n; ++i) {
    if (out_counts[i] < 0) continue;
    if (!gathered[(size_t)i]) { out_counts[i] = 0; continue; }
    finish_read(d, rc_flags && rc_flags[i] ? 1 : 0, host.data() + (size_t)i * rec_words,
                out_msgs + (size_t)i * L * d->code[0].msg_len, out_scores ? out_scores + (size_t)i * L : nullptr,
                &out_counts[i]);
  }
  return LVA_OK;
}

int lva_decode_batch_device(lva_decoder* d, const float* post_dev, const int64_t* row_offsets, int32_t n_reads,
                            const uint8_t* rc_flags, uint8_t* out_msgs, float* out_scores, int32_t* out_counts) {
  if (!d || n_reads < 0 || !row_offsets || (n_reads > 0 && (!post_dev || !out_msgs || !out_counts))) return LVA_ERR_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  std::vector<int64_t> len((size_t)n_reads);
  for (int32_t i = 0; i < n_reads; ++i) len[i] = row_offsets[i + 1] - row_offsets[i];
  return decode_impl(d, post_dev, row_offsets, len.data(), n_reads, rc_flags, out_msgs, out_scores, out_counts, false);
}

int lva_decode_windows_device(lva_decoder* d, const float* post_dev, const int64_t* first_block, const int64_t* n_blocks,
                              int32_t n_reads, const uint8_t* rc_flags, uint8_t* out_msgs, float* out_scores,
                              int32_t* out_counts) {
  if (!d || n_reads < 0 || (n_reads > 0 && (!post_dev || !first_block || !n_blocks || !out_msgs || !out_counts))) return LVA_ERR_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  return decode_impl(d, post_dev, first_block, n_blocks, n_reads, rc_flags, out_msgs, out_scores, out_counts, false);
}

int lva_decode_batch(lva_decoder* d, const float* post, const int64_t* row_offsets, int32_t n_reads,
                     const uint8_t* rc_flags, uint8_t* out_msgs, float* out_scores, int32_t* out_counts) {
  if (!d || n_reads < 0 || !row_offsets || (n_reads > 0 && (!post || !out_msgs || !out_counts))) return LVA_ERR_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  const int64_t blocks = n_reads > 0 ? row_offsets[n_reads] : 0;
  if (blocks < 0) return LVA_ERR_ARG;
  float* dev = nullptr;
  const size_t bytes = (size_t)std::max<int64_t>(blocks, 1) * 40 * sizeof(float);
  HIP_TRY(hipMalloc(&dev, bytes));
  hipError_t e = hipEventRecord(d->ev_total0, d->stream);
  if (e == hipSuccess && blocks > 0)
    e = hipMemcpyAsync(dev, post, (size_t)blocks * 40 * sizeof(float), hipMemcpyHostToDevice, d->stream);
  if (e == hipSuccess) e = hipEventRecord(d->ev_h2d, d->stream);
  if (e != hipSuccess) { g_hip_error = hipGetErrorString(e); (void)hipStreamSynchronize(d->stream); (void)hipFree(dev); return LVA_ERR_HIP; }
  std::vector<int64_t> len((size_t)n_reads);
  for (int32_t i = 0; i < n_reads; ++i) len[i] = row_offsets[i + 1] - row_offsets[i];
  const int st = decode_impl(d, dev, row_offsets, len.data(), n_reads, rc_flags, out_msgs, out_scores, out_counts, true);
  (void)hipStreamSynchronize(d->stream);
  float ms = 0;
  d->prof.h2d_ms = (st == LVA_OK && hipEventElapsedTime(&ms, d->ev_total0, d->ev_h2d) == hipSuccess) ? ms : 0.0;
  d->prof.h2d_bytes = (uint64_t)blocks * 40 * sizeof(float);
  (void)hipFree(dev);
  return st;
}

int lva_device_alloc(lva_decoder* d, uint64_t bytes, void** out_dev_ptr) {
  if (!d || !out_dev_ptr) return LVA_ERR_ARG;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipMalloc(out_dev_ptr, (size_t)std::max<uint64_t>(bytes, 1)));
  return LVA_OK;
}

int lva_device_free(lva_decoder* d, void* dev_ptr) {
  if (!d) return LVA_ERR_ARG;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipFree(dev_ptr));
  return LVA_OK;
}

int lva_device_upload(lva_decoder* d, void* dev_dst, const void* host_src, uint64_t bytes) {
  if (!d || (bytes && (!dev_dst || !host_src))) return LVA_ERR_ARG;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipMemcpyAsync(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return LVA_OK;
}

int lva_device_synchronize(lva_decoder* d) {
  if (!d) return LVA_ERR_ARG;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return LVA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// basecall of the posterior matrix and barcode localisation.
// ---------------------------------------------------------------------------------------------
namespace {

// bc_search packs (edit distance << 20 | window index) into one word for its minimum reduction
// (bc_kernels.hip): a read may have at most 2^20 blocks / called bases.  Real reads have a few thousand.
constexpr int64_t kBcMaxBlocks = (int64_t)1 << 20;

struct DevBlock {              // one device allocation carved into 256-byte aligned pieces
  char* base = nullptr;
  size_t used = 0, cap = 0;
  ~DevBlock() { if (base) (void)hipFree(base); }
  static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
  template <typename T> T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + used);
    used += pad(std::max<size_t>(count, 1) * sizeof(T));
    return p;
  }
};

bool rc_pattern(const char* src, int len, char* dst) {        // helper.reverse_complement (helper.py:227-229)
  for (int i = 0; i < len; ++i) {
    char c;
    switch (src[len - 1 - i]) {
      case 'A': c = 'T'; break;
      case 'C': c = 'G'; break;
      case 'G': c = 'C'; break;
      case 'T': c = 'A'; break;
      case 'N': c = 'N'; break;
      default: return false;
    }
    dst[i] = c;
  }
  return true;
}

int make_patterns(const char* start_bc, const char* end_bc, int n_orient, BcPatterns* p) {
  if (!start_bc || !end_bc) return LVA_ERR_ARG;
  const size_t ls = std::strlen(start_bc), le = std::strlen(end_bc);
  if (ls == 0 || le == 0 || ls > (size_t)kMaxBarcode || le > (size_t)kMaxBarcode) return LVA_ERR_ARG;
  std::memset(p, 0, sizeof *p);
  p->len[0] = (uint8_t)ls; p->len[1] = (uint8_t)le;
  std::memcpy(p->pat[0], start_bc, ls);
  std::memcpy(p->pat[1], end_bc, le);
  if (n_orient == 2) {         // generate_decoded_lists.py:33-34: START_BARCODE_RC = rc(END), END_BARCODE_RC = rc(START)
    p->len[2] = (uint8_t)le; p->len[3] = (uint8_t)ls;
    if (!rc_pattern(end_bc, (int)le, p->pat[2]) || !rc_pattern(start_bc, (int)ls, p->pat[3])) return LVA_ERR_ARG;
  }
  return LVA_OK;
}

// basecall (and, when n_orient > 0, barcode localisation) of n reads whose posteriors are resident
int bc_run(lva_decoder* d, const float* post_dev, const int64_t* row_offsets, int32_t n, const BcPatterns* pat,
           int n_orient, uint32_t min_len, char* bases_out, uint32_t* trans_out, int32_t* nbases_out,
           lva_payload_pos* pos_out) {
  static_assert(sizeof(lva_payload_pos) == sizeof(BcResult), "lva_payload_pos layout");
  if (n == 0) return LVA_OK;
  for (int32_t i = 0; i < n; ++i)
    if (row_offsets[i + 1] < row_offsets[i] || row_offsets[i + 1] - row_offsets[i] > kBcMaxBlocks) return LVA_ERR_ARG;
  const size_t T = (size_t)(row_offsets[n] - row_offsets[0]);
  if (T >= ((size_t)1 << 31)) return LVA_ERR_ARG;          // 32-bit block offsets inside the kernels
  if (row_offsets[0] != 0) return LVA_ERR_ARG;
  DevBlock blk;
  blk.cap = DevBlock::pad(8 * ((size_t)n + 1)) + DevBlock::pad(8 * T + 8) + DevBlock::pad(T + n + 1) + DevBlock::pad(T + 1) +
            DevBlock::pad(4 * T + 4) + DevBlock::pad(4 * (size_t)n) + DevBlock::pad(16 * (size_t)n) + DevBlock::pad(24 * (size_t)n);
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&blk.base), blk.cap));
  int64_t* d_off = blk.take<int64_t>((size_t)n + 1);
  uint32_t* d_tb = blk.take<uint32_t>(2 * T);          // 8 back-pointer bytes per block
  uint8_t* d_path = blk.take<uint8_t>(T + n);
  char* d_bases = blk.take<char>(T);
  uint32_t* d_trans = blk.take<uint32_t>(T);
  int32_t* d_nb = blk.take<int32_t>(n);
  uint32_t* d_best = blk.take<uint32_t>(4 * (size_t)n);
  BcResult* d_res = blk.take<BcResult>(n);
  HIP_TRY(hipMemcpyAsync(d_off, row_offsets, 8 * ((size_t)n + 1), hipMemcpyHostToDevice, d->stream));
  int e = launch_bc_basecall(post_dev, d_off, n, d_tb, d_path, d_bases, d_trans, d_nb, d->stream);
  if (!e && n_orient > 0) e = launch_bc_search(d_bases, d_off, d_nb, n, *pat, n_orient, d_best, d->stream);
  if (!e && n_orient > 0) e = launch_bc_finalize(d_trans, d_off, d_nb, n, *pat, n_orient, min_len, d_best, d_res, d->stream);
  if (e) { g_hip_error = hipGetErrorString((hipError_t)e); return LVA_ERR_HIP; }
  if (bases_out && T) HIP_TRY(hipMemcpyAsync(bases_out, d_bases, T, hipMemcpyDeviceToHost, d->stream));
  if (trans_out && T) HIP_TRY(hipMemcpyAsync(trans_out, d_trans, 4 * T, hipMemcpyDeviceToHost, d->stream));
  if (nbases_out) HIP_TRY(hipMemcpyAsync(nbases_out, d_nb, 4 * (size_t)n, hipMemcpyDeviceToHost, d->stream));
  if (pos_out && n_orient > 0) HIP_TRY(hipMemcpyAsync(pos_out, d_res, sizeof(BcResult) * (size_t)n, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return LVA_OK;
}

// host posteriors -> device copy for the duration of the call
struct HostPost {
  float* dev = nullptr;
  ~HostPost() { if (dev) (void)hipFree(dev); }
};

int upload_post(lva_decoder* d, const float* post, const int64_t* row_offsets, int32_t n, HostPost* hp) {
  const int64_t blocks = n > 0 ? row_offsets[n] : 0;
  if (blocks < 0) return LVA_ERR_ARG;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&hp->dev), (size_t)std::max<int64_t>(blocks, 1) * 160));
  if (blocks > 0) HIP_TRY(hipMemcpyAsync(hp->dev, post, (size_t)blocks * 160, hipMemcpyHostToDevice, d->stream));
  return LVA_OK;
}

}  // namespace

extern "C" {

int lva_basecall_batch_device(lva_decoder* d, const float* post_dev, const int64_t* row_offsets, int32_t n_reads,
                              char* bases_out, uint32_t* trans_out, int32_t* nbases_out) {
  if (!d || n_reads < 0 || !row_offsets || (n_reads > 0 && (!post_dev || !nbases_out))) return LVA_ERR_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  return bc_run(d, post_dev, row_offsets, n_reads, nullptr, 0, 0, bases_out, trans_out, nbases_out, nullptr);
}

int lva_basecall_batch(lva_decoder* d, const float* post, const int64_t* row_offsets, int32_t n_reads, char* bases_out,
                       uint32_t* trans_out, int32_t* nbases_out) {
  if (!d || n_reads < 0 || !row_offsets || (n_reads > 0 && (!post || !nbases_out))) return LVA_ERR_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  HostPost hp;
  const int st = upload_post(d, post, row_offsets, n_reads, &hp);
  if (st != LVA_OK) return st;
  return bc_run(d, hp.dev, row_offsets, n_reads, nullptr, 0, 0, bases_out, trans_out, nbases_out, nullptr);
}

int lva_locate_payload_batch_device(lva_decoder* d, const float* post_dev, const int64_t* row_offsets, int32_t n_reads,
                                    const char* start_barcode, const char* end_barcode, uint32_t min_len,
                                    lva_payload_pos* out) {
  if (!d || n_reads < 0 || !row_offsets || (n_reads > 0 && (!post_dev || !out))) return LVA_ERR_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  BcPatterns pat;
  const int st = make_patterns(start_barcode, end_barcode, 2, &pat);
  if (st != LVA_OK) return st;
  return bc_run(d, post_dev, row_offsets, n_reads, &pat, 2, min_len, nullptr, nullptr, nullptr, out);
}

int lva_locate_payload_batch(lva_decoder* d, const float* post, const int64_t* row_offsets, int32_t n_reads,
                             const char* start_barcode, const char* end_barcode, uint32_t min_len, lva_payload_pos* out) {
  if (!d || n_reads < 0 || !row_offsets || (n_reads > 0 && (!post || !out))) return LVA_ERR_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  BcPatterns pat;
  int st = make_patterns(start_barcode, end_barcode, 2, &pat);
  if (st != LVA_OK) return st;
  HostPost hp;
  st = upload_post(d, post, row_offsets, n_reads, &hp);
  if (st != LVA_OK) return st;
  return bc_run(d, hp.dev, row_offsets, n_reads, &pat, 2, min_len, nullptr, nullptr, nullptr, out);
}

int lva_find_barcode_batch(lva_decoder* d, const char* bases, const uint32_t* trans, const int64_t* base_offsets,
                           int32_t n_reads, const char* start_barcode, const char* end_barcode, lva_payload_pos* out) {
  if (!d || n_reads < 0 || !base_offsets || (n_reads > 0 && (!bases || !trans || !out))) return LVA_ERR_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  if (n_reads == 0) return LVA_OK;
  BcPatterns pat;
  const int st = make_patterns(start_barcode, end_barcode, 1, &pat);
  if (st != LVA_OK) return st;
  if (base_offsets[0] != 0) return LVA_ERR_ARG;
  std::vector<int32_t> nb(n_reads);
  for (int32_t i = 0; i < n_reads; ++i) {
    if (base_offsets[i + 1] < base_offsets[i] || base_offsets[i + 1] - base_offsets[i] > kBcMaxBlocks) return LVA_ERR_ARG;
    nb[i] = (int32_t)(base_offsets[i + 1] - base_offsets[i]);
  }
  const size_t T = (size_t)base_offsets[n_reads], n = (size_t)n_reads;
  DevBlock blk;
  blk.cap = DevBlock::pad(8 * (n + 1)) + DevBlock::pad(T + 1) + DevBlock::pad(4 * T + 4) + DevBlock::pad(4 * n) +
            DevBlock::pad(16 * n) + DevBlock::pad(24 * n);
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&blk.base), blk.cap));
  int64_t* d_off = blk.take<int64_t>(n + 1);
  char* d_bases = blk.take<char>(T);
  uint32_t* d_trans = blk.take<uint32_t>(T);
  int32_t* d_nb = blk.take<int32_t>(n);
  uint32_t* d_best = blk.take<uint32_t>(4 * n);
  BcResult* d_res = blk.take<BcResult>(n);
  HIP_TRY(hipMemcpyAsync(d_off, base_offsets, 8 * (n + 1), hipMemcpyHostToDevice, d->stream));
  if (T) HIP_TRY(hipMemcpyAsync(d_bases, bases, T, hipMemcpyHostToDevice, d->stream));
  if (T) HIP_TRY(hipMemcpyAsync(d_trans, trans, 4 * T, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(hipMemcpyAsync(d_nb, nb.data(), 4 * n, hipMemcpyHostToDevice, d->stream));
  int e = launch_bc_search(d_bases, d_off, d_nb, n_reads, pat, 1, d_best, d->stream);
  if (!e) e = launch_bc_finalize(d_trans, d_off, d_nb, n_reads, pat, 1, 0, d_best, d_res, d->stream);
  if (e) { g_hip_error = hipGetErrorString((hipError_t)e); return LVA_ERR_HIP; }
  HIP_TRY(hipMemcpyAsync(out, d_res, sizeof(BcResult) * n, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return LVA_OK;
}

}  // extern "C"
