// lva_code.cpp -- code parameters, kernel tables and the encoder (host only).
// Reference behaviour: viterbi/viterbi_convolutional_code.cpp, lines cited as ":NNN".
#include "lva_code.h"

#include <cmath>
#include <cstring>

#include "../../include/lva_decoder.h"

namespace lva {

namespace {

// fixed pseudo-random word of message bit index i (message fingerprints, lva_device.h)
uint32_t fp_word(uint32_t i) {
  uint64_t z = (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 16) | 1u;
}

uint32_t reverse_bits(uint32_t v, uint32_t n) {           // :417-424
  uint32_t r = 0;
  for (uint32_t i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
  return r;
}

struct RateDef { int rate; int len; uint8_t types[5]; };
// block types per rate (:296-339).  type 0: one message bit -> one base (both parity bits
// kept); types 1,2,3: two message bits -> one base (two of the four parity bits kept).
const RateDef kRates[] = {
    {1, 1, {0}}, {2, 3, {0, 2, 0}}, {3, 2, {0, 1}}, {4, 5, {0, 3, 0, 2, 1}}, {5, 3, {0, 1, 2}}, {7, 4, {0, 3, 1, 1}}};

struct MemDef { int m; uint32_t g0, g1, init; };
const MemDef kMems[] = {                                    // :269-289
    {6, 0171, 0133, 0b100101},
    {8, 0515, 0677, 0b10010110},
    {11, 05537, 06131, 0b10010110001},
    {14, 075063, 056711, 0b10010110001101}};

}  // namespace

uint32_t Code::out_bit(int k, uint32_t st, uint32_t bit) const {
  const uint32_t reg = st | (bit ? nconv : 0u);
  return (uint32_t)__builtin_parity(reg & g[k]) ^ (uint32_t)rc;
}

int build_code(Code* c, int mem_conv, int rate, uint32_t msg_len, int rc, const char* sync_marker,
               uint32_t sync_period) {
  *c = Code();
  const MemDef* md = nullptr;
  for (const auto& d : kMems) if (d.m == mem_conv) md = &d;
  if (!md) return LVA_ERR_MEM_CONV;
  const RateDef* rd = nullptr;
  for (const auto& d : kRates) if (d.rate == rate) rd = &d;
  if (!rd) return LVA_ERR_RATE;

  c->mem_conv = mem_conv; c->rate = rate; c->rc = rc ? 1 : 0; c->msg_len = msg_len;
  c->nconv = 1u << mem_conv;
  c->g[0] = md->g0; c->g[1] = md->g1; c->init = md->init;
  c->fin = reverse_bits(md->init, (uint32_t)mem_conv);       // :294
  c->plen = rd->len;
  std::memcpy(c->pattern, rd->types, (size_t)rd->len);

  // positions (:344-357): a base boundary must coincide with the end of message+termination
  const uint32_t total = msg_len + (uint32_t)mem_conv;
  uint32_t used = 0;
  c->npos = 1;
  while (used < total) {
    used += c->pattern[(c->npos - 1) % (uint32_t)c->plen] == 0 ? 1u : 2u;
    if (c->npos >= kMaxPos) return LVA_ERR_MSG_TOO_LONG;
    c->pos2msg[c->npos++] = used;
  }
  if (used != total) return LVA_ERR_MSG_LEN;

  if (c->rc) {                                               // :359-386
    c->g[0] = reverse_bits(c->g[0], (uint32_t)mem_conv + 1);
    c->g[1] = reverse_bits(c->g[1], (uint32_t)mem_conv + 1);
    const uint32_t a = reverse_bits(c->fin, (uint32_t)mem_conv), b = reverse_bits(c->init, (uint32_t)mem_conv);
    c->init = a; c->fin = b;
    uint8_t fwd[16];
    std::memcpy(fwd, c->pattern, sizeof fwd);
    const uint8_t seen_backwards[4] = {0, 2, 1, 3};
    const uint32_t last_type_idx = (c->npos - 2) % (uint32_t)c->plen;
    for (uint32_t i = 0; i < (uint32_t)c->plen; ++i)
      c->pattern[i] = seen_backwards[fwd[((uint32_t)c->plen - i + last_type_idx) % (uint32_t)c->plen]];
    uint32_t tmp[kMaxPos + 1];
    for (uint32_t i = 0; i < c->npos; ++i) tmp[i] = total - c->pos2msg[c->npos - 1 - i];
    std::memcpy(c->pos2msg, tmp, c->npos * sizeof(uint32_t));
  }

  if (sync_marker && sync_marker[0]) {                       // :388-414
    const size_t n = std::strlen(sync_marker);
    if (n >= kMaxPos) return LVA_ERR_SYNC;
    if (sync_period < n) return LVA_ERR_SYNC;
    for (size_t i = 0; i < n; ++i) {
      if (sync_marker[i] != '0' && sync_marker[i] != '1') return LVA_ERR_SYNC;
      c->sync[i] = (uint8_t)(sync_marker[i] - '0');
    }
    c->sync_len = (uint32_t)n; c->sync_period = sync_period;
  }

  // ---- per-position tables -------------------------------------------------------------
  const int m = mem_conv;
  for (uint32_t pos = 0; pos < c->npos; ++pos) {
    c->ptype[pos] = pos == 0 ? 0 : c->pattern[(pos - 1) % (uint32_t)c->plen];   // :693-696
    // is_valid_state (:944-978) folded into mask/value: register bit (m-1-age) holds message
    // index pos2msg[pos]-1-age; indices before the message pin to the initial state, indices
    // past it to the terminating bits, sync positions to the marker.
    uint32_t mask = 0, val = 0;
    for (int age = 0; age < m; ++age) {
      const int64_t idx = (int64_t)c->pos2msg[pos] - 1 - age;
      const int64_t idx_fwd = c->rc ? (int64_t)msg_len - 1 - idx : idx;
      const uint32_t bitpos = (uint32_t)(m - 1 - age);
      int want = -1;
      if (idx < 0) want = (int)((c->init >> (m + idx)) & 1u);
      else if (idx >= (int64_t)msg_len) want = (int)((c->fin >> (idx - (int64_t)msg_len)) & 1u);
      else if (c->sync_len > 0 && (idx_fwd % (int64_t)c->sync_period) < (int64_t)c->sync_len)
        want = c->sync[idx_fwd % (int64_t)c->sync_period];
      if (want >= 0) { mask |= 1u << bitpos; val |= (uint32_t)want << bitpos; }
    }
    c->vmask[pos] = mask; c->vval[pos] = val;
  }
  // bits consumed per position and the fingerprint delta of each step (kernel bookkeeping,
  // no counterpart in the reference)
  for (uint32_t pos = 1; pos < c->npos; ++pos) {
    const uint32_t sh = c->shift_of(c->ptype[pos]), n0 = c->nbits[pos - 1];
    c->nbits[pos] = n0 + sh;
    for (uint32_t nb = 0; nb < 4; ++nb) {
      uint32_t d = 0;
      if (sh == 1) d = (nb & 1u) ? fp_word(n0) : 0u;
      else d = ((nb & 2u) ? fp_word(n0) : 0u) ^ ((nb & 1u) ? fp_word(n0 + 1) : 0u);
      c->fpc[pos][nb] = d;
    }
  }

  // ---- predecessor table (find_prev_states :860-942) -------------------------------------
  // For target conv state t entered under block type T by emitting base b, enumerate the
  // candidate sources in the reference's order (lost bits ascending) and record the one that
  // emits b.  More than one match would make the reference push several heap entries for the
  // same source crf state; the kernels assume at most one, so such a code is refused.
  bool used_type[4] = {false, false, false, false};
  for (int i = 0; i < c->plen; ++i) used_type[c->pattern[i]] = true;
  for (int T = 0; T < 4; ++T) {
    if (!used_type[T]) continue;
    c->predtab[T].assign(c->nconv, 0);
    const uint32_t sh = c->shift_of(T);
    for (uint32_t t = 0; t < c->nconv; ++t) {
      const uint32_t newest = t >> (m - 1), second = (t >> (m - 2)) & 1u;
      uint16_t packed = 0;
      for (uint32_t y = 0; y < (1u << sh); ++y) {
        const uint32_t src = ((t << sh) | y) & (c->nconv - 1);
        uint32_t base;
        if (T == 0) {
          base = 2 * c->out_bit(0, src, newest) + c->out_bit(1, src, newest);              // :893-894
        } else {
          const uint32_t mid = ((t << 1) | (y >> 1)) & (c->nconv - 1);   // after the first of the two bits
          const uint32_t o0 = c->out_bit(0, src, second), o1 = c->out_bit(1, src, second);
          const uint32_t o2 = c->out_bit(0, mid, newest), o3 = c->out_bit(1, mid, newest);
          uint32_t hi, lo;
          if (T == 1) { hi = o1; lo = o2; } else if (T == 2) { hi = o0; lo = o3; } else { hi = o1; lo = o3; }
          base = c->rc ? 2 * lo + hi : 2 * hi + lo;                                       // :916-926
        }
        const uint16_t nib = (uint16_t)((packed >> (4 * base)) & 0xF);
        if (nib & 8) return LVA_ERR_UNSUPPORTED;   // two sources emit the same base
        packed = (uint16_t)(packed | ((8u | y) << (4 * base)));
      }
      c->predtab[T][t] = packed;
    }
  }

  // ---- structurally reachable states per position ----------------------------
  c->reach_per_pos.assign(c->npos, 0);
  for (uint32_t pos = 0; pos < c->npos; ++pos) {
    uint32_t n = 0;
    for (uint32_t t = 0; t < c->nconv; ++t) {
      if (!c->valid(pos, t)) continue;
      if (pos == 0) { n += kCrf; continue; }
      const uint16_t pk = c->predtab[c->ptype[pos]][t];
      for (int b = 0; b < 4; ++b) if ((pk >> (4 * b)) & 8) n += 2;   // flip b and flop b+4
    }
    c->reach_per_pos[pos] = n;
  }
  return LVA_OK;
}

int Code::encode(const uint8_t* msg, uint8_t* bases_out) const {
  if (rc) return LVA_ERR_ARG;
  const uint32_t total = msg_len + (uint32_t)mem_conv;
  std::vector<uint8_t> parity(2 * (size_t)total + 4, 0);
  uint32_t st = init;
  for (uint32_t i = 0; i < total; ++i) {
    const uint32_t bit = i < msg_len ? (uint32_t)(msg[i] & 1) : (fin >> (i - msg_len)) & 1u;   // :458-464
    parity[2 * i] = (uint8_t)out_bit(0, st, bit);
    parity[2 * i + 1] = (uint8_t)out_bit(1, st, bit);
    st = (st | (bit ? nconv : 0u)) >> 1;                                                        // :426-431
  }
  if (st != fin) return LVA_ERR_MSG_LEN;                                                        // :465-467
  // which parity bits of each block survive puncturing (:473-494)
  static const uint8_t keep_hi[4] = {0, 1, 0, 1}, keep_lo[4] = {1, 2, 3, 3}, width[4] = {2, 4, 4, 4};
  uint32_t k = 0;
  for (uint32_t pos = 0; pos + 1 < npos; ++pos) {
    const uint8_t T = pattern[pos % (uint32_t)plen];
    bases_out[pos] = (uint8_t)(2 * parity[k + keep_hi[T]] + parity[k + keep_lo[T]]);
    k += width[T];
  }
  return k == 2 * total ? LVA_OK : LVA_ERR_MSG_LEN;                                             // :496-497
}

void Code::band(uint32_t t, uint32_t nblk, uint32_t max_dev, uint32_t* lo, uint32_t* hi) const {
  // :677-679.  `(double)t / nblk * nstate_pos - max_deviation` is contracted to one fused
  // multiply-subtract by g++ -O3 -march=native (install.sh:9) on FMA hosts, and the reference
  // results depend on it (about 1 step in 1400 lands on a different integer otherwise).
  const double q = (double)t / (double)nblk;
  const double centre = std::fma(q, (double)npos, -(double)max_dev);
  int64_t s = (int64_t)centre;
  if (s < 0) s = 0;
  const uint32_t start = (uint32_t)s;
  uint32_t end = start + 2u * max_dev;          // uint32_t arithmetic as in the reference
  if (end > npos) end = npos;
  *lo = start; *hi = end;
}

// The band of time step t the kernels work on: the reference's (:677-679, Code::band) without the positions whose lists
// cannot matter.
//  * A path advances at most one position per time step, so after step t only positions <= t + 1 hold a finite score: the
//    reference writes -inf lists above (:799), the kernels neither write nor read them -- the band ends at t + 2 for them, and
//    "beyond the previous band end" already reads as -inf (the upper band edge).  3.8 % of a read's (step, position) pairs.
//  * A state at position p after step t can still reach the final position only if p >= npos - nblk + t; states below feed
//    nothing that the final selection (:806-824) reads -- a state's predecessors lie one position lower or one step earlier, so
//    states that matter depend on states that matter only -- and are neither written nor read: the band starts there.  1.2 %.
// (tests/test_host_logic.py checks both against a reachability computation on the reference's band.)
void Code::working_band(uint32_t t, uint32_t nblk, uint32_t max_dev, uint32_t* lo_out, uint32_t* hi_out) const {
  uint32_t lo, hi;
  band(t, nblk, max_dev, &lo, &hi);
  hi = std::min<uint32_t>(hi, t + 2);
  const int64_t alive = (int64_t)npos - (int64_t)nblk + (int64_t)t;
  if (alive > (int64_t)lo) lo = (uint32_t)std::min<int64_t>(alive, hi);
  *lo_out = lo; *hi_out = hi;
}

static double band_bytes(const Code& c, uint32_t nblk, uint32_t list_size, uint32_t max_dev, bool working) {
  const double entry = 4.0 + 4.0 * c.msg_words();
  std::vector<uint64_t> prefix(c.npos + 1, 0);
  for (uint32_t p = 0; p < c.npos; ++p) prefix[p + 1] = prefix[p] + c.reach_per_pos[p];
  double total = 0;
  for (uint32_t t = 0; t < nblk; ++t) {
    uint32_t lo, hi;
    if (working) c.working_band(t, nblk, max_dev, &lo, &hi); else c.band(t, nblk, max_dev, &lo, &hi);
    const uint64_t R = hi > lo ? prefix[hi] - prefix[lo] : 0;
    total += 2.0 * (double)R * list_size * entry + 160.0;
  }
  return total;
}

double Code::algorithmic_bytes(uint32_t nblk, uint32_t list_size, uint32_t max_dev) const { return band_bytes(*this, nblk, list_size, max_dev, false); }
double Code::working_bytes(uint32_t nblk, uint32_t list_size, uint32_t max_dev) const { return band_bytes(*this, nblk, list_size, max_dev, true); }

}  // namespace lva
