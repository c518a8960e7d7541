// rs_kernels.hip -- SURVEY.md section 8(f) row N4: the Reed-Solomon outer code on the device.
//
// The reference decodes the outer code one 16-bit column of the oligo payloads at a time: for every column
// RSCode_schifra/RSCode_16bit_fileio.py (MainDecoder :289-299 -> RS_decode_16bit :87-137) recompiles
// schifra_RS_16bit_fileio.cpp and runs it on a full-length RS(65535, 65535 - redundancy) block over GF(2^16)
// whose first 65535 - n_total symbols are padding (ASCII "00").  Columns are independent codewords that share
// one erasure list (a missing oligo erases its symbol in every column), so here ONE launch decodes all columns:
// one workgroup per codeword, the polynomials of schifra's decoder (schifra_reed_solomon_decoder.hpp:64-424;
// citations ":NNN" below are lines of that file) in LDS, the loops over syndromes / coefficients / field
// elements spread over the 256 threads:
//   syndromes        S_i = R(alpha^i), i < fec (:227-240); the padding's contribution is a closed-form geometric sum
//   erasure locator  Gamma = prod (1 + alpha^loc x) (:211-225, :242-248)
//   modified Berlekamp-Massey from round = #erasures (:296-337), discrepancy by a workgroup reduction (:275-294)
//   Chien search     over alpha^1 .. alpha^65535 (:250-273), 256 field elements per thread, roots kept in order
//   Forney           omega = Lambda S mod x^fec, Lambda', error values (:339-385)
// with schifra's own success / failure decisions (:66-75, :105-146, :362-383), so that a column the reference gives
// up on comes back as the reference's fill bytes and a miscorrection is the same miscorrection.
// Field: primitive polynomial x^16+x^12+x^3+x+1 (schifra_galois_field.hpp:511), generator roots alpha^0..alpha^(fec-1)
// (schifra_RS_16bit_fileio.cpp:60-75 with generator_polynomial_index 0).  Integer work throughout: bit-exact.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lva_decoder.h"

namespace {

constexpr int kN = 65535;           // code length = field size - 1
constexpr int kMaxFec = 4096;       // LDS budget: 4 polynomials of 2 fec + 8 coefficients, syndromes, omega, roots
constexpr int kThreads = 256;

struct Gf {
  const uint16_t* ex;               // alpha^i for i in [0, 2N)  (doubled: no reduction after adding two logarithms)
  const uint16_t* lg;               // log_alpha(v), v in [1, 65535]
  __device__ __forceinline__ uint32_t mul(uint32_t a, uint32_t b) const { return (a && b) ? ex[lg[a] + lg[b]] : 0u; }
  __device__ __forceinline__ uint32_t div(uint32_t a, uint32_t b) const { return (a && b) ? ex[lg[a] + kN - lg[b]] : 0u; }
  __device__ __forceinline__ uint32_t pw(uint32_t e) const { return ex[e % kN]; }       // alpha^e
};

// XOR of one value per thread over the workgroup; every thread gets the result.  `red` = 8 words of LDS.
__device__ __forceinline__ uint32_t block_xor(uint32_t v, uint32_t* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] ^ red[1] ^ red[2] ^ red[3];
}

__device__ __forceinline__ uint32_t block_max(uint32_t v, uint32_t* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t w = __shfl_xor(v, o); v = w > v ? w : v; }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t m = red[0];
  for (int i = 1; i < 4; ++i) m = red[i] > m ? red[i] : m;
  return m;
}

// grid = codewords, block = 256 threads, dynamic LDS (rs_lds_bytes).
// sym [cw][n_total] received symbols; er [S] erased positions in [0, n_total); out [cw][n_out]; ok [cw].
__global__ __launch_bounds__(kThreads) void rs_decode_kernel(const uint16_t* __restrict__ sym, int n_total, int fec,
                                                              const int* __restrict__ er, int S, uint32_t pad_sym, uint32_t fail_sym,
                                                              const uint16_t* __restrict__ gexp, const uint16_t* __restrict__ glog,
                                                              uint16_t* __restrict__ out, int n_out, int* __restrict__ ok_out) {
  extern __shared__ uint32_t lds[];
  const int tid = threadIdx.x, cw = blockIdx.x;
  const int cap = 2 * fec + 8;
  uint16_t* lam = reinterpret_cast<uint16_t*>(lds);          // Lambda
  uint16_t* tmp = lam + cap;                                 // tau / scratch
  uint16_t* prv = tmp + cap;                                 // previous_lambda without its x^shift factor
  uint16_t* llg = prv + cap;                                 // log of the coefficients of the polynomial being evaluated
  uint16_t* syn = llg + cap;                                 // [fec]
  uint16_t* omg = syn + fec;                                 // [fec]
  uint32_t* roots = reinterpret_cast<uint32_t*>(omg + fec);   // [cap]  (4 cap + 2 fec halfwords before it: aligned)
  uint32_t* red = roots + cap;                               // [8] reductions, [8..] flags / scan
  uint32_t* scan = red + 16;                                 // [kThreads + 1]
  const Gf gf{gexp, glog};
  const uint16_t* r = sym + (size_t)cw * n_total;
  uint16_t* o = out + (size_t)cw * n_out;
  const int pad = kN - n_total;
  auto finish = [&](bool ok) {                               // uniform over the workgroup
    if (!ok) for (int j = tid; j < n_out; j += kThreads) o[j] = (uint16_t)fail_sym;
    if (tid == 0) ok_out[cw] = ok ? 1 : 0;
  };

  // ---- syndromes (:227-240) ----
  uint32_t any = 0;
  for (int i = tid; i < fec; i += kThreads) {
    uint32_t acc = 0;
    uint32_t t = (uint32_t)(((uint64_t)i * (uint64_t)(n_total - 1)) % kN);   // i * exponent of symbol 0
    for (int j = 0; j < n_total; ++j) {
      const uint32_t v = r[j];
      if (v) acc ^= gexp[glog[v] + t];
      t = t >= (uint32_t)i ? t - i : t + kN - i;
    }
    if (pad > 0 && pad_sym) {            // sum over the padding of pad_sym * x^e, e = n_total .. N-1, x = alpha^i
      if (i == 0) { if (pad & 1) acc ^= pad_sym; }
      else {
        const uint32_t num = gf.pw((uint32_t)(((uint64_t)i * pad) % kN)) ^ 1u;
        const uint32_t den = gf.pw(i) ^ 1u;
        acc ^= gf.mul(gf.mul(pad_sym, gf.pw((uint32_t)(((uint64_t)i * n_total) % kN))), gf.div(num, den));
      }
    }
    syn[i] = (uint16_t)acc;
    any |= acc;
  }
  for (int j = tid; j < n_out; j += kThreads) o[j] = r[j];   // the word as received; corrections are XORed in below
  any = block_max(any, red);
  if (S > fec) { finish(false); return; }                    // more erasures than parity symbols: refused before anything else (:66-75)
  if (any == 0) { finish(true); return; }                    // already a codeword (:82-90)

  // ---- erasure locator (:211-225, :242-248): Lambda = prod (1 + alpha^loc x), loc = N-1-(pad+p) = n_total-1-p ----
  if (tid == 0) lam[0] = 1;
  int len = 1;
  __syncthreads();
  for (int e = 0; e < S; ++e) {
    const uint32_t alog = (uint32_t)(n_total - 1 - er[e]);
    for (int k = tid; k <= len; k += kThreads) {
      const uint32_t a = k < len ? lam[k] : 0u, b = k >= 1 ? lam[k - 1] : 0u;
      tmp[k] = (uint16_t)(a ^ (b ? gexp[glog[b] + alog] : 0u));
    }
    __syncthreads();
    { uint16_t* sw = lam; lam = tmp; tmp = sw; }
    len += 1;                                                // the leading coefficient is a product of non-zero elements
  }

  // ---- modified Berlekamp-Massey (:296-337) ----
  int psh = 1, plen = len;                                   // previous_lambda = prv * x^psh
  for (int k = tid; k < len; k += kThreads) prv[k] = lam[k];
  __syncthreads();
  if (S < fec) {
    int bi = -1, l = S;
    for (int rnd = S; rnd < fec; ++rnd) {
      const int ub = min(l, len - 1);                        // compute_discrepancy (:275-294)
      uint32_t part = 0;
      for (int k = tid; k <= ub; k += kThreads) part ^= gf.mul(lam[k], syn[rnd - k]);
      const uint32_t d = block_xor(part, red);
      if (d != 0) {
        const int n = max(len, plen + psh);
        uint32_t top = 0;
        for (int k = tid; k < n; k += kThreads) {            // tau = lambda - d * previous_lambda, then simplify
          const uint32_t a = k < len ? lam[k] : 0u;
          const uint32_t b = (k >= psh && k - psh < plen) ? prv[k - psh] : 0u;
          const uint32_t v = a ^ gf.mul(b, d);
          tmp[k] = (uint16_t)v;
          if (v) top = (uint32_t)k + 1;
        }
        const int nlen = (int)block_max(top, red);           // (barriers inside: tmp is complete, lam / prv fully read)
        if (l < rnd - bi) {
          const int t2 = rnd - bi;
          bi = rnd - l;
          l = t2;
          for (int k = tid; k < len; k += kThreads) prv[k] = (uint16_t)gf.div(lam[k], d);   // lambda / discrepancy (:327)
          plen = len; psh = 0;
        }
        __syncthreads();
        { uint16_t* sw = lam; lam = tmp; tmp = sw; }
        len = nlen;
      }
      psh += 1;                                              // previous_lambda <<= 1
      if (plen + psh > cap - 2) { finish(false); return; }   // cannot happen for fec <= kMaxFec (guard only)
    }
  }
  const int deg = len - 1;

  // ---- Chien search (:250-273): roots alpha^i, i = 1..N in increasing order (at most deg of them exist) ----
  for (int k = tid; k < len; k += kThreads) llg[k] = lam[k] ? glog[lam[k]] : (uint16_t)0xFFFF;
  __syncthreads();
  uint32_t found[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t cnt = 0;
  for (int q = 0; q < 256; q += 4) {
    uint32_t iv[4], tv[4], acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { iv[u] = (uint32_t)tid * 256u + 1u + q + u; tv[u] = 0; acc[u] = 0; }
    for (int k = 0; k < len; ++k) {
      const uint32_t lgk = llg[k];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (lgk != 0xFFFFu) acc[u] ^= gexp[lgk + tv[u]];
        tv[u] += iv[u] % kN;
        tv[u] = tv[u] >= (uint32_t)kN ? tv[u] - kN : tv[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (iv[u] <= (uint32_t)kN && acc[u] == 0) { found[(q + u) >> 5] |= 1u << ((q + u) & 31); ++cnt; }
  }
  scan[tid + 1] = cnt;
  if (tid == 0) scan[0] = 0;
  __syncthreads();
  if (tid == 0) for (int i = 1; i <= kThreads; ++i) scan[i] += scan[i - 1];
  __syncthreads();
  const int L = (int)scan[kThreads];
  if (L > cap) { finish(false); return; }                    // (a polynomial of degree deg has at most deg roots)
  {
    uint32_t at = scan[tid];
    for (int w = 0; w < 8; ++w) {
      uint32_t bits = found[w];
      while (bits) {
        const int b = __builtin_ctz(bits);
        bits &= bits - 1;
        roots[at++] = (uint32_t)tid * 256u + 1u + 32u * w + b;
      }
    }
  }
  __syncthreads();
  if (L == 0) { finish(false); return; }                                         // :105-121
  if ((uint64_t)(2ull * (uint64_t)L - (uint64_t)S) > (uint64_t)fec) { finish(false); return; }   // size_t arithmetic (:122-146)

  // ---- Forney (:339-385): omega = first fec coefficients of Lambda * S; derivative = odd coefficients of Lambda ----
  for (int k = tid; k < fec; k += kThreads) {
    uint32_t acc = 0;
    const int amax = min(k, len - 1);
    for (int a = 0; a <= amax; ++a) acc ^= gf.mul(lam[a], syn[k - a]);
    omg[k] = acc ? glog[acc] : (uint16_t)0xFFFF;                                 // kept as logarithms for the evaluation
  }
  __syncthreads();
  uint32_t bad = 0;
  for (int x = tid; x < L; x += kThreads) {
    const uint32_t i = roots[x], im = i % kN;
    uint32_t ov = 0, dv = 0, t = 0;
    for (int k = 0; k < fec; ++k) {                                               // omega(alpha^i)
      const uint32_t lgk = omg[k];
      if (lgk != 0xFFFFu) ov ^= gexp[lgk + t];
      t += im; t = t >= (uint32_t)kN ? t - kN : t;
    }
    t = 0;
    const uint32_t i2 = (2u * im) % kN;
    for (int k = 0; k + 1 < len; k += 2) {                                        // Lambda'(alpha^i): coefficient k = Lambda[k+1], k even
      const uint32_t lgk = llg[k + 1];
      if (lgk != 0xFFFFu) dv ^= gexp[lgk + t];
      t += i2; t = t >= (uint32_t)kN ? t - kN : t;
    }
    const uint32_t num = gf.mul(ov, gf.pw((uint32_t)kN - im));                    // root_exponent_table_[i] = alpha^(N-i) (:193-196)
    if (num != 0) {
      if (dv != 0) {
        const int p = (int)i - 1 - pad;                                          // rsblock[error_location - 1] (:364)
        if (p >= 0 && p < n_out) o[p] ^= (uint16_t)gf.div(num, dv);
      } else bad = 1;                                                             // e_decoder_error3
    }
  }
  bad = block_max(bad, red);
  finish(bad == 0 && deg == L);                                                   // :376-383
}

size_t rs_lds_bytes(int fec) {
  const size_t cap = 2 * (size_t)fec + 8;
  return (4 * cap + 2 * (size_t)fec + 2) * sizeof(uint16_t) + (cap + 16 + kThreads + 1 + 4) * sizeof(uint32_t);
}

// GF(2^16) tables, generated as schifra_galois_field.hpp:317-357 does
void make_tables(std::vector<uint16_t>* ex, std::vector<uint16_t>* lg) {
  ex->assign(2 * (size_t)kN + 2, 0);
  lg->assign(65536, 0);
  uint32_t x = 1;
  for (int i = 0; i < kN; ++i) {
    (*ex)[i] = (uint16_t)x;
    (*lg)[x] = (uint16_t)i;
    x <<= 1;
    if (x & 0x10000u) x ^= 0x1100Bu;
  }
  for (int i = kN; i < 2 * kN + 2; ++i) (*ex)[i] = (*ex)[i - kN];
}

thread_local std::string g_rs_error;

int rs_run(int device, const uint16_t* symbols, int n_cw, int n_total, int fec, const int32_t* erasures, int S,
           uint16_t pad_sym, uint16_t fail_sym, uint16_t* out, int n_out, int32_t* ok) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return LVA_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return LVA_ERR_NO_DEVICE;
  std::vector<uint16_t> ex, lg;
  make_tables(&ex, &lg);
  const size_t sym_b = (size_t)n_cw * n_total * 2, out_b = (size_t)n_cw * n_out * 2, ex_b = ex.size() * 2, lg_b = lg.size() * 2;
  const size_t er_b = (size_t)std::max(S, 1) * 4, ok_b = (size_t)n_cw * 4;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  char* base = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&base), up(sym_b) + up(out_b) + up(ex_b) + up(lg_b) + up(er_b) + up(ok_b)) != hipSuccess)
    return LVA_ERR_NOMEM;
  char* p = base;
  uint16_t* d_sym = reinterpret_cast<uint16_t*>(p); p += up(sym_b);
  uint16_t* d_out = reinterpret_cast<uint16_t*>(p); p += up(out_b);
  uint16_t* d_ex = reinterpret_cast<uint16_t*>(p); p += up(ex_b);
  uint16_t* d_lg = reinterpret_cast<uint16_t*>(p); p += up(lg_b);
  int* d_er = reinterpret_cast<int*>(p); p += up(er_b);
  int* d_ok = reinterpret_cast<int*>(p);
  hipError_t e = hipMemcpy(d_sym, symbols, sym_b, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ex, ex.data(), ex_b, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_lg, lg.data(), lg_b, hipMemcpyHostToDevice);
  if (e == hipSuccess && S > 0) e = hipMemcpy(d_er, erasures, (size_t)S * 4, hipMemcpyHostToDevice);
  const size_t lds = rs_lds_bytes(fec);
  if (e == hipSuccess && lds > 48 * 1024)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(rs_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rs_decode_kernel, dim3(n_cw), dim3(kThreads), lds, nullptr, d_sym, n_total, fec, d_er, S, (uint32_t)pad_sym,
                       (uint32_t)fail_sym, d_ex, d_lg, d_out, n_out, d_ok);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, d_out, out_b, hipMemcpyDeviceToHost);
  if (e == hipSuccess && ok) e = hipMemcpy(ok, d_ok, ok_b, hipMemcpyDeviceToHost);
  (void)hipFree(base);
  if (e != hipSuccess) { g_rs_error = hipGetErrorString(e); return LVA_ERR_HIP; }
  return LVA_OK;
}

int check_common(int n_cw, int n_total, int fec) {
  if (n_cw < 0 || fec < 1 || fec > kMaxFec || n_total <= fec || n_total > kN) return LVA_ERR_ARG;
  return LVA_OK;
}

}  // namespace

extern "C" {

const char* lva_rs_last_error(void) { return g_rs_error.c_str(); }

int lva_rs_decode(int32_t device, const uint16_t* symbols, int32_t n_codewords, int32_t n_total, int32_t redundancy,
                  const int32_t* erasures, int32_t n_erasures, uint16_t pad_symbol, uint16_t fail_symbol, uint16_t* out,
                  int32_t* ok) {
  const int st = check_common(n_codewords, n_total, redundancy);
  if (st != LVA_OK) return st;
  if (n_erasures < 0 || (n_codewords > 0 && (!symbols || !out)) || (n_erasures > 0 && !erasures)) return LVA_ERR_ARG;
  std::vector<uint8_t> seen((size_t)n_total, 0);             // "erasure positions must be unique and inside the block" (:213-218)
  for (int i = 0; i < n_erasures; ++i) {
    if (erasures[i] < 0 || erasures[i] >= n_total || seen[(size_t)erasures[i]]) return LVA_ERR_ARG;
    seen[(size_t)erasures[i]] = 1;
  }
  if (n_codewords == 0) return LVA_OK;
  return rs_run(device, symbols, n_codewords, n_total, redundancy, erasures, n_erasures, pad_symbol, fail_symbol, out,
                n_total - redundancy, ok);
}

int lva_rs_encode(int32_t device, const uint16_t* data, int32_t n_codewords, int32_t n_data, int32_t redundancy,
                  uint16_t pad_symbol, uint16_t* out) {
  if (n_data < 1) return LVA_ERR_ARG;
  const int n_total = n_data + redundancy;
  const int st = check_common(n_codewords, n_total, redundancy);
  if (st != LVA_OK) return st;
  if (n_codewords > 0 && (!data || !out)) return LVA_ERR_ARG;
  if (n_codewords == 0) return LVA_OK;
  // systematic encoding = filling in the parity symbols as erasures: the unique codeword with these data symbols
  std::vector<uint16_t> rx((size_t)n_codewords * n_total, 0);
  for (int c = 0; c < n_codewords; ++c) std::memcpy(&rx[(size_t)c * n_total], data + (size_t)c * n_data, (size_t)n_data * 2);
  std::vector<int32_t> er((size_t)redundancy), ok((size_t)n_codewords, 0);
  for (int i = 0; i < redundancy; ++i) er[(size_t)i] = n_data + i;
  const int rc = rs_run(device, rx.data(), n_codewords, n_total, redundancy, er.data(), redundancy, pad_symbol, 0, out, n_total, ok.data());
  if (rc != LVA_OK) return rc;
  for (int c = 0; c < n_codewords; ++c)
    if (!ok[(size_t)c]) { g_rs_error = "encode: the erasure decode of the parity symbols failed"; return LVA_ERR_HIP; }
  return LVA_OK;
}

}  // extern "C"
