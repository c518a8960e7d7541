// What a wavefront-level SCATTERED load costs on gfx950 (every lane its own 128-byte line), the access shape of the big-list
// kernel's walk and gathers: cycles per wave-instruction per CU as a function of bytes per lane, active lanes, loads in flight
// per wavefront, resident wavefronts and where the lines come from (a footprint that fits L2 / the Infinity Cache / neither).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter_rates scripts/ubench/scatter_rates.hip && /tmp/scatter_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// Every lane walks its own pseudo-random sequence of lines inside `lines` 128-byte lines; INFL independent loads are in flight,
// the next batch's addresses depend on nothing loaded (pure throughput), `dep` = 1 makes them depend on the data (latency chain).
template <int BYTES, int INFL>
__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ buf, uint32_t lines_mask, int iters, int lanes, int dep,
                                                 uint32_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = gid * 2654435761u + 12345u, acc = 0;
  if ((int)lane >= lanes) { out[gid] = 0; return; }
  for (int i = 0; i < iters; ++i) {
    uint32_t v[INFL];
#pragma unroll
    for (int u = 0; u < INFL; ++u) {
      x = x * 1664525u + 1013904223u;
      const uint32_t line = ((x >> 7) + (dep ? acc : 0u)) & lines_mask;
      const uint32_t* p = buf + (size_t)line * 32u + ((x >> 3) & (BYTES == 16 ? 28u : 30u) & ~(BYTES / 4u - 1u));
      if (BYTES == 16) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        v[u] = q.x ^ q.y ^ q.z ^ q.w;
      } else {
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        v[u] = q.x ^ q.y;
      }
    }
#pragma unroll
    for (int u = 0; u < INFL; ++u) acc += v[u];
  }
  out[gid] = acc;
}


// Streaming counterparts (what a request costs when the lanes of an instruction ask for neighbouring bytes): 16 bytes per lane,
// RUN bytes contiguous per group of RUN/16 lanes, the groups' runs at pseudo-random RUN-aligned places of the footprint
// (RUN = 1024: the whole instruction contiguous; 128 / 64: the big-list kernel's output runs).  WRITE: non-temporal stores.
template <int RUN, bool WRITE>
__global__ __launch_bounds__(256) void k_runs(uint32_t* __restrict__ buf, uint32_t runs_mask, int iters, uint32_t* __restrict__ out) {
  constexpr uint32_t LPR = RUN / 16;                       // lanes per run
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    uint32_t x = (wave * 1315423911u) ^ ((uint32_t)i * 2654435761u) ^ ((lane / LPR) * 40503u);
    x = x * 1664525u + 1013904223u;
    const uint32_t run = (x >> 5) & runs_mask;
    uint32_t* p = buf + (size_t)run * (RUN / 4) + (lane % LPR) * 4u;
    if (WRITE) {
      __builtin_nontemporal_store(x, p); __builtin_nontemporal_store(x + 1, p + 1);
      __builtin_nontemporal_store(x + 2, p + 2); __builtin_nontemporal_store(x + 3, p + 3);
    } else {
      const uint4 q = *reinterpret_cast<const uint4*>(p);
      acc += q.x ^ q.y ^ q.z ^ q.w;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int RUN, bool WRITE>
static void time_runs(const char* foot, size_t bytes, uint32_t* buf, uint32_t* out, hipEvent_t e0, hipEvent_t e1) {
  const uint32_t mask = (uint32_t)(bytes / RUN - 1);
  const int nblk = 256 * 3, iters = 600;
  hipLaunchKernelGGL((k_runs<RUN, WRITE>), dim3(nblk), dim3(256), 0, 0, buf, mask, 8, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_runs<RUN, WRITE>), dim3(nblk), dim3(256), 0, 0, buf, mask, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)nblk * 4 * iters, by = instr * 1024.0;
  printf("%-10s %-6s run %-5d | %8.1f ns per wave-instr per CU  %8.3g runs/s  %6.2f TB/s\n", foot, WRITE ? "store" : "load", RUN,
         ms * 1e6 / (instr / 256), instr * (1024 / RUN) / (ms * 1e-3), by / (ms * 1e-3) * 1e-12);
}

int main() {
  const size_t maxbytes = (size_t)4 << 30;
  uint32_t* buf; uint32_t* out;
  if (hipMalloc(&buf, maxbytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, maxbytes);
  hipMalloc(&out, 256 * 16 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("%-10s %-6s %-6s %-5s %-6s %-4s | %12s %14s %12s\n", "footprint", "bytes", "lanes", "infl", "waves", "dep", "ns/wave-instr", "cyc/instr/CU", "lines/s chip");
  struct Foot { const char* name; size_t bytes; } foots[] = {{"2MB", (size_t)2 << 20}, {"128MB", (size_t)128 << 20}, {"4GB", (size_t)4 << 30}};
  for (const Foot& f : foots) {
    const uint32_t mask = (uint32_t)(f.bytes / 128 - 1);
    for (int bytes : {8, 16})
      for (int lanes : {64, 32})
        for (int infl : {1, 8})
          for (int wgs : {3, 6})          // workgroups of four wavefronts per CU: 12 / 24 resident wavefronts
            for (int dep : {0, 1}) {
              if (dep && infl != 1) continue;
              if (lanes == 32 && (bytes == 16 || wgs == 6)) continue;
              const int nblk = 256 * wgs, iters = dep ? 400 : 1200 / infl;
              auto launch = [&](int it) {
                if (bytes == 8 && infl == 1) hipLaunchKernelGGL((k_scatter<8, 1>), dim3(nblk), dim3(256), 0, 0, buf, mask, it, lanes, dep, out);
                if (bytes == 8 && infl == 8) hipLaunchKernelGGL((k_scatter<8, 8>), dim3(nblk), dim3(256), 0, 0, buf, mask, it, lanes, dep, out);
                if (bytes == 16 && infl == 1) hipLaunchKernelGGL((k_scatter<16, 1>), dim3(nblk), dim3(256), 0, 0, buf, mask, it, lanes, dep, out);
                if (bytes == 16 && infl == 8) hipLaunchKernelGGL((k_scatter<16, 8>), dim3(nblk), dim3(256), 0, 0, buf, mask, it, lanes, dep, out);
              };
              launch(8);
              hipDeviceSynchronize();
              hipEventRecord(e0);
              launch(iters);
              hipEventRecord(e1);
              hipEventSynchronize(e1);
              float ms = 0; hipEventElapsedTime(&ms, e0, e1);
              const double winstr_per_cu = (double)wgs * 4 * iters * infl;           // wave-instructions issued per CU
              const double ns = ms * 1e6 / winstr_per_cu;
              const double lines = (double)nblk * 4 * lanes * iters * infl / (ms * 1e-3);
              printf("%-10s %-6d %-6d %-5d %-6d %-4d | %12.1f %14.1f %12.3g\n", f.name, bytes, lanes, infl, wgs * 4, dep, ns, ns * clk_khz * 1e-6, lines);
            }
  }
  printf("\nneighbouring lanes (16 bytes per lane; runs at random RUN-aligned places; 12 wavefronts per CU)\n");
  for (const Foot& f : foots) {
    time_runs<1024, false>(f.name, f.bytes, buf, out, e0, e1);
    time_runs<128, false>(f.name, f.bytes, buf, out, e0, e1);
    time_runs<64, false>(f.name, f.bytes, buf, out, e0, e1);
    time_runs<1024, true>(f.name, f.bytes, buf, out, e0, e1);
    time_runs<128, true>(f.name, f.bytes, buf, out, e0, e1);
    time_runs<64, true>(f.name, f.bytes, buf, out, e0, e1);
  }
  return 0;
}
