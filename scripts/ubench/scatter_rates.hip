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
  return 0;
}
