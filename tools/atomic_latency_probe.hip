// Latency of the ticket draws of the tile queue when 256 blocks draw at about the same time: returning device-scope atomics and
// device-scope loads on 1 / 8 / 32 / 256 different 128-byte lines, vector and scalar form.
//   hipcc --offload-arch=gfx950 -O2 tools/atomic_latency_probe.hip -o /tmp/alp && /tmp/alp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
// mode 0: vector atomic add (lane 0), 1: scalar atomic add, 2: vector load sc1 (lane 0), 3: vector atomic add 0 on 8 lines (lanes 8..15) + 1 on own
__global__ void probe(unsigned* ctr, int lines, int mode, int iters, unsigned long long* out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  unsigned* p = ctr + (b % lines) * 32;
  unsigned long long sum = 0, mx = 0;
  for (int it = 0; it < iters; ++it) {
    // lockstep: every block waits for the same wall-clock boundary (s_memrealtime: 100 MHz)
    unsigned long long t;
    do { t = __builtin_amdgcn_s_memrealtime(); } while ((t & 0x3ff) > 8);  // every 10.24 us
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned v = 0;
    if (mode == 0) {
      if (lane == 0) v = __hip_atomic_fetch_add(p + (lane & 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (mode == 1) {
      int sv = 1;
      asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(sv) : "s"(p) : "memory");
      v = sv;
    } else if (mode == 2) {
      if (lane == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0 || (lane >= 8 && lane < 16)) v = __hip_atomic_fetch_add(lane == 0 ? p : ctr + (lane & 7) * 32, lane == 0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(v) : "memory");
    const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
    sum += dt; mx = dt > mx ? dt : mx;
    while ((__builtin_amdgcn_s_memrealtime() & 0x3ff) <= 8) {}
  }
  if (lane == 0) { out[2 * b] = sum; out[2 * b + 1] = mx; }
}
int main() {
  unsigned* ctr; unsigned long long* out;
  const int blocks = 256, iters = 50;
  hipMalloc(&ctr, 256 * 128); hipMalloc(&out, blocks * 16);
  const char* names[] = {"vector atomic add (1 lane)", "scalar atomic add", "vector load sc1 (1 lane)", "vector atomic: 1 on own + 0 on 8 lines"};
  for (int mode = 0; mode < 4; ++mode)
    for (int lines : {1, 8, 32, 256}) {
      hipMemset(ctr, 0, 256 * 128);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, ctr, lines, mode, iters, out);
      std::vector<unsigned long long> h(blocks * 2);
      if (hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
      double mean = 0; unsigned long long mx = 0;
      for (int b = 0; b < blocks; ++b) { mean += (double)h[2 * b] / iters; mx = std::max(mx, h[2 * b + 1]); }
      printf("%-42s %3d line(s): mean %7.0f ns, worst single draw %7.0f ns  (256 blocks drawing together, %d rounds)\n", names[mode], lines, mean / blocks * 10.0, mx * 10.0, iters);
    }
  return 0;
}
