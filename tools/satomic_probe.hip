// Does gfx950 execute scalar memory atomics (s_atomic_add, tracked by lgkmcnt instead of vmcnt)?  Every wave of a 1024-block grid takes one ticket.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(int* ctr, int* out) {
  int v = 1;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = v;
}
int main() {
  int *ctr, *out; const int blocks = 1024, waves = 4;
  hipMalloc(&ctr, 4); hipMalloc(&out, blocks * waves * 4); hipMemset(ctr, 0, 4);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 0, 0, ctr, out);
  std::vector<int> h(blocks * waves); int c;
  if (hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
  hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = c == blocks * waves;
  for (int i = 0; i < (int)h.size(); ++i) ok = ok && h[i] == i;
  printf("counter %d (expect %d), tickets %s\n", c, blocks * waves, ok ? "a permutation of 0..n-1: scalar atomics WORK" : "WRONG");
  return 0;
}
