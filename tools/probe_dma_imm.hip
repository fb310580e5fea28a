// Does the immediate offset of `buffer_load ... lds` move BOTH the global address and the LDS destination?  (If so, the four 1-KiB pieces
// of a 4-KiB LDS region can share one M0 value: imm = 0, 1024, 2048, 3072 with the lane offsets reduced by the same amounts.)
// hipcc --offload-arch=gfx950 -O3 tools/probe_dma_imm.hip -o tools/bin/probe_dma_imm
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) float lds[1024];  // 4 KiB
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  const int voff = threadIdx.x * 16;  // lane l -> bytes [16 l, 16 l + 16)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 1024, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff + 4096 - 2048, 0, 2048, 0);  // global 4096.. -> lds 2048..
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)0x80000000 - 3072, 0, 3072, 0);  // padding row
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  float *g, *o; static float h[1024], src[4096];
  for (int i = 0; i < 4096; ++i) src[i] = (float)i;
  hipMalloc(&g, sizeof(src)); hipMalloc(&o, sizeof(h));
  hipMemcpy(g, src, sizeof(src), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, (int)sizeof(src));
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  for (int q = 0; q < 4; ++q) printf("lds KiB %d: first %g %g ... last %g   (piece %d)\n", q, h[q * 256], h[q * 256 + 1], h[q * 256 + 255], q);
  printf("expected: 0 1 .. 255 | 256 257 .. 511 (global +1024 with the LDS +1024) | 1024 1025 .. 1279 | 0 0 .. 0\n");
  return 0;
}
