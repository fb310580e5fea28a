// Does an out-of-range `buffer_load ... lds` (LDS-DMA) write ZEROS into LDS, or leave LDS untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out, int nbytes, int soff) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  lds[threadIdx.x] = -7.f; lds[threadIdx.x + 64] = -7.f; lds[threadIdx.x + 128] = -7.f; lds[threadIdx.x + 192] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x7fffff00;        // far out of range
  if (threadIdx.x == 2) voff = nbytes - 8;       // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = lds[threadIdx.x * 4 + i];
}
int main() {
  float *g, *o; float h[256], src[512];
  for (int i = 0; i < 512; ++i) src[i] = 100.f + i;
  hipMalloc(&g, sizeof(src)); hipMalloc(&o, sizeof(h));
  hipMemcpy(g, src, sizeof(src), hipMemcpyHostToDevice);
  for (int soff = 0; soff <= 1024; soff += 1024) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, 1024, soff);   // buffer = first 1024 bytes (256 floats)
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("soffset=%d\n", soff);
    for (int t = 0; t < 8; ++t) printf(" lane %d: %g %g %g %g\n", t, h[t*4], h[t*4+1], h[t*4+2], h[t*4+3]);
  }
  return 0;
}
