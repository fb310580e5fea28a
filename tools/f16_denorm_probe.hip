// Does the 16-bit MFMA keep fp16 SUBNORMAL inputs (A / B operands)?  hipcc --offload-arch=gfx950 -O2 tools/f16_denorm_probe.hip -o /tmp/f16_denorm_probe && /tmp/f16_denorm_probe
// A = 2^-20 (an fp16 subnormal: the smallest normal is 2^-14) in every element, B = 1: D[i][j] = 32 * 2^-20 = 3.0518e-05 if subnormals are
// honoured, 0 if they are flushed.  Also B subnormal / A normal, and the bf16 MFMA with a bf16 subnormal (2^-130).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  f16x8 a, b, one;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; one[i] = (_Float16)1.0f; }
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
  f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(one, a, z, 0, 0, 0);
  bf16x8 c, e;
  for (int i = 0; i < 8; ++i) { c[i] = (__bf16)7.3468e-40f; e[i] = (__bf16)1.0f; }
  f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, e, z, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = d1[0]; out[1] = d2[0]; out[2] = d3[0]; out[3] = (float)a[0]; }
}
int main() {
  float* d; float h[4];
  hipMalloc(&d, 16);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("fp16 subnormal A (2^-20) x 1, K = 32: %.6e (kept: 3.051758e-05, flushed: 0)\n", h[0]);
  printf("1 x fp16 subnormal B            : %.6e\n", h[1]);
  printf("bf16 subnormal A (2^-130ish) x 1: %.6e (kept: ~2.35e-38)\n", h[2]);
  printf("the fp16 operand as converted   : %.6e\n", h[3]);
  return 0;
}
