// RFM head and feature-consistency losses (stage 3, revise_net.py / revise_pseudo_labels.py): affinity
// attention (batched q^T k + column softmax), CAM normalisation, RFM application, and the cls / rfm / ecr
// loss reductions with their gradients.  All f32 arithmetic; these are <0.2 % of the step's FLOPs and
// HBM/L2-bound, so they are plain wave64 VALU kernels (LDS tiles, shuffles for the reductions).
#include <math.h>

#include "ps_internal.h"

namespace {

__device__ __forceinline__ float ld(const void* p, int dtype, long long i) {
  return ps_ld_dt(p, dtype, i);
}
__device__ __forceinline__ void st(void* p, int dtype, long long i, float v) {
  ps_st_dt(p, dtype, i, v);
}

// ------------------------------------------------------------------------------------------------
// Batched strided GEMM  C[b][m][n] = alpha * sum_k A[b][m*sam + k*sak] * B[b][k*sbk + n*sbn]
// 64x64 tile, 16-deep K steps staged in LDS as f32 (any input dtype), four waves of 32x32 on the exact-f32 MFMA 16x16x4
// (a lane supplies A[row = lane & 15][k = lane >> 4] and B[k][col = lane & 15]; LDS rows of 80 floats: the four k-groups of a
// fragment read hit disjoint banks).
// ------------------------------------------------------------------------------------------------
struct BgemmArgs {
  const void* A; const void* B; void* C;
  int adt, bdt, cdt;
  int batch, M, N, K;
  long long sab, sam, sak, sbb, sbk, sbn, scb, scm, scn;
  float alpha;
};
typedef float bg_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void bgemm_kernel(const BgemmArgs a) {
  __shared__ float As[16][80], Bs[16][80];
  const int b = blockIdx.z, m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, g = lane >> 4;
  const unsigned char* Ab = static_cast<const unsigned char*>(a.A);
  const unsigned char* Bb = static_cast<const unsigned char*>(a.B);
  bg_f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = bg_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < a.K; k0 += 16) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      // pick the index that is contiguous in memory as the fast one
      int kk, mm;
      if (a.sak == 1) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
      const int m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < a.M && k < a.K) ? ld(Ab, a.adt, b * a.sab + m * a.sam + k * a.sak) : 0.f;
      int kb, nn;
      if (a.sbk == 1) { kb = e & 15; nn = e >> 4; } else { nn = e & 63; kb = e >> 6; }
      const int n = n0 + nn, k2 = k0 + kb;
      Bs[kb][nn] = (n < a.N && k2 < a.K) ? ld(Bb, a.bdt, b * a.sbb + k2 * a.sbk + n * a.sbn) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = As[4 * k4 + g][wm * 32 + i * 16 + l16];
        bv[i] = Bs[4 * k4 + g][wn * 32 + i * 16 + l16];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + 4 * g + r, n = n0 + wn * 32 + j * 16 + l16;
        if (m < a.M && n < a.N) st(a.C, a.cdt, b * a.scb + m * a.scm + n * a.scn, a.alpha * acc[i][j][r]);
      }
}

// The same GEMM, same tile, same order of the K sum (bit-identical results), for operands that can be staged in 4-element vectors
// (one of (m, k) resp. (k, n) contiguous, extents / strides / pointers multiples of 4 elements; dtypes at compile time): 32-deep K steps,
// the NEXT step's global vectors are in registers while this step's MFMAs run, two LDS buffers and one barrier per step.  The
// element-wise version above stages through a generic accessor (a dtype switch per load: one load per basic block, NOTES 7.17) with
// two barriers per 16 MFMAs: 213 us per launch of the affinity products at bs = 32, this one see profiles/.
template <int DT>
__device__ __forceinline__ float4 bg_load4(const unsigned char* p, long long i) {  // 4 consecutive elements from element index i
  if constexpr (DT == PS_F32) {
    return *reinterpret_cast<const float4*>(p + i * 4);
  } else {
    const uint2 r = *reinterpret_cast<const uint2*>(p + i * 2);
    if constexpr (DT == PS_BF16)
      return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    else
      return make_float4(ps_f16_to_f32((uint16_t)(r.x & 0xffff)), ps_f16_to_f32((uint16_t)(r.x >> 16)), ps_f16_to_f32((uint16_t)(r.y & 0xffff)),
                         ps_f16_to_f32((uint16_t)(r.y >> 16)));
  }
}
template <int ADT, int BDT>
__global__ __launch_bounds__(256) void bgemm_vec_kernel(const BgemmArgs a) {
  constexpr int KS = 32, LD = 80;
  __shared__ __attribute__((aligned(16))) float As[2][KS][LD], Bs[2][KS][LD];
  const int b = blockIdx.z, m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, g = lane >> 4;
  const unsigned char* Ab = static_cast<const unsigned char*>(a.A);
  const unsigned char* Bb = static_cast<const unsigned char*>(a.B);
  const bool a_kfast = a.sak == 1, b_kfast = a.sbk == 1;
  // this thread's two vectors per operand: (row in the 64-wide tile side, k offset in the step); the 4 elements run along k or along m / n
  int am[2], ak[2], bn[2], bk[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u;
    if (a_kfast) { ak[u] = (e & 7) * 4; am[u] = e >> 3; } else { am[u] = (e & 15) * 4; ak[u] = e >> 4; }
    if (b_kfast) { bk[u] = (e & 7) * 4; bn[u] = e >> 3; } else { bn[u] = (e & 15) * 4; bk[u] = e >> 4; }
  }
  float4 ra[2], rb[2];
  auto load_regs = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + am[u], k = k0 + ak[u];
      ra[u] = (m < a.M && k < a.K) ? bg_load4<ADT>(Ab, b * a.sab + m * a.sam + k * a.sak) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int n = n0 + bn[u], k2 = k0 + bk[u];
      rb[u] = (n < a.N && k2 < a.K) ? bg_load4<BDT>(Bb, b * a.sbb + k2 * a.sbk + n * a.sbn) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (a_kfast) {
        As[buf][ak[u] + 0][am[u]] = ra[u].x; As[buf][ak[u] + 1][am[u]] = ra[u].y; As[buf][ak[u] + 2][am[u]] = ra[u].z; As[buf][ak[u] + 3][am[u]] = ra[u].w;
      } else {
        *reinterpret_cast<float4*>(&As[buf][ak[u]][am[u]]) = ra[u];
      }
      if (b_kfast) {
        Bs[buf][bk[u] + 0][bn[u]] = rb[u].x; Bs[buf][bk[u] + 1][bn[u]] = rb[u].y; Bs[buf][bk[u] + 2][bn[u]] = rb[u].z; Bs[buf][bk[u] + 3][bn[u]] = rb[u].w;
      } else {
        *reinterpret_cast<float4*>(&Bs[buf][bk[u]][bn[u]]) = rb[u];
      }
    }
  };
  bg_f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = bg_f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = (a.K + KS - 1) / KS;
  load_regs(0);
  store_lds(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_regs((ks + 1) * KS);
#pragma unroll
    for (int k4 = 0; k4 < KS / 4; ++k4) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = As[buf][4 * k4 + g][wm * 32 + i * 16 + l16];
        bv[i] = Bs[buf][4 * k4 + g][wn * 32 + i * 16 + l16];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (ks + 1 < nk) store_lds(buf ^ 1);  // (every wave left that buffer before the previous barrier)
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + 4 * g + r, n = n0 + wn * 32 + j * 16 + l16;
        if (m < a.M && n < a.N) st(a.C, a.cdt, b * a.scb + m * a.scm + n * a.scn, a.alpha * acc[i][j][r]);
      }
}

// Both operands 16-bit and K-contiguous (the affinity logits k q^T of the low-precision path, revise_net.py:69-72): the MFMA fragments
// ARE 16 consecutive bytes of a row, so they go from memory straight into registers -- no LDS, no barrier -- and the products run on
// the 16-bit MFMA (16x16x32: products of two bf16 / f16 numbers are exact in f32 and the accumulation is f32, i.e. the arithmetic of
// the f32 kernels above with another summation order) at 16x the rate of the exact-f32 instruction.  64 x 64 tile, four waves of
// 32 x 32, K a multiple of 32; the next K step's fragments are loaded before this step's MFMAs.
template <bool F16>
__global__ __launch_bounds__(256) void bgemm_kk16_kernel(const BgemmArgs a) {
  const int b = blockIdx.z, m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, g = lane >> 4;
  const uint16_t* Ab = static_cast<const uint16_t*>(a.A) + (long long)b * a.sab;
  const uint16_t* Bb = static_cast<const uint16_t*>(a.B) + (long long)b * a.sbb;
  const uint16_t* pa[2];
  const uint16_t* pb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // rows past the end are clamped (their results are not stored)
    pa[i] = Ab + (long long)min(m0 + wm * 32 + i * 16 + l16, a.M - 1) * a.sam + 8 * g;
    pb[i] = Bb + (long long)min(n0 + wn * 32 + i * 16 + l16, a.N - 1) * a.sbn + 8 * g;
  }
  bg_f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = bg_f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 fa[2], fb[2], na[2], nb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    fa[i] = *reinterpret_cast<const uint4*>(pa[i]);
    fb[i] = *reinterpret_cast<const uint4*>(pb[i]);
  }
  for (int k0 = 0; k0 < a.K; k0 += 32) {
    const int kn = k0 + 32 < a.K ? k0 + 32 : k0;  // (the last step re-reads itself)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      na[i] = *reinterpret_cast<const uint4*>(pa[i] + kn);
      nb[i] = *reinterpret_cast<const uint4*>(pb[i] + kn);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i) { fa[i] = na[i]; fb[i] = nb[i]; }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + 4 * g + r, n = n0 + wn * 32 + j * 16 + l16;
        if (m < a.M && n < a.N) st(a.C, a.cdt, b * a.scb + m * a.scm + n * a.scn, a.alpha * acc[i][j][r]);
      }
}

// in-place softmax of rows of length len (one wave per row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long long rows, int len) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* r = x + row * len;
  float mx = -INFINITY;
  for (int i = lane; i < len; i += 64) mx = fmaxf(mx, r[i]);
  mx = ps_wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < len; i += 64) s += expf(r[i] - mx);
  s = ps_wave_sum(s);
  const float inv = 1.f / s;
  for (int i = lane; i < len; i += 64) r[i] = expf(r[i] - mx) * inv;
}
// The same for rows of up to 64 x NV elements: the row is read ONCE, NV loads per lane issued back to back, and lives in registers for
// the three passes (same operations in the same order: bit-identical).  The three-pass form re-reads the row and pays a memory latency
// per element and pass.
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float* __restrict__ x, long long rows, int len) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* r = x + row * len;
  float v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = r[min(lane + 64 * j, len - 1)];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (lane + 64 * j < len) mx = fmaxf(mx, v[j]);
  mx = ps_wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (lane + 64 * j < len) s += expf(v[j] - mx);
  s = ps_wave_sum(s);
  const float inv = 1.f / s;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (lane + 64 * j < len) r[lane + 64 * j] = expf(v[j] - mx) * inv;
}

// Channel loops of the two affinity kernels: the maps' channel count cc = 3 x classes is a template argument for 9 / 12 / 15 (CT > 0: a
// pixel's V row is loaded back to back), any other count up to RFM_MAXCC takes the predicated form (CT == 0: `if (c < cc)` puts every
// load into its own basic block, where it is consumed before the next is issued -- one L1 latency per load, NOTES 7.17).
constexpr int RFM_MAXCC = 24;
template <int CT, class F>
__device__ __forceinline__ void for_cc(int cc, F&& f) {
  if constexpr (CT > 0) {
#pragma unroll
    for (int c = 0; c < CT; ++c) f(c);
  } else {
#pragma unroll
    for (int c = 0; c < RFM_MAXCC; ++c)
      if (c < cc) f(c);
  }
}

// R[n,j,cc] = sum_i P[n][j][i] * V[n,i,cc]     (one wave per (n,j); cc <= 24)
template <int CT>
__global__ __launch_bounds__(256) void rfm_apply_kernel(const float* __restrict__ P, const float* __restrict__ V, float* __restrict__ R,
                                                        long long rows, int np, int cc_rt) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cc = CT ? CT : cc_rt;
  const int lane = threadIdx.x & 63;
  const long long n = row / np;
  const float* p = P + row * np;
  const float* v = V + n * np * cc;
  float acc[CT ? CT : RFM_MAXCC];
  for_cc<CT>(cc, [&](int c) { acc[c] = 0.f; });
  for (int i = lane; i < np; i += 64) {
    const float pv = p[i];
    const float* vi = v + (long long)i * cc;
    for_cc<CT>(cc, [&](int c) { acc[c] = fmaf(pv, vi[c], acc[c]); });
  }
  for_cc<CT>(cc, [&](int c) {
    const float s = ps_wave_sum(acc[c]);
    if (lane == 0) R[row * cc + c] = s;
  });
}

// dS[n][j][i] = P[n][j][i] * (sum_c dR[n,j,c]*V[n,i,c] - sum_c dR[n,j,c]*R[n,j,c])   (in place over P)
template <int CT>
__global__ __launch_bounds__(256) void affinity_softmax_bwd_kernel(float* __restrict__ P, const float* __restrict__ dR,
                                                                   const float* __restrict__ V, const float* __restrict__ R,
                                                                   long long rows, int np, int cc_rt) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cc = CT ? CT : cc_rt;
  const int lane = threadIdx.x & 63;
  const long long n = row / np;
  float g[CT ? CT : RFM_MAXCC];
  float dot = 0.f;
  for_cc<CT>(cc, [&](int c) { g[c] = dR[row * cc + c]; });
  for_cc<CT>(cc, [&](int c) { dot = fmaf(g[c], R[row * cc + c], dot); });
  float* p = P + row * np;
  const float* v = V + n * np * cc;
  for (int i = lane; i < np; i += 64) {
    const float* vi = v + (long long)i * cc;
    float s = 0.f;
    for_cc<CT>(cc, [&](int c) { s = fmaf(g[c], vi[c], s); });
    p[i] = p[i] * (s - dot);
  }
}

// ------------------------------------------------------------------------------------------------
// CAM normalisation.  mode 0: get_norm_cam_d (revise_net.py:29-41); mode 1: max_norm(.)*label with the
// background channel rebuilt as 1 - max fg (revise_pseudo_labels.py:268-272).  One block per sample.
// src: strided (n,c,h,w) view; dst: f32 with strides (dn, dc, dp) over (n, c, pixel).
// ------------------------------------------------------------------------------------------------
struct NormArgs {
  const void* src; int sdt; long long sn, sc, sh, sw;
  float* dst; long long dn, dc, dp;
  const float* label;  // [n, c] or null (mode 1)
  int c, h, w, mode;
};
__global__ __launch_bounds__(256) void norm_cam_kernel(const NormArgs a) {
  __shared__ float smin[8][4], smax[8][4], fmn[8], fmx[8];
  const int n = blockIdx.x, hw = a.h * a.w, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = 0; c < a.c; ++c) {
    float mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < hw; p += 256) {
      const float v = ld(a.src, a.sdt, n * a.sn + c * a.sc + (p / a.w) * a.sh + (p % a.w) * a.sw);
      mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    mn = ps_wave_min(mn); mx = ps_wave_max(mx);
    if (lane == 0) { smin[c][wave] = mn; smax[c][wave] = mx; }
  }
  __syncthreads();
  if (threadIdx.x < a.c) {
    const int c = threadIdx.x;
    fmn[c] = fminf(fminf(smin[c][0], smin[c][1]), fminf(smin[c][2], smin[c][3]));
    fmx[c] = fmaxf(fmaxf(smax[c][0], smax[c][1]), fmaxf(smax[c][2], smax[c][3]));
  }
  __syncthreads();
  for (int p = threadIdx.x; p < hw; p += 256) {
    float v[8];
    float fgmax = -INFINITY;
    for (int c = 0; c < a.c; ++c) {
      const float x = ld(a.src, a.sdt, n * a.sn + c * a.sc + (p / a.w) * a.sh + (p % a.w) * a.sw);
      if (a.mode == 0) v[c] = (x - fmn[c]) / ((fmx[c] + 1e-5f) - fmn[c]);
      else v[c] = (x - fmn[c]) / (fmx[c] - fmn[c] + 1e-5f) * (a.label ? a.label[n * a.c + c] : 1.f);
      if (c >= 1) fgmax = fmaxf(fgmax, v[c]);
    }
    v[0] = 1.f - fgmax;
    if (a.mode == 0)
      for (int c = 1; c < a.c; ++c)
        if (v[c] < fgmax) v[c] = 0.f;
    for (int c = 0; c < a.c; ++c) a.dst[n * a.dn + c * a.dc + p * a.dp] = v[c];
  }
}

// The same for maps of up to 1024 positions (every call site: 28 x 28 CAMs, 32 x 32 pseudo-masks): a thread's pixels (<= 4) and their
// C <= 8 values are loaded ONCE, unconditionally (channel index clamped, dtype at compile time: 32 loads back to back), kept in
// registers for the normalisation pass, and a pixel's offset is computed once.  The general kernel above pays a memory latency and two
// integer divisions per element and pass through its generic accessor: 17.5 us per launch for a few hundred KB, five launches per step.
template <int SDT>
__global__ __launch_bounds__(256) void norm_cam_small_kernel(const NormArgs a) {
  __shared__ float smin[8][4], smax[8][4], fmn[8], fmx[8];
  const int n = blockIdx.x, hw = a.h * a.w, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float x[4][8];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int p = min((int)threadIdx.x + 256 * sl, hw - 1);
    const long long off = n * a.sn + (long long)(p / a.w) * a.sh + (long long)(p % a.w) * a.sw;
#pragma unroll
    for (int c = 0; c < 8; ++c) x[sl][c] = ps_ld_dt(a.src, SDT, off + (long long)min(c, a.c - 1) * a.sc);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
      if ((int)threadIdx.x + 256 * sl < hw) { mn = fminf(mn, x[sl][c]); mx = fmaxf(mx, x[sl][c]); }
    mn = ps_wave_min(mn); mx = ps_wave_max(mx);
    if (lane == 0) { smin[c][wave] = mn; smax[c][wave] = mx; }
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int c = threadIdx.x;
    fmn[c] = fminf(fminf(smin[c][0], smin[c][1]), fminf(smin[c][2], smin[c][3]));
    fmx[c] = fmaxf(fmaxf(smax[c][0], smax[c][1]), fmaxf(smax[c][2], smax[c][3]));
  }
  __syncthreads();
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int p = (int)threadIdx.x + 256 * sl;
    if (p >= hw) continue;
    float v[8];
    float fgmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c >= a.c) continue;
      const float xv = x[sl][c];
      if (a.mode == 0) v[c] = (xv - fmn[c]) / ((fmx[c] + 1e-5f) - fmn[c]);
      else v[c] = (xv - fmn[c]) / (fmx[c] - fmn[c] + 1e-5f) * (a.label ? a.label[n * a.c + c] : 1.f);
      if (c >= 1) fgmax = fmaxf(fgmax, v[c]);
    }
    v[0] = 1.f - fgmax;
#pragma unroll
    for (int c = 1; c < 8; ++c)
      if (a.mode == 0 && c < a.c && v[c] < fgmax) v[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < a.c) a.dst[n * a.dn + c * a.dc + p * a.dp] = v[c];
  }
}

// ------------------------------------------------------------------------------------------------
// Loss reductions.  All produce per-block partial sums (deterministic two-stage reduction).
// ------------------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 1024;
__device__ __forceinline__ void block_partial(float local, float* partials) {
  __shared__ float red[4];
  local = ps_wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void finish_sum_kernel(const float* __restrict__ partials, int nparts, float scale, float* __restrict__ out,
                                                         int accumulate) {
  __shared__ float red[4];
  float v = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) v += partials[i];
  v = ps_wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
    out[0] = accumulate ? out[0] + r : r;
  }
}

// loss_rfm = mean over [n, 1.., h, w] of |a*label - b*label|;  da/db (+)= +-sign * label * gscale
__global__ __launch_bounds__(256) void l1_masked_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ label,
                                                        float* __restrict__ da, float* __restrict__ db, float* __restrict__ partials, int n,
                                                        int c, long long hw, float gscale) {
  const long long total = (long long)n * (c - 1) * hw;
  float local = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i % hw;
    const long long t = i / hw;
    const int ch = (int)(t % (c - 1)) + 1, img = (int)(t / (c - 1));
    const long long off = ((long long)img * c + ch) * hw + pix;
    const float l = label[img * c + ch];
    const float d = a[off] * l - b[off] * l;
    local += fabsf(d);
    if (da) {
      const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      da[off] += s * l * gscale;
      db[off] -= s * l * gscale;
    }
  }
  block_partial(local, partials);
}

// tensor_ecr[n, c, p] = | max_onehot(ref)[n,c,p] - rv[n,c,p]*label[n,c] |   (revise_pseudo_labels.py:125-130,275-276)
__global__ __launch_bounds__(256) void ecr_tensor_kernel(const float* __restrict__ ref, const float* __restrict__ rv, const float* __restrict__ label,
                                                         float* __restrict__ out, int n, int c, long long hw) {
  const long long total = (long long)n * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long img = i / hw, pix = i - img * hw;
    const float* rp = ref + img * c * hw + pix;
    float fgmax = -INFINITY;
    for (int k = 1; k < c; ++k) fgmax = fmaxf(fgmax, rp[k * hw]);
    for (int k = 0; k < c; ++k) {
      float oh = rp[k * hw];
      if (k >= 1 && oh != fgmax) oh = 0.f;
      const long long off = (img * c + k) * hw + pix;
      out[off] = fabsf(oh - rv[off] * label[img * c + k]);
    }
  }
}
// Tie tickets for a top-k selection: of the elements EQUAL to the threshold, take[img] belong to the selection (which ones is free).
// The tie lanes of a wave draw their tickets with ONE returning atomic (ballot + rank), and with none once the image's quota is used
// up: per-lane returning atomics on one address per image serialised these kernels (the maps hold few distinct values -- ReLU zeros,
// bf16-derived numbers -- so ties come by the thousand).  Must be called by every active lane of the wave.
__device__ __forceinline__ bool tie_ticket(bool tie, long long img, int* __restrict__ counter, const int* __restrict__ take) {
  const unsigned long long tmask = __ballot(tie);
  if (!tmask) return false;
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)tmask) - 1;
  const long long img0 = __shfl(img, leader);
  if (__ballot(tie && img != img0) == 0) {
    int base = 0;
    if (lane == leader) {
      base = __atomic_load_n(&counter[img0], __ATOMIC_RELAXED);
      if (base < take[img0]) base = atomicAdd(&counter[img0], __popcll(tmask));
    }
    base = __shfl(base, leader);
    return tie && base + __popcll(tmask & ((1ull << lane) - 1ull)) < take[img];
  }
  return tie && atomicAdd(&counter[img], 1) < take[img];  // the wave straddles two images: one ticket per lane
}

// d rv (+)= -sign(oh - rv*label) * label * gscale on the selected (top-k) elements
__global__ __launch_bounds__(256) void ecr_bwd_kernel(const float* __restrict__ ref, const float* __restrict__ rv, const float* __restrict__ label,
                                                      const float* __restrict__ t, const float* __restrict__ thr, const int* __restrict__ ties_take,
                                                      int* __restrict__ tie_counter, float* __restrict__ drv, int n, int c, long long hw, float gscale) {
  const long long total = (long long)n * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long img = i / hw, pix = i - img * hw;
    const float* rp = ref + img * c * hw + pix;
    float fgmax = -INFINITY;
    for (int k = 1; k < c; ++k) fgmax = fmaxf(fgmax, rp[k * hw]);
    for (int k = 0; k < c; ++k) {
      const long long off = (img * c + k) * hw + pix;
      const float tv = t[off], th = thr[img];
      const bool sel = tv > th || tie_ticket(tv == th, img, tie_counter, ties_take);
      if (!sel) continue;
      float oh = rp[k * hw];
      if (k >= 1 && oh != fgmax) oh = 0.f;
      const float l = label[img * c + k];
      const float d = oh - rv[off] * l;
      const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      drv[off] -= s * l * gscale;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// top-k by radix select: one block per row; keys are order-preserving uint32 images of the floats.
// largest != 0: k largest; else k smallest.  Outputs thr[row] (the k-th value), take[row] = how many
// elements EQUAL to thr belong to the selection, and sum[row] = sum of the selected values (optionally
// of relu(values)), which does not depend on which ties are taken.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fkey(float f, int largest) {
  uint32_t u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order
  return largest ? ~u : u;                          // select the k SMALLEST keys
}
__device__ __forceinline__ float fkey_inv(uint32_t key, int largest) {
  uint32_t u = largest ? ~key : key;
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __uint_as_float(u);
}
__global__ __launch_bounds__(1024) void topk_select_kernel(const float* __restrict__ x, long long row_len, int k, int largest, int relu,
                                                           float* __restrict__ thr, int* __restrict__ take, float* __restrict__ sums) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_remaining;
  __shared__ float red[16];
  const float* r = x + (long long)blockIdx.x * row_len;
  if (threadIdx.x == 0) { s_prefix = 0; s_remaining = (unsigned)k; }
  __syncthreads();
  for (int pass = 3; pass >= 0; --pass) {
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
    for (long long i = threadIdx.x; i < row_len; i += blockDim.x) {
      const uint32_t key = fkey(r[i], largest);
      // one LDS atomic per distinct bin per wave, not per lane: the maps hold few distinct values (ReLU zeros, bf16-derived
      // numbers), so per-lane atomics all landed on one or two bins and serialised
      const bool act = (key & hi_mask) == prefix;
      const unsigned bin = (key >> (8 * pass)) & 255u;
      unsigned long long todo = __ballot(act);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned b0 = __shfl(bin, leader);
        const unsigned long long same = __ballot(act && bin == b0);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b0], (unsigned)__popcll(same));
        todo &= ~same;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned rem = s_remaining, b = 0;
      for (; b < 256; ++b) {
        if (hist[b] >= rem) break;
        rem -= hist[b];
      }
      s_prefix = prefix | (b << (8 * pass));
      s_remaining = rem;
    }
    __syncthreads();
  }
  const uint32_t kth = s_prefix;  // key of the k-th element; s_remaining = how many equal keys are taken
  const float t = fkey_inv(kth, largest);
  float local = 0.f;
  for (long long i = threadIdx.x; i < row_len; i += blockDim.x) {
    const float v = r[i];
    if (fkey(v, largest) < kth) local += relu ? fmaxf(v, 0.f) : v;
  }
  local = ps_wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
    thr[blockIdx.x] = t;
    take[blockIdx.x] = (int)s_remaining;
    sums[blockIdx.x] = s + (relu ? fmaxf(t, 0.f) : t) * (float)s_remaining;
  }
}

// The same selection with each row spread over TOPK_SPLIT-able blocks (the single-block kernel above walks a 50 K - 200 K element row
// five times with 1024 threads on ONE CU: 440 us for 32 rows): one launch per radix pass builds the row's 256-bin histogram in a
// global workspace (LDS histogram per block, then <= 256 atomics), every block re-derives the prefix chosen so far from the finished
// histograms of the earlier passes (a 256-entry scan), a fifth launch sums the selected values per block and a last one adds the
// partial sums in block order (deterministic) and writes thr / take / sums.
struct TopkState { uint32_t prefix, remaining; };
// Block-cooperative (blocks of >= 256 threads; all threads must call it; returns the state to every thread): for each pass already
// histogrammed, threads 0..255 scan the 256 bins in parallel (wave-level inclusive scan + the four wave totals through LDS) and the one
// thread whose bin holds the k-th remaining key publishes bin and remainder.  A single thread walking the bins -- one dependent LDS
// load per bin, up to 256 per pass, up to four passes per launch -- was most of every launch of this family (19-34 us each for
// 5 us of data).  lds: TOPK_LDS uints.
constexpr int TOPK_LDS = 264;
__device__ __forceinline__ TopkState topk_state(const unsigned* __restrict__ ghist /* [4][256] of this row */, int k, int upto_pass, unsigned* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) { lds[256] = 0u; lds[257] = (unsigned)k; }
  for (int pass = 3; pass > upto_pass; --pass) {  // passes already histogrammed
    __syncthreads();
    const unsigned c = tid < 256 ? ghist[pass * 256 + tid] : 0u;
    unsigned incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (tid < 256 && lane == 63) lds[260 + (tid >> 6)] = incl;
    __syncthreads();
    if (tid < 256) {
      for (int w = 0; w < (tid >> 6); ++w) incl += lds[260 + w];
      const unsigned rem = lds[257];
      if (incl >= rem && incl - c < rem) {  // exactly one bin (k <= row length): the first whose running count reaches rem
        lds[258] = (unsigned)tid;
        lds[259] = rem - (incl - c);
      }
    }
    __syncthreads();
    if (tid == 0) {
      lds[256] |= lds[258] << (8 * pass);
      lds[257] = lds[259];
    }
  }
  __syncthreads();
  const TopkState st{lds[256], lds[257]};
  __syncthreads();
  return st;
}
__global__ __launch_bounds__(1024) void topk_hist_kernel(const float* __restrict__ x, long long row_len, long long chunk, int k, int largest, int pass,
                                                         unsigned* __restrict__ ghist) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int st_lds[TOPK_LDS];
  const int row = blockIdx.y;
  const float* r = x + (long long)row * row_len;
  unsigned* gh = ghist + (long long)row * 1024;
  const unsigned prefix = topk_state(gh, k, pass, st_lds).prefix;
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  __syncthreads();
  const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
  const long long lo = (long long)blockIdx.x * chunk, hi = min(row_len, lo + chunk);
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const uint32_t key = fkey(r[i], largest);
    const bool act = (key & hi_mask) == prefix;
    const unsigned bin = (key >> (8 * pass)) & 255u;
    unsigned long long todo = __ballot(act);
    while (todo) {  // one LDS atomic per distinct bin per wave
      const int leader = __ffsll((long long)todo) - 1;
      const unsigned b0 = __shfl(bin, leader);
      const unsigned long long same = __ballot(act && bin == b0);
      if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b0], (unsigned)__popcll(same));
      todo &= ~same;
    }
  }
  __syncthreads();
  if (threadIdx.x < 256 && hist[threadIdx.x]) atomicAdd(&gh[pass * 256 + threadIdx.x], hist[threadIdx.x]);
}
__global__ __launch_bounds__(1024) void topk_sum_kernel(const float* __restrict__ x, long long row_len, long long chunk, int k, int largest, int relu,
                                                        const unsigned* __restrict__ ghist, float* __restrict__ partial) {
  __shared__ unsigned int st_lds[TOPK_LDS];
  __shared__ float red[16];
  const int row = blockIdx.y;
  const float* r = x + (long long)row * row_len;
  const uint32_t kth = topk_state(ghist + (long long)row * 1024, k, -1, st_lds).prefix;
  const long long lo = (long long)blockIdx.x * chunk, hi = min(row_len, lo + chunk);
  float local = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float v = r[i];
    if (fkey(v, largest) < kth) local += relu ? fmaxf(v, 0.f) : v;
  }
  local = ps_wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float sm = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sm += red[w];
    partial[(long long)row * gridDim.x + blockIdx.x] = sm;
  }
}
__global__ __launch_bounds__(256) void topk_finish_kernel(const unsigned* __restrict__ ghist, const float* __restrict__ partial, int nblk, int k,
                                                          int largest, int relu, float* __restrict__ thr, int* __restrict__ take, float* __restrict__ sums) {
  __shared__ unsigned int st_lds[TOPK_LDS];
  const int row = blockIdx.x;  // one block per row
  const TopkState st = topk_state(ghist + (long long)row * 1024, k, -1, st_lds);
  if (threadIdx.x != 0) return;
  const float t = fkey_inv(st.prefix, largest);
  float sm = 0.f;
  for (int b = 0; b < nblk; ++b) sm += partial[(long long)row * nblk + b];  // block order: deterministic
  thr[row] = t;
  take[row] = (int)st.remaining;
  sums[row] = sm + (relu ? fmaxf(t, 0.f) : t) * (float)st.remaining;
}

// ------------------------------------------------------------------------------------------------
// classification pieces of loss_cls (revise_pseudo_labels.py:253-256)
// ------------------------------------------------------------------------------------------------
// per-(n,c) mean over pixels of an NCHW tensor (adaptive_avg_pool2d(cam, 1)): one block per (n,c)
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ x, float* __restrict__ out, long long hw) {
  __shared__ float red[4];
  const float* p = x + (long long)blockIdx.x * hw;
  float s = 0.f;
  // (eight loads in flight per thread, added in index order: same sum as the plain loop, which paid a memory latency per element)
  long long i = threadIdx.x;
  for (; i + 7 * 256 < hw; i += 8 * 256) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[i + j * 256];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; i < hw; i += 256) s += p[i];
  s = ps_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)hw;
}
// multilabel_soft_margin_loss on gap[:,1:] vs label[:,1:] (mean over classes, mean over batch) and its gradient
// w.r.t. gap; single block.  loss_out[0] (+)= loss.  dgap[n,c] = gscale * (sigmoid(x) - y) / ((c-1)*n), dgap[n,0] = 0.
__global__ __launch_bounds__(256) void softmargin_kernel(const float* __restrict__ gap, const float* __restrict__ label, float* __restrict__ dgap,
                                                         float* __restrict__ loss_out, int n, int c, float gscale, int accumulate) {
  __shared__ float red[4];
  float local = 0.f;
  for (int i = threadIdx.x; i < n * c; i += 256) {
    const int ch = i % c;
    float g = 0.f;
    if (ch >= 1) {
      const float x = gap[i], y = label[i];
      // -(y*logsigmoid(x) + (1-y)*logsigmoid(-x)); logsigmoid(x) = min(x,0) - log1p(exp(-|x|))
      const float ls_pos = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
      const float ls_neg = fminf(-x, 0.f) - log1pf(expf(-fabsf(x)));
      local += -(y * ls_pos + (1.f - y) * ls_neg);
      const float sig = 1.f / (1.f + expf(-x));
      g = (sig - y) * gscale / (float)((c - 1) * n);
    }
    if (dgap) dgap[i] = g;
  }
  local = ps_wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = ((red[0] + red[1]) + (red[2] + red[3])) / (float)((c - 1) * n);
    loss_out[0] = accumulate ? loss_out[0] + r : r;
  }
}
// dx[n,c,p] (+)= dgap[n,c] / hw    (gradient of the global average pool, broadcast)
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dgap, float* __restrict__ dx, long long total, long long hw) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) dx[i] += dgap[i / hw] / (float)hw;
}
// m[n,p] = max over fg channels of x[n,c,p]*label[n,c]; arg[n,p] = its channel (first max)
__global__ __launch_bounds__(256) void chmax_kernel(const float* __restrict__ x, const float* __restrict__ label, float* __restrict__ m,
                                                    uint8_t* __restrict__ arg, int n, int c, long long hw) {
  const long long total = (long long)n * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long img = i / hw, pix = i - img * hw;
    float best = 0.f;
    int bi = 1;
    for (int k = 1; k < c; ++k) {
      const float v = x[(img * c + k) * hw + pix] * label[img * c + k];
      if (k == 1 || v > best) { best = v; bi = k; }
    }
    m[i] = best;
    arg[i] = (uint8_t)bi;
  }
}
// adaptive_min_pooling backward: the k smallest channel-max values with value > 0 route gscale*label to their argmax channel
__global__ __launch_bounds__(256) void minpool_bwd_kernel(const float* __restrict__ m, const uint8_t* __restrict__ arg, const float* __restrict__ label,
                                                          const float* __restrict__ thr, const int* __restrict__ take, int* __restrict__ tie_counter,
                                                          float* __restrict__ dx, int n, int c, long long hw, float gscale) {
  const long long total = (long long)n * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long img = i / hw, pix = i - img * hw;
    const float v = m[i], th = thr[img];
    // relu: no gradient -- and no tie ticket for v <= 0: the threshold is usually 0 with tens of thousands of ties, and a returning
    // atomic per tie on one address per image serialised the whole kernel (11 ms at bs = 32).  Ties at a positive threshold are common
    // too (the maps come from bf16 convolutions: few distinct values): see tie_ticket.
    const bool sel = (v > 0.f && v < th) || tie_ticket(v > 0.f && v == th, img, tie_counter, take);
    if (!sel) continue;
    const int k = arg[i];
    dx[(img * c + k) * hw + pix] += gscale * label[img * c + k];
  }
}

// ---- deterministic choice among the elements EQUAL to a top-k threshold: the first take[img] of them in a fixed order ----------
// (tie_ticket above is first come, first served: which tied elements carry the gradient then depends on wave timing.  The reference asks
// for deterministic algorithms in stage 3, revise_pseudo_labels.py:140-146.)  An image's pixels are cut into `nseg` segments of `seg` pixels;
// pass 1 counts the ties per segment, pass 2 gives every tied element the rank (ties in earlier segments) + (ties before it inside the
// segment, in the order the block walks it: 256-pixel chunks, channel by channel, lane order) and selects ranks < take[img].
// MODE 0: adaptive-min-pool map m[n][hw] (tie: v > 0 && v == thr);  MODE 1: ECR tensor t[n][c][hw] (tie: v == thr).
template <int MODE>
__global__ __launch_bounds__(256) void tie_count_kernel(const float* __restrict__ v, const float* __restrict__ thr, int* __restrict__ counts, int c,
                                                        long long hw, int seg, int nseg) {
  const int img = blockIdx.x / nseg, sg = blockIdx.x - img * nseg;
  const long long p0 = (long long)sg * seg, p1 = min(hw, p0 + seg);
  const float th = thr[img];
  int cnt = 0;
  for (long long pix = p0 + threadIdx.x; pix < p1; pix += 256) {
    if constexpr (MODE == 0) {
      const float x = v[img * hw + pix];
      cnt += (x > 0.f && x == th) ? 1 : 0;
    } else {
      for (int k = 0; k < c; ++k) cnt += (v[(img * c + k) * hw + pix] == th) ? 1 : 0;
    }
  }
  __shared__ int red[4];
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// Block-wide exclusive prefix of a per-thread flag (thread order) and the block total; `scratch` = 4 ints of LDS; two barriers.
__device__ __forceinline__ int block_prefix_flag(bool flag, int* scratch, int& total) {
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) scratch[wv] = __popcll(m);
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wv; ++w) before += scratch[w];
  total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  __syncthreads();
  return before + __popcll(m & ((1ull << lane) - 1ull));
}
__global__ __launch_bounds__(256) void minpool_bwd_det_kernel(const float* __restrict__ m, const uint8_t* __restrict__ arg, const float* __restrict__ label,
                                                              const float* __restrict__ thr, const int* __restrict__ take, const int* __restrict__ counts,
                                                              float* __restrict__ dx, int c, long long hw, int seg, int nseg, float gscale) {
  __shared__ int scratch[4];
  const int img = blockIdx.x / nseg, sg = blockIdx.x - img * nseg;
  const long long p0 = (long long)sg * seg, p1 = min(hw, p0 + seg);
  int running = 0;
  for (int s0 = 0; s0 < sg; ++s0) running += counts[img * nseg + s0];
  const float th = thr[img];
  const int quota = take[img];
  for (long long base = p0; base < p1; base += 256) {
    const long long pix = base + threadIdx.x;
    const bool in = pix < p1;
    const float v = in ? m[img * hw + pix] : 0.f;
    const bool tie = in && v > 0.f && v == th;
    int total;
    const int rank = running + block_prefix_flag(tie, scratch, total);
    running += total;
    const bool sel = in && ((v > 0.f && v < th) || (tie && rank < quota));
    if (sel) {
      const int k = arg[img * hw + pix];
      dx[(img * c + k) * hw + pix] += gscale * label[img * c + k];
    }
  }
}
__global__ __launch_bounds__(256) void ecr_bwd_det_kernel(const float* __restrict__ ref, const float* __restrict__ rv, const float* __restrict__ label,
                                                          const float* __restrict__ t, const float* __restrict__ thr, const int* __restrict__ take,
                                                          const int* __restrict__ counts, float* __restrict__ drv, int c, long long hw, int seg, int nseg,
                                                          float gscale) {
  __shared__ int scratch[4];
  const int img = blockIdx.x / nseg, sg = blockIdx.x - img * nseg;
  const long long p0 = (long long)sg * seg, p1 = min(hw, p0 + seg);
  int running = 0;
  for (int s0 = 0; s0 < sg; ++s0) running += counts[img * nseg + s0];
  const float th = thr[img];
  const int quota = take[img];
  for (long long base = p0; base < p1; base += 256) {
    const long long pix = base + threadIdx.x;
    const bool in = pix < p1;
    const float* rp = ref + (long long)img * c * hw + (in ? pix : p0);
    float fgmax = -INFINITY;
    for (int k = 1; k < c; ++k) fgmax = fmaxf(fgmax, rp[k * hw]);
    for (int k = 0; k < c; ++k) {
      const long long off = ((long long)img * c + k) * hw + (in ? pix : p0);
      const float tv = t[off];
      const bool tie = in && tv == th;
      int total;
      const int rank = running + block_prefix_flag(tie, scratch, total);
      running += total;
      const bool sel = in && (tv > th || (tie && rank < quota));
      if (!sel) continue;
      float oh = rp[k * hw];
      if (k >= 1 && oh != fgmax) oh = 0.f;
      const float l = label[img * c + k];
      const float d = oh - rv[off] * l;
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      drv[off] -= sgn * l * gscale;
    }
  }
}
// segments per image: ~2048 blocks in all, at least 1024 pixels per segment
static inline void tie_segments(int n, long long hw, int& seg, int& nseg) {
  long long want = 2048 / (n > 0 ? n : 1);
  if (want < 1) want = 1;
  long long sl = (hw + want - 1) / want;
  if (sl < 1024) sl = 1024;
  sl = (sl + 255) / 256 * 256;
  seg = (int)sl;
  nseg = (int)((hw + sl - 1) / sl);
}

static inline int grid_for(long long items, int per_block, int cap = 2048) {
  long long b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int ps_bgemm(int32_t adt, int32_t bdt, int32_t cdt, const void* A, const void* B, void* C, int32_t batch, int32_t M, int32_t N,
                        int32_t K, int64_t sab, int64_t sam, int64_t sak, int64_t sbb, int64_t sbk, int64_t sbn, int64_t scb, int64_t scm,
                        int64_t scn, float alpha, void* stream) {
  PS_REQUIRE(A && B && C && batch > 0 && M > 0 && N > 0 && K > 0, "bgemm: bad argument");
  PS_REQUIRE(batch <= 65535, "bgemm: batch %d too large", batch);
  BgemmArgs a{A, B, C, adt, bdt, cdt, batch, M, N, K, sab, sam, sak, sbb, sbk, sbn, scb, scm, scn, alpha};
  const dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  // vector staging: K a multiple of 4, and per operand either k contiguous, or the other index contiguous with an extent that is a
  // multiple of 4; every other stride and the base pointer a multiple of 4 elements (a vector is then entirely inside or outside)
  auto vec_ok = [](const void* p, int dt, long long s_slow, long long s_batch) {
    return reinterpret_cast<uintptr_t>(p) % (4 * ps_esize(dt)) == 0 && s_slow % 4 == 0 && s_batch % 4 == 0;
  };
  bool fast = K % 4 == 0 && (sak == 1 || (sam == 1 && M % 4 == 0)) && (sbk == 1 || (sbn == 1 && N % 4 == 0));
  fast = fast && vec_ok(A, adt, sak == 1 ? sam : sak, sab) && vec_ok(B, bdt, sbk == 1 ? sbn : sbk, sbb);
  // 16-bit x 16-bit with k contiguous in both: fragments straight from memory (16-byte aligned rows, K a multiple of 32)
  const bool kk16 = adt == bdt && (adt == PS_BF16 || adt == PS_F16) && sak == 1 && sbk == 1 && K % 32 == 0 && sam % 8 == 0 && sbn % 8 == 0 &&
                    sab % 8 == 0 && sbb % 8 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0;
  if (kk16 && adt == PS_BF16) hipLaunchKernelGGL(bgemm_kk16_kernel<false>, grid, dim3(256), 0, st_, a);
  else if (kk16) hipLaunchKernelGGL(bgemm_kk16_kernel<true>, grid, dim3(256), 0, st_, a);
  else if (fast && adt == PS_F32 && bdt == PS_F32) hipLaunchKernelGGL((bgemm_vec_kernel<PS_F32, PS_F32>), grid, dim3(256), 0, st_, a);
  else if (fast && adt == PS_BF16 && bdt == PS_BF16) hipLaunchKernelGGL((bgemm_vec_kernel<PS_BF16, PS_BF16>), grid, dim3(256), 0, st_, a);
  else if (fast && adt == PS_F32 && bdt == PS_BF16) hipLaunchKernelGGL((bgemm_vec_kernel<PS_F32, PS_BF16>), grid, dim3(256), 0, st_, a);
  else if (fast && adt == PS_F16 && bdt == PS_F16) hipLaunchKernelGGL((bgemm_vec_kernel<PS_F16, PS_F16>), grid, dim3(256), 0, st_, a);
  else if (fast && adt == PS_F32 && bdt == PS_F16) hipLaunchKernelGGL((bgemm_vec_kernel<PS_F32, PS_F16>), grid, dim3(256), 0, st_, a);
  else hipLaunchKernelGGL(bgemm_kernel, grid, dim3(256), 0, st_, a);
  PS_CHECK_LAUNCH("bgemm");
  return PS_OK;
}

extern "C" int ps_softmax_rows(float* x, int64_t rows, int32_t len, void* stream) {
  PS_REQUIRE(x && rows > 0 && len > 0, "softmax_rows: bad argument");
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  if (len <= 64 * 13) hipLaunchKernelGGL(softmax_rows_reg_kernel<13>, grid, dim3(256), 0, st_, x, (long long)rows, len);  // 28 x 28 = 784 positions
  else if (len <= 64 * 16) hipLaunchKernelGGL(softmax_rows_reg_kernel<16>, grid, dim3(256), 0, st_, x, (long long)rows, len);  // 32 x 32
  else hipLaunchKernelGGL(softmax_rows_kernel, grid, dim3(256), 0, st_, x, (long long)rows, len);
  PS_CHECK_LAUNCH("softmax_rows");
  return PS_OK;
}

extern "C" int ps_rfm_apply(const float* P, const float* V, float* R, int32_t n, int32_t np, int32_t cc, void* stream) {
  PS_REQUIRE(P && V && R && n > 0 && np > 0 && cc > 0 && cc <= RFM_MAXCC, "rfm_apply: bad argument (cc <= %d)", RFM_MAXCC);
  const long long rows = (long long)n * np;
  auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), P, V, R, rows, np, cc); };
  switch (cc) {
    case 9: launch(rfm_apply_kernel<9>); break;
    case 12: launch(rfm_apply_kernel<12>); break;
    case 15: launch(rfm_apply_kernel<15>); break;
    default: launch(rfm_apply_kernel<0>); break;
  }
  PS_CHECK_LAUNCH("rfm_apply");
  return PS_OK;
}

extern "C" int ps_affinity_softmax_bwd(float* P_inout, const float* dR, const float* V, const float* R, int32_t n, int32_t np, int32_t cc,
                                       void* stream) {
  PS_REQUIRE(P_inout && dR && V && R && n > 0 && np > 0 && cc > 0 && cc <= RFM_MAXCC, "affinity_softmax_bwd: bad argument");
  const long long rows = (long long)n * np;
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), P_inout, dR, V, R, rows, np, cc);
  };
  switch (cc) {
    case 9: launch(affinity_softmax_bwd_kernel<9>); break;
    case 12: launch(affinity_softmax_bwd_kernel<12>); break;
    case 15: launch(affinity_softmax_bwd_kernel<15>); break;
    default: launch(affinity_softmax_bwd_kernel<0>); break;
  }
  PS_CHECK_LAUNCH("affinity_softmax_bwd");
  return PS_OK;
}

extern "C" int ps_norm_cam(const ps_tensor4* src, float* dst, int64_t dn, int64_t dc, int64_t dp, const float* label, int32_t mode, void* stream) {
  PS_REQUIRE(src && src->ptr && dst, "norm_cam: null argument");
  PS_REQUIRE(src->c >= 2 && src->c <= 8 && (mode == 0 || mode == 1), "norm_cam: C=%d unsupported (2..8) or bad mode", src->c);
  NormArgs a{src->ptr, src->dtype, src->sn, src->sc, src->sh, src->sw, dst, dn, dc, dp, label, src->c, src->h, src->w, mode};
  hipStream_t st_ = static_cast<hipStream_t>(stream);
  if (src->h * src->w <= 1024 && src->dtype == PS_F32) hipLaunchKernelGGL(norm_cam_small_kernel<PS_F32>, dim3(src->n), dim3(256), 0, st_, a);
  else if (src->h * src->w <= 1024 && src->dtype == PS_BF16) hipLaunchKernelGGL(norm_cam_small_kernel<PS_BF16>, dim3(src->n), dim3(256), 0, st_, a);
  else if (src->h * src->w <= 1024 && src->dtype == PS_F16) hipLaunchKernelGGL(norm_cam_small_kernel<PS_F16>, dim3(src->n), dim3(256), 0, st_, a);
  else hipLaunchKernelGGL(norm_cam_kernel, dim3(src->n), dim3(256), 0, st_, a);
  PS_CHECK_LAUNCH("norm_cam");
  return PS_OK;
}

extern "C" int64_t ps_loss_workspace_floats(void) { return LOSS_BLOCKS; }

extern "C" int ps_l1_masked(const float* a, const float* b, const float* label, float* da, float* db, float* loss_out, int32_t accumulate,
                            float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w, float* partials, void* stream) {
  PS_REQUIRE(a && b && label && loss_out && partials && n > 0 && c >= 2 && (!da == !db), "l1_masked: bad argument");
  const long long hw = (long long)h * w, total = (long long)n * (c - 1) * hw;
  const int grid = grid_for(total, 256, LOSS_BLOCKS);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(l1_masked_kernel, dim3(grid), dim3(256), 0, s, a, b, label, da, db, partials, n, c, hw, grad_scale / (float)total);
  PS_CHECK_LAUNCH("l1_masked");
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, partials, grid, 1.f / (float)total, loss_out, accumulate);
  PS_CHECK_LAUNCH("l1_masked_finish");
  return PS_OK;
}

extern "C" int ps_ecr_tensor(const float* ref, const float* rv, const float* label, float* out, int32_t n, int32_t c, int32_t h, int32_t w,
                             void* stream) {
  PS_REQUIRE(ref && rv && label && out && n > 0 && c >= 2, "ecr_tensor: bad argument");
  const long long hw = (long long)h * w;
  hipLaunchKernelGGL(ecr_tensor_kernel, dim3(grid_for((long long)n * hw, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), ref, rv, label, out,
                     n, c, hw);
  PS_CHECK_LAUNCH("ecr_tensor");
  return PS_OK;
}

extern "C" int ps_ecr_bwd(const float* ref, const float* rv, const float* label, const float* t, const float* thr, const int32_t* take,
                          int32_t* tie_counter, float* drv, float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
  PS_REQUIRE(ref && rv && label && t && thr && take && tie_counter && drv && n > 0, "ecr_bwd: bad argument");
  const long long hw = (long long)h * w;
  hipLaunchKernelGGL(ecr_bwd_kernel, dim3(grid_for((long long)n * hw, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), ref, rv, label, t, thr,
                     take, tie_counter, drv, n, c, hw, grad_scale);
  PS_CHECK_LAUNCH("ecr_bwd");
  return PS_OK;
}

extern "C" int64_t ps_tie_workspace_ints(int32_t n, int32_t h, int32_t w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  int seg, nseg;
  tie_segments(n, (long long)h * w, seg, nseg);
  return (int64_t)n * nseg;
}

extern "C" int ps_ecr_bwd_det(const float* ref, const float* rv, const float* label, const float* t, const float* thr, const int32_t* take,
                              int32_t* seg_counts, int64_t seg_counts_ints, float* drv, float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w,
                              void* stream) {
  PS_REQUIRE(ref && rv && label && t && thr && take && seg_counts && drv && n > 0 && c > 0, "ecr_bwd_det: bad argument");
  const long long hw = (long long)h * w;
  int seg, nseg;
  tie_segments(n, hw, seg, nseg);
  PS_REQUIRE(seg_counts_ints >= (int64_t)n * nseg, "ecr_bwd_det: workspace of %lld ints, need %lld (ps_tie_workspace_ints)", (long long)seg_counts_ints,
             (long long)n * nseg);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(tie_count_kernel<1>, dim3(n * nseg), dim3(256), 0, s, t, thr, seg_counts, c, hw, seg, nseg);
  hipLaunchKernelGGL(ecr_bwd_det_kernel, dim3(n * nseg), dim3(256), 0, s, ref, rv, label, t, thr, take, seg_counts, drv, c, hw, seg, nseg, grad_scale);
  PS_CHECK_LAUNCH("ecr_bwd_det");
  return PS_OK;
}

extern "C" int ps_topk_select(const float* x, int32_t rows, int64_t row_len, int32_t k, int32_t largest, int32_t relu, float* thr, int32_t* take,
                              float* sums, void* stream) {
  PS_REQUIRE(x && thr && take && sums && rows > 0 && row_len > 0 && k >= 1 && k <= row_len, "topk_select: bad argument (k=%d, len=%lld)", k,
             (long long)row_len);
  hipLaunchKernelGGL(topk_select_kernel, dim3(rows), dim3(1024), 0, static_cast<hipStream_t>(stream), x, (long long)row_len, k, largest, relu, thr,
                     take, sums);
  PS_CHECK_LAUNCH("topk_select");
  return PS_OK;
}

static int topk_blocks_per_row(int rows, long long row_len) {
  long long b = (2LL * ps_num_cus() + rows - 1) / rows;         // ~2 blocks per CU in total
  const long long maxb = (row_len + 4095) / 4096;               // at least 4 elements per thread
  if (b > maxb) b = maxb;
  if (b > 64) b = 64;
  return b < 1 ? 1 : (int)b;
}
extern "C" int64_t ps_topk_select_workspace_bytes(int32_t rows) { return rows <= 0 ? 0 : (int64_t)rows * (1024 + 64) * 4; }

extern "C" int ps_topk_select_ws(const float* x, int32_t rows, int64_t row_len, int32_t k, int32_t largest, int32_t relu, float* thr, int32_t* take,
                                 float* sums, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!workspace) return ps_topk_select(x, rows, row_len, k, largest, relu, thr, take, sums, stream);
  PS_REQUIRE(x && thr && take && sums && rows > 0 && row_len > 0 && k >= 1 && k <= row_len, "topk_select: bad argument (k=%d, len=%lld)", k,
             (long long)row_len);
  PS_REQUIRE(workspace_bytes >= ps_topk_select_workspace_bytes(rows) && ps_aligned16(workspace), "topk_select: workspace of %lld bytes, need %lld",
             (long long)workspace_bytes, (long long)ps_topk_select_workspace_bytes(rows));
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned* ghist = static_cast<unsigned*>(workspace);
  float* partial = reinterpret_cast<float*>(ghist + (long long)rows * 1024);
  const int nblk = topk_blocks_per_row(rows, row_len);
  const long long chunk = (row_len + nblk - 1) / nblk;
  if (hipMemsetAsync(ghist, 0, (size_t)rows * 1024 * 4, s) != hipSuccess) {
    ps_set_error("topk_select: hipMemsetAsync failed");
    return PS_ERR_LAUNCH;
  }
  const dim3 grid(nblk, rows);
  for (int pass = 3; pass >= 0; --pass) {
    hipLaunchKernelGGL(topk_hist_kernel, grid, dim3(1024), 0, s, x, (long long)row_len, chunk, k, largest, pass, ghist);
    PS_CHECK_LAUNCH("topk_hist");
  }
  hipLaunchKernelGGL(topk_sum_kernel, grid, dim3(1024), 0, s, x, (long long)row_len, chunk, k, largest, relu, ghist, partial);
  PS_CHECK_LAUNCH("topk_sum");
  hipLaunchKernelGGL(topk_finish_kernel, dim3(rows), dim3(256), 0, s, ghist, partial, nblk, k, largest, relu, thr, take, sums);
  PS_CHECK_LAUNCH("topk_finish");
  return PS_OK;
}

extern "C" int ps_sum_scaled(const float* x, int32_t n, float scale, float* out, int32_t accumulate, void* stream) {
  PS_REQUIRE(x && out && n > 0, "sum_scaled: bad argument");
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, scale, out, accumulate);
  PS_CHECK_LAUNCH("sum_scaled");
  return PS_OK;
}

extern "C" int ps_gap(const float* x, float* out, int32_t nc, int64_t hw, void* stream) {
  PS_REQUIRE(x && out && nc > 0 && hw > 0, "gap: bad argument");
  hipLaunchKernelGGL(gap_kernel, dim3(nc), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, (long long)hw);
  PS_CHECK_LAUNCH("gap");
  return PS_OK;
}

extern "C" int ps_softmargin(const float* gap, const float* label, float* dgap, float* loss_out, int32_t accumulate, float grad_scale, int32_t n,
                             int32_t c, void* stream) {
  PS_REQUIRE(gap && label && loss_out && n > 0 && c >= 2, "softmargin: bad argument");
  hipLaunchKernelGGL(softmargin_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), gap, label, dgap, loss_out, n, c, grad_scale,
                     accumulate);
  PS_CHECK_LAUNCH("softmargin");
  return PS_OK;
}

extern "C" int ps_gap_bwd(const float* dgap, float* dx, int32_t nc, int64_t hw, void* stream) {
  PS_REQUIRE(dgap && dx && nc > 0 && hw > 0, "gap_bwd: bad argument");
  const long long total = (long long)nc * hw;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dgap, dx, total, (long long)hw);
  PS_CHECK_LAUNCH("gap_bwd");
  return PS_OK;
}

extern "C" int ps_chmax(const float* x, const float* label, float* m, uint8_t* arg, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
  PS_REQUIRE(x && label && m && arg && n > 0 && c >= 2, "chmax: bad argument");
  const long long hw = (long long)h * w;
  hipLaunchKernelGGL(chmax_kernel, dim3(grid_for((long long)n * hw, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, label, m, arg, n, c, hw);
  PS_CHECK_LAUNCH("chmax");
  return PS_OK;
}

extern "C" int ps_minpool_bwd_det(const float* m, const uint8_t* arg, const float* label, const float* thr, const int32_t* take, int32_t* seg_counts,
                                  int64_t seg_counts_ints, float* dx, float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
  PS_REQUIRE(m && arg && label && thr && take && seg_counts && dx && n > 0 && c > 0, "minpool_bwd_det: bad argument");
  const long long hw = (long long)h * w;
  int seg, nseg;
  tie_segments(n, hw, seg, nseg);
  PS_REQUIRE(seg_counts_ints >= (int64_t)n * nseg, "minpool_bwd_det: workspace of %lld ints, need %lld (ps_tie_workspace_ints)",
             (long long)seg_counts_ints, (long long)n * nseg);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(tie_count_kernel<0>, dim3(n * nseg), dim3(256), 0, s, m, thr, seg_counts, c, hw, seg, nseg);
  hipLaunchKernelGGL(minpool_bwd_det_kernel, dim3(n * nseg), dim3(256), 0, s, m, arg, label, thr, take, seg_counts, dx, c, hw, seg, nseg, grad_scale);
  PS_CHECK_LAUNCH("minpool_bwd_det");
  return PS_OK;
}

extern "C" int ps_minpool_bwd(const float* m, const uint8_t* arg, const float* label, const float* thr, const int32_t* take, int32_t* tie_counter,
                              float* dx, float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
  PS_REQUIRE(m && arg && label && thr && take && tie_counter && dx && n > 0, "minpool_bwd: bad argument");
  const long long hw = (long long)h * w;
  hipLaunchKernelGGL(minpool_bwd_kernel, dim3(grid_for((long long)n * hw, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), m, arg, label, thr,
                     take, tie_counter, dx, n, c, hw, grad_scale);
  PS_CHECK_LAUNCH("minpool_bwd");
  return PS_OK;
}
