// Internal helpers shared by the kernel translation units of libpistoseg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pistoseg_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

void ps_set_error(const char* fmt, ...);
int ps_num_cus(void);  // compute units of the current device (api.cpp)
// Kernel-selection / staging tunables.  In the product library they are compile-time constants (no process-global mutable state
// behind the C-ABI); the testing build (-DPS_DEBUG_HOOKS -> libpistoseg_hip_debug.so, include/pistoseg_hip_debug.h) makes them
// settable so that the parity suite can force every staging variant and the tools can run ablations.
#ifdef PS_DEBUG_HOOKS
#define PS_TUNABLE static int
#else
#define PS_TUNABLE [[maybe_unused]] static constexpr int
#endif
// grid of a persistent kernel over nitems work items (see ps_block_items)
static inline unsigned ps_persistent_grid(long long nitems, int nb, int tpb) {
  if (tpb <= 0) return (unsigned)(nitems < nb ? nitems : nb);
  const long long per_batch = (long long)nb * tpb, full = nitems / per_batch, rem = nitems - full * per_batch;
  return (unsigned)(full * nb + (rem < nb ? rem : nb));
}

// The ablation selectors of the conv kernels (IgemmArgs::ablate / WgradArgs::ablate: timing experiments with WRONG results) exist in the
// debug library only.  In the product build they are the constant 0: every per-K-step test of them in a loader's issue path -- which is
// part of the K-step's critical path, ~0.1 % of the kernel per scalar instruction (NOTES 7.20) -- compiles away.
#ifdef PS_DEBUG_HOOKS
#define PS_ABLATE(x) (x)
#else
#define PS_ABLATE(x) 0
#endif

// Static wave priority of the LOADER waves of the wave-specialised kernels (A/B builds: -DPS_LOADER_PRIO=n; see NOTES 7).
#ifndef PS_LOADER_PRIO
#define PS_LOADER_PRIO 0
#endif
#if PS_LOADER_PRIO > 0
#define PS_LOADER_SETPRIO() __builtin_amdgcn_s_setprio(PS_LOADER_PRIO)
#else
#define PS_LOADER_SETPRIO() ((void)0)
#endif

#define PS_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ps_set_error(__VA_ARGS__);         \
      return PS_ERR_ARG;                 \
    }                                    \
  } while (0)

#define PS_CHECK_LAUNCH(what)                                                        \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      ps_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e__));       \
      return PS_ERR_LAUNCH;                                                          \
    }                                                                                \
  } while (0)

static inline int ps_esize(int dtype) { return dtype == PS_F32 ? 4 : 2; }
static inline bool ps_dtype_ok(int dtype) { return dtype == PS_F32 || dtype == PS_BF16 || dtype == PS_F16; }
// the convolutions also take the split 16-bit formats; ps_planes = stored 16-bit values per logical channel (hi + lo)
static inline bool ps_conv_dtype_ok(int dtype) { return ps_dtype_ok(dtype) || dtype == PS_BF16X3 || dtype == PS_F16X3; }
static inline int ps_planes(int dtype) { return (dtype == PS_BF16X3 || dtype == PS_F16X3) ? 2 : 1; }
static inline bool ps_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float ps_bf16_to_f32(uint16_t b) { return __uint_as_float(static_cast<uint32_t>(b) << 16); }
__device__ __forceinline__ uint16_t ps_f32_to_bf16(float f) {
  __bf16 h = static_cast<__bf16>(f);  // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(uint16_t, h);
}

__device__ __forceinline__ float ps_f16_to_f32(uint16_t b) { return static_cast<float>(__builtin_bit_cast(_Float16, b)); }
__device__ __forceinline__ uint16_t ps_f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f)); }
// generic 16-bit / f32 scalar access by runtime dtype (per-pixel kernels)
__device__ __forceinline__ float ps_ld_dt(const void* p, int dtype, long long i) {
  if (dtype == PS_BF16) return ps_bf16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
  if (dtype == PS_F16) return ps_f16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
  return reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void ps_st_dt(void* p, int dtype, long long i, float v) {
  if (dtype == PS_BF16) reinterpret_cast<uint16_t*>(p)[i] = ps_f32_to_bf16(v);
  else if (dtype == PS_F16) reinterpret_cast<uint16_t*>(p)[i] = ps_f32_to_f16(v);
  else reinterpret_cast<float*>(p)[i] = v;
}

// 8 consecutive channels <-> 8 floats
template <typename T>
__device__ __forceinline__ void ps_load8(const T* p, float* v);
template <>
__device__ __forceinline__ void ps_load8<float>(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void ps_load8<__bf16>(const __bf16* p, float* v) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
template <>
__device__ __forceinline__ void ps_load8<_Float16>(const _Float16* p, float* v) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  const f16x8 h = __builtin_bit_cast(f16x8, a);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(h[i]);
}
template <typename T>
__device__ __forceinline__ void ps_store8(T* p, const float* v);
template <>
__device__ __forceinline__ void ps_store8<_Float16>(_Float16* p, const float* v) {
  f16x8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = static_cast<_Float16>(v[i]);
  *reinterpret_cast<uint4*>(p) = __builtin_bit_cast(uint4, h);
}
template <>
__device__ __forceinline__ void ps_store8<float>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void ps_store8<__bf16>(__bf16* p, const float* v) {
  // one vector conversion = four v_cvt_pk_bf16_f32 of adjacent pairs (written element by element with shifts and ORs, hipcc paired
  // (v0, v2) / (v1, v3) and then re-interleaved the halves with 16 extra instructions)
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
  const f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  *reinterpret_cast<uint4*>(p) = __builtin_bit_cast(uint4, __builtin_convertvector(f, b16x8));
}

// Division by a launch-invariant divisor as multiply-high + shift (the device has no integer divide: a 32-bit `/` is ~40
// instructions, a 64-bit one several hundred).  Exact for dividends < 2^31.
struct FastDiv {
  uint32_t magic, shift, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  if (d > 0x80000000u) d = 0x80000000u;  // callers divide 31-bit quantities only; keeps the shift below defined for any argument
  f.d = d;
  uint32_t l = 0;
  while (l < 31 && (1u << l) < d) ++l;
  f.shift = 31 + l;
  f.magic = static_cast<uint32_t>(((1ull << f.shift) + d - 1) / d);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return static_cast<uint32_t>((static_cast<unsigned long long>(n) * f.magic) >> f.shift);
}

// Eight consecutive elements held as loaded (no conversion): lets a kernel issue loads well ahead of their use.
template <typename T>
struct PsRaw8 {
  uint4 q[sizeof(T) / 2];
  __device__ __forceinline__ void load(const T* p) {
#pragma unroll
    for (int k = 0; k < (int)sizeof(T) / 2; ++k) q[k] = reinterpret_cast<const uint4*>(p)[k];
  }
  __device__ __forceinline__ void unpack(float* v) const {
    if constexpr (sizeof(T) == 4) {
      v[0] = __uint_as_float(q[0].x); v[1] = __uint_as_float(q[0].y); v[2] = __uint_as_float(q[0].z); v[3] = __uint_as_float(q[0].w);
      v[4] = __uint_as_float(q[1].x); v[5] = __uint_as_float(q[1].y); v[6] = __uint_as_float(q[1].z); v[7] = __uint_as_float(q[1].w);
    } else if constexpr (sizeof(T) == 2 && !__is_same(T, _Float16)) {
      v[0] = __uint_as_float(q[0].x << 16); v[1] = __uint_as_float(q[0].x & 0xffff0000u);
      v[2] = __uint_as_float(q[0].y << 16); v[3] = __uint_as_float(q[0].y & 0xffff0000u);
      v[4] = __uint_as_float(q[0].z << 16); v[5] = __uint_as_float(q[0].z & 0xffff0000u);
      v[6] = __uint_as_float(q[0].w << 16); v[7] = __uint_as_float(q[0].w & 0xffff0000u);
    } else {
      const f16x8 h = __builtin_bit_cast(f16x8, q[0]);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(h[i]);
    }
  }
};

__device__ __forceinline__ float ps_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ps_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float ps_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a 1-D block id: blocks that share `id % 8` share an XCD (observed
// round-robin dispatch; speed only, never correctness), so give each XCD a contiguous chunk of tiles.
__device__ __forceinline__ int ps_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ------------------------------------------------------------------------------------------------
// In-kernel tile queue of the persistent kernels (launch option `tile_queue`).  The FIRST item of block b is static
// (ps_xcd_remap(b, G): the first G items, exactly the first round of the static schedule); the R = nitems - G items behind them are
// handed out by tickets.  They are cut into chunks of 32 consecutive items, chunk c belonging to CLASS c % 8 -- the items the static
// schedule gives to the blocks that share `blockIdx % 8` (= an XCD and its L2) -- and every class has its own ticket counter: a
// block draws from its own class while that lasts (same L2 locality as the static schedule, order of the raster kept) and then
// steals from the others.  Nine counters per launch (tickets per class, blocks that have left); the last block to leave zeroes all nine
// (launches that share a counter block are ordered by their stream).
// ONE wave of the block draws, one item ahead and without waiting: ps_q_draw_begin issues the atomic when an item starts, ps_q_resolve turns the returned ticket into an item several K-steps later and the wave publishes it
// through an LDS mailbox in front of a block barrier; every other wave reads the mailbox behind that barrier.  The only blocking
// memory operations are the steals (a class that looked non-empty), i.e. only where the static schedule would have left a CU idle.
unsigned* ps_queue_slot(hipStream_t stream);  // host: this launch's counter block (conv_igemm.hip); nullptr if the pool cannot be allocated
constexpr int PS_Q_CHUNK_SHIFT = 5;
__device__ __forceinline__ int ps_q_count(int R, int x) {  // items of class x among the R queued items
  const int r = (R & 255) - (x << PS_Q_CHUNK_SHIFT);
  return ((R >> 8) << PS_Q_CHUNK_SHIFT) + (r < 0 ? 0 : r > 32 ? 32 : r);
}
__device__ __forceinline__ int ps_q_item(int G, int x, unsigned k) {  // k-th item of class x
  return G + (int)((((k >> PS_Q_CHUNK_SHIFT) << 3) + (unsigned)x) << PS_Q_CHUNK_SHIFT) + (int)(k & 31u);
}
// Counter layout: class x's tickets at ctr[x * PS_Q_STRIDE] -- one 128-byte line per class: 256 blocks draw at nearly the same time, and
// atomics on ONE line are served one after the other (the first version kept all nine counters in 64 bytes and added 1 / 0 to all eight class
// counters per draw: +3-6 % on the 3x3 layers of a training step) -- and the blocks-done count at ctr[8 * PS_Q_STRIDE].
constexpr int PS_Q_STRIDE = 32;
constexpr int PS_Q_SLOT_DWORDS = 9 * PS_Q_STRIDE;
// ONE atomic instruction per draw, one VGPR (`tk`): lane 0 takes a ticket of the own class; when the class is about to run out (`peek`,
// wave-uniform: the previous ticket was within 64 of its end -- two rounds of the class's blocks) lanes 8..15 add 0 to the eight class counters,
// i.e. read them: the look that decides where to steal.  Draws in the body of a launch are therefore single-lane atomics on the class's own
// line.  The address is lane-dependent on purpose: on a wave-uniform address hipcc's atomic optimiser folds the lanes into one atomic plus a
// readfirstlane of its result, i.e. a wait right behind the instruction.  `tk` must not be read before ps_q_resolve (it is in flight).
__device__ __forceinline__ void ps_q_draw_begin(unsigned* ctr, int x, int lane, bool peek, unsigned& tk) {
  // (an opaque zero in the offset: a loop-invariant per-lane ADDRESS would be hoisted out of the item loop -- two registers that the
  // weight-gradient kernel does not have; spilled, their reload's vmcnt(0) waits for the previous item's atomics at every draw)
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  if (lane == 0 || (peek && lane >= 8 && lane < 16))
    tk = __hip_atomic_fetch_add(ctr + (((lane < 8 ? lane + x : lane) & 7) * PS_Q_STRIDE + z), lane == 0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave-uniform: the next item of this block, or -1 when every class is exhausted; `peek` in: whether the draw looked at the other classes,
// out: whether the next draw should
__device__ __forceinline__ int ps_q_resolve(unsigned* ctr, int x, int lane, int G, int nitems, unsigned tk, bool& peek) {
  const int R = nitems - G, nx = ps_q_count(R, x);
  const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
  if ((int)k < nx) {
    peek = (int)k + 64 >= nx;
    return ps_q_item(G, x, k);
  }
  // own class exhausted.  Counters only grow: a class that looked empty stays empty, one that did not is asked (blocking).
  if (!peek) {  // (it ran out earlier than the previous ticket suggested: look now)
    if (lane >= 8 && lane < 16) tk = __hip_atomic_fetch_add(ctr + (lane & 7) * PS_Q_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    peek = true;
  }
  unsigned m = (unsigned)(__builtin_amdgcn_ballot_w64(lane >= 8 && lane < 16 && (int)tk < ps_q_count(R, lane & 7)) >> 8) & 0xffu & ~(1u << x);
  while (m) {
    const unsigned rot = ((m >> x) | (m << (8 - x))) & 0xffu;  // the classes behind x first (its neighbours in the raster)
    const int y = (x + __builtin_ctz(rot)) & 7;
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ctr + y * PS_Q_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    if ((int)t < ps_q_count(R, y)) return ps_q_item(G, y, t);
    m &= ~(1u << y);
  }
  return -1;
}
// a block leaves (all of its draws have returned): the last one re-arms the counters for the next launch that uses them
__device__ __forceinline__ void ps_q_block_done(unsigned* ctr, int lane, int nblocks) {
  if (lane == 0) {
    const unsigned d = __hip_atomic_fetch_add(ctr + 8 * PS_Q_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == (unsigned)nblocks - 1u) {
#pragma unroll
      for (int i = 0; i < 9; ++i) __hip_atomic_store(ctr + i * PS_Q_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// the mailbox: four ints of LDS; entry (seq & 3) holds the block's seq-th item (seq >= 1).  `mb` = its LDS byte address (ps_q_mbox_addr).
// Explicit DS instructions: a generic-pointer access compiles to a FLAT load, whose wait (vmcnt AND lgkmcnt) would drain a loader wave's DMA queue.
__device__ __forceinline__ unsigned ps_q_mbox_addr(unsigned char* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)p;
}
__device__ __forceinline__ int ps_q_mbox_read(unsigned mb, int seq) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(mb + ((unsigned)(seq & 3) << 2)) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void ps_q_mbox_write(unsigned mb, int seq, int item) {  // wave-uniform item: every lane stores the same dword
  asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(mb + ((unsigned)(seq & 3) << 2)), "v"(item) : "memory");
}

// Work split of the persistent kernels.  tpb <= 0: ONE batch -- block b takes items remap(b), remap(b) + nblocks, ... (a block
// lives for the whole launch: best when the GPU is ours alone).  tpb > 0: blocks come in batches of nb (= #CUs); a batch covers
// nb * tpb consecutive items and each of its blocks takes tpb of them, so the hardware dispatcher re-balances every tpb items when
// some CUs are held by another kernel (a communication kernel beside the backward: a block that starts late would otherwise
// serialise its whole static share behind the others, tools/hog_probe.py).
__device__ __forceinline__ void ps_block_items(int bid, int nblocks, int nitems, int nb, int tpb, int& first, int& stride, int& end) {
  if (tpb <= 0) {
    stride = nblocks;
    first = ps_xcd_remap(bid, nblocks);
    end = nitems;
    return;
  }
  const int per_batch = nb * tpb;
  const int batch = bid / nb, lane = bid - batch * nb;
  const int base = batch * per_batch;
  const int left = nitems - base;
  const int in_batch = left < per_batch ? left : per_batch;
  stride = in_batch < nb ? in_batch : nb;
  first = base + ps_xcd_remap(lane, stride);
  end = base + in_batch;
}
