// mickey_amd -- shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes; every wave-width constant below is hard-coded to 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mickey_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// last error text (thread-local on the host side; the ABI is re-entrant, kernels hold no globals)
void mk_set_error(const char* fmt, ...);
#define MK_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      mk_set_error(__VA_ARGS__);         \
      return MK_ERR_INVALID_ARGUMENT;    \
    }                                    \
  } while (0)
#define MK_CHECK_LAUNCH()                                      \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) {                                   \
      mk_set_error("%s: %s", __func__, hipGetErrorString(e__)); \
      return MK_ERR_LAUNCH;                                    \
    }                                                          \
  } while (0)

namespace mk {

// ---- MFMA wrappers: 16-bit operand type selects the instruction ---------------------------------
template <typename T> struct Lp;  // low-precision operand traits
template <> struct Lp<__bf16> {
  using V8 = bf16x8;
  using V4 = bf16x4;
  static __device__ __forceinline__ f32x4 mma16(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mma32(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Lp<_Float16> {
  using V8 = f16x8;
  using V4 = f16x4;
  static __device__ __forceinline__ f32x4 mma16(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mma32(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// ---- wave64 reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// direct global -> LDS copy of 16 B per lane.  LDS destination = wave-uniform base + lane*16
// (the hardware adds the lane offset); the global source address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(MK_GLOBAL_PTR(gsrc), MK_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// XOR swizzle of 16-byte chunks inside 128-byte LDS rows: chunk' = chunk ^ ((row >> 1) & 7).
// Conflict-free for ds_read_b128 fragment reads where a 16-lane group covers 16 distinct rows at one
// logical chunk (MI355X: 64 banks x 4 B, b128 reads serviced in 4 groups of 16 lanes).
__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// XCD-aware bijective remap of a linear block id: blocks that share an L2 get a contiguous id range
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// exact (erf) GELU, nn.GELU's default (reference DINO_modules/layers/mlp.py:23).  erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, i.e. fp32 round-off level) in ~14 instructions: libm's erff costs ~40 and sat un-overlapped in
// the fc1 epilogue (+30 % on that GEMM).
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = 1.0f - p * t * e;
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

}  // namespace mk
