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
typedef __attribute__((ext_vector_type(8))) float f32x8;

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

// Split-operand planes (x * scale = hi + lo in fp16) saturate at fp16's largest finite value; a NaN stays a NaN (both planes), so
// that a non-finite head activation reaches the outputs and the callers' finite checks.  flag: the per-call watcher word of the
// plane-writing entry points (mickey_hip.h: sat_flag; null = nobody watches: no per-element work) gets bit 0 set by any kernel
// that had to clamp or met a NaN -- a head activation beyond the planes' range (|x| > 1023 at the activation scale 64) cannot
// pass silently.  The word is read first: once set, no further atomics (widespread saturation does not serialise on one address).
__device__ __forceinline__ float sat16(float sv, int* flag) {
  const float c = fminf(fmaxf(sv, -65504.f), 65504.f);
  if (flag && !(c == sv)) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
  }
  return sv != sv ? sv : c;
}

namespace gemm { int num_cus(); }   // CUs of the current device (mk_gemm.hip; cached)

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

// MK_F32: the exact-fp32 parity mode -- "low precision" buffers hold fp32 and contractions run on the fp32-input MFMA
// (v_mfma_f32_16x16x4_f32: bitwise an fp32 fma chain, 1/16 of the bf16 rate); only the vector types are needed here
template <> struct Lp<float> {
  using V8 = f32x8;
  using V4 = f32x4;
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
// same with cache-policy bits (gfx940+ encoding: 1 = sc0, 2 = nt, 16 = sc1); AUX must be a compile-time constant
template <int AUX>
__device__ __forceinline__ void glds16_cp(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(MK_GLOBAL_PTR(gsrc), MK_LDS_PTR(lds_wave_base), 16, 0, AUX);
}

// The same copy with the address split as the hardware takes it: wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte
// offset (global_load_lds_dwordx4 v_off, s[base:base+1]).  The builtin always materialises a 64-bit per-lane address (one
// v_lshl_add_u64 + often a v_add per piece: measured ~30 cycles per piece of matrix-pipe idle time when a piece follows
// every MFMA); here a piece costs two SALU moves and the DMA instruction.  M0 (the LDS destination) is written and
// restored inside the statement (hipcc reserves it); the DMA is invisible to hipcc's vmcnt bookkeeping: the caller waits
// with its own s_waitcnt vmcnt(N).  sbase and lds_dst must be provably wave-uniform (kernel arguments, blockIdx
// expressions, readfirstlane results).
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, void* lds_dst) {
  unsigned keep;
  const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds)
               : "memory");
}

// XOR swizzle of 16-byte chunks inside 128-byte LDS rows: chunk' = chunk ^ ((row >> 1) & 7).
// Conflict-free for ds_read_b128 fragment reads where a 16-lane group covers 16 distinct rows at one
// logical chunk (MI355X: 64 banks x 4 B, b128 reads serviced in 4 groups of 16 lanes).
__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// ---- bordered NHWC feature maps (the activations the 3x3 convolutions read) -----------------------------------------
// Pixel (b, y, x) of a [nimg, H, Wd] grid lives at row  ((b (H+1) + y + 1) (Wd+1) + x + 1)  of a buffer whose other rows
// are zero and are never written: grid rows are Wd+1 pixels apart (the gap is the right border of one row AND the left
// border of the next), images H+1 grid rows apart (the gap row is the bottom border of one image and the top border of
// the next).  A 3x3 tap (dy, dx) of ANY pixel is then the row  +dy (Wd+1) + dx  -- a wave-uniform shift, no border test,
// no zero page: an A piece of the implicit GEMM is one SGPR-base LDS-DMA instruction like a dense operand's.
// Buffer rows: (nimg (H+1) + 1)(Wd+1) + 1.  With m = (b H + y) Wd + x the dense row index:
//   row(m) = m + q + (b + 1)(Wd + 1) + 1,   q = m / Wd,  b = q / H.
__host__ __device__ __forceinline__ long long bordered_rows(long long nimg, int H, int Wd) {
  return (nimg * (H + 1) + 1) * (Wd + 1) + 1;
}
// walks the bordered row of dense row m in increasing steps (one pair of integer divisions at init, none per step)
struct BorderedRow {
  int x, y, extra;   // extra = row(m) - m
  __device__ __forceinline__ void init(int m, int H, int Wd) {
    const int q = m / Wd, b = q / H;
    x = m - q * Wd;
    y = q - b * H;
    extra = q + (b + 1) * (Wd + 1) + 1;
  }
  __device__ __forceinline__ void step(int d, int H, int Wd) {
    x += d;
    while (x >= Wd) {
      x -= Wd;
      ++extra;
      if (++y == H) {
        y = 0;
        extra += Wd + 1;
      }
    }
  }
};
__device__ __forceinline__ long long bordered_row(int m, int H, int Wd) {
  BorderedRow w;
  w.init(m, H, Wd);
  return (long long)m + w.extra;
}

// XCD-aware bijective remap of a linear block id: blocks that share an L2 get a contiguous id range
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// exact (erf) GELU, nn.GELU's default (reference DINO_modules/layers/mlp.py:23):  gelu(v) = v * Phi(v)
//   = max(v, 0) - |v| * Phi(-|v|),   Phi(-u) = 2^-Q(u)  for u in [0, 6]  (Phi(-6) = 1e-9: beyond, the term is dropped
// below fp32 resolution of max(v, 0) anyway, so u is clamped).  Q is a degree-6 weighted-minimax fit of -log2 Phi(-u)
// (tools/fit_gelu.py): max |gelu error| = 9.8e-8 over [-8, 8] evaluated in fp32, i.e. round-off level.  One v_exp_f32
// and 9 full-rate ops per element (no reciprocal, no select); written on 4-vectors so the Horner chain can use
// v_pk_fma_f32.  The GELU runs un-overlapped in the fc1 epilogue: libm erff cost +30 % of that GEMM, the previous
// Abramowitz-Stegun 7.1.26 form (rcp + exp + 20 ops) +20 %.
// The Horner chain runs on v_pk_fma_f32 (two elements per issue).  Left to itself hipcc emits one v_fmaak_f32 per element and
// step instead (a 32-bit literal is cheaper to materialise than a register pair, and VOP3P takes no literals): the
// coefficients are therefore pinned into SGPR pairs (one scalar operand per packed FMA is allowed), once per call site.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 sgpr_pair(float c) {
  f32x2 v = {c, c};
  asm("" : "+s"(v));   // not volatile: identical pins of the unrolled call sites merge
  return v;
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
  const f32x2 c6 = sgpr_pair(-3.19620214e-05f), c5 = sgpr_pair(0.000758801579f), c4 = sgpr_pair(-0.00804438837f),
              c3 = sgpr_pair(0.0533519151f), c2 = sgpr_pair(0.45881945f), c1 = sgpr_pair(1.15118468f), c0 = sgpr_pair(0.999994836f);
  f32x4 r;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x2 u = {fminf(fabsf(v[2 * h]), 6.0f), fminf(fabsf(v[2 * h + 1]), 6.0f)};
    f32x2 q = __builtin_elementwise_fma(c6, u, c5);
    q = __builtin_elementwise_fma(q, u, c4);
    q = __builtin_elementwise_fma(q, u, c3);
    q = __builtin_elementwise_fma(q, u, c2);
    q = __builtin_elementwise_fma(q, u, c1);
    q = __builtin_elementwise_fma(q, u, c0);
#pragma unroll
    for (int e = 0; e < 2; ++e) r[2 * h + e] = fmaxf(v[2 * h + e], 0.f) - fabsf(v[2 * h + e]) * __builtin_amdgcn_exp2f(-q[e]);
  }
  return r;
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf4(f32x4{x, x, x, x})[0]; }

}  // namespace mk
