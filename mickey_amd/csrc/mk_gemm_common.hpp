// mickey_amd -- shared pieces of the 16-bit-operand MFMA GEMMs (mk_gemm.hip: dispatch + 128x128 kernel,
// mk_gemm_pp64.hip: 8-wave ping-pong): parameters, LDS-DMA stager, epilogues.
#pragma once
#include "mk_common.hpp"

namespace mk {
namespace gemm {

constexpr int BK = 64;   // K tile of the 16-bit kernels, in elements: LDS rows of 128 B
// element-size generic forms (the fp32 parity kernel shares the stager and the epilogue): 16-byte chunk / 128-byte row
template <typename T> constexpr int EPC = 16 / (int)sizeof(T);
template <typename T> constexpr int KT = 128 / (int)sizeof(T);

enum AMode { A_DENSE = 0, A_CONV3 = 1 };

struct GemmParams {
  // operands
  const void* A;       // dense: [M, lda]; conv: bordered NHWC activation of source 1 (mk_common.hpp: bordered_rows)
  const void* A2;      // conv only: bordered NHWC activation of source 2 (1x1 shortcut), may be null
  const void* W;       // [N, ldw]
  int M, N, K, lda, ldw;
  long long strideA_g, strideA2_g, strideW_g;  // element strides per group (blockIdx.y)
  // conv geometry
  int H, Wd, C1, C2;   // image grid, channels of source 1 / source 2
  int bord_out;        // conv: the output is written in the bordered layout too (it feeds the next conv)
  // SPLIT operands (mk_conv3x3_split, mk_gemm_grouped_split; npass = 3 marks them): x = hi + lo 16-bit planes of activations and
  // weights, the product evaluated as hi.hi + lo.hi + hi.lo from operands staged ONCE per K step of 32: an LDS row of 128 bytes
  // holds [32 hi | 32 lo] elements of one operand row, so K (and ldw) count 2 x the contraction length and W is the
  // interleaved plane layout of mickey_hip.h.  A / A2 point at the LOWER plane of each source, pl1 / pl2 = byte offsets of
  // (hi, lo) from there (one of each pair is 0; the planes of a source lie within 2 GiB of each other: checked by the entry
  // points).  acc_scale undoes the power-of-two scaling of the planes in the epilogue.
  unsigned pl1[2], pl2[2];
  int npass;           // 0 / 1: plain; 3: split operands; 2: split operands whose activation lo plane is identically zero (two products)
  float acc_scale;
  // ... whose output feeds another split conv: written straight as the next conv's operand planes, out_lp = hi, out_lo = lo
  // (fp16, v * plane_scale = hi + lo) instead of fp32 rows + a separate mk_split_planes pass
  void* out_lo;
  float plane_scale;
  int* sat_flag;       // watcher word of the planes' saturation (mk_common.hpp: sat16), or null
  // epilogue
  int epi;
  int act;
  const float* bias;   // [N]
  const float* gamma;  // [N]
  long long strideBias_g;
  float* out_f32;
  void* out_lp;
  int ldc;
  long long strideOut_g;
  const void* resid_lp;  // identity residual, [M, ldc] low precision; conv: bordered like A, group stride strideResid_g
  long long strideResid_g;
  // qkv split
  void* q;
  void* k;
  void* vt;
  int ntok, ntok_pad, heads;
  float qscale;
  // patch embed
  const float* pos;
  int npatch;
  // ---- LayerNorm folded into the GEMMs around it (the encoder's pre-norm blocks) ----
  // producer side (LS_RESIDUAL / PATCH epilogues with xh set): the residual stream lives in HBM as TWO 16-bit planes,
  //   x = hi + lo,  hi = rn16(x),  lo = rn16(x - hi)    (17 / 22 mantissa bits for bf16 / fp16 planes),
  // the same 4 bytes per element as an fp32 stream, but hi IS the raw A operand of the next GEMM: no copy, no LayerNorm
  // pass.  LS_RESIDUAL reads hi + lo, adds gamma * (acc + bias) in fp32 and writes the new hi / lo plus the per-slot
  // (sum, sum of squares) of the new fp32 rows; with out_f32 also set it writes the fp32 rows there instead (last block:
  // the final norm reads fp32).  PATCH writes hi / lo / statistics of the freshly embedded rows.
  void* xh;
  void* xl;               // both [rows, ldxs] 16 bit
  int ldxs;
  float* stats_out;       // [rows][nslot_out][2]
  int nslot_out;          // N / 64
  // consumer side (QKV / STORE epilogues, 16-bit outputs): A = raw rows, W = W.diag(ln_weight) (folded on the host),
  //   out[m][n] = rstd_m * acc[m][n] - rstd_m * mean_m * colsum[n] + bias[n],   bias = b + W.ln_bias (folded on the host)
  const float* ln_stats;  // [M][ln_nslot][2] as written by the producer; null = plain GEMM
  const float* ln_colsum; // [N]: sum_k W'[n][k] of the 16-bit-rounded folded weights
  int ln_nslot;           // K / 64
  float ln_eps;
  // ---- row centring of the split stream.  LayerNorm does not see a constant added to all channels of a row, and nothing
  // but LayerNorm (norm1 / norm2 / the final norm) ever reads the residual stream: the stream may carry any per-row offset.
  // The consumer knows each row's mean (from the statistics) and publishes it (ln_shift_out, written by the tiles of the
  // first column); the NEXT producer subtracts it while it adds the branch output (ln_shift_in): rows stay centred to within
  // one residual update, so the 16-bit hi plane rounds (x - mean) rather than x -- the operand rounding of the folded form
  // is then that of a LayerNorm OUTPUT (2^-9 |x - mean| for bf16), whatever the common-mode level of the row.
  float* ln_shift_out;        // [M], consumer side; null = not published
  const float* ln_shift_in;   // [M], producer side; null = no centring
};

// Row statistics of the folded LayerNorm, ONE summation order in every epilogue (so that a row's statistics -- hence its
// normalisation, hence every bit downstream -- do not depend on which schedule, tile or tile position produced it: the
// batch-invariance tests compare pair i of a 32-pair forward with a 1-pair forward bit for bit).  A 64-column slot is 16
// column quads q = 0..15 (columns 4q..4q+3):  quad sum sequentially, squares by fma;  then the tree
//   (q, q^1), (.., q^2)  ->  Q_j = sum of quads 4j..4j+3;   then (Q0 + Q1) + (Q2 + Q3).
// LDS epilogue: lane c of a 16-lane DPP row holds quad c  -> quad_stats + row16_sum (quad_perm, quad_perm, half mirror, mirror).
// direct epilogue: lane (fr, fg) holds quads fg + 4 ni, ni = 0..3 of its row -> quad_stats per ni, lane ^ 16, lane ^ 32, then in-lane.
__device__ __forceinline__ void quad_stats(const f32x4& x, float& s, float& q) {
  s = x[0];
  s += x[1];
  s += x[2];
  s += x[3];
  q = __builtin_fmaf(x[0], x[0], 0.0f);
  q = __builtin_fmaf(x[1], x[1], q);
  q = __builtin_fmaf(x[2], x[2], q);
  q = __builtin_fmaf(x[3], x[3], q);
}

// 16-lane (one DPP row) all-reduce: every lane of the row ends up with the same sum
__device__ __forceinline__ float row16_sum(float v) {
#define MK_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
  MK_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  MK_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  MK_DPP_ADD(0x141);   // row_half_mirror
  MK_DPP_ADD(0x140);   // row_mirror
#undef MK_DPP_ADD
  return v;
}

// Consumer prologue of the folded LayerNorm: (rstd, -mean * rstd) of the tile's BM rows into LDS (prm[BM]), from the
// producer's per-slot partial sums.  NT = 2 * BM threads: thread t sums one half of the slots of row t >> 1 (fixed
// order: the result does not depend on the schedule), the pair is combined in fp64 (E[x^2] - mean^2 without the fp32
// cancellation).  The K loop's barriers order the LDS writes before the epilogue that reads them.
// In the 256x256 kernel the D = 1024 case is split in two so that the loads are issued BEFORE the first LDS-DMA stage and
// consumed while it is in flight: LnRowLoads16::issue() (inline-asm loads, invisible to hipcc's vmcnt bookkeeping like the
// DMA pieces themselves) ... DMA pieces ... finish<n_younger>() waits with a counted vmcnt for exactly these loads.
// Every other width takes the plain path below (ln_params_to_lds).
// 16 slots per row (D = 1024, ViT-L): the tile's statistics are one contiguous block of 256 rows x 128 B.  Each wave reads
// 4 KiB of it with 4 fully coalesced 16-byte loads per lane (a per-row gather -- 64 lanes x 8 B on 32 different lines per
// instruction -- cost ~1.5 us of address coalescing per tile in front of the first barrier); lane l of load j then holds
// slots 2c, 2c + 1 (c = l & 7) of row 32 * wave + 8 * j + (l >> 3): an 8-lane DPP reduction finishes the row.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
struct LnRowLoads16 {
  u32x4 v[4];
  __device__ __forceinline__ void issue(const GemmParams& p, int m0, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    const long long last = (long long)p.M * 128 - 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long off = ((long long)(m0 + wave * 32 + j * 8) * 128) + lane * 16;
      off = off < last ? off : last;   // rows past M: any valid address (their parameters are never used)
      const char* a = (const char*)p.ln_stats + off;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(a) : "memory");
    }
  }
  // YOUNGER = VMEM instructions this wave issued after issue() that may still be in flight
  // publish: this tile belongs to the first tile column and writes the row means for the next producer (rows m0 ...)
  template <int YOUNGER>
  __device__ __forceinline__ void finish(const GemmParams& p, int tid, float2* prm, int m0 = 0, bool publish = false) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(YOUNGER) : "memory");
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 f = __builtin_bit_cast(f32x4, v[j]);
      float s = f[0] + f[2], q = f[1] + f[3];
#define MK_DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
      MK_DPP_ADD(s, 0xB1);  MK_DPP_ADD(q, 0xB1);     // quad_perm [1,0,3,2]
      MK_DPP_ADD(s, 0x4E);  MK_DPP_ADD(q, 0x4E);     // quad_perm [2,3,0,1]
      MK_DPP_ADD(s, 0x141); MK_DPP_ADD(q, 0x141);    // row_half_mirror: the other quad of the 8-lane group
#undef MK_DPP_ADD
      const double mean = (double)s * (1.0 / 1024.0);
      double var = (double)q * (1.0 / 1024.0) - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const float rstd = 1.0f / sqrtf((float)var + p.ln_eps);
      if ((lane & 7) == 0) {
        const int r = wave * 32 + j * 8 + (lane >> 3);
        prm[r] = make_float2(rstd, -(float)mean * rstd);
        if (publish && m0 + r < p.M) p.ln_shift_out[m0 + r] = (float)mean;
      }
    }
  }
};

// The same statistics through LDS (persistent kernel, tiles after a workgroup's first): the wave's four 1-KiB pieces are copied
// HBM/L2 -> LDS by the DMA engine, two at a time into a 2-KiB staging area of its own (16 KiB for the workgroup), while the
// K loop of the tile runs -- no register is held across a stage (as registers, even 8 per lane pushed the consumer kernel
// over the 256-register limit).  Same reduction, same order, same bits as LnRowLoads16.
struct LnRowDma16 {
  template <int J0>
  __device__ __forceinline__ static void issue(const GemmParams& p, int m0, int wave, int lane, char* stage_lds) {
    const long long last = (long long)p.M * 128 - 16;
#pragma unroll
    for (int j = J0; j < J0 + 2; ++j) {
      const long long base = (long long)(m0 + wave * 32 + j * 8) * 128;   // wave-uniform
      long long off = base + lane * 16;
      off = off < last ? off : last;   // rows past M: any valid address (their parameters are never used)
      glds16_sv((const char*)p.ln_stats, (unsigned)off, stage_lds + (wave * 2 + (j - J0)) * 1024);
    }
  }
  template <int J0>
  __device__ __forceinline__ static void finish(const GemmParams& p, int wave, int lane, float2* prm, const char* stage_lds, int m0,
                                                bool publish) {
#pragma unroll
    for (int j = J0; j < J0 + 2; ++j) {
      const f32x4 f = *(const f32x4*)(stage_lds + (wave * 2 + (j - J0)) * 1024 + lane * 16);
      float s = f[0] + f[2], q = f[1] + f[3];
#define MK_DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
      MK_DPP_ADD(s, 0xB1);  MK_DPP_ADD(q, 0xB1);
      MK_DPP_ADD(s, 0x4E);  MK_DPP_ADD(q, 0x4E);
      MK_DPP_ADD(s, 0x141); MK_DPP_ADD(q, 0x141);
#undef MK_DPP_ADD
      const double mean = (double)s * (1.0 / 1024.0);
      double var = (double)q * (1.0 / 1024.0) - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const float rstd = 1.0f / sqrtf((float)var + p.ln_eps);
      if ((lane & 7) == 0) {
        const int r = wave * 32 + j * 8 + (lane >> 3);
        prm[r] = make_float2(rstd, -(float)mean * rstd);
        if (publish && m0 + r < p.M) p.ln_shift_out[m0 + r] = (float)mean;
      }
    }
  }
};

// Producer prologue of the row centring: the tile's BM row shifts into LDS (zeros without centring), loaded by inline asm
// BEFORE the first LDS-DMA pieces and written after them (a load hipcc knows about would be waited for with vmcnt(0)).
struct ShiftLoad {
  float v;
  __device__ __forceinline__ void issue(const GemmParams& p, int m0, int tid, int BM) {
    v = 0.f;
    if (p.ln_shift_in && tid < BM) {
      int m = m0 + tid;
      m = m < p.M ? m : p.M - 1;
      const float* a = p.ln_shift_in + m;
      asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(a) : "memory");
    }
  }
  template <int YOUNGER>   // YOUNGER < 0: already waited for by the caller
  __device__ __forceinline__ void finish(int tid, int BM, float* shl) {
    if constexpr (YOUNGER >= 0)
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(YOUNGER) : "memory");
    else
      asm volatile("" : "+v"(v) : : "memory");
    if (tid < BM) shl[tid] = v;
  }
};

// Generic widths (and the 128x128 kernel).  ONE summation order with LnRowLoads16: a balanced binary tree over the slots in
// natural order, in fp32 -- pairs (2c, 2c+1), quads, eights, then the two halves -- with absent slots as zeros (x + 0 = x), so a
// row's (rstd, mean) do not depend on which schedule normalises it.  Up to 16 slots (D <= 1024) take the tree; wider rows
// fall back to a sequential sum per half (their own, still schedule-independent, order).
template <int BM, int NT>
__device__ __forceinline__ void ln_params_to_lds(const GemmParams& p, int m0, int tid, float2* prm, bool publish = false) {
  static_assert(NT == 2 * BM, "two threads per row");
  const int r = tid >> 1, h = tid & 1;
  int m = m0 + r;
  m = m < p.M ? m : p.M - 1;
  const float2* st = (const float2*)p.ln_stats + (long long)m * p.ln_nslot;
  float s, q;
  if (p.ln_nslot <= 16) {
    // thread h of the pair takes slots 8h .. 8h+7: 8 independent 8-byte loads in flight
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 8 * h + j < p.ln_nslot ? st[8 * h + j] : make_float2(0.f, 0.f);
    const float s01 = v[0].x + v[1].x, s23 = v[2].x + v[3].x, s45 = v[4].x + v[5].x, s67 = v[6].x + v[7].x;
    const float q01 = v[0].y + v[1].y, q23 = v[2].y + v[3].y, q45 = v[4].y + v[5].y, q67 = v[6].y + v[7].y;
    s = (s01 + s23) + (s45 + s67);
    q = (q01 + q23) + (q45 + q67);
  } else {
    const int per = (p.ln_nslot + 1) >> 1;
    const int lo = h * per, hi = min(p.ln_nslot, lo + per);
    s = q = 0.f;
    for (int i0 = lo; i0 < hi; i0 += 8) {
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = i0 + j < hi ? st[i0 + j] : make_float2(0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s += v[j].x;
        q += v[j].y;
      }
    }
  }
  s += __shfl_xor(s, 1, 64);   // fp32, as the DPP tree of LnRowLoads16 (commutative: both threads hold the same bits)
  q += __shfl_xor(q, 1, 64);
  const double invn = 1.0 / (64.0 * p.ln_nslot);
  const double mean = (double)s * invn;
  double var = (double)q * invn - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = 1.0f / sqrtf((float)var + p.ln_eps);
  if (h == 0) {
    prm[r] = make_float2(rstd, -(float)mean * rstd);
    if (publish && m0 + r < p.M) p.ln_shift_out[m0 + r] = (float)mean;
  }
}

template <typename T>
__device__ __forceinline__ T to_lp(float v) { return (T)v; }

// four 16-bit values <-> two dwords, by hand: arrays of bf16x4 / f16x4 kept live across a wait made hipcc keep every
// element in its own VGPR (the split-stream epilogue then spilled the accumulators of EVERY tile to scratch: +25 % on all
// GEMM launches).  As raw dwords the 32 pre-loaded chunks of a wave are 64 + 64 registers, like the fp32 form.
template <typename T>
__device__ __forceinline__ f32x4 unpack4(uint2 u);
template <>
__device__ __forceinline__ f32x4 unpack4<__bf16>(uint2 u) {
  return f32x4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
               __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
}
template <>
__device__ __forceinline__ f32x4 unpack4<_Float16>(uint2 u) {
  const f16x4 h = __builtin_bit_cast(f16x4, u);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
template <>
__device__ __forceinline__ f32x4 unpack4<float>(uint2) { return f32x4{0.f, 0.f, 0.f, 0.f}; }   // fp32 mode has no split stream
template <typename T>
__device__ __forceinline__ uint2 pack4(f32x4 v) {
  typename Lp<T>::V4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = to_lp<T>(v[e]);
  if constexpr (sizeof(T) == 2) return __builtin_bit_cast(uint2, o);
  else return uint2{0u, 0u};
}

// swap bits 2 and 3 of a token index: the V^T image is stored key-permuted so that the 8 keys a lane
// owns after the 32x32 S^T MFMA are one contiguous 16-B chunk (see mk_attention.hip)
__device__ __forceinline__ int vperm(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }

// Per-lane LDS-DMA state of one workgroup tile.  Piece (wave*J + j) is 8 rows x 128 B = 1 KiB of the LDS image; this
// lane feeds row +(lane>>3), 16-byte chunk lane&7 of it, fetching the XOR-swizzled source chunk.
template <typename T, int AMODE, int NW, int AJ, int WJ, bool SP = false>
struct Stager {
  const T* A;
  const T* A2;
  const T* wrow[WJ];
  long long aoff[AJ];  // dense: element offset of (row, swizzled chunk) (SP: + the chunk's plane); conv: bordered row of the pixel
  int wave, srow, sp;

  __device__ __forceinline__ void init(const GemmParams& p, int g, int m0, int n0, int wave_, int lane) {
    wave = wave_;
    srow = lane >> 3;
    sp = lane & 7;
    A = (const T*)p.A + (long long)g * p.strideA_g;
    A2 = p.A2 ? (const T*)p.A2 + (long long)g * p.strideA2_g : nullptr;
    const T* W = (const T*)p.W + (long long)g * p.strideW_g;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int r = (wave * WJ + j) * 8 + srow;
      int n = n0 + r;
      n = n < p.N ? n : p.N - 1;
      wrow[j] = W + (long long)n * p.ldw + swz8(r, sp) * EPC<T>;
    }
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int r = (wave * AJ + j) * 8 + srow;
      int m = m0 + r;
      m = m < p.M ? m : p.M - 1;
      const int c = swz8(r, sp);
      // SP: source chunks 0..3 of an LDS row are 32 elements of the HI plane, 4..7 the same 32 elements of the LO plane
      if (AMODE == A_DENSE) aoff[j] = SP ? (long long)m * p.lda + (c & 3) * EPC<T> + (long long)(((c >> 2) ? p.pl1[1] : p.pl1[0]) / sizeof(T))
                                         : (long long)m * p.lda + c * EPC<T>;
      else aoff[j] = bordered_row(m, p.H, p.Wd);
    }
  }

  __device__ __forceinline__ void issue(const GemmParams& p, char* sA, char* sW, int kt) const {
    const int k0 = kt * KT<T>;                     // W columns of this stage
    const int ka = SP ? kt * (KT<T> / 2) : k0;     // contraction index of its A columns
    if (AMODE == A_DENSE) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) glds16(A + aoff[j] + ka, sA + (wave * AJ + j) * 1024);
    } else {
      // wave-uniform: which source / tap does this K tile belong to
      const int kc = 9 * p.C1;
      // (the source pointers as values first: a select between two MEMBER loads keeps the whole Stager in scratch)
      const T* const s1 = A;
      const T* const s2 = A2;
      const T* src;
      int cs, c0, shift = 0;   // shift: the tap in bordered rows (out-of-image taps land on zero border rows)
      unsigned plh, pll;
      if (ka < kc) {
        const int tap = ka / p.C1;
        c0 = ka - tap * p.C1;
        shift = (tap / 3 - 1) * (p.Wd + 1) + tap % 3 - 1;
        src = s1;
        cs = p.C1;
        plh = p.pl1[0];
        pll = p.pl1[1];
      } else {
        c0 = ka - kc;
        src = s2;
        cs = p.C2;
        plh = p.pl2[0];
        pll = p.pl2[1];
      }
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int r = (wave * AJ + j) * 8 + srow;
        const int c = swz8(r, sp);
        if (SP) glds16(src + (aoff[j] + shift) * cs + c0 + (c & 3) * EPC<T> + (long long)(((c >> 2) ? pll : plh) / sizeof(T)), sA + (wave * AJ + j) * 1024);
        else glds16(src + (aoff[j] + shift) * cs + c0 + c * EPC<T>, sA + (wave * AJ + j) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) glds16(wrow[j] + k0, sW + (wave * WJ + j) * 1024);
  }
};

// ---- epilogue: lane owns row m = ...+(lane&15), features n..n+3 with n = ...+(lane>>4)*4 ----
// EPI / ACT / HAS_BIAS are compile-time inside the 32x unrolled store loop; epilogue() dispatches once per tile.
template <typename T, int WMF, int EPI, int ACT, bool HAS_BIAS, bool LN = false, bool SPLIT = false>
__device__ __forceinline__ void epilogue_impl(const GemmParams& p, f32x4 (&acc)[WMF][4], int m0, int n0, int wm, int wn, int lane,
                                              int g, const float2* lnp = nullptr) {
  using V4 = typename Lp<T>::V4;
  const int fr = lane & 15, fg = lane >> 4;
  const float* bias = HAS_BIAS ? p.bias + (long long)g * p.strideBias_g : nullptr;
  const int nb = n0 + wn * 64 + fg * 4;
  f32x4 bv[4], cs[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = nb + ni * 16;
    bv[ni] = (HAS_BIAS && n < p.N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    cs[ni] = (LN && n < p.N) ? *(const f32x4*)(p.ln_colsum + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // producer side of the folded LayerNorm (LS_RESIDUAL / PATCH on the split residual stream)
  const bool fin = SPLIT && EPI == MK_EPI_LS_RESIDUAL && p.out_f32 != nullptr;   // last block: fp32 rows out, no by-products
#pragma unroll
  for (int mi = 0; mi < WMF; ++mi) {
    const int m = m0 + wm * (WMF * 16) + mi * 16 + fr;
    if (SPLIT) {   // no early exit: every lane takes part in the cross-lane sums (invalid rows / columns contribute 0)
      const bool mok = m < p.M;
      long long xrow = m;
      int tok1 = 0;   // PATCH: row of the position table
      if (EPI == MK_EPI_PATCH) {
        const int img = m / p.npatch;
        tok1 = 1 + (m - img * p.npatch);
        xrow = (long long)img * (p.npatch + 1) + tok1;
      }
      // row centring: the mean this row had before the update (published by the consumer in between) comes off
      const float shv = (EPI == MK_EPI_LS_RESIDUAL && p.ln_shift_in && mok) ? p.ln_shift_in[m] : 0.f;
      float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};   // per column quad fg + 4 ni (see quad_stats)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = nb + ni * 16;
        if (!(mok && n < p.N)) continue;
        f32x4 v = acc[mi][ni] + bv[ni];
        T* ph = (T*)p.xh + xrow * p.ldxs + n;
        T* pl = (T*)p.xl + xrow * p.ldxs + n;
        if (EPI == MK_EPI_LS_RESIDUAL) {
          const f32x4 gm = *(const f32x4*)(p.gamma + n);
          const V4 h = *(const V4*)ph, l = *(const V4*)pl;
          f32x4 r;
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = (float)h[e] + (float)l[e];
          r += gm * v;
          r -= shv;
          v = r;
        } else {
          v += *(const f32x4*)(p.pos + (long long)tok1 * p.N + n);
        }
        if (fin) {
          *(f32x4*)(p.out_f32 + xrow * p.ldc + n) = v;
          continue;
        }
        V4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          oh[e] = to_lp<T>(v[e]);
          ol[e] = to_lp<T>(v[e] - (float)oh[e]);
        }
        quad_stats(v, s4[ni], q4[ni]);
        *(V4*)ph = oh;
        *(V4*)pl = ol;
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {   // (q, q^1) then (.., q^2): quads fg^1, fg^2 of the same ni live in lane^16, lane^32
        s4[ni] += __shfl_xor(s4[ni], 16, 64);
        q4[ni] += __shfl_xor(q4[ni], 16, 64);
        s4[ni] += __shfl_xor(s4[ni], 32, 64);
        q4[ni] += __shfl_xor(q4[ni], 32, 64);
      }
      const float ssum = (s4[0] + s4[1]) + (s4[2] + s4[3]), qsum = (q4[0] + q4[1]) + (q4[2] + q4[3]);
      if (!fin && fg == 0 && mok && n0 + wn * 64 < p.N)
        ((float2*)p.stats_out)[xrow * p.nslot_out + ((n0 + wn * 64) >> 6)] = make_float2(ssum, qsum);
      continue;
    }
    if (m >= p.M) continue;
    float2 prm = make_float2(1.f, 0.f);
    if (LN) prm = lnp[wm * (WMF * 16) + mi * 16 + fr];
    int img = 0, tok = 0;
    if (EPI == MK_EPI_QKV) {
      img = m / p.ntok;
      tok = m - img * p.ntok;
    } else if (EPI == MK_EPI_PATCH) {
      img = m / p.npatch;
      tok = m - img * p.npatch;
    }
    // conv (H > 0): the identity residual and, with bord_out, the 16-bit output are bordered feature maps
    const long long brow = (EPI == MK_EPI_STORE && p.H > 0) ? bordered_row(m, p.H, p.Wd) : (long long)m;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = nb + ni * 16;
      if (n >= p.N) continue;  // N is a multiple of 4 (checked on the host)
      f32x4 v = acc[mi][ni];
      if (EPI == MK_EPI_STORE && p.npass > 1) {   // split-operand conv: undo the planes' power-of-two scaling
        const float sc = p.acc_scale;             // (element-wise: `v *= p.acc_scale` keeps a slice of GemmParams in scratch)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= sc;
      }
      if (LN) v = v * prm.x + (cs[ni] * prm.y + bv[ni]);
      else if (HAS_BIAS) v += bv[ni];
      if (EPI == MK_EPI_STORE) {
        if (p.resid_lp) {
          const V4 r = *(const V4*)((const T*)p.resid_lp + (long long)g * p.strideResid_g + brow * p.ldc + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
        }
        if (ACT == MK_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (ACT == MK_ACT_GELU) {
          v = gelu_erf4(v);
        }
        if (p.out_lo) {   // split-operand conv chain: the next conv's (hi, lo) planes
          const float ps = p.plane_scale;
          f16x4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sv = sat16(v[e] * ps, p.sat_flag);   // saturate at fp16's largest finite value
            oh[e] = (_Float16)sv;
            ol[e] = (_Float16)(sv - (float)oh[e]);
          }
          const long long o = (long long)g * p.strideOut_g + (p.bord_out ? brow : (long long)m) * p.ldc + n;
          *(f16x4*)((_Float16*)p.out_lp + o) = oh;
          *(f16x4*)((_Float16*)p.out_lo + o) = ol;
        } else if (p.out_f32) {
          *(f32x4*)(p.out_f32 + (long long)g * p.strideOut_g + (p.bord_out ? brow : (long long)m) * p.ldc + n) = v;
        } else {
          V4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = to_lp<T>(v[e]);
          *(V4*)((T*)p.out_lp + (long long)g * p.strideOut_g + (p.bord_out ? brow : (long long)m) * p.ldc + n) = o;
        }
      } else if (EPI == MK_EPI_LS_RESIDUAL) {
        float* x = p.out_f32 + (long long)m * p.ldc + n;
        const f32x4 gm = *(const f32x4*)(p.gamma + n);
        f32x4 r = *(const f32x4*)x;
        r += gm * v;
        *(f32x4*)x = r;
      } else if (EPI == MK_EPI_PATCH) {
        const f32x4 pe = *(const f32x4*)(p.pos + (long long)(1 + tok) * p.N + n);
        *(f32x4*)(p.out_f32 + ((long long)img * (p.npatch + 1) + 1 + tok) * p.ldc + n) = v + pe;
      } else {  // MK_EPI_QKV
        const int D = p.heads * 64;
        const int which = n / D;
        const int rem = n - which * D;
        const int head = rem >> 6, d = rem & 63;
        const long long hb = (long long)img * p.heads + head;
        if (which == 2) {
          T* dst = (T*)p.vt + (hb * 64 + d) * p.ntok_pad + vperm(tok);
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[(long long)e * p.ntok_pad] = to_lp<T>(v[e]);
        } else {
          if (which == 0) { const float qs = p.qscale; _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] *= qs; }
          V4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = to_lp<T>(v[e]);
          T* base = (T*)(which == 0 ? p.q : p.k);
          *(V4*)(base + (hb * p.ntok_pad + tok) * 64 + d) = o;
        }
      }
    }
  }
}

template <typename T, int WMF>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[WMF][4], int m0, int n0, int wm, int wn, int lane,
                                         int g, const float2* lnp = nullptr) {
  if (p.xh) {   // folded LayerNorm (producer): split residual stream
    if (p.epi == MK_EPI_PATCH) epilogue_impl<T, WMF, MK_EPI_PATCH, MK_ACT_NONE, true, false, true>(p, acc, m0, n0, wm, wn, lane, g);
    else epilogue_impl<T, WMF, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true, false, true>(p, acc, m0, n0, wm, wn, lane, g);
    return;
  }
  if (p.ln_stats) {   // folded LayerNorm (consumer): QKV split, or bias (+ GELU) with a 16-bit output
    if (p.epi == MK_EPI_QKV) epilogue_impl<T, WMF, MK_EPI_QKV, MK_ACT_NONE, true, true>(p, acc, m0, n0, wm, wn, lane, g, lnp);
    else if (p.act == MK_ACT_GELU) epilogue_impl<T, WMF, MK_EPI_STORE, MK_ACT_GELU, true, true>(p, acc, m0, n0, wm, wn, lane, g, lnp);
    else epilogue_impl<T, WMF, MK_EPI_STORE, MK_ACT_NONE, true, true>(p, acc, m0, n0, wm, wn, lane, g, lnp);
    return;
  }
  switch (p.epi) {   // wave-uniform, once per output tile
    case MK_EPI_LS_RESIDUAL: epilogue_impl<T, WMF, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true>(p, acc, m0, n0, wm, wn, lane, g); break;
    case MK_EPI_QKV: epilogue_impl<T, WMF, MK_EPI_QKV, MK_ACT_NONE, true>(p, acc, m0, n0, wm, wn, lane, g); break;
    case MK_EPI_PATCH: epilogue_impl<T, WMF, MK_EPI_PATCH, MK_ACT_NONE, true>(p, acc, m0, n0, wm, wn, lane, g); break;
    default:
      if (!p.bias) {
        if (p.act == MK_ACT_RELU) epilogue_impl<T, WMF, MK_EPI_STORE, MK_ACT_RELU, false>(p, acc, m0, n0, wm, wn, lane, g);
        else if (p.act == MK_ACT_GELU) epilogue_impl<T, WMF, MK_EPI_STORE, MK_ACT_GELU, false>(p, acc, m0, n0, wm, wn, lane, g);
        else epilogue_impl<T, WMF, MK_EPI_STORE, MK_ACT_NONE, false>(p, acc, m0, n0, wm, wn, lane, g);
      } else {
        if (p.act == MK_ACT_RELU) epilogue_impl<T, WMF, MK_EPI_STORE, MK_ACT_RELU, true>(p, acc, m0, n0, wm, wn, lane, g);
        else if (p.act == MK_ACT_GELU) epilogue_impl<T, WMF, MK_EPI_STORE, MK_ACT_GELU, true>(p, acc, m0, n0, wm, wn, lane, g);
        else epilogue_impl<T, WMF, MK_EPI_STORE, MK_ACT_NONE, true>(p, acc, m0, n0, wm, wn, lane, g);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// LDS-staged epilogue of the full-line ping-pong kernel.  In the accumulator layout a lane owns 4 features of one
// row, so a direct store instruction touches 16 rows x 32 B -- measured ~65 cycles per instruction and 4.4-7.2 us
// per 256x256 tile (13-21 % of a K = 1024 tile).  Each wave bounces its 128x64 block through a private LDS slice
// (wave-local, LDS is in-order per wave: no barrier) and then moves whole rows: 16-bit outputs 16 B per lane = 8 full
// 128-byte lines per instruction, fp32 outputs 4 x 256 B.
// Round 5: the slice is 8 KiB (16-bit outputs: two rounds of 64 rows; fp32: four of 32), not 16, and it is two 4-KiB CHUNKS
// -- wl0 = stage 1's A region + 4 KiB x wave, wl1 = stage 1's W region + 4 KiB x wave: exactly where THIS wave's own LDS-DMA
// pieces of a stage land (piece = (4 x wave + j) KiB of the A half / of W).  The eight slices fill stage 1 of the ring and
// nothing else: a persistent tile loop has the NEXT tile's stage 0 in flight into the other half of the ring while the
// epilogue drains, and a wave that is done may issue its next A pieces without waiting for its neighbours.  The residual-stream read-modify-write and the q / k head-major stores become fully coalesced the same way;
// only the V^T part of the qkv split keeps element stores (its rows are tokens at an arbitrary 16-group alignment).
// XOR swizzles: 16-bit rows of 128 B, chunk ^ (row & 7); fp32 rows of 256 B, chunk ^ (row & 15).
// SPLIT (split residual stream, §2.1b of LABNOTES.md) is instantiated for INTERIOR tiles only (no row / column predicates:
// straight-line code; with per-row branches this variant pushed the whole kernel over 256 VGPRs and hipcc spilled half the
// accumulators of every tile of every launch) -- edge tiles take the direct epilogue; FIN: fp32 rows out (last block).
template <typename T, int EPI, int ACT, bool HAS_BIAS, bool LN = false, bool SPLIT = false, bool FIN = false, bool CONV = false>
__device__ __forceinline__ void epilogue_lds_impl(const GemmParams& p, f32x4 (&acc)[8][4], char* wl0, char* wl1, int m0, int n0, int wm,
                                                  int wn, int lane, int g, const float2* lnp = nullptr) {
  using V4 = typename Lp<T>::V4;
  using V8 = typename Lp<T>::V8;
  const int fr = lane & 15, fg = lane >> 4;
  const float* bias = HAS_BIAS ? p.bias + (long long)g * p.strideBias_g : nullptr;
  const int nw = n0 + wn * 64;        // first feature of this wave's block
  const int mw = m0 + wm * 128;       // first row
  // qkv split: (image, token) of row mw + r without a per-row integer division (~25 VALU incl. quarter-rate ops, 16 per
  // lane before): one wave-uniform division; the wave's 128 rows cross at most one image boundary when ntok >= 128
  int img0 = 0, tok0 = 0;
  if (EPI == MK_EPI_QKV) {
    img0 = mw / p.ntok;
    tok0 = mw - img0 * p.ntok;
  }
  auto img_tok = [&](int r, int& img, int& tok) {
    if (p.ntok >= 128) {
      const int t = tok0 + r;
      const bool wrap = t >= p.ntok;
      img = img0 + (wrap ? 1 : 0);
      tok = wrap ? t - p.ntok : t;
    } else {
      const int m = mw + r;
      img = m / p.ntok;
      tok = m - img * p.ntok;
    }
  };
  f32x4 bv[4], cs[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = nw + fg * 4 + ni * 16;
    bv[ni] = (HAS_BIAS && n < p.N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    cs[ni] = (LN && n < p.N) ? *(const f32x4*)(p.ln_colsum + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // folded LayerNorm (consumer): (rstd, -mean * rstd) of this lane's 8 rows, left in LDS by the kernel prologue
  float2 prm[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) prm[mi] = LN ? lnp[wm * 128 + mi * 16 + fr] : make_float2(1.f, 0.f);
  const bool lp_out = EPI == MK_EPI_QKV || (EPI == MK_EPI_STORE && !p.out_f32 && !p.out_lo);
  if (lp_out) {
    int which = 0, head = 0;
    if (EPI == MK_EPI_QKV) {
      const int D = p.heads * 64;
      which = nw / D;
      head = (nw - which * D) >> 6;
      if (which == 2) {   // V^T, key-permuted: element stores straight from the accumulators
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const int m = mw + mi * 16 + fr;
          if (m >= p.M) continue;
          int img, tok;
          img_tok(mi * 16 + fr, img, tok);
          const long long hb = (long long)img * p.heads + head;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const f32x4 v = LN ? acc[mi][ni] * prm[mi].x + (cs[ni] * prm[mi].y + bv[ni]) : acc[mi][ni] + bv[ni];
            T* dst = (T*)p.vt + (hb * 64 + ni * 16 + fg * 4) * p.ntok_pad + vperm(tok);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(long long)e * p.ntok_pad] = to_lp<T>(v[e]);
          }
        }
        return;
      }
    }
    const int rr = lane >> 3, c = lane & 7;
    const int n = nw + c * 8;
    // conv output that feeds the next conv: bordered rows (walked: the lane's rows are 8 apart)
    const bool bord = CONV && p.bord_out;
    BorderedRow bw;
    if (bord) bw.init(mw + rr, p.H, p.Wd);
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // two rounds of 64 rows through the 8-KiB slice (rows 0..31 in chunk 0, 32..63 in chunk 1)
#pragma unroll
      for (int mq = 0; mq < 4; ++mq) {
        const int mi = h * 4 + mq;
        const int r = mi * 16 + fr;
        char* wrow = (mq < 2 ? wl0 : wl1) + ((mq & 1) * 16 + fr) * 128;
        // identity residual of a conv: a bordered feature map (one launch per forward takes this path)
        const long long rrow = (CONV && p.resid_lp) ? bordered_row(min(mw + r, p.M - 1), p.H, p.Wd) : 0;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          f32x4 v = acc[mi][ni];
          if (LN) v = v * prm[mi].x + (cs[ni] * prm[mi].y + bv[ni]);
          else if (HAS_BIAS) v += bv[ni];
          if (EPI == MK_EPI_QKV) {
            if (which == 0) { const float qs = p.qscale; _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] *= qs; }
          } else {
            if (CONV && p.resid_lp) {
              const int m = mw + r, n2 = nw + fg * 4 + ni * 16;
              if (m < p.M && n2 < p.N) {
                const V4 rs = *(const V4*)((const T*)p.resid_lp + (long long)g * p.strideResid_g + rrow * p.ldc + n2);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rs[e];
              }
            }
            if (ACT == MK_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (ACT == MK_ACT_GELU) {
              v = gelu_erf4(v);
            }
          }
          V4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = to_lp<T>(v[e]);
          const int cc = ni * 2 + (fg >> 1);
          *(V4*)(wrow + ((cc ^ (fr & 7)) << 4) + (fg & 1) * 8) = o;
        }
      }
#pragma unroll
      for (int i8 = 0; i8 < 8; ++i8) {
        const int it = h * 8 + i8;
        const int r = it * 8 + rr;
        const int m = mw + r;
        const V8 val = *(const V8*)((i8 < 4 ? wl0 : wl1) + ((i8 & 3) * 8 + rr) * 128 + ((c ^ (rr & 7)) << 4));
        const int extra = bord ? bw.extra : 0;
        if (bord) bw.step(8, p.H, p.Wd);
        if (m >= p.M || n >= p.N) continue;
        T* dst;
        if (EPI == MK_EPI_QKV) {
          int img, tok;
          img_tok(r, img, tok);
          dst = (T*)(which == 0 ? p.q : p.k) + (((long long)img * p.heads + head) * p.ntok_pad + tok) * 64 + c * 8;
        } else {
          dst = (T*)p.out_lp + (long long)g * p.strideOut_g + ((long long)m + extra) * p.ldc + n;
        }
        if (n + 8 <= p.N) {
          *(V8*)dst = val;
        } else {   // N % 8 == 4: the last chunk is half wide
          V4 lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) lo[e] = val[e];
          *(V4*)dst = lo;
        }
      }
    }
  } else {
    const int rr = lane >> 4, c = lane & 15;
    const int n = nw + c * 4;
    f32x4 gm = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI == MK_EPI_LS_RESIDUAL && n < p.N) gm = *(const f32x4*)(p.gamma + n);
    // split stream: the bias is added in the drain layout (a lane's column quad is fixed there: 4 registers instead of the
    // 16 of the accumulator layout; same operation order, bit-identical) -- this variant runs at the 256-register limit
    f32x4 bvd = f32x4{0.f, 0.f, 0.f, 0.f};
    if (SPLIT && HAS_BIAS && n < p.N) bvd = *(const f32x4*)(bias + n);
    // read-modify-write of the residual stream: all 16 loads of a half are issued before anything waits on them (one
    // HBM round trip per half instead of one per row group: measured 0.68 us per dependent load -> 22 us per tile);
    // the second half's loads go out while the first half is still being stored
    // SPLIT (folded LayerNorm, producer): the stream is two 16-bit planes (x = hi + lo), 8 + 8 bytes per lane and row
    f32x4 xr[2][SPLIT ? 1 : 16];
    uint2 xh[2][SPLIT ? 16 : 1], xl[2][SPLIT ? 16 : 1];
    // split planes: wave-uniform row base (SGPRs) + one 32-bit per-lane element offset, so that the 64 loads and 64 stores
    // of a wave share ONE address register (kept as 64-bit per-lane addresses they pushed the kernel over 256 VGPRs and
    // hipcc spilled the accumulators of every tile)
    const unsigned lane_off = (unsigned)rr * (unsigned)p.ldxs + (unsigned)n;
    // the same offset, opaque to the optimiser, for the stores: otherwise hipcc keeps the 64 per-lane 64-bit addresses of
    // the loads alive for the stores to the same places (128 registers, most of them spilled) instead of re-deriving each
    // from the scalar row base with one v_lshl_add_u64
    unsigned lane_off_st = lane_off;
    asm volatile("" : "+v"(lane_off_st));
    auto plane_row = [&](void* plane, int half, int it) -> T* {   // uniform part: first row of the 4-row group
      return (T*)plane + (long long)(mw + half * 64 + it * 4) * p.ldxs;
    };
    // Round 4: the planes are addressed as BUFFERS (SGPR descriptor + wave-uniform byte offset in an SGPR + the lane's constant
    // 32-bit byte offset): a plane load / store is one instruction with no VALU address arithmetic (as 64-bit per-lane
    // addresses each cost a v_lshl_add_u64: 6 of the ~70 VALU issues per 4 elements of this VALU-bound epilogue).  Byte
    // offsets fit 32 bits: the launcher routes operands of 2^31 elements or more to the 128x128 kernel.
    __amdgpu_buffer_rsrc_t rs_h, rs_l;
    unsigned voff_b = 0;
    if constexpr (SPLIT && EPI == MK_EPI_LS_RESIDUAL && sizeof(T) == 2) {
      rs_h = __builtin_amdgcn_make_buffer_rsrc(p.xh, 0, 0x7fffffff, 0x00020000);
      rs_l = __builtin_amdgcn_make_buffer_rsrc(p.xl, 0, 0x7fffffff, 0x00020000);
      voff_b = lane_off * 2u;
    }
    auto soff_b = [&](int half, int it) -> unsigned {   // wave-uniform: byte offset of the first row of the 4-row group
      return (unsigned)(mw + half * 64 + it * 4) * (unsigned)p.ldxs * 2u;
    };
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    // it0 / it1: range of 4-row groups of the half (the split variant works in quarters, see the end of the function)
    auto preload = [&](int half, int it0 = 0, int it1 = 16) {
      if (EPI != MK_EPI_LS_RESIDUAL) return;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        if (it < it0 || it >= it1) continue;
        const int m = mw + half * 64 + it * 4 + rr;
        const bool ok = SPLIT || (m < p.M && n < p.N);
        if constexpr (SPLIT && EPI == MK_EPI_LS_RESIDUAL && sizeof(T) == 2) {
          const u32x2_t a = __builtin_amdgcn_raw_buffer_load_b64(rs_h, voff_b, soff_b(half, it), 0);
          const u32x2_t b = __builtin_amdgcn_raw_buffer_load_b64(rs_l, voff_b, soff_b(half, it), 0);
          xh[half][it] = uint2{a[0], a[1]};
          xl[half][it] = uint2{b[0], b[1]};
        } else if (SPLIT) {
          xh[half][it] = ok ? *(const uint2*)(plane_row(p.xh, half, it) + lane_off) : uint2{0u, 0u};
          xl[half][it] = ok ? *(const uint2*)(plane_row(p.xl, half, it) + lane_off) : uint2{0u, 0u};
        } else {
          xr[half][it] = ok ? *(const f32x4*)(p.out_f32 + (long long)m * p.ldc + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    // one QUARTER (32 rows: accumulator rows mi = 2 sub, 2 sub + 1 of the half) into the 8-KiB slice: rows 0..15 in chunk 0,
    // 16..31 in chunk 1
    auto stageq = [&](int half, int sub) {
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        const int mi = sub * 2 + m2;
        const int r = mi * 16 + fr;
        char* wrow = (m2 == 0 ? wl0 : wl1) + fr * 256;
        const long long rrow = (CONV && p.resid_lp) ? bordered_row(min(mw + half * 64 + r, p.M - 1), p.H, p.Wd) : 0;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          f32x4 v = acc[half * 4 + mi][ni];
          if (EPI == MK_EPI_STORE && p.npass > 1) {   // split operands: undo the planes' power-of-two scaling
            const float sc = p.acc_scale;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= sc;
          }
          if (HAS_BIAS && !SPLIT) v += bv[ni];
          if (EPI == MK_EPI_STORE) {
            if (CONV && p.resid_lp) {
              const int m = mw + half * 64 + r, nn = nw + fg * 4 + ni * 16;
              if (m < p.M && nn < p.N) {
                const V4 rs = *(const V4*)((const T*)p.resid_lp + (long long)g * p.strideResid_g + rrow * p.ldc + nn);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rs[e];
              }
            }
            if (ACT == MK_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (ACT == MK_ACT_GELU) {
              v = gelu_erf4(v);
            }
          }
          const int cw = ni * 4 + fg;
          *(f32x4*)(wrow + ((cw ^ fr) << 4)) = v;
        }
      }
    };
    // producer side of the folded LayerNorm: the new rows as hi / lo planes + (sum, sum of squares) of the fp32 values
    // over this wave's 64 columns (one DPP row of 16 lanes holds one row segment); all lanes take part, invalid ones
    // contribute zeros.  fin: last block, fp32 rows out instead.
    constexpr bool fin = FIN;
    auto emit_row = [&](long long xrow, T* dh, T* dl, f32x4 x, bool ok, unsigned soff = 0xffffffffu) {
      float ssum = 0.f, qsum = 0.f;
      if (ok) {
        const uint2 oh = pack4<T>(x);
        const uint2 ol = pack4<T>(x - unpack4<T>(oh));
        quad_stats(x, ssum, qsum);
        if constexpr (SPLIT && EPI == MK_EPI_LS_RESIDUAL && sizeof(T) == 2) {
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{oh.x, oh.y}, rs_h, voff_b, soff, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{ol.x, ol.y}, rs_l, voff_b, soff, 0);
        } else {
          *(uint2*)dh = oh;
          *(uint2*)dl = ol;
        }
      }
      ssum = row16_sum(ssum);
      qsum = row16_sum(qsum);
      if (ok && c == 0) ((float2*)p.stats_out)[xrow * p.nslot_out + (nw >> 6)] = make_float2(ssum, qsum);
    };
    auto drain = [&](int half, int it0 = 0, int it1 = 16) {
      // conv output that feeds the next conv (split-operand mode: fp32 rows, re-split by mk_split_planes): bordered rows
      const bool bordf = CONV && p.bord_out;
      BorderedRow bwf;
      if (bordf) bwf.init(mw + half * 64 + it0 * 4 + rr, p.H, p.Wd);
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        if (it < it0 || it >= it1) continue;
        const int r = it * 4 + rr;
        const int m = mw + half * 64 + r;
        const int extraf = bordf ? bwf.extra : 0;
        if (bordf) bwf.step(4, p.H, p.Wd);
        // slice row (it & 7) * 4 + rr of the quarter it >> 3 (written by stageq(half, it >> 3))
        f32x4 val = *(const f32x4*)(((it & 4) ? wl1 : wl0) + ((it & 3) * 4 + rr) * 256 + ((c ^ ((it & 3) * 4 + rr)) << 4));
        if (SPLIT) val += bvd;
        const bool ok = SPLIT || (m < p.M && n < p.N);   // SPLIT: interior tiles only
        if (!ok) continue;
        if (EPI == MK_EPI_LS_RESIDUAL) {
          f32x4 x;
          if (SPLIT) {
            x = unpack4<T>(xh[half][it]) + unpack4<T>(xl[half][it]);
          } else {
            x = xr[half][it];
          }
          x += gm * val;
          if (SPLIT) x -= ((const float*)lnp)[wm * 128 + half * 64 + r];   // row centring (zeros when off): kernel prologue
          if (!SPLIT || fin) {
            if (ok) *(f32x4*)(p.out_f32 + (long long)m * p.ldc + n) = x;
          } else {
            if constexpr (sizeof(T) == 2) emit_row(m, nullptr, nullptr, x, ok, soff_b(half, it));
            else emit_row(m, plane_row(p.xh, half, it) + lane_off_st, plane_row(p.xl, half, it) + lane_off_st, x, ok);
          }
        } else if (EPI == MK_EPI_PATCH) {
          const int mc = ok ? m : 0;
          const int img = mc / p.npatch, tok = mc - img * p.npatch;
          const long long xrow = (long long)img * (p.npatch + 1) + 1 + tok;
          f32x4 x = val;
          if (ok) x += *(const f32x4*)(p.pos + (long long)(1 + tok) * p.N + n);
          if (SPLIT) emit_row(xrow, (T*)p.xh + xrow * p.ldxs + n, (T*)p.xl + xrow * p.ldxs + n, x, ok);
          else if (ok) *(f32x4*)(p.out_f32 + xrow * p.ldc + n) = x;
        } else if (EPI == MK_EPI_STORE && p.out_lo) {   // split-operand chain: the next contraction's (hi, lo) planes, 8 bytes per lane each
          const float ps = p.plane_scale;
          f16x4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sv = sat16(val[e] * ps, p.sat_flag);
            oh[e] = (_Float16)sv;
            ol[e] = (_Float16)(sv - (float)oh[e]);
          }
          const long long o = (long long)g * p.strideOut_g + ((long long)m + extraf) * p.ldc + n;
          *(f16x4*)((_Float16*)p.out_lp + o) = oh;
          *(f16x4*)((_Float16*)p.out_lo + o) = ol;
        } else {
          *(f32x4*)(p.out_f32 + (long long)g * p.strideOut_g + ((long long)m + extraf) * p.ldc + n) = val;
        }
      }
    };
    if (SPLIT && !FIN && EPI == MK_EPI_LS_RESIDUAL) {
      // the variant that also splits, sums and DPP-reduces what it stores: with both halves pre-loaded (128 registers
      // next to the 64 accumulators still waiting) hipcc spilled.  Quarters, two of them in flight (64 registers):
      // every load still has at least one quarter of draining to land.
      preload(0, 0, 16);
      stageq(0, 0);
      drain(0, 0, 8);
      preload(1, 0, 8);
      stageq(0, 1);
      drain(0, 8, 16);
      preload(1, 8, 16);
      stageq(1, 0);
      drain(1, 0, 8);
      stageq(1, 1);
      drain(1, 8, 16);
    } else {
      preload(0);
      stageq(0, 0);
      drain(0, 0, 8);
      stageq(0, 1);
      preload(1);
      drain(0, 8, 16);
      stageq(1, 0);
      drain(1, 0, 8);
      stageq(1, 1);
      drain(1, 8, 16);
    }
  }
}

// KIND selects which epilogues a kernel instantiation carries.  All epilogues of an instantiation are inlined into ONE
// register allocation: a rarely taken variant that needs more registers than the rest makes hipcc spill the accumulators of
// every tile of every launch (measured: +25 % on all GEMMs of the forward when the split-stream epilogue shared the kernel
// with the plain ones).  0 = the plain epilogues, 1 = folded-LayerNorm consumer (QKV / bias (+GELU) with row parameters),
// 2 = folded-LayerNorm producer (split residual stream), 3 = the same writing fp32 rows (last block): 2 and 3 together in
// one kernel spill again.
template <typename T, int KIND, bool CONV = false, int BN = 256>
__device__ __forceinline__ void epilogue_lds(const GemmParams& p, f32x4 (&acc)[8][4], char* wl0, char* wl1, int m0, int n0, int wm,
                                             int wn, int lane, int g, const float2* lnp = nullptr) {
  if constexpr (KIND == 2 || KIND == 3) {
    const bool interior = m0 + 256 <= p.M && n0 + BN <= p.N;   // workgroup-uniform
    if (KIND == 3) {   // LS_RESIDUAL with fp32 rows out (last block)
      if (interior) epilogue_lds_impl<T, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true, false, true, true>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g, lnp);
      else epilogue_impl<T, 8, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true, false, true>(p, acc, m0, n0, wm, wn, lane, g);
    } else if (p.epi == MK_EPI_PATCH) {
      if (interior) epilogue_lds_impl<T, MK_EPI_PATCH, MK_ACT_NONE, true, false, true>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g);
      else epilogue_impl<T, 8, MK_EPI_PATCH, MK_ACT_NONE, true, false, true>(p, acc, m0, n0, wm, wn, lane, g);
    } else {
      if (interior) epilogue_lds_impl<T, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true, false, true, false>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g, lnp);
      else epilogue_impl<T, 8, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true, false, true>(p, acc, m0, n0, wm, wn, lane, g);
    }
  } else if constexpr (KIND == 1) {
#if defined(MK_LN_ABL) && MK_LN_ABL == 2
    constexpr bool LNE = false;   // timing ablation (wrong results): prologue only, plain epilogue arithmetic
#else
    constexpr bool LNE = true;
#endif
    if (p.epi == MK_EPI_QKV) epilogue_lds_impl<T, MK_EPI_QKV, MK_ACT_NONE, true, LNE>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g, lnp);
    else if (p.act == MK_ACT_GELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_GELU, true, LNE>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g, lnp);
    else epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_NONE, true, LNE>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g, lnp);
  } else {
    switch (CONV ? MK_EPI_STORE : p.epi) {   // wave-uniform, once per output tile (a conv only ever stores)
      case MK_EPI_LS_RESIDUAL: epilogue_lds_impl<T, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g); break;
      case MK_EPI_QKV: epilogue_lds_impl<T, MK_EPI_QKV, MK_ACT_NONE, true>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g); break;
      case MK_EPI_PATCH: epilogue_lds_impl<T, MK_EPI_PATCH, MK_ACT_NONE, true>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g); break;
      default:
        if (!p.bias) {
          if (p.act == MK_ACT_RELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_RELU, false, false, false, false, CONV>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g);
          else if (!CONV && p.act == MK_ACT_GELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_GELU, false>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g);
          else epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_NONE, false, false, false, false, CONV>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g);
        } else {
          if (p.act == MK_ACT_RELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_RELU, true, false, false, false, CONV>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g);
          else if (!CONV && p.act == MK_ACT_GELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_GELU, true>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g);
          else epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_NONE, true, false, false, false, CONV>(p, acc, wl0, wl1, m0, n0, wm, wn, lane, g);
        }
    }
  }
}

// Tile order of the 256x256 kernels (on top of the XCD remap: an XCD walks a contiguous range of ids, 32 at a time).
//   PP_GM > 0  bands of PP_GM m-tiles walked n-major: the 32 tiles in flight form an 8 x 4 block of the output; the band's A
//              panels are fetched once per 4 n-tiles, the 4 W panels once per 32 tiles;
//   PP_GM < 0  groups of -PP_GM n-tiles walked m-major (n fastest): the group's W panels (-PP_GM x 256 x K) stay in the XCD's
//              L2 for the whole walk down M, every A panel is fetched once per GROUP -- less L2-miss traffic when N is several
//              groups wide and a group's W fits the 4-MB L2 (qkv, fc1: K = 1024, 512 KB per n-tile).
__device__ __forceinline__ void pp_tile_coords(int id, int ntm, int ntn, int PP_GM, int& tm, int& tn) {
  if (PP_GM < 0) {
    const int ng = -PP_GM;
    const int grp = id / (ntm * ng);
    const int rem = id - grp * (ntm * ng);
    const int gn = min(ng, ntn - grp * ng);
    tm = rem / gn;
    tn = grp * ng + rem - tm * gn;
    return;
  }
  const int band = id / (PP_GM * ntn);
  const int rem = id - band * (PP_GM * ntn);
  const int gm = min(PP_GM, ntm - band * PP_GM);
  tm = band * PP_GM + rem % gm;
  tn = rem / gm;
}

int num_cus();
extern int g_pp64_persist;
// schedule launchers (one translation unit each); amode = A_DENSE | A_CONV3, dtype = MK_BF16 | MK_F16
int launch_pp64(const GemmParams& p, int groups, int dtype, int amode, hipStream_t st, int band_m);
int launch_f32(const GemmParams& p, int groups, int amode, hipStream_t st);   // exact-fp32 parity mode (mk_gemm_f32.hip)

}  // namespace gemm
}  // namespace mk
