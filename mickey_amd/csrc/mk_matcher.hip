// mickey_amd -- dense descriptor matcher for gfx950 (reference utils/feature_matcher.py).
//
// dual-softmax (feature_matcher.py:64-83):  S = dsc0^T dsc1 / T, a learned scalar dustbin appended as
// last row / column / corner, P = softmax_rows(S+) * softmax_cols(S+), cropped back to n0 x n1.
// `final_scores` feeds a sampler and must match the fp32 reference to round-off; the (n0+1) x (n1+1) coupling
// matrix is never materialised.  Two correlation paths (AMD.MATCHER_CORR):
//   split (default for L2-normalised descriptors)  descriptors as split-fp16 planes (hi + lo, fp32-grade products on the
//           16-bit matrix cores); pass 1 = correlation -> maximum-free row / column sums only (lse_split_kernel); merge
//           (lse_final_kernel); pass 2 = the SAME correlation again -> scores / kp_scores / final_scores from registers
//           through LDS (dual_softmax_split_apply_kernel).  No stored correlation.
//   exact   v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain) for descriptors of any norm: pass 1 stores the scaled
//           correlation and online (max, sum) partials (lse_partial_kernel), pass 2 is element-wise, in place
//           (dual_softmax_apply_kernel).
// Sinkhorn (feature_matcher.py:93-137) and mutual-NN (:19-46) follow below.
#include <type_traits>

#include "mk_common.hpp"

namespace {
using namespace mk;

constexpr int MT = 64;      // tile edge
constexpr int NCHUNK = 4;   // column chunks of pass 1
constexpr int CMAX = 128;   // descriptor channels held in LDS

// stage dsc[b][c][i0 .. i0+63] (c < C) into s[c][64]; columns >= n are zero
__device__ __forceinline__ void stage_desc(float* s, const float* __restrict__ dsc, int C, int n, int i0) {
  for (int i = threadIdx.x; i < C * MT; i += blockDim.x) {
    const int c = i >> 6, col = i & 63;
    s[i] = (i0 + col < n) ? dsc[(long long)c * n + i0 + col] : 0.f;
  }
}

// 32x32 sub-tile of S = A^T B (A tile sA[c][64], B tile sB[c][64]): lane -> column jj = lane&31,
// rows (r&3) + 8*(r>>2) + 4*(lane>>5), r = 0..15
__device__ __forceinline__ f32x16 corr_tile(const float* sA, const float* sB, int C, int wi, int wj, int lane) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int l31 = lane & 31, hi = lane >> 5;
  const float* pa = sA + hi * MT + wi * 32 + l31;
  const float* pb = sB + hi * MT + wj * 32 + l31;
  for (int kk = 0; kk < C / 2; ++kk) {   // (C is a runtime value: the unroll is hipcc's call)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk * 2 * MT], pb[kk * 2 * MT], acc, 0, 0, 0);
  }
  return acc;
}

// ---- dual softmax: register-resident correlation ---------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 takes ONE float per lane per operand (lane l: row/column l & 31, k = 2 kk + (l >> 5)), so a
// wave keeps the descriptors of its 32 rows in 64 VGPRs for its whole life and streams 32-column tiles of the other
// image straight from L2 (descriptors are ~1 MB per image): 64 coalesced 4-byte loads + 64 MFMAs per tile, no LDS, no
// barriers.  (The previous version staged 64 KiB through LDS with scalar loads per 64x64 tile and ran at ~10 % of the
// fp32 matrix rate.)  Everything is kept in the log2 domain: v2 = S / T * log2(e), exp2 is one v_exp_f32.
constexpr int RT = 32;       // rows per wave, columns per streamed tile

__device__ __forceinline__ void lse2_merge(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  s = s * __builtin_amdgcn_exp2f(m - M) + s2 * __builtin_amdgcn_exp2f(m2 - M);
  m = M;
}

// A-side operand: this lane's 64 k-values of row i0 + (lane & 31); rows >= n are zero
template <bool FULLC>
__device__ __forceinline__ void load_operand(float (&a)[CMAX / 2], const float* __restrict__ d, int C, int n, int i, int hi) {
  const bool ok = i < n;
  const float* pa = d + (long long)hi * n + (ok ? i : n - 1);
#pragma unroll
  for (int kk = 0; kk < CMAX / 2; ++kk) {
    if (!FULLC && kk >= (C >> 1)) {
      a[kk] = 0.f;
    } else {
      const float v = pa[(long long)kk * 2 * n];
      a[kk] = ok ? v : 0.f;
    }
  }
}

// XCD-aware decode of a 1-D grid into (bx, by, unit): workgroups are dealt round-robin to the 8 XCDs, so the linear id
// is re-read as (xcd, slot) and ALL gx*gy workgroups of a unit (an image pair [x side]) land on one XCD, whose 4-MiB L2
// then holds that unit's ~2 MB of descriptors.  (Dealt naively, every XCD serves 8 pairs at a time, thrashes its L2 and
// pulls 1.9 GB of 128-byte pieces from memory per pass: measured 1.9 ms instead of 0.4.)  The grid is padded to a
// multiple of 8 units; returns false for padding.
// Y_FASTEST: consecutive workgroups of a unit walk the y index (the column chunks of the split passes) first: the workgroups
// in flight at one time then cover WHOLE rows of the output between them (kernels that write [rows, n1] matrices tile by tile:
// a wave's 256-byte row pieces meet their neighbours' in the same DRAM pages while those are open).
template <bool Y_FASTEST = false>
__device__ __forceinline__ bool decode_unit_grid(int gx, int gy, int nunits, int& bx, int& by, int& unit) {
  const int L = blockIdx.x, per = gx * gy;
  int within;
  if (nunits < 8) {   // too few units to give every XCD one: spread each unit over the whole chip instead
    unit = L / per;
    within = L - unit * per;
  } else {
    const int xcd = L & 7, slot = L >> 3;
    unit = (slot / per) * 8 + xcd;
    within = slot - (slot / per) * per;
  }
  if (Y_FASTEST) {
    by = within % gy;
    bx = within / gy;
  } else {
    bx = within % gx;
    by = within / gx;
  }
  return unit < nunits;
}

// Scheduling directive for the tile body (one basic block): all 64 operand loads first, then the 64 MFMAs.  Left alone,
// the scheduler keeps ONE operand register and emits load -> s_waitcnt vmcnt(0) -> MFMA, i.e. 64 serial memory round
// trips per tile (measured 32 us per tile instead of ~3).
#define MK_LOADS_THEN_MFMAS()                                  \
  do {                                                         \
    __builtin_amdgcn_sched_group_barrier(0x020, CMAX / 2, 0);  \
    __builtin_amdgcn_sched_group_barrier(0x002, CMAX / 2, 0);  \
    __builtin_amdgcn_sched_group_barrier(0x008, CMAX / 2, 0);  \
  } while (0)

template <bool FULLC>
__device__ __forceinline__ f32x16 corr_regs(const float (&a)[CMAX / 2], const float (&bq)[CMAX / 2], int C) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int kk = 0; kk < CMAX / 2; ++kk) {
    if (FULLC || kk < (C >> 1)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bq[kk], acc, 0, 0, 0);
  }
  return acc;
}

// pass 1.  grid (row blocks of 128, NCHUNK, B), one wave per 32 rows and one chunk of columns.  ONE correlation serves
// both softmax directions (round 1 ran it twice, once for S and once for S^T):
//   * rows of S: per-lane online (max, sum) over the columns this lane sees, merged across the 32 lanes at the end;
//     partr[((b*NCHUNK + chunk)*n0 + row)][2]
//   * columns of S: per tile, (max, sum) of this lane's column over the wave's 32 rows (16 in-lane values + one exchange
//     with lane ^ 32); every (row block, column) pair is produced by exactly one wave: partc[((b*nrb + rb)*n1 + j)][2]
//   * v2 = S / T * log2(e) itself is stored (vbuf [B, n0, n1]): pass 2 becomes an element-wise kernel instead of a second
//     correlation (the fp32 matrix pipe is the bound of this stage: 64 MFMAs x 64 cycles per 32 x 32 tile).
// amdgpu_waves_per_eu(2, 2): without it the scheduler chases occupancy, keeps ONE operand register and emits
// load -> s_waitcnt vmcnt(0) -> MFMA 64 times per tile (64 serial memory round trips, measured 32 us per tile)
template <bool FULLC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void lse_partial_kernel(const float* __restrict__ dsc0, const float* __restrict__ dsc1,
                                                          float scale2, float* __restrict__ partr, float* __restrict__ partc,
                                                          float* __restrict__ vbuf, int C, int n0, int n1, int nrb, int gx,
                                                          int nunits) {
  int bx, by, b;
  if (!decode_unit_grid(gx, NCHUNK, nunits, bx, by, b)) return;
  const float* dA = dsc0 + (long long)b * C * n0;
  const float* dB = dsc1 + (long long)b * C * n1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int rb = bx * 4 + wave, i0 = rb * RT;
  if (i0 >= n0) return;
  float a[CMAX / 2], bq[CMAX / 2];
  load_operand<FULLC>(a, dA, C, n0, i0 + l31, hi);
  const int ntB = (n1 + RT - 1) / RT;
  const int per = (ntB + NCHUNK - 1) / NCHUNK;
  const int jt0 = by * per, jt1 = min(ntB, jt0 + per);
  float rm[16], rs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { rm[r] = -1e30f; rs[r] = 0.f; }
  for (int jt = jt0; jt < jt1; ++jt) {
    const int j = jt * RT + l31;
    load_operand<FULLC>(bq, dB, C, n1, j, hi);
    const f32x16 acc = corr_regs<FULLC>(a, bq, C);
    MK_LOADS_THEN_MFMAS();
    float v[16];
    float cm = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      v[r] = acc[r] * scale2;
      if (i < n0) cm = fmaxf(cm, v[r]);
    }
    if (j < n1) {
      float cs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float M = fmaxf(rm[r], v[r]);
        rs[r] = rs[r] * __builtin_amdgcn_exp2f(rm[r] - M) + __builtin_amdgcn_exp2f(v[r] - M);
        rm[r] = M;
        if (i < n0) {
          cs += __builtin_amdgcn_exp2f(v[r] - cm);
          vbuf[((long long)b * n0 + i) * n1 + j] = v[r];
        }
      }
      // the other 16 rows of this column live in lane ^ 32
      const float m2 = __shfl_xor(cm, 32, 64), s2 = __shfl_xor(cs, 32, 64);
      lse2_merge(cm, cs, m2, s2);
      if (hi == 0) {
        float* o = partc + (((long long)b * nrb + rb) * n1 + j) * 2;
        o[0] = cm;
        o[1] = cs;
      }
    } else {
      (void)__shfl_xor(cm, 32, 64);   // keep the exchange wave-uniform
      (void)__shfl_xor(cm, 32, 64);
    }
  }
  // combine the 32 lanes that share rows (same hi)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(rm[r], o, 64), s2 = __shfl_xor(rs[r], o, 64);
      lse2_merge(rm[r], rs[r], m2, s2);
    }
  }
  if (l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < n0) {
        float* o = partr + (((long long)b * NCHUNK + by) * n0 + row) * 2;
        o[0] = rm[r];
        o[1] = rs[r];
      }
    }
  }
}

// merge the partials (+ the dustbin term): lse2[(b*2+side)*nmax + idx] = log2 sum 2^v2.  grid (nmax/256, 2, B);
// side 0: rows (NCHUNK column-chunk partials each), side 1: columns (nrb row-block partials each)
__global__ __launch_bounds__(256) void lse_final_kernel(const float* __restrict__ partr, const float* __restrict__ partc,
                                                        float* __restrict__ lse2, int use_dustbin, float beta2, int n0, int n1,
                                                        int nmax, int nrb, int nchunk) {
  const int idx = blockIdx.x * 256 + threadIdx.x, side = blockIdx.y, b = blockIdx.z;
  if (idx >= (side ? n1 : n0)) return;
  float m = -1e30f, s = 0.f;
  if (side == 0) {
    for (int c = 0; c < nchunk; ++c) {
      const float* q = partr + (((long long)b * nchunk + c) * n0 + idx) * 2;
      lse2_merge(m, s, q[0], q[1]);
    }
  } else {
    for (int rb = 0; rb < nrb; ++rb) {
      const float* q = partc + (((long long)b * nrb + rb) * n1 + idx) * 2;
      lse2_merge(m, s, q[0], q[1]);
    }
  }
  if (use_dustbin) lse2_merge(m, s, beta2, 1.0f);
  lse2[((long long)b * 2 + side) * nmax + idx] = m + __builtin_amdgcn_logf(s);   // v_log_f32 is log2
}

// pass 2, element-wise.  grid (column blocks of 256*VEC, n0, B): scores = softmax_rows * softmax_cols =
// 2^((v2 - lc2) + (v2 - lr2)), kp = scr0 (x) scr1, final = scores * kp.  `scores` or `fin` may alias vbuf (in place).
template <int VEC>
__global__ __launch_bounds__(256) void dual_softmax_apply_kernel(const float* vbuf, const float* __restrict__ scr0,
                                                                 const float* __restrict__ scr1, const float* __restrict__ lse2,
                                                                 float* scores, float* __restrict__ kp, float* fin, int n0, int n1,
                                                                 int nmax) {
  const int i = blockIdx.y, b = blockIdx.z;
  const int j = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (j >= n1) return;
  const float lr = lse2[((long long)b * 2 + 0) * nmax + i];
  const float s0 = scr0 ? scr0[(long long)b * n0 + i] : 0.f;
  const long long o = ((long long)b * n0 + i) * n1 + j;
  typedef float VT __attribute__((ext_vector_type(VEC)));
  const VT v = __builtin_nontemporal_load((const VT*)(vbuf + o));   // read once, overwritten in place
  const VT lc = *(const VT*)(lse2 + ((long long)b * 2 + 1) * nmax + j);
  VT s1, pr, kk, ff;
  if (scr1) s1 = *(const VT*)(scr1 + (long long)b * n1 + j);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float p = __builtin_amdgcn_exp2f((v[e] - lc[e]) + (v[e] - lr));
    const float k = scr1 ? s0 * s1[e] : 0.f;
    pr[e] = p;
    kk[e] = k;
    ff[e] = p * k;
  }
  if (scores) *(VT*)(scores + o) = pr;
  if (kp) *(VT*)(kp + o) = kk;
  if (fin) *(VT*)(fin + o) = ff;
}

// ---- dual softmax on the 16-bit matrix cores: split-fp16 correlation ------------------------------------------------
// BASELINE.json configs[4] names an "fp16 MFMA descriptor correlation"; `final_scores` feeds a sampler and must stay at fp32
// round-off of the reference's fp32 matmul (feature_matcher.py:65).  Both hold with SPLIT operands: every descriptor entry
// (|x| <= 1: L2-normalised descriptors, extractor_utils.py:6-10) is scaled by 2^10 (exact; keeps the low part out of fp16's
// subnormals) and written as hi = rn16(x'), lo = rn16(x' - hi): 22 mantissa bits.  x0.x1 = (hi0 + lo0)(hi1 + lo1) is
// evaluated as lo0.hi1 + hi0.lo1 + hi0.hi1 -- three v_mfma_f32_32x32x16_f16 passes, every product exact in fp32, fp32
// accumulation; the dropped lo.lo term is <= 2^-22 |x0||x1| (measured against the fp32 chain: tests/test_kernels_gpu.py).
// 24 MFMAs of 32 cycles per 32 x 32 tile instead of 64 of 64: the stage is no longer matrix-bound, so
//   pre    mk split planes: each image side as blocks of 32 keypoints in MFMA-operand order -- block = 8 K-steps x {hi, lo} x
//          (64 lanes x 16 B): a wave fetches a tile's operands with 16 fully coalesced 1-KiB loads;
//   pass 1 correlation -> row sums and per-32-row-block column sums of 2^v2 ONLY (no stored correlation: round 3 wrote and
//          re-read 480 MB at 32 pairs).  |v2| <= inv_T log2(e) is bounded for unit-norm descriptors, so the sums need no
//          running maximum: one v_exp_f32 per element;
//   merge  lse_final_kernel as for the exact path (partials carry max = 0);
//   pass 2 the SAME correlation again (bit-identical accumulators) -> scores, kp_scores, final_scores straight from registers.
constexpr int SP_KS = 8;        // K steps of 16 channels: C = 128
constexpr int SP_BLK_U4 = SP_KS * 2 * 64;   // uint4 per block of 32 keypoints (16 KiB)
constexpr int NCHUNK_S = 8;     // column chunks of the split passes (FIXED: the summation order of a row does not depend on B)
constexpr float SP_SCALE = 1024.0f;

// fp32 [B, C = 128, n] -> split planes [B, nblk, 8, 2, 64] x 16 B.  grid (nblk, B), 512 threads: thread = (K step, lane)
__global__ __launch_bounds__(512) void dsc_split_kernel(const float* __restrict__ dsc, uint4* __restrict__ planes, int n, int nblk) {
  const int blk = blockIdx.x, b = blockIdx.y;
  const int l = threadIdx.x & 63, st = threadIdx.x >> 6;
  const int j = blk * 32 + (l & 31), k0 = 16 * st + 8 * (l >> 5);
  const float* src = dsc + ((long long)b * 128 + k0) * n + j;
  f16x8 h, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = (j < n ? src[(long long)e * n] : 0.f) * SP_SCALE;
    const _Float16 xh = (_Float16)x;
    h[e] = xh;
    lo[e] = (_Float16)(x - (float)xh);
  }
  uint4* o = planes + ((long long)b * nblk + blk) * SP_BLK_U4 + (st * 2) * 64 + l;
  o[0] = __builtin_bit_cast(uint4, h);
  o[64] = __builtin_bit_cast(uint4, lo);
}

struct SplitOperand {
  uint4 h[SP_KS], l[SP_KS];
  __device__ __forceinline__ void load(const uint4* __restrict__ blk, int lane) {
#pragma unroll
    for (int st = 0; st < SP_KS; ++st) {
      h[st] = blk[(st * 2) * 64 + lane];
      l[st] = blk[(st * 2 + 1) * 64 + lane];
    }
  }
};

// S' = 2^20 x (32 rows of a) . (32 columns of b): cross terms first, the large term last
__device__ __forceinline__ f32x16 corr_split(const SplitOperand& a, const SplitOperand& b) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int st = 0; st < SP_KS; ++st)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.l[st]), __builtin_bit_cast(f16x8, b.h[st]), acc, 0, 0, 0);
#pragma unroll
  for (int st = 0; st < SP_KS; ++st)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.h[st]), __builtin_bit_cast(f16x8, b.l[st]), acc, 0, 0, 0);
#pragma unroll
  for (int st = 0; st < SP_KS; ++st)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.h[st]), __builtin_bit_cast(f16x8, b.h[st]), acc, 0, 0, 0);
  return acc;
}

// pass 1.  grid: decode_unit_grid(gx = row blocks / 4, NCHUNK_S, B); one wave per 32 rows and one chunk of column tiles.
// partr[((b*NCHUNK_S + chunk)*n0 + row)] = (0, sum_j 2^v2), partc[((b*nrb + rb)*n1 + j)] = (0, sum over the block's rows)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void lse_split_kernel(
    const uint4* __restrict__ P0, const uint4* __restrict__ P1, float scale2, float* __restrict__ partr, float* __restrict__ partc,
    int n0, int n1, int nrb, int ntb, int gx, int nunits) {
  int bx, by, b;
  if (!decode_unit_grid(gx, NCHUNK_S, nunits, bx, by, b)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int rb = bx * 4 + wave, i0 = rb * RT;
  if (rb >= nrb) return;
  SplitOperand a, bq;
  a.load(P0 + ((long long)b * nrb + rb) * SP_BLK_U4, lane);
  const int per = (ntb + NCHUNK_S - 1) / NCHUNK_S;
  const int jt0 = by * per, jt1 = min(ntb, jt0 + per);
  float rs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rs[r] = 0.f;
  // rows of this block that exist (the last block is padded with zero descriptors: 2^0 = 1 must not enter a column sum)
  unsigned rowok = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) rowok |= (i0 + (r & 3) + 8 * (r >> 2) + 4 * hi < n0 ? 1u : 0u) << r;
  const uint4* pb = P1 + (long long)b * ntb * SP_BLK_U4;
  auto tile = [&](const SplitOperand& bt, int jt) {
    const f32x16 acc = corr_split(a, bt);
    const int j = jt * RT + l31;
    const bool jok = j < n1;
    float cs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = jok ? __builtin_amdgcn_exp2f(acc[r] * scale2) : 0.f;
      rs[r] += e;
      cs += ((rowok >> r) & 1u) ? e : 0.f;
    }
    cs += __shfl_xor(cs, 32, 64);   // the other 16 rows of this column live in lane ^ 32
    if (hi == 0 && jok) {
      float* o = partc + (((long long)b * nrb + rb) * n1 + j) * 2;
      o[0] = 0.f;
      o[1] = cs;
    }
  };
  // (a second operand set in flight -- the loads of tile jt + 1 in front of the MFMAs of tile jt -- needs 128 more registers
  // than two waves per SIMD leave: hipcc spills; as written it issues the next tile's loads as the current tile's operand
  // registers die, and the other wave of the SIMD covers the rest of the L2 round trip)
  for (int jt = jt0; jt < jt1; ++jt) {
    bq.load(pb + (long long)jt * SP_BLK_U4, lane);
    tile(bq, jt);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {   // the 32 lanes that share rows (same hi): a fixed butterfly
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) rs[r] += __shfl_xor(rs[r], o, 64);
  }
  if (l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < n0) {
        float* o = partr + (((long long)b * NCHUNK_S + by) * n0 + row) * 2;
        o[0] = 0.f;
        o[1] = rs[r];
      }
    }
  }
}

// pass 2: the same correlation, outputs from registers -- through LDS.  In the accumulator layout a lane owns ONE column of
// the tile: stored from there an instruction writes 4 bytes per lane, 128 contiguous bytes per output row (round 4's first
// version: 1.9 TB/s, 0.75 ms for the three 480-MB outputs of 32 pairs -- the vector memory path, not HBM, was the limit).
// Here a wave computes TWO column tiles (32 rows x 64 columns), transposes each output through an 8-KiB private LDS slice
// (wave-local, LDS is in-order per wave: no barrier) and writes 16 bytes per lane, 256 contiguous bytes per row.
// Round 5 built the writer round 4's notes asked for -- the 4 (8) waves of a workgroup on ONE row block, their 32 x 64 blocks
// staged in a shared LDS tile whose rows are shifted by the row's offset inside its 128-byte line, stored as whole aligned
// lines (9 lines touched per 8 written instead of 3 per 2), two tiles alternating, one barrier per output: commit 5 of round 5,
// profiles/r05a_matcher_whole_line_writer.txt -- and it is SLOWER (stage 0.80 ms against 0.69 full, 0.54 against 0.44 lean):
// half of this kernel's time is the latency of re-computing the correlation (three dependent L2 round trips per wave for 48
// MFMAs), which eight independent waves per CU overlap with each other's stores and four barrier-coupled ones do not.
// scores = 2^((v2 - lc2) + (v2 - lr2)), kp = scr0 (x) scr1, final = scores kp.
template <int VEC>   // floats per lane of an output store: 4 when the output rows are 16-byte aligned (n1 % 4 == 0), 2 for even n1
                     // (n1 = 1938, the Map-free grid), else 1
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dual_softmax_split_apply_kernel(
    const uint4* __restrict__ P0, const uint4* __restrict__ P1, float scale2, const float* __restrict__ scr0,
    const float* __restrict__ scr1, const float* __restrict__ lse2, float* __restrict__ scores, float* __restrict__ kp,
    float* __restrict__ fin, int n0, int n1, int nmax, int nrb, int ntb, int gx, int nunits, int nchunk) {
  __shared__ __attribute__((aligned(16))) float stage[4][RT * 64];
  int bx, by, b;
  if (!decode_unit_grid<true>(gx, nchunk, nunits, bx, by, b)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int rb = bx * 4 + wave, i0 = rb * RT;
  if (rb >= nrb) return;
  float* st = stage[wave];
  SplitOperand a, bq;
  a.load(P0 + ((long long)b * nrb + rb) * SP_BLK_U4, lane);
  const int per = (ntb + nchunk - 1) / nchunk;
  const int jt0 = by * per, jt1 = min(ntb, jt0 + per);
  float lr[16], s0[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    const int ic = i < n0 ? i : n0 - 1;
    lr[r] = lse2[((long long)b * 2 + 0) * nmax + ic];
    s0[r] = scr0 ? scr0[(long long)b * n0 + ic] : 0.f;
  }
  const uint4* pb = P1 + (long long)b * ntb * SP_BLK_U4;
  typedef float VT __attribute__((ext_vector_type(VEC)));
  constexpr int LPR = 64 / VEC, RPI = 64 / LPR;   // lanes per 64-column row, rows per store instruction
  const int dr = lane / LPR, dc = (lane % LPR) * VEC;
  for (int jt = jt0; jt < jt1; jt += 2) {
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bq.load(pb + (long long)min(jt + t, ntb - 1) * SP_BLK_U4, lane);
      acc[t] = corr_split(a, bq);
    }
    float lc[2], s1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = (jt + t) * RT + l31;
      const int jc = j < n1 ? j : n1 - 1;
      lc[t] = lse2[((long long)b * 2 + 1) * nmax + jc];
      s1[t] = scr1 ? scr1[(long long)b * n1 + jc] : 0.f;
    }
    const int jbase = jt * RT;   // first column of the pair of tiles; columns of tile jt + 1 past jt1 / n1 are never stored
    const int jlim = min(n1, min(jt + 2, jt1) * RT);
#pragma unroll 1
    for (int which = 0; which < 3; ++which) {   // 0 scores, 1 kp_scores, 2 final_scores: one LDS round trip each
      float* out = which == 0 ? scores : which == 1 ? kp : fin;
      if (!out) continue;   // wave-uniform
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r] * scale2;
          const float pv = __builtin_amdgcn_exp2f((v - lc[t]) + (v - lr[r]));
          const float kv = s0[r] * s1[t];
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
          st[row * 64 + t * 32 + l31] = which == 0 ? pv : which == 1 ? kv : pv * kv;
        }
#pragma unroll 8   // (VEC = 1 has 32 iterations: all of them in flight at once spill)
      for (int it = 0; it < RT / RPI; ++it) {
        const int row = dr + RPI * it, i = i0 + row, j = jbase + dc;
        const VT val = *(const VT*)(st + row * 64 + dc);
        if (i < n0 && j + VEC <= jlim) *(VT*)(out + ((long long)b * n0 + i) * n1 + j) = val;   // (jlim is a multiple of VEC)
      }
    }
  }
}

// ---- sinkhorn ---------------------------------------------------------------------------------------
// log_optimal_transport (feature_matcher.py:89-123) in the log2 domain: Z2 = Z log2(e), u2 = u log2(e), v2 = v log2(e), so that
// every exponential / logarithm is one v_exp_f32 / v_log_f32.  The bound is HBM: 2 passes over the (n0+1) x (n1+1) coupling
// matrix per iteration (20 LSE passes + 1 write = 301 MB per pair at 540x720).  Rows are padded to a multiple of 4
// floats (ldz; the pad holds -1e30 = contributes 0) so that both passes move 16 bytes per lane; the running maximum of an online
// log-sum-exp is updated per BLOCK of values (4 per lane in the row pass, 8 rows in the column pass): 1.25 / 1.125 exponentials
// per element instead of 2; the column pass is split over 256-row chunks (2900 workgroups at 1280x720 instead of 150) whose
// partial (max, sum) pairs a small kernel merges.
constexpr int SK_RCH = 256;   // rows per chunk of the column pass

// Z2[(n0+1) x ldz] = couplings * log2(e) (S / sqrt(C), alpha on the last row / column / corner, -1e30 in the row pad)
__global__ __launch_bounds__(256) void couplings_kernel(const float* __restrict__ dsc0, const float* __restrict__ dsc1,
                                                        float scale2, float alpha2, float* __restrict__ Z, int C, int n0,
                                                        int n1, int ldz) {
  __shared__ __attribute__((aligned(16))) float sA[CMAX * MT];
  __shared__ __attribute__((aligned(16))) float sB[CMAX * MT];
  const int b = blockIdx.z, i0 = blockIdx.y * MT, j0 = blockIdx.x * MT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wi = wave >> 1, wj = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  stage_desc(sA, dsc0 + (long long)b * C * n0, C, n0, i0);
  stage_desc(sB, dsc1 + (long long)b * C * n1, C, n1, j0);
  __syncthreads();
  const f32x16 acc = corr_tile(sA, sB, C, wi, wj, lane);
  const int jx = j0 + wj * 32 + l31;
  if (jx >= ldz) return;
  float* Zb = Z + (long long)b * (n0 + 1) * ldz;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (i > n0) continue;
    Zb[(long long)i * ldz + jx] = jx > n1 ? -1e30f : ((i == n0 || jx == n1) ? alpha2 : acc[r] * scale2);
  }
}

// u2[i] = log_mu2[i] - LSE2_j(Z2[i][j] + v2[j]); one wave per row, 16 bytes per lane
template <bool NT>
__global__ __launch_bounds__(256) void sink_row_kernel(const float* __restrict__ Z, const float* __restrict__ v,
                                                       float* __restrict__ u, int n0, int ldz, int ldu, float norm2, float last2) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i > n0) return;
  const f32x4* z = (const f32x4*)(Z + ((long long)b * (n0 + 1) + i) * ldz);
  const f32x4* vb = (const f32x4*)(v + (long long)b * ldz);
  float m = -1e30f, s = 0.f;
  for (int j4 = lane; j4 < (ldz >> 2); j4 += 64) {
    const f32x4 x = (NT ? __builtin_nontemporal_load(z + j4) : z[j4]) + vb[j4];
    const float M = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), m);
    s = s * __builtin_amdgcn_exp2f(m - M) + ((__builtin_amdgcn_exp2f(x[0] - M) + __builtin_amdgcn_exp2f(x[1] - M)) +
                                             (__builtin_amdgcn_exp2f(x[2] - M) + __builtin_amdgcn_exp2f(x[3] - M)));
    m = M;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    lse2_merge(m, s, m2, s2);
  }
  if (lane == 0) u[(long long)b * ldu + i] = (i == n0 ? last2 : norm2) - (m + __builtin_amdgcn_logf(s));
}

// column pass, part 1: (max, sum) of 2^(Z2[i][j] + u2[i]) over the rows of one 256-row chunk for 256 columns; a wave takes 64 of
// the rows, 8 at a time (one running-maximum update per 8 values of a column), a lane 4 adjacent columns
template <bool NT>
__global__ __launch_bounds__(256) void sink_col_part_kernel(const float* __restrict__ Z, const float* __restrict__ u,
                                                            float2* __restrict__ part, int n0, int ldz, int ldu, int nrc) {
  __shared__ f32x4 sm[4][64], ss[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.z, rc = blockIdx.y;
  const int j = (blockIdx.x * 64 + lane) * 4;
  const bool jok = j < ldz;
  const float* Zb = Z + (long long)b * (n0 + 1) * ldz + (jok ? j : 0);
  const float* ub = u + (long long)b * ldu;
  f32x4 m = f32x4{-1e30f, -1e30f, -1e30f, -1e30f}, sacc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r0 = rc * SK_RCH + wave * 64;
  for (int r = r0; r < r0 + 64 && r <= n0; r += 8) {
    f32x4 x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = r + k;
      // rows past the end repeat the last row with -inf added: no branch in the loads
      const int ic = i <= n0 ? i : n0;
      const float ui = i <= n0 ? ub[ic] : -1e30f;
      const f32x4* zp = (const f32x4*)(Zb + (long long)ic * ldz);
      x[k] = (NT ? __builtin_nontemporal_load(zp) : *zp) + ui;
    }
    f32x4 M = m;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) M[e] = fmaxf(M[e], x[k][e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += __builtin_amdgcn_exp2f(x[k][e] - M[e]);
      sacc[e] = sacc[e] * __builtin_amdgcn_exp2f(m[e] - M[e]) + t;
    }
    m = M;
  }
  sm[wave][lane] = m;
  ss[wave][lane] = sacc;
  __syncthreads();
  if (wave == 0 && jok) {
#pragma unroll
    for (int g = 1; g < 4; ++g) {
      const f32x4 m2 = sm[g][lane], s2 = ss[g][lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float me = m[e], se = sacc[e];
        lse2_merge(me, se, m2[e], s2[e]);
        m[e] = me;
        sacc[e] = se;
      }
    }
    float2* o = part + ((long long)b * nrc + rc) * ldz + j;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = make_float2(m[e], sacc[e]);
  }
}

// column pass, part 2: v2[j] = log_nu2[j] - log2 of the merged chunk partials
__global__ __launch_bounds__(256) void sink_col_fin_kernel(const float2* __restrict__ part, float* __restrict__ v, int n1, int ldz,
                                                           int nrc, float norm2, float last2) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ldz) return;
  float m = -1e30f, s = 0.f;
  for (int rc = 0; rc < nrc; ++rc) {
    const float2 p = part[((long long)b * nrc + rc) * ldz + j];
    lse2_merge(m, s, p.x, p.y);
  }
  v[(long long)b * ldz + j] = j > n1 ? 0.f : (j == n1 ? last2 : norm2) - (m + __builtin_amdgcn_logf(s));
}

__global__ __launch_bounds__(256) void sink_final_kernel(const float* __restrict__ Z, const float* __restrict__ u,
                                                         const float* __restrict__ v, float norm2,
                                                         const float* __restrict__ scr0, const float* __restrict__ scr1,
                                                         float* __restrict__ out, float* __restrict__ kp,
                                                         float* __restrict__ fin, int n0, int n1, int ldz, int ldu) {
  const int b = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n1) return;
  const float z = Z[((long long)b * (n0 + 1) + i) * ldz + j];
  const float pr = __builtin_amdgcn_exp2f(z + u[(long long)b * ldu + i] + v[(long long)b * ldz + j] - norm2);
  const long long o = ((long long)b * n0 + i) * n1 + j;
  if (out) out[o] = pr;
  if (scr0) {
    const float kk = scr0[(long long)b * n0 + i] * scr1[(long long)b * n1 + j];
    if (kp) kp[o] = kk;
    if (fin) fin[o] = pr * kk;
  }
}

__global__ void fill_kernel(float* p, float v, long long n) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- mutual nearest neighbours ----------------------------------------------------------------------
// row arg-max over columns [0, n1-1): one wave per row; first index wins ties
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ sc, int* __restrict__ arg,
                                                         float* __restrict__ val, int n0, int n1) {
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n0 - 1) return;
  const float* row = sc + ((long long)b * n0 + i) * n1;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < n1 - 1; j += 64) {
    const float x = row[j];
    if (x > bv) { bv = x; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(bv, o, 64);
    const int i2 = __shfl_xor(bi, o, 64);
    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
  }
  if (lane == 0) { arg[(long long)b * n0 + i] = bi; val[(long long)b * n0 + i] = bv; }
}

// column arg-max over rows [0, n0-1): block = 64 columns x 4 row groups
__global__ __launch_bounds__(256) void col_argmax_kernel(const float* __restrict__ sc, int* __restrict__ arg, int n0, int n1) {
  __shared__ float sv[4][64];
  __shared__ int si[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6, b = blockIdx.y;
  const int j = blockIdx.x * 64 + col;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  if (j < n1 - 1) {
    for (int i = rg; i < n0 - 1; i += 4) {
      const float x = sc[((long long)b * n0 + i) * n1 + j];
      if (x > bv) { bv = x; bi = i; }
    }
  }
  sv[rg][col] = bv;
  si[rg][col] = bi;
  __syncthreads();
  if (rg == 0 && j < n1 - 1) {
#pragma unroll
    for (int g = 1; g < 4; ++g) {
      const float v2 = sv[g][col];
      const int i2 = si[g][col];
      if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
    }
    arg[(long long)b * n1 + j] = bi;
  }
}

// one workgroup per pair: mutual check, compaction, bitonic sort by score (descending; ties: lower row first)
__global__ __launch_bounds__(1024) void mutual_collect_kernel(const int* __restrict__ rarg, const float* __restrict__ rval,
                                                              const int* __restrict__ carg, int* __restrict__ matches,
                                                              int* __restrict__ count, int n0, int n1, int npow2) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // [npow2]
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    unsigned long long key = 0ull;  // sorts last
    if (i < n0 - 1) {
      const int j = rarg[(long long)b * n0 + i];
      const float v = rval[(long long)b * n0 + i];
      // valid0 = mutual & (exp(max) > min_conf = 0): always true for finite scores (feature_matcher.py:29-30)
      if (j >= 0 && j < n1 - 1 && carg[(long long)b * n1 + j] == i && expf(v) > 0.f) {
        unsigned int u = __float_as_uint(v);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving map
        key = ((unsigned long long)u << 32) | (unsigned int)(0xffffffffu - (unsigned)i);
      }
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        const int ixj = i ^ jj;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  int local = 0;
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    const unsigned long long key = keys[i];
    if (key != 0ull && i < n0) {
      const int row = (int)(0xffffffffu - (unsigned int)(key & 0xffffffffu));
      matches[((long long)b * n0 + i) * 2 + 0] = row;
      matches[((long long)b * n0 + i) * 2 + 1] = rarg[(long long)b * n0 + row];
      ++local;
    }
  }
  __shared__ int scount;
  if (threadIdx.x == 0) scount = 0;
  __syncthreads();
  atomicAdd(&scount, local);
  __syncthreads();
  if (threadIdx.x == 0) count[b] = scount;
}

}  // namespace

static int g_ds_chunk2 = 0;   // dual softmax (split path): column chunks of pass 2 per row block (0 = one pair of tiles per wave)
static int g_sk_group = 0;   // Sinkhorn: pairs iterated together (0: the whole batch per pass with non-temporal reads)

extern "C" {

long long mk_dual_softmax_work_floats(int B, int n0, int n1, int own_copy) {
  const long long nmax = n0 > n1 ? n0 : n1, nrb = (n0 + RT - 1) / RT;
  // row partials + column partials + final log2-sum-exp vectors (+ the stored correlation: only a call that asks for
  // neither `scores` nor `final_scores` keeps it in `work`; at B = 32, n = 1938 that term alone is 480 MB)
  return (long long)B * NCHUNK * n0 * 2 + (long long)B * nrb * n1 * 2 + (long long)B * 2 * nmax + 4 +
         (own_copy ? (long long)B * n0 * n1 : 0);
}

int mk_dual_softmax(const float* dsc0, const float* dsc1, const float* scr0, const float* scr1, float inv_temperature,
                    int use_dustbin, float dustbin, float* scores, float* kp_scores, float* final_scores, float* work, int B,
                    int C, int n0, int n1, mk_stream_t stream) {
  MK_CHECK_ARG(dsc0 && dsc1 && work, "mk_dual_softmax: null pointer");
  MK_CHECK_ARG(B > 0 && n0 > 0 && n1 > 0 && C > 0 && C <= CMAX && C % 2 == 0, "mk_dual_softmax: need 0 < C <= %d, C even", CMAX);
  MK_CHECK_ARG((scr0 && scr1) || (!kp_scores && !final_scores), "mk_dual_softmax: kp/final scores need scr0 and scr1");
  hipStream_t st = (hipStream_t)stream;
  const int nmax = n0 > n1 ? n0 : n1, nrb = (n0 + RT - 1) / RT;
  const float LOG2E = 1.4426950408889634f;
  const float scale2 = inv_temperature * LOG2E;
  float* partr = work;
  float* partc = partr + (long long)B * NCHUNK * n0 * 2;
  float* lse2 = partc + (long long)B * nrb * n1 * 2;
  float* vown = lse2 + (((long long)B * 2 * nmax + 3) & ~3LL);   // 16-byte aligned when `work` is
  // in place when the caller wants `scores` or `final_scores` anyway (pass 2 reads an element, then overwrites it)
  float* vbuf = scores ? scores : final_scores ? final_scores : vown;
  const int gx1 = (n0 + 4 * RT - 1) / (4 * RT);
  const dim3 g1((unsigned)gx1 * NCHUNK * ((B + 7) / 8 * 8));   // see decode_unit_grid
  if (C == CMAX)
    hipLaunchKernelGGL(lse_partial_kernel<true>, g1, dim3(256), 0, st, dsc0, dsc1, scale2, partr, partc, vbuf, C, n0, n1, nrb, gx1, B);
  else
    hipLaunchKernelGGL(lse_partial_kernel<false>, g1, dim3(256), 0, st, dsc0, dsc1, scale2, partr, partc, vbuf, C, n0, n1, nrb, gx1, B);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(lse_final_kernel, dim3((nmax + 255) / 256, 2, B), dim3(256), 0, st, partr, partc, lse2, use_dustbin,
                     dustbin * LOG2E, n0, n1, nmax, nrb, NCHUNK);
  MK_CHECK_LAUNCH();
  if (scores || kp_scores || final_scores) {
    // rows of vbuf / outputs start at multiples of n1 floats: 8-byte vectors need n1 even and 8-byte-aligned bases
    const bool v2 = (n1 % 2 == 0) && ((((uintptr_t)vbuf | (uintptr_t)scores | (uintptr_t)kp_scores | (uintptr_t)final_scores |
                                        (uintptr_t)scr1 | (uintptr_t)lse2) & 7) == 0) && (nmax % 2 == 0);
    if (v2)
      hipLaunchKernelGGL(dual_softmax_apply_kernel<2>, dim3((n1 / 2 + 255) / 256, n0, B), dim3(256), 0, st, vbuf, scr0, scr1, lse2,
                         scores, kp_scores, final_scores, n0, n1, nmax);
    else
      hipLaunchKernelGGL(dual_softmax_apply_kernel<1>, dim3((n1 + 255) / 256, n0, B), dim3(256), 0, st, vbuf, scr0, scr1, lse2, scores,
                         kp_scores, final_scores, n0, n1, nmax);
    MK_CHECK_LAUNCH();
  }
  return MK_OK;
}

long long mk_dual_softmax_split_work_floats(int B, int n0, int n1) {
  const long long nmax = n0 > n1 ? n0 : n1, nrb = (n0 + RT - 1) / RT, ntb = (n1 + RT - 1) / RT;
  // the two split-plane images (16 KiB per block of 32 keypoints) + row / column partials + log2-sum-exp vectors
  return (long long)B * (nrb + ntb) * SP_BLK_U4 * 4 + (long long)B * NCHUNK_S * n0 * 2 + (long long)B * nrb * n1 * 2 +
         (long long)B * 2 * nmax + 8;
}

int mk_dual_softmax_split(const float* dsc0, const float* dsc1, const float* scr0, const float* scr1, float inv_temperature,
                          int use_dustbin, float dustbin, float* scores, float* kp_scores, float* final_scores, float* work,
                          int B, int C, int n0, int n1, mk_stream_t stream) {
  MK_CHECK_ARG(dsc0 && dsc1 && work, "mk_dual_softmax_split: null pointer");
  MK_CHECK_ARG(B > 0 && n0 > 0 && n1 > 0 && C == 16 * SP_KS, "mk_dual_softmax_split: C must be %d (use mk_dual_softmax otherwise)", 16 * SP_KS);
  MK_CHECK_ARG((scr0 && scr1) || (!kp_scores && !final_scores), "mk_dual_softmax_split: kp/final scores need scr0 and scr1");
  MK_CHECK_ARG(((uintptr_t)work & 15) == 0, "mk_dual_softmax_split: work must be 16-byte aligned");
  const float LOG2E = 1.4426950408889634f;
  // unit-norm descriptors: |v2| <= inv_T log2(e); the maximum-free sums of pass 1 need 2^v2 and n 2^v2 inside fp32
  MK_CHECK_ARG(inv_temperature > 0.f && inv_temperature * LOG2E <= 100.f,
               "mk_dual_softmax_split: temperature %g too small for the maximum-free sums (use mk_dual_softmax)", 1.0 / inv_temperature);
  hipStream_t st = (hipStream_t)stream;
  const int nmax = n0 > n1 ? n0 : n1, nrb = (n0 + RT - 1) / RT, ntb = (n1 + RT - 1) / RT;
  uint4* P0 = (uint4*)work;
  uint4* P1 = P0 + (long long)B * nrb * SP_BLK_U4;
  float* partr = (float*)(P1 + (long long)B * ntb * SP_BLK_U4);
  float* partc = partr + (long long)B * NCHUNK_S * n0 * 2;
  float* lse2 = partc + (long long)B * nrb * n1 * 2;
  hipLaunchKernelGGL(dsc_split_kernel, dim3(nrb, B), dim3(512), 0, st, dsc0, P0, n0, nrb);
  hipLaunchKernelGGL(dsc_split_kernel, dim3(ntb, B), dim3(512), 0, st, dsc1, P1, n1, ntb);
  MK_CHECK_LAUNCH();
  const float scale2 = inv_temperature * LOG2E / (SP_SCALE * SP_SCALE);
  const int gx = (nrb + 3) / 4;
  const dim3 g((unsigned)gx * NCHUNK_S * ((B + 7) / 8 * 8));   // see decode_unit_grid
  hipLaunchKernelGGL(lse_split_kernel, g, dim3(256), 0, st, P0, P1, scale2, partr, partc, n0, n1, nrb, ntb, gx, B);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(lse_final_kernel, dim3((nmax + 255) / 256, 2, B), dim3(256), 0, st, partr, partc, lse2, use_dustbin,
                     dustbin * LOG2E, n0, n1, nmax, nrb, NCHUNK_S);
  MK_CHECK_LAUNCH();
  if (scores || kp_scores || final_scores) {
    // pass 2 is a WRITER: the fewer output rows the chip has in flight at a time, the more of its 256-byte row pieces meet open
    // DRAM pages.  g_ds_chunk2 column chunks per row block (dev knob; 0 = one pair of column tiles per wave: the smallest window)
    const int nchunk2 = g_ds_chunk2 > 0 ? g_ds_chunk2 : (ntb + 1) / 2;
    const dim3 g2((unsigned)gx * nchunk2 * ((B + 7) / 8 * 8));
    const bool a16 = ((((uintptr_t)scores | (uintptr_t)kp_scores | (uintptr_t)final_scores) & 15) == 0);
    if ((n1 & 3) == 0 && a16)
      hipLaunchKernelGGL(dual_softmax_split_apply_kernel<4>, g2, dim3(256), 0, st, P0, P1, scale2, scr0, scr1, lse2, scores, kp_scores,
                         final_scores, n0, n1, nmax, nrb, ntb, gx, B, nchunk2);
    else if ((n1 & 1) == 0 && ((((uintptr_t)scores | (uintptr_t)kp_scores | (uintptr_t)final_scores) & 7) == 0))
      hipLaunchKernelGGL(dual_softmax_split_apply_kernel<2>, g2, dim3(256), 0, st, P0, P1, scale2, scr0, scr1, lse2, scores, kp_scores,
                         final_scores, n0, n1, nmax, nrb, ntb, gx, B, nchunk2);
    else
      hipLaunchKernelGGL(dual_softmax_split_apply_kernel<1>, g2, dim3(256), 0, st, P0, P1, scale2, scr0, scr1, lse2, scores, kp_scores,
                         final_scores, n0, n1, nmax, nrb, ntb, gx, B, nchunk2);
    MK_CHECK_LAUNCH();
  }
  return MK_OK;
}

static inline int sk_ldz(int n1) { return (n1 + 1 + 3) & ~3; }
static inline int sk_ldu(int n0) { return (n0 + 1 + 3) & ~3; }
static inline int sk_nrc(int n0) { return (n0 + 1 + SK_RCH - 1) / SK_RCH; }

long long mk_sinkhorn_work_floats(int B, int n0, int n1) {
  // Z2 (rows padded to 16 bytes) + u2 + v2 + the column pass' chunk partials (max, sum)
  const long long ldz = sk_ldz(n1);
  return (long long)B * ((long long)(n0 + 1) * ldz + sk_ldu(n0) + ldz + 2LL * sk_nrc(n0) * ldz);
}

int mk_sinkhorn(const float* dsc0, const float* dsc1, const float* scr0, const float* scr1, float alpha, int iters, float* scores,
                float* kp_scores, float* final_scores, float* work, int B, int C, int n0, int n1, mk_stream_t stream) {
  MK_CHECK_ARG(dsc0 && dsc1 && work && (scores || final_scores), "mk_sinkhorn: null pointer");
  MK_CHECK_ARG((scr0 && scr1) || (!kp_scores && !final_scores), "mk_sinkhorn: kp/final scores need scr0 and scr1");
  MK_CHECK_ARG(B > 0 && n0 > 0 && n1 > 0 && C > 0 && C <= CMAX && C % 2 == 0 && iters >= 0, "mk_sinkhorn: bad args");
  MK_CHECK_ARG(((uintptr_t)work & 15) == 0, "mk_sinkhorn: work must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int ldz = sk_ldz(n1), ldu = sk_ldu(n0), nrc = sk_nrc(n0);
  float* Z = work;
  float* u = Z + (long long)B * (n0 + 1) * ldz;
  float* v = u + (long long)B * ldu;
  float2* part = (float2*)(v + (long long)B * ldz);
  // constants of log_optimal_transport (feature_matcher.py:116-118), times log2(e)
  const float LOG2E = 1.4426950408889634f;
  const float norm = -logf((float)n0 + (float)n1);
  const float norm2 = norm * LOG2E, mu_last2 = (logf((float)n1) + norm) * LOG2E, nu_last2 = (logf((float)n0) + norm) * LOG2E;
  // The 20 LSE passes of a pair re-read its (n0+1) x ldz coupling matrix (15 MB at 540x720, 86 MB at 1280x720).  Round 4
  // measured whether cutting the batch into groups of pairs whose matrices fit the 256-MB Infinity Cache together -- a group
  // runs ALL its iterations before the next starts, reads without the non-temporal hint -- turns HBM passes into cache hits:
  // it does not (8 pairs of 1280x720: 5.49 ms batch-wide, 5.64 / 5.64 / 6.09 ms in groups of 4 / 2 / 1,
  // profiles/r04c_bench_matcher.txt).  The batch-wide order stays the default; g_sk_group > 0 (dev knob
  // mk_sinkhorn_set_group) selects the grouped one.
  const long long zbytes = (long long)(n0 + 1) * ldz * 4;
  int group = g_sk_group > 0 ? g_sk_group : B;
  if (group > B) group = B;
  const bool nt = g_sk_group <= 0 || (long long)group * zbytes > (240LL << 20);
  for (int b0 = 0; b0 < B; b0 += group) {
    const int nb = group < B - b0 ? group : B - b0;
    const float* d0 = dsc0 + (long long)b0 * C * n0;
    const float* d1 = dsc1 + (long long)b0 * C * n1;
    float* Zg = Z + (long long)b0 * (n0 + 1) * ldz;
    float* ug = u + (long long)b0 * ldu;
    float* vg = v + (long long)b0 * ldz;
    float2* pg = part + (long long)b0 * nrc * ldz;
    hipLaunchKernelGGL(couplings_kernel, dim3((ldz + MT - 1) / MT, (n0 + 1 + MT - 1) / MT, nb), dim3(256), 0, st, d0, d1,
                       LOG2E / sqrtf((float)C), alpha * LOG2E, Zg, C, n0, n1, ldz);
    MK_CHECK_LAUNCH();
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long long)nb * ldu + 255) / 256)), dim3(256), 0, st, ug, 0.f, (long long)nb * ldu);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long long)nb * ldz + 255) / 256)), dim3(256), 0, st, vg, 0.f, (long long)nb * ldz);
    MK_CHECK_LAUNCH();
    for (int it = 0; it < iters; ++it) {
      if (nt) {
        hipLaunchKernelGGL(sink_row_kernel<true>, dim3((n0 + 1 + 3) / 4, nb), dim3(256), 0, st, Zg, vg, ug, n0, ldz, ldu, norm2, mu_last2);
        hipLaunchKernelGGL(sink_col_part_kernel<true>, dim3((ldz / 4 + 63) / 64, nrc, nb), dim3(256), 0, st, Zg, ug, pg, n0, ldz, ldu, nrc);
      } else {
        hipLaunchKernelGGL(sink_row_kernel<false>, dim3((n0 + 1 + 3) / 4, nb), dim3(256), 0, st, Zg, vg, ug, n0, ldz, ldu, norm2, mu_last2);
        hipLaunchKernelGGL(sink_col_part_kernel<false>, dim3((ldz / 4 + 63) / 64, nrc, nb), dim3(256), 0, st, Zg, ug, pg, n0, ldz, ldu, nrc);
      }
      hipLaunchKernelGGL(sink_col_fin_kernel, dim3((ldz + 255) / 256, nb), dim3(256), 0, st, pg, vg, n1, ldz, nrc, norm2, nu_last2);
    }
    MK_CHECK_LAUNCH();
    hipLaunchKernelGGL(sink_final_kernel, dim3((n1 + 255) / 256, n0, nb), dim3(256), 0, st, Zg, ug, vg, norm2,
                       scr0 ? scr0 + (long long)b0 * n0 : nullptr, scr1 ? scr1 + (long long)b0 * n1 : nullptr,
                       scores ? scores + (long long)b0 * n0 * n1 : nullptr, kp_scores ? kp_scores + (long long)b0 * n0 * n1 : nullptr,
                       final_scores ? final_scores + (long long)b0 * n0 * n1 : nullptr, n0, n1, ldz, ldu);
    MK_CHECK_LAUNCH();
  }
  return MK_OK;
}

int mk_dual_softmax_set_chunks(int chunks) {
  g_ds_chunk2 = chunks;
  return MK_OK;
}

int mk_sinkhorn_set_group(int pairs) {
  g_sk_group = pairs;
  return MK_OK;
}

int mk_mutual_nn(const float* scores, int* matches, int* count, int* work, int B, int n0, int n1, mk_stream_t stream) {
  MK_CHECK_ARG(scores && matches && count && work && B > 0 && n0 > 1 && n1 > 1, "mk_mutual_nn: bad args");
  int npow2 = 1;
  while (npow2 < n0) npow2 <<= 1;
  MK_CHECK_ARG((size_t)npow2 * 8 <= 128 * 1024, "mk_mutual_nn: n0=%d too large", n0);
  hipStream_t st = (hipStream_t)stream;
  int* rarg = work;
  float* rval = (float*)(work + (long long)B * n0);
  int* carg = work + 2LL * B * n0;
  hipLaunchKernelGGL(row_argmax_kernel, dim3((n0 + 3) / 4, B), dim3(256), 0, st, scores, rarg, rval, n0, n1);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(col_argmax_kernel, dim3((n1 + 63) / 64, B), dim3(256), 0, st, scores, carg, n0, n1);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(mutual_collect_kernel, dim3(B), dim3(1024), (size_t)npow2 * 8, st, rarg, rval, carg, matches, count, n0, n1,
                     npow2);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // extern "C"
