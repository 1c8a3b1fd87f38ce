// mickey_amd -- flash-style multi-head attention forward for gfx950, head_dim 64, non-causal.
// Replaces q@k^T -> softmax -> @v of reference DINO_modules/layers/attention.py:53-59 (the xformers
// path :72-76 is the same maths).  The ntok x ntok score matrix never leaves registers.
//
// Design (wave64, v_mfma_f32_32x32x16):
//  * workgroup = 4 waves of one (image, head); each wave owns 32 or 64 queries (QB sub-blocks of 32).
//  * four variants share this design (mk_attn_set_mode); attn_fwd_lean_kernel strips the softmax to ~70 VALU
//    instructions per tile (max folded into the accumulator init, row sums on the matrix pipe).
//  * K tile [64 keys][64 d] and V^T tile [64 d][64 keys] go HBM -> LDS with global_load_lds
//    (lane-linear image; XOR swizzle on the source address + on the ds_read_b128), double-buffered.
//  * S^T = K.Q^T ("swapped" product): after the MFMA a lane holds 32 scores of ONE query, so the
//    row max / row sum are register-local plus one lane<->lane+32 exchange.
//  * O^T = V^T.P^T: the P^T B-operand is exactly the registers the lane already holds (converted
//    to 16 bit) -- no cross-lane movement -- because mk_gemm_qkv stores V^T with token bits 2<->3
//    swapped, which makes the 8 keys a lane owns per k-step one contiguous 16-byte LDS chunk.
//  * q arrives pre-multiplied by 64^-0.5*log2(e); softmax uses exp2.
#include <type_traits>

#include "mk_common.hpp"

namespace mk {
// exact-fp32 parity mode (mk_attention_f32.hip)
void launch_attn_f32(const float* q, const float* k, const float* vt, float* out, int ldo, int nimg, int heads, int ntok,
                     int ntok_pad, hipStream_t st);
// one wave per SIMD, 64 queries per wave, hand-placed MFMA / softmax interleave (mk_attention_w1.hip)
// -> false when the problem is outside its range (fewer than 4 KV tiles): the caller runs the 64-query kernel instead
bool launch_attn_w1(const void* q, const void* k, const void* vt, void* out, int ldo, int nimg, int heads, int ntok, int ntok_pad,
                    int dtype, hipStream_t st);
}

#ifndef MK_ATTN_DEFAULT_BIG
#define MK_ATTN_DEFAULT_BIG 2   // kernel of large grids under mk_attn_set_mode(0): 2 = 64 q/wave, two waves per SIMD; 7 = one wave per SIMD
#endif

namespace {
using namespace mk;

constexpr int KV_TILE_BYTES = 64 * 64 * 2;  // 8 KiB
constexpr float ATT_REBASE_THR = 8.0f;    // lean softmax: the running maximum is re-based when a tile exceeds it by 2^8

// XCD-aware decode of the workgroup id.  Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest), so
// with the natural (query block, head, image) grid the 16 query blocks of one (image, head) land on all 8 XCDs and every
// XCD's private L2 streams the K / V^T of EVERY head from HBM: rocprofv3 FETCH_SIZE 2.2 GB per launch against 0.76 GB
// of q, k, v.  Remapped, the workgroups an XCD runs back to back are the query blocks of one (image, head): its 0.5 MB
// of K / V^T is fetched once and then hit in that XCD's L2 by the other 15 blocks.
struct AttnBlock {
  int qblk, head, img;
};
__device__ __forceinline__ AttnBlock attn_block() {
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int r = xcd_remap(lin, gridDim.x * gridDim.y * gridDim.z);
  const int hg = r / gridDim.x;
  return AttnBlock{r - hg * (int)gridDim.x, hg % (int)gridDim.y, hg / (int)gridDim.y};
}

// QB = 32-query sub-blocks per wave (1 or 2).  With QB = 2 every K / V^T fragment read from LDS feeds two MFMAs
// and a workgroup covers 256 queries per staged K/V tile (half the LDS-DMA, ds_read and barrier work per MFMA).
template <typename T, int QB>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ vt, T* __restrict__ out, int ldo, int heads,
                                                          int ntok, int ntok_pad) {
  using V8 = typename Lp<T>::V8;
  using V4 = typename Lp<T>::V4;
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // [stage][K | Vt]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlock ab = attn_block();
  const int head = ab.head, img = ab.img;
  const long long hb = (long long)img * heads + head;
  const T* Qh = q + hb * ntok_pad * 64;
  const T* Kh = k + hb * ntok_pad * 64;
  const T* Vh = vt + hb * 64 * ntok_pad;
  const int q0 = ab.qblk * (128 * QB) + wave * (32 * QB);
  const int j = lane & 31, hi = lane >> 5;
  // 1939 tokens are padded to a multiple of the query block: in the last block whole waves hold no valid query (1/32 of
  // all waves at 256 queries per block); they skip the MFMA / softmax work
  const bool wave_has_queries = q0 < ntok;

  V8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qrow = q0 + qb * 32 + j;
    qrow = qrow < ntok_pad ? qrow : ntok_pad - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const V8*)(Qh + (long long)qrow * 64 + ks * 16 + hi * 8);
  }

  const int srow = lane >> 3, sp = lane & 7;
  auto stage = [&](int buf, int kt) {
    char* sK = smem + buf * 2 * KV_TILE_BYTES;
    char* sV = sK + KV_TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ii = wave * 2 + t;
      const int r = ii * 8 + srow;
      glds16(Kh + (long long)(kt * 64 + r) * 64 + swz8(r, sp) * 8, sK + ii * 1024);
      glds16(Vh + (long long)r * ntok_pad + kt * 64 + swz8(r, sp) * 8, sV + ii * 1024);
    }
  };

  f32x16 o[QB][2];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) o[qb][0][i] = o[qb][1][i] = 0.f;
    m_run[qb] = -1e30f;
    l_run[qb] = 0.f;
  }

  const int nkt = (ntok + 63) >> 6;
  stage(0, 0);
  // One KV tile.  LAST is a compile-time tag: only the peeled final tile carries the key mask -- written as a runtime
  // condition inside one loop it was if-converted into 32 v_cmp + 32 v_cndmask per query block on EVERY tile (~12 % of
  // the VALU work of a loop that is VALU-bound).
  auto kv_tile = [&](const int kt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!LAST) stage((kt + 1) & 1, kt + 1);
    if (!wave_has_queries) return;   // pad-only wave of the last query block: stages K/V and keeps the barriers, nothing else
    const char* sK = smem + (kt & 1) * 2 * KV_TILE_BYTES;
    const char* sV = sK + KV_TILE_BYTES;

    f32x16 s[QB][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[qb][kb][i] = 0.f;
      const int row = kb * 32 + j;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 kf = *(const V8*)(sK + row * 128 + swz8(row, ks * 2 + hi) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][kb] = Lp<T>::mma32(kf, qf[qb][ks], s[qb][kb]);
      }
    }
    V8 pf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (LAST && (ntok & 63)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= ntok) s[qb][kb][r] = -1e30f;
          }
      }
      // max over the lane's 32 scores as a shallow tree of 3-input maxima (v_max3_f32), not a 31-deep chain
      float t8[8];
#pragma unroll
      for (int r = 0; r < 8; ++r)
        t8[r] = fmaxf(fmaxf(s[qb][0][r], s[qb][0][r + 8]), fmaxf(s[qb][1][r], s[qb][1][r + 8]));
      float mx = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), t8[2]), fmaxf(fmaxf(t8[3], t8[4]), t8[5]));
      mx = fmaxf(fmaxf(mx, t8[6]), t8[7]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // raw v_exp_f32 (exp2f() would add ~5 range-fixup instructions per element; arguments here are <= 0 and
      // results below the normal range may flush to zero, which is what a softmax tail wants anyway)
      const float m_new = fmaxf(m_run[qb], mx);
      const bool grew = m_new > m_run[qb];
      float rs4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent partial sums: no 32-deep dependent add chain
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[qb][kb][r] - m_new);
          s[qb][kb][r] = pv;
          rs4[r & 3] += pv;
        }
      const float rs = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
      if (__any(grew)) {   // wave-uniform: after the first tiles the running max rarely moves -> no O rescale
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          o[qb][0][i] *= alpha;
          o[qb][1][i] *= alpha;
        }
      }
      m_run[qb] = m_new;
      l_run[qb] += rs;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[qb][s4][e] = (T)s[qb][s4 >> 1][(s4 & 1) * 8 + e];
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int row = dt * 32 + j;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const V8 vf = *(const V8*)(sV + row * 128 + swz8(row, s4 * 2 + hi) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[qb][dt] = Lp<T>::mma32(vf, pf[qb][s4], o[qb][dt]);
      }
    }
  };
  for (int kt = 0; kt < nkt - 1; ++kt) kv_tile(kt, std::false_type{});
  kv_tile(nkt - 1, std::true_type{});

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + qb * 32 + j;
    if (qi < ntok) {
      T* orow = out + ((long long)img * ntok + qi) * ldo + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          V4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (T)(o[qb][dt][r4 * 4 + e] * inv);
          *(V4*)(orow + dt * 32 + r4 * 8 + hi * 4) = w;
        }
    }
  }
}

// Production kernel (mode 9 / automatic): QB sub-blocks of 32 queries per wave, the lean softmax of attn_fwd_lean_kernel with
// the row sums on the VALU.  What decides between variants on this part is ENERGY per tile, not cycles: the chip runs every
// attention kernel (and the GEMM) at its power limit -- profiles/r03_pmc_clock_attention_gemm.json: variants that need fewer
// cycles run at a lower clock and finish at the same wall time -- so the kernel that issues the least work wins:
//   * no subtraction of the running maximum (folded into the MFMA accumulator init), no per-tile rescale of O (re-base only
//     when a tile exceeds the maximum by 2^8), 16 MFMAs per 32 x 64 tile (the matrix-pipe row sums of the lean kernel were a
//     fifth of its matrix work: 805-825 -> 885 TFLOP/s when they went back to 32 fp32 adds);
//   * QB = 2: every K / V^T fragment read from LDS feeds two MFMAs and a workgroup covers 256 queries per staged tile.
template <typename T, int QB>
__global__ __launch_bounds__(256, QB == 2 ? 2 : 3) void attn_fwd_fold_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                              const T* __restrict__ vt, T* __restrict__ out,
                                                                              int ldo, int heads, int ntok, int ntok_pad) {
  using V8 = typename Lp<T>::V8;
  using V4 = typename Lp<T>::V4;
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // [stage][K | Vt]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlock ab = attn_block();
  const int head = ab.head, img = ab.img;
  const long long hb = (long long)img * heads + head;
  const T* Qh = q + hb * ntok_pad * 64;
  const T* Kh = k + hb * ntok_pad * 64;
  const T* Vh = vt + hb * 64 * ntok_pad;
  const int q0 = ab.qblk * (128 * QB) + wave * (32 * QB);
  const int j = lane & 31, hi = lane >> 5;
  const bool wave_has_queries = q0 < ntok;   // pad-only waves of the last query block stage K/V and keep the barriers, nothing else

  V8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qrow = q0 + qb * 32 + j;
    qrow = qrow < ntok_pad ? qrow : ntok_pad - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const V8*)(Qh + (long long)qrow * 64 + ks * 16 + hi * 8);
  }
  const int srow = lane >> 3, sp = lane & 7;
  auto stage = [&](int buf, int kt) {
    char* sK = smem + buf * 2 * KV_TILE_BYTES;
    char* sV = sK + KV_TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ii = wave * 2 + t;
      const int r = ii * 8 + srow;
      glds16(Kh + (long long)(kt * 64 + r) * 64 + swz8(r, sp) * 8, sK + ii * 1024);
      glds16(Vh + (long long)r * ntok_pad + kt * 64 + swz8(r, sp) * 8, sV + ii * 1024);
    }
  };

  f32x16 o[QB][2], negm[QB];
  float m_run[QB], l_run[QB];   // l_run: this lane's share of the row sum (lanes j and j + 32 are added at the end)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      o[qb][0][i] = o[qb][1][i] = 0.f;
      negm[qb][i] = 0.f;      // m = 0 to start with; the first tile re-bases
    }
    m_run[qb] = l_run[qb] = 0.f;
  }
  bool first = true;

  const int nkt = (ntok + 63) >> 6;
  stage(0, 0);
  auto kv_tile = [&](const int kt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;   // compile-time: only the peeled final tile carries the key mask
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!LAST) stage((kt + 1) & 1, kt + 1);
    if (!wave_has_queries) return;
    const char* sK = smem + (kt & 1) * 2 * KV_TILE_BYTES;
    const char* sV = sK + KV_TILE_BYTES;

    f32x16 s[QB][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = kb * 32 + j;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 kf = *(const V8*)(sK + row * 128 + swz8(row, ks * 2 + hi) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][kb] = Lp<T>::mma32(kf, qf[qb][ks], ks == 0 ? negm[qb] : s[qb][kb]);   // S' = K.Q^T - m
      }
    }
    V8 pf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (LAST && (ntok & 63)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= ntok) s[qb][kb][r] = -1e30f;
          }
      }
      float t8[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) t8[r] = fmaxf(fmaxf(s[qb][0][r], s[qb][0][r + 8]), fmaxf(s[qb][1][r], s[qb][1][r + 8]));
      float mx = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), t8[2]), fmaxf(fmaxf(t8[3], t8[4]), t8[5]));
      mx = fmaxf(fmaxf(mx, t8[6]), t8[7]);
      if (__any(mx > ATT_REBASE_THR) || first) {   // wave-uniform, rare after the first tile: re-base m to this tile's maximum
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float shift = first ? mx : fmaxf(mx, 0.f);     // never lower m after the first tile
        const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          o[qb][0][i] *= alpha;
          o[qb][1][i] *= alpha;
          s[qb][0][i] -= shift;
          s[qb][1][i] -= shift;
        }
        l_run[qb] *= alpha;
        m_run[qb] += shift;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[qb][i] = -m_run[qb];
      }
      float rs4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent partial sums: no 32-deep dependent add chain
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pv = __builtin_amdgcn_exp2f(s[qb][s4 >> 1][(s4 & 1) * 8 + e]);
          pf[qb][s4][e] = (T)pv;
          rs4[e & 3] += pv;
        }
      l_run[qb] += (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
    }
    first = false;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int row = dt * 32 + j;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const V8 vf = *(const V8*)(sV + row * 128 + swz8(row, s4 * 2 + hi) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[qb][dt] = Lp<T>::mma32(vf, pf[qb][s4], o[qb][dt]);
      }
    }
  };
  for (int kt = 0; kt < nkt - 1; ++kt) kv_tile(kt, std::false_type{});
  kv_tile(nkt - 1, std::true_type{});

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.0f / (l_run[qb] + __shfl_xor(l_run[qb], 32, 64));
    const int qi = q0 + qb * 32 + j;
    if (qi < ntok) {
      T* orow = out + ((long long)img * ntok + qi) * ldo + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          V4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (T)(o[qb][dt][r4 * 4 + e] * inv);
          *(V4*)(orow + dt * 32 + r4 * 8 + hi * 4) = w;
        }
    }
  }
}

// VALU-lean variant (32 queries per wave).  A wave64 VALU instruction costs ~4 issue cycles and a 32-cycle MFMA hides
// only a handful of them, so the softmax (~170 VALU instructions per 16 MFMAs) bounds the kernels above.  Here:
//  * the running maximum is folded into the QK^T accumulator init: S' = K.Q^T + (-m) comes out of the MFMA already
//    shifted (a persistent 16-register vector holds -m; no per-element subtraction);
//  * m is only re-based when a tile's maximum exceeds it by more than 2^8 (then O and the row sums are rescaled);
//    otherwise P = exp2(S') <= 256 is used as is -- the common case after the first tile;
//  * the row sums are computed on the matrix pipe (ones . P^T, 4 extra MFMAs per tile) instead of 32 VALU adds; they
//    sum the same 16-bit P that multiplies V, and need no cross-lane exchange.
// Per tile and wave: 20 MFMAs and ~70 VALU instructions (16 max3, 32 exp, 16 cvt).

// MSUM: row sums on the matrix pipe (ones . P^T); false: as 32 fp32 adds per tile on the VALU.  The part runs this kernel at
// its POWER limit (profiles/r03_pmc_clock.json: every attention variant ends at the same wall time, the ones that need fewer
// cycles at a lower clock), where what counts is energy per tile: the 4 row-sum MFMAs are a fifth of the matrix work.
template <typename T, bool MSUM>
__global__ __launch_bounds__(256, 3) void attn_fwd_lean_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ vt, T* __restrict__ out, int ldo,
                                                               int heads, int ntok, int ntok_pad) {
  using V8 = typename Lp<T>::V8;
  using V4 = typename Lp<T>::V4;
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // [stage][K | Vt]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlock ab = attn_block();
  const int head = ab.head, img = ab.img;
  const long long hb = (long long)img * heads + head;
  const T* Qh = q + hb * ntok_pad * 64;
  const T* Kh = k + hb * ntok_pad * 64;
  const T* Vh = vt + hb * 64 * ntok_pad;
  const int q0 = ab.qblk * 128 + wave * 32;
  const int j = lane & 31, hi = lane >> 5;

  V8 qf[4];
  {
    int qrow = q0 + j;
    qrow = qrow < ntok_pad ? qrow : ntok_pad - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const V8*)(Qh + (long long)qrow * 64 + ks * 16 + hi * 8);
  }
  V8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;

  const int srow = lane >> 3, sp = lane & 7;
  auto stage = [&](int buf, int kt) {
    char* sK = smem + buf * 2 * KV_TILE_BYTES;
    char* sV = sK + KV_TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {   // 8 one-KiB pieces per operand tile, 2 per wave
      const int ii = wave * 2 + t;
      const int r = ii * 8 + srow;
      glds16(Kh + (long long)(kt * 64 + r) * 64 + swz8(r, sp) * 8, sK + ii * 1024);
      glds16(Vh + (long long)r * ntok_pad + kt * 64 + swz8(r, sp) * 8, sV + ii * 1024);
    }
  };

  f32x16 o[2], lsum, negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    o[0][i] = o[1][i] = 0.f;
    lsum[i] = 0.f;
    negm[i] = 0.f;          // m_run = 0 to start with; the first tile re-bases (scores are bounded by |q||k|)
  }
  float m_run = 0.f, l_run = 0.f;   // l_run: this lane's share of the row sum (VALU form; lanes j and j + 32 are added at the end)
  bool first = true;

  const int nkt = (ntok + 63) >> 6;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    if (q0 >= ntok) continue;   // pad-only wave of the last query block: stages K/V and keeps the barriers, nothing else
    const char* sK = smem + (kt & 1) * 2 * KV_TILE_BYTES;
    const char* sV = sK + KV_TILE_BYTES;

    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = kb * 32 + j;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 kf = *(const V8*)(sK + row * 128 + swz8(row, ks * 2 + hi) * 16);
        s[kb] = Lp<T>::mma32(kf, qf[ks], ks == 0 ? negm : s[kb]);   // S' = K.Q^T - m
      }
    }
    if (kt == nkt - 1 && (ntok & 63)) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= ntok) s[kb][r] = -1e30f;
        }
    }
    float t8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) t8[r] = fmaxf(fmaxf(s[0][r], s[0][r + 8]), fmaxf(s[1][r], s[1][r + 8]));
    float mx = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), t8[2]), fmaxf(fmaxf(t8[3], t8[4]), t8[5]));
    mx = fmaxf(fmaxf(mx, t8[6]), t8[7]);
    if (__any(mx > ATT_REBASE_THR) || first) {   // wave-uniform, rare after the first tile: re-base m to this tile's maximum
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float shift = first ? mx : fmaxf(mx, 0.f);     // never lower m after the first tile
      const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] *= alpha;
        o[1][i] *= alpha;
        if (MSUM) lsum[i] *= alpha;
        s[0][i] -= shift;
        s[1][i] -= shift;
      }
      l_run *= alpha;
      m_run += shift;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = -m_run;
      first = false;
    }
    V8 pf[4];
    float rs4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent partial sums: no 32-deep dependent add chain
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pv = __builtin_amdgcn_exp2f(s[s4 >> 1][(s4 & 1) * 8 + e]);
        pf[s4][e] = (T)pv;
        if (!MSUM) rs4[e & 3] += pv;
      }
    if (MSUM) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) lsum = Lp<T>::mma32(ones, pf[s4], lsum);   // row sums on the matrix pipe
    } else {
      l_run += (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int row = dt * 32 + j;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const V8 vf = *(const V8*)(sV + row * 128 + swz8(row, s4 * 2 + hi) * 16);
        o[dt] = Lp<T>::mma32(vf, pf[s4], o[dt]);
      }
    }
  }

  // MSUM: every accumulator row of ones.P^T holds the full row sum of this lane's query
  const float inv = 1.0f / (MSUM ? lsum[0] : l_run + __shfl_xor(l_run, 32, 64));
  const int qi = q0 + j;
  if (qi < ntok) {
    T* orow = out + ((long long)img * ntok + qi) * ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        V4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (T)(o[dt][r4 * 4 + e] * inv);
        *(V4*)(orow + dt * 32 + r4 * 8 + hi * 4) = w;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Ping-pong kernel (mode 8): 8 waves = two wave-rows of 4, 32 queries per wave, two waves per SIMD -- one of each row.
// Why: with two free-running waves per SIMD the per-tile barrier starts both of them in the SAME phase (both issue their
// QK^T MFMAs, then both run their softmax on the VALU, then both their P.V MFMAs), so the matrix pipe and the VALU /
// transcendental unit of a SIMD are never busy at the same time: the kernels above take MFMA time + VALU time per tile
// (~1400 cycles per 32-query x 64-key wave-tile = 640 + ~700).  One wave per SIMD does not fix it either
// (mk_attention_w1.hip): a lone wave issues a v_exp_f32 only every ~16 cycles (two waves: ~9), and at head_dim 64 there
// are 1.6 exponentials per MFMA.  Here the two rows run HALF A TILE APART, as the ping-pong GEMM's wave-rows do:
//     slot 2t   : row 0  M(t) = { O += V(t-1).P(t-1), l += 1.P(t-1), S'(t) = K(t).Q^T - m }   (20 MFMAs, matrix pipe)
//                 row 1  X(t-1) = { max S', (rare) re-base, P = exp2(S') -> 16 bit }          (VALU / transcendental)
//     slot 2t+1 : row 0  X(t) ,  row 1  M(t)
// with a workgroup barrier at every slot boundary.  K(u+1) and V(u) are fetched by LDS-DMA (one K piece and one V piece per
// wave, SGPR-addressed, invisible to hipcc's vmcnt) at the start of the even slot 2u and waited for at the end of slot
// 2u+1; K(t) is read in slots 2t, 2t+1 and V(t) in slots 2t+2, 2t+3, so two buffers of each are enough.
template <typename T>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const T* __restrict__ vt, T* __restrict__ out, int ldo, int heads,
                                                             int ntok, int ntok_pad) {
  using V8 = typename Lp<T>::V8;
  using V4 = typename Lp<T>::V4;
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // K[2] | Vt[2]
  char* const sKb = smem;
  char* const sVb = smem + 2 * KV_TILE_BYTES;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = wave >> 2;
  const AttnBlock ab = attn_block();
  const int head = ab.head, img = ab.img;
  const long long hb = (long long)img * heads + head;
  const T* Qh = q + hb * ntok_pad * 64;
  const T* Kh = k + hb * ntok_pad * 64;
  const T* Vh = vt + hb * 64 * ntok_pad;
  const int q0 = ab.qblk * 256 + wave * 32;
  const int j = lane & 31, hi = lane >> 5;
  const int nkt = (ntok + 63) >> 6;

  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 qraw[4];
  {
    int qrow = q0 + j;
    qrow = qrow < ntok_pad ? qrow : ntok_pad - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qraw[ks] = *(const u32x4*)(Qh + (long long)qrow * 64 + ks * 16 + hi * 8);
  }
  // the only loads hipcc knows about: waited for HERE (at their first use inside the loop it would emit a vmcnt(0) per
  // iteration, which also drains the LDS-DMA pieces it does not know about)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]) :: "memory");
  V8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(V8, qraw[ks]);
  V8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;

  // LDS-DMA: 8 one-KiB pieces per operand tile, one per wave
  const int srow = lane >> 3, sp = lane & 7;
  const int pr = wave * 8 + srow;
  const unsigned voffK = (unsigned)((pr * 64 + swz8(pr, sp) * 8) * (int)sizeof(T));
  const unsigned voffV = (unsigned)((pr * ntok_pad + swz8(pr, sp) * 8) * (int)sizeof(T));
  auto dma_k = [&](int tile) { glds16_sv(Kh + (long long)tile * 4096, voffK, sKb + (tile & 1) * KV_TILE_BYTES + wave * 1024); };
  auto dma_v = [&](int tile) { glds16_sv(Vh + tile * 64, voffV, sVb + (tile & 1) * KV_TILE_BYTES + wave * 1024); };
  // start of the even slot 2u: K(u+1) and V(u) go out (every wave its piece)
  auto dma_even = [&](int u) {
    if (u + 1 < nkt) dma_k(u + 1);
    if (u < nkt) dma_v(u);
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); };
  auto bar_landed = [&]() {   // end of an odd slot: this wave's pieces of the next even slot's tiles have landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  f32x16 o[2], lsum, negm, s[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    o[0][i] = o[1][i] = 0.f;
    lsum[i] = 0.f;
    negm[i] = 0.f;          // m = 0 to start with; the first tile re-bases
    s[0][i] = s[1][i] = 0.f;
  }
  V8 pf[4];
  float m_run = 0.f;
  bool first = true;

  // M(t): the matrix-pipe half of tile t (P.V and the row sums of tile t-1, S' of tile t)
  auto mphase = [&](int t) {
    __builtin_amdgcn_s_setprio(1);
    if (t > 0) {
      const char* sV = sVb + ((t - 1) & 1) * KV_TILE_BYTES;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int r = dt * 32 + j;
          const V8 vf = *(const V8*)(sV + r * 128 + swz8(r, s4 * 2 + hi) * 16);
          o[dt] = Lp<T>::mma32(vf, pf[s4], o[dt]);
        }
        lsum = Lp<T>::mma32(ones, pf[s4], lsum);   // row sums on the matrix pipe
      }
    }
    if (t < nkt) {
      const char* sK = sKb + (t & 1) * KV_TILE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int r = kb * 32 + j;
          const V8 kf = *(const V8*)(sK + r * 128 + swz8(r, ks * 2 + hi) * 16);
          s[kb] = Lp<T>::mma32(kf, qf[ks], ks == 0 ? negm : s[kb]);   // S' = K.Q^T - m
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // X(t): the VALU half of tile t
  auto xphase = [&](int t) {
    if (t == nkt - 1 && (ntok & 63)) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= ntok) s[kb][r] = -1e30f;
        }
    }
    float t8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) t8[r] = fmaxf(fmaxf(s[0][r], s[0][r + 8]), fmaxf(s[1][r], s[1][r + 8]));
    float mx = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), t8[2]), fmaxf(fmaxf(t8[3], t8[4]), t8[5]));
    mx = fmaxf(fmaxf(mx, t8[6]), t8[7]);
    if (__any(mx > ATT_REBASE_THR) || first) {   // wave-uniform, rare after the first tile
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float shift = first ? mx : fmaxf(mx, 0.f);     // never lower m after the first tile
      const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] *= alpha;
        o[1][i] *= alpha;
        lsum[i] *= alpha;
        s[0][i] -= shift;
        s[1][i] -= shift;
      }
      m_run += shift;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = -m_run;
      first = false;
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[s4][e] = (T)__builtin_amdgcn_exp2f(s[s4 >> 1][(s4 & 1) * 8 + e]);
  };

  // prologue: K(0)
  dma_k(0);
  bar_landed();
  if (row == 0) {
    for (int t = 0; t < nkt; ++t) {
      dma_even(t);      // slot 2t
      mphase(t);
      bar();
      xphase(t);        // slot 2t+1
      bar_landed();
    }
    dma_even(nkt);      // slot 2 nkt (nothing left to fetch: keeps the code symmetric)
    mphase(nkt);
    bar();
  } else {
    dma_even(0);        // slot 0: this row idles, its share of the pieces goes out all the same
    bar();
    for (int t = 0; t < nkt; ++t) {
      mphase(t);        // slot 2t+1
      bar_landed();
      dma_even(t + 1);  // slot 2t+2
      xphase(t);
      bar();
    }
    mphase(nkt);        // slot 2 nkt + 1
  }

  const float inv = 1.0f / lsum[0];   // every accumulator row of ones.P^T holds the full row sum of this lane's query
  const int qi = q0 + j;
  if (qi < ntok) {
    T* orow = out + ((long long)img * ntok + qi) * ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        V4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (T)(o[dt][r4 * 4 + e] * inv);
        *(V4*)(orow + dt * 32 + r4 * 8 + hi * 4) = w;
      }
  }
}

int g_attn_mode = 0;   // mk_attn_set_mode: 0 automatic, 1: 32 q/wave, 2: 64 q/wave, 4: VALU-lean, 7: one wave per SIMD (mk_attention_w1.hip)

template <typename T>
void launch_attn(const void* q, const void* k, const void* vt, void* out, int ldo, int nimg, int heads, int ntok, int ntok_pad,
                 hipStream_t st) {
  const long long blocks2 = (long long)((ntok + 255) / 256) * heads * nimg;
  int mode = g_attn_mode;
  // automatic: large grids (>= 2 workgroups of 256 queries per CU) take the one-wave-per-SIMD kernel, small ones (a single
  // image pair) the VALU-lean 32-query kernel, which fills the part with 4x as many workgroups
  if (mode == 0) mode = blocks2 >= 512 ? MK_ATTN_DEFAULT_BIG : 4;
  if (mode == 7 && !mk::launch_attn_w1(q, k, vt, out, ldo, nimg, heads, ntok, ntok_pad,
                                       std::is_same<T, __bf16>::value ? MK_BF16 : MK_F16, st))
    mode = 2;
  if (mode == 7) {
  } else if (mode == 8) {
    hipLaunchKernelGGL((attn_fwd_pp_kernel<T>), dim3((ntok + 255) / 256, heads, nimg), dim3(512), 0, st, (const T*)q, (const T*)k,
                       (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  } else if (mode == 4) {
    hipLaunchKernelGGL((attn_fwd_lean_kernel<T, true>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  } else if (mode == 9) {
    hipLaunchKernelGGL((attn_fwd_lean_kernel<T, false>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  } else if (mode == 10) {
    hipLaunchKernelGGL((attn_fwd_fold_kernel<T, 1>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  } else if (mode == 11) {
    hipLaunchKernelGGL((attn_fwd_fold_kernel<T, 2>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  } else if (mode == 2) {
    hipLaunchKernelGGL((attn_fwd_kernel<T, 2>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  } else {
    hipLaunchKernelGGL((attn_fwd_kernel<T, 1>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  }
}

}  // namespace

extern "C" int mk_attn_set_mode(int mode) {
  MK_CHECK_ARG(mode == 0 || mode == 1 || mode == 2 || mode == 4 || mode == 7 || mode == 8 || mode == 9 || mode == 10 || mode == 11,
               "mk_attn_set_mode: 0 automatic, 1 = 32 q/wave, 2 = 64 q/wave, 4 = VALU-lean, 7 = one wave per SIMD, 8 = ping-pong");
  g_attn_mode = mode;
  return MK_OK;
}

extern "C" int mk_flash_attn_fwd(const void* q, const void* k, const void* vt, void* out, int ldo, int nimg, int heads,
                                 int ntok, int ntok_pad, int dtype, mk_stream_t stream) {
  MK_CHECK_ARG(q && k && vt && out, "mk_flash_attn_fwd: null pointer");
  MK_CHECK_ARG(nimg > 0 && heads > 0 && ntok > 0 && ntok_pad % 64 == 0 && ntok_pad >= ntok && ldo % 4 == 0 &&
                   ldo >= heads * 64,
               "mk_flash_attn_fwd: bad geometry");
  if (dtype == MK_BF16)
    launch_attn<__bf16>(q, k, vt, out, ldo, nimg, heads, ntok, ntok_pad, (hipStream_t)stream);
  else if (dtype == MK_F16)
    launch_attn<_Float16>(q, k, vt, out, ldo, nimg, heads, ntok, ntok_pad, (hipStream_t)stream);
  else if (dtype == MK_F32)
    mk::launch_attn_f32((const float*)q, (const float*)k, (const float*)vt, (float*)out, ldo, nimg, heads, ntok, ntok_pad,
                        (hipStream_t)stream);
  else
    MK_CHECK_ARG(false, "mk_flash_attn_fwd: bad dtype");
  MK_CHECK_LAUNCH();
  return MK_OK;
}
