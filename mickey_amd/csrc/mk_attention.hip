// mickey_amd -- flash-style multi-head attention forward for gfx950, head_dim 64, non-causal.
// Replaces q@k^T -> softmax -> @v of reference DINO_modules/layers/attention.py:53-59 (the xformers
// path :72-76 is the same maths).  The ntok x ntok score matrix never leaves registers.
//
// Design (wave64, v_mfma_f32_32x32x16):
//  * workgroup = 4 waves of one (image, head); each wave owns 32 or 64 queries (QB sub-blocks of 32).
//  * two kernels share this design (mk_attn_set_mode): attn_fwd_fold_kernel (production: the lean softmax -- running maximum
//    folded into the accumulator init, re-based only when a tile outgrows it, detected from the row sums -- with fp32 row sums on the VALU) and
//    attn_fwd_kernel (classic online softmax, kept as the A/B partner).  Round 3 built and measured three more structures
//    (one wave per SIMD with an asm-owned accumulator file, ping-pong wave-rows, matrix-pipe row sums): LABNOTES.md 2.2.
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
}

namespace {
using namespace mk;

constexpr int KV_TILE_BYTES = 64 * 64 * 2;  // 8 KiB
constexpr float ATT_REBASE_SUM = 4096.0f; // lean softmax: the running maximum is re-based when a lane's share of a tile's row sum exceeds 2^12

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
// attention kernel (and the GEMM) at its power limit -- profiles/r03_pmc_clock_attention_variants.json: variants that need fewer
// cycles run at a lower clock and finish at the same wall time -- so the kernel that issues the least work wins:
//   * no subtraction of the running maximum (folded into the MFMA accumulator init), no per-tile rescale of O (re-base only
//     when a lane's share of a row sum exceeds 2^12: no per-tile maximum at all), 16 MFMAs per 32 x 64 tile (the matrix-pipe row sums of the lean kernel were a
//     fifth of its matrix work: 805-825 -> 885 TFLOP/s when they went back to 32 fp32 adds);
//   * QB = 2: every K / V^T fragment read from LDS feeds two MFMAs and a workgroup covers 256 queries per staged tile.
// One wave's share of a workgroup: QB sub-blocks of 32 queries from q0 on (all four waves stage K / V^T and meet at the barriers
// whatever QB they run with: one barrier per KV tile in either instantiation).
template <typename T, int QB>
__device__ __forceinline__ void attn_fold_wave(const T* __restrict__ Qh, const T* __restrict__ Kh, const T* __restrict__ Vh,
                                               T* __restrict__ out, int ldo, int img, int head, int ntok, int ntok_pad, char* smem,
                                               int lane, int wave, int q0) {
  using V8 = typename Lp<T>::V8;
  using V4 = typename Lp<T>::V4;
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
    // the peeled final tile with at most 32 valid keys (1939 tokens: 19): its second 32-key block is all padding -- P = 0 there,
    // exactly -- so neither its scores nor its half of P.V are computed (round 6: 1/62 of the kernel's matrix work; adding the
    // zeros changed no bit, leaving them out changes none)
    const bool half_tile = LAST && (ntok & 63) != 0 && (ntok & 63) <= 32;   // wave-uniform
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (LAST && kb == 1 && half_tile) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[qb][1][r] = -1e30f;
        continue;
      }
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
      // P = 2^S' and this lane's share of the row sums.  No maximum is taken on the way (32 scores cost ~20 v_max, as many VALU
      // issues as their conversion to 16 bit): a score that outgrew the running maximum shows in the SUM -- any P > 2^12
      // makes its lane's sum > 2^12 -- and only then (wave-uniform, after the first tile practically never) is the maximum
      // computed and m re-based; S' is still in registers, nothing is recomputed on the matrix pipe.  2^12 keeps P inside
      // fp16 and the sums far from overflow; softmax is invariant to where m sits, fp32 sums are relative to their largest term.
      auto exp_sum = [&]() {
        float rs4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent partial sums: no 32-deep dependent add chain
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float pv = __builtin_amdgcn_exp2f(s[qb][s4 >> 1][(s4 & 1) * 8 + e]);
            pf[qb][s4][e] = (T)pv;
            rs4[e & 3] += pv;
          }
        return (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
      };
      float rs = 0.f;
      bool rebase = first;
      if (!first) {
        rs = exp_sum();
        rebase = __any(!(rs <= ATT_REBASE_SUM));   // also true for +inf (an overflowed sum); the file is built with
                                                   // -fno-honor-nans, so nothing is claimed for NaN: finite q, k cannot make one
      }
      if (rebase) {   // the first tile, and tiles that outgrew m: re-base m to this tile's maximum
        float t8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) t8[r] = fmaxf(fmaxf(s[qb][0][r], s[qb][0][r + 8]), fmaxf(s[qb][1][r], s[qb][1][r + 8]));
        float mx = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), t8[2]), fmaxf(fmaxf(t8[3], t8[4]), t8[5]));
        mx = fmaxf(fmaxf(mx, t8[6]), t8[7]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float shift = first ? mx : fmaxf(mx, 0.f);     // never lower m after the first tile (rows that did not outgrow
                                                             // it: shift 0, bit-identical to not re-basing)
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-shift);   // first tile: O = l = 0 (2^-shift may be +inf there: 0 x inf)
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
        rs = exp_sum();
      }
      l_run[qb] += rs;
    }
    first = false;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int row = dt * 32 + j;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (LAST && s4 >= 2 && half_tile) break;   // keys 32..63 of the tile: P = 0
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

template <typename T, int QB, bool TAIL1 = true>   // TAIL1 = false: dev mode 6, the A/B partner without the one-sub-block tail wave
__global__ __launch_bounds__(256, QB == 2 ? 2 : 3) void attn_fwd_fold_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                              const T* __restrict__ vt, T* __restrict__ out,
                                                                              int ldo, int heads, int ntok, int ntok_pad) {
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
  if constexpr (QB == 2 && TAIL1) {
    // the one wave per (image, head) whose SECOND 32-query sub-block is all padding (1939 tokens: queries 1920..1938 of the last
    // block) runs the one-sub-block body: 1/62 of the kernel's matrix work, decided once per wave, outside the tile loop (as a
    // wave-uniform `if` inside the loop it re-scheduled the whole loop: -5 %, profiles/r06r_attn_padding_skips.txt).  The
    // 32- and 64-queries-per-wave bodies are bit-identical per query (tests/test_kernels_gpu.py)
    if (q0 < ntok && q0 + 32 >= ntok) {
      attn_fold_wave<T, 1>(Qh, Kh, Vh, out, ldo, img, head, ntok, ntok_pad, smem, lane, wave, q0);
      return;
    }
  }
  attn_fold_wave<T, QB>(Qh, Kh, Vh, out, ldo, img, head, ntok, ntok_pad, smem, lane, wave, q0);
}

// The same kernel on v_mfma_f32_16x16x32 (modes 4 / 5).  At the socket power limit the 16x16x32 shape gets ~15 % more flops
// through the matrix pipe than 32x32x16 (tools/micro/mfma_power.hip: 2.05 vs 1.79 PFLOP/s on random bf16 operands), and this
// kernel is bound by energy like the GEMMs.  Layout: S^T = K.Q^T in 16-key x 16-query blocks -- after the MFMA lane (n, g) =
// (lane & 15, lane >> 4) holds keys 16 kb + 4 g .. + 3 of query n, so a query's 64 scores of a tile sit in 4 lanes (row sums /
// maxima: two lane exchanges, once per kernel / per re-base).  O^T = V^T.P^T in 32-key steps: k-slot 8 g + e of a step is key
// 32 s + 4 g + e (e < 4) or 32 s + 16 + 4 g + e - 4 -- what the lane already holds of key blocks 2 s and 2 s + 1 -- so the
// V^T operand is two 8-byte LDS reads (the quads 4 g of both blocks, at their key-permuted positions) instead of one 16-byte
// read; everything else (staging, lean softmax, re-base rule, operation order per query) is the kernel above.
template <typename T, int QB>
__global__ __launch_bounds__(256, QB == 2 ? 2 : 3) void attn_fwd_fold16_kernel(const T* __restrict__ q, const T* __restrict__ k,
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
  const int n = lane & 15, g = lane >> 4;
  const bool wave_has_queries = q0 < ntok;

  V8 qf[QB][2][2];   // [32-query block][16-query half][32-wide d step]
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      int qrow = q0 + qb * 32 + nb * 16 + n;
      qrow = qrow < ntok_pad ? qrow : ntok_pad - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[qb][nb][ks] = *(const V8*)(Qh + (long long)qrow * 64 + ks * 32 + g * 8);
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

  f32x4 o[QB][2][4], negm[QB][2];
  float m_run[QB][2], l_run[QB][2];   // l_run: this lane's share of the row sum (the 4 lanes of a query are added at the end)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int db = 0; db < 4; ++db) o[qb][nb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
      negm[qb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      m_run[qb][nb] = l_run[qb][nb] = 0.f;
    }
  bool first = true;
  // position quad of key quad g inside a 16-key block (mk_gemm_qkv stores V^T with token bits 2 <-> 3 swapped)
  const int pq = ((g & 1) << 1) | (g >> 1);

  const int nkt = (ntok + 63) >> 6;
  stage(0, 0);
  auto kv_tile = [&](const int kt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!LAST) stage((kt + 1) & 1, kt + 1);
    if (!wave_has_queries) return;
    const char* sK = smem + (kt & 1) * 2 * KV_TILE_BYTES;
    const char* sV = sK + KV_TILE_BYTES;

    f32x4 s[QB][2][4];   // [qb][nb][16-key block]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {   // (d step outside: consecutive MFMAs go to different accumulators)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int row = kb * 16 + n;
        const V8 kf = *(const V8*)(sK + row * 128 + swz8(row, ks * 4 + g) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            s[qb][nb][kb] = Lp<T>::mma16(kf, qf[qb][nb][ks], ks == 0 ? negm[qb][nb] : s[qb][nb][kb]);   // S' = K.Q^T - m
      }
    }
    V8 pf[QB][2][2];   // [qb][nb][32-key step]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        if (LAST && (ntok & 63)) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kt * 64 + kb * 16 + 4 * g + r >= ntok) s[qb][nb][kb][r] = -1e30f;
        }
        auto exp_sum = [&]() {   // P = 2^S' and this lane's share of the row sum (see the kernel above)
          float rs4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = __builtin_amdgcn_exp2f(s[qb][nb][kb][r]);
              pf[qb][nb][kb >> 1][(kb & 1) * 4 + r] = (T)pv;
              rs4[r] += pv;
            }
          return (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
        };
        float rs = 0.f;
        bool rebase = first;
        if (!first) {
          rs = exp_sum();
          rebase = __any(!(rs <= ATT_REBASE_SUM));
        }
        if (rebase) {
          float t4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) t4[r] = fmaxf(fmaxf(s[qb][nb][0][r], s[qb][nb][1][r]), fmaxf(s[qb][nb][2][r], s[qb][nb][3][r]));
          float mx = fmaxf(fmaxf(t4[0], t4[1]), fmaxf(t4[2], t4[3]));
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float shift = first ? mx : fmaxf(mx, 0.f);
          const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-shift);   // first tile: O = l = 0 (2^-shift may be +inf there: 0 x inf)
#pragma unroll
          for (int b4 = 0; b4 < 4; ++b4)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              o[qb][nb][b4][i] *= alpha;
              s[qb][nb][b4][i] -= shift;
            }
          l_run[qb][nb] *= alpha;
          m_run[qb][nb] += shift;
#pragma unroll
          for (int i = 0; i < 4; ++i) negm[qb][nb][i] = -m_run[qb][nb];
          rs = exp_sum();
        }
        l_run[qb][nb] += rs;
      }
    first = false;
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int st = 0; st < 2; ++st) {   // (key step outside, as above)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int row = db * 16 + n;
        const int c1 = st * 4 + (pq >> 1);
        const u32x2_t lo = *(const u32x2_t*)(sV + row * 128 + swz8(row, c1) * 16 + (pq & 1) * 8);
        const u32x2_t hi = *(const u32x2_t*)(sV + row * 128 + swz8(row, c1 + 2) * 16 + (pq & 1) * 8);
        const V8 vf = __builtin_bit_cast(V8, u32x4_t{lo[0], lo[1], hi[0], hi[1]});
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) o[qb][nb][db] = Lp<T>::mma16(vf, pf[qb][nb][st], o[qb][nb][db]);
      }
    }
  };
  for (int kt = 0; kt < nkt - 1; ++kt) kv_tile(kt, std::false_type{});
  kv_tile(nkt - 1, std::true_type{});

#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float l = l_run[qb][nb];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      const int qi = q0 + qb * 32 + nb * 16 + n;
      if (qi < ntok) {
        T* orow = out + ((long long)img * ntok + qi) * ldo + head * 64;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          V4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (T)(o[qb][nb][db][e] * inv);
          *(V4*)(orow + db * 16 + g * 4) = w;
        }
      }
    }
}

int g_attn_mode = 0;   // mk_attn_set_mode: 0 automatic, 1 / 2: the fold kernel with 32 / 64 queries per wave, 3: classic online softmax (64 q/wave)

template <typename T>
void launch_attn(const void* q, const void* k, const void* vt, void* out, int ldo, int nimg, int heads, int ntok, int ntok_pad,
                 hipStream_t st) {
  const long long blocks2 = (long long)((ntok + 255) / 256) * heads * nimg;
  int mode = g_attn_mode;
  // automatic: 64 queries per wave once that still leaves >= 2 workgroups per CU (halves the LDS reads and the staged K / V^T
  // bytes per MFMA), 32 queries per wave for small grids (a single image pair: 4x the workgroups)
  if (mode == 0) mode = blocks2 >= 512 ? 2 : 1;
  if (mode == 1)
    hipLaunchKernelGGL((attn_fwd_fold_kernel<T, 1>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  else if (mode == 2)
    hipLaunchKernelGGL((attn_fwd_fold_kernel<T, 2>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  else if (mode == 6)
    hipLaunchKernelGGL((attn_fwd_fold_kernel<T, 2, false>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  else if (mode == 4)
    hipLaunchKernelGGL((attn_fwd_fold16_kernel<T, 1>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  else if (mode == 5)
    hipLaunchKernelGGL((attn_fwd_fold16_kernel<T, 2>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<T, 2>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
}

}  // namespace

extern "C" int mk_attn_set_mode(int mode) {
  MK_CHECK_ARG(mode >= 0 && mode <= 6, "mk_attn_set_mode: 0 automatic, 1 / 2 = lean softmax with 32 / 64 queries per wave, 3 = classic online softmax, 4 / 5 = 1 / 2 on the 16x16x32 MFMA, 6 = 2 without the one-sub-block tail wave");
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
