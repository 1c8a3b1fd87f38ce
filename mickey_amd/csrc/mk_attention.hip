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
}

namespace {
using namespace mk;

constexpr int KV_TILE_BYTES = 64 * 64 * 2;  // 8 KiB

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
// ABL (timing ablations, wrong results): 1 = no per-tile barrier, 2 = no per-tile barrier and no DMA wait
template <typename T, int QB, int ABL = 0>
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
    if (ABL < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ABL < 1) __syncthreads();
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

// Software-pipelined variant (32 queries per wave): the QK^T MFMAs of KV tile t+1 are issued BEFORE the softmax VALU
// work of tile t (S is double-buffered in registers), so within one wave the matrix pipe computes S(t+1) while the
// VALU exponentiates S(t); then P(t).V(t).  K therefore runs one tile ahead of V in the LDS double buffers.
template <typename T>
__global__ __launch_bounds__(256, 3) void attn_fwd_pipe_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ vt, T* __restrict__ out, int ldo,
                                                               int heads, int ntok, int ntok_pad) {
  using V8 = typename Lp<T>::V8;
  using V4 = typename Lp<T>::V4;
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // K[2] | Vt[2]

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
  const int srow = lane >> 3, sp = lane & 7;
  auto stage_k = [&](int kt) {
    char* sK = smem + (kt & 1) * KV_TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ii = wave * 2 + t, r = ii * 8 + srow;
      glds16(Kh + (long long)(kt * 64 + r) * 64 + swz8(r, sp) * 8, sK + ii * 1024);
    }
  };
  auto stage_v = [&](int kt) {
    char* sV = smem + (2 + (kt & 1)) * KV_TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ii = wave * 2 + t, r = ii * 8 + srow;
      glds16(Vh + (long long)r * ntok_pad + kt * 64 + swz8(r, sp) * 8, sV + ii * 1024);
    }
  };
  auto qk = [&](int kt, f32x16 (&s)[2]) {   // S^T(kt) = K(kt) . Q^T
    const char* sK = smem + (kt & 1) * KV_TILE_BYTES;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
      const int row = kb * 32 + j;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 kf = *(const V8*)(sK + row * 128 + swz8(row, ks * 2 + hi) * 16);
        s[kb] = Lp<T>::mma32(kf, qf[ks], s[kb]);
      }
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[0][i] = o[1][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const int nkt = (ntok + 63) >> 6;

  // one pipeline step: s_cur = S(t) (already computed), s_nxt receives S(t+1)
  auto step = [&](int t, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // V(t) and K(t+1) (issued one step ago) have landed
    __syncthreads();
    if (t + 1 < nkt) stage_v(t + 1);   // buffer of V(t-1): its readers are behind the barrier
    if (t + 2 < nkt) stage_k(t + 2);   // buffer of K(t): read one step ago
    if (t + 1 < nkt) qk(t + 1, s_nxt); // matrix pipe works on S(t+1) ...
    __builtin_amdgcn_sched_barrier(0);
    // ... while the VALU does the softmax of S(t)
    if (t == nkt - 1 && (ntok & 63)) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= ntok) s_cur[kb][r] = -1e30f;
        }
    }
    float t8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) t8[r] = fmaxf(fmaxf(s_cur[0][r], s_cur[0][r + 8]), fmaxf(s_cur[1][r], s_cur[1][r + 8]));
    float mx = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), t8[2]), fmaxf(fmaxf(t8[3], t8[4]), t8[5]));
    mx = fmaxf(fmaxf(mx, t8[6]), t8[7]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const bool grew = m_new > m_run;
    float rs4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(s_cur[kb][r] - m_new);
        s_cur[kb][r] = pv;
        rs4[r & 3] += pv;
      }
    if (__any(grew)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] *= alpha;
        o[1][i] *= alpha;
      }
    }
    m_run = m_new;
    l_run += (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
    V8 pf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[s4][e] = (T)s_cur[s4 >> 1][(s4 & 1) * 8 + e];
    const char* sV = smem + (2 + (t & 1)) * KV_TILE_BYTES;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int row = dt * 32 + j;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const V8 vf = *(const V8*)(sV + row * 128 + swz8(row, s4 * 2 + hi) * 16);
        o[dt] = Lp<T>::mma32(vf, pf[s4], o[dt]);
      }
    }
  };

  f32x16 sa[2], sb[2];
  stage_k(0);
  stage_v(0);
  if (nkt > 1) stage_k(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  qk(0, sa);
  for (int t = 0; t < nkt; t += 2) {
    step(t, sa, sb);
    if (t + 1 < nkt) step(t + 1, sb, sa);
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
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

// VALU-lean variant (32 queries per wave).  A wave64 VALU instruction costs ~4 issue cycles and a 32-cycle MFMA hides
// only a handful of them, so the softmax (~170 VALU instructions per 16 MFMAs) bounds the kernels above.  Here:
//  * the running maximum is folded into the QK^T accumulator init: S' = K.Q^T + (-m) comes out of the MFMA already
//    shifted (a persistent 16-register vector holds -m; no per-element subtraction);
//  * m is only re-based when a tile's maximum exceeds it by more than 2^8 (then O and the row sums are rescaled);
//    otherwise P = exp2(S') <= 256 is used as is -- the common case after the first tile;
//  * the row sums are computed on the matrix pipe (ones . P^T, 4 extra MFMAs per tile) instead of 32 VALU adds; they
//    sum the same 16-bit P that multiplies V, and need no cross-lane exchange.
// Per tile and wave: 20 MFMAs and ~70 VALU instructions (16 max3, 32 exp, 16 cvt).
constexpr float ATT_REBASE_THR = 8.0f;

template <typename T, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 4 ? 3 : 2) void attn_fwd_lean_kernel(const T* __restrict__ q, const T* __restrict__ k,
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
  const int q0 = ab.qblk * (NWAVES * 32) + wave * 32;
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
    for (int t = 0; t < 8 / NWAVES; ++t) {   // 8 one-KiB pieces per operand tile, spread over the waves
      const int ii = wave * (8 / NWAVES) + t;
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
  float m_run = 0.f;
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
        lsum[i] *= alpha;
        s[0][i] -= shift;
        s[1][i] -= shift;
      }
      m_run += shift;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = -m_run;
      first = false;
    }
    V8 pf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[s4][e] = (T)__builtin_amdgcn_exp2f(s[s4 >> 1][(s4 & 1) * 8 + e]);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) lsum = Lp<T>::mma32(ones, pf[s4], lsum);   // row sums on the matrix pipe
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

// ---------------------------------------------------------------------------------------------------------
// Software-pipelined VALU-lean kernel with a PINNED instruction interleave (mode 6).
// Measured (tools/micro/r2_probe.hip): VALU-class issue bandwidth is per SIMD, ~1 instruction per 4.7 cycles however
// many waves share it; a 32x32x16 MFMA occupies the matrix pipe for 32 cycles but one issue slot, so ~5 plain VALU
// instructions (v_exp_f32 counts ~2) fit in its shadow -- IF they sit next to it in program order and do not depend on
// it.  The kernels above issue S = K.Q^T, then the softmax, then P.V: each phase waits for the one before, and two
// free-running waves per SIMD only partly fill each other's gaps (31 % of the MFMA roof).  Here one wave's iteration t
// works on three different KV tiles, none depending on another inside the iteration:
//     matrix pipe:  O += V(t-1).P(t-1)   rowsum += 1.P(t-1)   S'(t+1) = K(t+1).Q^T - m      (20 MFMAs)
//     VALU:         P(t) = bf16(exp2(S'(t)))  and the tile maximum of S'(t)                    (32 exp, 16 cvt, 11 max3)
// written out as 20 slots of {1 MFMA, <= 1 fragment read, 2-4 VALU} with a scheduling barrier between slots, so the
// order survives the compiler.  Per slot the VALU work is ~6 issue slots: the loop is matrix-pipe bound by construction.
// Running maximum as in the lean kernel (folded into the accumulator init, re-based only when a tile exceeds it by 2^8):
// the decision for tile t is taken at the head of iteration t, BEFORE any S'(t+1) MFMA (which therefore already starts
// from the new -m); S'(t) is shifted in place; O and the row sums, which still receive P(t-1) in this iteration, are
// rescaled at the head of iteration t+1.  K tiles are staged two iterations ahead, V tiles one (same 32 KiB of LDS), by
// SGPR-addressed LDS-DMA invisible to hipcc's vmcnt bookkeeping: one explicit vmcnt(0) + barrier per iteration.
struct LpSched {
  int n, nfrag;
  int chain[20], step[20], frag[20];   // per slot: accumulator chain (0,1 = S' halves; 2,3 = O halves; 4 = row sums), k-step, index of its LDS fragment (-1: none)
  int fchain[16], fstep[16];           // per fragment of the stream
};
constexpr LpSched lp_sched(bool pv, bool sn) {
  LpSched s{};
  // chain-major: the four MFMAs of an accumulator back to back.  Measured: round-robin over the five chains (no two
  // consecutive MFMAs on one accumulator) is 14 % SLOWER -- a chain keeps its accumulator inside the matrix pipe, a switch
  // re-reads 16 registers x 64 lanes of SrcC from the VGPR file, competing with the VALU's operand reads.  (Accumulators
  // in AGPRs would take that traffic off the VGPR ports, but at 2 waves per SIMD hipcc splits the 256 registers 128 + 128
  // as soon as one AGPR is used and the 164 live VGPRs of this kernel spill.)
  for (int c = 0; c < 5; ++c)
    for (int st = 0; st < 4; ++st) {
      if ((c < 2 && !sn) || (c >= 2 && !pv)) continue;
      s.chain[s.n] = c;
      s.step[s.n] = st;
      if (c < 4) {
        s.fchain[s.nfrag] = c;
        s.fstep[s.nfrag] = st;
        s.frag[s.n] = s.nfrag++;
      } else {
        s.frag[s.n] = -1;
      }
      ++s.n;
    }
  return s;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_lp_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const T* __restrict__ vt, T* __restrict__ out, int ldo, int heads,
                                                             int ntok, int ntok_pad) {
  using V8 = typename Lp<T>::V8;
  using V4 = typename Lp<T>::V4;
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // K[2] | Vt[2]
  char* const sKb = smem;
  char* const sVb = smem + 2 * KV_TILE_BYTES;

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
  const bool has_q = q0 < ntok;   // pad-only waves of the last query block stage K/V and keep the barriers, nothing else

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

  // LDS-DMA: 8 one-KiB pieces per operand tile, 2 per wave; per-lane byte offsets are constant, the tile base is uniform
  const int srow = lane >> 3, sp = lane & 7;
  unsigned voffK[2], voffV[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = (wave * 2 + t) * 8 + srow;
    voffK[t] = (unsigned)((r * 64 + swz8(r, sp) * 8) * (int)sizeof(T));
    voffV[t] = (unsigned)((r * ntok_pad + swz8(r, sp) * 8) * (int)sizeof(T));
  }
  auto stageK = [&](int kt) {
#pragma unroll
    for (int t = 0; t < 2; ++t) glds16_sv(Kh + (long long)kt * 4096, voffK[t], sKb + (kt & 1) * KV_TILE_BYTES + (wave * 2 + t) * 1024);
  };
  auto stageV = [&](int kt) {
#pragma unroll
    for (int t = 0; t < 2; ++t) glds16_sv(Vh + kt * 64, voffV[t], sVb + (kt & 1) * KV_TILE_BYTES + (wave * 2 + t) * 1024);
  };
  // fragment g of an iteration's LDS stream: 0..7 = V^T(kt) fragment (dt = g >> 2, s4 = g & 3), 8..15 = K(kt) (kb, ks)
  auto vfrag = [&](int kt, int g) {
    const int row = (g >> 2) * 32 + j;
    return *(const V8*)(sVb + (kt & 1) * KV_TILE_BYTES + row * 128 + swz8(row, (g & 3) * 2 + hi) * 16);
  };
  auto kfrag = [&](int kt, int g) {
    const int row = (g >> 2) * 32 + j;
    return *(const V8*)(sKb + (kt & 1) * KV_TILE_BYTES + row * 128 + swz8(row, (g & 3) * 2 + hi) * 16);
  };

  f32x16 o[2], lsum, negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    o[0][i] = o[1][i] = 0.f;
    lsum[i] = 0.f;
    negm[i] = 0.f;          // m = 0 to start with; iteration 0 re-bases to the first tile's maximum
  }
  float m_run = 0.f, alpha_pend = 1.f;
  bool first = true, pend = false;
  const int nkt = (ntok + 63) >> 6;

  f32x16 sA[2], sB[2];   // S' of the tile being exponentiated / of the tile being accumulated (roles alternate)
  // P of the previous tile (feeds P.V) / of the current tile (being produced), as packed 16-bit pairs: a pair is pinned
  // into the slot that produces it by an empty asm (left to itself the compiler sinks all 16 conversions to the end)
  typedef T T2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 pA[4], pB[4];
  auto pfrag = [](const u32x4& w) { return __builtin_bit_cast(V8, w); };

  // prologue: S'(0)
  stageK(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nkt > 1) stageK(1);
#pragma unroll
  for (int g = 0; g < 8; ++g) sA[g >> 2] = Lp<T>::mma32(kfrag(0, g), qf[g & 3], (g & 3) == 0 ? negm : sA[g >> 2]);

  // One iteration.  PV: tile t-1 exists; SN: tile t+1 exists; LAST: tile t is the (possibly ragged) last one.
#ifdef MK_ATTN_LP_DBG
  unsigned long long dbg_sync = 0, dbg_head = 0, dbg_slots = 0;
#define LP_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#else
#define LP_T(x)
#endif
  auto iter = [&](int t, f32x16 (&sC)[2], f32x16 (&sN)[2], u32x4 (&pP)[4], u32x4 (&pC)[4], auto pv_tag, auto sn_tag, auto last_tag) {
    constexpr bool PV = decltype(pv_tag)::value, SN = decltype(sn_tag)::value, LAST = decltype(last_tag)::value;
    LP_T(t0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K(t+1), V(t-1) (and everything older) landed
    __syncthreads();                                     // ... for every wave; K(t), V(t-2) are no longer read
    LP_T(t1);
    if (t + 2 < nkt) stageK(t + 2);
    stageV(t);
    if (!has_q) return;
    if (pend) {   // wave-uniform, rare: the re-base decided one iteration ago, now that P(t-2).V has been accumulated
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] *= alpha_pend;
        o[1][i] *= alpha_pend;
        lsum[i] *= alpha_pend;
      }
      pend = false;
    }
    if (LAST && (ntok & 63)) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= ntok) sC[kb][r] = -1e30f;
        }
    }
    // ---- head: tile maximum of S'(t) and the (rare) re-base, before any S'(t+1) MFMA is issued
    float t8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) t8[r] = fmaxf(fmaxf(sC[0][r], sC[0][r + 8]), fmaxf(sC[1][r], sC[1][r + 8]));
    float mx = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), t8[2]), fmaxf(fmaxf(t8[3], t8[4]), t8[5]));
    mx = fmaxf(fmaxf(mx, t8[6]), t8[7]);
    if (__any(mx > ATT_REBASE_THR) || first) {
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float shift = first ? mx : fmaxf(mx, 0.f);     // never lower m after the first tile
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sC[0][i] -= shift;
        sC[1][i] -= shift;
      }
      m_run += shift;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = -m_run;
      if (!first) {   // O / row sums are rescaled at the head of the next iteration (they still receive P(t-1) below)
        alpha_pend = __builtin_amdgcn_exp2f(-shift);
        pend = true;
      }
      first = false;
    }
    // ---- 20 (PV) / 8 pinned slots; fragments are read four fragments ahead of the MFMA that consumes them.
    constexpr LpSched S = lp_sched(PV, SN);
    auto frag = [&](int k) {   // k-th fragment of the iteration's LDS stream, in consumption order
      const int c = S.fchain[k], st = S.fstep[k];
      return c < 2 ? kfrag(t + 1, c * 4 + st) : vfrag(t - 1, (c - 2) * 4 + st);
    };
    LP_T(t2);
    V8 fr[4];
#pragma unroll
    for (int g = 0; g < 4 && g < S.nfrag; ++g) fr[g] = frag(g);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NSLOT = S.n > 16 ? S.n : 16;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      if (i < S.n) {
        const int c = S.chain[i], st = S.step[i], fk = S.frag[i];
        if (c < 2) {
          sN[c] = Lp<T>::mma32(fr[fk & 3], qf[st], st == 0 ? negm : sN[c]);
        } else if (c < 4) {
          o[c - 2] = Lp<T>::mma32(fr[fk & 3], pfrag(pP[st]), o[c - 2]);
        } else {
          lsum = Lp<T>::mma32(ones, pfrag(pP[st]), lsum);
        }
        if (fk >= 0 && fk + 4 < S.nfrag) fr[fk & 3] = frag(fk + 4);   // MFMA i has taken its operands: refill its register
      }
      if (i < 16) {   // exponentiate and convert elements 2i, 2i+1 of the flattened S'(t) (the order P feeds P.V in)
        const int f = 2 * i;
        const float x0 = __builtin_amdgcn_exp2f(sC[f >> 4][f & 15]);
        const float x1 = __builtin_amdgcn_exp2f(sC[f >> 4][(f & 15) + 1]);
        unsigned bits = __builtin_bit_cast(unsigned, T2{(T)x0, (T)x1});
        asm volatile("" : "+v"(bits));
        pC[f >> 3][(f & 7) >> 1] = bits;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef MK_ATTN_LP_DBG
    LP_T(t3);
    dbg_sync += t1 - t0;
    dbg_head += t2 - t1;
    dbg_slots += t3 - t2;
#endif
  };
  using Y = std::true_type;
  using N = std::false_type;
  // iterations alternate the register roles (static indexing only): even t: sA -> pB ... odd t: sB -> pA
  if (nkt == 1) {
    iter(0, sA, sB, pA, pB, N{}, N{}, Y{});
  } else {
    iter(0, sA, sB, pA, pB, N{}, Y{}, N{});
    int t = 1;
    for (; t + 2 < nkt; t += 2) {
      iter(t, sB, sA, pB, pA, Y{}, Y{}, N{});
      iter(t + 1, sA, sB, pA, pB, Y{}, Y{}, N{});
    }
    if (t + 1 < nkt) {        // two tiles left: t (odd, has a successor), t+1 (last)
      iter(t, sB, sA, pB, pA, Y{}, Y{}, N{});
      iter(t + 1, sA, sB, pA, pB, Y{}, N{}, Y{});
    } else {                  // one tile left: t (odd), last
      iter(t, sB, sA, pB, pA, Y{}, N{}, Y{});
    }
  }
  // epilogue: P(nkt-1).V(nkt-1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!has_q) return;
  if (pend) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      o[0][i] *= alpha_pend;
      o[1][i] *= alpha_pend;
      lsum[i] *= alpha_pend;
    }
  }
  {
    // P of the last tile sits in pB after an even last index, in pA after an odd one
    const bool in_b = ((nkt - 1) & 1) == 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const V8 vf = vfrag(nkt - 1, g);
      o[g >> 2] = Lp<T>::mma32(vf, pfrag(in_b ? pB[g & 3] : pA[g & 3]), o[g >> 2]);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) lsum = Lp<T>::mma32(ones, pfrag(in_b ? pB[s4] : pA[s4]), lsum);
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
#ifdef MK_ATTN_LP_DBG
  // dev: first 16 bytes-per-element of the first output row of image 0 / head 0 / block 0 receive the timing sums
  if (img == 0 && head == 0 && ab.qblk == 1 && lane == 0) {
    float* d = (float*)(out + (long long)(q0) * ldo);
    d[0] = (float)dbg_sync / (float)nkt;
    d[1] = (float)dbg_head / (float)nkt;
    d[2] = (float)dbg_slots / (float)nkt;
  }
#endif
}

int g_attn_mode = 0;   // 0 auto, 1: 32 q/wave, 2: 64 q/wave, 3: software-pipelined (mk_attn_set_mode)

template <typename T>
void launch_attn(const void* q, const void* k, const void* vt, void* out, int ldo, int nimg, int heads, int ntok, int ntok_pad,
                 hipStream_t st) {
  // 64 queries per wave once that still leaves >= 2 workgroups per CU; 32 queries per wave for small batches
  const long long blocks2 = (long long)((ntok + 255) / 256) * heads * nimg;
  // default: 64-query waves for large grids (fastest inside the full forward, bench.py --attn-mode A/B), the VALU-lean
  // kernel for small ones (fastest at B = 1)
  if (g_attn_mode == 6) {
    hipLaunchKernelGGL((attn_fwd_lp_kernel<T>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
    return;
  }
  if (g_attn_mode == 4 || (g_attn_mode == 0 && blocks2 < 512)) {
    hipLaunchKernelGGL((attn_fwd_lean_kernel<T, 4>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
    return;
  }
  if (g_attn_mode == 5) {
    hipLaunchKernelGGL((attn_fwd_lean_kernel<T, 8>), dim3((ntok + 255) / 256, heads, nimg), dim3(512), 0, st, (const T*)q,
                       (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
    return;
  }
  if (g_attn_mode == 3) {
    hipLaunchKernelGGL((attn_fwd_pipe_kernel<T>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
    return;
  }
#ifdef MK_ATTN_ABLATIONS
  if (g_attn_mode == 13 || g_attn_mode == 14) {
    if (g_attn_mode == 13)
      hipLaunchKernelGGL((attn_fwd_kernel<T, 2, 1>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q,
                         (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<T, 2, 2>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q,
                         (const T*)k, (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
    return;
  }
#endif
  if (g_attn_mode == 2 || (g_attn_mode == 0 && blocks2 >= 512))
    hipLaunchKernelGGL((attn_fwd_kernel<T, 2>), dim3((ntok + 255) / 256, heads, nimg), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<T, 1>), dim3((ntok + 127) / 128, heads, nimg), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)vt, (T*)out, ldo, heads, ntok, ntok_pad);
}

}  // namespace

extern "C" int mk_attn_set_mode(int mode) {
  #ifdef MK_ATTN_ABLATIONS
  if (mode == 13 || mode == 14) { g_attn_mode = mode; return MK_OK; }
#endif
  MK_CHECK_ARG(mode >= 0 && mode <= 6, "mk_attn_set_mode: 0 auto, 1 = 32 q/wave, 2 = 64 q/wave, 3 = pipelined, 4 = VALU-lean, 5 = VALU-lean 8 waves, 6 = pipelined lean with pinned interleave");
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
