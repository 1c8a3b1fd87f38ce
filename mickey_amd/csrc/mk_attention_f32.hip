// mickey_amd -- exact-fp32 multi-head attention (the parity mode, dtype MK_F32; head_dim 64).
//
// softmax(q k^T) v of reference DINO_modules/layers/attention.py:53-59 in plain fp32 VALU arithmetic: one query row per
// lane (q and the output row in registers), key / value tiles of 32 tokens broadcast from LDS, block-wise online softmax
// (one rescale per 32 keys).  q arrives pre-scaled by 64^-1/2 * log2(e) (the QKV epilogue), so probabilities are
// exp2(s - max); exp2f is the accurate library form.  Operand layouts are those the QKV epilogue writes for every dtype:
// q, k [image, head, ntok_pad, 64]; v^T [image, head, 64, ntok_pad] with token t at column vperm(t).
// Speed is not a goal here (fp32 VALU rate, LDS-broadcast bound): this is correctness evidence for the 16-bit kernels.
#include "mk_common.hpp"

namespace mk {
namespace {

__device__ __forceinline__ int vperm(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }

constexpr int KT = 32;   // keys per tile

__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ vt, float* __restrict__ out, int ldo, int heads,
                                                       int ntok, int ntok_pad) {
  __shared__ __attribute__((aligned(16))) float sK[KT][64];
  __shared__ __attribute__((aligned(16))) float sV[KT][64];
  const int lane = threadIdx.x;
  const int head = blockIdx.y, img = blockIdx.z;
  const long long hb = (long long)img * heads + head;
  const int row = blockIdx.x * 64 + lane;
  const bool live = row < ntok;
  const float* qp = q + (hb * ntok_pad + (live ? row : 0)) * 64;
  float qr[64], o[64];
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const f32x4 t = *(const f32x4*)(qp + d);
    qr[d] = t[0]; qr[d + 1] = t[1]; qr[d + 2] = t[2]; qr[d + 3] = t[3];
    o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int t0 = 0; t0 < ntok; t0 += KT) {
    __syncthreads();
    // K tile: 32 rows x 256 B, coalesced;  V tile: element (key j, d) = vt[d][vperm(t0 + j)]
    for (int i = lane; i < KT * 16; i += 64) {
      const int j = i >> 4, c = i & 15;
      *(f32x4*)&sK[j][c * 4] = *(const f32x4*)(k + (hb * ntok_pad + t0 + j) * 64 + c * 4);
    }
    for (int i = lane; i < KT * 64; i += 64) {
      const int d = i >> 5, j = i & 31;
      sV[j][d] = vt[(hb * 64 + d) * ntok_pad + vperm(t0 + j)];
    }
    __syncthreads();
    const int nj = min(KT, ntok - t0);
    float s[KT];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < 64; d += 4) {
        const f32x4 kv = *(const f32x4*)&sK[j][d];
        a = fmaf(qr[d], kv[0], a);
        a = fmaf(qr[d + 1], kv[1], a);
        a = fmaf(qr[d + 2], kv[2], a);
        a = fmaf(qr[d + 3], kv[3], a);
      }
      s[j] = j < nj ? a : -INFINITY;
      tmax = fmaxf(tmax, s[j]);
    }
    const float mn = fmaxf(m, tmax);
    const float alpha = exp2f(m - mn);   // m = -inf on the first tile: exp2f(-inf) = 0
    l *= alpha;
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] *= alpha;
    m = mn;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const float pj = exp2f(s[j] - m);   // masked keys: exp2f(-inf) = 0
      l += pj;
#pragma unroll
      for (int d = 0; d < 64; d += 4) {
        const f32x4 vv = *(const f32x4*)&sV[j][d];
        o[d] = fmaf(pj, vv[0], o[d]);
        o[d + 1] = fmaf(pj, vv[1], o[d + 1]);
        o[d + 2] = fmaf(pj, vv[2], o[d + 2]);
        o[d + 3] = fmaf(pj, vv[3], o[d + 3]);
      }
    }
  }
  if (!live) return;
  const float inv = 1.0f / l;
  float* op = out + ((long long)img * ntok + row) * ldo + head * 64;
#pragma unroll
  for (int d = 0; d < 64; d += 4) *(f32x4*)(op + d) = f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
}

}  // namespace

void launch_attn_f32(const float* q, const float* k, const float* vt, float* out, int ldo, int nimg, int heads, int ntok,
                     int ntok_pad, hipStream_t st) {
  hipLaunchKernelGGL(attn_f32_kernel, dim3((ntok + 63) / 64, heads, nimg), dim3(64), 0, st, q, k, vt, out, ldo, heads, ntok,
                     ntok_pad);
}

}  // namespace mk
