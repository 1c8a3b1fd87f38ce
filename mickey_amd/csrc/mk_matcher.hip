// mickey_amd -- dense descriptor matcher for gfx950 (reference utils/feature_matcher.py).
//
// dual-softmax (feature_matcher.py:64-83):  S = dsc0^T dsc1 / T, a learned scalar dustbin appended as
// last row / column / corner, P = softmax_rows(S+) * softmax_cols(S+), cropped back to n0 x n1.
//
// The N x M x C correlation runs on the EXACT-fp32 matrix core path (v_mfma_f32_32x32x2_f32, bitwise
// an fp32 fma chain), because `final_scores` feeds a sampler and must match the fp32 reference to
// round-off.  The (n0+1) x (n1+1) coupling matrix is never materialised:
//   pass 1  row log-sum-exp of S and of S^T (= column LSE) as online (max, sum) partials per column
//           chunk; descriptor tiles staged in LDS; 64x64 tile per workgroup, 32x32 per wave;
//   pass 2  recompute the tile, merge the partials (+ dustbin term), write
//           scores, kp_scores = scr0 (x) scr1, final_scores = scores * kp_scores   (each optional)
// Pass 2 is HBM-write-bound (3 x n0 x n1 x 4 B); stores are 128 B contiguous per 32 lanes.
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
#pragma unroll 8
  for (int kk = 0; kk < C / 2; ++kk) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk * 2 * MT], pb[kk * 2 * MT], acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  s = s * expf(m - M) + s2 * expf(m2 - M);
  m = M;
}

// pass 1.  grid (row_tiles, NCHUNK, B*2): side 0 -> rows of S, side 1 -> rows of S^T.
// part[((b*2+side)*NCHUNK + chunk)*nmax + row][2] = (max, sum exp(S - max)) over the chunk's columns
__global__ __launch_bounds__(256) void lse_partial_kernel(const float* __restrict__ dsc0, const float* __restrict__ dsc1,
                                                          float inv_t, float* __restrict__ part, int C, int n0, int n1,
                                                          int nmax) {
  __shared__ __attribute__((aligned(16))) float sA[CMAX * MT];
  __shared__ __attribute__((aligned(16))) float sB[CMAX * MT];
  float(*sred)[MT][2] = (float(*)[MT][2])sB;  // [2][MT][2], reuses sB after the last tile
  const int b = blockIdx.z >> 1, side = blockIdx.z & 1;
  const int nA = side ? n1 : n0, nB = side ? n0 : n1;
  const float* dA = (side ? dsc1 + (long long)b * C * n1 : dsc0 + (long long)b * C * n0);
  const float* dB = (side ? dsc0 + (long long)b * C * n0 : dsc1 + (long long)b * C * n1);
  const int i0 = blockIdx.x * MT;
  if (i0 >= nA) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wi = wave >> 1, wj = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int ntB = (nB + MT - 1) / MT;
  const int per = (ntB + NCHUNK - 1) / NCHUNK;
  const int jt0 = blockIdx.y * per, jt1 = min(ntB, jt0 + per);

  stage_desc(sA, dA, C, nA, i0);
  float rm[16], rsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { rm[r] = -1e30f; rsum[r] = 0.f; }
  for (int jt = jt0; jt < jt1; ++jt) {
    __syncthreads();
    stage_desc(sB, dB, C, nB, jt * MT);
    __syncthreads();
    const f32x16 acc = corr_tile(sA, sB, C, wi, wj, lane);
    const bool colok = jt * MT + wj * 32 + l31 < nB;
    if (colok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[r] * inv_t;
        const float M = fmaxf(rm[r], v);
        rsum[r] = rsum[r] * expf(rm[r] - M) + expf(v - M);
        rm[r] = M;
      }
    }
  }
  // combine the 32 lanes that share rows (same hi), then the two column waves
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(rm[r], o, 64), s2 = __shfl_xor(rsum[r], o, 64);
      lse_merge(rm[r], rsum[r], m2, s2);
    }
  }
  __syncthreads();
  if (l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      sred[wj][row][0] = rm[r];
      sred[wj][row][1] = rsum[r];
    }
  }
  __syncthreads();
  if (threadIdx.x < MT) {
    const int row = threadIdx.x;
    float m = sred[0][row][0], s = sred[0][row][1];
    lse_merge(m, s, sred[1][row][0], sred[1][row][1]);
    if (i0 + row < nA) {
      float* o = part + ((((long long)b * 2 + side) * NCHUNK + blockIdx.y) * nmax + i0 + row) * 2;
      o[0] = m;
      o[1] = s;
    }
  }
}

__device__ __forceinline__ float final_lse(const float* part, long long bs, int nmax, int row, int use_dustbin, float beta) {
  float m = -1e30f, s = 0.f;
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const float* q = part + ((bs * NCHUNK + c) * nmax + row) * 2;
    lse_merge(m, s, q[0], q[1]);
  }
  if (use_dustbin) lse_merge(m, s, beta, 1.0f);
  return m + logf(s);
}

// pass 2.  grid (col_tiles, row_tiles, B)
__global__ __launch_bounds__(256) void dual_softmax_write_kernel(const float* __restrict__ dsc0, const float* __restrict__ dsc1,
                                                                 const float* __restrict__ scr0, const float* __restrict__ scr1,
                                                                 float inv_t, int use_dustbin, float beta,
                                                                 const float* __restrict__ part, float* __restrict__ scores,
                                                                 float* __restrict__ kp, float* __restrict__ fin, int C, int n0,
                                                                 int n1, int nmax) {
  __shared__ __attribute__((aligned(16))) float sA[CMAX * MT];
  __shared__ __attribute__((aligned(16))) float sB[CMAX * MT];
  float *slr = sA, *slc = sA + MT, *ss0 = sA + 2 * MT, *ss1 = sA + 3 * MT;  // reuse sA after the MFMAs
  const int b = blockIdx.z, i0 = blockIdx.y * MT, j0 = blockIdx.x * MT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wi = wave >> 1, wj = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  stage_desc(sA, dsc0 + (long long)b * C * n0, C, n0, i0);
  stage_desc(sB, dsc1 + (long long)b * C * n1, C, n1, j0);
  __syncthreads();
  const f32x16 acc = corr_tile(sA, sB, C, wi, wj, lane);
  __syncthreads();
  if (threadIdx.x < MT) {
    const int i = i0 + threadIdx.x;
    slr[threadIdx.x] = i < n0 ? final_lse(part, (long long)b * 2 + 0, nmax, i, use_dustbin, beta) : 0.f;
    ss0[threadIdx.x] = (scr0 && i < n0) ? scr0[(long long)b * n0 + i] : 0.f;
  } else if (threadIdx.x < 2 * MT) {
    const int t = threadIdx.x - MT, jx = j0 + t;
    slc[t] = jx < n1 ? final_lse(part, (long long)b * 2 + 1, nmax, jx, use_dustbin, beta) : 0.f;
    ss1[t] = (scr1 && jx < n1) ? scr1[(long long)b * n1 + jx] : 0.f;
  }
  __syncthreads();
  const int jl = wj * 32 + l31, jx = j0 + jl;
  if (jx >= n1) return;
  const float lc = slc[jl], s1 = ss1[jl];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int il = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, i = i0 + il;
    if (i >= n0) continue;
    const float v = acc[r] * inv_t;
    const float pr = expf(v - lc) * expf(v - slr[il]);  // softmax over dim 1 (column-normalised) * dim 2
    const long long o = ((long long)b * n0 + i) * n1 + jx;
    const float kk = ss0[il] * s1;
    if (scores) scores[o] = pr;
    if (kp) kp[o] = kk;
    if (fin) fin[o] = pr * kk;
  }
}

// ---- sinkhorn ---------------------------------------------------------------------------------------
// Z[(n0+1) x (n1+1)] = couplings (S / sqrt(C) with alpha on the last row / column / corner)
__global__ __launch_bounds__(256) void couplings_kernel(const float* __restrict__ dsc0, const float* __restrict__ dsc1,
                                                        float scale, float alpha, float* __restrict__ Z, int C, int n0,
                                                        int n1) {
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
  if (jx > n1) return;
  float* Zb = Z + (long long)b * (n0 + 1) * (n1 + 1);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (i > n0) continue;
    Zb[(long long)i * (n1 + 1) + jx] = (i == n0 || jx == n1) ? alpha : acc[r] * scale;
  }
}

// u[i] = log_mu[i] - LSE_j(Z[i][j] + v[j]); one wave per row
__global__ __launch_bounds__(256) void sink_row_kernel(const float* __restrict__ Z, const float* __restrict__ v,
                                                       float* __restrict__ u, int n0, int n1, float norm, float log_last) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i > n0) return;
  const float* z = Z + ((long long)b * (n0 + 1) + i) * (n1 + 1);
  const float* vb = v + (long long)b * (n1 + 1);
  float m = -1e30f, s = 0.f;
  for (int j = lane; j <= n1; j += 64) {
    const float x = z[j] + vb[j];
    const float M = fmaxf(m, x);
    s = s * expf(m - M) + expf(x - M);
    m = M;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    lse_merge(m, s, m2, s2);
  }
  if (lane == 0) u[(long long)b * (n0 + 1) + i] = (i == n0 ? log_last : norm) - (m + logf(s));
}

// v[j] = log_nu[j] - LSE_i(Z[i][j] + u[i]); block = 64 columns x 4 row groups
__global__ __launch_bounds__(256) void sink_col_kernel(const float* __restrict__ Z, const float* __restrict__ u,
                                                       float* __restrict__ v, int n0, int n1, float norm, float log_last) {
  __shared__ float sm[4][64], ss[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6, b = blockIdx.y;
  const int j = blockIdx.x * 64 + col;
  const float* Zb = Z + (long long)b * (n0 + 1) * (n1 + 1);
  const float* ub = u + (long long)b * (n0 + 1);
  float m = -1e30f, s = 0.f;
  if (j <= n1) {
    for (int i = rg; i <= n0; i += 4) {
      const float x = Zb[(long long)i * (n1 + 1) + j] + ub[i];
      const float M = fmaxf(m, x);
      s = s * expf(m - M) + expf(x - M);
      m = M;
    }
  }
  sm[rg][col] = m;
  ss[rg][col] = s;
  __syncthreads();
  if (rg == 0 && j <= n1) {
#pragma unroll
    for (int g = 1; g < 4; ++g) lse_merge(m, s, sm[g][col], ss[g][col]);
    v[(long long)b * (n1 + 1) + j] = (j == n1 ? log_last : norm) - (m + logf(s));
  }
}

__global__ __launch_bounds__(256) void sink_final_kernel(const float* __restrict__ Z, const float* __restrict__ u,
                                                         const float* __restrict__ v, float norm,
                                                         const float* __restrict__ scr0, const float* __restrict__ scr1,
                                                         float* __restrict__ out, float* __restrict__ kp,
                                                         float* __restrict__ fin, int n0, int n1) {
  const int b = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n1) return;
  const float z = Z[((long long)b * (n0 + 1) + i) * (n1 + 1) + j];
  const float pr = expf(z + u[(long long)b * (n0 + 1) + i] + v[(long long)b * (n1 + 1) + j] - norm);
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

extern "C" {

long long mk_dual_softmax_work_floats(int B, int n0, int n1) {
  const int nmax = n0 > n1 ? n0 : n1;
  return (long long)B * 2 * NCHUNK * nmax * 2;
}

int mk_dual_softmax(const float* dsc0, const float* dsc1, const float* scr0, const float* scr1, float inv_temperature,
                    int use_dustbin, float dustbin, float* scores, float* kp_scores, float* final_scores, float* work, int B,
                    int C, int n0, int n1, mk_stream_t stream) {
  MK_CHECK_ARG(dsc0 && dsc1 && work, "mk_dual_softmax: null pointer");
  MK_CHECK_ARG(B > 0 && n0 > 0 && n1 > 0 && C > 0 && C <= CMAX && C % 2 == 0, "mk_dual_softmax: need 0 < C <= %d, C even", CMAX);
  MK_CHECK_ARG((scr0 && scr1) || (!kp_scores && !final_scores), "mk_dual_softmax: kp/final scores need scr0 and scr1");
  hipStream_t st = (hipStream_t)stream;
  const int nmax = n0 > n1 ? n0 : n1;
  const int rt = (nmax + MT - 1) / MT;
  hipLaunchKernelGGL(lse_partial_kernel, dim3(rt, NCHUNK, B * 2), dim3(256), 0, st, dsc0, dsc1, inv_temperature, work, C, n0, n1,
                     nmax);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(dual_softmax_write_kernel, dim3((n1 + MT - 1) / MT, (n0 + MT - 1) / MT, B), dim3(256), 0, st, dsc0, dsc1,
                     scr0, scr1, inv_temperature, use_dustbin, dustbin, work, scores, kp_scores, final_scores, C, n0, n1, nmax);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

long long mk_sinkhorn_work_floats(int B, int n0, int n1) {
  return (long long)B * ((long long)(n0 + 1) * (n1 + 1) + (n0 + 1) + (n1 + 1));
}

int mk_sinkhorn(const float* dsc0, const float* dsc1, const float* scr0, const float* scr1, float alpha, int iters, float* scores,
                float* kp_scores, float* final_scores, float* work, int B, int C, int n0, int n1, mk_stream_t stream) {
  MK_CHECK_ARG(dsc0 && dsc1 && work && (scores || final_scores), "mk_sinkhorn: null pointer");
  MK_CHECK_ARG((scr0 && scr1) || (!kp_scores && !final_scores), "mk_sinkhorn: kp/final scores need scr0 and scr1");
  MK_CHECK_ARG(B > 0 && n0 > 0 && n1 > 0 && C > 0 && C <= CMAX && C % 2 == 0 && iters >= 0, "mk_sinkhorn: bad args");
  hipStream_t st = (hipStream_t)stream;
  float* Z = work;
  float* u = Z + (long long)B * (n0 + 1) * (n1 + 1);
  float* v = u + (long long)B * (n0 + 1);
  // constants of log_optimal_transport (feature_matcher.py:116-118)
  const float norm = -logf((float)n0 + (float)n1);
  const float log_mu_last = logf((float)n1) + norm, log_nu_last = logf((float)n0) + norm;
  hipLaunchKernelGGL(couplings_kernel, dim3((n1 + 1 + MT - 1) / MT, (n0 + 1 + MT - 1) / MT, B), dim3(256), 0, st, dsc0, dsc1,
                     1.0f / sqrtf((float)C), alpha, Z, C, n0, n1);
  MK_CHECK_LAUNCH();
  const long long nuv = (long long)B * (n0 + n1 + 2);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nuv + 255) / 256)), dim3(256), 0, st, u, 0.f, nuv);
  MK_CHECK_LAUNCH();
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(sink_row_kernel, dim3((n0 + 1 + 3) / 4, B), dim3(256), 0, st, Z, v, u, n0, n1, norm, log_mu_last);
    hipLaunchKernelGGL(sink_col_kernel, dim3((n1 + 1 + 63) / 64, B), dim3(256), 0, st, Z, u, v, n0, n1, norm, log_nu_last);
  }
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sink_final_kernel, dim3((n1 + 255) / 256, n0, B), dim3(256), 0, st, Z, u, v, norm, scr0, scr1, scores, kp_scores,
                     final_scores, n0, n1);
  MK_CHECK_LAUNCH();
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
