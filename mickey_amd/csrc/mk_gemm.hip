// mickey_amd -- 16-bit-operand MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] (+)= A[M,K] . W[N,K]^T        A, W: bf16 or fp16, K contiguous; fp32 accumulate
//
// One kernel body serves every dense contraction of the MicKey hot path:
//   * ViT linears (reference DINO_modules/layers/attention.py:44,51,60; mlp.py:30-39) with the
//     bias / GELU / LayerScale+residual / QKV-split epilogues fused,
//   * the 14x14 patch embedding (reference layers/patch_embed.py:66,76) after an im2col pass,
//   * the heads' 3x3 convolutions as an implicit GEMM (reference utils/extractor_utils.py:18-35):
//     the A tile is gathered straight from the NHWC activation (one 3x3 tap per 64-wide K tile,
//     out-of-image taps read a zero page), BatchNorm is folded into W/bias on the host and the
//     1x1 shortcut conv rides along as extra K columns.
//
// Structure (MI355X): 128x128x64 tile, 256 threads = 4 waves in a 2x2 grid, each wave 64x64 via
// 4x4 v_mfma_f32_16x16x32 fragments.  Operands go HBM -> LDS directly (global_load_lds, 16 B/lane),
// double-buffered, one barrier per K tile.  The LDS image is lane-linear, so the XOR swizzle that
// makes the ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address and
// to the read address (never to the LDS destination).  MFMA operands are swapped (A-operand = W rows,
// B-operand = activation rows) so that each lane ends up with 4 CONSECUTIVE output features of one
// row: epilogue loads/stores are 8-16 B per lane.  Block ids are remapped so that the blocks of one
// XCD (private L2) walk a contiguous range of tiles.
#include "mk_common.hpp"

namespace {

using namespace mk;

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

enum AMode { A_DENSE = 0, A_CONV3 = 1 };

struct GemmParams {
  // operands
  const void* A;       // dense: [M, lda]; conv: NHWC activation of source 1
  const void* A2;      // conv only: NHWC activation of source 2 (1x1 shortcut), may be null
  const void* W;       // [N, ldw]
  int M, N, K, lda, ldw;
  long long strideA_g, strideA2_g, strideW_g;  // element strides per group (blockIdx.y)
  // conv geometry
  int H, Wd, C1, C2;   // image grid, channels of source 1 / source 2
  const void* zero_page;
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
  const void* resid_lp;  // identity residual, [M, ldc] low precision
  // qkv split
  void* q;
  void* k;
  void* vt;
  int ntok, ntok_pad, heads;
  float qscale;
  // patch embed
  const float* pos;
  int npatch;
};

template <typename T>
__device__ __forceinline__ T to_lp(float v) { return (T)v; }

// swap bits 2 and 3 of a token index: the V^T image is stored key-permuted so that the 8 keys a lane
// owns after the 32x32 S^T MFMA are one contiguous 16-B chunk (see mk_attention.hip)
__device__ __forceinline__ int vperm(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }

template <typename T, int AMODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(GemmParams p) {
  using V8 = typename Lp<T>::V8;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [stage][A|W]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.y;

  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int tile_m = id / ntn, tile_n = id % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* A = (const T*)p.A + (long long)g * p.strideA_g;
  const T* A2 = p.A2 ? (const T*)p.A2 + (long long)g * p.strideA2_g : nullptr;
  const T* W = (const T*)p.W + (long long)g * p.strideW_g;

  // ---- per-lane staging state: this lane feeds rows wave*32 + j*8 + (lane>>3), chunk lane&7 ----
  const int srow = lane >> 3, sp = lane & 7;
  const T* wrow[4];
  long long aoff[4];  // dense: element offset of (row, swizzled chunk); conv: pixel index of the row
  int ay[4], ax[4];
  bool avalid[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 32 + j * 8 + srow;
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    wrow[j] = W + (long long)n * p.ldw + swz8(r, sp) * 8;
    int m = m0 + r;
    avalid[j] = m < p.M;
    m = avalid[j] ? m : p.M - 1;
    if (AMODE == A_DENSE) {
      aoff[j] = (long long)m * p.lda + swz8(r, sp) * 8;
      ay[j] = ax[j] = 0;
    } else {
      const int pix = m % (p.H * p.Wd);
      ay[j] = pix / p.Wd;
      ax[j] = pix % p.Wd;
      aoff[j] = m;
    }
  }

  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * 2 * TILE_BYTES;
    char* sW = sA + TILE_BYTES;
    const int k0 = kt * BK;
    if (AMODE == A_DENSE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) glds16(A + aoff[j] + k0, sA + (wave * 32 + j * 8) * 128);
    } else {
      // wave-uniform: which source / tap does this K tile belong to
      const int kc = 9 * p.C1;
      const T* src;
      int cs, c0, dy, dx;
      if (k0 < kc) {
        const int tap = k0 / p.C1;
        c0 = k0 - tap * p.C1;
        dy = tap / 3 - 1;
        dx = tap % 3 - 1;
        src = A;
        cs = p.C1;
      } else {
        c0 = k0 - kc;
        dy = dx = 0;
        src = A2;
        cs = p.C2;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wave * 32 + j * 8 + srow;
        const long long pixm = aoff[j];
        const int yy = ay[j] + dy, xx = ax[j] + dx;
        const bool ok = avalid[j] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
        const T* s = ok ? src + (pixm + dy * p.Wd + dx) * cs + c0 + swz8(r, sp) * 8 : (const T*)p.zero_page + sp * 8;
        glds16(s, sA + (wave * 32 + j * 8) * 128);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(wrow[j] + k0, sW + (wave * 32 + j * 8) * 128);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sA = smem + (kt & 1) * 2 * TILE_BYTES;
    const char* sW = sA + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      V8 wf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rw = wn * 64 + i * 16 + fr;
        wf[i] = *(const V8*)(sW + rw * 128 + swz8(rw, ks * 4 + fg) * 16);
        const int rx = wm * 64 + i * 16 + fr;
        xf[i] = *(const V8*)(sA + rx * 128 + swz8(rx, ks * 4 + fg) * 16);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Lp<T>::mma16(wf[ni], xf[mi], acc[mi][ni]);
    }
  }

  // ---- epilogue: lane owns row m = ...+(lane&15), features n..n+3 with n = ...+(lane>>4)*4 ----
  const float* bias = p.bias ? p.bias + (long long)g * p.strideBias_g : nullptr;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + fr;
    if (m >= p.M) continue;
    int img = 0, tok = 0;
    if (p.epi == MK_EPI_QKV) {
      img = m / p.ntok;
      tok = m - img * p.ntok;
    } else if (p.epi == MK_EPI_PATCH) {
      img = m / p.npatch;
      tok = m - img * p.npatch;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + fg * 4;
      if (n >= p.N) continue;  // N is a multiple of 4 (checked on the host)
      f32x4 v = acc[mi][ni];
      if (bias) {
        const f32x4 b = *(const f32x4*)(bias + n);
        v += b;
      }
      switch (p.epi) {
        case MK_EPI_STORE: {
          if (p.resid_lp) {
            const typename Lp<T>::V4 r = *(const typename Lp<T>::V4*)((const T*)p.resid_lp + (long long)g * p.strideOut_g +
                                                                  (long long)m * p.ldc + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
          }
          if (p.act == MK_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (p.act == MK_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          }
          if (p.out_f32) {
            *(f32x4*)(p.out_f32 + (long long)g * p.strideOut_g + (long long)m * p.ldc + n) = v;
          } else {
            typename Lp<T>::V4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = to_lp<T>(v[e]);
            *(typename Lp<T>::V4*)((T*)p.out_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + n) = o;
          }
        } break;
        case MK_EPI_LS_RESIDUAL: {
          float* x = p.out_f32 + (long long)m * p.ldc + n;
          const f32x4 gm = *(const f32x4*)(p.gamma + n);
          f32x4 r = *(const f32x4*)x;
          r += gm * v;
          *(f32x4*)x = r;
        } break;
        case MK_EPI_PATCH: {
          const f32x4 pe = *(const f32x4*)(p.pos + (long long)(1 + tok) * p.N + n);
          *(f32x4*)(p.out_f32 + ((long long)img * (p.npatch + 1) + 1 + tok) * p.ldc + n) = v + pe;
        } break;
        case MK_EPI_QKV: {
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
            if (which == 0) v *= p.qscale;
            typename Lp<T>::V4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = to_lp<T>(v[e]);
            T* base = (T*)(which == 0 ? p.q : p.k);
            *(typename Lp<T>::V4*)(base + (hb * p.ntok_pad + tok) * 64 + d) = o;
          }
        } break;
      }
    }
  }
}

template <int AMODE>
int launch(const GemmParams& p, int groups, int dtype, hipStream_t st) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, groups, 1);
  if (dtype == MK_BF16)
    hipLaunchKernelGGL((gemm_kernel<__bf16, AMODE>), grid, dim3(NTHREADS), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_kernel<_Float16, AMODE>), grid, dim3(NTHREADS), 0, st, p);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int check_common(const GemmParams& p, int dtype) {
  MK_CHECK_ARG(dtype == MK_BF16 || dtype == MK_F16, "gemm: dtype must be MK_BF16 or MK_F16");
  MK_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  MK_CHECK_ARG(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
  MK_CHECK_ARG(p.N % 4 == 0, "gemm: N=%d must be a multiple of 4", p.N);
  MK_CHECK_ARG(p.ldw % 8 == 0 && p.ldw >= p.K, "gemm: ldw=%d must be >= K and a multiple of 8", p.ldw);
  MK_CHECK_ARG(p.A && p.W, "gemm: null operand");
  return MK_OK;
}

}  // namespace

extern "C" {

int mk_gemm(const void* A, int lda, const void* W, int ldw, const float* bias, void* out, int ldc, int M, int N, int K,
            int act, int out_is_f32, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_STORE; p.act = act; p.bias = bias; p.ldc = ldc;
  if (out_is_f32) p.out_f32 = (float*)out; else p.out_lp = out;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(lda % 8 == 0 && lda >= K && ldc % 4 == 0 && ldc >= N && out, "mk_gemm: bad lda/ldc/out");
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

int mk_gemm_grouped(const void* A, int lda, long long strideA, const void* W, int ldw, long long strideW, const float* bias,
                    long long strideBias, void* out, int ldc, long long strideOut, int groups, int M, int N, int K, int act,
                    int out_is_f32, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
  p.strideA_g = strideA; p.strideW_g = strideW; p.strideBias_g = strideBias; p.strideOut_g = strideOut;
  p.epi = MK_EPI_STORE; p.act = act; p.bias = bias; p.ldc = ldc;
  if (out_is_f32) p.out_f32 = (float*)out; else p.out_lp = out;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(groups > 0 && lda % 8 == 0 && lda >= K && ldc % 4 == 0 && ldc >= N && out, "mk_gemm_grouped: bad args");
  return launch<A_DENSE>(p, groups, dtype, (hipStream_t)stream);
}

int mk_gemm_ls_residual(const void* A, int lda, const void* W, int ldw, const float* bias, const float* gamma, float* x,
                        int ldx, int M, int N, int K, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_LS_RESIDUAL; p.bias = bias; p.gamma = gamma; p.out_f32 = x; p.ldc = ldx;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(gamma && x && lda % 8 == 0 && lda >= K && ldx % 4 == 0 && ldx >= N, "mk_gemm_ls_residual: bad args");
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

int mk_gemm_qkv(const void* A, int lda, const void* W, int ldw, const float* bias, void* q, void* k, void* vt, int nimg,
                int ntok, int ntok_pad, int heads, float qscale, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  const int D = heads * 64;
  p.A = A; p.W = W; p.M = nimg * ntok; p.N = 3 * D; p.K = D; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_QKV; p.bias = bias; p.q = q; p.k = k; p.vt = vt;
  p.ntok = ntok; p.ntok_pad = ntok_pad; p.heads = heads; p.qscale = qscale;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(q && k && vt && ntok_pad % 64 == 0 && ntok_pad >= ntok && lda % 8 == 0 && lda >= D, "mk_gemm_qkv: bad args");
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

int mk_gemm_patch_embed(const void* A, int lda, const void* W, int ldw, const float* bias, const float* pos, float* x,
                        int nimg, int npatch, int D, int K, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = A; p.W = W; p.M = nimg * npatch; p.N = D; p.K = K; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_PATCH; p.bias = bias; p.pos = pos; p.npatch = npatch; p.out_f32 = x; p.ldc = D;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(pos && x && lda % 8 == 0 && lda >= K, "mk_gemm_patch_embed: bad args");
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

int mk_conv3x3(const void* in1, long long stride_in1, int C1, const void* in2, long long stride_in2, int C2, const void* W,
               int ldw, long long strideW, const float* bias, long long strideBias, const void* resid, void* out, int Cout,
               long long strideOut, int groups, int nimg, int H, int Wd, int act, int out_is_f32, const void* zero_page,
               int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = in1; p.A2 = in2; p.W = W;
  p.M = nimg * H * Wd; p.N = Cout; p.K = 9 * C1 + (in2 ? C2 : 0);
  p.ldw = ldw; p.strideA_g = stride_in1; p.strideA2_g = stride_in2; p.strideW_g = strideW;
  p.strideBias_g = strideBias; p.strideOut_g = strideOut;
  p.H = H; p.Wd = Wd; p.C1 = C1; p.C2 = in2 ? C2 : 0; p.zero_page = zero_page;
  p.epi = MK_EPI_STORE; p.act = act; p.bias = bias; p.ldc = Cout; p.resid_lp = resid;
  if (out_is_f32) p.out_f32 = (float*)out; else p.out_lp = out;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(C1 % BK == 0 && (!in2 || C2 % BK == 0), "mk_conv3x3: channel counts must be multiples of %d", BK);
  MK_CHECK_ARG(zero_page && out && groups > 0 && H > 0 && Wd > 0, "mk_conv3x3: bad args");
  return launch<A_CONV3>(p, groups, dtype, (hipStream_t)stream);
}

}  // extern "C"
