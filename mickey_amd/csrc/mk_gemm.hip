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
// Structure (MI355X): two instantiations of one body -- 128x128x64 tile / 4 waves (2x2, 64x64 per wave) and
// 256x256x64 tile / 8 waves (2x4, 128x64 per wave) -- built from v_mfma_f32_16x16x32 fragments.  Operands go HBM -> LDS directly (global_load_lds, 16 B/lane),
// double-buffered, one barrier per K tile.  The LDS image is lane-linear, so the XOR swizzle that
// makes the ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address and
// to the read address (never to the LDS destination).  MFMA operands are swapped (A-operand = W rows,
// B-operand = activation rows) so that each lane ends up with 4 CONSECUTIVE output features of one
// row: epilogue loads/stores are 8-16 B per lane.  Block ids are remapped so that the blocks of one
// XCD (private L2) walk a contiguous range of tiles.
#include "mk_common.hpp"

namespace {

using namespace mk;

constexpr int BK = 64;

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

// Per-lane LDS-DMA state of one workgroup tile.  Piece (wave*J + j) is 8 rows x 128 B = 1 KiB of the LDS image; this
// lane feeds row +(lane>>3), 16-byte chunk lane&7 of it, fetching the XOR-swizzled source chunk.
template <typename T, int AMODE, int NW, int AJ, int WJ>
struct Stager {
  const T* A;
  const T* A2;
  const T* wrow[WJ];
  long long aoff[AJ];  // dense: element offset of (row, swizzled chunk); conv: pixel index of the row
  int ay[AJ], ax[AJ];
  bool avalid[AJ];
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
      wrow[j] = W + (long long)n * p.ldw + swz8(r, sp) * 8;
    }
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int r = (wave * AJ + j) * 8 + srow;
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
  }

  __device__ __forceinline__ void issue(const GemmParams& p, char* sA, char* sW, int kt) const {
    const int k0 = kt * BK;
    if (AMODE == A_DENSE) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) glds16(A + aoff[j] + k0, sA + (wave * AJ + j) * 1024);
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
      for (int j = 0; j < AJ; ++j) {
        const int r = (wave * AJ + j) * 8 + srow;
        const int yy = ay[j] + dy, xx = ax[j] + dx;
        const bool ok = avalid[j] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
        const T* s = ok ? src + (aoff[j] + dy * p.Wd + dx) * cs + c0 + swz8(r, sp) * 8 : (const T*)p.zero_page + sp * 8;
        glds16(s, sA + (wave * AJ + j) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) glds16(wrow[j] + k0, sW + (wave * WJ + j) * 1024);
  }
};

// ---- epilogue: lane owns row m = ...+(lane&15), features n..n+3 with n = ...+(lane>>4)*4 ----
// EPI / ACT / HAS_BIAS are compile-time inside the 32x unrolled store loop; epilogue() dispatches once per tile.
template <typename T, int WMF, int EPI, int ACT, bool HAS_BIAS>
__device__ __forceinline__ void epilogue_impl(const GemmParams& p, f32x4 (&acc)[WMF][4], int m0, int n0, int wm, int wn, int lane,
                                              int g) {
  using V4 = typename Lp<T>::V4;
  const int fr = lane & 15, fg = lane >> 4;
  const float* bias = HAS_BIAS ? p.bias + (long long)g * p.strideBias_g : nullptr;
  const int nb = n0 + wn * 64 + fg * 4;
  f32x4 bv[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = nb + ni * 16;
    bv[ni] = (HAS_BIAS && n < p.N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int mi = 0; mi < WMF; ++mi) {
    const int m = m0 + wm * (WMF * 16) + mi * 16 + fr;
    if (m >= p.M) continue;
    int img = 0, tok = 0;
    if (EPI == MK_EPI_QKV) {
      img = m / p.ntok;
      tok = m - img * p.ntok;
    } else if (EPI == MK_EPI_PATCH) {
      img = m / p.npatch;
      tok = m - img * p.npatch;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = nb + ni * 16;
      if (n >= p.N) continue;  // N is a multiple of 4 (checked on the host)
      f32x4 v = acc[mi][ni];
      if (HAS_BIAS) v += bv[ni];
      if (EPI == MK_EPI_STORE) {
        if (p.resid_lp) {
          const V4 r = *(const V4*)((const T*)p.resid_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
        }
        if (ACT == MK_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (ACT == MK_ACT_GELU) {
          v = gelu_erf4(v);
        }
        if (p.out_f32) {
          *(f32x4*)(p.out_f32 + (long long)g * p.strideOut_g + (long long)m * p.ldc + n) = v;
        } else {
          V4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = to_lp<T>(v[e]);
          *(V4*)((T*)p.out_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + n) = o;
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
          if (which == 0) v *= p.qscale;
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
                                         int g) {
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

// WMF = 16-row M fragments per wave (4 -> 64 rows, 8 -> 128 rows); waves in an NWM x NWN grid, each wave 64 columns.
//   <4,2,2>: 128x128 tile, 256 threads, 64 KiB LDS  (2 workgroups / CU)  -- small / skinny problems
//   <8,2,4>: 256x256 tile, 512 threads, 128 KiB LDS (1 workgroup / CU)
// Two LDS stages, the LDS-DMA of the next K tile is issued before the MFMAs of the current one, one barrier per K
// tile.  PERSIST: the workgroup walks a sequence of output tiles and treats their K tiles as ONE stream -- the DMA of
// the next tile's first K tile is issued before the last MFMAs of the current tile, and the epilogue's stores drain
// under the next tile's main loop -- which removes the per-tile prologue bubble (no other workgroup shares the CU
// to hide it when the tile needs 128 KiB of LDS).
template <typename T, int AMODE, int WMF, int NWM, int NWN, bool PERSIST>
__global__ __launch_bounds__(NWM* NWN * 64, (NWM * NWN) / 4) void gemm_kernel(GemmParams p) {
  using V8 = typename Lp<T>::V8;
  constexpr int NW = NWM * NWN, BM = NWM * WMF * 16, BN = NWN * 64;
  constexpr int AJ = BM / 8 / NW, WJ = BN / 8 / NW;          // 1-KiB LDS-DMA pieces per wave per K tile
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A tile | W tile], rows of 128 B

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int ntiles = ntm * ntn;
  const int nk = p.K / BK;
  const int fr = lane & 15, fg = lane >> 4;

  // tile sequence of this workgroup: ids blockIdx.x, +gridDim.x, ...; xcd_remap keeps the tiles that one XCD works on
  // at any time adjacent (same A panel / neighbouring W panels in its L2)
  int seq = blockIdx.x;
  int id = xcd_remap(seq, ntiles);
  int m0 = (id / ntn) * BM, n0 = (id % ntn) * BN;
  Stager<T, AMODE, NW, AJ, WJ> st;
  st.init(p, g, m0, n0, wave, lane);
  st.issue(p, smem, smem + A_BYTES, 0);
  int gi = 0;  // position in the K-tile stream (selects the LDS stage)
  for (;;) {
    f32x4 acc[WMF][4];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int cm0 = m0, cn0 = n0;
    const int nseq = seq + gridDim.x;
    const bool more = PERSIST && nseq < ntiles;
    for (int kt = 0; kt < nk; ++kt, ++gi) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      char* nA = smem + ((gi + 1) & 1) * STAGE_BYTES;
      if (kt + 1 < nk) {
        st.issue(p, nA, nA + A_BYTES, kt + 1);
      } else if (more) {  // first K tile of the next output tile
        seq = nseq;
        id = xcd_remap(seq, ntiles);
        m0 = (id / ntn) * BM;
        n0 = (id % ntn) * BN;
        st.init(p, g, m0, n0, wave, lane);
        st.issue(p, nA, nA + A_BYTES, 0);
      }
      const char* sA = smem + (gi & 1) * STAGE_BYTES;
      const char* sW = sA + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        V8 wf[4], xf[WMF];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rw = wn * 64 + i * 16 + fr;
          wf[i] = *(const V8*)(sW + rw * 128 + swz8(rw, ks * 4 + fg) * 16);
        }
#pragma unroll
        for (int i = 0; i < WMF; ++i) {
          const int rx = wm * (WMF * 16) + i * 16 + fr;
          xf[i] = *(const V8*)(sA + rx * 128 + swz8(rx, ks * 4 + fg) * 16);
        }
#pragma unroll
        for (int mi = 0; mi < WMF; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Lp<T>::mma16(wf[ni], xf[mi], acc[mi][ni]);
      }
    }
    epilogue<T, WMF>(p, acc, cm0, cn0, wm, wn, lane, g);
    if (!more) break;
  }
}

// ---------------------------------------------------------------------------------------------------------
// "Ping-pong" 256x256 kernel: 8 waves = 2 wave-rows x 4, 128x64 outputs per wave, K tiles of 32, 4-deep LDS ring.
// The two wave-rows -- whose waves share SIMDs pairwise (w, w+4) -- run half a K tile apart: in every
// barrier-delimited slot one wave-row executes its 32 MFMAs of a K tile from REGISTERS (its 12 fragments were
// preloaded in the previous slot) while the other one issues LDS-DMA and reads its fragments, so each SIMD's matrix
// pipe is fed by one wave while its partner does the memory work (s_setprio favours the MFMA wave).
//   even slot 2kt  : wait(stage kt landed: counted vmcnt), barrier | row0: read frags(kt) + DMA a(kt+3) | row1: MFMA(kt-1) + DMA w(kt+2)
//   odd  slot 2kt+1: barrier                                       | row0: MFMA(kt) + DMA w(kt+3)       | row1: read frags(kt) + DMA a(kt+3)
// Each wave issues its own 4 LDS-DMA pieces of a tile (2 A pieces in its read slot, 2 W pieces in the following MFMA
// slot) into the ring slot of a tile whose last reader is at least one barrier behind; a DMA has >= 4 slots to land.
// Barriers are bare s_barrier: nothing ever drains the DMA queue to zero inside the loop.
constexpr int PK = 32;          // K tile of the ping-pong kernel
constexpr int PSTAGES = 4;

// 64-byte LDS rows (4 chunks of 16 B): chunk' = chunk ^ ((-(row >> 2)) & 3) is conflict-free for ds_read_b128
__device__ __forceinline__ int swz4(int row, int chunk) { return chunk ^ ((0 - (row >> 2)) & 3); }

// ABL: timing-ablation bits for tools/ (results are wrong when != 0): 1 = no LDS-DMA, 2 = no fragment reads, 4 = no barriers
template <typename T, int AMODE, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmParams p) {
  using V8 = typename Lp<T>::V8;
  constexpr int WMF = 8, NWN = 4, BM = 256, BN = 256;
  constexpr int A_BYTES = BM * 64, STAGE_BYTES = (BM + BN) * 64;   // 32 KiB per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;   // wm = wave-row = ping-pong group
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BN;

  // ---- LDS-DMA state: piece = 16 rows x 64 B; this wave owns pieces 2*wave, 2*wave+1 of A and of W ----
  const T* A = (const T*)p.A + (long long)g * p.strideA_g;
  const T* A2 = p.A2 ? (const T*)p.A2 + (long long)g * p.strideA2_g : nullptr;
  const T* W = (const T*)p.W + (long long)g * p.strideW_g;
  const int srow = lane >> 2, sp = lane & 3;
  const T* wrow[2];
  long long aoff[2];
  int ay[2], ax[2];
  bool avalid[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + srow;
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    wrow[j] = W + (long long)n * p.ldw + swz4(r, sp) * 8;
    int m = m0 + r;
    avalid[j] = m < p.M;
    m = avalid[j] ? m : p.M - 1;
    if (AMODE == A_DENSE) {
      aoff[j] = (long long)m * p.lda + swz4(r, sp) * 8;
      ay[j] = ax[j] = 0;
    } else {
      const int pix = m % (p.H * p.Wd);
      ay[j] = pix / p.Wd;
      ax[j] = pix % p.Wd;
      aoff[j] = m;
    }
  }
  // one LDS-DMA instruction: piece q of K tile kt (q = 0,1: A pieces; 2,3: W pieces).  Every wave issues exactly
  // these 4 per K tile, in this order (the vmcnt arithmetic relies on it)
  f32x4 dummy = {0.f, 0.f, 0.f, 0.f};
  auto issue_piece = [&](int kt, int q) {
    if (ABL & 1) return;
    char* sA = smem + (kt % PSTAGES) * STAGE_BYTES;
    char* sW = sA + A_BYTES;
    const int k0 = (ABL & 8) ? 0 : kt * PK;   // ABL 8: always the same (cache-hot) source addresses
    if (ABL & 32) {   // timing only: the same bytes through plain 16-B global loads into VGPRs (no LDS write)
      const T* src = (q >= 2 ? wrow[q - 2] : A + aoff[q]) + k0;
      const f32x4 v = *(const f32x4*)src;
      dummy += v;
      return;
    }
    if (ABL & 16) {   // timing only: same instruction count, but each piece touches 8 full 128-B lines instead of 16 halves
      const T* base = (q >= 2 ? wrow[q - 2] : A + aoff[q]) - (long long)(lane >> 2) * (q >= 2 ? p.ldw : p.lda);
      glds16(base + (long long)(lane >> 3) * (q >= 2 ? p.ldw : p.lda) + ((lane >> 2) & 1) * 32 + k0,
             (q >= 2 ? sW + (wave * 2 + (q - 2)) * 1024 : sA + (wave * 2 + q) * 1024));
      return;
    }
    if (q >= 2) {
      glds16(wrow[q - 2] + k0, sW + (wave * 2 + (q - 2)) * 1024);
    } else if (AMODE == A_DENSE) {
      glds16(A + aoff[q] + k0, sA + (wave * 2 + q) * 1024);
    } else {
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
      const int r = (wave * 2 + q) * 16 + srow;
      const int yy = ay[q] + dy, xx = ax[q] + dx;
      const bool ok = avalid[q] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
      const T* s = ok ? src + (aoff[q] + dy * p.Wd + dx) * cs + c0 + swz4(r, sp) * 8 : (const T*)p.zero_page + sp * 8;
      glds16(s, sA + (wave * 2 + q) * 1024);
    }
  };
  auto issue = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(kt, q);
  };

  f32x4 acc[WMF][4];
#pragma unroll
  for (int i = 0; i < WMF; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  V8 wf[4], xf[WMF];

  const int fr = lane & 15, fg = lane >> 4;
  auto load_frags = [&](int kt) {
    if ((ABL & 2) && kt > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(wf[i]));
#pragma unroll
      for (int i = 0; i < WMF; ++i) asm volatile("" : "+v"(xf[i]));
      return;
    }
    const char* sA = smem + (kt % PSTAGES) * STAGE_BYTES;
    const char* sW = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rw = wn * 64 + i * 16 + fr;
      wf[i] = *(const V8*)(sW + rw * 64 + swz4(rw, fg) * 16);
    }
#pragma unroll
    for (int i = 0; i < WMF; ++i) {
      const int rx = wm * (WMF * 16) + i * 16 + fr;
      xf[i] = *(const V8*)(sA + rx * 64 + swz4(rx, fg) * 16);
    }
  };
  const int nk = p.K / PK;
  // 32 MFMAs from registers.  An LDS-DMA instruction stalls the issuing wave's MFMA stream for ~60-100 cycles
  // (measured: all 4 pieces of a tile in this slot cost 19 %), so only pieces 2,3 (W) of tile `kt_dma` ride here;
  // pieces 0,1 (A) are issued from the fragment-read slot, which has issue slack
  auto mfma_tile = [&](int kt_dma) {
    const bool dma = kt_dma < nk;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < WMF; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Lp<T>::mma16(wf[ni], xf[mi], acc[mi][ni]);
      if (dma && mi == 2) issue_piece(kt_dma, 2);
      if (dma && mi == 5) issue_piece(kt_dma, 3);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto read_slot = [&](int kt) {   // fragment reads of tile kt + the first half of the DMA of tile kt+3
    load_frags(kt);
    if (kt + 3 < nk) {
      issue_piece(kt + 3, 0);
      issue_piece(kt + 3, 1);
    }
  };

  issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
  // even-slot entry: wait until this wave's LDS-DMA of tile kt has landed = at most `allow` newer DMA instructions
  // outstanding, then the barrier makes every wave's share of tile kt visible.  Wave-row 0 has issued tiles <= kt+2
  // completely at that point; wave-row 1 has issued tile kt+1 completely and pieces 0,1 of tile kt+2.
  auto even_entry = [&](int kt, int half_issued) {
    const int newer = nk - 1 - kt;
    if (newer >= 2) {
      if (half_issued)
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    } else if (newer == 1) {
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
  };
  auto odd_entry = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
  };
  // the two wave-rows run separate straight-line loops (same number of barriers per K tile), so the accumulators
  // never flow through a conditional merge
  if (wm == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      even_entry(kt, 0);
      read_slot(kt);       // slot 2kt  : ring slot (kt+3)%4 was last read in slot 2kt-1
      odd_entry();
      mfma_tile(kt + 3);   // slot 2kt+1
    }
  } else {
    // same barrier sequence, loop boundary shifted by one slot so that fragments are loaded and consumed inside
    // one iteration (no loop-carried fragment registers)
    even_entry(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      odd_entry();
      read_slot(kt);                          // slot 2kt+1
      if (kt + 1 < nk) even_entry(kt + 1, 1);
      mfma_tile(kt + 3);                      // slot 2kt+2
    }
  }
  if (ABL & 32) acc[0][0] += dummy * 1e-30f;
  epilogue<T, WMF>(p, acc, m0, n0, wm, wn, lane, g);
}

// ---------------------------------------------------------------------------------------------------------
// LDS-staged epilogue of the full-line ping-pong kernel.  In the accumulator layout a lane owns 4 features of one
// row, so a direct store instruction touches 16 rows x 32 B -- measured ~65 cycles per instruction and 4.4-7.2 us
// per 256x256 tile (13-21 % of a K = 1024 tile).  After the K loop the 128-KiB ring is idle: each wave bounces its
// 128x64 block through a private 16-KiB slice (wave-local, LDS is in-order per wave: no barrier) and then moves whole
// rows: 16-bit outputs 16 B per lane = 8 full 128-byte lines per instruction, fp32 outputs (two 64-row halves) 4 x
// 256 B.  The residual-stream read-modify-write and the q / k head-major stores become fully coalesced the same way;
// only the V^T part of the qkv split keeps element stores (its rows are tokens at an arbitrary 16-group alignment).
// XOR swizzles: 16-bit rows of 128 B, chunk ^ (row & 7); fp32 rows of 256 B, chunk ^ (row & 15).
template <typename T, int EPI, int ACT, bool HAS_BIAS>
__device__ __forceinline__ void epilogue_lds_impl(const GemmParams& p, f32x4 (&acc)[8][4], char* wl, int m0, int n0, int wm,
                                                  int wn, int lane, int g) {
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
  f32x4 bv[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = nw + fg * 4 + ni * 16;
    bv[ni] = (HAS_BIAS && n < p.N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool lp_out = EPI == MK_EPI_QKV || (EPI == MK_EPI_STORE && !p.out_f32);
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
            const f32x4 v = acc[mi][ni] + bv[ni];
            T* dst = (T*)p.vt + (hb * 64 + ni * 16 + fg * 4) * p.ntok_pad + vperm(tok);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(long long)e * p.ntok_pad] = to_lp<T>(v[e]);
          }
        }
        return;
      }
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int r = mi * 16 + fr;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        f32x4 v = acc[mi][ni];
        if (HAS_BIAS) v += bv[ni];
        if (EPI == MK_EPI_QKV) {
          if (which == 0) v *= p.qscale;
        } else {
          if (p.resid_lp) {
            const int m = mw + r, n = nw + fg * 4 + ni * 16;
            if (m < p.M && n < p.N) {
              const V4 rs = *(const V4*)((const T*)p.resid_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + n);
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
        const int c = ni * 2 + (fg >> 1);
        *(V4*)(wl + r * 128 + ((c ^ (r & 7)) << 4) + (fg & 1) * 8) = o;
      }
    }
    const int rr = lane >> 3, c = lane & 7;
    const int n = nw + c * 8;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = it * 8 + rr;
      const int m = mw + r;
      const V8 val = *(const V8*)(wl + r * 128 + ((c ^ (r & 7)) << 4));
      if (m >= p.M || n >= p.N) continue;
      T* dst;
      if (EPI == MK_EPI_QKV) {
        int img, tok;
        img_tok(r, img, tok);
        dst = (T*)(which == 0 ? p.q : p.k) + (((long long)img * p.heads + head) * p.ntok_pad + tok) * 64 + c * 8;
      } else {
        dst = (T*)p.out_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + n;
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
  } else {
    const int rr = lane >> 4, c = lane & 15;
    const int n = nw + c * 4;
    f32x4 gm = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI == MK_EPI_LS_RESIDUAL && n < p.N) gm = *(const f32x4*)(p.gamma + n);
    // read-modify-write of the residual stream: all 16 loads of a half are issued before anything waits on them (one
    // HBM round trip per half instead of one per row group: measured 0.68 us per dependent load -> 22 us per tile);
    // the second half's loads go out while the first half is still being stored
    f32x4 xr[2][16];
    auto preload = [&](int half) {
      if (EPI != MK_EPI_LS_RESIDUAL) return;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int m = mw + half * 64 + it * 4 + rr;
        xr[half][it] = (m < p.M && n < p.N) ? *(const f32x4*)(p.out_f32 + (long long)m * p.ldc + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto stage = [&](int half) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int r = mi * 16 + fr;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          f32x4 v = acc[half * 4 + mi][ni];
          if (HAS_BIAS) v += bv[ni];
          if (EPI == MK_EPI_STORE) {
            if (p.resid_lp) {
              const int m = mw + half * 64 + r, nn = nw + fg * 4 + ni * 16;
              if (m < p.M && nn < p.N) {
                const V4 rs = *(const V4*)((const T*)p.resid_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + nn);
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
          *(f32x4*)(wl + r * 256 + ((cw ^ (r & 15)) << 4)) = v;
        }
      }
    };
    auto drain = [&](int half) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r = it * 4 + rr;
        const int m = mw + half * 64 + r;
        const f32x4 val = *(const f32x4*)(wl + r * 256 + ((c ^ (r & 15)) << 4));
        if (m >= p.M || n >= p.N) continue;
        if (EPI == MK_EPI_LS_RESIDUAL) {
          f32x4 x = xr[half][it];
          x += gm * val;
          *(f32x4*)(p.out_f32 + (long long)m * p.ldc + n) = x;
        } else if (EPI == MK_EPI_PATCH) {
          const int img = m / p.npatch, tok = m - img * p.npatch;
          const f32x4 pe = *(const f32x4*)(p.pos + (long long)(1 + tok) * p.N + n);
          *(f32x4*)(p.out_f32 + ((long long)img * (p.npatch + 1) + 1 + tok) * p.ldc + n) = val + pe;
        } else {
          *(f32x4*)(p.out_f32 + (long long)g * p.strideOut_g + (long long)m * p.ldc + n) = val;
        }
      }
    };
    preload(0);
    stage(0);
    preload(1);
    drain(0);
    stage(1);
    drain(1);
  }
}

// Same staging through a 4-KiB slice per wave (the 32 KiB of LDS beside the 128-KiB ring): for the persistent variant,
// whose ring already holds the first stages of the next tile while the epilogue runs.  16-bit outputs go in 4 passes
// of 32 rows, fp32 outputs in 8 passes of 16 rows; LDS executes a wave's accesses in order, so pass q+1 may overwrite
// the slice as soon as pass q's reads are issued.
template <typename T, int EPI, int ACT, bool HAS_BIAS>
__device__ __forceinline__ void epilogue_lds4k_impl(const GemmParams& p, f32x4 (&acc)[8][4], char* wl, int m0, int n0, int wm,
                                                    int wn, int lane, int g) {
  using V4 = typename Lp<T>::V4;
  using V8 = typename Lp<T>::V8;
  const int fr = lane & 15, fg = lane >> 4;
  const float* bias = HAS_BIAS ? p.bias + (long long)g * p.strideBias_g : nullptr;
  const int nw = n0 + wn * 64;
  const int mw = m0 + wm * 128;
  f32x4 bv[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = nw + fg * 4 + ni * 16;
    bv[ni] = (HAS_BIAS && n < p.N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool lp_out = EPI == MK_EPI_QKV || (EPI == MK_EPI_STORE && !p.out_f32);
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
          const int img = m / p.ntok, tok = m - img * p.ntok;
          const long long hb = (long long)img * p.heads + head;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const f32x4 v = acc[mi][ni] + bv[ni];
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
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const int mi = q * 2 + mh;
        const int r = mh * 16 + fr;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          f32x4 v = acc[mi][ni];
          if (HAS_BIAS) v += bv[ni];
          if (EPI == MK_EPI_QKV) {
            if (which == 0) v *= p.qscale;
          } else {
            if (p.resid_lp) {
              const int m = mw + q * 32 + r, nn = nw + fg * 4 + ni * 16;
              if (m < p.M && nn < p.N) {
                const V4 rs = *(const V4*)((const T*)p.resid_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + nn);
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
          *(V4*)(wl + r * 128 + ((cc ^ (r & 7)) << 4) + (fg & 1) * 8) = o;
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + rr;
        const int m = mw + q * 32 + r;
        const V8 val = *(const V8*)(wl + r * 128 + ((c ^ (r & 7)) << 4));
        if (m >= p.M || n >= p.N) continue;
        T* dst;
        if (EPI == MK_EPI_QKV) {
          const int img = m / p.ntok, tok = m - img * p.ntok;
          dst = (T*)(which == 0 ? p.q : p.k) + (((long long)img * p.heads + head) * p.ntok_pad + tok) * 64 + c * 8;
        } else {
          dst = (T*)p.out_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + n;
        }
        if (n + 8 <= p.N) {
          *(V8*)dst = val;
        } else {
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
    f32x4 xr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (EPI == MK_EPI_LS_RESIDUAL && (q & 1) == 0) {   // the 8 row groups of these 32 rows in one round trip
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int m = mw + q * 16 + j * 4 + rr;
          xr[j] = (m < p.M && n < p.N) ? *(const f32x4*)(p.out_f32 + (long long)m * p.ldc + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        f32x4 v = acc[q][ni];
        if (HAS_BIAS) v += bv[ni];
        if (EPI == MK_EPI_STORE) {
          if (p.resid_lp) {
            const int m = mw + q * 16 + fr, nn = nw + fg * 4 + ni * 16;
            if (m < p.M && nn < p.N) {
              const V4 rs = *(const V4*)((const T*)p.resid_lp + (long long)g * p.strideOut_g + (long long)m * p.ldc + nn);
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
        *(f32x4*)(wl + fr * 256 + ((cw ^ fr) << 4)) = v;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 4 + rr;
        const int m = mw + q * 16 + r;
        const f32x4 val = *(const f32x4*)(wl + r * 256 + ((c ^ r) << 4));
        if (m >= p.M || n >= p.N) continue;
        if (EPI == MK_EPI_LS_RESIDUAL) {
          f32x4 x = xr[(q & 1) * 4 + it];
          x += gm * val;
          *(f32x4*)(p.out_f32 + (long long)m * p.ldc + n) = x;
        } else if (EPI == MK_EPI_PATCH) {
          const int img = m / p.npatch, tok = m - img * p.npatch;
          const f32x4 pe = *(const f32x4*)(p.pos + (long long)(1 + tok) * p.N + n);
          *(f32x4*)(p.out_f32 + ((long long)img * (p.npatch + 1) + 1 + tok) * p.ldc + n) = val + pe;
        } else {
          *(f32x4*)(p.out_f32 + (long long)g * p.strideOut_g + (long long)m * p.ldc + n) = val;
        }
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void epilogue_lds4k(const GemmParams& p, f32x4 (&acc)[8][4], char* wl, int m0, int n0, int wm, int wn,
                                               int lane, int g) {
  switch (p.epi) {   // wave-uniform, once per output tile
    case MK_EPI_LS_RESIDUAL: epilogue_lds4k_impl<T, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    case MK_EPI_QKV: epilogue_lds4k_impl<T, MK_EPI_QKV, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    case MK_EPI_PATCH: epilogue_lds4k_impl<T, MK_EPI_PATCH, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    default:
      if (!p.bias) {
        if (p.act == MK_ACT_RELU) epilogue_lds4k_impl<T, MK_EPI_STORE, MK_ACT_RELU, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else if (p.act == MK_ACT_GELU) epilogue_lds4k_impl<T, MK_EPI_STORE, MK_ACT_GELU, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else epilogue_lds4k_impl<T, MK_EPI_STORE, MK_ACT_NONE, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
      } else {
        if (p.act == MK_ACT_RELU) epilogue_lds4k_impl<T, MK_EPI_STORE, MK_ACT_RELU, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else if (p.act == MK_ACT_GELU) epilogue_lds4k_impl<T, MK_EPI_STORE, MK_ACT_GELU, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else epilogue_lds4k_impl<T, MK_EPI_STORE, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
      }
  }
}

template <typename T>
__device__ __forceinline__ void epilogue_lds(const GemmParams& p, f32x4 (&acc)[8][4], char* wl, int m0, int n0, int wm, int wn,
                                             int lane, int g) {
  switch (p.epi) {   // wave-uniform, once per output tile
    case MK_EPI_LS_RESIDUAL: epilogue_lds_impl<T, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    case MK_EPI_QKV: epilogue_lds_impl<T, MK_EPI_QKV, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    case MK_EPI_PATCH: epilogue_lds_impl<T, MK_EPI_PATCH, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    default:
      if (!p.bias) {
        if (p.act == MK_ACT_RELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_RELU, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else if (p.act == MK_ACT_GELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_GELU, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_NONE, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
      } else {
        if (p.act == MK_ACT_RELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_RELU, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else if (p.act == MK_ACT_GELU) epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_GELU, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else epilogue_lds_impl<T, MK_EPI_STORE, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Ping-pong with FULL-LINE LDS-DMA pieces: 256x256 tile, 8 waves (wave-row g = wave>>2), LDS stages of K = 64
// (128-byte rows, the swz8 swizzle of the plain kernel), computed in two K = 32 sub-steps h.  A piece is 8 rows x
// 128 B (8 full cache lines; the K = 32 ring above moves 16 half lines per piece, measured 12 % slower).  Only two
// 64-KiB stages fit, which is deep enough because the operand halves are released at different times:
//   * wave-row g loads AND reads only its own A half (rows 128g..128g+127); W is loaded and read by everyone;
//   * slots (barrier at every boundary):  row g does  L(kt,h) [12 fragment reads] in slot 4kt+2h+g  and
//     C(kt,h) [32 MFMAs from registers] in slot 4kt+2h+g+1;
//   * A_g(kt+2) is issued in row g's C(kt,1) slot (its last reader, L(kt,1) of the same row, is one barrier behind);
//     W(kt+1) is issued in row g's L(kt,0) slot (the last reader of W(kt-1), row 1 in slot 4kt-1, is behind);
//   * every DMA has 3-5 slots to land; once per stage a counted vmcnt at the end of slot 4kt+3 (row 0: 4 newer DMAs
//     may stay in flight; row 1: 0) precedes the barrier that opens stage kt+1.  (Epilogue stores also count in
//     vmcnt; loads return in order among themselves, so "<= 4 outstanding" still implies the older DMAs landed.)
// Persistent: a workgroup walks tiles blockIdx.x, +gridDim.x, ... and the K stages of successive tiles form ONE stream
// (stage parity carries over), so the first stages of the next tile are in flight during the epilogue, which both
// wave-rows run in the slot that opens the next tile.  Tile order: bands of PP_GM m-tiles walked n-major, so the 32
// tiles an XCD works on at one time form an 8 x 4 block of the output (A and W panels shared in its L2).
__device__ __forceinline__ void pp_tile_coords(int id, int ntm, int ntn, int PP_GM, int& tm, int& tn) {
  const int band = id / (PP_GM * ntn);
  const int rem = id - band * (PP_GM * ntn);
  const int gm = min(PP_GM, ntm - band * PP_GM);
  tm = band * PP_GM + rem % gm;
  tn = rem / gm;
}

// DBG: dev-only instantiation that records per wave-row {entry, first stage landed, K loop done, epilogue issued}
// on the 100-MHz wall clock plus HW_ID / XCC_ID of the first tile a workgroup runs (tools/gemm_timeline.py)
// ABL (experiments, -DMK_PP64_ABLATIONS): 1 = no LDS-DMA inside the K loop, 2 = fragments read once, 4 = no barriers
// (1, 2, 4: wrong results); cache policy of the DMA loads (results stay correct): 8 = A nt, 16 = W nt, 32 = sc0, 64 = sc0 sc1
template <typename T, int AMODE, bool PERSIST, bool DBG = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp64_kernel(GemmParams p, int band_m, int stagger, unsigned long long* dbg) {
  bool abl_loop = false, abl_loaded = false;
  unsigned long long t_entry = 0, t_landed = 0, t_loop = 0;
  if (DBG) t_entry = __builtin_amdgcn_s_memrealtime();
  // All CUs start together and would hit their (HBM-heavy, un-overlapped) epilogues in lockstep while HBM idles during
  // the K loops.  Spreading the first round of workgroups over `stagger` ticks of the 100-MHz clock de-phases them.
  if (stagger > 0 && blockIdx.x < 256) {
    const unsigned h = (blockIdx.x * 2654435761u) >> 24;
    const unsigned long long until = __builtin_amdgcn_s_memrealtime() + ((unsigned long long)h * stagger >> 8);
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(16);
  }
  using V8 = typename Lp<T>::V8;
  constexpr int WMF = 8, BM = 256, BN = 256;
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;   // 64 KiB per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int ntiles = ntm * ntn;
  const int nk = p.K / BK;

  const T* A = (const T*)p.A + (long long)g * p.strideA_g;
  const T* A2 = p.A2 ? (const T*)p.A2 + (long long)g * p.strideA2_g : nullptr;
  const T* W = (const T*)p.W + (long long)g * p.strideW_g;
  const int srow = lane >> 3, sp = lane & 7;
  // this wave's 4 A pieces (own half) and 4 W pieces of a stage; piece = 8 rows x 128 B
  // (32-bit element offsets: the launcher routes operands of 2^31 elements or more to the other kernels)
  unsigned woff[4];
  unsigned aoff[4];
  int ay[4], ax[4];
  bool avalid[4];
  auto set_w = [&](int n0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rw = (wave * 4 + j) * 8 + srow;
      int n = n0 + rw;
      n = n < p.N ? n : p.N - 1;
      woff[j] = (unsigned)n * (unsigned)p.ldw + swz8(rw, sp) * 8;
    }
  };
  auto set_a = [&](int m0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ra = wm * 128 + (wn * 4 + j) * 8 + srow;
      int m = m0 + ra;
      avalid[j] = m < p.M;
      m = avalid[j] ? m : p.M - 1;
      if (AMODE == A_DENSE) {
        aoff[j] = (unsigned)m * (unsigned)p.lda + swz8(ra, sp) * 8;
      } else {
        const int pix = m % (p.H * p.Wd);
        ay[j] = pix / p.Wd;
        ax[j] = pix % p.Wd;
        aoff[j] = m;
      }
    }
  };
  // s = stage index relative to the current tile; s >= nk addresses the next tile (registers already switched)
  auto dma_w = [&](int s, int pb, bool more) {   // 4 instructions
    if ((ABL & 1) && abl_loop) return;
    if (s >= nk && !more) return;
    char* sW = smem + ((pb + s) & 1) * STAGE_BYTES + A_BYTES;
    const int k0 = (s < nk ? s : s - nk) * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_cp<(ABL & 16) ? 2 : (ABL & 32) ? 1 : (ABL & 64) ? 17 : 0>(W + (woff[j] + (unsigned)k0), sW + (wave * 4 + j) * 1024);
  };
  auto dma_a = [&](int s, int pb, bool more) {   // 4 instructions
    if ((ABL & 1) && abl_loop) return;
    if (s >= nk && !more) return;
    char* sA = smem + ((pb + s) & 1) * STAGE_BYTES;
    const int k0 = (s < nk ? s : s - nk) * BK;
    if (AMODE == A_DENSE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) glds16_cp<(ABL & 8) ? 2 : (ABL & 32) ? 1 : (ABL & 64) ? 17 : 0>(A + (aoff[j] + (unsigned)k0), sA + (wm * 16 + wn * 4 + j) * 1024);
    } else {
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
        const int ra = wm * 128 + (wn * 4 + j) * 8 + srow;
        const int yy = ay[j] + dy, xx = ax[j] + dx;
        const bool ok = avalid[j] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
        const T* sp_ = ok ? src + ((long long)aoff[j] + dy * p.Wd + dx) * cs + c0 + swz8(ra, sp) * 8 : (const T*)p.zero_page + sp * 8;
        glds16(sp_, sA + (wm * 16 + wn * 4 + j) * 1024);
      }
    }
  };

  const int fr = lane & 15, fg = lane >> 4;
  auto load_frags = [&](V8* wf, V8* xf, int par, int h) {
    if ((ABL & 2) && abl_loaded) return;
    abl_loaded = true;
    const char* sA = smem + par * STAGE_BYTES;
    const char* sW = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rw = wn * 64 + i * 16 + fr;
      wf[i] = *(const V8*)(sW + rw * 128 + swz8(rw, h * 4 + fg) * 16);
    }
#pragma unroll
    for (int i = 0; i < WMF; ++i) {
      const int rx = wm * 128 + i * 16 + fr;
      xf[i] = *(const V8*)(sA + rx * 128 + swz8(rx, h * 4 + fg) * 16);
    }
  };
  auto mfma32 = [&](f32x4 (*acc)[4], const V8* wf, const V8* xf) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < WMF; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Lp<T>::mma16(wf[ni], xf[mi], acc[mi][ni]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
  };

  // next-tile addresses must be computed in the K-loop iteration that switches to them: hoisted in front of the loop
  // (they are loop-invariant) they cost 8 more live VGPRs, which spills accumulators inside the loop
  auto opaque = [](int v) {
    asm volatile("" : "+s"(v));
    return v;
  };
  int seq = blockIdx.x;
  int tm, tn;
  pp_tile_coords(xcd_remap(seq, ntiles), ntm, ntn, band_m, tm, tn);
  int m0 = tm * BM, n0 = tn * BN;
  set_a(m0);
  set_w(n0);
  // prologue: stage 0 (own A half + W share) and the own A half of stage 1
  dma_a(0, 0, false);
  dma_w(0, 0, false);
  dma_a(1, 0, false);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  abl_loop = true;
  if (DBG) t_landed = __builtin_amdgcn_s_memrealtime();
  int pb = 0;   // LDS stage parity of the current tile's stage 0
  if (wm == 1) bar();   // slot 0: this wave-row idles
  bool skipbar = false;
  for (;;) {
    f32x4 acc[WMF][4];
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    V8 wf[4], xf[WMF];
    const int nseq = seq + gridDim.x;
    const bool more = PERSIST && nseq < ntiles;
    int nm0 = 0, nn0 = 0;
    if (more) {
      pp_tile_coords(xcd_remap(nseq, ntiles), ntm, ntn, band_m, tm, tn);
      nm0 = tm * BM;
      nn0 = tn * BN;
    }
    if (wm == 0) {
      for (int kt = 0; kt < nk; ++kt) {
        if (more && kt == nk - 2) set_a(opaque(nm0));   // A DMAs from here on belong to the next tile
        if (more && kt == nk - 1) set_w(opaque(nn0));
        if (!(kt == 0 && skipbar)) bar();   // slot 4kt
        load_frags(wf, xf, (pb + kt) & 1, 0);
        dma_w(kt + 1, pb, more);
        bar();                      // slot 4kt+1
        mfma32(acc, wf, xf);
        bar();                      // slot 4kt+2
        load_frags(wf, xf, (pb + kt) & 1, 1);
        bar();                      // slot 4kt+3
        mfma32(acc, wf, xf);
        dma_a(kt + 2, pb, more);
        if (kt + 2 < nk || more)
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // stage kt+1 landed; A0(kt+2) may still fly
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (more || !PERSIST) bar();  // opens slot 0 of the next tile (row 1 runs its last C slot there); without
                                    // persistence: row 1's last fragment reads are done, the LDS ring is free
      skipbar = true;
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        if (more && kt == nk - 2) set_a(opaque(nm0));
        if (more && kt == nk - 1) set_w(opaque(nn0));
        bar();                      // slot 4kt+1
        load_frags(wf, xf, (pb + kt) & 1, 0);
        dma_w(kt + 1, pb, more);
        bar();                      // slot 4kt+2
        mfma32(acc, wf, xf);
        bar();                      // slot 4kt+3
        load_frags(wf, xf, (pb + kt) & 1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own share of stage kt+1 (A1 and W) landed
        if (kt + 1 < nk || more || !PERSIST) bar();   // slot 4kt+4
        mfma32(acc, wf, xf);
        dma_a(kt + 2, pb, more);
      }
    }
    if (DBG) t_loop = __builtin_amdgcn_s_memrealtime();
    if (PERSIST) {   // the ring already holds the next tile's first stages: stage through the 32 KiB beside it
      // opaque copy of the lane id: keeps the epilogue's per-lane address arithmetic from being hoisted out of the tile
      // loop, where it would stay live across the K loop and spill (scratch traffic drains the counted DMA queue)
      int elane = lane;
      asm volatile("" : "+v"(elane));
      epilogue_lds4k<T>(p, acc, smem + 2 * STAGE_BYTES + wave * 4096, m0, n0, wm, wn, elane, g);
    } else
      epilogue_lds<T>(p, acc, smem + wave * 16384, m0, n0, wm, wn, lane, g);
    if (DBG && wn == 0 && lane == 0) {
      unsigned long long* d = dbg + ((long long)seq * 2 + wm) * 6;
      d[0] = t_entry;
      d[1] = t_landed;
      d[2] = t_loop;
      d[3] = __builtin_amdgcn_s_memrealtime();
      d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
      d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
      t_entry = t_landed = d[3];
    }
    if (!more) break;
    seq = nseq;
    m0 = nm0;
    n0 = nn0;
    pb = (pb + nk) & 1;
  }
}

int g_num_cus = 0;
int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cus = n;
  }
  return g_num_cus;
}

unsigned long long* g_dbg = nullptr;   // mk_gemm_debug_timeline
int g_band_m = 8;                      // m-tiles per band of the tile order (set_tile 400+b)
int g_stagger_us = 0;                  // start-time spread of the first workgroup round for RMW epilogues (set_tile 100+us)

template <typename T, int AMODE, bool PERSIST, bool DBG = false, int ABL = 0>
int launch_pp64(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  constexpr int LDS = 2 * 512 * 128 + (PERSIST ? 8 * 4096 : 0);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_pp64_kernel<T, AMODE, PERSIST, DBG, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  const int cap = (num_cus() + groups - 1) / groups;   // one resident workgroup per CU in total
  const int gx = (!PERSIST || ntm * ntn < cap) ? ntm * ntn : cap;
  const int stagger = (p.epi == MK_EPI_LS_RESIDUAL && ntm * ntn >= 4 * cap) ? g_stagger_us * 100 : 0;
  hipLaunchKernelGGL((gemm_pp64_kernel<T, AMODE, PERSIST, DBG, ABL>), dim3(gx, groups, 1), dim3(512), LDS, st, p, band_m, stagger, g_dbg);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

template <typename T, int AMODE, int WMF, int NWM, int NWN>
int launch_cfg(const GemmParams& p, int groups, hipStream_t st) {
  constexpr int BM = NWM * WMF * 16, BN = NWN * 64;
  constexpr int LDS = 2 * (BM + BN) * 128;
  constexpr bool PERSIST = WMF == 8;   // the 128-KiB-LDS tile owns its CU: walk the tiles persistently
  static bool attr_done = false;  // benign race: the attribute call is idempotent
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<T, AMODE, WMF, NWM, NWN, PERSIST>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  int gx = ntm * ntn;
  if (PERSIST) {
    const int cap = (num_cus() + groups - 1) / groups;   // one resident workgroup per CU in total
    if (gx > cap) gx = cap;
  }
  hipLaunchKernelGGL((gemm_kernel<T, AMODE, WMF, NWM, NWN, PERSIST>), dim3(gx, groups, 1), dim3(NWM * NWN * 64), LDS, st, p);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

template <typename T, int AMODE, int ABL = 0>
int launch_pp(const GemmParams& p, int groups, hipStream_t st) {
  constexpr int LDS = PSTAGES * 512 * 64;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_pp_kernel<T, AMODE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  hipLaunchKernelGGL((gemm_pp_kernel<T, AMODE, ABL>), dim3(ntm * ntn, groups, 1), dim3(512), LDS, st, p);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int g_force_tile = 0;  // 0 auto (128x128 or full-line ping-pong), 1 force 128x128, 2 force 256x256 K-stream, 3 force K=32 ping-pong ring,
                       // 4 auto with the 256x256 K-stream kernel for large problems, 5/7/8/9 full-line ping-pong variants, 10+ ablations

template <int AMODE>
int launch(const GemmParams& p, int groups, int dtype, hipStream_t st) {
  // 256x256 tiles need enough of them to fill 256 CUs (1 workgroup per CU); otherwise 128x128 (2 per CU)
  const long long big_tiles = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256) * groups;
  bool big = p.N >= 256 && big_tiles >= 224;  // measured: +12..18 % over 128x128 at M >= 31k (profiles/r01_gemm_pmc.md)
  if (g_force_tile == 1) big = false;
  if (g_force_tile == 2) big = true;
#ifdef MK_PP64_ABLATIONS
  if (g_force_tile >= 30 && g_force_tile <= 39 && AMODE == A_DENSE && dtype == MK_BF16) {
    switch (g_force_tile - 30) {
      case 1: return launch_pp64<__bf16, A_DENSE, false, false, 1>(p, groups, st, 8);
      case 3: return launch_pp64<__bf16, A_DENSE, false, false, 3>(p, groups, st, 8);
      case 4: return launch_pp64<__bf16, A_DENSE, false, false, 4>(p, groups, st, 8);
      case 5: return launch_pp64<__bf16, A_DENSE, false, false, 8>(p, groups, st, 8);    // A nt
      case 6: return launch_pp64<__bf16, A_DENSE, false, false, 24>(p, groups, st, 8);   // A and W nt
      case 7: return launch_pp64<__bf16, A_DENSE, false, false, 32>(p, groups, st, 8);   // sc0
      case 8: return launch_pp64<__bf16, A_DENSE, false, false, 64>(p, groups, st, 8);   // sc0 sc1
      default: return launch_pp64<__bf16, A_DENSE, false, false, 16>(p, groups, st, 8);  // W nt
    }
  }
#endif
  const bool k_ok = p.K >= 2 * BK && (long long)p.M * p.lda < (1ll << 31) && (long long)p.N * p.ldw < (1ll << 31) &&
                    (AMODE == A_DENSE || (long long)p.M * (p.C1 > p.C2 ? p.C1 : p.C2) < (1ll << 31));
  const bool forced64 = g_force_tile == 5 || (g_force_tile >= 7 && g_force_tile <= 9);
  if (k_ok && (forced64 || ((g_force_tile == 0 || g_force_tile == 6) && big))) {
    // auto (0/6) and 7: one tile per workgroup, bands of 8 m-tiles (measured best: 6-16 % over the K=32 ring and the
    // K-stream kernel, 8-10 % over its own persistent variant); 5: persistent + bands; 8: persistent, row-major; 9: neither
    const bool persist = g_force_tile == 5 || g_force_tile == 8;
    const int band_m = (g_force_tile == 8 || g_force_tile == 9) ? 1 : g_band_m;
    if (g_dbg && AMODE == A_DENSE && dtype == MK_BF16 && groups == 1)
      return persist ? launch_pp64<__bf16, A_DENSE, true, true>(p, groups, st, band_m)
                     : launch_pp64<__bf16, A_DENSE, false, true>(p, groups, st, band_m);
    if (persist)
      return dtype == MK_BF16 ? launch_pp64<__bf16, AMODE, true>(p, groups, st, band_m)
                              : launch_pp64<_Float16, AMODE, true>(p, groups, st, band_m);
    return dtype == MK_BF16 ? launch_pp64<__bf16, AMODE, false>(p, groups, st, band_m)
                            : launch_pp64<_Float16, AMODE, false>(p, groups, st, band_m);
  }
  if (g_force_tile == 3)
    return dtype == MK_BF16 ? launch_pp<__bf16, AMODE>(p, groups, st) : launch_pp<_Float16, AMODE>(p, groups, st);
#ifdef MK_GEMM_ABLATIONS   // timing ablations of the K=32 ring kernel for tools/ablate_gemm.py (results are wrong); not in the default build
  if (g_force_tile >= 10 && g_force_tile < 22 && AMODE == A_DENSE && dtype == MK_BF16) {  // timing ablations (wrong results)
    switch (g_force_tile - 10) {
      case 1: return launch_pp<__bf16, A_DENSE, 1>(p, groups, st);
      case 2: return launch_pp<__bf16, A_DENSE, 2>(p, groups, st);
      case 3: return launch_pp<__bf16, A_DENSE, 3>(p, groups, st);
      case 4: return launch_pp<__bf16, A_DENSE, 4>(p, groups, st);
      case 5: return launch_pp<__bf16, A_DENSE, 5>(p, groups, st);
      case 6: return launch_pp<__bf16, A_DENSE, 6>(p, groups, st);
      case 7: return launch_pp<__bf16, A_DENSE, 7>(p, groups, st);
      case 8: return launch_pp<__bf16, A_DENSE, 8>(p, groups, st);
      case 9: return launch_pp<__bf16, A_DENSE, 16>(p, groups, st);
      case 10: return launch_pp<__bf16, A_DENSE, 24>(p, groups, st);
      case 11: return launch_pp<__bf16, A_DENSE, 32>(p, groups, st);
      default: break;
    }
  }
#endif
  if (dtype == MK_BF16)
    return big ? launch_cfg<__bf16, AMODE, 8, 2, 4>(p, groups, st) : launch_cfg<__bf16, AMODE, 4, 2, 2>(p, groups, st);
  return big ? launch_cfg<_Float16, AMODE, 8, 2, 4>(p, groups, st) : launch_cfg<_Float16, AMODE, 4, 2, 2>(p, groups, st);
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

int mk_gemm_debug_timeline(void* buf) {
  g_dbg = (unsigned long long*)buf;
  return MK_OK;
}

int mk_gemm_set_tile(int mode) {
#ifdef MK_PP64_ABLATIONS
  if (mode >= 30 && mode <= 39) { g_force_tile = mode; return MK_OK; }
#endif
  if (mode >= 400 && mode < 528) {   // dev: band height of the tile order
    g_band_m = mode - 400 > 0 ? mode - 400 : 1;
    return MK_OK;
  }
  if (mode >= 100 && mode < 400) {   // dev: stagger window in microseconds
    g_stagger_us = mode - 100;
    return MK_OK;
  }
  #ifdef MK_GEMM_ABLATIONS
  MK_CHECK_ARG((mode >= 0 && mode <= 9) || (mode >= 10 && mode < 22), "mk_gemm_set_tile: unknown mode %d", mode);
#else
  MK_CHECK_ARG(mode >= 0 && mode <= 9, "mk_gemm_set_tile: unknown mode %d (0 auto, 1..5 / 7..9 schedules; ablations need -DMK_GEMM_ABLATIONS)", mode);
#endif
  g_force_tile = mode;
  return MK_OK;
}

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
  MK_CHECK_ARG(bias && gamma && x && lda % 8 == 0 && lda >= K && ldx % 4 == 0 && ldx >= N, "mk_gemm_ls_residual: bad args");
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
  MK_CHECK_ARG(bias && q && k && vt && ntok_pad % 64 == 0 && ntok_pad >= ntok && lda % 8 == 0 && lda >= D, "mk_gemm_qkv: bad args");
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

int mk_gemm_patch_embed(const void* A, int lda, const void* W, int ldw, const float* bias, const float* pos, float* x,
                        int nimg, int npatch, int D, int K, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = A; p.W = W; p.M = nimg * npatch; p.N = D; p.K = K; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_PATCH; p.bias = bias; p.pos = pos; p.npatch = npatch; p.out_f32 = x; p.ldc = D;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(bias && pos && x && lda % 8 == 0 && lda >= K, "mk_gemm_patch_embed: bad args");
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
