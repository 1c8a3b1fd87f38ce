// mickey_amd -- 16-bit-operand MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] (+)= A[M,K] . W[N,K]^T        A, W: bf16 or fp16, K contiguous; fp32 accumulate
//
// One kernel body serves every dense contraction of the MicKey hot path:
//   * ViT linears (reference DINO_modules/layers/attention.py:44,51,60; mlp.py:30-39) with the
//     bias / GELU / LayerScale+residual / QKV-split epilogues fused,
//   * the 14x14 patch embedding (reference layers/patch_embed.py:66,76) after an im2col pass,
//   * the heads' 3x3 convolutions as an implicit GEMM (reference utils/extractor_utils.py:18-35):
//     the A tile comes straight from the BORDERED NHWC activation (mk_common.hpp; one 3x3 tap per 64-wide K
//     tile = one row shift, the padding zeros are part of the buffer), BatchNorm is folded into W/bias on the
//     host and the 1x1 shortcut conv rides along as extra K columns.
//
// Structure (MI355X): two instantiations of one body -- 128x128x64 tile / 4 waves (2x2, 64x64 per wave) and
// 256x256x64 tile / 8 waves (2x4, 128x64 per wave) -- built from v_mfma_f32_16x16x32 fragments.  Operands go HBM -> LDS directly (global_load_lds, 16 B/lane),
// double-buffered, one barrier per K tile.  The LDS image is lane-linear, so the XOR swizzle that
// makes the ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address and
// to the read address (never to the LDS destination).  MFMA operands are swapped (A-operand = W rows,
// B-operand = activation rows) so that each lane ends up with 4 CONSECUTIVE output features of one
// row: epilogue loads/stores are 8-16 B per lane.  Block ids are remapped so that the blocks of one
// XCD (private L2) walk a contiguous range of tiles.
#include "mk_gemm_common.hpp"

namespace mk {
namespace gemm {
namespace {

// 128x128 tile, 4 waves (2 x 2, 64x64 per wave), 64 KiB LDS -> 2 workgroups per CU: small / skinny problems (a single
// image pair, the 128-channel linears of the heads) and the fallback for operands of 2^31 elements or more.
// Two LDS stages, the LDS-DMA of the next K tile is issued before the MFMAs of the current one, one barrier per K tile.
// WMF = 16-row fragments per wave: 4 -> 128x128 tiles; 2 -> 64x128 tiles (48 KiB of LDS, 3 workgroups per CU) for problems
// whose 128x128 tiling would leave CUs without work (proj / fc2 of a single image pair: 248 -> 488 workgroups).
// SP: split operands staged once (mk_gemm_common.hpp, GemmParams): a K tile is 32 contraction columns, its LDS rows hold
// [32 hi | 32 lo] -- sub-step 0 reads the HI fragments of W and A, sub-step 1 the LO fragments, and the three products
// hi.hi, lo.hi, hi.lo come from those registers (the plain kernel's two sub-steps are the two K halves of a 64-column tile).
// (measured and removed in round 6: a 128x128 form with FOUR LDS stages -- 129 KiB, one workgroup per CU -- for launches of <= one
// workgroup per CU, i.e. proj / fc2 of a single image pair: 2/3 of the 64-row form's L2 -> LDS bytes, three K tiles in flight, and NO
// faster: fc2 44.2 us against 41.7 (64x128, three stages), 45.5 (128x128, two stages) and hipBLASLt's 42.9; profiles/r06k_gemm_b1.txt)
template <typename T, int AMODE, int WMF, bool SP = false>
__global__ __launch_bounds__(256, 1) void gemm_kernel(GemmParams p) {
  using V8 = typename Lp<T>::V8;
  constexpr int NWM = 2, NWN = 2;
  constexpr int NW = NWM * NWN, BM = NWM * WMF * 16, BN = NWN * 64;
  constexpr int AJ = BM / 8 / NW, WJ = BN / 8 / NW;          // 1-KiB LDS-DMA pieces per wave per K tile
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
  // LDS stages.  The 64-row form is what under-filled launches get (one image pair): every workgroup is resident at once
  // and a launch lasts as long as ONE workgroup's K loop, which with two stages is one L2 -> LDS round trip (~1 us) per
  // K tile against ~0.15 us of MFMA work (fc2 of one pair: 64 tiles = 65 us).  Three stages keep two DMA stages in
  // flight behind the one being consumed (72 KiB: two workgroups per CU, i.e. four stages in flight per CU).
  constexpr int NS = WMF == 2 ? 3 : 2;
  constexpr int PIECES = AJ + WJ;   // LDS-DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NS stages][A tile | W tile], rows of 128 B

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int nk = p.K / BK;
  const int fr = lane & 15, fg = lane >> 4;

  // xcd_remap keeps the tiles that one XCD works on at any time adjacent (same A panel / neighbouring W panels in its L2)
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BN;
  Stager<T, AMODE, NW, AJ, WJ, SP> st;
  st.init(p, g, m0, n0, wave, lane);
  st.issue(p, smem, smem + A_BYTES, 0);
  if (NS == 3 && nk > 1) st.issue(p, smem + STAGE_BYTES, smem + STAGE_BYTES + A_BYTES, 1);
  // folded LayerNorm (consumer): row parameters of the tile into LDS while the first stage is in flight
  float2* lnp = (float2*)(smem + NS * STAGE_BYTES);
  if (AMODE == A_DENSE && p.ln_stats && tid < 2 * BM) ln_params_to_lds<BM, 2 * BM>(p, m0, tid, lnp, p.ln_shift_out != nullptr && n0 == 0);
  f32x4 acc[WMF][4];
#pragma unroll
  for (int i = 0; i < WMF; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cur = 0, nxt = NS - 1;   // stage buffers: being consumed / receiving stage kt + NS - 1
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed; with three stages the pieces of stage kt + 1 (if it exists) may still be in flight
    if (NS == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... for every wave, and everybody is done reading the buffer that is refilled next.  A bare s_barrier: __syncthreads()
    // is a workgroup fence and makes hipcc drain ALL LDS-DMA traffic (s_waitcnt vmcnt(0)) in front of it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* nA = smem + nxt * STAGE_BYTES;
    if (kt + NS - 1 < nk) st.issue(p, nA, nA + A_BYTES, kt + NS - 1);
    const char* sA = smem + cur * STAGE_BYTES;
    const char* sW = sA + A_BYTES;
    cur = cur + 1 == NS ? 0 : cur + 1;
    nxt = nxt + 1 == NS ? 0 : nxt + 1;
    if constexpr (SP) {
      V8 wf[2][4], xf[2][WMF];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rw = wn * 64 + i * 16 + fr;
          wf[ks][i] = *(const V8*)(sW + rw * 128 + swz8(rw, ks * 4 + fg) * 16);
        }
#pragma unroll
        for (int i = 0; i < WMF; ++i) {
          const int rx = wm * (WMF * 16) + i * 16 + fr;
          xf[ks][i] = *(const V8*)(sA + rx * 128 + swz8(rx, ks * 4 + fg) * 16);
        }
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {   // hi.hi, W_lo . A_hi, W_hi . A_lo
        if (pr == 2 && p.npass == 2) break;   // the activations' lo plane is identically zero (mk_conv3x3_split, in1_lo == NULL)
#pragma unroll
        for (int mi = 0; mi < WMF; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Lp<T>::mma16(wf[pr == 1][ni], xf[pr == 2][mi], acc[mi][ni]);
      }
    } else {
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
  }
  epilogue<T, WMF>(p, acc, m0, n0, wm, wn, lane, g, lnp);
}

template <typename T, int AMODE, int WMF, bool SP = false>
int launch_small(const GemmParams& p, int groups, hipStream_t st) {
  constexpr int BM = 32 * WMF;
  constexpr int LDS = (WMF == 2 ? 3 : 2) * (BM + 128) * 128 + BM * 8;   // the stages + the folded LayerNorm's row parameters
  static bool attr_done = false;  // benign race: the attribute call is idempotent
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<T, AMODE, WMF, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + 127) / 128;
  hipLaunchKernelGGL((gemm_kernel<T, AMODE, WMF, SP>), dim3(ntm * ntn, groups, 1), dim3(256), LDS, st, p);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int g_num_cus = 0;
int g_band_m = 0;      // tile order of the 256x256 kernels: 0 automatic, b > 0 bands of b m-tiles, -g groups of g n-tiles (dev: mk_gemm_set_tile 400 + b / 464 + g)
int g_half_rows = 1;   // automatic choice may use the 64x128 tiling for under-filled launches (dev: 500 off / 501 on)
int g_schedule = 0;    // mk_gemm_set_tile: 0 automatic, 1 force 128x128, 2 force 64x128, 7 force the 8-wave ping-pong

template <int AMODE>
int launch(const GemmParams& p, int groups, int dtype, hipStream_t st) {
  // 256x256 tiles need enough of them to fill 256 CUs (1 workgroup per CU); otherwise 128x128 (2 per CU)
  const long long big_tiles = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256) * groups;
  // measured: +12..18 % over 128x128 at M >= 31k (profiles/r01_gemm_pmc.md); at 192 tiles (one pair, qkv: 16 x 12) the
  // ping-pong kernel on 75 % of the CUs still beats 744 tiles of 128x128 (30.6 vs 35.6 us); at 64 tiles (proj / fc2 of one
  // pair) it loses 2x
  bool big = p.N >= 256 && big_tiles >= 192;
  // the 256x256 kernels address operands with 32-bit element offsets and run a software pipeline of >= 2 K stages
  // (byte offsets of 16-bit elements in 32 bits: < 2^31 elements)
  // (split operands: the byte offset of a chunk's plane rides in the same 32 bits)
  const long long plmax = p.npass > 1 ? (long long)max(max(p.pl1[0], p.pl1[1]), max(p.pl2[0], p.pl2[1])) / 2 : 0;
  const bool k_ok = p.K >= 2 * BK && (long long)p.M * p.lda + plmax < (1ll << 31) && (long long)p.N * p.ldw < (1ll << 31) &&
                    (AMODE == A_DENSE || bordered_rows(p.M / (p.H * p.Wd), p.H, p.Wd) * (p.C1 > p.C2 ? p.C1 : p.C2) + plmax < (1ll << 31));
  const bool ln_fold = p.ln_stats || p.xh;
  if (dtype == MK_F32) {   // exact parity mode: one plain schedule, LayerNorm as its own kernel
    MK_CHECK_ARG(!ln_fold, "gemm: the folded-LayerNorm epilogues exist for 16-bit operands only");
    return launch_f32(p, groups, AMODE, st);
  }
  int sched = g_schedule;
  if (sched == 0) sched = (big && k_ok) ? 7 : 1;
  if (!k_ok) sched = 1;
  if (sched == 7) return launch_pp64(p, groups, dtype, AMODE, st, g_band_m);
  // 128x128 tiles fill a 256-CU part (2 workgroups per CU) from 512 tiles on; below that 64-row tiles double the count
  const long long small_tiles = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * groups;
  const bool half_rows = g_schedule == 2 || (g_schedule != 1 && g_half_rows && small_tiles < 2 * num_cus() && p.M > 64);
  if (p.npass > 1)   // split operands (fp16 planes)
    return half_rows ? launch_small<_Float16, AMODE, 2, true>(p, groups, st) : launch_small<_Float16, AMODE, 4, true>(p, groups, st);
  if (half_rows) return dtype == MK_BF16 ? launch_small<__bf16, AMODE, 2>(p, groups, st) : launch_small<_Float16, AMODE, 2>(p, groups, st);
  return dtype == MK_BF16 ? launch_small<__bf16, AMODE, 4>(p, groups, st) : launch_small<_Float16, AMODE, 4>(p, groups, st);
}

int check_common(const GemmParams& p, int dtype) {
  MK_CHECK_ARG(dtype == MK_BF16 || dtype == MK_F16 || dtype == MK_F32, "gemm: dtype must be MK_BF16, MK_F16 or MK_F32");
  MK_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  const int kt = dtype == MK_F32 ? KT<float> : BK;
  MK_CHECK_ARG(p.K % kt == 0, "gemm: K=%d must be a multiple of %d", p.K, kt);
  MK_CHECK_ARG(p.N % 4 == 0, "gemm: N=%d must be a multiple of 4", p.N);
  MK_CHECK_ARG(p.ldw % 8 == 0 && p.ldw >= p.K, "gemm: ldw=%d must be >= K and a multiple of 8", p.ldw);
  MK_CHECK_ARG(p.A && p.W, "gemm: null operand");
  return MK_OK;
}

}  // namespace

int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cus = n;
  }
  return g_num_cus;
}

}  // namespace gemm
}  // namespace mk

using namespace mk;
using namespace mk::gemm;

extern "C" {

int mk_gemm_set_tile(int mode) {
  if (mode >= 400 && mode < 464) {   // dev: band height of the 256x256 tile order
    g_band_m = mode - 400;   // 400: automatic
    return MK_OK;
  }
  if (mode >= 464 && mode < 496) {   // dev: n-group tile order (groups of mode - 464 n-tiles walked m-major); 464: automatic
    g_band_m = -(mode - 464);
    return MK_OK;
  }
  if (mode == 500 || mode == 501) {   // dev: automatic use of the 64x128 tiling off / on
    g_half_rows = mode - 500;
    return MK_OK;
  }
  if (mode >= 600 && mode <= 602) {   // dev: persistent tile loop of the 256x256 kernel wherever it applies / off / producers only (default)
    g_pp64_persist = mode - 600;
    return MK_OK;
  }
  MK_CHECK_ARG(mode == 0 || mode == 1 || mode == 2 || mode == 7,
               "mk_gemm_set_tile: unknown mode %d (0 automatic, 1 128x128, 2 64x128, 7 8-wave ping-pong)", mode);
  g_schedule = mode;
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


// ---- LayerNorm folded into the GEMMs around it (reference block.py:84-88,105-106: x + ls(f(norm(x)))) ----
// producer: the residual / patch-embed epilogue also emits the new rows in 16 bit (raw) and their partial statistics;
// consumer: A = those raw rows, W = W.diag(ln_weight), and the epilogue applies rstd / mean per row.
static int ln_consumer_args(GemmParams& p, const float* colsum, const float* stats, float eps, float* shift_out, const char* who) {
  MK_CHECK_ARG(colsum && stats && p.bias, "%s: colsum, stats and bias are required", who);
  MK_CHECK_ARG(p.lda == p.K, "%s: the normalised width is K: lda must equal K", who);
  p.ln_colsum = colsum; p.ln_stats = stats; p.ln_nslot = p.K / 64; p.ln_eps = eps; p.ln_shift_out = shift_out;
  return MK_OK;
}

int mk_gemm_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const float* colsum, const float* stats,
               float eps, float* shift_out, void* out, int ldc, int M, int N, int K, int act, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_STORE; p.act = act; p.bias = bias; p.ldc = ldc; p.out_lp = out;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(lda % 8 == 0 && ldc % 4 == 0 && ldc >= N && out && (act == MK_ACT_NONE || act == MK_ACT_GELU), "mk_gemm_ln: bad args");
  if (int e = ln_consumer_args(p, colsum, stats, eps, shift_out, "mk_gemm_ln")) return e;
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

int mk_gemm_qkv_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const float* colsum, const float* stats,
                   float eps, float* shift_out, void* q, void* k, void* vt, int nimg, int ntok, int ntok_pad, int heads,
                   float qscale, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  const int D = heads * 64;
  p.A = A; p.W = W; p.M = nimg * ntok; p.N = 3 * D; p.K = D; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_QKV; p.bias = bias; p.q = q; p.k = k; p.vt = vt;
  p.ntok = ntok; p.ntok_pad = ntok_pad; p.heads = heads; p.qscale = qscale;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(q && k && vt && ntok_pad % 64 == 0 && ntok_pad >= ntok && lda % 8 == 0, "mk_gemm_qkv_ln: bad args");
  if (int e = ln_consumer_args(p, colsum, stats, eps, shift_out, "mk_gemm_qkv_ln")) return e;
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

int mk_gemm_ls_residual_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const float* gamma, void* xh,
                           void* xl, int ldxs, float* stats, const float* shift_in, float* x_f32_out, int ldx, int M, int N,
                           int K, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_LS_RESIDUAL; p.bias = bias; p.gamma = gamma; p.out_f32 = x_f32_out; p.ldc = ldx;
  p.xh = xh; p.xl = xl; p.ldxs = ldxs; p.stats_out = stats; p.nslot_out = N / 64; p.ln_shift_in = shift_in;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(bias && gamma && lda % 8 == 0 && lda >= K, "mk_gemm_ls_residual_ln: bad args");
  MK_CHECK_ARG(xh && xl && N % 64 == 0 && ldxs % 8 == 0 && ldxs >= N, "mk_gemm_ls_residual_ln: xh / xl / N %% 64");
  MK_CHECK_ARG(x_f32_out ? (ldx % 4 == 0 && ldx >= N) : stats != nullptr, "mk_gemm_ls_residual_ln: stats, or x_f32_out with ldx");
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

int mk_gemm_patch_embed_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const float* pos, void* xh,
                           void* xl, float* stats, int nimg, int npatch, int D, int K, int dtype, mk_stream_t stream) {
  GemmParams p = {};
  p.A = A; p.W = W; p.M = nimg * npatch; p.N = D; p.K = K; p.lda = lda; p.ldw = ldw;
  p.epi = MK_EPI_PATCH; p.bias = bias; p.pos = pos; p.npatch = npatch; p.ldc = D;
  p.xh = xh; p.xl = xl; p.ldxs = D; p.stats_out = stats; p.nslot_out = D / 64;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(bias && pos && lda % 8 == 0 && lda >= K, "mk_gemm_patch_embed_ln: bad args");
  MK_CHECK_ARG(xh && xl && stats && D % 64 == 0, "mk_gemm_patch_embed_ln: xh / xl / stats / D %% 64");
  return launch<A_DENSE>(p, 1, dtype, (hipStream_t)stream);
}

long long mk_bordered_rows(int nimg, int H, int Wd) { return bordered_rows(nimg, H, Wd); }

int mk_conv3x3(const void* in1, long long stride_in1, int C1, const void* in2, long long stride_in2, int C2, const void* W,
               int ldw, long long strideW, const float* bias, long long strideBias, const void* resid, long long strideResid,
               void* out, int Cout, long long strideOut, int groups, int nimg, int H, int Wd, int act, int out_kind, int dtype,
               mk_stream_t stream) {
  GemmParams p = {};
  p.A = in1; p.A2 = in2; p.W = W;
  p.M = nimg * H * Wd; p.N = Cout; p.K = 9 * C1 + (in2 ? C2 : 0);
  p.ldw = ldw; p.strideA_g = stride_in1; p.strideA2_g = stride_in2; p.strideW_g = strideW;
  p.strideBias_g = strideBias; p.strideOut_g = strideOut;
  p.H = H; p.Wd = Wd; p.C1 = C1; p.C2 = in2 ? C2 : 0;
  p.epi = MK_EPI_STORE; p.act = act; p.bias = bias; p.ldc = Cout; p.resid_lp = resid; p.strideResid_g = strideResid;
  if (out_kind == MK_CONV_OUT_F32) p.out_f32 = (float*)out; else p.out_lp = out;
  p.bord_out = out_kind == MK_CONV_OUT_BORDERED;
  if (int e = check_common(p, dtype)) return e;
  MK_CHECK_ARG(C1 % (dtype == MK_F32 ? 32 : BK) == 0 && (!in2 || C2 % (dtype == MK_F32 ? 32 : BK) == 0),
               "mk_conv3x3: channel counts must be multiples of the K tile (%d)", dtype == MK_F32 ? 32 : BK);
  MK_CHECK_ARG(out && groups > 0 && nimg > 0 && H > 0 && Wd > 0 && out_kind >= 0 && out_kind <= 2, "mk_conv3x3: bad args");
  return launch<A_CONV3>(p, groups, dtype, (hipStream_t)stream);
}

// fp32 -> two fp16 planes, x * scale = hi + lo (22 mantissa bits); |x * scale| is saturated at fp16's largest finite value
namespace {
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, long long rows, int cols4, long long ld_src4,
                                                           float scale, uint2* __restrict__ hi, uint2* __restrict__ lo, long long ld_dst4, int* sat_flag) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  if (t >= rows * cols4) return;
  const long long r = t / cols4;
  const int c = (int)(t - r * cols4);
  const long long i = r * ld_dst4 + c;
  const f32x4 x = __builtin_nontemporal_load((const f32x4*)src + r * ld_src4 + c) * scale;
  f16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float v = sat16(x[e], sat_flag);
    h[e] = (_Float16)v;
    l[e] = (_Float16)(v - (float)h[e]);
  }
  hi[i] = __builtin_bit_cast(uint2, h);
  lo[i] = __builtin_bit_cast(uint2, l);
}
}  // namespace

int mk_split_planes(const float* src, long long rows, int cols, long long ld_src, float scale, void* hi, void* lo,
                    long long ld_dst, int* sat_flag, mk_stream_t stream) {
  MK_CHECK_ARG(src && hi && lo && rows > 0 && cols > 0 && cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= cols &&
                   ld_dst >= cols, "mk_split_planes: cols / ld_src / ld_dst must be multiples of 4 and ld >= cols");
  MK_CHECK_ARG((((uintptr_t)src | (uintptr_t)hi * 2 | (uintptr_t)lo * 2) & 15) == 0, "mk_split_planes: src must be 16-byte, planes 8-byte aligned");
  const long long n4 = rows * (cols / 4);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, rows, cols / 4,
                     ld_src / 4, scale, (uint2*)hi, (uint2*)lo, ld_dst / 4, sat_flag);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

// the two planes of a split source: base = the lower address, pl = byte offsets of (hi, lo) from it (< 2 GiB)
static int split_source(const void* hi, const void* lo, const void*& base, unsigned (&pl)[2], const char* who) {
  const uintptr_t h = (uintptr_t)hi, l = (uintptr_t)lo, b = h < l ? h : l;
  MK_CHECK_ARG((h - b) < (1ull << 31) && (l - b) < (1ull << 31) && ((h | l) & 15) == 0,
               "%s: the (hi, lo) planes of a source must be 16-byte aligned and lie within 2 GiB of each other", who);
  base = (const void*)b;
  pl[0] = (unsigned)(h - b);
  pl[1] = (unsigned)(l - b);
  return MK_OK;
}

int mk_gemm_grouped_split(const void* A_hi, const void* A_lo, int lda, long long strideA, const void* W, int ldw, long long strideW,
                          const float* bias, long long strideBias, void* out, void* out_lo, int ldc, long long strideOut, int groups,
                          int M, int N, int K, int act, float acc_scale, float plane_scale, int* sat_flag, mk_stream_t stream) {
  GemmParams p = {};
  MK_CHECK_ARG(A_hi && A_lo, "mk_gemm_grouped_split: null plane");
  if (int e = split_source(A_hi, A_lo, p.A, p.pl1, "mk_gemm_grouped_split")) return e;
  p.W = W; p.M = M; p.N = N; p.K = 2 * K; p.lda = lda; p.ldw = ldw;
  p.npass = 3; p.acc_scale = acc_scale;
  p.strideA_g = strideA; p.strideW_g = strideW; p.strideBias_g = strideBias; p.strideOut_g = strideOut;
  p.epi = MK_EPI_STORE; p.act = act; p.bias = bias; p.ldc = ldc;
  if (out_lo) { p.out_lp = out; p.out_lo = out_lo; p.plane_scale = plane_scale; p.sat_flag = sat_flag; } else { p.out_f32 = (float*)out; }
  if (int e = check_common(p, MK_F16)) return e;
  MK_CHECK_ARG(out && groups > 0 && K % 32 == 0 && lda % 8 == 0 && lda >= K && ldc % 4 == 0 && ldc >= N,
               "mk_gemm_grouped_split: bad args (K must be a multiple of 32)");
  return launch<A_DENSE>(p, groups, MK_F16, (hipStream_t)stream);
}

int mk_conv3x3_split(const void* in1_hi, const void* in1_lo, long long stride_in1, int C1, const void* in2_hi, const void* in2_lo,
                     long long stride_in2, int C2, const void* W, int ldw, long long strideW, const float* bias,
                     long long strideBias, void* out, void* out_lo, int Cout, long long strideOut, int groups, int nimg, int H,
                     int Wd, int act, int out_bordered, float acc_scale, float plane_scale, int* sat_flag, mk_stream_t stream) {
  GemmParams p = {};
  // in1_lo == NULL: the source IS its hi plane (features of an fp16 encoder: fp16 values) -- two products instead of three
  const bool hi_only = in1_hi && !in1_lo;
  MK_CHECK_ARG(in1_hi && (hi_only ? (!in2_hi && !in2_lo) : (!in2_hi == !in2_lo)),
               "mk_conv3x3_split: every source needs both planes (in1_lo == NULL = an all-zero lo plane: then without in2)");
  if (int e = split_source(in1_hi, hi_only ? in1_hi : in1_lo, p.A, p.pl1, "mk_conv3x3_split")) return e;
  if (in2_hi) {
    if (int e = split_source(in2_hi, in2_lo, p.A2, p.pl2, "mk_conv3x3_split")) return e;
  }
  p.W = W;
  p.npass = hi_only ? 2 : 3; p.acc_scale = acc_scale;
  p.M = nimg * H * Wd; p.N = Cout; p.K = 2 * (9 * C1 + (in2_hi ? C2 : 0));
  p.ldw = ldw; p.strideA_g = stride_in1; p.strideA2_g = stride_in2; p.strideW_g = strideW;
  p.strideBias_g = strideBias; p.strideOut_g = strideOut;
  p.H = H; p.Wd = Wd; p.C1 = C1; p.C2 = in2_hi ? C2 : 0;
  p.epi = MK_EPI_STORE; p.act = act; p.bias = bias; p.ldc = Cout;
  if (out_lo) { p.out_lp = out; p.out_lo = out_lo; p.plane_scale = plane_scale; p.sat_flag = sat_flag; } else { p.out_f32 = (float*)out; }
  p.bord_out = out_bordered ? 1 : 0;
  if (int e = check_common(p, MK_F16)) return e;
  MK_CHECK_ARG(C1 % 32 == 0 && (!in2_hi || C2 % 32 == 0), "mk_conv3x3_split: channel counts must be multiples of 32");
  MK_CHECK_ARG(out && groups > 0 && nimg > 0 && H > 0 && Wd > 0, "mk_conv3x3_split: bad args");
  return launch<A_CONV3>(p, groups, MK_F16, (hipStream_t)stream);
}

}  // extern "C"
