// mickey_amd -- 8-wave full-line ping-pong GEMM schedule (256x256 tile, two waves per SIMD).
#include <type_traits>

#include "mk_gemm_common.hpp"

namespace mk {
namespace gemm {
namespace {

// Ping-pong with FULL-LINE LDS-DMA pieces: 256x256 tile, 8 waves (wave-row g = wave>>2), LDS stages of K = 64
// (128-byte rows, the swz8 swizzle of the plain kernel), computed in two K = 32 sub-steps h.  A piece is 8 rows x
// 128 B (8 full cache lines).  Only two 64-KiB stages fit, which is deep enough because the operand halves are released
// at different times:
//   * wave-row g loads AND reads only its own A half (rows 128g..128g+127); W is loaded and read by everyone;
//   * slots (barrier at every boundary):  row g does  L(kt,h) [12 fragment reads] in slot 4kt+2h+g  and
//     C(kt,h) [32 MFMAs from registers] in slot 4kt+2h+g+1;
//   * A_g(kt+2) is issued in row g's C(kt,1) slot (its last reader, L(kt,1) of the same row, is one barrier behind) --
//     BETWEEN the MFMAs of that slot (a piece issued between a wave's MFMAs costs ~12 cycles, four of them issued
//     after the MFMAs ~240 cycles of the critical slot: tools/micro/r2_probe.hip);
//     W(kt+1) is issued in row g's L(kt,0) slot (the last reader of W(kt-1), row 1 in slot 4kt-1, is behind);
//   * every DMA has 3-5 slots to land; once per stage a counted vmcnt at the end of slot 4kt+3 (row 0: 4 newer DMAs
//     may stay in flight; row 1: 0) precedes the barrier that opens stage kt+1.
template <typename T, int AMODE, int KIND>
__global__ __launch_bounds__(512, 2) void gemm_pp64_kernel(GemmParams p, int band_m) {
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
  const int nk = p.K / BK;

  const T* A = (const T*)p.A + (long long)g * p.strideA_g;
  const T* A2 = p.A2 ? (const T*)p.A2 + (long long)g * p.strideA2_g : nullptr;
  const T* W = (const T*)p.W + (long long)g * p.strideW_g;
  const int srow = lane >> 3, sp = lane & 7;
  int tm, tn;
  pp_tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, band_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  // this wave's 4 A pieces (own half) and 4 W pieces of a stage; piece = 8 rows x 128 B
  // (32-bit element offsets: the launcher routes operands of 2^31 elements or more to the 128x128 kernel)
  unsigned woff[4], aoff[4], aoff2[4];   // aoff2: conv, the same rows of source 2 (other channel count)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rw = (wave * 4 + j) * 8 + srow;
    int n = n0 + rw;
    n = n < p.N ? n : p.N - 1;
    woff[j] = ((unsigned)n * (unsigned)p.ldw + swz8(rw, sp) * 8) * (unsigned)sizeof(T);   // bytes (< 2^32: see launch())
    const int ra = wm * 128 + (wn * 4 + j) * 8 + srow;
    int m = m0 + ra;
    m = m < p.M ? m : p.M - 1;
    if (AMODE == A_DENSE) {
      aoff[j] = ((unsigned)m * (unsigned)p.lda + swz8(ra, sp) * 8) * (unsigned)sizeof(T);
      aoff2[j] = 0;
    } else {   // the pixel's row in the bordered feature maps (mk_common.hpp): a tap is a wave-uniform shift of it
      const unsigned br = (unsigned)bordered_row(m, p.H, p.Wd);
      aoff[j] = (br * (unsigned)p.C1 + swz8(ra, sp) * 8) * (unsigned)sizeof(T);
      aoff2[j] = (br * (unsigned)p.C2 + swz8(ra, sp) * 8) * (unsigned)sizeof(T);
    }
  }
  auto dma_w1 = [&](int s, int j) {
    // wave-uniform base (SGPRs) + the lane's constant 32-bit byte offset: no VALU instruction per piece
    glds16_sv(W + s * BK, woff[j], smem + (s & 1) * STAGE_BYTES + A_BYTES + (wave * 4 + j) * 1024);
  };
  // conv: A stages are issued in K order (0, 1, 2, ...), so where the NEXT stage reads -- source, 3x3 tap, first channel,
  // folded into one wave-uniform base pointer -- is running scalar state, advanced once per stage.  K = 9 taps x C1 channels of source 1, then C2 channels of source 2
  // (the 1x1 shortcut) at the pixel itself.
  // Split operands (p.npass == 3, mk_conv3x3_split): the K range is swept three times -- sweep 0 over the LO activation
  // planes, sweeps 1 and 2 over the HI planes (W holds [W_hi | W_lo | W_hi] along K, so the W side just keeps streaming).
  const bool split = AMODE != A_DENSE && p.npass > 1;
  const T* Acur = split ? (const T*)p.A_lo + (long long)g * p.strideA_g : A;
  const T* A2cur = split && p.A2_lo ? (const T*)p.A2_lo + (long long)g * p.strideA2_g : A2;
  const int ntap = A2 ? 10 : 9;   // K segments of one sweep: nine taps (+ the shortcut source)
  const T* cbase = AMODE == A_DENSE ? A : Acur - (long long)(p.Wd + 2) * p.C1;   // tap (-1, -1), channel 0
  int cleft = p.C1, ctap = 0;
  auto conv_advance = [&]() {
    cleft -= BK;
    const bool wrap = cleft == 0;
    ctap += wrap ? 1 : 0;
    if (wrap && ctap == ntap) {   // end of a sweep (only the split form goes on from here): HI planes from now on
      ctap = 0;
      Acur = A;
      A2cur = A2;
    }
    const int ty = (ctap * 11) >> 5, tx = ctap - 3 * ty;   // ctap / 3, ctap % 3 for 0 <= ctap < 9
    const T* tapbase = ctap < 9 ? Acur + (long long)((ty - 1) * (p.Wd + 1) + tx - 1) * p.C1 : A2cur;
    cbase = wrap ? tapbase : cbase + BK;
    cleft = wrap ? (ctap < 9 ? p.C1 : p.C2) : cleft;
  };
  auto dma_a1 = [&](int s, int j) {
    char* dst = smem + (s & 1) * STAGE_BYTES + (wm * 16 + wn * 4 + j) * 1024;
    if (AMODE == A_DENSE) {
      if constexpr (KIND == 0) {   // plain kernel: may run with split operands (mk_gemm_grouped_split): sweep 0 reads the LO plane
        if (p.npass > 1) {
          const int nkp = nk / 3, sweep = s >= 2 * nkp ? 2 : (s >= nkp ? 1 : 0);
          const T* base = sweep == 0 ? (const T*)p.A_lo + (long long)g * p.strideA_g : A;
          glds16_sv(base + (s - sweep * nkp) * BK, aoff[j], dst);
          return;
        }
      }
      glds16_sv(A + s * BK, aoff[j], dst);
    } else {
      glds16_sv(cbase, ctap < 9 ? aoff[j] : aoff2[j], dst);   // the caller advances behind the 4th piece
    }
  };
  auto dma_w = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_w1(s, j);
  };
  auto dma_a = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a1(s, j);
    if (AMODE != A_DENSE) conv_advance();
  };

  const int fr = lane & 15, fg = lane >> 4;
  auto load_frags = [&](V8* wf, V8* xf, int par, int h) {
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
  f32x4 acc[WMF][4];
#pragma unroll
  for (int i = 0; i < WMF; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // 32 MFMAs of a C slot; DMA: the 4 A pieces of stage s.  A piece is one SALU-addressed instruction (dense rows, and
  // conv rows alike since the feature maps are bordered: round 2's conv pieces were ~15 VALU instructions of per-lane tap /
  // border arithmetic each and had to go out in the wave-row's non-MFMA slot) and goes out BETWEEN the MFMAs (behind MFMA
  // 3, 11, 19, 27), pinned.
  auto mfma32 = [&](const V8* wf, const V8* xf, auto dma, int s) {
    constexpr bool DMA = decltype(dma)::value;
    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int mi = q >> 2, ni = q & 3;
      acc[mi][ni] = Lp<T>::mma16(wf[ni], xf[mi], acc[mi][ni]);
      if (DMA && (q & 7) == 3) {
        __builtin_amdgcn_sched_barrier(0);
        dma_a1(s, q >> 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    if (DMA && AMODE != A_DENSE) conv_advance();   // behind the slot's MFMAs: hipcc turns its selects into a branch
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  V8 wf[4], xf[WMF];
  // one K = 64 stage of wave-row 0 / 1; NEXT1: stage kt+1 exists, NEXT2: stage kt+2 exists (compile time: the slots stay
  // single basic blocks, so the DMA pieces can be pinned between the MFMAs)
  auto stage0 = [&](int kt, auto next1, auto next2) {
    bar();   // slot 4kt
    load_frags(wf, xf, kt & 1, 0);
    if constexpr (decltype(next1)::value) dma_w(kt + 1);
    bar();                      // slot 4kt+1
    mfma32(wf, xf, No{}, 0);
    bar();                      // slot 4kt+2
    load_frags(wf, xf, kt & 1, 1);
    bar();                      // slot 4kt+3
    mfma32(wf, xf, next2, kt + 2);
    if constexpr (decltype(next2)::value)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // stage kt+1 landed; A0(kt+2) may still fly
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto stage1 = [&](int kt, auto next1, auto next2) {
    bar();                      // slot 4kt+1
    load_frags(wf, xf, kt & 1, 0);
    if constexpr (decltype(next1)::value) dma_w(kt + 1);
    bar();                      // slot 4kt+2
    mfma32(wf, xf, No{}, 0);
    bar();                      // slot 4kt+3
    load_frags(wf, xf, kt & 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own share of stage kt+1 (A1 and W) landed
    bar();                      // slot 4kt+4
    mfma32(wf, xf, next2, kt + 2);
  };

  // prologue: stage 0 (own A half + W share) and the own A half of stage 1 (nk >= 2, see launch()).
  // Folded LayerNorm (consumer): the row statistics of the tile are requested first and turned into row parameters in LDS
  // while the DMA pieces are in flight.
  const bool ln = KIND == 1 && p.ln_stats != nullptr;
#if defined(MK_LN_ABL) && MK_LN_ABL == 1
  const bool ln_fast = false && ln;   // ablation: no prologue work
#define MK_LN_NO_SLOW 1
#else
  const bool ln_fast = ln && p.ln_nslot == 16;
#endif
  LnRowLoads16 lnl;
  if (ln_fast) lnl.issue(p, m0, tid);
  // folded LayerNorm (producer): the row shifts of the tile (row centring, zeros when off) take the same route into LDS
  constexpr bool PRODUCER = AMODE == A_DENSE && (KIND == 2 || KIND == 3);
  ShiftLoad shl;
  if constexpr (PRODUCER) shl.issue(p, m0, tid, 256);
  dma_a(0);
  dma_w(0);
  dma_a(1);
  const bool publish = ln && p.ln_shift_out != nullptr && n0 == 0;   // first tile column: the row means for the next producer
  if (ln_fast) lnl.template finish<12>(p, tid, (float2*)(smem + 2 * STAGE_BYTES), m0, publish);   // 12 DMA pieces are younger than the loads
#ifndef MK_LN_NO_SLOW
  else if (ln) ln_params_to_lds<256, 512>(p, m0, tid, (float2*)(smem + 2 * STAGE_BYTES), publish);
#endif
  if constexpr (PRODUCER) shl.template finish<12>(tid, 256, (float*)(smem + 2 * STAGE_BYTES));
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if (wm == 0) {
    for (int kt = 0; kt < nk - 2; ++kt) stage0(kt, Yes{}, Yes{});
    stage0(nk - 2, Yes{}, No{});
    stage0(nk - 1, No{}, No{});
    bar();   // row 1's last fragment reads are done, the LDS ring is free
  } else {
    bar();   // slot 0: this wave-row idles
    for (int kt = 0; kt < nk - 2; ++kt) stage1(kt, Yes{}, Yes{});
    stage1(nk - 2, Yes{}, No{});
    stage1(nk - 1, No{}, No{});
  }
  epilogue_lds<T, KIND, AMODE == A_CONV3>(p, acc, smem + wave * 16384, m0, n0, wm, wn, lane, g, (const float2*)(smem + 2 * STAGE_BYTES));
}

template <typename T, int AMODE, int KIND>
int launch_k(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  constexpr int LDS = 2 * 512 * 128 + 256 * 8;   // two stages + the folded LayerNorm's row parameters
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_pp64_kernel<T, AMODE, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  // tile order (pp_tile_coords): outputs at most 4 tiles wide (proj, fc2: N = 1024) are walked m-major with n fastest -- the
  // 32 tiles in flight are the same 8 x 4 block as in a band, but neighbouring CUs share the (large) A panel: -19 % L2-miss
  // traffic and +2.6 % on fc2 (profiles/r04j_gemm_order.txt); wider outputs keep bands of 8 m-tiles (n-groups there cut the
  // traffic as much and cost 1-6 % of time)
  if (band_m == 0) band_m = (AMODE == A_DENSE && ntn <= 4) ? -4 : 8;
  hipLaunchKernelGGL((gemm_pp64_kernel<T, AMODE, KIND>), dim3(ntm * ntn, groups, 1), dim3(512), LDS, st, p, band_m);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

template <typename T>
int launch_dense(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  if (p.xh && p.epi == MK_EPI_LS_RESIDUAL && p.out_f32) return launch_k<T, A_DENSE, 3>(p, groups, st, band_m);
  if (p.xh) return launch_k<T, A_DENSE, 2>(p, groups, st, band_m);
  if (p.ln_stats) return launch_k<T, A_DENSE, 1>(p, groups, st, band_m);
  return launch_k<T, A_DENSE, 0>(p, groups, st, band_m);
}

}  // namespace

int launch_pp64(const GemmParams& p, int groups, int dtype, int amode, hipStream_t st, int band_m) {
  if (amode == A_DENSE) return dtype == MK_BF16 ? launch_dense<__bf16>(p, groups, st, band_m) : launch_dense<_Float16>(p, groups, st, band_m);
  return dtype == MK_BF16 ? launch_k<__bf16, A_CONV3, 0>(p, groups, st, band_m) : launch_k<_Float16, A_CONV3, 0>(p, groups, st, band_m);
}

}  // namespace gemm
}  // namespace mk
