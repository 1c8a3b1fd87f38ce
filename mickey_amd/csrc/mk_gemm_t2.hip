// mickey_amd -- 256x128-tile GEMM schedule with TWO resident workgroups per CU (4 waves each).
#include "mk_gemm_common.hpp"

namespace mk {
namespace gemm {
namespace {

// The ping-pong kernel (mk_gemm_pp64.hip) owns a CU with one 8-wave workgroup: while that workgroup fills its first stage
// or drains its epilogue the matrix pipe idles.  At K = 1024 and 30 tiles per CU that is the price of the 256x256 tile
// (and at the socket's power limit it costs little); at K = 384 (ViT-S: 6 K-stages per tile) or with one
// round of tiles per launch (a single image pair) fill and drain are most of a tile's life.  Here a workgroup is HALF of
// that kernel -- 4 waves, 2 x 2, the same 128 x 64 block and the same 128 accumulator registers per wave, so every
// epilogue of mk_gemm_common.hpp applies unchanged -- with a 256 x 128 tile and 74 KiB of LDS, so two workgroups share a
// CU and one's fill / drain overlaps the other's K loop; and a launch has twice the tiles (N = 384 is 3 tile columns
// instead of 1.5).
//   * K stages of 32 (LDS rows of 64 B; 16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3): a 16-lane group of a
//     fragment read covers 256 B = all 64 banks once), ring of three 24-KiB stages, ONE barrier per stage;
//   * per stage and wave: 6 LDS-DMA pieces (1 KiB = 16 rows x 64 B; SGPR base + constant lane offset, glds16_sv) of stage
//     kt+2, 12 fragment reads, 32 MFMAs; the other workgroup's wave on the same SIMD fills the read / barrier gaps;
//   * price: 1.5x the L2 -> LDS bytes per flop of the 256x256 tile (24 KiB per 2 x 256 x 128 x 32 flop).
constexpr int T2_BK = 32;
template <typename T, int KIND>
__global__ __launch_bounds__(256, 2) void gemm_t2_kernel(GemmParams p, int band_m) {
  using V8 = typename Lp<T>::V8;
  constexpr int BM = 256, BN = 128;
  constexpr int A_BYTES = BM * 64, STAGE_BYTES = (BM + BN) * 64;   // 24 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int nk = p.K / T2_BK;

  const T* A = (const T*)p.A + (long long)g * p.strideA_g;
  const T* W = (const T*)p.W + (long long)g * p.strideW_g;
  int tm, tn;
  pp_tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, band_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  // pieces: 16 rows x 64 B; lane -> row lane >> 2, chunk lane & 3 of the LDS image, fetching the swizzled source chunk
  const int prow = lane >> 2, pch = lane & 3;
  unsigned aoff[4], woff[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 16 + prow;
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    aoff[j] = ((unsigned)m * (unsigned)p.lda + (pch ^ ((r >> 2) & 3)) * 8) * (unsigned)sizeof(T);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + prow;
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    woff[j] = ((unsigned)n * (unsigned)p.ldw + (pch ^ ((r >> 2) & 3)) * 8) * (unsigned)sizeof(T);
  }
  auto dma = [&](int s, int slot) {   // stage s -> ring slot (= s % 3, tracked by the caller)
    char* dst = smem + slot * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_sv(A + s * T2_BK, aoff[j], dst + (wave * 4 + j) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_sv(W + s * T2_BK, woff[j], dst + A_BYTES + (wave * 2 + j) * 1024);
  };
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto step = [&](int slot) {
    const char* sA = smem + slot * STAGE_BYTES;
    const char* sW = sA + A_BYTES;
    V8 wf[4], xf[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rw = wn * 64 + i * 16 + fr;
      wf[i] = *(const V8*)(sW + rw * 64 + ((fg ^ ((rw >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rx = wm * 128 + i * 16 + fr;
      xf[i] = *(const V8*)(sA + rx * 64 + ((fg ^ ((rx >> 2) & 3)) << 4));
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Lp<T>::mma16(wf[ni], xf[mi], acc[mi][ni]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // folded LayerNorm: the consumer's row parameters / the producer's row shifts go to LDS behind the ring
  float2* lnp = (float2*)(smem + 3 * STAGE_BYTES);
  const bool ln = KIND == 1 && p.ln_stats != nullptr;
  const bool publish = ln && p.ln_shift_out != nullptr && n0 == 0;
  if (ln) {
    ln_params_to_lds<128, 256>(p, m0, tid, lnp, publish);
    ln_params_to_lds<128, 256>(p, m0 + 128, tid, lnp + 128, publish);
  }
  if constexpr (KIND == 2 || KIND == 3) {   // row centring (zeros when off)
    const int m = m0 + tid;
    ((float*)lnp)[tid] = (p.ln_shift_in && m < p.M) ? p.ln_shift_in[m] : 0.f;
  }
  dma(0, 0);
  if (nk > 1) dma(1, 1);
  int slot = 0, wslot = 2;
  for (int kt = 0; kt < nk - 2; ++kt) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // stage kt landed; the 6 pieces of stage kt+1 may still fly
    bar();
    dma(kt + 2, wslot);                                 // into the slot whose readers all passed this barrier
    step(slot);
    slot = slot == 2 ? 0 : slot + 1;
    wslot = wslot == 2 ? 0 : wslot + 1;
  }
  if (nk > 1) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    bar();
    step(slot);
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bar();
  step(slot);
  bar();   // the ring is free: the epilogue stages through it (16 KiB per wave)
  epilogue_lds<T, KIND, false, BN>(p, acc, smem + wave * 16384, m0, n0, wm, wn, lane, g, lnp);
}

template <typename T, int KIND>
int launch_k(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  constexpr int LDS = 3 * (256 + 128) * 64 + 256 * 8;   // the ring + row parameters of the folded LayerNorm
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_t2_kernel<T, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 127) / 128;
  hipLaunchKernelGGL((gemm_t2_kernel<T, KIND>), dim3(ntm * ntn, groups, 1), dim3(256), LDS, st, p, band_m);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

template <typename T>
int launch_dense(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  if (p.xh && p.epi == MK_EPI_LS_RESIDUAL && p.out_f32) return launch_k<T, 3>(p, groups, st, band_m);
  if (p.xh) return launch_k<T, 2>(p, groups, st, band_m);
  if (p.ln_stats) return launch_k<T, 1>(p, groups, st, band_m);
  return launch_k<T, 0>(p, groups, st, band_m);
}

}  // namespace

int launch_t2(const GemmParams& p, int groups, int dtype, hipStream_t st, int band_m) {
  return dtype == MK_BF16 ? launch_dense<__bf16>(p, groups, st, band_m) : launch_dense<_Float16>(p, groups, st, band_m);
}

}  // namespace gemm
}  // namespace mk
