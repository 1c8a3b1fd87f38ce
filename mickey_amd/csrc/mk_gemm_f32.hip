// mickey_amd -- exact-fp32 GEMM / implicit-GEMM conv (the parity mode, dtype MK_F32).
//
// The reference runs the four head stacks in fp32 always and the encoder in fp32 when MICKEY.DINOV2.FLOAT16 is False
// (reference mickey_extractor.py:31-35,49-56).  This kernel gives that arithmetic on the GPU: fp32 operands, products and
// sums on v_mfma_f32_16x16x4_f32, which is bitwise an fp32 fma chain (no reduced-precision step anywhere), so encoder and
// head outputs agree with the fp32 CPU reference to summation-order round-off (~1e-6).  It is 1/16 of the bf16 MFMA rate
// and not tuned beyond a plain two-stage LDS pipeline: correctness evidence, not the product path.
//
// Same tile, LDS image, stager, swizzle and epilogue as the 128x128 16-bit kernel (mk_gemm.hip) with a K tile of 32
// floats (128-byte rows).  A lane's 16-byte fragment chunk holds 4 CONSECUTIVE k; the 16x16x4 MFMA contracts over the
// lane group index (lane >> 4), so MFMA e = 0..3 of a chunk contracts k = {e, 4 + e, 8 + e, 12 + e}: both operands use
// the same k per lane and the four of them cover the 16 k of the chunk row exactly once.
#include "mk_gemm_common.hpp"

namespace mk {
namespace gemm {
namespace {

template <int AMODE>
__global__ __launch_bounds__(256, 1) void gemm_f32_kernel(GemmParams p) {
  using T = float;
  constexpr int WMF = 4, NWN = 2;
  constexpr int NW = 4, BM = 128, BN = 128;
  constexpr int AJ = BM / 8 / NW, WJ = BN / 8 / NW;
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int nk = p.K / KT<T>;
  const int fr = lane & 15, fg = lane >> 4;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BN;
  Stager<T, AMODE, NW, AJ, WJ> st;
  st.init(p, g, m0, n0, wave, lane);
  st.issue(p, smem, smem + A_BYTES, 0);
  f32x4 acc[WMF][4];
#pragma unroll
  for (int i = 0; i < WMF; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* nA = smem + ((kt + 1) & 1) * STAGE_BYTES;
    if (kt + 1 < nk) st.issue(p, nA, nA + A_BYTES, kt + 1);
    const char* sA = smem + (kt & 1) * STAGE_BYTES;
    const char* sW = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 wf[4], xf[WMF];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rw = wn * 64 + i * 16 + fr;
        wf[i] = *(const f32x4*)(sW + rw * 128 + swz8(rw, ks * 4 + fg) * 16);
      }
#pragma unroll
      for (int i = 0; i < WMF; ++i) {
        const int rx = wm * (WMF * 16) + i * 16 + fr;
        xf[i] = *(const f32x4*)(sA + rx * 128 + swz8(rx, ks * 4 + fg) * 16);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < WMF; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni][e], xf[mi][e], acc[mi][ni], 0, 0, 0);
    }
  }
  epilogue<T, WMF>(p, acc, m0, n0, wm, wn, lane, g);
}

template <int AMODE>
int launch_t(const GemmParams& p, int groups, hipStream_t st) {
  constexpr int LDS = 2 * 256 * 128;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_f32_kernel<AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + 127) / 128, ntn = (p.N + 127) / 128;
  hipLaunchKernelGGL((gemm_f32_kernel<AMODE>), dim3(ntm * ntn, groups, 1), dim3(256), LDS, st, p);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // namespace

int launch_f32(const GemmParams& p, int groups, int amode, hipStream_t st) {
  return amode == A_DENSE ? launch_t<A_DENSE>(p, groups, st) : launch_t<A_CONV3>(p, groups, st);
}

}  // namespace gemm
}  // namespace mk
