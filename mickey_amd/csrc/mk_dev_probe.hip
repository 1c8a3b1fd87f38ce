// mickey_amd -- development probe (include/mickey_hip_dev.h): what the socket power limit lets through the matrix pipe ALONE.
//
// Back-to-back v_mfma_f32_16x16x32 (bf16) on operands held in registers -- no LDS, no L2, no HBM traffic -- two waves per SIMD on
// every CU.  bench.py times one launch with HIP events and reports the rate as `roofline.peak_sustained`: the ceiling of the
// instruction the GEMM kernels are built from on THIS box on THIS day (LABNOTES R4.11: ~2.05 PFLOP/s on pseudo-random operands,
// 2.5 on zeros), next to the 2.5 PFLOP/s spec every fraction is quoted against.  Not product code: nothing of a forward calls it.
#include "mk_common.hpp"

namespace mk {
namespace {

__device__ __forceinline__ unsigned lcg_next(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return s;
}

// 8 independent 16x16 accumulators per wave (32 registers), four operand sets rotating so that consecutive MFMAs see different bits
template <bool ZERO>
__global__ __launch_bounds__(512) void mfma_sustained_kernel(float* __restrict__ out, int iters) {
  unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[i][j] = ZERO ? (__bf16)0.f : (__bf16)(((int)(lcg_next(s) >> 16) - 32768) * (1.f / 32768.f));
      b[i][j] = ZERO ? (__bf16)0.f : (__bf16)(((int)(lcg_next(s) >> 16) - 32768) * (1.f / 32768.f));
    }
  f32x4 c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c[2 * r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b[r], c[2 * r], 0, 0, 0);
      c[2 * r + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b[(r + 1) & 3], c[2 * r + 1], 0, 0, 0);
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += c[i][0] + c[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}

}  // namespace
}  // namespace mk

using namespace mk;

extern "C" {

// scratch: >= workgroups * 512 floats.  Flops of one launch = workgroups * 8 waves * iters * 8 MFMAs * 16384.
int mk_dev_mfma_sustained(float* scratch, int workgroups, int iters, int zero_operands, mk_stream_t stream) {
  MK_CHECK_ARG(scratch && workgroups > 0 && iters > 0, "mk_dev_mfma_sustained: bad args");
  if (zero_operands) hipLaunchKernelGGL(mfma_sustained_kernel<true>, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, scratch, iters);
  else hipLaunchKernelGGL(mfma_sustained_kernel<false>, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, scratch, iters);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // extern "C"
