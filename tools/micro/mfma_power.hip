// Sustained bf16 MFMA rate of the whole chip at the socket power limit (dev micro-benchmark): the same flops issued as
// v_mfma_f32_16x16x32_bf16 (what the GEMM kernels use) and as v_mfma_f32_32x32x16_bf16 (half the instructions and half the A/B
// register reads per flop), operands in registers -- pseudo-random values rotating over four sets, or zeros -- two waves per
// SIMD on every CU, ~100 ms per measurement.  Question: does the instruction shape / the data change what the power limit lets
// through?   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/micro/mfma_power.hip && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int SHAPE, bool ZERO>   // SHAPE 0: 16x16x32, 8 accumulators of 4; 1: 32x32x16, 4 accumulators of 16 (64 registers either way)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = ZERO ? (__bf16)0.f : (__bf16)(((int)(lcg(s) >> 16) - 32768) * (1.f / 32768.f));
      b[i][j] = ZERO ? (__bf16)0.f : (__bf16)(((int)(lcg(s) >> 16) - 32768) * (1.f / 32768.f));
    }
  f32x4 c4[8];
  f32x16 c16[4];
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) c16[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // operand set r: consecutive MFMAs see different A / B bits
      if (SHAPE == 0) {
        c4[2 * r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b[r], c4[2 * r], 0, 0, 0);
        c4[2 * r + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b[(r + 1) & 3], c4[2 * r + 1], 0, 0, 0);
      } else {
        c16[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b[r], c16[r], 0, 0, 0);
      }
    }
  }
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += c4[i][0] + c4[i][3];
  for (int i = 0; i < 4; ++i) t += c16[i][0] + c16[i][15];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int SHAPE, bool ZERO>
double run(float* out, int iters, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(blocks), dim3(512), 0, 0, out, iters / 8);   // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 8.0 * iters * (SHAPE == 0 ? 8.0 * 16384.0 : 4.0 * 32768.0);
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount;   // one 8-wave workgroup per CU = two waves per SIMD
  float* out;
  hipMalloc(&out, (size_t)blocks * 512 * 4);
  const int iters = 400000;   // 256 CUs x 8 waves x 400 k x 131 k flop = 107 Tflop per launch: ~60-100 ms
  for (int rep = 0; rep < 3; ++rep) {
    const double r16 = run<0, false>(out, iters, blocks), r32 = run<1, false>(out, iters, blocks);
    const double z16 = run<0, true>(out, iters, blocks), z32 = run<1, true>(out, iters, blocks);
    printf("sustained bf16 MFMA, %d CUs x 8 waves: 16x16x32 %7.1f TF   32x32x16 %7.1f TF   | zero operands: 16x16x32 %7.1f TF   32x32x16 %7.1f TF\n",
           blocks, r16, r32, z16, z32);
  }
  return 0;
}
