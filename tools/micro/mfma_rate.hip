// Issue rate of the fp32 / bf16 MFMAs on gfx950 (dev micro-benchmark): cycles per instruction for a dependent chain
// and for NACC independent accumulators, one wave per SIMD and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int KIND, int NACC>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x16 acc[NACC];
  f32x4 acc4[NACC];
  for (int i = 0; i < NACC; ++i) {
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int j = 0; j < 4; ++j) acc4[i][j] = 0.f;
  }
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)a; bb[j] = (__bf16)b; }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      if (KIND == 1) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
      if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i) {
    for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int j = 0; j < 4; ++j) s += acc4[i][j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int NACC>
void run(const char* name, int threads, int blocks) {
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double flops_per = KIND == 0 ? 4096.0 : KIND == 1 ? 2048.0 : 32768.0;
  const double n = (double)iters * NACC;
  const double waves = (double)blocks * threads / 64;
  printf("%-26s nacc %d  %d thr x %d blk: %.1f shader-clock ticks / MFMA / wave, %.1f TFLOP/s chip\n", name, NACC, threads, blocks,
         (double)c / n, n * waves * flops_per / (ms * 1e-3) / 1e12);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, 1>("mfma_f32_32x32x2_f32", 256, 256);
  run<0, 4>("mfma_f32_32x32x2_f32", 256, 256);
  run<0, 4>("mfma_f32_32x32x2_f32", 512, 256);
  run<1, 1>("mfma_f32_16x16x4_f32", 256, 256);
  run<1, 4>("mfma_f32_16x16x4_f32", 256, 256);
  run<1, 4>("mfma_f32_16x16x4_f32", 512, 256);
  run<2, 1>("mfma_f32_32x32x16_bf16", 256, 256);
  run<2, 4>("mfma_f32_32x32x16_bf16", 256, 256);
  run<2, 4>("mfma_f32_32x32x16_bf16", 512, 256);
  return 0;
}
