// Do the matrix pipe and the VALU / transcendental unit of one SIMD run concurrently?  (dev micro-benchmark)
//  A: 16 independent v_mfma_f32_32x32x16_bf16 per iteration        B: 64 v_exp_f32 + 128 v_fma_f32 per iteration
//  C: A and B interleaved in ONE wave's program order (1 MFMA : 4 exp : 8 fma)
//  D: two waves per SIMD, one runs A, the other runs B             E: two waves per SIMD, both run C
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[8];
  float x[16], y[16];
  for (int i = 0; i < 16; ++i) {
    for (int j = 0; j < 16; ++j) acc[i & 7][j] = 0.f;
    x[i] = threadIdx.x * 1e-3f - i;
    y[i] = 1.f + i;
  }
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f); b[j] = (__bf16)1.0f; }
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = KIND == 0 || KIND == 2 || KIND == 4 || (KIND == 3 && (wave & 4) == 0);
  const bool do_valu = KIND == 1 || KIND == 2 || KIND == 4 || (KIND == 3 && (wave & 4) != 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (do_mfma) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 7], 0, 0, 0);
      if (do_valu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = (i * 4 + e) & 15;
          x[j] = __builtin_amdgcn_exp2f(x[j]);
          y[j] = __builtin_fmaf(y[j], 1.0001f, 0.5f);
          y[(j + 7) & 15] = __builtin_fmaf(y[(j + 7) & 15], 0.9999f, 0.25f);
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) { s += x[i] + y[i]; for (int j = 0; j < 16; ++j) s += acc[i & 7][j]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int threads) {
  float* out;
  (void)hipMalloc(&out, sizeof(float) * threads * 256);
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, out, 50);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %d waves/SIMD: %7.1f ns per iteration (16 MFMA and/or 64 exp + 128 fma)\n", name, threads / 256, ms * 1e6 / iters);
  (void)hipFree(out);
}

int main() {
  run<0>("A  MFMA only", 256);
  run<1>("B  exp + fma only", 256);
  run<2>("C  both, interleaved in one wave", 256);
  run<0>("A  MFMA only", 512);
  run<1>("B  exp + fma only", 512);
  run<3>("D  wave 0-3: MFMA only, wave 4-7: exp + fma only", 512);
  run<4>("E  both waves run the interleaved stream", 512);
  return 0;
}
