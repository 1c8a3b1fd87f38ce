// Issue cost (shader-clock cycles per wave64 instruction) of the VALU ops the attention softmax is made of (dev tool):
// 16 independent dependency chains per wave, 1..4 waves per SIMD, v_exp_f32 alone and interleaved with v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(2))) float f32x2;

// 16 independent 2-wide chains: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 throughput
template <int KIND>
__global__ void kp(float* out, long long* cyc, int iters) {
  f32x2 x[16];
  for (int i = 0; i < 16; ++i) x[i] = f32x2{threadIdx.x * 1e-3f + i, 1.0f + i};
  const f32x2 c = f32x2{1.0001f, 0.9999f}, d = f32x2{0.5f, 0.25f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(d));
      if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
      if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i][0] + x[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
__global__ void k(float* out, long long* cyc, int iters) {
  float x[16], y[16];
  for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = x[i] + 1.f; }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0 || KIND == 2 || KIND == 3) x[i] = __builtin_amdgcn_exp2f(x[i]);
      if (KIND == 1 || KIND == 2) y[i] = __builtin_fmaf(y[i], 1.0001f, 0.5f);
      if (KIND == 3) { y[i] = __builtin_fmaf(y[i], 1.0001f, 0.5f); y[i] = __builtin_fmaf(y[i], 0.9999f, 0.25f); y[i] = fmaxf(y[i], 0.1f); }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i] + y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, bool PK = false>
void run(const char* name, int threads) {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * threads * 256);
  (void)hipMalloc(&cyc, 8);
  const int iters = 40000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  if (PK) hipLaunchKernelGGL((kp<KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, 100);
  else hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, 100);
  (void)hipEventRecord(e0);
  if (PK) hipLaunchKernelGGL((kp<KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  else hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long c;
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double groups_per_simd = (double)iters * 16.0 * (threads / 256);   // one block per CU, threads/256 waves per SIMD
  printf("%-20s %d waves/SIMD: %6.2f counter ticks per group per wave | wall: %.3f ms -> %.2f ns per group per SIMD (counter says %.2f GHz)\n",
         name, threads / 256, (double)c / (iters * 16.0), ms, ms * 1e6 / groups_per_simd, (double)c / (ms * 1e6));
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 4; ++w) run<0>("exp", 256 * w);
  for (int w = 1; w <= 4; ++w) run<1>("fma", 256 * w);
  for (int w = 1; w <= 4; ++w) run<2>("exp + fma", 256 * w);
  for (int w = 1; w <= 4; ++w) run<3>("exp + 2 fma + max", 256 * w);
  for (int w = 1; w <= 4; w *= 2) run<0, true>("v_pk_add_f32", 256 * w);
  for (int w = 1; w <= 4; w *= 2) run<1, true>("v_pk_mul_f32", 256 * w);
  for (int w = 1; w <= 4; w *= 2) run<2, true>("v_pk_fma_f32", 256 * w);
  return 0;
}
