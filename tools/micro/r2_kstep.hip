// What does a k-step of the one-wave-per-SIMD GEMM cost, piece by piece?  (dev micro-benchmark, MI355X)
// One workgroup of 4 waves per CU (1 wave per SIMD), 128 KiB of LDS, per iteration = one k-step of mk_gemm_w4.hip:
//   V0  16 x v_mfma_f32_32x32x16_bf16 (8 accumulators used twice)
//   V1  + 8 x ds_read_b128 behind MFMAs 0..7 (results unused)
//   V2  the MFMAs consume the fragments read in the PREVIOUS iteration (double-buffered sets, counted lgkmcnt)
//   V3  V2 + 6 LDS-DMA pieces (global_load_lds, saddr form) behind MFMAs 8..13, <= 12 in flight
//   V4  V3 + s_barrier every 4th iteration
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, const char* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 f0[8], f1[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { f0[i][j] = (__bf16)(lane * 1e-3f + i); f1[i][j] = (__bf16)(1.f + j); }
  // fragment addresses as in the GEMM: row = i*32 + (lane & 31), 128-byte rows, swizzled 16-byte chunk
  const int r32 = lane & 31, hi = lane >> 5;
  const unsigned abase = (unsigned)(size_t)smem + (unsigned)((wave & 1) * 16384 + r32 * 128 + ((hi ^ ((r32 >> 1) & 7)) << 4));
  const unsigned voff = (unsigned)((lane >> 3) * 2048 + (lane & 7) * 16);
  const char* gsrc = src + (size_t)blockIdx.x * 131072;
  const unsigned ldsdma = (unsigned)(size_t)smem + 65536 + wave * 16384;
  auto mfma = [&](f32x16& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  };
  auto dsr = [&](bf16x8& d, int i) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(abase), "n"(0), "n"(0));
  };
  (void)dsr;
  int pc = 0;
  auto step = [&](int it, bf16x8* cur, bf16x8* nxt) {
    if (V >= 1 && V != 5 && V != 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (V == 3 || V == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (V >= 4 && (it & 3) == 3) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (V >= 2) mfma(acc[q & 7], cur[q & 3], cur[4 + (q >> 2)]);
      else mfma(acc[q & 7], f0[q & 3], f0[4 + (q >> 2)]);
      if (V >= 1 && q < 8) {
        bf16x8 t;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(abase), "n"(q * 4096));
        if (V >= 2) nxt[q] = t; else asm volatile("" ::"v"(t));
      }
      if (V == 6 && q == 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if ((V == 3 || V == 4) && q >= 8 && q < 14) {
        unsigned keep;
        const unsigned dst = ldsdma + (pc & 15) * 1024;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff + (unsigned)((pc & 7) * 128)), "s"(gsrc + ((pc >> 3) & 7) * 16384), "s"(dst) : "memory");
        ++pc;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int it = 0; it < iters; it += 2) {
    step(it, f0, f1);
    step(it + 1, f1, f0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 8; ++i) s += (float)f0[i][0] + (float)f1[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(const char* name, float* out, const char* src) {
  (void)hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 131072, 0, out, src, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 131072, 0, out, src, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("  !! %s\n", hipGetErrorString(e));
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-72s %7.1f ns per k-step (16 MFMA: %5.1f ns each)\n", name, ms * 1e6 / iters, ms * 1e6 / iters / 16);
}


typedef __attribute__((ext_vector_type(4))) float f32x4;
// 16x16x32 form of the same work: k32-step of a 128x128 wave tile = 64 MFMAs, 16 fragment reads, 12 DMA pieces
template <int V>
__global__ __launch_bounds__(256, 1) void k16(float* out, const char* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc[64];
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  bf16x8 f0[16], f1[16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 8; ++j) { f0[i][j] = (__bf16)(lane * 1e-3f + i); f1[i][j] = (__bf16)(1.f + j); }
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned abase = (unsigned)(size_t)smem + (unsigned)((wave & 1) * 16384 + fr * 128 + ((fg ^ ((fr >> 1) & 7)) << 4));
  const unsigned voff = (unsigned)((lane >> 3) * 2048 + (lane & 7) * 16);
  const char* gsrc = src + (size_t)blockIdx.x * 131072;
  const unsigned ldsdma = (unsigned)(size_t)smem + 65536 + wave * 16384;
  int pc = 0;
  auto step = [&](bf16x8* cur, bf16x8* nxt) {
    if (V >= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (V >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 64; ++q) {
      const bf16x8& a = V >= 1 ? cur[q & 7] : f0[q & 7];
      const bf16x8& b = V >= 1 ? cur[8 + (q >> 3)] : f0[8 + (q >> 3)];
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
      if (V >= 1 && q < 32 && (q & 1) == 0) {
        bf16x8 t;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(abase), "n"((q >> 1) * 2048));
        nxt[q >> 1] = t;
      }
      if (V >= 2 && q >= 32 && q < 56 && (q & 1) == 0) {
        unsigned keep;
        const unsigned dst = ldsdma + (pc & 15) * 1024;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff + (unsigned)((pc & 7) * 128)), "s"(gsrc + ((pc >> 3) & 7) * 16384), "s"(dst) : "memory");
        ++pc;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int it = 0; it < iters; it += 2) {
    step(f0, f1);
    step(f1, f0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0;
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  for (int i = 0; i < 16; ++i) s += (float)f0[i][0] + (float)f1[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run16(const char* name, float* out, const char* src) {
  (void)hipFuncSetAttribute((const void*)k16<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 10000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k16<V>, dim3(256), dim3(256), 131072, 0, out, src, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k16<V>, dim3(256), dim3(256), 131072, 0, out, src, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("  !! %s\n", hipGetErrorString(e));
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-72s %7.1f ns per k32-step = %6.1f ns per k16-equivalent (64 MFMA: %5.2f ns each)\n", name, ms * 1e6 / iters, ms * 1e6 / iters / 2,
         ms * 1e6 / iters / 64);
}

int main() {
  float* out;
  char* src;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256);
  (void)hipMalloc(&src, 256 * 131072);
  (void)hipMemset(src, 1, 256 * 131072);
  run<0>("V0 16 MFMA", out, src);
  run<1>("V1 + 8 ds_read_b128 behind MFMAs 0..7 (unused)", out, src);
  run<2>("V2 MFMAs consume the fragments of the previous iteration", out, src);
  run<3>("V3 + 6 LDS-DMA pieces behind MFMAs 8..13", out, src);
  run<4>("V4 + s_barrier every 4th iteration", out, src);
  run<5>("V5 = V2 without any lgkmcnt wait (timing only)", out, src);
  run<6>("V6 = V2 with the lgkmcnt(0) behind MFMA 11 of the issuing step", out, src);
  run16<0>("W0 64 x v_mfma_f32_16x16x32_bf16 (same flops as 2 k-steps)", out, src);
  run16<1>("W1 + 16 ds_read_b128 behind MFMAs 0,2,..30, consumed next iteration", out, src);
  run16<2>("W2 + 12 LDS-DMA pieces behind MFMAs 32,34,..54", out, src);
  return 0;
}
