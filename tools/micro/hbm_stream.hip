// What the HBM side of this part sustains for the access patterns of the bandwidth-bound stages (dev micro-benchmark, round 6):
//   write    1.44 GB of fp32 in 16-byte stores, fully contiguous                      (an upper bound for any output writer)
//   rows     the same bytes as rows of 1938 floats = 7752 B (60.56 cache lines: every row starts and ends inside a 128-byte line),
//            one wave per row segment -- the matcher's scores / kp_scores / final_scores [B, 1938, 1938]
//   rows3    three such matrices written by the same workgroup, row by row (what dual_softmax_split_apply_kernel does)
//   copy     read fp32, write 16 bit (the row kernels: LayerNorm-like, 6 B per element)
//   read     a 481-MB fp32 matrix read once (the sampler's histogram pass)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_stream tools/micro/hbm_stream.hip && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

__global__ __launch_bounds__(256) void k_write(f32x4* out, long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += stride) out[i] = f32x4{1.f, 2.f, 3.f, (float)i};
}

// rows of `n` floats (n % 2 == 0): workgroup = 32 consecutive rows of one matrix, wave w writes rows w, w + 4, ... in 8-byte stores
// (7752 B rows are 8-byte but not 16-byte aligned)
template <int NMAT>
__global__ __launch_bounds__(256) void k_rows(float* o0, float* o1, float* o2, long long rows, int n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = blockIdx.x * 32LL;
  for (int r = wave; r < 32; r += 4) {
    const long long row = r0 + r;
    if (row >= rows) return;
    for (int c = lane * 2; c < n; c += 128) {
      const float2 v = make_float2((float)c, (float)row);
      *(float2*)(o0 + row * n + c) = v;
      if (NMAT > 1) *(float2*)(o1 + row * n + c) = v;
      if (NMAT > 2) *(float2*)(o2 + row * n + c) = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_copy(const f32x4* in, f16x4* out, long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = __builtin_nontemporal_load(in + i);
    out[i] = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  }
}

__global__ __launch_bounds__(256) void k_read(const f32x4* in, float* sink, long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  float a = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += 4 * stride) {   // four 16-byte loads in flight per thread
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i + u * stride < n4 ? __builtin_nontemporal_load(in + i + u * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) a += v[u][0] + v[u][3];
  }
  if (a == 12345.678f) sink[0] = a;
}

template <typename F>
double timed(F launch, double bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double best = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    if (rep > 0 && tbs > best) best = tbs;   // rep 0 = warm-up
  }
  return best;
}

int main() {
  const int B = 32, n = 1938;
  const long long rows = (long long)B * n, elems = rows * n;           // one [32, 1938, 1938] fp32 matrix: 481 MB
  float *a, *b, *c;
  hipMalloc(&a, elems * 4);
  hipMalloc(&b, elems * 4);
  hipMalloc(&c, elems * 4);
  hipMemset(a, 0, elems * 4);
  const int grid = 256 * 16;
  const double one = (double)elems * 4;
  const double w1 = timed([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, (f32x4*)a, elems / 4); }, one);
  const double r1 = timed([&] { hipLaunchKernelGGL(k_rows<1>, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, 0, a, b, c, rows, n); }, one);
  const double r3 = timed([&] { hipLaunchKernelGGL(k_rows<3>, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, 0, a, b, c, rows, n); }, 3 * one);
  const double cp = timed([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const f32x4*)a, (f16x4*)b, elems / 4); }, one * 1.5);
  const double rd = timed([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const f32x4*)a, c, elems / 4); }, one);
  printf("HBM streaming, %.0f MB per matrix (best of 4 after a warm-up): contiguous 16-B writes %.2f TB/s | rows of 7752 B, one matrix %.2f TB/s | "
         "three matrices per workgroup %.2f TB/s | fp32 -> fp16 copy %.2f TB/s (read + write) | read once %.2f TB/s\n",
         one / 1e6, w1, r1, r3, cp, rd);
  return 0;
}
