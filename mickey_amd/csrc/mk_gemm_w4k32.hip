// mickey_amd -- one-wave-per-SIMD GEMM schedule on a deep ring of K = 32 stages (experimental: measured 5-6 % slower
// than the two-stage K = 64 form of mk_gemm_w4.hip, i.e. DMA latency is NOT what limits that schedule).
#include <type_traits>

#include "mk_gemm_epi32.hpp"

namespace mk {
namespace gemm {
namespace {

// ---------------------------------------------------------------------------------------------------------
// The same wave layout on a DEEP ring: stages of K = 32 (64-byte LDS rows, 32 KiB per stage), FIVE of them = all 160 KiB
// of LDS.  With two 64-KiB stages at most 64 KiB are in flight per CU and the youngest DMA piece of a stage has ~600
// cycles to land before the once-per-stage vmcnt(0), which under full load (every CU streaming, L2 misses going to
// HBM at 1-2 us) stalls every stage: measured 3800-3900 cycles per K = 64 instead of the 2048 of its MFMAs.  Here a
// stage is issued FIVE stages before it is read: 128 KiB in flight, ~4000 cycles to land, and the per-stage wait is a
// COUNTED vmcnt(24) that only retires the oldest stage.
//   stage s (K = 32, buffer s % 5) = k-steps (s,0), (s,1) of K = 16
//   step (s,0):  16 MFMAs | behind the first 8: fragment reads of (s,1) | then the 4 W pieces of stage s+4
//   step (s,1):  vmcnt(24) [stage s+1 landed]  lgkmcnt(0)  s_barrier  [every wave is done reading stage s]
//                16 MFMAs | fragment reads of (s+1,0) | the 4 A pieces of stage s+5 (into the buffer of stage s)
// A DMA piece is 16 rows x 64 B (half cache lines: the other half is the next stage's piece, an L2 hit).  64-byte rows,
// 16-byte chunk c of row r at c ^ ((r >> 2) & 3): conflict-free for the 32-row fragment reads (ds_read_b128 is served in
// four 16-lane groups whose rows hit each (r & 3) four times, with four different (r >> 2) & 3).
template <typename T, int AMODE>
__global__ __launch_bounds__(256, 1) void gemm_w4k32_kernel(GemmParams p, int band_m) {
  using V8 = typename Lp<T>::V8;
  constexpr int BM = 256, BN = 256, KS = 32, NST = 5;
  constexpr int A_BYTES = BM * 64, STAGE_BYTES = (BM + BN) * 64;   // 32 KiB per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int nk = p.K / KS;
  int tm, tn;
  pp_tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, band_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const T* A = (const T*)p.A + (long long)g * p.strideA_g;
  const T* A2 = p.A2 ? (const T*)p.A2 + (long long)g * p.strideA2_g : nullptr;
  const T* W = (const T*)p.W + (long long)g * p.strideW_g;
  const int srow = lane >> 2, sc = lane & 3;
  // this wave's 4 A pieces and 4 W pieces of a stage; piece = 16 rows x 64 B
  unsigned woff[4], aoff[4];
  int ay[AMODE == A_CONV3 ? 4 : 1], ax[AMODE == A_CONV3 ? 4 : 1];
  bool avalid[AMODE == A_CONV3 ? 4 : 1];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 16 + srow;
    const int csw = (sc ^ ((r >> 2) & 3)) * 8;   // swizzled source chunk, in elements
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    woff[j] = (unsigned)n * (unsigned)p.ldw + csw;
    int m = m0 + r;
    const bool ok = m < p.M;
    m = ok ? m : p.M - 1;
    if (AMODE == A_DENSE) {
      aoff[j] = (unsigned)m * (unsigned)p.lda + csw;
    } else {
      const int pix = m % (p.H * p.Wd);
      ay[j] = pix / p.Wd;
      ax[j] = pix % p.Wd;
      avalid[j] = ok;
      aoff[j] = m;
    }
  }
  // piece j of operand A (which = 0) or W (which = 1) of stage s into ring slot `slot`
  auto dma1 = [&](int s, int slot, int which, int j) {
    char* dst = smem + slot * STAGE_BYTES + which * A_BYTES + (wave * 4 + j) * 1024;
    const int k0 = s * KS;
    if (which == 1) {
      glds16(W + (woff[j] + (unsigned)k0), dst);
    } else if (AMODE == A_DENSE) {
      glds16(A + (aoff[j] + (unsigned)k0), dst);
    } else {
      const int kc = 9 * p.C1;
      const T* src;
      int cs, c0, dy, dx;
      if (k0 < kc) {
        const int tap = k0 / p.C1;
        c0 = k0 - tap * p.C1;
        dy = tap / 3 - 1;
        dx = tap % 3 - 1;
        src = A;
        cs = p.C1;
      } else {
        c0 = k0 - kc;
        dy = dx = 0;
        src = A2;
        cs = p.C2;
      }
      const int r = (wave * 4 + j) * 16 + srow;
      const int yy = ay[j] + dy, xx = ax[j] + dx;
      const bool ok = avalid[j] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
      const T* sp_ = ok ? src + ((long long)aoff[j] + dy * p.Wd + dx) * cs + c0 + (sc ^ ((r >> 2) & 3)) * 8
                        : (const T*)p.zero_page + sc * 8;
      glds16(sp_, dst);
    }
  };
  const int r32 = lane & 31, hi = lane >> 5;
  // fragment i = 0..3: W rows (A operand of the MFMA), 4..7: activation rows (B operand), k-step kk of ring slot `slot`
  auto load_frag = [&](int slot, int kk, int i) {
    const char* sA = smem + slot * STAGE_BYTES;
    const int row = (i < 4 ? wn * 128 + i * 32 : wm * 128 + (i - 4) * 32) + r32;
    return *(const V8*)(sA + (i < 4 ? A_BYTES : 0) + row * 64 + (((kk * 2 + hi) ^ ((r32 >> 2) & 3)) << 4));
  };
  f32x16 acc[4][4];   // [row block][feature block]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;

  V8 f0[8], f1[8];   // fragment sets: [0..3] W, [4..7] activations
  // One k-step: 16 MFMAs on `cur`; behind each of the first 8 one fragment read of the next k-step (the youngest is 8
  // MFMAs old when it is needed), behind the next 4 one LDS-DMA piece; a scheduling barrier pins exactly this order.
  auto kstep = [&](const V8* cur, V8* nxt, auto reads, int rslot, int rkk, auto dma, int ds, int dslot, int dwhich) {
    constexpr bool READS = decltype(reads)::value, DMA = decltype(dma)::value;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int mi = q >> 2, ni = q & 3;
      acc[mi][ni] = Lp<T>::mma32(cur[ni], cur[4 + mi], acc[mi][ni]);
      if (READS && q < 8) {   // in the order the next k-step consumes them: W0, X0, W1, W2, W3, X1, X2, X3
        constexpr int order[8] = {0, 4, 1, 2, 3, 5, 6, 7};
        nxt[order[q]] = load_frag(rslot, rkk, order[q]);
      }
      if (DMA && q >= 8 && q < 12) dma1(ds, dslot, dwhich, q - 8);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // VM = DMA pieces that may stay in flight at the barrier (8 per stage issued beyond s+1); NEXT = stage s+1 exists
  auto stage = [&](int s, int slot, auto dma_w, auto dma_a, auto next, auto vm) {
    constexpr int VM = decltype(vm)::value;
    const int s4 = slot == 0 ? 4 : slot - 1;   // ring slot of stage s+4
    const int s1 = slot == 4 ? 0 : slot + 1;   // ring slot of stage s+1
    kstep(f0, f1, Yes{}, slot, 1, dma_w, s + 4, s4, 1);
    if constexpr (VM == 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    else if constexpr (VM == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if constexpr (VM == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    kstep(f1, f0, next, s1, 0, dma_a, s + 5, slot, 0);
  };
  using V24 = std::integral_constant<int, 24>;
  using V16 = std::integral_constant<int, 16>;
  using V8_ = std::integral_constant<int, 8>;
  using V0 = std::integral_constant<int, 0>;
  // prologue: stages 0..3 and the A pieces of stage 4 (nk >= 6: the launcher sends shorter K elsewhere)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma1(t, t, 0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma1(t, t, 1, j);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) dma1(4, 4, 0, j);
  asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    constexpr int order[8] = {0, 4, 1, 2, 3, 5, 6, 7};
    f0[order[i]] = load_frag(0, 0, order[i]);
  }
  int slot = 0;
  auto adv = [&]() { slot = slot == 4 ? 0 : slot + 1; };
  int s = 0;
  for (; s < nk - 5; ++s, adv()) stage(s, slot, Yes{}, Yes{}, Yes{}, V24{});
  stage(s, slot, Yes{}, No{}, Yes{}, V24{}); ++s; adv();    // s = nk-5: stage nk-1's W pieces still to issue
  stage(s, slot, No{}, No{}, Yes{}, V16{}); ++s; adv();     // nk-4
  stage(s, slot, No{}, No{}, Yes{}, V8_{}); ++s; adv();     // nk-3
  stage(s, slot, No{}, No{}, Yes{}, V0{}); ++s; adv();      // nk-2
  stage(s, slot, No{}, No{}, No{}, V0{});                   // nk-1 (its barrier: everyone is done reading the ring)
  epilogue32<T>(p, acc, smem + wave * 32768, m0, n0, wm, wn, lane, g);
}

template <typename T, int AMODE>
int launch_k32(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  constexpr int LDS = 5 * 512 * 64;   // 160 KiB
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_w4k32_kernel<T, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  hipLaunchKernelGGL((gemm_w4k32_kernel<T, AMODE>), dim3(ntm * ntn, groups, 1), dim3(256), LDS, st, p, band_m);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // namespace

int launch_w4k32(const GemmParams& p, int groups, int dtype, int amode, hipStream_t st, int band_m) {
  if (amode == A_DENSE)
    return dtype == MK_BF16 ? launch_k32<__bf16, A_DENSE>(p, groups, st, band_m) : launch_k32<_Float16, A_DENSE>(p, groups, st, band_m);
  return dtype == MK_BF16 ? launch_k32<__bf16, A_CONV3>(p, groups, st, band_m) : launch_k32<_Float16, A_CONV3>(p, groups, st, band_m);
}

}  // namespace gemm
}  // namespace mk
