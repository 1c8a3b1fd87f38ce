// mickey_amd -- one-wave-per-SIMD GEMM schedule (256x256 tile, four waves, 128x128 per wave).
#include <type_traits>

#include "mk_gemm_epi32.hpp"

namespace mk {
namespace gemm {
namespace {

// ---------------------------------------------------------------------------------------------------------
// One wave per SIMD: 256x256 tile, FOUR waves (2 x 2), each wave owns a 128x128 block of C = 16 accumulators of
// v_mfma_f32_32x32x16 = 256 registers (the whole 512-entry register file of its SIMD is this wave's: 256 AGPR
// accumulators, fragments and addresses in VGPRs).  The ALTERNATIVE large-problem schedule (mk_gemm_set_tile(10)); the
// default is the 8-wave ping-pong of mk_gemm_pp64.hip, which measures 7-15 % faster on the encoder shapes.
// Idea (tools/micro/r2_probe.hip): VALU-class issue bandwidth is per SIMD, ~1 instruction per 4.7 cycles; a 32x32x16 MFMA
// takes one of those slots but 32 cycles of matrix pipe, so a single in-order wave should be able to issue the 0.5
// ds_read_b128 + 0.25 LDS-DMA piece per MFMA this loop needs in the shadow of its own MFMAs -- no wave-role alternation,
// no priorities, ONE barrier per K = 64 stage placed in front of the stage's LAST k-step, whose 16 MFMAs run from
// registers while the next stage's first fragments are read; the 128x128 wave tile reads a third less LDS per flop.
//   stage s (K = 64, LDS buffer s & 1) = k-steps (s,0..3) of K = 16;  fragment sets F0 / F1 (8 x b128 each)
//   step (s,kk), kk < 3:  16 MFMAs, behind each of the first 8 a fragment read of (s,kk+1), behind the next ones a DMA piece
//   step (s,3):  vmcnt(0) [stage s+1 landed]  lgkmcnt(0)  s_barrier   [every wave is done reading stage s]
//                16 MFMAs | reads (s+1,0) | DMA pieces of stage s+2 (into the buffer of stage s)
// The 16 DMA pieces of stage s+2 are issued in steps (s,3): 6, (s+1,0): 6, (s+1,1): 4: a wave has at most 16 in flight
// and drains them once per stage.
// What it measures (tools/micro/r2_kstep.hip replays one k-step of this loop, ns per k-step on MI355X, random operands):
// 16 MFMAs alone 290 (= 18.1 ns per MFMA: the matrix pipe at the sustained clock) | + 8 fragment reads 279 (free) | the
// MFMAs consuming the fragments read one step earlier 308 (wherever the lgkmcnt wait sits, or without it) | + 6 DMA pieces
// 364 (9 ns each even in the SGPR-base form: a vector-memory instruction is NOT absorbed by the MFMA it follows).  The
// kernel's K loop runs at exactly that rate (2740 cycles per stage; s_memtime brackets put only ~130 of them at the
// stage-boundary waits), i.e. ~1.4 PFLOP/s in the loop and 0.94-1.14 PFLOP/s per launch with prologue, epilogue and tail.
// Tried on top (all measured, none faster): a 5-deep ring of K = 32 stages with counted vmcnt(24) (-5 %), all 16 pieces
// right behind the barrier (-3..-5 %), 8 + 8 (same), sched_group_barrier instead of pinned program order (the compiler
// clusters reads and pieces behind the second MFMA: -4 %), 16x16x32 MFMAs with 256 f32x4 accumulators (register
// allocator shuffles accumulators between AGPRs and VGPRs inside the loop).
template <typename T, int AMODE>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(GemmParams p, int band_m) {
  using V8 = typename Lp<T>::V8;
  constexpr int BM = 256, BN = 256;
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;   // 64 KiB per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int nk = p.K / BK;
  int tm, tn;
  pp_tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, band_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const T* A = (const T*)p.A + (long long)g * p.strideA_g;
  const T* A2 = p.A2 ? (const T*)p.A2 + (long long)g * p.strideA2_g : nullptr;
  const T* W = (const T*)p.W + (long long)g * p.strideW_g;
  const int srow = lane >> 3, sp = lane & 7;
  // this wave's 8 A pieces and 8 W pieces of a stage; piece = 8 rows x 128 B (32-bit element offsets, see launch())
  unsigned woff[8], aoff[8];
  int ay[AMODE == A_CONV3 ? 8 : 1], ax[AMODE == A_CONV3 ? 8 : 1];
  bool avalid[AMODE == A_CONV3 ? 8 : 1];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = (wave * 8 + j) * 8 + srow;
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    woff[j] = ((unsigned)n * (unsigned)p.ldw + swz8(r, sp) * 8) * (unsigned)sizeof(T);   // bytes (< 2^32: see launch())
    int m = m0 + r;
    const bool ok = m < p.M;
    m = ok ? m : p.M - 1;
    if (AMODE == A_DENSE) {
      aoff[j] = ((unsigned)m * (unsigned)p.lda + swz8(r, sp) * 8) * (unsigned)sizeof(T);
    } else {
      const int pix = m % (p.H * p.Wd);
      ay[j] = pix / p.Wd;
      ax[j] = pix % p.Wd;
      avalid[j] = ok;
      aoff[j] = m;
    }
  }
  // piece j of operand A (which = 0) or W (which = 1) of stage s
  auto dma1 = [&](int s, int which, int j) {
    char* dst = smem + (s & 1) * STAGE_BYTES + which * A_BYTES + (wave * 8 + j) * 1024;
    const int k0 = s * BK;
    // dense operands: wave-uniform base (operand + k offset, SGPRs) + this lane's constant 32-bit byte offset, so the
    // address costs no VALU instruction per piece (global_load_lds v_off, s[base:base+1])
    if (which == 1) {
      glds16_sv(W + k0, woff[j], dst);
    } else if (AMODE == A_DENSE) {
      glds16_sv(A + k0, aoff[j], dst);
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
      const int r = (wave * 8 + j) * 8 + srow;
      const int yy = ay[j] + dy, xx = ax[j] + dx;
      const bool ok = avalid[j] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
      const T* sp_ = ok ? src + ((long long)aoff[j] + dy * p.Wd + dx) * cs + c0 + swz8(r, sp) * 8 : (const T*)p.zero_page + sp * 8;
      glds16(sp_, dst);
    }
  };
  // piece idx = 0..15 of stage s: A pieces 0..7, then W pieces 0..7
  auto dma_piece = [&](int s, int idx) { dma1(s, idx >> 3, idx & 7); };
  const int r32 = lane & 31, hi = lane >> 5;
  // fragment i = 0..3: W rows (A operand of the MFMA), 4..7: activation rows (B operand), k-step kk of LDS buffer par
  auto load_frag = [&](int par, int kk, int i) {
    const char* sA = smem + par * STAGE_BYTES;
    const int row = (i < 4 ? wn * 128 + i * 32 : wm * 128 + (i - 4) * 32) + r32;
    return *(const V8*)(sA + (i < 4 ? A_BYTES : 0) + row * 128 + swz8(row, kk * 2 + hi) * 16);
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
  // One k-step: 16 MFMAs on `cur`, each followed by at most ONE memory instruction -- a fragment read of the next
  // k-step behind each of the first 8 (the youngest read is 8 MFMAs old when the next step needs it), an LDS-DMA piece
  // behind the following ones -- and a scheduling barrier that pins exactly this order.
  auto kstep = [&](const V8* cur, V8* nxt, auto reads, int rpar, int rkk, int ds, auto dq0, auto d0, auto ndma) {
    constexpr bool READS = decltype(reads)::value;
    constexpr int DQ0 = decltype(dq0)::value, D0 = decltype(d0)::value, NDMA = decltype(ndma)::value;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int mi = q >> 2, ni = q & 3;
      acc[mi][ni] = Lp<T>::mma32(cur[ni], cur[4 + mi], acc[mi][ni]);
      if (READS && q < 8) {   // in the order the next k-step consumes them: W0, X0, W1, W2, W3, X1, X2, X3
        constexpr int order[8] = {0, 4, 1, 2, 3, 5, 6, 7};
        nxt[order[q]] = load_frag(rpar, rkk, order[q]);
      }
      if (q >= DQ0 && q - DQ0 < NDMA) dma_piece(ds, D0 + q - DQ0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // pieces issued in steps (s,3) | (s+1,0) | (s+1,1) and the MFMA slot of the first one
  constexpr int N3 = 6, NA = 6, NB = 4;
  constexpr int Q3 = 8;
  using I0 = std::integral_constant<int, 0>;
  using I8 = std::integral_constant<int, 8>;
  // one K = 64 stage; NEXT1: stage s+1 exists, NEXT2: stage s+2 exists
  auto stage = [&](int s, auto next1, auto next2) {
    constexpr bool NEXT1 = decltype(next1)::value, NEXT2 = decltype(next2)::value;
    const int par = s & 1;
    kstep(f0, f1, Yes{}, par, 1, s + 1, I8{}, std::integral_constant<int, N3>{}, std::integral_constant<int, NEXT1 ? NA : 0>{});
    kstep(f1, f0, Yes{}, par, 2, s + 1, I8{}, std::integral_constant<int, N3 + NA>{}, std::integral_constant<int, NEXT1 ? NB : 0>{});
    kstep(f0, f1, Yes{}, par, 3, 0, I0{}, I0{}, I0{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // stage s+1 landed; my reads of stage s are done
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    kstep(f1, f0, next1, par ^ 1, 0, s + 2, std::integral_constant<int, Q3>{}, I0{}, std::integral_constant<int, NEXT2 ? N3 : 0>{});
  };
  // prologue: all of stage 0, the step-(s,3) share of stage 1 (nk >= 2: the launcher sends shorter K to the 128x128 kernel)
#pragma unroll
  for (int i = 0; i < 16; ++i) dma_piece(0, i);
#pragma unroll
  for (int i = 0; i < N3; ++i) dma_piece(1, i);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) f0[i] = load_frag(0, 0, i);
  for (int s = 0; s < nk - 2; ++s) stage(s, Yes{}, Yes{});
  stage(nk - 2, Yes{}, No{});
  stage(nk - 1, No{}, No{});
  // every wave passed the last barrier after its final fragment reads: the ring is free, 32 KiB of it per wave
  epilogue32<T>(p, acc, smem + wave * 32768, m0, n0, wm, wn, lane, g);
}

template <typename T, int AMODE>
int launch_t(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  constexpr int LDS = 2 * 512 * 128;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_w4_kernel<T, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  hipLaunchKernelGGL((gemm_w4_kernel<T, AMODE>), dim3(ntm * ntn, groups, 1), dim3(256), LDS, st, p, band_m);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // namespace

int launch_w4(const GemmParams& p, int groups, int dtype, int amode, hipStream_t st, int band_m) {
  if (amode == A_DENSE)
    return dtype == MK_BF16 ? launch_t<__bf16, A_DENSE>(p, groups, st, band_m) : launch_t<_Float16, A_DENSE>(p, groups, st, band_m);
  return dtype == MK_BF16 ? launch_t<__bf16, A_CONV3>(p, groups, st, band_m) : launch_t<_Float16, A_CONV3>(p, groups, st, band_m);
}

}  // namespace gemm
}  // namespace mk
