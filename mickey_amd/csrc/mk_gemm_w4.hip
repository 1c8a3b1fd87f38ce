// mickey_amd -- one-wave-per-SIMD GEMM schedule (256x256 tile, four waves, 128x128 per wave).
#include <type_traits>

#include "mk_gemm_epi32.hpp"

namespace mk {
namespace gemm {
namespace {

// ---------------------------------------------------------------------------------------------------------
// One wave per SIMD: 256x256 tile, FOUR waves (2 x 2), each wave owns a 128x128 block of C = 16 accumulators of
// v_mfma_f32_32x32x16 = 256 registers (the whole 512-entry register file of its SIMD is this wave's: 256 AGPR
// accumulators, fragments and addresses in VGPRs).
// Why (tools/micro/r2_probe.hip, measured on MI355X): (1) VALU-class issue bandwidth is per SIMD, ~1 instruction per
// 4.7 cycles; a v_mfma_f32_32x32x16 takes one of those slots but 32 cycles of matrix pipe, so a single in-order wave can
// issue ~5 other instructions per MFMA for free, and this loop needs 0.5 ds_read_b128 + 0.25 LDS-DMA piece per MFMA;
// (2) an LDS-DMA piece issued BETWEEN a wave's MFMAs costs ~12 cycles, issued after them (as the 8-wave ping-pong did at
// the end of its compute slot) it is serial; (3) the L2 -> LDS DMA path sustains 70 B/clk/CU, this loop needs 32.
// The 128x128 wave tile also reads a third less LDS per flop than 128x64.  No wave-role alternation, no priorities, ONE
// barrier per K = 64 stage, and no latency is exposed at it: the barrier sits in front of the LAST k-step of a stage,
// whose 16 MFMAs run from registers while the first fragments of the next stage are being read.
//   stage s (K = 64, LDS buffer s & 1) = k-steps (s,0..3) of K = 16;  fragment sets F0 / F1 (8 x b128 each)
//   step (s,kk), kk < 3:  16 MFMAs, behind each of the first 8 a fragment read of (s,kk+1), behind the next ones a DMA piece
//   step (s,3):  vmcnt(0) [stage s+1 landed]  lgkmcnt(0)  s_barrier   [every wave is done reading stage s]
//                16 MFMAs | reads (s+1,0) | DMA pieces of stage s+2 (into the buffer of stage s)
// The 16 DMA pieces of stage s+2 are issued in steps (s,3): 6, (s+1,0): 6, (s+1,1): 4: a wave has at most 16 in flight
// and drains them once per stage.
// DSCH: how the 16 DMA pieces of stage s+2 are spread over the k-steps (s,3), (s+1,0), (s+1,1): 0 = 6 + 6 + 4,
// 1 = all 16 in (s,3) (one behind every MFMA), 2 = 8 + 8
template <typename T, int AMODE, int DSCH>
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
    woff[j] = (unsigned)n * (unsigned)p.ldw + swz8(r, sp) * 8;
    int m = m0 + r;
    const bool ok = m < p.M;
    m = ok ? m : p.M - 1;
    if (AMODE == A_DENSE) {
      aoff[j] = (unsigned)m * (unsigned)p.lda + swz8(r, sp) * 8;
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
  // behind the following ones -- and a scheduling barrier that pins exactly this order (left to itself, or with sched_group_barrier, the compiler clusters the 8 reads and the DMA pieces behind the
  // second MFMA and the matrix pipe idles for ~400 of the step's ~900 cycles while they issue: 929 vs 1104 TFLOP/s).
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
  constexpr int N3 = DSCH == 0 ? 6 : DSCH == 1 ? 16 : 8, NA = DSCH == 0 ? 6 : DSCH == 1 ? 0 : 8, NB = DSCH == 0 ? 4 : 0;
  constexpr int Q3 = DSCH == 1 ? 0 : 8;
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
  if constexpr (N3 == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N3 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
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

template <typename T, int AMODE, int DSCH = 0>
int launch_t(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  constexpr int LDS = 2 * 512 * 128;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_w4_kernel<T, AMODE, DSCH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  hipLaunchKernelGGL((gemm_w4_kernel<T, AMODE, DSCH>), dim3(ntm * ntn, groups, 1), dim3(256), LDS, st, p, band_m);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // namespace

// dev: DMA-distribution variants of the bf16 dense kernel (mk_gemm_set_tile 12 / 13)
int launch_w4_variant(const GemmParams& p, int groups, hipStream_t st, int band_m, int dsch) {
  return dsch == 1 ? launch_t<__bf16, A_DENSE, 1>(p, groups, st, band_m) : launch_t<__bf16, A_DENSE, 2>(p, groups, st, band_m);
}

int launch_w4(const GemmParams& p, int groups, int dtype, int amode, hipStream_t st, int band_m) {
  if (amode == A_DENSE)
    return dtype == MK_BF16 ? launch_t<__bf16, A_DENSE>(p, groups, st, band_m) : launch_t<_Float16, A_DENSE>(p, groups, st, band_m);
  return dtype == MK_BF16 ? launch_t<__bf16, A_CONV3>(p, groups, st, band_m) : launch_t<_Float16, A_CONV3>(p, groups, st, band_m);
}

}  // namespace gemm
}  // namespace mk
