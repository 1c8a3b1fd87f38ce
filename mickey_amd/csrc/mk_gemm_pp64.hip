// mickey_amd -- 8-wave full-line ping-pong GEMM schedule (256x256 tile, two waves per SIMD).
#include <type_traits>

#include "mk_gemm_common.hpp"

namespace mk {
namespace gemm {
// persistent tile loop of the dense encoder launches: 2 (default, mk_gemm_set_tile 602) = the producers (proj, fc2, patch embed: KIND 2 / 3),
// 0 (600) = wherever it applies, consumers (qkv, fc1) too, 1 (601) = never.  Round 6, three alternating runs of the whole forward on one
// box (profiles/r06u_persist_ab.txt): encoder GEMMs 76.6 ms per step with the producers only, 77.1 with all, 77.2 with none -- fc1 loses
// 0.7 - 1.8 % to the tile loop on every box measured, qkv is level, proj gains 3 - 4 %, fc2 0.5 - 1 % (r05 / r06_gemm_persistent.txt).
int g_pp64_persist = 2;
namespace {

// Ping-pong with FULL-LINE LDS-DMA pieces: 256x256 tile, 8 waves (wave-row g = wave>>2), LDS stages of K = 64
// (128-byte rows, the swz8 swizzle of the plain kernel), computed in two K = 32 sub-steps h.  A piece is 8 rows x
// 128 B (8 full cache lines).  Only two 64-KiB stages fit, which is deep enough because the operand halves are released
// at different times:
//   * wave-row g loads AND reads only its own A half (rows 128g..128g+127); W is loaded and read by everyone;
//   * slots (barrier at every boundary):  row g does  L(kt,h) [12 fragment reads] in slot 4kt+2h+g  and
//     C(kt,h) [32 MFMAs from registers] in slot 4kt+2h+g+1;
//   * A_g(kt+2) is issued in row g's C(kt,1) slot (its last reader, L(kt,1) of the same row, is one barrier behind) --
//     BETWEEN the MFMAs of that slot (a piece issued between a wave's MFMAs costs ~12 cycles, four of them issued
//     after the MFMAs ~240 cycles of the critical slot: tools/micro/r2_probe.hip);
//     W(kt+1) is issued in row g's L(kt,0) slot (the last reader of W(kt-1), row 1 in slot 4kt-1, is behind);
//   * every DMA has 3-5 slots to land; once per stage a counted vmcnt at the end of slot 4kt+3 (row 0: 4 newer DMAs
//     may stay in flight; row 1: 0) precedes the barrier that opens stage kt+1.
//
// PERSIST (round 5; dense operands, even number of K stages): gridDim.x = #CUs workgroups, each walking the tiles
// blockIdx.x + t * gridDim.x (gridDim.x is a multiple of 8: a workgroup stays on its XCD's contiguous range of tile ids, and
// the 32 workgroups of an XCD work on the same 32 neighbouring tiles as 32 non-persistent workgroups would).  The tiles of a
// workgroup are ONE continuous K stream: in the last two stages of a tile the "stage kt+2 / kt+1" DMA slots carry the NEXT
// tile's A(0) / W(0) into the ring half the current tile has finished with, waited for by the same counted vmcnt as any other
// stage -- the next tile's first fragments are in LDS before the current epilogue starts.  The epilogue works in the OTHER ring
// half (8-KiB wave slices laid exactly where the wave's own stage-1 pieces land, mk_gemm_common.hpp), so a wave that has
// drained issues its A(1) pieces at once; the row parameters / row shifts of the folded LayerNorm are requested there too and
// turned into LDS entries behind the first stage's wait.  What a tile boundary still costs is the epilogue itself; gone are
// the workgroup launch, the argument loads, the address set-up and the L2 / HBM round trip to the first fragments (2-3 us per
// ~35-us tile, LABNOTES R4.12).  Tile order, summation order and every epilogue are the non-persistent kernel's: bit-identical.
//
// SP (round 6): SPLIT operands staged once (GemmParams; mk_conv3x3_split / mk_gemm_grouped_split).  A stage is 32 contraction
// columns; its 128-byte LDS rows hold [32 hi | 32 lo] of an operand row, so the K = 32 sub-step h = 0 reads HI fragments and
// h = 1 LO fragments, and a stage is THREE (L, C) slot pairs instead of two:
//     L0: W_hi, A_hi fragments (12 reads)     C0: W_hi . A_hi
//     L1: W_lo fragments (4 reads) + W DMA    C1: W_lo . A_hi
//     L2: A_lo fragments (8 reads)            C2: W_hi . A_lo  + A DMA of stage kt+2 between the MFMAs
// -- 96 MFMAs per wave on the 8 DMA pieces and 24 fragment reads that 64 MFMAs of a plain stage (or of one of the three
// sweeps this replaces) take.  Ring hand-over is the plain kernel's with "last reader" = L2 for A and L1 for W: row g's
// A(kt+2) pieces go out in its C2(kt) slot, one barrier behind its L2(kt); the same counted waits at the end of a stage.
// SP2 (with SP): the activations' LO plane is identically zero -- the first conv of a head stack behind an fp16 encoder, whose
// features ARE fp16 values in the reference (mickey_extractor.py:49-52: forward_features(x.to(amp_dtype)) ... .float()) -- so the
// W_hi . A_lo product is not computed: TWO MFMA sets per stage in the plain kernel's four slots (the second load slot brings the
// W_lo fragments only, the A_hi fragments stay).  The lo half of an A row in LDS is staged (from wherever the caller's lo offset
// points: the hi plane) and never read.
template <typename T, int AMODE, int KIND, bool PERSIST = false, bool SP = false, bool SP2 = false>
__global__ __launch_bounds__(512, 2) void gemm_pp64_kernel(GemmParams p, int band_m) {
  static_assert(!SP2 || SP, "SP2 is a form of the split-operand kernel");
  static_assert(!PERSIST || AMODE == A_DENSE, "the persistent tile loop is built for dense operands");
  static_assert(!SP || (KIND == 0 && !PERSIST && sizeof(T) == 2), "split operands: plain epilogues, one tile per workgroup");
  constexpr int BKA = SP ? BK / 2 : BK;   // contraction columns of A per stage
  using V8 = typename Lp<T>::V8;
  constexpr int WMF = 8, BM = 256, BN = 256;
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;   // 64 KiB per stage
  constexpr int LN_STAGE_OFF = 2 * STAGE_BYTES + 256 * 8;            // PERSIST: the statistics' staging area (LnRowDma16), 16 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  // PERSIST: everything per-lane (offsets, LDS addresses) is re-derived from an OPAQUE copy of the lane id at the top of every
  // tile, so that nothing per-lane is loop-invariant: hoisted out of the tile loop it would live through the epilogue, which
  // runs at the 256-register limit (hipcc spilled 600-1000 registers, scratch traffic in every stage)
  int lane = tid0 & 63, tid = tid0;
  auto refresh_lane = [&]() {
    if constexpr (PERSIST) {
      // the lane id from the hardware, in a volatile statement: no source register to keep alive, nothing to hoist or merge
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
      tid = wave * 64 + lane;
    }
  };
  const int wm = wave >> 2, wn = wave & 3;
  const int g = blockIdx.y;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int nk = p.K / BK;

  const T* A = (const T*)p.A + (long long)g * p.strideA_g;
  const T* A2 = p.A2 ? (const T*)p.A2 + (long long)g * p.strideA2_g : nullptr;
  const T* W = (const T*)p.W + (long long)g * p.strideW_g;
  int srow = lane >> 3, sp = lane & 7;
  const int ntiles = ntm * ntn;
  auto tile_origin = [&](int id, int& mm, int& nn) {
    int tm, tn;
    pp_tile_coords(xcd_remap(id, ntiles), ntm, ntn, band_m, tm, tn);
    mm = tm * BM;
    nn = tn * BN;
  };
  int bid = blockIdx.x;   // the tile (PERSIST: this workgroup's current tile)
  int m0, n0;
  tile_origin(bid, m0, n0);
  // this wave's 4 A pieces (own half) and 4 W pieces of a stage; piece = 8 rows x 128 B
  // (32-bit element offsets: the launcher routes operands of 2^31 elements or more to the 128x128 kernel)
  unsigned woff[4], aoff[4], aoff2[4];   // aoff2: conv, the same rows of source 2 (other channel count)
  auto set_woff = [&](int nn0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rw = (wave * 4 + j) * 8 + srow;
      int n = nn0 + rw;
      n = n < p.N ? n : p.N - 1;
      woff[j] = ((unsigned)n * (unsigned)p.ldw + swz8(rw, sp) * 8) * (unsigned)sizeof(T);   // bytes (< 2^32: see launch())
    }
  };
  auto set_aoff = [&](int mm0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ra = wm * 128 + (wn * 4 + j) * 8 + srow;
      int m = mm0 + ra;
      m = m < p.M ? m : p.M - 1;
      // SP: source chunks 0..3 of an LDS row come from the HI plane, 4..7 (the same 32 columns) from the LO plane
      const int c = swz8(ra, sp);
      const unsigned cc = SP ? (unsigned)(c & 3) : (unsigned)c;
      const bool lo = SP && (c >> 2) != 0;   // (selects, not an indexed load: a per-lane index into the kernel arguments makes them vector loads)
      const unsigned pb1 = !SP ? 0u : (lo ? p.pl1[1] : p.pl1[0]), pb2 = !SP ? 0u : (lo ? p.pl2[1] : p.pl2[0]);
      if (AMODE == A_DENSE) {
        aoff[j] = ((unsigned)m * (unsigned)p.lda + cc * 8) * (unsigned)sizeof(T) + pb1;
        aoff2[j] = 0;
      } else {   // the pixel's row in the bordered feature maps (mk_common.hpp): a tap is a wave-uniform shift of it
        const unsigned br = (unsigned)bordered_row(m, p.H, p.Wd);
        aoff[j] = (br * (unsigned)p.C1 + cc * 8) * (unsigned)sizeof(T) + pb1;
        aoff2[j] = (br * (unsigned)p.C2 + cc * 8) * (unsigned)sizeof(T) + pb2;
      }
    }
  };
  set_woff(n0);
  set_aoff(m0);
  auto dma_w1 = [&](int s, int j) {
    // wave-uniform base (SGPRs) + the lane's constant 32-bit byte offset: no VALU instruction per piece
    glds16_sv(W + s * BK, woff[j], smem + (s & 1) * STAGE_BYTES + A_BYTES + (wave * 4 + j) * 1024);
  };
  // conv: A stages are issued in K order (0, 1, 2, ...), so where the NEXT stage reads -- source, 3x3 tap, first channel,
  // folded into one wave-uniform base pointer -- is running scalar state, advanced once per stage.  K = 9 taps x C1 channels of source 1, then C2 channels of source 2
  // (the 1x1 shortcut) at the pixel itself.
  const int ntap = A2 ? 10 : 9;   // K segments: nine taps (+ the shortcut source)
  const T* cbase = AMODE == A_DENSE ? A : A - (long long)(p.Wd + 2) * p.C1;   // tap (-1, -1), channel 0
  int cleft = p.C1, ctap = 0;
  auto conv_advance = [&]() {
    cleft -= BKA;
    const bool wrap = cleft == 0;
    ctap += wrap ? 1 : 0;
    const int ty = (ctap * 11) >> 5, tx = ctap - 3 * ty;   // ctap / 3, ctap % 3 for 0 <= ctap < 9
    const T* tapbase = ctap < 9 ? A + (long long)((ty - 1) * (p.Wd + 1) + tx - 1) * p.C1 : A2;
    cbase = wrap ? tapbase : cbase + BKA;
    cleft = wrap ? (ctap < 9 ? p.C1 : p.C2) : cleft;
  };
  auto dma_a1 = [&](int s, int j) {
    char* dst = smem + (s & 1) * STAGE_BYTES + (wm * 16 + wn * 4 + j) * 1024;
    if (AMODE == A_DENSE) {
      glds16_sv(A + s * BKA, aoff[j], dst);
    } else {
      glds16_sv(cbase, ctap < 9 ? aoff[j] : aoff2[j], dst);   // the caller advances behind the 4th piece
    }
  };
  auto dma_w = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_w1(s, j);
  };
  auto dma_a = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a1(s, j);
    if (AMODE != A_DENSE) conv_advance();
  };

  int fr = lane & 15, fg = lane >> 4;
  auto derive_lane = [&]() {
    srow = lane >> 3;
    sp = lane & 7;
    fr = lane & 15;
    fg = lane >> 4;
  };
  auto load_w = [&](V8* wf, int par, int h) {
    const char* sW = smem + par * STAGE_BYTES + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rw = wn * 64 + i * 16 + fr;
      wf[i] = *(const V8*)(sW + rw * 128 + swz8(rw, h * 4 + fg) * 16);
    }
  };
  auto load_x = [&](V8* xf, int par, int h) {
    const char* sA = smem + par * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < WMF; ++i) {
      const int rx = wm * 128 + i * 16 + fr;
      xf[i] = *(const V8*)(sA + rx * 128 + swz8(rx, h * 4 + fg) * 16);
    }
  };
  auto load_frags = [&](V8* wf, V8* xf, int par, int h) {
    load_w(wf, par, h);
    load_x(xf, par, h);
  };
  f32x4 acc[WMF][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  if constexpr (!PERSIST) zero_acc();
  // 32 MFMAs of a C slot; DMA: the 4 A pieces of stage s.  A piece is one SALU-addressed instruction (dense rows, and
  // conv rows alike since the feature maps are bordered: round 2's conv pieces were ~15 VALU instructions of per-lane tap /
  // border arithmetic each and had to go out in the wave-row's non-MFMA slot) and goes out BETWEEN the MFMAs (behind MFMA
  // 3, 11, 19, 27), pinned.
  // FIRST (PERSIST: the first slot of a tile): C = 0 as the MFMA's inline constant -- the accumulators are DEFINED here and dead
  // behind the epilogue, not carried around the tile loop (zeroed at the loop's end they were 128 loop-carried registers next
  // to 128 fresh results: hipcc spilled a tile's worth of accumulators per stage)
  auto mfma32 = [&](const V8* wf, const V8* xf, auto dma, int s, auto first) {
    constexpr bool DMA = decltype(dma)::value;
    constexpr bool FIRST = decltype(first)::value;
    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int mi = q >> 2, ni = q & 3;
      acc[mi][ni] = Lp<T>::mma16(wf[ni], xf[mi], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mi][ni]);
      if (DMA && (q & 7) == 3) {
        __builtin_amdgcn_sched_barrier(0);
        dma_a1(s, q >> 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    if (DMA && AMODE != A_DENSE) conv_advance();   // behind the slot's MFMAs: hipcc turns its selects into a branch
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  V8 wf[4], xf[WMF];
  V8 wfl[SP ? 4 : 1];   // SP: the W_lo fragments (wf = W_hi stays live through the stage)
  // one K = 64 stage of wave-row 0 / 1; NEXT1: stage kt+1 exists, NEXT2: stage kt+2 exists (compile time: the slots stay
  // single basic blocks, so the DMA pieces can be pinned between the MFMAs)
  // sw / sa: the stage whose W / A pieces this stage issues -- kt + 1 / kt + 2, or (PERSIST, last two stages of a tile, woff /
  // aoff already re-pointed) stage 0 of the workgroup's NEXT tile; nk is even there, so the ring half is the same either way
  auto stage0 = [&](int kt, auto next1, auto next2, int sw, int sa, auto first) {
    if constexpr (SP && SP2) {
      bar();   // slot 4kt
      load_frags(wf, xf, kt & 1, 0);
      if constexpr (decltype(next1)::value) dma_w(sw);
      bar();                      // slot 4kt+1
      mfma32(wf, xf, No{}, 0, first);     // W_hi . A_hi
      bar();                      // slot 4kt+2
      load_w(wf, kt & 1, 1);              // W_lo; the A_hi fragments stay
      bar();                      // slot 4kt+3
      mfma32(wf, xf, next2, sa, No{});    // W_lo . A_hi
      if constexpr (decltype(next2)::value)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // stage kt+1 landed; A0(kt+2) may still fly
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    if constexpr (SP) {
      bar();                      // slot 6kt
      load_frags(wf, xf, kt & 1, 0);
      bar();                      // slot 6kt+1
      mfma32(wf, xf, No{}, 0, first);     // W_hi . A_hi
      bar();                      // slot 6kt+2
      load_w(wfl, kt & 1, 1);
      if constexpr (decltype(next1)::value) dma_w(sw);
      bar();                      // slot 6kt+3
      mfma32(wfl, xf, No{}, 0, No{});     // W_lo . A_hi
      bar();                      // slot 6kt+4
      load_x(xf, kt & 1, 1);
      bar();                      // slot 6kt+5
      mfma32(wf, xf, next2, sa, No{});    // W_hi . A_lo
      if constexpr (decltype(next2)::value)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // stage kt+1 landed; A0(kt+2) may still fly
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    bar();   // slot 4kt
    load_frags(wf, xf, kt & 1, 0);
    if constexpr (decltype(next1)::value) dma_w(sw);
    bar();                      // slot 4kt+1
    mfma32(wf, xf, No{}, 0, first);
    bar();                      // slot 4kt+2
    load_frags(wf, xf, kt & 1, 1);
    bar();                      // slot 4kt+3
    mfma32(wf, xf, next2, sa, No{});
    if constexpr (decltype(next2)::value)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // stage kt+1 landed; A0(kt+2) may still fly
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto stage1 = [&](int kt, auto next1, auto next2, int sw, int sa, auto first) {
    if constexpr (SP && SP2) {
      bar();                      // slot 4kt+1
      load_frags(wf, xf, kt & 1, 0);
      if constexpr (decltype(next1)::value) dma_w(sw);
      bar();                      // slot 4kt+2
      mfma32(wf, xf, No{}, 0, first);
      bar();                      // slot 4kt+3
      load_w(wf, kt & 1, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own share of stage kt+1 (A1 and W) landed
      bar();                      // slot 4kt+4
      mfma32(wf, xf, next2, sa, No{});
      return;
    }
    if constexpr (SP) {
      bar();                      // slot 6kt+1
      load_frags(wf, xf, kt & 1, 0);
      bar();                      // slot 6kt+2
      mfma32(wf, xf, No{}, 0, first);
      bar();                      // slot 6kt+3
      load_w(wfl, kt & 1, 1);
      if constexpr (decltype(next1)::value) dma_w(sw);
      bar();                      // slot 6kt+4
      mfma32(wfl, xf, No{}, 0, No{});
      bar();                      // slot 6kt+5
      load_x(xf, kt & 1, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own share of stage kt+1 (A1 and W) landed
      bar();                      // slot 6kt+6
      mfma32(wf, xf, next2, sa, No{});
      return;
    }
    bar();                      // slot 4kt+1
    load_frags(wf, xf, kt & 1, 0);
    if constexpr (decltype(next1)::value) dma_w(sw);
    bar();                      // slot 4kt+2
    mfma32(wf, xf, No{}, 0, first);
    bar();                      // slot 4kt+3
    load_frags(wf, xf, kt & 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own share of stage kt+1 (A1 and W) landed
    bar();                      // slot 4kt+4
    mfma32(wf, xf, next2, sa, No{});
  };

  // prologue: stage 0 (own A half + W share) and the own A half of stage 1 (nk >= 2, see launch()).
  // Folded LayerNorm (consumer): the row statistics of the tile are requested first and turned into row parameters in LDS
  // while the DMA pieces are in flight.
  const bool ln = KIND == 1 && p.ln_stats != nullptr;
#if defined(MK_LN_ABL) && MK_LN_ABL == 1
  const bool ln_fast = false && ln;   // ablation: no prologue work
#define MK_LN_NO_SLOW 1
#else
  const bool ln_fast = ln && p.ln_nslot == 16;
#endif
  LnRowLoads16 lnl;
  if (ln_fast) lnl.issue(p, m0, tid);
  // folded LayerNorm (producer): the row shifts of the tile (row centring, zeros when off) take the same route into LDS
  constexpr bool PRODUCER = AMODE == A_DENSE && (KIND == 2 || KIND == 3);
  ShiftLoad shl;
  if constexpr (PRODUCER) shl.issue(p, m0, tid, 256);
  dma_a(0);
  dma_w(0);
  if constexpr (!PERSIST) dma_a(1);   // (PERSIST: at the top of the tile loop, for every tile alike)
  constexpr int YOUNGER = PERSIST ? 8 : 12;   // DMA pieces younger than the loads above
  const bool publish = ln && p.ln_shift_out != nullptr && n0 == 0;   // first tile column: the row means for the next producer
  if (ln_fast) lnl.template finish<YOUNGER>(p, tid, (float2*)(smem + 2 * STAGE_BYTES), m0, publish);
#ifndef MK_LN_NO_SLOW
  else if (ln) ln_params_to_lds<256, 512>(p, m0, tid, (float2*)(smem + 2 * STAGE_BYTES), publish);
#endif
  if constexpr (PRODUCER) shl.template finish<YOUNGER>(tid, 256, (float*)(smem + 2 * STAGE_BYTES));
  if constexpr (!PERSIST) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  const int bstep = PERSIST ? (int)gridDim.x : 0;
  // (measured and removed: every other workgroup of an XCD starting 10-60 k cycles late, so that the CUs' epilogue bursts do not
  // coincide -- +-1 % on all four encoder shapes, profiles/r05f_gemm_persistent_stagger.txt)
  for (;;) {
    const int nbid = bid + bstep;
    const bool has_next = PERSIST && nbid < ntiles;   // workgroup-uniform
    // the tile whose stage 0 is prefetched in the last two stages: the next one -- or, behind the last tile, the current one
    // again (64 KiB of valid operands nobody reads, once per workgroup: keeps ONE code path through the tail; with a branch
    // around it hipcc renamed all 128 accumulators at the join and spilled a tile's worth of them per stage)
    int m1 = m0, n1 = n0;
    if (PERSIST && has_next) tile_origin(nbid, m1, n1);
    if constexpr (PERSIST) {
      // every tile alike (ONE definition of the per-lane values inside the loop: two -- prologue and loop -- met in a PHI that
      // hipcc spilled and re-loaded here, behind the DMA pieces, with a vmcnt wait for all of them)
      refresh_lane();
      derive_lane();
      set_aoff(m0);
      set_woff(n0);
      dma_a(1);
      // the workgroup's first tile: its stage 0 went out in the prologue (later tiles: waited for inside the previous K loop;
      // a wait here would be a wait for the previous epilogue's stores)
      if (bid == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    // PERSIST, tiles after the first: the row parameters / shifts requested behind the previous epilogue become LDS entries
    // behind the first stage's wait (every older memory operation of the wave has completed there; the other waves left
    // their epilogues -- the last readers of the previous entries -- before this tile's first barrier)
    bool ln_pending = false;
    auto first_stage_done = [&]() {    // (two of the wave's four statistics pieces were requested behind the epilogue ...
      if constexpr (PERSIST) {
        if (ln_pending) {
          const bool publish1 = ln && p.ln_shift_out != nullptr && n0 == 0;
          if (ln_fast) {
            LnRowDma16::finish<0>(p, wave, lane, (float2*)(smem + 2 * STAGE_BYTES), smem + LN_STAGE_OFF, m0, publish1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging area has been read
            LnRowDma16::issue<2>(p, m0, wave, lane, smem + LN_STAGE_OFF);
          }
          // (consumers of other widths than 1024 -- ViT-S -- are not launched persistent: launch_k)
          if constexpr (PRODUCER) shl.template finish<-1>(tid, 256, (float*)(smem + 2 * STAGE_BYTES));
        }
      }
    };
    auto second_stage_done = [&]() {   // ... the other two behind the first stage)
      if constexpr (PERSIST) {
        if (ln_pending && ln_fast)
          LnRowDma16::finish<2>(p, wave, lane, (float2*)(smem + 2 * STAGE_BYTES), smem + LN_STAGE_OFF, m0,
                                ln && p.ln_shift_out != nullptr && n0 == 0);
      }
    };
    ln_pending = PERSIST && bid != (int)blockIdx.x;
    int kt0 = 0;
    if (wm == 0) {
      if constexpr (PERSIST) {   // (nk >= 4: launch_k)
        stage0(0, Yes{}, Yes{}, 1, 2, Yes{});
        first_stage_done();
        stage0(1, Yes{}, Yes{}, 2, 3, No{});
        second_stage_done();
        kt0 = 2;
      }
      for (int kt = kt0; kt < nk - 2; ++kt) stage0(kt, Yes{}, Yes{}, kt + 1, kt + 2, No{});
      if constexpr (PERSIST) {
        set_aoff(m1);
        stage0(nk - 2, Yes{}, Yes{}, nk - 1, 0, No{});
        set_woff(n1);
        stage0(nk - 1, Yes{}, No{}, 0, 0, No{});
      } else {
        stage0(nk - 2, Yes{}, No{}, nk - 1, 0, No{});
        stage0(nk - 1, No{}, No{}, 0, 0, No{});
      }
      bar();   // row 1's last fragment reads are done, the LDS ring is free
    } else {
      bar();   // slot 0: this wave-row idles
      if constexpr (PERSIST) {
        stage1(0, Yes{}, Yes{}, 1, 2, Yes{});
        first_stage_done();
        stage1(1, Yes{}, Yes{}, 2, 3, No{});
        second_stage_done();
        kt0 = 2;
      }
      for (int kt = kt0; kt < nk - 2; ++kt) stage1(kt, Yes{}, Yes{}, kt + 1, kt + 2, No{});
      if constexpr (PERSIST) {
        set_aoff(m1);
        stage1(nk - 2, Yes{}, Yes{}, nk - 1, 0, No{});
        set_woff(n1);
        stage1(nk - 1, Yes{}, No{}, 0, 0, No{});
      } else {
        stage1(nk - 2, Yes{}, No{}, nk - 1, 0, No{});
        stage1(nk - 1, No{}, No{}, 0, 0, No{});
      }
    }
    char* wl0 = smem + STAGE_BYTES + wave * 4096;   // the wave's own A pieces / W pieces of stage 1 (mk_gemm_common.hpp)
    refresh_lane();   // (neither the K loop's per-lane values nor the epilogue's are to be shared / hoisted)
    epilogue_lds<T, KIND, AMODE == A_CONV3>(p, acc, wl0, wl0 + A_BYTES, m0, n0, wm, wn, lane, g, (const float2*)(smem + 2 * STAGE_BYTES));
    if (!(PERSIST && has_next)) break;
    // the next tile: its stage 0 is in LDS (waited for in the last stage above); the own A half of stage 1 and the row
    // parameters go out now, the accumulators restart
    bid = nbid;
    m0 = m1;
    n0 = n1;
    asm volatile("" : "+s"(m0), "+s"(n0));   // not the values the tail above derived aoff / woff from: those die with the tail
    refresh_lane();
    if (ln_fast) LnRowDma16::issue<0>(p, m0, wave, lane, smem + LN_STAGE_OFF);
    if constexpr (PRODUCER) shl.issue(p, m0, tid, 256);
  }
}

template <typename T, int AMODE, int KIND, bool PERSIST, bool SP = false, bool SP2 = false>
int launch_k2(const GemmParams& p, int groups, hipStream_t st, int band_m, int grid) {
  constexpr int LDS = 2 * 512 * 128 + 256 * 8 + (PERSIST && KIND == 1 ? 16384 : 0);   // two stages + the folded LayerNorm's row parameters (+ the statistics' staging area)
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_pp64_kernel<T, AMODE, KIND, PERSIST, SP, SP2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      mk_set_error("gemm: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
      return MK_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_pp64_kernel<T, AMODE, KIND, PERSIST, SP, SP2>), dim3(grid, groups, 1), dim3(512), LDS, st, p, band_m);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

template <typename T, int AMODE, int KIND>
int launch_k(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  // tile order (pp_tile_coords): outputs at most 4 tiles wide (proj, fc2: N = 1024) are walked m-major with n fastest -- the
  // 32 tiles in flight are the same 8 x 4 block as in a band, but neighbouring CUs share the (large) A panel: -19 % L2-miss
  // traffic and +2.6 % on fc2 (profiles/r04j_gemm_order.txt); wider outputs keep bands of 8 m-tiles (n-groups there cut the
  // traffic as much and cost 1-6 % of time)
  if (band_m == 0) band_m = (AMODE == A_DENSE && ntn <= 4) ? -4 : 8;
  if constexpr (KIND == 0 && sizeof(T) == 2 && std::is_same<T, _Float16>::value) {
    if (p.npass == 2) {   // split operands, activations' lo plane identically zero (conv only: mk_conv3x3_split with in1_lo == NULL)
      if constexpr (AMODE == A_CONV3) return launch_k2<T, AMODE, 0, false, true, true>(p, groups, st, band_m, ntm * ntn);
    }
    if (p.npass > 1) return launch_k2<T, AMODE, 0, false, true>(p, groups, st, band_m, ntm * ntn);   // split operands
  }
  if constexpr (AMODE == A_DENSE && KIND != 0) {   // (KIND 0: the plain epilogues -- head linears, patch embed of the unfolded
                                                    // mode: not the encoder's hot launches; its eight inlined variants spill in the loop)
    // persistent tile loop: one workgroup per CU (a multiple of 8: a workgroup keeps its XCD) walking a continuous K stream over
    // its tiles; needs an even number (>= 4) of K stages, plain operands, and more tiles than CUs to be worth anything
    const int nk = p.K / BK, grid = num_cus() & ~7;
    const bool ok = groups == 1 && p.npass <= 1 && (nk & 1) == 0 && nk >= 4 && grid >= 8 && ntm * ntn > grid &&
                    !(KIND == 1 && p.ln_stats && p.ln_nslot != 16);   // the consumer's statistics staging is built for 16 slots
    if (g_pp64_persist != 1 && ok && !(g_pp64_persist == 2 && KIND == 1)) return launch_k2<T, AMODE, KIND, true>(p, groups, st, band_m, grid);
  }
  return launch_k2<T, AMODE, KIND, false>(p, groups, st, band_m, ntm * ntn);
}

template <typename T>
int launch_dense(const GemmParams& p, int groups, hipStream_t st, int band_m) {
  if (p.xh && p.epi == MK_EPI_LS_RESIDUAL && p.out_f32) return launch_k<T, A_DENSE, 3>(p, groups, st, band_m);
  if (p.xh) return launch_k<T, A_DENSE, 2>(p, groups, st, band_m);
  if (p.ln_stats) return launch_k<T, A_DENSE, 1>(p, groups, st, band_m);
  return launch_k<T, A_DENSE, 0>(p, groups, st, band_m);
}

}  // namespace

int launch_pp64(const GemmParams& p, int groups, int dtype, int amode, hipStream_t st, int band_m) {
  if (amode == A_DENSE) return dtype == MK_BF16 ? launch_dense<__bf16>(p, groups, st, band_m) : launch_dense<_Float16>(p, groups, st, band_m);
  return dtype == MK_BF16 ? launch_k<__bf16, A_CONV3, 0>(p, groups, st, band_m) : launch_k<_Float16, A_CONV3, 0>(p, groups, st, band_m);
}

}  // namespace gemm
}  // namespace mk
