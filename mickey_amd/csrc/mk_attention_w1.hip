// mickey_amd -- flash attention forward, head_dim 64, ONE WAVE PER SIMD (mk_attn_set_mode 7; reference
// DINO_modules/layers/attention.py:53-59).  Same maths and fragment conventions as mk_attention.hip (S^T = K.Q^T so that a
// lane owns 32 scores of one query; O^T = V^T.P^T with the key-permuted V^T image; exp2 with the scale folded into q;
// running maximum folded into the MFMA accumulator init and re-based only when a tile exceeds it by 2^8; row sums as
// ones.P^T on the matrix pipe), re-built around what bounds that kernel: at two waves per SIMD the softmax VALU work and
// the MFMAs of a wave run one after the other (~1400 cycles per 32-query x 64-key wave-tile against 512 cycles of matrix
// pipe).  Here:
//  * a workgroup is 4 waves, one per SIMD, 64 queries each (two 32-query blocks qb = 0, 1); a wave owns the whole 512-entry
//    register file: the output accumulators O^T (64), the row sums (32) and DOUBLE-BUFFERED K and V^T fragments of a KV
//    tile (64 + 64) live in the accumulator half (a[0:223], written only by inline asm: ds_read_b128 straight into AGPRs,
//    MFMAs with AGPR A / C / D operands); the architectural VGPRs hold S' (64), P (32), -m (32), Q (32);
//  * the two query blocks run HALF A TILE APART, which makes every slot of 20 MFMAs independent of the VALU work beside it:
//        slot A(t):  matrix pipe  S'1(t) = K(t).Q1^T - m1 ,  O1 += V(t-1).P1(t-1) ,  l1 += 1.P1(t-1)
//                    VALU         P0(t) = exp2(S'0(t))  (32 exp + 16 cvt) ,  then max S'1(t)
//        slot B(t):  matrix pipe  S'0(t+1) = K(t+1).Q0^T - m0 ,  O0 += V(t).P0(t) ,  l0 += 1.P0(t)
//                    VALU         P1(t) = exp2(S'1(t)) ,  then max S'0(t+1)
//    written out as 20 gaps of {1 MFMA, <= 5 single-issue fillers} pinned by sched_barrier (one wave per SIMD hides about
//    5 fillers per 32-cycle MFMA, v_exp_f32 counting as 2: MI355X_MICROARCH.md); no MFMA waits for a VALU result of its own
//    slot and vice versa, so S' and P need ONE buffer per query block;
//  * K / V^T tiles travel HBM -> LDS by SGPR-addressed LDS-DMA three tiles ahead into a 3-deep ring (48 KiB), one counted
//    vmcnt + one barrier per tile; fragments are read from LDS one slot before use (K(t+1) in slot A(t), V(t+1) in B(t)).
// The launcher hands problems with fewer than 4 KV tiles back to the caller (mk_attention.hip, 64-query kernel).
#include <type_traits>
#include <utility>

#include "mk_common.hpp"

namespace {
using namespace mk;

constexpr int TILE_BYTES = 64 * 64 * 2;    // one K or V^T tile: 8 KiB
constexpr int RING = 3;
constexpr int V_RING_OFF = RING * TILE_BYTES;
constexpr float REBASE_THR = 8.0f;

// accumulator-file map (asm-owned)
constexpr int A_O = 0;      // O^T(qb, dt): 16 registers at A_O + (qb * 2 + dt) * 16
constexpr int A_L = 64;     // row sums l(qb): 16 registers at A_L + qb * 16
constexpr int A_K = 96;     // K fragments, two buffers of 8 x 4: A_K + buf * 32 + (kb * 4 + ks) * 4
constexpr int A_V = 160;    // V^T fragments, two buffers: A_V + buf * 32 + (dt * 4 + s4) * 4
constexpr int A_END = 224;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define MK_A16(b) "a" #b
// every accumulator register this kernel names: listed once as clobbers so that the kernel descriptor allocates them
#define MK_ACC_CLOBBERS                                                                                                      \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18",   \
  "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",      \
  "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52",      \
  "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69",      \
  "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86",      \
  "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103",  \
  "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118",     \
  "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133",     \
  "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148",     \
  "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163",     \
  "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178",     \
  "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193",     \
  "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208",     \
  "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223"

// ---- the instructions the compiler must not see inside (operands in the accumulator file, hand-counted waits) ----------
// S'(first k-step) = K_frag(AGPR) . Q_frag + (-m):  D, C in VGPRs (the softmax reads S' with VALU instructions)
template <bool BF, int KA>
__device__ __forceinline__ void mfma_qk_first(f32x16& s, const u32x4& q, const f32x16& negm) {
  if constexpr (BF)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c3:%c4], %1, %2" : "=&v"(s) : "v"(q), "v"(negm), "n"(KA), "n"(KA + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c3:%c4], %1, %2" : "=&v"(s) : "v"(q), "v"(negm), "n"(KA), "n"(KA + 3));
}
template <bool BF, int KA>
__device__ __forceinline__ void mfma_qk_acc(f32x16& s, const u32x4& q) {
  if constexpr (BF)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], %1, %0" : "+v"(s) : "v"(q), "n"(KA), "n"(KA + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, %0" : "+v"(s) : "v"(q), "n"(KA), "n"(KA + 3));
}
// O^T(AGPR) += V_frag(AGPR) . P_frag
template <bool BF, int OA, int VA>
__device__ __forceinline__ void mfma_pv(const u32x4& p) {
  if constexpr (BF)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(p), "n"(OA), "n"(OA + 15), "n"(VA), "n"(VA + 3) : MK_ACC_CLOBBERS);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(p), "n"(OA), "n"(OA + 15), "n"(VA), "n"(VA + 3) : MK_ACC_CLOBBERS);
}
// l(AGPR) += ones . P_frag  (every row of the result is the row sum of the lane's query)
template <bool BF, int LA>
__device__ __forceinline__ void mfma_ls(const u32x4& ones, const u32x4& p) {
  if constexpr (BF)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(ones), "v"(p), "n"(LA), "n"(LA + 15) : MK_ACC_CLOBBERS);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(ones), "v"(p), "n"(LA), "n"(LA + 15) : MK_ACC_CLOBBERS);
}
// LDS -> accumulator file, 16 B per lane (counted by the kernel's own lgkmcnt waits)
template <int AA, int OFF>
__device__ __forceinline__ void lds_to_acc(unsigned addr) {
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "n"(AA), "n"(AA + 3), "n"(OFF) : "memory", MK_ACC_CLOBBERS);
}
template <int AA>
__device__ __forceinline__ void acc_zero() {
  asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"n"(AA) : MK_ACC_CLOBBERS);
}
template <int AA>
__device__ __forceinline__ float acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "n"(AA));
  return v;
}
template <int AA>
__device__ __forceinline__ void acc_scale(float alpha) {
  float tmp;
  asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[%c2], %0" : "=&v"(tmp) : "v"(alpha), "n"(AA) : MK_ACC_CLOBBERS);
}
// idle states in front of a VALU / accvgpr read of something the last MFMAs wrote (the compiler pads nothing for asm)
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }

template <typename T>
__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, T2{(T)a, (T)b});
}

template <typename T>
__global__ __launch_bounds__(256, 1) void attn_w1_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
                                                         T* __restrict__ out, int ldo, int heads, int ntok, int ntok_pad) {
  constexpr bool BF = std::is_same<T, __bf16>::value;
  using V4 = typename Lp<T>::V4;
  __shared__ __attribute__((aligned(16))) char smem[2 * RING * TILE_BYTES];   // K ring | V^T ring
  asm volatile("" ::: MK_ACC_CLOBBERS);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware decode as in mk_attention.hip: the query blocks of one (image, head) run back to back on one XCD
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int rr = xcd_remap(lin, gridDim.x * gridDim.y * gridDim.z);
  const int hg = rr / (int)gridDim.x;
  const int qblk = rr - hg * (int)gridDim.x, head = hg % (int)gridDim.y, img = hg / (int)gridDim.y;
  const long long hb = (long long)img * heads + head;
  const T* Qh = q + hb * ntok_pad * 64;
  const T* Kh = k + hb * ntok_pad * 64;
  const T* Vh = vt + hb * 64 * ntok_pad;
  const int q0 = qblk * 256 + wave * 64;
  const int j = lane & 31, hi = lane >> 5;
  const int nkt = (ntok + 63) >> 6;

  // ---- LDS-DMA: 8 one-KiB pieces per operand tile, 2 per wave; per-lane byte offsets are constant, the base is uniform
  const int srow = lane >> 3, sp = lane & 7;
  unsigned voffK[2], voffV[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = (wave * 2 + t) * 8 + srow;
    voffK[t] = (unsigned)((r * 64 + swz8(r, sp) * 8) * 2);
    voffV[t] = (unsigned)((r * ntok_pad + swz8(r, sp) * 8) * 2);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma_k = [&](int tile, int ring_off, int t) {
    glds16_sv(Kh + (long long)tile * 4096, voffK[t], smem + ring_off + (wave * 2 + t) * 1024);
  };
  auto dma_v = [&](int tile, int ring_off, int t) {
    glds16_sv(Vh + tile * 64, voffV[t], smem + V_RING_OFF + ring_off + (wave * 2 + t) * 1024);
  };
  // fragment reads: row j (+32 for the second 32-row block: immediate), logical chunk (ks * 2 + hi) ^ ((j >> 1) & 7)
  unsigned lo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) lo[ks] = lds0 + (unsigned)(j * 128 + (((ks * 2 + hi) ^ ((j >> 1) & 7)) << 4));
  unsigned ad[4];   // lo + ring slot of the tile whose fragments are read next
  int rd_off = 0, wr_off = 2 * TILE_BYTES;   // byte offsets of the ring slots of tiles t+1 / t+3, advanced at the top of slot A(t)

  // ---- registers
  u32x4 qf[2][4];   // Q fragments (B operand of S^T = K.Q^T)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int qrow = q0 + qb * 32 + j;
    qrow = qrow < ntok_pad ? qrow : ntok_pad - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const u32x4*)(Qh + (long long)qrow * 64 + ks * 16 + hi * 8);
  }
  // the Q loads are the only loads hipcc knows about: make it wait for them HERE (left alone it waits at their first use,
  // inside the tile loop, with a vmcnt(0) that drains the LDS-DMA ring every slot)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]),
               "+v"(qf[1][2]), "+v"(qf[1][3]) :: "memory");
  u32x4 ones;
#pragma unroll
  for (int e = 0; e < 4; ++e) ones[e] = pack2<T>(1.0f, 1.0f);
  asm volatile("" : "+v"(ones));   // a VGPR operand for good: re-materialised from constants it would be written right in front
                                   // of the MFMA that reads it (no wait states are inserted in front of an asm statement)
  f32x16 S0[2], S1[2];       // S' of query block 0 / 1: [key half kb]
  f32x16 negm0, negm1;       // -m of the lane's query, 16 copies (the C operand of the first k-step)
  u32x4 P0[4], P1[4];        // P of query block 0 / 1, packed 16-bit pairs: [k-step s4]
  float m0 = 0.f, m1 = 0.f, mx0 = 0.f, mx1 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm0[i] = negm1[i] = 0.f;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) P0[s4] = P1[s4] = u32x4{0u, 0u, 0u, 0u};   // tile "-1" of query block 1 contributes P = 0
  static_for<96>([&](auto i) { acc_zero<A_O + decltype(i)::value>(); });                    // O^T, l = 0
  static_for<32>([&](auto i) { acc_zero<A_V + 32 + decltype(i)::value>(); });               // V^T buffer 1 = 0 (times P = 0)

  // ---- the re-base of a query block (rare after the first tile): S' -= shift, m += shift, O and l *= 2^-shift
  auto rebase = [&](f32x16 (&S)[2], f32x16& negm, float& m, float mx, bool first, auto qb_tag) {
    constexpr int QB = decltype(qb_tag)::value;
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                 // lanes j and j + 32 hold the two key halves of one query
    const float shift = first ? mx : fmaxf(mx, 0.f);        // never lower m after the first tile
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      S[0][i] -= shift;
      S[1][i] -= shift;
    }
    m += shift;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = -m;
    if (!first) {
      const float alpha = __builtin_amdgcn_exp2f(-shift);
      mfma_drain();   // the last MFMAs of the previous slot wrote O / l of this query block
      static_for<32>([&](auto i) { acc_scale<A_O + QB * 32 + decltype(i)::value>(alpha); });
      static_for<16>([&](auto i) { acc_scale<A_L + QB * 16 + decltype(i)::value>(alpha); });
      asm volatile("s_nop 4" ::: "memory");
    }
  };
  // key mask of the ragged last tile (applied before the tile maximum): keys >= ntok get -inf
  auto mask_keys = [&](f32x16 (&S)[2], int tile) {
    int lim = ntok - tile * 64 - 4 * hi;   // first masked key of the tile, in this lane's register numbering
    asm volatile("" : "+v"(lim));          // opaque: otherwise the 32 comparisons are hoisted out of the tile loop as 32 live SGPR pairs
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + (r & 3) + 8 * (r >> 2) >= lim) S[kb][r] = -1e30f;
  };
  auto tile_max_part = [&](const f32x16 (&S)[2], float (&t8)[8], int r0, int r1) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r >= r0 && r < r1) t8[r] = fmaxf(fmaxf(S[0][r], S[0][r + 8]), fmaxf(S[1][r], S[1][r + 8]));
  };
  auto tile_max_fin = [&](const float (&t8)[8]) {
    float mx = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), t8[2]), fmaxf(fmaxf(t8[3], t8[4]), t8[5]));
    return fmaxf(fmaxf(mx, t8[6]), t8[7]);
  };

  // One slot of 20 MFMAs.  SLOT_A: the exponentials are query block 0's, the MFMAs query block 1's (see the header).
  // PAR = parity of the tile index t: K(t), V(t) sit in fragment buffer PAR, K(t+1) / V(t+1) go to buffer PAR ^ 1.
  auto slot = [&](int t, bool first, auto slot_a_tag, auto par_tag) {
    constexpr bool SLOT_A = decltype(slot_a_tag)::value;
    constexpr int PAR = decltype(par_tag)::value;
    constexpr int QBM = SLOT_A ? 1 : 0;                      // query block of this slot's MFMAs
    constexpr int KBUF = A_K + 32 * (SLOT_A ? PAR : PAR ^ 1);         // K(t) for S'1(t)   |  K(t+1) for S'0(t+1)
    constexpr int VBUF = A_V + 32 * (SLOT_A ? PAR ^ 1 : PAR);         // V(t-1) for O1     |  V(t) for O0
    constexpr int RBUF = (SLOT_A ? A_K : A_V) + 32 * (PAR ^ 1);       // fragments read in this slot: K(t+1) | V(t+1)
    constexpr int ROFF = SLOT_A ? 0 : V_RING_OFF;
    f32x16(&SX)[2] = SLOT_A ? S0 : S1;     // exponentiated in this slot
    f32x16(&SM)[2] = SLOT_A ? S1 : S0;     // produced in this slot
    u32x4(&PX)[4] = SLOT_A ? P0 : P1;      // produced in this slot
    u32x4(&PM)[4] = SLOT_A ? P1 : P0;      // consumed by this slot's MFMAs
    f32x16& negmM = SLOT_A ? negm1 : negm0;

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments read during the previous slot have landed
    if constexpr (SLOT_A) {
      // tile t+1 has landed (this wave's pieces; the barrier makes it everybody's), the 4 pieces issued during tile t-1
      // may still be in flight.  (Every tile issues 4 pieces: past the end the last tile is fetched again into a ring slot
      // nobody reads any more, which keeps this count -- and the slot a single basic block -- free of tail cases.)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      rd_off = rd_off + TILE_BYTES == RING * TILE_BYTES ? 0 : rd_off + TILE_BYTES;   // ring slot of tile t+1
      wr_off = wr_off + TILE_BYTES == RING * TILE_BYTES ? 0 : wr_off + TILE_BYTES;   // ring slot of tile t+3 (= that of tile t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ad[ks] = lo[ks] + (unsigned)rd_off;
    }
    // head: the (rare) re-base of the query block whose S' is exponentiated in this slot
    {
      float& mxX = SLOT_A ? mx0 : mx1;
      if (first || __any(mxX > REBASE_THR)) {
        if constexpr (SLOT_A) rebase(S0, negm0, m0, mxX, first, std::integral_constant<int, 0>{});
        else rebase(S1, negm1, m1, mxX, first, std::integral_constant<int, 1>{});
      }
    }
    const int tile_w = t + 3 < nkt ? t + 3 : nkt - 1;   // the tile whose pieces go out in this tile's two slots
    float t8[8];
    __builtin_amdgcn_sched_barrier(0);
    static_for<20>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      // ---- the MFMA of this gap
      if constexpr (g < 8) {
        constexpr int kb = g & 1, ks = g >> 1;
        if constexpr (ks == 0) mfma_qk_first<BF, KBUF + (kb * 4 + ks) * 4>(SM[kb], qf[QBM][ks], negmM);
        else mfma_qk_acc<BF, KBUF + (kb * 4 + ks) * 4>(SM[kb], qf[QBM][ks]);
      } else {
        constexpr int s4 = (g - 8) / 3, c = (g - 8) % 3;
        if constexpr (c < 2) mfma_pv<BF, A_O + (QBM * 2 + c) * 16, VBUF + (c * 4 + s4) * 4>(PM[s4]);
        else mfma_ls<BF, A_L + QBM * 16>(ones, PM[s4]);
      }
      // ---- fillers
      if constexpr (g < 16) {   // P dword g = (exp2 S'[2g], exp2 S'[2g + 1]) in the flattened order P.V consumes
        constexpr int f = 2 * g;
        const float x0 = __builtin_amdgcn_exp2f(SX[f >> 4][f & 15]);
        const float x1 = __builtin_amdgcn_exp2f(SX[f >> 4][(f & 15) + 1]);
        unsigned bits = pack2<T>(x0, x1);
        asm volatile("" : "+v"(bits));   // pins the pair into this gap (left alone all 16 conversions sink to P's first use)
        PX[g >> 2][g & 3] = bits;
      }
      if constexpr (g >= 8 && g < 16) {   // one fragment of the next tile: LDS -> accumulator file
        constexpr int fr = g - 8;          // (kb | dt) = fr >> 2, (ks | s4) = fr & 3
        lds_to_acc<RBUF + fr * 4, ROFF + (fr >> 2) * 4096>(ad[fr & 3]);
      }
      if constexpr (g == 1 || g == 3) {
        if constexpr (SLOT_A) dma_k(tile_w, wr_off, g >> 1);
        else dma_v(tile_w, wr_off, g >> 1);
      }
      if constexpr (g == 15) {   // the ragged last tile: mask S' of the tile just produced before its maximum is taken
        const int tile_m = SLOT_A ? t : t + 1;
        if (tile_m == nkt - 1 && (ntok & 63)) mask_keys(SM, tile_m);
      }
      if constexpr (g == 16) tile_max_part(SM, t8, 0, 3);
      if constexpr (g == 17) tile_max_part(SM, t8, 3, 6);
      if constexpr (g == 18) tile_max_part(SM, t8, 6, 8);
      if constexpr (g == 19) (SLOT_A ? mx1 : mx0) = tile_max_fin(t8);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue: tiles 0, 1, 2 on their way; K(0), V(0) fragments; S'0(0)
  for (int tile = 0; tile < 3; ++tile)   // nkt >= 4 (launcher)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      dma_k(tile, tile * TILE_BYTES, t);
      dma_v(tile, tile * TILE_BYTES, t);
    }
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tiles 0 and 1 (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  static_for<8>([&](auto fi) {
    constexpr int fr = decltype(fi)::value;
    lds_to_acc<A_K + fr * 4, (fr >> 2) * 4096>(lo[fr & 3]);
    lds_to_acc<A_V + fr * 4, V_RING_OFF + (fr >> 2) * 4096>(lo[fr & 3]);
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  static_for<8>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    constexpr int kb = g & 1, ks = g >> 1;
    if constexpr (ks == 0) mfma_qk_first<BF, A_K + (kb * 4 + ks) * 4>(S0[kb], qf[0][ks], negm0);
    else mfma_qk_acc<BF, A_K + (kb * 4 + ks) * 4>(S0[kb], qf[0][ks]);
  });
  mfma_drain();
  __builtin_amdgcn_sched_barrier(0);
  {
    float t8[8];
    tile_max_part(S0, t8, 0, 8);
    mx0 = tile_max_fin(t8);
  }

  using TA = std::true_type;
  using TB = std::false_type;
  using P0T = std::integral_constant<int, 0>;
  using P1T = std::integral_constant<int, 1>;
  for (int t = 0; t < nkt; t += 2) {
    slot(t, t == 0, TA{}, P0T{});
    slot(t, t == 0, TB{}, P0T{});
    if (t + 1 < nkt) {
      slot(t + 1, false, TA{}, P1T{});
      slot(t + 1, false, TB{}, P1T{});
    }
  }
  // ---- last half slot: O1 += V(nkt-1).P1(nkt-1), l1 += 1.P1(nkt-1)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the re-fetched tail pieces too: nothing in flight at exit)
  __builtin_amdgcn_sched_barrier(0);
  auto fin = [&](auto par_tag) {
    constexpr int VBUF = A_V + 32 * decltype(par_tag)::value;
    static_for<12>([&](auto gi) {
      constexpr int s4 = decltype(gi)::value / 3, c = decltype(gi)::value % 3;
      if constexpr (c < 2) mfma_pv<BF, A_O + (2 + c) * 16, VBUF + (c * 4 + s4) * 4>(P1[s4]);
      else mfma_ls<BF, A_L + 16>(ones, P1[s4]);
    });
  };
  if ((nkt - 1) & 1) fin(P1T{});
  else fin(P0T{});
  mfma_drain();
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: O^T / l -> 16-bit rows of `out` (a lane owns 8 groups of 4 consecutive features of one query)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = q0 + qb * 32 + j;
    float l = 0.f;
    if (qb == 0) l = acc_read<A_L>();
    else l = acc_read<A_L + 16>();
    const float inv = 1.0f / l;
    T* orow = out + ((long long)img * ntok + (qi < ntok ? qi : 0)) * ldo + head * 64;
    static_for<8>([&](auto ci) {
      constexpr int c = decltype(ci)::value, dt = c >> 2, r4 = c & 3;
      float o[4];
      if (qb == 0) {
        o[0] = acc_read<A_O + dt * 16 + r4 * 4 + 0>(); o[1] = acc_read<A_O + dt * 16 + r4 * 4 + 1>();
        o[2] = acc_read<A_O + dt * 16 + r4 * 4 + 2>(); o[3] = acc_read<A_O + dt * 16 + r4 * 4 + 3>();
      } else {
        o[0] = acc_read<A_O + 32 + dt * 16 + r4 * 4 + 0>(); o[1] = acc_read<A_O + 32 + dt * 16 + r4 * 4 + 1>();
        o[2] = acc_read<A_O + 32 + dt * 16 + r4 * 4 + 2>(); o[3] = acc_read<A_O + 32 + dt * 16 + r4 * 4 + 3>();
      }
      V4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (T)(o[e] * inv);
      if (qi < ntok) *(V4*)(orow + dt * 32 + r4 * 8 + hi * 4) = w;
    });
  }
}

}  // namespace

namespace mk {
bool launch_attn_w1(const void* q, const void* k, const void* vt, void* out, int ldo, int nimg, int heads, int ntok, int ntok_pad,
                    int dtype, hipStream_t st) {
  if ((ntok + 63) / 64 < 4) return false;
  const dim3 grid((ntok + 255) / 256, heads, nimg);
  if (dtype == MK_BF16)
    hipLaunchKernelGGL((attn_w1_kernel<__bf16>), grid, dim3(256), 0, st, (const __bf16*)q, (const __bf16*)k, (const __bf16*)vt,
                       (__bf16*)out, ldo, heads, ntok, ntok_pad);
  else
    hipLaunchKernelGGL((attn_w1_kernel<_Float16>), grid, dim3(256), 0, st, (const _Float16*)q, (const _Float16*)k,
                       (const _Float16*)vt, (_Float16*)out, ldo, heads, ntok, ntok_pad);
  return true;
}
}  // namespace mk
