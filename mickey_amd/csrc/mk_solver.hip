// mickey_amd -- probabilistic-Procrustes RANSAC on gfx950
// (reference lib/models/MicKey/modules/utils/probabilisticProcrustes.py:183-348).
//
// The reference materialises a [20B, n*n] tiled copy of final_scores (300 MB / pair), a same-sized
// Exp(1) noise tensor and a full top-k for its outer torch.multinomial, then tiles X/Y 100x for the
// inner one.  Here:
//   * mk_exprace_topk: one read of p gives its histogram -- hence the threshold T of the race keys whose EXPECTED tail count
//     is 1.25 k, a function of p alone -- and the maximum of every 16 cells; the candidates {p / e > T} of a row are then
//     generated, not searched for (exprace_skip_kernel: geometric skipping under the 16-cell bound, thinning, keys drawn from
//     Exp(1) conditioned on clearing T: the law of drawing every e, RNG work proportional to the ~2600 candidates of a row
//     instead of its 3.76 M cells); small in-LDS bitonic sort -> the same "top-k of p / Exp(1)" selection, in the same
//     (descending key) order torch.topk returns; an exact radix-histogram path takes over on device if a row collected too
//     few / too many.  Injected noise (tests): every (row, cell) is keyed, one streamed read per group of 4 rows.  The
//     round-3 generator (every cell tested behind a 6-bit pre-filter) is kept as the A/B partner (mk_exprace_set_mode).
//   * mk_train_ransac_masks / mk_reinforce_scatter: the training-time RANSAC of loss/loss_class.py (8-point hypotheses,
//     refinement of every hypothesis, REINFORCE bookkeeping).
//   * mk_ransac_hypotheses: a correspondence set (X, Y, w: 56 KB) is staged once in LDS and shared by
//     all its hypotheses; a hypothesis draws its 3 correspondences ~ w without replacement with three uniforms (prefix
//     sums of the set's weights in LDS, binary search with the chosen ones masked out = the top-3 of an exponential race,
//     which the injected-noise path still runs), 3x3 Kabsch via one-sided Jacobi SVD in fp64 (one per lane, no MFMA), soft
//     inlier count by wave64 reduction.
//   * mk_refine_pose: one workgroup per pair: arg-max, <= 4 masked-Kabsch refits with the reference's
//     per-pair early exit, final confidence.  No host synchronisation anywhere.
// Noise can be INJECTED (fp32 Exp(1) tensors / explicit indices) so that tests are bit-comparable
// with torch; the product path uses Philox4x32-10.
#include "mk_common.hpp"

#pragma clang fp contract(off)  // keep p/e, dist, sigmoid arithmetic un-fused: comparable with ATen

namespace {
using namespace mk;

// ---- Philox4x32-10 ------------------------------------------------------------------------------
struct U4 { unsigned x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(unsigned k0, unsigned k1, U4 c) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    // one 32x32 -> 64 product per multiplier (the compiler can then use v_mad_u64_u32) instead of separate
    // v_mul_hi_u32 + v_mul_lo_u32: integer multiplies are quarter rate and were 37 of the ~105 instructions of the key loop
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * (unsigned long long)c.x;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * (unsigned long long)c.z;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
    const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
// Exp(1) draw from 32 random bits: u in (0,1) on a 2^-24 grid, e = -log(u) > 0
__device__ __forceinline__ float exp1(unsigned r) { return -logf(((float)(r >> 8) + 0.5f) * 5.9604644775390625e-8f); }
// Race key p / e for the on-device (Philox) path: only the ORDER of the keys matters, so hardware log2 / rcp
// (1 ulp-class) replace libm logf and the IEEE divide (~35 -> ~10 instructions per key).  The injected-noise path
// keeps the exact p / e so that it stays bit-comparable with torch.
__device__ __forceinline__ float race_key(float p, unsigned r) {
  const float u = ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-8f;
  const float e = -0.69314718055994531f * __builtin_amdgcn_logf(u);
  return p * __builtin_amdgcn_rcpf(e);
}

// Optional device-resident part of the Philox stream offset (a captured hipGraph re-reads it at every replay)
__device__ __forceinline__ void add_device_offset(unsigned& off_lo, unsigned& off_hi, const unsigned long long* offp) {
  if (!offp) return;
  const unsigned long long o = (((unsigned long long)off_hi << 32) | off_lo) + *offp;
  off_lo = (unsigned)o;
  off_hi = (unsigned)(o >> 32);
}

__global__ void counter_add_kernel(unsigned long long* ctr, unsigned long long inc) { *ctr += inc; }

// ---- exponential-race top-k -------------------------------------------------------------------------
constexpr int NBINS = 2048;     // bits 30..20 of a positive float: exponent + 3 mantissa bits
constexpr int RG = 4;           // rows per group = draws per Philox call
constexpr int CAND_MAX = 8192;  // candidates kept per row (expected ~k * 1.1)
constexpr int CELL_BLOCKS = 128;

// Workspace of mk_exprace_topk.  The words of `ncand | redo | phist | done1` are SELF-CLEANING state: they must be zero
// when a call starts and every call leaves them zero (each is reset by its last reader), so that no zero-fill launch stands in
// front of the chain -- mickey_hip.h: the caller zero-initialises the buffer once and never shares it between streams.
struct TopkWork {
  unsigned* ncand;           // [R]          (state) candidates appended per row; reset by the select kernel
  int* redo;                 // [B]          (state) per pair: set when one of its rows collected fewer than k candidates above an
                             //     analytic threshold or a workgroup's queue overflowed -- the exact fallback then redoes THAT pair only
                             //     (the others keep their skip-sampler draws: a pair's result never depends on its batch)
  unsigned* phist;           // [B][NBINS]   (state) histogram of p itself (analytic threshold); reset by its pair's last workgroup
  unsigned* done1;           // [B]          (state) workgroups of the histogram pass that have finished, per pair
  int* thr;                  // [R]
  unsigned long long* cand;  // [R][CAND_MAX]
  int* invalid;              // [1] or null
  int pair_base;             // global index of pair 0 of this call (keys the Philox streams)
  float* pmax;               // [B][nblk]    largest valid p of every 16 consecutive cells (bound of the skip sampler)
  long long nblk;            // ceil(ncell / 16)
};

__device__ __forceinline__ void row_keys(const float* __restrict__ noise, unsigned k0, unsigned k1, unsigned off_lo,
                                         unsigned off_hi, float p, long long c, long long ncell, int b, int rows_per_pair,
                                         int grp, float key[RG], int pair_base) {
  if (noise) {
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const int r = grp * RG + q;
      key[q] = r < rows_per_pair ? p / noise[((long long)b * rows_per_pair + r) * ncell + c] : 0.f;
    }
  } else {
    const U4 rnd = philox4x32(k0, k1, U4{(unsigned)c, (unsigned)(c >> 32) ^ off_hi, (unsigned)((b + pair_base) * 64 + grp), off_lo});
    key[0] = race_key(p, rnd.x);
    key[1] = race_key(p, rnd.y);
    key[2] = race_key(p, rnd.z);
    key[3] = race_key(p, rnd.w);
  }
}

// Collect pass: candidates are appended to a per-block LDS buffer (LDS atomics return in ~100 cycles) and flushed with ONE
// global atomic per row per block at the end.  Appending straight to global memory stalls the whole wave for a memory
// round trip whenever any of its 256 keys is a candidate -- inside the Philox loop that was ~25 % of the pass.
constexpr int LCAP = 960;   // LDS candidate slots per row and block (expected ~25 at k = 2048, 128 blocks); overflow goes direct

// (the every-cell collect pass: injected noise -- tests -- and more rows per pair than the generators below are built for)
__global__ __launch_bounds__(256) void exprace_scan_kernel(const float* __restrict__ p, const float* __restrict__ noise,
                                                           unsigned k0, unsigned k1, unsigned off_lo, unsigned off_hi,
                                                           const unsigned long long* __restrict__ offp, TopkWork w,
                                                           int rows_per_pair, long long ncell) {
  __shared__ unsigned long long lbuf[RG * LCAP];
  __shared__ unsigned lcount[RG], lbase[RG];
  add_device_offset(off_lo, off_hi, offp);
  const int b = blockIdx.z, grp = blockIdx.y;
  const long long per = (ncell + gridDim.x - 1) / gridDim.x;
  const long long c0 = blockIdx.x * per, c1 = min(ncell, c0 + per);
  int thr[RG];
  if (threadIdx.x < RG) lcount[threadIdx.x] = 0;
#pragma unroll
  for (int q = 0; q < RG; ++q) {
    const int r = grp * RG + q;
    thr[q] = r < rows_per_pair ? w.thr[b * rows_per_pair + r] : NBINS;
  }
  __syncthreads();
  const float* pb = p + (long long)b * ncell;
  for (long long c = c0 + threadIdx.x; c < c1; c += 256) {
    const float pv = pb[c];
    if (!(pv > 0.f) || isinf(pv)) continue;
    float key[RG];
    row_keys(noise, k0, k1, off_lo, off_hi, pv, c, ncell, b, rows_per_pair, grp, key, w.pair_base);
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const int r = grp * RG + q;
      if (r >= rows_per_pair) continue;
      const unsigned bits = __float_as_uint(key[q]);
      const int bin = (int)((bits & 0x7fffffffu) >> 20);
      if (bin >= thr[q]) {
        const unsigned long long item = ((unsigned long long)bits << 32) | (unsigned)(0xffffffffu - (unsigned)c);
        const unsigned ls = atomicAdd(&lcount[q], 1u);
        if (ls < (unsigned)LCAP) {
          lbuf[q * LCAP + ls] = item;
        } else {   // block-local overflow (pathological inputs): append directly
          const int row = b * rows_per_pair + r;
          const unsigned slot = atomicAdd(&w.ncand[row], 1u);
          if (slot < CAND_MAX) w.cand[(long long)row * CAND_MAX + slot] = item;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < RG) {
    const int q = threadIdx.x, r = grp * RG + q;
    const unsigned nloc = min(lcount[q], (unsigned)LCAP);
    lbase[q] = (r < rows_per_pair && nloc) ? atomicAdd(&w.ncand[b * rows_per_pair + r], nloc) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RG; ++q) {
    const int r = grp * RG + q;
    if (r >= rows_per_pair) continue;
    const unsigned nloc = min(lcount[q], (unsigned)LCAP), base = lbase[q];
    const long long row = (long long)b * rows_per_pair + r;
    for (unsigned i = threadIdx.x; i < nloc; i += 256)
      if (base + i < (unsigned)CAND_MAX) w.cand[row * CAND_MAX + base + i] = lbuf[q * LCAP + i];
  }
}

// ---- exact fallback: ONE workgroup does RG rows of a pair start to finish (early exit unless the pair's `redo` is set: the
// common case pays one launch of idle workgroups).  Pass 0: key histograms of its rows in LDS; the largest bin whose tail holds
// >= k keys per row; pass 1: the keys at or above it are the candidates -- the exact top-k whatever the distribution of p or of
// the (injected) noise.  Slow by design (a workgroup walks all cells twice: ~1 ms at 3.8 M cells); it never runs on a matcher's
// output, it is what makes the result EXACT for adversarial inputs.
__global__ __launch_bounds__(1024) void exprace_fallback_kernel(const float* __restrict__ p, const float* __restrict__ noise,
                                                                unsigned k0, unsigned k1, unsigned off_lo, unsigned off_hi,
                                                                const unsigned long long* __restrict__ offp, TopkWork w,
                                                                int rows_per_pair, long long ncell, int k) {
  __shared__ unsigned sh[RG * NBINS];
  __shared__ unsigned part[256];
  __shared__ int thr_s[RG];
  __shared__ unsigned cnt_s[RG];
  const int b = blockIdx.y, grp = blockIdx.x, t = threadIdx.x;
  // does this pair need the fallback?  A queue of the generator overflowed (`redo`, raised there), or one of the pair's rows fell
  // short of k candidates although its threshold was not "everything", or overflowed its candidate buffer (noise that is not
  // Exp(1)-distributed can do either).  Every workgroup of the pair evaluates the same rows_per_pair counts: no hand-over between
  // workgroups, no check launch, no tail in the collect pass (round 6).
  __shared__ int need_s;
  if (t == 0) need_s = w.redo[b];
  __syncthreads();
  if (t < rows_per_pair) {
    const unsigned nc = w.ncand[b * rows_per_pair + t];
    if ((w.thr[b * rows_per_pair + t] > 0 && nc < (unsigned)k) || nc > (unsigned)CAND_MAX) atomicOr(&need_s, 1);
  }
  __syncthreads();
  if (need_s == 0) return;
  add_device_offset(off_lo, off_hi, offp);
  for (int i = t; i < RG * NBINS; i += 1024) sh[i] = 0;
  if (t < RG) cnt_s[t] = 0;
  __syncthreads();
  const float* pb = p + (long long)b * ncell;
  for (long long c = t; c < ncell; c += 1024) {
    const float pv = pb[c];
    if (!(pv > 0.f) || isinf(pv)) continue;
    float key[RG];
    row_keys(noise, k0, k1, off_lo, off_hi, pv, c, ncell, b, rows_per_pair, grp, key, w.pair_base);
#pragma unroll
    for (int q = 0; q < RG; ++q)
      if (grp * RG + q < rows_per_pair) atomicAdd(&sh[q * NBINS + (int)((__float_as_uint(key[q]) & 0x7fffffffu) >> 20)], 1u);
  }
  __syncthreads();
  // per row: largest bin tb with count(bins >= tb) >= k (0 if fewer than k non-zero keys); thread t < 256 owns 8 bins
  for (int q = 0; q < RG; ++q) {
    const unsigned* h = sh + q * NBINS;
    if (t < 256) {
      unsigned loc = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) loc += h[t * 8 + i];
      part[t] = loc;
    }
    __syncthreads();
    if (t == 0) {
      unsigned run = 0;
      int tb = 0;
      for (int s8 = 255; s8 >= 0; --s8) {
        if (run + part[s8] >= (unsigned)k) {
          for (int i = 7; i >= 0; --i) {
            run += h[s8 * 8 + i];
            if (run >= (unsigned)k) { tb = s8 * 8 + i; break; }
          }
          break;
        }
        run += part[s8];
      }
      thr_s[q] = tb;
      if (grp * RG + q < rows_per_pair) w.thr[b * rows_per_pair + grp * RG + q] = tb;
    }
    __syncthreads();
  }
  for (long long c = t; c < ncell; c += 1024) {
    const float pv = pb[c];
    if (!(pv > 0.f) || isinf(pv)) continue;
    float key[RG];
    row_keys(noise, k0, k1, off_lo, off_hi, pv, c, ncell, b, rows_per_pair, grp, key, w.pair_base);
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const int r = grp * RG + q;
      if (r >= rows_per_pair) continue;
      const unsigned bits = __float_as_uint(key[q]);
      if ((int)((bits & 0x7fffffffu) >> 20) < thr_s[q]) continue;
      const unsigned slot = atomicAdd(&cnt_s[q], 1u);
      if (slot < (unsigned)CAND_MAX)
        w.cand[((long long)b * rows_per_pair + r) * CAND_MAX + slot] = ((unsigned long long)bits << 32) | (unsigned)(0xffffffffu - (unsigned)c);
    }
  }
  __syncthreads();
  if (t < RG && grp * RG + t < rows_per_pair) w.ncand[b * rows_per_pair + grp * RG + t] = cnt_s[t];   // replaces what the generator counted
}

// ---- Philox collect pass with a 6-bit pre-filter ---------------------------------------------------------------------
// The scan above draws one full uniform per (row, cell): 5 Philox calls per cell for 20 rows, 2.4 G draws per 32-pair
// step, of which one in ~1400 becomes a candidate.  A key p/e reaches the threshold T only if e = -ln(u) <= p/T; for a
// cell with p < 0.0155 T that needs u > 1 - 2^-6, i.e. the top 6 bits of the 24-bit uniform all ones.  So:
//   stage 1  one Philox call per cell yields the TOP 6 bits of the uniforms of 20 rows (5 x 6 bits per 32-bit word); only
//            (cell, row) pairs whose 6 bits are all ones (1 in 64) are queued in LDS;
//   stage 2  the queue is processed densely: one Philox call per queued pair, keyed by (cell, row), yields the LOW 18
//            bits; u = (top6 << 18 | low18 + 0.5) 2^-24, key = p / -ln(u), candidate test, append (as in the scan).
// The joint distribution is that of independent 24-bit uniforms per (row, cell) -- the sampler is unchanged, the RNG work
// drops from 5 to ~1.3 Philox calls per cell.  Cells with p >= 0.0155 T (a few thousand per pair) take stage 2 for all
// their rows.  Counter layout: z = (pair + pair_base) * 512 + {256 + row / 20 (stage 1) | row (stage 2)}.
constexpr int PF_CPT = 3;         // cells per thread per iteration: 768 cells queue ~240 (cell, row) pairs = ONE dense round of stage 2
                                  // (4 cells: 320 pairs = two rounds, the second a quarter full)
constexpr int PF_QCAP = 2048;     // queued (cell, row) pairs per iteration (expected 320 + 20 per large-p cell)
constexpr int PF_LCAP = 128;      // LDS candidate slots per row and block (expected ~20)
constexpr int PF_MAXROWS = 48;   // LDS: rows x 1 KiB of candidates + 8 KiB queue <= 56 KiB

__global__ __launch_bounds__(256) void exprace_prefilter_kernel(const float* __restrict__ p, unsigned k0, unsigned k1,
                                                                unsigned off_lo, unsigned off_hi,
                                                                const unsigned long long* __restrict__ offp, TopkWork w,
                                                                int rows_per_pair, long long ncell) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pf_smem[];
  unsigned long long* lbuf = (unsigned long long*)pf_smem;                       // [rows][PF_LCAP]
  unsigned* queue = (unsigned*)(pf_smem + (size_t)rows_per_pair * PF_LCAP * 8);   // [PF_QCAP]
  __shared__ unsigned lcount[PF_MAXROWS], lbase[PF_MAXROWS];
  __shared__ unsigned qcount;
  add_device_offset(off_lo, off_hi, offp);
  const int b = blockIdx.y;
  const long long per = (ncell + gridDim.x - 1) / gridDim.x;
  const long long c0 = blockIdx.x * per, c1 = min(ncell, c0 + per);
  const unsigned zb = (unsigned)(b + w.pair_base) * 512u;
  const int thr0 = w.thr[b * rows_per_pair];          // one analytic threshold per pair (the tail of exprace_phist_kernel)
  const float T = __uint_as_float((unsigned)thr0 << 20);
  const float pfast = 0.0155f * T;                     // -ln(1 - 2^-6) = 0.015748: margin for the 1-ulp log / rcp
  const float pnever = 2.9e-8f * T;                    // smallest e: -ln(1 - 2^-25) = 2.98e-8
  if (threadIdx.x < rows_per_pair) lcount[threadIdx.x] = 0;
  if (threadIdx.x == 0) qcount = 0;
  __syncthreads();
  const float* pb = p + (long long)b * ncell;

  auto finish = [&](unsigned cl, int r, unsigned top6) {   // stage 2 for one (cell, row)
    const long long c = c0 + cl;
    const float pv = pb[c];
    const U4 rnd = philox4x32(k0, k1, U4{(unsigned)c, off_hi, zb + (unsigned)r, off_lo});
    const unsigned r24 = (top6 << 18) | (rnd.x >> 14);
    const float key = race_key(pv, r24 << 8);
    const unsigned bits = __float_as_uint(key);
    if ((int)((bits & 0x7fffffffu) >> 20) < thr0) return;
    const unsigned long long item = ((unsigned long long)bits << 32) | (unsigned)(0xffffffffu - (unsigned)c);
    const unsigned ls = atomicAdd(&lcount[r], 1u);
    if (ls < (unsigned)PF_LCAP) {
      lbuf[r * PF_LCAP + ls] = item;
    } else {
      const int row = b * rows_per_pair + r;
      const unsigned slot = atomicAdd(&w.ncand[row], 1u);
      if (slot < CAND_MAX) w.cand[(long long)row * CAND_MAX + slot] = item;
    }
  };

  for (long long base = c0; base < c1; base += 256 * PF_CPT) {
#pragma unroll
    for (int u = 0; u < PF_CPT; ++u) {
      const long long c = base + u * 256 + threadIdx.x;
      if (c >= c1) continue;
      const float pv = pb[c];
      if (!(pv > pnever) || isinf(pv)) continue;   // key <= p / e_min < T for every possible draw (and p <= 0)
      const unsigned cl = (unsigned)(c - c0);
      const bool fast = pv < pfast;
      for (int rc = 0; rc * 20 < rows_per_pair; ++rc) {
        const U4 rnd = philox4x32(k0, k1, U4{(unsigned)c, off_hi, zb + 256u + (unsigned)rc, off_lo});
        const unsigned wd[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
        // 20-bit mask of the rows to queue: all of them for a large-p cell, otherwise those whose 6-bit field is all ones
        // (bit 0 of each field of ~x OR-folded over the field is 0 exactly then) -- ~12 ops per word instead of a
        // compare + branch per row
        unsigned mask = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const unsigned y = ~wd[v];
          const unsigned z = (y | (y >> 1) | (y >> 2) | (y >> 3) | (y >> 4) | (y >> 5)) & 0x01041041u;   // 1 = field not all ones
          const unsigned pass = ~z & 0x01041041u;
          const unsigned m5 = (pass & 1u) | ((pass >> 5) & 2u) | ((pass >> 10) & 4u) | ((pass >> 15) & 8u) | ((pass >> 20) & 16u);
          mask |= m5 << (5 * v);
        }
        const int nr = min(20, rows_per_pair - rc * 20);
        if (!fast) mask = 0xfffffu;
        mask &= (1u << nr) - 1u;
        while (mask) {
          const int q = __builtin_ctz(mask);
          mask &= mask - 1u;
          const unsigned top6 = (wd[q / 5] >> (6 * (q % 5))) & 63u;
          const int r = rc * 20 + q;
          const unsigned pos = atomicAdd(&qcount, 1u);
          if (pos < (unsigned)PF_QCAP) queue[pos] = cl | ((unsigned)r << 16) | (top6 << 24);
          else finish(cl, r, top6);   // queue full (many large-p cells in one iteration): do it in place
        }
      }
    }
    __syncthreads();
    const unsigned n = min(qcount, (unsigned)PF_QCAP);
    for (unsigned i = threadIdx.x; i < n; i += 256) {
      const unsigned it = queue[i];
      finish(it & 0xffffu, (int)((it >> 16) & 0xffu), it >> 24);
    }
    __syncthreads();
    if (threadIdx.x == 0) qcount = 0;
    __syncthreads();
  }
  if (threadIdx.x < rows_per_pair) {
    const int r = threadIdx.x;
    const unsigned nloc = min(lcount[r], (unsigned)PF_LCAP);
    lbase[r] = nloc ? atomicAdd(&w.ncand[b * rows_per_pair + r], nloc) : 0u;
  }
  __syncthreads();
  for (int r = 0; r < rows_per_pair; ++r) {
    const unsigned nloc = min(lcount[r], (unsigned)PF_LCAP), bs = lbase[r];
    const long long row = (long long)b * rows_per_pair + r;
    for (unsigned i = threadIdx.x; i < nloc; i += 256)
      if (bs + i < (unsigned)CAND_MAX) w.cand[row * CAND_MAX + bs + i] = lbuf[r * PF_LCAP + i];
  }
}

// ---- Philox collect pass by geometric skipping (the product path) ------------------------------------------------------------
// With the analytic threshold T a (row, cell) becomes a candidate iff its Exp(1) draw e < p / T: independent Bernoulli events
// of probability s = 1 - exp(-p / T), ~7e-4 on average (1.25 k candidates per row out of n^2 cells).  Testing every (row, cell)
// costs a Philox call per cell even with the pre-filter above; here the candidates of a row are generated DIRECTLY, as a thinned
// Bernoulli process:
//   bound     pmax = largest p of every 16 consecutive cells (written by the histogram pass): every cell of the block is
//             "proposed" with probability S = 1 - exp(-lam), lam = pmax / T >= p / T;
//   walk      proposals are found by skipping: an Exp(1) budget x is spent at the rate lam per cell, the next proposal is cell
//             floor(x / lam) of the block, or -- memorylessness -- the rest of the budget carries over to the thread's next
//             block (two Philox calls per thread for 5 rows x 16 blocks, not one per cell); after a proposal the process
//             resumes at the next cell with a fresh budget;
//   thinning  a proposed cell is accepted with probability s / S (second uniform, keyed by (cell, row)), so it survives with
//             probability s exactly; its race key is p / e with e drawn from Exp(1) conditioned on e < p / T:
//             e = -log1p(-u s);
//   dense     blocks with lam > 0.1 (around a dominant cell; everything when T = 0) skip the skipping: their 16 cells are
//             tested directly with probability s.
// Proposals are queued in LDS and tested densely (all lanes busy), candidates are appended as in the passes above, and the
// select kernel sorts them: the result is the top-k of the race keys of a row -- the same sampling law, with the RNG work
// proportional to the number of candidates instead of the number of cells.  The walk geometry depends on ncell only (not on
// the batch), so a pair's draws do not depend on the batch it is in.
// Counter layout: x = walk thread (+ call << 26) | cell, z = (pair + pair_base) * 512 + {288 + row group (walk) | row (test)}.
constexpr int SK_CELLS = 16;                 // cells per bound block
constexpr int SK_NBW = 16;                   // blocks per thread (their rates live in registers for all rows of the workgroup)
constexpr int SK_RANGE = 256 * SK_NBW;       // blocks per workgroup (65536 cells: a queue entry holds the local cell in 16 bits)
constexpr int SK_ROWS = 5;                   // rows per workgroup
constexpr int SK_QCAP = 1024;                // queued proposals (expected ~300)
constexpr int SK_DCAP = 2048;                // queued dense (block, row) entries
constexpr int SK_LCAP = 128;                 // LDS candidate slots per row and workgroup (expected ~45); beyond: appended directly
constexpr int SK_HSLOTS = 4;                 // LDS slots of a thread for its hits (expected 0.9 per thread)
constexpr float SK_DENSE = 0.1f;
// A queue that overflows (a workgroup range with thousands of proposals: not a distribution a matcher produces) raises
// `redo` of its pair: the exact histogram passes then redo THAT PAIR from scratch (the other pairs of the call keep what the
// walk gave them).

__global__ __launch_bounds__(256) void exprace_skip_kernel(const float* __restrict__ p, unsigned k0, unsigned k1, unsigned off_lo,
                                                           unsigned off_hi, const unsigned long long* __restrict__ offp,
                                                           TopkWork w, int rows_per_pair, long long ncell) {
  __shared__ unsigned long long lbuf[SK_ROWS * SK_LCAP], hits[SK_QCAP], slots[SK_HSLOTS * 256];
  __shared__ unsigned dqueue[SK_DCAP];
  __shared__ unsigned lcount[SK_ROWS], lbase[SK_ROWS], qcount, dcount;
  add_device_offset(off_lo, off_hi, offp);
  const int b = blockIdx.z, r0 = blockIdx.y * SK_ROWS, nr = min(SK_ROWS, rows_per_pair - r0), t = threadIdx.x;
  const long long blk0 = (long long)blockIdx.x * SK_RANGE;
  const int nb = (int)min((long long)SK_RANGE, w.nblk - blk0);
  const long long cbase = blk0 * SK_CELLS;
  const unsigned zb = (unsigned)(b + w.pair_base) * 512u;
  const int thr0 = w.thr[b * rows_per_pair];          // one analytic threshold per pair (the tail of exprace_phist_kernel)
  const float T = __uint_as_float((unsigned)thr0 << 20);
  const float invT = T > 0.f ? 1.f / T : __builtin_inff();   // T = 0: "collect every positive cell"
  const float* pb = p + (long long)b * ncell;
  const float* pmb = w.pmax + (long long)b * w.nblk + blk0;
  // budget a block takes off the walk: 16 x its rate bound (block j * 256 + t of the range); +inf marks a dense block
  float span[SK_NBW];
#pragma unroll
  for (int j = 0; j < SK_NBW; ++j) {
    const int bl = j * 256 + t;
    const float pm = bl < nb ? pmb[bl] : 0.f;
    const float lj = pm > 0.f ? pm * invT : 0.f;
    span[j] = lj > SK_DENSE ? __builtin_inff() : (float)SK_CELLS * lj;
  }
  if (t < SK_ROWS) lcount[t] = 0;
  if (t == 0) { qcount = 0; dcount = 0; }
  __syncthreads();

  // ---- the walk.  It only FINDS the blocks with a proposal: a hit (block, row, budget left at the block) goes to the
  // thread's own LDS slots (no return value to wait for) and the walk goes on at the next block with a fresh budget
  // (memorylessness again); the rest of the hit block belongs to phase 2.  A wave step covers 1024 cells and half of the
  // steps have a hit in some lane, so whatever a hit costs is paid by all 64 lanes: no LDS round trip, no RNG call and no
  // logarithm inside the loop in the common case.
  // Budgets: hardware log2 (1 ulp-class; they only place the proposals).  Two Philox calls per thread hold the first budget of
  // each of the workgroup's rows and three spares; a thread with more hits draws again.
  auto budget_of = [](unsigned r) { return -0.69314718055994531f * __builtin_amdgcn_logf(((float)(r >> 8) + 0.5f) * 5.9604644775390625e-8f); };
  const unsigned walk = blockIdx.x * 256u + (unsigned)t, zw = zb + 288u + blockIdx.y;
  const U4 ra = philox4x32(k0, k1, U4{walk, off_hi, zw, off_lo});
  const U4 rb = philox4x32(k0, k1, U4{walk | (1u << 26), off_hi, zw, off_lo});
  const float first[SK_ROWS] = {budget_of(ra.x), budget_of(ra.y), budget_of(ra.z), budget_of(ra.w), budget_of(rb.x)};
  float sp0 = budget_of(rb.y), sp1 = budget_of(rb.z), sp2 = budget_of(rb.w), sp3 = 0.f;   // spare budgets, used from sp0 up
  int left = 3;          // spares left
  unsigned refill = 1;
  bool over = false;
  int nh = 0;
  static_assert(SK_ROWS == 5, "first budgets: ra.x .. ra.w, rb.x");
#pragma unroll 1
  for (int rl = 0; rl < nr; ++rl) {
    float budget = rl == 0 ? first[0] : rl == 1 ? first[1] : rl == 2 ? first[2] : rl == 3 ? first[3] : first[4];
#pragma unroll
    for (int j = 0; j < SK_NBW; ++j) {
      if (!(budget < span[j])) {   // (the common case, also every empty block: no proposal)
        budget -= span[j];
        continue;
      }
      const unsigned meta = (unsigned)(j * 256 + t) | ((unsigned)rl << 16);
      if (span[j] == __builtin_inff()) {   // dense: spends no budget (it is not part of the skipping process)
        const unsigned pos = atomicAdd(&dcount, 1u);
        if (pos < (unsigned)SK_DCAP) dqueue[pos] = meta;
        else over = true;
        continue;
      }
      const unsigned long long e = ((unsigned long long)meta << 32) | __float_as_uint(budget);
      if (nh < SK_HSLOTS) {
        slots[nh * 256 + t] = e;
      } else {   // more hits than slots in one thread: straight to the queue
        const unsigned pos = atomicAdd(&qcount, 1u);
        if (pos < (unsigned)SK_QCAP) hits[pos] = e;
        else over = true;
      }
      ++nh;
      if (left == 0) {   // the spares are spent: next call of this walk
        ++refill;
        const U4 rc = philox4x32(k0, k1, U4{walk | (refill << 26), off_hi, zw, off_lo});
        sp0 = budget_of(rc.x); sp1 = budget_of(rc.y); sp2 = budget_of(rc.z); sp3 = budget_of(rc.w);
        left = 4;
        budget = sp0; sp0 = sp1; sp1 = sp2; sp2 = sp3;
      } else {
        budget = sp0; sp0 = sp1; sp1 = sp2; sp2 = sp3;
      }
      --left;
    }
  }
  const int nown = min(nh, SK_HSLOTS);
  if (nown) {
    const unsigned base = atomicAdd(&qcount, (unsigned)nown);
    if (base + nown > (unsigned)SK_QCAP) over = true;
#pragma unroll
    for (int h = 0; h < SK_HSLOTS; ++h)
      if (h < nown && base + h < (unsigned)SK_QCAP) hits[base + h] = slots[h * 256 + t];
  }
  if (over) atomicOr(&w.redo[b], 1);
  __syncthreads();

  // thinning + conditional race key of one proposed (cell, row); rnd = Philox keyed by (cell, row)
  auto test = [&](long long c, int rl, float bound, const U4& rnd) {
    if (c >= ncell) return;
    const float pv = pb[c];
    if (!(pv > 0.f) || isinf(pv)) return;
    const float s = -expm1f(-pv * invT);
    const float ua = ((float)(rnd.x >> 8) + 0.5f) * 5.9604644775390625e-8f;
    if (!(ua * bound < s)) return;
    const float uk = ((float)(rnd.y >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float e = -log1pf(-uk * s);
    if (!(e > 0.f)) return;
    const unsigned bits = __float_as_uint(pv / e);
    const unsigned long long item = ((unsigned long long)bits << 32) | (unsigned)(0xffffffffu - (unsigned)c);
    const unsigned ls = atomicAdd(&lcount[rl], 1u);
    if (ls < (unsigned)SK_LCAP) {
      lbuf[rl * SK_LCAP + ls] = item;
    } else {
      const int row = b * rows_per_pair + r0 + rl;
      const unsigned slot = atomicAdd(&w.ncand[row], 1u);
      if (slot < CAND_MAX) w.cand[(long long)row * CAND_MAX + slot] = item;
    }
  };
  // ---- phase 2a: one lane per hit block.  The first proposal sits where the walk's budget ran out; after every proposal the
  // rest of the block is walked with a fresh budget (third word of the cell's Philox call).
  const unsigned nq = min(qcount, (unsigned)SK_QCAP), nd = min(dcount, (unsigned)SK_DCAP);
  for (unsigned i = t; i < nq; i += 256) {
    const unsigned long long e = hits[i];
    const unsigned meta = (unsigned)(e >> 32);
    const int bl = (int)(meta & 0xffffu), rl = (int)(meta >> 16);
    const float lj = pmb[bl] * invT;
    const float bound = -expm1f(-lj);
    float budget = __uint_as_float((unsigned)e), left = (float)SK_CELLS;
#pragma unroll 1
    while (true) {
      const int cell = SK_CELLS - (int)left + min((int)(budget / lj), (int)left - 1);
      const long long c = cbase + (long long)bl * SK_CELLS + cell;
      const U4 rnd = philox4x32(k0, k1, U4{(unsigned)c, off_hi, zb + (unsigned)(r0 + rl), off_lo});
      test(c, rl, bound, rnd);
      left = (float)(SK_CELLS - 1 - cell);
      budget = budget_of(rnd.z);
      if (!(budget < left * lj)) break;
    }
  }
  // ---- phase 2b: the 16 cells of every dense (block, row) entry, one lane each, accepted with probability s
  for (unsigned i = t; i < nd * SK_CELLS; i += 256) {
    const unsigned meta = dqueue[i / SK_CELLS];
    const long long c = cbase + (long long)(meta & 0xffffu) * SK_CELLS + (i % SK_CELLS);
    const int rl = (int)(meta >> 16);
    const U4 rnd = philox4x32(k0, k1, U4{(unsigned)c, off_hi, zb + (unsigned)(r0 + rl), off_lo});
    test(c, rl, 1.f, rnd);
  }
  __syncthreads();
  if (t < nr) {
    const unsigned nloc = min(lcount[t], (unsigned)SK_LCAP);
    lbase[t] = nloc ? atomicAdd(&w.ncand[b * rows_per_pair + r0 + t], nloc) : 0u;
  }
  __syncthreads();
  for (int rl = 0; rl < nr; ++rl) {
    const unsigned nloc = min(lcount[rl], (unsigned)SK_LCAP), bs = lbase[rl];
    const long long row = (long long)b * rows_per_pair + r0 + rl;
    for (unsigned i = t; i < nloc; i += 256)
      if (bs + i < (unsigned)CAND_MAX) w.cand[row * CAND_MAX + bs + i] = lbuf[rl * SK_LCAP + i];
  }
}

// ---- analytic threshold --------------------------------------------------------------------------------------
// The number of race keys p_i / E_i (E_i ~ Exp(1)) above T is a sum of independent Bernoulli(1 - exp(-p_i / T)): its
// mean is a function of p alone, shared by all draws of a pair.  So instead of generating all rows_per_pair x ncell
// keys once just to histogram them (a full Philox pass), histogram p (one RNG-free read), pick the key bin whose
// expected tail count is >= 1.25 k (k = 2048: +11 sigma) and go straight to the collect pass.  The result is still
// the EXACT top-k of the keys as long as a row collected >= k candidates; otherwise `redo` is raised and the exact
// histogram passes below run (they early-exit on the flag, so the common case pays only their launch).
__device__ __forceinline__ float row16_max(float m) {   // maximum over the lane's DPP row (16 lanes); every lane gets it
#define MK_ROR_MAX(n) m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x120 + n, 0xf, 0xf, false)))
  MK_ROR_MAX(1); MK_ROR_MAX(2); MK_ROR_MAX(4); MK_ROR_MAX(8);
#undef MK_ROR_MAX
  return m;
}

// histogram of p (analytic threshold) and, in the same read, the largest valid p of every 16 consecutive cells (w.pmax: the
// rate bound of the skip sampler below).  A workgroup's cell range starts at a multiple of 256, a wave reads 64 consecutive
// cells per iteration: a 16-cell block is one DPP row.
// Round 6: (a) for large matrices (>= 2^20 cells) only every fourth 256-cell group enters the histogram, weighted 4: the threshold
// only has to land the expected candidate count near 1.25 k (any T that leaves every row between k and CAND_MAX candidates gives
// the exact top-k of the keys; a shortfall raises `redo`), while the LDS atomics -- most cells share a handful of bins, i.e. one
// address per wave -- were the pass's bottleneck, not the read; the rule depends on ncell alone (never on the batch);
// (b) the threshold search is the TAIL of this kernel: the last workgroup of a pair to finish reduces the pair's histogram
// (one launch less in front of the collect pass, and the pairs' searches overlap the other pairs' reads).
// one pair's analytic threshold: largest key bin t whose expected tail count sum_bins h[pb] * (1 - exp(-p_mid(pb) / T_t)) >= need
__device__ __forceinline__ void athresh_block(const TopkWork& w, int b, int rows_per_pair, float need, float* red) {
  const int t = threadIdx.x;
  float hp[8], pm[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int bin = t * 8 + i;
    hp[i] = (float)__hip_atomic_load(&w.phist[(long long)b * NBINS + bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pm[i] = __uint_as_float(((unsigned)bin << 20) | (1u << 19));   // middle of the bin
    w.phist[(long long)b * NBINS + bin] = 0;                        // self-cleaning state (TopkWork)
  }
  auto expected = [&](int tb) {   // block-wide; every thread returns the total
    const float T = __uint_as_float((unsigned)tb << 20);           // lower edge of key bin tb
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (hp[i] > 0.f) a += hp[i] * -expm1f(-pm[i] / T);
    red[t] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) red[t] += red[t + o];
      __syncthreads();
    }
    const float tot = red[0];
    __syncthreads();
    return tot;
  };
  // bins >= 0x7f8 are inf / nan; bin 0 means "collect every positive key"
  int lo = 0, hi = 0x7f7;
  if (expected(1) >= need) {
    lo = 1;
    while (lo < hi) {   // invariant: expected(lo) >= need
      const int mid = (lo + hi + 1) >> 1;
      if (expected(mid) >= need) lo = mid; else hi = mid - 1;
    }
  }
  for (int r = t; r < rows_per_pair; r += 256) w.thr[b * rows_per_pair + r] = lo;
}

__global__ __launch_bounds__(256) void exprace_phist_kernel(const float* __restrict__ p, TopkWork w, long long ncell, int rows_per_pair,
                                                            float need) {
  __shared__ unsigned sh[NBINS];
  __shared__ float red[256];
  __shared__ bool last_wg;
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < NBINS; i += 256) sh[i] = 0;
  __syncthreads();
  const long long per = ((ncell + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const long long c0 = blockIdx.x * per, c1 = min(ncell, c0 + per);
  const float* pb = p + (long long)b * ncell;
  float* pm = w.pmax + (long long)b * w.nblk;
  const bool sub = ncell >= (1LL << 20);   // subsampled histogram (see above)
  bool bad = false;
  if ((ncell & 3) == 0 && ((uintptr_t)p & 15) == 0) {
    // 16-byte loads, four per thread in flight (64 B per lane: with 4-byte loads the pass had 32 KB per CU in flight and ran at
    // 3 TB/s -- latency-bound, not bandwidth-bound); a 16-cell block is four consecutive lanes: two quad-permute steps
    for (long long base = c0; base < c1; base += 4096) {
      f32x4 pv[4];
      long long c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c[u] = base + (u * 256 + threadIdx.x) * 4;
        pv[u] = c[u] < c1 ? __builtin_nontemporal_load((const f32x4*)(pb + c[u])) : f32x4{0.f, 0.f, 0.f, 0.f};   // (c1 % 4 == 0)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float m = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = pv[u][e];
          bad |= !(v >= 0.f) || isinf(v);
          const bool ok = v > 0.f && !isinf(v);
          if (ok && (!sub || u == 0)) atomicAdd(&sh[__float_as_uint(v) >> 20], sub ? 4u : 1u);
          m = fmaxf(m, ok ? v : 0.f);
        }
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
        if ((threadIdx.x & 3) == 0 && c[u] < c1) pm[c[u] >> 4] = m;
      }
    }
  } else {
    for (long long base = c0; base < c1; base += 1024) {   // (uniform trip count: the DPP reduction needs whole rows)
      float pv[4];
      bool in[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {   // four loads in flight per thread
        const long long c = base + u * 256 + threadIdx.x;
        in[u] = c < c1;
        pv[u] = in[u] ? pb[c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long c = base + u * 256 + threadIdx.x;
        bad |= !(pv[u] >= 0.f) || isinf(pv[u]);
        const bool ok = pv[u] > 0.f && !isinf(pv[u]);
        if (ok && (!sub || u == 0)) atomicAdd(&sh[__float_as_uint(pv[u]) >> 20], sub ? 4u : 1u);
        const float m = row16_max(ok ? pv[u] : 0.f);
        if ((threadIdx.x & 15) == 0 && in[u]) pm[c >> 4] = m;
      }
    }
  }
  if (bad && w.invalid) atomicOr(w.invalid, 1);
  __syncthreads();
  for (int i = threadIdx.x; i < NBINS; i += 256)
    if (sh[i]) atomicAdd(&w.phist[(long long)b * NBINS + i], sh[i]);
  // ---- tail: the pair's last workgroup turns the histogram into the pair's threshold (hand-over by device-scope atomics only,
  // performed where every XCD sees them and complete -- waited for below -- before this workgroup's arrival is counted; the last
  // workgroup reads them with device-scope loads.  No __threadfence(): its L2 write-back, once per workgroup, cost four times the pass)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) last_wg = atomicAdd(&w.done1[b], 1u) == gridDim.x - 1u;
  __syncthreads();
  if (!last_wg) return;
  athresh_block(w, b, rows_per_pair, need, red);
  if (threadIdx.x == 0) w.done1[b] = 0;   // self-cleaning state (TopkWork)
}

// one block per row: sort the candidates (key desc, index asc), emit the top k.
// Bitonic network over np2 = 2^m >= nc slots (empty slots hold 0: they sink to the end), thread t owning the E = np2 / 1024
// consecutive slots t E .. t E + E - 1 IN REGISTERS.  A compare-exchange distance j is
//   j < E            inside the thread: register to register;
//   E <= j < 64 E    inside the wave: the partner's value comes by a lane exchange (two 32-bit __shfl_xor), no barrier;
//   j >= 64 E        across waves: through LDS (store, barrier, read the partner slot, barrier).
// For ~2600 candidates (np2 = 4096, E = 4) that is 23 + 45 + 10 passes instead of 78 LDS passes with a workgroup barrier each
// (round 4: 133 us per launch, and the same 133 us of latency for ONE pair's 20 rows); every slot's new value is computed by its
// owner from (own, partner) -- max or min of the pair by its side of the exchange -- so no slot is written by two threads.
// The order is a total order on distinct 64-bit words (key bits | inverted cell index): the result does not depend on the network.
template <int E, int J>   // compare-exchange at distance J < E inside the thread (register indices are compile-time)
__device__ __forceinline__ void select_thread_pass(unsigned long long (&v)[E], int kk, int t) {
  if constexpr (J < E) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if ((e & J) == 0) {
        const int i = t * E + e;
        const bool desc = (i & kk) == 0;
        const unsigned long long x = v[e], y = v[e | J];
        const bool sw = desc ? (x < y) : (x > y);
        v[e] = sw ? y : x;
        v[e | J] = sw ? x : y;
      }
    }
  }
}

constexpr int SEL_T = 1024;       // threads of the select kernel (512 threads with 8 slots each and 32 KiB of LDS -- every row of a 32-pair
                                  // batch resident at once -- measured 100 us against 89: profiles/r06f_sampler_kernel_stats.txt)
constexpr int SEL_LDS = 8192;     // slots of its LDS exchange buffer (= CAND_MAX)

template <int E>
__device__ __forceinline__ void select_sort(unsigned long long (&v)[E], unsigned long long* keys, int np2, int t) {
  for (int kk = 2; kk <= np2; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      if (j < E) {
        if (j == 1) select_thread_pass<E, 1>(v, kk, t);
        else if (j == 2) select_thread_pass<E, 2>(v, kk, t);
        else if (j == 4) select_thread_pass<E, 4>(v, kk, t);
        else select_thread_pass<E, 8>(v, kk, t);
      } else if (j < 64 * E) {
        const int lm = j / E;            // lane distance
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = t * E + e;
          const unsigned long long x = v[e];
          const unsigned lo = __shfl_xor((unsigned)x, lm, 64), hi = __shfl_xor((unsigned)(x >> 32), lm, 64);
          const unsigned long long y = ((unsigned long long)hi << 32) | lo;
          const bool lower = (i & j) == 0, desc = (i & kk) == 0;
          const bool want_max = lower == desc;
          v[e] = want_max ? (x > y ? x : y) : (x < y ? x : y);
        }
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) keys[t * E + e] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = t * E + e;
          const unsigned long long x = v[e], y = keys[i ^ j];
          const bool lower = (i & j) == 0, desc = (i & kk) == 0;
          const bool want_max = lower == desc;
          v[e] = want_max ? (x > y ? x : y) : (x < y ? x : y);
        }
        __syncthreads();
      }
    }
  }
}

template <int E>
__device__ __forceinline__ void select_run(unsigned long long* cand, unsigned long long* keys, int nc, int np2,
                                           int* __restrict__ out, int take) {
  const int t = threadIdx.x;
  unsigned long long v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = t * E + e;
    v[e] = i < nc ? cand[i] : 0ull;
  }
  select_sort<E>(v, keys, np2, t);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = t * E + e;
    if (i < take) out[i] = (int)(0xffffffffu - (unsigned)(v[e] & 0xffffffffu));
  }
}

__global__ __launch_bounds__(SEL_T) void exprace_select_kernel(const float* __restrict__ p, TopkWork w, int* __restrict__ idx,
                                                               int* __restrict__ cnt, int rows_per_pair, long long ncell,
                                                               int k) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // [SEL_LDS]
  const int row = blockIdx.x;
  const unsigned nc_raw = w.ncand[row];
  const int nc = (int)min(nc_raw, (unsigned)CAND_MAX);
  int np2 = SEL_T;                   // at least one slot per thread (E = 1); CAND_MAX = 8192 -> E <= 8
  while (np2 < nc) np2 <<= 1;
  unsigned long long* cand = w.cand + (long long)row * CAND_MAX;
  const int take = min(nc, k);
  int* out = idx + (long long)row * k;
  if (np2 == 1024) select_run<1>(cand, keys, nc, np2, out, take);        // (workgroup-uniform)
  else if (np2 == 2048) select_run<2>(cand, keys, nc, np2, out, take);
  else if (np2 == 4096) select_run<4>(cand, keys, nc, np2, out, take);
  else select_run<8>(cand, keys, nc, np2, out, take);
  if (threadIdx.x == 0) {
    cnt[row] = nc_raw > (unsigned)CAND_MAX ? -1 : take;
    if (nc_raw == 0 && w.invalid) atomicOr(w.invalid, 1);
    if (take < k) {  // degenerate: fewer than k cells with p > 0 -> pad with zero-probability cells
      const float* pb = p + (long long)(row / rows_per_pair) * ncell;
      int f = take;
      for (long long c = 0; c < ncell && f < k; ++c)
        if (!(pb[c] > 0.f)) idx[(long long)row * k + f++] = (int)c;
      for (; f < k; ++f) idx[(long long)row * k + f] = 0;
    }
  }
  // self-cleaning state (TopkWork): this workgroup was the last reader of its row's count, the chain's last kernel of the pair's `redo`
  __syncthreads();
  if (threadIdx.x == 0) {
    w.ncand[row] = 0;
    if (row % rows_per_pair == 0) w.redo[row / rows_per_pair] = 0;
  }
}

// ---- gather + back-projection -----------------------------------------------------------------------
__device__ __forceinline__ void inv3(const float* K, float* o) {
  const float a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
  const float A = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
  const float det = a * A + b * Bc + c * Cc;
  const float id = 1.0f / det;
  o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
  o[3] = Bc * id; o[4] = (a * i - c * g) * id; o[5] = -(a * f - c * d) * id;
  o[6] = Cc * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}

__global__ __launch_bounds__(256) void gather_backproject_kernel(const int* __restrict__ idx, const float* __restrict__ fs,
                                                                 const float* __restrict__ kps0, const float* __restrict__ dep0,
                                                                 const float* __restrict__ kps1, const float* __restrict__ dep1,
                                                                 const float* __restrict__ K0, const float* __restrict__ K1,
                                                                 float* __restrict__ X, float* __restrict__ Y,
                                                                 float* __restrict__ wts, float* __restrict__ corr,
                                                                 int rows_per_pair, int k, int n0, int n1) {
  const int r = blockIdx.y, b = r / rows_per_pair;
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= k) return;
  float Ki0[9], Ki1[9];
  inv3(K0 + b * 9, Ki0);
  inv3(K1 + b * 9, Ki1);
  const int c = idx[(long long)r * k + s];
  const int i = c / n1, j = c - i * n1;
  const float u0 = kps0[((long long)b * 2 + 0) * n0 + i], v0 = kps0[((long long)b * 2 + 1) * n0 + i], d0 = dep0[(long long)b * n0 + i];
  const float u1 = kps1[((long long)b * 2 + 0) * n1 + j], v1 = kps1[((long long)b * 2 + 1) * n1 + j], d1 = dep1[(long long)b * n1 + j];
  const long long o = (long long)r * k + s;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    X[o * 3 + a] = d0 * (Ki0[a * 3 + 0] * u0 + Ki0[a * 3 + 1] * v0 + Ki0[a * 3 + 2]);
    Y[o * 3 + a] = d1 * (Ki1[a * 3 + 0] * u1 + Ki1[a * 3 + 1] * v1 + Ki1[a * 3 + 2]);
  }
  wts[o] = fs[(long long)b * n0 * n1 + c];
  float* cr = corr + o * 6;
  cr[0] = u0; cr[1] = v0; cr[2] = u1; cr[3] = v1; cr[4] = d0; cr[5] = d1;
}

// Backward of the above w.r.t. the keypoints and depths (training: reference loss_class.py:139-146 under autograd; the
// intrinsics are detached).  X_a = d (Ki[a,0] u + Ki[a,1] v + Ki[a,2]):  dL/du = d sum_a gX_a Ki[a,0], dL/dv = d sum_a gX_a Ki[a,1],
// dL/dd = sum_a gX_a (Ki[a,:] . [u, v, 1]).  A keypoint is drawn by many cells of many rows: fp32 atomic adds into zeroed
// outputs (what torch's index backward does on a GPU too).
__global__ __launch_bounds__(256) void gather_backproject_bwd_kernel(const int* __restrict__ idx, const float* __restrict__ corr,
                                                                     const float* __restrict__ gX, const float* __restrict__ gY,
                                                                     const float* __restrict__ K0, const float* __restrict__ K1,
                                                                     float* __restrict__ gkps0, float* __restrict__ gdep0,
                                                                     float* __restrict__ gkps1, float* __restrict__ gdep1,
                                                                     int rows_per_pair, int k, int n0, int n1) {
  const int r = blockIdx.y, b = r / rows_per_pair;
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= k) return;
  float Ki0[9], Ki1[9];
  inv3(K0 + b * 9, Ki0);
  inv3(K1 + b * 9, Ki1);
  const long long o = (long long)r * k + s;
  const int c = idx[o];
  const int i = c / n1, j = c - i * n1;
  const float* cr = corr + o * 6;
  const float u0 = cr[0], v0 = cr[1], u1 = cr[2], v1 = cr[3], d0 = cr[4], d1 = cr[5];
  float gu0 = 0.f, gv0 = 0.f, gd0 = 0.f, gu1 = 0.f, gv1 = 0.f, gd1 = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float gx = gX[o * 3 + a], gy = gY[o * 3 + a];
    gu0 += gx * Ki0[a * 3 + 0];
    gv0 += gx * Ki0[a * 3 + 1];
    gd0 += gx * (Ki0[a * 3 + 0] * u0 + Ki0[a * 3 + 1] * v0 + Ki0[a * 3 + 2]);
    gu1 += gy * Ki1[a * 3 + 0];
    gv1 += gy * Ki1[a * 3 + 1];
    gd1 += gy * (Ki1[a * 3 + 0] * u1 + Ki1[a * 3 + 1] * v1 + Ki1[a * 3 + 2]);
  }
  atomicAdd(gkps0 + ((long long)b * 2 + 0) * n0 + i, d0 * gu0);
  atomicAdd(gkps0 + ((long long)b * 2 + 1) * n0 + i, d0 * gv0);
  atomicAdd(gdep0 + (long long)b * n0 + i, gd0);
  atomicAdd(gkps1 + ((long long)b * 2 + 0) * n1 + j, d1 * gu1);
  atomicAdd(gkps1 + ((long long)b * 2 + 1) * n1 + j, d1 * gv1);
  atomicAdd(gdep1 + (long long)b * n1 + j, gd1);
}

// ---- 3x3 Kabsch: R = V diag(1,1,det(V U^T)) U^T for H = U S V^T (reference loss/solvers.py:45-50) ----
// One-sided Jacobi on the columns of H (no H^T H: keeps fp32-level relative accuracy of the small
// singular directions), fp64, then the two leading singular pairs + right-handed completion, which
// equals the reflection-fixed Kabsch rotation and is finite for degenerate (collinear) input.
__device__ void kabsch_rotation(const double Hin[9], double R[9]) {
  double G[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 9; ++i) G[i] = Hin[i];
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      double al = 0, be = 0, ga = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        al += G[r * 3 + p] * G[r * 3 + p];
        be += G[r * 3 + q] * G[r * 3 + q];
        ga += G[r * 3 + p] * G[r * 3 + q];
      }
      if (fabs(ga) <= 1e-30 || ga * ga <= 1e-32 * al * be) continue;
      off += fabs(ga);
      const double zeta = (be - al) / (2.0 * ga);
      const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double gp = G[r * 3 + p], gq = G[r * 3 + q];
        G[r * 3 + p] = cs * gp - sn * gq;
        G[r * 3 + q] = sn * gp + cs * gq;
        const double vp = V[r * 3 + p], vq = V[r * 3 + q];
        V[r * 3 + p] = cs * vp - sn * vq;
        V[r * 3 + q] = sn * vp + cs * vq;
      }
    }
    if (off == 0.0) break;
  }
  double nrm[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) nrm[c] = G[c] * G[c] + G[3 + c] * G[3 + c] + G[6 + c] * G[6 + c];
  int i0 = 0, i1 = 1, i2 = 2;
  if (nrm[i0] < nrm[i1]) { int t = i0; i0 = i1; i1 = t; }
  if (nrm[i0] < nrm[i2]) { int t = i0; i0 = i2; i2 = t; }
  if (nrm[i1] < nrm[i2]) { int t = i1; i1 = i2; i2 = t; }
  double u1[3], u2[3], v1[3], v2[3];
  const double s1 = sqrt(nrm[i0]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    v1[r] = V[r * 3 + i0];
    v2[r] = V[r * 3 + i1];
    u1[r] = s1 > 1e-300 ? G[r * 3 + i0] / s1 : (r == 0 ? 1.0 : 0.0);
    u2[r] = G[r * 3 + i1];
  }
  // u2: orthogonalise against u1 and normalise; fall back to any perpendicular if it vanishes
  double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
#pragma unroll
  for (int r = 0; r < 3; ++r) u2[r] -= d12 * u1[r];
  double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  if (!(n2 > 1e-12 * s1) || !(n2 > 1e-300)) {
    const int ax = fabs(u1[0]) <= fabs(u1[1]) && fabs(u1[0]) <= fabs(u1[2]) ? 0 : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[ax] = 1.0;
    d12 = u1[ax];
#pragma unroll
    for (int r = 0; r < 3; ++r) u2[r] = e[r] - d12 * u1[r];
    n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) u2[r] /= n2;
  const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
  const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) R[a * 3 + b] = v1[a] * u1[b] + v2[a] * u2[b] + v3[a] * u3[b];
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// dist_j = sqrt(|R x + t - y|^2 + 1e-6)   (reference training_utils.py:58-59)
__device__ __forceinline__ float pt_dist(const float* R, const float* t, const float* x, const float* y) {
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = (R[a * 3 + 0] * x[0] + R[a * 3 + 1] * x[1] + R[a * 3 + 2] * x[2]) + t[a] - y[a];
    s += v * v;
  }
  return sqrtf(s + 1e-6f);
}

// ---- hypotheses: block = (set r, slice of its hypotheses); wave = a few hypotheses, their SVDs one per lane ---
__global__ __launch_bounds__(256) void hypotheses_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                         const float* __restrict__ wts, const float* __restrict__ noise3,
                                                         const int* __restrict__ idx3_in, unsigned k0, unsigned k1,
                                                         unsigned off_lo, unsigned off_hi,
                                                         const unsigned long long* __restrict__ offp, float th_soft,
                                                         float* __restrict__ Rh, float* __restrict__ th,
                                                         float* __restrict__ score, int* __restrict__ idx3, int it_ransac, int k,
                                                         int nsplit, long long set_base) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // X[k*3] | Y[k*3] | w[k] | cdf[k] | scan scratch[256]
  add_device_offset(off_lo, off_hi, offp);
  float* sX = lds;
  float* sY = lds + (size_t)k * 3;
  float* sW = lds + (size_t)k * 6;
  float* sC = lds + (size_t)k * 7;   // inclusive prefix sums of the weights (the on-device 3-sample draws through it)
  float* sS = lds + (size_t)k * 8;
  const int r = blockIdx.x / nsplit, part = blockIdx.x % nsplit;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < k * 3; i += 256) {
    sX[i] = X[(long long)r * k * 3 + i];
    sY[i] = Y[(long long)r * k * 3 + i];
  }
  for (int i = threadIdx.x; i < k; i += 256) sW[i] = wts[(long long)r * k + i];
  __syncthreads();
  const bool cdf_draw = !idx3_in && !noise3;
  if (cdf_draw) {   // block scan: thread t owns the run [t * run, (t + 1) * run)
    const int run = (k + 255) / 256, j0 = threadIdx.x * run;
    float acc = 0.f;
    for (int j = j0; j < min(k, j0 + run); ++j) acc += fmaxf(sW[j], 0.f);
    sS[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {   // Hillis-Steele over the 256 run totals
      const float v = threadIdx.x >= o ? sS[threadIdx.x - o] : 0.f;
      __syncthreads();
      sS[threadIdx.x] += v;
      __syncthreads();
    }
    float c = threadIdx.x ? sS[threadIdx.x - 1] : 0.f;
    for (int j = j0; j < min(k, j0 + run); ++j) {
      c += fmaxf(sW[j], 0.f);
      sC[j] = c;
    }
    __syncthreads();
  }
  const int per = (it_ransac + nsplit - 1) / nsplit;
  const int h0 = part * per, h1 = min(it_ransac, h0 + per);
  const float beta = 5.0f / th_soft;
  // A wave owns the hypotheses h0 + wave, + 4, ...  Three phases per pass of up to HYP_PASS of them:
  //   1. selection (wave-cooperative arg-max over the k matches) and the 3 x 3 cross-covariance, parked in lane i;
  //   2. the fp64 Jacobi SVD of ALL parked hypotheses at once, one per lane (computed redundantly by 64 lanes it was most of
  //      this kernel's time: divisions and square roots in fp64 at a fraction of the fp32 rate, hypothesis after hypothesis);
  //   3. soft-inlier scoring, wave-cooperative again, R and t broadcast from lane i.
  // Every hypothesis sees exactly the arithmetic it saw before: the results are bit-identical.
  constexpr int HYP_PASS = 8;
  for (int hb = h0 + wave; hb < h1; hb += 4 * HYP_PASS) {
    double myH[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float myam[3] = {0.f, 0.f, 0.f}, mybm[3] = {0.f, 0.f, 0.f};
    int mysel[3] = {0, 0, 0};
    // On-device draws: lane i draws the triple of ITS hypothesis (hb + 4 i) by sequential sampling without replacement
    // through the prefix sums -- the same (Plackett-Luce) distribution as the top-3 of the exponential race, order included,
    // from 3 uniforms instead of k Exp(1) draws (the race over 2048 matches was two thirds of this kernel: 8 Philox calls,
    // 32 logarithms and 32 divisions per lane and hypothesis).  Injected noise / indices keep the race (torch-comparable).
    int draw3[3] = {0, 0, 0};
    if (cdf_draw && lane < HYP_PASS && hb + 4 * lane < h1) {
      const long long gh = (long long)r * it_ransac + hb + 4 * lane + set_base * it_ransac;   // GLOBAL hypothesis index
      const U4 rnd = philox4x32(k0, k1, U4{0x3c6ef372u, (unsigned)gh, (unsigned)(gh >> 32) ^ off_hi ^ 0x5bd1e995u, off_lo});
      const unsigned rr[3] = {rnd.x, rnd.y, rnd.z};
      int ex[2] = {-1, -1};        // chosen so far, ascending
      float exw[2] = {0.f, 0.f};   // their weights
      float rem = sC[k - 1];
      for (int d = 0; d < 3; ++d) {
        const float u = ((float)(rr[d] >> 8) + 0.5f) * 5.9604644775390625e-8f;   // (0, 1) on a 2^-24 grid
        const float target = u * rem;
        // smallest j with F'(j) > target, F'(j) = cdf[j] - (weights of the chosen indices <= j)
        int lo = 0, hi = k - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          float f = sC[mid];
          if (ex[0] >= 0 && ex[0] <= mid) f -= exw[0];
          if (ex[1] >= 0 && ex[1] <= mid) f -= exw[1];
          if (f > target) hi = mid; else lo = mid + 1;
        }
        // round-off next to a chosen index / zero-weight runs: step to the next index that can be drawn; if nothing is
        // left (fewer than three positive weights) the lowest index not yet chosen, as the race's tie rule does
        int j = lo;
        for (int tries = 0; tries < k && (j == ex[0] || j == ex[1] || !(sW[j] > 0.f)); ++tries) j = j + 1 < k ? j + 1 : 0;
        if (j == ex[0] || j == ex[1] || !(sW[j] > 0.f)) {
          j = 0;
          while (j == ex[0] || j == ex[1]) ++j;
        }
        draw3[d] = j;
        const float wj = fmaxf(sW[j], 0.f);
        rem = fmaxf(rem - wj, 0.f);
        if (ex[0] < 0) { ex[0] = j; exw[0] = wj; }
        else if (j < ex[0]) { ex[1] = ex[0]; exw[1] = exw[0]; ex[0] = j; exw[0] = wj; }
        else { ex[1] = j; exw[1] = wj; }
      }
    }
    for (int i = 0; i < HYP_PASS; ++i) {
      const int h = hb + 4 * i;
      if (h >= h1) break;
      const long long hyp = (long long)r * it_ransac + h;
      int sel[3];
      if (cdf_draw) {
        sel[0] = __shfl(draw3[0], i, 64);
        sel[1] = __shfl(draw3[1], i, 64);
        sel[2] = __shfl(draw3[2], i, 64);
      } else if (idx3_in) {
        sel[0] = idx3_in[hyp * 3 + 0];
        sel[1] = idx3_in[hyp * 3 + 1];
        sel[2] = idx3_in[hyp * 3 + 2];
      } else {
        // per-lane top-3 of key = w / e, then 3 wave arg-max rounds
        float bk[3] = {-1.f, -1.f, -1.f};
        int bi[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        for (int base = lane * 4; base < k; base += 256) {
          float e[4];
          if (noise3) {
  #pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = base + q < k ? noise3[hyp * k + base + q] : 1.f;
          } else {
            const long long gh = hyp + set_base * it_ransac;   // GLOBAL hypothesis index: draws do not depend on how a batch is split
            const U4 rnd = philox4x32(k0, k1, U4{(unsigned)(base >> 2), (unsigned)gh, (unsigned)(gh >> 32) ^ off_hi ^ 0x5bd1e995u, off_lo});
            e[0] = exp1(rnd.x); e[1] = exp1(rnd.y); e[2] = exp1(rnd.z); e[3] = exp1(rnd.w);
          }
  #pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j = base + q;
            if (j >= k) continue;
            const float key = sW[j] / e[q];
            if (key > bk[2]) {  // strict: earlier (lower) index wins ties
              if (key > bk[0]) { bk[2] = bk[1]; bi[2] = bi[1]; bk[1] = bk[0]; bi[1] = bi[0]; bk[0] = key; bi[0] = j; }
              else if (key > bk[1]) { bk[2] = bk[1]; bi[2] = bi[1]; bk[1] = key; bi[1] = j; }
              else { bk[2] = key; bi[2] = j; }
            }
          }
        }
  #pragma unroll
        for (int round = 0; round < 3; ++round) {
          float v = bk[0];
          int ix = bi[0];
  #pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(v, o, 64);
            const int i2 = __shfl_xor(ix, o, 64);
            if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
          }
          sel[round] = ix;
          if (bi[0] == ix) { bk[0] = bk[1]; bi[0] = bi[1]; bk[1] = bk[2]; bi[1] = bi[2]; bk[2] = -1.f; bi[2] = 0x7fffffff; }
        }
      }
      // means and cross-covariance of the 3 pairs (reference loss/solvers.py:31-39,45-52)
      float am[3], bm[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        am[a] = (sX[sel[0] * 3 + a] + sX[sel[1] * 3 + a] + sX[sel[2] * 3 + a]) / 3.0f;
        bm[a] = (sY[sel[0] * 3 + a] + sY[sel[1] * 3 + a] + sY[sel[2] * 3 + a]) / 3.0f;
      }
      const bool mine = lane == i;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float sm = 0.f;
#pragma unroll
          for (int pnt = 0; pnt < 3; ++pnt) sm += (sX[sel[pnt] * 3 + a] - am[a]) * (sY[sel[pnt] * 3 + c] - bm[c]);
          if (mine) myH[a * 3 + c] = (double)sm;
        }
        if (mine) {
          myam[a] = am[a];
          mybm[a] = bm[a];
          mysel[a] = sel[a];
        }
      }
    }
    double Rd[9];
    kabsch_rotation(myH, Rd);
    float myR[9], myt[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) myR[q] = (float)Rd[q];
#pragma unroll
    for (int a = 0; a < 3; ++a) myt[a] = mybm[a] - (myam[0] * myR[a * 3 + 0] + myam[1] * myR[a * 3 + 1] + myam[2] * myR[a * 3 + 2]);
    for (int i = 0; i < HYP_PASS; ++i) {
      const int h = hb + 4 * i;
      if (h >= h1) break;
      const long long hyp = (long long)r * it_ransac + h;
      float Rf[9], tf[3];
      int sel[3];
#pragma unroll
      for (int q = 0; q < 9; ++q) Rf[q] = __shfl(myR[q], i, 64);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        tf[a] = __shfl(myt[a], i, 64);
        sel[a] = __shfl(mysel[a], i, 64);
      }
      float sc = 0.f;
      for (int jj = lane; jj < k; jj += 64) sc += sigmoidf(beta * (th_soft - pt_dist(Rf, tf, sX + jj * 3, sY + jj * 3)));
      sc = wave_sum(sc);
      if (lane < 9) Rh[hyp * 9 + lane] = Rf[lane];
      if (lane < 3) { th[hyp * 3 + lane] = tf[lane]; idx3[hyp * 3 + lane] = sel[lane]; }
      if (lane == 0) score[hyp] = sc;
    }
  }
}

// ---- arg-max + refinement: one workgroup per pair -----------------------------------------------------
constexpr int RT = 512;
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
  for (int i = 0; i < RT / 64; ++i) s += red[i];
  return s;
}

__global__ __launch_bounds__(RT) void refine_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                    const float* __restrict__ Rh, const float* __restrict__ th,
                                                    const float* __restrict__ score, float th_in, int num_ref, int min_inl,
                                                    float* __restrict__ Ro, float* __restrict__ to, float* __restrict__ conf,
                                                    int* __restrict__ best, unsigned char* __restrict__ mask,
                                                    int* __restrict__ rounds, int* __restrict__ invalid, int it_matches,
                                                    int it_ransac, int k) {
  __shared__ double red[RT / 64];
  __shared__ float sv[RT / 64];
  __shared__ int si[RT / 64];
  __shared__ float sR[9], st[3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HP = it_matches * it_ransac;
  // non-finite hypothesis anywhere in the batch -> the reference returns the zero pose for all pairs
  bool bad = false;
  for (int i = tid; i < HP * 9; i += RT) bad |= !isfinite(Rh[(long long)b * HP * 9 + i]);
  for (int i = tid; i < HP * 3; i += RT) bad |= !isfinite(th[(long long)b * HP * 3 + i]);
  if (bad) atomicOr(invalid, 1);
  // arg-max (first index among equal maxima)
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < HP; i += RT) {
    const float v = score[(long long)b * HP + i];
    if (v > bv) { bv = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(bv, o, 64);
    const int i2 = __shfl_xor(bi, o, 64);
    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < RT / 64; ++i)
      if (sv[i] > bv || (sv[i] == bv && si[i] < bi)) { bv = sv[i]; bi = si[i]; }
    if (bi == 0x7fffffff) bi = 0;  // all-NaN scores
    si[0] = bi;
    for (int i = 0; i < 9; ++i) sR[i] = Rh[((long long)b * HP + bi) * 9 + i];
    for (int i = 0; i < 3; ++i) st[i] = th[((long long)b * HP + bi) * 3 + i];
  }
  __syncthreads();
  const int hb = si[0];
  const long long set = (long long)b * it_matches + hb / it_ransac;
  const float* Xb = X + set * k * 3;
  const float* Yb = Y + set * k * 3;
  float best_cnt = (float)min_inl;
  int nround = 0;
  for (int it = 0; it < num_ref; ++it) {
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = sR[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = st[i];
    double cnt = 0, sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};
    for (int j = tid; j < k; j += RT) {
      if (th_in - pt_dist(R, t, Xb + j * 3, Yb + j * 3) >= 0.f) {
        cnt += 1.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { sa[a] += Xb[j * 3 + a]; sb[a] += Yb[j * 3 + a]; }
      }
    }
    const double C = block_sum_d(cnt, red);
    if (!(C >= (double)min_inl && C > (double)best_cnt)) break;  // block-uniform
    best_cnt = (float)C;
    ++nround;
    // weighted centroids with w / (sum|w| + 1e-16), covariance with the RAW 0/1 mask (solvers.py:14-26)
    double am[3], bm[3];
    const float wn = 1.0f / ((float)C + 1e-16f);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      am[a] = block_sum_d(sa[a], red) * (double)wn;
      bm[a] = block_sum_d(sb[a], red) * (double)wn;
    }
    double hl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = tid; j < k; j += RT) {
      if (th_in - pt_dist(R, t, Xb + j * 3, Yb + j * 3) >= 0.f) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) hl[a * 3 + c] += ((double)Xb[j * 3 + a] - am[a]) * ((double)Yb[j * 3 + c] - bm[c]);
      }
    }
    double H[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = block_sum_d(hl[i], red);
    __syncthreads();
    if (tid == 0) {
      double Rd[9];
      kabsch_rotation(H, Rd);
      for (int i = 0; i < 9; ++i) sR[i] = (float)Rd[i];
      for (int a = 0; a < 3; ++a)
        st[a] = (float)bm[a] - ((float)am[0] * sR[a * 3 + 0] + (float)am[1] * sR[a * 3 + 1] + (float)am[2] * sR[a * 3 + 2]);
    }
    __syncthreads();
  }
  __syncthreads();
  float R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = sR[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = st[i];
  const float beta = 5.0f / th_in;
  double cs = 0;
  for (int j = tid; j < k; j += RT) {
    const float d = pt_dist(R, t, Xb + j * 3, Yb + j * 3);
    cs += (double)sigmoidf(beta * (th_in - d));
    mask[(long long)b * k + j] = (th_in - d) >= 0.f ? 1 : 0;
  }
  const double ctot = block_sum_d(cs, red);
  if (tid < 9) Ro[b * 9 + tid] = R[tid];
  if (tid < 3) to[b * 3 + tid] = t[tid];
  if (tid == 0) { conf[b] = (float)ctot; best[b] = hb; rounds[b] = nround; }
}

// ---- training-time RANSAC (reference loss/loss_class.py:141-184) -------------------------------
// One workgroup per sampled match set (row r: S matches X, Y, w staged in LDS once), one wave per hypothesis.  A hypothesis
// draws NUM_CORR matches without replacement ~ w (exponential race: top-NUM_CORR of w / Exp(1), Philox in registers or
// injected noise / indices), then runs the reference's refinement state machine:
//     cur = sample, fin = sample, pre = NUM_CORR
//     repeat NUM_REF_STEPS: (R, t) = masked Procrustes(cur); ref = {|R x + t - y| <= th}; stop unless |ref| > pre;
//                           pre = |ref|, fin = cur, cur = ref
// and emits fin as a 0/1 float mask [S] -- the input of the differentiable Procrustes the loss is built on.  Match j of a
// row lives in lane j % 64, slot j / 64 (S <= 1024): the three sets are 16-bit masks per lane, no memory traffic.
constexpr int TR_SLOTS = 16;

__global__ __launch_bounds__(256) void train_refine_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                           const float* __restrict__ wts, const float* __restrict__ noise,
                                                           const int* __restrict__ idx_in, unsigned k0, unsigned k1,
                                                           unsigned off_lo, unsigned off_hi,
                                                           const unsigned long long* __restrict__ offp, float th_ref,
                                                           int num_ref, int nc, float* __restrict__ fin_mask,
                                                           int* __restrict__ idx_out, int* __restrict__ rounds_out,
                                                           int it_ransac, int S, long long set_base) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // X[S*3] | Y[S*3] | w[S]
  add_device_offset(off_lo, off_hi, offp);
  float* sX = lds;
  float* sY = lds + (size_t)S * 3;
  float* sW = lds + (size_t)S * 6;
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < S * 3; i += 256) {
    sX[i] = X[(long long)r * S * 3 + i];
    sY[i] = Y[(long long)r * S * 3 + i];
  }
  for (int i = threadIdx.x; i < S; i += 256) sW[i] = wts[(long long)r * S + i];
  __syncthreads();
  const int nq = (S + 63) >> 6;
  for (int h = wave; h < it_ransac; h += 4) {
    const long long hyp = (long long)r * it_ransac + h;
    unsigned cur = 0;
    if (idx_in) {
      for (int c = 0; c < nc; ++c) {
        const int j = idx_in[hyp * nc + c];
        if ((j & 63) == lane) cur |= 1u << (j >> 6);
        if (lane == 0) idx_out[hyp * nc + c] = j;
      }
    } else {
      float key[TR_SLOTS];
#pragma unroll
      for (int q4 = 0; q4 < TR_SLOTS / 4; ++q4) {
        float e[4] = {1.f, 1.f, 1.f, 1.f};
        if (q4 * 4 < nq) {
          if (noise) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = (q4 * 4 + u) * 64 + lane;
              if (j < S) e[u] = noise[hyp * S + j];
            }
          } else {
            const long long gh = hyp + set_base * it_ransac;   // GLOBAL hypothesis index (sharding-invariant draws)
            const U4 rnd = philox4x32(k0, k1, U4{(unsigned)(q4 * 64 + lane), (unsigned)gh, (unsigned)(gh >> 32) ^ off_hi ^ 0x2545f491u, off_lo});
            e[0] = exp1(rnd.x); e[1] = exp1(rnd.y); e[2] = exp1(rnd.z); e[3] = exp1(rnd.w);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = (q4 * 4 + u) * 64 + lane;
          key[q4 * 4 + u] = j < S ? sW[j] / e[u] : -1.f;
        }
      }
      for (int c = 0; c < nc; ++c) {
        float v = -1.f;
        int ix = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < TR_SLOTS; ++q)
          if (!((cur >> q) & 1u) && key[q] > v) { v = key[q]; ix = q * 64 + lane; }   // ascending j within a lane: first wins ties
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float v2 = __shfl_xor(v, o, 64);
          const int i2 = __shfl_xor(ix, o, 64);
          if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
        }
        if (ix != 0x7fffffff && (ix & 63) == lane) cur |= 1u << (ix >> 6);
        if (lane == 0) idx_out[hyp * nc + c] = ix == 0x7fffffff ? 0 : ix;
      }
    }
    unsigned fin = cur;
    int pre = nc, nround = 0;
    for (int it = 0; it < num_ref; ++it) {
      // masked Procrustes over cur: centroids with w / (sum|w| + 1e-16), covariance with the RAW 0/1 mask (solvers.py:14-26)
      double sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};
      int cl = 0;
#pragma unroll
      for (int q = 0; q < TR_SLOTS; ++q)
        if ((cur >> q) & 1u) {
          const int j = q * 64 + lane;
          ++cl;
#pragma unroll
          for (int a = 0; a < 3; ++a) { sa[a] += sX[j * 3 + a]; sb[a] += sY[j * 3 + a]; }
        }
      const double C = wave_sum_d((double)cl);
      const float wn = 1.0f / ((float)C + 1e-16f);
      double am[3], bm[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        am[a] = wave_sum_d(sa[a]) * (double)wn;
        bm[a] = wave_sum_d(sb[a]) * (double)wn;
      }
      double hl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < TR_SLOTS; ++q)
        if ((cur >> q) & 1u) {
          const int j = q * 64 + lane;
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) hl[a * 3 + c] += ((double)sX[j * 3 + a] - am[a]) * ((double)sY[j * 3 + c] - bm[c]);
        }
      double H[9], Rd[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) H[i] = wave_sum_d(hl[i]);
      kabsch_rotation(H, Rd);
      float Rf[9], tf[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) Rf[i] = (float)Rd[i];
#pragma unroll
      for (int a = 0; a < 3; ++a) tf[a] = (float)bm[a] - ((float)am[0] * Rf[a * 3 + 0] + (float)am[1] * Rf[a * 3 + 1] + (float)am[2] * Rf[a * 3 + 2]);
      unsigned ref = 0;
      int rl = 0;
#pragma unroll
      for (int q = 0; q < TR_SLOTS; ++q) {
        const int j = q * 64 + lane;
        if (j < S && th_ref - pt_dist(Rf, tf, sX + j * 3, sY + j * 3) >= 0.f) { ref |= 1u << q; ++rl; }
      }
      const int cnt = (int)wave_sum((float)rl);
      if (!(cnt > pre)) break;   // wave-uniform
      pre = cnt;
      fin = cur;
      cur = ref;
      ++nround;
    }
#pragma unroll
    for (int q = 0; q < TR_SLOTS; ++q) {
      const int j = q * 64 + lane;
      if (j < S) fin_mask[hyp * S + j] = (fin >> q) & 1u ? 1.f : 0.f;
    }
    if (lane == 0) rounds_out[hyp] = nround;
  }
}

// REINFORCE bookkeeping (loss_class.py:251-261): gradients[b, c] += loss_value[row], gradients_b[b, c] += 1 for the S
// sampled cells c of every row of pair b, ROW AFTER ROW like the reference's python loop (a cell drawn by several rows
// receives its fp32 additions in the same order: bit-identical sums).  Cells within a row are distinct (sampling without
// replacement), so a row is a plain read-modify-write; one workgroup per pair, a barrier between rows.
__global__ __launch_bounds__(512) void reinforce_scatter_kernel(const int* __restrict__ idx, const float* __restrict__ loss_value,
                                                                float* __restrict__ grads, float* __restrict__ grads_b,
                                                                int it_matches, int S, long long ncell) {
  const int b = blockIdx.x;
  float* g = grads + (long long)b * ncell;
  float* gb = grads_b + (long long)b * ncell;
  for (int r = 0; r < it_matches; ++r) {
    const long long row = (long long)b * it_matches + r;
    const float lv = loss_value[row];
    for (int s = threadIdx.x; s < S; s += 512) {
      const int c = idx[row * S + s];
      g[c] += lv;
      gb[c] += 1.0f;
    }
    __syncthreads();
  }
}

__global__ void finalize_kernel(float* R, float* t, float* conf, const int* invalid, int B) {
  if (*invalid == 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 9) R[i] = 0.f;
  if (i < B * 3) t[i] = 0.f;
  if (i < B) conf[i] = 0.f;
}

int g_exprace_mode = 0;   // dev (mk_exprace_set_mode): 0 = skip sampler, 1 = the 6-bit pre-filter pass

// bytes of the self-cleaning state at the head of the workspace (TopkWork)
long long topk_state_bytes(long long R, long long B) { return (R * 4 + B * 4 + B * NBINS * 4 + B * 4 + 15) / 16 * 16; }

TopkWork carve(void* work, int R, int B, long long ncell) {
  TopkWork w;
  char* p = (char*)work;
  w.ncand = (unsigned*)p; p += (size_t)R * 4;
  w.redo = (int*)p;       p += (size_t)B * 4;
  w.phist = (unsigned*)p; p += (size_t)B * NBINS * 4;
  w.done1 = (unsigned*)p; p += (size_t)B * 4;
  p = (char*)work + topk_state_bytes(R, B);
  w.thr = (int*)p;        p += ((size_t)R * 4 + 15) / 16 * 16;
  w.cand = (unsigned long long*)p;  p += (size_t)R * CAND_MAX * 8;
  w.nblk = (ncell + SK_CELLS - 1) / SK_CELLS;
  w.pmax = (float*)p;
  return w;
}

}  // namespace

extern "C" {

long long mk_exprace_topk_work_bytes(int B, int rows_per_pair, int k, long long ncell) {
  (void)k;
  const long long R = (long long)B * rows_per_pair;
  return topk_state_bytes(R, B) + (R * 4 + 15) / 16 * 16 + R * CAND_MAX * 8 + (long long)B * ((ncell + SK_CELLS - 1) / SK_CELLS) * 4;
}
long long mk_exprace_topk_state_bytes(int B, int rows_per_pair) { return topk_state_bytes((long long)B * rows_per_pair, B); }
int mk_exprace_set_mode(int mode) {
  MK_CHECK_ARG(mode == 0 || mode == 1, "mk_exprace_set_mode: 0 (skip sampler) or 1 (pre-filter pass)");
  g_exprace_mode = mode;
  return MK_OK;
}

int mk_counter_add(unsigned long long* counter, unsigned long long inc, mk_stream_t stream) {
  MK_CHECK_ARG(counter, "mk_counter_add: null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, inc);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_exprace_topk(const float* p, const float* noise, unsigned long long seed, unsigned long long offset,
                    const unsigned long long* offset_dev, int* idx, int* cnt, int* invalid, void* work, int B, int rows_per_pair,
                    long long ncell, int k, int pair_base, mk_stream_t stream) {
  MK_CHECK_ARG(p && idx && cnt && work, "mk_exprace_topk: null pointer");
  MK_CHECK_ARG(B > 0 && rows_per_pair > 0 && rows_per_pair <= 64 * RG && ncell > 0 && ncell < (1LL << 31) && k > 0 &&
                   k <= CAND_MAX / 2,
               "mk_exprace_topk: bad sizes (k <= %d, ncell < 2^31)", CAND_MAX / 2);
  MK_CHECK_ARG(((uintptr_t)work & 15) == 0, "mk_exprace_topk: work must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int R = B * rows_per_pair;
  TopkWork w = carve(work, R, B, ncell);
  w.invalid = invalid;
  w.pair_base = pair_base;
  const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32), ol = (unsigned)offset, oh = (unsigned)(offset >> 32);
  const int groups = (rows_per_pair + RG - 1) / RG;
  // cell blocks per pair: 128 at the bench batch; small batches (one pair: 128 workgroups on a 256-CU part, each walking 29 k
  // cells) take more, so that B * cb fills the chip a few times over.  The candidate SET does not depend on the split (the
  // select kernel orders it), so a pair's draws do not depend on the batch it is in.
  int cb = CELL_BLOCKS;
  while (cb < 1024 && (long long)B * cb < 2048) cb *= 2;
  if ((long long)cb * 256 > ncell) cb = (int)((ncell + 255) / 256);
  // The chain is FOUR launches (round 6; ten before): histogram of p + the 16-cell maxima with the analytic threshold as its
  // tail -> ONE noise pass (collect) -> the exact fallback (its workgroups check their pair's rows and leave unless the pair
  // came up short) -> select.  No zero-fill: the state words clean themselves (TopkWork).
  hipLaunchKernelGGL(exprace_phist_kernel, dim3(cb, B), dim3(256), 0, st, p, w, ncell, rows_per_pair, 1.25f * (float)k);
  MK_CHECK_LAUNCH();
  // the block-local cell index must fit 16 bits of a queue entry: more cell blocks for very large matrices
  int pcb = cb;
  while ((ncell + pcb - 1) / pcb > 65536) pcb *= 2;
  if (!noise && rows_per_pair <= PF_MAXROWS && g_exprace_mode == 0) {
    // (the walk geometry is a function of ncell alone: a pair's draws do not depend on the batch)
    hipLaunchKernelGGL(exprace_skip_kernel, dim3((unsigned)((w.nblk + SK_RANGE - 1) / SK_RANGE), (rows_per_pair + SK_ROWS - 1) / SK_ROWS, B),
                       dim3(256), 0, st, p, k0, k1, ol, oh, offset_dev, w, rows_per_pair, ncell);
  } else {
    if (!noise && rows_per_pair <= PF_MAXROWS) {
      const size_t lds = (size_t)rows_per_pair * PF_LCAP * 8 + (size_t)PF_QCAP * 4;
      hipLaunchKernelGGL(exprace_prefilter_kernel, dim3(pcb, B), dim3(256), lds, st, p, k0, k1, ol, oh, offset_dev, w, rows_per_pair,
                         ncell);
    } else {
      hipLaunchKernelGGL(exprace_scan_kernel, dim3(cb, groups, B), dim3(256), 0, st, p, noise, k0, k1, ol, oh, offset_dev, w, rows_per_pair, ncell);
    }
  }
  MK_CHECK_LAUNCH();
  // exact fallback (runs only if a row came up short: never observed on a matcher's output, kept for adversarial inputs / injected noise)
  hipLaunchKernelGGL(exprace_fallback_kernel, dim3(groups, B), dim3(1024), 0, st, p, noise, k0, k1, ol, oh, offset_dev, w, rows_per_pair,
                     ncell, k);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(exprace_select_kernel, dim3(R), dim3(SEL_T), (size_t)SEL_LDS * 8, st, p, w, idx, cnt, rows_per_pair, ncell,
                     k);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_gather_backproject(const int* idx, const float* final_scores, const float* kps0, const float* depth0, const float* kps1,
                          const float* depth1, const float* K0, const float* K1, float* X, float* Y, float* wts, float* corr,
                          int B, int rows_per_pair, int k, int n0, int n1, mk_stream_t stream) {
  MK_CHECK_ARG(idx && final_scores && kps0 && depth0 && kps1 && depth1 && K0 && K1 && X && Y && wts && corr,
               "mk_gather_backproject: null pointer");
  MK_CHECK_ARG(B > 0 && rows_per_pair > 0 && k > 0 && n0 > 0 && n1 > 0, "mk_gather_backproject: bad sizes");
  hipLaunchKernelGGL(gather_backproject_kernel, dim3((k + 255) / 256, B * rows_per_pair), dim3(256), 0, (hipStream_t)stream, idx,
                     final_scores, kps0, depth0, kps1, depth1, K0, K1, X, Y, wts, corr, rows_per_pair, k, n0, n1);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_gather_backproject_bwd(const int* idx, const float* corr, const float* gX, const float* gY, const float* K0, const float* K1,
                              float* gkps0, float* gdepth0, float* gkps1, float* gdepth1, int B, int rows_per_pair, int k, int n0,
                              int n1, mk_stream_t stream) {
  MK_CHECK_ARG(idx && corr && gX && gY && K0 && K1 && gkps0 && gdepth0 && gkps1 && gdepth1, "mk_gather_backproject_bwd: null pointer");
  MK_CHECK_ARG(B > 0 && rows_per_pair > 0 && k > 0 && n0 > 0 && n1 > 0, "mk_gather_backproject_bwd: bad sizes");
  hipLaunchKernelGGL(gather_backproject_bwd_kernel, dim3((k + 255) / 256, B * rows_per_pair), dim3(256), 0, (hipStream_t)stream, idx,
                     corr, gX, gY, K0, K1, gkps0, gdepth0, gkps1, gdepth1, rows_per_pair, k, n0, n1);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_ransac_hypotheses(const float* X, const float* Y, const float* wts, const float* noise3, const int* idx3_in,
                         unsigned long long seed, unsigned long long offset, const unsigned long long* offset_dev, float th_soft,
                         float* Rh, float* th, float* score, int* idx3, int nsets, int it_ransac, int k, long long set_base,
                         mk_stream_t stream) {
  MK_CHECK_ARG(X && Y && wts && Rh && th && score && idx3, "mk_ransac_hypotheses: null pointer");
  MK_CHECK_ARG(nsets > 0 && it_ransac > 0 && k >= 3 && (size_t)k * 32 + 1024 <= 150 * 1024, "mk_ransac_hypotheses: bad sizes (k <= 4768)");
  // blocks per correspondence set: 4 (25 hypotheses per block, 6 - 7 per wave) when the sets alone fill the part; with few sets
  // (one pair: 20) as many as keep every wave at >= one hypothesis and the launch at ~2 blocks per CU -- a hypothesis' arithmetic and
  // draws do not depend on the block it is computed in (keyed by its global index): bit-identical for any split
  int nsplit = it_ransac >= 16 ? 4 : 1;
  if (it_ransac >= 16) nsplit = max(4, min((it_ransac + 3) / 4, (2 * mk::gemm::num_cus() + nsets - 1) / nsets));
  const size_t lds = ((size_t)k * 8 + 256) * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)hypotheses_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { mk_set_error("mk_ransac_hypotheses: cannot reserve %zu B of LDS", lds); return MK_ERR_LAUNCH; }
  }
  hipLaunchKernelGGL(hypotheses_kernel, dim3(nsets * nsplit), dim3(256), lds, (hipStream_t)stream, X, Y, wts, noise3, idx3_in,
                     (unsigned)seed, (unsigned)(seed >> 32), (unsigned)offset, (unsigned)(offset >> 32), offset_dev, th_soft, Rh, th,
                     score, idx3, it_ransac, k, nsplit, set_base);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_refine_pose(const float* X, const float* Y, const float* Rh, const float* th, const float* score, float th_inlier,
                   int num_ref, int min_inliers, float* R, float* t, float* conf, int* best, unsigned char* inl_mask, int* rounds,
                   int* invalid, int B, int it_matches, int it_ransac, int k, mk_stream_t stream) {
  MK_CHECK_ARG(X && Y && Rh && th && score && R && t && conf && best && inl_mask && rounds && invalid,
               "mk_refine_pose: null pointer");
  MK_CHECK_ARG(B > 0 && it_matches > 0 && it_ransac > 0 && k > 0 && num_ref >= 0, "mk_refine_pose: bad sizes");
  hipLaunchKernelGGL(refine_kernel, dim3(B), dim3(RT), 0, (hipStream_t)stream, X, Y, Rh, th, score, th_inlier, num_ref,
                     min_inliers, R, t, conf, best, inl_mask, rounds, invalid, it_matches, it_ransac, k);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_pose_finalize(float* R, float* t, float* conf, const int* invalid, int B, mk_stream_t stream) {
  MK_CHECK_ARG(R && t && conf && invalid && B > 0, "mk_pose_finalize: bad args");
  hipLaunchKernelGGL(finalize_kernel, dim3((B * 9 + 255) / 256), dim3(256), 0, (hipStream_t)stream, R, t, conf, invalid, B);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_train_ransac_masks(const float* X, const float* Y, const float* wts, const float* noise, const int* idx_in,
                          unsigned long long seed, unsigned long long offset, const unsigned long long* offset_dev, float th_ref,
                          int num_ref, int num_corr, float* final_mask, int* idx_out, int* rounds, int nsets, int it_ransac,
                          int S, long long set_base, mk_stream_t stream) {
  MK_CHECK_ARG(X && Y && wts && final_mask && idx_out && rounds, "mk_train_ransac_masks: null pointer");
  MK_CHECK_ARG(nsets > 0 && it_ransac > 0 && S > 0 && S <= 64 * TR_SLOTS && num_corr >= 3 && num_corr <= S && num_ref >= 0,
               "mk_train_ransac_masks: bad sizes (3 <= num_corr <= S <= %d)", 64 * TR_SLOTS);
  const size_t lds = (size_t)S * 7 * sizeof(float);
  hipLaunchKernelGGL(train_refine_kernel, dim3(nsets), dim3(256), lds, (hipStream_t)stream, X, Y, wts, noise, idx_in,
                     (unsigned)seed, (unsigned)(seed >> 32), (unsigned)offset, (unsigned)(offset >> 32), offset_dev, th_ref,
                     num_ref, num_corr, final_mask, idx_out, rounds, it_ransac, S, set_base);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_reinforce_scatter(const int* idx, const float* loss_value, float* gradients, float* gradients_b, int B, int it_matches,
                         int S, long long ncell, mk_stream_t stream) {
  MK_CHECK_ARG(idx && loss_value && gradients && gradients_b, "mk_reinforce_scatter: null pointer");
  MK_CHECK_ARG(B > 0 && it_matches > 0 && S > 0 && ncell > 0 && ncell < (1LL << 31), "mk_reinforce_scatter: bad sizes");
  hipLaunchKernelGGL(reinforce_scatter_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, idx, loss_value, gradients,
                     gradients_b, it_matches, S, ncell);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // extern "C"
