// mickey_amd -- epilogue of the one-wave-per-SIMD GEMM schedules (32x32x16 accumulator layout), shared by
// mk_gemm_w4.hip and mk_gemm_w4k32.hip.
#pragma once
#include "mk_gemm_common.hpp"

namespace mk {
namespace gemm {

// ---------------------------------------------------------------------------------------------------------
// Epilogue of the 32x32x16 accumulator layout.  acc[mi][ni] (f32x16) covers rows m = mi*32 + (lane & 31) and features
// ni*32 + 8*(e >> 2) + 4*(lane >> 5) + (e & 3): per register quad a lane owns 4 CONSECUTIVE features of one row.
// After the K loop the 128-KiB ring is idle and each wave owns 32 KiB of it: a 32-row x 128-feature slab (mi) of RAW fp32
// accumulators is bounced through LDS (wave-local, LDS is in order per wave: no barrier; rows of 512 B, 16-byte chunk c
// of row r at c ^ r) and leaves as whole rows.  All arithmetic (bias, activation, LayerScale + residual, q scale, identity
// residual, 16-bit conversion) happens on the way out, where a lane owns the SAME features in every row it touches -- its
// bias / gamma are 1-2 registers instead of a 64-register table -- and every global access is a full-row access: fp32
// outputs 512 B per row per half-wave, 16-bit outputs 256 B per row per 16 lanes.  Two slab buffers alternate so the
// write pass of slab mi+1 can be scheduled under the stores of slab mi.
template <typename T, int EPI, int ACT, bool HAS_BIAS>
__device__ __forceinline__ void epilogue32_impl(const GemmParams& p, f32x16 (&acc)[4][4], char* wl, int m0, int n0, int wm, int wn,
                                                int lane, int g) {
  using V4 = typename Lp<T>::V4;
  using V8 = typename Lp<T>::V8;
  const int r32 = lane & 31, hi = lane >> 5;
  const float* bias = HAS_BIAS ? p.bias + (long long)g * p.strideBias_g : nullptr;
  const int nw = n0 + wn * 128;   // first feature of this wave's block
  const int mw = m0 + wm * 128;   // first row
  int img0 = 0, tok0 = 0;
  if (EPI == MK_EPI_QKV) {
    img0 = mw / p.ntok;
    tok0 = mw - img0 * p.ntok;
  }
  auto img_tok = [&](int r, int& img, int& tok) {   // row mw + r: one wave-uniform division, at most one image boundary
    if (p.ntok >= 128) {
      const int t = tok0 + r;
      const bool wrap = t >= p.ntok;
      img = img0 + (wrap ? 1 : 0);
      tok = wrap ? t - p.ntok : t;
    } else {
      const int m = mw + r;
      img = m / p.ntok;
      tok = m - img * p.ntok;
    }
  };
  auto quad = [&](int mi, int ni, int gq) {
    return f32x4{acc[mi][ni][gq * 4], acc[mi][ni][gq * 4 + 1], acc[mi][ni][gq * 4 + 2], acc[mi][ni][gq * 4 + 3]};
  };
  auto stage = [&](int mi) {
    char* buf = wl + (mi & 1) * 16384;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int cw = ni * 8 + gq * 2 + hi;
        *(f32x4*)(buf + r32 * 512 + ((cw ^ r32) << 4)) = quad(mi, ni, gq);
      }
  };
  const bool lp_out = EPI == MK_EPI_QKV || (EPI == MK_EPI_STORE && !p.out_f32);
  if (lp_out) {
    int which = 0, head0 = 0;
    if (EPI == MK_EPI_QKV) {
      const int D = p.heads * 64;
      which = nw / D;
      head0 = (nw - which * D) >> 6;
      if (which == 2) {   // V^T, key-permuted: element stores straight from the accumulators
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int m = mw + mi * 32 + r32;
          if (m >= p.M) continue;
          int img, tok;
          img_tok(mi * 32 + r32, img, tok);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const long long hb = (long long)img * p.heads + head0 + (ni >> 1);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int nf = ni * 32 + gq * 8 + hi * 4;
              const f32x4 v = quad(mi, ni, gq) + *(const f32x4*)(bias + nw + nf);
              T* dst = (T*)p.vt + (hb * 64 + (nf & 63)) * p.ntok_pad + vperm(tok);
#pragma unroll
              for (int e = 0; e < 4; ++e) dst[(long long)e * p.ntok_pad] = to_lp<T>(v[e]);
            }
          }
        }
        return;
      }
    }
    const int rr = lane >> 4, c = lane & 15;
    const int n = nw + c * 8;
    f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (HAS_BIAS && n < p.N) {
      b0 = *(const f32x4*)(bias + n);
      if (n + 8 <= p.N) b1 = *(const f32x4*)(bias + n + 4);
    }
    auto drain = [&](int mi) {
      const char* buf = wl + (mi & 1) * 16384;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 4 + rr;
        const int m = mw + mi * 32 + r;
        f32x4 v0 = *(const f32x4*)(buf + r * 512 + (((2 * c) ^ r) << 4));
        f32x4 v1 = *(const f32x4*)(buf + r * 512 + (((2 * c + 1) ^ r) << 4));
        if (m >= p.M || n >= p.N) continue;
        v0 += b0;
        v1 += b1;
        T* dst;
        if (EPI == MK_EPI_QKV) {
          if (which == 0) {
            v0 *= p.qscale;
            v1 *= p.qscale;
          }
          int img, tok;
          img_tok(mi * 32 + r, img, tok);
          dst = (T*)(which == 0 ? p.q : p.k) + (((long long)img * p.heads + head0 + (c >> 3)) * p.ntok_pad + tok) * 64 + (c & 7) * 8;
        } else {
          const long long off = (long long)g * p.strideOut_g + (long long)m * p.ldc + n;
          if (p.resid_lp) {
            if (n + 8 <= p.N) {
              const V8 rs = *(const V8*)((const T*)p.resid_lp + off);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v0[e] += (float)rs[e];
                v1[e] += (float)rs[4 + e];
              }
            } else {
              const V4 rs = *(const V4*)((const T*)p.resid_lp + off);
#pragma unroll
              for (int e = 0; e < 4; ++e) v0[e] += (float)rs[e];
            }
          }
          if (ACT == MK_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v0[e] = fmaxf(v0[e], 0.f);
              v1[e] = fmaxf(v1[e], 0.f);
            }
          } else if (ACT == MK_ACT_GELU) {
            v0 = gelu_erf4(v0);
            v1 = gelu_erf4(v1);
          }
          dst = (T*)p.out_lp + off;
        }
        if (n + 8 <= p.N) {
          V8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = to_lp<T>(v0[e]);
            o[4 + e] = to_lp<T>(v1[e]);
          }
          *(V8*)dst = o;
        } else {   // N % 8 == 4: the last chunk is half wide
          V4 lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) lo[e] = to_lp<T>(v0[e]);
          *(V4*)dst = lo;
        }
      }
    };
    stage(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if (mi + 1 < 4) stage(mi + 1);
      drain(mi);
    }
  } else {
    const int c = lane & 31;
    const int n = nw + c * 4;
    f32x4 gm = f32x4{0.f, 0.f, 0.f, 0.f}, b4 = gm;
    if (EPI == MK_EPI_LS_RESIDUAL && n < p.N) gm = *(const f32x4*)(p.gamma + n);
    if (HAS_BIAS && n < p.N) b4 = *(const f32x4*)(bias + n);
    // read-modify-write of the residual stream: the 16 loads of a slab are issued before anything waits on them, the
    // next slab's loads go out while this one is being stored (one HBM round trip per slab)
    f32x4 xr[2][16];
    auto preload = [&](int mi) {
      if (EPI != MK_EPI_LS_RESIDUAL) return;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int m = mw + mi * 32 + it * 2 + hi;
        xr[mi & 1][it] = (m < p.M && n < p.N) ? *(const f32x4*)(p.out_f32 + (long long)m * p.ldc + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto drain = [&](int mi) {
      const char* buf = wl + (mi & 1) * 16384;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r = it * 2 + hi;
        const int m = mw + mi * 32 + r;
        f32x4 val = *(const f32x4*)(buf + r * 512 + ((c ^ r) << 4));
        if (m >= p.M || n >= p.N) continue;
        val += b4;
        if (EPI == MK_EPI_LS_RESIDUAL) {
          f32x4 x = xr[mi & 1][it];
          x += gm * val;
          *(f32x4*)(p.out_f32 + (long long)m * p.ldc + n) = x;
        } else if (EPI == MK_EPI_PATCH) {
          const int img = m / p.npatch, tok = m - img * p.npatch;
          const f32x4 pe = *(const f32x4*)(p.pos + (long long)(1 + tok) * p.N + n);
          *(f32x4*)(p.out_f32 + ((long long)img * (p.npatch + 1) + 1 + tok) * p.ldc + n) = val + pe;
        } else {
          const long long off = (long long)g * p.strideOut_g + (long long)m * p.ldc + n;
          if (p.resid_lp) {
            const V4 rs = *(const V4*)((const T*)p.resid_lp + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] += (float)rs[e];
          }
          if (ACT == MK_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
          } else if (ACT == MK_ACT_GELU) {
            val = gelu_erf4(val);
          }
          *(f32x4*)(p.out_f32 + off) = val;
        }
      }
    };
    preload(0);
    stage(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if (mi + 1 < 4) {
        preload(mi + 1);
        stage(mi + 1);
      }
      drain(mi);
    }
  }
}

template <typename T>
__device__ __forceinline__ void epilogue32(const GemmParams& p, f32x16 (&acc)[4][4], char* wl, int m0, int n0, int wm, int wn,
                                           int lane, int g) {
  switch (p.epi) {   // wave-uniform, once per output tile
    case MK_EPI_LS_RESIDUAL: epilogue32_impl<T, MK_EPI_LS_RESIDUAL, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    case MK_EPI_QKV: epilogue32_impl<T, MK_EPI_QKV, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    case MK_EPI_PATCH: epilogue32_impl<T, MK_EPI_PATCH, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g); break;
    default:
      if (!p.bias) {
        if (p.act == MK_ACT_RELU) epilogue32_impl<T, MK_EPI_STORE, MK_ACT_RELU, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else if (p.act == MK_ACT_GELU) epilogue32_impl<T, MK_EPI_STORE, MK_ACT_GELU, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else epilogue32_impl<T, MK_EPI_STORE, MK_ACT_NONE, false>(p, acc, wl, m0, n0, wm, wn, lane, g);
      } else {
        if (p.act == MK_ACT_RELU) epilogue32_impl<T, MK_EPI_STORE, MK_ACT_RELU, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else if (p.act == MK_ACT_GELU) epilogue32_impl<T, MK_EPI_STORE, MK_ACT_GELU, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
        else epilogue32_impl<T, MK_EPI_STORE, MK_ACT_NONE, true>(p, acc, wl, m0, n0, wm, wn, lane, g);
      }
  }
}

}  // namespace gemm
}  // namespace mk
