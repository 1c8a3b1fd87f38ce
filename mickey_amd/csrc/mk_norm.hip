// mickey_amd -- row-wise kernels of the encoder: LayerNorm, patch im2col, CLS row.  All HBM-bound:
// one wave per row, 16-byte loads/stores, wave64 shuffle reductions, values held in registers
// between the statistics and the normalisation (single read of x).
#include "mk_common.hpp"
#include "mk_gemm_common.hpp"   // quad_stats / row16_sum: the one summation order of the row statistics

#include <stdarg.h>
#include <stdio.h>

namespace {
using namespace mk;

// out_is_f32 == 2: the row goes out as the (hi, lo) fp16 operand planes of the split-operand head kernels
// (mk_conv3x3_split / mk_gemm_grouped_split): y * scale = hi + lo, saturating at fp16's largest finite value
__device__ __forceinline__ void store_planes(void* hi, void* lo, long long o, const f32x4& y, float scale, int* sat_flag) {
  f16x4 oh, ol;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float sv = sat16(y[e] * scale, sat_flag);
    oh[e] = (_Float16)sv;
    ol[e] = (_Float16)(sv - (float)oh[e]);
  }
  *(f16x4*)((_Float16*)hi + o) = oh;
  if (lo) *(f16x4*)((_Float16*)lo + o) = ol;   // lo == nullptr: the row rounded to fp16 (what an fp16 LayerNorm outputs)
}

constexpr int LN_MAXV = 8;  // float4 per lane -> D <= 2048 (instantiated for 4 as well: D <= 1024 needs half the registers)

constexpr int LN_RPW = 4;   // rows per wave: the loads of row r+1 are in flight while row r is reduced and stored
                            // (a third buffer / 6 rows per wave measured 7 % slower in the forward)

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, void* out, int ldo,
                                                        int out_is_f32, float* resid, int ldr, int rows_out, int D,
                                                        int rows_per_img, int skip, int wgroup_rows, int bord_h, int bord_w,
                                                        int bord_m, void* out_lo, float plane_scale, int* sat_flag) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_RPW;
  if (r0 >= rows_out) return;
  const int rpo = rows_per_img - skip;
  f32x4 v[2][MAXV];
  auto load_row = [&](int r, f32x4 (&dst)[MAXV]) {
    const long long rin = (long long)(r / rpo) * rows_per_img + skip + (r % rpo);
    const float* xr = x + rin * ldx;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) dst[i] = __builtin_nontemporal_load((const f32x4*)(xr + c));   // streamed once: do not displace the weights in L2
    }
  };
  load_row(r0, v[0]);
  // weight / bias shared by all rows (the encoder's case): once per wave, not once per row -- the compiler cannot hoist
  // them itself because `out` may alias them for all it knows (8 of the 12 load instructions per row)
  f32x4 wreg[MAXV], breg[MAXV];
  const bool shared_wb = wgroup_rows <= 0;
  if (shared_wb) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        wreg[i] = *(const f32x4*)(w + c);
        breg[i] = *(const f32x4*)(b + c);
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < LN_RPW; ++rr) {
    const int r = r0 + rr;
    if (r >= rows_out) break;
    if (rr + 1 < LN_RPW && r + 1 < rows_out) load_row(r + 1, v[(rr + 1) & 1]);
    f32x4 (&cur)[MAXV] = v[rr & 1];
    // bordered output (feeds a 3x3 conv, mk_common.hpp): buffers of bord_m pixels each, one behind the other
    long long ro = r;
    if (bord_h > 0) {
      const int gb = r / bord_m, m = r - gb * bord_m;
      ro = gb * bordered_rows(bord_m / (bord_h * bord_w), bord_h, bord_w) + bordered_row(m, bord_h, bord_w);
    }
    const float* wr = w;
    const float* br = b;
    if (wgroup_rows > 0) {  // one (w, b) per block of wgroup_rows output rows
      wr += (long long)(r / wgroup_rows) * D;
      br += (long long)(r / wgroup_rows) * D;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) s += (cur[i][0] + cur[i][1]) + (cur[i][2] + cur[i][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = cur[i][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        const f32x4 ww = shared_wb ? wreg[i] : *(const f32x4*)(wr + c), bb = shared_wb ? breg[i] : *(const f32x4*)(br + c);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (cur[i][e] - mean) * rstd * ww[e] + bb[e];
        if (resid) {
          float* rp = resid + (long long)r * ldr + c;
          y += *(const f32x4*)rp;
          *(f32x4*)rp = y;
        }
        if (out) {
          if (out_is_f32 == 2) {
            store_planes(out, out_lo, ro * ldo + c, y, plane_scale, sat_flag);
          } else if (out_is_f32) {
            *(f32x4*)((float*)out + ro * ldo + c) = y;
          } else {
            typename Lp<T>::V4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (T)y[e];
            *(typename Lp<T>::V4*)((T*)out + ro * ldo + c) = o;
          }
        }
      }
    }
  }
}

// Narrow rows (D <= 128: the 128-channel LayerNorms of the heads' attention layers, 6 of the 7 stand-alone LayerNorms of a
// forward).  In the kernel above such a row occupies half a wave and a wave has one 512-byte row in flight: 2.2 TB/s.
// Here a row is 32 lanes x 4 floats, a wave works on TWO rows at a time and requests all 8 rows it owns (4 trips) before
// it reduces the first.  Same arithmetic per row (sum, then squares about the mean, both over the row's 32 lanes).
constexpr int LNN_TRIPS = 4;   // row pairs per wave
template <typename T>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                               const float* __restrict__ b, float eps, void* out, int ldo,
                                                               int out_is_f32, float* resid, int ldr, int rows_out, int D,
                                                               int rows_per_img, int skip, int wgroup_rows, int bord_h,
                                                               int bord_w, int bord_m, void* out_lo, float plane_scale, int* sat_flag) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = (lane & 31) * 4;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (2 * LNN_TRIPS);
  if (r0 >= rows_out) return;
  const bool cok = c < D;
  const int rpo = rows_per_img - skip;
  f32x4 v[LNN_TRIPS];
#pragma unroll
  for (int t = 0; t < LNN_TRIPS; ++t) {
    const int r = r0 + 2 * t + half;
    v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < rows_out && cok) {
      const long long rin = (long long)(r / rpo) * rows_per_img + skip + (r % rpo);
      v[t] = __builtin_nontemporal_load((const f32x4*)(x + rin * ldx + c));
    }
  }
  auto half_sum = [](float a) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    return a;
  };
#pragma unroll
  for (int t = 0; t < LNN_TRIPS; ++t) {
    const int r = r0 + 2 * t + half;
    const bool ok = r < rows_out && cok;
    const float mean = half_sum((v[t][0] + v[t][1]) + (v[t][2] + v[t][3])) / (float)D;
    f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cok) d = v[t] - mean;
    const float rstd = 1.0f / sqrtf(half_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) / (float)D + eps);
    if (!ok) continue;
    const long long wo = wgroup_rows > 0 ? (long long)(r / wgroup_rows) * D : 0;
    const f32x4 ww = *(const f32x4*)(w + wo + c), bb = *(const f32x4*)(b + wo + c);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = d[e] * rstd * ww[e] + bb[e];
    if (resid) {
      float* rp = resid + (long long)r * ldr + c;
      y += *(const f32x4*)rp;
      *(f32x4*)rp = y;
    }
    if (out) {
      long long ro = r;
      if (bord_h > 0) {
        const int gb = r / bord_m, m = r - gb * bord_m;
        ro = gb * bordered_rows(bord_m / (bord_h * bord_w), bord_h, bord_w) + bordered_row(m, bord_h, bord_w);
      }
      if (out_is_f32 == 2) {
        store_planes(out, out_lo, ro * ldo + c, y, plane_scale, sat_flag);
      } else if (out_is_f32) {
        *(f32x4*)((float*)out + ro * ldo + c) = y;
      } else {
        typename Lp<T>::V4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (T)y[e];
        *(typename Lp<T>::V4*)((T*)out + ro * ldo + c) = o;
      }
    }
  }
}

// one thread per 4 output columns (k = ch*196 + dy*14 + dx); rows = patches
template <typename T>
__global__ __launch_bounds__(256) void im2col14_kernel(const float* __restrict__ img, long long stride_img,
                                                       long long stride_ch, int stride_row, int nimg, int gh, int gw,
                                                       T* __restrict__ out, int ldo) {
  const int cols4 = ldo >> 2;
  const long long total = (long long)nimg * gh * gw * cols4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cols4);
    const long long m = i / cols4;
    const int pc = (int)(m % gw);
    const int pr = (int)((m / gw) % gh);
    const int im = (int)(m / ((long long)gw * gh));
    typename Lp<T>::V4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = c4 * 4 + e;
      float v = 0.f;
      if (k < 588) {
        const int ch = k / 196, rem = k - ch * 196, dy = rem / 14, dx = rem - dy * 14;
        v = img[im * stride_img + ch * stride_ch + (long long)(pr * 14 + dy) * stride_row + pc * 14 + dx];
      }
      o[e] = (T)v;
    }
    *(typename Lp<T>::V4*)(out + m * ldo + c4 * 4) = o;
  }
}

__global__ void cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x, int nimg,
                           int ntok, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nimg * D) return;
  const int im = i / D, c = i - im * D;
  x[(long long)im * ntok * D + c] = cls[c] + pos[c];
}

// CLS row of the split residual stream (hi / lo planes) with its slot statistics: one wave per image; lane l owns columns
// c = l, l + 64, ...: slot c / 64 of the row statistics is a plain wave sum.
template <typename T>
__global__ __launch_bounds__(64) void cls_ln_kernel(const float* __restrict__ cls, const float* __restrict__ pos, T* __restrict__ xh,
                                                    T* __restrict__ xl, float* __restrict__ stats, int ntok, int D) {
  const int im = blockIdx.x, lane = threadIdx.x;
  const long long row = (long long)im * ntok;
  for (int c0 = 0; c0 < D; c0 += 64) {
    const float v = cls[c0 + lane] + pos[c0 + lane];
    const T h = (T)v;
    xh[row * D + c0 + lane] = h;
    xl[row * D + c0 + lane] = (T)(v - (float)h);
    const float s = mk::wave_sum(v), q = mk::wave_sum(v * v);
    if (lane == 0) ((float2*)stats)[row * (D / 64) + (c0 >> 6)] = make_float2(s, q);
  }
}

// Row centring of a freshly produced split stream (mickey_hip.h, mk_recentre_split): every row gets its own mean taken off
// -- LayerNorm, the stream's only reader, cannot tell -- so that the FIRST consumer of the folded LayerNorm (qkv of block 0, on
// the patch embedding's output) also multiplies rows whose 16-bit hi plane rounds x - mean rather than x; from then on the
// consumers publish the means and the producers keep the rows centred (GemmParams::ln_shift_*).  One wave per row, two reads
// of the row (the second one hits L1 / L2); statistics in the epilogues' order (quad_stats + row16_sum).
template <typename T>
__global__ __launch_bounds__(256) void recentre_kernel(T* __restrict__ xh, T* __restrict__ xl, float* __restrict__ stats,
                                                       long long rows, int D) {
  using V4 = typename mk::Lp<T>::V4;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nslot = D >> 6;
  T* ph = xh + row * D;
  T* pl = xl + row * D;
  const int quad = lane & 15, sub = lane >> 4;   // a 16-lane DPP row owns one 64-column slot at a time
  float sum = 0.f;
  for (int s0 = 0; s0 < nslot; s0 += 4) {
    const int slot = s0 + sub;
    if (slot < nslot) {
      const V4 h = *(const V4*)(ph + slot * 64 + quad * 4), l = *(const V4*)(pl + slot * 64 + quad * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += (float)h[e] + (float)l[e];
    }
  }
  const float mean = mk::wave_sum(sum) / (float)D;
  for (int s0 = 0; s0 < nslot; s0 += 4) {
    const int slot = s0 + sub;
    const bool ok = slot < nslot;
    f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
      const V4 h = *(const V4*)(ph + slot * 64 + quad * 4), l = *(const V4*)(pl + slot * 64 + quad * 4);
      V4 oh, ol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[e] = ((float)h[e] + (float)l[e]) - mean;
        oh[e] = (T)x[e];
        ol[e] = (T)(x[e] - (float)oh[e]);
      }
      *(V4*)(ph + slot * 64 + quad * 4) = oh;
      *(V4*)(pl + slot * 64 + quad * 4) = ol;
    }
    float s, q;
    mk::gemm::quad_stats(x, s, q);
    s = mk::gemm::row16_sum(s);
    q = mk::gemm::row16_sum(q);
    if (ok && quad == 0) ((float2*)stats)[row * nslot + slot] = make_float2(s, q);
  }
}

}  // namespace

// ---- error string + version (host side) ---------------------------------------------------------------
static thread_local char g_err[512] = "";
void mk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int mk_version(void) { return 100; }

const char* mk_last_error(void) { return g_err; }

static int layernorm_launch(const float* x, int ldx, const float* w, const float* b, float eps, void* out, int ldo, int out_is_f32,
                            float* resid, int ldr, int rows_out, int D, int rows_per_img, int skip, int wgroup_rows, int bord_h,
                            int bord_w, int bord_m, int dtype, void* out_lo, float plane_scale, int* sat_flag, mk_stream_t stream) {
  MK_CHECK_ARG(x && w && b && (out || resid), "mk_layernorm: null pointer");
  MK_CHECK_ARG(bord_h == 0 || (bord_h > 0 && bord_w > 0 && bord_m > 0 && bord_m % (bord_h * bord_w) == 0 && rows_out % bord_m == 0),
               "mk_layernorm: bordered output needs rows_out = k * bord_m, bord_m = nimg * bord_h * bord_w");
  MK_CHECK_ARG(D > 0 && D % 4 == 0 && D <= LN_MAXV * 256, "mk_layernorm: D=%d must be a multiple of 4 and <= %d", D,
               LN_MAXV * 256);
  MK_CHECK_ARG(rows_out > 0 && rows_per_img > skip && skip >= 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldr % 4 == 0,
               "mk_layernorm: bad geometry");
  if (D <= 128) {
    dim3 gridn((rows_out + 8 * LNN_TRIPS - 1) / (8 * LNN_TRIPS));
#define MK_LNN(T_)                                                                                                         \
  hipLaunchKernelGGL((layernorm_narrow_kernel<T_>), gridn, dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, eps, out, ldo, \
                     out_is_f32, resid, ldr, rows_out, D, rows_per_img, skip, wgroup_rows, bord_h, bord_w, bord_m, out_lo, plane_scale, sat_flag)
    if (dtype == MK_BF16) MK_LNN(__bf16);
    else if (dtype == MK_F16) MK_LNN(_Float16);
    else MK_LNN(float);
#undef MK_LNN
    MK_CHECK_LAUNCH();
    return MK_OK;
  }
  dim3 grid((rows_out + 4 * LN_RPW - 1) / (4 * LN_RPW));
#define MK_LN(T_, V_)                                                                                                  \
  hipLaunchKernelGGL((layernorm_kernel<T_, V_>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, eps, out, ldo, \
                     out_is_f32, resid, ldr, rows_out, D, rows_per_img, skip, wgroup_rows, bord_h, bord_w, bord_m, out_lo, plane_scale, sat_flag)
  if (dtype == MK_BF16) {
    if (D <= 1024) MK_LN(__bf16, 4); else MK_LN(__bf16, LN_MAXV);
  } else if (dtype == MK_F16) {
    if (D <= 1024) MK_LN(_Float16, 4); else MK_LN(_Float16, LN_MAXV);
  } else {
    if (D <= 1024) MK_LN(float, 4); else MK_LN(float, LN_MAXV);
  }
#undef MK_LN
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_layernorm(const float* x, int ldx, const float* w, const float* b, float eps, void* out, int ldo, int out_is_f32,
                 float* resid, int ldr, int rows_out, int D, int rows_per_img, int skip, int wgroup_rows, int bord_h,
                 int bord_w, int bord_m, int dtype, mk_stream_t stream) {
  return layernorm_launch(x, ldx, w, b, eps, out, ldo, out_is_f32 ? 1 : 0, resid, ldr, rows_out, D, rows_per_img, skip, wgroup_rows,
                          bord_h, bord_w, bord_m, dtype, nullptr, 1.0f, nullptr, stream);
}

int mk_layernorm_planes(const float* x, int ldx, const float* w, const float* b, float eps, void* out_hi, void* out_lo, int ldo,
                        float plane_scale, float* resid, int ldr, int rows_out, int D, int rows_per_img, int skip, int wgroup_rows,
                        int bord_h, int bord_w, int bord_m, int* sat_flag, mk_stream_t stream) {
  MK_CHECK_ARG(out_hi, "mk_layernorm_planes: null plane pointer");
  return layernorm_launch(x, ldx, w, b, eps, out_hi, ldo, 2, resid, ldr, rows_out, D, rows_per_img, skip, wgroup_rows, bord_h, bord_w,
                          bord_m, MK_F32, out_lo, plane_scale, sat_flag, stream);
}

int mk_im2col_patch14(const float* img, long long stride_img, long long stride_ch, int stride_row, int nimg, int gh,
                      int gw, void* out, int ldo, int dtype, mk_stream_t stream) {
  MK_CHECK_ARG(img && out && nimg > 0 && gh > 0 && gw > 0, "mk_im2col_patch14: bad args");
  MK_CHECK_ARG(ldo >= 588 && ldo % 8 == 0, "mk_im2col_patch14: ldo=%d must be >= 588 and a multiple of 8", ldo);
  const long long total = (long long)nimg * gh * gw * (ldo / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  if (dtype == MK_BF16)
    hipLaunchKernelGGL(im2col14_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, stride_img, stride_ch,
                       stride_row, nimg, gh, gw, (__bf16*)out, ldo);
  else if (dtype == MK_F16)
    hipLaunchKernelGGL(im2col14_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, stride_img,
                       stride_ch, stride_row, nimg, gh, gw, (_Float16*)out, ldo);
  else
    hipLaunchKernelGGL(im2col14_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, stride_img,
                       stride_ch, stride_row, nimg, gh, gw, (float*)out, ldo);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_cls_token(const float* cls, const float* pos, float* x, int nimg, int ntok, int D, mk_stream_t stream) {
  MK_CHECK_ARG(cls && pos && x && nimg > 0 && ntok > 0 && D > 0, "mk_cls_token: bad args");
  hipLaunchKernelGGL(cls_kernel, dim3((nimg * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls, pos, x, nimg, ntok, D);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

/* the CLS rows of the split residual stream (hi / lo planes) + their per-slot statistics */
int mk_cls_token_ln(const float* cls, const float* pos, void* xh, void* xl, float* stats, int nimg, int ntok, int D, int dtype,
                    mk_stream_t stream) {
  MK_CHECK_ARG(cls && pos && xh && xl && stats && nimg > 0 && ntok > 0 && D > 0 && D % 64 == 0, "mk_cls_token_ln: bad args");
  MK_CHECK_ARG(dtype == MK_BF16 || dtype == MK_F16, "mk_cls_token_ln: 16-bit dtypes only");
  if (dtype == MK_BF16)
    hipLaunchKernelGGL((cls_ln_kernel<__bf16>), dim3(nimg), dim3(64), 0, (hipStream_t)stream, cls, pos, (__bf16*)xh, (__bf16*)xl, stats, ntok, D);
  else
    hipLaunchKernelGGL((cls_ln_kernel<_Float16>), dim3(nimg), dim3(64), 0, (hipStream_t)stream, cls, pos, (_Float16*)xh, (_Float16*)xl, stats, ntok, D);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

/* centre every row of a split stream on its own mean (in place) and rewrite its per-slot statistics */
int mk_recentre_split(void* xh, void* xl, float* stats, long long rows, int D, int dtype, mk_stream_t stream) {
  MK_CHECK_ARG(xh && xl && stats && rows > 0 && D > 0 && D % 64 == 0, "mk_recentre_split: bad args");
  MK_CHECK_ARG(dtype == MK_BF16 || dtype == MK_F16, "mk_recentre_split: 16-bit dtypes only");
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == MK_BF16)
    hipLaunchKernelGGL((recentre_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, (__bf16*)xh, (__bf16*)xl, stats, rows, D);
  else
    hipLaunchKernelGGL((recentre_kernel<_Float16>), grid, dim3(256), 0, (hipStream_t)stream, (_Float16*)xh, (_Float16*)xl, stats, rows, D);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // extern "C"
