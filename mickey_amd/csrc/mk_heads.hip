// mickey_amd -- small kernels of the four MicKey heads (reference mickey_extractor.py:67-251):
// sine positional encoding add, LoFTR linear attention (att_layers/attention.py:46-64) and the head
// tails.  None of these is FLOP-heavy (the heads' FLOPs live in mk_conv3x3 / mk_gemm_grouped);
// they are written for few launches, coalesced 16-byte accesses and deterministic reductions.
#include "mk_common.hpp"

namespace {
using namespace mk;

constexpr int KVW = 272;     // 16x16 KV + 16 Ksum per (group, image, head)
constexpr int KV_CHUNK = 64; // tokens per partial block (64 KiB of staged rows at C = 128; 32 measured the same here and doubled the reduce)

template <typename T>
__global__ __launch_bounds__(256) void posenc_kernel(const T* __restrict__ x, const float* __restrict__ pe,
                                                     float* __restrict__ xs, T* __restrict__ cat, int ld_cat, long long rows,
                                                     int npix, int C, long long total4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4));
    const long long r = i / (C / 4);  // row over [G*rows)
    const long long rr = r % rows;
    const int pix = (int)(rr % npix);
    const typename Lp<T>::V4 v = *(const typename Lp<T>::V4*)(x + r * C + c4 * 4);
    f32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = (float)v[e];
    if (pe) f += *(const f32x4*)(pe + (long long)pix * C + c4 * 4);
    *(f32x4*)(xs + r * C + c4 * 4) = f;
    typename Lp<T>::V4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (T)f[e];
    *(typename Lp<T>::V4*)(cat + r * ld_cat + c4 * 4) = o;
  }
}

__device__ __forceinline__ float phi(float x) { return x > 0.f ? x + 1.0f : expf(x); }  // elu(x) + 1

// partial KV over a chunk of KV_CHUNK tokens for ALL heads of one (group, image) (C = 128: 8 heads of 16).
// The chunk's k and v rows (1 KiB per token, contiguous in the 3C-wide qkv row) are staged into LDS by LDS-DMA (no register
// round trip, everything in flight at once), phi() is applied to the k half in place (once per element), then ONE wave works on a token:
// lane = (head, half of v, quarter of d) holds a 4 x 8 block of the head's 16 x 16 outer product, 3 LDS reads per 32 FMAs
// -- the first version had one thread per (head, d, half of v) read k and v straight from global memory: every v float4
// was requested by 16 lanes and every k by 2 (1 KiB of requests per 128 unique bytes), 2.4 TB/s.  The block's 4 waves take
// every 4th token and their partial sums are combined in wave order.
__global__ __launch_bounds__(256) void linattn_kv_partial(const float* __restrict__ qkv, float* __restrict__ part, int L, int C,
                                                          int nchunk) {
  extern __shared__ __attribute__((aligned(16))) float skv[];   // [KV_CHUNK][2C]: phi(k) | v ; reused for the wave partials
  const int H = C >> 4;
  const long long gi = blockIdx.y;      // g*nimg + img
  const int chunk = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int s0 = chunk * KV_CHUNK, ntok = min(L, s0 + KV_CHUNK) - s0;
  const float invL = 1.0f / (float)L;
  const float* base = qkv + (gi * (long long)L + s0) * 3 * C + C;   // k of the chunk's first token
  const int c4 = 2 * C / 4;                                          // float4 per token (k | v)
  // LDS-DMA, 16 B per lane: float4 i of the staged image <- token i / c4, column 4 (i % c4); a wave instruction fills 1 KiB
  // of LDS (one token at C = 128).  KV_CHUNK * c4 is a multiple of 256: nothing waits until all trips are issued.
  for (int it = 0; it < KV_CHUNK * c4 / 256; ++it) {
    const int i = it * 256 + t;
    const int s = min(i / c4, ntok - 1), c = (i % c4) * 4;
    glds16(base + (long long)s * 3 * C + c, (char*)skv + (it * 256 + wave * 64) * 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = t; i < ntok * (C / 4); i += 256) {   // phi() on the k half, in place, once per element
    const int s = i / (C / 4), c = (i - s * (C / 4)) * 4;
    f32x4 v = *(const f32x4*)(skv + s * 2 * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = phi(v[e]);
    *(f32x4*)(skv + s * 2 * C + c) = v;
  }
  __syncthreads();
  const int h = lane >> 3, vh = (lane >> 2) & 1, dg = lane & 3;
  float acc[4][8];
  float ks[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[a][e] = 0.f;
  if (h < H) {
    for (int s = wave; s < ntok; s += 4) {
      const float* row = skv + s * 2 * C + h * 16;
      const f32x4 k4 = *(const f32x4*)(row + dg * 4);
      const f32x4 v0 = *(const f32x4*)(row + C + vh * 8), v1 = *(const f32x4*)(row + C + vh * 8 + 4);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[a][e] += k4[a] * (v0[e] * invL);
          acc[a][4 + e] += k4[a] * (v1[e] * invL);
        }
        ks[a] += k4[a];
      }
    }
  }
  __syncthreads();   // everybody is done reading the staged rows: the buffer now takes the 4 wave partials [wave][H][KVW]
  if (h < H) {
    float* o = skv + (wave * H + h) * KVW;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[(dg * 4 + a) * 16 + vh * 8 + e] = acc[a][e];
      if (vh == 0) o[256 + dg * 4 + a] = ks[a];
    }
  }
  __syncthreads();
  for (int i = t; i < H * KVW; i += 256) {
    const int hh = i / KVW, e = i - hh * KVW;
    const float r = ((skv[(0 * H + hh) * KVW + e] + skv[(1 * H + hh) * KVW + e]) + skv[(2 * H + hh) * KVW + e]) + skv[(3 * H + hh) * KVW + e];
    part[(((gi * H + hh) * nchunk) + chunk) * KVW + e] = r;
  }
}

__global__ __launch_bounds__(KVW) void linattn_kv_reduce(const float* __restrict__ part, float* __restrict__ kv, int nchunk) {
  const long long gih = blockIdx.x;
  const int t = threadIdx.x;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[(gih * nchunk + c) * KVW + t];
  kv[gih * KVW + t] = s;
}

// block = 32 tokens x 8... generally (256/H) tokens x H heads of one (g, img)
template <typename T>
__global__ __launch_bounds__(256) void linattn_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ kv,
                                                            T* __restrict__ out, int ldo, int L, int C) {
  extern __shared__ __attribute__((aligned(16))) float skv[];  // [H][273]
  const int H = C >> 4;
  const long long gi = blockIdx.y;
  for (int i = threadIdx.x; i < H * KVW; i += blockDim.x) skv[(i / KVW) * 273 + (i % KVW)] = kv[gi * H * KVW + i];
  __syncthreads();
  const int tpb = 256 / H;
  const int h = threadIdx.x % H;
  const int s = blockIdx.x * tpb + threadIdx.x / H;
  if (s >= L || threadIdx.x >= tpb * H) return;
  const float* qrow = qkv + (gi * L + s) * 3 * C + h * 16;
  float Q[16];
#pragma unroll
  for (int d4 = 0; d4 < 4; ++d4) {
    const f32x4 t4 = *(const f32x4*)(qrow + d4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) Q[d4 * 4 + e] = phi(t4[e]);
  }
  const float* K = skv + h * 273;
  float z = 0.f;
#pragma unroll
  for (int d = 0; d < 16; ++d) z += Q[d] * K[256 + d];
  const float scale = (float)L / (z + 1e-6f);
  typename Lp<T>::V8 o0, o1;
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) a += Q[d] * K[d * 16 + v];
    a *= scale;
    if (v < 8) o0[v] = (T)a; else o1[v - 8] = (T)a;
  }
  T* orow = out + (gi * L + s) * ldo + h * 16;
  *(typename Lp<T>::V8*)orow = o0;
  *(typename Lp<T>::V8*)(orow + 8) = o1;
}

// ---- tails ---------------------------------------------------------------------------------------

// detector logits + offset + depth: one wave per pixel, three C-long dot products from the three kp heads' features
// (reference mickey_extractor.py:134-138,172-178,213-218).  The raw detector logit goes to scr, which det_norm_kernel
// then normalises in place (round 6: the logits used to be computed by the one-workgroup-per-image normaliser itself, 121
// dependent row reads per wave = 67 us for ANY batch size)
__global__ __launch_bounds__(256) void kp_depth_tail_kernel(const float* __restrict__ fdet, const float* __restrict__ wsc,
                                                            const float* __restrict__ foff, const float* __restrict__ wxy,
                                                            const float* __restrict__ fdep, const float* __restrict__ wd,
                                                            float* __restrict__ scr, float* __restrict__ kps,
                                                            float* __restrict__ depth, int nimg, int h, int w, int C,
                                                            int use_depth_sigmoid, float max_depth, float down) {
  const int n = h * w;
  const long long gp = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (gp >= (long long)nimg * n) return;
  const int lane = threadIdx.x & 63;
  const int img = (int)(gp / n), p = (int)(gp % n);
  float as = 0.f, ax = 0.f, ay = 0.f, ad = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float fo = foff[gp * C + c];
    as += fdet[gp * C + c] * wsc[c];
    ax += fo * wxy[c];
    ay += fo * wxy[C + c];
    ad += fdep[gp * C + c] * wd[c];
  }
  as = wave_sum(as);
  ax = wave_sum(ax);
  ay = wave_sum(ay);
  ad = wave_sum(ad);
  if (lane == 0) {
    const int y = p / w, x = p % w;
    scr[gp] = as;
    kps[((long long)img * 2 + 0) * n + p] = (1.0f / (1.0f + expf(-ax)) + (float)x) * down;
    kps[((long long)img * 2 + 1) * n + p] = (1.0f / (1.0f + expf(-ay)) + (float)y) * down;
    depth[gp] = use_depth_sigmoid ? max_depth * (1.0f / (1.0f + expf(-ad))) : ad;
  }
}

// detector: the logits of one image (in scr) -> scores, in place; one workgroup per image (reference mickey_extractor.py:98-124)
__global__ __launch_bounds__(1024) void det_norm_kernel(float* __restrict__ scr, int h, int w, int border, int use_softmax) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [n] scores + [32] reduction scratch
  const int n = h * w;
  float* sred = sm + n;
  const int img = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int p = threadIdx.x; p < n; p += blockDim.x) sm[p] = scr[(long long)img * n + p];
  __syncthreads();
  if (!use_softmax) {
    for (int p = threadIdx.x; p < n; p += blockDim.x) {
      const int y = p / w, x = p % w;
      const bool in = y >= border && y < h - border && x >= border && x < w - border;
      scr[(long long)img * n + p] = in ? 1.0f / (1.0f + expf(-sm[p])) : 0.f;
    }
    return;
  }
  float a = 0.f;
  for (int p = threadIdx.x; p < n; p += blockDim.x) a += sm[p];
  a = wave_sum(a);
  if (lane == 0) sred[wave] = a;
  __syncthreads();
  float tot = 0.f;
  for (int i = 0; i < nw; ++i) tot += sred[i];
  const float mean = tot / (float)n + 1e-16f;
  __syncthreads();
  float es = 0.f;
  for (int p = threadIdx.x; p < n; p += blockDim.x) {
    const int y = p / w, x = p % w;
    const bool in = y >= border && y < h - border && x >= border && x < w - border;
    const float e = in ? expf((sm[p] - mean) / 100.0f) : 0.f;
    sm[p] = e;
    es += e;
  }
  es = wave_sum(es);
  if (lane == 0) sred[wave] = es;
  __syncthreads();
  float sum = 0.f;
  for (int i = 0; i < nw; ++i) sum += sred[i];
  const float inv = 1.0f / (sum + 1e-16f);
  for (int p = threadIdx.x; p < n; p += blockDim.x) scr[(long long)img * n + p] = sm[p] * inv;
}

// descriptors: l2-normalise over channels and transpose [pix, Cd] -> [Cd, pix]; block = 64 pixels
__global__ __launch_bounds__(256) void dsc_tail_kernel(const float* __restrict__ fdsc, float* __restrict__ dsc, int n, int Cd,
                                                       int norm_dsc) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [Cd][65]
  const int img = blockIdx.y, p0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int pp = wave; pp < 64; pp += 4) {
    const int p = p0 + pp;
    if (p >= n) break;
    const float* f = fdsc + ((long long)img * n + p) * Cd;
    float q = 0.f;
    for (int c = lane; c < Cd; c += 64) q += f[c] * f[c];
    q = wave_sum(q);
    const float sc = norm_dsc ? 1.0f / sqrtf(q + 1e-10f) : 1.0f;
    for (int c = lane; c < Cd; c += 64) tile[c * 65 + pp] = f[c] * sc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cd * 64; i += blockDim.x) {
    const int c = i >> 6, pp = i & 63;
    if (p0 + pp < n) dsc[((long long)img * Cd + c) * n + p0 + pp] = tile[c * 65 + pp];
  }
}

// ---- a 128-wide linear + LayerNorm (+ residual) in one pass (round 6) ----------------------------------------------------------
// The linear-attention layers close two of their sub-blocks with `Linear(K -> 128, no bias)` followed by LayerNorm over those 128
// features (reference att_layers/transformer_utils.py:58-66: merge -> norm1, mlp -> norm2 (+ the layer's residual)).  As two
// launches the fp32 GEMM output [4 x 124 k rows, 128] is written and read back (254 + 254 MB, twice per layer); here a wave keeps
// the 16 x 128 block of its 16 rows in the MFMA accumulators, normalises it there and writes only what the next kernel reads.
//   workgroup = 4 waves x 16 rows; the group's W [128, K] stays in LDS (K <= 256: 64 KiB, chunk-swizzled: the fragment reads of 16
//   lanes hit 16 different 16-byte bank groups) while the workgroup walks row tiles of its group; activations go from global memory
//   straight into the MFMA B-operand registers (16 bytes per lane and K step).  Operands swapped as in the GEMM kernels (A-operand =
//   W rows): a lane ends up with 4 consecutive features of ONE row per 16-feature block, 32 of the row's 128 values in all, the
//   other 96 in lanes +16, +32, +48 -- the row statistics are two lane exchanges.  K steps are visited in order: the accumulators
//   are those of the GEMM kernels bit for bit.  HBM-bound (MFMA work: 0.45 us per 64-row tile against ~4 us of its bytes).
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_ln128_kernel(const T* __restrict__ A, int lda, long long strideA, const T* __restrict__ W,
                                                            int ldw, long long strideW, const float* __restrict__ lnw,
                                                            const float* __restrict__ lnb, float eps, float* __restrict__ resid, int ldr,
                                                            T* __restrict__ out, int ldo, int groups, int M, int K, int bord_h, int bord_w) {
  using V8 = typename Lp<T>::V8;
  using V4 = typename Lp<T>::V4;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // W: [K / 32][128 rows][64 B, chunk-swizzled] | ln weight [128] | ln bias [128]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.x % groups, wg = blockIdx.x / groups, nwg = gridDim.x / groups;
  const int ks = K >> 5;
  const T* Wg = W + (long long)g * strideW;
  for (int i = threadIdx.x; i < 128 * ks * 4; i += 256) {   // 16-byte chunks, a W row's chunks consecutive
    const int c = i & 3, s = (i >> 2) % ks, n = i / (4 * ks);
    const uint4 v = *(const uint4*)(Wg + (long long)n * ldw + s * 32 + c * 8);
    *(uint4*)(smem + ((s * 128 + n) * 64 + ((c ^ ((n >> 2) & 3)) << 4))) = v;
  }
  float* slw = (float*)(smem + 128 * K * 2);
  float* slb = slw + 128;
  if (threadIdx.x < 128) {
    slw[threadIdx.x] = lnw[g * 128 + threadIdx.x];
    slb[threadIdx.x] = lnb[g * 128 + threadIdx.x];
  }
  __syncthreads();
  const int r16 = lane & 15, q = lane >> 4;
  const int ntile = (M + 63) >> 6;
  const long long brows = bord_h > 0 ? bordered_rows(M / (bord_h * bord_w), bord_h, bord_w) : 0;
  // software pipeline over the workgroup's tiles: a tile's activations are requested while the previous tile is normalised and stored
  // (into the registers its MFMAs have just released), its residual rows at the end of the previous epilogue
  V8 xf[8];
  f32x4 rv[8];
  auto row_of = [&](int tile) { return tile * 64 + wave * 16 + r16; };   // this lane's row inside the group
  auto load_x = [&](int tile) {
    const int row = row_of(tile);
    const T* ar = A + (long long)g * strideA + (long long)(row < M ? row : M - 1) * lda + q * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < ks) xf[s] = *(const V8*)(ar + s * 32);
  };
  auto load_r = [&](int tile) {
    const int row = row_of(tile);
    const float* rp = resid + ((long long)g * M + (row < M ? row : M - 1)) * ldr + q * 4;
#pragma unroll
    for (int f = 0; f < 8; ++f) rv[f] = *(const f32x4*)(rp + f * 16);
  };
  if (wg < ntile) {
    load_x(wg);
    if (resid) load_r(wg);
  }
  for (int tile = wg; tile < ntile; tile += nwg) {
    const int row = row_of(tile);
    const bool ok = row < M;
    const bool more = tile + nwg < ntile;   // workgroup-uniform
    f32x4 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s >= ks) break;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int n = f * 16 + r16;
        const V8 wf = *(const V8*)(smem + ((s * 128 + n) * 64 + ((q ^ ((n >> 2) & 3)) << 4)));
        acc[f] = Lp<T>::mma16(wf, xf[s], acc[f]);
      }
    }
    if (more) load_x(tile + nwg);
    // LayerNorm over the row's 128 features: this lane's 32 + lanes ^16, ^32
    float sm = 0.f;
#pragma unroll
    for (int f = 0; f < 8; ++f) sm += (acc[f][0] + acc[f][1]) + (acc[f][2] + acc[f][3]);
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / 128.0f);
    float qs = 0.f;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[f][e] -= mean;
        qs += acc[f][e] * acc[f][e];
      }
    }
    qs += __shfl_xor(qs, 16, 64);
    qs += __shfl_xor(qs, 32, 64);
    const float rstd = 1.0f / sqrtf(qs * (1.0f / 128.0f) + eps);
    if (ok) {
      const long long grow = (long long)g * M + row;
      const long long orow = bord_h > 0 ? (long long)g * brows + bordered_row(row, bord_h, bord_w) : grow;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int fe = f * 16 + q * 4;
        const f32x4 ww = *(const f32x4*)(slw + fe), bb = *(const f32x4*)(slb + fe);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = acc[f][e] * rstd * ww[e] + bb[e];
        if (resid) {
          y += rv[f];
          *(f32x4*)(resid + grow * ldr + fe) = y;
        }
        V4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (T)y[e];
        *(V4*)(out + orow * ldo + fe) = o;
      }
    }
    if (resid && more) load_r(tile + nwg);
  }
}

}  // namespace

extern "C" {

int mk_posenc_add(const void* x, const float* pe, float* xs, void* cat, int ld_cat, int groups, int nimg, int npix, int C,
                  int dtype, mk_stream_t stream) {
  MK_CHECK_ARG(x && xs && cat && groups > 0 && nimg > 0 && npix > 0 && C % 8 == 0 && ld_cat % 4 == 0 && ld_cat >= C,
               "mk_posenc_add: bad args");
  const long long rows = (long long)nimg * npix;
  const long long total4 = (long long)groups * rows * (C / 4);
  int blocks = (int)((total4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (dtype == MK_BF16)
    hipLaunchKernelGGL(posenc_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, pe, xs,
                       (__bf16*)cat, ld_cat, rows, npix, C, total4);
  else if (dtype == MK_F16)
    hipLaunchKernelGGL(posenc_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, pe, xs,
                       (_Float16*)cat, ld_cat, rows, npix, C, total4);
  else
    hipLaunchKernelGGL(posenc_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, pe, xs,
                       (float*)cat, ld_cat, rows, npix, C, total4);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

long long mk_linattn_work_floats(int groups, int nimg, int L, int C) {
  const int nchunk = (L + KV_CHUNK - 1) / KV_CHUNK;
  return (long long)groups * nimg * (C / 16) * nchunk * KVW;
}

int mk_linattn_kv(const float* qkv, float* kv, float* work, int groups, int nimg, int L, int C, mk_stream_t stream) {
  MK_CHECK_ARG(qkv && kv && work && groups > 0 && nimg > 0 && L > 0 && C % 16 == 0 && C <= 128, "mk_linattn_kv: bad args (C <= 128)");
  const int nchunk = (L + KV_CHUNK - 1) / KV_CHUNK;
  const int gih = groups * nimg * (C / 16);
  size_t lds = (size_t)KV_CHUNK * 2 * C * sizeof(float);         // the staged rows, then reused for
  if (lds < (size_t)4 * (C / 16) * KVW * sizeof(float)) lds = (size_t)4 * (C / 16) * KVW * sizeof(float);   // the 4 wave partials
  hipLaunchKernelGGL(linattn_kv_partial, dim3(nchunk, groups * nimg), dim3(256), lds, (hipStream_t)stream, qkv, work, L, C,
                     nchunk);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(linattn_kv_reduce, dim3(gih), dim3(KVW), 0, (hipStream_t)stream, work, kv, nchunk);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_linattn_apply(const float* qkv, const float* kv, void* out, int ldo, int groups, int nimg, int L, int C, int dtype,
                     mk_stream_t stream) {
  const int H = C / 16;
  MK_CHECK_ARG(qkv && kv && out && groups > 0 && nimg > 0 && L > 0 && C % 16 == 0 && H <= 64 && ldo % 8 == 0 && ldo >= C,
               "mk_linattn_apply: bad args");
  const int tpb = 256 / H;
  dim3 grid((L + tpb - 1) / tpb, groups * nimg);
  const size_t lds = (size_t)H * 273 * sizeof(float);
  if (dtype == MK_BF16)
    hipLaunchKernelGGL(linattn_apply_kernel<__bf16>, grid, dim3(256), lds, (hipStream_t)stream, qkv, kv, (__bf16*)out, ldo, L,
                       C);
  else if (dtype == MK_F16)
    hipLaunchKernelGGL(linattn_apply_kernel<_Float16>, grid, dim3(256), lds, (hipStream_t)stream, qkv, kv, (_Float16*)out,
                       ldo, L, C);
  else
    hipLaunchKernelGGL(linattn_apply_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, qkv, kv, (float*)out, ldo, L, C);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_gemm_ln128(const void* A, int lda, long long strideA, const void* W, int ldw, long long strideW, const float* ln_w,
                  const float* ln_b, float eps, float* resid, int ldr, void* out, int ldo, int groups, int M, int K, int bord_h,
                  int bord_w, int dtype, mk_stream_t stream) {
  MK_CHECK_ARG(A && W && ln_w && ln_b && out, "mk_gemm_ln128: null pointer");
  MK_CHECK_ARG(dtype == MK_BF16 || dtype == MK_F16, "mk_gemm_ln128: 16-bit operands only");
  MK_CHECK_ARG(groups > 0 && M > 0 && K >= 32 && K <= 256 && K % 32 == 0 && lda % 8 == 0 && lda >= K && ldw % 8 == 0 && ldw >= K &&
                   ldo % 4 == 0 && ldo >= 128 && (!resid || (ldr % 4 == 0 && ldr >= 128)),
               "mk_gemm_ln128: bad geometry (K a multiple of 32 up to 256, 128 output features)");
  MK_CHECK_ARG((((uintptr_t)A | (uintptr_t)W) & 15) == 0 && ((uintptr_t)out & 7) == 0 && (strideA % 8) == 0 && (strideW % 8) == 0,
               "mk_gemm_ln128: operands must be 16-byte aligned");
  MK_CHECK_ARG(bord_h == 0 || (bord_w > 0 && M % (bord_h * bord_w) == 0), "mk_gemm_ln128: bordered output needs M = nimg * bord_h * bord_w");
  const int lds = 128 * K * 2 + 1024;
  const int ntile = (M + 63) / 64;
  // workgroups per CU: two with a 64-KiB W (K = 256), three with 32 KiB (K <= 128: the registers' limit); each walks its share of a group's tiles
  int per_group = ((K > 128 ? 2 : 3) * mk::gemm::num_cus() + groups - 1) / groups;
  if (per_group > ntile) per_group = ntile;
  const dim3 grid((unsigned)(per_group * groups));
  hipError_t e = hipSuccess;
  if (dtype == MK_BF16) {
    static bool done = false;
    if (!done) { e = hipFuncSetAttribute((const void*)gemm_ln128_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 256 * 2 + 1024); done = e == hipSuccess; }
    if (e == hipSuccess)
      hipLaunchKernelGGL(gemm_ln128_kernel<__bf16>, grid, dim3(256), lds, (hipStream_t)stream, (const __bf16*)A, lda, strideA, (const __bf16*)W,
                         ldw, strideW, ln_w, ln_b, eps, resid, ldr, (__bf16*)out, ldo, groups, M, K, bord_h, bord_w);
  } else {
    static bool done = false;
    if (!done) { e = hipFuncSetAttribute((const void*)gemm_ln128_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 256 * 2 + 1024); done = e == hipSuccess; }
    if (e == hipSuccess)
      hipLaunchKernelGGL(gemm_ln128_kernel<_Float16>, grid, dim3(256), lds, (hipStream_t)stream, (const _Float16*)A, lda, strideA,
                         (const _Float16*)W, ldw, strideW, ln_w, ln_b, eps, resid, ldr, (_Float16*)out, ldo, groups, M, K, bord_h, bord_w);
  }
  if (e != hipSuccess) { mk_set_error("mk_gemm_ln128: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e)); return MK_ERR_LAUNCH; }
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_head_tails(const float* feat_det, const float* w_score, const float* feat_off, const float* w_xy,
                  const float* feat_depth, const float* w_depth, const float* feat_dsc, float* scr, float* kps, float* depth,
                  float* dsc, int nimg, int h, int w, int C, int Cd, int border, int use_softmax, int use_depth_sigmoid,
                  float max_depth, int norm_dsc, float down, mk_stream_t stream) {
  MK_CHECK_ARG(feat_det && w_score && feat_off && w_xy && feat_depth && w_depth && feat_dsc && scr && kps && depth && dsc,
               "mk_head_tails: null pointer");
  const int n = h * w;
  MK_CHECK_ARG(nimg > 0 && n > 0 && C > 0 && Cd > 0 && (size_t)(n + 32) * 4 <= 160 * 1024 && Cd * 65 * 4 <= 160 * 1024,
               "mk_head_tails: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(kp_depth_tail_kernel, dim3((unsigned)(((long long)nimg * n + 3) / 4)), dim3(256), 0, st, feat_det, w_score,
                     feat_off, w_xy, feat_depth, w_depth, scr, kps, depth, nimg, h, w, C, use_depth_sigmoid, max_depth, down);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(det_norm_kernel, dim3(nimg), dim3(1024), (size_t)(n + 32) * sizeof(float), st, scr, h, w, border, use_softmax);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(dsc_tail_kernel, dim3((n + 63) / 64, nimg), dim3(256), (size_t)Cd * 65 * sizeof(float), st, feat_dsc, dsc,
                     n, Cd, norm_dsc);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // extern "C"
