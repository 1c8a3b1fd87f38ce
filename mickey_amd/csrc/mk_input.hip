// mickey_amd -- input pipeline kernel: decoded uint8 RGB frames -> the fp32 CHW tensors the model eats.
//
// reference lib/datasets/utils.py:61-78 (read_color_image): cv2 decode -> RGB -> cv2.resize(image, (w, h)) on the UINT8 frame
// (default INTER_LINEAR) -> .float() -> permute(2, 0, 1) -> / 255.  Decoding stays on host cores (no JPEG engine is exposed on
// this stack); everything after it is this one HBM-bound pass: <= 12 B read + 12 B written per output pixel, one thread per
// output pixel (x fastest: the three channel planes are written with coalesced 4-byte stores, the 3-byte source pixels of a
// row are read once through L1).
//
// The resize is byte work and is BIT-EXACT with cv2 as the reference pins it (opencv-python 4.8.0.74,
// resources/environment.yml:16; OpenCV 4.8.0 modules/imgproc/src/resize.cpp, CV_8UC3), restated in oracle/input_oracle.py:
//   * equal sizes: a copy;
//   * both scales exactly 2: the fast area path that hal::resize substitutes for INTER_LINEAR, (a + b + c + d + 2) >> 2;
//   * otherwise 11-bit fixed-point bilinear: f = (float)((d + 0.5) * scale - 0.5) with scale a double, s = floor(f), f -= s,
//     horizontally clamped with the weight reset (s < 0 -> (0, 0); s >= src - 1 -> (src - 1, 0)), vertically the two row
//     indices clipped with the weights kept; weights = rint((1 - f, f) * 2048) (round half to even);
//     row = S[s] * a0 + S[s + 1] * a1 (int32), out = (((b0 * (row0 >> 4)) >> 16) + ((b1 * (row1 >> 4)) >> 16) + 2) >> 2.
// The coordinate arithmetic uses the explicitly rounded intrinsics (__dmul_rn ...): a contracted fma would move f by an ulp
// and flip a weight.  The result byte is converted exactly as the reference does: float(v) / 255 (IEEE division).
#include "mk_common.hpp"

namespace {

struct Tap {
  int s0, s1, w0, w1;
};

// the xofs / ialpha (horizontal) and yofs / ibeta (vertical) tables of hal::resize, evaluated per destination index
__device__ __forceinline__ Tap linear_tap(int d, double scale, int n_src, bool horizontal) {
#pragma clang fp contract(off)
  float f = (float)__dsub_rn(__dmul_rn(__dadd_rn((double)d, 0.5), scale), 0.5);
  int s = (int)floorf(f);
  f = __fsub_rn(f, (float)s);
  if (horizontal) {
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
  }
  Tap t;
  t.w0 = __float2int_rn(__fmul_rn(__fsub_rn(1.0f, f), 2048.0f));   // saturate_cast<short>(cvRound(.)): |.| <= 2048, no saturation
  t.w1 = __float2int_rn(__fmul_rn(f, 2048.0f));
  t.s0 = min(max(s, 0), n_src - 1);
  t.s1 = min(max(s + 1, 0), n_src - 1);
  return t;
}

// MODE 0: copy, 1: exact 2 x 2 decimation (area fast), 2: fixed-point bilinear
template <int MODE>
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ src, long long stride_img, int Hs,
                                                            int Ws, float* __restrict__ dst, int H, int W, double sy, double sx) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int n = blockIdx.z;
  if (x >= W) return;
  const unsigned char* im = src + (long long)n * stride_img;
  float* o = dst + ((long long)n * 3 * H + y) * W + x;
  int v[3];
  if (MODE == 0) {
    const unsigned char* p = im + ((long long)y * Ws + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = p[c];
  } else if (MODE == 1) {
    const unsigned char* p0 = im + ((long long)(2 * y) * Ws + 2 * x) * 3;
    const unsigned char* p1 = p0 + (long long)Ws * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = ((int)p0[c] + (int)p0[c + 3] + (int)p1[c] + (int)p1[c + 3] + 2) >> 2;
  } else {
    const Tap tx = linear_tap(x, sx, Ws, true);
    const Tap ty = linear_tap(y, sy, Hs, false);
    const unsigned char* r0 = im + (long long)ty.s0 * Ws * 3;
    const unsigned char* r1 = im + (long long)ty.s1 * Ws * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = (int)r0[tx.s0 * 3 + c] * tx.w0 + (int)r0[tx.s1 * 3 + c] * tx.w1;   // HResizeLinear, int32
      const int h1 = (int)r1[tx.s0 * 3 + c] * tx.w0 + (int)r1[tx.s1 * 3 + c] * tx.w1;
      v[c] = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;     // VResizeLinear, 8-bit specialisation
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) o[(long long)c * H * W] = __fdiv_rn((float)v[c], 255.0f);
}

}  // namespace

extern "C" int mk_preprocess_u8(const unsigned char* src, long long stride_img, int n, int Hs, int Ws, float* dst, int H, int W,
                                mk_stream_t stream) {
  MK_CHECK_ARG(src && dst && n > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0, "mk_preprocess_u8: bad args");
  MK_CHECK_ARG(stride_img >= (long long)Hs * Ws * 3, "mk_preprocess_u8: stride_img smaller than a frame");
  // cv::resize: inv_scale = (double) dsize / ssize; hal::resize: scale = 1. / inv_scale
  const double sx = 1.0 / ((double)W / (double)Ws), sy = 1.0 / ((double)H / (double)Hs);
  const dim3 grid((W + 255) / 256, H, n), block(256);
  if (H == Hs && W == Ws)
    hipLaunchKernelGGL(preprocess_u8_kernel<0>, grid, block, 0, (hipStream_t)stream, src, stride_img, Hs, Ws, dst, H, W, sy, sx);
  else if (Hs == 2 * H && Ws == 2 * W)
    hipLaunchKernelGGL(preprocess_u8_kernel<1>, grid, block, 0, (hipStream_t)stream, src, stride_img, Hs, Ws, dst, H, W, sy, sx);
  else
    hipLaunchKernelGGL(preprocess_u8_kernel<2>, grid, block, 0, (hipStream_t)stream, src, stride_img, Hs, Ws, dst, H, W, sy, sx);
  MK_CHECK_LAUNCH();
  return MK_OK;
}
