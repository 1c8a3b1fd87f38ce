// mickey_amd -- input pipeline kernel: decoded uint8 RGB frames -> the fp32 CHW tensors the model eats.
//
// reference lib/datasets/utils.py:61-78 (read_color_image): cv2 decode -> RGB -> cv2.resize(image, (w, h)) -> float
// -> permute(2, 0, 1) -> / 255.  Decoding stays on host cores (no JPEG engine is exposed on this stack); everything after it
// is this one HBM-bound pass: 3 B read + 12 B written per output pixel, one thread per output pixel (x fastest: the three
// channel planes are written with coalesced 4-byte stores, the 3-byte source pixels of a row are read once through L1).
// Resize = bilinear with half-pixel centres and edge clamping, i.e. cv2.resize's INTER_LINEAR sampling rule evaluated in
// fp32 (OpenCV's uint8 path quantises the weights to 11 bits and rounds the result to uint8: differences <= 1/255).  When
// source and target sizes agree -- Map-free frames are stored at 540 x 720, the size the model is run at
// (config/datasets/mapfree.yaml:6-7) -- every weight is exactly 0 or 1 and the result is bit-identical to the reference's
// float(v) / 255.
#include "mk_common.hpp"

namespace {

__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ src, long long stride_img, int Hs,
                                                            int Ws, float* __restrict__ dst, int H, int W, float sy, float sx) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int n = blockIdx.z;
  if (x >= W) return;
  // cv2.resize INTER_LINEAR: f = (d + 0.5) * scale - 0.5; s = floor(f); f -= s; clamp to the image
  float fy = ((float)y + 0.5f) * sy - 0.5f;
  int y0 = (int)floorf(fy);
  fy -= (float)y0;
  if (y0 < 0) { y0 = 0; fy = 0.f; }
  if (y0 >= Hs - 1) { y0 = Hs - 1; fy = 0.f; }
  float fx = ((float)x + 0.5f) * sx - 0.5f;
  int x0 = (int)floorf(fx);
  fx -= (float)x0;
  if (x0 < 0) { x0 = 0; fx = 0.f; }
  if (x0 >= Ws - 1) { x0 = Ws - 1; fx = 0.f; }
  const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
  const unsigned char* im = src + (long long)n * stride_img;
  const unsigned char* p00 = im + ((long long)y0 * Ws + x0) * 3;
  const unsigned char* p01 = im + ((long long)y0 * Ws + x1) * 3;
  const unsigned char* p10 = im + ((long long)y1 * Ws + x0) * 3;
  const unsigned char* p11 = im + ((long long)y1 * Ws + x1) * 3;
  float* o = dst + ((long long)n * 3 * H + y) * W + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a = (float)p00[c], b = (float)p01[c], d = (float)p10[c], e = (float)p11[c];
    const float top = a + fx * (b - a), bot = d + fx * (e - d);
    o[(long long)c * H * W] = (top + fy * (bot - top)) / 255.0f;
  }
}

}  // namespace

extern "C" int mk_preprocess_u8(const unsigned char* src, long long stride_img, int n, int Hs, int Ws, float* dst, int H, int W,
                                mk_stream_t stream) {
  MK_CHECK_ARG(src && dst && n > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0, "mk_preprocess_u8: bad args");
  MK_CHECK_ARG(stride_img >= (long long)Hs * Ws * 3, "mk_preprocess_u8: stride_img smaller than a frame");
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3((W + 255) / 256, H, n), dim3(256), 0, (hipStream_t)stream, src, stride_img, Hs, Ws,
                     dst, H, W, (float)Hs / (float)H, (float)Ws / (float)W);
  MK_CHECK_LAUNCH();
  return MK_OK;
}
