"""CPU restatement of the reference's input preparation -- TEST INFRASTRUCTURE, like the rest of
oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

  read_color_image        reference lib/datasets/utils.py:61-78 (and demo_inference.py:12-29): decode, RGB, resize, /255, CHW
  correct_intrinsic_scale reference lib/datasets/utils.py:86-99

cv2 is absent from this container, so `resize_bilinear` restates the SAMPLING RULE of cv2.resize(..., INTER_LINEAR)
(half-pixel centres, edge clamp: OpenCV modules/imgproc/src/resize.cpp, resizeGeneric_ / HResizeLinear) with fp32 weights.
OpenCV's uint8 path additionally quantises the weights to 11 bits and rounds the result to uint8; that rounding is NOT
restated (difference <= 1/255 per pixel): parity of the resize is pinned only up to that bound ("parity unpinned" beyond
it).  The identity resize -- the Map-free case, frames are stored at the model's 540 x 720 -- is exact by construction."""
import numpy as np
import torch


def resize_bilinear(img, w, h):
    """img uint8 / float [Hs, Ws, C] -> float32 [h, w, C]; cv2.resize(img, (w, h)) sampling in fp32."""
    Hs, Ws = img.shape[:2]
    src = img.astype(np.float32)

    def axis(n_dst, n_src):
        f = (np.arange(n_dst, dtype=np.float32) + np.float32(0.5)) * (np.float32(n_src) / np.float32(n_dst)) - np.float32(0.5)
        i0 = np.floor(f).astype(np.int64)
        fr = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        i0[lo], fr[lo] = 0, 0.0
        hi = i0 >= n_src - 1
        i0[hi], fr[hi] = n_src - 1, 0.0
        return i0, np.minimum(i0 + 1, n_src - 1), fr

    y0, y1, fy = axis(h, Hs)
    x0, x1, fx = axis(w, Ws)
    fx = fx[None, :, None]
    fy = fy[:, None, None]
    top = src[y0][:, x0] + fx * (src[y0][:, x1] - src[y0][:, x0])
    bot = src[y1][:, x0] + fx * (src[y1][:, x1] - src[y1][:, x0])
    return (top + fy * (bot - top)).astype(np.float32)


def read_color_image(rgb_u8, resize=None):
    """rgb_u8: decoded uint8 [Hs, Ws, 3] RGB; resize = (w, h) or None.  Returns fp32 [3, h, w] in [0, 1]."""
    img = rgb_u8
    if resize is not None and (resize[0] != img.shape[1] or resize[1] != img.shape[0]):
        img = resize_bilinear(img, resize[0], resize[1])
    return torch.from_numpy(np.ascontiguousarray(img)).float().permute(2, 0, 1) / 255


def correct_intrinsic_scale(K, scale_x, scale_y):
    transform = torch.eye(3)
    transform[0, 0] = scale_x
    transform[0, 2] = scale_x / 2 - 0.5
    transform[1, 1] = scale_y
    transform[1, 2] = scale_y / 2 - 0.5
    return transform @ K
