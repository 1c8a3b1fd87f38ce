"""CPU restatement of the reference's input preparation -- TEST INFRASTRUCTURE, like the rest of
oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

  read_color_image        reference lib/datasets/utils.py:61-78 (and demo_inference.py:12-29): decode, RGB, resize, /255, CHW
  correct_intrinsic_scale reference lib/datasets/utils.py:86-99

The resize is `cv2.resize(image, (w, h))` on the decoded uint8 frame (utils.py:71: default interpolation INTER_LINEAR), then
`.float() / 255` (utils.py:74).  cv2 is a third-party dependency that is ABSENT from this container and from
/root/reference; the reference pins it as opencv-python==4.8.0.74 (resources/environment.yml:16).  `resize_u8` below restates
the PUBLISHED integer algorithm of that version (OpenCV 4.8.0, modules/imgproc/src/resize.cpp) for CV_8UC3 -- it is byte/integer
work, so the restatement is exact, not approximate:

  * hal::resize: scale = 1. / ((double) dst / src) per axis; when both scales are exactly 2 an INTER_LINEAR request is
    served by the fast area path (resize.cpp: "INTER_AREA (fast) also is equal to INTER_LINEAR"): out = (a + b + c + d + 2) >> 2
    over each 2 x 2 block (ResizeAreaFast_Invoker / ResizeAreaFastVec, cn = 3);
  * otherwise resizeGeneric_ with 11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS = 11): per destination index
    f = (float)((d + 0.5) * scale - 0.5); s = cvFloor(f); f -= s; horizontally s < 0 -> (s, f) = (0, 0) and
    s >= src - 1 -> (s, f) = (src - 1, 0); vertically the two ROW indices are clipped to the image and the weights kept;
    weights = saturate_cast<short>((1 - f, f) * 2048) (cvRound: round half to even);
  * HResizeLinear<uchar, int, short, 2048>:  row[x] = S[s] * a0 + S[s + 1] * a1  (int32);
  * VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>, VResizeLinearVec_32s8u> (the 8-bit specialisation):
        dst = uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)
    (the SIMD body computes the same integers); the IPP resize is not taken for 8-bit linear (ipp_resize: "doesn't match
    OpenCV exactly" unless useIPP_NotExact()).

PARITY PINNING: no cv2 binary can be run here, so this restatement is pinned by hand-derivable known answers
(tests/test_input_oracle_cpu.py: constant images, 2x2 -> 3x3 and 2x1 -> 4x1 worked by hand from the formulas above, exact 2x
decimation = rounded block mean, identity) -- "parity unpinned" against a cv2 BINARY, pinned against the published algorithm.
The HIP kernel (csrc/mk_input.hip) must equal this oracle byte for byte."""
import numpy as np
import torch

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _round_half_even_to_short(v):
    """saturate_cast<short>(float): cvRound (round half to even) then saturation."""
    return np.clip(np.rint(v.astype(np.float32)), -32768, 32767).astype(np.int32)


def linear_coeffs(n_dst, n_src, horizontal):
    """-> (s0, s1, w0, w1): source indices (already clipped) and 11-bit weights per destination index (resize.cpp,
    hal::resize, the xofs / ialpha and yofs / ibeta tables)."""
    scale = 1.0 / (float(n_dst) / float(n_src))                 # double, as computed by cv::resize + hal::resize
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)            # fx = (float)((dx + 0.5) * scale_x - 0.5)
    s = np.floor(f).astype(np.int64)                            # cvFloor
    f = (f - s.astype(np.float32)).astype(np.float32)           # fx -= sx   (float)
    if horizontal:
        lo = s < 0
        s[lo], f[lo] = 0, 0.0
        hi = s >= n_src - 1
        s[hi], f[hi] = n_src - 1, 0.0
    w0 = _round_half_even_to_short((np.float32(1.0) - f) * np.float32(COEF_SCALE))
    w1 = _round_half_even_to_short(f * np.float32(COEF_SCALE))
    s0 = np.clip(s, 0, n_src - 1)
    s1 = np.clip(s + 1, 0, n_src - 1)
    return s0, s1, w0, w1


def resize_u8(img, w, h):
    """uint8 [Hs, Ws, C] -> uint8 [h, w, C]: cv2.resize(img, (w, h)) with the default INTER_LINEAR, OpenCV 4.8.0."""
    assert img.dtype == np.uint8 and img.ndim == 3
    Hs, Ws = img.shape[:2]
    if (w, h) == (Ws, Hs):
        return img.copy()                                       # cv::resize of equal sizes degenerates to a copy
    if Ws == 2 * w and Hs == 2 * h:                             # both scales exactly 2: the fast area path
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = linear_coeffs(w, Ws, horizontal=True)
    y0, y1, b0, b1 = linear_coeffs(h, Hs, horizontal=False)
    s = img.astype(np.int32)
    rows = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]          # HResizeLinear: int32 [Hs, w, C]
    S0, S1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    assert out.min() >= 0 and out.max() <= 255
    return out.astype(np.uint8)


def read_color_image(rgb_u8, resize=None):
    """rgb_u8: decoded uint8 [Hs, Ws, 3] RGB; resize = (w, h) or None.  Returns fp32 [3, h, w] in [0, 1]
    (utils.py:70-74: cv2.resize on the uint8 frame, THEN float / 255)."""
    img = rgb_u8
    if resize is not None:
        img = resize_u8(img, int(resize[0]), int(resize[1]))
    return torch.from_numpy(np.ascontiguousarray(img)).float().permute(2, 0, 1) / 255


def correct_intrinsic_scale(K, scale_x, scale_y):
    transform = torch.eye(3)
    transform[0, 0] = scale_x
    transform[0, 2] = scale_x / 2 - 0.5
    transform[1, 1] = scale_y
    transform[1, 2] = scale_y / 2 - 0.5
    return transform @ K
