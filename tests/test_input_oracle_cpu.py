"""CPU: oracle/input_oracle.py (the restatement of cv2.resize INTER_LINEAR for uint8 frames, OpenCV 4.8.0 resize.cpp, as the
reference calls it at lib/datasets/utils.py:70-74) against known answers worked BY HAND from the published integer formulas.
No cv2 binary exists in this image; these vectors are what pins the oracle (see its header)."""
import numpy as np

from oracle import input_oracle as IO


def test_coefficient_tables_by_hand():
    # 2 -> 4 (scale 0.5): f = -0.25, 0.25, 0.75, 1.25 -> horizontally (s, f) = (0, 0), (0, .25), (0, .75), (1, 0)
    s0, s1, w0, w1 = IO.linear_coeffs(4, 2, horizontal=True)
    assert s0.tolist() == [0, 0, 0, 1] and w0.tolist() == [2048, 1536, 512, 2048] and w1.tolist() == [0, 512, 1536, 0]
    assert s1.tolist() == [1, 1, 1, 1]
    # vertically the row indices are clipped but the weights are kept: d = 0 -> rows (0, 0) with (512, 1536); d = 3 -> s = 1,
    # f = 0.25 -> rows (1, 1) with (1536, 512)
    s0, s1, w0, w1 = IO.linear_coeffs(4, 2, horizontal=False)
    assert s0.tolist() == [0, 0, 0, 1] and s1.tolist() == [0, 1, 1, 1]
    assert w0.tolist() == [512, 1536, 512, 1536] and w1.tolist() == [1536, 512, 1536, 512]
    # 3 -> 2 (scale 1.5): f = 0.25, 1.75 -> (0, .25), (1, .75)
    s0, s1, w0, w1 = IO.linear_coeffs(2, 3, horizontal=True)
    assert s0.tolist() == [0, 1] and s1.tolist() == [1, 2] and w0.tolist() == [1536, 512] and w1.tolist() == [512, 1536]
    # weights are rounded half to even (cvRound): 5 -> 3: scale 5/3; d = 1: f = (float)(1.5 * 5/3 - 0.5) = 2.0 exactly
    s0, s1, w0, w1 = IO.linear_coeffs(3, 5, horizontal=True)
    assert s0.tolist() == [0, 2, 3] and (w0 + w1 == 2048).all()


def test_row_2_to_4_by_hand():
    """One row [100, 200] stretched to width 4 (height 1 -> 2, so the generic path runs in both directions)."""
    img = np.array([[[100, 7, 0], [200, 9, 255]]], dtype=np.uint8)              # [1, 2, 3]
    out = IO.resize_u8(img, 4, 2)
    assert out.shape == (2, 4, 3)
    # channel 0: rows = 100*2048, 100*1536 + 200*512, 100*512 + 200*1536, 200*2048 = 204800, 256000, 358400, 409600
    # vertical (1 -> 2, scale 0.5): d = 0: rows (0, 0), weights (512, 1536): ((512 * (r >> 4)) >> 16) + ((1536 * (r >> 4)) >> 16)
    #   r = 256000: r >> 4 = 16000 -> 125 + 375 = 500 -> (500 + 2) >> 2 = 125
    assert out[0, :, 0].tolist() == [100, 125, 175, 200] and out[1, :, 0].tolist() == [100, 125, 175, 200]
    # channel 1: 7, (7*1536 + 9*512) = 15360 -> >> 4 = 960 -> (512*960 >> 16) + (1536*960 >> 16) = 7 + 22 = 29 -> 31 >> 2 = 7
    assert out[0, :, 1].tolist() == [7, 7, 8, 9]
    # channel 2: 0, 255*512 = 130560 -> 8160 -> 63 + 191 = 254 -> 256 >> 2 = 64; 255*1536 = 391680 -> 24480 -> 191 + 573 = 764 -> 191
    assert out[0, :, 2].tolist() == [0, 64, 191, 255]


def test_2x2_to_3x3_by_hand():
    img = np.array([[[0], [255]], [[255], [0]]], dtype=np.uint8)
    out = IO.resize_u8(img, 3, 3)[:, :, 0]
    # 2 -> 3 (scale 2/3): f = -1/6, 0.5, 7/6 -> horizontally (0, 0), (0, 0.5), (1, 0) -> weights (2048, 0), (1024, 1024), (2048, 0)
    # rows of source row 0: 0, 255*1024 = 261120, 255*2048 = 522240; of source row 1: 522240, 261120, 0
    # vertically: d = 0: s = -1, f = (float)(5/6): weights rint((1 - f, f) * 2048) = (341, 1707) on rows (0, 0);
    #   d = 1: (1024, 1024) on rows (0, 1); d = 2: s = 1, f = (float)(1/6): (1707, 341) on rows (1, 1)
    # corner (0, 0): 0.  (0, 2): r = 522240 -> 32640: (341*32640 >> 16) + (1707*32640 >> 16) = 169 + 850 = 1019 -> 1021 >> 2 = 255
    # centre: rows 261120, 261120 -> 16320: (1024*16320 >> 16) * 2 = 255 * 2 = 510 -> 512 >> 2 = 128
    # (1, 0): rows 0 and 522240: 0 + (1024*32640 >> 16) = 510 -> 128;  (0, 1): r = 261120 -> 16320: 84 + 425 = 509 -> 511 >> 2 = 127
    assert out.tolist() == [[0, 127, 255], [128, 128, 128], [255, 127, 0]]


def test_constant_images_stay_constant_except_where_the_truncation_bites():
    """A constant frame v maps to v wherever a weight pair sums to 2048 on ONE row value (every case): the two shifted
    products can lose at most one unit before the +2 >> 2 rounding, which absorbs it."""
    for v in (0, 1, 17, 128, 254, 255):
        img = np.full((13, 9, 3), v, dtype=np.uint8)
        for (w, h) in ((20, 31), (5, 4), (9, 14), (18, 13)):
            assert (IO.resize_u8(img, w, h) == v).all(), (v, w, h)


def test_identity_and_exact_halving():
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (12, 10, 3), dtype=np.uint8)
    assert np.array_equal(IO.resize_u8(img, 10, 12), img)
    half = IO.resize_u8(img, 5, 6)      # both scales exactly 2: hal::resize takes the fast area path for INTER_LINEAR
    blocks = img.astype(np.int64).reshape(6, 2, 5, 2, 3).sum(axis=(1, 3))
    assert np.array_equal(half, ((blocks + 2) >> 2).astype(np.uint8))
    # halving in ONE direction only is not the area path
    one = IO.resize_u8(img, 5, 12)
    x0, x1, a0, a1 = IO.linear_coeffs(5, 10, horizontal=True)
    assert x0.tolist() == [0, 2, 4, 6, 8] and a0.tolist() == [1024] * 5     # f = 0.5 everywhere: plain 2-tap average
    assert np.array_equal(one, ((img[:, 0::2].astype(np.int64) + img[:, 1::2] + 1) >> 1).astype(np.uint8)) or \
        np.abs(one.astype(int) - ((img[:, 0::2].astype(int) + img[:, 1::2]) // 2)).max() <= 1


def test_read_color_image_is_resize_then_divide():
    g = np.random.default_rng(1)
    img = g.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    t = IO.read_color_image(img, resize=(28, 42))
    assert t.shape == (3, 42, 28) and t.dtype.is_floating_point
    q = (t * 255).round()
    assert float((t - q / 255).abs().max()) == 0.0                      # every value is exactly byte / 255
    assert np.array_equal(q.permute(1, 2, 0).numpy().astype(np.uint8), IO.resize_u8(img, 28, 42))


def test_against_an_independent_bilinear_implementation():
    """Not a pin of bit-exactness (only a cv2 binary could give that, and none exists here -- the row stays "parity unpinned"), but an
    INDEPENDENT implementation of the same sampling rule: torch's F.interpolate(mode="bilinear", align_corners=False, antialias=False)
    computes real-valued bilinear interpolation at half-pixel centres with edge clamp, exactly the function cv2's INTER_LINEAR
    approximates in 11-bit fixed point.  On random uint8 frames the restatement must stay within 1 LSB of it everywhere (the 11-bit
    coefficient rounding + the two truncating passes move a value by < 1), agree exactly on > 80 % of the pixels (random noise images: the worst case for rounding ties), and show only the small negative bias of cv2's truncating shifts (S >> 4, product >> 16 in VResizeLinear: about -0.1 LSB) -- for
    up- and down-scales, the Map-free native size and non-integer ratios.  Exact 2x decimation is excluded: cv2 (and the
    restatement) switch to the area-average fast path there, which is a different function by design."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    for (hs, ws, h, w) in ((720, 540, 360, 271), (97, 131, 64, 80), (48, 64, 720, 540), (30, 40, 45, 60), (101, 77, 101, 153),
                            (1280, 720, 720, 405)):
        img = rng.integers(0, 256, size=(hs, ws, 3), dtype=np.uint8)
        got = IO.resize_u8(img, w, h).astype(np.int64)
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(h, w), mode="bilinear", align_corners=False,
                            antialias=False)[0].permute(1, 2, 0).numpy()
        d = got - ref
        assert np.abs(d).max() < 1.0 + 1e-9, ((hs, ws, h, w), float(np.abs(d).max()))
        assert (got == np.rint(ref)).mean() > 0.8, ((hs, ws, h, w), float((got == np.rint(ref)).mean()))
        assert -0.3 < d.mean() < 0.05, ((hs, ws, h, w), float(d.mean()))
