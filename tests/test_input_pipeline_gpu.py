"""-m gpu: the input pipeline kernel / feeder against the oracle's restatement of the reference's
read_color_image / correct_intrinsic_scale (lib/datasets/utils.py:61-99)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_identity_resize_is_bit_exact_with_the_reference_expression():
    """Map-free frames are stored at the model's 540 x 720: resize is the identity and the kernel must give exactly
    torch.from_numpy(img).float().permute(2, 0, 1) / 255 (utils.py:74)."""
    from mickey_amd import ops
    dev = _dev()
    g = np.random.default_rng(1)
    frames = g.integers(0, 256, (5, 720, 540, 3), dtype=np.uint8)
    out = ops.preprocess_u8(torch.from_numpy(frames).to(dev), 720, 540)
    ref = torch.from_numpy(frames).float().permute(0, 3, 1, 2) / 255
    assert out.shape == (5, 3, 720, 540) and torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("src,dst", [((480, 640), (720, 540)), ((1080, 1920), (360, 640)), ((33, 47), (91, 13)), ((720, 540), (719, 541)),
                                     ((1440, 1080), (720, 540)), ((720, 1080), (720, 540)), ((2, 2), (3, 3)), ((1, 2), (2, 4)),
                                     ((256, 192), (720, 540))])
def test_resize_is_byte_exact_with_the_cv2_algorithm(src, dst):
    """Non-identity sizes: the kernel == oracle/input_oracle.py (cv2.resize INTER_LINEAR on uint8 as OpenCV 4.8.0 computes it:
    11-bit fixed-point weights, two truncating products, +2 >> 2; exact 2 x 2 decimation = the fast area path; then
    float / 255), byte for byte -- torch.equal on the fp32 result, i.e. on byte / 255.  Up- and down-scales, one axis halved
    only (NOT the area path), degenerate 1- and 2-pixel frames (clipped rows keep their weights)."""
    from mickey_amd import ops
    from oracle import input_oracle as IO
    dev = _dev()
    g = np.random.default_rng(2)
    frames = g.integers(0, 256, (2,) + src + (3,), dtype=np.uint8)
    frames[1, :, : max(1, src[1] // 3)] = 255                     # saturated and black regions: no overshoot, exact corners
    frames[1, : max(1, src[0] // 4)] = 0
    out = ops.preprocess_u8(torch.from_numpy(frames).to(dev), dst[0], dst[1]).cpu()
    for i in range(2):
        ref = IO.read_color_image(frames[i], resize=(dst[1], dst[0]))
        assert out[i].shape == ref.shape
        bad = int((out[i] != ref).sum())
        assert torch.equal(out[i], ref), "%d of %d values differ, max %.3g" % (bad, ref.numel(), float((out[i] - ref).abs().max()))


def test_upscale_agrees_with_pil_bilinear_to_one_grey_level():
    """Independent cross-check of the sampling rule: PIL's bilinear filter has the same half-pixel geometry when
    UPscaling (its support only widens for downscaling); it rounds to uint8, hence the 1/255 + eps bound."""
    from PIL import Image
    from mickey_amd import ops
    dev = _dev()
    g = np.random.default_rng(3)
    img = g.integers(0, 256, (60, 45, 3), dtype=np.uint8)
    out = ops.preprocess_u8(torch.from_numpy(img[None]).to(dev), 144, 108)[0].cpu()
    pil = np.asarray(Image.fromarray(img).resize((108, 144), Image.BILINEAR)).astype(np.float32) / 255
    assert float((out.permute(1, 2, 0).numpy() - pil).__abs__().max()) <= 1.0 / 255 + 1e-6


def test_pair_feeder_batches_match_the_reference_preparation():
    """JPEG files on disk -> PairFeeder -> device batches == decode + the reference's resize / normalise / intrinsics
    rescale (restated in oracle/input_oracle.py), including the ragged last batch and the pass-through metadata."""
    import tempfile, os
    from PIL import Image
    from mickey_amd import input_pipeline as ip
    from oracle import input_oracle as IO
    dev = _dev()
    g = np.random.default_rng(4)
    K = np.array([[500.0, 0, 160.0], [0, 505.0, 120.0], [0, 0, 1.0]], dtype=np.float32)
    with tempfile.TemporaryDirectory() as d:
        recs, decoded = [], []
        for i in range(7):
            names = []
            for j in range(2):
                arr = g.integers(0, 256, (240, 320, 3), dtype=np.uint8)
                p = os.path.join(d, "p%d_%d.png" % (i, j))
                Image.fromarray(arr).save(p)          # PNG: lossless, so the decoded frame is known exactly
                names.append(p)
                decoded.append(arr)
            recs.append({"image0": names[0], "image1": names[1], "K_color0": K, "K_color1": K * np.array([[1.1], [1.0], [1.0]], np.float32),
                         "scene_id": "s%02d" % (i // 4), "pair_names": ("seq0/frame_00000.jpg", "seq1/frame_%05d.jpg" % i)})
        feeder = ip.PairFeeder(recs, batch_size=3, resize=(196, 182), device=dev, workers=4)
        assert len(feeder) == 3
        seen = 0
        for data in feeder:
            n = data["image0"].shape[0]
            assert data["image0"].shape == (n, 3, 182, 196) and data["K_color0"].shape == (n, 3, 3)
            for i in range(n):
                r = recs[seen + i]
                for key, j in (("image0", 0), ("image1", 1)):
                    ref = IO.read_color_image(decoded[2 * (seen + i) + j], resize=(196, 182))
                    assert torch.equal(data[key][i].cpu(), ref)          # byte-exact resize
                Kref = IO.correct_intrinsic_scale(torch.from_numpy(r["K_color0"]), 196 / 320, 182 / 240)
                assert torch.allclose(data["K_color0"][i].cpu(), Kref, atol=1e-5)
            assert data["scene_id"] == [r["scene_id"] for r in recs[seen:seen + n]]
            assert data["pair_names"][1] == [r["pair_names"][1] for r in recs[seen:seen + n]]
            seen += n
        assert seen == 7
