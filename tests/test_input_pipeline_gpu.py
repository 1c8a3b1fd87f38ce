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


@pytest.mark.parametrize("src,dst", [((480, 640), (720, 540)), ((1080, 1920), (360, 640)), ((33, 47), (91, 13)), ((720, 540), (719, 541))])
def test_bilinear_resize_vs_oracle(src, dst):
    from mickey_amd import ops
    from oracle import input_oracle as IO
    dev = _dev()
    g = np.random.default_rng(2)
    frames = g.integers(0, 256, (2,) + src + (3,), dtype=np.uint8)
    out = ops.preprocess_u8(torch.from_numpy(frames).to(dev), dst[0], dst[1]).cpu()
    for i in range(2):
        ref = IO.read_color_image(frames[i], resize=(dst[1], dst[0]))
        assert out[i].shape == ref.shape
        assert float((out[i] - ref).abs().max()) < 1e-4, float((out[i] - ref).abs().max())   # fp32 coordinate round-off (fma contraction): 0.03 grey levels; 1/255 = 3.9e-3
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0


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
                    assert float((data[key][i].cpu() - ref).abs().max()) < 1e-4
                Kref = IO.correct_intrinsic_scale(torch.from_numpy(r["K_color0"]), 196 / 320, 182 / 240)
                assert torch.allclose(data["K_color0"][i].cpu(), Kref, atol=1e-5)
            assert data["scene_id"] == [r["scene_id"] for r in recs[seen:seen + n]]
            assert data["pair_names"][1] == [r["pair_names"][1] for r in recs[seen:seen + n]]
            seen += n
        assert seen == 7
