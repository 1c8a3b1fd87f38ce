"""A tiny on-disk dataset in the Map-free layout (<root>/<split>/<scene>/{intrinsics.txt, poses.txt, seq0/, seq1/}):
TEST INFRASTRUCTURE for the evaluation harness (mickey_amd/mapfree_eval.py)."""
import os

import numpy as np


def make(root, split="val", scenes=("s00460", "s00461"), queries=11, size=(126, 168), seed=0):
    """`queries` query frames per scene (seq1/frame_00000 .. ) + the key frame seq0/frame_00000; size = (W, H) stored."""
    from PIL import Image
    g = np.random.default_rng(seed)
    W, H = size
    for s in scenes:
        d = os.path.join(root, split, s)
        os.makedirs(os.path.join(d, "seq0"))
        os.makedirs(os.path.join(d, "seq1"))
        names = ["seq0/frame_00000.jpg"] + ["seq1/frame_%05d.jpg" % i for i in range(queries)]
        with open(os.path.join(d, "intrinsics.txt"), "w") as fi, open(os.path.join(d, "poses.txt"), "w") as fp:
            fi.write("# name fx fy cx cy W H\n")
            fp.write("# name qw qx qy qz tx ty tz\n")
            for k, nm in enumerate(names):
                base = g.integers(0, 256, (H // 6 + 1, W // 6 + 1, 3), dtype=np.uint8)
                img = np.asarray(Image.fromarray(base).resize((W, H), Image.BICUBIC))
                Image.fromarray(img).save(os.path.join(d, nm), quality=92)
                f = 0.8 * W + float(g.uniform(0, 5))
                fi.write("%s %.4f %.4f %.4f %.4f %d %d\n" % (nm, f, f, W / 2 + float(g.uniform(-2, 2)), H / 2 + float(g.uniform(-2, 2)), W, H))
                q = g.normal(size=4) if k else np.array([1.0, 0, 0, 0])
                q = q / np.linalg.norm(q)
                t = g.normal(size=3) * (0.5 if k else 0.0)
                fp.write("%s %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n" % ((nm,) + tuple(q) + tuple(t)))
    return os.path.join(root, split)
