"""Iterate the REFERENCE's MapFreeScene (lib/datasets/mapfree.py) over a dataset directory and dump what it yields.
Subprocess of tests/test_mapfree_eval_cpu.py with PYTHONPATH = <reference checkout>; third-party modules this container
lacks are stood in for by ref_env_shims (cv2 -> PIL, transforms3d: unused for test_scene=True)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_env_shims  # noqa: E402


def main():
    split_root, W, H, factor = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    ref_env_shims.install()
    from lib.datasets.mapfree import MapFreeScene
    out = []
    for s in sorted(os.listdir(split_root)):
        ds = MapFreeScene(os.path.join(split_root, s), (W, H), factor, (0.2, 0.7), None, True)
        for i in range(len(ds)):
            d = ds[i]
            out.append({"scene_id": d["scene_id"], "pair_names": list(d["pair_names"]),
                        "K_color0": np.asarray(d["K_color0"]).tolist(), "K_color1": np.asarray(d["K_color1"]).tolist(),
                        "Kori_color0": np.asarray(d["Kori_color0"]).tolist(),
                        "image0_shape": list(d["image0"].shape), "image0_sum": float(d["image0"].double().sum()),
                        "image1_sum": float(d["image1"].double().sum())})
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
