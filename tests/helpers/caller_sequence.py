"""The CALL SEQUENCE of the reference's two inference callers, written out stand-alone (own code, same calls, same argument
kinds) so that it can run on a box without the reference tree -- i.e. on the GPU box, against the real HIP forward:

  demo_sequence        reference demo_inference.py:79-138 (run_demo_inference): cfg.merge_from_file -> build_model(cfg, ckpt)
                       -> data dict with float64 intrinsics from torch.from_numpy(...).unsqueeze(0) -> model(data,
                       return_inliers=...) -> reads data['depth0_map'], data['scr0'], data['image0'], data['inliers_list'],
                       data['R'], data['t'], model.e2e_Procrustes.num_samples_matches
  submission_sequence  reference submission.py:32-68 (predict + save_submission): data_to_model_device -> no_grad ->
                       R, t = model(data) -> per item .detach().cpu().numpy(), data['inliers'][i].item(), NaN filter,
                       quaternion, 'name qw qx qy qz tx ty tz conf' lines zipped per scene

The reference's own files run unmodified against the drop-in in tests/test_reference_callers.py (build container, no GPU);
this module is the same traffic for tests/test_callers_gpu.py."""
import zipfile

import numpy as np
import torch


def data_to_model_device(data, model):   # reference lib/utils/data.py:3-16
    device = next(model.parameters()).device
    for k, v in data.items():
        if torch.is_tensor(v):
            data[k] = v.to(device)
    return data


def demo_sequence(build_model, cfg, config_yaml, checkpoint, im0, im1, K0, K1, return_inliers):
    """im0 / im1: fp32 [1, 3, H, W] in [0, 1] (what read_color_image returns + unsqueeze); K0 / K1: float64 numpy 3x3."""
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    cfg.merge_from_file(config_yaml)
    model = build_model(cfg, checkpoint=checkpoint)
    data = {"image0": im0.to(device), "image1": im1.to(device),
            "K_color0": torch.from_numpy(K0).unsqueeze(0).to(device), "K_color1": torch.from_numpy(K1).unsqueeze(0).to(device)}
    model(data, return_inliers=return_inliers)
    read = {"depth0": data["depth0_map"][0], "invalid0": (data["depth0_map"][0] < 0.001).cpu()[0], "scr0": data["scr0"][0],
            "image0": data["image0"][0], "R": data["R"], "t": data["t"], "inliers": data["inliers"],
            "n_matches": model.e2e_Procrustes.num_samples_matches}
    if return_inliers:
        read["inliers_list"] = data["inliers_list"][0]
    return model, data, read


def submission_sequence(model, loader, mat2quat, zip_path):
    results = {}
    for data in loader:
        data = data_to_model_device(data, model)
        with torch.no_grad():
            R_batched, t_batched = model(data)
        for i in range(len(data["scene_id"])):
            R = R_batched[i].unsqueeze(0).detach().cpu().numpy()
            t = t_batched[i].reshape(-1).detach().cpu().numpy()
            inliers = data["inliers"][i].item()
            if np.isnan(R).any() or np.isnan(t).any() or np.isinf(t).any():
                continue
            q = mat2quat(R[0]).reshape(-1)
            fmt = lambda v: " ".join("%.6f" % x for x in v)  # noqa: E731
            results.setdefault(data["scene_id"][i], []).append("%s %s %s %s" % (data["pair_names"][1][i], fmt(q), fmt(t), inliers))
    with zipfile.ZipFile(zip_path, "w") as z:
        for scene, lines in results.items():
            z.writestr("pose_%s.txt" % scene, "\n".join(lines).encode("utf-8"))
    return results
