"""On-disk products of a groupwise evaluation, as scripts/groupwise_register_eval.py:346-431, 478-527 writes them:
    <group_dir>/img_m/*.npz ("img", (1,1,D,H,W))           in
    <group_dir>/seg_m/*.npz ("seg", (1,C,D,H,W) one-hot)   in (optional)
    <group_dir>/registration_results/{type}_grid_{i:03}.npy        KeyMorph.groupwise_register(save_results_to_disk=True)
    <group_dir>/img_a_{type}/img_a_{type}_{i:03}.npy               aligned images
    <group_dir>/seg_a_{type}/seg_a_{type}_{i:03}.npy               aligned segmentations (bilinear)
    <group_dir>/metrics-{type}.json                                {"mse", "softdice", "harddice", "harddiceroi", "jdstd", "jdlessthan0"}
    <group_dir>/points_m-{aug}.npy, points_a-{aug}-{type}.npy      keypoints of the first subject before / after
Host-side file plumbing; every warp and metric runs on the GPU (align_img, loss_ops)."""
import json
import os

import numpy as np
import torch

from .. import loss_ops
from ..utils import align_img


def save_dict_as_json(d, save_path):
    """scripts/script_utils.py:118-120"""
    with open(save_path, "w") as f:
        json.dump(d, f, sort_keys=True, indent=4)


def evaluate_group(registration_model, group_dir, transform_types, device, metrics=("mse", "softdice", "harddice",
                   "jdstd", "jdlessthan0"), seg_available=True, num_iters=5, aug="rot0", log=None):
    """groupwise_register_eval.py:375-527 for one group directory; returns {type: metrics dict}."""
    group_dir = str(group_dir)
    log = log or (lambda *a: None)
    img_dir, seg_dir = os.path.join(group_dir, "img_m"), os.path.join(group_dir, "seg_m")
    res_dir = os.path.join(group_dir, "registration_results")
    os.makedirs(res_dir, exist_ok=True)
    img_paths = sorted(os.path.join(img_dir, f) for f in os.listdir(img_dir))
    seg_paths = sorted(os.path.join(seg_dir, f) for f in os.listdir(seg_dir)) if seg_available else []
    with torch.no_grad():
        results = registration_model.groupwise_register(img_dir, transform_type=list(transform_types), device=device,
                                                        save_results_to_disk=True, save_dir=res_dir, plot=False,
                                                        num_iters=num_iters, log_to_console=False,
                                                        shard_subjects=False)   # every rank needs every grid below
    out = {}
    for tt, res in results.items():
        # exact prefix: "tps_0" must not pick up the grids of "tps_0.1"
        grids = sorted(os.path.join(res_dir, f) for f in os.listdir(res_dir) if f.startswith(f"{tt}_grid_"))
        assert len(grids) == len(img_paths), (tt, len(grids), len(img_paths))
        img_a_dir, seg_a_dir = os.path.join(group_dir, f"img_a_{tt}"), os.path.join(group_dir, f"seg_a_{tt}")
        os.makedirs(img_a_dir, exist_ok=True)
        os.makedirs(seg_a_dir, exist_ok=True)
        img_a_paths, seg_a_paths = [], []
        for i, p in enumerate(img_paths):
            grid = torch.tensor(np.load(grids[i])).to(device)
            img_a = align_img(grid, torch.tensor(np.load(p)["img"]).float().to(device))
            img_a_paths.append(os.path.join(img_a_dir, f"img_a_{tt}_{i:03}.npy"))
            np.save(img_a_paths[-1], img_a.cpu().numpy())
            if seg_available:
                seg_a = align_img(grid, torch.tensor(np.load(seg_paths[i])["seg"]).float().to(device))
                seg_a_paths.append(os.path.join(seg_a_dir, f"seg_a_{tt}_{i:03}.npy"))
                np.save(seg_a_paths[-1], seg_a.cpu().numpy())
        m, seg_names, grid_names = {}, [], []
        for name in metrics:
            if name == "mse":
                m["mse"] = loss_ops.MSEPairwiseLoss()(img_a_paths).item()
            elif name in ("softdice", "harddice", "harddiceroi"):
                assert seg_available
                seg_names.append(name)
            elif name in ("jdstd", "jdlessthan0"):
                grid_names.append(name)
            else:
                raise ValueError('Invalid metric "{}"'.format(name))
        seg_m = loss_ops.MultipleAvgSegPairwiseMetric()(seg_a_paths, seg_names) if seg_names else {}
        for k in ("harddice", "softdice"):
            if k in seg_m:
                seg_m[k] = (1 - seg_m[k]).item()
        if "harddiceroi" in seg_m:
            seg_m["harddiceroi"] = (1 - seg_m["harddiceroi"]).tolist()
        grid_m = loss_ops.MultipleAvgGridMetric()(grids, grid_names) if grid_names else {}
        m = m | seg_m | grid_m
        res["metrics"] = m
        save_dict_as_json(m, os.path.join(group_dir, f"metrics-{tt}.json"))
        pm_path = os.path.join(group_dir, f"points_m-{aug}.npy")
        if not os.path.exists(pm_path):
            np.save(pm_path, res["grouppoints_m"][0].cpu().numpy())
        np.save(os.path.join(group_dir, f"points_a-{aug}-{tt}.npy"), res["grouppoints_a"][0].cpu().numpy())
        log(f"{tt}: {m}")
        out[tt] = m
    return out
