"""KeyMorph registration model with the reference's call surface (keymorph/model.py:22-530).

    KeyMorph(backbone, num_keypoints, dim, keypoint_layer="com", max_train_keypoints=None, use_amp=False,
             use_checkpoint=False, weight_keypoints=None, align_keypoints_in_real_world_coords=False,
             max_rand_tps_lmbda=10)
    model(img_f, img_m, transform_type="affine" | "rigid" | "tps_<lambda>" | [..], return_aligned_points=bool,
          aff_f=..., aff_m=..., **ignored) -> {type: {"grid", "points_f", "points_m", "points_weights",
          "tps_lmbda", "time_keypoint_extract", "time_align", "time", ["matrix"], ["points_a"]}}

Differences that are deliberate: any batch size works (row i == the reference's bs=1
result for pair i); fixed and moving volumes go through the backbone as ONE batch; the
TPS system is solved on the GPU once per direction instead of 3x per call on the host (F6).
"""
from __future__ import annotations

import os
import re
import time

import numpy as np
import torch
import torch.nn as nn

from .keypoint_aligners import AffineKeypointAligner, RigidKeypointAligner, TPS
from .layers import CenterOfMass2d, CenterOfMass3d
from .utils import str_or_float


class KeyMorph(nn.Module):
    def __init__(self, backbone, num_keypoints, dim, keypoint_layer="com", max_train_keypoints=None, use_amp=False,
                 use_checkpoint=False, weight_keypoints=None, align_keypoints_in_real_world_coords=False,
                 max_rand_tps_lmbda=10):
        super().__init__()
        self.backbone = backbone
        self.num_keypoints = num_keypoints
        self.dim = dim
        if keypoint_layer != "com":
            # keymorph/layers.py:6-27 (LinearRegressor*) is broken upstream (self.num_keypoints undefined)
            raise NotImplementedError("only the center-of-mass keypoint layer is implemented")
        self.keypoint_layer = CenterOfMass2d(indexing="ij") if dim == 2 else CenterOfMass3d(indexing="ij")
        self.max_train_keypoints = max_train_keypoints
        # use_amp (keymorph/model.py:176-191: the reference runs the keypoint extractor under fp16 autocast): the one-product
        # fp16 arithmetic of the backbone's matrix kernels -- fp16 inputs (11 significant bits), fp32 accumulation and fp32
        # tensors, a third of the MFMA work -- instead of the fp32-class split-operand default.  Per call, not process-wide:
        # get_keypoints() runs the backbone inside backbone_ops.amp_scope(self.use_amp), every operator records the setting
        # of its forward and its backward re-opens it (another model, with another setting, between the two changes nothing);
        # aligners, warp and losses are fp32 either way, as in the reference.
        self.use_amp = bool(use_amp)
        self.use_checkpoint = use_checkpoint
        self.max_rand_tps_lmbda = max_rand_tps_lmbda
        self.supported_transform_type = ["rigid", "affine", "tps"]
        assert weight_keypoints in [None, "variance", "power"]
        self.weight_keypoints = weight_keypoints
        if weight_keypoints == "variance":           # model.py:70-72 (same parameter names: checkpoints load)
            self.scales = nn.Parameter(torch.ones(num_keypoints))
            self.biases = nn.Parameter(torch.zeros(num_keypoints))
        self.align_keypoints_in_real_world_coords = align_keypoints_in_real_world_coords

    # ------------------------------------------------------------------
    def get_keypoints(self, img, return_feat=False):
        """model.py:111-117"""
        if img.is_cuda:
            from . import backbone_ops
            with backbone_ops.amp_scope(self.use_amp):
                return self._get_keypoints(img, return_feat)
        return self._get_keypoints(img, return_feat)

    def _get_keypoints(self, img, return_feat):
        net = getattr(self.backbone, "module", self.backbone)   # nn.DataParallel wrapper (run.py:390)
        if (not return_feat and self.dim == 3 and hasattr(net, "keypoints_ij")
                and (net.final_activation is None or net.training)):
            return net.keypoints_ij(img)       # fused 1x1x1 head + ReLU + center of mass, no heat-map
        feat = self.backbone(img)
        points = self.keypoint_layer(feat)
        if return_feat:
            return points, feat
        return points

    def _keypoint_weights(self, power_f, power_m, sq_f, sq_m, nvox):
        """model.py:75-109 from per-channel moments of relu(heat-map): 'power' = product of the two heat-map
        masses, 'variance' = product of 1 / (scale * var + bias) with torch.var's unbiased estimator; normalised
        per sample (the reference divides by sum(dim=1) without keepdim, which is the same thing at bs = 1)."""
        if self.weight_keypoints == "power":
            w = power_f * power_m
        else:
            def var(s1, s2):
                s1, s2 = s1.double(), s2.double()
                return ((s2 - s1 * s1 / nvox) / (nvox - 1)).float()
            w = 1.0 / (self.scales * var(power_f, sq_f) + self.biases) / (self.scales * var(power_m, sq_m) + self.biases)
        return w / w.sum(dim=1, keepdim=True)

    def _convert_tps_lmbda(self, num_samples, tps_lmbda):
        """model.py:119-132"""
        if tps_lmbda == "uniform":
            return torch.rand(num_samples) * self.max_rand_tps_lmbda
        if tps_lmbda == "loguniform":
            from scipy.stats import loguniform
            return torch.tensor(loguniform.rvs(1e-6, self.max_rand_tps_lmbda, size=num_samples))
        return torch.tensor(tps_lmbda).repeat(num_samples)

    def _tps_lmbda_on(self, num_samples, tps_lmbda, device):
        """_convert_tps_lmbda(...).to(device).float(); a FIXED lambda is created on the device once per (value, count) -- the
        host-to-device copy of a fresh CPU tensor is a blocking point in every step otherwise.  Callers get a device-side
        CLONE of the cached tensor (no host synchronisation): an in-place edit downstream cannot corrupt later steps, and
        the clone is an ordinary tensor even when the cache entry was first created under `torch.inference_mode()` (an
        inference tensor cannot be saved for a later backward).  dtype: float32 always (the reference passes float64 for
        the random-lambda strings and casts inside `TPS.fit`)."""
        if isinstance(tps_lmbda, str):
            return self._convert_tps_lmbda(num_samples, tps_lmbda).to(device).float()
        if torch.is_inference_mode_enabled():      # (nothing created here may enter the cache)
            return self._convert_tps_lmbda(num_samples, tps_lmbda).to(device).float()
        cache = self.__dict__.setdefault("_lmbda_cache", {})
        key = (float(tps_lmbda), int(num_samples), str(device))
        t = cache.get(key)
        if t is None:
            t = cache[key] = self._convert_tps_lmbda(num_samples, tps_lmbda).to(device).float()
        return t.clone()

    @staticmethod
    def is_supported_transform_type(s):
        return s in ["affine", "rigid"] or bool(re.match(r"^tps_.*$", s))

    def _make_aligner(self, align_type, points_m, points_f, tps_lmbda, weights, aff):
        aff_f, aff_m, shape_f, shape_m = aff
        common = dict(points_m=points_m, points_f=points_f, w=weights, aff_f=aff_f, aff_m=aff_m, shape_f=shape_f,
                      shape_m=shape_m, dim=self.dim,
                      align_in_real_world_coords=self.align_keypoints_in_real_world_coords)
        if align_type == "rigid":
            return RigidKeypointAligner(**common)
        if align_type == "affine":
            return AffineKeypointAligner(**common)
        return TPS(lmbda=tps_lmbda, use_checkpoint=self.use_checkpoint, **common)

    # ------------------------------------------------------------------
    def forward(self, img_f, img_m, transform_type="affine", **kwargs):
        """model.py:142-289"""
        return_aligned_points = kwargs["return_aligned_points"]
        if not isinstance(transform_type, (list, tuple)):
            transform_type = [transform_type]
        if self.training:
            assert len(transform_type) == 1, "Only one alignment type allowed in training"
        assert all(self.is_supported_transform_type(s) for s in transform_type), "Invalid transform_type"

        if self.align_keypoints_in_real_world_coords:
            aff = (kwargs["aff_f"], kwargs["aff_m"], torch.tensor(img_f.shape[2:]).to(img_f),
                   torch.tensor(img_m.shape[2:]).to(img_m))
        else:
            aff = (None, None, None, None)
        assert img_f.shape[1] == 1, "Image dimension must be 1"
        assert img_m.shape[1] == 1, "Image dimension must be 1"

        start_time = time.time()
        weights = None
        if self.weight_keypoints == "power":
            # (upstream applies only "power" in forward -- model.py:183-193; with "variance" the scales / biases
            # parameters exist but the weights stay None, which is mirrored here)
            net = getattr(self.backbone, "module", self.backbone)
            if not hasattr(net, "keypoints_and_power"):
                raise NotImplementedError("keypoint weighting needs a backbone with the fused keypoint head")
            nf = img_f.shape[0]
            if img_f.shape == img_m.shape:
                pts, power = net.keypoints_and_power(torch.cat([img_f, img_m], dim=0))
                points_f, points_m = pts[:nf], pts[nf:]
                weights = self._keypoint_weights(power[:nf], power[nf:], None, None, None)
            else:
                points_f, pf = net.keypoints_and_power(img_f)
                points_m, pm = net.keypoints_and_power(img_m)
                weights = self._keypoint_weights(pf, pm, None, None, None)
        elif img_f.shape == img_m.shape:
            # one backbone pass over [fixed; moving] (norms are per-sample, so results are unchanged)
            pts = self.get_keypoints(torch.cat([img_f, img_m], dim=0))
            points_f, points_m = pts[: img_f.shape[0]], pts[img_f.shape[0]:]
        else:
            points_f = self.get_keypoints(img_f)
            points_m = self.get_keypoints(img_m)
        keypoint_extract_time = time.time() - start_time

        # Evaluation with several thin-plate-spline types (pairwise_register_eval.py:116-171 passes a list): every fit the
        # loop below would launch one after the other -- (lambda, direction) per type, one CU each -- is solved in ONE
        # batched launch up front; the aligners find their coefficients already in place.
        prefit = {}
        tps_types = [t for t in transform_type if t.startswith("tps") and not isinstance(str_or_float(t[4:]), str)]
        # (numeric lambdas only: "uniform" / "loguniform" draw random values, which must be drawn once, in the loop)
        if (not self.training and not torch.is_grad_enabled() and not self.align_keypoints_in_real_world_coords
                and len(tps_types) * (2 if return_aligned_points else 1) > 1):
            from . import ops
            n = len(img_f)
            ctrl, tgt, lam = [], [], []
            for t in tps_types:
                lm = self._tps_lmbda_on(n, str_or_float(t[4:]), img_f.device)
                ctrl.append(points_f); tgt.append(points_m); lam.append(lm)             # inverse map: fixed -> moving
                if return_aligned_points:
                    ctrl.append(points_m); tgt.append(points_f); lam.append(lm)         # forward map: moving -> fixed
            theta = ops.tps_fit(torch.cat(ctrl), torch.cat(tgt), torch.cat(lam),
                                None if weights is None else weights.repeat(len(ctrl), 1))
            per = 2 if return_aligned_points else 1
            for i, t in enumerate(tps_types):
                prefit[t] = (theta[(per * i) * n:(per * i + 1) * n],
                             theta[(per * i + 1) * n:(per * i + 2) * n] if return_aligned_points else None)

        result_dict = {}
        for align_type_str in transform_type:
            start_time = time.time()
            if align_type_str.startswith("tps"):
                align_type = "tps"
                tps_lmbda = self._tps_lmbda_on(len(img_f), str_or_float(align_type_str[4:]), img_f.device)
            else:
                align_type = align_type_str
                tps_lmbda = None
            if (self.training and align_type == "tps" and self.max_train_keypoints
                    and self.num_keypoints > self.max_train_keypoints):
                idx = np.random.choice(self.num_keypoints, size=self.max_train_keypoints, replace=False)
                points_f = points_f[:, idx]
                points_m = points_m[:, idx]
                if weights is not None:      # model.py:221-222: the weights follow their keypoints
                    weights = weights[:, idx]
            aligner = self._make_aligner(align_type, points_m, points_f, tps_lmbda, weights, aff)
            if align_type_str in prefit:
                aligner._inverse_theta, aligner.theta = prefit[align_type_str]
            grid = aligner.get_flow_field(img_f.shape, compute_on_subgrids=not self.training)
            if return_aligned_points:
                points_a = aligner.get_forward_transformed_points(points_m)
            align_time = time.time() - start_time
            res = {
                "grid": grid,
                "points_f": points_f,
                "points_m": points_m,
                "points_weights": weights,
                "tps_lmbda": tps_lmbda,
                "time_keypoint_extract": keypoint_extract_time,
                "time_align": align_time,
                "time": keypoint_extract_time + align_time,
            }
            if align_type in ["rigid", "affine"]:
                res["matrix"] = aligner.transform_matrix
            if return_aligned_points:
                res["points_a"] = points_a
            result_dict[align_type_str] = res
        return result_dict

    def pairwise_register(self, *args, **kwargs):
        """Alias for forward()."""
        return self.forward(*args, **kwargs)

    # ------------------------------------------------------------------
    def groupwise_register(self, inputs, transform_type="affine", **kwargs):
        """Iterative mean-keypoint groupwise registration (model.py:295-530).

        ``inputs``: directory with ``*.npz`` files (key ``img``, (1,1,D,H,W)), a list of such paths, or a
        tensor stack (N,1,D,H,W).  Grids are written to ``save_dir`` as ``{type}_grid_{i:03}.npy`` when
        ``save_results_to_disk`` is set (the only mode the reference's scripts use), else returned under
        ``"groupgrids"``."""
        device = kwargs["device"]
        num_iters = kwargs["num_iters"]
        log = print if kwargs.get("log_to_console", False) else (lambda *a, **k: None)
        if not isinstance(transform_type, (list, tuple)):
            transform_type = [transform_type]
        if isinstance(inputs, str):
            save_dir = kwargs["save_dir"]
            inputs = sorted(os.path.join(inputs, f) for f in os.listdir(inputs) if f.endswith(".npz"))
            if len(inputs) == 0:
                raise ValueError("No .npz files found")
        else:
            save_dir = kwargs.get("save_dir")

        # One process per GPU (BASELINE config 5): every rank extracts the keypoints of its contiguous block of subjects,
        # ONE all-gather of (n, K, 3) makes the whole set available everywhere, the (cheap) iterations are replicated
        # and every rank produces the grids of its own subjects.  Single process: `mine` is everything.
        from . import parallel
        num_subjects = len(inputs)
        sharded = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1 and kwargs.get("shard_subjects", True)
        mine = (parallel.shard_indices(num_subjects, torch.distributed.get_rank(), torch.distributed.get_world_size())
                if sharded else list(range(num_subjects)))
        group_points = []
        log("Extracting keypoints...")
        img_m = None
        for i in mine:
            if isinstance(inputs[i], str):
                img_m = torch.tensor(np.load(inputs[i])["img"]).float()
            else:
                img_m = inputs[i:i + 1]
            img_m = img_m.to(device)
            group_points.append(self.get_keypoints(img_m).detach())
            log(f"-> Extracted keypoints from subject {i + 1}/{num_subjects}")
        if img_m is None:      # more ranks than subjects: this rank only needs the volume shape
            img_m = (torch.tensor(np.load(inputs[0])["img"]) if isinstance(inputs[0], str) else inputs[0:1])
            local = torch.zeros((0, self.num_keypoints, self.dim), dtype=torch.float32, device=device)
        else:
            local = torch.cat(group_points, dim=0)
        group_points = parallel.gather_group_points(local, num_subjects) if sharded else local
        grid_shape = img_m.shape
        no_aff = (None, None, None, None)

        result_dict = {}
        for align_type_str in transform_type:
            log(f"\nAligning keypoints via {align_type_str}...")
            start_time = time.time()
            if align_type_str.startswith("tps"):
                align_type = "tps"
                tps_lmbda = self._tps_lmbda_on(1, str_or_float(align_type_str[4:]), device)
            else:
                align_type, tps_lmbda = align_type_str, None

            curr_points = group_points.clone()
            mean_points = curr_points.mean(dim=0, keepdim=True)
            n_sub = len(curr_points)
            lm_all = None if tps_lmbda is None else tps_lmbda.reshape(-1)[:1].expand(n_sub).contiguous()
            for j in range(num_iters):
                # model.py:331-444 loops over the subjects; the aligners are batched (row i == the bs = 1 result for
                # subject i), so one fit launch solves all subjects' systems side by side, one CU each
                mean_points = curr_points.mean(dim=0, keepdim=True)
                al = self._make_aligner_plain(align_type, curr_points, mean_points.expand_as(curr_points).contiguous(),
                                              lm_all)
                curr_points = al.get_forward_transformed_points(curr_points)
                log(f"-> Iteration {j + 1}/{num_iters}")
            res = {"time": time.time() - start_time, "grouppoints_m": group_points, "grouppoints_a": curr_points}

            grids = []
            # the final maps (model.py:453-510: mean -> subject): all of this rank's subjects are fitted in ONE launch
            # (one CU per system), each grid is then evaluated and saved / kept on its own
            from . import ops
            if len(mine):
                sel = torch.as_tensor(list(mine), device=group_points.device)
                pm_all = group_points.index_select(0, sel)
                al_all = self._make_aligner_plain(align_type, pm_all, mean_points.expand_as(pm_all).contiguous(),
                                                  None if tps_lmbda is None else lm_all[:len(mine)].contiguous())
            for j, i in enumerate(mine):
                if align_type == "tps":
                    grid = ops.tps_grid(al_all.inverse_theta[j:j + 1].contiguous(), al_all.points_f[j:j + 1].contiguous(),
                                        grid_shape[2:])
                else:
                    grid = ops.affine_grid(al_all.inverse_transform_matrix[j:j + 1, :3, :].contiguous(), grid_shape[2:])
                if kwargs.get("save_results_to_disk") and save_dir:
                    path = f"{save_dir}/{align_type_str}_grid_{i:03}.npy"
                    log(f"-> Saving grid {i + 1}/{len(curr_points)} to {path}")
                    np.save(path, grid.cpu().detach().numpy())
                else:
                    grids.append(grid)
            if grids:
                res["groupgrids"] = torch.cat(grids, dim=0)
            if sharded:
                res["grid_subjects"] = list(mine)       # which subjects this rank's grids / files belong to
            result_dict[align_type_str] = res
        log("Groupwise registration complete!")
        return result_dict

    def _make_aligner_plain(self, align_type, points_m, points_f, tps_lmbda):
        saved = self.align_keypoints_in_real_world_coords
        self.align_keypoints_in_real_world_coords = False
        try:
            return self._make_aligner(align_type, points_m, points_f, tps_lmbda, None, (None, None, None, None))
        finally:
            self.align_keypoints_in_real_world_coords = saved


# keymorph/model.py:533-640 also holds a brain-mask U-Net and its post-processing, unrelated to registration: names only.
from ._absent import absent_class as _absent_class, absent_function as _absent_function   # noqa: E402

Simple_Unet = _absent_class("Simple_Unet", "keymorph/model.py:533", nn.Module)
simple_block = _absent_class("simple_block", "keymorph/model.py:598", nn.Module)
clean_mask = _absent_function("clean_mask", "keymorph/model.py:622")
