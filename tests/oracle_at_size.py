"""TEST INFRASTRUCTURE: the oracle (oracle/keymorph_oracle.py, pinned against the reference by tests/golden/) run ONCE at
the metric's volume size on the host cores, and the HIP path run on the same pair and the same weights, compared.

Used by tests/test_fullsize_gpu.py::test_fullsize_vs_oracle_256_affine and by bench.py's cpu_baseline leg (which times the
oracle run and reports the comparison as `parity_at_size`); nothing in keymorph_amd/ imports this.  Restates, in the oracle's
functional form, KeyMorph.forward + align_img + MSELoss + loss.backward() of scripts/train.py:129-176 for one pair
(keymorph/model.py:142-289, keymorph/utils.py:14-21, keymorph/loss_ops.py:9-13)."""
import time

import torch

from tests.util import seeded_state_dict, unet_shapes


def oracle_pair(size, keypoints, threads=32, tt="affine", seed=100, sd_seed=23):
    """One synthetic pair (keymorph_amd.synthetic's recipe evaluated with the ORACLE's sampler on the CPU), seeded weights
    of TruncatedUNet3D(1, K, f_maps 32, 4 levels, 1 truncated), forward + MSE + autograd backward on the host.
    Returns CPU tensors and timings; ~2 min and ~50 GB of host RAM at 256^3."""
    from keymorph_amd import synthetic
    from oracle import keymorph_oracle as O
    if threads:
        torch.set_num_threads(threads)
    cpu = torch.device("cpu")
    img_f = synthetic.blob_volume(size, seed, cpu)
    g = O.affine_grid(torch.inverse(synthetic.random_affine_matrix(seed, cpu)), (size,) * 3)
    img_m = O.align_img(g, img_f)
    del g
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(unet_shapes(keypoints, 32, trunc=1), sd_seed).items()}
    t0 = time.time()
    r = O.keymorph_forward(lambda x: O.unet3d_forward(sd, x, 4, 1, 8), img_f, img_m, tt)
    t_fwd = time.time() - t0
    img_a = O.align_img(r["grid"], img_m)
    mse = O.mse_loss(img_f, img_a)
    mse.backward()
    dt = time.time() - t0
    out = {"img_f": img_f, "img_m": img_m, "sd": {k: v.detach() for k, v in sd.items()},
           "grads": {k: v.grad.detach() for k, v in sd.items()}, "points_f": r["points_f"].detach(),
           "points_m": r["points_m"].detach(), "grid": r["grid"].detach(), "img_a": img_a.detach(), "mse": float(mse.detach()),
           "seconds": dt, "forward_seconds": t_fwd, "size": size, "keypoints": keypoints, "transform": tt}
    if "matrix" in r:
        out["matrix"] = r["matrix"].detach()
    return out


def hip_model(sd, keypoints, dev):
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    net = TruncatedUNet3D(1, keypoints, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    net.load_state_dict({k: v.detach().clone() for k, v in sd.items()}, strict=True)
    return KeyMorph(net, keypoints, 3, max_train_keypoints=None).to(dev).train()


def compare_with_hip(ref, dev="cuda"):
    """The HIP path (the bench's train step minus the optimizer) on `ref`'s pair and weights.  Returns max-abs differences of
    keypoints / matrix / grid / warped volume, |MSE difference|, and relative-L2 gradient differences (whole vector, per tensor)."""
    from keymorph_amd import ops
    tt, K = ref["transform"], ref["keypoints"]
    km = hip_model(ref["sd"], K, dev)
    f, m = ref["img_f"].to(dev), ref["img_m"].to(dev)
    r = km(f, m, transform_type=tt, return_aligned_points=False)[tt]
    loss, img_a = ops.warp_mse(m, r["grid"], f)
    loss.backward()
    mx = lambda a, b: float((a.detach().cpu().double() - b.double()).abs().max())      # noqa: E731
    out = {"size": ref["size"], "keypoints_n": K, "transform": tt,
           "keypoints": max(mx(r["points_f"], ref["points_f"]), mx(r["points_m"], ref["points_m"])),
           "grid": mx(r["grid"], ref["grid"]), "warped": mx(img_a, ref["img_a"]),
           "mse": abs(float(loss.detach()) - ref["mse"]), "mse_oracle": ref["mse"]}
    if "matrix" in ref:
        out["matrix"] = mx(r["matrix"], ref["matrix"])
    per = {}
    num = den = 0.0
    for k, p in km.backbone.named_parameters():
        a, b = p.grad.detach().cpu().double(), ref["grads"][k].double()
        n, d = float((a - b).pow(2).sum()), float(b.pow(2).sum())
        per[k] = (n / (d + 1e-300)) ** 0.5
        num, den = num + n, den + d
    worst = max(per, key=per.get)
    out.update({"gradient_rel_l2": (num / den) ** 0.5, "gradient_worst_tensor": worst, "gradient_worst_rel_l2": per[worst],
                "gradient_per_tensor": per})
    del km
    torch.cuda.empty_cache()
    return out
