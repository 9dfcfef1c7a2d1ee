"""TEST INFRASTRUCTURE: the oracle (oracle/keymorph_oracle.py, pinned against the reference by tests/golden/) run ONCE at
the metric's volume size on the host cores, and the HIP path run on the same pair and the same weights, compared.

Used by tests/test_fullsize_gpu.py::test_fullsize_vs_oracle_256_affine and by bench.py's cpu_baseline leg (which times the
oracle run and reports the comparison as `parity_at_size`); nothing in keymorph_amd/ imports this.  Restates, in the oracle's
functional form, KeyMorph.forward + align_img + MSELoss + loss.backward() of scripts/train.py:129-176 for one pair
(keymorph/model.py:142-289, keymorph/utils.py:14-21, keymorph/loss_ops.py:9-13).

The backward is taken in the two pieces the chain rule gives, so that they can be judged separately:
  tail      (keypoints -> fit -> grid -> warp -> MSE): d(loss)/d(keypoints), 2 x K x 3 numbers.  Cheap, and badly conditioned
            (a sum over 16.8 M voxels pushed through the inverse of a fit to clumped keypoints), so it is ALSO run in fp64
            from the same fp32 keypoints: the "truth" both fp32 tails are measured against;
  backbone  (image -> keypoints): the 24 TFLOP part.  The oracle's autograd runs it ONCE with its own fp32 tail gradient as
            the cotangent -- arithmetically the same as loss.backward() end to end."""
import time

import torch

from tests.util import seeded_state_dict, unet_shapes


def _tail(O, pf, pm, img_f, img_m, tt):
    """-> (register() result, warped, mse, d mse / d points_f, d mse / d points_m) in the dtype of the inputs"""
    pf = pf.detach().clone().requires_grad_(True)
    pm = pm.detach().clone().requires_grad_(True)
    r = O.register(pf, pm, tt, img_f.shape[2:])
    img_a = O.align_img(r["grid"], img_m)
    mse = O.mse_loss(img_f, img_a)
    dpf, dpm = torch.autograd.grad(mse, [pf, pm])
    return r, img_a.detach(), mse.detach(), dpf, dpm


def oracle_pair(size, keypoints, threads=32, tt="affine", seed=100, sd_seed=23, in_subprocess=True):
    """One synthetic pair (keymorph_amd.synthetic's recipe evaluated with the ORACLE's sampler on the CPU), seeded weights
    of TruncatedUNet3D(1, K, f_maps 32, 4 levels, 1 truncated), forward + MSE + autograd backward on the host.
    Returns CPU tensors and timings; ~2 min and ~50 GB of host RAM at 256^3.
    in_subprocess (default): the run happens in a CHILD interpreter with its own thread pool and the result comes back
    through a file.  Changing torch's intra-op thread count inside a long-lived process is not safe to rely on: with the
    oracle run in-process, a later CPU `torch.linalg.solve` of the same pytest session returned a wrong answer on the MI355X
    box (tests/test_ops_gpu.py::test_tps_fit_vs_fp64[200-1.0], only after this test), and the child also returns its 50 GB."""
    if in_subprocess:
        import os
        import subprocess
        import sys
        import tempfile
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "oracle_pair.pt")
            env = dict(os.environ, OMP_NUM_THREADS=str(threads or os.cpu_count()), MKL_NUM_THREADS=str(threads or os.cpu_count()))
            code = (f"import sys; sys.path.insert(0, {root!r}); import torch; from tests.oracle_at_size import oracle_pair; "
                    f"torch.save(oracle_pair({size}, {keypoints}, {threads}, {tt!r}, {seed}, {sd_seed}, in_subprocess=False), {out!r})")
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("the oracle child process failed:\n" + r.stderr[-3000:])
            return torch.load(out, weights_only=False)
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
    pf = O.center_of_mass(O.unet3d_forward(sd, img_f, 4, 1, 8), "ij")           # keymorph/model.py:111-117
    pm = O.center_of_mass(O.unet3d_forward(sd, img_m, 4, 1, 8), "ij")
    r, img_a, mse, dpf, dpm = _tail(O, pf, pm, img_f, img_m, tt)
    t_fwd = time.time() - t0
    torch.autograd.backward([pf, pm], [dpf, dpm])                               # == mse.backward() end to end
    dt = time.time() - t0
    # the same tail in fp64 from the same (fp32) keypoints: what both fp32 tails are measured against
    r64, _, mse64, dpf64, dpm64 = _tail(O, pf.double(), pm.double(), img_f.double(), img_m.double(), tt)
    out = {"img_f": img_f, "img_m": img_m, "sd": {k: v.detach() for k, v in sd.items()},
           "grads": {k: v.grad.detach() for k, v in sd.items()}, "points_f": pf.detach(), "points_m": pm.detach(),
           "grid": r["grid"].detach(), "img_a": img_a, "mse": float(mse), "dpoints": torch.stack([dpf, dpm]),
           "dpoints_fp64": torch.stack([dpf64, dpm64]), "mse_fp64": float(mse64), "grid_fp64_err": float(
               (r["grid"].detach().double() - r64["grid"].detach()).abs().max()),
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


def _grad_table(named_params, ref_grads):
    per, num, den = {}, 0.0, 0.0
    for k, p in named_params:
        a, b = p.grad.detach().cpu().double(), ref_grads[k].double()
        n, d = float((a - b).pow(2).sum()), float(b.pow(2).sum())
        per[k] = (n / (d + 1e-300)) ** 0.5
        num, den = num + n, den + d
    return (num / den) ** 0.5, per


def compare_with_hip(ref, dev="cuda", mode=None, extras=True):
    """The HIP path (the bench's train step minus the optimizer) on `ref`'s pair and weights.  Returns max-abs differences of
    keypoints / matrix / grid / warped volume, |MSE difference|, and relative-L2 gradient differences:
      gradient_rel_l2            end to end (HIP loss.backward() vs the oracle's), whole vector + per tensor;
      backbone_gradient_rel_l2   the backbone's backward alone: HIP and oracle given the SAME cotangent d(loss)/d(keypoints)
                                 (the oracle's fp32 one);
      tail_rel_l2_{hip,oracle}   d(loss)/d(keypoints) of each against the fp64 tail.
    mode: run the comparison under another convolution arithmetic (backbone_ops.set_conv_mode), restored afterwards."""
    from keymorph_amd import backbone_ops, ops
    if mode is not None:
        # another arithmetic of the 27-tap / 1x1x1 kernels ("f32": the exact fp32 MFMA) on the same pair: separates what the
        # split-operand arithmetic contributes from what the implementation does (no forward-only extra legs)
        old_mode = backbone_ops.CONV_MODE
        backbone_ops.set_conv_mode(mode)
        try:
            return compare_with_hip(ref, dev, mode=None, extras=False)
        finally:
            backbone_ops.set_conv_mode(old_mode)
    tt, K = ref["transform"], ref["keypoints"]
    km = hip_model(ref["sd"], K, dev)
    f, m = ref["img_f"].to(dev), ref["img_m"].to(dev)
    r = km(f, m, transform_type=tt, return_aligned_points=False)[tt]
    got = {}
    r["points_f"].register_hook(lambda g: got.__setitem__("f", g.detach().clone()))
    r["points_m"].register_hook(lambda g: got.__setitem__("m", g.detach().clone()))
    loss, img_a = ops.warp_mse(m, r["grid"], f)
    loss.backward()
    mx = lambda a, b: float((a.detach().cpu().double() - b.double()).abs().max())      # noqa: E731
    out = {"size": ref["size"], "keypoints_n": K, "transform": tt,
           "keypoints": max(mx(r["points_f"], ref["points_f"]), mx(r["points_m"], ref["points_m"])),
           "grid": mx(r["grid"], ref["grid"]), "warped": mx(img_a, ref["img_a"]),
           "mse": abs(float(loss.detach()) - ref["mse"]), "mse_oracle": ref["mse"],
           "oracle_fp32_vs_fp64_grid": ref["grid_fp64_err"], "oracle_fp32_vs_fp64_mse": abs(ref["mse"] - ref["mse_fp64"])}
    if "matrix" in ref:
        out["matrix"] = mx(r["matrix"], ref["matrix"])
    e2e, per = _grad_table(km.backbone.named_parameters(), ref["grads"])
    worst = max(per, key=per.get)
    out.update({"gradient_rel_l2": e2e, "gradient_worst_tensor": worst, "gradient_worst_rel_l2": per[worst],
                "gradient_per_tensor": per})
    # the tails against fp64
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())                      # noqa: E731
    dp_hip = torch.stack([got["f"], got["m"]]).cpu()
    out["tail_rel_l2_hip"] = rel(dp_hip, ref["dpoints_fp64"])
    out["tail_rel_l2_oracle"] = rel(ref["dpoints"], ref["dpoints_fp64"])
    # the backbone's backward alone, on the oracle's cotangent
    km.zero_grad(set_to_none=True)
    pts = km.get_keypoints(torch.cat([f, m]))
    torch.autograd.backward([pts], [ref["dpoints"].reshape(pts.shape).to(dev)])
    bb, per_bb = _grad_table(km.backbone.named_parameters(), ref["grads"])
    wb = max(per_bb, key=per_bb.get)
    out.update({"backbone_gradient_rel_l2": bb, "backbone_gradient_worst_tensor": wb,
                "backbone_gradient_worst_rel_l2": per_bb[wb], "backbone_gradient_per_tensor": per_bb})
    if extras:
        out.update(extra_legs(ref, km, f, m, dev))
    del km
    torch.cuda.empty_cache()
    return out


def extra_legs(ref, km, f, m, dev, samples=4096):
    """Forward-only legs on the same pair, weights and (oracle) keypoints:
      tps_1 / tps_0  the HIP grid at `samples` random voxels against the oracle's TPS (keypoint_aligners.py:276-433) on the
                     ORACLE's keypoints, in fp32 (the reference arithmetic) and in fp64 (truth).  lambda = 0 with 512
                     clumped keypoints is the metric's own configuration and is ill-conditioned: the statement
                     there is |ours - truth| <= max(1e-4, 1.25 |reference fp32 - truth|), reported as tps_0_* below;
      dice           soft Dice of a 14-class one-hot segmentation pair warped by the affine grid: the fused
                     loss_ops.warp_dice_loss (nothing materialised) against the oracle's align_img + DiceLoss
                     (utils.py:14-21, loss_ops.py:16-63) -- north_star: "matching reference Dice within 1e-4"."""
    from keymorph_amd import loss_ops, utils
    from keymorph_amd.keypoint_aligners import TPS
    from oracle import keymorph_oracle as O
    size, out = ref["size"], {}
    gen = torch.Generator().manual_seed(5)
    idx = [torch.randint(0, size, (samples,), generator=gen) for _ in range(3)]
    g32 = O.base_grid((size,) * 3)[idx[0], idx[1], idx[2]].reshape(1, -1, 3)
    pf, pm = ref["points_f"], ref["points_m"]
    km.eval()
    try:
        for lam in (1.0, 0.0):
            tt = f"tps_{lam:g}"
            with torch.no_grad():
                grid = km(f, m, transform_type=tt, return_aligned_points=False)[tt]["grid"]
            got = grid[0][idx[0].to(dev), idx[1].to(dev), idx[2].to(dev)].cpu().double()
            lm = torch.full((1,), lam)
            w32 = O.tps_transform_points(O.tps_fit(pf, pm, lm), pf, g32).flip(-1)[0].double()
            w64 = O.tps_transform_points(O.tps_fit(pf.double(), pm.double(), lm.double()), pf.double(), g32.double()).flip(-1)[0]
            # the aligner alone on the ORACLE's keypoints (what the F7 statement is about: the 1.5e-6 keypoint difference of the
            # end-to-end run is amplified by the same conditioning and is not the aligner's arithmetic)
            with torch.no_grad():
                ga = TPS(points_m=pm.to(dev), points_f=pf.to(dev), lmbda=lm.to(dev)).get_flow_field(f.shape)
            gota = ga[0][idx[0].to(dev), idx[1].to(dev), idx[2].to(dev)].cpu().double()
            out[f"{tt}_e2e_grid_vs_oracle_fp32"] = float((got - w32).abs().max())
            out[f"{tt}_e2e_grid_vs_fp64"] = float((got - w64).abs().max())
            out[f"{tt}_grid_vs_oracle_fp32"] = float((gota - w32).abs().max())
            out[f"{tt}_grid_vs_fp64"] = float((gota - w64).abs().max())
            out[f"{tt}_oracle_fp32_vs_fp64"] = float((w32 - w64).abs().max())
    finally:
        km.train()
    # Dice on a 14-class pair: labels = intensity bands of the two volumes (bench.py::synthetic_segmentation's recipe)
    C = 14
    lab = lambda v: torch.clamp((v * C).long(), 0, C - 1)                               # noqa: E731
    seg_f = torch.zeros(1, C, size, size, size).scatter_(1, lab(ref["img_f"]), 1.0)
    seg_m = torch.zeros(1, C, size, size, size).scatter_(1, lab(ref["img_m"]), 1.0)
    want = float(O.dice_loss(O.align_img(ref["grid"], seg_m), seg_f))
    with torch.no_grad():
        aff = km(f, m, transform_type="affine", return_aligned_points=False)["affine"]["grid"]
        got_fused = float(loss_ops.warp_dice_loss(aff, seg_m.to(dev), seg_f.to(dev)))
        got_plain = float(loss_ops.DiceLoss()(utils.align_img(aff, seg_m.to(dev)), seg_f.to(dev)))
    out.update({"dice_oracle": want, "dice_fused": abs(got_fused - want), "dice_unfused": abs(got_plain - want)})
    return out


def _child(call, threads):
    """evaluate `tests.oracle_at_size.<call>` in a child interpreter with its own thread pool; returns what it returned"""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "result.pt")
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        code = (f"import sys; sys.path.insert(0, {root!r}); import torch; torch.set_num_threads({threads}); "
                f"import tests.oracle_at_size as M; torch.save(M.{call}, {out!r})")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("the oracle child process failed:\n" + r.stderr[-3000:])
        return torch.load(out, weights_only=False)


def oracle_convnet(size, keypoints, sd_seed=77, vol_seed=11, cot_seed=4, threads=32, in_subprocess=True):
    """ConvNet(instance norm) of keymorph/net.py:7-36 on one blob volume, center of mass, and the autograd gradients of
    sum(keypoints * cot) with respect to every weight -- on the host, in a child process by default."""
    if in_subprocess:
        return _child(f"oracle_convnet({size}, {keypoints}, {sd_seed}, {vol_seed}, {cot_seed}, {threads}, False)", threads)
    from keymorph_amd import synthetic
    from oracle import keymorph_oracle as O
    from tests.util import convnet_shapes
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(convnet_shapes(keypoints), sd_seed).items()}
    x = synthetic.blob_volume(size, vol_seed, torch.device("cpu"))
    y = O.convnet_forward(sd, x, "instance")
    pts = O.center_of_mass(y, "ij")
    cot = torch.randn(pts.shape, generator=torch.Generator().manual_seed(cot_seed))
    (pts * cot).sum().backward()
    return {"x": x, "sd": {k: v.detach() for k, v in sd.items()}, "y": y.detach(), "pts": pts.detach(), "cot": cot,
            "grads": {k: v.grad.detach() for k, v in sd.items()}}


def oracle_backbone_fp64(size, keypoints, sd_seed=23, vol_seed=100, cot_seed=6, threads=32, in_subprocess=True):
    """The BACKBONE's backward (image -> TruncatedUNet3D -> center of mass, keymorph/model.py:111-117) for one blob volume and
    one seeded cotangent d(loss)/d(keypoints), by the oracle's autograd in fp32 (the reference arithmetic) AND in fp64 (truth)
    from the same fp32 weights and image: who is closer to the truth, tensor by tensor, when the HIP path and the fp32 oracle
    disagree.  ~12 GB and under a minute at 128^3 / 512 keypoints; child process by default (see oracle_pair)."""
    if in_subprocess:
        return _child(f"oracle_backbone_fp64({size}, {keypoints}, {sd_seed}, {vol_seed}, {cot_seed}, {threads}, False)", threads)
    from keymorph_amd import synthetic
    from oracle import keymorph_oracle as O
    x = synthetic.blob_volume(size, vol_seed, torch.device("cpu"))
    sd0 = seeded_state_dict(unet_shapes(keypoints, 32, trunc=1), sd_seed)
    cot = torch.randn(1, keypoints, 3, generator=torch.Generator().manual_seed(cot_seed))
    out = {"x": x, "sd": sd0, "cot": cot}
    for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
        sd = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in sd0.items()}
        pts = O.center_of_mass(O.unet3d_forward(sd, x.to(dt), 4, 1, 8), "ij")
        (pts * cot.to(dt)).sum().backward()
        out["pts_" + name] = pts.detach()
        out["grads_" + name] = {k: v.grad.detach() for k, v in sd.items()}
    return out


def oracle_tps_step(size, keypoints, lam, sd_seed=23, seed=100, threads=32, in_subprocess=True):
    """ONE training step with a TPS transform (scripts/train.py:129-176: KeyMorph.forward with `tps_<lam>`, align_img, MSELoss,
    loss.backward()) by the oracle's autograd on the host: the grid evaluation is chunked over z slabs under
    torch.utils.checkpoint (the reference's own `use_checkpoint` / subgrid device for the (K, N, 3) temporaries,
    keymorph/keypoint_aligners.py:365-433).  Returns the pair, the weights, loss, keypoints, grid samples and every parameter
    gradient."""
    if in_subprocess:
        return _child(f"oracle_tps_step({size}, {keypoints}, {lam}, {sd_seed}, {seed}, {threads}, False)", threads)
    from torch.utils.checkpoint import checkpoint
    from keymorph_amd import synthetic
    from oracle import keymorph_oracle as O
    cpu = torch.device("cpu")
    img_f = synthetic.blob_volume(size, seed, cpu)
    g = O.affine_grid(torch.inverse(synthetic.random_affine_matrix(seed, cpu)), (size,) * 3)
    img_m = O.align_img(g, img_f)
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(unet_shapes(keypoints, 32, trunc=1), sd_seed).items()}
    pf = O.center_of_mass(O.unet3d_forward(sd, img_f, 4, 1, 8), "ij")
    pm = O.center_of_mass(O.unet3d_forward(sd, img_m, 4, 1, 8), "ij")
    lm = torch.full((1,), float(lam))
    theta = O.tps_fit(pf, pm, lm)                                               # ctrl = points_f, tgt = points_m: the inverse map
    base = O.base_grid((size,) * 3)                                             # (D, H, W, 3) ij
    slabs = []
    for z0 in range(0, size, 8):
        pts = base[z0:z0 + 8].reshape(1, -1, 3)
        slabs.append(checkpoint(lambda th, c, p: O.tps_transform_points(th, c, p), theta, pf, pts, use_reentrant=False)
                     .reshape(1, -1, size, size, 3))
    grid = torch.cat(slabs, dim=1).flip(-1)
    img_a = O.align_img(grid, img_m)
    loss = O.mse_loss(img_f, img_a)
    loss.backward()
    gen = torch.Generator().manual_seed(5)
    idx = [torch.randint(0, size, (4096,), generator=gen) for _ in range(3)]
    return {"img_f": img_f, "img_m": img_m, "sd": {k: v.detach() for k, v in sd.items()},
            "grads": {k: v.grad.detach() for k, v in sd.items()}, "loss": float(loss.detach()), "points_f": pf.detach(),
            "points_m": pm.detach(), "idx": idx, "grid_samples": grid.detach()[0][idx[0], idx[1], idx[2]],
            "size": size, "keypoints": keypoints, "lam": float(lam)}


def oracle_cfg1_fp64(threads=32, in_subprocess=True):
    """BASELINE configs[0] end to end -- the example_data_half pair at 128^3 (label / 13 intensities from the fixture
    tests/golden/cfg1_example_half_128.npz), 128 keypoints, affine, MSE, loss.backward() (scripts/train.py:129-176) -- by the
    oracle's autograd in fp32 (the reference arithmetic) AND in fp64 (truth) from the same fp32 weights and images: the
    per-tensor parameter gradients both ways.  ~15 GB of host RAM, about a minute; child process by default (see oracle_pair)."""
    if in_subprocess:
        return _child(f"oracle_cfg1_fp64({threads}, False)", threads)
    from oracle import keymorph_oracle as O
    from tests.util import golden
    K = 128
    g = golden("cfg1_example_half_128.npz")
    img_f = (torch.from_numpy(g["label_0"]).float() / 13.0)[None, None]
    img_m = (torch.from_numpy(g["label_1"]).float() / 13.0)[None, None]
    sd0 = seeded_state_dict(unet_shapes(K, 32, trunc=1), 23)
    out = {"sd": sd0}
    for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
        sd = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in sd0.items()}
        f, m = img_f.to(dt), img_m.to(dt)
        r = O.keymorph_forward(lambda x: O.unet3d_forward(sd, x, 4, 1, 8), f, m, "affine", True)
        mse = O.mse_loss(f, O.align_img(r["grid"], m))
        mse.backward()
        out["mse_" + name] = mse.detach()
        out["points_f_" + name] = r["points_f"].detach()
        out["grads_" + name] = {k: v.grad.detach() for k, v in sd.items()}
    return out
