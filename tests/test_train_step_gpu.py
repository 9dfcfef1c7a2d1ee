"""GPU: the reference's training iteration (scripts/train.py:39-176) restated on this package, end to end --
loader dict (io.make_subject) -> one_hot_subsampled_pair -> random_affine_augment (image bilinear, seg nearest,
same seeded draw) -> KeyMorph.forward -> align_img image + seg -> MSE and Dice branches -> backward -> Adam ->
checkpoint in the reference's format (run.py:588-602) -> --resume_latest style reload (script_utils.py:59-81,
129-154) -> an identical next step.  Golden: tests/golden/trainstep_tiny.npz, produced by tools/make_golden.py from
the reference's OWN functions with the script's seeds."""
import os
import re

import numpy as np
import pytest
import torch

from tests.util import T, golden, sd_checksum, seeded_state_dict, unet_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, atol=1e-5, rtol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def rel_l2(a, b):
    a, b = a.detach().cpu().double().reshape(-1), torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _model():
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    net = TruncatedUNet3D(1, 8, 1, final_sigmoid=False, f_maps=8, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    net.load_state_dict(seeded_state_dict(unet_shapes(8, 8, trunc=1), 320), strict=True)
    return KeyMorph(net, 8, 3, max_train_keypoints=None).to(DEV).train()


def _subjects(g):
    from keymorph_amd.io import make_subject
    subs = []
    for i in range(2):
        s = make_subject(g[f"sub{i}::img"][0, 0], affine=g[f"sub{i}::affine"][0], seg=g[f"sub{i}::seg"][0, 0],
                         rescale=False)
        subs.append(s)
    return subs


def _iteration(km, fixed, moving, loss_fn, args):
    """scripts/train.py:39-176, one pass of the loop body."""
    from keymorph_amd import loss_ops
    from keymorph_amd.augmentation import random_affine_augment
    from keymorph_amd.io import AFFINE, DATA
    from keymorph_amd.utils import align_img, one_hot_subsampled_pair
    img_f, img_m = fixed["img"][DATA], moving["img"][DATA]
    aff_f, aff_m = fixed["img"][AFFINE], moving["img"][AFFINE]
    seg_f, seg_m = one_hot_subsampled_pair(fixed["seg"][DATA].long(), moving["seg"][DATA].long(),
                                           args["max_train_seg_channels"])
    assert img_f.shape[1] == 1 and img_m.shape[1] == 1
    img_f, img_m = img_f.float().to(DEV), img_m.float().to(DEV)
    aff_f, aff_m = aff_f.float().to(DEV), aff_m.float().to(DEV)
    seg_f, seg_m = seg_f.float().to(DEV), seg_m.float().to(DEV)
    img_m, seg_m, aug = random_affine_augment(img_m, seg=seg_m, max_random_params=(0.2, 0.2, 3.1416, 0.1),
                                              scale_params=args["scale_augment"], return_affine_matrix=True)
    aff_m = torch.bmm(aff_m, aug)
    r = km(img_f, img_m, transform_type="affine", return_aligned_points=False, aff_f=aff_f, aff_m=aff_m)["affine"]
    img_a = align_img(r["grid"], img_m)
    seg_a = align_img(r["grid"], seg_m)
    metrics = {"mse": loss_ops.MSELoss()(img_f, img_a), "softdiceloss": loss_ops.DiceLoss()(seg_a, seg_f)}
    metrics["loss"] = metrics["mse"] if loss_fn == "mse" else metrics["softdiceloss"]
    return metrics, dict(aug=aug, img_m=img_m, seg_f=seg_f, seg_m=seg_m, grid=r["grid"])


@pytest.mark.parametrize("loss_fn", ["mse", "dice"])
def test_two_training_iterations_and_resume(loss_fn, tmp_path):
    from keymorph_amd import parallel
    from keymorph_amd.io import load_checkpoint, save_checkpoint
    g = golden("trainstep_tiny.npz")
    assert abs(sd_checksum(seeded_state_dict(unet_shapes(8, 8, trunc=1), 320)) - float(g["sdsum"])) < 1e-6 * float(g["sdsum"])
    subs = _subjects(g)
    args = {"max_train_seg_channels": 3, "scale_augment": 0.3}
    km = _model()
    flat = parallel.FlatParams(km.parameters())
    opt = parallel.FusedAdam(flat, lr=1e-3)
    torch.manual_seed(23)            # scripts/run.py:217-218 (set_seed)
    np.random.seed(23)
    net = km.backbone
    for step in range(2):
        fixed, moving = subs[step % 2], subs[(step + 1) % 2]
        flat.zero_grad()
        metrics, aux = _iteration(km, fixed, moving, loss_fn, args)
        t = f"{loss_fn}::step{step}"
        # step 0 sees identical parameters; step 1 sees parameters after one Adam step (a near-zero gradient component
        # may take a +-lr step in either direction, see below), hence the wider bars
        tol = 1e-5 if step == 0 else 2e-4
        close(aux["aug"], g[f"{t}::aug_matrix"], 1e-6)
        close(aux["img_m"], g[f"{t}::img_m_aug"], 1e-5)
        close(aux["seg_f"], g[f"{t}::seg_f"], 0, 0)
        close(aux["seg_m"], g[f"{t}::seg_m_aug"], 0, 0)          # nearest-mode warp of a one-hot map: exact
        close(aux["grid"], g[f"{t}::grid"], 10 * tol if step == 0 else 5e-3)     # (affine fit on clumped keypoints: ~100x)
        close(metrics["mse"], g[f"{t}::mse"], tol)
        close(metrics["softdiceloss"], g[f"{t}::softdiceloss"], 10 * tol)
        metrics["loss"].backward()
        e = rel_l2(net.final_conv.weight.grad, g[f"{t}::grad::final_conv.weight"])
        assert e < (1e-3 if step == 0 else 3e-2), (t, e)
        opt.step(1.0)
        # Adam's first update is lr * g / (|g| + eps): components with |g| ~ 1e-8 can differ by O(lr); everything else
        # must land on the reference's parameters
        for name, p in (("final_conv.weight", net.final_conv.weight),
                        ("enc0", net.encoders[0].basic_module.SingleConv1.conv.weight)):
            d = (p.detach().cpu() - T(g[f"{t}::after::{name}"])).abs()
            # (step 1: Adam's second update divides by sqrt(v) of two gradients that already differ at the 1e-4 level)
            assert float((d > (2e-5 if step == 0 else 1e-4)).float().mean()) < (2e-3 if step == 0 else 5e-2), \
                (t, name, float(d.max()))
            assert float(d.max()) <= 2.5e-3 * (step + 1), (t, name, float(d.max()))
        if step == 0:
            path = tmp_path / "epoch1_trained_model.pth.tar"
            state = save_checkpoint(path, km, opt, epoch=1, args=None)
            assert sorted(state.keys()) == list(g[f"{loss_fn}::ckpt_keys"])
            assert list(state["state_dict"].keys()) == list(g[f"{loss_fn}::ckpt_sd_keys"])
            assert [float(v["step"]) for v in state["optimizer"]["state"].values()] == list(g[f"{loss_fn}::ckpt_opt_step"])
            rng_t, rng_n = torch.get_rng_state(), np.random.get_state()

    # --resume_latest: newest epoch file by the reference's pattern, strict reload, identical next step
    (tmp_path / "epoch0_trained_model.pth.tar").write_bytes(b"")
    pat = re.compile(r"epoch(\d+)_trained_model.pth.tar")
    latest = max((f for f in os.listdir(tmp_path) if pat.match(f)), key=lambda f: int(pat.match(f).group(1)))
    assert latest == "epoch1_trained_model.pth.tar"
    km2 = _model()
    flat2 = parallel.FlatParams(km2.parameters())
    opt2 = parallel.FusedAdam(flat2, lr=123.0)
    state, km2, opt2 = load_checkpoint(tmp_path / latest, km2, opt2, device=DEV)
    assert state["epoch"] == 1 and opt2.t == 1 and opt2.lr == 1e-3
    torch.set_rng_state(rng_t)
    np.random.set_state(rng_n)
    flat2.zero_grad()
    metrics2, aux2 = _iteration(km2, subs[1], subs[0], loss_fn, args)
    metrics2["loss"].backward()
    opt2.step(1.0)
    assert float(metrics2["loss"]) == float(metrics["loss"])                      # bit-identical replay
    assert torch.equal(aux2["grid"], aux["grid"])
    assert torch.equal(flat2.flat, flat.flat)
    # and the state interchanges with torch.optim.Adam (what the reference's loop would load)
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in km2.parameters()], lr=1.0)
    ref_opt.load_state_dict(opt2.state_dict())
    assert ref_opt.param_groups[0]["lr"] == 1e-3


@pytest.mark.parametrize("size,levels,trunc,shrink", [(64, 4, 1, 0.95), (40, 3, 0, 0.99)])
def test_use_checkpoint_bit_identical_gradients_and_smaller_peak(size, levels, trunc, shrink):
    """use_checkpoint=True (keymorph/unet3d/model.py:113-144: every encoder / decoder block under torch.utils.checkpoint,
    non-reentrant) recomputes a block's activations during the backward instead of keeping them.  One training step (tps_1,
    warp + MSE) with and without it from the same weights: loss, keypoints and EVERY parameter gradient bit-identical (the
    kernels are deterministic, and the gradient hand-offs between blocks -- channel-blocked, pre-split, lazy GroupNorm
    backward, pool_fork -- cross the checkpoint boundaries unchanged), and the peak of allocated device memory over the step
    is lower with it."""
    from keymorph_amd import ops, synthetic
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D, UNet3D
    K = 32
    img_f, img_m = synthetic.make_pair(size, 3, torch.device(DEV))

    def run(ckpt):
        torch.manual_seed(5)
        if trunc:
            net = TruncatedUNet3D(1, K, trunc, final_sigmoid=False, f_maps=16, layer_order="gcr", num_groups=8, num_levels=levels,
                                  is_segmentation=False, conv_padding=1, use_checkpoint=ckpt)
        else:
            net = UNet3D(1, K, final_sigmoid=False, f_maps=16, layer_order="gcr", num_groups=8, num_levels=levels,
                         is_segmentation=False, conv_padding=1, use_checkpoint=ckpt)
        assert net.use_checkpoint is ckpt
        km = KeyMorph(net, K, 3, max_train_keypoints=None, use_checkpoint=ckpt).to(DEV).train()
        ops._WS.clear()                   # both runs allocate their scratch buffers afresh (whatever ran before in this process)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        r = km(img_f, img_m, transform_type="tps_1", return_aligned_points=False)["tps_1"]
        loss, _ = ops.warp_mse(img_m, r["grid"], img_f)
        loss.backward()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        return float(loss.detach()), r["points_f"].detach().clone(), {k: p.grad.clone() for k, p in km.named_parameters()}, peak

    l0, p0, g0, m0 = run(False)
    l1, p1, g1, m1 = run(True)
    assert l0 == l1 and torch.equal(p0, p1)
    for k in g0:
        assert torch.isfinite(g0[k]).all() and torch.equal(g0[k], g1[k]), k
    print(f"use_checkpoint at {size}^3, {levels} levels: peak device memory over the step {m0 / 2**20:.0f} MiB -> {m1 / 2**20:.0f} MiB")
    assert m1 < shrink * m0, (m0, m1)      # (40^3, three levels: the step's peak is mostly the TPS grid and the workspaces)
