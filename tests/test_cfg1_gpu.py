"""GPU: BASELINE.json configs[0] -- the example_data_half pair at 128^3 (scripts/hyperparameters.py:4-11), 128
keypoints, affine aligner -- HIP path vs (i) what the REFERENCE computed for it on the CPU
(tests/golden/cfg1_example_half_128.npz, tools/make_golden.py::gen_cfg1) and (ii) the oracle run here on the host.
The intensity images are missing from the reference mount, so intensity = label / 13 from the two label
maps of example_data_half/seg_m, nearest-down-sampled 256^3 -> 128^3; the fixture carries those label maps (data)."""
import numpy as np
import pytest
import torch

from tests.util import T, golden, sd_checksum, seeded_state_dict, unet_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda"
K = 128


def close(a, b, atol, rtol=0):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def rel_l2(a, b):
    a, b = a.detach().cpu().double().reshape(-1), torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_cfg1_128_affine_vs_reference_and_oracle():
    from keymorph_amd import loss_ops, utils
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    from oracle import keymorph_oracle as O
    g = golden("cfg1_example_half_128.npz")
    lab_f, lab_m = torch.from_numpy(g["label_0"]), torch.from_numpy(g["label_1"])
    assert lab_f.shape == (128, 128, 128) and int(lab_f.max()) == 13
    img_f = (lab_f.float() / 13.0)[None, None].to(DEV)
    img_m = (lab_m.float() / 13.0)[None, None].to(DEV)
    seg_f = utils.one_hot(lab_f.long()[None, None]).float()          # 14 channels, HIP one-hot kernel
    seg_m = utils.one_hot(lab_m.long()[None, None]).float()
    assert seg_f.shape == (1, 14, 128, 128, 128)
    sd = seeded_state_dict(unet_shapes(K, 32, trunc=1), 23)
    assert abs(sd_checksum(sd) - float(g["sdsum"])) < 1e-6 * float(g["sdsum"])
    net = TruncatedUNet3D(1, K, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    net.load_state_dict(sd, strict=True)
    km = KeyMorph(net, K, 3, max_train_keypoints=None).to(DEV).train()
    r = km(img_f, img_m, transform_type="affine", return_aligned_points=True)["affine"]
    img_a = utils.align_img(r["grid"], img_m)
    seg_a = utils.align_img(r["grid"], seg_m)
    mse = loss_ops.MSELoss()(img_f, img_a)
    dice = loss_ops.DiceLoss()(seg_a, seg_f)
    with torch.no_grad():
        hard = loss_ops.DiceLoss(hard=True)(seg_a, seg_f, ign_first_ch=True)
    mse.backward()

    # (i) against the reference's own numbers -- north-star bar 1e-4 on keypoints / grid / warped volume / Dice
    close(r["points_f"], g["points_f"], 1e-4)
    close(r["points_m"], g["points_m"], 1e-4)
    close(r["points_a"], g["points_a"], 1e-4)
    close(r["matrix"], g["matrix"], 1e-4)
    close(r["grid"][:, ::8, ::8, ::8], g["grid_sub8"], 1e-4)
    # the warped label image is piecewise constant (steps of 1/13): away from label boundaries it is exact, at a
    # boundary a 1e-5 grid difference moves the bilinear blend by <= 1e-5 * 64 voxels * (1/13)
    close(img_a[:, :, ::4, ::4, ::4], g["img_a_sub4"], 1e-4)
    close(mse, g["mse"], 1e-5)
    close(dice, g["softdiceloss"], 1e-4)
    close(hard, g["harddiceloss"], 1e-4)
    e_b = rel_l2(net.final_conv.bias.grad, g["gradfull::final_conv.bias"])
    e_0 = rel_l2(net.encoders[0].basic_module.SingleConv1.conv.weight.grad, g["gradfull::enc0"])
    print(f"cfg1: rel-L2 against the two full gradient tensors the REFERENCE run left in the fixture: final bias {e_b:.2e}, "
          f"first conv {e_0:.2e}")
    assert e_b < 1e-2 and e_0 < 1e-2, (e_b, e_0)
    # Every parameter-gradient tensor against the FP64 run of the oracle on this very pair (round 6; until round 5 the bar
    # here was "gradient norms within 3e-2 of the reference's").  128^3 x 32..256 channels of piecewise-constant label
    # images put a few hundred pre-ReLU values at rounding distance from zero, where any fp32 implementation masks
    # differently from fp64 -- the oracle's own fp32 autograd is measured against the same truth and sets the scale:
    # per tensor |hip - fp64| <= max(3e-3, 3 |oracle fp32 - fp64|) relative L2, the whole vector <= max(1e-3, 2 x the oracle's).
    from tests.oracle_at_size import oracle_cfg1_fp64
    ref = oracle_cfg1_fp64()
    g64, g32 = ref["grads_fp64"], ref["grads_fp32"]
    assert abs(float(mse) - float(ref["mse_fp64"])) <= 1e-6
    rows, nh, no_, den = [], 0.0, 0.0, 0.0
    for k, p in net.named_parameters():
        t = g64[k].double()
        eh = float((p.grad.detach().cpu().double() - t).norm() / (t.norm() + 1e-300))
        eo = float((g32[k].double() - t).norm() / (t.norm() + 1e-300))
        rows.append((k, eh, eo))
        nh += float((p.grad.detach().cpu().double() - t).pow(2).sum()); no_ += float((g32[k].double() - t).pow(2).sum())
        den += float(t.pow(2).sum())
    whole_h, whole_o = (nh / den) ** 0.5, (no_ / den) ** 0.5
    print(f"cfg1 parameter gradients vs fp64: whole vector hip {whole_h:.2e}, oracle fp32 {whole_o:.2e}")
    for k, eh, eo in sorted(rows, key=lambda r_: -r_[1])[:8]:
        print(f"   {k:62s} hip {eh:.2e}   oracle fp32 {eo:.2e}")
    worse = [(k, eh, eo) for k, eh, eo in rows if eh > max(3e-3, 3 * eo)]
    assert not worse, worse
    assert whole_h <= max(1e-3, 2 * whole_o), (whole_h, whole_o)

    # (ii) against the oracle on the host cores (forward only: ~10 s)
    with torch.no_grad():
        ro = O.keymorph_forward(lambda x: O.unet3d_forward(sd, x, 4, 1, 8), img_f.cpu(), img_m.cpu(), "affine", True)
        img_ao = O.align_img(ro["grid"], img_m.cpu())
        mse_o = O.mse_loss(img_f.cpu(), img_ao)
        dice_o = O.dice_loss(O.align_img(ro["grid"], seg_m.cpu()), seg_f.cpu())
    close(r["points_f"], ro["points_f"], 1e-4)
    close(r["points_m"], ro["points_m"], 1e-4)
    close(r["grid"], ro["grid"], 1e-4)
    close(img_a, img_ao, 1e-4)
    close(mse, mse_o, 1e-5)
    close(dice, dice_o, 1e-4)
