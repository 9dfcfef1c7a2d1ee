"""GPU, round 2: the branches round 1 left untested, against reference-generated goldens.

* real-world-coordinate alignment for the three aligners and through KeyMorph.forward
  (keymorph/keypoint_aligners.py:47-66, 116-148, 255-268, 431-465; utils.py:243-354)
* one_hot / one_hot_subsampled_pair as HIP kernels (utils.py:200-240)
* weighted TPS training with max_train_keypoints (model.py:209-222)
* EVERY parameter-gradient tensor of tiny backbones and of the end-to-end step vs the reference's autograd, with the
  ReLU-kink accounting that justifies (or refuses) a looser bar
* two iterations of the reference's training loop (scripts/train.py:39-176) + checkpoint round trip
"""
import numpy as np
import pytest
import torch

from oracle import keymorph_oracle as O
from tests.util import T, golden, sd_checksum, seeded_state_dict, unet_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, atol=1e-5, rtol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def rel_l2(a, b):
    a, b = a.detach().cpu().double().reshape(-1), torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def tunet(K, f_maps=8, levels=4, trunc=1):
    from keymorph_amd.unet3d.model import TruncatedUNet3D, UNet3D
    if trunc:
        return TruncatedUNet3D(1, K, trunc, final_sigmoid=False, f_maps=f_maps, layer_order="gcr", num_groups=8,
                               num_levels=levels, is_segmentation=False, conv_padding=1)
    return UNet3D(1, K, final_sigmoid=False, f_maps=f_maps, layer_order="gcr", num_groups=8, num_levels=levels,
                  is_segmentation=False, conv_padding=1)


# ------------------------------------------------------------------ real-world coordinates
def _rw(g):
    return dict(dim=3, align_in_real_world_coords=True, aff_f=T(g["aff_f"]).to(DEV), aff_m=T(g["aff_m"]).to(DEV),
                shape_f=T(g["shape_f"]).to(DEV), shape_m=T(g["shape_m"]).to(DEV))


def _aligner(name, pm, pf, w, kw):
    from keymorph_amd.keypoint_aligners import AffineKeypointAligner, RigidKeypointAligner, TPS
    if name == "affine":
        return AffineKeypointAligner(points_m=pm, points_f=pf, w=w, **kw)
    if name == "rigid":
        return RigidKeypointAligner(points_m=pm, points_f=pf, w=w, **kw)
    return TPS(points_m=pm, points_f=pf, lmbda=torch.tensor(float(name[4:])).repeat(1).to(DEV), w=w, **kw)


@pytest.mark.parametrize("name", ["affine", "rigid", "tps_10", "tps_1000"])
@pytest.mark.parametrize("weighted", [False, True])
def test_real_world_aligners_golden(name, weighted):
    g = golden("realworld_small.npz")
    pf, pm = T(g["pf"]).to(DEV), T(g["pm"]).to(DEV)
    w = T(g["w"]).to(DEV) if weighted else None
    al = _aligner(name, pm, pf, w, _rw(g))
    tag = name + ("_w" if weighted else "")
    # mm-scale thin-plate systems (r ~ 10 mm, U ~ 1e2..1e3): the reference's own fp32 solve is 1e-4-accurate there
    tol = 3e-4 if name.startswith("tps") else 1e-4
    close(al.get_flow_field((1, 1, 6, 7, 8)), g[f"{tag}::grid"], tol)
    close(al.get_forward_transformed_points(pm), g[f"{tag}::points_a"], tol)
    close(al.get_inverse_transformed_points(pf), g[f"{tag}::points_inv"], tol)
    if name in ("affine", "rigid"):
        close(al.transform_matrix, g[f"{tag}::matrix"], 1e-4, 1e-4)
        # affine_grid() is the ij-ordered view of the same field (transformations.py:37-58)
        close(al.affine_grid((1, 1, 6, 7, 8)), np.flip(g[f"{tag}::grid"], -1).copy(), tol)


@pytest.mark.parametrize("name", ["affine", "rigid", "tps_10"])
def test_real_world_aligner_gradients(name):
    g = golden("realworld_small.npz")
    pf, pm = T(g["pf"]).to(DEV).requires_grad_(True), T(g["pm"]).to(DEV).requires_grad_(True)
    grid = _aligner(name, pm, pf, None, _rw(g)).get_flow_field((1, 1, 6, 7, 8))
    (grid * T(g[f"{name}::gridcot"]).to(DEV)).sum().backward()
    assert rel_l2(pf.grad, g[f"{name}::dpf"]) < 2e-3, rel_l2(pf.grad, g[f"{name}::dpf"])
    assert rel_l2(pm.grad, g[f"{name}::dpm"]) < 2e-3, rel_l2(pm.grad, g[f"{name}::dpm"])


def test_real_world_through_keymorph_forward():
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.utils import align_img
    from keymorph_amd import loss_ops
    g, e = golden("realworld_small.npz"), golden("e2e_tiny.npz")
    sd = {k[4:]: T(e[k]) for k in e.files if k.startswith("sd::")}
    net = tunet(16)
    net.load_state_dict(sd, strict=True)
    km = KeyMorph(net, 16, 3, max_train_keypoints=None, align_keypoints_in_real_world_coords=True).to(DEV).eval()
    img_f, img_m = T(e["img_f"]).to(DEV), T(e["img_m"]).to(DEV)
    aff_f, aff_m = T(g["aff_f"]).to(DEV), T(g["aff_m"]).to(DEV)
    with torch.no_grad():
        rr = km(img_f, img_m, transform_type=["rigid", "affine", "tps_10"], return_aligned_points=True,
                aff_f=aff_f, aff_m=aff_m)
    # The backbone's keypoints are clumped (std 0.1) and the fit runs on mm coordinates, so 1e-6 keypoint rounding
    # differences are amplified ~100x in the grid: the reference's own affine grid is 3.0e-4 from the all-fp64
    # restatement here (rigid 3e-6, tps_10 1.4e-5).  As for the groupwise grids ("keypoint noise floor") the
    # two factors are checked separately: our keypoints equal the reference's at the fp32 rounding level, and the
    # aligner is exact (fp64 oracle) for the keypoints it was given; the reference's grid within 5e-4 as a sanity bound.
    close(rr["affine"]["points_f"], e["affine::points_f"], 2e-6)
    close(rr["affine"]["points_m"], e["affine::points_m"], 2e-6)
    pf64, pm64 = rr["affine"]["points_f"].cpu().double(), rr["affine"]["points_m"].cpu().double()
    s64 = torch.tensor([32.0, 32.0, 32.0], dtype=torch.float64)
    for tt in ("rigid", "affine", "tps_10"):
        exact = O.register_real_world(pf64, pm64, tt, (32, 32, 32), aff_f.cpu().double(), aff_m.cpu().double(), s64, s64)
        tol = 3e-5 if tt.startswith("tps") else 1e-5
        close(rr[tt]["grid"], exact["grid"].float(), tol)
        close(rr[tt]["points_a"], exact["points_a"].float(), tol)
        close(rr[tt]["grid"], g[f"km::{tt}::grid"], 5e-4)
        close(rr[tt]["points_a"], g[f"km::{tt}::points_a"], 5e-4)
        print(tt, "real-world grid vs fp64 oracle on our keypoints", float((rr[tt]["grid"].cpu().double() - exact["grid"]).abs().max()),
              "vs the reference", float((rr[tt]["grid"].cpu() - T(g[f"km::{tt}::grid"])).abs().max()))
    with pytest.raises(KeyError):
        km(img_f, img_m, transform_type="affine", return_aligned_points=False)       # aff_f / aff_m are required
    km.train()
    for tt in ("affine", "tps_10"):
        km.zero_grad()
        r = km(img_f, img_m, transform_type=tt, return_aligned_points=False, aff_f=aff_f, aff_m=aff_m)[tt]
        mse = loss_ops.MSELoss()(img_f, align_img(r["grid"], img_m))
        close(mse, g[f"km_train::{tt}::mse"], 1e-5)
        mse.backward()
        e_ = rel_l2(net.final_conv.weight.grad, g[f"km_train::{tt}::gradfull::final_conv.weight"])
        assert e_ < 1e-2, (tt, e_)


# ------------------------------------------------------------------ one-hot encodings
def test_one_hot_kernels_golden():
    from keymorph_amd import utils
    g = golden("onehot_small.npz")
    seg = T(g["seg"], torch.int64)
    for inp in (seg, seg.to(DEV)):                       # the loops pass CPU tensors (train.py:54-61)
        oh = utils.one_hot(inp)
        assert oh.dtype == torch.int64 and oh.is_cuda
        close(oh, g["one_hot"], 0, 0)
    s1, s2 = T(g["seg1"], torch.int64).to(DEV), T(g["seg2"], torch.int64).to(DEV)
    for num in (5, 14, 9):
        np.random.seed(int(g[f"sub{num}::seed"][0]))
        a, b = utils.one_hot_subsampled_pair(s1, s2, num)
        assert a.dtype == torch.float32
        close(a, g[f"sub{num}::a"], 0, 0)
        close(b, g[f"sub{num}::b"], 0, 0)
    with pytest.raises(ValueError):
        utils.one_hot(torch.tensor([[[[[-1, 2]]]]]))
    # vs the oracle at a realistic size (14 labels, ragged volume)
    lab = torch.randint(0, 14, (2, 1, 33, 20, 47), generator=torch.Generator().manual_seed(2))
    close(utils.one_hot(lab), O.one_hot(lab), 0, 0)
    np.random.seed(3)
    a, b = utils.one_hot_subsampled_pair(lab[:1], lab[1:], 6)
    np.random.seed(3)
    ao, bo = O.one_hot_subsampled_pair(lab[:1], lab[1:], 6)
    close(a, ao, 0, 0)
    close(b, bo, 0, 0)


# ------------------------------------------------------------------ weighted TPS training, keypoint subsampling
@pytest.mark.parametrize("weighting", ["power", None])
def test_tps_training_subsamples_points_and_weights(weighting):
    from keymorph_amd import loss_ops
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.utils import align_img
    g, e = golden("weighted_subsample.npz"), golden("e2e_tiny.npz")
    sd = {k[4:]: T(e[k]) for k in e.files if k.startswith("sd::")}
    net = tunet(16)
    net.load_state_dict(sd, strict=True)
    km = KeyMorph(net, 16, 3, max_train_keypoints=6, weight_keypoints=weighting).to(DEV).train()
    np.random.seed(int(g["np_seed"][0]))
    r = km(T(e["img_f"]).to(DEV), T(e["img_m"]).to(DEV), transform_type="tps_1", return_aligned_points=True)["tps_1"]
    t = str(weighting)
    assert r["points_f"].shape == (1, 6, 3)
    close(r["points_f"], g[f"{t}::points_f"], 1e-5)
    close(r["points_m"], g[f"{t}::points_m"], 1e-5)
    if weighting:
        assert r["points_weights"].shape == (1, 6)
        close(r["points_weights"], g[f"{t}::weights"], 1e-6, 2e-4)
    close(r["grid"], g[f"{t}::grid"], 1e-4)
    close(r["points_a"], g[f"{t}::points_a"], 3e-4)
    mse = loss_ops.MSELoss()(T(e["img_f"]).to(DEV), align_img(r["grid"], T(e["img_m"]).to(DEV)))
    close(mse, g[f"{t}::mse"], 1e-5)
    mse.backward()
    assert rel_l2(net.final_conv.weight.grad, g[f"{t}::gradfull::final_conv.weight"]) < 1e-2
    assert rel_l2(net.final_conv.bias.grad, g[f"{t}::gradfull::final_conv.bias"]) < 1e-2


# ------------------------------------------------------------------ full gradient vectors
def _record_relu_masks(monkeypatch):
    """Every SingleConv output of the HIP network in execution order (NDHWC, post-ReLU) -> list of bool masks."""
    from keymorph_amd import backbone_ops as B
    masks = []
    orig_single, orig_up = B.single_conv_gcr, B.upcat_conv_gcr

    def single(*a, **k):
        y = orig_single(*a, **k)
        masks.append((y.detach() > 0).permute(0, 4, 1, 2, 3).cpu())
        return y

    def up(*a, **k):
        y = orig_up(*a, **k)
        masks.append((y.detach() > 0).permute(0, 4, 1, 2, 3).cpu())
        return y

    monkeypatch.setattr(B, "single_conv_gcr", single)
    monkeypatch.setattr(B, "upcat_conv_gcr", up)
    return masks


def _kink_flips(masks, sd, x, levels, trunc):
    """Voxels whose ReLU mask differs from the fp64 oracle's, and the largest |pre-ReLU| (relative to the layer's
    maximum) among them: a flip is only excusable where the true pre-activation is at rounding distance from 0."""
    taps = []
    with torch.no_grad():
        O.unet3d_forward({k: v.double() for k, v in sd.items()}, x.double(), levels, trunc, 8, taps)
    assert len(taps) == len(masks), (len(taps), len(masks))
    flips, worst = 0, 0.0
    for pre, m in zip(taps, masks):
        bad = (pre > 0) != m
        n = int(bad.sum())
        flips += n
        if n:
            worst = max(worst, float(pre[bad].abs().max() / pre.abs().max()))
    return flips, worst


def _check_all_gradients(named_grads, ref_of, flips, worst_flip, tight, what):
    """rel-L2 <= `tight` on EVERY tensor; a looser 3e-2 is granted only if ReLU masks provably flipped at voxels whose
    fp64 pre-activation is below 1e-5 of the layer maximum (at most 8 of them)."""
    errs = {k: rel_l2(v, ref_of(k)) for k, v in named_grads}
    worst = max(errs, key=errs.get)
    print(f"{what}: worst parameter-gradient rel-L2 {errs[worst]:.2e} ({worst}); ReLU-mask flips vs fp64: {flips}")
    # The first layer's one-element GroupNorm weight / bias are whole-volume sums whose terms cancel to ~1e-3 of their size
    # (rstd (sum dxn x - mean sum dxn) over every voxel): the per-tensor worst of every pairing of implementations and
    # arithmetics.  Measured on the kink-free network: 4.4e-6 with the VALU correlation kernel,
    # 0.9e-5..1.2e-5 with the fp32 matrix-core kernel (another summation order) -- granted 3e-5, every other tensor `tight`.
    cancelling = ("encoders.0.basic_module.SingleConv1.groupnorm.weight", "encoders.0.basic_module.SingleConv1.groupnorm.bias")
    bad = {k: e for k, e in errs.items() if e > (max(tight, 3e-5) if k.endswith(cancelling) else tight)}
    if bad:
        assert 0 < flips <= 8 and worst_flip < 1e-5, (bad, flips, worst_flip)
        assert max(bad.values()) < 3e-2, bad
    return errs


@pytest.mark.parametrize("tag,levels,trunc,size,tight", [("tunet16", 4, 1, 16, 1e-3), ("unet16", 4, 0, 16, 1e-3),
                                                         ("kinkfree", 3, 1, 8, 1e-5)])
def test_backbone_every_parameter_gradient(tag, levels, trunc, size, tight, monkeypatch):
    """sum(net(x) * cot).backward(): all parameter gradients vs the reference's autograd (gradients_tiny.npz).
    `kinkfree`: no fp64 pre-ReLU value within 2.4e-4 of zero, so no implementation can flip a mask: 1e-5."""
    g = golden("gradients_tiny.npz")
    seed = 300 if tag != "kinkfree" else 1000 + int(g["kinkfree::seed"][0])
    shapes = unet_shapes(8, 8, levels=levels, trunc=trunc or None)
    sd = seeded_state_dict(shapes, seed)
    assert abs(sd_checksum(sd) - float(g[f"{tag}::sdsum"])) < 1e-6 * float(g[f"{tag}::sdsum"])
    net = tunet(8, levels=levels, trunc=trunc)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).train()
    masks = _record_relu_masks(monkeypatch)
    x = T(g[f"{tag}::x"])
    y = net(x.to(DEV))
    close(y, g[f"{tag}::out"], 1e-4, 1e-4)
    (y * T(g[f"{tag}::cot"]).to(DEV)).sum().backward()
    flips, wf = _kink_flips(masks, sd, x, levels, trunc)
    if tag == "kinkfree":
        assert flips == 0 and float(g["kinkfree::margin"]) > 2e-4
    _check_all_gradients([(k, p.grad) for k, p in net.named_parameters()], lambda k: g[f"{tag}::grad::{k}"],
                         flips, wf, tight, tag)


@pytest.mark.parametrize("tt,loss_name", [("affine", "mse"), ("affine", "dice"), ("rigid", "mse"), ("tps_1", "mse")])
def test_end_to_end_every_parameter_gradient(tt, loss_name, monkeypatch):
    """KeyMorph.forward -> align_img -> MSE | Dice -> backward at 16^3: every parameter gradient vs the reference."""
    from keymorph_amd import loss_ops
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.utils import align_img
    g = golden("gradients_tiny.npz")
    shapes = unet_shapes(8, 8, trunc=1)
    sd = seeded_state_dict(shapes, 310)
    assert abs(sd_checksum(sd) - float(g["e2e16::sdsum"])) < 1e-6 * float(g["e2e16::sdsum"])
    net = tunet(8)
    net.load_state_dict(sd, strict=True)
    km = KeyMorph(net, 8, 3, max_train_keypoints=None).to(DEV).train()
    img_f, img_m = T(g["e2e16::img_f"]), T(g["e2e16::img_m"])
    masks = _record_relu_masks(monkeypatch)
    r = km(img_f.to(DEV), img_m.to(DEV), transform_type=tt, return_aligned_points=False)[tt]
    close(r["grid"], g[f"e2e16::{tt}::{loss_name}::grid"], 1e-4)
    if loss_name == "mse":
        loss = loss_ops.MSELoss()(img_f.to(DEV), align_img(r["grid"], img_m.to(DEV)))
    else:
        loss = loss_ops.DiceLoss()(align_img(r["grid"], T(g["e2e16::seg_m"]).to(DEV)), T(g["e2e16::seg_f"]).to(DEV))
    close(loss, g[f"e2e16::{tt}::{loss_name}::loss"], 1e-5)
    loss.backward()
    flips, wf = _kink_flips(masks, sd, torch.cat([img_f, img_m]), 4, 1)
    _check_all_gradients([(k, p.grad) for k, p in net.named_parameters()],
                         lambda k: g[f"e2e16::{tt}::{loss_name}::grad::{k}"], flips, wf, 1e-3, f"e2e16 {tt} {loss_name}")
