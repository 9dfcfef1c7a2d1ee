#!/usr/bin/env python3
"""Generate golden vectors by importing the REAL reference (build container only).

Usage (in the container that has /root/reference):
    python tools/make_golden.py            # writes tests/golden/*.npz
    python tools/make_golden.py augment    # only the named fixture(s)

The reference cannot travel to the GPU box in any form, so what is committed is
data only: seeded inputs and the outputs the reference produced for them.  The
reference needs three empty stub packages on sys.path (nibabel, skimage, h5py:
imported eagerly by keymorph/__init__.py, never touched by the hot path --
SURVEY.md section 8c); they are created in a temp dir here.

Weights for the big fixed-width ConvNet are not stored: both this script and the
tests regenerate them with ``seeded_state_dict`` (deterministic torch CPU RNG,
same image on both boxes) and a checksum in the fixture guards the assumption.
"""
import os
import sys
import tempfile

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

REF = os.environ.get("KEYMORPH_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _install_stubs():
    d = tempfile.mkdtemp(prefix="km_stubs_")
    for name in ("nibabel", "skimage", "h5py"):
        os.makedirs(os.path.join(d, name))
        with open(os.path.join(d, name, "__init__.py"), "w") as f:
            f.write("morphology = None\n" if name == "skimage" else "")
    open(os.path.join(d, "skimage", "morphology.py"), "w").close()
    sys.path.insert(0, d)
    sys.path.insert(0, REF)


_install_stubs()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from keymorph.layers import CenterOfMass2d, CenterOfMass3d  # noqa: E402
from keymorph.keypoint_aligners import (  # noqa: E402
    AffineKeypointAligner, RigidKeypointAligner, TPS)
from keymorph.transformations import AffineTransform  # noqa: E402
from keymorph.utils import align_img  # noqa: E402
from keymorph import loss_ops  # noqa: E402
from keymorph.model import KeyMorph  # noqa: E402
from keymorph.net import ConvNet  # noqa: E402
from keymorph.unet3d.model import UNet3D, TruncatedUNet3D  # noqa: E402


def seeded_state_dict(ref_sd, seed):
    """Deterministic weights keyed like ``ref_sd``: N(0,1)/sqrt(fan_in) for
    >=2-D tensors, 1 + 0.1 N for norm weights, 0.1 N for biases."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(ref_sd.keys()):
        shp = tuple(ref_sd[k].shape)
        r = torch.randn(shp, generator=g)
        if len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            out[k] = r / np.sqrt(fan_in) * 1.4
        elif k.endswith("norm.weight"):
            out[k] = 1 + 0.1 * r
        else:
            out[k] = 0.1 * r
    return out


def sd_checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def npy(t):
    # a COPY: .numpy() of a CPU tensor shares its memory, and parameters / .grad buffers are updated in place later
    return np.array(t.detach().cpu().numpy(), copy=True)


def blob_volume(shape, seed):
    """Smooth synthetic 'anatomy': sum of anisotropic Gaussians in [0, 1]."""
    g = torch.Generator().manual_seed(seed)
    axes = torch.meshgrid(*[torch.linspace(-1, 1, s) for s in shape], indexing="ij")
    vol = torch.zeros(shape)
    for _ in range(12):
        c = torch.rand(3, generator=g) * 1.2 - 0.6
        s = torch.rand(3, generator=g) * 0.25 + 0.1
        a = torch.rand(1, generator=g) * 0.8 + 0.2
        e = sum(((axes[i] - c[i]) / s[i]) ** 2 for i in range(3))
        vol += a * torch.exp(-0.5 * e)
    vol += 0.01 * torch.rand(shape, generator=g)
    vol = (vol - vol.min()) / (vol.max() - vol.min())
    return vol[None, None].float()


# --------------------------------------------------------------------------
def gen_ops():
    g = torch.Generator().manual_seed(1)
    d = {}
    # center of mass (3D both index orders, 2D)
    hm = torch.randn(2, 5, 6, 7, 8, generator=g)
    d["com_in"] = npy(hm)
    d["com_ij"] = npy(CenterOfMass3d("ij")(hm))
    d["com_xy"] = npy(CenterOfMass3d("xy")(hm))
    hm2 = torch.randn(2, 3, 9, 11, generator=g)
    d["com2d_in"] = npy(hm2)
    d["com2d_ij"] = npy(CenterOfMass2d("ij")(hm2))

    # affine / rigid
    shape5 = (1, 1, 6, 7, 8)
    for K, tag in ((12, "k12"), (64, "k64")):
        pf = torch.rand(1, K, 3, generator=g) * 1.6 - 0.8
        A = torch.eye(3) + 0.15 * torch.randn(3, 3, generator=g)
        pm = pf @ A.T + 0.1 * torch.randn(1, 1, 3, generator=g) + 0.03 * torch.randn(1, K, 3, generator=g)
        w = torch.rand(1, K, generator=g)
        w = w / w.sum()
        d[f"{tag}_pf"], d[f"{tag}_pm"], d[f"{tag}_w"] = npy(pf), npy(pm), npy(w)
        for name, cls in (("affine", AffineKeypointAligner), ("rigid", RigidKeypointAligner)):
            for wt, wtag in ((None, ""), (w, "_w")):
                al = cls(points_m=pm, points_f=pf, w=wt, dim=3)
                d[f"{tag}_{name}{wtag}_matrix"] = npy(al.transform_matrix)
                d[f"{tag}_{name}{wtag}_inv"] = npy(al.inverse_transform_matrix)
                d[f"{tag}_{name}{wtag}_grid"] = npy(al.get_flow_field(shape5))
                d[f"{tag}_{name}{wtag}_points_a"] = npy(al.get_forward_transformed_points(pm))
        for lam in (0.0, 0.1, 10.0):
            lm = torch.tensor(lam).repeat(1)
            tps = TPS(points_m=pm, points_f=pf, lmbda=lm, dim=3)
            ltag = str(lam).replace(".", "p")
            d[f"{tag}_tps{ltag}_theta"] = npy(tps.inverse_theta)
            d[f"{tag}_tps{ltag}_grid"] = npy(tps.get_flow_field(shape5))
            d[f"{tag}_tps{ltag}_grid_sub"] = npy(tps.get_flow_field(shape5, compute_on_subgrids=True))
            d[f"{tag}_tps{ltag}_points_a"] = npy(tps.get_forward_transformed_points(pm))
        # gradients of a scalar through fit + grid (autograd of the reference)
        for name in ("affine", "rigid", "tps0p0", "tps1p0"):
            pf_ = pf.clone().requires_grad_(True)
            pm_ = pm.clone().requires_grad_(True)
            if name == "affine":
                al = AffineKeypointAligner(points_m=pm_, points_f=pf_, dim=3)
            elif name == "rigid":
                al = RigidKeypointAligner(points_m=pm_, points_f=pf_, dim=3)
            else:
                lam = float(name[3:].replace("p", "."))
                al = TPS(points_m=pm_, points_f=pf_, lmbda=torch.tensor(lam).repeat(1), dim=3)
            grid = al.get_flow_field(shape5)
            cot = torch.randn(grid.shape, generator=torch.Generator().manual_seed(7))
            (grid * cot).sum().backward()
            d[f"{tag}_{name}_gridcot"] = npy(cot)
            d[f"{tag}_{name}_dpf"] = npy(pf_.grad)
            d[f"{tag}_{name}_dpm"] = npy(pm_.grad)

    # AffineTransform(matrix=...) as used by augmentation
    M = torch.eye(4)[None].clone()
    M[0, :3, :] += 0.1 * torch.randn(3, 4, generator=g)
    at = AffineTransform(matrix=M, dim=3)
    d["at_matrix"] = npy(M)
    d["at_grid"] = npy(at.get_flow_field(shape5))

    # warp: values + gradient wrt grid; includes out-of-range coordinates
    x = torch.rand(1, 3, 6, 7, 8, generator=g)
    grid = (torch.rand(1, 5, 6, 7, 3, generator=g) * 2.6 - 1.3).requires_grad_(True)
    out = align_img(grid, x)
    cot = torch.randn(out.shape, generator=g)
    (out * cot).sum().backward()
    d["warp_x"], d["warp_grid"], d["warp_out"] = npy(x), npy(grid), npy(out)
    d["warp_cot"], d["warp_dgrid"] = npy(cot), npy(grid.grad)
    d["warp_out_nearest"] = npy(align_img(grid.detach(), x, mode="nearest"))

    # losses
    a = torch.rand(2, 4, 5, 6, 7, generator=g)
    b = torch.rand(2, 4, 5, 6, 7, generator=g)
    d["loss_a"], d["loss_b"] = npy(a), npy(b)
    d["mse"] = npy(loss_ops.MSELoss()(a, b))
    d["dice_soft"] = npy(loss_ops.DiceLoss()(a, b))
    d["dice_soft_ign"] = npy(loss_ops.DiceLoss()(a, b, ign_first_ch=True))
    d["dice_hard"] = npy(loss_ops.DiceLoss(hard=True)(a, b))
    d["dice_hard_regions"] = npy(loss_ops.DiceLoss(hard=True, return_regions=True)(a, b))
    a_ = a.clone().requires_grad_(True)
    loss_ops.DiceLoss()(a_, b).backward()
    d["dice_soft_dpred"] = npy(a_.grad)
    np.savez_compressed(os.path.join(OUT, "ops_small.npz"), **d)
    print("ops_small.npz", len(d), "arrays")


# --------------------------------------------------------------------------
def make_tunet(K, f_maps, trunc=1, levels=4):
    return TruncatedUNet3D(1, K, trunc, final_sigmoid=False, f_maps=f_maps, layer_order="gcr",
                           num_groups=8, num_levels=levels, is_segmentation=False, conv_padding=1)


def make_unet(K, f_maps, levels=4):
    return UNet3D(1, K, final_sigmoid=False, f_maps=f_maps, layer_order="gcr", num_groups=8,
                  num_levels=levels, is_segmentation=False, conv_padding=1)


def gen_backbones():
    d = {}
    x = blob_volume((32, 32, 32), 11)
    d["x"] = npy(x)
    for name, net in (("tunet", make_tunet(16, 8)), ("unet", make_unet(8, 8)),
                      ("convnet", ConvNet(3, 1, 8, "instance")),
                      ("convnet_none", ConvNet(3, 1, 8, "none"))):
        sd = seeded_state_dict(net.state_dict(), 100)
        net.load_state_dict(sd, strict=True)
        net.train()
        xx = x.clone().requires_grad_(False)
        y = net(xx)
        d[f"{name}_sdsum"] = np.float64(sd_checksum(sd))
        d[f"{name}_out"] = npy(y)
        # parameter gradients of sum(y * cot)
        cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
        (y * cot).sum().backward()
        d[f"{name}_cot"] = npy(cot)
        names = [k for k, _ in net.named_parameters()]
        # keep fixtures small: gradient of every parameter reduced to (sum, abs-sum, first 8)
        for k, p in net.named_parameters():
            gflat = p.grad.reshape(-1)
            d[f"{name}_grad::{k}"] = npy(torch.cat([gflat.sum()[None], gflat.abs().sum()[None], gflat[:8]]))
        if name == "tunet":
            for k, p in net.named_parameters():
                if "decoders.1" in k or k.startswith("final_conv") or "encoders.0" in k:
                    d[f"{name}_gradfull::{k}"] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "backbones_32.npz"), **d)
    print("backbones_32.npz", len(d), "arrays")


def gen_e2e():
    """KeyMorph.forward + align_img + MSE (+Dice) + backward, train mode, bs=1."""
    d = {}
    K = 16
    img_f = blob_volume((32, 32, 32), 21)
    # moving = fixed warped by a small affine through the reference's own classes
    M = torch.eye(4)[None].clone()
    M[0, :3, :3] += torch.tensor([[0.05, 0.08, -0.03], [-0.06, -0.04, 0.05], [0.02, -0.07, 0.06]])
    M[0, :3, 3] = torch.tensor([0.06, -0.05, 0.04])
    img_m = align_img(AffineTransform(matrix=M, dim=3).get_flow_field(img_f.shape), img_f)
    seg_f = torch.stack([(img_f[0, 0] > t).float() for t in (0.0, 0.3, 0.5, 0.7)])[None]
    seg_f = torch.cat([seg_f[:, :-1] - seg_f[:, 1:], seg_f[:, -1:]], 1)
    seg_m = align_img(AffineTransform(matrix=M, dim=3).get_flow_field(img_f.shape), seg_f)
    d["img_f"], d["img_m"], d["seg_f"], d["seg_m"] = npy(img_f), npy(img_m), npy(seg_f), npy(seg_m)
    net = make_tunet(K, 8)
    sd = seeded_state_dict(net.state_dict(), 200)
    # spread the keypoints: bias the final conv so heat-maps are not all-positive noise
    net.load_state_dict(sd, strict=True)
    for k, v in sd.items():
        d[f"sd::{k}"] = npy(v)
    km = KeyMorph(net, K, 3, max_train_keypoints=None).train()
    for tt in ("affine", "rigid", "tps_0", "tps_0.1", "tps_10"):
        km.zero_grad()
        r = km(img_f, img_m, transform_type=tt, return_aligned_points=True)[tt]
        img_a = align_img(r["grid"], img_m)
        seg_a = align_img(r["grid"], seg_m)
        mse = loss_ops.MSELoss()(img_f, img_a)
        dice = loss_ops.DiceLoss()(seg_a, seg_f)
        (mse + dice).backward()
        t = tt.replace(".", "p")
        d[f"{t}::points_f"], d[f"{t}::points_m"] = npy(r["points_f"]), npy(r["points_m"])
        d[f"{t}::points_a"] = npy(r["points_a"])
        d[f"{t}::grid"] = npy(r["grid"]).astype(np.float32)
        if "matrix" in r:
            d[f"{t}::matrix"] = npy(r["matrix"])
        d[f"{t}::img_a"] = npy(img_a)
        d[f"{t}::mse"], d[f"{t}::dice"] = npy(mse), npy(dice)
        for k, p in net.named_parameters():
            gflat = p.grad.reshape(-1)
            d[f"{t}::gradsum::{k}"] = npy(torch.cat([gflat.sum()[None], gflat.abs().sum()[None], gflat[:8]]))
        d[f"{t}::gradfull::final_conv.weight"] = npy(net.final_conv.weight.grad)
        d[f"{t}::gradfull::enc0"] = npy(net.encoders[0].basic_module.SingleConv1.conv.weight.grad)
    # eval mode (subgrid chunking, several types in one call)
    km.eval()
    with torch.no_grad():
        rr = km(img_f, img_m, transform_type=["affine", "tps_1"], return_aligned_points=False)
    d["eval::affine::grid"] = npy(rr["affine"]["grid"])
    d["eval::tps_1::grid"] = npy(rr["tps_1"]["grid"])
    np.savez_compressed(os.path.join(OUT, "e2e_tiny.npz"), **d)
    print("e2e_tiny.npz", len(d), "arrays")


def gen_groupwise():
    d = {}
    K = 16
    net = make_tunet(K, 8)
    net.load_state_dict(seeded_state_dict(net.state_dict(), 200), strict=True)
    km = KeyMorph(net, K, 3, max_train_keypoints=None).eval()
    with tempfile.TemporaryDirectory() as td:
        for i in range(3):
            v = blob_volume((24, 24, 24), 300 + i)
            d[f"img_{i}"] = npy(v)
            np.savez(os.path.join(td, f"img_m_{i:03}.npz"), img=npy(v))
        sdir = os.path.join(td, "out")
        os.makedirs(sdir)
        with torch.no_grad():
            res = km.groupwise_register(td, transform_type=["affine", "rigid", "tps_1"], device="cpu",
                                        save_results_to_disk=True, save_dir=sdir, plot=False,
                                        num_iters=3, log_to_console=False,
                                        num_resolutions_for_itkelastix=None)
        for tt in ("affine", "rigid", "tps_1"):
            d[f"{tt}::grouppoints_m"] = npy(res[tt]["grouppoints_m"])
            d[f"{tt}::grouppoints_a"] = npy(res[tt]["grouppoints_a"])
            for i in range(3):
                d[f"{tt}::grid_{i}"] = np.load(os.path.join(sdir, f"{tt}_grid_{i:03}.npy"))
    np.savez_compressed(os.path.join(OUT, "groupwise_tiny.npz"), **d)
    print("groupwise_tiny.npz", len(d), "arrays")


def gen_tps_illcond():
    """TPS lambda=0 at K=512: reference fp32 result + inputs; the fp64
    truth is recomputed by the oracle in the tests."""
    g = torch.Generator().manual_seed(42)
    K = 512
    pf = torch.rand(1, K, 3, generator=g) * 1.6 - 0.8
    pm = pf + 0.05 * torch.randn(1, K, 3, generator=g)
    d = {"pf": npy(pf), "pm": npy(pm)}
    shape5 = (1, 1, 10, 12, 14)
    for lam in (0.0, 1.0):
        tps = TPS(points_m=pm, points_f=pf, lmbda=torch.tensor(lam).repeat(1), dim=3)
        t = str(lam).replace(".", "p")
        d[f"theta_{t}"] = npy(tps.inverse_theta)
        d[f"grid_{t}"] = npy(tps.get_flow_field(shape5, compute_on_subgrids=True))
    np.savez_compressed(os.path.join(OUT, "tps_k512.npz"), **d)
    print("tps_k512.npz")


def gen_weighted():
    """KeyMorph(weight_keypoints='power' | 'variance') in eval mode on the e2e_tiny inputs and weights."""
    g = np.load(os.path.join(OUT, "e2e_tiny.npz"))
    K = 16
    img_f, img_m = torch.from_numpy(g["img_f"]), torch.from_numpy(g["img_m"])
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    gen = torch.Generator().manual_seed(9)
    scales, biases = 0.5 + torch.rand(K, generator=gen), 0.05 + 0.1 * torch.rand(K, generator=gen)
    d = {"scales": npy(scales), "biases": npy(biases)}
    for mode in ("power", "variance"):
        net = make_tunet(K, 8)
        net.load_state_dict(sd, strict=True)
        km = KeyMorph(net, K, 3, max_train_keypoints=None, weight_keypoints=mode).eval()
        if mode == "variance":
            km.scales.data.copy_(scales)
            km.biases.data.copy_(biases)
        with torch.no_grad():
            rr = km(img_f, img_m, transform_type=["rigid", "affine", "tps_1"], return_aligned_points=True)
        if rr["affine"]["points_weights"] is not None:      # upstream: None for "variance" (model.py:183-193)
            d[f"{mode}::weights"] = npy(rr["affine"]["points_weights"])
        for tt in ("rigid", "affine", "tps_1"):
            d[f"{mode}::{tt}::grid"] = npy(rr[tt]["grid"])
            d[f"{mode}::{tt}::points_a"] = npy(rr[tt]["points_a"])
    # training with weights: the loss gradient also flows through the weights into the head (model.py:183-191)
    seg_f, seg_m = torch.from_numpy(g["seg_f"]), torch.from_numpy(g["seg_m"])
    net = make_tunet(K, 8)
    net.load_state_dict(sd, strict=True)
    km = KeyMorph(net, K, 3, max_train_keypoints=None, weight_keypoints="power").train()
    for tt in ("rigid", "affine", "tps_1"):
        km.zero_grad()
        r = km(img_f, img_m, transform_type=tt, return_aligned_points=True)[tt]
        mse = loss_ops.MSELoss()(img_f, align_img(r["grid"], img_m))
        dice = loss_ops.DiceLoss()(align_img(r["grid"], seg_m), seg_f)
        (mse + dice).backward()
        d[f"train::{tt}::grid"] = npy(r["grid"])
        d[f"train::{tt}::mse"], d[f"train::{tt}::dice"] = npy(mse), npy(dice)
        d[f"train::{tt}::gradfull::final_conv.weight"] = npy(net.final_conv.weight.grad)
        d[f"train::{tt}::gradfull::final_conv.bias"] = npy(net.final_conv.bias.grad)
        d[f"train::{tt}::gradfull::enc0"] = npy(net.encoders[0].basic_module.SingleConv1.conv.weight.grad)
    # the variance formula itself (model.py:75-94), called directly on the heat-maps
    net = make_tunet(K, 8)
    net.load_state_dict(sd, strict=True)
    km = KeyMorph(net, K, 3, weight_keypoints="variance").eval()
    km.scales.data.copy_(scales)
    km.biases.data.copy_(biases)
    with torch.no_grad():
        d["variance::direct_weights"] = npy(km.weight_by_variance(net(img_f), net(img_m)))
    np.savez_compressed(os.path.join(OUT, "weighted_tiny.npz"), **d)
    print("weighted_tiny.npz", len(d), "arrays")


def gen_groupwise_truth():
    """fp64 restatement (oracle, double precision end to end) of the groupwise fixture: the noise floor.  A random-init
    backbone clumps its keypoints, so the final grids amplify ~1e-7 keypoint rounding differences ~100x; the reference
    itself is 1e-5..3e-5 away from these grids.  Tests compare against THIS with max(1e-4, reference error) as bar."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import keymorph_oracle as O
    from tests.util import unet_shapes
    from tests.util import seeded_state_dict as ssd
    g = np.load(os.path.join(OUT, "groupwise_tiny.npz"))
    sd = {k: v.double() for k, v in ssd(unet_shapes(16, 8, trunc=1), 200).items()}
    d = {}
    with torch.no_grad():
        pts = torch.cat([O.center_of_mass(O.unet3d_forward(sd, torch.from_numpy(g[f"img_{i}"]).double(), 4, 1, 8), "ij")
                         for i in range(3)])
        d["grouppoints_m"] = pts.numpy()
        for tt in ("affine", "rigid", "tps_1"):
            cur, mean = O.groupwise_points(pts, tt, 3)
            d[f"{tt}::grouppoints_a"] = cur.numpy()
            for i in range(3):
                grid = O.groupwise_grid(pts[i:i + 1], mean, tt, (24, 24, 24)).numpy()
                d[f"{tt}::grid_{i}"] = grid.astype(np.float32)
                d[f"{tt}::ref_err_{i}"] = np.asarray([np.abs(g[f"{tt}::grid_{i}"] - grid).max()])
    np.savez_compressed(os.path.join(OUT, "groupwise_truth_tiny.npz"), **d)
    print("groupwise_truth_tiny.npz", len(d), "arrays")


def gen_augment():
    """keymorph/augmentation.py: fixed and random affine augmentation of an image, a label map and keypoints."""
    from keymorph.augmentation import AffineDeformation3d, affine_augment, random_affine_augment
    g = torch.Generator().manual_seed(77)
    img = blob_volume((12, 14, 10), 5).reshape(1, 1, 12, 14, 10)
    seg = torch.randint(0, 5, (1, 1, 12, 14, 10), generator=g).float()
    pts = torch.rand(1, 9, 3, generator=g) * 1.6 - 0.8
    d = {"img": npy(img), "seg": npy(seg), "pts": npy(pts)}
    fixed = (0.1, -0.05, 0.3, 0.04)
    a, b, c = affine_augment(img, fixed, seg=seg, points=pts)
    d["fixed_params"] = np.asarray(fixed, np.float32)
    d["fixed_img"], d["fixed_seg"], d["fixed_pts"] = npy(a), npy(b), npy(c)
    torch.manual_seed(1234)
    a, b, c, m = random_affine_augment(img, seg=seg, points=pts, max_random_params=(0.2, 0.2, 3.1416, 0.1),
                                       scale_params=0.5, return_affine_matrix=True)
    d["rand_seed"] = np.asarray([1234])
    d["rand_img"], d["rand_seg"], d["rand_pts"], d["rand_matrix"] = npy(a), npy(b), npy(c), npy(m)
    params = (torch.tensor([[1.1, 0.9, 1.05]]), torch.tensor([[0.1, -0.2, 0.05]]), torch.tensor([[0.3, -0.7, 1.9]]),
              torch.tensor([[0.02, -0.05, 0.08, 0.01, -0.03, 0.06]]))
    d["params_scale"], d["params_offset"], d["params_theta"], d["params_shear"] = (npy(p) for p in params)
    d["params_matrix"] = npy(AffineDeformation3d(device="cpu").build_affine_matrix(1, params))
    # eval metrics on a dense map (loss_ops.py:161-247), called like pairwise_register_eval.py:337-345 does
    from keymorph import loss_ops as L
    grid = TPS(points_m=pts + 0.08 * torch.randn(1, 9, 3, generator=g), points_f=pts,
               lmbda=torch.tensor(0.1).repeat(1), dim=3).get_flow_field((1, 1, 12, 14, 10))
    gp = grid.permute(0, 4, 1, 2, 3)
    d["jd_grid"] = npy(grid)
    d["jd_det"] = np.asarray(L._jacobian_determinant(gp.numpy()), np.float32)
    d["jd_std"] = np.asarray([L.jdstd(gp)], np.float64)
    d["jd_neg"] = np.asarray([L.jdlessthan0(gp), L.jdlessthan0(gp, as_percentage=True)], np.float64)
    fold = gp * torch.tensor([9.0, -14.0, 11.0]).reshape(1, 3, 1, 1, 1)    # scaled + mirrored: determinants of both signs
    d["jd_fold_scale"] = np.asarray([9.0, -14.0, 11.0], np.float32)
    d["jd_fold_neg"] = np.asarray([L.jdlessthan0(fold)], np.float64)
    d["jd_fold_std"] = np.asarray([L.jdstd(fold)], np.float64)
    np.savez_compressed(os.path.join(OUT, "augment_small.npz"), **d)
    print("augment_small.npz", len(d), "arrays")


def _world_affines(g):
    """Two different voxel->world matrices (anisotropic spacing, a small rotation, an origin): (1, 4, 4) each."""
    def one(spacing, angle, origin):
        c, s_ = np.cos(angle), np.sin(angle)
        R = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
        A = torch.eye(4)
        A[:3, :3] = R @ torch.diag(torch.tensor(spacing, dtype=torch.float32))
        A[:3, 3] = torch.tensor(origin, dtype=torch.float32)
        return A[None]
    return one((1.0, 1.2, 0.9), 0.15, (-4.0, 3.0, 1.5)), one((1.1, 0.8, 1.3), -0.1, (2.0, -1.0, 0.5))


def gen_realworld():
    """align_in_real_world_coords=True (keypoint_aligners.py:47-66, 134-148, 255-268, 431-465; utils.py:243-354):
    the three aligners on point sets, and KeyMorph(align_keypoints_in_real_world_coords=True) on the e2e_tiny pair."""
    from keymorph.utils import (convert_points_norm2real, convert_points_real2norm, convert_points_norm2voxel,
                                convert_points_voxel2norm)
    g = torch.Generator().manual_seed(31)
    d = {}
    aff_f, aff_m = _world_affines(g)
    d["aff_f"], d["aff_m"] = npy(aff_f), npy(aff_m)
    K = 14
    shape5 = (1, 1, 6, 7, 8)
    shape_f = torch.tensor([6.0, 7.0, 8.0])
    shape_m = torch.tensor([8.0, 6.0, 7.0])
    pf = torch.rand(1, K, 3, generator=g) * 1.6 - 0.8
    A = torch.eye(3) + 0.12 * torch.randn(3, 3, generator=g)
    pm = pf @ A.T + 0.08 * torch.randn(1, 1, 3, generator=g) + 0.03 * torch.randn(1, K, 3, generator=g)
    w = torch.rand(1, K, generator=g)
    w = w / w.sum()
    d["pf"], d["pm"], d["w"], d["shape_f"], d["shape_m"] = npy(pf), npy(pm), npy(w), npy(shape_f), npy(shape_m)
    d["pf_real"] = npy(convert_points_norm2real(pf, aff_f, shape_f))
    d["pf_voxel"] = npy(convert_points_norm2voxel(pf, shape_f))
    d["pf_back"] = npy(convert_points_real2norm(convert_points_norm2real(pf, aff_f, shape_f), aff_f, shape_f))
    d["pm_in_f"] = npy(convert_points_real2norm(convert_points_norm2real(pm, aff_m, shape_m), aff_f, shape_f))
    d["pf_vox2norm"] = npy(convert_points_voxel2norm(convert_points_norm2voxel(pf, shape_f), shape_f))
    common = dict(dim=3, align_in_real_world_coords=True, aff_f=aff_f, aff_m=aff_m, shape_f=shape_f, shape_m=shape_m)

    def make(name, pm_, pf_, wt):
        if name == "affine":
            return AffineKeypointAligner(points_m=pm_, points_f=pf_, w=wt, **common)
        if name == "rigid":
            return RigidKeypointAligner(points_m=pm_, points_f=pf_, w=wt, **common)
        return TPS(points_m=pm_, points_f=pf_, lmbda=torch.tensor(float(name[4:])).repeat(1), w=wt, **common)

    for name in ("affine", "rigid", "tps_10", "tps_1000"):
        for wt, wtag in ((None, ""), (w, "_w")):
            al = make(name, pm, pf, wt)
            if name in ("affine", "rigid"):
                d[f"{name}{wtag}::matrix"] = npy(al.transform_matrix)
            d[f"{name}{wtag}::grid"] = npy(al.get_flow_field(shape5))
            d[f"{name}{wtag}::points_a"] = npy(al.get_forward_transformed_points(pm))
            d[f"{name}{wtag}::points_inv"] = npy(al.get_inverse_transformed_points(pf))
        pf_ = pf.clone().requires_grad_(True)
        pm_ = pm.clone().requires_grad_(True)
        grid = make(name, pm_, pf_, None).get_flow_field(shape5)
        cot = torch.randn(grid.shape, generator=torch.Generator().manual_seed(8))
        (grid * cot).sum().backward()
        d[f"{name}::gridcot"], d[f"{name}::dpf"], d[f"{name}::dpm"] = npy(cot), npy(pf_.grad), npy(pm_.grad)

    # through KeyMorph.forward (model.py:163-170, 230-268) on the e2e_tiny pair and weights
    e = np.load(os.path.join(OUT, "e2e_tiny.npz"))
    img_f, img_m = torch.from_numpy(e["img_f"]), torch.from_numpy(e["img_m"])
    sd = {k[4:]: torch.from_numpy(e[k]) for k in e.files if k.startswith("sd::")}
    net = make_tunet(16, 8)
    net.load_state_dict(sd, strict=True)
    km = KeyMorph(net, 16, 3, max_train_keypoints=None, align_keypoints_in_real_world_coords=True).eval()
    with torch.no_grad():
        rr = km(img_f, img_m, transform_type=["rigid", "affine", "tps_10"], return_aligned_points=True,
                aff_f=aff_f, aff_m=aff_m)
    for tt in ("rigid", "affine", "tps_10"):
        d[f"km::{tt}::grid"] = npy(rr[tt]["grid"])
        d[f"km::{tt}::points_a"] = npy(rr[tt]["points_a"])
    km.train()
    for tt in ("affine", "tps_10"):
        km.zero_grad()
        r = km(img_f, img_m, transform_type=tt, return_aligned_points=False, aff_f=aff_f, aff_m=aff_m)[tt]
        mse = loss_ops.MSELoss()(img_f, align_img(r["grid"], img_m))
        mse.backward()
        d[f"km_train::{tt}::mse"] = npy(mse)
        d[f"km_train::{tt}::gradfull::final_conv.weight"] = npy(net.final_conv.weight.grad)
    np.savez_compressed(os.path.join(OUT, "realworld_small.npz"), **d)
    print("realworld_small.npz", len(d), "arrays")


def gen_onehot():
    """keymorph/utils.py:200-240: one_hot and one_hot_subsampled_pair (np.random.choice on the shared labels)."""
    from keymorph.utils import one_hot, one_hot_subsampled_pair
    g = torch.Generator().manual_seed(13)
    d = {}
    seg = torch.randint(0, 5, (2, 1, 4, 5, 6), generator=g)
    d["seg"], d["one_hot"] = npy(seg), npy(one_hot(seg))
    lab1 = torch.tensor([0, 2, 3, 5, 7, 8, 11, 12, 17, 20, 21, 30])
    lab2 = torch.tensor([0, 1, 3, 5, 7, 9, 11, 12, 17, 21, 25, 30, 31])
    seg1 = lab1[torch.randint(0, len(lab1), (1, 1, 6, 7, 8), generator=g)]
    seg2 = lab2[torch.randint(0, len(lab2), (1, 1, 6, 7, 8), generator=g)]
    d["seg1"], d["seg2"] = npy(seg1), npy(seg2)
    for num, seed in ((5, 3), (14, 4), (9, 5)):
        np.random.seed(seed)
        a, b = one_hot_subsampled_pair(seg1, seg2, num)
        d[f"sub{num}::seed"] = np.asarray([seed])
        d[f"sub{num}::a"], d[f"sub{num}::b"] = npy(a).astype(np.uint8), npy(b).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "onehot_small.npz"), **d)
    print("onehot_small.npz", len(d), "arrays")


def gen_gradients():
    """Full parameter-gradient vectors (every tensor) of tiny backbones and of the end-to-end training step.

    * `tunet16` / `unet16`: (Truncated)UNet3D, f_maps 8, K = 8, 16^3 input, loss = sum(y * cot).
    * `kinkfree`: the same TruncatedUNet3D on an input for which NO pre-ReLU value (fp64) lies within `margin` of 0
      (searched over seeds here), so that no implementation can flip a ReLU mask: gradients must agree to rounding.
    * `e2e16`: KeyMorph.forward + align_img + MSE (+ Dice), 16^3, affine / tps_1, full gradients."""
    import torch.nn as nn
    d = {}

    def pre_relu_margin(net, x):
        vals = []
        hooks = [m.register_forward_pre_hook(lambda mod, inp: vals.append(float(inp[0].double().abs().min())))
                 for m in net.modules() if isinstance(m, nn.ReLU)]      # pre-hook: the ReLUs are in-place
        with torch.no_grad():
            net(x)
        for h in hooks:
            h.remove()
        return min(vals)

    def record(tag, net, x, seed_cot):
        net.train()
        net.zero_grad()
        y = net(x)
        cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed_cot))
        (y * cot).sum().backward()
        d[f"{tag}::x"], d[f"{tag}::out"], d[f"{tag}::cot"] = npy(x), npy(y), npy(cot)
        for k, p in net.named_parameters():
            d[f"{tag}::grad::{k}"] = npy(p.grad)

    x16 = blob_volume((16, 16, 16), 61)
    for tag, net in (("tunet16", make_tunet(8, 8)), ("unet16", make_unet(8, 8))):
        sd = seeded_state_dict(net.state_dict(), 300)
        net.load_state_dict(sd, strict=True)
        d[f"{tag}::sdsum"] = np.float64(sd_checksum(sd))
        record(tag, net, x16, 5)

    # kink-free: search (input seed, weight seed) for the largest fp64 margin
    best = None
    net = make_tunet(8, 8, levels=3).double()
    for seed in range(400):
        sd = seeded_state_dict(net.state_dict(), 1000 + seed)
        net.load_state_dict({k: v.double() for k, v in sd.items()}, strict=True)
        x = blob_volume((8, 8, 8), 2000 + seed).double()
        m = pre_relu_margin(net, x)
        if best is None or m > best[0]:
            best = (m, seed)
        if m > 2e-4:
            break
    margin, seed = best
    print("kink-free margin", margin, "seed", seed)
    net = make_tunet(8, 8, levels=3)
    sd = seeded_state_dict(net.state_dict(), 1000 + seed)
    net.load_state_dict(sd, strict=True)
    d["kinkfree::margin"] = np.float64(margin)
    d["kinkfree::seed"] = np.asarray([seed])
    d["kinkfree::sdsum"] = np.float64(sd_checksum(sd))
    record("kinkfree", net, blob_volume((8, 8, 8), 2000 + seed), 6)

    # end to end at 16^3 (weights stored: the tests rebuild them from the same seeded recipe, checksum guards)
    K = 8
    img_f = blob_volume((16, 16, 16), 71)
    M = torch.eye(4)[None].clone()
    M[0, :3, :3] += torch.tensor([[0.05, 0.08, -0.03], [-0.06, -0.04, 0.05], [0.02, -0.07, 0.06]])
    M[0, :3, 3] = torch.tensor([0.06, -0.05, 0.04])
    flow = AffineTransform(matrix=M, dim=3).get_flow_field(img_f.shape)
    img_m = align_img(flow, img_f)
    seg_f = torch.stack([(img_f[0, 0] > t).float() for t in (0.0, 0.35, 0.6)])[None]
    seg_f = torch.cat([seg_f[:, :-1] - seg_f[:, 1:], seg_f[:, -1:]], 1)
    seg_m = align_img(flow, seg_f)
    d["e2e16::img_f"], d["e2e16::img_m"] = npy(img_f), npy(img_m)
    d["e2e16::seg_f"], d["e2e16::seg_m"] = npy(seg_f), npy(seg_m)
    net = make_tunet(K, 8)
    sd = seeded_state_dict(net.state_dict(), 310)
    net.load_state_dict(sd, strict=True)
    d["e2e16::sdsum"] = np.float64(sd_checksum(sd))
    km = KeyMorph(net, K, 3, max_train_keypoints=None).train()
    for tt in ("affine", "rigid", "tps_1"):
        for loss_name in (("mse", "dice") if tt == "affine" else ("mse",)):
            km.zero_grad()
            r = km(img_f, img_m, transform_type=tt, return_aligned_points=False)[tt]
            if loss_name == "mse":
                loss = loss_ops.MSELoss()(img_f, align_img(r["grid"], img_m))
            else:
                loss = loss_ops.DiceLoss()(align_img(r["grid"], seg_m), seg_f)
            loss.backward()
            d[f"e2e16::{tt}::{loss_name}::loss"] = npy(loss)
            d[f"e2e16::{tt}::{loss_name}::grid"] = npy(r["grid"])
            for k, p in net.named_parameters():
                d[f"e2e16::{tt}::{loss_name}::grad::{k}"] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "gradients_tiny.npz"), **d)
    print("gradients_tiny.npz", len(d), "arrays")


def gen_weighted_subsample():
    """Training with weight_keypoints='power', a TPS transform and max_train_keypoints < num_keypoints
    (keymorph/model.py:209-222: points AND weights are subsampled with one np.random.choice draw)."""
    g = np.load(os.path.join(OUT, "e2e_tiny.npz"))
    K = 16
    img_f, img_m = torch.from_numpy(g["img_f"]), torch.from_numpy(g["img_m"])
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    d = {}
    for weighting in ("power", None):
        net = make_tunet(K, 8)
        net.load_state_dict(sd, strict=True)
        km = KeyMorph(net, K, 3, max_train_keypoints=6, weight_keypoints=weighting).train()
        np.random.seed(17)
        r = km(img_f, img_m, transform_type="tps_1", return_aligned_points=True)["tps_1"]
        mse = loss_ops.MSELoss()(img_f, align_img(r["grid"], img_m))
        mse.backward()
        t = str(weighting)
        d[f"{t}::points_f"], d[f"{t}::points_m"] = npy(r["points_f"]), npy(r["points_m"])
        if weighting:
            d[f"{t}::weights"] = npy(r["points_weights"])
        d[f"{t}::grid"], d[f"{t}::points_a"], d[f"{t}::mse"] = npy(r["grid"]), npy(r["points_a"]), npy(mse)
        d[f"{t}::gradfull::final_conv.weight"] = npy(net.final_conv.weight.grad)
        d[f"{t}::gradfull::final_conv.bias"] = npy(net.final_conv.bias.grad)
    d["np_seed"] = np.asarray([17])
    np.savez_compressed(os.path.join(OUT, "weighted_subsample.npz"), **d)
    print("weighted_subsample.npz", len(d), "arrays")


def gen_trainstep():
    """Two iterations of scripts/train.py:39-176 composed from the reference's own functions (loader dict ->
    one_hot_subsampled_pair -> random_affine_augment -> KeyMorph.forward -> align_img img + seg -> MSE / Dice ->
    backward -> torch.optim.Adam), with the script's seeds, for both loss branches; plus the checkpoint the
    loop would write after step 1 (run.py:588-602 keys).  The loader is the torchio-free dict shape the loop reads."""
    from keymorph.utils import one_hot_subsampled_pair
    from keymorph.augmentation import random_affine_augment
    d = {}
    K, S = 8, 16
    g = torch.Generator().manual_seed(91)
    subjects = []
    for i in range(2):
        img = blob_volume((S, S, S), 400 + i)
        lab = torch.zeros((1, 1, S, S, S), dtype=torch.int64)
        for j, t in enumerate((0.2, 0.4, 0.6, 0.8)):
            lab[img > t] = j + 1
        aff = torch.eye(4)[None].clone()
        aff[0, :3, 3] = torch.tensor([1.0 * i, -2.0, 0.5])
        subjects.append({"img": {"data": img, "affine": aff}, "seg": {"data": lab}})
        d[f"sub{i}::img"], d[f"sub{i}::seg"], d[f"sub{i}::affine"] = npy(img), npy(lab).astype(np.uint8), npy(aff)
    for loss_fn in ("mse", "dice"):
        net = make_tunet(K, 8)
        sd = seeded_state_dict(net.state_dict(), 320)
        net.load_state_dict(sd, strict=True)
        d["sdsum"] = np.float64(sd_checksum(sd))
        km = KeyMorph(net, K, 3, max_train_keypoints=None).train()
        opt = torch.optim.Adam(km.parameters(), lr=1e-3)
        torch.manual_seed(23)
        np.random.seed(23)
        for step in range(2):
            fixed, moving = subjects[step % 2], subjects[(step + 1) % 2]
            img_f, img_m = fixed["img"]["data"], moving["img"]["data"]
            aff_f, aff_m = fixed["img"]["affine"], moving["img"]["affine"]
            seg_f, seg_m = one_hot_subsampled_pair(fixed["seg"]["data"].long(), moving["seg"]["data"].long(), 3)
            img_f, img_m, seg_f, seg_m = img_f.float(), img_m.float(), seg_f.float(), seg_m.float()
            img_m, seg_m, aug = random_affine_augment(img_m, seg=seg_m, max_random_params=(0.2, 0.2, 3.1416, 0.1),
                                                      scale_params=0.3, return_affine_matrix=True)
            aff_m = torch.bmm(aff_m, aug)
            opt.zero_grad()
            r = km(img_f, img_m, transform_type="affine", return_aligned_points=False, aff_f=aff_f, aff_m=aff_m)["affine"]
            img_a = align_img(r["grid"], img_m)
            seg_a = align_img(r["grid"], seg_m)
            mse = loss_ops.MSELoss()(img_f, img_a)
            dice = loss_ops.DiceLoss()(seg_a, seg_f)
            loss = mse if loss_fn == "mse" else dice
            loss.backward()
            t = f"{loss_fn}::step{step}"
            d[f"{t}::aug_matrix"], d[f"{t}::img_m_aug"] = npy(aug), npy(img_m)
            d[f"{t}::seg_f"], d[f"{t}::seg_m_aug"] = npy(seg_f).astype(np.uint8), npy(seg_m)
            d[f"{t}::mse"], d[f"{t}::softdiceloss"] = npy(mse), npy(dice)
            d[f"{t}::grid"] = npy(r["grid"])
            d[f"{t}::grad::final_conv.weight"] = npy(net.final_conv.weight.grad)
            opt.step()
            d[f"{t}::after::final_conv.weight"] = npy(net.final_conv.weight)
            d[f"{t}::after::enc0"] = npy(net.encoders[0].basic_module.SingleConv1.conv.weight)
            if step == 0:
                state = {"epoch": 1, "args": None, "state_dict": km.backbone.state_dict(), "optimizer": opt.state_dict()}
                d[f"{loss_fn}::ckpt_keys"] = np.asarray(sorted(state.keys()))
                d[f"{loss_fn}::ckpt_sd_keys"] = np.asarray(list(state["state_dict"].keys()))
                d[f"{loss_fn}::ckpt_opt_step"] = np.asarray([float(v["step"]) for v in state["optimizer"]["state"].values()])
    np.savez_compressed(os.path.join(OUT, "trainstep_tiny.npz"), **d)
    print("trainstep_tiny.npz", len(d), "arrays")


def gen_cfg1():
    """BASELINE.json configs[0]: the example_data_half pair (scripts/hyperparameters.py:4-11 resizes to 128^3), 128
    keypoints, affine aligner, reference CPU path.  The intensity images are not in the mount: intensity
    = label / 13 from example_data_half/seg_m/*.nii.gz, nearest-down-sampled 256^3 -> 128^3 (every second voxel).
    The fixture holds the two 128^3 label maps (uint8, data) and what the reference computes for them with the
    seeded TruncatedUNet3D(f_maps 32) weights: keypoints, matrix, losses, sub-sampled grid / warped volume, and
    summaries of every parameter gradient."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from keymorph_amd.io.nifti import read_nifti
    d = {}
    labs = []
    for i, name in enumerate(("IXI_001_128x128x128.nii.gz", "IXI_002_128x128x128.nii.gz")):
        arr, aff = read_nifti(os.path.join(REF, "example_data_half", "seg_m", name), dtype=None)
        lab = np.ascontiguousarray(np.asarray(arr)[::2, ::2, ::2]).astype(np.uint8)
        assert lab.shape == (128, 128, 128) and lab.max() == 13
        labs.append(lab)
        d[f"label_{i}"], d[f"affine_{i}"] = lab, aff
    K = 128
    img_f = torch.from_numpy(labs[0].astype(np.float32) / 13.0)[None, None]
    img_m = torch.from_numpy(labs[1].astype(np.float32) / 13.0)[None, None]
    seg_f = torch.stack([torch.from_numpy((labs[0] == c).astype(np.float32)) for c in range(14)])[None]
    seg_m = torch.stack([torch.from_numpy((labs[1] == c).astype(np.float32)) for c in range(14)])[None]
    net = make_tunet(K, 32)
    sd = seeded_state_dict(net.state_dict(), 23)
    net.load_state_dict(sd, strict=True)
    d["sdsum"] = np.float64(sd_checksum(sd))
    torch.set_num_threads(8)
    km = KeyMorph(net, K, 3, max_train_keypoints=None).train()
    import time
    t0 = time.time()
    r = km(img_f, img_m, transform_type="affine", return_aligned_points=True)["affine"]
    img_a = align_img(r["grid"], img_m)
    seg_a = align_img(r["grid"], seg_m)
    mse = loss_ops.MSELoss()(img_f, img_a)
    dice = loss_ops.DiceLoss()(seg_a, seg_f)
    mse.backward()
    d["ref_seconds_fwd_bwd_8_threads"] = np.asarray([time.time() - t0])
    d["points_f"], d["points_m"], d["points_a"] = npy(r["points_f"]), npy(r["points_m"]), npy(r["points_a"])
    d["matrix"], d["mse"], d["softdiceloss"] = npy(r["matrix"]), npy(mse), npy(dice)
    with torch.no_grad():
        d["harddiceloss"] = npy(loss_ops.DiceLoss(hard=True)(seg_a, seg_f, ign_first_ch=True))
    d["grid_sub8"] = npy(r["grid"][:, ::8, ::8, ::8])
    d["img_a_sub4"] = npy(img_a[:, :, ::4, ::4, ::4])
    for k, p in net.named_parameters():
        gflat = p.grad.reshape(-1)
        d[f"gradsum::{k}"] = npy(torch.cat([gflat.sum()[None], gflat.abs().sum()[None], gflat.norm()[None], gflat[:8]]))
    d["gradfull::final_conv.bias"] = npy(net.final_conv.bias.grad)
    d["gradfull::enc0"] = npy(net.encoders[0].basic_module.SingleConv1.conv.weight.grad)
    np.savez_compressed(os.path.join(OUT, "cfg1_example_half_128.npz"), **d)
    print("cfg1_example_half_128.npz", len(d), "arrays; reference fwd+bwd", float(d["ref_seconds_fwd_bwd_8_threads"][0]), "s")


def gen_groupwise_eval():
    """scripts/groupwise_register_eval.py:375-527 composed from the reference's own functions on a 3-subject group:
    grids from KeyMorph.groupwise_register, aligned images / segmentations, and the metrics dictionary it would write
    to metrics-{type}.json (MSEPairwiseLoss, MultipleAvgSegPairwiseMetric, MultipleAvgGridMetric)."""
    import json
    g = np.load(os.path.join(OUT, "groupwise_tiny.npz"))
    d = {}
    K = 16
    net = make_tunet(K, 8)
    net.load_state_dict(seeded_state_dict(net.state_dict(), 200), strict=True)
    km = KeyMorph(net, K, 3, max_train_keypoints=None).eval()
    types = ["affine", "tps_1"]
    with tempfile.TemporaryDirectory() as td:
        img_dir, seg_dir, res_dir = (os.path.join(td, n) for n in ("img_m", "seg_m", "registration_results"))
        for p in (img_dir, seg_dir, res_dir):
            os.makedirs(p)
        for i in range(3):
            img = torch.from_numpy(g[f"img_{i}"])
            seg = torch.stack([(img[0, 0] > t).float() for t in (-1.0, 0.3, 0.5, 0.7)])[None]
            seg = torch.cat([seg[:, :-1] - seg[:, 1:], seg[:, -1:]], 1)
            d[f"seg_{i}"] = npy(seg).astype(np.uint8)
            np.savez(os.path.join(img_dir, f"img_m_{i:03}.npz"), img=npy(img))
            np.savez(os.path.join(seg_dir, f"seg_m_{i:03}.npz"), seg=npy(seg))
        with torch.no_grad():
            res = km.groupwise_register(img_dir, transform_type=types, device="cpu", save_results_to_disk=True,
                                        save_dir=res_dir, plot=False, num_iters=5, log_to_console=False,
                                        num_resolutions_for_itkelastix=None)
        img_paths = sorted(os.path.join(img_dir, f) for f in os.listdir(img_dir))
        seg_paths = sorted(os.path.join(seg_dir, f) for f in os.listdir(seg_dir))
        for tt in types:
            grids = sorted(os.path.join(res_dir, f) for f in os.listdir(res_dir) if f.startswith(tt))
            ia, sa = [], []
            for i in range(3):
                grid = torch.tensor(np.load(grids[i]))
                img_a = align_img(grid, torch.tensor(np.load(img_paths[i])["img"]))
                seg_a = align_img(grid, torch.tensor(np.load(seg_paths[i])["seg"]))
                ia.append(os.path.join(td, f"img_a_{tt}_{i:03}.npy"))
                sa.append(os.path.join(td, f"seg_a_{tt}_{i:03}.npy"))
                np.save(ia[-1], npy(img_a))
                np.save(sa[-1], npy(seg_a))
                if i == 1:
                    d[f"{tt}::img_a_1"], d[f"{tt}::seg_a_1"] = npy(img_a), npy(seg_a)
            m = {"mse": loss_ops.MSEPairwiseLoss()(ia).item()}
            sm = loss_ops.MultipleAvgSegPairwiseMetric()(sa, ["softdice", "harddice", "harddiceroi"])
            sm["harddice"] = (1 - sm["harddice"]).item()
            sm["harddiceroi"] = (1 - sm["harddiceroi"]).tolist()
            sm["softdice"] = (1 - sm["softdice"]).item()
            gm = loss_ops.MultipleAvgGridMetric()(grids, ["jdstd", "jdlessthan0"])
            m = m | sm | gm
            d[f"{tt}::metrics_json"] = np.asarray(json.dumps(m, sort_keys=True))
            d[f"{tt}::points_a0"] = npy(res[tt]["grouppoints_a"][0])
        d["points_m0"] = npy(res[types[0]]["grouppoints_m"][0])
    np.savez_compressed(os.path.join(OUT, "groupwise_eval_tiny.npz"), **d)
    print("groupwise_eval_tiny.npz", len(d), "arrays")


def gen_valid_coords():
    """keymorph/utils.py:97-162: sample_valid_coordinates (rejection sampling with np.random.randint draws; used by
    scripts/run.py:528-548 to draw the pre-training reference keypoints)."""
    import contextlib, io
    from keymorph.utils import sample_valid_coordinates
    g = torch.Generator().manual_seed(41)
    d = {}
    x3 = torch.rand(1, 1, 7, 9, 11, generator=g) * (torch.rand(1, 1, 7, 9, 11, generator=g) > 0.6)
    x2 = torch.rand(1, 1, 8, 13, generator=g) * (torch.rand(1, 1, 8, 13, generator=g) > 0.5)
    d["x3"], d["x2"] = npy(x3), npy(x2)
    for tag, x, dim, seed, space, indexing in (("a", x3, 3, 5, "norm", "xy"), ("b", x3, 3, 6, "norm", "ij"),
                                               ("c", x3, 3, 7, "voxel", "xy"), ("d", x2, 2, 8, "norm", "xy"),
                                               ("e", x2, 2, 9, "voxel", "ij")):
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):            # the reference prints a progress line per point
            pts = sample_valid_coordinates(x, 6, dim, point_space=space, indexing=indexing)
        d[f"{tag}::seed"] = np.asarray([seed])
        d[f"{tag}::points"] = np.asarray(pts.numpy(), dtype=np.float64)
        d[f"{tag}::dtype"] = np.asarray(str(pts.dtype))
        d[f"{tag}::after"] = np.asarray([np.random.randint(0, 1 << 30)])   # the generator state the caller is left with
    np.savez_compressed(os.path.join(OUT, "valid_coords_small.npz"), **d)
    print("valid_coords_small.npz", len(d), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gens = {"ops": gen_ops, "backbones": gen_backbones, "e2e": gen_e2e, "groupwise": gen_groupwise,
            "tps_illcond": gen_tps_illcond, "augment": gen_augment, "weighted": gen_weighted,
            "groupwise_truth": gen_groupwise_truth, "realworld": gen_realworld, "onehot": gen_onehot,
            "gradients": gen_gradients, "weighted_subsample": gen_weighted_subsample, "trainstep": gen_trainstep,
            "cfg1": gen_cfg1, "groupwise_eval": gen_groupwise_eval, "valid_coords": gen_valid_coords}
    for name in (sys.argv[1:] or list(gens)):      # e.g. `make_golden.py augment` regenerates one fixture
        torch.manual_seed(0)
        np.random.seed(0)
        gens[name]()
