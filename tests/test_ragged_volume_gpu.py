"""GPU: the whole path on a reference-shaped RAGGED volume -- 120 x 120 x 90, half of example_data's 241 x 240 x 180
 -- against the oracle run on the host cores.  Nothing here is a multiple of the 32 x 8 x 4 convolution
brick, the pooled levels are 60 x 60 x 45 -> 30 x 30 x 22 -> 15 x 15 x 11 (the second decoder joins 30 x 30 x 22
up to 60 x 60 x 45: NOT an exact 2x, so the plain upsample + concat route runs next to the fused operator), and the
256-brick threshold puts the LDS-DMA kernel on the top level only -- the test therefore runs once with the default
kernel selection and once with conv3_fwd_g_kernel forced everywhere it is legal.
North-star bar: keypoints, grid, warped volume, MSE within 1e-4 of the reference arithmetic (fp32 CPU path)."""
import numpy as np
import pytest
import torch

from tests.util import seeded_state_dict, unet_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda"
K = 128
SHAPE = (120, 120, 90)


def close(a, b, atol, rtol=0):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.fixture(scope="module")
def world():
    from keymorph_amd import synthetic
    from oracle import keymorph_oracle as O
    sd = seeded_state_dict(unet_shapes(K, 32, trunc=1), 23)
    img_f, img_m = synthetic.make_pair(SHAPE, 11, torch.device(DEV))
    with torch.no_grad():        # the oracle's backbone once (~10 s on the host cores), its aligners per transform
        heat = O.unet3d_forward(sd, torch.cat([img_f, img_m]).cpu(), 4, 1, 8)
        assert heat.shape == (2, K, 60, 60, 45)
        pts = O.center_of_mass(heat, "ij")
    return dict(sd=sd, img_f=img_f, img_m=img_m, opf=pts[:1], opm=pts[1:])


@pytest.mark.parametrize("force_g", [False, True])
def test_ragged_120x120x90_vs_oracle(world, force_g):
    from keymorph_amd import _lib, ops
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    from oracle import keymorph_oracle as O
    lib = _lib.load()
    net = TruncatedUNet3D(1, K, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    net.load_state_dict(world["sd"], strict=True)
    km = KeyMorph(net, K, 3, max_train_keypoints=None).to(DEV).train()
    img_f, img_m = world["img_f"], world["img_m"]
    old = lib.kmh_conv3d_fwd_bf_set_dispatch(2 if force_g else 1)
    try:
        # top level (N = 2 images in one batch): 2 * 3 * 15 * 30 bricks -> the LDS-DMA kernel by default as well
        assert lib.kmh_conv3d_fwd_bf_variant(2, *SHAPE, 16, 32, 2, 0, 0) == 1
        assert (lib.kmh_conv3d_fwd_bf_variant(2, 60, 60, 45, 32, 64, 2, 0, 0) == 2) == force_g
        res = {}
        for tt in ("affine", "tps_1"):          # training mode takes one transform type per call (model.py:165)
            r = res[tt] = km(img_f, img_m, transform_type=tt, return_aligned_points=True)[tt]
            ro = O.register(world["opf"], world["opm"], tt, SHAPE, True)
            close(r["points_f"], world["opf"], 1e-4)
            close(r["points_m"], world["opm"], 1e-4)
            close(r["points_a"], ro["points_a"], 1e-4)
            close(r["grid"], ro["grid"], 1e-4)
            loss, img_a = ops.warp_mse(img_m, r["grid"], img_f)
            img_ao = O.align_img(ro["grid"], img_m.cpu())
            close(img_a, img_ao, 1e-4)
            close(loss, O.mse_loss(img_f.cpu(), img_ao), 1e-6)
            e_pts = float((r["points_f"].cpu() - world["opf"]).abs().max())
            print(f"ragged {SHAPE} {tt} force_g={force_g}: keypoints {e_pts:.2e}, grid "
                  f"{float((r['grid'].detach().cpu() - ro['grid']).abs().max()):.2e}")
        # and a backward through all of it: finite, non-zero gradients in every parameter tensor
        loss, _ = ops.warp_mse(img_m, res["tps_1"]["grid"], img_f)
        loss.backward()
        for k, p in net.named_parameters():
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, k
    finally:
        lib.kmh_conv3d_fwd_bf_set_dispatch(old)


def test_ragged_gradients_identical_under_both_kernel_selections(world):
    """conv3_fwd_g_kernel and conv3_fwd_bf_kernel are bit-identical per launch, so the whole training step must be:
    every parameter gradient of the 120 x 120 x 90 pair equal under both dispatch modes."""
    from keymorph_amd import _lib, ops
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    lib = _lib.load()
    net = TruncatedUNet3D(1, K, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    net.load_state_dict(world["sd"], strict=True)
    km = KeyMorph(net, K, 3, max_train_keypoints=None).to(DEV).train()
    grads = {}
    old = lib.kmh_conv3d_fwd_bf_set_dispatch(1)
    try:
        for mode in (0, 2):
            lib.kmh_conv3d_fwd_bf_set_dispatch(mode)
            for p in net.parameters():
                p.grad = None
            r = km(world["img_f"], world["img_m"], transform_type="affine", return_aligned_points=False)["affine"]
            loss, _ = ops.warp_mse(world["img_m"], r["grid"], world["img_f"])
            loss.backward()
            grads[mode] = {k: p.grad.clone() for k, p in net.named_parameters()}
    finally:
        lib.kmh_conv3d_fwd_bf_set_dispatch(old)
    # Per launch the two kernels are bit-identical (tests/test_conv_dispatch_gpu.py); over a whole step the epilogue
    # statistics are summed over bricks of another height (fp32 partials of <= 64 values), GroupNorm coefficients move
    # in the last bit, keypoints by ~1e-7, and the affine fit of a random-init network's clumped keypoints amplifies
    # that 100-300x ("keypoint noise floor").  So: the whole gradient vector to 1e-3, every tensor
    # to 3e-2 (the worst is the one-element GroupNorm weight of the first layer, a 1.3 M-term sum that cancels to ~0).
    errs = {k: float((grads[0][k].double() - grads[2][k].double()).norm() / (grads[2][k].double().norm() + 1e-300))
            for k in grads[0]}
    va = torch.cat([grads[0][k].reshape(-1).double() for k in grads[0]])
    vb = torch.cat([grads[2][k].reshape(-1).double() for k in grads[2]])
    whole = float((va - vb).norm() / vb.norm())
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(f"ragged: gradients under the two kernel selections: whole vector {whole:.2e}; worst tensors {top}")
    assert whole < 1e-3 and top[0][1] < 3e-2, (whole, top)
