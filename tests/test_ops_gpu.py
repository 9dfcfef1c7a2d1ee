"""GPU: every HIP op vs the CPU oracle on the same seeded inputs (and vs the reference goldens).

Tolerances are written per test; 1e-4 fp32 is the north-star bar, most ops are far inside it.
"""
import numpy as np
import pytest
import torch

from oracle import keymorph_oracle as O
from tests.util import T, golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from keymorph_amd import ops as _ops
    return _ops


def close(a, b, atol=1e-5, rtol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------- sampler
@pytest.mark.parametrize("shape", [((1, 3, 6, 7, 8), (5, 6, 7)), ((2, 1, 16, 12, 20), (16, 12, 20)),
                                   ((1, 2, 9, 9, 9), (3, 5, 7)),
                                   # many channels (14 = the one-hot segmentation of the Dice branch)
                                   ((2, 14, 11, 12, 21), (11, 12, 21)), ((1, 6, 7, 9, 33), (8, 10, 40)),
                                   ((1, 4, 5, 6, 7), (5, 6, 7))])
def test_grid_sample_fwd_bwd(shape):
    xs, gs = shape
    g = gen(3)
    x = torch.rand(xs, generator=g)
    grid = (torch.rand((xs[0],) + gs + (3,), generator=g) * 2.6 - 1.3)
    cot = torch.randn((xs[0], xs[1]) + gs, generator=g)
    gr = grid.clone().requires_grad_(True)
    ref = O.align_img(gr, x)
    (ref * cot).sum().backward()
    gh = grid.to(DEV).requires_grad_(True)
    out = ops().grid_sample3d(x.to(DEV), gh)
    (out * cot.to(DEV)).sum().backward()
    close(out, ref, 1e-6)
    close(gh.grad, gr.grad, 2e-5 * max(1, xs[1] // 2), 3e-5)      # a sum over the channels: rounding grows with C
    close(ops().grid_sample3d(x.to(DEV), grid.to(DEV), "nearest"), O.align_img(grid, x, "nearest"), 0)


@pytest.mark.parametrize("kind", ["random", "affine", "zoom_in", "border"])
def test_multichannel_sampler_equals_per_channel(kind):
    """C >= 2 bilinear warps run sample_fwd_mc_kernel (persistent 16 x 8 x 8 tiles; per channel the tile's source box goes
    through LDS, tiles whose box does not fit gather from memory; align_img of a one-hot segmentation,
    keymorph/utils.py:14-21): bit-equal to warping the channels one by one (the single-channel kernel) and within 3e-6 of
    the oracle -- for a random grid (every tile takes the memory gathers), a three-axis affine grid (box path, ragged edge
    tiles), a zoom-in (tiny boxes), and a grid that leaves the volume on every side (clipped corners, x0 = W - 1 pairs)."""
    from keymorph_amd import synthetic
    from keymorph_amd.transformations import AffineTransform
    g = gen(8)
    x = torch.rand(2, 5, 36, 45, 52, generator=g)
    x[0, 1, :, :, 0] = float("inf")          # a column the x0 = W - 1 pairs must not touch through the next row's first voxel
    if kind == "random":
        grid = torch.rand(2, 19, 33, 47, 3, generator=g) * 2.4 - 1.2
    else:
        mats = torch.cat([synthetic.random_affine_matrix(s_, DEV) for s_ in (3, 4)])
        grid = AffineTransform(matrix=mats, dim=3).get_flow_field((2, 1, 35, 41, 50)).contiguous().cpu()
        if kind == "zoom_in":
            grid = grid * 0.3
        if kind == "border":
            grid = grid * 1.5 + 0.2
    xd, gd = x.to(DEV), grid.to(DEV)
    out = ops().grid_sample3d(xd, gd)
    for c in range(x.shape[1]):
        one = ops().grid_sample3d(xd[:, c:c + 1].contiguous(), gd)
        same = (out[:, c:c + 1] == one) | (torch.isnan(one) & torch.isnan(out[:, c:c + 1]))
        assert bool(same.all()), (kind, c, int((~same).sum()))
    ref = O.align_img(grid, x)
    ok = torch.isfinite(ref)
    close(torch.where(ok, out.cpu(), torch.zeros(())), torch.where(ok, ref, torch.zeros(())), 3e-6)      # (steep grids: the coordinate rounding)
    assert bool((torch.isfinite(out.cpu()) == ok).all())          # and infinities / NaNs exactly where the reference has them


def test_grid_sample_golden():
    o = golden("ops_small.npz")
    x, grid = T(o["warp_x"]).to(DEV), T(o["warp_grid"]).to(DEV).requires_grad_(True)
    out = ops().grid_sample3d(x, grid)
    close(out, o["warp_out"], 1e-6)
    (out * T(o["warp_cot"]).to(DEV)).sum().backward()
    close(grid.grad, o["warp_dgrid"], 2e-5)
    close(ops().grid_sample3d(x, grid.detach(), "nearest"), o["warp_out_nearest"], 0)


def test_grid_sample_bwd_input():
    g = gen(4)
    x = torch.rand(1, 2, 5, 6, 7, generator=g)
    grid = torch.rand(1, 4, 5, 6, 3, generator=g) * 2.4 - 1.2
    cot = torch.randn(1, 2, 4, 5, 6, generator=g)
    xr = x.clone().requires_grad_(True)
    (O.align_img(grid, xr) * cot).sum().backward()
    xh = x.to(DEV).requires_grad_(True)
    (ops().grid_sample3d(xh, grid.to(DEV)) * cot.to(DEV)).sum().backward()
    close(xh.grad, xr.grad, 1e-5)


def test_identity_is_not_identity():
    """linspace(-1,1) grid sampled with align_corners=False is a slight zoom."""
    x = torch.rand(1, 1, 8, 10, 12, generator=gen(5))
    eye = torch.eye(3, 4)[None]
    grid = ops().affine_grid(eye.to(DEV), (8, 10, 12))
    out = ops().grid_sample3d(x.to(DEV), grid)
    close(out, O.align_img(O.affine_grid(O.square(eye), (8, 10, 12)), x), 1e-6)
    assert (out.cpu() - x).abs().max() > 0.05


# ---------------------------------------------------------------- losses
def test_mse_and_fused_warp_mse():
    g = gen(6)
    a = torch.rand(2, 1, 17, 13, 11, generator=g)
    b = torch.rand(2, 1, 17, 13, 11, generator=g)
    ah = a.to(DEV).requires_grad_(True)
    l = ops().mse_loss(ah, b.to(DEV))
    l.backward()
    ar = a.clone().requires_grad_(True)
    lr = O.mse_loss(ar, b)
    lr.backward()
    close(l, lr, 1e-7)
    close(ah.grad, ar.grad, 1e-9, 1e-5)
    # fused
    grid = torch.rand(2, 17, 13, 11, 3, generator=g) * 2.2 - 1.1
    gh = grid.to(DEV).requires_grad_(True)
    lf, warped = ops().warp_mse(a.to(DEV), gh, b.to(DEV))
    lf.backward()
    gr = grid.clone().requires_grad_(True)
    wr = O.align_img(gr, a)
    lref = O.mse_loss(b, wr)
    lref.backward()
    close(lf, lref, 1e-7)
    close(warped, wr, 1e-6)
    close(gh.grad, gr.grad, 1e-8, 2e-4)


@pytest.mark.parametrize("shape,scale", [((2, 1, 16, 12, 20), 1.0), ((1, 3, 8, 24, 16), 2.5), ((2, 1, 17, 13, 11), 1.0)])
def test_warp_mse_single_pass_gradient(shape, scale):
    """warp + MSE + d(loss)/d(grid) in ONE launch (kmh_warp_mse_fwd_grad; the third shape has no 16-byte-aligned grid
    and takes the three-launch route): loss, warped volume and grid gradient vs autograd of the oracle, also under a
    non-unit upstream cotangent (the device-side conditional rescale)."""
    g = gen(16)
    n, c, d, h, w = shape
    a, b = torch.rand(shape, generator=g), torch.rand(shape, generator=g)
    grid = torch.rand(n, d, h, w, 3, generator=g) * 2.4 - 1.2          # includes out-of-range coordinates (border clamp)
    gh = grid.to(DEV).requires_grad_(True)
    lf, warped = ops().warp_mse(a.to(DEV), gh, b.to(DEV))
    (lf * scale).backward()
    gr = grid.clone().requires_grad_(True)
    wr = O.align_img(gr, a)
    lref = O.mse_loss(b, wr)
    (lref * scale).backward()
    close(lf, lref, 1e-7)
    close(warped, wr, 1e-6)
    close(gh.grad, gr.grad, 1e-8, 2e-4)
    with torch.no_grad():                                              # no gradient requested: plain fused forward
        l2, w2 = ops().warp_mse(a.to(DEV), grid.to(DEV), b.to(DEV))
    close(l2, lref, 1e-7)
    close(w2, wr, 1e-6)


def test_dice_golden_and_grad():
    from keymorph_amd import loss_ops
    o = golden("ops_small.npz")
    a, b = T(o["loss_a"]).to(DEV), T(o["loss_b"]).to(DEV)
    close(loss_ops.MSELoss()(a, b), o["mse"], 1e-7)
    close(loss_ops.DiceLoss()(a, b), o["dice_soft"], 1e-6)
    close(loss_ops.DiceLoss()(a, b, ign_first_ch=True), o["dice_soft_ign"], 1e-6)
    close(loss_ops.DiceLoss(hard=True)(a, b), o["dice_hard"], 1e-6)
    close(loss_ops.DiceLoss(hard=True, return_regions=True)(a, b), o["dice_hard_regions"], 1e-6)
    a_ = a.clone().requires_grad_(True)
    loss_ops.DiceLoss()(a_, b).backward()
    close(a_.grad, o["dice_soft_dpred"], 1e-8, 1e-4)


@pytest.mark.parametrize("shape,C,N", [((9, 10, 11), 3, 2), ((16, 16, 16), 14, 1), ((5, 33, 70), 5, 2), ((24, 20, 50), 1, 1)])
def test_fused_warp_dice_vs_oracle_and_unfused(shape, C, N, monkeypatch):
    """loss_ops.warp_dice_loss == DiceLoss()(align_img(grid, seg_m), seg_f) (scripts/train.py:146-164) in value and in
    d/d(grid): against the oracle's autograd (keymorph/utils.py:14-21 + loss_ops.py:16-63 restated) and against this
    package's unfused three-launch route; one-hot and soft segmentations, ragged sizes, grids that leave the volume
    (border padding), the 1024-voxel chunk tail, ign_first_ch / return_regions."""
    from keymorph_amd import loss_ops, utils
    g = gen(61 + C)
    D, H, W = shape
    lab_m = torch.randint(0, C, (N, 1, D, H, W), generator=g)
    lab_f = torch.randint(0, C, (N, 1, D, H, W), generator=g)
    hot = lambda lab: torch.zeros(N, C, D, H, W).scatter_(1, lab, 1.0)       # noqa: E731
    for soft in (False, True):
        seg_m = hot(lab_m) if not soft else torch.rand(N, C, D, H, W, generator=g)
        seg_f = hot(lab_f) if not soft else torch.rand(N, C, D, H, W, generator=g)
        grid = (O.base_grid(shape).flip(-1)[None].repeat(N, 1, 1, 1, 1) * 1.08
                + 0.05 * torch.randn(N, D, H, W, 3, generator=g))              # some samples fall outside [-1, 1]
        for kw in ({}, {"ign_first_ch": True}, {"return_regions": True}):
            if C == 1 and kw.get("ign_first_ch"):
                continue
            cot = torch.randn(C, generator=g) if kw.get("return_regions") else torch.tensor(1.0)
            gr = grid.clone().requires_grad_(True)
            ref = O.dice_loss(O.align_img(gr, seg_m), seg_f, **kw)
            (ref * cot).sum().backward()
            gh = grid.to(DEV).requires_grad_(True)
            out = loss_ops.warp_dice_loss(gh, seg_m.to(DEV), seg_f.to(DEV), **kw)
            (out * cot.to(DEV)).sum().backward()
            gu = grid.to(DEV).requires_grad_(True)
            unf = loss_ops.DiceLoss(return_regions=bool(kw.get("return_regions")))(
                utils.align_img(gu, seg_m.to(DEV)), seg_f.to(DEV), ign_first_ch=bool(kw.get("ign_first_ch")))
            (unf * cot.to(DEV)).sum().backward()
            close(out, ref, 2e-6)
            close(out, unf, 1e-6)
            scale = max(float(gr.grad.abs().max()), 1e-6)      # (C = 1 one-hot: a constant volume, zero gradient)
            close(gh.grad, gr.grad, 2e-5 * scale, 1e-4)
            close(gh.grad, gu.grad, 2e-6 * scale, 1e-5)
            if not soft and not kw:
                # exactly one-hot inputs took the label-map kernels (one byte per voxel): the dense kernels, forced by
                # KEYMORPH_DICE_NO_LABELS, must give the same bits
                monkeypatch.setenv("KEYMORPH_DICE_NO_LABELS", "1")
                gd = grid.to(DEV).requires_grad_(True)
                dense = loss_ops.warp_dice_loss(gd, seg_m.to(DEV), seg_f.to(DEV))
                dense.backward()
                monkeypatch.delenv("KEYMORPH_DICE_NO_LABELS")
                assert torch.equal(dense, out) and torch.equal(gd.grad, gh.grad)


def test_fused_warp_dice_falls_back_when_the_moving_segmentation_needs_a_gradient():
    from keymorph_amd import loss_ops, ops as kops
    g = gen(9)
    seg_m = torch.rand(1, 2, 6, 6, 6, generator=g).to(DEV).requires_grad_(True)
    seg_f = torch.rand(1, 2, 6, 6, 6, generator=g).to(DEV)
    grid = O.base_grid((6, 6, 6)).flip(-1)[None].to(DEV).requires_grad_(True)
    assert not kops.warp_dice_ok(seg_m, grid)
    loss_ops.warp_dice_loss(grid, seg_m, seg_f).backward()
    assert seg_m.grad is not None and grid.grad is not None


def test_fused_warp_dice_predicate_is_the_kernels_own_and_the_fallback_is_taken():
    """`ops.warp_dice_ok` is the library's predicate (the one its entry points return -22 on), so a shape the fused kernels do
    not serve -- > 128 channels; N * C beyond the partial-sum workspace, where the block count used to be rounded UP past
    it -- falls back to align_img + DiceLoss instead of raising; a many-row shape that IS served agrees with the fallback."""
    from keymorph_amd import _lib, loss_ops, ops as kops
    from keymorph_amd.utils import align_img
    lib = _lib.load()
    assert lib.kmh_warp_dice_ok(2, 14, 256, 256, 256) == 1
    assert lib.kmh_warp_dice_ok(1, 129, 8, 8, 8) == 0            # channels
    assert lib.kmh_warp_dice_ok(1, 4, 8, 8, 1) == 0              # W < 2
    assert lib.kmh_warp_dice_ok(1024, 128, 4, 4, 4) == 0         # N * C > 65536 rows
    assert lib.kmh_warp_dice_ok(512, 128, 4, 4, 4) == 1
    g = gen(10)
    # C = 130: not served -> the documented composition, with a gradient
    seg_m = torch.rand(1, 130, 4, 4, 4, generator=g).to(DEV)
    seg_f = torch.rand(1, 130, 4, 4, 4, generator=g).to(DEV)
    grid = (O.base_grid((4, 4, 4)).flip(-1)[None] * 0.9).to(DEV).requires_grad_(True)
    assert not kops.warp_dice_ok(seg_m, grid)
    loss = loss_ops.warp_dice_loss(grid, seg_m, seg_f)
    loss.backward()
    ref = loss_ops.DiceLoss()(align_img(grid.detach(), seg_m), seg_f)
    close(loss.detach(), ref, 1e-6)
    assert grid.grad is not None and torch.isfinite(grid.grad).all()
    # N = 100 (> 96: per-sample block count 0 before the fix), C = 100: N * C = 10000 rows, workspace cap 6 blocks per row
    N, C = 100, 100
    seg_m = torch.rand(N, C, 4, 4, 6, generator=g).to(DEV)
    seg_f = torch.rand(N, C, 4, 4, 6, generator=g).to(DEV)
    grid = (O.base_grid((4, 4, 6)).flip(-1)[None] * 0.9).repeat(N, 1, 1, 1, 1).to(DEV).requires_grad_(True)
    assert kops.warp_dice_ok(seg_m, grid)
    fused = loss_ops.warp_dice_loss(grid, seg_m, seg_f)
    fused.backward()
    g2 = grid.detach().clone().requires_grad_(True)
    unfused = loss_ops.DiceLoss()(align_img(g2, seg_m), seg_f)
    unfused.backward()
    close(fused.detach(), unfused.detach(), 1e-6)
    close(grid.grad, g2.grad, 1e-7 + 1e-4 * float(g2.grad.abs().max()))


# ---------------------------------------------------------------- grids
@pytest.mark.parametrize("shape", [(6, 7, 8), (16, 16, 16), (5, 9, 13)])
def test_affine_grid(shape):
    g = gen(7)
    M = torch.eye(3, 4)[None].repeat(2, 1, 1) + 0.2 * torch.randn(2, 3, 4, generator=g)
    cot = torch.randn((2,) + shape + (3,), generator=g)
    Mr = M.clone().requires_grad_(True)
    ref = O.affine_grid(O.square(Mr), shape)
    (ref * cot).sum().backward()
    Mh = M.to(DEV).requires_grad_(True)
    out = ops().affine_grid(Mh, shape)
    (out * cot.to(DEV)).sum().backward()
    close(out, ref, 2e-6)
    close(Mh.grad, Mr.grad, 1e-3, 1e-4)


# (the last two shapes take the row-hoisted evaluators with the weighted sums on the 4x4x1 MFMA: whole 1024-voxel blocks and
# W % 4 == 0 forward, W % 256 == 0 backward; T = 130 leaves idle keypoint lanes in the backward's MFMA blocks)
@pytest.mark.parametrize("T_,shape", [(16, (6, 7, 8)), (64, (12, 10, 9)), (130, (8, 8, 8)), (64, (8, 16, 16)), (130, (2, 4, 256)),
                                      (512, (4, 8, 256))])
def test_tps_grid_given_theta(T_, shape):
    g = gen(8)
    ctrl = torch.rand(2, T_, 3, generator=g) * 1.6 - 0.8
    theta = torch.randn(2, T_ + 4, 3, generator=g) * 0.3
    cot = torch.randn((2,) + shape + (3,), generator=g)
    cr, tr = ctrl.clone().requires_grad_(True), theta.clone().requires_grad_(True)
    flat = O.base_grid(shape).reshape(1, -1, 3).expand(2, -1, -1)
    ref = O.tps_transform_points(tr, cr, flat).reshape(2, *shape, 3).flip(-1)
    (ref * cot).sum().backward()
    ch, th = ctrl.to(DEV).requires_grad_(True), theta.to(DEV).requires_grad_(True)
    out = ops().tps_grid(th, ch, shape)
    (out * cot.to(DEV)).sum().backward()
    scale = float(ref.abs().max())
    close(out, ref, 2e-5 * max(1.0, scale))
    close(th.grad, tr.grad, 2e-4 * float(tr.grad.abs().max()), 1e-4)
    close(ch.grad, cr.grad, 2e-4 * float(cr.grad.abs().max()), 1e-4)


def test_tps_points_fwd_bwd():
    g = gen(9)
    T_, P = 40, 70
    ctrl = torch.rand(2, T_, 3, generator=g) * 1.6 - 0.8
    theta = torch.randn(2, T_ + 4, 3, generator=g) * 0.3
    pts = torch.rand(2, P, 3, generator=g) * 1.8 - 0.9
    cot = torch.randn(2, P, 3, generator=g)
    a = [t.clone().requires_grad_(True) for t in (theta, ctrl, pts)]
    (O.tps_transform_points(*a) * cot).sum().backward()
    b = [t.to(DEV).requires_grad_(True) for t in (theta, ctrl, pts)]
    out = ops().tps_points(*b)
    (out * cot.to(DEV)).sum().backward()
    close(out, O.tps_transform_points(theta, ctrl, pts), 2e-5)
    for x, y in zip(b, a):
        close(x.grad, y.grad, 2e-4 * float(y.grad.abs().max()), 1e-4)


# ---------------------------------------------------------------- fits
@pytest.mark.parametrize("kind", ["affine", "rigid"])
@pytest.mark.parametrize("weighted", [False, True])
def test_matrix_fit_fwd_bwd(kind, weighted):
    g = gen(10)
    K = 50
    pf = torch.rand(2, K, 3, generator=g) * 1.6 - 0.8
    A = torch.eye(3) + 0.2 * torch.randn(3, 3, generator=g)
    pm = pf @ A.T + 0.1 + 0.05 * torch.randn(2, K, 3, generator=g)
    w = torch.rand(2, K, generator=g)
    w = (w / w.sum(1, keepdim=True)) if weighted else None
    cot = torch.randn(2, 3, 4, generator=g)
    fit_o = O.affine_fit if kind == "affine" else O.rigid_fit
    fit_h = ops().affine_fit if kind == "affine" else ops().rigid_fit
    xr, yr = pf.clone().requires_grad_(True), pm.clone().requires_grad_(True)
    wr = None if w is None else w.clone().requires_grad_(True)
    Mr = fit_o(xr, yr, wr)
    (Mr * cot).sum().backward()
    xh, yh = pf.to(DEV).requires_grad_(True), pm.to(DEV).requires_grad_(True)
    wh = None if w is None else w.to(DEV).requires_grad_(True)
    Mh = fit_h(xh, yh, wh)
    (Mh * cot.to(DEV)).sum().backward()
    close(Mh, Mr, 2e-5)
    close(xh.grad, xr.grad, 2e-4 * float(xr.grad.abs().max()), 1e-3)
    close(yh.grad, yr.grad, 2e-4 * float(yr.grad.abs().max()), 1e-3)
    if weighted:      # d/d(weights): training with weight_keypoints (keymorph/model.py:183-191)
        close(wh.grad, wr.grad, 2e-4 * float(wr.grad.abs().max()), 1e-3)


def test_matrix_fit_golden():
    o = golden("ops_small.npz")
    for tag in ("k12", "k64"):
        pf, pm = T(o[f"{tag}_pf"]).to(DEV), T(o[f"{tag}_pm"]).to(DEV)
        w = T(o[f"{tag}_w"]).to(DEV)
        for kind, fit in (("affine", ops().affine_fit), ("rigid", ops().rigid_fit)):
            for wt, sfx in ((None, ""), (w, "_w")):
                inv34 = fit(pf, pm, wt)
                close(inv34, o[f"{tag}_{kind}{sfx}_inv"][:, :3], 2e-5)
                fwd34 = ops().affine_inverse(inv34)
                close(fwd34, o[f"{tag}_{kind}{sfx}_matrix"][:, :3], 2e-5)
                close(ops().affine_grid(inv34, (6, 7, 8)), o[f"{tag}_{kind}{sfx}_grid"], 2e-5)
                close(ops().affine_points(fwd34, pm), o[f"{tag}_{kind}{sfx}_points_a"], 2e-5)


def test_affine_inverse_bwd():
    g = gen(11)
    M = torch.eye(3, 4)[None].repeat(3, 1, 1) + 0.2 * torch.randn(3, 3, 4, generator=g)
    cot = torch.randn(3, 3, 4, generator=g)
    Mr = M.clone().requires_grad_(True)
    (torch.inverse(O.square(Mr))[:, :3] * cot).sum().backward()
    Mh = M.to(DEV).requires_grad_(True)
    out = ops().affine_inverse(Mh)
    (out * cot.to(DEV)).sum().backward()
    close(out, torch.inverse(O.square(M))[:, :3], 1e-5)
    close(Mh.grad, Mr.grad, 1e-4, 1e-4)


@pytest.mark.parametrize("T_,lam", [(12, 0.0), (64, 0.1), (64, 10.0), (200, 1.0), (512, 1.0), (700, 1.0)])
def test_tps_fit_vs_fp64(T_, lam):
    """theta against the fp64 oracle solve of the same system (the HIP path factorises in fp64)."""
    g = gen(12 + T_)
    ctrl = torch.rand(2, T_, 3, generator=g) * 1.6 - 0.8
    tgt = ctrl + 0.05 * torch.randn(2, T_, 3, generator=g)
    lm = torch.full((2,), lam)
    ref = O.tps_fit(ctrl.double(), tgt.double(), lm.double())
    out = ops().tps_fit(ctrl.to(DEV), tgt.to(DEV), lm.to(DEV))
    scale = float(ref.abs().max())
    # fp32 assembly of U (reference-faithful) limits agreement with an all-fp64 assembly
    close(out, ref, 5e-3 * scale if lam == 0.0 else 2e-4 * scale, 1e-3)


@pytest.mark.parametrize("T_,lam", [(16, 0.5), (100, 1.0), (512, 1.0), (512, 0.05)])
def test_tps_fit_bwd(T_, lam):
    """dctrl, dtgt against the fp64 oracle's autograd; T = 512 runs the workgroup-cluster factorisation the headline uses."""
    g = gen(40 + T_)
    ctrl = torch.rand(2, T_, 3, generator=g) * 1.6 - 0.8
    tgt = ctrl + 0.05 * torch.randn(2, T_, 3, generator=g)
    lm = torch.full((2,), lam)
    cot = torch.randn(2, T_ + 4, 3, generator=g)
    cr, tr = ctrl.double().requires_grad_(True), tgt.double().requires_grad_(True)
    (O.tps_fit(cr, tr, lm.double()) * cot.double()).sum().backward()
    ch, th = ctrl.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
    (ops().tps_fit(ch, th, lm.to(DEV)) * cot.to(DEV)).sum().backward()
    close(th.grad, tr.grad.float(), 1e-3 * float(tr.grad.abs().max()), 1e-3)
    close(ch.grad, cr.grad.float(), 1e-3 * float(cr.grad.abs().max()), 1e-3)


def test_tps_fit_cluster_retry_pass_equals_one_workgroup_result():
    """A cluster factorisation that gives up (bounded waits: its workgroups were not all resident) is redone on the
    one-workgroup kernel inside the same call.  The test hook marks every system as "gave up": the retry pass then does all
    the work, and theta / the backward's factors must equal the fp64 solve like the normal path's do (never NaN)."""
    from keymorph_amd import _lib
    g = gen(77)
    T_ = 512
    ctrl = torch.rand(2, T_, 3, generator=g) * 1.6 - 0.8
    tgt = ctrl + 0.05 * torch.randn(2, T_, 3, generator=g)
    lm = torch.full((2,), 1.0)
    cot = torch.randn(2, T_ + 4, 3, generator=g)
    ref = O.tps_fit(ctrl.double(), tgt.double(), lm.double())
    normal = ops().tps_fit(ctrl.to(DEV), tgt.to(DEV), lm.to(DEV))
    prev = _lib.load().kmh_tps_fit_force_retry(1)
    try:
        ch, th = ctrl.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
        retried = ops().tps_fit(ch, th, lm.to(DEV))
        (retried * cot.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        left = _lib.load().kmh_tps_fit_force_retry(0)
    finally:
        _lib.load().kmh_tps_fit_force_retry(prev)
    assert left == 0, "the forced-retry credit was not consumed: the cluster path did not run for T = 512"
    assert torch.isfinite(retried).all()
    scale = float(ref.abs().max())
    close(retried, ref.float(), 2e-4 * scale, 1e-3)
    close(retried, normal, 1e-5 * scale, 1e-5)      # two fp64 factorisations of one matrix (pivot order may differ)
    cr, tr = ctrl.double().requires_grad_(True), tgt.double().requires_grad_(True)
    (O.tps_fit(cr, tr, lm.double()) * cot.double()).sum().backward()
    close(th.grad, tr.grad.float(), 1e-3 * float(tr.grad.abs().max()), 1e-3)
    close(ch.grad, cr.grad.float(), 1e-3 * float(cr.grad.abs().max()), 1e-3)


@pytest.mark.parametrize("T_,lam", [(16, 0.5), (100, 0.05)])
def test_tps_fit_weighted_bwd(T_, lam):
    """weighted TPS (lmbda / (diag_embed(w) + 1e-6), keypoint_aligners.py:298-302): theta and all gradients,
    d/d(weights) included, vs the oracle in fp64."""
    g = gen(60 + T_)
    ctrl = torch.rand(2, T_, 3, generator=g) * 1.6 - 0.8
    tgt = ctrl + 0.05 * torch.randn(2, T_, 3, generator=g)
    w = 0.2 + torch.rand(2, T_, generator=g)
    w = w / w.sum(1, keepdim=True)
    lm = torch.full((2,), lam)
    cot = torch.randn(2, T_ + 4, 3, generator=g)
    cr, tr, wr = (t.double().requires_grad_(True) for t in (ctrl, tgt, w))
    ref = O.tps_fit(cr, tr, lm.double(), wr)
    (ref * cot.double()).sum().backward()
    ch, th, wh = (t.to(DEV).requires_grad_(True) for t in (ctrl, tgt, w))
    out = ops().tps_fit(ch, th, lm.to(DEV), wh)
    (out * cot.to(DEV)).sum().backward()
    close(out, ref.float(), 2e-4 * float(ref.abs().max()), 1e-3)
    for h, r in ((th, tr), (ch, cr), (wh, wr)):
        close(h.grad, r.grad.float(), 1e-3 * float(r.grad.abs().max()), 1e-3)


def test_tps_golden_end_to_end():
    o = golden("ops_small.npz")
    for tag in ("k12", "k64"):
        pf, pm = T(o[f"{tag}_pf"]).to(DEV), T(o[f"{tag}_pm"]).to(DEV)
        for lam in (0.0, 0.1, 10.0):
            lt = str(lam).replace(".", "p")
            lm = torch.full((1,), lam, device=DEV)
            theta = ops().tps_fit(pf, pm, lm)
            grid = ops().tps_grid(theta, pf, (6, 7, 8))
            close(grid, o[f"{tag}_tps{lt}_grid"], 1e-4)
            th2 = ops().tps_fit(pm, pf, lm)
            close(ops().tps_points(th2, pm, pm), o[f"{tag}_tps{lt}_points_a"], 1e-4)


def test_tps_k512_lambda0_vs_truth():
    """SURVEY F7 parity definition: |ours - fp64 truth| vs |reference fp32 - truth| at cond(A) ~ 4.5e5.  Both are single
    draws of the same fp32 rounding noise (an ulp-level change in how U is evaluated moves either by a few per cent:
    3.3e-3 .. 3.7e-3 observed for ours across equivalent formulations, 3.37e-3 for the reference), so the bar is the
    reference's own error with a 25 % band -- not "1e-4", which no fp32 implementation can meet here."""
    o = golden("tps_k512.npz")
    pf, pm = T(o["pf"]), T(o["pm"])
    shape = (10, 12, 14)
    truth = O.tps_grid(pm.double(), pf.double(), torch.zeros(1, dtype=torch.float64), shape).float()
    ref_err = float((T(o["grid_0p0"]) - truth).abs().max())
    theta = ops().tps_fit(pf.to(DEV), pm.to(DEV), torch.zeros(1, device=DEV))
    ours = ops().tps_grid(theta, pf.to(DEV), shape).cpu()
    err = float((ours - truth).abs().max())
    print("tps k512 lambda0: ours", err, "reference", ref_err)
    assert err <= max(1e-4, 1.25 * ref_err), (err, ref_err)
    theta1 = ops().tps_fit(pf.to(DEV), pm.to(DEV), torch.ones(1, device=DEV))
    close(ops().tps_grid(theta1, pf.to(DEV), shape), o["grid_1p0"], 1e-4)


# ---------------------------------------------------------------- center of mass
def test_com3d():
    o = golden("ops_small.npz")
    hm = T(o["com_in"]).to(DEV).requires_grad_(True)
    pts = ops().com3d(hm)
    close(pts, o["com_ij"], 1e-6)
    cot = torch.randn(pts.shape, generator=gen(13))
    (pts * cot.to(DEV)).sum().backward()
    hr = T(o["com_in"]).requires_grad_(True)
    (O.center_of_mass(hr, "ij") * cot).sum().backward()
    close(hm.grad, hr.grad, 1e-7, 1e-4)
    big = torch.randn(1, 3, 40, 33, 47, generator=gen(14))
    close(ops().com3d(big.to(DEV)), O.center_of_mass(big, "ij"), 2e-6)


# ---------------------------------------------------------------- optimizer
def test_fused_adam_matches_torch():
    from keymorph_amd import parallel
    torch.manual_seed(0)
    m1 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(DEV)
    m2 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(DEV)
    m2.load_state_dict(m1.state_dict())
    flat = parallel.FlatParams(m1.parameters())
    opt1 = parallel.FusedAdam(flat, lr=1e-2)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-2)
    x = torch.randn(4, 7, device=DEV)
    for _ in range(5):
        flat.zero_grad()
        m1(x).pow(2).sum().backward()
        opt1.step(flat.allreduce_grads())
        opt2.zero_grad()
        m2(x).pow(2).sum().backward()
        opt2.step()
    for a, b in zip(m1.parameters(), m2.parameters()):
        close(a, b, 1e-6, 1e-5)


# ---------------------------------------------------------------- f-1: affine augmentation (augmentation.py:81-277)
def test_augmentation_golden():
    """HIP augmentation vs the reference's outputs: matrix builder, fixed parameters, and the SAME random draw
    (the parameters come from torch's global CPU generator in the reference's order)."""
    from keymorph_amd import augmentation as A
    a = golden("augment_small.npz")
    img, seg, pts = (T(a[k]).to(DEV) for k in ("img", "seg", "pts"))
    aug = A.AffineDeformation3d(device=DEV)
    params = tuple(T(a[k]) for k in ("params_scale", "params_offset", "params_theta", "params_shear"))
    close(aug.build_affine_matrix(1, params), a["params_matrix"], 1e-6)
    fixed = tuple(float(v) for v in a["fixed_params"])
    i2, s2, p2 = A.affine_augment(img, fixed, seg=seg, points=pts)
    close(i2, a["fixed_img"], 1e-5)
    assert float((s2.cpu() != T(a["fixed_seg"])).float().mean()) <= 1e-3      # nearest: label flips only at exact ties
    close(p2, a["fixed_pts"], 1e-5)
    torch.manual_seed(int(a["rand_seed"][0]))
    i3, s3, p3, m3 = A.random_affine_augment(img, seg=seg, points=pts, max_random_params=(0.2, 0.2, 3.1416, 0.1),
                                             scale_params=0.5, return_affine_matrix=True)
    close(m3, a["rand_matrix"], 1e-6)
    close(i3, a["rand_img"], 1e-5)
    assert float((s3.cpu() != T(a["rand_seg"])).float().mean()) <= 1e-3
    close(p3, a["rand_pts"], 1e-5)
    # image only -> a bare tensor, pair variant shares one transform
    torch.manual_seed(5)
    one = A.random_affine_augment(img)
    torch.manual_seed(5)
    u, v = A.random_affine_augment_pair(img, img)
    assert isinstance(one, torch.Tensor) and one.shape == img.shape
    close(u, one, 0, 0)
    close(u, v, 0, 0)


def test_augmentation_vs_oracle_128():
    """128^3: HIP augmentation == oracle (bilinear image, nearest labels, points); and the blob's centre of mass
    follows deform_points (within the half-voxel rescale the reference's linspace grid + align_corners=False
    implies, cf. test_identity_is_not_identity)."""
    from keymorph_amd import augmentation as A
    S = 128
    lin = torch.linspace(-1, 1, S)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    img = torch.exp(-((zz - 0.1) ** 2 + (yy + 0.15) ** 2 + (xx - 0.05) ** 2) / 0.08)[None, None].contiguous()
    seg = (img * 6).floor()
    pts = torch.rand(1, 32, 3, generator=gen(4)) * 1.6 - 0.8
    params = (torch.tensor([[1.1, 0.95, 1.05]]), torch.tensor([[0.05, -0.08, 0.03]]), torch.tensor([[0.2, -0.1, 0.15]]),
              torch.tensor([[0.02, -0.03, 0.01, 0.02, -0.01, 0.03]]))
    aug = A.AffineDeformation3d(device=DEV)
    M = aug.build_affine_matrix(1, params)
    Mo = O.augment_matrix(*params)
    close(M, Mo, 1e-6)
    ri, rs, rp = O.augment(img, Mo, seg, pts)
    gi = aug.deform_img(img.to(DEV), params)
    gs = aug.deform_img(seg.to(DEV), params, interp_mode="nearest")
    gp = aug.deform_points(pts.to(DEV), params)
    close(gi, ri, 1e-5)
    assert float((gs.cpu() != rs).float().mean()) < 1e-4                  # nearest: flips only at exact rounding ties
    close(gp, rp, 1e-5)
    w = gi[0, 0].cpu() / gi.sum().cpu()
    com = torch.stack([(w * zz).sum(), (w * yy).sum(), (w * xx).sum()])
    centre = torch.tensor([[[0.1, -0.15, 0.05]]], device=DEV)
    close(com, aug.deform_points(centre, params)[0, 0], 2e-2)


# ---------------------------------------------------------------- f-4: Jacobian-determinant eval metrics
def test_jacobian_metrics_golden_and_oracle():
    """loss_ops.{_jacobian_determinant, jdstd, jdlessthan0} on the GPU vs the reference's values (golden) and, at
    96^3 on a TPS-like smooth map, vs the oracle; the permuted channels-last grid view is consumed without a copy."""
    from keymorph_amd import loss_ops as L
    a = golden("augment_small.npz")
    grid = T(a["jd_grid"]).to(DEV)
    gp = grid.permute(0, 4, 1, 2, 3)                       # exactly what pairwise_register_eval.py:337 passes
    close(L._jacobian_determinant(gp), a["jd_det"], 1e-6)
    assert abs(L.jdstd(gp) - float(a["jd_std"][0])) < 1e-7
    assert L.jdlessthan0(gp) == int(a["jd_neg"][0]) and L.jdlessthan0(gp, as_percentage=True) == float(a["jd_neg"][1])
    fold = (gp * T(a["jd_fold_scale"]).to(DEV).reshape(1, 3, 1, 1, 1))
    assert L.jdlessthan0(fold) == int(a["jd_fold_neg"][0])
    assert abs(L.jdstd(fold) - float(a["jd_fold_std"][0])) < 1e-5 * float(a["jd_fold_std"][0])
    assert abs(L.jdlessthan0(fold, as_percentage=True) - int(a["jd_fold_neg"][0]) / a["jd_det"].size) < 1e-12
    # contiguous NCDHW input, bigger volume, vs the oracle
    S = 96
    lin = torch.linspace(-1, 1, S)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    disp = torch.stack([4 * torch.sin(3 * yy) * xx, 5 * torch.cos(2 * zz + xx), 3 * zz * yy + torch.sin(4 * xx)])[None]
    ref = O.jacobian_determinant(disp)
    got = L._jacobian_determinant(disp.to(DEV).contiguous())
    close(got, ref, 1e-5, 1e-5)
    assert abs(L.jdstd(disp.to(DEV)) - float(ref.double().std(unbiased=False))) < 1e-5
    assert abs(L.jdlessthan0(disp.to(DEV)) - int((ref <= 0).sum())) <= 2            # ties at exactly 0 in fp32


# ---------------------------------------------------------------- round-2 advisor findings, pinned
def test_warp_mse_backward_twice_over_a_retained_graph():
    """the fused warp + MSE pass writes d(loss)/d(grid) in the forward; a second backward over the same graph
    (retain_graph) used to crash on empty saved tensors -- it now recomputes through the plain three-launch route"""
    g = gen(41)
    x = torch.rand(1, 1, 9, 10, 12, generator=g).to(DEV)
    f = torch.rand(1, 1, 9, 10, 12, generator=g).to(DEV)
    grid = (torch.rand(1, 9, 10, 12, 3, generator=g) * 2.2 - 1.1).to(DEV).requires_grad_(True)
    loss, _ = ops().warp_mse(x, grid, f)
    (g1,) = torch.autograd.grad(loss, grid, retain_graph=True)
    g1 = g1.clone()
    (g2,) = torch.autograd.grad(loss, grid, grad_outputs=torch.tensor(2.0, device=DEV), retain_graph=True)
    (g3,) = torch.autograd.grad(loss, grid)
    gr = grid.detach().cpu().requires_grad_(True)
    O.mse_loss(f.cpu(), O.align_img(gr, x.cpu())).backward()
    close(g1, gr.grad, 1e-7, 1e-4)
    close(g2, 2 * gr.grad, 2e-7, 1e-4)
    close(g3, gr.grad, 1e-7, 1e-4)


def test_one_hot_many_labels_empty_intersection_and_wrong_device():
    """utils.one_hot / one_hot_subsampled_pair at the edges (keymorph/utils.py:200-240): more than 256 channels (FreeSurfer
    aparc+aseg ids reach 2035) is encoded in slices; label maps without a shared label give the reference's (N, 0, ...)
    tensors instead of an error"""
    from keymorph_amd import utils
    g = gen(42)
    seg = torch.randint(0, 700, (2, 1, 5, 6, 7), generator=g)
    seg[0, 0, 0, 0, 0] = 699
    oh = utils.one_hot(seg)
    ref = O.one_hot(seg)
    assert oh.shape == ref.shape == (2, 700, 5, 6, 7) and oh.dtype == torch.int64
    assert torch.equal(oh.cpu(), ref)
    a = torch.randint(0, 5, (1, 1, 4, 4, 4), generator=g)
    b = torch.randint(10, 15, (1, 1, 4, 4, 4), generator=g)
    ea, eb = utils.one_hot_subsampled_pair(a, b)
    assert ea.shape == eb.shape == (1, 0, 4, 4, 4) and ea.dtype == torch.float32
