"""GPU: size-independent properties at BASELINE.json's full configuration (256^3, 512 keypoints,
TruncatedUNet3D f_maps=32) -- the oracle cannot run these sizes in seconds, so parity here is through
invariants the domain offers.  One model / one pair is shared by the whole module."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
SIZE, K = 256, 512


def close(a, b, atol, rtol=0):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.fixture(scope="module")
def world():
    from keymorph_amd import synthetic
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    torch.manual_seed(23)
    net = TruncatedUNet3D(1, K, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    km = KeyMorph(net, K, 3, max_train_keypoints=None).to(DEV).eval()
    img_f, img_m = synthetic.make_pair(SIZE, 3, torch.device(DEV))
    with torch.no_grad():
        pts = km.get_keypoints(torch.cat([img_f, img_m]))
    return dict(km=km, img_f=img_f, img_m=img_m, pts_f=pts[:1], pts_m=pts[1:])


def test_keypoints_in_range_and_batch_consistent(world):
    km, f, m = world["km"], world["img_f"], world["img_m"]
    assert world["pts_f"].shape == (1, K, 3)
    assert float(world["pts_f"].abs().max()) <= 1.0 and float(world["pts_m"].abs().max()) <= 1.0
    with torch.no_grad():
        solo = km.get_keypoints(f)
    close(solo, world["pts_f"], 2e-6)          # [f; m] batch == f alone (per-sample norms)


def test_conv_modes_agree_at_full_size(world):
    """fp32-MFMA, split-bf16 (6 products) and the default range-scaled split-fp16 (3 products) convolutions give the
    same 512 keypoints at 256^3 (every layer, the first-layer kernels and the fused head included)."""
    from keymorph_amd import backbone_ops as B
    km, f = world["km"], world["img_f"]
    prev = B.CONV_MODE
    pts = {}
    try:
        for mode in ("f32", "bf16x6", "f16x3"):
            B.set_conv_mode(mode)
            with torch.no_grad():
                pts[mode] = km.get_keypoints(f)
    finally:
        B.set_conv_mode(prev)
    close(pts["bf16x6"], pts["f32"], 1e-5)
    close(pts["f16x3"], pts["f32"], 1e-5)


def test_fused_head_equals_heatmap_path(world):
    """fused 1x1x1 head + ReLU + CoM == materialised (1,512,128^3) heat-map + CenterOfMass3d."""
    from keymorph_amd.layers import CenterOfMass3d
    km, f = world["km"], world["img_f"]
    with torch.no_grad():
        heat = km.backbone(f)
        assert heat.shape == (1, K, SIZE // 2, SIZE // 2, SIZE // 2)
        pts = CenterOfMass3d("ij")(heat)
    close(pts, world["pts_f"], 5e-6)


def test_self_registration_is_identity(world):
    """img_m == img_f  =>  identical keypoints  =>  affine/rigid matrix = I and every grid = the base grid."""
    from keymorph_amd import ops
    km, f = world["km"], world["img_f"]
    eye34 = torch.eye(3, 4, device=DEV)[None]
    base = ops.affine_grid(eye34, (SIZE, SIZE, SIZE))
    with torch.no_grad():
        res = km(f, f, transform_type=["affine", "rigid", "tps_0", "tps_1"], return_aligned_points=True)
    for tt in ("affine", "rigid"):
        close(res[tt]["matrix"], torch.eye(4)[None], 2e-5)
    for tt, tol in (("affine", 5e-5), ("rigid", 5e-5), ("tps_0", 2e-4), ("tps_1", 1e-4)):
        close(res[tt]["grid"][0, ::37, ::41, ::43], base[0, ::37, ::41, ::43], tol)
        close(res[tt]["points_a"], res[tt]["points_m"], tol)


def test_tps_interpolates_keypoints(world):
    """lambda = 0 thin-plate spline maps every control point onto its target at K = 512.

    Uses well-spread keypoints (what a trained extractor produces).  The random-init network of this module
    collapses its 512 keypoints into a cluster (min distance ~1e-3): the lambda = 0 weights then reach 1e3..1e4
    and NO fp32 evaluation of sum_j w_j U_ij (the reference's bmm included) interpolates to 1e-4 -- that case is
    only checked for boundedness."""
    from keymorph_amd.keypoint_aligners import TPS
    g = torch.Generator().manual_seed(11)
    pf = (torch.rand(1, K, 3, generator=g) * 1.6 - 0.8).to(DEV)
    pm = pf + 0.05 * torch.randn(1, K, 3, generator=g).to(DEV)
    tps = TPS(points_m=pm, points_f=pf, lmbda=torch.zeros(1, device=DEV), dim=3)
    # cond(A) ~ 4.5e5 here: 3e-4 is the fp32 evaluation floor, the reference's own is 3.8e-4
    close(tps.get_inverse_transformed_points(pf), pm, 3e-4)     # grid direction: fixed -> moving
    close(tps.get_forward_transformed_points(pm), pf, 3e-4)     # forward fit: moving -> fixed
    grid = tps.get_flow_field((1, 1, SIZE, SIZE, SIZE))
    assert grid.shape == (1, SIZE, SIZE, SIZE, 3) and bool(torch.isfinite(grid).all())
    # clustered keypoints of the random-init net: finite, and accurate to the fp32 floor of this conditioning
    tc = TPS(points_m=world["pts_m"], points_f=world["pts_f"], lmbda=torch.zeros(1, device=DEV), dim=3)
    err = float((tc.get_inverse_transformed_points(world["pts_f"]) - world["pts_m"]).abs().max())
    assert err < 5e-3, err


def test_warp_linearity_and_fusion(world):
    from keymorph_amd import ops
    from keymorph_amd.keypoint_aligners import AffineKeypointAligner
    f, m = world["img_f"], world["img_m"]
    grid = AffineKeypointAligner(points_m=world["pts_m"], points_f=world["pts_f"], dim=3).get_flow_field(f.shape)
    w1, w2 = ops.grid_sample3d(f, grid), ops.grid_sample3d(m, grid)
    close(ops.grid_sample3d(0.3 * f - 1.7 * m, grid), 0.3 * w1 - 1.7 * w2, 2e-6)
    loss, warped = ops.warp_mse(m, grid, f)
    close(warped, w2, 0)
    close(loss, ops.mse_loss(f, w2), 1e-7)
    assert float(w2.min()) >= float(m.min()) - 1e-6 and float(w2.max()) <= float(m.max()) + 1e-6  # convex blend


def test_affine_inverse_consistency(world):
    from keymorph_amd.keypoint_aligners import AffineKeypointAligner, RigidKeypointAligner
    pf, pm = world["pts_f"], world["pts_m"]
    for cls in (AffineKeypointAligner, RigidKeypointAligner):
        a = cls(points_m=pm, points_f=pf, dim=3)
        close(torch.bmm(a.transform_matrix, a.inverse_transform_matrix), torch.eye(4)[None], 2e-5)
        rt = a.get_inverse_transformed_points(a.get_forward_transformed_points(pm))
        close(rt, pm, 2e-5)
    # Kabsch is symmetric under swapping the clouds (R21 = R12^T) even for noisy points; a noisy affine
    # least-squares fit is not (LS(f->m)^-1 != LS(m->f)), so that half of the reference's exact-motion test
    # (test/test.py test_rigid_0_forward_inverse) applies to the rigid aligner only.
    a = RigidKeypointAligner(points_m=pm, points_f=pf, dim=3)
    b = RigidKeypointAligner(points_m=pf, points_f=pm, dim=3)
    close(a.transform_matrix, b.inverse_transform_matrix, 2e-5)
    R = a.transform_matrix[0, :3, :3]
    close(R @ R.T, torch.eye(3), 2e-5)
    assert abs(float(torch.det(R.cpu())) - 1) < 1e-4


def test_gradients_agree_across_arithmetic_modes_at_full_size(world):
    """one full forward + backward (256^3, 512 kp, TPS 0.1, MSE) in the default f16x3 arithmetic and in bf16x6: same
    loss, and parameter gradients equal in relative L2 (a few ReLU-kink flips among 10^9 activations are the only
    legitimate difference, DESIGN.md "gradient parity")."""
    from keymorph_amd import backbone_ops as B, ops
    km = world["km"].train()
    prev = B.CONV_MODE
    out = {}
    try:
        for mode in ("bf16x6", "f16x3"):
            B.set_conv_mode(mode)
            km.zero_grad(set_to_none=True)
            r = km(world["img_f"], world["img_m"], transform_type="tps_0.1", return_aligned_points=False)["tps_0.1"]
            loss, _ = ops.warp_mse(world["img_m"], r["grid"], world["img_f"])
            loss.backward()
            out[mode] = (float(loss.detach()), torch.cat([p.grad.reshape(-1).double() for p in km.parameters()]))
    finally:
        B.set_conv_mode(prev)
        km.zero_grad(set_to_none=True)
        km.eval()
    (l6, g6), (l3, g3) = out["bf16x6"], out["f16x3"]
    assert abs(l3 - l6) <= 1e-6 * max(1.0, abs(l6)), (l3, l6)
    rel = float((g3 - g6).norm() / g6.norm())
    assert bool(torch.isfinite(g3).all()) and rel < 2e-2, rel


def test_training_step_decreases_loss(world):
    """three full fwd+bwd+Adam steps at 256^3 / 512 kp / TPS: finite gradients, loss goes down."""
    from keymorph_amd import ops, parallel
    from keymorph_amd.model import KeyMorph
    km = world["km"].train()
    flat = parallel.FlatParams(km.parameters())
    opt = parallel.FusedAdam(flat, lr=1e-4)
    losses = []
    for _ in range(3):
        flat.zero_grad()
        r = km(world["img_f"], world["img_m"], transform_type="tps_0", return_aligned_points=False)["tps_0"]
        loss, _ = ops.warp_mse(world["img_m"], r["grid"], world["img_f"])
        loss.backward()
        assert bool(torch.isfinite(flat.grad).all()) and float(flat.grad.abs().max()) > 0
        opt.step(flat.allreduce_grads())
        losses.append(float(loss.detach()))
    km.eval()
    assert losses[-1] < losses[0], losses


def test_two_pairs_per_gpu_equal_two_single_pairs(world):
    """BASELINE configs[1] runs 2 pairs per GPU: one pass over the batch of 2 pairs gives each pair's keypoints, grid and
    loss, and the SUM of the two single-pair parameter gradients (norms are per sample; nothing couples the pairs).
    The gradient is taken through a fixed linear read-out of the keypoints: through the keypoint fit the 512 clustered
    keypoints of this random-init network amplify 1e-7 keypoint differences by the fit's condition number
    (test_tps_interpolates_keypoints), which says nothing about the kernels.  Grid / loss go through the affine aligner
    for the same reason; the tps_0 leg checks finiteness."""
    from keymorph_amd import ops, synthetic
    km = world["km"].train()
    f2, m2 = synthetic.make_pair(SIZE, 5, torch.device(DEV))
    F, M = torch.cat([world["img_f"], f2]), torch.cat([world["img_m"], m2])
    g = torch.Generator().manual_seed(3)
    cot = torch.randn(2, 2, K, 3, generator=g).to(DEV)            # (pair, fixed / moving, keypoint, axis)

    def backbone_grads(pairs):
        km.zero_grad(set_to_none=True)
        imgs = torch.cat([F[pairs], M[pairs]])                   # [fixed...; moving...] as KeyMorph.forward batches them
        pts = km.get_keypoints(imgs)
        c = torch.cat([cot[pairs, 0], cot[pairs, 1]])
        (pts * c).sum().backward()
        return pts.detach().clone(), torch.cat([p.grad.reshape(-1) for p in km.parameters()]).clone()

    def register(f, m, tt):
        with torch.no_grad():
            r = km(f, m, transform_type=tt, return_aligned_points=False)[tt]
            loss, _ = ops.warp_mse(m, r["grid"], f)
        return float(loss), r["grid"]

    try:
        pb, gb = backbone_grads([0, 1])
        singles = [backbone_grads([i]) for i in range(2)]
        km.zero_grad(set_to_none=True)
        lb, grid_b = register(F, M, "affine")
        ls = [register(F[i:i + 1], M[i:i + 1], "affine") for i in range(2)]
        lt, grid_t = register(F, M, "tps_0")
    finally:
        km.zero_grad(set_to_none=True)
        km.eval()
    for i, (pi, gi) in enumerate(singles):
        # (not bit-equal: the f16x3 split takes ONE range scale per tensor, i.e. over the whole batch, so a sample's rounding
        # depends on its batch mates; observed 1.5e-6 .. 2.2e-6 over the weights the earlier tests of this file leave behind)
        close(pb[[i, 2 + i]], pi, 4e-6)
        close(grid_b[i:i + 1], ls[i][1], 2e-4)
    gs = singles[0][1] + singles[1][1]
    rel = float((gb - gs).norm() / gs.norm())
    assert bool(torch.isfinite(gb).all()) and float(gb.abs().max()) > 0 and rel < 1e-3, rel
    lm = 0.5 * (ls[0][0] + ls[1][0])
    assert abs(lb - lm) <= 1e-5 * max(1.0, abs(lm)), (lb, lm)
    assert np.isfinite(lt) and bool(torch.isfinite(grid_t).all())


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: groupwise registration of 8 subjects x 256^3, 512 keypoints, TPS
# (keymorph/model.py:295-530, scripts/groupwise_register_eval.py:379-431), on one GPU, evaluation mode, no_grad.
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_grid_samples(pts_m, mean, tt, idx):
    """the oracle's final grid of one subject at the lattice points idx = (iz, iy, ix) index tensors (fp32 CPU path):
    keymorph/model.py:453-510 -> register(points_f = mean, points_m = subject)"""
    from oracle import keymorph_oracle as O
    kind, lam = O.parse_transform(tt)
    g = O.base_grid((SIZE, SIZE, SIZE))[idx[0], idx[1], idx[2]].reshape(1, -1, 3)
    if kind == "tps":
        lm = torch.full((1,), lam)
        out = O.tps_transform_points(O.tps_fit(mean, pts_m, lm), mean, g)
    else:
        inv = O.square((O.affine_fit if kind == "affine" else O.rigid_fit)(mean, pts_m))
        out = O.matrix_transform_points(inv, g)
    return out.flip(-1)[0]


def test_groupwise_8x256(world, tmp_path):
    from keymorph_amd import synthetic
    from keymorph_amd.transformations import AffineTransform
    from keymorph_amd.utils import align_img
    from oracle import keymorph_oracle as O
    km = world["km"].eval()
    dev = torch.device(DEV)
    n_sub, iters = 8, 5
    base = synthetic.blob_volume(SIZE, 3, dev)
    with torch.no_grad():
        subs = [base] + [align_img(AffineTransform(matrix=synthetic.random_affine_matrix(40 + i, dev, scale=0.1, shift=0.1,
                                                                                          rot=0.2, shear=0.05),
                                                   dim=3).get_flow_field(base.shape), base) for i in range(1, n_sub)]
        stack = torch.cat(subs).contiguous()
        del subs
        types = ["affine", "tps_1", "tps_0"]
        res = km.groupwise_register(stack, transform_type=types, device=dev, num_iters=iters, save_results_to_disk=True,
                                    save_dir=str(tmp_path), log_to_console=False)
        solo = torch.cat([km.get_keypoints(stack[i:i + 1]) for i in (0, 3, 7)])
    pts = res["affine"]["grouppoints_m"]
    assert pts.shape == (n_sub, K, 3)
    print(f"groupwise 8x256: stack keypoints vs solo {float((pts[[0, 3, 7]] - solo).abs().max()):.2e}")
    close(pts[[0, 3, 7]], solo, 2e-6)                 # the stack's keypoints are each subject's own
    g = torch.Generator().manual_seed(5)
    idx = tuple(torch.randint(0, SIZE, (4096,), generator=g) for _ in range(3))
    for tt in types:
        r = res[tt]
        assert r["grouppoints_a"].shape == (n_sub, K, 3) and bool(torch.isfinite(r["grouppoints_a"]).all())
        files = sorted(f for f in os.listdir(tmp_path) if f.startswith(f"{tt}_grid_"))
        assert files == [f"{tt}_grid_{i:03}.npy" for i in range(n_sub)], files
        # the iterations against the oracle's (fp32 CPU) on OUR keypoints: model.py:331-444
        if tt != "tps_0":       # lambda = 0 on the clumped keypoints of a random-init net: conditioning, not arithmetic
            pa, mean = O.groupwise_points(pts.cpu(), tt, iters)
            print(f"groupwise 8x256 {tt}: aligned points vs oracle {float((r['grouppoints_a'].cpu() - pa).abs().max()):.2e}")
            close(r["grouppoints_a"], pa, 1e-4)
            for i in (0, 5):
                grid = np.load(os.path.join(tmp_path, files[i]), mmap_mode="r")
                assert grid.shape == (1, SIZE, SIZE, SIZE, 3) and grid.dtype == np.float32
                ours = torch.from_numpy(np.ascontiguousarray(grid[0][idx[0].numpy(), idx[1].numpy(), idx[2].numpy()]))
                want = _oracle_grid_samples(pts[i:i + 1].cpu(), mean, tt, idx)
                print(f"groupwise 8x256 {tt} subject {i}: grid samples vs oracle {float((ours - want).abs().max()):.2e}")
                close(ours, want, 1e-4)
        else:
            grid = np.load(os.path.join(tmp_path, files[2]), mmap_mode="r")
            assert grid.shape == (1, SIZE, SIZE, SIZE, 3) and bool(np.isfinite(grid[0, ::8, ::8, ::8]).all())
    # tps_0 against the oracle where a comparison means something: ONE iteration (the mean is then the plain mean of the
    # keypoints, the same numbers on both sides; five chaotic lambda = 0 iterations amplify 1e-7 into anything).  The statement is
    # SURVEY F7's: |ours - fp64| <= max(1e-4, 1.25 |reference fp32 - fp64|), for the aligned points of the iteration
    # (model.py:331-444: subject -> mean) and for the final map's grid (model.py:453-510: mean -> subject) at 4096 lattice points.
    with torch.no_grad():
        one = km.groupwise_register(stack, transform_type=["tps_0"], device=dev, num_iters=1, save_results_to_disk=False)["tps_0"]
    p32 = pts.cpu()
    mean32 = p32.mean(dim=0, keepdim=True)
    p64, mean64 = p32.double(), p32.double().mean(dim=0, keepdim=True)
    z32, z64 = torch.zeros(1), torch.zeros(1, dtype=torch.float64)
    g32 = O.base_grid((SIZE, SIZE, SIZE))[idx[0], idx[1], idx[2]].reshape(1, -1, 3)
    worst = {"points": (0.0, 0.0), "grid": (0.0, 0.0)}
    for i in range(n_sub):
        a32 = O.tps_transform_points(O.tps_fit(p32[i:i + 1], mean32, z32), p32[i:i + 1], p32[i:i + 1])[0].double()
        a64 = O.tps_transform_points(O.tps_fit(p64[i:i + 1], mean64, z64), p64[i:i + 1], p64[i:i + 1])[0]
        ours = one["grouppoints_a"][i].cpu().double()
        e_ours, e_ref = float((ours - a64).abs().max()), float((a32 - a64).abs().max())
        worst["points"] = max(worst["points"], (e_ours, e_ref))
        assert e_ours <= max(1e-4, 1.25 * e_ref), ("tps_0 aligned points", i, e_ours, e_ref)
        if i in (0, 5):
            w32 = O.tps_transform_points(O.tps_fit(mean32, p32[i:i + 1], z32), mean32, g32).flip(-1)[0].double()
            w64 = O.tps_transform_points(O.tps_fit(mean64, p64[i:i + 1], z64), mean64, g32.double()).flip(-1)[0]
            og = one["groupgrids"][i][idx[0].to(dev), idx[1].to(dev), idx[2].to(dev)].cpu().double()
            e_ours, e_ref = float((og - w64).abs().max()), float((w32 - w64).abs().max())
            worst["grid"] = max(worst["grid"], (e_ours, e_ref))
            assert e_ours <= max(1e-4, 1.25 * e_ref), ("tps_0 final grid", i, e_ours, e_ref)
    print(f"groupwise 8x256 tps_0, one iteration: |ours - fp64| / |reference fp32 - fp64|: aligned points "
          f"{worst['points'][0]:.2e} / {worst['points'][1]:.2e}, grid samples {worst['grid'][0]:.2e} / {worst['grid'][1]:.2e}")
    # a group of identical subjects: identical keypoints = their own mean, every map the identity
    from keymorph_amd import ops
    with torch.no_grad():
        same = km.groupwise_register(base.expand(3, -1, -1, -1, -1).contiguous(), transform_type=["affine", "tps_1"],
                                     device=dev, num_iters=2, save_results_to_disk=False)
    eye = ops.affine_grid(torch.eye(3, 4, device=DEV)[None], (SIZE, SIZE, SIZE))
    for tt, tol in (("affine", 5e-5), ("tps_1", 1e-4)):
        close(same[tt]["grouppoints_a"], same[tt]["grouppoints_m"], tol)
        close(same[tt]["groupgrids"][1, ::37, ::41, ::43], eye[0, ::37, ::41, ::43], tol)


def test_eval_path_at_full_size_uses_the_fused_decoder_operator(world, monkeypatch):
    """scripts/register.py / pairwise_register_eval.py:116-171: model.eval(), no_grad, a list of transform types,
    return_aligned_points=True.  Under no_grad the decoder must take the same fused upsample+concat+conv operator as
    the training forward (it was gated on torch.is_grad_enabled() before round 3) and give the same keypoints as the
    plain upsample + concat + 27-tap route."""
    from keymorph_amd import backbone_ops as B
    km = world["km"].eval()
    before = B.UPCONV_STATS["calls"]
    with torch.no_grad():
        res = km(world["img_f"], world["img_m"], transform_type=["rigid", "affine", "tps_10", "tps_0"],
                 return_aligned_points=True)
    assert B.UPCONV_STATS["calls"] == before + 2, "the fused decoder operator did not run under no_grad"
    for tt in ("rigid", "affine", "tps_10", "tps_0"):
        r = res[tt]
        assert r["grid"].shape == (1, SIZE, SIZE, SIZE, 3) and bool(torch.isfinite(r["grid"]).all())
        assert r["points_a"].shape == (1, K, 3)
    monkeypatch.setenv("KEYMORPH_NO_UPCONV", "1")
    with torch.no_grad():
        plain = km.get_keypoints(world["img_f"])
    assert B.UPCONV_STATS["calls"] == before + 2
    # (two routes to the same numbers: different convolutions -> different epilogue statistics groupings -> GroupNorm
    # coefficients that differ in the last bits; observed 2e-6 .. 6.4e-6 over the weights the earlier tests leave behind)
    close(plain, res["affine"]["points_f"], 1e-5)      # (the module's earlier training test moved the weights: compare
    close(res["tps_0"]["points_f"], res["rigid"]["points_f"], 0)       # within this test, not with world["pts_f"])


def _fresh_model():
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    torch.manual_seed(23)
    net = TruncatedUNet3D(1, K, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    return KeyMorph(net, K, 3, max_train_keypoints=None).to(DEV).train()


def _step_grads(km, f, m, tt, mode):
    from keymorph_amd import backbone_ops as B, ops
    prev = B.CONV_MODE
    try:
        B.set_conv_mode(mode)
        km.zero_grad(set_to_none=True)
        r = km(f, m, transform_type=tt, return_aligned_points=False)[tt]
        loss, _ = ops.warp_mse(m, r["grid"], f)
        loss.backward()
        return float(loss.detach()), {k: p.grad.detach().double().clone() for k, p in km.named_parameters()}
    finally:
        B.set_conv_mode(prev)
        km.zero_grad(set_to_none=True)


def _compare_grads(ga, gb):
    per = {k: float((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-300)) for k in gb}
    va, vb = torch.cat([ga[k].reshape(-1) for k in gb]), torch.cat([gb[k].reshape(-1) for k in gb])
    return float((va - vb).norm() / vb.norm()), per


@pytest.mark.parametrize("tt", ["affine", "tps_1"])
def test_default_arithmetic_vs_exact_fp32_mfma_at_full_size(world, tt):
    """BASELINE configs[1] (256^3, 512 keypoints, affine, bs = 1) and the same with tps_1, forward + backward: the default
    split-fp16 arithmetic (f16x3) against the EXACT fp32 matrix-core kernels (v_mfma_f32_32x32x2_f32, no operand
    splitting) on a fresh initialisation -- same loss to 1e-6, the whole parameter-gradient vector to 2e-3 relative L2
    (two fp32-class arithmetics differ by ~1e-7 in keypoints; the fit of this random-init network's clumped keypoints
    amplifies that 100-300x), with the per-tensor table printed."""
    km = _fresh_model()
    f, m = world["img_f"], world["img_m"]
    l32, g32 = _step_grads(km, f, m, tt, "f32")
    l16, g16 = _step_grads(km, f, m, tt, "f16x3")
    whole, per = _compare_grads(g16, g32)
    print(f"\n256^3 {tt}: loss f16x3 {l16:.9f} vs f32 {l32:.9f}; whole gradient vector rel-L2 {whole:.2e}")
    for k, e in sorted(per.items(), key=lambda kv: -kv[1]):
        print(f"    {e:9.2e}  {k}")
    assert abs(l16 - l32) <= 1e-6 * max(1.0, abs(l32)), (l16, l32)
    assert whole <= 2e-3, whole


def test_bs2_tps0_forward_backward_vs_exact_fp32_mfma(world):
    """BASELINE configs[2] exactly as bench.py runs it (256^3, 512 keypoints, TPS lambda = 0, bs = 2, fwd + bwd), asserted:
    finite, batch loss = mean of the two single-pair losses, and f16x3 against the exact fp32-MFMA kernels.  lambda = 0
    on 512 clumped keypoints is a conditioning statement (cond(A) ~ 1e6 and worse): the reference's own fp32
    path is 3.8e-4 from the fp64 truth ON THE GRID there, so the loss is held to 1e-4 and the gradient to 3e-2."""
    from keymorph_amd import synthetic
    km = _fresh_model()
    f2, m2 = synthetic.make_pair(SIZE, 5, torch.device(DEV))
    F, M = torch.cat([world["img_f"], f2]), torch.cat([world["img_m"], m2])
    l16, g16 = _step_grads(km, F, M, "tps_0", "f16x3")
    l32, g32 = _step_grads(km, F, M, "tps_0", "f32")
    singles = [_step_grads(km, F[i:i + 1], M[i:i + 1], "tps_0", "f16x3")[0] for i in range(2)]
    whole, per = _compare_grads(g16, g32)
    top = sorted(per.items(), key=lambda kv: -kv[1])[:3]
    print(f"\n256^3 tps_0 bs=2: loss f16x3 {l16:.9f} vs f32 {l32:.9f} (singles {singles}); gradient rel-L2 {whole:.2e}; "
          f"worst tensors {top}")
    assert all(bool(torch.isfinite(v).all()) for v in g16.values()) and np.isfinite(l16)
    assert abs(l16 - 0.5 * (singles[0] + singles[1])) <= 1e-4 * max(1.0, abs(l16))     # measured 1.2e-5 (conditioning)
    assert abs(l16 - l32) <= 1e-4 * max(1.0, abs(l32)), (l16, l32)
    assert whole <= 3e-2, whole


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the ORACLE at the metric's size.  Everything above compares the HIP path with itself (properties, arithmetic
# modes); this runs oracle/keymorph_oracle.py's forward and autograd backward ONCE at 256^3 / 512 keypoints on the host
# (~2 min, ~50 GB of RAM) and holds the HIP path to it: keymorph/model.py:142-289, utils.py:14-21, loss_ops.py:9-13.
# ---------------------------------------------------------------------------------------------------------------------
def _host_ram_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return float(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def test_fullsize_vs_oracle_256_affine():
    """BASELINE configs[1] (256^3, 512 keypoints, affine, bs = 1), forward + backward, HIP vs the oracle on the same pair
    and the same seeded weights.  Forward: keypoints / matrix / grid / warped volume <= 1e-4 (north_star's bar), MSE <= 1e-6.
    Backward, against the oracle's AUTOGRAD, in the two pieces of the chain rule (tests/oracle_at_size.py):
      * the BACKBONE's backward (24 TFLOP) on an identical cotangent d(loss)/d(keypoints): whole gradient vector <= 2e-3
        relative L2, every tensor <= 1e-2 (two fp32 implementations; pre-activations at rounding distance from zero are
        masked differently by each -- the per-tensor table is printed);
      * the TAIL (keypoints -> affine fit -> grid -> warp -> MSE), whose gradient is 3072 numbers obtained by pushing a sum
        over 16.8 M voxels through the inverse of a fit to clumped keypoints: both fp32 tails are measured against the fp64
        tail, and ours must be no further from it than 1.25 x the oracle's own fp32 error (or 1e-4);
      * end to end the difference is then bounded by what the two tails differ by: <= 2e-3 + 2 (tail_hip + tail_oracle).
    Plus a forward-only tps_1 leg: the oracle's TPS fit on the ORACLE's keypoints at 4096 sampled voxels vs the HIP grid."""
    from tests.oracle_at_size import compare_with_hip, hip_model, oracle_pair
    from oracle import keymorph_oracle as O
    if _host_ram_gib() < 96:
        pytest.skip("needs ~50 GB of host RAM for the oracle's 256^3 autograd graph")
    ref = oracle_pair(SIZE, K, threads=32, tt="affine", seed=100, sd_seed=23)
    par = compare_with_hip(ref, DEV)
    per, per_bb = par.pop("gradient_per_tensor"), par.pop("backbone_gradient_per_tensor")
    print(f"256^3 affine vs oracle ({ref['seconds']:.0f} s on the host): " +
          ", ".join(f"{k} {v:.2e}" for k, v in par.items() if isinstance(v, float)))
    for k, v in sorted(per_bb.items(), key=lambda kv: -kv[1])[:8]:
        print(f"   backbone-only grad {k:60s} rel-L2 {v:.2e}   (end to end {per[k]:.2e})")
    assert par["keypoints"] <= 1e-4 and par["matrix"] <= 1e-4 and par["grid"] <= 1e-4 and par["warped"] <= 1e-4, par
    assert par["mse"] <= 1e-6, par
    # (round 5, after the fp64 comparison at 128^3 -- test_backbone_backward_vs_fp64_oracle_128 -- said who is off where: the
    # whole vector and every tensor outside the first encoder block are held to half of round 4's bars; the first block's
    # cancelling sums -- 6.7e-3 on its one-element GroupNorm weight here, the 22-bit operands' doing -- keep 1e-2)
    assert par["backbone_gradient_rel_l2"] <= 1e-3, par
    assert max(v for k, v in per_bb.items() if not k.startswith("encoders.0.")) <= 3e-3, per_bb
    assert max(per_bb.values()) <= 1e-2, per_bb
    assert par["tail_rel_l2_hip"] <= max(1e-4, 1.25 * par["tail_rel_l2_oracle"]), par
    assert par["gradient_rel_l2"] <= 2e-3 + 2 * (par["tail_rel_l2_hip"] + par["tail_rel_l2_oracle"]), par

    # forward-only legs from the same oracle run (tests/oracle_at_size.py::extra_legs): TPS grids at 4096 sampled voxels on
    # the ORACLE's keypoints, and the 14-class Dice (north_star: "matching reference Dice within 1e-4")
    assert par["tps_1_grid_vs_oracle_fp32"] <= 1e-4 and par["tps_1_e2e_grid_vs_oracle_fp32"] <= 1e-4, par
    # lambda = 0, 512 clumped keypoints = the metric's own configuration, ill-conditioned: no further from the
    # fp64 truth than 1.25 x the reference arithmetic's own fp32 error (or 1e-4)
    assert par["tps_0_grid_vs_fp64"] <= max(1e-4, 1.25 * par["tps_0_oracle_fp32_vs_fp64"]), par
    assert par["dice_fused"] <= 1e-4 and par["dice_unfused"] <= 1e-4, par


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: who is right when the HIP backward and the oracle's fp32 autograd disagree?  The backbone's backward against an
# fp64 run of the oracle at 128^3 / 512 keypoints (tests/oracle_at_size.py::oracle_backbone_fp64).
# ---------------------------------------------------------------------------------------------------------------------
def test_backbone_backward_vs_fp64_oracle_128():
    """image -> TruncatedUNet3D(f_maps 32, 4 levels, 1 truncated) -> center of mass (keymorph/model.py:111-117), one seeded
    cotangent: every parameter-gradient tensor of the HIP path against the oracle's FP64 autograd (truth), next to the
    oracle's own fp32 autograd (the reference arithmetic).  Bar per tensor: |hip - fp64| <= max(1e-3, 1.25 |oracle fp32 -
    fp64|) relative L2; the whole vector <= 5e-4 (measured: hip 3.3e-4, oracle fp32 3.8e-4 -- closer than the reference
    arithmetic overall).  The exception, measured and bounded separately: the FIRST encoder block's tensors, sums over 2 M
    voxels that cancel to ~1e-4 of their terms.  There the HIP path is the one further from the truth -- first GroupNorm
    (bias, weight) 4.9e-3 / 4.6e-3 where the reference arithmetic is 1.0e-3 / 1.4e-3 -- and the cause is the split-operand
    arithmetic's 22 significant bits: tools/diag_backbone_fp64.py gives 2.0e-3 / 2.1e-3 with bf16x6 (24 bits) and
    3.9e-4 / 5.7e-4 with the exact fp32 MFMA, unchanged by the statistics fold, the lazy first layer or the fused pooling.
    Bar for those: 8e-3."""
    from tests.oracle_at_size import hip_model, oracle_backbone_fp64
    if _host_ram_gib() < 48:
        pytest.skip("needs ~12 GB of host RAM for the fp64 autograd graph at 128^3")
    S, Kk = 128, 512
    ref = oracle_backbone_fp64(S, Kk)
    km = hip_model(ref["sd"], Kk, DEV)
    pts = km.get_keypoints(ref["x"].to(DEV))
    assert float((pts.detach().cpu().double() - ref["pts_fp64"]).abs().max()) <= 1e-5
    torch.autograd.backward([pts], [ref["cot"].to(DEV)])
    g64, g32 = ref["grads_fp64"], ref["grads_fp32"]
    rows, num_h, num_o, den = [], 0.0, 0.0, 0.0
    for k, p in km.backbone.named_parameters():
        t = g64[k].double()
        eh = float((p.grad.detach().cpu().double() - t).norm() / (t.norm() + 1e-300))
        eo = float((g32[k].double() - t).norm() / (t.norm() + 1e-300))
        rows.append((k, eh, eo))
        num_h += float((p.grad.detach().cpu().double() - t).pow(2).sum())
        num_o += float((g32[k].double() - t).pow(2).sum())
        den += float(t.pow(2).sum())
    whole_h, whole_o = (num_h / den) ** 0.5, (num_o / den) ** 0.5
    print(f"backbone backward vs fp64 at 128^3 / 512 kp: whole vector hip {whole_h:.2e}, oracle fp32 {whole_o:.2e}")
    for k, eh, eo in sorted(rows, key=lambda r: -max(r[1], r[2]))[:8]:
        print(f"   {k:62s} hip {eh:.2e}   oracle fp32 {eo:.2e}   closer: {'hip' if eh <= eo else 'oracle'}")
    first_block = lambda k: k.startswith("encoders.0.")      # noqa: E731
    worse = [(k, eh, eo) for k, eh, eo in rows if eh > (8e-3 if first_block(k) else max(1e-3, 1.25 * eo))]
    assert not worse, worse
    assert whole_h <= 5e-4, (whole_h, whole_o)


def test_first_block_dgrad_selector_vs_fp64_oracle_128():
    """Round 6: `KEYMORPH_FIRST_BLOCK_DGRAD=bf16x6` / backbone_ops.set_first_block_dgrad runs the
    first encoder block's 32 -> 16 data gradient -- the launch behind the first GroupNorm's cancelling sums -- with three bf16
    terms (24 bits at every magnitude) while everything else stays f16x3.  Same comparison as above: the selector must have
    taken the launch, every tensor outside the first block is unchanged bit for bit (only that launch differs), the first
    block's tensors stay inside the default's bar, and the table (default / selector / reference fp32, against fp64) is
    printed: DESIGN section 4 quotes it as the price / benefit of the option."""
    from keymorph_amd import backbone_ops as B
    from tests.oracle_at_size import hip_model, oracle_backbone_fp64
    if _host_ram_gib() < 48:
        pytest.skip("needs ~12 GB of host RAM for the fp64 autograd graph at 128^3")
    S, Kk = 128, 512
    ref = oracle_backbone_fp64(S, Kk)
    g64, g32 = ref["grads_fp64"], ref["grads_fp32"]
    got = {}
    try:
        for mode in ("", "bf16x6"):
            B.set_first_block_dgrad(mode)
            n0 = B.FIRST_BLOCK_STATS["exact_dgrads"]
            km = hip_model(ref["sd"], Kk, DEV)
            pts = km.get_keypoints(ref["x"].to(DEV))
            torch.autograd.backward([pts], [ref["cot"].to(DEV)])
            assert (B.FIRST_BLOCK_STATS["exact_dgrads"] - n0) == (1 if mode else 0)
            got[mode] = {k: p.grad.detach().cpu() for k, p in km.backbone.named_parameters()}
            del km
    finally:
        B.set_first_block_dgrad("")
    err = lambda g, k: float((g.double() - g64[k].double()).norm() / (g64[k].double().norm() + 1e-300))      # noqa: E731
    print("first-block data-gradient arithmetic vs fp64 at 128^3 / 512 kp (relative L2):   f16x3   bf16x6-selector   reference fp32")
    for k in got[""]:
        if k.startswith("encoders.0."):
            print(f"   {k:58s} {err(got[''][k], k):.2e}   {err(got['bf16x6'][k], k):.2e}   {err(g32[k], k):.2e}")
    for k in got[""]:
        # the selector changes ONE launch: tensors whose gradient does not pass through it are bit-identical
        if not k.startswith(("encoders.0.basic_module.SingleConv1.", "encoders.0.basic_module.SingleConv2.groupnorm.")):
            assert torch.equal(got[""][k], got["bf16x6"][k]), k
        assert err(got["bf16x6"][k], k) <= (8e-3 if k.startswith("encoders.0.") else max(1e-3, 1.25 * err(g32[k], k))), k


def test_tps_training_step_vs_oracle_autograd_64_k512():
    """ONE training step with a TPS transform and the production kernel variants that do not need a large volume -- K = 512
    keypoints, so the workgroup-cluster LU and the row-hoisted T = 512 grid evaluators run, forward AND backward -- against
    the oracle's autograd of the same step (scripts/train.py:129-176; keymorph/keypoint_aligners.py:276-433 chunked under
    torch.utils.checkpoint): 64^3 pair, tps_1.  Loss <= 1e-6, keypoints / sampled grid <= 1e-4, whole parameter-gradient
    vector <= 3e-3 relative L2 (a random-init backbone clumps its keypoints: the TPS tail amplifies fp32 rounding of the
    keypoints 100-300x)."""
    from keymorph_amd import ops
    from tests.oracle_at_size import hip_model, oracle_tps_step
    S, Kk, lam = 64, 512, 1.0
    ref = oracle_tps_step(S, Kk, lam)
    km = hip_model(ref["sd"], Kk, DEV)
    f, m = ref["img_f"].to(DEV), ref["img_m"].to(DEV)
    tt = "tps_1"
    r = km(f, m, transform_type=tt, return_aligned_points=False)[tt]
    loss, _ = ops.warp_mse(m, r["grid"], f)
    loss.backward()
    i0, i1, i2 = (t.to(DEV) for t in ref["idx"])
    e_pts = max(float((r["points_f"].detach().cpu() - ref["points_f"]).abs().max()),
                float((r["points_m"].detach().cpu() - ref["points_m"]).abs().max()))
    e_grid = float((r["grid"].detach()[0][i0, i1, i2].cpu() - ref["grid_samples"]).abs().max())
    e_loss = abs(float(loss.detach()) - ref["loss"])
    num = den = 0.0
    per = {}
    for k, p in km.backbone.named_parameters():
        a, b = p.grad.detach().cpu().double(), ref["grads"][k].double()
        per[k] = float((a - b).norm() / (b.norm() + 1e-300))
        num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
    whole = (num / den) ** 0.5
    print(f"tps_1 training step 64^3 / 512 kp vs oracle autograd: keypoints {e_pts:.2e} grid {e_grid:.2e} loss {e_loss:.2e} "
          f"gradient {whole:.2e}; worst tensors {sorted(per.items(), key=lambda kv: -kv[1])[:3]}")
    assert e_pts <= 1e-4 and e_grid <= 1e-4 and e_loss <= 1e-6, (e_pts, e_grid, e_loss)
    assert whole <= 3e-3, (whole, sorted(per.items(), key=lambda kv: -kv[1])[:5])      # measured 1.0e-3


def test_use_amp_one_product_backbone_vs_oracle_128():
    """use_amp=True (keymorph/model.py:176-191: the reference autocasts the keypoint extractor to fp16): the one-product fp16
    arithmetic of the 27-tap / weight-gradient / decoder kernels -- fp16 inputs, fp32 accumulation, fp32 tensors.  At 128^3 /
    512 keypoints against the fp32 oracle: keypoints <= 2e-3, every gradient finite, whole gradient vector <= 5e-2 relative
    L2; it is a different arithmetic from the default (results differ).  The setting is per call: a model with the OPPOSITE
    setting run between the forward and the backward changes nothing (the gradient bars below hold for both), and nothing of
    it is left when get_keypoints() returns."""
    from keymorph_amd import backbone_ops as B
    from tests.oracle_at_size import hip_model, oracle_backbone_fp64
    if _host_ram_gib() < 48:
        pytest.skip("needs ~12 GB of host RAM for the oracle's autograd graph at 128^3")
    S, Kk = 128, 512
    ref = oracle_backbone_fp64(S, Kk)
    out = {}
    other = hip_model(ref["sd"], Kk, DEV)                      # a second model with the OPPOSITE setting, called between the
    try:                                                       # forward and the backward of the one under test
        for amp in (True, False):
            km = hip_model(ref["sd"], Kk, DEV)
            km.use_amp = amp
            pts = km.get_keypoints(ref["x"].to(DEV))
            assert B.amp_enabled() is False                    # the setting ended with get_keypoints()
            other.use_amp = not amp
            with torch.no_grad():
                other.get_keypoints(ref["x"].to(DEV))
            torch.autograd.backward([pts], [ref["cot"].to(DEV)])
            num = den = 0.0
            for k, p in km.backbone.named_parameters():
                assert torch.isfinite(p.grad).all(), k
                a, b = p.grad.detach().cpu().double(), ref["grads_fp32"][k].double()
                num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
            out[amp] = (float((pts.detach().cpu() - ref["pts_fp32"]).abs().max()), (num / den) ** 0.5, pts.detach().clone())
            del km
    finally:
        del other
    print(f"use_amp at 128^3 / 512 kp vs the fp32 oracle: keypoints {out[True][0]:.2e} (default {out[False][0]:.2e}), "
          f"gradient {out[True][1]:.2e} (default {out[False][1]:.2e})")
    assert out[True][0] <= 2e-3 and out[True][1] <= 5e-2, out[True][:2]
    assert out[False][0] <= 1e-5 and out[False][1] <= 2e-3, out[False][:2]
    assert not torch.equal(out[True][2], out[False][2]) and out[True][0] > 4 * out[False][0], "the one-product kernels did not run"
