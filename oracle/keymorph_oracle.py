"""CPU oracle for the KeyMorph forward-registration hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain torch-CPU restatement of the
reference algorithm (alanqrwang/keymorph @ 2.0.1).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline -- never as the product
path.  The product path is ``keymorph_amd`` (hand-written HIP behind a C ABI)
and fails loudly when its shared library is missing.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors that
``tools/make_golden.py`` produced by importing the real reference from
``/root/reference`` in the build container (fixtures in ``tests/golden``), and
against the analytic known-answer cases of the reference's ``test/test.py``.

Everything is written functionally (weights come in as a ``state_dict``) and is
dtype-generic: run it in ``torch.float64`` to get the "fp64 truth" used for the
ill-conditioned TPS lambda=0 parity definition.

Citations are ``path:line`` under ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# coordinate helpers
# --------------------------------------------------------------------------
def base_grid(shape: Sequence[int], dtype=torch.float32) -> Tensor:
    """ij-ordered identity grid, linspace(-1, 1, n) per axis -> (D, H, W, 3).

    keymorph/utils.py:387-398 (uniform_norm_grid).  Note the align_corners=True
    style spacing that is later sampled with align_corners=False.
    """
    axes = [torch.linspace(-1, 1, int(n)).to(dtype) for n in shape]
    mesh = torch.meshgrid(*axes, indexing="ij")
    return torch.stack(mesh, dim=-1)


def homogeneous(points: Tensor) -> Tensor:
    one = torch.ones(points.shape[:-1] + (1,), dtype=points.dtype)
    return torch.cat([points, one], dim=-1)


# --------------------------------------------------------------------------
# a4  center of mass
# --------------------------------------------------------------------------
def center_of_mass(feat: Tensor, indexing: str = "ij") -> Tensor:
    """keymorph/layers.py:78-134 (3D) and :30-75 (2D).

    feat (n, K, *spatial) -> (n, K, dim) in [-1, 1].
    """
    v = feat.clamp_min(0)
    nd = v.dim() - 2
    out = []
    for ax in range(nd):  # ax 0 = z (or y in 2D) ... last = x
        other = tuple(2 + a for a in range(nd) if a != ax)
        marg = v.sum(dim=other)  # (n, K, size_ax)
        tot = marg.sum(dim=-1, keepdim=True) + 1e-8
        lin = torch.linspace(0, 1, v.shape[2 + ax]).to(v.dtype)
        out.append((lin * marg).sum(dim=-1, keepdim=True) / tot)
    if indexing == "xy":
        out = out[::-1]
    return torch.cat(out, dim=-1) * 2 - 1


# --------------------------------------------------------------------------
# a5 / a6  closed-form matrix fits
# --------------------------------------------------------------------------
def affine_fit(x: Tensor, y: Tensor, w: Optional[Tensor] = None) -> Tensor:
    """argmin_A ||A [x;1] - y||: keymorph/keypoint_aligners.py:76-114.

    x, y (n, K, d); w (n, K) or None  ->  (n, d, d+1).
    """
    X = homogeneous(x).transpose(1, 2)  # (n, d+1, K)
    Y = y.transpose(1, 2)
    if w is None:
        XW = X
    else:
        XW = X * w[:, None, :]
    S = XW @ X.transpose(1, 2)
    Sinv = torch.inverse(S)
    return Y @ (XW.transpose(1, 2) @ Sinv)


def rigid_fit(p1: Tensor, p2: Tensor, w: Optional[Tensor] = None) -> Tensor:
    """Kabsch: keymorph/keypoint_aligners.py:151-213 (incl. the row-scaling
    reflection fix at :199-206, which scales the LAST ROW of V)."""
    a = p1.transpose(1, 2)
    b = p2.transpose(1, 2)
    if w is None:
        ca = a.mean(dim=2, keepdim=True)
        cb = b.mean(dim=2, keepdim=True)
    else:
        ww = w[:, None, :]
        ca = (a * ww).sum(dim=2, keepdim=True)
        cb = (b * ww).sum(dim=2, keepdim=True)
    qa, qb = a - ca, b - cb
    if w is not None:
        qa, qb = qa * ww, qb * ww
    H = qa @ qb.transpose(1, 2)
    U, _, Vt = torch.linalg.svd(H)
    V = Vt.transpose(1, 2)
    R0 = V @ U.transpose(1, 2)
    s = torch.sign(torch.det(R0))  # (n,)
    d = p1.shape[-1]
    rowscale = torch.ones(p1.shape[0], d, 1, dtype=p1.dtype)
    rowscale[:, -1, 0] = s
    V = V * rowscale
    R = V @ U.transpose(1, 2)
    T = cb - R @ ca
    return torch.cat([R, T], dim=-1)


def square(mat: Tensor) -> Tensor:
    """(n, d, d+1) -> (n, d+1, d+1): keymorph/transformations.py:32-35 (the
    reference can only do n=1; batched here = per-sample stacking)."""
    n, d, _ = mat.shape
    out = torch.eye(d + 1, dtype=mat.dtype).repeat(n, 1, 1)
    out[:, :d, :] = mat
    return out


def matrix_transform_points(mat: Tensor, points: Tensor) -> Tensor:
    """points (n, P, d) -> mat[:, :d, :] @ [p;1]: keymorph/transformations.py:81-114."""
    d = points.shape[-1]
    return (mat[:, :d, :] @ homogeneous(points).transpose(1, 2)).transpose(1, 2)


def affine_grid(inv_matrix: Tensor, shape: Sequence[int]) -> Tensor:
    """Dense sampling grid of an affine map: keymorph/transformations.py:37-79.

    inv_matrix (n, d+1, d+1) maps fixed -> moving (ij coords).  Returns
    (n, *shape, d) with the last axis flipped to xyz for grid_sample.
    """
    g = base_grid(shape, inv_matrix.dtype)
    flat = g.reshape(1, -1, g.shape[-1]).expand(inv_matrix.shape[0], -1, -1)
    out = matrix_transform_points(inv_matrix, flat)
    return out.reshape(inv_matrix.shape[0], *shape, g.shape[-1]).flip(-1)


# --------------------------------------------------------------------------
# f-4  Jacobian-determinant eval metrics (keymorph/loss_ops.py:161-247)
# --------------------------------------------------------------------------
def jacobian_determinant(disp: Tensor) -> Tensor:
    """disp (1, 3, D, H, W) -> det(J + I) on the volume cropped by 2, J[a][c] = central difference of component c
    along axis a with zero padding (what scipy.ndimage.correlate(mode='constant') does in the reference)."""
    f = F.pad(disp[0], (1, 1, 1, 1, 1, 1))                  # (3, D+2, H+2, W+2), zeros outside
    D, H, W = disp.shape[2:]
    c = (slice(None), slice(1, D + 1), slice(1, H + 1), slice(1, W + 1))
    gz = 0.5 * f[:, 2:, 1:H + 1, 1:W + 1] - 0.5 * f[:, :D, 1:H + 1, 1:W + 1]
    gy = 0.5 * f[:, 1:D + 1, 2:, 1:W + 1] - 0.5 * f[:, 1:D + 1, :H, 1:W + 1]
    gx = 0.5 * f[:, 1:D + 1, 1:H + 1, 2:] - 0.5 * f[:, 1:D + 1, 1:H + 1, :W]
    J = torch.stack([gz, gy, gx], 0) + torch.eye(3, dtype=disp.dtype).reshape(3, 3, 1, 1, 1)   # [axis][component]
    J = J[:, :, 2:-2, 2:-2, 2:-2]
    return (J[0, 0] * (J[1, 1] * J[2, 2] - J[1, 2] * J[2, 1]) - J[1, 0] * (J[0, 1] * J[2, 2] - J[0, 2] * J[2, 1])
            + J[2, 0] * (J[0, 1] * J[1, 2] - J[0, 2] * J[1, 1]))


# --------------------------------------------------------------------------
# f-1  affine augmentation (the step in front of the path: scripts/train.py:84-98)
# --------------------------------------------------------------------------
def augment_matrix(scale: Tensor, offset: Tensor, theta: Tensor, shear: Tensor) -> Tensor:
    """(b,3),(b,3),(b,3),(b,6) -> (b,4,4) = Mz Ms Mt (R3 R2 R1): keymorph/augmentation.py:85-158."""
    b, dt = scale.shape[0], scale.dtype

    def eye():
        return torch.eye(4, dtype=dt).repeat(b, 1, 1)

    c, s = torch.cos(theta), torch.sin(theta)
    r1, r2, r3, ms, mt, mz = eye(), eye(), eye(), eye(), eye(), eye()
    r1[:, 1, 1], r1[:, 1, 2], r1[:, 2, 1], r1[:, 2, 2] = c[:, 0], -s[:, 0], s[:, 0], c[:, 0]
    r2[:, 0, 0], r2[:, 0, 2], r2[:, 2, 0], r2[:, 2, 2] = c[:, 1], s[:, 1], -s[:, 1], c[:, 1]
    r3[:, 0, 0], r3[:, 0, 1], r3[:, 1, 0], r3[:, 1, 1] = c[:, 2], -s[:, 2], s[:, 2], c[:, 2]
    for k in range(3):
        ms[:, k, k] = scale[:, k]
        mt[:, k, 3] = offset[:, k]
    for k, (i, j) in enumerate(((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1))):
        mz[:, i, j] = shear[:, k]
    return mz @ (ms @ (mt @ (r3 @ (r2 @ r1))))


def augment(img: Tensor, matrix: Tensor, seg: Optional[Tensor] = None, points: Optional[Tensor] = None):
    """Warp img (bilinear), seg (nearest) and points with the forward matrix: keymorph/augmentation.py:148-160
    (deform_img samples through the INVERSE matrix, deform_points applies the forward one)."""
    grid = affine_grid(torch.inverse(matrix), img.shape[2:])
    out = [align_img(grid, img, "bilinear")]
    if seg is not None:
        out.append(align_img(grid, seg, "nearest"))
    if points is not None:
        out.append(matrix_transform_points(matrix, points))
    return out[0] if len(out) == 1 else tuple(out)


# --------------------------------------------------------------------------
# a7 / a8 / a10  thin-plate spline
# --------------------------------------------------------------------------
def tps_dist(a: Tensor, b: Tensor) -> Tensor:
    """keymorph/keypoint_aligners.py:322-334: sqrt(|a_i - b_j|^2 + 1e-6)."""
    diff = a[:, :, None, :] - b[:, None, :, :]
    return torch.sqrt((diff * diff).sum(-1) + 1e-6)


def tps_u(r: Tensor) -> Tensor:
    """keymorph/keypoint_aligners.py:336-339: r^2 log(r + 1e-6)."""
    return r ** 2 * torch.log(r + 1e-6)


def tps_system(ctrl: Tensor, lmbda: Tensor, w: Optional[Tensor] = None) -> Tensor:
    """Assemble A = [[K, P], [P^T, 0]]: keymorph/keypoint_aligners.py:293-318."""
    n, T, d = ctrl.shape
    K = tps_u(tps_dist(ctrl, ctrl))
    lam = lmbda.to(ctrl.dtype).view(n, 1, 1)
    if w is None:
        K = K + torch.eye(T, dtype=ctrl.dtype)[None] * lam
    else:
        # reciprocal of the WHOLE diag-embedded matrix
        K = K + torch.reciprocal(torch.diag_embed(w) + 1e-6) * lam
    P = torch.cat([torch.ones(n, T, 1, dtype=ctrl.dtype), ctrl], dim=-1)
    A = torch.zeros(n, T + d + 1, T + d + 1, dtype=ctrl.dtype)
    A[:, :T, :T] = K
    A[:, :T, T:] = P
    A[:, T:, :T] = P.transpose(1, 2)
    return A


def tps_fit(ctrl: Tensor, tgt: Tensor, lmbda: Tensor, w: Optional[Tensor] = None) -> Tensor:
    """theta (n, T+d+1, d): keymorph/keypoint_aligners.py:276-320, 341-363.

    The reference solves one system per output dim; the matrix is the same, so
    a multi-RHS solve is the same arithmetic up to LAPACK blocking."""
    n, T, d = ctrl.shape
    A = tps_system(ctrl, lmbda, w)
    rhs = torch.zeros(n, T + d + 1, d, dtype=ctrl.dtype)
    rhs[:, :T, :] = tgt
    return torch.linalg.solve(A, rhs)


def tps_transform_points(theta: Tensor, ctrl: Tensor, points: Tensor,
                         chunk: int = 1 << 16) -> Tensor:
    """f(p) = [1,p] theta_aff + sum_t U(|c_t - p|) theta_w[t]:
    keymorph/keypoint_aligners.py:399-433.  Chunked over points only to bound
    memory (results are chunk-independent; reference chunks in 4, :375-388)."""
    n, T, d = ctrl.shape
    wts, aff = theta[:, :T, :], theta[:, T:, :]
    outs = []
    for s in range(0, points.shape[1], chunk):
        p = points[:, s:s + chunk]
        U = tps_u(tps_dist(ctrl, p))  # (n, T, P)
        one_p = torch.cat([torch.ones(p.shape[:-1] + (1,), dtype=p.dtype), p], -1)  # [1, p]
        outs.append(one_p @ aff + U.transpose(1, 2) @ wts)
    return torch.cat(outs, dim=1)


def tps_grid(points_m: Tensor, points_f: Tensor, lmbda: Tensor, shape: Sequence[int],
             w: Optional[Tensor] = None, chunk: int = 1 << 15) -> Tensor:
    """TPS.get_flow_field: keymorph/keypoint_aligners.py:365-397.  ctrl =
    points_f, target = points_m (inverse map), result flipped to xyz."""
    theta = tps_fit(points_f, points_m, lmbda, w)
    g = base_grid(shape, points_f.dtype)
    n = points_f.shape[0]
    flat = g.reshape(1, -1, g.shape[-1]).expand(n, -1, -1)
    out = tps_transform_points(theta, points_f, flat, chunk=chunk)
    return out.reshape(n, *shape, g.shape[-1]).flip(-1)


# --------------------------------------------------------------------------
# a11-a13  warp + losses
# --------------------------------------------------------------------------
def align_img(grid: Tensor, x: Tensor, mode: str = "bilinear") -> Tensor:
    """keymorph/utils.py:14-21."""
    return F.grid_sample(x, grid=grid, mode=mode, padding_mode="border", align_corners=False)


def grid_sample_3d_manual(x: Tensor, grid: Tensor) -> Tensor:
    """Index-level restatement of ATen grid_sampler_3d (bilinear, border,
    align_corners=False) used to pin the arithmetic the HIP sampler follows
    (SURVEY Appendix B).  x (n,C,D,H,W), grid (n,Do,Ho,Wo,3) xyz."""
    n, C, D, H, W = x.shape
    gx, gy, gz = grid[..., 0], grid[..., 1], grid[..., 2]
    ix = (((gx + 1) * W - 1) / 2).clamp(0, W - 1)
    iy = (((gy + 1) * H - 1) / 2).clamp(0, H - 1)
    iz = (((gz + 1) * D - 1) / 2).clamp(0, D - 1)
    x0, y0, z0 = ix.floor(), iy.floor(), iz.floor()
    fx, fy, fz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    out = torch.zeros(n, C, *grid.shape[1:4], dtype=x.dtype)
    flat = x.reshape(n, C, -1)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                zz, yy, xx = z0 + dz, y0 + dy, x0 + dx
                wgt = ((fz if dz else 1 - fz) * (fy if dy else 1 - fy) * (fx if dx else 1 - fx))
                ok = (zz < D) & (yy < H) & (xx < W)
                idx = (zz.clamp_max(D - 1) * H + yy.clamp_max(H - 1)) * W + xx.clamp_max(W - 1)
                val = torch.gather(flat, 2, idx.reshape(n, 1, -1).expand(-1, C, -1))
                out += (val.reshape(out.shape) * (wgt * ok)[:, None])
    return out


def mse_loss(pred: Tensor, target: Tensor) -> Tensor:
    """keymorph/loss_ops.py:9-13."""
    return ((pred - target) ** 2).mean()


def dice_loss(pred: Tensor, target: Tensor, hard: bool = False, return_regions: bool = False,
              ign_first_ch: bool = False) -> Tensor:
    """keymorph/loss_ops.py:16-63 (eps = 1 in numerator AND denominator)."""
    n, c = target.shape[:2]
    t = target.reshape(n, c, -1)
    p = pred.reshape(n, c, -1)
    if hard:
        am = p.argmax(dim=1, keepdim=True)
        p = torch.zeros_like(p).scatter(1, am, 1.0)
    if ign_first_ch:
        t, p = t[:, 1:], p[:, 1:]
    num = (2 * t * p).sum(2) + 1
    den = (p * p).sum(2) + (t * t).sum(2) + 1
    loss = 1 - num / den
    return loss.mean(0) if return_regions else loss.mean()


# --------------------------------------------------------------------------
# a2 / a3  backbones (functional, weights from a reference-keyed state_dict)
# --------------------------------------------------------------------------
def _gn_groups(channels: int, num_groups: int) -> int:
    # keymorph/unet3d/buildingblocks.py:66-68
    return 1 if channels < num_groups else num_groups


def single_conv_gcr(sd: Dict[str, Tensor], prefix: str, x: Tensor, num_groups: int,
                    taps: Optional[List[Tensor]] = None) -> Tensor:
    """GroupNorm -> Conv3d(k3,p1,no bias) -> ReLU: keymorph/unet3d/buildingblocks.py:10-93.
    ``taps``: if given, the pre-ReLU tensor is appended (tests use it to locate ReLU-kink voxels)."""
    g = _gn_groups(x.shape[1], num_groups)
    x = F.group_norm(x, g, sd[prefix + "groupnorm.weight"], sd[prefix + "groupnorm.bias"], 1e-5)
    x = F.conv3d(x, sd[prefix + "conv.weight"], None, padding=1)
    if taps is not None:
        taps.append(x.detach())
    return F.relu(x)


def double_conv(sd, prefix, x, num_groups, taps=None):
    x = single_conv_gcr(sd, prefix + "SingleConv1.", x, num_groups, taps)
    return single_conv_gcr(sd, prefix + "SingleConv2.", x, num_groups, taps)


def unet3d_forward(sd: Dict[str, Tensor], x: Tensor, num_levels: int = 4,
                   num_truncated: int = 0, num_groups: int = 8, taps: Optional[List[Tensor]] = None) -> Tensor:
    """UNet3D / TruncatedUNet3D forward ("gcr" DoubleConv, max-pool encoders,
    nearest-upsample + concat decoders, 1x1x1 final conv; logits returned):
    keymorph/unet3d/model.py:117-151, 307-391; buildingblocks.py:321-475,568-582."""
    feats: List[Tensor] = []
    for i in range(num_levels):
        if i > 0:
            x = F.max_pool3d(x, 2)
        x = double_conv(sd, f"encoders.{i}.basic_module.", x, num_groups, taps)
        feats.insert(0, x)
    feats = feats[1:]
    n_dec = num_levels - 1 - num_truncated
    for j in range(n_dec):
        skip = feats[j]
        x = F.interpolate(x, size=skip.shape[2:], mode="nearest")
        x = torch.cat([skip, x], dim=1)
        x = double_conv(sd, f"decoders.{j}.basic_module.", x, num_groups, taps)
    return F.conv3d(x, sd["final_conv.weight"], sd["final_conv.bias"])


CONVNET_POOL_AFTER = (2, 4, 6, 8)


def convnet_forward(sd: Dict[str, Tensor], x: Tensor, norm_type: str = "instance") -> Tensor:
    """ConvNet: 9 x [Conv3d(k3,p1,bias) -> norm -> ReLU (-> MaxPool 2)]:
    keymorph/net.py:7-36, keymorph/layers.py:137-187."""
    for b in range(1, 10):
        x = F.conv3d(x, sd[f"block{b}.conv.weight"], sd[f"block{b}.conv.bias"], padding=1)
        if norm_type == "instance":
            x = F.instance_norm(x, eps=1e-5)
        elif norm_type == "group":
            x = F.group_norm(x, 8, sd[f"block{b}.norm.weight"], sd[f"block{b}.norm.bias"], 1e-5)
        elif norm_type != "none":
            raise NotImplementedError(norm_type)
        x = F.relu(x)
        if b in CONVNET_POOL_AFTER:
            x = F.max_pool3d(x, 2)
    return x


# --------------------------------------------------------------------------
# a1  the registration step
# --------------------------------------------------------------------------
def keypoint_weights(feat_f: Tensor, feat_m: Tensor, mode: str, scales: Optional[Tensor] = None,
                     biases: Optional[Tensor] = None) -> Tensor:
    """keymorph/model.py:75-109 on materialised heat-maps (n, K, D, H, W): 'power' or 'variance' weights."""
    f, m = F.relu(feat_f), F.relu(feat_m)
    if mode == "power":
        w = f.flatten(2).sum(-1) * m.flatten(2).sum(-1)
    else:
        w = 1 / (scales * torch.var(f, dim=(2, 3, 4)) + biases) * (1 / (scales * torch.var(m, dim=(2, 3, 4)) + biases))
    return w / w.sum(dim=1, keepdim=True)


def parse_transform(t: str) -> Tuple[str, Optional[float]]:
    if t in ("affine", "rigid"):
        return t, None
    assert t.startswith("tps_"), t
    return "tps", float(t[4:])


def register(points_f: Tensor, points_m: Tensor, transform_type: str, shape: Sequence[int],
             return_aligned_points: bool = False, w: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """Keypoints -> grid (+matrix / aligned points), per sample:
    keymorph/model.py:198-288.  Batched input = per-sample bs=1 results
    concatenated."""
    kind, lam = parse_transform(transform_type)
    res: Dict[str, Tensor] = {"points_f": points_f, "points_m": points_m}
    n = points_f.shape[0]
    if kind in ("affine", "rigid"):
        fit = affine_fit if kind == "affine" else rigid_fit
        inv = square(fit(points_f, points_m, w))  # fixed -> moving
        fwd = torch.inverse(inv)
        res["matrix"] = fwd
        res["grid"] = affine_grid(inv, shape)
        if return_aligned_points:
            res["points_a"] = matrix_transform_points(fwd, points_m)
    else:
        lm = torch.full((n,), lam, dtype=points_f.dtype)
        res["grid"] = tps_grid(points_m, points_f, lm, shape, w)
        if return_aligned_points:
            th = tps_fit(points_m, points_f, lm, w)
            res["points_a"] = tps_transform_points(th, points_m, points_m)
    return res


# --------------------------------------------------------------------------
# a16  real-world-coordinate alignment (bs = 1)
# --------------------------------------------------------------------------
def norm2real(points: Tensor, affine: Tensor, sizes: Tensor) -> Tensor:
    """[-1,1] -> voxel ((p+1)*S/2 - 1/2) -> world (affine @ [v;1]): keymorph/utils.py:243-259, 275-291, 320-335."""
    vox = (points + 1) * sizes.to(points.dtype) / 2 - 0.5
    return matrix_transform_points(affine.to(points.dtype), vox)


def real2norm(points: Tensor, affine: Tensor, sizes: Tensor) -> Tensor:
    """world -> voxel (affine^-1) -> [-1,1] (2*(v+1/2)/S - 1): keymorph/utils.py:262-272, 294-317, 338-354."""
    vox = matrix_transform_points(torch.inverse(affine.to(points.dtype)), points)
    return 2 * (vox + 0.5) / sizes.to(points.dtype) - 1


def register_real_world(points_f: Tensor, points_m: Tensor, transform_type: str, shape: Sequence[int],
                        aff_f: Tensor, aff_m: Tensor, shape_f: Tensor, shape_m: Tensor,
                        w: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """align_in_real_world_coords=True: fit on the world-space keypoints, and move every query point
    norm -> world (its own image's affine) -> fitted map -> world -> norm (the other image's affine):
    keymorph/keypoint_aligners.py:47-66, 116-148 (matrix aligners), 255-268, 431-465 (TPS)."""
    kind, lam = parse_transform(transform_type)
    rf, rm = norm2real(points_f, aff_f, shape_f), norm2real(points_m, aff_m, shape_m)
    g = base_grid(shape, points_f.dtype)
    flat = norm2real(g.reshape(1, -1, 3), aff_f, shape_f)
    res: Dict[str, Tensor] = {}
    if kind in ("affine", "rigid"):
        fit = affine_fit if kind == "affine" else rigid_fit
        inv = square(fit(rf, rm, w))
        fwd = torch.inverse(inv)
        res["matrix"] = fwd
        moved = matrix_transform_points(inv, flat)
        res["points_a"] = real2norm(matrix_transform_points(fwd, rm), aff_f, shape_f)
    else:
        lm = torch.full((1,), lam, dtype=points_f.dtype)
        moved = tps_transform_points(tps_fit(rf, rm, lm, w), rf, flat)
        res["points_a"] = real2norm(tps_transform_points(tps_fit(rm, rf, lm, w), rm, rm), aff_f, shape_f)
    res["grid"] = real2norm(moved, aff_m, shape_m).reshape(1, *shape, 3).flip(-1)
    return res


# --------------------------------------------------------------------------
# a14  one-hot encodings
# --------------------------------------------------------------------------
def one_hot(seg: Tensor) -> Tensor:
    """(N,1,D,H,W) integer labels -> (N,C,D,H,W), C = max label + 1: keymorph/utils.py:200-205."""
    return F.one_hot(seg)[:, 0].permute(0, 4, 1, 2, 3)


def one_hot_subsampled_pair(seg1: Tensor, seg2: Tensor, subsample_num: int = 14):
    """keymorph/utils.py:208-240: channels = (np.random.choice of) the labels both maps contain, in that order."""
    import numpy as np
    shared = np.intersect1d(np.unique(seg1.numpy()), np.unique(seg2.numpy()), assume_unique=True)
    chosen = np.random.choice(shared, subsample_num, replace=False) if len(shared) > subsample_num else shared

    def enc(seg):
        return torch.stack([(seg[:, 0] == int(v)).float() for v in chosen], dim=1)
    return enc(seg1), enc(seg2)


def keymorph_forward(backbone, img_f: Tensor, img_m: Tensor, transform_type: str,
                     return_aligned_points: bool = False) -> Dict[str, Tensor]:
    """KeyMorph.forward for one transform type: keymorph/model.py:142-289.
    ``backbone`` is a callable img -> heat-map logits."""
    pf = center_of_mass(backbone(img_f), "ij")
    pm = center_of_mass(backbone(img_m), "ij")
    return register(pf, pm, transform_type, img_f.shape[2:], return_aligned_points)


def groupwise_points(group_points: Tensor, transform_type: str, num_iters: int
                     ) -> Tuple[Tensor, Tensor]:
    """Iterative mean-keypoint alignment: keymorph/model.py:331-444.  Returns
    (aligned points after num_iters, mean_points used by the LAST iteration)."""
    kind, lam = parse_transform(transform_type)
    cur = group_points.clone()
    mean = cur.mean(dim=0, keepdim=True)
    for _ in range(num_iters):
        mean = cur.mean(dim=0, keepdim=True)
        nxt = torch.zeros_like(cur)
        for i in range(cur.shape[0]):
            pm = cur[i:i + 1]
            if kind == "tps":
                lm = torch.full((1,), lam, dtype=pm.dtype)
                th = tps_fit(pm, mean, lm)
                nxt[i:i + 1] = tps_transform_points(th, pm, pm)
            else:
                fit = affine_fit if kind == "affine" else rigid_fit
                fwd = torch.inverse(square(fit(mean, pm)))
                nxt[i:i + 1] = matrix_transform_points(fwd, pm)
        cur = nxt
    return cur, mean


def groupwise_grid(points_m: Tensor, mean_points: Tensor, transform_type: str,
                   shape: Sequence[int]) -> Tensor:
    """Final per-subject grid: keymorph/model.py:453-510."""
    return register(mean_points, points_m, transform_type, shape)["grid"]
