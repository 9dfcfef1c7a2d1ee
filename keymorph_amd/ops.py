"""torch.autograd.Function wrappers over the C ABI (include/keymorph_hip.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and autograd
bookkeeping.  Every FLOP of the hot path runs in libkeymorph_hip.so; there is no
eager/PyTorch fallback -- a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import os

from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------
def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _prep(t: Tensor, name: str = "tensor") -> Tensor:
    if not t.is_cuda:
        raise _lib.KeymorphHipError(
            f"{name} is on {t.device}: keymorph_amd ops run only on an AMD GPU (no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        # the C ABI launches on the CURRENT device's stream and never calls hipSetDevice: a tensor that lives on
        # another GPU would be touched by kernels queued on the wrong device
        raise _lib.KeymorphHipError(
            f"{name} is on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
            "call torch.cuda.set_device(...) (one process per GPU) before using keymorph_amd ops")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_WS = {}


def workspace(nbytes: int, device: torch.device, slot: str = "reduce") -> Tensor:
    """Stream-ordered scratch (grown on demand, reused by consecutive launches)."""
    key = (device.index, slot)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _reduce_ws(device) -> Tensor:
    return workspace(int(_lib.load().kmh_reduce_ws_bytes()), device, "reduce")


# --------------------------------------------------------------------------
# a11  sampler
# --------------------------------------------------------------------------
class _GridSample3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grid, mode):
        lib = _lib.load()
        x, grid = _prep(x, "x"), _prep(grid, "grid")
        N, C, D, H, W = x.shape
        n2, Do, Ho, Wo, three = grid.shape
        assert n2 == N and three == 3, "grid must be (N, Do, Ho, Wo, 3)"
        out = torch.empty((N, C, Do, Ho, Wo), dtype=torch.float32, device=x.device)
        if _lib.profiler.enabled:  # grid 12 B + C*(volume 4 B + out 4 B) per output voxel
            _lib.profiler.meta = {"bytes": float(N * Do * Ho * Wo) * (12 + 8 * C)}
        check(lib.kmh_grid_sample3d_fwd(_p(x), _p(grid), _p(out), N, C, D, H, W, Do, Ho, Wo, mode, _stream()),
              "kmh_grid_sample3d_fwd")
        ctx.save_for_backward(x, grid)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        x, grid = ctx.saved_tensors
        gout = _prep(gout)
        N, C, D, H, W = x.shape
        _, Do, Ho, Wo, _ = grid.shape
        dx = dgrid = None
        if ctx.needs_input_grad[1]:
            if ctx.mode != 0:
                dgrid = torch.zeros_like(grid)
            else:
                dgrid = torch.empty_like(grid)
                if _lib.profiler.enabled:  # gout 4C + grid 12 + volume 4C + dgrid 12 B per voxel
                    _lib.profiler.meta = {"bytes": float(N * Do * Ho * Wo) * (24 + 8 * C)}
                check(lib.kmh_grid_sample3d_bwd_grid(_p(x), _p(grid), _p(gout), _p(dgrid), N, C, D, H, W, Do, Ho,
                                                     Wo, _stream()), "kmh_grid_sample3d_bwd_grid")
        if ctx.needs_input_grad[0]:
            if ctx.mode != 0:
                raise NotImplementedError("gradient wrt the volume is implemented for bilinear mode only")
            dx = torch.zeros_like(x)
            check(lib.kmh_grid_sample3d_bwd_input(_p(grid), _p(gout), _p(dx), N, C, D, H, W, Do, Ho, Wo,
                                                  _stream()), "kmh_grid_sample3d_bwd_input")
        return dx, dgrid, None


def grid_sample3d(x: Tensor, grid: Tensor, mode: str = "bilinear") -> Tensor:
    return _GridSample3d.apply(x, grid, {"bilinear": 0, "nearest": 1}[mode])


class _WarpMSE(torch.autograd.Function):
    """Fused align_img + MSELoss (one pass; the warped volume is still returned).  When the grid needs a gradient the
    same pass also writes d(loss)/d(grid) -- the MSE cotangent is known inside the warp -- so the backward is a no-op
    for the default cotangent 1 (a device-side check) instead of an MSE-backward pass plus a grid-backward pass."""

    @staticmethod
    def forward(ctx, x, grid, fixed):
        lib = _lib.load()
        x, grid, fixed = _prep(x), _prep(grid), _prep(fixed)
        N, C, D, H, W = x.shape
        _, Do, Ho, Wo, _ = grid.shape
        out = torch.empty((N, C, Do, Ho, Wo), dtype=torch.float32, device=x.device)
        assert fixed.shape == out.shape
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        ctx.dgrid = None
        if ctx.needs_input_grad[1] and (grid.numel() & 3) == 0:
            dgrid = torch.empty_like(grid)
            if _lib.profiler.enabled:  # grid 12 + dgrid 12 + C * (volume 4 + fixed 4 + out 4) B per output voxel
                _lib.profiler.meta = {"bytes": float(N * Do * Ho * Wo) * (24 + 12 * C)}
            rc = lib.kmh_warp_mse_fwd_grad(_p(x), _p(grid), _p(fixed), _p(out), _p(loss), _p(dgrid), N, C, D, H, W, Do,
                                           Ho, Wo, _p(_reduce_ws(x.device)), _stream())
            if rc == 0:
                ctx.dgrid = dgrid
            elif rc != -22:
                check(rc, "kmh_warp_mse_fwd_grad")
        if ctx.dgrid is None:
            if _lib.profiler.enabled:  # grid 12 B + C*(volume 4 + fixed 4 + out 4) B per output voxel
                _lib.profiler.meta = {"bytes": float(N * Do * Ho * Wo) * (12 + 12 * C)}
            check(lib.kmh_warp_mse_fwd(_p(x), _p(grid), _p(fixed), _p(out), _p(loss), N, C, D, H, W, Do, Ho, Wo,
                                       _p(_reduce_ws(x.device)), _stream()), "kmh_warp_mse_fwd")
        # always saved (inputs and the returned output: no extra memory): the plain three-launch backward serves a
        # grid that needed the non-fused route and every backward after the first one
        ctx.save_for_backward(x, grid, fixed, out)
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, gloss, _gout):
        lib = _lib.load()
        if ctx.dgrid is not None:
            # d(loss)/d(grid) was written by the forward pass: the first backward hands that buffer out, scaled in place
            # by the cotangent (a device-side check makes the usual cotangent 1 a no-op: no pass over 200 MB, no host
            # synchronisation).  A second backward over a retained graph recomputes it from the saved tensors below.
            dgrid, ctx.dgrid = ctx.dgrid, None
            gl = _prep(gloss).reshape(1)
            check(lib.kmh_scale_unless_one(_p(dgrid), dgrid.numel(), _p(gl), _stream()), "kmh_scale_unless_one")
            return None, dgrid, None
        x, grid, fixed, out = ctx.saved_tensors
        N, C, D, H, W = x.shape
        _, Do, Ho, Wo, _ = grid.shape
        gout = torch.empty_like(out)
        gl = _prep(gloss).reshape(1)
        check(lib.kmh_mse_bwd(_p(out), _p(fixed), _p(gl), out.numel(), _p(gout), _stream()), "kmh_mse_bwd")
        dgrid = torch.empty_like(grid)
        if _lib.profiler.enabled:
            _lib.profiler.meta = {"bytes": float(N * Do * Ho * Wo) * (24 + 8 * C)}
        check(lib.kmh_grid_sample3d_bwd_grid(_p(x), _p(grid), _p(gout), _p(dgrid), N, C, D, H, W, Do, Ho, Wo,
                                             _stream()), "kmh_grid_sample3d_bwd_grid")
        return None, dgrid, None


def warp_mse(x: Tensor, grid: Tensor, fixed: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (mse(fixed, warp(x, grid)), warp(x, grid)); gradient flows to ``grid`` only."""
    return _WarpMSE.apply(x, grid, fixed)


# --------------------------------------------------------------------------
# a12 / a13  losses
# --------------------------------------------------------------------------
class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        a, b = _prep(a), _prep(b)
        assert a.shape == b.shape
        out = torch.empty((), dtype=torch.float32, device=a.device)
        check(lib.kmh_mse_fwd(_p(a), _p(b), a.numel(), _p(out), _p(_reduce_ws(a.device)), _stream()), "kmh_mse_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = _prep(g).reshape(1)
        da = torch.empty_like(a)
        check(lib.kmh_mse_bwd(_p(a), _p(b), _p(g), a.numel(), _p(da), _stream()), "kmh_mse_bwd")
        return (da if ctx.needs_input_grad[0] else None), (-da if ctx.needs_input_grad[1] else None)


def mse_loss(a: Tensor, b: Tensor) -> Tensor:
    return _MSE.apply(a, b)


class _DiceRows(torch.autograd.Function):
    """per-row Dice loss 1 - (2 sum tp + 1)/(sum p^2 + sum t^2 + 1); rows = n*c."""

    @staticmethod
    def forward(ctx, pred, target):
        lib = _lib.load()
        pred, target = _prep(pred), _prep(target)
        R, V = pred.shape
        sums = torch.empty((R, 3), dtype=torch.float32, device=pred.device)
        check(lib.kmh_dice_sums(_p(pred), _p(target), R, V, _p(sums), _p(_reduce_ws(pred.device)), _stream()),
              "kmh_dice_sums")
        num = 2 * sums[:, 0] + 1
        den = sums[:, 1] + sums[:, 2] + 1
        ctx.save_for_backward(pred, target, num, den)
        return 1 - num / den

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        pred, target, num, den = ctx.saved_tensors
        R, V = pred.shape
        g = _prep(g)
        ca = (-2.0 * g / den).contiguous()
        cb = (2.0 * g * num / (den * den)).contiguous()
        dpred = torch.empty_like(pred)
        check(lib.kmh_rows_axpby(_p(target), _p(pred), _p(ca), _p(cb), R, V, _p(dpred), _stream()), "kmh_rows_axpby")
        return dpred, None


def dice_rows(pred: Tensor, target: Tensor) -> Tensor:
    return _DiceRows.apply(pred, target)


class _WarpDiceRows(torch.autograd.Function):
    """Dice rows of align_img(grid, x) against `fixed` WITHOUT the warped tensor: one pass for the three sums per (n, c)
    (kmh_warp_dice_sums), one pass for d/d(grid) (kmh_warp_dice_bwd_grid; the warp is recomputed, nothing is stored).
    scripts/train.py:146-164 with loss_fn == "dice"; keymorph/utils.py:14-21 + keymorph/loss_ops.py:16-63.
    Both segmentations are first checked for being exactly one-hot ON THE DEVICE (kmh_onehot_to_labels: one streaming
    read each, writes byte label maps + a flag); when they are -- one_hot() of a label map, nearest-sampled augmentation --
    the two passes read one byte per voxel instead of 4 C; a soft segmentation clears the flag and the same launches read
    the float tensors.  No host synchronisation decides; the results are bit-identical either way."""

    @staticmethod
    def forward(ctx, x, grid, fixed):
        lib = _lib.load()
        x, grid, fixed = _prep(x), _prep(grid), _prep(fixed)
        N, C, D, H, W = x.shape
        _, Do, Ho, Wo, _ = grid.shape
        assert fixed.shape == (N, C, Do, Ho, Wo), "the fixed segmentation must have the warped tensor's shape"
        sums = torch.empty((N * C, 3), dtype=torch.float32, device=x.device)
        labx = labf = gate = None
        # (label maps pay from ~4 channels on: at C = 1 the class loop costs more than the one gather it saves -- 0.175 vs 0.102 ms)
        if 4 <= C <= 255 and not os.environ.get("KEYMORPH_DICE_NO_LABELS"):
            labx = torch.empty((N, D * H * W), dtype=torch.uint8, device=x.device)
            labf = torch.empty((N, Do * Ho * Wo), dtype=torch.uint8, device=x.device)
            gate = torch.ones(1, dtype=torch.int32, device=x.device)
            check(lib.kmh_onehot_to_labels(_p(x), N, C, D * H * W, _p(labx), _p(gate), _stream()), "kmh_onehot_to_labels")
            check(lib.kmh_onehot_to_labels(_p(fixed), N, C, Do * Ho * Wo, _p(labf), _p(gate), _stream()),
                  "kmh_onehot_to_labels")
        if _lib.profiler.enabled:      # grid 12 B + C * (gathered volume 4 + fixed 4) per output voxel
            _lib.profiler.meta = {"bytes": float(N * Do * Ho * Wo) * (12 + 8 * C)}
        check(lib.kmh_warp_dice_sums(_p(x), _p(grid), _p(fixed), _p(sums), N, C, D, H, W, Do, Ho, Wo, _p(labx), _p(labf),
                                     _p(gate), _p(_reduce_ws(x.device)), _stream()), "kmh_warp_dice_sums")
        num = 2 * sums[:, 0] + 1
        den = sums[:, 1] + sums[:, 2] + 1
        ctx.labels = (labx, labf, gate)
        ctx.save_for_backward(x, grid, fixed, num, den)
        return (1 - num / den).view(N, C)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, grid, fixed, num, den = ctx.saved_tensors
        labx, labf, gate = ctx.labels
        N, C, D, H, W = x.shape
        _, Do, Ho, Wo, _ = grid.shape
        g = _prep(g).reshape(-1)
        ca = (-2.0 * g / den).contiguous()
        cb = (2.0 * g * num / (den * den)).contiguous()
        dgrid = torch.empty_like(grid)
        if _lib.profiler.enabled:      # the same reads + 12 B of grid gradient
            _lib.profiler.meta = {"bytes": float(N * Do * Ho * Wo) * (24 + 8 * C)}
        check(lib.kmh_warp_dice_bwd_grid(_p(x), _p(grid), _p(fixed), _p(ca), _p(cb), _p(dgrid), N, C, D, H, W, Do, Ho, Wo,
                                         _p(labx), _p(labf), _p(gate), _stream()), "kmh_warp_dice_bwd_grid")
        return None, dgrid, None


def warp_dice_ok(x: Tensor, grid: Tensor) -> bool:
    """does the fused warp + Dice pass apply?  The size conditions are the library's own (`kmh_warp_dice_ok`, the same
    predicate its two entry points return -22 on: W >= 2, < 2^30 voxels per channel plane, <= 128 channels, N * C <= 65536
    rows, the lane-contiguous sampler not switched off by KMH_SAMPLER_OLD); on top of them the grid must be the only input
    that needs a gradient (`fixed.requires_grad` is the caller's check: loss_ops.warp_dice_loss)."""
    if not (x.dim() == 5 and grid.dim() == 5 and not x.requires_grad):
        return False
    N, C, D, H, W = (int(v) for v in x.shape)
    return bool(_lib.load().kmh_warp_dice_ok(N, C, D, H, W))


def warp_dice_rows(x: Tensor, grid: Tensor, fixed: Tensor) -> Tensor:
    """-> (N, C) rows 1 - (2 sum t p + 1) / (sum p^2 + sum t^2 + 1) with p = align_img(grid, x), t = fixed; the gradient
    flows to ``grid`` only."""
    return _WarpDiceRows.apply(x, grid, fixed)


def argmax_onehot(pred: Tensor) -> Tensor:
    lib = _lib.load()
    pred = _prep(pred)
    N, C, V = pred.shape
    out = torch.empty_like(pred)
    check(lib.kmh_argmax_onehot(_p(pred), N, C, V, _p(out), _stream()), "kmh_argmax_onehot")
    return out


# --------------------------------------------------------------------------
# a9 / a8  grid generators
# --------------------------------------------------------------------------
class _AffineGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mat, D, H, W):
        lib = _lib.load()
        mat = _prep(mat)
        N = mat.shape[0]
        assert mat.shape[1:] == (3, 4)
        out = torch.empty((N, D, H, W, 3), dtype=torch.float32, device=mat.device)
        check(lib.kmh_affine_grid_fwd(_p(mat), _p(out), N, D, H, W, _stream()), "kmh_affine_grid_fwd")
        ctx.dims = (N, D, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        N, D, H, W = ctx.dims
        g = _prep(g)
        dmat = torch.empty((N, 3, 4), dtype=torch.float32, device=g.device)
        check(lib.kmh_affine_grid_bwd(_p(g), _p(dmat), N, D, H, W, _p(_reduce_ws(g.device)), _stream()),
              "kmh_affine_grid_bwd")
        return dmat, None, None, None


def affine_grid(mat34: Tensor, shape: Sequence[int]) -> Tensor:
    D, H, W = (int(s) for s in shape)
    return _AffineGrid.apply(mat34, D, H, W)


class _TpsGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, ctrl, D, H, W):
        lib = _lib.load()
        theta, ctrl = _prep(theta), _prep(ctrl)
        N, T, _ = ctrl.shape
        assert theta.shape == (N, T + 4, 3)
        out = torch.empty((N, D, H, W, 3), dtype=torch.float32, device=ctrl.device)
        check(lib.kmh_tps_grid_fwd(_p(theta), _p(ctrl), _p(out), N, T, D, H, W, _stream()), "kmh_tps_grid_fwd")
        ctx.save_for_backward(theta, ctrl)
        ctx.dims = (N, T, D, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        theta, ctrl = ctx.saved_tensors
        N, T, D, H, W = ctx.dims
        g = _prep(g)
        dtheta = torch.empty_like(theta)
        dctrl = torch.empty_like(ctrl)
        ws = workspace(int(lib.kmh_tps_grid_bwd_ws_bytes(N, T, D, H, W)), g.device, "tps_bwd")
        check(lib.kmh_tps_grid_bwd(_p(g), _p(theta), _p(ctrl), _p(dtheta), _p(dctrl), N, T, D, H, W, _p(ws),
                                   _stream()), "kmh_tps_grid_bwd")
        return dtheta, dctrl, None, None, None


def tps_grid(theta: Tensor, ctrl: Tensor, shape: Sequence[int]) -> Tensor:
    D, H, W = (int(s) for s in shape)
    return _TpsGrid.apply(theta, ctrl, D, H, W)


class _TpsPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, ctrl, pts):
        lib = _lib.load()
        theta, ctrl, pts = _prep(theta), _prep(ctrl), _prep(pts)
        N, T, _ = ctrl.shape
        P = pts.shape[1]
        out = torch.empty_like(pts)
        check(lib.kmh_tps_points_fwd(_p(theta), _p(ctrl), _p(pts), _p(out), N, T, P, _stream()), "kmh_tps_points_fwd")
        ctx.save_for_backward(theta, ctrl, pts)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        theta, ctrl, pts = ctx.saved_tensors
        N, T, _ = ctrl.shape
        P = pts.shape[1]
        g = _prep(g)
        dtheta, dctrl, dpts = torch.empty_like(theta), torch.empty_like(ctrl), torch.empty_like(pts)
        ws = workspace(int(lib.kmh_tps_points_bwd_ws_bytes(N, T, P)), g.device, "tps_bwd")
        check(lib.kmh_tps_points_bwd(_p(g), _p(theta), _p(ctrl), _p(pts), _p(dtheta), _p(dctrl), _p(dpts), N, T, P,
                                     _p(ws), _stream()), "kmh_tps_points_bwd")
        return dtheta, dctrl, dpts


def tps_points(theta: Tensor, ctrl: Tensor, pts: Tensor) -> Tensor:
    return _TpsPoints.apply(theta, ctrl, pts)


class _AffinePoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M, pts):
        lib = _lib.load()
        M, pts = _prep(M), _prep(pts)
        N, P, _ = pts.shape
        out = torch.empty_like(pts)
        check(lib.kmh_affine_points_fwd(_p(M), _p(pts), _p(out), N, P, _stream()), "kmh_affine_points_fwd")
        ctx.save_for_backward(M, pts)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        M, pts = ctx.saved_tensors
        N, P, _ = pts.shape
        g = _prep(g)
        dM, dpts = torch.empty_like(M), torch.empty_like(pts)
        check(lib.kmh_affine_points_bwd(_p(g), _p(M), _p(pts), _p(dM), _p(dpts), N, P, _stream()),
              "kmh_affine_points_bwd")
        return dM, dpts


def affine_points(mat34: Tensor, pts: Tensor) -> Tensor:
    return _AffinePoints.apply(mat34, pts)


# --------------------------------------------------------------------------
# a5 / a6 / a7  fits
# --------------------------------------------------------------------------
class _MatrixFit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, w, kind):
        lib = _lib.load()
        x, y = _prep(x), _prep(y)
        w = None if w is None else _prep(w)
        N, K, _ = x.shape
        M = torch.empty((N, 3, 4), dtype=torch.float32, device=x.device)
        fn = lib.kmh_affine_fit_fwd if kind == "affine" else lib.kmh_rigid_fit_fwd
        check(fn(_p(x), _p(y), _p(w), _p(M), N, K, _stream()), f"kmh_{kind}_fit_fwd")
        ctx.save_for_backward(x, y, M) if w is None else ctx.save_for_backward(x, y, M, w)
        ctx.kind = kind
        return M

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        saved = ctx.saved_tensors
        x, y, M = saved[:3]
        w = saved[3] if len(saved) > 3 else None
        N, K, _ = x.shape
        g = _prep(g)
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        dw = torch.empty_like(w) if (w is not None and ctx.needs_input_grad[2]) else None
        if ctx.kind == "affine":
            check(lib.kmh_affine_fit_bwd(_p(g), _p(x), _p(y), _p(w), _p(M), _p(dx), _p(dy), _p(dw), N, K, _stream()),
                  "kmh_affine_fit_bwd")
        else:
            check(lib.kmh_rigid_fit_bwd(_p(g), _p(x), _p(y), _p(w), _p(dx), _p(dy), _p(dw), N, K, _stream()),
                  "kmh_rigid_fit_bwd")
        return dx, dy, dw, None


def affine_fit(x: Tensor, y: Tensor, w: Optional[Tensor] = None) -> Tensor:
    return _MatrixFit.apply(x, y, w, "affine")


def rigid_fit(x: Tensor, y: Tensor, w: Optional[Tensor] = None) -> Tensor:
    return _MatrixFit.apply(x, y, w, "rigid")


class _AffineInverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M):
        lib = _lib.load()
        M = _prep(M)
        Mi = torch.empty_like(M)
        check(lib.kmh_affine_inverse_fwd(_p(M), _p(Mi), M.shape[0], _stream()), "kmh_affine_inverse_fwd")
        ctx.save_for_backward(Mi)
        return Mi

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (Mi,) = ctx.saved_tensors
        g = _prep(g)
        dM = torch.empty_like(Mi)
        check(lib.kmh_affine_inverse_bwd(_p(g), _p(Mi), _p(dM), Mi.shape[0], _stream()), "kmh_affine_inverse_bwd")
        return dM


def affine_inverse(mat34: Tensor) -> Tensor:
    """(N,3,4) -> top 3 rows of inverse([M; 0 0 0 1])."""
    return _AffineInverse.apply(mat34)


class _TpsFit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ctrl, tgt, lmbda, w):
        lib = _lib.load()
        ctrl, tgt, lmbda = _prep(ctrl), _prep(tgt), _prep(lmbda)
        w = None if w is None else _prep(w)
        N, T, _ = ctrl.shape
        assert lmbda.numel() == N
        theta = torch.empty((N, T + 4, 3), dtype=torch.float32, device=ctrl.device)
        # the LU factors must outlive the forward (the backward re-uses them): dedicated buffer
        ws = torch.empty(int(lib.kmh_tps_fit_ws_bytes(N, T)), dtype=torch.uint8, device=ctrl.device)
        check(lib.kmh_tps_fit_fwd(_p(ctrl), _p(tgt), _p(lmbda), _p(w), _p(theta), N, T, _p(ws), _stream()),
              "kmh_tps_fit_fwd")
        ctx.save_for_backward(ctrl, lmbda, theta, ws) if w is None else ctx.save_for_backward(ctrl, lmbda, theta, ws, w)
        return theta

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        saved = ctx.saved_tensors
        ctrl, lmbda, theta, ws = saved[:4]
        w = saved[4] if len(saved) > 4 else None
        N, T, _ = ctrl.shape
        g = _prep(g)
        dctrl, dtgt = torch.empty_like(ctrl), torch.empty_like(ctrl)
        dw = torch.empty_like(w) if (w is not None and ctx.needs_input_grad[3]) else None
        check(lib.kmh_tps_fit_bwd(_p(g), _p(theta), _p(ctrl), _p(lmbda), _p(w), _p(dctrl), _p(dtgt), _p(dw), N, T,
                                  _p(ws), _stream()), "kmh_tps_fit_bwd")
        return dctrl, dtgt, None, dw


def tps_fit(ctrl: Tensor, tgt: Tensor, lmbda: Tensor, w: Optional[Tensor] = None) -> Tensor:
    return _TpsFit.apply(ctrl, tgt, lmbda, w)


# --------------------------------------------------------------------------
# a4  center of mass on a materialised heat-map
# --------------------------------------------------------------------------
class _Com3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat):
        lib = _lib.load()
        feat = _prep(feat)
        N, K, D, H, W = feat.shape
        pts = torch.empty((N, K, 3), dtype=torch.float32, device=feat.device)
        sums = torch.empty((N, K, 4), dtype=torch.float32, device=feat.device)
        check(lib.kmh_com3d_fwd(_p(feat), _p(pts), _p(sums), N, K, D, H, W, _p(_reduce_ws(feat.device)), _stream()),
              "kmh_com3d_fwd")
        ctx.save_for_backward(feat, sums)
        return pts

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        feat, sums = ctx.saved_tensors
        N, K, D, H, W = feat.shape
        g = _prep(g)
        dfeat = torch.empty_like(feat)
        check(lib.kmh_com3d_bwd(_p(g), _p(feat), _p(sums), _p(dfeat), N, K, D, H, W, _stream()), "kmh_com3d_bwd")
        return dfeat


def com3d(feat: Tensor) -> Tensor:
    """(N,K,D,H,W) -> (N,K,3) in (z,y,x) order, [-1,1]."""
    return _Com3d.apply(feat)
