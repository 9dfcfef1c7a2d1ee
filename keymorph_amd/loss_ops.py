"""Losses with the reference's call surface (keymorph/loss_ops.py:9-63), HIP underneath."""
import torch

from . import ops


class MSELoss(torch.nn.Module):
    """keymorph/loss_ops.py:9-13"""

    def forward(self, pred, target):
        return ops.mse_loss(pred, target)


class DiceLoss(torch.nn.Module):
    """Soft / hard Dice loss (lower is better), keymorph/loss_ops.py:16-63.

    eps = 1 is added to numerator and denominator; the denominator uses squared sums.
    """

    def __init__(self, hard=False, return_regions=False):
        super().__init__()
        self.hard = hard
        self.return_regions = return_regions

    def forward(self, pred, target, ign_first_ch=False):
        assert pred.size() == target.size(), "Input and target are different dim"
        assert target.dim() in (4, 5)
        n, c = target.shape[:2]
        target = target.contiguous().view(n, c, -1)
        pred = pred.contiguous().view(n, c, -1)
        if self.hard:
            pred = ops.argmax_onehot(pred)
        if ign_first_ch:
            target = target[:, 1:, :]
            pred = pred[:, 1:, :]
            c -= 1
        v = target.shape[-1]
        rows = ops.dice_rows(pred.reshape(n * c, v), target.reshape(n * c, v)).view(n, c)
        if self.return_regions:
            return rows.mean(0)
        return rows.mean()


def warp_dice_loss(grid, seg_m, seg_f, ign_first_ch=False, return_regions=False):
    """`DiceLoss(return_regions=...)(align_img(grid, seg_m), seg_f, ign_first_ch)` -- the Dice branch of
    scripts/train.py:146-164 -- as ONE fused operator: the warped segmentation is never materialised (forward: one pass
    over grid + both segmentations; backward: one more pass that writes d(loss)/d(grid)).  Falls back to exactly that
    composition when the fused kernels do not apply (seg_m needing a gradient, 4-D inputs, > 128 channels)."""
    if not ops.warp_dice_ok(seg_m, grid) or seg_f.requires_grad:
        from .utils import align_img
        return DiceLoss(return_regions=return_regions)(align_img(grid, seg_m), seg_f, ign_first_ch=ign_first_ch)
    rows = ops.warp_dice_rows(seg_m, grid, seg_f)
    if ign_first_ch:
        rows = rows[:, 1:]
    return rows.mean(0) if return_regions else rows.mean()


# --------------------------------------------------------------------------
# eval-only metrics on the GPU (keymorph/loss_ops.py:161-247; callers pairwise_register_eval.py:332-345)
# --------------------------------------------------------------------------
def _jacdet(disp, want_map):
    from . import _lib
    from .ops import _p, _reduce_ws, _stream, check
    lib = _lib.load()
    assert disp.dim() == 5 and disp.shape[0] == 1 and disp.shape[1] == 3, "expected a (1, 3, D, H, W) map"
    if disp.dtype != torch.float32 or not disp.is_cuda:
        raise _lib.KeymorphHipError("jacobian determinant: expected a float32 tensor on the GPU")
    _, _, D, H, W = disp.shape
    st = disp.stride()
    if st[2:] == (H * W * st[4], W * st[4], st[4]) and st[4] in (1, 3) and (st[1] == 1 or st[1] == D * H * W):
        src, cs, vs = disp, st[1], st[4]          # NCDHW, or the permuted view of a (1, D, H, W, 3) grid: no copy
    else:
        src = disp.contiguous()
        cs, vs = D * H * W, 1
    jd = torch.empty((D - 4, H - 4, W - 4), dtype=torch.float32, device=disp.device) if want_map else None
    stats = torch.empty(4, dtype=torch.float64, device=disp.device)
    check(lib.kmh_jacobian_det(_p(src), cs, vs, D, H, W, _p(jd), _p(stats), _p(_reduce_ws(disp.device)), _stream()),
          "kmh_jacobian_det")
    return jd, stats


def _jacobian_determinant(disp):
    """(1, 3, D, H, W) -> (D-4, H-4, W-4) determinants of d(disp)/d(z,y,x) + I (loss_ops.py:161-228)."""
    return _jacdet(disp, True)[0]


def jdstd(disp):
    """Population standard deviation of the Jacobian determinant (loss_ops.py:231-234); Python float."""
    return float(_jacdet(disp, False)[1][1])


def jdlessthan0(disp, as_percentage=False):
    """Number (or fraction) of voxels with a non-positive Jacobian determinant (loss_ops.py:237-242)."""
    st = _jacdet(disp, False)[1]
    return float(st[2] / st[3]) if as_percentage else int(st[2])


# --------------------------------------------------------------------------
# groupwise evaluation metrics over files or tensor stacks (keymorph/loss_ops.py:406-551; callers
# scripts/groupwise_register_eval.py:478-515): the pairwise / per-grid averages, HIP losses underneath
# --------------------------------------------------------------------------
def _load_file(path, device=None):
    """loss_ops.py:406-412 (.npy; NIfTI through keymorph_amd.io.read_nifti instead of nibabel); lands on the GPU."""
    import numpy as np
    path = str(path)
    if path.endswith(".nii") or path.endswith(".nii.gz"):
        from .io import read_nifti
        arr = read_nifti(path)[0]
    elif path.endswith(".npy"):
        arr = np.load(path)
    else:
        raise ValueError("File format not supported")
    return torch.tensor(arr).to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))


def _item(batch, i):
    return _load_file(batch[i]) if isinstance(batch[0], (str, bytes)) or hasattr(batch[0], "__fspath__") else batch[i:i + 1]


class _AvgPairwiseLoss(torch.nn.Module):
    """Mean of metric_fn over all unordered pairs (loss_ops.py:415-435)."""

    def __init__(self, metric_fn):
        super().__init__()
        self.metric_fn = metric_fn

    def forward(self, batch_of_imgs):
        loss, num = 0, 0
        for i in range(len(batch_of_imgs)):
            for j in range(i + 1, len(batch_of_imgs)):
                loss = loss + self.metric_fn(_item(batch_of_imgs, i), _item(batch_of_imgs, j))
                num += 1
        return loss / num


class MSEPairwiseLoss(_AvgPairwiseLoss):
    def __init__(self):
        super().__init__(MSELoss().forward)


class SoftDicePairwiseLoss(_AvgPairwiseLoss):
    def __init__(self):
        super().__init__(DiceLoss().forward)


class HardDicePairwiseLoss(_AvgPairwiseLoss):
    def __init__(self):
        super().__init__(DiceLoss(hard=True).forward)


class MultipleAvgSegPairwiseMetric(torch.nn.Module):
    """Several pairwise segmentation metrics in one sweep over the files (loss_ops.py:499-527).  'hausd' (scipy /
    skimage surface distances on the host) is outside the registration path and not provided."""

    def __init__(self):
        super().__init__()
        self.name2fn = {"harddice": DiceLoss(hard=True).forward,
                        "harddiceroi": DiceLoss(hard=True, return_regions=True).forward,
                        "softdice": DiceLoss().forward}

    def forward(self, batch_of_imgs, fn_names):
        for name in fn_names:
            if name not in self.name2fn:
                raise NotImplementedError(f"metric '{name}' is not part of the MI355X registration path")
        res, num = {name: 0 for name in fn_names}, 0
        for i in range(len(batch_of_imgs)):
            for j in range(i + 1, len(batch_of_imgs)):
                a, b = _item(batch_of_imgs, i), _item(batch_of_imgs, j)
                for name in fn_names:
                    res[name] = res[name] + self.name2fn[name](a, b)
                num += 1
        return {name: res[name] / num for name in fn_names}


class MultipleAvgGridMetric(torch.nn.Module):
    """Jacobian-determinant metrics averaged over the grids of a group (loss_ops.py:530-551)."""

    def __init__(self):
        super().__init__()
        self.name2fn = {"jdstd": jdstd, "jdlessthan0": jdlessthan0}

    def forward(self, batch_of_grids, fn_names):
        res = {name: 0 for name in fn_names}
        for i in range(len(batch_of_grids)):
            grid = _item(batch_of_grids, i).float()
            gp = grid.permute(0, 4, 1, 2, 3)
            for name in fn_names:
                res[name] += self.name2fn[name](gp)
        return {name: res[name] / len(batch_of_grids) for name in fn_names}


class AvgJDStd(MultipleAvgGridMetric):
    def forward(self, batch_of_grids):
        return super().forward(batch_of_grids, ["jdstd"])["jdstd"]


class AvgJDLessThan0(MultipleAvgGridMetric):
    def forward(self, batch_of_grids):
        return super().forward(batch_of_grids, ["jdlessthan0"])["jdlessthan0"]


# Host-side scipy / skimage metrics of the reference that are not on the path: the names exist so that
# `import keymorph.loss_ops as loss_ops` users can reference them; calling / constructing raises NotImplementedError.
from ._absent import absent_class as _absent_class, absent_function as _absent_function   # noqa: E402

fast_dice = _absent_function("fast_dice", "keymorph/loss_ops.py:66")
dice = _absent_function("dice", "keymorph/loss_ops.py:109")
hausdorff_distance = _absent_function("hausdorff_distance", "keymorph/loss_ops.py:142")
LC2 = _absent_class("LC2", "keymorph/loss_ops.py:250", torch.nn.Module)
ImageLC2 = _absent_class("ImageLC2", "keymorph/loss_ops.py:305", torch.nn.Module)
HausdorffPairwiseLoss = _absent_class("HausdorffPairwiseLoss", "keymorph/loss_ops.py:459", torch.nn.Module)
