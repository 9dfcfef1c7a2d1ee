"""Affine augmentation on the GPU -- the step in front of the registration path in every training iteration
(reference: keymorph/augmentation.py:81-277, callers scripts/train.py:84-98, pairwise_register_eval.py:112-114).

Same surface and the same random draws as the reference (parameters are drawn on the host from torch's global
CPU generator in the order scale, offset, theta, shear, so a seeded script reproduces the reference's
augmentations); the 4x4 composition, the sampling grid and the bilinear / nearest warp run as HIP kernels
(kmh_affine_build_matrix, kmh_affine_grid_fwd, kmh_grid_sample3d_fwd).  3-D only."""
import torch

from . import _lib
from .ops import _p, _stream, check
from .transformations import AffineTransform
from .utils import align_img


class AffineDeformation2d:
    def __init__(self, device="cuda:0"):
        raise NotImplementedError("keymorph_amd implements the 3-D registration path")


class AffineDeformation3d:
    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)

    def build_affine_matrix(self, batch_size, params):
        """params = (scale (bs,3), offset (bs,3), theta (bs,3), shear (bs,6)) -> (bs,4,4);
        M = Mz Ms Mt Mr with Mr = R3 R2 R1 (augmentation.py:85-158)."""
        lib = _lib.load()
        # (1, k) parameters broadcast over the batch like the reference's (1,)-shaped tensors do (augmentation.py:96-158)
        scale, offset, theta, shear = (p.to(self.device, torch.float32).expand(batch_size, -1).contiguous()
                                       for p in params)
        for p, k in ((scale, 3), (offset, 3), (theta, 3), (shear, 6)):
            assert tuple(p.shape) == (batch_size, k), f"expected ({batch_size}, {k}) parameters, got {tuple(p.shape)}"
        out = torch.empty((batch_size, 4, 4), dtype=torch.float32, device=self.device)
        check(lib.kmh_affine_build_matrix(_p(scale), _p(offset), _p(theta), _p(shear), _p(out), batch_size, _stream()),
              "kmh_affine_build_matrix")
        return out

    def deform_img(self, img, params, interp_mode="bilinear"):
        Ma = self.build_affine_matrix(len(img), params)
        phi_inv = AffineTransform(matrix=Ma).get_flow_field(img.size())
        return align_img(phi_inv, img, mode=interp_mode)

    def deform_points(self, points, params):
        Ma = self.build_affine_matrix(len(points), params)
        return AffineTransform(matrix=Ma).get_forward_transformed_points(points)

    def __call__(self, img, **kwargs):
        return self.deform_img(img, kwargs["params"], kwargs["interp_mode"])


def _draw(img, lo_hi):
    """One host draw per parameter group, in the reference's order and from the same generator."""
    if img.dim() != 5:
        raise NotImplementedError("keymorph_amd implements the 3-D registration path")
    return tuple(torch.empty(1, n, dtype=torch.float32).uniform_(lo, hi) for n, (lo, hi) in zip((3, 3, 3, 6), lo_hi))


def _apply(augmenter, params, img, seg, points, matrix):
    out = (augmenter(img, params=params, interp_mode="bilinear"),)
    if seg is not None:
        out += (augmenter(seg, params=params, interp_mode="nearest"),)
    if points is not None:
        out += (augmenter.deform_points(points, params),)
    if matrix:
        out += (augmenter.build_affine_matrix(len(img), params),)
    return out[0] if len(out) == 1 else out


def random_affine_augment(img, seg=None, points=None, max_random_params=(0.2, 0.2, 3.1416, 0.1), scale_params=1,
                          return_affine_matrix=False):
    """augmentation.py:162-207.  img (bs, nch, D, H, W); the one (1, k) parameter draw is applied to every sample of the batch."""
    s, o, a, z = (p * scale_params for p in max_random_params)
    params = _draw(img, ((1 - s, 1 + s), (-o, o), (-a, a), (-z, z)))
    return _apply(AffineDeformation3d(device=img.device), params, img, seg, points, return_affine_matrix)


def affine_augment(img, fixed_params, seg=None, points=None):
    """augmentation.py:210-245: the same scale / offset / angle / shear on every axis."""
    if img.dim() != 5:
        raise NotImplementedError("keymorph_amd implements the 3-D registration path")
    s, o, a, z = fixed_params
    params = (torch.full((1, 3), 1.0 + s), torch.full((1, 3), float(o)), torch.full((1, 3), float(a)),
              torch.full((1, 6), float(z)))
    return _apply(AffineDeformation3d(device=img.device), params, img, seg, points, False)


def random_affine_augment_pair(img1, img2, max_random_params=(0.2, 0.2, 3.1416, 0.1), scale_params=1):
    """augmentation.py:248-277: one random transform applied to both images."""
    s, o, a, z = (p * scale_params for p in max_random_params)
    params = _draw(img1, ((1 - s, 1 + s), (-o, o), (-a, a), (-z, z)))
    aug = AffineDeformation3d(device=img2.device)
    return aug(img1, params=params, interp_mode="bilinear"), aug(img2, params=params, interp_mode="bilinear")
