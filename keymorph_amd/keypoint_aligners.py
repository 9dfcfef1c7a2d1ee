"""Closed-form keypoint aligners with the reference's surface
(keymorph/keypoint_aligners.py:14-465).  All points are (bs, K, 3) in ij (z, y, x) order.

Every aligner fits the INVERSE map (fixed -> moving), because that is what grid_sample needs
(keypoint_aligners.py:67-70, 274).  Batched by construction.
"""
import torch
import torch.nn as nn

from . import ops
from .transformations import AffineTransform
from .utils import convert_points_norm2real, convert_points_real2norm


class AffineKeypointAligner(AffineTransform):
    def __init__(self, points_m, points_f, w=None, dim=3, align_in_real_world_coords=False, aff_m=None,
                 aff_f=None, shape_m=None, shape_f=None):
        if dim != 3:
            raise NotImplementedError("keymorph_amd implements the 3-D registration path")
        self.dim = dim
        self.align_in_real_world_coords = align_in_real_world_coords
        self.points_f, self.points_m = points_f, points_m
        self.shape_f, self.shape_m = shape_f, shape_m
        if align_in_real_world_coords:
            for name, v in (("aff_f", aff_f), ("aff_m", aff_m), ("shape_f", shape_f), ("shape_m", shape_m)):
                assert v is not None, f"Need to provide {name} for real-world coords"
            assert points_f.shape[0] == 1 and points_m.shape[0] == 1, "Batch size must be 1 for real-world coords"
            self.aff_f, self.aff_m = aff_f, aff_m
            self.points_m = convert_points_norm2real(self.points_m, aff_m, shape_m)
            self.points_f = convert_points_norm2real(self.points_f, aff_f, shape_f)
        inverse = self._square(self.fit(self.points_f, self.points_m, w=w))
        super().__init__(inverse_matrix=inverse, dim=dim)

    def fit(self, x, y, w=None):
        """argmin_A ||A [x;1] - y||  ->  (bs, 3, 4)   (keypoint_aligners.py:76-114)."""
        return ops.affine_fit(x, y, w)

    def get_forward_transformed_points(self, points):
        if self.align_in_real_world_coords:
            points = convert_points_norm2real(points, self.aff_m, self.shape_m)
        points = super().get_forward_transformed_points(points)
        if self.align_in_real_world_coords:
            points = convert_points_real2norm(points, self.aff_f, self.shape_f)
        return points

    def get_inverse_transformed_points(self, points):
        if self.align_in_real_world_coords:
            points = convert_points_norm2real(points, self.aff_f, self.shape_f)
        points = super().get_inverse_transformed_points(points)
        if self.align_in_real_world_coords:
            points = convert_points_real2norm(points, self.aff_m, self.shape_m)
        return points

    def _norm_to_norm_inverse_matrix(self):
        """Real-world mode: the map the sampling grid needs is fixed-norm -> fixed-mm -> moving-mm -> moving-norm
        (keypoint_aligners.py:134-148 applied to every grid point by transformations.py:37-58).  All three stages
        are affine, so they compose into ONE (1, 3, 4) matrix for the grid kernel:
            norm2real_f = aff_f . [diag(S_f/2) | S_f/2 - 1/2]        (utils.py:243-259, 275-291)
            real2norm_m = [diag(2/S_m) | 1/S_m - 1] . aff_m^-1       (utils.py:262-272, 294-317)
        composed in fp64 (mm-scale entries; the product is back in [-1, 1] units), differentiable in the fit."""
        dd = torch.float64
        dev = self.inverse_transform_matrix.device
        sf = torch.as_tensor(self.shape_f).to(dev, dd).reshape(-1)
        sm = torch.as_tensor(self.shape_m).to(dev, dd).reshape(-1)
        n2v = torch.eye(4, dtype=dd, device=dev)
        n2v[:3, :3] = torch.diag(sf / 2)
        n2v[:3, 3] = sf / 2 - 0.5
        v2n = torch.eye(4, dtype=dd, device=dev)
        v2n[:3, :3] = torch.diag(2 / sm)
        v2n[:3, 3] = 1 / sm - 1
        n2r_f = self.aff_f.to(dev, dd) @ n2v
        r2n_m = v2n @ torch.inverse(self.aff_m.to(dev, dd))
        return (r2n_m @ self.inverse_transform_matrix.to(dd) @ n2r_f).float()

    def get_flow_field(self, grid_shape, **kwargs):
        if not self.align_in_real_world_coords:
            return super().get_flow_field(grid_shape, **kwargs)
        return ops.affine_grid(self._norm_to_norm_inverse_matrix()[:, :3, :].contiguous(), grid_shape[2:])

    def grid_from_points(self, points_m, points_f, grid_shape, lmbda=None, weights=None, compute_on_subgrids=False):
        """README.md:74 compatibility shim: construct + get_flow_field."""
        return type(self)(points_m=points_m, points_f=points_f, w=weights, dim=self.dim).get_flow_field(grid_shape)


class RigidKeypointAligner(AffineKeypointAligner):
    def fit(self, p1, p2, w=None):
        """Kabsch [R|T] with the reference's row-scaled reflection fix (keypoint_aligners.py:151-213)."""
        return ops.rigid_fit(p1, p2, w)


class TPS(nn.Module):
    """Thin-plate spline aligner (keypoint_aligners.py:216-465).  The linear system is assembled,
    LU-factorised (fp64) and solved on the GPU once per direction and cached -- the reference
    re-fits on the host on every call; results are identical."""

    def __init__(self, points_m, points_f, lmbda, w=None, dim=3, num_subgrids=4, use_checkpoint=False,
                 align_in_real_world_coords=False, aff_m=None, aff_f=None, shape_m=None, shape_f=None):
        super().__init__()
        if dim != 3:
            raise NotImplementedError("keymorph_amd implements the 3-D registration path")
        self.dim = dim
        self.num_subgrids = num_subgrids      # kept for signature parity: the fused evaluator never chunks
        self.use_checkpoint = use_checkpoint  # idem: nothing large is materialised, nothing to checkpoint
        self.lmbda = lmbda
        self.weights = w
        self.align_in_real_world_coords = align_in_real_world_coords
        self.points_f, self.points_m = points_f, points_m
        self.shape_f, self.shape_m = shape_f, shape_m
        if align_in_real_world_coords:
            for name, v in (("aff_f", aff_f), ("aff_m", aff_m), ("shape_f", shape_f), ("shape_m", shape_m)):
                assert v is not None, f"Need to provide {name} for real-world coords"
            assert points_f.shape[0] == 1 and points_m.shape[0] == 1, "Batch size must be 1 for real-world coords"
            self.aff_f, self.aff_m = aff_f, aff_m
            self.points_m = convert_points_norm2real(self.points_m, aff_m, shape_m)
            self.points_f = convert_points_norm2real(self.points_f, aff_f, shape_f)
        # both directions are fitted on first use and cached: the groupwise iterations (model.py:331-444) only ever
        # need the forward map, a registration without aligned points only the inverse one
        self._inverse_theta = None
        self.theta = None

    @property
    def inverse_theta(self):
        """theta of the fixed -> moving map the sampling grid needs (keypoint_aligners.py:274)"""
        if self._inverse_theta is None:
            self._inverse_theta = self.fit(self.points_f, self.points_m, self.lmbda, weights=self.weights)
        return self._inverse_theta

    @staticmethod
    def _lmbda_vec(lmbda, n, device):
        lm = torch.as_tensor(lmbda, dtype=torch.float32, device=device).reshape(-1)
        return lm.expand(n).contiguous() if lm.numel() == 1 else lm

    def fit(self, c_src, c_dst, lmbda, weights=None):
        """theta (bs, T+4, 3): rows [w_0..w_{T-1}, a_1, a_z, a_y, a_x] (keypoint_aligners.py:276-363)."""
        return ops.tps_fit(c_src, c_dst, self._lmbda_vec(lmbda, c_src.shape[0], c_src.device), weights)

    def transform_points(self, theta, ctrl, points):
        """keypoint_aligners.py:399-433"""
        return ops.tps_points(theta, ctrl, points)

    def get_flow_field(self, grid_shape, compute_on_subgrids=False):
        """(bs, D, H, W, 3) xyz grid (keypoint_aligners.py:365-397).  ``compute_on_subgrids`` only bounds
        the reference's (K, N, 3) temporaries; the fused kernel has none, so it is accepted and ignored."""
        if self.align_in_real_world_coords:
            flat = torch.stack(torch.meshgrid(*[torch.linspace(-1, 1, int(s)) for s in grid_shape[2:]],
                                              indexing="ij"), -1).reshape(1, -1, 3).to(self.points_f)
            out = self.get_inverse_transformed_points(flat)
            return out.reshape(1, *grid_shape[2:], 3).flip(-1)
        return ops.tps_grid(self.inverse_theta, self.points_f, grid_shape[2:])

    def get_inverse_transformed_points(self, points):
        if self.align_in_real_world_coords:
            points = convert_points_norm2real(points, self.aff_f, self.shape_f)
        points = self.transform_points(self.inverse_theta, self.points_f, points)
        if self.align_in_real_world_coords:
            points = convert_points_real2norm(points, self.aff_m, self.shape_m)
        return points

    def get_forward_transformed_points(self, points):
        if self.theta is None:
            self.theta = self.fit(self.points_m, self.points_f, self.lmbda, weights=self.weights)
        if self.align_in_real_world_coords:
            points = convert_points_norm2real(points, self.aff_m, self.shape_m)
        points = self.transform_points(self.theta, self.points_m, points)
        if self.align_in_real_world_coords:
            points = convert_points_real2norm(points, self.aff_f, self.shape_f)
        return points

    def grid_from_points(self, points_m, points_f, grid_shape, lmbda=None, weights=None, compute_on_subgrids=False):
        """README.md:74 compatibility shim."""
        return TPS(points_m=points_m, points_f=points_f, lmbda=self.lmbda if lmbda is None else lmbda, w=weights,
                   dim=self.dim).get_flow_field(grid_shape, compute_on_subgrids=compute_on_subgrids)
