"""Analytic known-answer cases in the spirit of the reference's test/test.py (center of mass of
blurred deltas, pure translations / 90-degree rotations for the rigid + affine aligners, coordinate
conversions).  The CPU half checks the oracle, the GPU half the HIP path."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import keymorph_oracle as O

DEV = "cuda"


def blurred_delta(shape, at, sigma=5):
    img = np.zeros(shape)
    img[at] = 1
    return torch.tensor(ndimage.gaussian_filter(img, sigma)).float()[None, None]


COM_CASES = [
    ((3, 3, 3), (1, 1, 1), 0, (0.0, 0.0, 0.0)),
    ((101, 101, 101), (50, 50, 50), 5, (0.0, 0.0, 0.0)),
    ((101, 51, 51), (50, 25, 25), 5, (0.0, 0.0, 0.0)),
    ((101, 101, 101), (50, 25, 25), 5, (0.0, -0.5, -0.5)),   # (z, y, x)
    ((101, 101, 101), (25, 50, 50), 5, (-0.5, 0.0, 0.0)),
]


@pytest.mark.parametrize("shape,at,sigma,want", COM_CASES)
def test_com_oracle(shape, at, sigma, want):
    vol = blurred_delta(shape, at, sigma) if sigma else torch.zeros(1, 1, *shape).index_put_(
        tuple(torch.tensor([i]) for i in (0, 0) + at), torch.tensor([1.0]))
    torch.testing.assert_close(O.center_of_mass(vol, "ij"), torch.tensor(want).view(1, 1, 3))
    torch.testing.assert_close(O.center_of_mass(vol, "xy"), torch.tensor(want[::-1]).view(1, 1, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,at,sigma,want", COM_CASES)
def test_com_hip(shape, at, sigma, want):
    from keymorph_amd.layers import CenterOfMass3d
    vol = blurred_delta(shape, at, sigma) if sigma else torch.zeros(1, 1, *shape).index_put_(
        tuple(torch.tensor([i]) for i in (0, 0) + at), torch.tensor([1.0]))
    torch.testing.assert_close(CenterOfMass3d("ij")(vol.to(DEV)).cpu(), torch.tensor(want).view(1, 1, 3))
    torch.testing.assert_close(CenterOfMass3d("xy")(vol.to(DEV)).cpu(), torch.tensor(want[::-1]).view(1, 1, 3))
    # batched + 2-D layer
    from keymorph_amd.layers import CenterOfMass2d
    img = torch.zeros(1, 1, 3, 3)
    img[0, 0, 0, 0] = img[0, 0, 2, 2] = 1
    torch.testing.assert_close(CenterOfMass2d()(img.to(DEV)).cpu(), torch.zeros(1, 1, 2))


def _rot90_z():
    return torch.tensor([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])


RIGID_CASES = []
_p = torch.tensor([[0.1, 0.2, 0.3], [0.5, -0.2, 0.1], [-0.3, 0.4, -0.6], [0.7, 0.1, -0.2], [-0.5, -0.5, 0.2]])
RIGID_CASES.append(("translate", _p, _p + torch.tensor([0.0, 0.0, 0.1]), torch.eye(3), torch.tensor([0.0, 0.0, 0.1])))
RIGID_CASES.append(("rot90", _p, _p @ _rot90_z().T, _rot90_z(), torch.zeros(3)))
RIGID_CASES.append(("rot+shift", _p, _p @ _rot90_z().T + torch.tensor([0.2, -0.1, 0.05]), _rot90_z(),
                    torch.tensor([0.2, -0.1, 0.05])))


def _expect(R, t):
    M = torch.eye(4)
    M[:3, :3], M[:3, 3] = R, t
    return M[None]


@pytest.mark.parametrize("name,pm,pf,R,t", RIGID_CASES)
def test_rigid_affine_oracle(name, pm, pf, R, t):
    """transform_matrix maps moving -> fixed points (reference test_rigid_* convention)."""
    for fit in (O.rigid_fit, O.affine_fit):
        inv = O.square(fit(pf[None], pm[None]))
        torch.testing.assert_close(torch.inverse(inv), _expect(R, t), atol=2e-5, rtol=1e-4)
    # rigid is scale-blind: scaling the moving cloud must not change R
    inv = O.square(O.rigid_fit(pf[None], pm[None] * 1.0))
    assert abs(float(torch.det(torch.inverse(inv)[0, :3, :3])) - 1) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name,pm,pf,R,t", RIGID_CASES)
def test_rigid_affine_hip(name, pm, pf, R, t):
    from keymorph_amd.keypoint_aligners import AffineKeypointAligner, RigidKeypointAligner
    for cls in (RigidKeypointAligner, AffineKeypointAligner):
        al = cls(pm[None].to(DEV), pf[None].to(DEV), dim=3)
        torch.testing.assert_close(al.transform_matrix.cpu(), _expect(R, t), atol=2e-5, rtol=1e-4)
        back = cls(pf[None].to(DEV), pm[None].to(DEV), dim=3)   # forward(m->f) == inverse(f->m)
        torch.testing.assert_close(al.transform_matrix.cpu(), back.inverse_transform_matrix.cpu(), atol=2e-5, rtol=1e-4)
        torch.testing.assert_close(al.get_forward_transformed_points(pm[None].to(DEV)).cpu(), pf[None], atol=2e-5,
                                   rtol=1e-4)


@pytest.mark.gpu
def test_rigid_collinear_translation_hip():
    """reference test_rigid_0: four collinear points translated along one axis (rank-1 covariance)."""
    from keymorph_amd.keypoint_aligners import RigidKeypointAligner
    a = torch.tensor([[0, 0, 0], [0, 0, 0.1], [0, 0, 0.2], [0, 0, 0.3]])[None].float()
    b = a + torch.tensor([0, 0, 0.1])
    al = RigidKeypointAligner(a.to(DEV), b.to(DEV), dim=3)
    torch.testing.assert_close(al.transform_matrix.cpu(), _expect(torch.eye(3), torch.tensor([0, 0, 0.1])), atol=1e-5,
                               rtol=1e-5)


def test_coordinate_conversions():
    from keymorph_amd import utils
    p = torch.tensor([[[-1.0, -1, -1], [1, 1, 1], [0, 0, 0]]])
    v = utils.convert_points_norm2voxel(p, (10, 20, 30))
    torch.testing.assert_close(v, torch.tensor([[[-0.5, -0.5, -0.5], [9.5, 19.5, 29.5], [4.5, 9.5, 14.5]]]))
    torch.testing.assert_close(utils.convert_points_voxel2norm(v, (10, 20, 30)), p)
    aff = torch.eye(4)[None].clone()
    aff[0, :3, :3] *= 2
    aff[0, :3, 3] = torch.tensor([1.0, 2, 3])
    r = utils.convert_points_norm2real(p, aff, (10, 20, 30))
    torch.testing.assert_close(r, v * 2 + torch.tensor([1.0, 2, 3]))
    torch.testing.assert_close(utils.convert_points_real2norm(r, aff, (10, 20, 30)), p, atol=1e-6, rtol=1e-6)
    flow = torch.zeros(1, 2, 2, 2, 3)
    out = utils.convert_flow_voxel2norm(flow.clone(), (4, 4, 4))
    torch.testing.assert_close(out, torch.full_like(flow, 2 * 0.5 / 4 - 1))
    assert utils.str_or_float("0.5") == 0.5 and utils.str_or_float("uniform") == "uniform"
    # one_hot / one_hot_subsampled_pair run on the GPU (HIP kernels): tests/test_parity_r2_gpu.py
