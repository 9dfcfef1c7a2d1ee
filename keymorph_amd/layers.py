"""Keypoint layers (keymorph/layers.py:30-134)."""
import torch.nn as nn

from . import ops


class CenterOfMass3d(nn.Module):
    """ReLU -> center of mass per channel in [-1,1]^3 (keymorph/layers.py:78-134).

    indexing='xy' returns (x, y, z); 'ij' returns (z, y, x) (what KeyMorph uses)."""

    def __init__(self, indexing="xy") -> None:
        super().__init__()
        assert indexing in ["xy", "ij"]
        self.indexing = indexing

    def forward(self, vol):
        pts = ops.com3d(vol)
        return pts.flip(-1) if self.indexing == "xy" else pts


class CenterOfMass2d(nn.Module):
    """2-D variant (keymorph/layers.py:30-75): a (n, K, H, W) map is a depth-1 volume."""

    def __init__(self, indexing="xy") -> None:
        super().__init__()
        assert indexing in ["xy", "ij"]
        self.indexing = indexing

    def forward(self, img):
        pts = ops.com3d(img.unsqueeze(2))[..., 1:]  # drop the degenerate z coordinate
        return pts.flip(-1) if self.indexing == "xy" else pts


# keymorph/layers.py also defines ConvBlock (here: keymorph_amd/net.py, next to its only user; re-exported) and the LinearRegressor
# keypoint layers, which are broken upstream (layers.py:6-27 read an undefined self.num_keypoints): names only.
from ._absent import absent_class as _absent_class   # noqa: E402

LinearRegressor2d = _absent_class("LinearRegressor2d", "keymorph/layers.py:6", nn.Module)
LinearRegressor3d = _absent_class("LinearRegressor3d", "keymorph/layers.py:18", nn.Module)

from .net import ConvBlock   # noqa: E402,F401
