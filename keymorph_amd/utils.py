"""Warp + coordinate utilities with the reference's names (keymorph/utils.py).

``align_img`` is the HIP sampler; the ``convert_points_*`` family works on (bs, K, dim)
point sets (microseconds, only used with --align_keypoints_in_real_world_coords) and stays
plain tensor algebra, as SURVEY.md section 8 row a16 prescribes.
"""
import numpy as np
import torch

from . import ops


def str_or_float(x):
    try:
        return float(x)
    except ValueError:
        return x


def align_img(grid, x, mode="bilinear"):
    """F.grid_sample(x, grid, mode, padding_mode='border', align_corners=False): utils.py:14-21."""
    if x.dim() == 4:  # (N,C,H,W) + (N,Ho,Wo,2): a depth-1 volume sampled at z = 0
        g3 = torch.cat([grid, torch.zeros_like(grid[..., :1])], dim=-1).unsqueeze(1)
        return ops.grid_sample3d(x.unsqueeze(2), g3, mode).squeeze(2)
    return ops.grid_sample3d(x, grid, mode)


def uniform_norm_grid(grid_shape, dim=3):
    """Identity grid in ij order, linspace(-1, 1, n) per axis: utils.py:387-398 (host tensor;
    the HIP grid generators build it from the voxel index instead of reading it)."""
    axes = [torch.linspace(-1, 1, int(n)) for n in grid_shape[2:2 + dim]]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).float()


def rescale_intensity(array, out_range=(0, 1), percentiles=(0, 100)):
    """utils.py:78-94 (in-place min-max rescale)."""
    if isinstance(array, torch.Tensor):
        array = array.float()
    if percentiles != (0, 100):
        cutoff = np.percentile(array, percentiles)
        np.clip(array, *cutoff, out=array)
    in_min = array.min()
    in_range = array.max() - in_min
    array -= in_min
    array /= in_range
    array *= out_range[1] - out_range[0]
    array += out_range[0]
    return array


_MAX_LABEL = 1 << 16     # label ids handled by the presence bitmap (FreeSurfer / BrainMorph label maps are < 2^16)



def sample_valid_coordinates(x, num_points, dim, point_space="norm", indexing="xy"):
    """`num_points` random foreground voxels of x (1,1,H,W) / (1,1,D,H,W) as (1, num_points, dim) coordinates -- what
    scripts/run.py:528-548 draws as pre-training reference keypoints (keymorph/utils.py:97-162).  Same draws from numpy's
    global generator as the reference (one np.random.randint per axis and attempt, slowest axis first; accepted when the
    voxel is foreground: > 0 in 2-D, > 0.1 in 3-D), same outputs: "norm" -> index / size per axis as float32, anything
    else -> the integer indices (int64); fastest axis first ("xy"), reversed for indexing == "ij".  The volume is read
    on the host once instead of building a one-hot volume per attempt."""
    import numpy as np
    if dim not in (2, 3):
        raise NotImplementedError
    if x.dim() != dim + 2 or x.shape[0] != 1 or x.shape[1] != 1:
        raise ValueError(f"sample_valid_coordinates expects a (1, 1, ...) volume with {dim} spatial axes, got {tuple(x.shape)}")
    sizes = [int(n) for n in x.shape[2:]]
    fg = (x[0, 0] > (0 if dim == 2 else 1e-1)).cpu().numpy()
    if not fg.any():
        raise ValueError("sample_valid_coordinates: the volume has no foreground voxel to sample")
    rows = []
    for _ in range(num_points):
        while True:
            idx = tuple(int(np.random.randint(0, n)) for n in sizes)
            if fg[idx]:
                break
        rev = idx[::-1]
        rows.append([i / n for i, n in zip(rev, sizes[::-1])] if point_space == "norm" else list(rev))
    coords = torch.tensor(rows).view(1, num_points, dim)
    return coords.flip(-1) if indexing == "ij" else coords


def _seg_on_gpu(seg):
    """The reference's loops call the encoders on the loader's CPU tensors and move the result to the device
    afterwards (scripts/train.py:54-79); here the label map goes to the current GPU first and the encoding is
    returned there (the later ``.float().to(device)`` is then a no-op)."""
    if not seg.is_cuda:
        seg = seg.to(torch.device("cuda", torch.cuda.current_device()))
    elif seg.device.index != torch.cuda.current_device():
        # same rule as ops._prep: the C ABI launches on the current device's stream and never calls hipSetDevice
        from ._lib import KeymorphHipError
        raise KeymorphHipError(f"label map is on {seg.device} but the current device is "
                               f"cuda:{torch.cuda.current_device()}: call torch.cuda.set_device(...) first")
    if seg.dtype != torch.int64:
        seg = seg.long()
    return seg.contiguous()


def _labels_present(seg):
    """Sorted label ids occurring in an int64 label map on the GPU (np.unique of the reference, utils.py:210-211)."""
    from . import _lib
    lib = _lib.load()
    flags = torch.zeros(_MAX_LABEL + 1, dtype=torch.int32, device=seg.device)
    ops.check(lib.kmh_label_presence(ops._p(seg), seg.numel(), _MAX_LABEL, ops._p(flags), ops._stream()),
              "kmh_label_presence")
    flags = flags.cpu().numpy()
    if flags[_MAX_LABEL]:
        raise ValueError(f"label ids must lie in [0, {_MAX_LABEL}) (F.one_hot also rejects negative labels)")
    return np.nonzero(flags[:_MAX_LABEL])[0].astype(np.int64)


def _encode(seg, labels, as_int64):
    """out[n, c] = (seg[n] == labels[c]); any number of channels (the kernel takes 256 labels per launch, so e.g. a
    FreeSurfer aparc+aseg map with ids up to 2035 is encoded in 8 launches into channel slices of one tensor); no
    shared label at all gives the reference's empty (N, 0, ...) tensor."""
    from . import _lib
    lib = _lib.load()
    n = seg.shape[0]
    v = seg.numel() // max(n, 1)
    C = len(labels)
    dt = torch.int64 if as_int64 else torch.float32
    if C == 0 or seg.numel() == 0:
        return torch.zeros((n, C, *seg.shape[2:]), dtype=dt, device=seg.device)
    lab = torch.as_tensor(np.asarray(labels, dtype=np.int64), device=seg.device)
    out = torch.empty((n, C, *seg.shape[2:]), dtype=dt, device=seg.device)
    if C <= 256:
        ops.check(lib.kmh_one_hot_select(ops._p(seg), n, v, ops._p(lab), C, ops._p(out), int(as_int64), ops._stream()),
                  "kmh_one_hot_select")
        return out
    for i in range(n):               # channel slices of one sample are contiguous
        for c0 in range(0, C, 256):
            c1 = min(C, c0 + 256)
            ops.check(lib.kmh_one_hot_select(ops._p(seg[i]), 1, v, ops._p(lab[c0:c1]), c1 - c0, ops._p(out[i, c0:c1]),
                                             int(as_int64), ops._stream()), "kmh_one_hot_select")
    return out


def one_hot(seg):
    """(N,1,D,H,W) integer labels -> (N,C,D,H,W) int64, C = largest label + 1: utils.py:200-205."""
    assert seg.shape[1] == 1, "expected a (N, 1, ...) label map"
    seg = _seg_on_gpu(seg)
    present = _labels_present(seg)
    return _encode(seg, np.arange(int(present.max()) + 1), True)


def one_hot_subsampled_pair(seg1, seg2, subsample_num=14):
    """utils.py:208-240: float32 one-hot over (an np.random.choice subset of) the labels both maps contain; the
    draw is the reference's (same generator, same arguments), so a seeded script selects the same channels."""
    seg1, seg2 = _seg_on_gpu(seg1), _seg_on_gpu(seg2)
    shared = np.intersect1d(_labels_present(seg1), _labels_present(seg2), assume_unique=True)
    chosen = np.random.choice(shared, subsample_num, replace=False) if len(shared) > subsample_num else shared
    return _encode(seg1, chosen, False), _encode(seg2, chosen, False)


def _homog(points):
    return torch.cat([points, torch.ones_like(points[..., :1])], dim=2)


def convert_points_norm2voxel(points, grid_sizes):
    """[-1,1] -> voxel: (p+1)*S/2 - 0.5 (utils.py:243-259)."""
    grid_sizes = torch.as_tensor(grid_sizes).to(points.device)
    assert grid_sizes.shape[-1] == points.shape[-1], "Dimensions don't match"
    return ((points + 1) * grid_sizes) / 2 - 0.5


def convert_points_voxel2norm(points, grid_sizes):
    grid_sizes = torch.as_tensor(grid_sizes).to(points.device)
    assert grid_sizes.shape[-1] == points.shape[-1], "Dimensions don't match"
    return (2 * (points + 0.5) / grid_sizes) - 1


def convert_points_voxel2real(points, affine):
    return torch.bmm(affine, _homog(points).permute(0, 2, 1)).permute(0, 2, 1)[:, :, :-1]


def convert_points_real2voxel(points, affine):
    return torch.bmm(torch.inverse(affine), _homog(points).permute(0, 2, 1)).permute(0, 2, 1)[:, :, :-1]


def convert_points_norm2real(points, affine_matrices, voxel_sizes):
    return convert_points_voxel2real(convert_points_norm2voxel(points, voxel_sizes), affine_matrices)


def convert_points_real2norm(real_world_points, affine_matrices, voxel_sizes):
    return convert_points_voxel2norm(convert_points_real2voxel(real_world_points, affine_matrices), voxel_sizes)


def convert_flow_voxel2norm(flow, dim_sizes):
    """utils.py:357-371 (in place)."""
    for i, dim_size in enumerate(dim_sizes):
        flow[..., i] = 2 * (flow[..., i] + 0.5) / dim_size - 1
    return flow


# Reference helpers outside the path (voxel-unit displacement fields of the baselines, SynthSeg label tables, 2-D sampling):
# present as names, NotImplementedError when called.
from ._absent import absent_function as _absent_function   # noqa: E402

displacement2pytorchflow = _absent_function("displacement2pytorchflow", "keymorph/utils.py:24")
pytorchflow2displacement = _absent_function("pytorchflow2displacement", "keymorph/utils.py:56")
one_hot_eval_synthseg = _absent_function("one_hot_eval_synthseg", "keymorph/utils.py:164")
uniform_voxel_grid = _absent_function("uniform_voxel_grid", "keymorph/utils.py:373")


def sample_valid_coordinates_2d(x, num_points, point_space="norm"):
    """keymorph/utils.py:117-137: the "xy" draw of `sample_valid_coordinates` for a (1, 1, H, W) image."""
    return sample_valid_coordinates(x, num_points, 2, point_space=point_space, indexing="xy")


def sample_valid_coordinates_3d(x, num_points, point_space="norm"):
    """keymorph/utils.py:139-162: the "xy" draw of `sample_valid_coordinates` for a (1, 1, D, H, W) volume."""
    return sample_valid_coordinates(x, num_points, 3, point_space=point_space, indexing="xy")
