"""Minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer -- what the reference gets from nibabel /
torchio for example_data* (`scripts/script_utils.py`, `dataset/*`): the voxel array in (x, y, z) index order and the
4x4 voxel-to-world affine (sform if present, else qform, else the pixdim diagonal), with scl_slope / scl_inter
applied.  Header layout: NIfTI-1.1 specification (348-byte header, data at vox_offset)."""
import gzip
import struct

import numpy as np

_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _quat_affine(b, c, d, qoff, pixdim):
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    qfac = -1.0 if pixdim[0] < 0 else 1.0
    A = np.eye(4)
    A[:3, :3] = R * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac])
    A[:3, 3] = qoff
    return A


def read_nifti(path, dtype=np.float32):
    """-> (array (X, Y, Z[, T]) of `dtype` (None = on-disk dtype, unscaled), affine (4, 4) float64)."""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 348:
        raise ValueError(f"{path}: not a NIfTI-1 file (shorter than the 348-byte header)")
    end = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[:4])[0] != 348 or raw[344:347] not in (b"n+1", b"ni1"):
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr / magic)")
    if raw[344:347] == b"ni1":
        raise ValueError(f"{path}: header/image pairs (.hdr + .img) are not supported")
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, = struct.unpack(end + "h", raw[70:72])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform, sform = struct.unpack(end + "2h", raw[252:256])
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {datatype}")
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    while len(shape) > 3 and shape[-1] == 1:
        shape = shape[:-1]
    n = int(np.prod(shape))
    dt = np.dtype(end + _DTYPES[datatype])
    off = int(vox_offset)
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order="F")
    if sform > 0:
        A = np.eye(4)
        A[:3, :] = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    elif qform > 0:
        b, c, d, qx, qy, qz = struct.unpack(end + "6f", raw[256:280])
        A = _quat_affine(b, c, d, (qx, qy, qz), pixdim)
    else:
        A = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0]).astype(np.float64)
    if dtype is None:
        return np.ascontiguousarray(arr), A
    out = arr.astype(dtype)
    # NIfTI-1: scaling applies only for a finite, non-zero scl_slope; nibabel writes NaN into both fields for
    # float images ("no scaling") and a non-finite scl_inter counts as 0.
    if np.isfinite(slope) and slope != 0.0:
        inter = inter if np.isfinite(inter) else 0.0
        if slope != 1.0 or inter != 0.0:
            out = out * dtype(slope) + dtype(inter)
    return np.ascontiguousarray(out), A


def write_nifti(path, array, affine=None):
    """(X, Y, Z) array + 4x4 affine -> NIfTI-1 single file (sform only); used for fixtures and for saving results."""
    array = np.asarray(array)
    key = array.dtype.str[1:]
    if key not in _CODES:
        raise ValueError(f"unsupported dtype {array.dtype}")
    affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [array.ndim] + list(array.shape) + [1] * (7 - array.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<h", hdr, 70, _CODES[key])
    struct.pack_into("<h", hdr, 72, array.dtype.itemsize * 8)
    vox = np.sqrt((affine[:3, :3] ** 2).sum(0))
    struct.pack_into("<8f", hdr, 76, 1.0, *vox, 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    struct.pack_into("<2h", hdr, 252, 0, 1)
    struct.pack_into("<12f", hdr, 280, *affine[:3, :].reshape(-1))
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(np.asfortranarray(array.astype(array.dtype.newbyteorder("<"))).tobytes(order="F"))
