"""CPU: host-side logic that needs no GPU -- module surface / state_dict compatibility with the reference,
transform-type parsing, sharding, and the N>1 data-parallel path on gloo with world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keymorph_amd import parallel
from keymorph_amd.model import KeyMorph
from keymorph_amd.unet3d.model import TruncatedUNet3D, UNet3D
from tests.util import unet_shapes


def test_state_dict_keys_match_reference_layout():
    for K, f, trunc in ((16, 8, 1), (512, 32, 1), (8, 8, 0)):
        cls = TruncatedUNet3D if trunc else UNet3D
        args = (1, K, trunc) if trunc else (1, K)
        net = cls(*args, final_sigmoid=False, f_maps=f, layer_order="gcr", num_groups=8, num_levels=4,
                  is_segmentation=False, conv_padding=1)
        got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert got == {k: tuple(v) for k, v in unet_shapes(K, f, trunc=trunc or None).items()}
    # TruncatedUNet3D(512 kp, f_maps 32) has 4 003 666 parameters
    net = TruncatedUNet3D(1, 512, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False, conv_padding=1)
    assert sum(p.numel() for p in net.parameters()) == 4003666


def test_dataparallel_prefixed_checkpoint_loads():
    """scripts/run.py:390 wraps the backbone in nn.DataParallel, so checkpoints carry 'module.' keys
    (scripts/script_utils.py:59-81); the backbone must be a plain nn.Module that survives that."""
    net = TruncatedUNet3D(1, 16, 1, final_sigmoid=False, f_maps=8, layer_order="gcr", num_groups=8, num_levels=4,
                          is_segmentation=False)
    wrapped = torch.nn.DataParallel(net)
    sd = wrapped.state_dict()
    assert all(k.startswith("module.") for k in sd)
    km = KeyMorph(wrapped, 16, 3)
    km.backbone.load_state_dict(sd, strict=True)
    assert hasattr(km, "backbone") and len(list(km.parameters())) == len(sd)


def test_transform_type_parsing_and_lambda():
    net = TruncatedUNet3D(1, 16, 1, final_sigmoid=False, f_maps=8, is_segmentation=False)
    km = KeyMorph(net, 16, 3, max_rand_tps_lmbda=10)
    assert km.is_supported_transform_type("affine") and km.is_supported_transform_type("rigid")
    assert km.is_supported_transform_type("tps_0") and km.is_supported_transform_type("tps_loguniform")
    assert not km.is_supported_transform_type("bspline") and not km.is_supported_transform_type("tps")
    assert torch.equal(km._convert_tps_lmbda(3, 0.5), torch.tensor([0.5, 0.5, 0.5]))
    u = km._convert_tps_lmbda(100, "uniform")
    assert u.shape == (100,) and float(u.min()) >= 0 and float(u.max()) <= 10
    lu = km._convert_tps_lmbda(50, "loguniform")
    assert lu.shape == (50,) and float(lu.min()) >= 1e-6 and float(lu.max()) <= 10
    kv = KeyMorph(net, 16, 3, weight_keypoints="variance")       # model.py:68-72: same parameter names as upstream
    assert kv.scales.shape == (16,) and kv.biases.shape == (16,) and "scales" in kv.state_dict()
    assert KeyMorph(net, 16, 3, weight_keypoints="power").weight_keypoints == "power"
    with pytest.raises(AssertionError):
        KeyMorph(net, 16, 3, weight_keypoints="entropy")
    with pytest.raises(NotImplementedError):
        KeyMorph(net, 16, 3, keypoint_layer="linear")


def test_unsupported_unet_config_is_loud():
    with pytest.raises(NotImplementedError):
        UNet3D(1, 4, layer_order="cr")


def test_shard_indices_partition():
    for n in (1, 7, 8, 16, 17):
        for w in (1, 2, 3, 8):
            parts = [parallel.shard_indices(n, r, w) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, l, w = parallel.init_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    torch.manual_seed(100 + rank)          # different init per rank -> broadcast must equalise
    model = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    flat = parallel.FlatParams(model.parameters())
    flat.broadcast(0)
    # parameters are views into the flat buffer
    assert model[0].weight.data_ptr() == flat.flat.data_ptr()
    flat.zero_grad()
    x = torch.full((2, 5), float(rank + 1))
    model(x).sum().backward()              # accumulates into the flat grad views
    local = flat.grad.clone()
    scale = flat.allreduce_grads()
    pts = parallel.allgather_points(torch.full((1, 4, 3), float(rank)))
    # groupwise sharding (BASELINE config 5): 5 subjects over 2 ranks -> blocks of 3 and 2, gathered in subject order
    mine = parallel.shard_indices(5, rank, world)
    group = parallel.gather_group_points(torch.stack([torch.full((4, 3), float(i)) for i in mine]), 5)
    out[rank] = dict(flat=flat.flat.clone(), local=local, summed=flat.grad.clone(), scale=scale, pts=pts, mine=mine,
                     group=group)
    dist.destroy_process_group()


def test_data_parallel_gloo_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["flat"], b["flat"])                       # broadcast equalised the weights
    assert torch.allclose(a["summed"], a["local"] + b["local"])   # sum all-reduce of the flat bucket
    assert torch.equal(a["summed"], b["summed"])
    assert a["scale"] == 0.5 and b["scale"] == 0.5
    assert a["pts"].shape == (2, 4, 3) and torch.equal(a["pts"][:, 0, 0], torch.tensor([0.0, 1.0]))
    assert torch.equal(a["pts"], b["pts"])
    assert a["mine"] == [0, 1, 2] and b["mine"] == [3, 4]
    assert a["group"].shape == (5, 4, 3) and torch.equal(a["group"][:, 0, 0], torch.arange(5.0))
    assert torch.equal(a["group"], b["group"])


def test_augmentation_surface_and_draw_order():
    """augmentation.py mirrors the reference's surface; random parameters come from torch's global CPU generator in
    the reference's order (scale, offset, theta, shear) -- no GPU needed to check that."""
    import torch
    from keymorph_amd import augmentation as A
    for name in ("AffineDeformation2d", "AffineDeformation3d", "random_affine_augment", "affine_augment",
                 "random_affine_augment_pair"):
        assert hasattr(A, name)
    torch.manual_seed(3)
    got = A._draw(torch.zeros(1, 1, 2, 2, 2), ((0.9, 1.1), (-0.2, 0.2), (-1.0, 1.0), (-0.1, 0.1)))
    torch.manual_seed(3)
    want = (torch.FloatTensor(1, 3).uniform_(0.9, 1.1), torch.FloatTensor(1, 3).uniform_(-0.2, 0.2),
            torch.FloatTensor(1, 3).uniform_(-1.0, 1.0), torch.FloatTensor(1, 6).uniform_(-0.1, 0.1))
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    import pytest
    with pytest.raises(NotImplementedError):
        A.random_affine_augment(torch.zeros(1, 1, 4, 4))          # 2-D path is out of scope
    with pytest.raises(NotImplementedError):
        A.AffineDeformation2d()


def test_nifti_roundtrip_and_header_variants(tmp_path):
    import gzip, struct
    import numpy as np
    from keymorph_amd.io import read_nifti, write_nifti
    rng = np.random.default_rng(0)
    A = np.array([[0.0, -1.2, 0.0, 90.0], [1.1, 0.0, 0.0, -120.0], [0.0, 0.0, 1.3, -70.0], [0, 0, 0, 1.0]])
    for dt, name in ((np.int32, "a.nii.gz"), (np.float64, "b.nii"), (np.uint8, "c.nii.gz")):
        vol = (rng.random((5, 6, 7)) * 100).astype(dt)
        write_nifti(tmp_path / name, vol, A)
        got, aff = read_nifti(tmp_path / name, dtype=None)
        assert got.dtype == dt and np.array_equal(got, vol)
        np.testing.assert_allclose(aff, A, atol=1e-5)
        assert read_nifti(tmp_path / name)[0].dtype == np.float32
    # x is the fastest axis on disk (Fortran order)
    raw = gzip.open(tmp_path / "a.nii.gz").read()
    first = np.frombuffer(raw, "<i4", count=5, offset=352)
    assert np.array_equal(first, read_nifti(tmp_path / "a.nii.gz", dtype=None)[0][:, 0, 0])
    # big-endian header, qform only, scl_slope / scl_inter
    vol = np.arange(24, dtype=">i2").reshape((2, 3, 4), order="F")
    hdr = bytearray(352)
    struct.pack_into(">i", hdr, 0, 348)
    struct.pack_into(">8h", hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into(">h", hdr, 70, 4)
    struct.pack_into(">h", hdr, 72, 16)
    struct.pack_into(">8f", hdr, 76, -1.0, 2.0, 3.0, 4.0, 1, 1, 1, 1)
    struct.pack_into(">3f", hdr, 108, 352.0, 0.5, 10.0)
    struct.pack_into(">2h", hdr, 252, 1, 0)
    struct.pack_into(">6f", hdr, 256, 0.0, 0.0, 0.0, 7.0, 8.0, 9.0)
    hdr[344:348] = b"n+1\0"
    (tmp_path / "be.nii").write_bytes(bytes(hdr) + vol.tobytes(order="F"))
    got, aff = read_nifti(tmp_path / "be.nii")
    np.testing.assert_allclose(got, vol.astype(np.float32) * 0.5 + 10.0)
    np.testing.assert_allclose(aff, np.array([[2.0, 0, 0, 7], [0, 3.0, 0, 8], [0, 0, -4.0, 9], [0, 0, 0, 1]]))
    import pytest
    (tmp_path / "bad.nii").write_bytes(b"\0" * 400)
    with pytest.raises(ValueError):
        read_nifti(tmp_path / "bad.nii")


def test_checkpoint_format_and_optimizer_state_interchange(tmp_path):
    """Reference-format checkpoints (run.py:588-602) round-trip, with '.backbone' / 'module.' key variants, and the
    Adam state is interchangeable between torch.optim.Adam and FusedAdam (state_dict layout only: no GPU needed)."""
    import torch
    from keymorph_amd.io import load_checkpoint, save_checkpoint
    from keymorph_amd.model import KeyMorph
    from keymorph_amd.parallel import FlatParams, FusedAdam
    from keymorph_amd.unet3d.model import TruncatedUNet3D

    def make():
        net = TruncatedUNet3D(1, 8, 1, final_sigmoid=False, f_maps=8, layer_order="gcr", num_groups=8, num_levels=3,
                              is_segmentation=False, conv_padding=1)
        return KeyMorph(net, 8, 3)

    torch.manual_seed(0)
    a, b = make(), make()
    opt = torch.optim.Adam(a.parameters(), lr=3e-4)
    for p in a.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    save_checkpoint(tmp_path / "ck.pth.tar", a, opt, epoch=7, args={"lr": 3e-4})
    flat = FlatParams(b.parameters())
    fused = FusedAdam(flat)
    state, _, fused = load_checkpoint(tmp_path / "ck.pth.tar", b, fused)
    assert state["epoch"] == 7 and fused.t == 1 and fused.lr == 3e-4
    for (k, p), q in zip(a.backbone.state_dict().items(), b.backbone.state_dict().values()):
        assert torch.equal(p, q), k
    o = 0
    for i, p in enumerate(a.parameters()):
        k = p.numel()
        assert torch.equal(fused.m[o:o + k], opt.state[p]["exp_avg"].reshape(-1))
        assert torch.equal(fused.v[o:o + k], opt.state[p]["exp_avg_sq"].reshape(-1))
        o += k
    # and back into torch.optim.Adam
    opt2 = torch.optim.Adam(make().parameters())
    opt2.load_state_dict(fused.state_dict())
    assert opt2.param_groups[0]["lr"] == 3e-4
    # key variants of published checkpoints
    sd = a.backbone.state_dict()
    for rename in (lambda k: "module." + k, lambda k: "backbone." + k, lambda k: "module.backbone." + k):
        torch.save({"epoch": 1, "state_dict": {rename(k): v for k, v in sd.items()}, "optimizer": opt.state_dict()},
                   tmp_path / "v.pth.tar")
        c = make()
        load_checkpoint(tmp_path / "v.pth.tar", c)
        assert all(torch.equal(p, q) for p, q in zip(sd.values(), c.backbone.state_dict().values()))


def test_pair_loader_shim(tmp_path):
    import numpy as np
    import torch
    from keymorph_amd.io import AFFINE, DATA, PairLoader, make_subject, write_nifti
    rng = np.random.default_rng(1)
    subs = []
    for i in range(3):
        vol = (rng.random((6, 7, 8)) * 50 + 10).astype(np.float32)
        write_nifti(tmp_path / f"s{i}.nii.gz", vol, np.diag([1.0, 2.0, 3.0, 1.0]))
        write_nifti(tmp_path / f"l{i}.nii.gz", (vol > 30).astype(np.int32))
        subs.append(make_subject(tmp_path / f"s{i}.nii.gz", seg=tmp_path / f"l{i}.nii.gz", modality="t1"))
    s = subs[0]
    assert s["img"][DATA].shape == (1, 1, 6, 7, 8) and s["img"][DATA].dtype == torch.float32
    assert float(s["img"][DATA].min()) == 0.0 and float(s["img"][DATA].max()) == 1.0
    assert s["img"][AFFINE].shape == (1, 4, 4) and float(s["img"][AFFINE][0, 1, 1]) == 2.0
    assert s["seg"][DATA].dtype == torch.int64 and s["img"]["path"].endswith("s0.nii.gz")
    loader = PairLoader(subs, steps=5, seed=3)
    first = [(f["img"]["path"], m["img"]["path"]) for f, m in loader]
    assert len(first) == 5 and all(f != m for f, m in first)
    assert first == [(f["img"]["path"], m["img"]["path"]) for f, m in PairLoader(subs, steps=5, seed=3)]


def test_nifti_reader_on_reference_example_data():
    """build container only (the GPU box has no /root/reference): example_data_half's label maps are 256^3 float64
    with 14 labels and an LPS-flipped identity affine."""
    import glob
    import numpy as np
    import pytest
    files = sorted(glob.glob("/root/reference/example_data_half/seg_m/*.nii.gz"))
    if not files:
        pytest.skip("reference example data not present")
    from keymorph_amd.io import read_nifti
    a, A = read_nifti(files[0], dtype=None)
    assert a.shape == (256, 256, 256) and a.dtype == np.float64
    assert np.array_equal(np.unique(a), np.arange(14.0))
    np.testing.assert_allclose(A, np.diag([-1.0, -1.0, 1.0, 1.0]))


def test_f16x3_arithmetic_claim_emulated():
    """The arithmetic behind the default convolution mode, emulated in numpy against fp64: fp16 hi+lo with 3 products
    and power-of-two range scaling is as accurate as the 6-product bf16 split and better than sequential fp32, for
    O(1) activations AND for tiny gradients -- while WITHOUT the scaling small gradients lose 3 digits."""
    import importlib.util, os
    import numpy as np
    spec = importlib.util.spec_from_file_location("f16x3_emulation", os.path.join(os.path.dirname(__file__), "..", "tools",
                                                                                "f16x3_emulation.py"))
    emu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emu)
    rng = np.random.default_rng(0)
    M, K = 256, 27 * 16
    w = rng.uniform(-0.05, 0.05, (M, K))
    for x in (rng.standard_normal((M, K)), rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, K))) * 1e-5):
        e = emu.errors(x, w)
        assert e["f16x3"] < 4e-7 and e["f16x3"] < 1.5 * e["bf16x6"] + 1e-8 and e["f16x3"] < e["fp32"], e
    assert e["f16x3_unscaled"] > 100 * e["f16x3"], e        # the tiny-gradient case: scaling is what makes it work


def test_use_amp_is_a_stored_flag_without_warning():
    """keymorph/model.py:176-191 autocasts the extractor under use_amp; here the flag selects the one-product fp16 arithmetic
    of the backbone's matrix kernels at every get_keypoints() call on a GPU tensor (tests/test_train_step_gpu.py); constructing
    the model does not touch the library and does not warn"""
    import warnings
    from keymorph_amd.model import KeyMorph
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        km = KeyMorph(torch.nn.Identity(), 4, 3, use_amp=True)
    assert km.use_amp is True and KeyMorph(torch.nn.Identity(), 4, 3).use_amp is False


def test_flat_params_gathers_gradients():
    """FlatParams.zero_grad() drops the gradients, autograd assigns fresh tensors, and the first access to .grad gathers
    them into the flat buffer (one multi-tensor copy) and re-points every p.grad at its slice: same observable
    behaviour as accumulating into views -- incl. accumulation over two backward passes and parameters that got none."""
    import torch.nn as nn
    torch.manual_seed(0)
    lin1, lin2, unused = nn.Linear(3, 4), nn.Linear(4, 2), nn.Linear(2, 2)
    params = list(lin1.parameters()) + list(lin2.parameters()) + list(unused.parameters())
    flat = parallel.FlatParams(params)
    x = torch.randn(5, 3)

    def loss():
        return lin2(torch.tanh(lin1(x))).pow(2).sum()

    ref = torch.autograd.grad(loss(), list(lin1.parameters()) + list(lin2.parameters()))
    flat.zero_grad()
    assert all(p.grad is None for p in params)
    loss().backward()
    g = flat.grad                                          # gathers
    o = 0
    for p, r in zip(params[:4], ref):
        assert torch.allclose(g[o:o + p.numel()].view_as(p), r) and p.grad.data_ptr() == g[o:o + p.numel()].data_ptr()
        o += p.numel()
    assert float(g[o:].abs().max()) == 0.0                 # the unused module's slices are zero
    loss().backward()                                      # no zero_grad: accumulates into the flat views
    assert torch.allclose(flat.grad[:params[0].numel()].view_as(params[0]), 2 * ref[0])
    assert flat.allreduce_grads() == 1.0
    flat.zero_grad()
    loss().backward()
    assert torch.allclose(flat.grad[:params[0].numel()].view_as(params[0]), ref[0])


def test_sample_valid_coordinates_golden():
    """keymorph/utils.py:97-162 (scripts/run.py:528-548 draws pre-training keypoints with it): the same points as the
    reference for the same numpy seed, the same dtype, and the global generator left in the same state."""
    from keymorph_amd.utils import sample_valid_coordinates
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "valid_coords_small.npz"))
    x3, x2 = torch.tensor(d["x3"]), torch.tensor(d["x2"])
    for tag, x, dim, space, indexing in (("a", x3, 3, "norm", "xy"), ("b", x3, 3, "norm", "ij"), ("c", x3, 3, "voxel", "xy"),
                                         ("d", x2, 2, "norm", "xy"), ("e", x2, 2, "voxel", "ij")):
        np.random.seed(int(d[f"{tag}::seed"][0]))
        pts = sample_valid_coordinates(x, 6, dim, point_space=space, indexing=indexing)
        assert str(pts.dtype) == str(d[f"{tag}::dtype"]) and tuple(pts.shape) == (1, 6, dim)
        np.testing.assert_array_equal(pts.numpy().astype(np.float64), d[f"{tag}::points"])
        assert int(np.random.randint(0, 1 << 30)) == int(d[f"{tag}::after"][0])
        # every point is a foreground voxel of x
        p = pts.flip(-1) if indexing == "xy" else pts                         # -> slowest axis first
        sizes = torch.tensor(x.shape[2:], dtype=torch.float64)
        idx = (p[0].double() * sizes).round().long() if space == "norm" else p[0].long()
        vals = x[0, 0][tuple(idx[:, k] for k in range(dim))]
        assert bool((vals > (0 if dim == 2 else 0.1)).all())
    with pytest.raises(NotImplementedError):
        sample_valid_coordinates(x3, 2, 4)
    with pytest.raises(ValueError):
        sample_valid_coordinates(torch.zeros(1, 1, 3, 3, 3), 2, 3)


# ---------------------------------------------------------------------------------------------------------------------
# bench.py started the way the driver starts `--gpus 1` (plain `python bench.py --gpus N`, no torch.distributed.run) must
# start N ranks by itself (replaces nn.DataParallel, scripts/run.py:390) -- or refuse; never print `n_gpus: 1` for N > 1.
# KEYMORPH_BENCH_LAUNCH_CHECK=1 stops every rank after the process group's first collective (no HIP work on a CPU box);
# the same launch WITH the HIP step runs on the GPU box in tests/test_multirank_gpu.py.
# ---------------------------------------------------------------------------------------------------------------------
def _bench(args, env_extra, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_self_launch_two_ranks_without_a_launcher():
    r, line = _bench(["--gpus", "2", "--size", "24", "--keypoints", "16", "--steps", "1"],
                     {"KEYMORPH_BENCH_LAUNCH_CHECK": "1", "KEYMORPH_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo" and line["self_launched"]
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1        # ONE line, from rank 0


def test_bench_refuses_more_ranks_than_devices():
    r, line = _bench(["--gpus", "8"], {})
    assert r.returncode == 2 and line is None
    assert "needs 8 visible devices" in r.stderr


def test_bench_rejects_world_size_mismatch():
    r, line = _bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0", "KEYMORPH_BENCH_LAUNCH_CHECK": "1"})
    assert r.returncode != 0 and line is None and "WORLD_SIZE=4 but --gpus 2" in r.stderr


def test_bench_rccl_transport_summary_and_sensor_fallback(tmp_path):
    """bench.py's self-explaining pieces for the first multi-GPU run: the RCCL debug log of rank 0's communicator set-up is
    folded into counts of its `via <transport>` channel lines (P2P/IPC = xGMI or PCIe peer access, SHM, NET/...), a missing
    log says so, and the hwmon reader returns (None, None) instead of raising where there is no amdgpu device."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    log = tmp_path / "rccl.log"
    log.write_text("h:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\n"
                   "h:1:1 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC\n"
                   "h:1:1 [0] NCCL INFO Channel 00/0 : 7[7] -> 0[0] [receive] via NET/Socket/0\n"
                   "h:1:1 [0] NCCL INFO Connected all rings\n"
                   "h:1:1 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 busId c1000 - Init COMPLETE\n")
    got = bench.rccl_transport_summary(str(log))
    assert got["via"] == {"P2P/IPC": 2, "NET/Socket/0": 1}
    assert any("Connected all rings" in ln for ln in got["lines"]) and any("nranks 8" in ln for ln in got["lines"])
    assert "error" in bench.rccl_transport_summary(str(tmp_path / "absent.log"))
    if not torch.cuda.is_available():
        assert bench.gpu_sensors(0) == (None, None)


def test_bench_transport_check_allreduce_expectation_and_profile_manifest(tmp_path, monkeypatch):
    """Round 6: (i) channels `via SHM` / `via NET` in RCCL's set-up log end a multi-GPU bench run loudly --
    a single node whose all-reduce bounces through host memory is a misconfigured box, not an MI355X number; (ii) the measured
    all-reduce is printed next to what 2 (n-1)/n of the bucket costs on one 153 GB/s xGMI link; (iii) the PMC traffic summary
    is the one profiles/LATEST declares to be HEAD's, not the lexicographically last tag."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    assert bench.check_rccl_transport({"via": {"P2P/IPC": 56}}, "nccl") == {"host_or_network_channels": {}}
    assert bench.check_rccl_transport({"backend": "gloo"}, "gloo") is None
    for bad in ({"P2P/IPC": 6, "SHM/direct/direct": 8}, {"NET/Socket/0": 2}):
        with pytest.raises(SystemExit) as e:
            bench.check_rccl_transport({"via": bad}, "nccl")
        assert "xGMI" in str(e.value)
    monkeypatch.setenv("KEYMORPH_BENCH_ALLOW_HOST_TRANSPORT", "1")
    assert bench.check_rccl_transport({"via": {"SHM/direct/direct": 8}}, "nccl") == {"host_or_network_channels": {"SHM/direct/direct": 8}}
    ex = bench.allreduce_expectation(16 * 2 ** 20, 8, 0.5)
    assert abs(ex["ring_one_link_ms"] - 1e3 * (2 * 7 / 8 * 16 * 2 ** 20) / 153e9) < 1e-9 and ex["all_links_ms"] == ex["ring_one_link_ms"] / 7
    assert abs(ex["measured_over_one_link"] - 0.5 / ex["ring_one_link_ms"]) < 1e-9 and bench.allreduce_expectation(1, 1, 0.0) is None
    # the committed manifest names a committed summary with conv3_fwd_* entries
    mb, src = bench.pmc_traffic("conv3_fwd_")
    tag = open(os.path.join(root, "profiles", "LATEST")).read().split()[0]
    assert src == f"{tag}_pmc_hbm_traffic.json" and mb and mb > 1e8, (mb, src)


def test_amp_scope_is_per_thread_and_travels_with_the_autograd_node():
    """use_amp is per call (round 6): `amp_scope` is a per-thread setting that ends with the `with` block, `_t` turns it
    into the `terms = 1` the C ABI takes, and a Function decorated with `_binds_amp` runs its BACKWARD under the setting of its
    own forward -- whatever another model set in between, and although autograd runs the backward on another thread."""
    import threading
    from keymorph_amd import backbone_ops as B
    assert B.amp_enabled() is B._AMP_DEFAULT
    with B.amp_scope(True):
        assert B.amp_enabled() and B._t(2) == 1 and B._t(3) == 3
        seen = []
        th = threading.Thread(target=lambda: seen.append(B.amp_enabled()))      # another thread: its own default
        th.start(); th.join()
        assert seen == [B._AMP_DEFAULT]
        with B.amp_scope(False):
            assert not B.amp_enabled() and B._t(2) == 2
        assert B.amp_enabled()
    assert B.amp_enabled() is B._AMP_DEFAULT

    log = []

    @B._binds_amp
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            log.append(("fwd", B.amp_enabled()))
            return x * 2

        @staticmethod
        def backward(ctx, g):
            log.append(("bwd", B.amp_enabled(), threading.current_thread() is threading.main_thread()))
            return g * 2

    x = torch.ones(3, requires_grad=True)
    with B.amp_scope(True):
        y = F.apply(x)
    with B.amp_scope(False):                       # "another model" between the forward and the backward
        F.apply(torch.ones(1, requires_grad=True))
    y.sum().backward()
    assert log[0] == ("fwd", True) and log[1] == ("fwd", False)
    assert log[2][0] == "bwd" and log[2][1] is True            # the forward's setting, not the last one seen
    assert torch.equal(x.grad, torch.full((3,), 2.0)) and B.amp_enabled() is B._AMP_DEFAULT
