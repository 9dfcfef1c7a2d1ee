"""GPU: the convolution kernels the bench times, forced onto ragged multi-sample volumes.

`conv3_fwd_g_kernel` (the LDS-DMA forward / data-gradient kernel, 38 % of the 256^3 step) is normally chosen only for
launches of >= 512 bricks, i.e. never by the small op tests.  `kmh_conv3d_fwd_bf_set_dispatch(2)` forces it whenever
its preconditions hold, so that its three instantiations (32-wide, 64-wide and z-paired output tiles) see partial
32x8x4 bricks, partial 8x8 brick patches and several samples in one persistent work list -- checked here against an
fp64 restatement of keymorph/unet3d/buildingblocks.py:46-78 (GroupNorm -> Conv3d(k3, p1, no bias) -> ReLU) and, bit for
bit, against `conv3_fwd_bf_kernel` (the source claims identical arithmetic).  The wave-specialised weight gradient
(`conv3_wgrad_ws_kernel`) runs on the same shapes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

G_KERNEL = {1: "conv3_fwd_g_kernel<1,false>", 2: "conv3_fwd_g_kernel<2,false>", 3: "conv3_fwd_g_kernel<1,true>"}


def gen(s):
    return torch.Generator().manual_seed(s)


def conv3_fp64(x, w):
    """x (N,D,H,W,Cin) fp64, w (Cout,Cin,3,3,3) fp64 -> (N,D,H,W,Cout): zero-padded 3x3x3 cross-correlation as 27
    shifted matrix products (differentiable; the fp64 truth of F.conv3d(padding=1), checked against it below)."""
    N, D, H, W, Cin = x.shape
    xp = F.pad(x, (0, 0, 1, 1, 1, 1, 1, 1))
    y = None
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                t = xp[:, kz:kz + D, ky:ky + H, kx:kx + W, :] @ w[:, :, kz, ky, kx].t()
                y = t if y is None else y + t
    return y


def group_norm_fp64(x, G, gamma, beta, eps=1e-5):
    N, D, H, W, C = x.shape
    xg = x.reshape(N, D * H * W, G, C // G)
    m = xg.mean(dim=(1, 3), keepdim=True)
    v = xg.var(dim=(1, 3), keepdim=True, unbiased=False)
    return ((xg - m) / torch.sqrt(v + eps)).reshape(N, D, H, W, C) * gamma + beta


def rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b).abs().max()) / (float(b.abs().max()) + 1e-300)


@pytest.fixture
def dispatch():
    """yields a setter for the forward kernel selection and restores the previous mode afterwards"""
    from keymorph_amd import _lib
    lib = _lib.load()
    old = lib.kmh_conv3d_fwd_bf_set_dispatch(1)
    lib.kmh_conv3d_fwd_bf_set_dispatch(old)

    def set_mode(m):
        assert lib.kmh_conv3d_fwd_bf_set_dispatch(m) >= 0
    yield set_mode
    lib.kmh_conv3d_fwd_bf_set_dispatch(old)


def variant(N, D, H, W, Cin, Cout, mask=False, addend=False):
    from keymorph_amd import _lib
    return _lib.load().kmh_conv3d_fwd_bf_variant(N, D, H, W, Cin, Cout, 2, int(mask), int(addend))


def test_fp64_restatement_is_conv3d():
    g = gen(0)
    x = torch.randn(2, 5, 4, 6, 7, generator=g, dtype=torch.float64)
    w = torch.randn(6, 5, 3, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv3d(x, w, None, padding=1).permute(0, 2, 3, 4, 1)
    assert rel(conv3_fp64(x.permute(0, 2, 3, 4, 1).contiguous(), w), ref) < 1e-14
    gam, bet = torch.randn(5, generator=g, dtype=torch.float64), torch.randn(5, generator=g, dtype=torch.float64)
    refn = F.group_norm(x, 1, gam, bet, 1e-5).permute(0, 2, 3, 4, 1)
    assert rel(group_norm_fp64(x.permute(0, 2, 3, 4, 1).contiguous(), 1, gam, bet), refn) < 1e-13


def test_dispatch_setter_and_variant_query(dispatch):
    from keymorph_amd import _lib
    lib = _lib.load()
    assert lib.kmh_conv3d_fwd_bf_set_dispatch(7) == -22 and lib.kmh_conv3d_fwd_bf_set_dispatch(-1) == -22
    dispatch(1)
    assert variant(1, 30, 61, 121, 64, 64) == 0            # 4 * 8 * 8 = 256 bricks of one channel group: below 512
    assert variant(4, 128, 128, 128, 64, 64) == 2 and variant(4, 256, 256, 256, 16, 32) == 1
    assert variant(4, 256, 256, 256, 32, 16) == 3
    dispatch(2)
    assert variant(1, 5, 7, 9, 16, 32) == 1 and variant(1, 5, 7, 9, 32, 16) == 3 and variant(1, 5, 7, 9, 48, 96) == 2
    assert variant(1, 5, 7, 9, 16, 32, mask=True) == 0      # a fused ReLU-mask operand belongs to conv3_fwd_bf_kernel
    assert variant(1, 5, 7, 9, 12, 32) == 0                 # whole 8-channel chunks only
    assert variant(1, 5, 7, 9, 32, 16, addend=True) == 0
    dispatch(0)
    assert variant(4, 128, 128, 128, 64, 64) == 0


# (N, (D, H, W), Cin, Cout): volumes that are multiples of NEITHER the 32x8x4 brick NOR the 8x8 brick patch, several
# samples in one work list.  (45,60,100): tiles (4, 8 of which the last is half, 12 of which the last holds 1 plane);
# (30,61,121): one x column of the last brick, a 5-row y brick, a 2-plane z brick; (130, 20, 33): 33 z bricks -> 5
# patches of 8 with one brick in the last, one voxel in the last x brick; (9, 243, 40): tiles_y = 31 -> 4 patches.
RAGGED = [
    (2, (45, 60, 100), 16, 32),       # g<1,false>: the 256^3 level's forward shape (16 -> 32)
    (2, (45, 60, 100), 32, 16),       # g<1,true>: its z-paired data gradient (32 -> 16)
    (2, (30, 61, 121), 64, 64),       # g<2,false>
    (3, (30, 61, 121), 48, 96),       # g<2,false>, two 64-wide channel groups, the second half empty; N = 3
    (2, (130, 20, 33), 32, 32),
    (1, (9, 243, 40), 24, 8),         # z-paired with Cout = 8
    (2, (7, 9, 31), 128, 72),         # 16 chunks, tiny volume: every brick partial
]
# whole-brick volumes with several bricks per persistent workgroup (768 bricks for 256 workgroups): the launches on which the
# 32-wide and z-paired tiles keep their output pieces in registers and store them under the next brick's first stage (round 5)
WHOLE = [
    (3, (32, 64, 128), 32, 32),
    (2, (16, 32, 64), 16, 32),
]


@pytest.mark.parametrize("cfg", RAGGED + WHOLE)
def test_forward_kernel_g_ragged_vs_fp64_and_bit_identical(cfg, dispatch):
    """relu(conv3(x * scale + shift)) with output statistics, through the C ABI entry the backbone uses
    (kmh_conv3d_fwd_bf via backbone_ops.conv3_raw): conv3_fwd_g_kernel forced == conv3_fwd_bf_kernel bit for bit, and
    both within 3e-6 of the tensor maximum of the fp64 result (the bar of test_conv_arithmetic_modes_vs_fp64)."""
    from keymorph_amd import backbone_ops as B
    N, dims, Cin, Cout = cfg
    D, H, W = dims
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        g = gen(100 + Cin + Cout)
        x = (torch.randn(N, D, H, W, Cin, generator=g).abs() + 0.1 * torch.randn(N, D, H, W, Cin, generator=g)).to(DEV)
        scale = (1 + 0.3 * torch.randn(N, Cin, generator=g)).to(DEV)
        shift = (0.3 * torch.randn(N, Cin, generator=g)).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / np.sqrt(27 * Cin)).to(DEV)
        xn64 = x.double() * scale.double()[:, None, None, None, :] + shift.double()[:, None, None, None, :]
        ref = torch.relu(conv3_fp64(xn64, w.double()))
        ascale = B.absmax_scale(xn64.float())        # any bound of the normalised tensor's magnitude will do
        out = {}
        for mode in (0, 2):
            dispatch(mode)
            v = variant(N, D, H, W, Cin, Cout)
            assert (v == 0) if mode == 0 else (v == (3 if Cout <= 16 else 2 if Cout > 32 else 1)), (mode, v)
            pk = B.pack_weight(w, False)
            st = torch.full((N, Cout, 2), float("nan"), dtype=torch.float64, device=DEV)
            y = B.conv3_raw(x, scale, shift, pk, None, N, D, H, W, Cin, Cout, False, True, ascale=ascale, stats_out=st)
            torch.cuda.synchronize()
            out[mode] = (y, st)
            assert rel(y, ref) < 3e-6, (mode, G_KERNEL.get(v), rel(y, ref))
            st_ref = torch.stack([ref.sum(dim=(1, 2, 3)), (ref * ref).sum(dim=(1, 2, 3))], dim=-1)
            assert rel(st, st_ref) < 2e-6, (mode, rel(st, st_ref))
        assert torch.equal(out[0][0], out[2][0]), "conv3_fwd_g_kernel is not bit-identical to conv3_fwd_bf_kernel"
        # the statistics are sums of the same values in a different order (bricks of another height)
        assert rel(out[2][1], out[0][1]) < 1e-6
    finally:
        B.set_conv_mode(old)


@pytest.mark.parametrize("cfg", RAGGED + WHOLE)
@pytest.mark.parametrize("blocked", [False, True])
def test_data_gradient_kernel_g_ragged_vs_fp64_and_bit_identical(cfg, blocked, dispatch):
    """the data gradient = the same kernel on tap-mirrored weights, no normalisation, no activation, premasked
    gradient (no mask operand), with the (sum dxn) statistics of its epilogue; optionally reading a channel-blocked
    gradient (N, C/8, D, H, W, 8) as the hand-off inside a DoubleConv does."""
    from keymorph_amd import backbone_ops as B
    N, dims, Cw_in, Cw_out = cfg           # the forward layer is Cw_in -> Cw_out; the data gradient maps Cw_out -> Cw_in
    D, H, W = dims
    if Cw_out % 8 or Cw_in % 4:
        pytest.skip("the LDS-DMA kernel needs whole 8-channel input chunks")
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        g = gen(200 + Cw_in + Cw_out)
        dz = torch.randn(N, D, H, W, Cw_out, generator=g)
        dz = (dz * (torch.rand(N, D, H, W, Cw_out, generator=g) > 0.4) * 3e-4).to(DEV)        # masked, small magnitudes
        w = (torch.randn(Cw_out, Cw_in, 3, 3, 3, generator=g) / np.sqrt(27 * Cw_in)).to(DEV)
        # dxn[v, ci] = sum_{tap, co} dz[v - tap, co] W[co, ci, tap]: correlation with the mirrored, transposed filter
        wt = w.double().flip(2, 3, 4).permute(1, 0, 2, 3, 4).contiguous()
        ref = conv3_fp64(dz.double(), wt)
        dscale = B.absmax_scale(dz)
        dz_in = dz
        if blocked:
            dz_in = dz.view(N, D, H, W, Cw_out // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()
        out = {}
        for mode in (0, 2):
            dispatch(mode)
            v = variant(N, D, H, W, Cw_out, Cw_in)
            assert (v == 0) if mode == 0 else (v == (3 if Cw_in <= 16 else 2 if Cw_in > 32 else 1)), (mode, v)
            pk = B.pack_weight(w, True)
            st = torch.full((N, Cw_in, 2), float("nan"), dtype=torch.float64, device=DEV)
            y = B.conv3_raw(dz_in, None, None, pk, None, N, D, H, W, Cw_out, Cw_in, False, False, ascale=dscale,
                            stats_out=st, in_blocked=blocked)
            torch.cuda.synchronize()
            out[mode] = (y, st)
            assert rel(y, ref) < 3e-6, (mode, G_KERNEL.get(v), rel(y, ref))
            # a cancelling sum: its error is measured against sum |dxn|
            err = float((st[..., 0] - ref.sum(dim=(1, 2, 3))).abs().max()) / float(ref.abs().sum(dim=(1, 2, 3)).max())
            assert err < 1e-6, (mode, err)
        assert torch.equal(out[0][0], out[2][0])
    finally:
        B.set_conv_mode(old)


@pytest.mark.parametrize("cfg", [(2, (30, 61, 121), 64, 64), (2, (45, 60, 100), 16, 32), (1, (20, 12, 70), 128, 128)])
def test_forward_kernel_g_addend_ragged(cfg, dispatch):
    """`addend` (the fused decoder operator's low-resolution contribution) is added before the ReLU and the statistics"""
    from keymorph_amd import backbone_ops as B
    N, dims, Cin, Cout = cfg
    D, H, W = dims
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        g = gen(300 + Cin)
        x = torch.randn(N, D, H, W, Cin, generator=g).to(DEV)
        add = torch.randn(N, D, H, W, Cout, generator=g).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / np.sqrt(27 * Cin)).to(DEV)
        ref = torch.relu(conv3_fp64(x.double(), w.double()) + add.double())
        out = {}
        for mode in (0, 2):
            dispatch(mode)
            assert (variant(N, D, H, W, Cin, Cout, addend=True) != 0) == (mode == 2)
            st = torch.empty((N, Cout, 2), dtype=torch.float64, device=DEV)
            y = B.conv3_raw(x, None, None, B.pack_weight(w, False), None, N, D, H, W, Cin, Cout, False, True,
                            stats_out=st, addend=add)
            out[mode] = y
            assert rel(y, ref) < 3e-6
            assert rel(st[..., 1], (ref * ref).sum(dim=(1, 2, 3))) < 2e-6
        assert torch.equal(out[0], out[2])
    finally:
        B.set_conv_mode(old)


@pytest.mark.parametrize("cfg", [(2, (45, 60, 100), 16, 32, 8), (2, (30, 61, 121), 64, 64, 8), (3, (21, 37, 50), 48, 96, 8),
                                 (2, (45, 60, 100), 32, 16, 8)])
def test_single_conv_layer_ragged_forced_kernel_g_all_gradients_vs_fp64(cfg, dispatch):
    """One whole SingleConv (GroupNorm -> conv -> ReLU), forward and every gradient, with the LDS-DMA kernel forced in
    both directions and the wave-specialised weight gradient (conv3_wgrad_ws_kernel) on the same ragged multi-sample
    volume, against fp64 autograd of the same expression.  The cotangent is premasked by (y > 0), as a downstream
    SingleConv hands it over -- the only case in which the data gradient takes no mask operand."""
    from keymorph_amd import backbone_ops as B
    N, dims, Cin, Cout, G = cfg
    D, H, W = dims
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        dispatch(2)
        g = gen(400 + Cin + Cout)
        x = (torch.randn(N, D, H, W, Cin, generator=g).abs() + 0.1 * torch.randn(N, D, H, W, Cin, generator=g)).to(DEV)
        gamma = (1 + 0.2 * torch.randn(Cin, generator=g)).to(DEV)
        beta = (0.2 * torch.randn(Cin, generator=g)).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / np.sqrt(27 * Cin)).to(DEV)
        cot = torch.randn(N, D, H, W, Cout, generator=g).to(DEV)
        Hh = [t.clone().requires_grad_(True) for t in (x, gamma, beta, w)]
        assert variant(N, D, H, W, Cin, Cout) != 0 and variant(N, D, H, W, Cout, Cin) != 0
        yh = B.single_conv_gcr(*Hh, G, x_from_relu=False, dy_premasked=True)
        # the cotangent is masked with OUR sign pattern (what a downstream layer hands over) and the fp64 reference
        # differentiates the pre-activation against the same masked cotangent, so that a sign flip of a |y| ~ 1e-7
        # output cannot enter the comparison
        cotm = cot * (yh.detach() > 0)
        (yh * cotm).sum().backward()
        R = [t.double().clone().requires_grad_(True) for t in (x, gamma, beta, w)]
        pre = conv3_fp64(group_norm_fp64(R[0], G, R[1], R[2]), R[3])
        (pre * cotm.double()).sum().backward()
        assert rel(yh, torch.relu(pre.detach())) < 3e-6
        assert rel(Hh[0].grad, R[0].grad) < 5e-6
        assert rel(Hh[3].grad, R[3].grad) < 3e-6
        assert rel(Hh[1].grad, R[1].grad) < 2e-5 and rel(Hh[2].grad, R[2].grad) < 2e-5
    finally:
        B.set_conv_mode(old)


@pytest.mark.parametrize("log2_ratio", [10, 20, 30])
def test_gradient_dynamic_range_across_samples(log2_ratio, dispatch):
    """The f16x3 convolutions carry ONE power-of-two range scale per gradient tensor.  Two samples of one batch whose
    cotangents differ by 2^r: the faint sample's data gradient keeps hi + lo fp16 terms whose quantum is 2^-24 of the
    scaled maximum, i.e. its own relative precision is max(2^-22, 2^(r - 39)) -- fp32-class up to r ~ 17, 2e-6 at
    r = 20, 2e-3 at r = 30.  This test measures it against fp64 (per sample, relative to that sample's own maximum) and
    pins the supported range: <= 1e-5 for ratios up to 2^20; beyond that `backbone_ops.range_audit` (below) is the
    detector and bf16x6 (8 exponent bits, no range scale) the arithmetic to switch to."""
    from keymorph_amd import backbone_ops as B
    N, D, H, W, Cin, Cout, G = 2, 12, 20, 40, 16, 32, 8
    old = B.CONV_MODE
    try:
        g = gen(77)
        x = (torch.randn(N, D, H, W, Cin, generator=g).abs() + 0.1 * torch.randn(N, D, H, W, Cin, generator=g)).to(DEV)
        gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).to(DEV), (0.2 * torch.randn(Cin, generator=g)).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / np.sqrt(27 * Cin)).to(DEV)
        cot = torch.randn(N, D, H, W, Cout, generator=g).to(DEV)
        cot[1] *= 2.0 ** -log2_ratio
        errs = {}
        for mode in ("f16x3", "bf16x6"):
            B.set_conv_mode(mode)
            Hh = [t.clone().requires_grad_(True) for t in (x, gamma, beta, w)]
            yh = B.single_conv_gcr(*Hh, G, x_from_relu=False, dy_premasked=True)
            # cotangent masked by THIS output's sign pattern, the fp64 reference differentiates the pre-activation against
            # the same masked cotangent (a ReLU flip at |y| ~ 1e-7 would otherwise move the weight gradient by 1e-3)
            cotm = cot * (yh.detach() > 0)
            (yh * cotm).sum().backward()
            R = [t.double().clone().requires_grad_(True) for t in (x, gamma, beta, w)]
            (conv3_fp64(group_norm_fp64(R[0], G, R[1], R[2]), R[3]) * cotm.double()).sum().backward()
            errs[mode] = [rel(Hh[0].grad[n], R[0].grad[n]) for n in range(N)]
            assert rel(Hh[3].grad, R[3].grad) < 3e-6            # the weight gradient is dominated by the loud sample
        print(f"\ncotangent ratio 2^{log2_ratio}: data-gradient error per sample (own maximum) f16x3 {errs['f16x3']}, "
              f"bf16x6 {errs['bf16x6']}")
        assert errs["f16x3"][0] < 5e-6 and errs["bf16x6"][0] < 5e-6 and errs["bf16x6"][1] < 5e-6
        if log2_ratio <= 20:
            assert errs["f16x3"][1] < 1e-5, errs
        else:
            assert errs["f16x3"][1] < 2.0 ** (log2_ratio - 36), errs      # degrades as predicted, never garbage
        # the detector: fraction of non-zero gradient elements whose lo term has left fp16's normal range
        audit = B.range_audit(cot)
        assert audit["per_sample_log2_below_max"][1] == pytest.approx(log2_ratio, abs=1.0)
        assert (audit["worst_relative_precision"] > 2.0 ** -21) == (log2_ratio > 18), audit
    finally:
        B.set_conv_mode(old)


@pytest.mark.parametrize("cfg", [(2, (12, 16, 64), 16, 32), (2, (45, 60, 100), 16, 32), (1, (30, 61, 121), 24, 24),
                                 (3, (9, 10, 35), 8, 20)])
def test_conv_epilogue_pooling_equals_conv_then_maxpool(cfg, dispatch, monkeypatch):
    """GroupNorm -> conv -> ReLU -> MaxPool3d(2) with the pooling done in the convolution's epilogue (the full-resolution
    output is never written; keymorph/unet3d/buildingblocks.py:46-78 + the next Encoder's pooling) against the same
    layer followed by the separate pooling kernel: pooled output BIT-identical, identical winners (so identical
    gradients: every parameter gradient and the input gradient bit for bit), the pooled tensor's statistics, on even,
    odd and ragged volumes; and the pooled output against F.max_pool3d of the fp64 layer."""
    from keymorph_amd import backbone_ops as B
    N, dims, Cin, Cout = cfg
    D, H, W = dims
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        dispatch(2)
        assert B.conv_pool_ok(N, D, H, W, Cin, Cout)
        g = gen(500 + Cin)
        x = (torch.randn(N, D, H, W, Cin, generator=g).abs() + 0.1 * torch.randn(N, D, H, W, Cin, generator=g)).to(DEV)
        gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).to(DEV), (0.2 * torch.randn(Cin, generator=g)).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / np.sqrt(27 * Cin)).to(DEV)
        cot = torch.randn(N, D // 2, H // 2, W // 2, Cout, generator=g).to(DEV)
        G = 8

        def run(fused):
            Hh = [t.clone().requires_grad_(True) for t in (x, gamma, beta, w)]
            before = B.POOL_STATS["fused"]
            if fused:
                yp = B.single_conv_gcr(*Hh, G, x_from_relu=False, dy_premasked=True, pool=True)
            else:
                yp = B.maxpool2(B.single_conv_gcr(*Hh, G, x_from_relu=False, dy_premasked=True))
            assert B.POOL_STATS["fused"] == before + int(fused)
            st = B._peek_stats(yp)
            (yp * (cot * (yp.detach() > 0))).sum().backward()
            return yp.detach(), st, [t.grad for t in Hh]

        yf, sf, gf = run(True)
        yu, su, gu = run(False)
        assert yf.shape == (N, D // 2, H // 2, W // 2, Cout)
        assert torch.equal(yf, yu), "pooled output differs from conv followed by maxpool"
        for a, b in zip(gf, gu):
            assert torch.equal(a, b), "a gradient differs: the recorded winners are not the pooling kernel's"
        # statistics of the pooled tensor (what the next GroupNorm needs) came with the epilogue
        assert sf is not None and su is None
        V = (D // 2) * (H // 2) * (W // 2)
        ref = B.channel_stats(yf, None, N, V, Cout)
        assert rel(sf, ref) < 1e-6
        # and against fp64
        R = [t.double() for t in (x, gamma, beta, w)]
        y64 = torch.relu(conv3_fp64(group_norm_fp64(R[0], G, R[1], R[2]), R[3]))
        p64 = F.max_pool3d(y64.permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1)
        assert rel(yf, p64) < 3e-6
    finally:
        B.set_conv_mode(old)


def test_conv_epilogue_pooling_tie_rule_and_whole_network(dispatch):
    """ties: a constant input makes every window's 8 children equal (ReLU of one value): the winner must be child 0, as
    ATen's max_pool3d and kmh_maxpool3d_fwd choose; then the whole TruncatedUNet3D with and without the fused
    pooling gives the same keypoints and parameter gradients."""
    from keymorph_amd import backbone_ops as B, _lib
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    import os
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        dispatch(2)
        N, D, H, W, Cin, Cout = 1, 8, 8, 32, 8, 32
        x = torch.ones(N, D, H, W, Cin, device=DEV)
        x[0, 4:, :, :, :] = 2.0                                       # two constant slabs: interior windows are all ties
        gamma, beta = torch.ones(Cin, device=DEV), torch.full((Cin,), 0.5, device=DEV)
        w = torch.full((Cout, Cin, 3, 3, 3), 0.01, device=DEV)
        xg = x.clone().requires_grad_(True)
        yp = B.single_conv_gcr(xg, gamma, beta, w, 8, x_from_relu=False, dy_premasked=True, pool=True)
        yu = B.maxpool2(B.single_conv_gcr(x, gamma, beta, w, 8, x_from_relu=False, dy_premasked=True))
        assert torch.equal(yp, yu)
        (yp.sum()).backward()
        xu = x.clone().requires_grad_(True)
        B.maxpool2(B.single_conv_gcr(xu, gamma, beta, w, 8, x_from_relu=False, dy_premasked=True)).sum().backward()
        assert torch.equal(xg.grad, xu.grad)
        # whole network
        torch.manual_seed(5)
        net = TruncatedUNet3D(1, 16, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=3,
                              is_segmentation=False, conv_padding=1).to(DEV).train()
        img = torch.rand(2, 1, 24, 40, 64, generator=gen(9)).to(DEV)
        outs = {}
        for fused in (True, False):
            if fused:
                os.environ.pop("KEYMORPH_NO_CONV_POOL", None)
            else:
                os.environ["KEYMORPH_NO_CONV_POOL"] = "1"
            try:
                net.zero_grad(set_to_none=True)
                before = B.POOL_STATS["fused"]
                pts = net.keypoints_ij(img)
                assert B.POOL_STATS["fused"] == before + int(fused)
                (pts * torch.linspace(-1, 1, pts.numel(), device=DEV).reshape(pts.shape)).sum().backward()
                outs[fused] = (pts.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()})
            finally:
                os.environ.pop("KEYMORPH_NO_CONV_POOL", None)
        # (not bit for bit: the next GroupNorm's statistics come from the epilogue's brick sums in one case and from a
        # separate pass over the pooled tensor in the other -- the same numbers in another summation order)
        assert float((outs[True][0] - outs[False][0]).abs().max()) < 2e-6
        va = torch.cat([v.reshape(-1).double() for v in outs[True][1].values()])
        vb = torch.cat([v.reshape(-1).double() for v in outs[False][1].values()])
        per = {k: float((outs[True][1][k].double() - outs[False][1][k].double()).norm() / (outs[False][1][k].double().norm() + 1e-300))
               for k in outs[True][1]}
        # (whole vector 0.9e-4 with the eight-wave kernel's pooling epilogue, 1.3e-4 with the one-wave kernel's: the pooled
        # tensor's statistics are grouped by wave in one and by (plane, row half) in the other; the worst tensors are the
        # GroupNorm affines of the next block and the first layer's one-element GroupNorm weight, a sum that cancels to ~1e-6)
        assert float((va - vb).norm() / vb.norm()) < 3e-4, (float((va - vb).norm() / vb.norm()), sorted(per.items(), key=lambda kv: -kv[1])[:4])
    finally:
        B.set_conv_mode(old)


@pytest.mark.parametrize("cfg", [(2, (6, 11, 64), 16), (1, (5, 9, 100), 16), (2, (4, 17, 33), 12), (1, (9, 8, 256), 16),
                                 (1, (3, 3, 7), 4)])
def test_first_layer_correlation_kernel_vs_fp64(cfg):
    """The first U-Net convolution (1 -> Cout <= 16): its weight / GroupNorm gradients come from the correlations of the
    gradient with the raw image, computed on the fp32 matrix cores (first_wgrad_mfma_kernel, round 3) -- against fp64
    autograd of GroupNorm(1, 1) -> Conv3d(1, Cout) -> ReLU on ragged rows (W = 100, 33, 7: scalar staging; 64, 256: the
    pipelined 16-byte path), several samples, Cout 16 / 12 / 4, with a premasked cotangent."""
    from keymorph_amd import backbone_ops as B
    N, dims, Cout = cfg
    D, H, W = dims
    g = gen(600 + W)
    img = torch.rand(N, D, H, W, 1, generator=g).to(DEV)
    gamma, beta = torch.tensor([1.3], device=DEV), torch.tensor([-0.2], device=DEV)
    w = (torch.randn(Cout, 1, 3, 3, 3, generator=g) / np.sqrt(27.0)).to(DEV)
    cot = torch.randn(N, D, H, W, Cout, generator=g).to(DEV)
    Hh = [t.clone().requires_grad_(True) for t in (gamma, beta, w)]
    yh = B.single_conv_gcr(img, Hh[0], Hh[1], Hh[2], 1, x_from_relu=False, dy_premasked=True)
    cotm = cot * (yh.detach() > 0)
    (yh * cotm).sum().backward()
    R = [t.double().clone().requires_grad_(True) for t in (gamma, beta, w)]
    pre = conv3_fp64(group_norm_fp64(img.double(), 1, R[0], R[1]), R[2])
    (pre * cotm.double()).sum().backward()
    assert rel(yh, torch.relu(pre.detach())) < 3e-6
    assert rel(Hh[2].grad, R[2].grad) < 3e-6, rel(Hh[2].grad, R[2].grad)
    # the one-element GroupNorm parameters: sums over the whole volume, compared against the size of their terms
    scale = float((cotm.double().abs() * pre.detach().abs()).sum()) + 1e-30
    assert abs(float(Hh[0].grad) - float(R[0].grad)) < 1e-6 * scale
    assert abs(float(Hh[1].grad) - float(R[1].grad)) < 1e-6 * scale


def test_forward_kernels_repeat_bit_for_bit_with_other_work_in_between():
    """conv3_fwd_s_kernel (one wave per SIMD, the default for 16 < Cout) waits for its loads with plain full drains; an earlier
    version counted them and gave run-to-run different results inside a training step (never back to back on warm caches).
    Here the same launch is repeated with a cache-flushing copy and another convolution in between: outputs and epilogue
    statistics must be bit-equal every time -- the 64-wide tile, the 32-wide one, a channel-blocked data gradient."""
    from keymorph_amd import backbone_ops as B
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        g = gen(77)
        junk = torch.empty(96 * 1024 * 1024, device=DEV)                      # 384 MB: past L2 + the infinity cache
        for (N, D, Cin, Cout, blocked) in ((2, 64, 64, 64, False), (4, 64, 32, 32, False), (2, 64, 64, 32, True)):
            x = torch.randn(N, D, D, D, Cin, generator=g).to(DEV)
            # (the data gradient Cin -> Cout uses the FORWARD layer's filter (Cin, Cout, 3, 3, 3), packed transposed)
            w = (torch.randn(*((Cin, Cout) if blocked else (Cout, Cin)), 3, 3, 3, generator=g) / np.sqrt(27 * Cin)).to(DEV)
            pk = B.pack_weight(w, blocked)
            xin = x.view(N, D, D, D, Cin // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous() if blocked else x
            asc = B.absmax_scale(x)
            other = torch.randn(1, 40, 40, 40, 16, generator=g).to(DEV)
            pko = B.pack_weight((torch.randn(32, 16, 3, 3, 3, generator=g) * 0.05).to(DEV), False)
            ref = None
            for rep in range(6):
                st = torch.full((N, Cout, 2), float("nan"), dtype=torch.float64, device=DEV)
                y = B.conv3_raw(xin, None, None, pk, None, N, D, D, D, Cin, Cout, False, not blocked, ascale=asc, stats_out=st,
                                in_blocked=blocked)
                if ref is None:
                    ref = (y.clone(), st.clone())
                else:
                    assert torch.equal(y, ref[0]) and torch.equal(st, ref[1]), (N, D, Cin, Cout, blocked, rep)
                junk.fill_(float(rep))                                            # evict the weights and the halo
                B.conv3_raw(other, None, None, pko, None, 1, 40, 40, 40, 16, 32, False, True)
    finally:
        B.set_conv_mode(old)


def test_the_eight_wave_kernel_still_matches(tmp_path):
    """KEYMORPH_FWD_S=0 keeps conv3_fwd_g_kernel<2,false> / <1,false> as the A/B arm of conv3_fwd_s_kernel: its outputs must
    stay bit-identical to the default's (the switch is read once per process, so the other arm runs in a child)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from keymorph_amd import backbone_ops as B, _lib\n"
        "B.set_conv_mode('f16x3'); lib = _lib.load(); lib.kmh_conv3d_fwd_bf_set_dispatch(2)\n"
        "g = torch.Generator().manual_seed(5); outs = {}\n"
        "for (N, dims, Cin, Cout) in ((2, (30, 61, 121), 64, 64), (2, (45, 60, 100), 32, 32)):\n"
        "    D, H, W = dims\n"
        "    x = torch.randn(N, D, H, W, Cin, generator=g).cuda(); w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.05).cuda()\n"
        "    sc = (1 + 0.3 * torch.randn(N, Cin, generator=g)).cuda(); sh = (0.3 * torch.randn(N, Cin, generator=g)).cuda()\n"
        "    asc = B.absmax_scale(x * 2.5)\n"
        "    y = B.conv3_raw(x, sc, sh, B.pack_weight(w, False), None, N, D, H, W, Cin, Cout, False, True, ascale=asc)\n"
        "    outs['%%d_%%d' %% (Cin, Cout)] = y.cpu().numpy()\n"
        "np.savez(sys.argv[1], **outs)\n" % root)
    files = {}
    for arm in ("0", "2"):
        files[arm] = str(tmp_path / f"arm{arm}.npz")
        r = subprocess.run([sys.executable, "-c", code, files[arm]], env=dict(os.environ, KEYMORPH_FWD_S=arm),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(files["0"]), np.load(files["2"])
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_training_steps_repeat_bit_for_bit(dispatch):
    """Three optimisation steps of a small TruncatedUNet3D keypoint model with the LDS-DMA / one-wave kernels forced onto its
    volumes, twice from the same initial state: identical parameters, bit for bit (every reduction has a fixed order; the
    forward kernel's waits are full drains -- with counted waits this is what differed from run to run)."""
    from keymorph_amd import backbone_ops as B, parallel
    from keymorph_amd.unet3d.model import TruncatedUNet3D
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        dispatch(2)
        img = torch.rand(2, 1, 24, 40, 64, generator=gen(31)).to(DEV)
        tgt = torch.randn(2, 16, 3, generator=gen(32)).to(DEV)
        junk = torch.empty(64 * 1024 * 1024, device=DEV)

        def run():
            torch.manual_seed(7)
            net = TruncatedUNet3D(1, 16, 1, final_sigmoid=False, f_maps=32, layer_order="gcr", num_groups=8, num_levels=3,
                                  is_segmentation=False, conv_padding=1).to(DEV).train()
            flat = parallel.FlatParams(net.parameters())
            opt = parallel.FusedAdam(flat, lr=1e-3)
            for it in range(3):
                flat.zero_grad()
                ((net.keypoints_ij(img) - tgt) ** 2).mean().backward()
                junk.fill_(float(it))                         # cold caches for the next step's first launches
                opt.step(flat.allreduce_grads())
            return flat.flat.clone()

        a, b = run(), run()
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    finally:
        B.set_conv_mode(old)


# ---------------------------------------------------------------------------------------------------------------
# Round 5: the pre-split operand path -- the pooling backward writes fp16 hi / lo records, the z-paired data gradient copies
# them (conv3_fwd_s_kernel<1, true, true>: LDS-DMA pieces, no conversion)
def _pool_bwd_pair(N, shape, C, seed):
    """-> (pooled gradient (N, D/2, H/2, W/2, C), winners (same, uint8 in 0..7), its device range scale {S, 1/S})"""
    from keymorph_amd import backbone_ops as B
    D, H, W = shape
    g = gen(seed)
    dy = torch.randn(N, D // 2, H // 2, W // 2, C, generator=g).to(DEV)
    dy[0, 0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-30, -3.5])          # zeros of both signs, a tiny value
    arg = torch.randint(0, 8, (N, D // 2, H // 2, W // 2, C), generator=g, dtype=torch.uint8).to(DEV)
    return dy.contiguous(), arg.contiguous(), B.absmax_scale(dy)


@pytest.mark.parametrize("N,shape,C", [(2, (8, 16, 64), 32), (1, (12, 20, 34), 32), (3, (6, 10, 70), 16), (1, (4, 8, 32), 64)])
def test_pool_backward_split_records_are_the_consumers_own_split(N, shape, C):
    """kmh_maxpool3d_bwd_split: every record = (fp16 hi, fp16 lo) of fmaf(scatter(dy), S, 0) exactly as split8<2> forms them
    (hi = round-to-nearest fp16, lo = fp16 of the exact residual), the plane's last record zero -- checked against the dense
    scatter (kmh_maxpool3d_bwd) split on the host."""
    from keymorph_amd import _lib
    from keymorph_amd.ops import _p, _stream, check
    lib = _lib.load()
    D, H, W = shape
    V = D * H * W
    dy, arg, sc = _pool_bwd_pair(N, shape, C, 5)
    dense = torch.empty(N, D, H, W, C, device=DEV)
    check(lib.kmh_maxpool3d_bwd(None, _p(arg), _p(dy), None, 0, _p(dense), N, D, H, W, C, 0, _stream()), "bwd")
    assert lib.kmh_maxpool3d_bwd_split_bytes(N, D, H, W, C) == N * (C // 8) * (V + 1) * 32
    rec = torch.full((N, C // 8, V + 1, 16), -1, dtype=torch.int16, device=DEV)
    check(lib.kmh_maxpool3d_bwd_split(_p(arg), _p(dy), _p(sc), _p(rec), N, D, H, W, C, _stream()), "split")
    S = float(sc[0])
    v = (dense.double() * S).reshape(N, V, C // 8, 8).permute(0, 2, 1, 3)            # (N, chunk, V, 8), exact in fp64
    hi = v.to(torch.float32).to(torch.float16)
    lo = (v - hi.double()).to(torch.float32).to(torch.float16)
    got = rec.view(torch.float16)
    assert torch.equal(got[:, :, :V, :8].float(), hi.float()) and torch.equal(got[:, :, :V, 8:].float(), lo.float())
    assert int(rec[:, :, V, :].abs().max()) == 0


@pytest.mark.parametrize("N,shape,Cin,Cout", [(2, (8, 16, 64), 32, 16), (1, (12, 20, 34), 32, 16), (3, (6, 10, 70), 16, 8),
                                              (1, (4, 8, 32), 64, 16), (2, (44, 60, 100), 32, 16), (3, (32, 64, 128), 32, 16)])
def test_presplit_data_gradient_is_bit_identical_to_the_fp32_operand(N, shape, Cin, Cout, dispatch):
    """The z-paired data gradient on the pre-split records (in_blocked = 2) against the same launch on the channel-blocked fp32
    scatter (in_blocked = 1: conv3_fwd_g_kernel<1,true>) and on the dense one (conv3_fwd_bf_kernel): identical bits, with the
    output statistics the GroupNorm backward takes from the epilogue."""
    from keymorph_amd import _lib, backbone_ops as B
    from keymorph_amd.ops import _p, _stream, check
    lib = _lib.load()
    B.set_conv_mode("f16x3")
    D, H, W = shape
    V = D * H * W
    dy, arg, sc = _pool_bwd_pair(N, shape, Cin, 7)
    w = (torch.randn(Cin, Cout, 3, 3, 3, generator=gen(8)) * 0.05).to(DEV)        # (Cout_w, Cin_w) = (Cin, Cout) of this view
    pk = B.pack_weight(w, True)
    dense = torch.empty(N, D, H, W, Cin, device=DEV)
    blocked = torch.empty(N, Cin // 8, D, H, W, 8, device=DEV)
    rec = torch.empty((N, Cin // 8, V + 1, 8), device=DEV)
    check(lib.kmh_maxpool3d_bwd(None, _p(arg), _p(dy), None, 0, _p(dense), N, D, H, W, Cin, 0, _stream()), "bwd")
    check(lib.kmh_maxpool3d_bwd(None, _p(arg), _p(dy), None, 0, _p(blocked), N, D, H, W, Cin, 1, _stream()), "bwd blocked")
    check(lib.kmh_maxpool3d_bwd_split(_p(arg), _p(dy), _p(sc), _p(rec), N, D, H, W, Cin, _stream()), "bwd split")
    dispatch(2)
    assert lib.kmh_conv3d_fwd_bf_split_ok(N, D, H, W, Cin, Cout, 2) == 1
    st = {k: torch.empty((N, Cout, 2), dtype=torch.float64, device=DEV) for k in ("blocked", "split")}
    y_b = B.conv3_raw(blocked, None, None, pk, None, N, D, H, W, Cin, Cout, False, False, ascale=sc, in_blocked=True,
                      stats_out=st["blocked"])
    y_s = B.conv3_raw(rec, None, None, pk, None, N, D, H, W, Cin, Cout, False, False, ascale=sc, in_blocked=2,
                      stats_out=st["split"])
    dispatch(0)
    y_d = B.conv3_raw(dense, None, None, pk, None, N, D, H, W, Cin, Cout, False, False, ascale=sc)
    assert torch.isfinite(y_s).all() and float(y_s.abs().max()) > 0
    assert torch.equal(y_s, y_b), float((y_s - y_b).abs().max())
    assert torch.equal(y_s, y_d)
    ref = conv3_fp64(dense.double().cpu(), w.double().cpu().permute(1, 0, 2, 3, 4).flip(2, 3, 4).contiguous())
    assert rel(y_s.cpu(), ref) < 3e-6
    # statistics: the same outputs summed (grouped by wave = plane pair in the one-wave kernel, by (pair, row quarter) before)
    assert rel(st["split"].cpu(), st["blocked"].cpu().double()) < 1e-6
    # repeated launches (the deep fragment ring and the sparse drains): same bits
    for _ in range(3):
        torch.empty(64 << 20, device=DEV).normal_()          # evict
        dispatch(2)
        again = B.conv3_raw(rec, None, None, pk, None, N, D, H, W, Cin, Cout, False, False, ascale=sc, in_blocked=2)
        assert torch.equal(again, y_s)


def test_presplit_operand_refused_where_it_is_not_served(dispatch):
    from keymorph_amd import _lib
    lib = _lib.load()
    dispatch(2)
    assert lib.kmh_conv3d_fwd_bf_split_ok(1, 8, 16, 64, 32, 32, 2) == 0           # not the z-paired tile
    assert lib.kmh_conv3d_fwd_bf_split_ok(1, 8, 16, 64, 32, 16, 3) == 0           # bf16x6
    assert lib.kmh_conv3d_fwd_bf_split_ok(1, 8, 16, 64, 12, 16, 2) == 0           # whole chunks only
    # (round 6, ADVICE r5: the answer is the SHAPE's, not the dispatch mode's -- in_blocked = 2 always launches the one-wave
    # z-paired kernel, and a producer that asked at forward time must get the same answer at backward time)
    for mode in (0, 1, 2):
        dispatch(mode)
        assert lib.kmh_conv3d_fwd_bf_split_ok(1, 8, 16, 64, 32, 16, 2) == 1
        assert lib.kmh_conv3d_fwd_bf_split_ok(4, 256, 256, 256, 32, 16, 2) == 1


@pytest.mark.parametrize("cfg", [(2, (12, 16, 64), 16, 32), (2, (44, 60, 100), 16, 32), (1, (30, 62, 122), 8, 24)])
def test_conv_pool_backward_with_presplit_scatter_equals_the_fp32_scatter(cfg, dispatch, monkeypatch):
    """The conv + pooling operator's backward with its pooled gradient scattered straight into pre-split records (both the
    z-paired data gradient and the wave-specialised weight gradient read them) against the same backward on the channel-
    blocked fp32 scatter (KEYMORPH_NO_SPLIT_POOLGRAD=1): every gradient -- weights, GroupNorm affine, input -- bit for bit,
    and the weight gradient against fp64 autograd of keymorph/unet3d/buildingblocks.py:46-78 + max_pool3d."""
    from keymorph_amd import backbone_ops as B
    N, dims, Cin, Cout = cfg
    D, H, W = dims
    old = B.CONV_MODE
    try:
        B.set_conv_mode("f16x3")
        dispatch(2)
        assert B.conv_pool_ok(N, D, H, W, Cin, Cout) and B.grad_blocked_ok(N, D, H, W, Cin, Cout)
        g = gen(600 + Cin)
        x = (torch.randn(N, D, H, W, Cin, generator=g).abs() + 0.1 * torch.randn(N, D, H, W, Cin, generator=g)).to(DEV)
        gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).to(DEV), (0.2 * torch.randn(Cin, generator=g)).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / np.sqrt(27 * Cin)).to(DEV)
        cot = torch.randn(N, D // 2, H // 2, W // 2, Cout, generator=g).to(DEV)
        G = 8

        def run(split):
            if split:
                monkeypatch.delenv("KEYMORPH_NO_SPLIT_POOLGRAD", raising=False)
            else:
                monkeypatch.setenv("KEYMORPH_NO_SPLIT_POOLGRAD", "1")
            Hh = [t.clone().requires_grad_(True) for t in (x, gamma, beta, w)]
            before = B.SPLIT_STATS["handoffs"]
            yp = B.single_conv_gcr(*Hh, G, x_from_relu=False, dy_premasked=True, dy_blocked=True, pool=True)
            (yp * (cot * (yp.detach() > 0))).sum().backward()
            assert B.SPLIT_STATS["handoffs"] == before + int(split)
            return yp.detach(), [t.grad for t in Hh]

        assert B.pool_grad_split_ok(N, D, H, W, Cin, Cout)
        ys, gs = run(True)
        yb, gb = run(False)
        assert torch.equal(ys, yb)
        # the weight gradient is the same sum of the same words: identical bits.  The normalised-input gradient dxn is too
        # (test_presplit_data_gradient_is_bit_identical_to_the_fp32_operand), but GroupNorm's backward takes sum(dxn) from the
        # convolution's epilogue, and the one-wave kernel groups those partial sums by wave = plane pair where the eight-wave
        # kernel of the fp32 route groups them by (pair, row quarter): the coefficients, and with them dx / dgamma / dbeta,
        # agree to fp32 rounding of the partial sums
        assert torch.isfinite(gs[3]).all() and torch.equal(gs[3], gb[3]), float((gs[3] - gb[3]).abs().max())
        for name, a, b in zip(("x", "gamma", "beta"), gs, gb):
            assert torch.isfinite(a).all() and rel(a, b.double()) < 2e-6, (name, rel(a, b.double()))
        R = [t.double().cpu().requires_grad_(True) for t in (x, gamma, beta, w)]
        y64 = torch.relu(conv3_fp64(group_norm_fp64(R[0], G, R[1], R[2]), R[3]))
        p64 = F.max_pool3d(y64.permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1)
        (p64 * (cot.cpu().double() * (p64.detach() > 0))).sum().backward()
        assert rel(gs[3].cpu(), R[3].grad) < 5e-6 and rel(gs[0].cpu(), R[0].grad) < 2e-5
    finally:
        B.set_conv_mode(old)
