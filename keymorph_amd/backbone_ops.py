"""Backbone operators (NDHWC internally) as autograd Functions over the C ABI.

    single_conv_gcr : GroupNorm -> Conv3d(k3,p1,no bias) -> ReLU  (buildingblocks.py:10-93)
    conv_block      : Conv3d(k3,p1,bias) [-> InstanceNorm -> ReLU -> MaxPool folded into the
                      NEXT block's loader]                         (keymorph/layers.py:137-187)
    maxpool2 / upcat / pointwise (final 1x1x1 conv, NDHWC -> NCDHW)
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check
from .ops import _p, _prep, _stream, workspace

Tensor = torch.Tensor
EPS = 1e-5

CONV_MODE = os.environ.get("KEYMORPH_HIP_CONV", "f16x3")
# arithmetic of the 3x3x3 convolutions (fp32 in, fp32 out, fp32 accumulate in every mode):
#   "f32"    v_mfma_f32_32x32x2_f32
#   "bf16x6" operands split into 3 bf16 terms, 6 products per fp32 product
#   "f16x3"  DEFAULT: operands range-scaled by a power of two and split into 2 fp16 terms, 3 products (half the MFMA
#            work of bf16x6 at the same measured accuracy: 5e-7 vs fp64, the fp32-MFMA kernel measures 6.5e-7;
#            csrc/conv_bf.hip, tests/test_backbone_gpu.py::test_conv_arithmetic_modes_vs_fp64)
_TERMS = {"f16x3": 2, "bf16x6": 3}
_HEAD_TERMS = {"f16x3": 2, "bf16x6": 3}       # csrc/headcom.hip: the same two split schemes
BF_ROWS_PER_WAVE = int(os.environ.get("KEYMORPH_HIP_BF_ROWS", "4"))   # 4 (32x8x2 brick, default) | 2 (32x4x2)


def set_conv_mode(mode: str):
    global CONV_MODE
    assert mode in ("f32", "f16x3", "bf16x6"), mode
    CONV_MODE = mode


# ---- use_amp: a per-call argument of the library, a per-thread scope here ------------------------------------------
# KeyMorph(use_amp=True).get_keypoints() opens `amp_scope(True)` around its backbone call; every autograd Function below
# records the setting of ITS forward in ctx and re-opens that scope around its backward (autograd runs backward on its own
# thread, long after the scope of the forward has closed), and every launching call passes `_t(terms)`: terms == 1 = "the
# fp16 kernels, hi x hi only" (include/keymorph_hip.h).  Nothing is process-wide: a use_amp=False model called between the
# forward and the backward of a use_amp=True one changes nothing for it, and the setting ends with get_keypoints().
# KEYMORPH_AMP=1 sets the default of threads that never opened a scope (A/B measurements of the operators alone).
import contextlib
import functools
import threading

_AMP_DEFAULT = os.environ.get("KEYMORPH_AMP", "0") not in ("", "0")
_amp_tls = threading.local()


def amp_enabled() -> bool:
    return bool(getattr(_amp_tls, "on", _AMP_DEFAULT))


@contextlib.contextmanager
def amp_scope(on: bool):
    """use_amp of KeyMorph (keymorph/model.py:176-191) for the operators called inside: True = the 27-tap forward /
    data-gradient kernels, the weight gradient, the fused decoder operator and the fused head multiply only the fp16 hi terms
    of their (range-scaled) operands -- fp16 inputs, fp32 accumulation, one MFMA per product block instead of three.  Only the
    default f16x3 mode has the variant."""
    prev = amp_enabled()
    _amp_tls.on = bool(on)
    try:
        yield
    finally:
        _amp_tls.on = prev


def set_amp(on: bool) -> bool:
    """Sets the calling thread's default (outside any amp_scope); returns the previous one.  Kept for scripts that profile the
    operators alone; models use amp_scope."""
    prev = amp_enabled()
    _amp_tls.on = bool(on)
    return prev


def _t(terms: int) -> int:
    """`terms` as the launching entry points take it: 1 = the terms-2 kernels with the hi x hi product only."""
    return 1 if (terms == 2 and amp_enabled()) else terms


def _binds_amp(cls):
    """Class decorator of an autograd Function whose kernels have the use_amp variant: the backward runs under the setting its
    forward saw."""
    fwd, bwd = cls.forward, cls.backward

    @staticmethod
    @functools.wraps(fwd)
    def forward(ctx, *a, **k):
        ctx._kmh_amp = amp_enabled()
        return fwd(ctx, *a, **k)

    @staticmethod
    @functools.wraps(bwd)
    def backward(ctx, *g):
        with amp_scope(ctx._kmh_amp):
            return bwd(ctx, *g)

    cls.forward, cls.backward = forward, backward
    return cls


# ---- arithmetic of the first encoder block's data gradient (round 6) ---------------------
# The 32 -> 16 data gradient at full resolution feeds the sums behind the first GroupNorm's weight / bias gradient; under
# f16x3 its gradient operand carries ONE range scale and elements far below the tensor maximum keep fewer than 22 bits
# (fp16's exponent range), which shows in those two sums (4.9e-3 / 4.6e-3 of the fp64 truth at 128^3 / 512 kp against the
# reference's fp32 1.0e-3 / 1.4e-3: tests/test_fullsize_gpu.py).  "bf16x6" runs THAT launch with three bf16 terms (24 bits at
# every magnitude, six products, the generic kernel) -- its price is bench.py's `first_block_exact_ms_per_step`.
FIRST_BLOCK_DGRAD = os.environ.get("KEYMORPH_FIRST_BLOCK_DGRAD", "")          # "" (the mode's arithmetic) | "bf16x6"


def set_first_block_dgrad(mode: str) -> str:
    global FIRST_BLOCK_DGRAD
    assert mode in ("", "bf16x6"), mode
    prev, FIRST_BLOCK_DGRAD = FIRST_BLOCK_DGRAD, mode
    return prev


def first_block_dgrad_terms() -> int:
    """`dgrad_terms` for the second convolution of the first encoder block: 3 = bf16x6 for its data gradient, 0 = the mode's."""
    return 3 if (FIRST_BLOCK_DGRAD == "bf16x6" and CONV_MODE == "f16x3") else 0


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def channel_stats(a: Tensor, b: Optional[Tensor], N: int, V: int, C: int, only_if: Optional[Tensor] = None) -> Tensor:
    """(N,C,2) float64: mode 0 (sum a, sum a^2) if b is None else (sum a, sum a*b).
    only_if: device int32 flag; when it reads 0 the launches return at once (a fallback path gated without a host sync)."""
    lib = _lib.load()
    out = torch.empty((N, C, 2), dtype=torch.float64, device=a.device)
    ws = workspace(int(lib.kmh_channel_stats_ws_bytes(N, C)), a.device, "stats")
    check(lib.kmh_channel_stats(_p(a), _p(b), 0 if b is None else 1, N, V, C, _p(out), _p(ws), _p(only_if), _stream()),
          "kmh_channel_stats")
    return out


def _needs_range_scales() -> bool:
    return _TERMS.get(CONV_MODE, 0) == 2


_CONSTS = {}


def _const(device, *values) -> Tensor:
    """a small constant tensor on `device`, created once: torch.tensor(list, device=cuda) is a blocking host-to-device copy,
    i.e. a point where the host waits for the stream to drain (seven of them per training step before round 4)"""
    key = (str(device), values)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(values, dtype=torch.float32, device=device)
    return t


def absmax_scale(x: Tensor, min_abs: float = 0.0) -> Tensor:
    """device float[2] = {S, 1/S}: the power-of-two range scale of one operand tensor of an f16x3 convolution."""
    lib = _lib.load()
    out = _f32((2,), x.device)
    check(lib.kmh_absmax_scale(_p(x), x.numel(), float(min_abs), _p(out), _stream()), "kmh_absmax_scale")
    return out


# ---- GroupNorm statistics that travel with activations ---------------------------------------------------------
# (sum, sum^2) per (n, channel) of a tensor, produced by the pass that wrote it (the conv epilogue) or derived from
# the statistics of its sources (upsample + concat), tagged on the tensor object and guarded by its version counter.
def conv_emits_stats() -> bool:
    """KEYMORPH_NO_EPILOGUE_STATS=1 (A/B measurements only) falls back to a separate statistics pass per layer."""
    return CONV_MODE != "f32" and not os.environ.get("KEYMORPH_NO_EPILOGUE_STATS")


def _tag_stats(t: Tensor, stats: Optional[Tensor]) -> None:
    if stats is not None:
        t._kmh_stats = (stats, t._version)


def _peek_stats(t: Tensor) -> Optional[Tensor]:
    tag = getattr(t, "_kmh_stats", None)
    return tag[0] if (tag is not None and tag[1] == t._version) else None


STATS_STATS = {"carried": 0, "measured": 0}


def input_stats(x: Tensor, N: int, V: int, C: int) -> Tensor:
    s = _peek_stats(x)
    if s is not None and tuple(s.shape) == (N, C, 2):
        STATS_STATS["carried"] += 1
        return s
    STATS_STATS["measured"] += 1
    return channel_stats(x, None, N, V, C)


# ---- range scales that travel with gradients -------------------------------------------------------------------
# A backward kernel that produces a gradient tensor can emit its f16x3 range scale in the same pass (gn_bwd_apply),
# and pooling / upsample+concat backward only move or add values, so a bound follows from the incoming scale.  The
# scale rides on the tensor object as an attribute, guarded by the tensor's version counter; whenever it is missing
# (autograd summed two gradients, a hook replaced the tensor, ...) the consumer measures max|dy| itself.
def _tag_grad_scale(t: Optional[Tensor], scale2: Optional[Tensor], loosen: float = 1.0) -> None:
    """loosen >= 1 (a power of two): |t| <= loosen * (the bound scale2 was made for)."""
    if t is None or scale2 is None:
        return
    if loosen != 1.0:
        scale2 = torch.stack([scale2[0] * (1.0 / loosen), scale2[1] * loosen])
    t._kmh_dscale = (scale2, t._version)


def _peek_grad_scale(t: Tensor) -> Optional[Tensor]:
    tag = getattr(t, "_kmh_dscale", None)
    return tag[0] if (tag is not None and tag[1] == t._version) else None


def _sum_bound(a: Optional[Tensor], b: Optional[Tensor]) -> Optional[Tensor]:
    """range scale valid for x + y given the scales of x and y: S = min(Sa, Sb) / 2."""
    if a is None or b is None:
        return None
    return torch.stack([torch.minimum(a[0], b[0]) * 0.5, torch.maximum(a[1], b[1]) * 2.0])


GRAD_SCALE_STATS = {"carried": 0, "measured": 0}
RANGE_AUDIT_LOG = []          # KEYMORPH_RANGE_AUDIT=1 (debug; synchronises): one record per gradient operand of a backward conv


def range_audit(t: Tensor) -> dict:
    """Debug detector for the ONE range scale an f16x3 gradient operand carries: per sample, how
    far its largest magnitude lies below the tensor's, and the relative precision its hi + lo fp16 terms keep --
    max(2^-22, 2^-39 M / m_n): the scaled maximum sits in (2^14, 2^15] and fp16's subnormal quantum is 2^-24, so a
    sample 2^17 below the maximum still has 22 bits, one 2^30 below 9.  Also the fraction of non-zero elements that
    flush to zero altogether (|x| S < 2^-25).  Plain torch reductions: a diagnostic, not part of the product path."""
    with torch.no_grad():
        a = t.detach().abs().reshape(t.shape[0], -1).float()
        m_n = a.amax(dim=1).double()
        M = float(m_n.max())
        if M == 0.0:
            return {"max": 0.0, "per_sample_log2_below_max": [0.0] * t.shape[0], "worst_relative_precision": 2.0 ** -22,
                    "flushed_fraction": 0.0}
        S = 2.0 ** (14 - int(torch.floor(torch.log2(torch.tensor(M))).item()))       # M * S in [2^14, 2^15)
        nz = a > 0
        flushed = float(((a * S < 2.0 ** -25) & nz).sum()) / max(1.0, float(nz.sum()))
        below = [float(torch.log2(torch.tensor(M) / m)) if float(m) > 0 else float("inf") for m in m_n]
        prec = max(max(2.0 ** -22, 2.0 ** (b - 39)) for b in below if b != float("inf"))
        return {"max": M, "per_sample_log2_below_max": below, "worst_relative_precision": prec,
                "flushed_fraction": flushed}


def grad_scale(dy: Tensor) -> Tensor:
    if os.environ.get("KEYMORPH_RANGE_AUDIT") and not _is_blocked(dy):
        RANGE_AUDIT_LOG.append(dict(range_audit(dy), shape=tuple(dy.shape)))
    s = _peek_grad_scale(dy)
    if s is not None:
        GRAD_SCALE_STATS["carried"] += 1
        return s
    GRAD_SCALE_STATS["measured"] += 1
    return absmax_scale(dy)


def norm_coeffs(stats: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], N: int, C: int, G: int, V: int,
                want_ascale: bool = False):
    """-> scale, shift (N,C), mean_rstd (N,G,2) [, ascale: range scale of the normalised tensor (f16x3 mode)]."""
    lib = _lib.load()
    dev = stats.device
    scale, shift, mr = _f32((N, C), dev), _f32((N, C), dev), _f32((N, G, 2), dev)
    ascale = _f32((2,), dev) if (want_ascale and _needs_range_scales()) else None
    check(lib.kmh_gn_fwd_coeffs(_p(stats), _p(gamma), _p(beta), N, C, G, float(V), EPS, _p(scale), _p(shift),
                                _p(mr), _p(ascale), _stream()), "kmh_gn_fwd_coeffs")
    return (scale, shift, mr, ascale) if want_ascale else (scale, shift, mr)


def pack_weight(w: Tensor, transposed: bool, wscale: Optional[Tensor] = None, terms: int = 0) -> Tensor:
    """wscale: the filter's f16x3 range scale when the caller already has it (the backward of a layer re-uses the one
    its forward measured: the weights of one autograd graph do not change in between).  terms: 0 = the mode's split, 2 / 3 =
    this packing's (one launch in another arithmetic: first_block_dgrad_terms)."""
    lib = _lib.load()
    Cout, Cin = w.shape[:2]
    if CONV_MODE != "f32":
        terms = terms or _TERMS[CONV_MODE]
        out = torch.empty(int(lib.kmh_conv3d_pack_bf_bytes(Cout, Cin, int(transposed), terms)), dtype=torch.uint8,
                          device=w.device)
        out._kmh_terms = terms
        out._kmh_wscale = (wscale if wscale is not None else absmax_scale(w)) if terms == 2 else None
        check(lib.kmh_conv3d_pack_weight_bf(_p(w), _p(out), Cout, Cin, int(transposed), terms, _p(out._kmh_wscale),
                                            _stream()), "kmh_conv3d_pack_weight_bf")
        return out
    out = _f32((27, Cout, Cin) if transposed else (27, Cin, Cout), w.device)
    check(lib.kmh_conv3d_pack_weight(_p(w), _p(out), Cout, Cin, int(transposed), _stream()), "kmh_conv3d_pack_weight")
    return out


def conv3_raw(x, scale, shift, packed, bias, N, D, H, W, Cin, Cout, relu_in, relu_out, mask=None,
              ascale=None, stats_out: Optional[Tensor] = None, in_blocked: bool = False,
              addend: Optional[Tensor] = None) -> Tensor:
    """ascale: range scale of the (normalised) input for the f16x3 mode; measured here when not supplied.
    stats_out (N,Cout,2) float64: filled with the per-channel (sum y, sum y^2) of the output by the split-operand
    kernels' epilogue (the caller checks `conv_emits_stats()` first)."""
    lib = _lib.load()
    y = _f32((N, D, H, W, Cout), x.device)
    if _lib.profiler.enabled:  # algorithmic work: 2*27*Cin*Cout flops per output voxel
        _lib.profiler.meta = {"flops": 2.0 * 27 * Cin * Cout * N * D * H * W, "shape": (N, D, H, W, Cin, Cout)}
    terms = getattr(packed, "_kmh_terms", 0)
    if terms:
        if _lib.profiler.enabled:
            _lib.profiler.meta = {"flops": 2.0 * 27 * Cin * Cout * N * D * H * W, "shape": (N, D, H, W, Cin, Cout)}
        if terms == 2 and ascale is None:
            assert scale is None, "a normalised input needs the range scale of the NORMALISED tensor (norm_coeffs)"
            ascale = absmax_scale(x)
        sws = None
        if stats_out is not None:
            sws = workspace(int(lib.kmh_conv3d_fwd_bf_stats_ws_bytes(N, D, H, W, Cout, BF_ROWS_PER_WAVE)), x.device,
                            "convstats")
        check(lib.kmh_conv3d_fwd_bf(_p(x), _p(scale), _p(shift), _p(mask), _p(packed), _p(bias), _p(y), N, D, H, W,
                                    Cin, Cout, int(relu_in), int(relu_out), _t(terms), BF_ROWS_PER_WAVE,
                                    _p(ascale if terms == 2 else None), _p(packed._kmh_wscale), _p(sws), _p(stats_out),
                                    int(in_blocked), _p(addend), _stream()), "kmh_conv3d_fwd_bf")
        return y
    assert stats_out is None and addend is None, "only the split-operand kernels emit output statistics / take an addend"
    assert not in_blocked, "the channel-blocked input layout belongs to the split-operand kernels"
    check(lib.kmh_conv3d_fwd(_p(x), _p(scale), _p(shift), _p(mask), _p(packed), _p(bias), _p(y), N, D, H, W, Cin,
                             Cout, int(relu_in), int(relu_out), _stream()), "kmh_conv3d_fwd")
    return y


def conv3_wgrad(x, scale, shift, dz, N, D, H, W, Cin, Cout, relu_in, dzmask=None, xscale=None, dscale=None,
                dz_blocked: bool = False, fold: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """fold = (weight (Cout,Cin,3,3,3), bhat (N,Cin) float64 zeros): bhat[n,c] += sum_{tap,co} W dW_n, which equals
    sum_v dxn[n,v,c] * xhat[n,v,c] for the data gradient dxn of the same dz (split-operand kernels only)."""
    """xscale / dscale: range scales of the (normalised) input and of dz for the f16x3 mode (measured if absent)."""
    lib = _lib.load()
    dw = _f32((Cout, Cin, 3, 3, 3), x.device)
    if _lib.profiler.enabled:
        _lib.profiler.meta = {"flops": 2.0 * 27 * Cin * Cout * N * D * H * W, "shape": (N, D, H, W, Cin, Cout)}
    if CONV_MODE != "f32":
        terms = _TERMS[CONV_MODE]
        ws = workspace(int(lib.kmh_conv3d_wgrad_bf_ws_bytes(N, D, H, W, Cin, Cout, terms)), x.device, "wgrad")
        if terms == 2:
            if xscale is None:
                assert scale is None, "a normalised input needs the range scale of the NORMALISED tensor"
                xscale = absmax_scale(x)
            if dscale is None:
                dscale = absmax_scale(dz)
        else:
            xscale = dscale = None
        check(lib.kmh_conv3d_wgrad_bf(_p(x), _p(scale), _p(shift), _p(dz), _p(dzmask), _p(dw), N, D, H, W, Cin, Cout,
                                      int(relu_in), 0, _t(terms), 0, _p(xscale), _p(dscale), int(dz_blocked),
                                      _p(fold[0] if fold else None), _p(fold[1] if fold else None), _p(ws),
                                      _stream()), "kmh_conv3d_wgrad_bf")
        return dw
    assert fold is None, "the per-sample fold belongs to the split-operand kernels"
    assert not dz_blocked, "the channel-blocked gradient layout belongs to the split-operand kernels"
    ws = workspace(int(lib.kmh_conv3d_wgrad_ws_bytes(N, D, H, W, Cin, Cout)), x.device, "wgrad")
    check(lib.kmh_conv3d_wgrad(_p(x), _p(scale), _p(shift), _p(dz), _p(dzmask), _p(dw), N, D, H, W, Cin, Cout,
                               int(relu_in), 0, _p(ws), _stream()), "kmh_conv3d_wgrad")
    return dw


def first_layer_grads(x, scale, shift, mr, gamma, weight, dy, ymask, N, D, H, W, Cout, G, dscale=None, lazy_c123=None):
    """Backward of the FIRST U-Net conv (Cin = 1, input image needs no gradient): the correlations of dz with the RAW
    input (R = x * dz) and with the volume's indicator (S = 1 * dz) give dW = scale R + shift S and GroupNorm's
    (sum dxn, sum dxn x) = (sum W S, sum W R) from 27 x Cout numbers per sample -- the 1-channel data gradient (a full
    256^3 conv launch) is never computed.  Cout <= 16: dedicated exact-fp32 kernel (csrc/firstlayer.hip), any
    arithmetic mode; otherwise the split-operand weight-gradient kernel over the virtual 2-channel input (x, 1)."""
    lib = _lib.load()
    V = D * H * W
    dw = _f32((Cout, 1, 3, 3, 3), x.device)
    ab = torch.empty((N, 1, 2), dtype=torch.float64, device=x.device)
    if Cout <= 16:
        rs = torch.empty((N, Cout, 2, 27), dtype=torch.float64, device=x.device)      # kept in fp64 until folded
        ws = workspace(int(lib.kmh_conv3d_first_layer_wgrad_ws_bytes(N, D, H, W, Cout)), x.device, "wgrad")
        if _lib.profiler.enabled:
            _lib.profiler.meta = {"flops": 2.0 * 27 * 2 * Cout * V * N, "shape": (N, D, H, W, 1, Cout)}
        # lazy_c123: dy is the next layer's normalised-input gradient and ymask that layer's input (= this layer's output):
        # its GroupNorm backward is applied while the kernel stages the gradient (see _SingleConvGCR.backward, dx_lazy)
        check(lib.kmh_conv3d_first_layer_wgrad(_p(x), _p(dy), _p(ymask), _p(lazy_c123), _p(rs), N, D, H, W, Cout, _p(ws),
                                               _stream()), "kmh_conv3d_first_layer_wgrad")
        for n in range(N):
            check(lib.kmh_conv3d_first_layer_fold(_p(rs[n]), 1, _p(weight), _p(scale[n]), _p(shift[n]), Cout, _p(dw),
                                                  _p(ab[n]), int(n > 0), _stream()), "kmh_conv3d_first_layer_fold")
    else:
        assert lazy_c123 is None
        terms = _TERMS[CONV_MODE]
        rs = _f32((Cout, 2, 3, 3, 3), x.device)
        ws = workspace(int(lib.kmh_conv3d_wgrad_bf_ws_bytes(1, D, H, W, 2, Cout, terms)), x.device, "wgrad")
        # f16x3: the virtual channel is the constant 1, so the input's range scale must cover max(|x|, 1)
        xscale = absmax_scale(x, 1.0) if terms == 2 else None
        dscale = (dscale if dscale is not None else absmax_scale(dy)) if terms == 2 else None
        for n in range(N):
            if _lib.profiler.enabled:
                _lib.profiler.meta = {"flops": 2.0 * 27 * 1 * Cout * V, "shape": (1, D, H, W, 1, Cout)}
            check(lib.kmh_conv3d_wgrad_bf(_p(x[n]), None, None, _p(dy[n]), _p(None if ymask is None else ymask[n]),
                                          _p(rs), 1, D, H, W, 2, Cout, 0, 0, _t(terms), 1, _p(xscale), _p(dscale), 0, None, None,
                                          _p(ws), _stream()), "kmh_conv3d_wgrad_bf")
            check(lib.kmh_conv3d_first_layer_fold(_p(rs), 0, _p(weight), _p(scale[n]), _p(shift[n]), Cout, _p(dw),
                                                  _p(ab[n]), int(n > 0), _stream()), "kmh_conv3d_first_layer_fold")
    c123 = _f32((N, 1, 3), x.device)
    dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
    check(lib.kmh_gn_bwd_coeffs(_p(ab), _p(gamma), _p(mr), N, 1, G, float(V), _p(c123), _p(dgamma), _p(dbeta),
                                None, _stream()), "kmh_gn_bwd_coeffs")
    return dw, dgamma, dbeta


def conv3_up2_dgrad(dz: Tensor, weight: Tensor, Cs: int, Cl: int, dscale: Optional[Tensor] = None,
                    stats_out: Optional[Tensor] = None, dz_blocked: bool = False) -> Tensor:
    """dz (N,D,H,W,Cout), weight (Cout, Cs+Cl, 3,3,3) -> (N,D/2,H/2,W/2,Cl): for every low voxel the sum over its 8
    children of the data gradient with respect to the nearest-x2 upsampled channels [Cs, Cs+Cl) -- computed at low
    resolution with 64 pre-summed taps (csrc/conv_bf.hip: conv3_up2_dgrad)."""
    lib = _lib.load()
    N, D, H, W, Cout = dz.shape      # (a channel-blocked dz keeps the dense tensor's nominal shape)
    terms = _TERMS[CONV_MODE]
    wsu = None
    if terms == 2:
        wsu = absmax_scale(weight[:, Cs:].contiguous()) * _const(dz.device, 0.125, 8.0)
        if dscale is None:
            dscale = absmax_scale(dz)
    pk = torch.empty(int(lib.kmh_conv3d_up2_dgrad_pack_bytes(Cout, Cl, terms)), dtype=torch.uint8, device=dz.device)
    check(lib.kmh_conv3d_up2_dgrad_pack_weight(_p(weight), _p(pk), Cout, Cs + Cl, Cs, Cl, terms, _p(wsu), _stream()),
          "kmh_conv3d_up2_dgrad_pack_weight")
    ds = _f32((N, D // 2, H // 2, W // 2, Cl), dz.device)
    if _lib.profiler.enabled:
        _lib.profiler.meta = {"flops": 2.0 * 8 * Cl * Cout * N * D * H * W, "shape": (N, D, H, W, Cout, Cl)}
    sws = None
    if stats_out is not None:        # (N, Cl, 2) float64: per-channel (sum, sum of squares) of ds, from the epilogue
        sws = workspace(int(lib.kmh_conv3d_up2_dgrad_stats_ws_bytes(N, D // 2, H // 2, W // 2, Cl)), dz.device, "convstats")
    check(lib.kmh_conv3d_up2_dgrad(_p(dz), _p(pk), _p(ds), N, D // 2, H // 2, W // 2, Cl, Cout, _t(terms),
                                   _p(dscale if terms == 2 else None), _p(wsu), _p(sws), _p(stats_out), int(dz_blocked),
                                   _stream()), "kmh_conv3d_up2_dgrad")
    return ds


UPCONV_STATS = {"calls": 0}          # decoder convolutions computed without the upsampled half (tests)


def _up_sources(x):
    """(skip, low) when x is the untouched output of upcat() with exact 2x upsampling, else None."""
    tag = getattr(x, "_kmh_upsrc", None)
    if tag is None or tag[2] != x._version or tag[0]._version != tag[3] or tag[1]._version != tag[4]:
        return None
    return tag[0], tag[1]


def grad_blocked_ok(N, D, H, W, Cin, Cout) -> bool:
    """May the gradient of a (Cin -> Cout) SingleConv's OUTPUT be handed to it channel-blocked, (N, Cout/8, D, H, W, 8)?
    (f16x3 mode, the wave-specialised weight gradient takes this shape, whole 8-channel chunks.)  The data-gradient
    loader then uses whole cache lines: 8-13 % on those launches, bit-identical results."""
    if CONV_MODE != "f16x3" or os.environ.get("KEYMORPH_NO_BLOCKED_GRADS"):
        return False
    return bool(_lib.load().kmh_conv3d_wgrad_bf_blocked_ok(N, D, H, W, Cin, Cout, 2))


BLOCKED_STATS = {"handoffs": 0}      # channel-blocked gradient hand-offs performed (tests)
SPLIT_STATS = {"handoffs": 0}        # pooled gradients scattered straight into pre-split fp16 records (tests)


UP2_STATS = {"fold": 0, "boxsum": 0}      # which weight-gradient route the fused upsample + concat convolution took (tests)


def up2_wgrad_fold_ok(Cl, Cout, terms):
    """The upsampled channels' weight gradient with the box sums formed inside the product (kmh_up2_wgrad_fold);
    KEYMORPH_NO_UP2_FOLD=1: the box-sum tensor + matrix product route (the A/B arm)."""
    if os.environ.get("KEYMORPH_NO_UP2_FOLD"):
        return False
    return bool(_lib.load().kmh_up2_wgrad_fold_ok(int(Cl), int(Cout), int(terms)))


def pool_grad_split_ok(N, D, H, W, Cin, Cout) -> bool:
    """May the backward of a (Cin -> Cout) conv + pooling operator scatter the pooled gradient straight into the pre-split
    records (`kmh_maxpool3d_bwd_split`)?  Both consumers must take them: the data gradient (Cout -> Cin, the z-paired tile of
    the one-wave kernel: `kmh_conv3d_fwd_bf_split_ok`) and the wave-specialised weight gradient.  KEYMORPH_NO_SPLIT_POOLGRAD=1
    keeps the fp32 scatter (A/B runs, tests)."""
    if CONV_MODE != "f16x3" or os.environ.get("KEYMORPH_NO_SPLIT_POOLGRAD") or Cout % 8:
        return False
    lib = _lib.load()
    return bool(lib.kmh_conv3d_fwd_bf_split_ok(N, D, H, W, Cout, Cin, 2)) and \
        bool(lib.kmh_conv3d_wgrad_bf_blocked_ok(N, D, H, W, Cin, Cout, 2))


# Layout of a gradient tensor handed from one operator's backward to the next (an attribute on the tensor object, guarded by its
# version counter): kind 1 = fp32 channel-blocked (N, C/8, D, H, W, 8); kind 2 = PRE-SPLIT records (N, C/8, V + 1, 8 floats =
# 8 fp16 hi + 8 fp16 lo), what kmh_maxpool3d_bwd_split writes.  Both are "blocked"; a consumer must know which.
def _tag_blocked(t, kind: int = 1) -> None:
    t._kmh_blocked = t._version
    t._kmh_blocked_kind = int(kind)


def _is_blocked(t) -> bool:
    tag = getattr(t, "_kmh_blocked", None)
    return tag is not None and tag == t._version


def _blocked_kind(t) -> int:
    """0 = (N,D,H,W,C); 1 = fp32 channel-blocked; 2 = pre-split records"""
    return int(getattr(t, "_kmh_blocked_kind", 1)) if _is_blocked(t) else 0


@_binds_amp
class _SingleConvGCR(torch.autograd.Function):
    """y = relu(conv3(group_norm(x)))  -- all NDHWC."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, num_groups, x_from_relu, dy_premasked, dy_blocked=False, dx_blocked=False,
                pool=False, dy_lazy=False, dx_lazy=False, dgrad_terms=0):
        """dy_blocked: the ONLY consumer of y is a SingleConv called with dx_blocked=True (it returns y's gradient
        channel-blocked, see grad_blocked_ok); dx_blocked: x is the output of a SingleConv called with dy_blocked=True.
        pool: return maxpool2(y) instead of y, computed in the convolution's epilogue (conv_pool_ok): y itself is never
        written; the backward scatters the pooled gradient through the recorded winners (channel-blocked when dy_blocked)
        and continues as usual.
        dx_lazy / dy_lazy (a hand-off between the second and the FIRST convolution of the first encoder block, whose input
        image needs no gradient): the second returns its normalised-input gradient dxn UNTOUCHED, tagged with GroupNorm's
        backward coefficients and its input; the first layer's correlation kernel applies them while it stages the
        gradient (lazy_first_layer_ok) -- the pass that would write the 256^3 x 16-channel gradient is gone.
        dgrad_terms: 3 = this layer's DATA gradient runs bf16x6 whatever the mode (first_block_dgrad_terms); 0 = the mode's."""
        ctx.dgrad_terms = int(dgrad_terms) if (CONV_MODE == "f16x3" and int(dgrad_terms) == 3) else 0
        upsrc = _up_sources(x)
        x, gamma, beta, weight = _prep(x), _prep(gamma), _prep(beta), _prep(weight)
        N, D, H, W, Cin = x.shape
        Cout = weight.shape[0]
        V = D * H * W
        if upsrc is not None and not (CONV_MODE in _TERMS and _TERMS[CONV_MODE] in (2, 3) and Cout > 16
                                       and upsrc[1].shape[-1] % 8 == 0 and upsrc[0].shape[-1] % 8 == 0
                                       and not os.environ.get("KEYMORPH_NO_UPCONV")):
            upsrc = None
        stats = input_stats(x, N, V, Cin)
        scale, shift, mr, ascale = norm_coeffs(stats, gamma, beta, N, Cin, num_groups, V, want_ascale=True)
        ystats = (torch.empty((N, Cout, 2), dtype=torch.float64, device=x.device) if conv_emits_stats() else None)
        if Cin == 1 and Cout <= 16:
            # the first U-Net convolution has its own exact-fp32 kernels, forward and backward (csrc/firstlayer.hip)
            lib = _lib.load()
            y = _f32((N, D, H, W, Cout), x.device)
            ws = workspace(int(lib.kmh_conv3d_first_layer_fwd_ws_bytes(N, D, H, W, Cout)), x.device, "convstats")
            if _lib.profiler.enabled:
                _lib.profiler.meta = {"flops": 2.0 * 27 * Cin * Cout * N * D * H * W, "shape": (N, D, H, W, Cin, Cout)}
            check(lib.kmh_conv3d_first_layer_fwd(_p(x), _p(scale), _p(shift), _p(weight), _p(y), N, D, H, W, Cout, _p(ws),
                                                 _p(ystats), _stream()), "kmh_conv3d_first_layer_fwd")
        elif upsrc is not None:
            # x = cat(skip, up2(low)): the upsampled channels' 27 taps fall on 2x2x2 low-resolution voxels per output
            # parity, so their contribution comes from `low` with 8 pre-summed taps (csrc/conv_bf.hip: conv3_up2) and
            # the 27-tap kernel runs over the skip channels only, adding it in its epilogue
            skip, low = upsrc
            y = _up2_forward(skip, low, scale, shift, ascale, weight, N, D, H, W, skip.shape[-1], low.shape[-1], Cout,
                             ystats)
        elif pool:
            assert dy_premasked and ystats is not None and conv_pool_ok(N, D, H, W, Cin, Cout)
            lib = _lib.load()
            pk = pack_weight(weight, False)
            ctx.wscale = getattr(pk, "_kmh_wscale", None)
            y = _f32((N, D // 2, H // 2, W // 2, Cout), x.device)
            arg = torch.empty((N, D // 2, H // 2, W // 2, Cout), dtype=torch.uint8, device=x.device)
            sws = workspace(int(lib.kmh_conv3d_fwd_bf_stats_ws_bytes(N, D, H, W, Cout, BF_ROWS_PER_WAVE)), x.device,
                            "convstats")
            if _lib.profiler.enabled:
                _lib.profiler.meta = {"flops": 2.0 * 27 * Cin * Cout * N * D * H * W, "shape": (N, D, H, W, Cin, Cout)}
            check(lib.kmh_conv3d_fwd_bf_pool(_p(x), _p(scale), _p(shift), _p(pk), _p(y), _p(arg), N, D, H, W, Cin, Cout, 0,
                                             _t(2), _p(ascale), _p(pk._kmh_wscale), _p(sws), _p(ystats), 0, _stream()),
                  "kmh_conv3d_fwd_bf_pool")
            ctx.pool_arg = arg
            POOL_STATS["fused"] += 1
        else:
            pk = pack_weight(weight, False)
            ctx.wscale = getattr(pk, "_kmh_wscale", None)   # the data-gradient packing of the backward re-uses it
            y = conv3_raw(x, scale, shift, pk, None, N, D, H, W, Cin, Cout, False, True, ascale=ascale,
                          stats_out=ystats)
        ctx.pool = bool(pool)
        # (pool: y is the POOLED output -- saved only because the slot exists; the backward never reads it, the
        # gradient it receives being masked already)
        ctx.save_for_backward(x, y, scale, shift, mr, gamma, weight, beta)
        ctx.ascale = ascale               # range scale of the normalised input (f16x3), reused by the weight gradient
        ctx.cfg = (num_groups, bool(x_from_relu), bool(dy_premasked))
        ctx.blocked = (bool(dy_blocked), bool(dx_blocked))
        ctx.lazy = (bool(dy_lazy), bool(dx_lazy))
        assert not dy_blocked or dy_premasked, "a channel-blocked gradient comes from a SingleConv, i.e. already masked"
        if ystats is None:
            return y, None
        ctx.mark_non_differentiable(ystats)
        return y, ystats

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        lib = _lib.load()
        x, y, scale, shift, mr, gamma, weight, beta = ctx.saved_tensors
        G, x_from_relu, dy_premasked = ctx.cfg
        N, D, H, W, Cin = x.shape
        Cout = weight.shape[0]
        V = D * H * W
        dy_blocked, dx_blocked = ctx.blocked
        if _is_blocked(dy) != (dy_blocked and not ctx.pool):   # a lost or unexpected layout tag would silently scramble channels
            raise RuntimeError("keymorph_amd: gradient layout mismatch (channel-blocked tag %s, expected %s); something "
                               "between two SingleConvs replaced the gradient tensor (a hook?) -- set "
                               "KEYMORPH_NO_BLOCKED_GRADS=1 to keep every gradient in (N,D,H,W,C)"
                               % (_is_blocked(dy), dy_blocked))
        sd_in = _peek_grad_scale(dy)
        dy_lazy, dx_lazy = ctx.lazy
        lazy_tag = getattr(dy, "_kmh_lazy_gn", None)
        if (lazy_tag is not None and lazy_tag[2] == dy._version) != dy_lazy:
            raise RuntimeError("keymorph_amd: the first encoder block's lazy GroupNorm-backward hand-off lost its tag (a "
                               "hook replaced the gradient?) -- set KEYMORPH_NO_LAZY_FIRST=1")
        dy = _prep(dy)
        dy_split = False
        if ctx.pool:
            # the pooled output's gradient -> the full-resolution one through the winners recorded by the epilogue
            # (the pooling layer's backward, kmh_maxpool3d_bwd), channel-blocked when the gradient kernels take it so
            odd = (D | H | W) & 1
            dy_split = (dy_blocked and not odd and _needs_range_scales() and pool_grad_split_ok(N, D, H, W, Cin, Cout)
                        and not (Cin == 1 and not ctx.needs_input_grad[0])
                        and not ctx.dgrad_terms)            # (a bf16x6 data gradient reads the fp32 operand)
            if dy_split:
                # ... and PRE-SPLIT into the fp16 hi / lo records both gradient kernels multiply with: a scatter keeps the
                # range scale of the pooled gradient, so the split can be done by the pass that writes the tensor and the
                # z-paired data gradient copies fragments instead of converting them (kmh_maxpool3d_bwd_split)
                sd_in = sd_in if sd_in is not None else absmax_scale(dy)
                full = torch.empty((N, Cout // 8, D * H * W + 1, 8), dtype=torch.float32, device=dy.device)
                check(lib.kmh_maxpool3d_bwd_split(_p(ctx.pool_arg), _p(dy), _p(sd_in), _p(full), N, D, H, W, Cout, _stream()),
                      "kmh_maxpool3d_bwd_split")
                SPLIT_STATS["handoffs"] += 1
            else:
                full = torch.empty((N, D, H, W, Cout), dtype=torch.float32, device=dy.device)
                if odd:
                    full.zero_()
                check(lib.kmh_maxpool3d_bwd(None, _p(ctx.pool_arg), _p(dy), None, 0, _p(full), N, D, H, W, Cout,
                                            int(dy_blocked), _stream()), "kmh_maxpool3d_bwd")
            _tag_grad_scale(full, sd_in)       # scattering moves values: the bound of the pooled gradient holds
            dy = full
            if dy_blocked:
                _tag_blocked(dy, 2 if dy_split else 1)
                BLOCKED_STATS["handoffs"] += 1
        # ReLU backward (dz = dy * [y > 0]) is fused into the loaders of both gradient kernels -- and is
        # skipped altogether when every consumer of y already returned a gradient masked by (y > 0)
        # (a downstream SingleConv with x_from_relu, possibly through max-pool / upsample+concat).
        if _blocked_kind(dy) != (2 if dy_split else (1 if (dy_blocked) else 0)):
            raise RuntimeError("keymorph_amd: gradient layout kind %d where %d was expected (1 = fp32 channel-blocked, 2 = pre-split "
                               "records); a hook or an in-place op between two operators changed the tensor"
                               % (_blocked_kind(dy), 2 if dy_split else (1 if dy_blocked else 0)))
        ymask = None if dy_premasked else y
        first = Cin == 1 and not ctx.needs_input_grad[0] and (Cout <= 16 or CONV_MODE != "f32")
        dscale = (grad_scale(dy) if (_needs_range_scales() and not (first and Cout <= 16)) else None)
        if first:
            if dy_lazy:      # dy = dxn of the next layer, to be combined with that layer's input (= y) on the fly
                dw, dgamma, dbeta = first_layer_grads(x, scale, shift, mr, gamma, weight, dy, lazy_tag[1], N, D, H, W, Cout,
                                                      G, dscale=dscale, lazy_c123=lazy_tag[0])
            else:
                dw, dgamma, dbeta = first_layer_grads(x, scale, shift, mr, gamma, weight, dy, ymask, N, D, H, W, Cout, G,
                                                      dscale=dscale)
            return None, dgamma, dbeta, dw, None, None, None, None, None, None, None, None, None
        assert not dy_lazy, "only the first layer's correlation kernel applies a pending GroupNorm backward"
        need_affine = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        need_dxn = ctx.needs_input_grad[0] or need_affine
        # GroupNorm's backward statistics without a pass over dxn and x: sum dxn from the data-gradient launch's
        # epilogue, sum dxn * xhat from the per-sample weight gradient contracted with the weights (the same trick as
        # the first layer's fold); a gamma that is exactly 0 flips a device flag that un-gates the direct path
        fold = need_dxn and ctx.needs_input_grad[3] and conv_emits_stats() and not os.environ.get("KEYMORPH_NO_STATS_FOLD")
        bhat = torch.zeros((N, Cin), dtype=torch.float64, device=x.device) if fold else None
        dw = (conv3_wgrad(x, scale, shift, dy, N, D, H, W, Cin, Cout, False, dzmask=ymask, xscale=ctx.ascale,
                          dscale=dscale, dz_blocked=2 if dy_split else dy_blocked, fold=(weight, bhat) if fold else None)
              if ctx.needs_input_grad[3] else None)
        dx = dgamma = dbeta = None
        if need_dxn:
            dstats = torch.empty((N, Cin, 2), dtype=torch.float64, device=x.device) if fold else None
            if ctx.dgrad_terms:
                FIRST_BLOCK_STATS["exact_dgrads"] += 1
            dxn = conv3_raw(dy, None, None,
                            pack_weight(weight, True, None if ctx.dgrad_terms else getattr(ctx, "wscale", None), terms=ctx.dgrad_terms),
                            None, N, D, H, W, Cout, Cin, False, False,
                            mask=ymask, ascale=dscale, in_blocked=2 if dy_split else dy_blocked, stats_out=dstats)
            c123 = _f32((N, Cin, 3), x.device)
            sc2 = (torch.zeros(2, dtype=torch.float32, device=x.device)
                   if (_needs_range_scales() and ctx.needs_input_grad[0]) else None)
            dgamma = torch.zeros_like(gamma)
            dbeta = torch.zeros_like(gamma)
            flag = None
            if fold:
                flag = torch.empty(1, dtype=torch.int32, device=x.device)
                check(lib.kmh_gn_bwd_coeffs_fold(_p(dstats), _p(bhat), _p(gamma), _p(beta), _p(mr), N, Cin, G, float(V),
                                                 _p(c123), _p(dgamma), _p(dbeta), _p(flag), _stream()),
                      "kmh_gn_bwd_coeffs_fold")
                STATS_STATS["folded"] = STATS_STATS.get("folded", 0) + 1
            ab = channel_stats(dxn, x, N, V, Cin, only_if=flag)
            check(lib.kmh_gn_bwd_coeffs(_p(ab), _p(gamma), _p(mr), N, Cin, G, float(V), _p(c123), _p(dgamma),
                                        _p(dbeta), _p(flag), _stream()), "kmh_gn_bwd_coeffs")
            if ctx.needs_input_grad[0] and dx_lazy:
                assert x_from_relu and not dx_blocked
                dx = dxn
                dx._kmh_lazy_gn = (c123, x, dx._version)      # applied by the first layer's correlation kernel
                LAZY_STATS["handoffs"] += 1
            elif ctx.needs_input_grad[0]:
                # in place on dxn; the (x > 0) mask is the upstream ReLU's backward (x is a ReLU output,
                # possibly pooled / upsampled / concatenated -- all of which commute with the mask)
                dx = torch.empty_like(dxn) if dx_blocked else dxn    # another layout cannot be written in place
                check(lib.kmh_gn_bwd_apply(_p(dxn), _p(x), _p(c123), N, V, Cin, int(x_from_relu), 0, _p(dx), _p(sc2),
                                           int(dx_blocked), _stream()), "kmh_gn_bwd_apply")
                _tag_grad_scale(dx, sc2)
                if dx_blocked:
                    _tag_blocked(dx, 1)
                    BLOCKED_STATS["handoffs"] += 1
        return dx, dgamma, dbeta, dw, None, None, None, None, None, None, None, None, None


def _up2_forward(skip, low, scale, shift, ascale, weight, N, D, H, W, Cs, Cl, Cout, ystats):
    """relu(conv3(cat(norm(skip), up2(norm(low))))) without the concatenated tensor (see _SingleConvGCR.forward)."""
    lib = _lib.load()
    terms = _TERMS[CONV_MODE]
    dev = skip.device
    pk_s = pack_weight(weight[:, :Cs].contiguous(), False)
    wsu = None
    if terms == 2:   # room for the sum of 8 taps
        wsu = absmax_scale(weight[:, Cs:].contiguous()) * _const(dev, 0.125, 8.0)
    pku = torch.empty(int(lib.kmh_conv3d_up2_pack_bytes(Cout, Cl, terms)), dtype=torch.uint8, device=dev)
    check(lib.kmh_conv3d_up2_pack_weight(_p(weight), _p(pku), Cout, Cs + Cl, Cs, Cl, terms, _p(wsu), _stream()),
          "kmh_conv3d_up2_pack_weight")
    part = _f32((N, D, H, W, Cout), dev)
    if _lib.profiler.enabled:   # the work actually done: 8 taps per upsampled channel
        _lib.profiler.meta = {"flops": 2.0 * 8 * Cl * Cout * N * D * H * W, "shape": (N, D, H, W, Cl, Cout)}
    check(lib.kmh_conv3d_up2_fwd(_p(low), _p(scale), _p(shift), Cs + Cl, Cs, _p(pku), _p(part), N, D // 2, H // 2,
                                 W // 2, Cl, Cout, _t(terms), _p(ascale if terms == 2 else None), _p(wsu), _stream()),
          "kmh_conv3d_up2_fwd")
    y = conv3_raw(skip, scale[:, :Cs].contiguous(), shift[:, :Cs].contiguous(), pk_s, None, N, D, H, W, Cs, Cout,
                  False, True, ascale=ascale, stats_out=ystats, addend=part)
    UPCONV_STATS["calls"] += 1
    return y


def upcat_conv_ok(skip, low, Cout) -> bool:
    """May relu(conv3(group_norm(cat(skip, up2(low))))) run as the fused operator (no concatenated tensor, the
    upsampled channels handled at low resolution, forward and backward)?  Also under no_grad: the evaluation scripts
    (scripts/register.py, pairwise_register_eval.py:116-171) run the same forward."""
    return (CONV_MODE in _TERMS and conv_emits_stats() and Cout > 16 and Cout % 4 == 0 and skip.shape[-1] % 8 == 0
            and low.shape[-1] % 8 == 0 and all(a == 2 * b for a, b in zip(skip.shape[1:4], low.shape[1:4]))
            and not os.environ.get("KEYMORPH_NO_UPCONV")
            and not os.environ.get("KEYMORPH_NO_UPCONV_BWD"))


@_binds_amp
class _UpCatConvGCR(torch.autograd.Function):
    """y = relu(conv3(group_norm(cat(skip, nearest_up2(low)))))  (the decoder's first SingleConv fused with the
    interpolate + cat in front of it, keymorph/unet3d/buildingblocks.py:471-475 + 46-78).  Neither direction forms the
    concatenated tensor except the weight gradient (for now): the upsampled channels' forward uses 8 pre-summed taps
    per output parity on `low`, their data gradient 64 pre-summed taps at low resolution (already summed over the 8
    children, i.e. interpolate's backward), and GroupNorm's backward is applied to the two halves separately."""

    @staticmethod
    def forward(ctx, skip, low, gamma, beta, weight, num_groups, dy_premasked, dskip_lazy=False, dy_blocked=False):
        """dskip_lazy: `skip` is the second output of pool_fork (its gradient goes to _PoolFork.backward and nowhere else):
        the skip half's normalised-input gradient is returned with GroupNorm's backward pending, and the pooling backward
        applies it while it sums the two gradients (kmh_maxpool3d_bwd_lazy).
        dy_blocked: the ONLY consumer of y is a SingleConv called with dx_blocked=True, which returns y's gradient
        channel-blocked (upcat_blocked_ok): all four gradient kernels then read whole cache lines of it."""
        ctx.dskip_lazy = bool(dskip_lazy)
        ctx.dy_blocked = bool(dy_blocked)
        assert not dy_blocked or dy_premasked, "a channel-blocked gradient comes from a SingleConv, i.e. already masked"
        skip, low, gamma, beta, weight = _prep(skip), _prep(low), _prep(gamma), _prep(beta), _prep(weight)
        N, D, H, W, Cs = skip.shape
        Cl, Cout = low.shape[-1], weight.shape[0]
        V = D * H * W
        stats = torch.cat([input_stats(skip, N, V, Cs), input_stats(low, N, V // 8, Cl) * 8.0], dim=1)
        scale, shift, mr, ascale = norm_coeffs(stats, gamma, beta, N, Cs + Cl, num_groups, V, want_ascale=True)
        ystats = torch.empty((N, Cout, 2), dtype=torch.float64, device=skip.device)
        y = _up2_forward(skip, low, scale, shift, ascale, weight, N, D, H, W, Cs, Cl, Cout, ystats)
        ctx.save_for_backward(skip, low, y, scale, shift, mr, gamma, weight, beta)
        ctx.ascale = ascale
        ctx.cfg = (num_groups, bool(dy_premasked))
        ctx.mark_non_differentiable(ystats)
        return y, ystats

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        lib = _lib.load()
        skip, low, y, scale, shift, mr, gamma, weight, beta = ctx.saved_tensors
        G, dy_premasked = ctx.cfg
        N, D, H, W, Cs = skip.shape
        Cl, Cout = low.shape[-1], weight.shape[0]
        C, V = Cs + Cl, D * H * W
        blk = ctx.dy_blocked
        if _is_blocked(dy) != blk:      # a lost or unexpected layout tag would silently scramble channels
            raise RuntimeError("keymorph_amd: gradient layout mismatch at the fused upsample+concat convolution (channel-"
                               "blocked tag %s, expected %s); something replaced the gradient tensor (a hook?) -- set "
                               "KEYMORPH_NO_BLOCKED_GRADS=1 to keep every gradient in (N,D,H,W,C)" % (_is_blocked(dy), blk))
        dy = _prep(dy)
        if not dy_premasked:     # fold the ReLU mask once (this operator's gradient kernels take no mask operand)
            dzm = torch.empty_like(dy)
            check(lib.kmh_relu_mask(_p(dy), _p(y), dy.numel(), _p(dzm), _stream()), "kmh_relu_mask")
            dy = dzm
        dscale = grad_scale(dy) if _needs_range_scales() else None
        # weight gradient (+ the per-sample fold for GroupNorm).  Skip channels: the 27-tap kernel on `skip`.  Upsampled
        # channels: dW[tap] = sum_m x_low[m] G[m][tap] with G the 2x2x2 box sums of dz (kmh_up2_boxsum) -- one plain
        # matrix product over the low-resolution voxels per sample (1/8 of the multiply-adds; library fp32 GEMM)
        bhat_s = torch.zeros((N, Cs), dtype=torch.float64, device=dy.device)
        dw_s = conv3_wgrad(skip, scale[:, :Cs].contiguous(), shift[:, :Cs].contiguous(), dy, N, D, H, W, Cs, Cout, False,
                           xscale=ctx.ascale, dscale=dscale, dz_blocked=blk, fold=(weight[:, :Cs].contiguous(), bhat_s))
        if Cout % 4 == 0 and not os.environ.get("KEYMORPH_NO_UPCONV_WGRAD"):
            Vl = V // 8
            sc_l, sh_l = scale[:, Cs:].contiguous(), shift[:, Cs:].contiguous()   # named: they must outlive the launch
            terms = _TERMS[CONV_MODE]
            dwn = _f32((N, Cl, 27, Cout), dy.device)
            if _lib.profiler.enabled:
                _lib.profiler.meta = {"flops": 2.0 * 27 * Cl * Cout * N * Vl, "shape": (N, Vl, Cl, 27 * Cout)}
            if up2_wgrad_fold_ok(Cl, Cout, terms):
                # round 5: the box sums are formed inside the product (from LDS), never stored
                UP2_STATS["fold"] += 1
                gws = workspace(int(lib.kmh_up2_wgrad_fold_ws_bytes(N, D // 2, H // 2, W // 2, Cl, Cout)), dy.device, "wgrad")
                check(lib.kmh_up2_wgrad_fold(_p(low), _p(dy), _p(dwn), N, D // 2, H // 2, W // 2, Cl, Cout, _t(terms), _p(ctx.ascale),
                                             _p(dscale), _p(sc_l), _p(sh_l), int(blk), _p(gws), _stream()), "kmh_up2_wgrad_fold")
            else:
                UP2_STATS["boxsum"] += 1
                boxes = _f32((N, Vl, 27 * Cout), dy.device)
                check(lib.kmh_up2_boxsum(_p(dy), _p(boxes), N, D // 2, H // 2, W // 2, Cout, int(blk), _stream()), "kmh_up2_boxsum")
                gws = workspace(int(lib.kmh_up2_wgrad_gemm_ws_bytes(N, Vl, Cl, 27 * Cout)), dy.device, "wgrad")
                bsc = (dscale * _const(dy.device, 0.125, 8.0)) if terms == 2 else None   # sums of 8
                # (the raw low tensor: GroupNorm's affine is applied while the product stages it)
                check(lib.kmh_up2_wgrad_gemm(_p(low), _p(boxes), _p(dwn), N, Vl, Cl, 27 * Cout, _t(terms),
                                             _p(ctx.ascale if terms == 2 else None), _p(bsc), _p(sc_l), _p(sh_l), _p(gws),
                                             _stream()), "kmh_up2_wgrad_gemm")
                del boxes
            dw_l = dwn.sum(0).permute(2, 0, 1).reshape(Cout, Cl, 3, 3, 3)
            wl = weight[:, Cs:].reshape(Cout, Cl, 27).permute(1, 2, 0)                         # (Cl, 27, Cout)
            bhat_l = (dwn.double() * wl.double().unsqueeze(0)).sum(dim=(2, 3))
        else:   # rebuild the upsampled half and use the 27-tap kernel
            xu = _f32((N, D, H, W, Cl), dy.device)
            check(lib.kmh_upcat_fwd(_p(skip), _p(low), _p(xu), N, D, H, W, 0, D // 2, H // 2, W // 2, Cl, _stream()),
                  "kmh_upcat_fwd")
            bhat_l = torch.zeros((N, Cl), dtype=torch.float64, device=dy.device)
            dw_l = conv3_wgrad(xu, scale[:, Cs:].contiguous(), shift[:, Cs:].contiguous(), dy, N, D, H, W, Cl, Cout,
                               False, xscale=ctx.ascale, dscale=dscale, dz_blocked=blk,
                               fold=(weight[:, Cs:].contiguous(), bhat_l))
            del xu
        dw = torch.cat([dw_s, dw_l], dim=1)
        bhat = torch.cat([bhat_s, bhat_l], dim=1)
        # data gradient: skip channels at full resolution (27 taps), upsampled channels at low resolution (64 taps)
        dst_s = torch.empty((N, Cs, 2), dtype=torch.float64, device=dy.device)
        dxn_s = conv3_raw(dy, None, None, pack_weight(weight[:, :Cs].contiguous(), True), None, N, D, H, W, Cout, Cs,
                          False, False, ascale=dscale, stats_out=dst_s, in_blocked=blk)
        dst_l = torch.empty((N, Cl, 2), dtype=torch.float64, device=dy.device)
        dsum_l = conv3_up2_dgrad(dy, weight, Cs, Cl, dscale, stats_out=dst_l, dz_blocked=blk)
        dstats = torch.cat([dst_s, dst_l], dim=1)
        c123 = _f32((N, C, 3), dy.device)
        dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
        flag = torch.empty(1, dtype=torch.int32, device=dy.device)
        check(lib.kmh_gn_bwd_coeffs_fold(_p(dstats), _p(bhat), _p(gamma), _p(beta), _p(mr), N, C, G, float(V), _p(c123),
                                         _p(dgamma), _p(dbeta), _p(flag), _stream()), "kmh_gn_bwd_coeffs_fold")
        STATS_STATS["folded"] = STATS_STATS.get("folded", 0) + 1
        # gamma == 0 somewhere: the direct statistics, gated on the device (sum dxn x over the upsampled channels is
        # sum over low voxels of (children's sum) * x_low)
        ab = torch.cat([channel_stats(dxn_s, skip, N, V, Cs, only_if=flag),
                        channel_stats(dsum_l, low, N, V // 8, Cl, only_if=flag)], dim=1)
        check(lib.kmh_gn_bwd_coeffs(_p(ab), _p(gamma), _p(mr), N, C, G, float(V), _p(c123), _p(dgamma), _p(dbeta),
                                    _p(flag), _stream()), "kmh_gn_bwd_coeffs")
        # dx = mask * (c1 dxn + c2 x + c3); summed over 8 children for the upsampled half: c1 S + 8 c2 x_low + 8 c3
        c_s = c123[:, :Cs].contiguous()
        c_l = (c123[:, Cs:] * _const(dy.device, 1.0, 8.0, 8.0)).contiguous()
        want = _needs_range_scales()
        sc_s = torch.zeros(2, dtype=torch.float32, device=dy.device) if want else None
        sc_l = torch.zeros(2, dtype=torch.float32, device=dy.device) if want else None
        dskip = dlow = None
        if ctx.needs_input_grad[0] and ctx.dskip_lazy:
            dskip = dxn_s
            dskip._kmh_lazy_gn = (c_s, skip, dskip._version)       # applied by _PoolFork.backward
            LAZY_STATS["handoffs"] += 1
        elif ctx.needs_input_grad[0]:
            check(lib.kmh_gn_bwd_apply(_p(dxn_s), _p(skip), _p(c_s), N, V, Cs, 1, 0, _p(dxn_s), _p(sc_s), 0, _stream()),
                  "kmh_gn_bwd_apply")
            dskip = dxn_s
            _tag_grad_scale(dskip, sc_s)
        if ctx.needs_input_grad[1]:
            check(lib.kmh_gn_bwd_apply(_p(dsum_l), _p(low), _p(c_l), N, V // 8, Cl, 1, 0, _p(dsum_l), _p(sc_l), 0,
                                       _stream()), "kmh_gn_bwd_apply")
            dlow = dsum_l
            _tag_grad_scale(dlow, sc_l)
        return dskip, dlow, dgamma, dbeta, dw, None, None, None, None


def upcat_blocked_ok(skip, low, Cout) -> bool:
    """May the fused operator's OUTPUT gradient arrive channel-blocked?  (the skip half's 27-tap weight gradient must take
    that layout -- grad_blocked_ok -- and the upsampled half goes through the box sums, Cout % 8 == 0.)"""
    N, D, H, W, Cs = skip.shape
    return (torch.is_grad_enabled() and Cout % 8 == 0 and not os.environ.get("KEYMORPH_NO_UPCONV_WGRAD")
            and not os.environ.get("KEYMORPH_NO_BLOCKED_UPCAT") and grad_blocked_ok(N, D, H, W, Cs, Cout))


def upcat_conv_gcr(skip, low, gamma, beta, weight, num_groups: int, dy_premasked: bool = False,
                   dskip_lazy: bool = False, dy_blocked: bool = False) -> Tensor:
    if dskip_lazy:
        # the hand-off is a contract between THIS operator's backward and pool_fork's: record it on pool_fork's node, so that
        # its backward can tell "no tag because nothing was pending" from "the tag was lost on the way" (a hook, retain_grad
        # or a second consumer of `skip` makes autograd re-wrap or sum the gradient, and the attribute does not survive)
        node = skip.grad_fn
        if node is not None and type(node).__name__ == "_PoolForkBackward":
            node._kmh_expect_lazy = True
        else:                          # not pool_fork's second output after all: apply GroupNorm's backward here
            dskip_lazy = False
    y, ystats = _UpCatConvGCR.apply(skip, low, gamma, beta, weight, num_groups, dy_premasked, dskip_lazy, dy_blocked)
    _tag_stats(y, ystats)
    return y


def single_conv_gcr(x, gamma, beta, weight, num_groups: int, x_from_relu: bool = True,
                    dy_premasked: bool = False, dy_blocked: bool = False, dx_blocked: bool = False,
                    pool: bool = False, dy_lazy: bool = False, dx_lazy: bool = False, dgrad_terms: int = 0) -> Tensor:
    """dy_premasked: promise that the gradient arriving for the output is already zero wherever the output is
    <= 0 (true when all consumers are SingleConvs with x_from_relu=True).
    pool: return maxpool2 of the output (see conv_pool_ok); the statistics tagged on it are the pooled tensor's."""
    y, ystats = _SingleConvGCR.apply(x, gamma, beta, weight, num_groups, x_from_relu, dy_premasked, dy_blocked,
                                     dx_blocked, pool, dy_lazy, dx_lazy, dgrad_terms)
    _tag_stats(y, ystats)       # the next GroupNorm's statistics came with the epilogue
    return y


POOL_STATS = {"fused": 0}           # convolutions that pooled in their epilogue (tests)
FIRST_BLOCK_STATS = {"exact_dgrads": 0}   # data gradients run bf16x6 by the first-block selector (tests)
LAZY_STATS = {"handoffs": 0}        # GroupNorm backwards applied inside the first layer's correlation kernel (tests)


def lazy_skip_ok(skip) -> bool:
    """May the fused decoder operator hand the skip half's gradient to pool_fork's backward with GroupNorm's backward
    pending?  (dense fp32 skip tensor with even D, H, W and whole channel quads)"""
    return (torch.is_grad_enabled() and skip.shape[1] % 2 == 0 and skip.shape[2] % 2 == 0 and skip.shape[3] % 2 == 0
            and skip.shape[4] % 4 == 0 and not os.environ.get("KEYMORPH_NO_LAZY_SKIP"))


def lazy_first_layer_ok(x, cout1: int) -> bool:
    """May the SECOND convolution of the first encoder block hand its normalised-input gradient to the FIRST one with
    GroupNorm's backward still pending?  The first layer must be the 1 -> Cout <= 16 layer with the dedicated kernels
    and its input must need no gradient (the image)."""
    return (x.shape[-1] == 1 and cout1 <= 16 and not x.requires_grad and torch.is_grad_enabled()
            and not os.environ.get("KEYMORPH_NO_LAZY_FIRST"))


def conv_pool_ok(N, D, H, W, Cin, Cout) -> bool:
    """May relu(conv3(group_norm(x))) be followed by MaxPool3d(2) inside the convolution's epilogue (the output feeds
    ONLY that pooling and its gradient arrives masked)?  f16x3 mode, 16 < Cout <= 32, the LDS-DMA kernel selected."""
    if CONV_MODE != "f16x3" or not conv_emits_stats() or os.environ.get("KEYMORPH_NO_CONV_POOL"):
        return False
    return bool(_lib.load().kmh_conv3d_fwd_bf_pool_ok(N, D, H, W, Cin, Cout, 2))


_NO_ADD = object()


def _maxpool_fwd(ctx, x):
    """y = maxpool2(x); when a gradient will be needed the winners are recorded (1 byte per pooled element) so that
    the backward does not re-read the full-resolution input."""
    lib = _lib.load()
    x = _prep(x)
    N, D, H, W, C = x.shape
    y = _f32((N, D // 2, H // 2, W // 2, C), x.device)
    arg = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None
    check(lib.kmh_maxpool3d_fwd(_p(x), _p(y), _p(arg), N, D, H, W, C, _stream()), "kmh_maxpool3d_fwd")
    ctx.xshape = tuple(x.shape)
    if arg is not None:
        ctx.save_for_backward(arg)
    return x, y


def _maxpool_bwd(ctx, dy, add, out_blocked=False):
    """dx = scatter(dy) [+ add]; add may be a channel-strided view (the first Cs channels of a wider NDHWC tensor).
    out_blocked: dx is written channel-blocked for a SingleConv that takes its gradient that way (even dims)."""
    lib = _lib.load()
    (arg,) = ctx.saved_tensors
    N, D, H, W, C = ctx.xshape
    odd = (D % 2) or (H % 2) or (W % 2)
    acs = 0
    add_tag = _NO_ADD if add is None else _peek_grad_scale(add)
    if add is not None:
        acs = add.stride(3)
        dense_voxels = add.stride() == (D * H * W * acs, H * W * acs, W * acs, acs, 1)
        if odd or add.dtype != torch.float32 or not dense_voxels:
            add, acs = add.contiguous().float().clone(), C      # pre-filled output, accumulated in place
    if add is not None and acs == C and odd:
        dx = add
    else:
        dx = (torch.zeros if odd else torch.empty)((N, D, H, W, C), dtype=torch.float32, device=arg.device)
    dy_c = _prep(dy)
    check(lib.kmh_maxpool3d_bwd(None, _p(arg), _p(dy_c), _p(add), acs, _p(dx), N, D, H, W, C, int(out_blocked),
                                _stream()), "kmh_maxpool3d_bwd")
    if out_blocked:
        _tag_blocked(dx, 1)
        BLOCKED_STATS["handoffs"] += 1
    # scattering moves values: the bound of dy holds for dx (plus the skip gradient's bound when that is added)
    sd = _peek_grad_scale(dy)
    _tag_grad_scale(dx, sd if add_tag is _NO_ADD else _sum_bound(sd, add_tag))
    return dx


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blocked_grad):
        ctx.blocked_grad = bool(blocked_grad)
        return _maxpool_fwd(ctx, x)[1]

    @staticmethod
    def backward(ctx, dy):
        return _maxpool_bwd(ctx, dy, None, ctx.blocked_grad), None


def maxpool2(x: Tensor, blocked_grad: bool = False) -> Tensor:
    """blocked_grad: x's ONLY other role is to be the output of a SingleConv called with dy_blocked=True, which then
    receives its gradient channel-blocked (see grad_blocked_ok); needs even D, H, W."""
    return _MaxPool2.apply(x, blocked_grad)


class _PoolFork(torch.autograd.Function):
    """(maxpool2(x), x): the encoder output that feeds both the next level and a decoder's skip connection.  One
    backward pass sums the two gradients (autograd would materialise the scattered pool gradient and add)."""

    @staticmethod
    def forward(ctx, x):
        x, y = _maxpool_fwd(ctx, x)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        lazy = getattr(dskip, "_kmh_lazy_gn", None) if dskip is not None else None
        if lazy is not None and lazy[2] != dskip._version:
            raise RuntimeError("keymorph_amd: the skip gradient's pending GroupNorm backward lost its tag (a hook modified "
                               "the gradient?) -- set KEYMORPH_NO_LAZY_SKIP=1")
        if lazy is None and dskip is not None and getattr(ctx, "_kmh_expect_lazy", False):
            # the decoder handed over dxn with GroupNorm's backward PENDING, but what arrived is not that tensor (a hook or
            # retain_grad on the skip tensor, or a second consumer whose gradient autograd added): summing it as a finished
            # gradient would be silently wrong
            raise RuntimeError("keymorph_amd: pool_fork expected the skip gradient with its GroupNorm backward pending, but the "
                               "tensor that arrived carries no such tag (hook / retain_grad / second consumer of the skip "
                               "tensor?) -- set KEYMORPH_NO_LAZY_SKIP=1 to apply GroupNorm's backward in the decoder")
        if lazy is not None:
            lib = _lib.load()
            c123, x, _ = lazy
            N, D, H, W, C = ctx.xshape
            dxn = _prep(dskip)
            sc2 = torch.zeros(2, dtype=torch.float32, device=dxn.device) if _needs_range_scales() else None
            if dy is None:                       # no pooled gradient: just the pending apply
                check(lib.kmh_gn_bwd_apply(_p(dxn), _p(x), _p(c123), N, D * H * W, C, 1, 0, _p(dxn), _p(sc2), 0, _stream()),
                      "kmh_gn_bwd_apply")
                _tag_grad_scale(dxn, sc2)
                return dxn
            (arg,) = ctx.saved_tensors
            dx = torch.empty((N, D, H, W, C), dtype=torch.float32, device=dxn.device)
            check(lib.kmh_maxpool3d_bwd_lazy(_p(arg), _p(_prep(dy)), _p(dxn), _p(x), _p(c123), _p(dx), N, D, H, W, C, _p(sc2),
                                             _stream()), "kmh_maxpool3d_bwd_lazy")
            _tag_grad_scale(dx, sc2)
            return dx
        if dy is None:
            return dskip
        return _maxpool_bwd(ctx, dy, dskip)


def pool_fork(x: Tensor):
    """-> (maxpool2(x), skip): use `skip` (not x) as the decoder's skip input."""
    y, skip = _PoolFork.apply(x)
    _tag_stats(skip, _peek_stats(x))
    return y, skip


class _UpCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, low, lazy_skip_grad):
        lib = _lib.load()
        skip, low = _prep(skip), _prep(low)
        N, D, H, W, Cs = skip.shape
        _, Dl, Hl, Wl, Cl = low.shape
        out = _f32((N, D, H, W, Cs + Cl), skip.device)
        check(lib.kmh_upcat_fwd(_p(skip), _p(low), _p(out), N, D, H, W, Cs, Dl, Hl, Wl, Cl, _stream()), "kmh_upcat_fwd")
        ctx.dims = (N, D, H, W, Cs, Dl, Hl, Wl, Cl)
        ctx.lazy = bool(lazy_skip_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        N, D, H, W, Cs, Dl, Hl, Wl, Cl = ctx.dims
        dout = _prep(dout)
        # lazy: the skip gradient is the channel-strided view dout[..., :Cs]; its consumer (pool_fork) reads it in place
        dskip = None if ctx.lazy else _f32((N, D, H, W, Cs), dout.device)
        dlow = _f32((N, Dl, Hl, Wl, Cl), dout.device)
        check(lib.kmh_upcat_bwd(_p(dout), _p(dskip), _p(dlow), N, D, H, W, Cs, Dl, Hl, Wl, Cl, 0, _stream()),
              "kmh_upcat_bwd")
        if ctx.lazy:
            dskip = dout[..., :Cs]
        sd = _peek_grad_scale(dout)
        _tag_grad_scale(dskip, sd)                      # a subset of dout's values
        if (D, H, W) == (2 * Dl, 2 * Hl, 2 * Wl):
            _tag_grad_scale(dlow, sd, loosen=8.0)       # each coarse voxel sums exactly 8 fine ones
        return dskip, dlow, None


def upcat(skip: Tensor, low: Tensor, lazy_skip_grad: bool = False) -> Tensor:
    """lazy_skip_grad: `skip` comes from pool_fork, whose backward consumes a strided gradient view without a copy."""
    out = _UpCat.apply(skip, low, lazy_skip_grad)
    ss, sl = _peek_stats(skip), _peek_stats(low)
    if ss is not None and sl is not None and all(a == 2 * b for a, b in zip(skip.shape[1:4], low.shape[1:4])):
        # exact 2x nearest upsampling replicates every coarse voxel 8 times: the sums are linear in the sources
        _tag_stats(out, torch.cat([ss, sl * 8.0], dim=1))
    if all(a == 2 * b for a, b in zip(skip.shape[1:4], low.shape[1:4])) and skip.is_contiguous() and low.is_contiguous():
        # the consumer (a SingleConv) can compute the upsampled half from `low` itself (8 taps instead of 27)
        out._kmh_upsrc = (skip.detach(), low.detach(), out._version, skip._version, low._version)
    return out


class _Pointwise(torch.autograd.Function):
    """x NDHWC (N,D,H,W,Cin), w (Cout,Cin,1,1,1), b (Cout) -> y NCDHW (N,Cout,D,H,W)."""

    @staticmethod
    def forward(ctx, x, w, b):
        lib = _lib.load()
        x, w = _prep(x), _prep(w)
        b = None if b is None else _prep(b)
        N, D, H, W, Cin = x.shape
        Cout = w.shape[0]
        V = D * H * W
        wt = _f32((Cin, Cout), x.device)
        check(lib.kmh_pointwise_pack(_p(w), _p(wt), Cout, Cin, _stream()), "kmh_pointwise_pack")
        y = _f32((N, Cout, D, H, W), x.device)
        check(lib.kmh_pointwise_fwd(_p(x), _p(wt), _p(b), _p(y), N, V, Cin, Cout, _stream()), "kmh_pointwise_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        N, D, H, W, Cin = x.shape
        Cout = w.shape[0]
        V = D * H * W
        dy = _prep(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.kmh_pointwise_dgrad(_p(dy), _p(w), _p(dx), N, V, Cin, Cout, _stream()), "kmh_pointwise_dgrad")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = _f32((Cout,), x.device) if ctx.has_bias else None
            ws = workspace(int(lib.kmh_pointwise_wgrad_ws_bytes(N, V, Cin, Cout)), x.device, "wgrad")
            check(lib.kmh_pointwise_wgrad(_p(dy), _p(x), _p(dw), _p(db), N, V, Cin, Cout, 0, _p(ws), _stream()),
                  "kmh_pointwise_wgrad")
        return dx, dw, db


def pointwise(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    return _Pointwise.apply(x, w, b)


def to_ndhwc(x: Tensor) -> Tensor:
    """(N,C,D,H,W) contiguous -> (N,D,H,W,C) contiguous (free for C == 1)."""
    x = _prep(x)
    N, C = x.shape[:2]
    if C == 1:
        return x.reshape(N, *x.shape[2:], 1)
    return _Layout.apply(x, False)


def to_ncdhw(x: Tensor) -> Tensor:
    x = _prep(x)
    if x.shape[-1] == 1:
        return x.reshape(x.shape[0], 1, *x.shape[1:4])
    return _Layout.apply(x, True)


class _Layout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, to_ncdhw_):
        lib = _lib.load()
        if to_ncdhw_:
            N, D, H, W, C = x.shape
            out = _f32((N, C, D, H, W), x.device)
        else:
            N, C, D, H, W = x.shape
            out = _f32((N, D, H, W, C), x.device)
        check(lib.kmh_layout_convert(_p(x), _p(out), N, D * H * W, C, int(to_ncdhw_), _stream()), "kmh_layout_convert")
        ctx.to_ncdhw_ = to_ncdhw_
        return out

    @staticmethod
    def backward(ctx, g):
        return _Layout.apply(_prep(g), not ctx.to_ncdhw_), None


@_binds_amp
class _ConvBlock(torch.autograd.Function):
    """ConvNet block (keymorph/layers.py:137-187): Conv3d(k3,p1,bias) -> [InstanceNorm3d(affine=False) |
    GroupNorm(8) | BatchNorm3d | none] -> ReLU.  x, y NDHWC.  (MaxPool is a separate op, like in the reference.)

    bn = 1: BatchNorm3d in training mode.  Its statistics run over (N, D, H, W) per channel, and an NDHWC batch IS one
    sample of N*D planes in memory, so it is the instance norm of that one "sample" with an affine (gamma, beta): the
    same kernels with N' = 1, V' = N*V.  The batch statistics are returned for the running averages.
    bn = 2: BatchNorm3d in evaluation mode: a fixed per-channel affine map from (running_mean, running_var)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, groups, bn=0, rmean=None, rvar=None, eps=EPS):
        lib = _lib.load()
        x, weight, bias = _prep(x), _prep(weight), _prep(bias)
        N, D, H, W, Cin = x.shape
        Cout = weight.shape[0]
        V = D * H * W
        z = conv3_raw(x, None, None, pack_weight(weight, False), bias, N, D, H, W, Cin, Cout, False, groups == 0 and bn == 0)
        ctx.bn = bn
        if groups == 0 and bn == 0:  # norm_type == "none": conv -> ReLU
            ctx.save_for_backward(x, weight, z)
            ctx.cfg = (0,)
            return z
        if bn == 2:
            rstd = torch.rsqrt(rvar.double() + eps)
            sc1 = (gamma.double() * rstd).float()
            sh1 = (beta.double() - rmean.double() * gamma.double() * rstd).float()
            scale, shift = sc1[None].expand(N, Cout).contiguous(), sh1[None].expand(N, Cout).contiguous()
            y = torch.empty_like(z)
            check(lib.kmh_norm_apply(_p(z), _p(scale), _p(shift), N, V, Cout, 1, _p(y), _stream()), "kmh_norm_apply")
            ctx.save_for_backward(x, weight, z, y, scale, gamma, rmean.float(), rstd.float())
            ctx.cfg = (0,)
            return y
        Nn, Vn = (1, N * V) if bn == 1 else (N, V)
        if bn == 1:
            groups = Cout
        stats = channel_stats(z, None, Nn, Vn, Cout)
        # (kmh_gn_fwd_coeffs takes its epsilon from the module constant: every norm layer of the reference uses 1e-5)
        scale, shift, mr = norm_coeffs(stats, gamma, beta, Nn, Cout, groups, Vn)
        y = torch.empty_like(z)
        check(lib.kmh_norm_apply(_p(z), _p(scale), _p(shift), Nn, Vn, Cout, 1, _p(y), _stream()), "kmh_norm_apply")
        saved = [x, weight, z, y, mr] + ([gamma] if gamma is not None else [])
        ctx.save_for_backward(*saved)
        ctx.cfg = (groups,)
        if bn == 1:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        lib = _lib.load()
        (groups,) = ctx.cfg
        dy = _prep(dy)
        dgamma = dbeta = None
        if ctx.bn == 2:
            x, weight, z, y, scale, gamma, rmean, rstd = ctx.saved_tensors
            N, D, H, W, Cin = x.shape
            Cout = weight.shape[0]
            V = D * H * W
            dym = torch.empty_like(dy)
            check(lib.kmh_relu_mask(_p(dy), _p(y), dy.numel(), _p(dym), _stream()), "kmh_relu_mask")
            ab = channel_stats(dym, z, N, V, Cout).sum(0)                 # (Cout, 2): sum dym, sum dym * z
            dbeta = ab[:, 0].float()
            dgamma = ((ab[:, 1] - rmean.double() * ab[:, 0]) * rstd.double()).float()
            c123 = torch.zeros((N, Cout, 3), dtype=torch.float32, device=x.device)
            c123[:, :, 0] = scale
            check(lib.kmh_gn_bwd_apply(_p(dym), _p(z), _p(c123), N, V, Cout, 0, 0, _p(dym), None, 0, _stream()),
                  "kmh_gn_bwd_apply")
            dz, dzmask = dym, None
        elif groups == 0:
            x, weight, y = ctx.saved_tensors
            N, D, H, W, Cin = x.shape
            Cout = weight.shape[0]
            V = D * H * W
            dz, dzmask = dy, y
        else:
            saved = ctx.saved_tensors
            x, weight, z, y, mr = saved[:5]
            gamma = saved[5] if len(saved) > 5 else None
            N, D, H, W, Cin = x.shape
            Cout = weight.shape[0]
            V = D * H * W
            Nn, Vn = (1, N * V) if ctx.bn == 1 else (N, V)
            dym = torch.empty_like(dy)
            check(lib.kmh_relu_mask(_p(dy), _p(y), dy.numel(), _p(dym), _stream()), "kmh_relu_mask")
            ab = channel_stats(dym, z, Nn, Vn, Cout)
            c123 = _f32((Nn, Cout, 3), x.device)
            if gamma is not None:
                dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
            check(lib.kmh_gn_bwd_coeffs(_p(ab), _p(gamma), _p(mr), Nn, Cout, groups, float(Vn), _p(c123), _p(dgamma),
                                        _p(dbeta), None, _stream()), "kmh_gn_bwd_coeffs")
            check(lib.kmh_gn_bwd_apply(_p(dym), _p(z), _p(c123), Nn, Vn, Cout, 0, 0, _p(dym), None, 0, _stream()),
                  "kmh_gn_bwd_apply")
            dz, dzmask = dym, None
        dw = conv3_wgrad(x, None, None, dz, N, D, H, W, Cin, Cout, False, dzmask=dzmask)
        if dzmask is not None:
            dzm = torch.empty_like(dz)
            check(lib.kmh_relu_mask(_p(dz), _p(dzmask), dz.numel(), _p(dzm), _stream()), "kmh_relu_mask")
            db = channel_stats(dzm, None, N, V, Cout)[:, :, 0].sum(0).float()
        else:
            db = channel_stats(dz, None, N, V, Cout)[:, :, 0].sum(0).float()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv3_raw(dz, None, None, pack_weight(weight, True), None, N, D, H, W, Cout, Cin, False, False,
                           mask=dzmask)
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None


def conv_block(x, weight, bias, gamma=None, beta=None, groups: int = 0) -> Tensor:
    """groups: 0 = no norm, Cout = instance norm (gamma/beta None), 8 = GroupNorm(8) with affine."""
    return _ConvBlock.apply(x, weight, bias, gamma, beta, groups)


def conv_block_batchnorm(x, weight, bias, bn: "torch.nn.modules.batchnorm._BatchNorm", training: bool) -> Tensor:
    """Conv3d(k3,p1,bias) -> BatchNorm3d -> ReLU with torch's BatchNorm semantics (keymorph/layers.py:166-187 with
    norm_type='batch'): batch statistics + running-average update in training mode (momentum, unbiased running
    variance, num_batches_tracked), the running statistics as a fixed affine map in evaluation mode."""
    assert abs(bn.eps - EPS) < 1e-12, "the norm kernels use eps = 1e-5 (what every norm layer of the reference uses)"
    use_batch = training or not bn.track_running_stats or bn.running_mean is None
    if not use_batch:
        return _ConvBlock.apply(x, weight, bias, bn.weight, bn.bias, 0, 2, bn.running_mean, bn.running_var, bn.eps)
    y, stats = _ConvBlock.apply(x, weight, bias, bn.weight, bn.bias, 0, 1, None, None, bn.eps)
    if training and bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            n = float(x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3])
            mean = stats[0, :, 0] / n
            var = (stats[0, :, 1] / n - mean * mean).clamp_min(0.0)
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1 - mom).add_((var * (n / max(n - 1.0, 1.0))).to(bn.running_var.dtype), alpha=mom)
    return y


# ---------------------------------------------------------------------------------------------------------------------
# ConvNet with InstanceNorm kept lazy (keymorph/net.py:7-36, keymorph/layers.py:137-187, norm_type "instance").
# A block of the reference is Conv3d -> InstanceNorm3d -> ReLU [-> MaxPool3d(2)].  Here the chain is cut at the RAW
# convolution outputs z_b instead: unit b = InstanceNorm(z_{b-1}) -> ReLU -> [pool] -> Conv3d_b, and the normalisation, the
# ReLU and -- because InstanceNorm without affine is increasing, so pooling commutes with it -- the pooling of the raw
# tensor are applied by the convolution's own loader (scale / shift / relu_in of kmh_conv3d_fwd_bf and _wgrad_bf).  The
# statistics of z_{b-1} come with the epilogue of the convolution that produced it.  No normalised tensor, no ReLU output
# and no full-resolution pooling gradient is ever stored; the backward of the cut chain is three launches
# (kmh_in_bwd_stats at the pooled resolution, kmh_gn_bwd_coeffs, kmh_in_bwd_apply[_pool]) around the two convolutions.
# bench.py's ConvNet leg: 299 ms -> see DESIGN.md section 6.
# ---------------------------------------------------------------------------------------------------------------------
LAZY_IN_STATS = {"units": 0}


def convnet_lazy_ok(x: Tensor, norm_type: str) -> bool:
    """(N, D, H, W, C) NDHWC input of the ConvNet: four poolings need D, H, W % 16 == 0 for the even-size kernels; the
    split-operand convolutions emit the statistics; KEYMORPH_NO_LAZY_IN=1 takes the block-by-block route (A/B, tests).
    The fused units never form the gradient w.r.t. the image (their first unit returns None for it), so an input that
    requires one -- saliency maps, adversarial inputs, augmentation differentiated through the image -- takes the
    block-by-block route, whose first block does compute it."""
    return (norm_type == "instance" and conv_emits_stats() and not os.environ.get("KEYMORPH_NO_LAZY_IN")
            and not (x.requires_grad and torch.is_grad_enabled())
            and all(int(d) % 16 == 0 for d in x.shape[1:4]))


@_binds_amp
class _ConvINUnit(torch.autograd.Function):
    """z = Conv3d_b(u) + bias with u = [MaxPool3d(2)](ReLU(InstanceNorm(zprev))) applied on the fly (zprev: the previous
    block's raw convolution output with its epilogue statistics `st` (N, Cin, 2)); first unit: u = the image, no norm.
    Returns (z, statistics of z)."""

    @staticmethod
    def forward(ctx, zprev, st, weight, bias, pool, first):
        lib = _lib.load()
        zprev, weight, bias = _prep(zprev), _prep(weight), _prep(bias)
        N, Dp, Hp, Wp, Cin = zprev.shape
        Cout = weight.shape[0]
        ctx.first, ctx.pool = bool(first), bool(pool)
        scale = shift = mr = ascale = arg = None
        u = zprev
        if not first:
            scale, shift, mr, ascale = norm_coeffs(st, None, None, N, Cin, Cin, Dp * Hp * Wp, want_ascale=True)
            if pool:
                u = _f32((N, Dp // 2, Hp // 2, Wp // 2, Cin), zprev.device)
                arg = torch.empty(u.shape, dtype=torch.uint8, device=zprev.device)
                check(lib.kmh_maxpool3d_fwd(_p(zprev), _p(u), _p(arg), N, Dp, Hp, Wp, Cin, _stream()), "kmh_maxpool3d_fwd")
        D, H, W = u.shape[1:4]
        pk = pack_weight(weight, False)
        ctx.wscale = getattr(pk, "_kmh_wscale", None)
        zst = torch.empty((N, Cout, 2), dtype=torch.float64, device=zprev.device)
        z = conv3_raw(u, scale, shift, pk, bias, N, D, H, W, Cin, Cout, not first, False, ascale=ascale, stats_out=zst)
        ctx.ascale = ascale
        ctx.dims = (N, Dp, Hp, Wp, D, H, W, Cin, Cout)
        saved = [zprev, weight] + ([] if first else [scale, shift, mr]) + ([u, arg] if (pool and not first) else [])
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(zst)
        LAZY_IN_STATS["units"] += 1
        return z, zst

    @staticmethod
    def backward(ctx, dz, _dst):
        lib = _lib.load()
        N, Dp, Hp, Wp, D, H, W, Cin, Cout = ctx.dims
        saved = ctx.saved_tensors
        zprev, weight = saved[0], saved[1]
        dscale = _peek_grad_scale(dz)
        dz = _prep(dz)
        if _needs_range_scales() and dscale is None:
            dscale = absmax_scale(dz)
        # InstanceNorm follows every convolution of this network: a bias shifts the mean and nothing else
        db = torch.zeros(Cout, dtype=torch.float32, device=dz.device)
        if ctx.first:
            xs = absmax_scale(zprev) if _needs_range_scales() else None
            dw = conv3_wgrad(zprev, None, None, dz, N, D, H, W, Cin, Cout, False, xscale=xs, dscale=dscale)
            return None, None, dw, db, None, None
        scale, shift, mr = saved[2:5]
        u, arg = (saved[5], saved[6]) if ctx.pool else (zprev, None)
        dw = conv3_wgrad(u, scale, shift, dz, N, D, H, W, Cin, Cout, True, xscale=ctx.ascale, dscale=dscale)
        du = conv3_raw(dz, None, None, pack_weight(weight, True, wscale=ctx.wscale), None, N, D, H, W, Cout, Cin, False, False,
                       ascale=dscale)
        # g = scatter(du) [zhat > 0] lives at u's resolution: both sums of InstanceNorm's backward are taken there
        ab = torch.empty((N, Cin, 2), dtype=torch.float64, device=dz.device)
        ws = workspace(int(lib.kmh_channel_stats_ws_bytes(N, Cin)), dz.device, "stats")
        check(lib.kmh_in_bwd_stats(_p(du), _p(u), _p(scale), _p(shift), N, D * H * W, Cin, _p(ab), _p(ws), _stream()),
              "kmh_in_bwd_stats")
        c123 = _f32((N, Cin, 3), dz.device)
        check(lib.kmh_gn_bwd_coeffs(_p(ab), None, _p(mr), N, Cin, Cin, float(Dp * Hp * Wp), _p(c123), None, None, None,
                                    _stream()), "kmh_gn_bwd_coeffs")
        sc2 = torch.zeros(2, dtype=torch.float32, device=dz.device) if _needs_range_scales() else None
        if ctx.pool:
            dzp = _f32((N, Dp, Hp, Wp, Cin), dz.device)
            check(lib.kmh_in_bwd_apply_pool(_p(arg), _p(du), _p(zprev), _p(scale), _p(shift), _p(c123), N, Dp, Hp, Wp, Cin,
                                            _p(dzp), _p(sc2), _stream()), "kmh_in_bwd_apply_pool")
        else:
            dzp = du
            check(lib.kmh_in_bwd_apply(_p(du), _p(zprev), _p(scale), _p(shift), _p(c123), N, Dp * Hp * Wp, Cin, _p(dzp),
                                       _p(sc2), _stream()), "kmh_in_bwd_apply")
        _tag_grad_scale(dzp, sc2)
        return dzp, None, dw, db, None, None


class _INReluOut(torch.autograd.Function):
    """y = ReLU(InstanceNorm(z)) materialised (the network's output in front of the center of mass: 16^3 voxels)."""

    @staticmethod
    def forward(ctx, z, st):
        lib = _lib.load()
        z = _prep(z)
        N, D, H, W, C = z.shape
        V = D * H * W
        scale, shift, mr = norm_coeffs(st, None, None, N, C, C, V)
        y = torch.empty_like(z)
        check(lib.kmh_norm_apply(_p(z), _p(scale), _p(shift), N, V, C, 1, _p(y), _stream()), "kmh_norm_apply")
        ctx.save_for_backward(z, scale, shift, mr)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        z, scale, shift, mr = ctx.saved_tensors
        N, D, H, W, C = z.shape
        V = D * H * W
        dy = _prep(dy)
        ab = torch.empty((N, C, 2), dtype=torch.float64, device=z.device)
        ws = workspace(int(lib.kmh_channel_stats_ws_bytes(N, C)), z.device, "stats")
        check(lib.kmh_in_bwd_stats(_p(dy), _p(z), _p(scale), _p(shift), N, V, C, _p(ab), _p(ws), _stream()), "kmh_in_bwd_stats")
        c123 = _f32((N, C, 3), z.device)
        check(lib.kmh_gn_bwd_coeffs(_p(ab), None, _p(mr), N, C, C, float(V), _p(c123), None, None, None, _stream()),
              "kmh_gn_bwd_coeffs")
        dz = torch.empty_like(z)
        sc2 = torch.zeros(2, dtype=torch.float32, device=z.device) if _needs_range_scales() else None
        check(lib.kmh_in_bwd_apply(_p(dy), _p(z), _p(scale), _p(shift), _p(c123), N, V, C, _p(dz), _p(sc2), _stream()),
              "kmh_in_bwd_apply")
        _tag_grad_scale(dz, sc2)
        return dz, None


def convnet_instance_lazy(x: Tensor, blocks) -> Tensor:
    """The whole ConvNet (instance norm) on an NDHWC image: blocks = [(weight, bias, down_sample), ...] in order.
    Returns ReLU(IN(conv_9(...))) [pooled if the last block pools], NDHWC."""
    z = st = None
    pool_prev = False
    for b, (w, bias, down) in enumerate(blocks):
        if b == 0:
            z, st = _ConvINUnit.apply(x, None, w, bias, False, True)
        else:
            z, st = _ConvINUnit.apply(z, st, w, bias, pool_prev, False)
        pool_prev = bool(down)
    y = _INReluOut.apply(z, st)
    return maxpool2(y) if pool_prev else y


# The forward of the fused head stores [h > 0] (1 bit per voxel and keypoint channel: 0.5 GB for 4 x 128^3 x 512) when a
# backward can follow, and the backward then skips recomputing the logits.  KEYMORPH_HEAD_MASK=0: always recompute.
HEAD_MASK = os.environ.get("KEYMORPH_HEAD_MASK", "1") != "0"
HEAD_STATS = {"mask": 0, "recompute": 0}


@_binds_amp
class _HeadCoM(torch.autograd.Function):
    """pts = CenterOfMass3d('ij')(conv1x1(feat) + b) without the heat-map (csrc/headcom.hip); the second output is
    power = sum relu(h) per channel (keymorph/model.py:96-109), differentiable through the same backward pass."""

    @staticmethod
    def forward(ctx, feat, w, b, feat_from_relu=False):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        ctx.mask_dfeat = int(bool(feat_from_relu))
        feat, w = _prep(feat), _prep(w)
        b = None if b is None else _prep(b)
        N, D, H, W, Cin = feat.shape
        Cout = w.shape[0]
        pts = _f32((N, Cout, 3), feat.device)
        sums = _f32((N, Cout, 4), feat.device)
        if CONV_MODE != "f32" and Cin % 4 == 0:
            terms = _HEAD_TERMS[CONV_MODE]
            ws = workspace(int(lib.kmh_headcom_fwd_bf_ws_bytes(N, D * H * W, Cout, terms)), feat.device, "head")
            hsc = _f32((4,), feat.device) if terms == 2 else None    # feat / filter range scales, re-used by the backward
            # [h > 0] bits for the backward (it then skips recomputing the logits): only when a backward can follow
            nmask = int(lib.kmh_headcom_mask_words(N, D, H, W, Cout)) if (HEAD_MASK and any(ctx.needs_input_grad[:3])) else 0
            hmask = torch.empty(nmask, dtype=torch.int32, device=feat.device) if nmask else None
            check(lib.kmh_headcom_fwd_bf(_p(feat), _p(w), _p(b), _p(pts), _p(sums), None, _p(hsc), N, D, H, W, Cin, Cout,
                                         _t(terms), _p(hmask), _p(ws), _stream()), "kmh_headcom_fwd_bf")
            ctx.hsc = hsc
            ctx.hmask = hmask
            HEAD_STATS["mask" if nmask else "recompute"] += 1
        else:
            ws = workspace(int(lib.kmh_headcom_fwd_ws_bytes(N, D * H * W, Cout)), feat.device, "head")
            check(lib.kmh_headcom_fwd(_p(feat), _p(w), _p(b), _p(pts), _p(sums), None, N, D, H, W, Cin, Cout, _p(ws),
                                      _stream()), "kmh_headcom_fwd")
        ctx.save_for_backward(feat, w, sums) if b is None else ctx.save_for_backward(feat, w, sums, b)
        return pts, sums[:, :, 0].contiguous()

    @staticmethod
    def backward(ctx, dpts, dpower):
        lib = _lib.load()
        saved = ctx.saved_tensors
        feat, w, sums = saved[:3]
        b = saved[3] if len(saved) > 3 else None
        N, D, H, W, Cin = feat.shape
        Cout = w.shape[0]
        dpts = torch.zeros((N, Cout, 3), dtype=torch.float32, device=feat.device) if dpts is None else _prep(dpts)
        dpower = None if dpower is None else _prep(dpower)
        dfeat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or (b is not None and ctx.needs_input_grad[2])
        dw = torch.empty_like(w) if need_w else None
        db = _f32((Cout,), feat.device) if (need_w and b is not None) else None
        if CONV_MODE != "f32" and Cin % 4 == 0:
            terms = _HEAD_TERMS[CONV_MODE]
            ws = workspace(int(lib.kmh_headcom_bwd_bf_ws_bytes(N, D * H * W, Cin, Cout, terms)), feat.device, "head")
            dsc = (torch.zeros(2, dtype=torch.float32, device=feat.device)
                   if (terms == 2 and dfeat is not None) else None)
            check(lib.kmh_headcom_bwd_bf(_p(dpts), _p(dpower), _p(feat), _p(w), _p(b), _p(sums), _p(dfeat), _p(dw), _p(db), N, D, H,
                                         W, Cin, Cout, _t(terms), ctx.mask_dfeat, _p(getattr(ctx, "hsc", None)), _p(dsc),
                                         _p(getattr(ctx, "hmask", None)), _p(ws), _stream()), "kmh_headcom_bwd_bf")
            _tag_grad_scale(dfeat, dsc)
        else:
            ws = workspace(int(lib.kmh_headcom_bwd_ws_bytes(N, D * H * W, Cin, Cout)), feat.device, "head")
            check(lib.kmh_headcom_bwd(_p(dpts), _p(dpower), _p(feat), _p(w), _p(b), _p(sums), _p(dfeat), _p(dw), _p(db), N, D, H,
                                      W, Cin, Cout, ctx.mask_dfeat, _p(ws), _stream()), "kmh_headcom_bwd")
        return dfeat, dw, db, None


HEAD_FUSED_MAX_CIN = 64


def head_moments(feat: Tensor, w: Tensor, b: Optional[Tensor]):
    """Inference-only companion of head_com for keypoint weighting (keymorph/model.py:75-109): keypoints plus the
    per-channel moments of relu(heat-map), still without materialising it.
    -> pts (N,K,3), power (N,K) = sum relu(h), sq (N,K) = sum relu(h)^2.  No autograd graph is recorded."""
    lib = _lib.load()
    with torch.no_grad():
        feat, w = _prep(feat), _prep(w)
        b = None if b is None else _prep(b)
        N, D, H, W, Cin = feat.shape
        Cout = w.shape[0]
        pts, sums, sq = _f32((N, Cout, 3), feat.device), _f32((N, Cout, 4), feat.device), _f32((N, Cout), feat.device)
        if CONV_MODE != "f32" and Cin % 4 == 0:
            terms = _HEAD_TERMS[CONV_MODE]
            ws = workspace(int(lib.kmh_headcom_fwd_bf_ws_bytes(N, D * H * W, Cout, terms)), feat.device, "head")
            check(lib.kmh_headcom_fwd_bf(_p(feat), _p(w), _p(b), _p(pts), _p(sums), _p(sq), None, N, D, H, W, Cin, Cout,
                                         _t(terms), None, _p(ws), _stream()), "kmh_headcom_fwd_bf")
        else:
            ws = workspace(int(lib.kmh_headcom_fwd_ws_bytes(N, D * H * W, Cout)), feat.device, "head")
            check(lib.kmh_headcom_fwd(_p(feat), _p(w), _p(b), _p(pts), _p(sums), _p(sq), N, D, H, W, Cin, Cout, _p(ws),
                                      _stream()), "kmh_headcom_fwd")
        return pts, sums[:, :, 0].contiguous(), sq


def head_com(feat: Tensor, w: Tensor, b: Optional[Tensor], feat_from_relu: bool = False) -> Tensor:
    """(N,D,H,W,Cin) features + final_conv parameters -> (N,K,3) keypoints in ij order.
    feat_from_relu: feat is a ReLU output whose producer was told `dy_premasked`: the feature gradient is returned
    already multiplied by (feat > 0), for free (the backward re-reads its own operand)."""
    return _HeadCoM.apply(feat, w, b, feat_from_relu)[0]


def head_com_power(feat: Tensor, w: Tensor, b: Optional[Tensor], feat_from_relu: bool = False):
    """head_com plus power (N,K) = sum relu(h), both with gradients (training with weight_keypoints='power')."""
    return _HeadCoM.apply(feat, w, b, feat_from_relu)
