"""ctypes binding of libkeymorph_hip.so (the C ABI declared in include/keymorph_hip.h).

There is NO fallback: if the library is missing the product path raises.  Build it with
``python -m keymorph_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

from .build import LIBPATH

_f = C.c_void_p  # device pointer (float*/double*/void*)
_i = C.c_int
_ll = C.c_longlong
_sz = C.c_size_t

# name -> (restype, argtypes).  Mirrors include/keymorph_hip.h one-to-one
# (tests/test_abi.py parses the header and checks both directions).
PROTOS = {
    "kmh_abi_version": (_i, []),
    "kmh_grid_sample3d_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f]),
    "kmh_grid_sample3d_bwd_grid": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f]),
    "kmh_grid_sample3d_bwd_input": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f]),
    "kmh_reduce_ws_bytes": (_sz, []),
    "kmh_mse_fwd": (_i, [_f, _f, _ll, _f, _f, _f]),
    "kmh_mse_bwd": (_i, [_f, _f, _f, _ll, _f, _f]),
    "kmh_warp_mse_fwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_warp_mse_fwd_grad": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_scale_unless_one": (_i, [_f, _ll, _f, _f]),
    "kmh_dice_sums": (_i, [_f, _f, _i, _ll, _f, _f, _f]),
    "kmh_rows_axpby": (_i, [_f, _f, _f, _f, _i, _ll, _f, _f]),
    "kmh_warp_dice_ok": (_i, [_i, _i, _i, _i, _i]),
    "kmh_warp_dice_sums": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f]),
    "kmh_warp_dice_bwd_grid": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "kmh_onehot_to_labels": (_i, [_f, _i, _i, _ll, _f, _f, _f]),
    "kmh_argmax_onehot": (_i, [_f, _i, _i, _ll, _f, _f]),
    "kmh_affine_grid_fwd": (_i, [_f, _f, _i, _i, _i, _i, _f]),
    "kmh_affine_grid_bwd": (_i, [_f, _f, _i, _i, _i, _i, _f, _f]),
    "kmh_tps_grid_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "kmh_tps_grid_bwd_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "kmh_tps_grid_bwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_tps_points_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _f]),
    "kmh_tps_points_bwd_ws_bytes": (_sz, [_i, _i, _i]),
    "kmh_tps_points_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f]),
    "kmh_tps_fit_ws_bytes": (_sz, [_i, _i]),
    "kmh_tps_fit_fwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _f, _f]),
    "kmh_tps_fit_force_retry": (_i, [_i]),
    "kmh_tps_fit_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _f, _f]),
    "kmh_affine_fit_fwd": (_i, [_f, _f, _f, _f, _i, _i, _f]),
    "kmh_affine_fit_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _f]),
    "kmh_rigid_fit_fwd": (_i, [_f, _f, _f, _f, _i, _i, _f]),
    "kmh_rigid_fit_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _i, _f]),
    "kmh_affine_inverse_fwd": (_i, [_f, _f, _i, _f]),
    "kmh_affine_inverse_bwd": (_i, [_f, _f, _f, _i, _f]),
    "kmh_affine_points_fwd": (_i, [_f, _f, _f, _i, _i, _f]),
    "kmh_jacobian_det": (_i, [_f, _ll, _ll, _i, _i, _i, _f, _f, _f, _f]),
    "kmh_label_presence": (_i, [_f, _ll, _i, _f, _f]),
    "kmh_one_hot_select": (_i, [_f, _i, _ll, _f, _i, _f, _i, _f]),
    "kmh_affine_build_matrix": (_i, [_f, _f, _f, _f, _f, _i, _f]),
    "kmh_affine_points_bwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _f]),
    "kmh_com3d_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_com3d_bwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "kmh_conv3d_pack_weight": (_i, [_f, _f, _i, _i, _i, _f]),
    "kmh_conv3d_fwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f]),
    "kmh_conv3d_pack_bf_bytes": (_sz, [_i, _i, _i, _i]),
    "kmh_conv3d_pack_weight_bf": (_i, [_f, _f, _i, _i, _i, _i, _f, _f]),
    "kmh_conv3d_fwd_bf": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _f, _f]),
    "kmh_conv3d_fwd_bf_set_dispatch": (_i, [_i]),
    "kmh_conv3d_fwd_bf_variant": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "kmh_conv3d_fwd_bf_pool_ok": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "kmh_conv3d_fwd_bf_split_ok": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "kmh_conv3d_fwd_bf_pool": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _f]),
    "kmh_conv3d_up2_dgrad_pack_bytes": (_sz, [_i, _i, _i]),
    "kmh_conv3d_up2_dgrad_pack_weight": (_i, [_f, _f, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_conv3d_up2_dgrad_stats_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "kmh_conv3d_up2_dgrad": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _f]),
    "kmh_up2_boxsum": (_i, [_f, _f, _i, _i, _i, _i, _i, _i, _f]),
    "kmh_up2_wgrad_gemm_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "kmh_up2_wgrad_gemm": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f]),
    "kmh_up2_wgrad_fold_ok": (_i, [_i, _i, _i]),
    "kmh_up2_wgrad_fold_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "kmh_up2_wgrad_fold": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _i, _f, _f]),
    "kmh_conv3d_up2_pack_bytes": (_sz, [_i, _i, _i]),
    "kmh_conv3d_up2_pack_weight": (_i, [_f, _f, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_conv3d_up2_fwd": (_i, [_f, _f, _f, _i, _i, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f]),
    "kmh_conv3d_fwd_bf_stats_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "kmh_conv3d_first_layer_fwd_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "kmh_conv3d_first_layer_fwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f]),
    "kmh_conv3d_wgrad_bf_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "kmh_conv3d_wgrad_bf": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _f, _f, _f, _f]),
    "kmh_conv3d_wgrad_bf_blocked_ok": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "kmh_conv3d_first_layer_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "kmh_conv3d_first_layer_wgrad": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_conv3d_first_layer_fold": (_i, [_f, _i, _f, _f, _f, _i, _f, _f, _i, _f]),
    "kmh_conv3d_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "kmh_conv3d_wgrad": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_channel_stats_ws_bytes": (_sz, [_i, _i]),
    "kmh_channel_stats": (_i, [_f, _f, _i, _i, _ll, _i, _f, _f, _f, _f]),
    "kmh_gn_fwd_coeffs": (_i, [_f, _f, _f, _i, _i, _i, C.c_double, C.c_float, _f, _f, _f, _f, _f]),
    "kmh_absmax_scale": (_i, [_f, _ll, C.c_float, _f, _f]),
    "kmh_gn_bwd_coeffs": (_i, [_f, _f, _f, _i, _i, _i, C.c_double, _f, _f, _f, _f, _f]),
    "kmh_gn_bwd_coeffs_fold": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, C.c_double, _f, _f, _f, _f, _f]),
    "kmh_gn_bwd_apply": (_i, [_f, _f, _f, _i, _ll, _i, _i, _i, _f, _f, _i, _f]),
    "kmh_in_bwd_stats": (_i, [_f, _f, _f, _f, _i, _ll, _i, _f, _f, _f]),
    "kmh_in_bwd_apply": (_i, [_f, _f, _f, _f, _f, _i, _ll, _i, _f, _f, _f]),
    "kmh_in_bwd_apply_pool": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f]),
    "kmh_relu_mask": (_i, [_f, _f, _ll, _f, _f]),
    "kmh_norm_apply": (_i, [_f, _f, _f, _i, _ll, _i, _i, _f, _f]),
    "kmh_maxpool3d_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "kmh_maxpool3d_bwd": (_i, [_f, _f, _f, _f, _i, _f, _i, _i, _i, _i, _i, _i, _f]),
    "kmh_maxpool3d_bwd_split_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "kmh_maxpool3d_bwd_split": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "kmh_maxpool3d_bwd_lazy": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_upcat_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f]),
    "kmh_upcat_bwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f]),
    "kmh_layout_convert": (_i, [_f, _f, _i, _ll, _i, _i, _f]),
    "kmh_pointwise_pack": (_i, [_f, _f, _i, _i, _f]),
    "kmh_pointwise_fwd": (_i, [_f, _f, _f, _f, _i, _ll, _i, _i, _f]),
    "kmh_pointwise_dgrad": (_i, [_f, _f, _f, _i, _ll, _i, _i, _f]),
    "kmh_pointwise_wgrad_ws_bytes": (_sz, [_i, _ll, _i, _i]),
    "kmh_pointwise_wgrad": (_i, [_f, _f, _f, _f, _i, _ll, _i, _i, _i, _f, _f]),
    "kmh_headcom_fwd_ws_bytes": (_sz, [_i, _ll, _i]),
    "kmh_headcom_bwd_ws_bytes": (_sz, [_i, _ll, _i, _i]),
    "kmh_headcom_fwd_bf_ws_bytes": (_sz, [_i, _ll, _i, _i]),
    "kmh_headcom_bwd_bf_ws_bytes": (_sz, [_i, _ll, _i, _i, _i]),
    "kmh_headcom_mask_words": (_sz, [_i, _i, _i, _i, _i]),
    "kmh_headcom_fwd_bf": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f]),
    "kmh_headcom_bwd_bf": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f]),
    "kmh_headcom_fwd": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_headcom_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f]),
    "kmh_adam_step": (_i, [_f, _f, _f, _f, _ll, C.c_float, C.c_float, C.c_float, C.c_float, _i, C.c_float, _f]),
}

_lib = None


class KeymorphHipError(RuntimeError):
    pass


class _Profiler:
    """Optional per-entry-point timing with HIP events on torch's current stream (the stream every
    kernel of this library is launched on).  Off by default; bench.py turns it on for one step."""

    def __init__(self):
        self.enabled = False
        self.records = []   # (name, start_event, end_event, meta)
        self.meta = None

    def reset(self):
        self.records = []

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, a, b, meta in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["calls"] += 1
            d["ms"] += a.elapsed_time(b)
            if meta:
                d["flops"] += meta.get("flops", 0.0)
                d["bytes"] += meta.get("bytes", 0.0)
        return out


profiler = _Profiler()


class _Proxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("kmh_") or name.endswith("_bytes") or name == "kmh_abi_version":
            return fn

        def call(*args):
            if not profiler.enabled:
                return fn(*args)
            import torch
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(*args)
            b.record()
            profiler.records.append((name, a, b, profiler.meta))
            profiler.meta = None
            return rc

        object.__setattr__(self, name, call)
        return call


_DEFAULT_LIBPATH = LIBPATH


def load():
    """dlopen the HIP library (idempotent).  Raises if it has not been built."""
    global _lib, LIBPATH
    if _lib is not None:
        return _lib
    if os.environ.get("KEYMORPH_HIP_LIB"):      # another build of the library (A/B runs: tools/build_exp_lib.sh)
        LIBPATH = os.environ["KEYMORPH_HIP_LIB"]
    if not os.path.exists(LIBPATH):
        raise KeymorphHipError(
            f"{LIBPATH} is missing: the HIP extension has not been built "
            "(run `python -m keymorph_amd.build`); there is no CPU/PyTorch fallback.")
    # One HIP runtime per process: bind to the runtime PyTorch-ROCm already loaded (the library is
    # linked without its own libamdhip64 dependency, see build.py), so streams / events / memory of
    # torch and of this library belong to the same runtime and launches on torch's current stream
    # are ordered with torch's own work.
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
    rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)
    else:  # system ROCm PyTorch build
        C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIBPATH)
    ab_build = os.path.abspath(LIBPATH) != os.path.abspath(_DEFAULT_LIBPATH)      # another build, named explicitly (A/B timing)
    for name, (res, args) in PROTOS.items():
        try:
            fn = getattr(lib, name)  # AttributeError => ABI mismatch, fail loudly ...
        except AttributeError:
            if not ab_build:
                raise
            continue                 # ... except for an explicitly named A/B build of an older commit: calling the symbol raises
        fn.restype = res
        fn.argtypes = args
    if lib.kmh_abi_version() != 1:
        raise KeymorphHipError("libkeymorph_hip.so ABI version mismatch")
    _lib = _Proxy(lib)
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise KeymorphHipError(f"{what} failed with status {rc}")
