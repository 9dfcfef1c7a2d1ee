"""ISA audit of the hand-counted inline-asm loads of csrc/conv_bf.hip (no GPU needed: hipcc cross-compiles gfx950).

An `asm volatile("global_load_dwordx4 %0, ...")` destination is defined, for the compiler, when the statement ends -- long before
the data lands.  Under register pressure the compiler has been seen to copy such a register (v_accvgpr_write) while the load was
still in flight, which crashed a kernel variant on the GPU.  This test fails the build of any conv3_fwd_[sg]_kernel instance whose
assembly touches a destination between its load and the (counted) wait that covers it (tools/scan_asm_inflight.py)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


needs_hipcc = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


def _audit(*flags):
    spec = importlib.util.spec_from_file_location("scan_asm_inflight", os.path.join(ROOT, "tools", "scan_asm_inflight.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.audit(flags)


@needs_hipcc
def test_no_inflight_asm_destination_is_copied():
    res = _audit()
    one_wave = {k: v for k, v in res.items() if "conv3_fwd_s_kernel" in k}
    assert len(one_wave) >= 8, sorted(res)                 # every instance of the one-wave-per-SIMD kernel was found ...
    assert all(loads > 0 for loads, _ in one_wave.values()), one_wave   # ... and its asm loads were recognised
    bad = {k: v for k, v in res.items() if v[1]}
    assert not bad, f"asm-load destinations touched while in flight: {bad}"


@needs_hipcc
def test_audit_flags_the_variant_that_crashed():
    """Positive control: the stage-deep fragment ring in "=v" registers (KMH_S_DEEP_RING_V=1, never built into the library) is the
    variant whose in-flight destinations the compiler moved into AGPRs and which faulted on the GPU -- the audit must see that."""
    res = _audit("-DKMH_S_DEEP_RING_V=1", "-DKMH_S_CW=0", "-DKMH_S_IL=0", "-DKMH_S_UNCOND=0")      # (the configuration it crashed in)
    assert any(bad for _, bad in res.values()), res
