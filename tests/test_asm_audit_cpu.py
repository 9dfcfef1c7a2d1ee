"""ISA audit of the hand-counted inline-asm loads of csrc/conv_bf.hip (no GPU needed: hipcc cross-compiles gfx950).

An `asm volatile("global_load_dwordx4 %0, ...")` destination is defined, for the compiler, when the statement ends -- long before
the data lands.  Under register pressure the compiler has been seen to copy such a register (v_accvgpr_write) while the load was
still in flight, which crashed a kernel variant on the GPU.  This test fails the build of any conv3_fwd_[sg]_kernel instance whose
assembly touches a destination between its load and the (counted) wait that covers it (tools/scan_asm_inflight.py)."""
import pytest

from keymorph_amd import build, isa_audit


def _have_hipcc():
    try:
        build._hipcc()                                   # the compiler the library itself is built with
        return True
    except RuntimeError:
        return False


needs_hipcc = pytest.mark.skipif(not _have_hipcc(), reason="needs hipcc")


def _audit(*flags):
    return isa_audit.audit(flags)


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


def test_scan_follows_loop_back_edges():
    """The walk is over the control-flow graph: a destination touched only on the path through a loop's back edge (invisible to a
    scan in text order) is found; the same loop with the covering wait in front of the touch is clean."""
    def kernel(first_in_loop):
        return "\n".join([
            "_ZN12_GLOBAL__N_118conv3_fwd_s_kernelILi1EEEvv: ; @x",
            "s_mov_b32 s0, 4",
            ".LBB0_1:",
            first_in_loop,                                  # runs again after the back edge, with the load below in flight
            "v_add_f32 v9, v9, v9",
            "s_waitcnt vmcnt(0)",
            "v_mul_f32 v8, v1, v1",
            "s_add_i32 s0, s0, -1",
            "s_cmp_lg_u32 s0, 0",
            ";;#ASMSTART",
            "global_load_dwordx4 v[0:3], v[4:5], off",
            ";;#ASMEND",
            "s_cbranch_scc1 .LBB0_1",
            "s_waitcnt vmcnt(0)",
            "s_endpgm",
            ".Lfunc_end0:", ""])
    assert isa_audit.scan(kernel("v_mov_b32 v7, v2")) == {"conv3_fwd_s_kernelILi1EEEvv": (1, 1)}
    assert isa_audit.scan(kernel("v_mov_b32 v7, v6")) == {"conv3_fwd_s_kernelILi1EEEvv": (1, 0)}
