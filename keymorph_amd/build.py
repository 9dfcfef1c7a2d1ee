"""Ahead-of-time build of libkeymorph_hip.so (hipcc, gfx950 only, in-tree).

The shared object lands in keymorph_amd/lib/ so it travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).  No JIT at import time.
"""
from __future__ import annotations

import functools
import hashlib
import re
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libkeymorph_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden",
         "-Wno-unused-result", "-ffp-contract=fast"]


# Per-file extra flags.  conv_wgrad.hip (= the weight-gradient section of conv_bf.hip as its own translation unit): LLVM's
# "max-ILP" scheduling strategy -- the wave-specialised weight-gradient kernels (several waves per SIMD, no explicit scheduling
# groups) run 2-4 % faster with it (profiles/r5z_llvm_sched_strategy_max_ilp.txt).  NOT for conv_bf.hip itself: under max-ILP
# hipcc spills an in-flight destination of conv3_fwd_g_kernel<2>'s inline-asm loads (tools/scan_asm_inflight.py finds it).
# norm.hip: its streaming kernels keep more loads in flight under max-ILP (kmh_maxpool3d_bwd_lazy -25 %, kmh_maxpool3d_bwd_split
# -7 %, GroupNorm kernels unchanged); headcom.hip, grids.hip and the fused decoder kernels get SLOWER with it (+18 % / +4 % / +5 %)
# and keep the default (profiles/r5z_llvm_sched_strategy_max_ilp.txt).
_MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
FILE_FLAGS = {"conv_wgrad.hip": _MAX_ILP, "norm.hip": _MAX_ILP}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


# Sources whose hand-counted inline-asm waits are only correct if the compiler never copies or spills a register a load is still
# in flight to (keymorph_amd/isa_audit.py): their assembly -- of THIS compilation, kept with -save-temps -- is scanned before the
# object is accepted.  A failing scan is retried with the full-drain arms of the kernels (no counted waits, no deep ring:
# slower, always correct); if that fails too the build fails.  KEYMORPH_SKIP_ISA_AUDIT=1 skips the scan (A/B experiment builds).
AUDITED = {"conv_bf.hip": ["-DKMH_S_CW=0", "-DKMH_S_DEEP=0"]}


@functools.lru_cache(maxsize=None)
def _hipcc_version(hipcc: str) -> str:
    try:
        return subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout.strip()
    except OSError:
        return "unknown"


def _digest(path: str) -> str:
    # flags + the compiler's own version string: another ROCm / hipcc rebuilds (and re-audits) everything
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items())) + repr(sorted(AUDITED.items()))
                        + _hipcc_version(_hipcc())).encode())
    with open(path, "rb") as f:
        text = f.read()
    h.update(text)
    for inc in re.findall(rb'^#include "([^"]+)"', text, re.M):      # local includes (common.h; conv_wgrad.hip includes conv_bf.hip)
        q = os.path.join(CSRC, inc.decode())
        if os.path.exists(q):
            with open(q, "rb") as f:
                h.update(f.read())
    with open(os.path.join(CSRC, "common.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _compile_audited(cmd, obj: str, safe_flags, verbose: bool) -> None:
    """Compile with -save-temps into a scratch directory, scan the gfx950 assembly the object was assembled from, move the object
    into place only if no in-flight asm destination is touched; else once more with `safe_flags`."""
    from . import isa_audit
    base = os.path.splitext(os.path.basename(obj))[0]
    for extra in ([], list(safe_flags)):
        with tempfile.TemporaryDirectory(prefix="kmh_build_") as tmp:
            tobj = os.path.join(tmp, base + ".o")
            c = [*cmd[:-1], tobj, "-save-temps=obj", *extra]
            if verbose:
                print(" ".join(c), file=sys.stderr)
            subprocess.run(c, check=True)
            asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "amdgcn" in f]
            if len(asm) != 1:
                raise RuntimeError(f"ISA audit: expected one device assembly file from -save-temps, found {asm}")
            with open(os.path.join(tmp, asm[0])) as f:
                res = isa_audit.scan(f.read())
            bad = isa_audit.failures(res)
            if not res or not any(n for n, _ in res.values()):
                raise RuntimeError("ISA audit: no inline-asm loads recognised in %s -- the scan no longer matches the compiler's output"
                                   % asm[0])
            if not bad:
                if verbose:
                    print(f"ISA audit of {base}: {len(res)} kernels, {sum(n for n, _ in res.values())} asm loads, "
                          f"0 in-flight destinations touched{' (full-drain build)' if extra else ''}", file=sys.stderr)
                shutil.move(tobj, obj)
                return
            print(f"ISA audit of {base}{' ' + ' '.join(extra) if extra else ''}: asm-load destinations touched while in flight: "
                  f"{bad}", file=sys.stderr)
    raise RuntimeError(f"ISA audit: {base} fails even with {' '.join(safe_flags)}; refusing to build a library whose results "
                       "could differ from run to run")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 and link them into one shared library."""
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    hipcc = _hipcc()
    objs, relink = [], force or not os.path.exists(LIBPATH)
    for src in sources():
        base = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(LIBDIR, "obj", base + ".o")
        stamp = obj + ".sha"
        dig = _digest(src)
        fresh = os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig
        if force or not fresh:
            name = os.path.basename(src)
            cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(name, []), "-c", src, "-o", obj]
            if name in AUDITED and not os.environ.get("KEYMORPH_SKIP_ISA_AUDIT"):
                _compile_audited(cmd, obj, AUDITED[name], verbose)
            else:
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                subprocess.run(cmd, check=True)
            with open(stamp, "w") as f:
                f.write(dig)
            relink = True
        objs.append(obj)
    if relink:
        # Link with the plain host linker and NO libamdhip64 dependency: the hip* symbols are bound at
        # dlopen time to the ONE HIP runtime already in the process (PyTorch-ROCm's, see _lib.load()).
        # `hipcc -shared` would record the system libamdhip64.so.7 next to torch's bundled runtime and
        # put two HIP runtimes (two sets of streams/queues) into one process.
        cmd = [shutil.which("g++") or "g++", "-shared", "-fPIC", "-o", LIBPATH, *objs]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIBPATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
