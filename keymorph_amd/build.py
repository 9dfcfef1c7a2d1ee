"""Ahead-of-time build of libkeymorph_hip.so (hipcc, gfx950 only, in-tree).

The shared object lands in keymorph_amd/lib/ so it travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).  No JIT at import time.
"""
from __future__ import annotations

import hashlib
import re
import os
import shutil
import subprocess
import sys

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


def _digest(path: str) -> str:
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
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
            cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
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
