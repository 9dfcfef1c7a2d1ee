"""Audit of the hand-waited inline-asm loads of csrc/conv_bf.hip (run by keymorph_amd.build on the assembly of the very
compilation that makes the library, by tools/scan_asm_inflight.py from the command line, and by tests/test_asm_audit_cpu.py).

An `asm volatile("global_load_dwordx4 %0, ...")` destination is "defined", for the compiler, the moment the statement ends -- long
before the data lands.  Under register pressure the compiler has been seen to COPY such a register (v_accvgpr_write / v_mov /
scratch store) while the load was in flight: the copy holds stale data and the load lands in a register that has meanwhile been
given to something else (a GPU fault in round 5; run-to-run different results in round 4).  `scan()` walks the gfx950 assembly of
every conv3_fwd_[sg]_kernel instance: between an inline-asm load and the wait that covers it NO instruction may mention its
destination registers.  A wait covers a load if it is `vmcnt(N)` with N <= the number of vector-memory instructions issued after
the load (loads return in issue order; stores in flight only make a counted wait stricter -- the kernels never wait with a count
for a load that is older than a store, see conv_bf.hip).

The walk follows the control-flow graph, not the text: at a conditional branch both the fall-through and the target are
explored, at `s_branch` only the target -- so a load issued near the end of a loop body is followed through the back edge into
the next iteration (round 5's scan was linear in text order and saw only the fall-through path).  A path ends at the covering
wait, at `s_endpgm`, or at an instruction already visited with no more younger loads than now (fewer younger loads = fewer waits
cover = the stricter state, so that one is the state kept).
"""
from __future__ import annotations

import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = re.compile(r'^(global_|buffer_|flat_|scratch_)(load|store|atomic)')
KERNEL = re.compile(r'^(_ZN12_GLOBAL__N_1\d+(conv3_fwd_[sg]_kernel\S*?)):\s', re.M)
BRANCH = re.compile(r'^s_(branch|cbranch_\w+)\s+(\S+)')
LABEL = re.compile(r'^(\.?[A-Za-z_][\w.$]*):')


def regs_of(line: str, kind: str):
    """register numbers of file `kind` ('v' / 'a') an instruction line mentions"""
    out = set()
    ops = line.split(None, 1)[1] if ' ' in line else ''
    for m in re.finditer(r'\b%s\[(\d+):(\d+)\]' % kind, ops):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\b%s(\d+)\b' % kind, ops):
        out.add(int(m.group(1)))
    return out


def _parse(raw):
    """instructions of one function (text, inside-inline-asm flag) and label -> index of the next instruction"""
    body, flag, labels, in_asm = [], [], {}, False
    for l in raw:
        t = l.strip()
        if '#ASMSTART' in t:
            in_asm = True
        elif '#ASMEND' in t:
            in_asm = False
        elif not t or t.startswith((';', '//')):
            continue
        else:
            m = LABEL.match(t)
            if m and not in_asm:
                labels[m.group(1)] = len(body)
            elif not t.startswith('.'):
                body.append(t.split(';')[0].strip())
                flag.append(in_asm)
    return body, flag, labels


def _touched(body, labels, start, kind, dst) -> bool:
    """does any path from instruction `start` mention a register of `dst` before a wait that covers the load?"""
    best = {}                                      # instruction index -> fewest younger loads it was reached with
    work = [(start, 0)]
    while work:
        i, younger = work.pop()
        while i < len(body):
            if best.get(i, 1 << 30) <= younger:
                break
            best[i] = younger
            t = body[i]
            if t.startswith('s_waitcnt'):
                w = re.search(r'vmcnt\((\d+)\)', t)
                if w and int(w.group(1)) <= younger:   # all but the `younger` youngest have landed: this one has
                    break
            if t.startswith('s_endpgm'):
                break
            if VMEM.match(t):
                younger += 1
            if regs_of(t, kind) & dst:
                return True
            b = BRANCH.match(t)
            if b:
                tgt = labels.get(b.group(2))
                if tgt is not None:
                    work.append((tgt, younger))
                if b.group(1) == 'branch':             # unconditional: no fall-through
                    break
            i += 1
    return False


def scan(txt: str):
    """{kernel name: (inline-asm loads scanned, loads whose destination is touched while in flight)} for every
    conv3_fwd_[sg]_kernel instance in the gfx950 assembly text `txt`."""
    res = {}
    for m in KERNEL.finditer(txt):
        end = txt.find('.Lfunc_end', m.end())
        body, flag, labels = _parse(txt[m.end():end if end >= 0 else len(txt)].split('\n'))
        loads = bad = 0
        for i, l in enumerate(body):
            mm = re.match(r'global_load_dwordx4 ([va])\[(\d+):(\d+)\]', l)
            if not mm or not flag[i]:               # only the inline-asm loads: the compiler waits for its own loads itself
                continue
            loads += 1
            dst = set(range(int(mm.group(2)), int(mm.group(3)) + 1))
            if _touched(body, labels, i + 1, mm.group(1), dst):
                bad += 1
        res[re.sub(r'^\d+', '', m.group(2))] = (loads, bad)
    return res


def failures(res) -> dict:
    return {k: v for k, v in res.items() if v[1]}


def audit(extra_flags=()):
    """Compile csrc/conv_bf.hip to gfx950 assembly with the library's compiler and flags (+ extra_flags) and scan it."""
    from . import build
    src = os.path.join(build.CSRC, "conv_bf.hip")
    fd, out = tempfile.mkstemp(prefix="kmh_conv_bf_scan_", suffix=".s")
    os.close(fd)
    try:
        cmd = [build._hipcc(), *build.FLAGS, *build.FILE_FLAGS.get("conv_bf.hip", []), "-S", "--cuda-device-only", *extra_flags,
               src, "-o", out]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError("ISA audit: %s failed (%d):\n%s" % (" ".join(cmd), r.returncode, r.stderr[-4000:]))
        with open(out) as f:
            return scan(f.read())
    finally:
        if os.path.exists(out):
            os.remove(out)
