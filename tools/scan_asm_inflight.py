#!/usr/bin/env python3
"""Audit of the hand-waited inline-asm loads of csrc/conv_bf.hip.

An `asm volatile("global_load_dwordx4 %0, ...")` destination is "defined", for the compiler, the moment the statement ends -- long
before the data lands.  Under register pressure the compiler has been seen to COPY such a register (v_accvgpr_write / v_mov /
scratch store) while the load was in flight: the copy holds stale data and the load lands in a register that has meanwhile been
given to something else.  This scans the gfx950 assembly of every conv3_fwd_[sg]_kernel instance: between an inline-asm load and
the wait that covers it NO instruction may mention its destination registers.  A wait covers a load if it is `vmcnt(N)` with N <=
the number of vector-memory instructions issued after the load (loads return in issue order; stores in flight only make a
counted wait stricter -- the kernels never wait with a count for a load that is older than a store, see conv_bf.hip).
The scan is linear in text order (the fall-through path of every branch).
usage: tools/scan_asm_inflight.py [extra hipcc flags]   (exit code 1 if any kernel touches an in-flight destination)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = re.compile(r'^(global_|buffer_|flat_|scratch_)(load|store|atomic)')


def regs_of(line, kind):
    """register numbers of file `kind` ('v' / 'a') an instruction line mentions"""
    out = set()
    ops = line.split(None, 1)[1] if ' ' in line else ''
    for m in re.finditer(r'\b%s\[(\d+):(\d+)\]' % kind, ops):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\b%s(\d+)\b' % kind, ops):
        out.add(int(m.group(1)))
    return out


def audit(extra_flags=()):
    """{kernel name: (inline-asm loads scanned, loads whose destination is touched while in flight)} for every conv3_fwd_[sg]_kernel instance."""
    out = os.path.join(tempfile.gettempdir(), "kmh_conv_bf_scan_%d.s" % os.getpid())
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden",
                    "-Wno-unused-result", "-ffp-contract=fast", "-S", "--cuda-device-only", *extra_flags,
                    os.path.join(ROOT, "keymorph_amd/csrc/conv_bf.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.remove(out)
    res = {}
    for m in re.finditer(r'^(_ZN12_GLOBAL__N_1\d+(conv3_fwd_[sg]_kernel\S*?)): ', txt, re.M):
        raw = txt[m.end():txt.index('.Lfunc_end', m.end())].split('\n')
        body, in_asm, flag = [], False, []
        for l in raw:                                   # instructions only, each with "is inside an inline-asm block"
            t = l.strip()
            if '#ASMSTART' in t:
                in_asm = True
            elif '#ASMEND' in t:
                in_asm = False
            elif t and not t.startswith((';', '.')) and not t.endswith(':'):
                body.append(t)
                flag.append(in_asm)
        loads = bad = 0
        for i, l in enumerate(body):
            mm = re.match(r'global_load_dwordx4 ([va])\[(\d+):(\d+)\]', l)
            if not mm or not flag[i]:                   # only the inline-asm loads: the compiler waits for its own loads itself
                continue
            loads += 1
            kind, dst = mm.group(1), set(range(int(mm.group(2)), int(mm.group(3)) + 1))
            younger = 0
            for t in body[i + 1:]:
                w = re.search(r'vmcnt\((\d+)\)', t) if t.startswith('s_waitcnt') else None
                if w and int(w.group(1)) <= younger:    # all but the `younger` youngest have landed: this one has
                    break
                if t.startswith('s_endpgm'):
                    break
                if VMEM.match(t):
                    younger += 1
                if regs_of(t, kind) & dst:
                    bad += 1
                    break
        res[re.sub(r'^\d+', '', m.group(2))] = (loads, bad)
    return res


if __name__ == "__main__":
    r = audit(sys.argv[1:])
    for name, (loads, bad) in r.items():
        print(f"{name[:60]:62s} asm loads {loads:4d}   destinations touched while in flight: {bad}")
    sys.exit(1 if any(b for _, b in r.values()) else 0)
