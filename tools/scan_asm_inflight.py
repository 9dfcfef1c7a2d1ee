#!/usr/bin/env python3
"""Audit of the hand-counted inline-asm loads: an `asm volatile("global_load_dwordx4 %0, ...")` destination is "defined" for the
compiler the moment the statement ends, so under register pressure it may COPY that register (v_accvgpr_write / v_mov / scratch
store) before the data has arrived -- and the load then lands in a register that has been given to something else.  This scans the
gfx950 assembly of csrc/conv_bf.hip for any copy of a global_load_dwordx4 destination between the load and the next full drain
(`s_waitcnt vmcnt(0)`), per kernel.  usage: tools/scan_asm_inflight.py [extra hipcc flags]   (exit code 1 if any kernel has one)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COPY_V = r'(v_accvgpr_write_b32 a\d+, v%d\b|v_mov_b32_e32 v\d+, v%d\b|scratch_store\w* \S+, v%d\b)'
COPY_A = r'(v_accvgpr_read_b32 v\d+, a%d\b|v_accvgpr_mov_b32 a\d+, a%d\b|scratch_store\w* \S+, a%d\b)'


def audit(extra_flags=()):
    """{kernel name: (inline-asm loads scanned, destinations copied while in flight)} for every conv3_fwd_[sg]_kernel instance."""
    out = os.path.join(tempfile.gettempdir(), "kmh_conv_bf_scan_%d.s" % os.getpid())
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden",
                    "-Wno-unused-result", "-ffp-contract=fast", "-S", "--cuda-device-only", *extra_flags,
                    os.path.join(ROOT, "keymorph_amd/csrc/conv_bf.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.remove(out)
    res = {}
    for m in re.finditer(r'^(_ZN12_GLOBAL__N_1\d+(conv3_fwd_[sg]_kernel\S*?)): ', txt, re.M):
        body = txt[m.end():txt.index('.Lfunc_end', m.end())].split('\n')
        loads = bad = 0
        in_asm = False
        for i, l in enumerate(body):
            if '#ASMSTART' in l:
                in_asm = True
            elif '#ASMEND' in l:
                in_asm = False
            mm = re.search(r'global_load_dwordx4 ([va])\[(\d+):(\d+)\]', l)
            if not mm or not in_asm:                 # only the inline-asm loads: the compiler waits for its own loads itself
                continue
            loads += 1
            pat = COPY_V if mm.group(1) == "v" else COPY_A
            regs = range(int(mm.group(2)), int(mm.group(3)) + 1)
            for t in body[i + 1:]:
                if 's_waitcnt' in t and 'vmcnt(0)' in t:              # only a FULL drain proves the asm load has landed
                    break
                if any(re.search(pat % (r, r, r), t) for r in regs):
                    bad += 1
                    break
        res[re.sub(r'^\d+', '', m.group(2))] = (loads, bad)
    return res


if __name__ == "__main__":
    r = audit(sys.argv[1:])
    for name, (loads, bad) in r.items():
        print(f"{name[:60]:62s} asm loads {loads:4d}   in-flight destinations copied before a full drain: {bad}")
    sys.exit(1 if any(b for _, b in r.values()) else 0)
