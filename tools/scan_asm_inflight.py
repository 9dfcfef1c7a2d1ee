#!/usr/bin/env python3
"""Audit of the hand-waited inline-asm loads of csrc/conv_bf.hip: command-line front end of keymorph_amd/isa_audit.py (the scan
itself, its rules and what it guards against are described there; keymorph_amd/build.py runs the same scan on the assembly of
every library build).
usage: tools/scan_asm_inflight.py [extra hipcc flags]   (exit code 1 if any kernel touches an in-flight destination)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keymorph_amd.isa_audit import audit, failures, regs_of, scan  # noqa: E402,F401

if __name__ == "__main__":
    r = audit(sys.argv[1:])
    for name, (loads, bad) in r.items():
        print(f"{name[:60]:62s} asm loads {loads:4d}   destinations touched while in flight: {bad}")
    sys.exit(1 if failures(r) else 0)
