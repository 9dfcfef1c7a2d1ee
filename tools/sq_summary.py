#!/usr/bin/env python3
"""Derived figures from a tools/profile_sq*.sh counter dump: per section and 27-tap / weight-gradient kernel
   mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs)   (both summed over the dispatches)
   per-MFMA    = other instructions issued per MFMA: VALU (SQ_INSTS_VALU counts MFMAs too), LDS, SALU, VMEM
   wait / stall/ active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of a wave's life)
   lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;  lds_stall = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
usage: tools/sq_summary.py gpurun_out/r5a_sq_counters_fwd_family.txt"""
import re, sys
sec, cur, rows = None, None, []
for line in open(sys.argv[1]):
    line = line.rstrip("\n")
    if line.startswith("== "):
        sec = line[3:]; cur = None
    elif line and not line.startswith(" "):
        cur = {"sec": sec, "k": line}
        rows.append(cur)
    elif cur is not None and line.startswith("   "):
        p = line.split()
        if len(p) == 2:
            try: cur[p[0]] = float(p[1])
            except ValueError: pass
print("| run | kernel | launches | MFMA busy | VALU / LDS / SALU / VMEM per MFMA | parked / issue-stalled / issuing | LDS conflict | LDS stall |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    if not re.match(r"conv3_(fwd|wgrad)", r["k"]) or not r.get("SQ_INSTS_MFMA"):
        continue
    m = r["SQ_INSTS_MFMA"]
    busy = r["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (r["GRBM_GUI_ACTIVE"] / 8.0)
    wc = r["SQ_WAVE_CYCLES"]
    wpk = {"conv3_fwd_s": 4, "conv3_fwd_g": 8, "conv3_fwd_bf": 4}.get(r["k"].split("_kernel")[0], None)
    launches = f"{r['SQ_WAVES']:.0f} waves"
    vm = r.get("SQ_INSTS_VMEM_RD", 0) + r.get("SQ_INSTS_VMEM_WR", 0)
    print(f"| {r['sec'][:46]} | `{r['k']}` | {launches} | {100 * busy:.1f} % | {(r['SQ_INSTS_VALU'] - m) / m:.2f} / {r['SQ_INSTS_LDS'] / m:.2f} / "
          f"{r['SQ_INSTS_SALU'] / m:.2f} / {vm / m:.2f} | {100 * r['SQ_WAIT_ANY'] / wc:.0f} / {100 * r['SQ_WAIT_INST_ANY'] / wc:.0f} / "
          f"{100 * r['SQ_ACTIVE_INST_ANY'] / wc:.0f} % | {100 * r['SQ_LDS_BANK_CONFLICT'] / max(r['SQ_LDS_IDX_ACTIVE'], 1):.0f} % | "
          f"{100 * r['SQ_WAIT_INST_LDS'] / wc:.1f} % |")
