#!/usr/bin/env python3
"""Sustained shader clock per kernel: GRBM_GUI_ACTIVE (cycles the GPU was active during the dispatch) / the dispatch's
duration, from `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- cmd`.
Usage: tools/effective_clock.py DIR"""
import collections, csv, glob, re, sys
d = sys.argv[1]
cnt = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[r["Dispatch_Id"]] = (float(r["Counter_Value"]), r["Kernel_Name"], float(r.get("Start_Timestamp", 0) or 0), float(r.get("End_Timestamp", 0) or 0))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for did, (c, k, s, e) in cnt.items():
    if e > s:
        k = re.sub(r"\(anonymous namespace\)::", "", k).split("(")[0][:70]
        agg[k][0] += c; agg[k][1] += e - s; agg[k][2] += 1
tot_c = tot_t = 0
for k, (c, t, n) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print(f"{c / t:6.3f} GHz  {t / 1e6:9.3f} ms  {n:5d}x  {k}")
    tot_c += c; tot_t += t
print(f"{tot_c / tot_t:6.3f} GHz over the kernels above (GRBM_GUI_ACTIVE counts per XCD? if the figures read 8x the clock, divide)")
