#!/usr/bin/env python3
"""Sum rocprofv3 --pmc csv counters per kernel (all dispatches): pmc_agg.py DIR [DIR...]"""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"^void ", "", k).split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
import os
ONLY = tuple(t for t in os.environ.get("KMH_PMC_ONLY", "").split(",") if t)      # e.g. KMH_PMC_ONLY=sample_,warp_
for k, v in agg.items():
    if not any(t in k for t in (ONLY or ("conv3", "headcom", "tps", "sample", "warp_dice", "dice_partial", "elementwise", "copy"))):
        continue
    print(f"{k}   [{len(cnt[k])} dispatches]")
    for c, x in sorted(v.items()):
        print(f"   {c:34s} {x:16.0f}")
