#!/bin/bash
# Round 6: a per-kernel survey of the headline step with a handful of counters (one rocprofv3 pass per group, --pmc with
# --kernel-trace only): matrix-pipe busy, VALU busy, issue-stall share, LDS busy / conflicts, TA busy -- where does each kernel sit?
#   tools/survey_step_counters.sh r6q  ->  gpurun_out/r6q_step_survey.txt
tag=${1:-r6x}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --also-f32 0 --first-block-exact 0 --amp 0 --dice 0 --eval-steps 0 --groupwise 0 --convnet 0 --sampler 0 --cpu-256 0 --steps 1 --warmup 1"
GROUPS_=(
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"
  "TA_TA_BUSY_sum TA_BUSY_avr"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES"
)
i=0; dirs=""
for P in "${GROUPS_[@]}"; do
  i=$((i+1)); rm -rf gpurun_out/sv_$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/sv_$i -- $B > /dev/null 2>&1 && dirs="$dirs gpurun_out/sv_$i"
done
python - $dirs <<'PY' > gpurun_out/${tag}_step_survey.txt
import collections, csv, glob, re, sys
agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k).split("(")[0]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); disp[(d,k)].add(r["Dispatch_Id"])
print("per kernel over the bench process (2 steps + setup); cycles = GRBM_GUI_ACTIVE / 8 XCDs; busy figures relative to 256 CUs x 4 SIMDs")
print("Mcycles | launches | MFMA busy | VALU busy | issue-stalled (WAIT_INST_ANY / WAVE_CYCLES) | LDS busy | LDS conflicts | TA busy | VALU per MFMA | kernel")
rows=[]
for k,v in agg.items():
    cyc=v.get("GRBM_GUI_ACTIVE",0)/8
    if cyc<=0: continue
    n=max(len(s) for (d,kk),s in disp.items() if kk==k)
    mf=v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/1024/cyc
    va=v.get("SQ_ACTIVE_INST_VALU",0)*4/1024/cyc      # quad-cycles -> cycles
    st=v.get("SQ_WAIT_INST_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1)
    lds=v.get("SQ_LDS_IDX_ACTIVE",0)/256/cyc
    cf=v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_LDS_IDX_ACTIVE",1),1)
    ta=v.get("TA_BUSY_avr",0)/max(n,1)/ (cyc/max(n,1)) if v.get("TA_BUSY_avr") else 0
    vpm=(v.get("SQ_INSTS_VALU",0)-v.get("SQ_INSTS_MFMA",0))/v["SQ_INSTS_MFMA"] if v.get("SQ_INSTS_MFMA") else float('nan')
    rows.append((cyc,n,mf,va,st,lds,cf,ta,vpm,k))
rows.sort(key=lambda r:-r[0])
for cyc,n,mf,va,st,lds,cf,ta,vpm,k in rows[:36]:
    print(f"{cyc/1e6:8.2f} | {n:4d} | {100*mf:5.1f} % | {100*va:5.1f} % | {100*st:5.1f} % | {100*lds:5.1f} % | {100*cf:5.1f} % | {100*ta:5.1f} % | {vpm:6.2f} | {k[:90]}")
PY
for j in $(seq 1 $i); do rm -rf gpurun_out/sv_$j; done
cat gpurun_out/${tag}_step_survey.txt
