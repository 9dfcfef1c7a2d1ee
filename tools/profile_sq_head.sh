#!/bin/bash
# SQ counters of the fused head's kernels (tools/prof_head.py: 64 -> 512 at 4 x 128^3), three --pmc passes:
#   tools/profile_sq_head.sh r3x  ->  gpurun_out/r3x_sq_counters_head.txt   (copy into profiles/)
set -e
tag=${1:-rX}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/${tag}_sq_counters_head.txt
: > $out
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_CVT SQ_WAVES SQ_INST_CYCLES_VMEM_RD"
echo "== python tools/prof_head.py (N = 4, 128^3, 64 -> 512, f16x3; mask and recompute backward alternating)" >> $out
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf gpurun_out/sq_$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/sq_$i -- python tools/prof_head.py > /dev/null 2>&1 || echo "pass $i failed" >> $out
done
python tools/pmc_agg.py gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 >> $out
rm -rf gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3
python tools/prof_head.py >> $out 2>&1
cat $out | head -120
