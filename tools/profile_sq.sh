#!/bin/bash
# SQ counters of the dominant kernel (conv3_fwd_g_kernel) on the shapes the bench step gives it, three passes of <= 8 SQ
# counters each (MI355X_MICROARCH.md "rocprofv3 PMC slots"), no --stats / sys-trace beside --pmc:
#   tools/profile_sq.sh r3x  ->  gpurun_out/r3x_sq_counters_conv3_fwd_g.txt   (copy into profiles/)
# Reading: matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs ... see README)
set -e
tag=${1:-rX}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/${tag}_sq_counters_conv3_fwd_g.txt
: > $out
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_CVT SQ_WAVES SQ_INST_CYCLES_VMEM_RD"
for shape in "128 64 64" "256 16 32"; do
  echo "== python tools/prof_layer.py $shape f16x3 nomask   (N = 2; forward + premasked data gradient = conv3_fwd_g_kernel, + the weight gradient)" >> $out
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rm -rf gpurun_out/sq_$i
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/sq_$i -- python tools/prof_layer.py $shape f16x3 nomask > /dev/null 2>&1 || echo "pass $i failed" >> $out
  done
  python tools/pmc_agg.py gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 >> $out
  rm -rf gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3
done
KMH_TIME=1 python tools/prof_layer.py 128 64 64 f16x3 nomask >> $out 2>&1
KMH_TIME=1 python tools/prof_layer.py 256 16 32 f16x3 nomask >> $out 2>&1
cat $out | head -80
