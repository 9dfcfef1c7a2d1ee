#!/bin/bash
# counters of the fused warp + Dice kernels and the C-channel sampler (tools/bench_warp_dice.py): HBM traffic + SQ view
tag=${1:-rX}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${tag}_counters_warp_dice.txt
: > $out
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/wd_$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/wd_$i -- python tools/bench_warp_dice.py > /dev/null 2>&1 || echo "pass $i ($P) failed" >> $out
done
python tools/pmc_agg.py gpurun_out/wd_* >> $out
rm -rf gpurun_out/wd_*
cat $out
