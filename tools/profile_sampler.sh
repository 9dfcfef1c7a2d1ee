#!/bin/bash
# Round 6: counters of the bilinear warp kernels (256^3, the bench's affine grid), three launches each of the stand-alone warp
# and of the fused warp + MSE + d loss / d grid.   tools/profile_sampler.sh r6d  ->  gpurun_out/r6d_sampler_counters.txt
# One rocprofv3 pass per SMALL counter group (the TA / TCP blocks take two counters at a time: a group of four aborted rocprofv3
# with "Request exceeds the capabilities of the hardware to collect" and then hung), every pass under its own timeout;
# --pmc with --kernel-trace only (no --stats / sys-trace beside --pmc).
tag=${1:-r6x}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/${tag}_sampler_counters.txt
echo "== python tools/prof_sampler_min.py  (3 dispatches per kernel)" > $out
GROUPS_=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU"
  "TA_TA_BUSY_sum TA_BUSY_avr"
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum"
  "TCC_HIT_sum TCC_MISS_sum"
  "MeanOccupancyPerCU"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0; dirs=""
for P in "${GROUPS_[@]}"; do
  i=$((i+1))
  rm -rf gpurun_out/sp_$i
  if timeout 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/sp_$i -- python tools/prof_sampler_min.py > gpurun_out/sp_log.txt 2>&1; then
    dirs="$dirs gpurun_out/sp_$i"
  else
    echo "   (pass failed or timed out: $P :: $(grep -i -m1 'error code\|invalid\|not found' gpurun_out/sp_log.txt | cut -c1-160))" >> $out
  fi
done
KMH_PMC_ONLY=sample_ python tools/pmc_agg.py $dirs >> $out
for j in $(seq 1 $i); do rm -rf gpurun_out/sp_$j; done
rm -f gpurun_out/sp_log.txt
