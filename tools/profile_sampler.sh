#!/bin/bash
# Round 6: counters of the trilinear sampler kernels (sample_fwd_lc and friends) on the bench's 256^3 shapes.
#   tools/profile_sampler.sh r6a  ->  gpurun_out/r6a_sampler_counters.txt  (+ the list of counters this box offers)
# One rocprofv3 pass per counter group, --pmc with --kernel-trace only (no --stats / sys-trace beside --pmc).
tag=${1:-r6x}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/${tag}_sampler_counters.txt
: > $out
rocprofv3 --list-avail > gpurun_out/${tag}_counters_avail.txt 2>&1 || true
GROUPS_=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"
  "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_WRITE_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"
  "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "MeanOccupancyPerCU"
  "TCC_BUBBLE_sum TCC_EA_RDREQ_DRAM_sum TCC_EA_WRREQ_DRAM_sum"
)
run_grp() {   # $1 label; rest command
  local label="$1"; shift
  echo "== $label" >> $out
  local i=0 dirs=""
  for P in "${GROUPS_[@]}"; do
    i=$((i+1))
    rm -rf gpurun_out/sp_$i
    if rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/sp_$i -- "$@" > gpurun_out/sp_log.txt 2>&1; then
      dirs="$dirs gpurun_out/sp_$i"
    else
      echo "   (pass failed: $P :: $(grep -i -m1 'error\|invalid\|not found' gpurun_out/sp_log.txt))" >> $out
    fi
  done
  KMH_PMC_ONLY=sample_,warp_ python tools/pmc_agg.py $dirs >> $out
  for i in $(seq 1 ${#GROUPS_[@]}); do rm -rf gpurun_out/sp_$i; done
  rm -f gpurun_out/sp_log.txt
}
echo "== python tools/bench_sampler.py 256 (times without a profiler)" >> $out
GRID=affine3 python tools/bench_sampler.py 256 >> $out 2>&1
KMH_SAMPLER_QUICK=1 GRID=affine3 run_grp "KMH_SAMPLER_QUICK=1 GRID=affine3 python tools/bench_sampler.py 256 (every launch of the script summed per kernel; dispatches per kernel in brackets)" python tools/bench_sampler.py 256
