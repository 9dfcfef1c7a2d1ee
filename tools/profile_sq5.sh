#!/bin/bash
# Round 5: SQ counters of every forward / data-gradient variant of the 27-tap family on the shapes the bench step gives
# them, plus cycle stamps (KMH_G_TRACE) of the same launches.  Three passes of <= 8 SQ counters each, no --stats beside --pmc.
#   tools/profile_sq5.sh r5a  ->  gpurun_out/r5a_sq_counters_fwd_family.txt , gpurun_out/r5a_cycle_stamps.txt
# Shapes (prof_layer.py D Cin Cout, N = 2): forward = Cin -> Cout, data gradient = Cout -> Cin
#   128 64 64 : conv3_fwd_s<2> both ways          128 32 32 : conv3_fwd_s<1> both ways
#   256 16 32 : conv3_fwd_s<1> forward, conv3_fwd_g<1,ZP> data gradient (KEYMORPH_FWD_S=3: conv3_fwd_s<1,ZP>)
#   prof_pool.py: conv3_fwd_g<1,false,POOL> (16 -> 32 at 256^3 + pooling)
set -e
tag=${1:-r5x}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/${tag}_sq_counters_fwd_family.txt
: > $out
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_CVT SQ_WAVES SQ_INST_CYCLES_VMEM_RD"
run_sq() {   # $1 = label, rest = command
  local label="$1"; shift
  echo "== $label" >> $out
  local i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rm -rf gpurun_out/sq_$i
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/sq_$i -- "$@" > /dev/null 2>&1 || echo "pass $i failed" >> $out
  done
  python tools/pmc_agg.py gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 >> $out
  rm -rf gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3
}
for shape in "128 64 64" "128 32 32" "256 16 32"; do
  run_sq "python tools/prof_layer.py $shape f16x3 nomask (N = 2)" python tools/prof_layer.py $shape f16x3 nomask
done
KEYMORPH_FWD_S=3 run_sq "KEYMORPH_FWD_S=3 python tools/prof_layer.py 256 16 32 f16x3 nomask (N = 2; the one-wave z-paired data gradient)" python tools/prof_layer.py 256 16 32 f16x3 nomask
run_sq "python tools/prof_pool.py (16 -> 32 at 2 x 256^3 with the pooling epilogue, then without)" python tools/prof_pool.py
run_sq "python tools/prof_split.py 256 (32 -> 16 data gradient at 2 x 256^3: blocked fp32 operand = conv3_fwd_g<1,ZP>, pre-split = conv3_fwd_s<1,ZP,SPLIT>)" python tools/prof_split.py 256
for shape in "128 64 64" "128 32 32" "256 16 32"; do
  echo "== KMH_TIME=1 python tools/prof_layer.py $shape f16x3 nomask" >> $out
  KMH_TIME=1 python tools/prof_layer.py $shape f16x3 nomask >> $out 2>&1
done
echo "== KEYMORPH_FWD_S=3 KMH_TIME=1 python tools/prof_layer.py 256 16 32 f16x3 nomask" >> $out
KEYMORPH_FWD_S=3 KMH_TIME=1 python tools/prof_layer.py 256 16 32 f16x3 nomask >> $out 2>&1
python tools/prof_pool.py >> $out 2>&1
# cycle stamps of workgroup 0, wave 0 (differences between consecutive stamps; see the stamp() calls in conv_bf.hip)
st=gpurun_out/${tag}_cycle_stamps.txt
: > $st
for shape in "128 64 64" "128 32 32" "256 16 32"; do
  KMH_G_TRACE=1 python tools/prof_layer.py $shape f16x3 nomask 2>&1 | grep KMH_G_TRACE | tail -3 >> $st
done
KEYMORPH_FWD_S=3 KMH_G_TRACE=1 python tools/prof_layer.py 256 16 32 f16x3 nomask 2>&1 | grep KMH_G_TRACE | tail -3 >> $st
KMH_G_TRACE=1 python tools/prof_pool.py 2>&1 | grep KMH_G_TRACE | head -3 >> $st
KMH_G_TRACE=1 python tools/prof_split.py 256 2>&1 | grep "SPLIT=1" | tail -1 >> $st
head -c 6000 $out
