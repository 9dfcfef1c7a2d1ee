cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r5g_sweep.txt
for r in 1 2; do
for v in default 9_8 6_5 6_3 18_8 18_16; do
  if [ $v = default ]; then L=""; else L="KEYMORPH_HIP_LIB=keymorph_amd/lib/ab/sp_$v.so"; fi
  echo "== $v: $(env $L timeout 300 python tools/prof_split.py 256 2>&1 | grep "data gradient\|bit-identical" | tr '\n' ' ')" >> gpurun_out/r5g_sweep.txt
done; done
cat gpurun_out/r5g_sweep.txt
bash tools/profile_round.sh r5g 2>&1 | tail -30
