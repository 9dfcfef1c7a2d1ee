cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r5n_steps.txt
for L in st_cw; do for sh in "128 64 64" "128 32 32"; do
  echo "== $L $sh" >> gpurun_out/r5n_steps.txt
  KEYMORPH_HIP_LIB=keymorph_amd/lib/ab/$L.so KMH_G_TRACE=1 timeout 300 python tools/prof_layer.py $sh f16x3 nomask 2>&1 | grep KMH_G_TRACE | tail -2 | cut -c1-1800 >> gpurun_out/r5n_steps.txt
done; done
cat gpurun_out/r5n_steps.txt
