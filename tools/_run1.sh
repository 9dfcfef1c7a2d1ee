cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r5j_stagger.txt
for r in 1 2; do for pct in 0 25 40 15; do
  echo "== pct $pct: 128^3 64->64 $(KMH_STAGGER_PCT=$pct KMH_TIME=1 timeout 300 python tools/prof_layer.py 128 64 64 f16x3 nomask 2>&1 | grep "fwd\|dgrad" | tr '\n' ' ') | 128^3 32->32 $(KMH_STAGGER_PCT=$pct KMH_TIME=1 timeout 300 python tools/prof_layer.py 128 32 32 f16x3 nomask 2>&1 | grep "fwd\|dgrad" | tr '\n' ' ') | split $(KMH_STAGGER_PCT=$pct timeout 300 python tools/prof_split.py 256 2>&1 | grep "data gradient" | sed 's/.*pre-split//') | pool $(KMH_STAGGER_PCT=$pct timeout 300 python tools/prof_pool.py 2>&1 | grep True | sed 's/(incl.*//')" >> gpurun_out/r5j_stagger.txt
done; done
KMH_STAGGER_PCT=25 KMH_G_TRACE=1 timeout 300 python tools/prof_layer.py 128 64 64 f16x3 nomask 2>&1 | grep KMH_G_TRACE | tail -1 | cut -c1-700 >> gpurun_out/r5j_stagger.txt
cat gpurun_out/r5j_stagger.txt
