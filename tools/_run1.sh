cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv_dispatch_gpu.py -x -q -k "pool" 2>&1 | tail -4 > gpurun_out/r5h_tests.txt
for i in 1 2; do timeout 300 python tools/prof_pool.py 2>&1 | grep True >> gpurun_out/r5h_tests.txt; KEYMORPH_POOL_G=1 timeout 300 python tools/prof_pool.py 2>&1 | grep True >> gpurun_out/r5h_tests.txt; done
KMH_G_TRACE=1 timeout 300 python tools/prof_pool.py 2>&1 | grep "POOL=1" | tail -1 | cut -c1-500 >> gpurun_out/r5h_tests.txt
cat gpurun_out/r5h_tests.txt
