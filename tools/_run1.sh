cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_dispatch_gpu.py -x -q -k "tie_rule" 2>&1 | grep -v "^  \|^$" | tail -12 | cut -c1-1500 > gpurun_out/r5d_tests.txt
KEYMORPH_POOL_G=1 timeout 600 python -m pytest tests/test_conv_dispatch_gpu.py -x -q -k "tie_rule" 2>&1 | tail -3 | cut -c1-600 >> gpurun_out/r5d_tests.txt
KEYMORPH_NO_SPLIT_POOLGRAD=1 timeout 600 python -m pytest tests/test_conv_dispatch_gpu.py -x -q -k "tie_rule" 2>&1 | tail -3 | cut -c1-600 >> gpurun_out/r5d_tests.txt
cat gpurun_out/r5d_tests.txt
