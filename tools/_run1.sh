cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5c_tests.txt
cat gpurun_out/r5c_tests.txt
