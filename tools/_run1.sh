cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python tools/diag_backbone_fp64.py 128 2>&1 | grep -v amdgpu.ids | cut -c1-400 > gpurun_out/r5e_diag.txt
cat gpurun_out/r5e_diag.txt
