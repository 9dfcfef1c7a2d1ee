python -m pytest tests -m gpu -q -x -k "pool or unet or backbone or gradients" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x -o b -- python bench.py --no-cpu-baseline --also-f32 0 --dice 0 --steps 2 --warmup 1 > /dev/null 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/prof_x/*/*results.db gpurun_out/prof_x/*results.db 2>/dev/null | head -1) --md gpurun_out/x_stats.md > /dev/null
grep -i "maxpool\|gn_bwd_apply\|upcat" gpurun_out/x_stats.md | cut -c1-150
rm -rf gpurun_out/prof_x
