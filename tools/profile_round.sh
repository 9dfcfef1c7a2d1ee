#!/bin/bash
# Per-round rocprofv3 evidence for bench.py's numbers (run on the MI355X box through gpurun):
#   tools/profile_round.sh r2c   ->  gpurun_out/<tag>_{bench.json,kernel_trace_stats.md,pmc_hbm_traffic.json}
# (copy them into profiles/ afterwards).  Counters are collected in their own passes, without --stats / sys-trace.
set -e
tag=${1:-rX}
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --also-f32 0 --first-block-exact 0 --amp 0 --dice 0 --eval-steps 0 --groupwise 0 --convnet 0 --sampler 0"
python bench.py --steps 5 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag} -o bench -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/prof_${tag}/*/*results.db gpurun_out/prof_${tag}/*results.db 2>/dev/null | head -1) --md gpurun_out/${tag}_kernel_trace_stats.md > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f_${tag} -- $B --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w_${tag} -- $B --steps 1 --warmup 1 > /dev/null 2>&1
python tools/pmc_traffic.py $(dirname $(ls gpurun_out/pmc_f_${tag}/*/*counter_collection.csv | head -1)) $(dirname $(ls gpurun_out/pmc_w_${tag}/*/*counter_collection.csv | head -1)) gpurun_out/${tag}_pmc_hbm_traffic.json | head -8
rm -rf gpurun_out/prof_${tag} gpurun_out/pmc_f_${tag} gpurun_out/pmc_w_${tag}
head -12 gpurun_out/${tag}_kernel_trace_stats.md
echo "after copying the four ${tag}_* files into profiles/:  echo ${tag} > profiles/LATEST   (bench.py reads the traffic summary it names)"
python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print({k: d[k] for k in ('value','ms_per_step','dice_pairs_per_s','f32_mfma_ms_per_step') if k in d}); print(d['roofline']); print(d.get('cpu_baseline'))"
