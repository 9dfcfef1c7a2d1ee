cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r5f_a.txt
for t in "128 64 64" "128 32 32" "256 16 32" "64 128 128"; do echo "== AMP $t" >> gpurun_out/r5f_a.txt; KEYMORPH_AMP=1 KMH_TIME=1 timeout 300 python tools/prof_layer.py $t f16x3 nomask 2>&1 | tail -4 >> gpurun_out/r5f_a.txt; done
KEYMORPH_AMP=1 timeout 300 python tools/prof_pool.py 2>&1 | tail -3 >> gpurun_out/r5f_a.txt
KEYMORPH_AMP=1 timeout 300 python tools/prof_split.py 256 2>&1 | tail -3 >> gpurun_out/r5f_a.txt
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -q -s -k "use_amp" 2>&1 | grep "use_amp at\|passed\|failed\|Error\|assert" | cut -c1-400 >> gpurun_out/r5f_a.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err
cat gpurun_out/r5f_a.txt; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5f_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: d.get(k) for k in ("amp_pairs_per_s", "amp_ms_per_step", "amp_loss", "loss", "amp_wgrad_tflops")}, d.get("amp_roofline", {}).get("achieved"))
print(d["config"].get("rank_power_w_min_max"), d["config"].get("rank_sclk_mhz_min_max"))
PY
tail -2 gpurun_out/r5f_bench.err | cut -c1-300
