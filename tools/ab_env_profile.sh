#!/bin/bash
# usage: tools/ab_env_profile.sh VAR   -- kernel-trace the headline step with VAR unset and VAR=1; print the kernels whose
# total time differs (run on the MI355X box through gpurun)
var=$1
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --also-f32 0 --dice 0 --eval-steps 0 --groupwise 0 --convnet 0 --sampler 0 --steps 3 --warmup 1"
for v in off on; do
  rm -rf gpurun_out/prof_ab_$v
  if [ $v = on ]; then export $var=1; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ab_$v -o bench -- $B > /dev/null 2>&1
  python tools/rocpd_summary.py $(ls gpurun_out/prof_ab_$v/*/*results.db gpurun_out/prof_ab_$v/*results.db 2>/dev/null | head -1) --md gpurun_out/ab_$v.md > /dev/null
  rm -rf gpurun_out/prof_ab_$v
done
python - <<'PY'
import re
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"\| `(.*)` \| (\d+) \| ([\d.]+) \|", l)
        if m: d[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load("gpurun_out/ab_off.md"), load("gpurun_out/ab_on.md")
tot = 0
for k in sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1])):
    da = a.get(k, (0, 0)); db = b.get(k, (0, 0))
    tot += db[1] - da[1]
    if abs(da[1] - db[1]) > 0.05:
        print(f"{db[1] - da[1]:+8.3f} ms  unset {da[0]}x {da[1]:.3f}  set {db[0]}x {db[1]:.3f}  {k[:90]}")
print("sum (set - unset), ms over 4 steps:", round(tot, 3))
PY
