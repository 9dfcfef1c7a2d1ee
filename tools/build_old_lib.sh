#!/bin/bash
# Build the library of another commit (default HEAD) into keymorph_amd/lib/ab/libkeymorph_hip_old.so for A/B runs
# (tools/ab_layers.sh, KMH_LIB=... in tools/prof_layer.py).  The directory is git-ignored but travels with gpurun.
set -e
rev=${1:-HEAD}
root=$(git rev-parse --show-toplevel)
out=$root/keymorph_amd/lib/ab
rm -rf $out/src && mkdir -p $out/src/csrc $out/src/include
git -C $root archive $rev keymorph_amd/csrc include | tar -x -C $out/src --strip-components=0
objs=""
for f in $out/src/keymorph_amd/csrc/*.hip; do
  o=$out/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-result -Wno-unused-value \
    -ffp-contract=fast -I$out/src/include $( case $(basename $f) in conv_wgrad.hip|norm.hip) echo "-mllvm -amdgpu-sched-strategy=max-ilp";; esac ) -c $f -o $o &
  objs="$objs $o"
done
wait
g++ -shared -fPIC -o $out/libkeymorph_hip_old.so $objs
rm -rf $out/src $out/*.o
ls -la $out
