#!/bin/bash
# Build an EXPERIMENT copy of the library into keymorph_amd/lib/ab/libkeymorph_hip_old.so: the working tree's sources
# with extra compiler flags (e.g. -DKMH_EXP_SOMETHING guarding a throw-away change).  For A/B timing only.
set -e
root=$(git rev-parse --show-toplevel)
out=$root/keymorph_amd/lib/ab
mkdir -p $out
objs=""
for f in $root/keymorph_amd/csrc/*.hip; do
  o=$out/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-result -Wno-unused-value \
    -ffp-contract=fast -I$root/include $( case $(basename $f) in conv_wgrad.hip|norm.hip) echo "-mllvm -amdgpu-sched-strategy=max-ilp";; esac ) "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
g++ -shared -fPIC -o $out/libkeymorph_hip_old.so $objs
rm -f $out/*.o
