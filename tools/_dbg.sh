export KMH_TIME=1
for cfg in "256 16 32" "128 32 32" "128 64 64"; do
  echo "== $cfg base   : $(python tools/prof_layer.py $cfg f16x3 nomask 2>/dev/null | grep -v "done\|wgrad" | tr '\n' ' ')"
  echo "== $cfg nostore: $(KMH_LIB=keymorph_amd/lib/ab/libkeymorph_hip_old.so python tools/prof_layer.py $cfg f16x3 nomask 2>/dev/null | grep -v "done\|wgrad" | tr '\n' ' ')"
done
