for r in 1 2; do
echo "new: $(python tools/prof_head.py 128 64 512 4)"
echo "old: $(KMH_LIB=keymorph_amd/lib/ab/libkeymorph_hip_old.so python tools/prof_head.py 128 64 512 4)"
done
