# A/B of the LDS-DMA forward / data-gradient kernel (KEYMORPH_FWD_G=1, default) against conv3_fwd_bf_kernel (=0),
# same library, same process layout: per-launch times of tools/prof_layer.py for the layers of the bench step
export KMH_TIME=1
for cfg in "128 192 64" "128 64 64" "128 32 64" "64 128 128" "64 64 128" "256 16 32" "128 32 32" "64 384 128" "256 32 32"; do
  for g in 0 1; do
    echo "== $cfg FWD_G=$g: $(KEYMORPH_FWD_G=$g python tools/prof_layer.py $cfg f16x3 nomask 2>/dev/null | grep -v done | tr '\n' ' ')"
  done
done
