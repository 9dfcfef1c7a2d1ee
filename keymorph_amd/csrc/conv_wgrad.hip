// The weight-gradient section of conv_bf.hip as its own translation unit (see the note at the top of that file):
// conv3_wgrad_ws_kernel / conv3_wgrad_bf_kernel, their reduce kernels and the kmh_conv3d_wgrad_bf* entry points, compiled
// with -mllvm -amdgpu-sched-strategy=max-ilp (keymorph_amd/build.py FILE_FLAGS).
#define KMH_TU_WGRAD 1
#include "conv_bf.hip"
