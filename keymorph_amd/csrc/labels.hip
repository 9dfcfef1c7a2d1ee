// Label-map encodings in front of the Dice branch of the registration step
// (reference: keymorph/utils.py:200-240 one_hot / one_hot_subsampled_pair; callers scripts/train.py:54-61,
// pairwise_register_eval.py).  Integer work, HBM bound: 8 B read + 4 (or 8) * C B written per voxel.
#include "common.h"

namespace {
constexpr int TPB = 256;

// flags[l] = 1 for every label l in [0, nflags) that occurs; flags[nflags] = 1 if any label falls outside
__global__ __launch_bounds__(TPB) void label_presence_kernel(const long long* __restrict__ seg, long long n, int nflags,
                                                             int* __restrict__ flags) {
  const long long stride = (long long)gridDim.x * TPB;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
    const long long l = seg[i];
    const int slot = (l >= 0 && l < nflags) ? (int)l : nflags;
    if (flags[slot] == 0) flags[slot] = 1;      // benign race: every writer stores the same value
  }
}

template <typename OUT>
__global__ __launch_bounds__(TPB) void one_hot_select_kernel(const long long* __restrict__ seg, long long V,
                                                             const long long* __restrict__ labels, int C,
                                                             OUT* __restrict__ out) {
  __shared__ long long lab[256];
  for (int c = threadIdx.x; c < C; c += TPB) lab[c] = labels[c];
  __syncthreads();
  const int n = blockIdx.y;
  const long long stride = (long long)gridDim.x * TPB;
  for (long long v = (long long)blockIdx.x * TPB + threadIdx.x; v < V; v += stride) {
    const long long l = seg[(long long)n * V + v];
    OUT* o = out + (long long)n * C * V + v;
    for (int c = 0; c < C; ++c) o[(long long)c * V] = (OUT)(l == lab[c]);
  }
}
}  // namespace

/* flags: nflags + 1 ints, zeroed by the caller; on return flags[l] = 1 iff label l occurs in seg (n int64 values),
 * flags[nflags] = 1 iff some label is < 0 or >= nflags. */
KMH_API int kmh_label_presence(const long long* seg, long long n, int nflags, int* flags, void* stream) {
  if (n <= 0 || nflags <= 0) return -22;
  long long nb = (n + TPB * 8 - 1) / (TPB * 8);
  if (nb > 4096) nb = 4096;
  label_presence_kernel<<<(int)nb, TPB, 0, (hipStream_t)stream>>>(seg, n, nflags, flags);
  return KMH_LAUNCH_CHECK();
}

/* out[n, c, v] = (seg[n, v] == labels[c]) for C <= 256 selected labels; out is float32 (out_i64 = 0: what
 * one_hot_subsampled_pair returns) or int64 (out_i64 = 1: what F.one_hot returns).  keymorph/utils.py:200-240 */
KMH_API int kmh_one_hot_select(const long long* seg, int N, long long V, const long long* labels, int C, void* out,
                               int out_i64, void* stream) {
  if (N <= 0 || V <= 0 || C <= 0 || C > 256) return -22;
  long long nb = (V + TPB * 4 - 1) / (TPB * 4);
  if (nb > 2048) nb = 2048;
  dim3 grid((int)nb, N);
  if (out_i64)
    one_hot_select_kernel<long long><<<grid, TPB, 0, (hipStream_t)stream>>>(seg, V, labels, C, (long long*)out);
  else
    one_hot_select_kernel<float><<<grid, TPB, 0, (hipStream_t)stream>>>(seg, V, labels, C, (float*)out);
  return KMH_LAUNCH_CHECK();
}
