// NDHWC helpers around the convolutions: per-(n,c) statistics, GroupNorm / InstanceNorm
// coefficient kernels (forward scale/shift, backward c1/c2/c3 + dgamma/dbeta), the elementwise
// GroupNorm-backward apply (fused with the upstream ReLU mask), MaxPool3d(2) and the decoder's
// nearest-upsample + channel-concat.  All HBM-bound streaming kernels with 16-B accesses.
//   GroupNorm   : keymorph/unet3d/buildingblocks.py:59-78 (F.group_norm, eps 1e-5)
//   InstanceNorm: keymorph/layers.py:165 (affine=False, eps 1e-5)
//   MaxPool3d   : keymorph/unet3d/buildingblocks.py:363, keymorph/layers.py:176
//   upsample+cat: keymorph/unet3d/buildingblocks.py:471-475, 568-582
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int STAT_BLOCKS = 512;   // voxel splits per sample

// ---------------------------------------------------------------------------------------------
// per-(n,c): MODE 0 -> (sum a, sum a*a) ; MODE 1 -> (sum a, sum a*b)
// a, b: (N, V, C).  partial: (N, nblk, C, 2) doubles.
template <int MODE, int VEC>
__global__ __launch_bounds__(TPB) void channel_stats_kernel(const float* __restrict__ a,
                                                            const float* __restrict__ b, long long V, int C,
                                                            double* __restrict__ partial,
                                                            const int* __restrict__ only_if) {
  if (only_if && *only_if == 0) return;        // uniform: device-side gate of a fallback path
  extern __shared__ __attribute__((aligned(16))) double sred[];  // [rows][CQ*VEC*2]
  const int n = blockIdx.y;
  const int CQ = C / VEC;                       // channel groups handled per thread
  const int rows = TPB / CQ;                    // voxel lanes (threads beyond rows*CQ idle)
  const int q = threadIdx.x % CQ, vl = threadIdx.x / CQ;
  const bool active = vl < rows;
  const long long per = (V + gridDim.x - 1) / gridDim.x;
  const long long vbeg = per * blockIdx.x;
  long long vend = vbeg + per;
  if (vend > V) vend = V;
  double d0[VEC], d1[VEC];
  float f0[VEC], f1[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { d0[j] = d1[j] = 0.0; f0[j] = f1[j] = 0.f; }
  if (active) {
    const float* ap = a + (long long)n * V * C + (long long)q * VEC;
    const float* bp = MODE ? b + (long long)n * V * C + (long long)q * VEC : nullptr;
    int cnt = 0;
    for (long long v = vbeg + vl; v < vend; v += rows) {
      float av[VEC], bv[VEC];
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(ap + v * C);
        av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
        if (MODE) {
          const float4 u = *reinterpret_cast<const float4*>(bp + v * C);
          bv[0] = u.x; bv[1] = u.y; bv[2] = u.z; bv[3] = u.w;
        }
      } else {
        av[0] = ap[v * C];
        if (MODE) bv[0] = bp[v * C];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        f0[j] += av[j];
        f1[j] += MODE ? av[j] * bv[j] : av[j] * av[j];
      }
      if (++cnt == 64) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { d0[j] += f0[j]; d1[j] += f1[j]; f0[j] = f1[j] = 0.f; }
        cnt = 0;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { d0[j] += f0[j]; d1[j] += f1[j]; }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      sred[((long long)vl * C + q * VEC + j) * 2] = d0[j];
      sred[((long long)vl * C + q * VEC + j) * 2 + 1] = d1[j];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C * 2; e += TPB) {
    double s = 0;
    for (int r = 0; r < rows; ++r) s += sred[(long long)r * C * 2 + e];
    partial[((long long)n * gridDim.x + blockIdx.x) * C * 2 + e] = s;
  }
}

// stats of nearest-upsampled / concatenated tensors are linear in the per-channel sums of the
// sources, so GroupNorm over cat(skip, up(x)) never needs the concatenated tensor for its statistics.

// forward coefficients: per (n, c) scale = rstd*gamma, shift = beta - mean*rstd*gamma
__global__ void gn_fwd_coeffs_kernel(const double* __restrict__ stats /* (N,C,2) */,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                     int G, double count /* voxels per channel */, float eps,
                                     float* __restrict__ scale, float* __restrict__ shift,
                                     float* __restrict__ mean_rstd /* (N,G,2) */,
                                     float* __restrict__ ascale /* {S, 1/S} | NULL */) {
  const int n = blockIdx.x;
  const int cpg = C / G;
  if (ascale && n == 0 && threadIdx.x == 0) {
    // guaranteed bound on the normalised tensor: |x_hat| < sqrt(m) for m = cpg * count elements per group
    // (sum of squares of the standardised values is <= m), so |gamma x_hat + beta| < max|gamma| sqrt(m) + max|beta|
    float gmax = gamma ? 0.f : 1.f, bmax = 0.f;
    for (int c = 0; c < C; ++c) {
      if (gamma) gmax = fmaxf(gmax, fabsf(gamma[c]));
      if (beta) bmax = fmaxf(bmax, fabsf(beta[c]));
    }
    range_scale(gmax * sqrtf((float)(count * cpg)) + bmax, ascale);
  }
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    double s = 0, ss = 0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += stats[((long long)n * C + c) * 2]; ss += stats[((long long)n * C + c) * 2 + 1]; }
    const double m = count * cpg;
    const double mean = s / m;
    double var = ss / m - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    mean_rstd[((long long)n * G + g) * 2] = (float)mean;
    mean_rstd[((long long)n * G + g) * 2 + 1] = (float)rstd;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
      scale[(long long)n * C + c] = (float)(rstd * ga);
      shift[(long long)n * C + c] = (float)(be - mean * rstd * ga);
    }
  }
}

// backward coefficients.  ab (N,C,2) = (sum dxn, sum dxn*x) ; dx = c1*dxn + c2*x + c3
// dgamma/dbeta (C) are ACCUMULATED (+=) so the same parameter can be used by several samples/calls.
__global__ void gn_bwd_coeffs_kernel(const double* __restrict__ ab, const float* __restrict__ gamma,
                                     const float* __restrict__ mean_rstd, int N, int C, int G, double count,
                                     float* __restrict__ c123 /* (N,C,3) */, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, const int* __restrict__ only_if) {
  if (only_if && *only_if == 0) return;        // device-side gate of a fallback path
  const int cpg = C / G;
  // one block; threads over (n, g)
  for (int ng = threadIdx.x; ng < N * G; ng += blockDim.x) {
    const int n = ng / G, g = ng % G;
    const double mean = mean_rstd[ng * 2], rstd = mean_rstd[ng * 2 + 1];
    double S1 = 0, S2 = 0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const double ga = gamma ? (double)gamma[c] : 1.0;
      const double A = ab[((long long)n * C + c) * 2], B = ab[((long long)n * C + c) * 2 + 1];
      S1 += ga * A;
      S2 += ga * rstd * (B - mean * A);
    }
    const double m = count * cpg;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const double ga = gamma ? (double)gamma[c] : 1.0;
      float* o = c123 + ((long long)n * C + c) * 3;
      o[0] = (float)(rstd * ga);
      o[1] = (float)(-rstd * rstd * S2 / m);
      o[2] = (float)(-rstd * S1 / m + rstd * rstd * S2 * mean / m);
    }
  }
  __syncthreads();
  if (dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const int g = c / cpg;
      double dg = 0, db = 0;
      for (int n = 0; n < N; ++n) {
        const double mean = mean_rstd[(n * G + g) * 2], rstd = mean_rstd[(n * G + g) * 2 + 1];
        const double A = ab[((long long)n * C + c) * 2], B = ab[((long long)n * C + c) * 2 + 1];
        dg += rstd * (B - mean * A);
        db += A;
      }
      dgamma[c] += (float)dg;
      dbeta[c] += (float)db;
    }
  }
}

// The same coefficients from statistics that cost no pass over the tensors:
//   dstats (N,C,2): [.][0] = A = sum_v dxn            (the data-gradient launch's epilogue statistics)
//   bhat   (N,C)  :          sum_v dxn * xhat         (kmh_conv3d_wgrad_bf's fold: sum_{tap,co} W dW_n)
// with xhat = gamma * xt + beta the convolution's input and xt = (x - mean) rstd:  gamma * sum dxn xt = bhat - beta A
// needs no division; only dgamma = sum_n (bhat - beta A) / gamma does.  If some gamma is exactly 0 that quotient does
// not exist: nothing is written, *fallback = 1, and the caller's gated direct path (kmh_channel_stats +
// kmh_gn_bwd_coeffs with only_if = fallback) does the work instead.
__global__ void gn_bwd_coeffs_fold_kernel(const double* __restrict__ dstats, const double* __restrict__ bhat,
                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const float* __restrict__ mean_rstd, int N, int C, int G, double count,
                                          float* __restrict__ c123, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, int* __restrict__ fallback) {
  __shared__ int zero_gamma;
  if (threadIdx.x == 0) zero_gamma = 0;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    if (gamma && gamma[c] == 0.f) zero_gamma = 1;
  __syncthreads();
  if (threadIdx.x == 0) *fallback = zero_gamma;
  if (zero_gamma) return;
  const int cpg = C / G;
  for (int ng = threadIdx.x; ng < N * G; ng += blockDim.x) {
    const int n = ng / G, g = ng % G;
    const double mean = mean_rstd[ng * 2], rstd = mean_rstd[ng * 2 + 1];
    double S1 = 0, S2 = 0;                       // sum_c gamma A ; sum_c gamma sum dxn xt
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
      const double A = dstats[((long long)n * C + c) * 2], Bh = bhat[(long long)n * C + c];
      S1 += ga * A;
      S2 += Bh - be * A;
    }
    const double m = count * cpg;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const double ga = gamma ? (double)gamma[c] : 1.0;
      float* o = c123 + ((long long)n * C + c) * 3;
      o[0] = (float)(rstd * ga);
      o[1] = (float)(-rstd * rstd * S2 / m);
      o[2] = (float)(-rstd * S1 / m + rstd * rstd * S2 * mean / m);
    }
  }
  if (dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
      double dg = 0, db = 0;
      for (int n = 0; n < N; ++n) {
        const double A = dstats[((long long)n * C + c) * 2], Bh = bhat[(long long)n * C + c];
        dg += (Bh - be * A) / ga;
        db += A;
      }
      dgamma[c] += (float)dg;
      dbeta[c] += (float)db;
    }
  }
}

// dx = (c1*dxn + c2*x + c3) * (x > 0 if relu_mask) ; optionally dx += (accumulate into dx)
__global__ __launch_bounds__(TPB) void gn_bwd_apply_kernel(const float* dxn, const float* __restrict__ x,
                                                           const float* __restrict__ c123, long long V, int C,
                                                           int relu_mask, int accumulate, float* dx,
                                                           unsigned* __restrict__ amax /* max |dx| bits | NULL */,
                                                           int out_blocked /* dx is (N, C/8, V, 8); C % 8 == 0 */) {
  const int n = blockIdx.y;
  const long long total = V * C;
  float mx = 0.f;
  const float* cc = c123 + (long long)n * C * 3;
  const long long base = (long long)n * total;
  if ((C & 3) == 0) {
    const long long t4 = total >> 2;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < t4; e += (long long)gridDim.x * TPB) {
      const int c = (int)((e * 4) % C);
      const float4 g = *reinterpret_cast<const float4*>(dxn + base + e * 4);
      const float4 xv = *reinterpret_cast<const float4*>(x + base + e * 4);
      float r[4];
      const float gg[4] = {g.x, g.y, g.z, g.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = cc[(c + j) * 3] * gg[j] + cc[(c + j) * 3 + 1] * xx[j] + cc[(c + j) * 3 + 2];
        if (relu_mask && !(xx[j] > 0.f)) v = 0.f;
        r[j] = v;
      }
      // channel-blocked output: quad (voxel v, channels c..c+3) goes to chunk plane c / 8, record v, half (c % 8) / 4
      float4* o = reinterpret_cast<float4*>(
          dx + base + (out_blocked ? ((long long)(c >> 3) * V + (e * 4) / C) * 8 + (c & 7) : e * 4));
      if (accumulate) { const float4 p = *o; r[0] += p.x; r[1] += p.y; r[2] += p.z; r[3] += p.w; }
      *o = make_float4(r[0], r[1], r[2], r[3]);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r[0]), fabsf(r[1]))), fmaxf(fabsf(r[2]), fabsf(r[3])));
    }
  } else {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
      const int c = (int)(e % C);
      float v = cc[c * 3] * dxn[base + e] + cc[c * 3 + 1] * x[base + e] + cc[c * 3 + 2];
      if (relu_mask && !(x[base + e] > 0.f)) v = 0.f;
      v = accumulate ? dx[base + e] + v : v;
      dx[base + e] = v;
      mx = fmaxf(mx, fabsf(v));
    }
  }
  if (amax) {   // the consumer's f16x3 range scale comes for free with the pass that produces the gradient
    kmh_absmax::publish(mx, amax);
  }
}

// y = act(x*scale[n,c] + shift[n,c])  (InstanceNorm / GroupNorm apply, optionally fused ReLU), in place ok
__global__ __launch_bounds__(TPB) void norm_apply_kernel(const float* x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, long long V, int C,
                                                         int relu, float* y) {
  const int n = blockIdx.y;
  const long long total = V * C, base = (long long)n * total;
  const float* sc = scale + (long long)n * C;
  const float* sh = shift + (long long)n * C;
  if ((C & 3) == 0) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < (total >> 2); e += (long long)gridDim.x * TPB) {
      const int c = (int)((e * 4) % C);
      float4 v = *reinterpret_cast<const float4*>(x + base + e * 4);
      v.x = v.x * sc[c] + sh[c]; v.y = v.y * sc[c + 1] + sh[c + 1];
      v.z = v.z * sc[c + 2] + sh[c + 2]; v.w = v.w * sc[c + 3] + sh[c + 3];
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(y + base + e * 4) = v;
    }
  } else {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
      const int c = (int)(e % C);
      float v = x[base + e] * sc[c] + sh[c];
      y[base + e] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

// relu backward alone: dz = dy * (y > 0)
__global__ __launch_bounds__(TPB) void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        long long n, float* __restrict__ dz) {
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < n; e += (long long)gridDim.x * TPB)
    dz[e] = y[e] > 0.f ? dy[e] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// ConvNet blocks with the normalisation kept LAZY (keymorph/layers.py:137-187, norm_type "instance"): a block's output
// u = MaxPool(ReLU(IN(z))) is never stored -- the next convolution's loader normalises, rectifies and reads the raw z
// (max-pooled raw: IN without affine is increasing, so pooling commutes with it).  Backward of that chain for an
// incoming du (at u's resolution), with zhat = z * scale + shift and g = scatter(du) * [zhat > 0]:
//     dz = rstd (g - mean g - zhat mean(g zhat)) = c1 g + c2 z + c3          (c123 from kmh_gn_bwd_coeffs with G = C)
// g is non-zero at the pooling winners only, so the two sums are taken at u's resolution (in_bwd_stats_kernel) and the
// apply pass reads du, the winners and z once and writes dz once (in_bwd_apply_pool_kernel): the unfused route ran a
// pooling backward, a ReLU mask, a statistics pass and an apply pass over full-resolution tensors.

// per (n, c): (sum g, sum g * z), g = du * [fma(z, scale, shift) > 0]; du, z (N, V, C) at the same resolution
template <int VEC>
__global__ __launch_bounds__(TPB) void in_bwd_stats_kernel(const float* __restrict__ du, const float* __restrict__ z,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           long long V, int C, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double sred[];  // [rows][C*2]
  const int n = blockIdx.y;
  const int CQ = C / VEC;
  const int rows = TPB / CQ;
  const int q = threadIdx.x % CQ, vl = threadIdx.x / CQ;
  const bool active = vl < rows;
  const long long per = (V + gridDim.x - 1) / gridDim.x;
  const long long vbeg = per * blockIdx.x;
  long long vend = vbeg + per;
  if (vend > V) vend = V;
  double d0[VEC], d1[VEC];
  float f0[VEC], f1[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    d0[j] = d1[j] = 0.0; f0[j] = f1[j] = 0.f;
    sc[j] = active ? scale[(long long)n * C + q * VEC + j] : 0.f;
    sh[j] = active ? shift[(long long)n * C + q * VEC + j] : 0.f;
  }
  if (active) {
    const float* ap = du + (long long)n * V * C + (long long)q * VEC;
    const float* bp = z + (long long)n * V * C + (long long)q * VEC;
    int cnt = 0;
    for (long long v = vbeg + vl; v < vend; v += rows) {
      float av[VEC], bv[VEC];
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(ap + v * C);
        const float4 u = *reinterpret_cast<const float4*>(bp + v * C);
        av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
        bv[0] = u.x; bv[1] = u.y; bv[2] = u.z; bv[3] = u.w;
      } else {
        av[0] = ap[v * C];
        bv[0] = bp[v * C];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float g = fmaf(bv[j], sc[j], sh[j]) > 0.f ? av[j] : 0.f;
        f0[j] += g;
        f1[j] += g * bv[j];
      }
      if (++cnt == 64) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { d0[j] += f0[j]; d1[j] += f1[j]; f0[j] = f1[j] = 0.f; }
        cnt = 0;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { d0[j] += f0[j]; d1[j] += f1[j]; }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      sred[((long long)vl * C + q * VEC + j) * 2] = d0[j];
      sred[((long long)vl * C + q * VEC + j) * 2 + 1] = d1[j];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C * 2; e += TPB) {
    double s = 0;
    for (int r = 0; r < rows; ++r) s += sred[(long long)r * C * 2 + e];
    partial[((long long)n * gridDim.x + blockIdx.x) * C * 2 + e] = s;
  }
}

// dz = c1 * du * [fma(z, scale, shift) > 0] + c2 * z + c3   (no pooling between z and u); dz may alias du
__global__ __launch_bounds__(TPB) void in_bwd_apply_kernel(const float* du, const float* __restrict__ z,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ c123, long long V, int C, float* dz,
                                                           unsigned* __restrict__ amax) {
  const int n = blockIdx.y;
  const long long total = V * C, base = (long long)n * total;
  const float* cc = c123 + (long long)n * C * 3;
  const float* sc = scale + (long long)n * C;
  const float* sh = shift + (long long)n * C;
  float mx = 0.f;
  if ((C & 3) == 0) {
    const long long t4 = total >> 2;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < t4; e += (long long)gridDim.x * TPB) {
      const int c = (int)((e * 4) % C);
      const float4 g4 = *reinterpret_cast<const float4*>(du + base + e * 4);
      const float4 z4 = *reinterpret_cast<const float4*>(z + base + e * 4);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w};
      float r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = fmaf(zz[j], sc[c + j], sh[c + j]) > 0.f ? gg[j] : 0.f;
        r[j] = cc[(c + j) * 3] * g + cc[(c + j) * 3 + 1] * zz[j] + cc[(c + j) * 3 + 2];
      }
      *reinterpret_cast<float4*>(dz + base + e * 4) = make_float4(r[0], r[1], r[2], r[3]);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r[0]), fabsf(r[1]))), fmaxf(fabsf(r[2]), fabsf(r[3])));
    }
  } else {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
      const int c = (int)(e % C);
      const float zv = z[base + e];
      const float g = fmaf(zv, sc[c], sh[c]) > 0.f ? du[base + e] : 0.f;
      const float v = cc[c * 3] * g + cc[c * 3 + 1] * zv + cc[c * 3 + 2];
      dz[base + e] = v;
      mx = fmaxf(mx, fabsf(v));
    }
  }
  if (amax) kmh_absmax::publish(mx, amax);
}

// dz (full resolution) = c1 * scatter(du * [zhat(winner) > 0]) + c2 * z + c3: MaxPool3d(2)'s backward (winners from
// kmh_maxpool3d_fwd on the RAW z), the ReLU mask and InstanceNorm's backward in ONE pass; even D, H, W, C % 4 == 0
__global__ __launch_bounds__(TPB) void in_bwd_apply_pool_kernel(const unsigned* __restrict__ argm, const float4* __restrict__ du,
                                                                const float4* __restrict__ z, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ c123,
                                                                float4* __restrict__ dz, int D, int H, int W, int C4, int Do,
                                                                int Ho, int Wo, unsigned* __restrict__ amax) {
  const int n = blockIdx.y;
  const int total = Do * Ho * Wo * C4;
  const long long V = (long long)D * H * W;
  const float4* zn = z + (long long)n * V * C4;
  float4* dzo = dz + (long long)n * V * C4;
  const float* cc = c123 + (long long)n * C4 * 12;
  const float* sc = scale + (long long)n * C4 * 4;
  const float* sh = shift + (long long)n * C4 * 4;
  float mx = 0.f;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < total; e += gridDim.x * TPB) {
    const int c = e % C4, v = e / C4;
    const int xo = v % Wo, r = v / Wo, yo = r % Ho, zo = r / Ho;
    const unsigned a = argm[(long long)n * total + e];
    const float4 g = du[(long long)n * total + e];
    const int aw[4] = {(int)(a & 255), (int)((a >> 8) & 255), (int)((a >> 16) & 255), (int)(a >> 24)};
    const float gg[4] = {g.x, g.y, g.z, g.w};
    float k1[4], k2[4], k3[4], s1[4], s0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      k1[j] = cc[(4 * c + j) * 3]; k2[j] = cc[(4 * c + j) * 3 + 1]; k3[j] = cc[(4 * c + j) * 3 + 2];
      s1[j] = sc[4 * c + j]; s0[j] = sh[4 * c + j];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      const long long o = ((long long)(zz * H + yy) * W + xx) * C4 + c;
      const float4 z4 = zn[o];
      const float zv[4] = {z4.x, z4.y, z4.z, z4.w};
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = (k == aw[j] && fmaf(zv[j], s1[j], s0[j]) > 0.f) ? gg[j] : 0.f;
        t[j] = k1[j] * gj + k2[j] * zv[j] + k3[j];
      }
      dzo[o] = make_float4(t[0], t[1], t[2], t[3]);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
    }
  }
  if (amax) kmh_absmax::publish(mx, amax);
}

// ---------------------------------------------------------------------------------------------
// MaxPool3d(2), floor mode, NDHWC
__global__ __launch_bounds__(TPB) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ arg /* window index | NULL */, int D,
                                                          int H, int W, int C, int Do, int Ho, int Wo) {
  const int n = blockIdx.y;
  const long long total = (long long)Do * Ho * Wo * C;
  const float* xn = x + (long long)n * D * H * W * C;
  float* yn = y + (long long)n * total;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
    const int c = (int)(e % C);
    long long v = e / C;
    const int xo = (int)(v % Wo), yo = (int)((v / Wo) % Ho), zo = (int)(v / ((long long)Wo * Ho));
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      const float val = xn[(((long long)zz * H + yy) * W + xx) * C + c];
      if (val > m || val != val) { m = val; am = k; }
    }
    yn[e] = m;
    if (arg) arg[(long long)n * total + e] = (unsigned char)am;    // same first-max rule as the backward's rescan
  }
}

// dx[child] = dy if child is the first max of its window (scan order z,y,x) else 0; optional accumulate
__global__ __launch_bounds__(TPB) void maxpool_bwd_kernel(const float* __restrict__ x,
                                                          const unsigned char* __restrict__ argm /* from the forward | NULL */,
                                                          const float* __restrict__ dy, const float* add, int acs,
                                                          float* dx, int D, int H, int W, int C, int Do, int Ho, int Wo,
                                                          int out_blocked /* dx is (N, C/8, D, H, W, 8) */) {
  // add (may alias dx): a second gradient of the pooled tensor's source (the skip connection), channel stride acs
  const int n = blockIdx.y;
  const long long total = (long long)Do * Ho * Wo * C;
  const float* xn = x ? x + (long long)n * D * H * W * C : nullptr;
  float* dxn = dx + (long long)n * D * H * W * C;
  const float* an = add ? add + (long long)n * D * H * W * acs : nullptr;
  const float* dyn = dy + (long long)n * total;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
    const int c = (int)(e % C);
    long long v = e / C;
    const int xo = (int)(v % Wo), yo = (int)((v / Wo) % Ho), zo = (int)(v / ((long long)Wo * Ho));
    int arg = 0;
    if (argm) {                       // the forward recorded the winner: x (a full-resolution tensor) is not re-read
      arg = argm[(long long)n * total + e];
    } else {
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
        const float val = xn[(((long long)zz * H + yy) * W + xx) * C + c];
        if (val > m || val != val) { m = val; arg = k; }
      }
    }
    const float g = dyn[e];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      const long long vox = ((long long)zz * H + yy) * W + xx;
      const float val = (k == arg) ? g : 0.f;
      const long long o = out_blocked ? ((long long)(c >> 3) * D * H * W + vox) * 8 + (c & 7) : vox * C + c;
      dxn[o] = an ? an[vox * acs + c] + val : val;
    }
  }
}

// 16-byte versions (C % 4 == 0, fewer than 2^31 pooled elements per sample): one thread owns 4 channels of a pooled
// voxel; a wave reads / writes whole runs of consecutive full-resolution voxels (both x children per thread).  Same
// first-max rule and the same results as the scalar kernels, a quarter of the memory instructions, 32-bit indices.
__device__ __forceinline__ void first_max(float v, int k, float& m, int& am) {
  if (v > m || v != v) { m = v; am = k; }
}
__global__ __launch_bounds__(TPB) void maxpool_fwd4_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                           unsigned* __restrict__ arg /* 4 window indices | NULL */, int D,
                                                           int H, int W, int C4, int Do, int Ho, int Wo) {
  const int n = blockIdx.y;
  const int total = Do * Ho * Wo * C4;
  const float4* xn = x + (long long)n * D * H * W * C4;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < total; e += gridDim.x * TPB) {
    const int c = e % C4, v = e / C4;
    const int xo = v % Wo, r = v / Wo, yo = r % Ho, zo = r / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      const float4 val = xn[((long long)(zz * H + yy) * W + xx) * C4 + c];
      first_max(val.x, k, m.x, a0); first_max(val.y, k, m.y, a1); first_max(val.z, k, m.z, a2); first_max(val.w, k, m.w, a3);
    }
    y[(long long)n * total + e] = m;
    if (arg) arg[(long long)n * total + e] = (unsigned)a0 | ((unsigned)a1 << 8) | ((unsigned)a2 << 16) | ((unsigned)a3 << 24);
  }
}

template <bool BLOCKED>
__global__ __launch_bounds__(TPB) void maxpool_bwd4_kernel(const unsigned* __restrict__ argm, const float4* __restrict__ dy,
                                                           const float* add, int acs, float* dx, int D, int H, int W,
                                                           int C4, int Do, int Ho, int Wo) {
  const int n = blockIdx.y;
  const int total = Do * Ho * Wo * C4;
  const long long V = (long long)D * H * W;
  float* dxn = dx + (long long)n * V * C4 * 4;
  const float* an = add ? add + (long long)n * V * acs : nullptr;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < total; e += gridDim.x * TPB) {
    const int c = e % C4, v = e / C4;
    const int xo = v % Wo, r = v / Wo, yo = r % Ho, zo = r / Ho;
    const unsigned a = argm[(long long)n * total + e];
    const float4 g = dy[(long long)n * total + e];
    const int a0 = a & 255, a1 = (a >> 8) & 255, a2 = (a >> 16) & 255, a3 = a >> 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      const long long vox = (long long)(zz * H + yy) * W + xx;
      float4 val = make_float4(k == a0 ? g.x : 0.f, k == a1 ? g.y : 0.f, k == a2 ? g.z : 0.f, k == a3 ? g.w : 0.f);
      if (an) {
        const float4 s4 = *reinterpret_cast<const float4*>(an + vox * acs + 4 * c);
        val.x += s4.x; val.y += s4.y; val.z += s4.z; val.w += s4.w;
      }
      const long long o = BLOCKED ? ((long long)(c >> 1) * V + vox) * 8 + 4 * (c & 1) : (vox * C4 + c) * 4;
      *reinterpret_cast<float4*>(dxn + o) = val;     // (non-temporal stores measured 2x slower here)
    }
  }
}

// Channel-blocked output without a second gradient (the 256^3 level: 8.6 GB written): lane = one 16-byte piece of an output
// row in memory order (piece p = 4 xo + 2 dx + quad-in-chunk), so every store instruction writes 1 KB of CONTIGUOUS output;
// the kernel above hands the same bytes over in 32-byte pieces on four planes (2.6 ms = 3.3 TB/s).  Each lane loads its
// (pooled voxel, quad) winners + gradient (shared with the dx twin through L1) and writes the four (dz, dy) children.
__global__ __launch_bounds__(TPB) void maxpool_bwd4_rows_kernel(const unsigned* __restrict__ argm, const float4* __restrict__ dy,
                                                                float* __restrict__ dx, int D, int H, int W, int C4, int Do,
                                                                int Ho, int Wo) {
  const int n = blockIdx.y;
  const long long V = (long long)D * H * W;
  const long long pooled = (long long)Do * Ho * Wo * C4;
  float* dxn = dx + (long long)n * V * C4 * 4;
  const int ppr = Wo * 4;                                   // pieces per pooled row
  const long long total = (long long)(C4 >> 1) * Do * Ho * ppr;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
    const int p = (int)(e % ppr);
    const long long r = e / ppr;
    const int yo = (int)(r % Ho), zo = (int)((r / Ho) % Do), chunk = (int)(r / ((long long)Ho * Do));
    const int xo = p >> 2, dxx = (p >> 1) & 1, qh = p & 1;
    const long long idx = (((long long)zo * Ho + yo) * Wo + xo) * C4 + 2 * chunk + qh;
    const unsigned a = argm[(long long)n * pooled + idx];
    const float4 g = dy[(long long)n * pooled + idx];
    const int a0 = a & 255, a1 = (a >> 8) & 255, a2 = (a >> 16) & 255, a3 = a >> 24;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 2 * kk + dxx;                            // window index (dz, dy, dx)
      const long long vox = ((long long)(2 * zo + (kk >> 1)) * H + 2 * yo + (kk & 1)) * W + 2 * xo + dxx;
      const float4 val = make_float4(k == a0 ? g.x : 0.f, k == a1 ? g.y : 0.f, k == a2 ? g.z : 0.f, k == a3 ? g.w : 0.f);
      *reinterpret_cast<float4*>(dxn + ((long long)chunk * V + vox) * 8 + 4 * qh) = val;
    }
  }
}

// The same scatter written ALREADY SPLIT for the 16-bit matrix cores (round 5): the consumer -- the data-gradient convolution of
// the block in front of the pooling, conv3_fwd_s_kernel<1, true, true> -- then copies fragments instead of converting them.
// A scatter moves values, so the range scale S of the pooled gradient (known before this pass, carried with the tensor) is the
// scale of its output: every 32-byte record (one voxel, one 8-channel chunk) holds the 8 fp16 "hi" then the 8 fp16 "lo" terms of
// fmaf(value, S, 0) -- exactly what the consumer's own conversion computes from the fp32 tensor, so results are bit-identical --
// and record V of every (sample, chunk) plane is zero (the source of the consumer's padding slots).  Same bytes, same store
// pattern as the kernel above (lane = one 16-byte piece in memory order; the two lanes of a record compute both halves and
// store one each).
__global__ __launch_bounds__(TPB) void maxpool_bwd4_split_kernel(const unsigned* __restrict__ argm, const float4* __restrict__ dy,
                                                                 const float* __restrict__ scale2, float* __restrict__ dx, int D,
                                                                 int H, int W, int C4, int Do, int Ho, int Wo) {
  const int n = blockIdx.y;
  const long long V = (long long)D * H * W;
  const long long pooled = (long long)Do * Ho * Wo * C4;
  const int nchunk = C4 >> 1;
  float* dxn = dx + (long long)n * nchunk * (V + 1) * 8;
  const float S = scale2[0];
  if (blockIdx.x == 0) {                                    // the zero records
    for (int e = threadIdx.x; e < nchunk * 2; e += TPB)
      *reinterpret_cast<float4*>(dxn + ((long long)(e >> 1) * (V + 1) + V) * 8 + 4 * (e & 1)) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int ppr = Wo * 4;                                   // pieces per pooled row
  const long long total = (long long)nchunk * Do * Ho * ppr;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
    const int p = (int)(e % ppr);
    const long long r = e / ppr;
    const int yo = (int)(r % Ho), zo = (int)((r / Ho) % Do), chunk = (int)(r / ((long long)Ho * Do));
    const int xo = p >> 2, dxx = (p >> 1) & 1, half = p & 1;
    const long long idx = (((long long)zo * Ho + yo) * Wo + xo) * C4 + 2 * chunk;
    const uint2 a2 = *reinterpret_cast<const uint2*>(argm + (long long)n * pooled + idx);
    const float4 g0 = dy[(long long)n * pooled + idx], g1 = dy[(long long)n * pooled + idx + 1];
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    int a[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = (a2.x >> (8 * j)) & 255; a[4 + j] = (a2.y >> (8 * j)) & 255; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 2 * kk + dxx;                            // window index (dz, dy, dx)
      const long long vox = ((long long)(2 * zo + (kk >> 1)) * H + 2 * yo + (kk & 1)) * W + 2 * xo + dxx;
      float val[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = fmaf(k == a[j] ? g[j] : 0.f, S, 0.f);
      kmh_bf16x8 parts[2];
      split8<2>(val, parts);
      *reinterpret_cast<kmh_bf16x8*>(dxn + ((long long)chunk * (V + 1) + vox) * 8 + 4 * half) = half ? parts[1] : parts[0];
    }
  }
}

// Pooling backward + skip gradient, with the skip gradient's GroupNorm backward still pending (round 3): the decoder's
// fused upsample + concat + conv operator returns its normalised-input gradient dxn for the skip half untouched, and this
// kernel forms  dx = scatter(dy) + [x > 0] (c1 dxn + c2 x + c3)  in one pass over the encoder output x -- the separate
// kmh_gn_bwd_apply pass (read dxn, read x, write dskip) and this kernel's read of dskip are gone.  It also publishes
// max |dx|, the next backward convolution's f16x3 range scale (exact, where the two-pass route carried a sum bound).
__global__ __launch_bounds__(TPB) void maxpool_bwd4_lazy_kernel(const unsigned* __restrict__ argm, const float4* __restrict__ dy,
                                                                const float4* __restrict__ dxn, const float4* __restrict__ x,
                                                                const float* __restrict__ c123, float4* __restrict__ dx,
                                                                int D, int H, int W, int C4, int Do, int Ho, int Wo,
                                                                unsigned* __restrict__ amax) {
  const int n = blockIdx.y;
  const int total = Do * Ho * Wo * C4;
  const long long V = (long long)D * H * W;
  const float4* gn = dxn + (long long)n * V * C4;
  const float4* xn = x + (long long)n * V * C4;
  float4* dxo = dx + (long long)n * V * C4;
  const float* cc = c123 + (long long)n * C4 * 12;
  float mx = 0.f;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < total; e += gridDim.x * TPB) {
    const int c = e % C4, v = e / C4;
    const int xo = v % Wo, r = v / Wo, yo = r % Ho, zo = r / Ho;
    const unsigned a = argm[(long long)n * total + e];
    const float4 g = dy[(long long)n * total + e];
    const int a0 = a & 255, a1 = (a >> 8) & 255, a2 = (a >> 16) & 255, a3 = a >> 24;
    float k1[4], k2[4], k3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { k1[j] = cc[(4 * c + j) * 3]; k2[j] = cc[(4 * c + j) * 3 + 1]; k3[j] = cc[(4 * c + j) * 3 + 2]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      const long long o = ((long long)(zz * H + yy) * W + xx) * C4 + c;
      const float4 d4 = gn[o], x4 = xn[o];
      const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
      float s4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {       // the same expression as gn_bwd_apply_kernel
        float t = k1[j] * dd[j] + k2[j] * xv[j] + k3[j];
        if (!(xv[j] > 0.f)) t = 0.f;
        s4[j] = t;
      }
      float4 val = make_float4(k == a0 ? g.x : 0.f, k == a1 ? g.y : 0.f, k == a2 ? g.z : 0.f, k == a3 ? g.w : 0.f);
      val.x += s4[0]; val.y += s4[1]; val.z += s4[2]; val.w += s4[3];
      dxo[o] = val;
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(val.x), fabsf(val.y))), fmaxf(fabsf(val.z), fabsf(val.w)));
    }
  }
  if (amax) kmh_absmax::publish(mx, amax);
}

// ---------------------------------------------------------------------------------------------
// out[n, z,y,x, :] = cat(skip[n,z,y,x,:Cs], low[n, nearest(z,y,x), :Cl])
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
  // ATen legacy 'nearest': floor(dst * (float)in/out), clamped
  const float scale = (float)in_size / (float)out_size;
  int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

__global__ __launch_bounds__(TPB) void upcat_fwd_kernel(const float* __restrict__ skip, const float* __restrict__ low,
                                                        float* __restrict__ out, int D, int H, int W, int Cs, int Dl,
                                                        int Hl, int Wl, int Cl) {
  const int n = blockIdx.y;
  const int C = Cs + Cl;
  const long long total = (long long)D * H * W * C;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
    const int c = (int)(e % C);
    const long long v = e / C;
    float val;
    if (c < Cs) {
      val = skip[((long long)n * D * H * W + v) * Cs + c];
    } else {
      const int xx = (int)(v % W), yy = (int)((v / W) % H), zz = (int)(v / ((long long)W * H));
      const int xs = nearest_src(xx, Wl, W), ys = nearest_src(yy, Hl, H), zs = nearest_src(zz, Dl, D);
      val = low[((((long long)n * Dl + zs) * Hl + ys) * Wl + xs) * Cl + (c - Cs)];
    }
    out[(long long)n * total + e] = val;
  }
}

// 16-byte version of the same (Cs % 4 == 0 and Cl % 4 == 0): one thread per (voxel, channel quad), 32-bit index math
__global__ __launch_bounds__(TPB) void upcat_fwd4_kernel(const float4* __restrict__ skip, const float4* __restrict__ low,
                                                         float4* __restrict__ out, int D, int H, int W, int Cs4, int Dl,
                                                         int Hl, int Wl, int Cl4) {
  const int n = blockIdx.y;
  const unsigned C4 = Cs4 + Cl4;
  const unsigned V = (unsigned)D * H * W;
  const unsigned long long total = (unsigned long long)V * C4;
  for (unsigned long long e = (unsigned long long)blockIdx.x * TPB + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * TPB) {
    const unsigned v = (unsigned)(e / C4), c = (unsigned)(e - (unsigned long long)v * C4);
    float4 val;
    if (c < (unsigned)Cs4) {
      val = skip[((unsigned long long)n * V + v) * Cs4 + c];
    } else {
      const unsigned xx = v % W, t = v / W, yy = t % H, zz = t / H;
      const int xs = nearest_src(xx, Wl, W), ys = nearest_src(yy, Hl, H), zs = nearest_src(zz, Dl, D);
      val = low[((((unsigned long long)n * Dl + zs) * Hl + ys) * Wl + xs) * Cl4 + (c - Cs4)];
    }
    out[(unsigned long long)n * total + e] = val;
  }
}

// exact 2x upsampling, 16-byte version: dlow = sum of the 8 fine voxels of dout[..., Cs:]
__global__ __launch_bounds__(TPB) void upcat_bwd_low2x4_kernel(const float4* __restrict__ dout, float4* __restrict__ dlow,
                                                               int H, int W, int Cs4, int Dl, int Hl, int Wl, int Cl4) {
  const int n = blockIdx.y;
  const unsigned C4 = Cs4 + Cl4;
  const unsigned Vl = (unsigned)Dl * Hl * Wl;
  const unsigned long long total = (unsigned long long)Vl * Cl4;
  const float4* dn = dout + (unsigned long long)n * (8ull * Vl) * C4 + Cs4;
  for (unsigned long long e = (unsigned long long)blockIdx.x * TPB + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * TPB) {
    const unsigned v = (unsigned)(e / Cl4), c = (unsigned)(e - (unsigned long long)v * Cl4);
    const unsigned xs = v % Wl, t = v / Wl, ys = t % Hl, zs = t / Hl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {      // same summation order as the generic kernel: z, then y, then x
      const unsigned zz = 2 * zs + (k >> 2), yy = 2 * ys + ((k >> 1) & 1), xx = 2 * xs + (k & 1);
      const float4 g = dn[(((unsigned long long)zz * H + yy) * W + xx) * C4 + c];
      s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
    }
    dlow[(unsigned long long)n * total + e] = s;
  }
}

// dskip (+)= dout[..., :Cs]
__global__ __launch_bounds__(TPB) void upcat_bwd_skip_kernel(const float* __restrict__ dout, float* __restrict__ dskip,
                                                             long long V, int Cs, int C, int accumulate) {
  const int n = blockIdx.y;
  const long long total = V * Cs;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
    const int c = (int)(e % Cs);
    const long long v = e / Cs;
    const float g = dout[((long long)n * V + v) * C + c];
    float* o = dskip + (long long)n * total + e;
    *o = accumulate ? *o + g : g;
  }
}

// dlow[n, zs,ys,xs, c] = sum over the destination voxels that map to (zs,ys,xs) of dout[..., Cs + c]
__global__ __launch_bounds__(TPB) void upcat_bwd_low_kernel(const float* __restrict__ dout, float* __restrict__ dlow,
                                                            int D, int H, int W, int Cs, int Dl, int Hl, int Wl,
                                                            int Cl) {
  const int n = blockIdx.y;
  const int C = Cs + Cl;
  const long long total = (long long)Dl * Hl * Wl * Cl;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
    const int c = (int)(e % Cl);
    const long long v = e / Cl;
    const int xs = (int)(v % Wl), ys = (int)((v / Wl) % Hl), zs = (int)(v / ((long long)Wl * Hl));
    // candidate destination ranges (conservative), filtered by the exact forward mapping
    const int z_lo = (int)((long long)zs * D / Dl) - 1, z_hi = (int)(((long long)zs + 1) * D / Dl) + 1;
    const int y_lo = (int)((long long)ys * H / Hl) - 1, y_hi = (int)(((long long)ys + 1) * H / Hl) + 1;
    const int x_lo = (int)((long long)xs * W / Wl) - 1, x_hi = (int)(((long long)xs + 1) * W / Wl) + 1;
    float s = 0.f;
    for (int zz = z_lo < 0 ? 0 : z_lo; zz <= z_hi && zz < D; ++zz) {
      if (nearest_src(zz, Dl, D) != zs) continue;
      for (int yy = y_lo < 0 ? 0 : y_lo; yy <= y_hi && yy < H; ++yy) {
        if (nearest_src(yy, Hl, H) != ys) continue;
        for (int xx = x_lo < 0 ? 0 : x_lo; xx <= x_hi && xx < W; ++xx) {
          if (nearest_src(xx, Wl, W) != xs) continue;
          s += dout[((((long long)n * D + zz) * H + yy) * W + xx) * C + Cs + c];
        }
      }
    }
    dlow[(long long)n * total + e] = s;
  }
}

// NCDHW <-> NDHWC (boundary conversions for C > 1)
__global__ __launch_bounds__(TPB) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           long long V, int C, int reverse) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const long long v0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = in + (long long)n * V * C;
  float* dst = out + (long long)n * V * C;
  if (!reverse) {  // in: (C, V) -> out: (V, C)
    for (int j = ty; j < 32; j += 8) {
      const int c = c0 + j;
      const long long v = v0 + tx;
      tile[j][tx] = (c < C && v < V) ? src[(long long)c * V + v] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
      const long long v = v0 + j;
      const int c = c0 + tx;
      if (c < C && v < V) dst[v * C + c] = tile[tx][j];
    }
  } else {  // in: (V, C) -> out: (C, V)
    for (int j = ty; j < 32; j += 8) {
      const long long v = v0 + j;
      const int c = c0 + tx;
      tile[j][tx] = (c < C && v < V) ? src[v * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
      const int c = c0 + j;
      const long long v = v0 + tx;
      if (c < C && v < V) dst[(long long)c * V + v] = tile[tx][j];
    }
  }
}

static inline int stream_blocks(long long elems) {
  int nb = ceil_div(elems, (long long)TPB * 8);
  if (nb > 4096) nb = 4096;
  return nb < 1 ? 1 : nb;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
KMH_API size_t kmh_channel_stats_ws_bytes(int N, int C) {
  return (size_t)N * STAT_BLOCKS * C * 2 * sizeof(double);
}

// mode 0: out (N,C,2) doubles = (sum a, sum a^2); mode 1: (sum a, sum a*b)
KMH_API int kmh_channel_stats(const float* a, const float* b, int mode, int N, long long V, int C, double* out,
                              void* ws, const int* only_if, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C > TPB * 4) return -22;
  int nblk = ceil_div(V, 2048);
  if (nblk > STAT_BLOCKS) nblk = STAT_BLOCKS;
  if (nblk < 1) nblk = 1;
  const bool v4 = (C % 4 == 0) && (C / 4 <= TPB);
  const int CQ = v4 ? C / 4 : C;
  if (CQ > TPB) return -22;
  const int rows = TPB / CQ;
  const size_t lds = (size_t)rows * C * 2 * sizeof(double);
  if (lds > 64 * 1024) return -22;
  dim3 g(nblk, N);
  if (v4) {
    if (mode == 0) channel_stats_kernel<0, 4><<<g, TPB, lds, s>>>(a, b, V, C, (double*)ws, only_if);
    else channel_stats_kernel<1, 4><<<g, TPB, lds, s>>>(a, b, V, C, (double*)ws, only_if);
  } else {
    if (mode == 0) channel_stats_kernel<0, 1><<<g, TPB, lds, s>>>(a, b, V, C, (double*)ws, only_if);
    else channel_stats_kernel<1, 1><<<g, TPB, lds, s>>>(a, b, V, C, (double*)ws, only_if);
  }
  kmh_stats::final_kernel<<<dim3(ceil_div(C * 2, 256 / kWave), N), 256, 0, s>>>((const double*)ws, nblk, C, out, only_if);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_gn_fwd_coeffs(const double* stats, const float* gamma, const float* beta, int N, int C, int G,
                              double count, float eps, float* scale, float* shift, float* mean_rstd,
                              float* ascale, void* stream) {
  if (G <= 0 || C % G) return -22;
  gn_fwd_coeffs_kernel<<<N, 64, 0, (hipStream_t)stream>>>(stats, gamma, beta, C, G, count, eps, scale, shift,
                                                         mean_rstd, ascale);
  return KMH_LAUNCH_CHECK();
}

/* out2[2] = {S, 1/S}: S = 2^k with max(max|x|, min_abs) * S in (2^14, 2^15]  (S = 1 for an all-zero or non-finite
 * tensor).  Range scale of one operand tensor for the fp16-split ("f16x3") convolutions; everything stays on the
 * device and on `stream`. */
KMH_API int kmh_absmax_scale(const float* x, long long n, float min_abs, float* out2, void* stream) {
  return kmh_absmax::launch(x, n, min_abs, out2, (hipStream_t)stream);
}

KMH_API int kmh_gn_bwd_coeffs(const double* ab, const float* gamma, const float* mean_rstd, int N, int C, int G,
                              double count, float* c123, float* dgamma, float* dbeta, const int* only_if, void* stream) {
  if (G <= 0 || C % G) return -22;
  gn_bwd_coeffs_kernel<<<1, 256, 0, (hipStream_t)stream>>>(ab, gamma, mean_rstd, N, C, G, count, c123, dgamma,
                                                          dbeta, only_if);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_gn_bwd_coeffs_fold(const double* dstats, const double* bhat, const float* gamma, const float* beta,
                                   const float* mean_rstd, int N, int C, int G, double count, float* c123,
                                   float* dgamma, float* dbeta, int* fallback, void* stream) {
  if (G <= 0 || C % G || !fallback) return -22;
  gn_bwd_coeffs_fold_kernel<<<1, 256, 0, (hipStream_t)stream>>>(dstats, bhat, gamma, beta, mean_rstd, N, C, G, count,
                                                               c123, dgamma, dbeta, fallback);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_gn_bwd_apply(const float* dxn, const float* x, const float* c123, int N, long long V, int C,
                             int relu_mask, int accumulate, float* dx, float* dx_scale2, int out_blocked, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (out_blocked && ((C & 7) || dx == dxn || dx == x)) return -22;      // another layout: not in place
  gn_bwd_apply_kernel<<<dim3(stream_blocks(V * C / 4), N), TPB, 0, s>>>(dxn, x, c123, V, C, relu_mask, accumulate, dx,
                                                                      reinterpret_cast<unsigned*>(dx_scale2), out_blocked);
  if (dx_scale2) kmh_absmax::final_kernel<<<1, 1, 0, s>>>(dx_scale2, 0.f);
  return KMH_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------
// Box sums for the weight gradient of a 3x3x3 convolution over a nearest-x2 upsampled tensor: with x_up[v] = x_low[v / 2],
//   dW[tap][ci][co] = sum_v x_up[v + tap][ci] dz[v][co] = sum_m x_low[m][ci] * G[m][tap][co],
//   G[m][tap][co]   = sum of dz over the 2 x 2 x 2 voxels v with (v + tap) / 2 == m, i.e. per axis v in {2m - tap, 2m + 1 - tap}
// so the correlation becomes ONE plain matrix product over the low-resolution voxels (1/8 of the multiply-adds).  This
// kernel forms G (N, V_low, 27, Cout) from dz (N, 2Dl, 2Hl, 2Wl, Cout): thread = (low voxel, channel quad), the 4 x 4 x 4
// window is streamed once and reduced separably (x pairs, then y pairs, then z pairs).
// address of channels [c, c + 4) of voxel `vox` in one sample of dz: (D,H,W,C), or channel-blocked (C/8, D,H,W, 8)
__device__ __forceinline__ const float* dz_quad(const float* dn, long long vox, int c, int Cout, long long V, int in_blocked) {
  return in_blocked ? dn + ((long long)(c >> 3) * V + vox) * 8 + (c & 7) : dn + vox * Cout + c;
}

__global__ __launch_bounds__(256, 3) void up2_boxsum_kernel(const float* __restrict__ dz, float* __restrict__ G, int Dl,
                                                            int Hl, int Wl, int Cout, int in_blocked) {
  // thread = (low voxel, z tap, channel quad): 2 of the window's 4 planes, 9 outputs -- ~100 registers instead of 256
  const int n = blockIdx.y;
  const int cq = Cout >> 2;
  const long long total = (long long)Dl * Hl * Wl * 3 * cq;
  const int D = 2 * Dl, H = 2 * Hl, W = 2 * Wl;
  const float* dn = dz + (long long)n * D * H * W * Cout;
  float* gn = G + (long long)n * Dl * Hl * Wl * 27 * Cout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int q = (int)(e % cq);
    const int kz = (int)((e / cq) % 3);
    const long long m = e / (3 * cq);
    const int mx = (int)(m % Wl), my = (int)((m / Wl) % Hl), mz = (int)(m / ((long long)Wl * Hl));
    float4 Y[3][3];
#pragma unroll
    for (int a = 0; a < 9; ++a) (&Y[0][0])[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    // window index i = u - (2m - 1) in 0..3 per axis; tap k (offset k - 1) sums i in {2 - k, 3 - k}
#pragma unroll
    for (int dzp = 0; dzp < 2; ++dzp) {
      const int uz = 2 * mz - 1 + (2 - kz) + dzp;
#pragma unroll
      for (int iy = 0; iy < 4; ++iy) {
        const int uy = 2 * my - 1 + iy;
        float4 a4[4];
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) {
          const int ux = 2 * mx - 1 + ix;
          a4[ix] = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((unsigned)ux < (unsigned)W && (unsigned)uy < (unsigned)H && (unsigned)uz < (unsigned)D)
            a4[ix] = *reinterpret_cast<const float4*>(dz_quad(dn, ((long long)uz * H + uy) * W + ux, 4 * q, Cout,
                                                              (long long)D * H * W, in_blocked));
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float4 u = a4[2 - kx], v = a4[3 - kx];
          const float4 xs = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
            if (iy == 2 - ky || iy == 3 - ky) {
              Y[ky][kx].x += xs.x; Y[ky][kx].y += xs.y; Y[ky][kx].z += xs.z; Y[ky][kx].w += xs.w;
            }
        }
      }
    }
    float* o = gn + (m * 27 + kz * 9) * Cout + 4 * q;
#pragma unroll
    for (int a = 0; a < 9; ++a) *reinterpret_cast<float4*>(o + (long long)a * Cout) = (&Y[0][0])[a];
  }
}

// LDS-tiled variant (the default): workgroup = a 4 x 4 x 2 tile of low voxels x 32 channels.  The kernel above reads every dz
// value ~12 times through L1 (26 GB of loads for a 2.1 GB tensor at the 128^3 decoder level); here the tile's 10 x 10 x 6 window is
// loaded once into LDS (one plane per channel quad, plane stride = 16 B mod 256 B so that the quads of a voxel fall on different
// banks) and the SAME additions in the SAME order give bit-identical box sums.
constexpr int BT_X = 4, BT_Y = 4, BT_Z = 2, BT_CQ = 8;
constexpr int BT_HX = 2 * BT_X + 2, BT_HY = 2 * BT_Y + 2, BT_HZ = 2 * BT_Z + 2, BT_VOX = BT_HX * BT_HY * BT_HZ;   // 600
constexpr int BT_PLANE = BT_VOX + 1;
constexpr int BT_TPB = BT_X * BT_Y * BT_Z * 3 * BT_CQ;                                                             // 768
__global__ __launch_bounds__(BT_TPB) void up2_boxsum_tiled_kernel(const float* __restrict__ dz, float* __restrict__ G, int Dl,
                                                                  int Hl, int Wl, int Cout, int tiles_x, int tiles_y,
                                                                  int in_blocked) {
  __shared__ float4 sx[BT_CQ * BT_PLANE];
  const int n = blockIdx.z, chunk = blockIdx.y, tid = threadIdx.x;
  const int bx = blockIdx.x % tiles_x, by = (blockIdx.x / tiles_x) % tiles_y, bz = blockIdx.x / (tiles_x * tiles_y);
  const int x0 = bx * BT_X, y0 = by * BT_Y, z0 = bz * BT_Z;
  const int D = 2 * Dl, H = 2 * Hl, W = 2 * Wl;
  const float* dn = dz + (long long)n * D * H * W * Cout;
  for (int e = tid; e < BT_VOX * BT_CQ; e += BT_TPB) {
    const int q = e & (BT_CQ - 1), v = e / BT_CQ;
    const int lx = v % BT_HX, ly = (v / BT_HX) % BT_HY, lz = v / (BT_HX * BT_HY);
    const int ux = 2 * x0 - 1 + lx, uy = 2 * y0 - 1 + ly, uz = 2 * z0 - 1 + lz, c = (chunk * BT_CQ + q) * 4;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)ux < (unsigned)W && (unsigned)uy < (unsigned)H && (unsigned)uz < (unsigned)D && c < Cout)
      val = *reinterpret_cast<const float4*>(dz_quad(dn, ((long long)uz * H + uy) * W + ux, c, Cout, (long long)D * H * W,
                                                     in_blocked));
    sx[q * BT_PLANE + v] = val;
  }
  __syncthreads();
  const int q = tid & (BT_CQ - 1), kz = (tid / BT_CQ) % 3, m = tid / (3 * BT_CQ);
  const int lmx = m % BT_X, lmy = (m / BT_X) % BT_Y, lmz = m / (BT_X * BT_Y);
  const int mx = x0 + lmx, my = y0 + lmy, mz = z0 + lmz, c = (chunk * BT_CQ + q) * 4;
  if (mx >= Wl || my >= Hl || mz >= Dl || c >= Cout) return;
  const float4* sq = sx + q * BT_PLANE;
  float4 Y[3][3];
#pragma unroll
  for (int a = 0; a < 9; ++a) (&Y[0][0])[a] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int dzp = 0; dzp < 2; ++dzp) {
    const int lz = 2 * lmz + (2 - kz) + dzp;
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {
      const float4* row = sq + (lz * BT_HY + 2 * lmy + iy) * BT_HX + 2 * lmx;
      const float4 a4[4] = {row[0], row[1], row[2], row[3]};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 u = a4[2 - kx], v = a4[3 - kx];
        const float4 xs = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
          if (iy == 2 - ky || iy == 3 - ky) {
            Y[ky][kx].x += xs.x; Y[ky][kx].y += xs.y; Y[ky][kx].z += xs.z; Y[ky][kx].w += xs.w;
          }
      }
    }
  }
  const long long mg = ((long long)mz * Hl + my) * Wl + mx;
  float* o = G + ((long long)n * Dl * Hl * Wl * 27 + mg * 27 + kz * 9) * Cout + c;
#pragma unroll
  for (int a = 0; a < 9; ++a) *reinterpret_cast<float4*>(o + (long long)a * Cout) = (&Y[0][0])[a];
}

/* G (N, Dl*Hl*Wl, 27, Cout) from dz (N, 2Dl, 2Hl, 2Wl, Cout), Cout % 4 == 0 (see the kernel comment): the weight gradient
 * of a 3x3x3 convolution with respect to nearest-x2 upsampled input channels is then x_low^T (Cl x V_low) times G
 * (V_low x 27 Cout) per sample -- one plain matrix product (the host uses the library GEMM). */
KMH_API int kmh_up2_boxsum(const float* dz, float* G, int N, int Dl, int Hl, int Wl, int Cout, int in_blocked, void* stream) {
  if ((Cout & 3) || (in_blocked && (Cout & 7))) return -22;
  static const bool plain = getenv("KEYMORPH_BOXSUM_PLAIN") != nullptr;       // A/B runs: the untiled kernel
  if (!plain) {
    const int tx = (Wl + BT_X - 1) / BT_X, ty = (Hl + BT_Y - 1) / BT_Y, tz = (Dl + BT_Z - 1) / BT_Z;
    const int chunks = (Cout + 4 * BT_CQ - 1) / (4 * BT_CQ);
    if ((long long)tx * ty * tz < (1ll << 31) && chunks < 65536 && N < 65536) {
      up2_boxsum_tiled_kernel<<<dim3(tx * ty * tz, chunks, N), BT_TPB, 0, (hipStream_t)stream>>>(dz, G, Dl, Hl, Wl, Cout, tx, ty, in_blocked);
      return KMH_LAUNCH_CHECK();
    }
  }
  const long long total = (long long)Dl * Hl * Wl * 3 * (Cout / 4);
  up2_boxsum_kernel<<<dim3(stream_blocks(total), N), 256, 0, (hipStream_t)stream>>>(dz, G, Dl, Hl, Wl, Cout, in_blocked);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_norm_apply(const float* x, const float* scale, const float* shift, int N, long long V, int C,
                           int relu, float* y, void* stream) {
  norm_apply_kernel<<<dim3(stream_blocks(V * C / 4), N), TPB, 0, (hipStream_t)stream>>>(x, scale, shift, V, C, relu,
                                                                                       y);
  return KMH_LAUNCH_CHECK();
}

/* Lazy InstanceNorm blocks (keymorph/layers.py:137-187 with norm_type "instance"; autograd of InstanceNorm3d -> ReLU ->
 * [MaxPool3d(2)] in one piece).  out (N,C,2) doubles = (sum g, sum g*z), g = du * [fma(z, scale, shift) > 0]; du, z (N,V,C). */
KMH_API int kmh_in_bwd_stats(const float* du, const float* z, const float* scale, const float* shift, int N, long long V,
                             int C, double* out, void* ws, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C > TPB * 4) return -22;
  int nblk = ceil_div(V, 2048);
  if (nblk > STAT_BLOCKS) nblk = STAT_BLOCKS;
  if (nblk < 1) nblk = 1;
  const bool v4 = (C % 4 == 0) && (C / 4 <= TPB);
  const int CQ = v4 ? C / 4 : C;
  if (CQ > TPB) return -22;
  const int rows = TPB / CQ;
  const size_t lds = (size_t)rows * C * 2 * sizeof(double);
  if (lds > 64 * 1024) return -22;
  dim3 g(nblk, N);
  if (v4) in_bwd_stats_kernel<4><<<g, TPB, lds, s>>>(du, z, scale, shift, V, C, (double*)ws);
  else in_bwd_stats_kernel<1><<<g, TPB, lds, s>>>(du, z, scale, shift, V, C, (double*)ws);
  kmh_stats::final_kernel<<<dim3(ceil_div(C * 2, 256 / kWave), N), 256, 0, s>>>((const double*)ws, nblk, C, out, nullptr);
  return KMH_LAUNCH_CHECK();
}

/* dz = c1 g + c2 z + c3 with g as above (c123 (N,C,3) from kmh_gn_bwd_coeffs with G = C); dz may alias du; dz_scale2
 * (2 floats) | NULL receives the f16x3 range scale {S, 1/S} of max |dz|. */
KMH_API int kmh_in_bwd_apply(const float* du, const float* z, const float* scale, const float* shift, const float* c123,
                             int N, long long V, int C, float* dz, float* dz_scale2, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (dz_scale2) {
    hipError_t e = hipMemsetAsync(dz_scale2, 0, 2 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  in_bwd_apply_kernel<<<dim3(stream_blocks(V * C / 4), N), TPB, 0, s>>>(du, z, scale, shift, c123, V, C, dz,
                                                                       reinterpret_cast<unsigned*>(dz_scale2));
  if (dz_scale2) kmh_absmax::final_kernel<<<1, 1, 0, s>>>(dz_scale2, 0.f);
  return KMH_LAUNCH_CHECK();
}

/* the same through MaxPool3d(2): du (N,D/2,H/2,W/2,C) and the winners of kmh_maxpool3d_fwd on the raw z (N,D,H,W,C);
 * dz (N,D,H,W,C) = c1 scatter(du [zhat(winner) > 0]) + c2 z + c3.  Even D, H, W; C % 4 == 0; 16-byte aligned tensors. */
KMH_API int kmh_in_bwd_apply_pool(const unsigned char* argmax, const float* du, const float* z, const float* scale,
                                  const float* shift, const float* c123, int N, int D, int H, int W, int C, float* dz,
                                  float* dz_scale2, void* stream) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long long pooled = (long long)Do * Ho * Wo * C;
  if (!argmax || !du || !z || !c123 || !dz || (C & 3) || (D & 1) || (H & 1) || (W & 1)) return -22;
  if (pooled / 4 >= (1ll << 31) || (long long)D * H * W >= (1ll << 31)) return -22;
  if ((((uintptr_t)du | (uintptr_t)z | (uintptr_t)dz) & 15) || ((uintptr_t)argmax & 3)) return -22;
  hipStream_t s = (hipStream_t)stream;
  if (dz_scale2) {
    hipError_t e = hipMemsetAsync(dz_scale2, 0, 2 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  in_bwd_apply_pool_kernel<<<dim3(stream_blocks(pooled), N), TPB, 0, s>>>(
      (const unsigned*)argmax, (const float4*)du, (const float4*)z, scale, shift, c123, (float4*)dz, D, H, W, C / 4, Do, Ho, Wo,
      reinterpret_cast<unsigned*>(dz_scale2));
  if (dz_scale2) kmh_absmax::final_kernel<<<1, 1, 0, s>>>(dz_scale2, 0.f);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_relu_mask(const float* dy, const float* y, long long n, float* dz, void* stream) {
  relu_mask_kernel<<<stream_blocks(n), TPB, 0, (hipStream_t)stream>>>(dy, y, n, dz);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_maxpool3d_fwd(const float* x, float* y, unsigned char* argmax, int N, int D, int H, int W, int C,
                              void* stream) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long long pooled = (long long)Do * Ho * Wo * C;
  if ((C & 3) == 0 && pooled / 4 < (1ll << 31) && (long long)D * H * W < (1ll << 31))
    maxpool_fwd4_kernel<<<dim3(stream_blocks(pooled), N), TPB, 0, (hipStream_t)stream>>>(
        (const float4*)x, (float4*)y, (unsigned*)argmax, D, H, W, C / 4, Do, Ho, Wo);
  else
    maxpool_fwd_kernel<<<dim3(stream_blocks(pooled), N), TPB, 0, (hipStream_t)stream>>>(x, y, argmax, D, H, W, C, Do, Ho, Wo);
  return KMH_LAUNCH_CHECK();
}

/* dx = scatter(dy) [+ add]; add (N,D,H,W,add_cstride >= C)|NULL is a second gradient of x that is summed in the same
 * pass (the U-Net skip connection; add may alias dx with add_cstride == C).  When any of D,H,W is odd the trailing
 * plane has no window: the caller pre-fills dx (zero, or a copy of add passed as add == dx). */
KMH_API int kmh_maxpool3d_bwd(const float* x, const unsigned char* argmax, const float* dy, const float* add,
                              int add_cstride, float* dx, int N, int D, int H, int W, int C, int out_blocked, void* stream) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  if ((add && add_cstride < C) || (!x && !argmax)) return -22;
  if (out_blocked && ((C & 7) || (D & 1) || (H & 1) || (W & 1) || add == dx)) return -22;   // whole chunks, every voxel written
  const long long pooled = (long long)Do * Ho * Wo * C;
  // the 16-byte kernel reads add / dy and writes dx as float4 and the winners as 4 packed bytes: a channel-slice view
  // whose storage offset is not a multiple of 4 floats takes the scalar kernel
  const bool aligned = (((uintptr_t)add | (uintptr_t)dx | (uintptr_t)dy) & 15) == 0 && ((uintptr_t)argmax & 3) == 0;
  if (argmax && aligned && (C & 3) == 0 && (add_cstride & 3) == 0 && pooled / 4 < (1ll << 31) &&
      (long long)D * H * W < (1ll << 31)) {
    static const bool pieces32 = getenv("KEYMORPH_POOL_BWD_PLANES") != nullptr;    // A/B runs: the per-plane 32-byte stores
    if (out_blocked && !add && !pieces32) {
      maxpool_bwd4_rows_kernel<<<dim3(stream_blocks(pooled), N), TPB, 0, (hipStream_t)stream>>>(
          (const unsigned*)argmax, (const float4*)dy, dx, D, H, W, C / 4, Do, Ho, Wo);
      return KMH_LAUNCH_CHECK();
    }
    auto kern = out_blocked ? maxpool_bwd4_kernel<true> : maxpool_bwd4_kernel<false>;
    kern<<<dim3(stream_blocks(pooled), N), TPB, 0, (hipStream_t)stream>>>((const unsigned*)argmax, (const float4*)dy, add,
                                                                         add_cstride, dx, D, H, W, C / 4, Do, Ho, Wo);
  } else {
    maxpool_bwd_kernel<<<dim3(stream_blocks(pooled), N), TPB, 0, (hipStream_t)stream>>>(
        x, argmax, dy, add, add_cstride, dx, D, H, W, C, Do, Ho, Wo, out_blocked);
  }
  return KMH_LAUNCH_CHECK();
}

/* MaxPool3d(2)'s backward written PRE-SPLIT for kmh_conv3d_fwd_bf(in_blocked = 2): dxs is (N, C/8, D*H*W + 1) records of 32
 * bytes -- 8 fp16 hi + 8 fp16 lo terms of fmaf(scatter(dy), S, 0), S = dy_scale2[0] (the {S, 1/S} range scale of dy, which a
 * scatter keeps), record D*H*W of every plane zero; kmh_maxpool3d_bwd_split_bytes gives its size.  Even D, H, W, C % 8 == 0,
 * winners as recorded by kmh_maxpool3d_fwd / kmh_conv3d_fwd_bf_pool.  (Autograd of max_pool3d,
 * keymorph/unet3d/buildingblocks.py:321-380, in the operand format of the data gradient that consumes it.) */
KMH_API size_t kmh_maxpool3d_bwd_split_bytes(int N, int D, int H, int W, int C) {
  return (size_t)N * (C / 8) * ((size_t)D * H * W + 1) * 32;
}
KMH_API int kmh_maxpool3d_bwd_split(const unsigned char* argmax, const float* dy, const float* dy_scale2, float* dxs, int N,
                                    int D, int H, int W, int C, void* stream) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long long pooled = (long long)Do * Ho * Wo * C;
  if (!argmax || !dy || !dy_scale2 || !dxs || (C & 7) || (D & 1) || (H & 1) || (W & 1) || N <= 0 || D <= 0 || H <= 0 || W <= 0)
    return -22;
  if (pooled / 4 >= (1ll << 31) || (long long)D * H * W >= (1ll << 31)) return -22;
  if ((((uintptr_t)dxs | (uintptr_t)dy) & 15) || ((uintptr_t)argmax & 7)) return -22;
  maxpool_bwd4_split_kernel<<<dim3(stream_blocks(pooled), N), TPB, 0, (hipStream_t)stream>>>(
      (const unsigned*)argmax, (const float4*)dy, dy_scale2, dxs, D, H, W, C / 4, Do, Ho, Wo);
  return KMH_LAUNCH_CHECK();
}

/* dx = scatter(dy) + [x > 0] (c1 dxn + c2 x + c3): MaxPool3d(2)'s backward (winners from kmh_maxpool3d_fwd) summed with a
 * second gradient of x whose GroupNorm backward (c123 (N,C,3), as kmh_gn_bwd_apply with relu_mask) is applied on the fly;
 * x, dxn, dx (N,D,H,W,C) dense with even D, H, W and C % 4 == 0; dx_scale2 (2 floats) | NULL receives {S, 1/S} for max |dx|
 * (autograd of max_pool3d + the skip connection of keymorph/unet3d/buildingblocks.py:363, 471-475). */
KMH_API int kmh_maxpool3d_bwd_lazy(const unsigned char* argmax, const float* dy, const float* dxn, const float* x,
                                   const float* c123, float* dx, int N, int D, int H, int W, int C, float* dx_scale2,
                                   void* stream) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long long pooled = (long long)Do * Ho * Wo * C;
  if (!argmax || !dy || !dxn || !x || !c123 || !dx || (C & 3) || (D & 1) || (H & 1) || (W & 1)) return -22;
  if (pooled / 4 >= (1ll << 31) || (long long)D * H * W >= (1ll << 31)) return -22;
  if ((((uintptr_t)dxn | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dy) & 15) || ((uintptr_t)argmax & 3)) return -22;
  hipStream_t s = (hipStream_t)stream;
  if (dx_scale2) {
    hipError_t e = hipMemsetAsync(dx_scale2, 0, 2 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  maxpool_bwd4_lazy_kernel<<<dim3(stream_blocks(pooled), N), TPB, 0, s>>>(
      (const unsigned*)argmax, (const float4*)dy, (const float4*)dxn, (const float4*)x, c123, (float4*)dx, D, H, W, C / 4, Do,
      Ho, Wo, reinterpret_cast<unsigned*>(dx_scale2));
  if (dx_scale2) kmh_absmax::final_kernel<<<1, 1, 0, s>>>(dx_scale2, 0.f);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_upcat_fwd(const float* skip, const float* low, float* out, int N, int D, int H, int W, int Cs,
                          int Dl, int Hl, int Wl, int Cl, void* stream) {
  if (((Cs | Cl) & 3) == 0 && (long long)D * H * W < (1ll << 32))
    upcat_fwd4_kernel<<<dim3(stream_blocks((long long)D * H * W * (Cs + Cl) / 4), N), TPB, 0, (hipStream_t)stream>>>(
        (const float4*)skip, (const float4*)low, (float4*)out, D, H, W, Cs / 4, Dl, Hl, Wl, Cl / 4);
  else
    upcat_fwd_kernel<<<dim3(stream_blocks((long long)D * H * W * (Cs + Cl)), N), TPB, 0, (hipStream_t)stream>>>(
        skip, low, out, D, H, W, Cs, Dl, Hl, Wl, Cl);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_upcat_bwd(const float* dout, float* dskip, float* dlow, int N, int D, int H, int W, int Cs, int Dl,
                          int Hl, int Wl, int Cl, int accumulate_skip, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long long V = (long long)D * H * W;
  if (dskip)   // NULL: the caller consumes dout[..., :Cs] in place (strided), no copy
    upcat_bwd_skip_kernel<<<dim3(stream_blocks(V * Cs), N), TPB, 0, s>>>(dout, dskip, V, Cs, Cs + Cl,
                                                                       accumulate_skip);
  if (((Cs | Cl) & 3) == 0 && D == 2 * Dl && H == 2 * Hl && W == 2 * Wl && V < (1ll << 32))
    upcat_bwd_low2x4_kernel<<<dim3(stream_blocks((long long)Dl * Hl * Wl * Cl / 4), N), TPB, 0, s>>>(
        (const float4*)dout, (float4*)dlow, H, W, Cs / 4, Dl, Hl, Wl, Cl / 4);
  else
    upcat_bwd_low_kernel<<<dim3(stream_blocks((long long)Dl * Hl * Wl * Cl), N), TPB, 0, s>>>(dout, dlow, D, H, W,
                                                                                             Cs, Dl, Hl, Wl, Cl);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_layout_convert(const float* in, float* out, int N, long long V, int C, int to_ncdhw,
                               void* stream) {
  dim3 g(ceil_div(V, 32), ceil_div(C, 32), N);
  nchw_to_nhwc_kernel<<<g, 256, 0, (hipStream_t)stream>>>(in, out, V, C, to_ncdhw);
  return KMH_LAUNCH_CHECK();
}
