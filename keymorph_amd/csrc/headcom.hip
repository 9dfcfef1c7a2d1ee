// Fused keypoint head: final 1x1x1 conv (+bias) -> ReLU -> center of mass, heat-map never written.
// Replaces  keymorph/unet3d/model.py:387-391 (final_conv)  +  keymorph/layers.py:92-134 (CenterOfMass3d)
// and their autograd.  At 256^3 / 512 keypoints the unfused path moves a 4.3 GB heat-map per image five
// times (write, CoM read, gradient write, two gradient reads); here the forward reads the 0.54 GB feature
// map (x4 keypoint groups, L2/MALL-resident) and the backward RECOMPUTES the logits instead of storing them.
//
//   fwd    : h[v,k] = b[k] + sum_c feat[v,c] W[k,c];  S[k] = sum_v relu(h[v,k]) * (1, cz(v), cy(v), cx(v))
//   bwd    : dh[v,k] = [h > 0] (g0[k] + gz[k] cz + gy[k] cy + gx[k] cx)       (g from dpts and S)
//            dfeat[v,c] = sum_k dh[v,k] W[k,c]     dW[k,c] = sum_v dh[v,k] feat[v,c]     db[k] = sum_v dh[v,k]
// All GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32).  M = 32 voxels, N = 32 keypoint channels: the MFMA C
// layout (col = channel = lane, row = voxel) makes the center-of-mass sums per-lane accumulations, and its
// registers are directly the A operand of the dW product (k-pair = voxels rho, rho+4).
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HTPB = 256;
constexpr int VT = 128;           // voxels per tile (one 32-voxel M-tile per wave)
constexpr int LD = 65;            // padded row (Cin <= 64)
constexpr int GC = 128;           // keypoint channels per group (4 N-tiles)

struct Dims { int D, H, W; };

__device__ __forceinline__ void stage_feat(const float* __restrict__ feat, long long v0, long long V, int Cin,
                                           Dims d, float* sX, float4* sC, int tid) {
  for (int e = tid; e < VT * 64; e += HTPB) {
    const int c = e & 63, v = e >> 6;
    float val = 0.f;
    if (c < Cin && v0 + v < V) val = feat[(v0 + v) * Cin + c];
    sX[v * LD + c] = val;
  }
  if (tid < VT) {
    const long long v = v0 + tid;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < V) {
      const int x = (int)(v % d.W), y = (int)((v / d.W) % d.H), z = (int)(v / ((long long)d.W * d.H));
      c.x = d.D > 1 ? (float)z / (float)(d.D - 1) : 0.f;
      c.y = d.H > 1 ? (float)y / (float)(d.H - 1) : 0.f;
      c.z = d.W > 1 ? (float)x / (float)(d.W - 1) : 0.f;
      c.w = 1.f;   // valid
    }
    sC[tid] = c;
  }
}

// W rows [co0, co0+128) -> sW[co][LD] (zero padded)
__device__ __forceinline__ void stage_w(const float* __restrict__ w, int co0, int Cout, int Cin, float* sW, int tid) {
  for (int e = tid; e < GC * 64; e += HTPB) {
    const int c = e & 63, k = e >> 6;
    sW[k * LD + c] = (c < Cin && co0 + k < Cout) ? w[(long long)(co0 + k) * Cin + c] : 0.f;
  }
}

// logits of this wave's 32 voxels x 128 channels of the group: acc[t] (C layout: col = channel, row = voxel)
__device__ __forceinline__ void gemm_logits(const float* sX, const float* sW, int Cin, int wv, int li, int lh,
                                            f32x16 acc[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int nk = (Cin + 1) >> 1;
  for (int kk = 0; kk < nk; ++kk) {
    const float a = sX[(wv * 32 + li) * LD + 2 * kk + lh];            // A[i = voxel][k = ci]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = sW[(32 * t + li) * LD + 2 * kk + lh];           // B[k = ci][j = co] = W[co][ci]
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HTPB, 2) void headcom_fwd_kernel(const float* __restrict__ feat,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              double* __restrict__ partial /* (N*Cout, nslab, 5) */,
                                                              long long V, int Cin, int Cout, Dims d,
                                                              int tiles_per_slab, int nslab) {
  __shared__ float sX[VT * LD];
  __shared__ float sW[GC * LD];
  __shared__ float4 sC[VT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int slab = blockIdx.x, co0 = blockIdx.y * GC, n = blockIdx.z;
  const float* fn = feat + (long long)n * V * Cin;
  stage_w(w, co0, Cout, Cin, sW, tid);
  float bv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) bv[t] = (bias && co0 + 32 * t + li < Cout) ? bias[co0 + 32 * t + li] : 0.f;
  float S[4][5];                  // per channel: sum h, sum h cz, sum h cy, sum h cx, sum h^2   (h = relu(logit))
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 5; ++k) S[t][k] = 0.f;
  const long long ntiles = (V + VT - 1) / VT;
  long long t_beg = (long long)slab * tiles_per_slab, t_end = t_beg + tiles_per_slab;
  if (t_end > ntiles) t_end = ntiles;
  for (long long tile = t_beg; tile < t_end; ++tile) {
    __syncthreads();
    stage_feat(fn, tile * VT, V, Cin, d, sX, sC, tid);
    __syncthreads();
    f32x16 acc[4];
    gemm_logits(sX, sW, Cin, wv, li, lh, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 c = sC[wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float h = fmaxf(acc[t][r] + bv[t], 0.f) * c.w;
        S[t][0] += h; S[t][1] += h * c.x; S[t][2] += h * c.y; S[t][3] += h * c.z; S[t][4] += h * h;
      }
    }
  }
  // combine the two half-waves (same channel, different rows), then the 4 waves through LDS
  __syncthreads();
  float* sR = sX;   // [4 waves][128 ch][5]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float v = S[t][k];
      v += __shfl_xor(v, 32, 64);
      if (lh == 0) sR[((wv * GC) + 32 * t + li) * 5 + k] = v;
    }
  __syncthreads();
  if (tid < GC && co0 + tid < Cout) {
    double* o = partial + (((long long)n * Cout + co0 + tid) * nslab + slab) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k)
      o[k] = (double)sR[(0 * GC + tid) * 5 + k] + (double)sR[(1 * GC + tid) * 5 + k] +
             (double)sR[(2 * GC + tid) * 5 + k] + (double)sR[(3 * GC + tid) * 5 + k];
  }
}

__global__ void headcom_final_kernel(const double* __restrict__ partial, int nslab, int NK, float* __restrict__ pts,
                                     float* __restrict__ sums, float* __restrict__ sq /* (N*K) sum relu(h)^2 | NULL */) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= NK) return;
  double s[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < nslab; ++b)
    for (int k = 0; k < 5; ++k) s[k] += partial[((long long)ch * nslab + b) * 5 + k];
  const double den = s[0] + 1e-8;
  for (int k = 0; k < 3; ++k) pts[ch * 3 + k] = (float)(s[1 + k] / den * 2.0 - 1.0);
  for (int k = 0; k < 4; ++k) sums[ch * 4 + k] = (float)s[k];
  if (sq) sq[ch] = (float)s[4];
}

// g (N*K, 4) = coefficients of d(loss)/d(relu(h)) = g0 + gz cz + gy cy + gx cx
// dpower (N*K)|NULL = d(loss)/d(sum relu(h)) (keypoint weighting by power) adds a constant to g0
__global__ void headcom_coef_kernel(const float* __restrict__ dpts, const float* __restrict__ dpower,
                                    const float* __restrict__ sums, int NK, float* __restrict__ g) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= NK) return;
  // A dead channel (sum relu(h) == 0 exactly, i.e. h <= 0 at every voxel) has dh = [h > 0] (...) = 0 whatever its
  // coefficients are; 2 / (0 + 1e-8) would make them 1e13 times a live channel's, and the range scale of the
  // split-fp16 head gradient (head_dh_scale_kernel: one power of two for all channels) would push every live channel
  // below fp16's range.  Zero coefficients give the same (zero) gradient and leave the scale to the live channels.
  if (sums[ch * 4] == 0.f) {
    g[ch * 4 + 0] = g[ch * 4 + 1] = g[ch * 4 + 2] = g[ch * 4 + 3] = 0.f;
    return;
  }
  const float den = sums[ch * 4] + 1e-8f;
  const float k2 = 2.f / den;
  const float gz = dpts[ch * 3] * k2, gy = dpts[ch * 3 + 1] * k2, gx = dpts[ch * 3 + 2] * k2;
  g[ch * 4 + 0] = -(gz * (sums[ch * 4 + 1] / den) + gy * (sums[ch * 4 + 2] / den) + gx * (sums[ch * 4 + 3] / den)) +
                  (dpower ? dpower[ch] : 0.f);
  g[ch * 4 + 1] = gz; g[ch * 4 + 2] = gy; g[ch * 4 + 3] = gx;
}

// in-place: logits (acc, C layout) -> dh = [h > 0] (g0 + gz cz + gy cy + gx cx)
__device__ __forceinline__ void logits_to_dh(f32x16 acc[4], const float bv[4], const float4 gv[4], const float4* sC,
                                             int wv, int lh) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float4 c = sC[wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float h = acc[t][r] + bv[t];
      const float gd = gv[t].x + gv[t].y * c.x + gv[t].z * c.y + gv[t].w * c.z;
      acc[t][r] = (h > 0.f && c.w > 0.f) ? gd : 0.f;
    }
  }
}

// dfeat: one workgroup per voxel tile, loops over all keypoint groups
__global__ __launch_bounds__(HTPB, 1) void headcom_bwd_feat_kernel(const float* __restrict__ feat,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ g,
                                                                   float* __restrict__ dfeat, long long V, int Cin,
                                                                   int Cout, Dims d, int mask_dfeat) {
  __shared__ float sX[VT * LD];
  __shared__ float sW[GC * LD];
  __shared__ float sH[4 * 32 * 33];
  __shared__ float4 sC[VT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y;
  const long long v0 = (long long)blockIdx.x * VT;
  const float* fn = feat + (long long)n * V * Cin;
  stage_feat(fn, v0, V, Cin, d, sX, sC, tid);
  f32x16 acc2[2];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[cc][r] = 0.f;
  float* myH = sH + wv * 32 * 33;
  for (int co0 = 0; co0 < Cout; co0 += GC) {
    __syncthreads();
    stage_w(w, co0, Cout, Cin, sW, tid);
    float bv[4];
    float4 gv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int co = co0 + 32 * t + li;
      bv[t] = (bias && co < Cout) ? bias[co] : 0.f;
      gv[t] = co < Cout ? *reinterpret_cast<const float4*>(g + ((long long)n * Cout + co) * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    f32x16 acc[4];
    gemm_logits(sX, sW, Cin, wv, li, lh, acc);
    logits_to_dh(acc, bv, gv, sC, wv, lh);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // transpose this 32 voxel x 32 channel block through the wave-private LDS tile
#pragma unroll
      for (int r = 0; r < 16; ++r) myH[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[t][r];
      // (same wave: LDS operations execute in order, no barrier needed)
      for (int s = 0; s < 16; ++s) {
        const float a = myH[li * 33 + 2 * s + lh];                          // A[i = voxel][k = channel]
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const float b = sW[(32 * t + 2 * s + lh) * LD + 32 * cc + li];   // B[k = channel][j = ci] = W[co][ci]
          acc2[cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2[cc], 0, 0, 0);
        }
      }
    }
  }
  float* dn = dfeat + (long long)n * V * Cin;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int c = 32 * cc + li;
    if (c >= Cin) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long v = v0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (v < V) dn[v * Cin + c] = (mask_dfeat && !(fn[v * Cin + c] > 0.f)) ? 0.f : acc2[cc][r];
    }
  }
}

// dW / db: workgroup = (slab of voxel tiles over all samples, keypoint group); per-wave partial slabs
__global__ __launch_bounds__(HTPB, 2) void headcom_bwd_w_kernel(const float* __restrict__ feat,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ g,
                                                                float* __restrict__ pw /* (nslab*4, Cout, Cin) */,
                                                                float* __restrict__ pb /* (nslab*4, Cout) */,
                                                                int N, long long V, int Cin, int Cout, Dims d,
                                                                int tiles_per_slab) {
  __shared__ float sX[VT * LD];
  __shared__ float sW[GC * LD];
  __shared__ float4 sC[VT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int slab = blockIdx.x, co0 = blockIdx.y * GC;
  stage_w(w, co0, Cout, Cin, sW, tid);
  float bv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) bv[t] = (bias && co0 + 32 * t + li < Cout) ? bias[co0 + 32 * t + li] : 0.f;
  f32x16 dw[4][2];
  float db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r) dw[t][cc][r] = 0.f;
  const long long tiles_per_n = (V + VT - 1) / VT, ntiles = tiles_per_n * N;
  long long t_beg = (long long)slab * tiles_per_slab, t_end = t_beg + tiles_per_slab;
  if (t_end > ntiles) t_end = ntiles;
  for (long long tile = t_beg; tile < t_end; ++tile) {
    const int n = (int)(tile / tiles_per_n);
    const long long v0 = (tile - (long long)n * tiles_per_n) * VT;
    __syncthreads();
    stage_feat(feat + (long long)n * V * Cin, v0, V, Cin, d, sX, sC, tid);
    float4 gv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int co = co0 + 32 * t + li;
      gv[t] = co < Cout ? *reinterpret_cast<const float4*>(g + ((long long)n * Cout + co) * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    f32x16 acc[4];
    gemm_logits(sX, sW, Cin, wv, li, lh, acc);
    logits_to_dh(acc, bv, gv, sC, wv, lh);
    // dW[k, c] += sum_v dh[v, k] feat[v, c]: the C-layout register r of lane (li, lh) IS A[i = k = li][kk = lh]
    // for the voxel pair (rho, rho + 4); B[kk][j = c] = feat[voxel rho + 4 kk][c]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int vox = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float b[2];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) b[cc] = sX[vox * LD + 32 * cc + li];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        db[t] += acc[t][r];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
          dw[t][cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[t][r], b[cc], dw[t][cc], 0, 0, 0);
      }
    }
  }
  float* ow = pw + ((long long)slab * 4 + wv) * Cout * Cin;
  float* ob = pb + ((long long)slab * 4 + wv) * Cout;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = 32 * cc + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < Cout && c < Cin) ow[(long long)co * Cin + c] = dw[t][cc][r];
      }
    }
    const float s = db[t] + __shfl_xor(db[t], 32, 64);
    const int co = co0 + 32 * t + li;
    if (lh == 0 && co < Cout) ob[co] = s;
  }
}

// out[e] = sum over the per-slab partials, fp64, fixed order: 64 outputs x 4 slices of the slabs per workgroup
__global__ __launch_bounds__(256) void headcom_reduce_kernel(const float* __restrict__ partial, int nparts,
                                                             long long total, float* __restrict__ out) {
  __shared__ double red[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (long long e0 = (long long)blockIdx.x * 64; e0 < total; e0 += (long long)gridDim.x * 64) {
    const long long e = e0 + o;
    double s = 0;
    if (e < total)
      for (int k = sl; k < nparts; k += 4) s += partial[(long long)k * total + e];
    __syncthreads();
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0 && e < total) out[e] = (float)((red[0][o] + red[1][o]) + (red[2][o] + red[3][o]));
  }
}

// =============================================================================================
// Split-bf16 head (default arithmetic, same scheme as csrc/conv_bf.hip): every fp32 operand is split exactly
// into TERMS bf16 terms and each product block is accumulated from the 6 (TERMS = 3) or 3 (TERMS = 2)
// significant cross terms with v_mfma_f32_32x32x16_bf16 -- fp32-class logits and gradients at ~2.2x the
// fp32-MFMA rate.  All LDS images are [rows][64 bf16] with 128-byte rows; the 16-byte chunk index is XOR-
// swizzled with (row >> 1) & 7 so that a ds_read_b128 of 16 different rows is conflict free.
//
// Wherever the K index of a GEMM is the ROW index of an accumulator tile that is consumed straight from
// registers (dW: K = voxels of the logits tile; dfeat: K = channels of the transposed logits tile), the other
// operand's image is stored with the in-block order  sigma(i) = i with bits 2 and 3 swapped, because register r
// of lane-half lh holds row (r & 3) + 8 (r >> 2) + 4 lh, i.e. k = 8 lh + e  <->  row sigma^-1(16 s2 + 8 lh + e).
typedef kmh_bf16x8 bf16x8;     // 8 x 16-bit fragment: bf16 (TERMS == 3) or range-scaled fp16 (TERMS == 2), see common.h

__device__ __forceinline__ int sig5(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Sign mask of the logits ([h > 0], 1 bit per (voxel, keypoint channel)), written by the ROWS forward kernel when the
// caller wants a backward pass and read by the two mask backward kernels instead of recomputing the logits GEMM there:
//   mask[(n V + v) / 32][CoutP] 32-bit words: bit b of word k = [h > 0] of voxel 32 * block + b, channel k.
// (Scalar loads of these words straight into v_cndmask's lane-mask operand were tried in the dfeat kernel: one VALU per
// element, but every 64-byte line is a first touch of HBM -- ~1 us exposed per s_load, four per 64-channel block, and the
// kernel ran at that latency.  The words now ride with the staged filter block, a block ahead.)

template <int TERMS>
__device__ __forceinline__ void split4(const float4 v, uint2 out[TERMS]) {
  if constexpr (TERMS == 2) {          // packed conversions (common.h split_pair): 3 VALU per pair
    unsigned a[2], b[2];
    split_pair<2>(v.x, v.y, a);
    split_pair<2>(v.z, v.w, b);
    out[0] = make_uint2(a[0], b[0]);
    out[1] = make_uint2(a[1], b[1]);
    return;
  }
  float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int t = 0; t < TERMS; ++t) {
    unsigned h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { float back; h[j] = to16<TERMS>(r[j], back); r[j] -= back; }
    out[t].x = h[0] | (h[1] << 16);
    out[t].y = h[2] | (h[3] << 16);
  }
}
// AMP (KeyMorph(use_amp=True), kmh_conv_set_amp; TERMS == 2 only): the hi x hi product alone, as the reference's autocast
// runs this convolution in fp16 (keymorph/model.py:176-191)
template <int TERMS, bool AMP = false>
__device__ __forceinline__ f32x16 mfma_split(const bf16x8 a[TERMS], const bf16x8 b[TERMS], f32x16 acc) {
  if constexpr (TERMS == 3) {
    acc = mfma16<TERMS>(a[2], b[0], acc);
    acc = mfma16<TERMS>(a[1], b[1], acc);
    acc = mfma16<TERMS>(a[0], b[2], acc);
  }
  if constexpr (!(AMP && TERMS == 2)) {
    acc = mfma16<TERMS>(a[1], b[0], acc);
    acc = mfma16<TERMS>(a[0], b[1], acc);
  }
  acc = mfma16<TERMS>(a[0], b[0], acc);
  return acc;
}

// Range scales of the head (device floats, written on the stream; all 1 for TERMS == 3):
//   hs[0..1] = {S_F, 1/S_F} features, hs[2..3] = {S_W, 1/S_W} weights, hs[4..5] = {S_dh, 1/S_dh} head gradient
// Pre-split weight images (one tiny launch per forward / backward):
//   wk  [t][CoutP][64]   wk [t][k][ci]                       (B operand of the logits GEMM, forward orientation)
//   wkp [t][CoutP][64]   wkp[t][k][16*(ci>>4) + sig(ci&15)]  (A operand of the transposed logits GEMM: feature
//                        fragments there hold ci with bit 2 == lane-half, see headcom_bwd_feat_bf_kernel)
//   wt  [t][64][CoutP]   wt [t][ci][32*(k>>5) + sig5(k&31)]  (A operand of dfeat^T = W^T dh^T)
template <int TERMS>
__global__ __launch_bounds__(256) void headcom_pack_bf_kernel(const float* __restrict__ w, int Cout, int Cin, int CoutP,
                                                             __bf16* __restrict__ wk, __bf16* __restrict__ wkp,
                                                             __bf16* __restrict__ wt, const float* __restrict__ hs) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= CoutP * 64) return;
  const int ci = e & 63, k = e >> 6;
  float r = ((k < Cout && ci < Cin) ? w[(long long)k * Cin + ci] : 0.f) * hs[2];
  const int cip = (ci & ~15) | sig5(ci & 15), kp = (k & ~31) | sig5(k & 31);
#pragma unroll
  for (int t = 0; t < TERMS; ++t) {
    float back;
    const unsigned short h = to16<TERMS>(r, back);
    r -= back;
    reinterpret_cast<unsigned short*>(wk)[((long long)t * CoutP + k) * 64 + ci] = h;
    reinterpret_cast<unsigned short*>(wkp)[((long long)t * CoutP + k) * 64 + cip] = h;
    reinterpret_cast<unsigned short*>(wt)[((long long)t * 64 + ci) * CoutP + kp] = h;
  }
}

// feature tile (VTT voxels x 64 ci, fp32 NDHWC) -> sF[t][voxel][ci] and optionally sFT[t][ci][sigma(voxel)]; coords
// Two halves so that a kernel can keep the NEXT tile's loads in flight under the current tile's MFMAs:
//   FeatRegs r; feat_fetch(r, tile + 1);  ... compute tile ...  barrier; feat_commit(r)
template <int VTT, int NTHR>
struct FeatRegs {
  static constexpr int ITEMS = (VTT / 2) * 16;         // (voxel pair, 4-channel quad)
  static constexpr int NIT = (ITEMS + NTHR - 1) / NTHR;
  float4 x0[NIT], x1[NIT];
};
template <int VTT, int NTHR>
__device__ __forceinline__ void feat_fetch(FeatRegs<VTT, NTHR>& r, const float* __restrict__ feat, long long v0,
                                           long long V, int Cin, int tid) {
#pragma unroll
  for (int k = 0; k < FeatRegs<VTT, NTHR>::NIT; ++k) {
    const int e = tid + k * NTHR;
    const int c4 = e & 15, v = 2 * (e >> 4);
    r.x0[k] = r.x1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < FeatRegs<VTT, NTHR>::ITEMS && 4 * c4 < Cin) {
      if (v0 + v < V) r.x0[k] = *reinterpret_cast<const float4*>(feat + (v0 + v) * Cin + 4 * c4);
      if (v0 + v + 1 < V) r.x1[k] = *reinterpret_cast<const float4*>(feat + (v0 + v + 1) * Cin + 4 * c4);
    }
  }
}
template <int TERMS, int VTT, int NTHR, bool TRANSPOSED, bool PLAIN = true>
__device__ __forceinline__ void feat_commit(const FeatRegs<VTT, NTHR>& r, long long v0, long long V, Dims d,
                                            unsigned char* sF, unsigned char* sFT, float4* sC, int tid, float sFs,
                                            bool coords);

template <int TERMS, int VTT, int NTHR, bool TRANSPOSED>
__device__ __forceinline__ void stage_feat_bf(const float* __restrict__ feat, long long v0, long long V, int Cin, Dims d,
                                              unsigned char* sF, unsigned char* sFT, float4* sC, int tid, float sFs) {
  FeatRegs<VTT, NTHR> r;
  feat_fetch<VTT, NTHR>(r, feat, v0, V, Cin, tid);
  feat_commit<TERMS, VTT, NTHR, TRANSPOSED>(r, v0, V, d, sF, sFT, sC, tid, sFs, true);
}

template <int TERMS, int VTT, int NTHR, bool TRANSPOSED, bool PLAIN>
__device__ __forceinline__ void feat_commit(const FeatRegs<VTT, NTHR>& r, long long v0, long long V, Dims d,
                                            unsigned char* sF, unsigned char* sFT, float4* sC, int tid, float sFs,
                                            bool coords) {
#pragma unroll
  for (int k = 0; k < FeatRegs<VTT, NTHR>::NIT; ++k) {
    const int e = tid + k * NTHR;
    if (e >= FeatRegs<VTT, NTHR>::ITEMS) break;
    const int c4 = e & 15, vp = e >> 4, v = 2 * vp;
    float4 x0 = r.x0[k], x1 = r.x1[k];
    x0.x *= sFs; x0.y *= sFs; x0.z *= sFs; x0.w *= sFs;          // power-of-two range scale (1 for TERMS == 3)
    x1.x *= sFs; x1.y *= sFs; x1.z *= sFs; x1.w *= sFs;
    uint2 s0[TERMS], s1[TERMS];
    split4<TERMS>(x0, s0);
    split4<TERMS>(x1, s1);
    if (PLAIN) {
#pragma unroll
      for (int t = 0; t < TERMS; ++t) {
        *reinterpret_cast<uint2*>(sF + t * (VTT * 128) + swz(v, c4 >> 1) + (c4 & 1) * 8) = s0[t];
        *reinterpret_cast<uint2*>(sF + t * (VTT * 128) + swz(v + 1, c4 >> 1) + (c4 & 1) * 8) = s1[t];
      }
    }
    if (TRANSPOSED) {
      const int pos = (v & ~31) | sig5(v & 31);        // v even -> pos even, voxel v+1 sits at pos+1
#pragma unroll
      for (int t = 0; t < TERMS; ++t) {
        const unsigned lo0 = s0[t].x & 0xffffu, lo1 = s0[t].x >> 16, lo2 = s0[t].y & 0xffffu, lo3 = s0[t].y >> 16;
        const unsigned hi0 = s1[t].x & 0xffffu, hi1 = s1[t].x >> 16, hi2 = s1[t].y & 0xffffu, hi3 = s1[t].y >> 16;
        unsigned char* base = sFT + t * (64 * 128) + (pos & 7) * 2;
        *reinterpret_cast<unsigned*>(base + swz(4 * c4 + 0, pos >> 3)) = lo0 | (hi0 << 16);
        *reinterpret_cast<unsigned*>(base + swz(4 * c4 + 1, pos >> 3)) = lo1 | (hi1 << 16);
        *reinterpret_cast<unsigned*>(base + swz(4 * c4 + 2, pos >> 3)) = lo2 | (hi2 << 16);
        *reinterpret_cast<unsigned*>(base + swz(4 * c4 + 3, pos >> 3)) = lo3 | (hi3 << 16);
      }
    }
  }
  if (coords && tid < VTT) {
    const long long v = v0 + tid;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < V) {
      const int x = (int)(v % d.W), y = (int)((v / d.W) % d.H), z = (int)(v / ((long long)d.W * d.H));
      c.x = d.D > 1 ? (float)z / (float)(d.D - 1) : 0.f;
      c.y = d.H > 1 ? (float)y / (float)(d.H - 1) : 0.f;
      c.z = d.W > 1 ? (float)x / (float)(d.W - 1) : 0.f;
      c.w = 1.f;
    }
    sC[tid] = c;
  }
}

// ROWS kernels walk 32-voxel blocks of whole x rows in order: the block's (x, y, z) origin is advanced with three
// compares instead of being re-derived with 64-bit divisions (~125 scalar instructions each) for every block.
struct RowCursor {
  int x, y, z;
  __device__ __forceinline__ void set(long long v, Dims d) {
    const unsigned u = (unsigned)v;                     // ROWS launches guarantee V < 2^31
    const unsigned row = u / (unsigned)d.W;
    x = (int)(u - row * (unsigned)d.W);
    z = (int)(row / (unsigned)d.H);
    y = (int)(row - (unsigned)z * (unsigned)d.H);
  }
  __device__ __forceinline__ void advance32(Dims d) {   // the next 32-voxel block of the same sample (wraps to 0,0,0)
    x += 32;
    if (x >= d.W) { x = 0; if (++y >= d.H) { y = 0; if (++z >= d.D) z = 0; } }
  }
};

constexpr int FVT = 128;   // forward: voxels per tile
constexpr int WVT = 64;    // dW kernel: voxels per tile (its transposed image has 64 columns)

// ---- forward: workgroup = (slab of voxel tiles, 128 keypoint channels, sample); wave = 32 channels --------
// ROWS: W % 32 == 0 and V % FVT == 0 -- every 32-voxel block is a piece of one x row and no voxel is padding, so the
// moment sums need the voxel coordinates only once per block (z, y) or as compile-time offsets (x): 4 VALU per logit
// instead of 9 and no coordinate reads.  (The epilogue, not the MFMAs, bounds this kernel: 16 logits per lane per
// 12 MFMAs.)  Coordinates are accumulated as voxel INDICES and scaled by 1/(dim-1) when the partials are written.
// NWV waves per workgroup = 32 NWV keypoint channels share one staged (converted) feature tile: 4 (128 channels, 3
// workgroups per CU) or 16 (all 512 channels of the headline config on one tile image -- the conversion, which is
// what bounds these kernels, is then done once per tile instead of once per 128-channel group).
template <int TERMS, bool ROWS, int NWV, bool AMP = false>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 3 : 1) void headcom_fwd_bf_kernel(const float* __restrict__ feat,
                                                                 const __bf16* __restrict__ wk,
                                                                 const float* __restrict__ bias,
                                                                 double* __restrict__ partial, long long V, int Cin,
                                                                 int Cout, int CoutP, Dims d, int tiles_per_slab,
                                                                 int nslab, int ngroups, const float* __restrict__ hs,
                                                                 int want_sq, unsigned* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  constexpr int FBUF = TERMS * FVT * 128 + FVT * 16;                   // one buffer: sF [TERMS][FVT][128 B] + sC [FVT]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int item = xcd_remap(blockIdx.x, gridDim.x);                  // channel groups of one slab share an L2
  const int grp = item % ngroups, slab = item / ngroups, n = blockIdx.y;
  constexpr int NT_ = 64 * NWV;
  const int co = grp * (32 * NWV) + 32 * wv + li;
  const int nks = (Cin + 15) >> 4;
  const float* fn = feat + (long long)n * V * Cin;
  bf16x8 bw[4][TERMS];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < TERMS; ++t)
      bw[s][t] = *reinterpret_cast<const bf16x8*>(wk + ((long long)t * CoutP + co) * 64 + 16 * s + 8 * lh);
  const float bv = (bias && co < Cout) ? bias[co] : 0.f;
  const float sFs = hs[0], desc = hs[1] * hs[3];
  float S[5] = {0.f, 0.f, 0.f, 0.f, 0.f};     // sum h, sum h cz, sum h cy, sum h cx, sum h^2
  const long long ntiles = (V + FVT - 1) / FVT;
  long long t_beg = (long long)slab * tiles_per_slab, t_end = t_beg + tiles_per_slab;
  if (t_end > ntiles) t_end = ntiles;
  const float bv_s = bv / desc;               // ROWS: the accumulator starts at bias / descale, h' = max(acc, 0) = h / descale
  f32x16 acc0;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = ROWS ? bv_s : 0.f;
  // Two LDS buffers, one barrier per tile: a wave converts tile t+1 (fetched under tile t's MFMAs) into the other
  // buffer as soon as ITS tile-t blocks are done, while slower waves are still on the matrix cores.
  FeatRegs<FVT, NT_> pre;
  auto commit = [&](long long tile) {
    unsigned char* b = hsm + (int)((tile - t_beg) & 1) * FBUF;
    feat_commit<TERMS, FVT, NT_, false>(pre, tile * FVT, V, d, b, nullptr, reinterpret_cast<float4*>(b + TERMS * FVT * 128),
                                        tid, sFs, !ROWS);
  };
  if (t_beg < t_end) {
    feat_fetch<FVT, NT_>(pre, fn, t_beg * FVT, V, Cin, tid);
    commit(t_beg);
    if (t_beg + 1 < t_end) feat_fetch<FVT, NT_>(pre, fn, (t_beg + 1) * FVT, V, Cin, tid);
  }
  __syncthreads();
  RowCursor cur;
  if (ROWS) cur.set(t_beg * FVT, d);
  for (long long tile = t_beg; tile < t_end; ++tile) {
    const unsigned char* sF = hsm + (int)((tile - t_beg) & 1) * FBUF;
    const float4* sC = reinterpret_cast<const float4*>(sF + TERMS * FVT * 128);
#pragma unroll 1
    for (int vb = 0; vb < FVT / 32; ++vb) {
      f32x16 acc = acc0;                     // the first product reads acc0 as its C operand: no per-block initialisation
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < nks) {
          bf16x8 a[TERMS];
#pragma unroll
          for (int t = 0; t < TERMS; ++t)
            a[t] = *reinterpret_cast<const bf16x8*>(sF + t * (FVT * 128) + swz(32 * vb + li, 2 * s + lh));
          acc = mfma_split<TERMS, AMP>(a, bw[s], acc);
        }
      }
      if (ROWS) {
        const int xb = cur.x, yb = cur.y, zb = cur.z;                  // wave-uniform origin of block tile * FVT + 32 vb
        cur.advance32(d);
        float R = 0.f, T = 0.f;
        if (want_sq) {                       // sum relu(h)^2: keypoint weighting at inference only (head_moments)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float h = fmaxf(acc[r], 0.f);
            R += h;
            T = fmaf(h, (float)((r & 3) + 8 * (r >> 2)), T);
            S[4] = fmaf(h, h, S[4]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float h = fmaxf(acc[r], 0.f);
            R += h;
            T = fmaf(h, (float)((r & 3) + 8 * (r >> 2)), T);
          }
        }
        S[0] += R;
        S[1] = fmaf(R, (float)zb, S[1]);
        S[2] = fmaf(R, (float)yb, S[2]);
        S[3] += fmaf(R, (float)(xb + 4 * lh), T);
        if (mask) {            // [h > 0] of this block for the backward kernels: two VALU per logit
          unsigned bits = 0u;
#pragma unroll
          for (int q = 3; q >= 0; --q) {
#pragma unroll
            for (int j = 3; j >= 0; --j)
              asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(acc[4 * q + j]) : "vcc");
            if (q) bits <<= 4;                                           // register 4 q + j = voxel row 8 q + j (+ 4 lh)
          }
          bits <<= 4 * lh;
          bits |= __shfl_xor(bits, 32, 64);
          if (lh == 0)
            mask[(((long long)n * V + tile * FVT) / 32 + vb) * CoutP + co] = bits;
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 c = sC[32 * vb + (r & 3) + 8 * (r >> 2) + 4 * lh];
        const float h = fmaxf(acc[r] * desc + bv, 0.f) * c.w;
        S[0] += h; S[1] += h * c.x; S[2] += h * c.y; S[3] += h * c.z; S[4] += h * h;
      }
    }
    if (tile + 1 < t_end) {
      commit(tile + 1);
      if (tile + 2 < t_end) feat_fetch<FVT, NT_>(pre, fn, (tile + 2) * FVT, V, Cin, tid);
    }
    __syncthreads();
  }
  if (ROWS) {   // back to descaled logits and normalised coordinates
    S[0] *= desc; S[4] *= desc * desc;
    S[1] *= desc / (float)(d.D > 1 ? d.D - 1 : 1); S[2] *= desc / (float)(d.H > 1 ? d.H - 1 : 1);
    S[3] *= desc / (float)(d.W > 1 ? d.W - 1 : 1);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) S[k] += __shfl_xor(S[k], 32, 64);
  if (lh == 0 && co < Cout) {
    double* o = partial + (((long long)n * Cout + co) * nslab + slab) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = (double)S[k];
  }
}

// ---- dW / db: workgroup = (slab of 64-voxel tiles over all samples, 128 channels); wave = 32 channels -----
// ROWS (W % 32 == 0, V % WVT == 0): as in the forward kernel -- the accumulator starts at bias / descale, the gradient
// coefficients carry the operand's range scale, and the per-voxel factor is one fma on the block's row constants.
template <int TERMS, bool ROWS, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void headcom_bwd_w_bf_kernel(const float* __restrict__ feat,
                                                                   const __bf16* __restrict__ wk,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ g,
                                                                   float* __restrict__ pw /* (nslab, Cout, Cin) */,
                                                                   float* __restrict__ pb /* (nslab, Cout) */, int N,
                                                                   long long V, int Cin, int Cout, int CoutP, Dims d,
                                                                   int tiles_per_slab, int ngroups,
                                                                   const float* __restrict__ hs,
                                                                   const int* __restrict__ gate, int want) {
  if (gate && *gate != want) return;         // the other arithmetic runs this backward (head_dh_scale_kernel)
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  // one buffer: sF [TERMS][64 voxels][128 B] + sFT [TERMS][64 ci][128 B] (columns = sigma(voxel)) + sC [64]
  constexpr int WBUF = 2 * TERMS * WVT * 128 + WVT * 16;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = item % ngroups, slab = item / ngroups;
  constexpr int NT_ = 64 * NWV;
  const int co = grp * (32 * NWV) + 32 * wv + li;
  const int nks = (Cin + 15) >> 4;
  bf16x8 bw[4][TERMS];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < TERMS; ++t)
      bw[s][t] = *reinterpret_cast<const bf16x8*>(wk + ((long long)t * CoutP + co) * 64 + 16 * s + 8 * lh);
  const float bv = (bias && co < Cout) ? bias[co] : 0.f;
  const float sFs = hs[0], desc = hs[1] * hs[3], sDh = hs[4], desc_w = hs[5] * hs[1];
  f32x16 dw[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[ct][r] = 0.f;
  float db = 0.f;
  f32x16 acc0;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = ROWS ? bv / desc : 0.f;
  const long long tiles_per_n = (V + WVT - 1) / WVT, ntiles = tiles_per_n * N;
  long long t_beg = (long long)slab * tiles_per_slab, t_end = t_beg + tiles_per_slab;
  if (t_end > ntiles) t_end = ntiles;
  FeatRegs<WVT, NT_> pre;                    // the next tile's features, in flight under this tile's MFMAs
  // (sample, first voxel) of the current tile, stepped without divisions: tiles never straddle samples
  int n = t_beg < t_end ? (int)(t_beg / tiles_per_n) : 0;
  long long v0 = (t_beg - (long long)n * tiles_per_n) * WVT;
  // Two LDS buffers, one barrier per tile (see the forward kernel): tile t+1 is converted into the other buffer by each
  // wave right after its own tile-t blocks; tile t+2 is then fetched.
  auto commit = [&](long long tile, long long vfirst) {
    unsigned char* b = hsm + (int)((tile - t_beg) & 1) * WBUF;
    feat_commit<TERMS, WVT, NT_, true>(pre, vfirst, V, d, b, b + TERMS * WVT * 128,
                                       reinterpret_cast<float4*>(b + 2 * TERMS * WVT * 128), tid, sFs, !ROWS);
  };
  auto step = [&](int& nn, long long& vv) { vv += WVT; if (vv >= V) { vv = 0; ++nn; } };
  int n_next = n;                            // (sample, first voxel) of the tile after the current one
  long long v_next = v0;
  step(n_next, v_next);
  if (t_beg < t_end) {
    feat_fetch<WVT, NT_>(pre, feat + (long long)n * V * Cin, v0, V, Cin, tid);
    commit(t_beg, v0);
    if (t_beg + 1 < t_end) feat_fetch<WVT, NT_>(pre, feat + (long long)n_next * V * Cin, v_next, V, Cin, tid);
  }
  __syncthreads();
  RowCursor cur;
  if (ROWS) cur.set(v0, d);
  for (long long tile = t_beg; tile < t_end; ++tile) {
    const unsigned char* sF = hsm + (int)((tile - t_beg) & 1) * WBUF;
    const unsigned char* sFT = sF + TERMS * WVT * 128;
    const float4* sC = reinterpret_cast<const float4*>(sF + 2 * TERMS * WVT * 128);
    const float4 gv = co < Cout ? *reinterpret_cast<const float4*>(g + ((long long)n * Cout + co) * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int vb = 0; vb < WVT / 32; ++vb) {
      f32x16 acc = acc0;                     // the first product reads acc0 as its C operand
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < nks) {
          bf16x8 a[TERMS];
#pragma unroll
          for (int t = 0; t < TERMS; ++t)
            a[t] = *reinterpret_cast<const bf16x8*>(sF + t * (WVT * 128) + swz(32 * vb + li, 2 * s + lh));
          acc = mfma_split<TERMS>(a, bw[s], acc);
        }
      }
      // dh = [h > 0] (g0 + gz cz + gy cy + gx cx): lane = channel, register r = voxel row
      float dh[16];
      if (ROWS) {
        const int xb = cur.x, yb = cur.y, zb = cur.z;                  // wave-uniform origin of block v0 + 32 vb
        cur.advance32(d);                                              // (wraps to the origin at a sample boundary)
        const float iz = d.D > 1 ? 1.f / (float)(d.D - 1) : 0.f, iy = d.H > 1 ? 1.f / (float)(d.H - 1) : 0.f,
                    ix = d.W > 1 ? 1.f / (float)(d.W - 1) : 0.f;
        const float gxs = gv.w * ix * sDh;                             // per x step, in the operand's range scale
        const float G = (gv.x + gv.y * ((float)zb * iz) + gv.z * ((float)yb * iy)) * sDh + gxs * (float)(xb + 4 * lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dh[r] = acc[r] > 0.f ? fmaf(gxs, (float)((r & 3) + 8 * (r >> 2)), G) : 0.f;
          db += dh[r];                                   // scaled by sDh: undone when the partial is written
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float4 c = sC[32 * vb + (r & 3) + 8 * (r >> 2) + 4 * lh];
          const float gd = gv.x + gv.y * c.x + gv.z * c.y + gv.w * c.z;
          dh[r] = (acc[r] * desc + bv > 0.f && c.w > 0.f) ? gd : 0.f;
          db += dh[r];
          dh[r] *= sDh;                                    // range scale of the gradient operand
        }
      }
      // dW[k, c] += sum_v dh[v, k] feat[v, c]: registers 8 s2 .. 8 s2 + 7 ARE the A fragment of K step s2
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 a[TERMS];
        split8<TERMS>(dh + 8 * s2, a);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          if (32 * ct < Cin) {                           // wave-uniform: no products against the zero padding of Cin <= 32
            bf16x8 b[TERMS];
#pragma unroll
            for (int t = 0; t < TERMS; ++t)
              b[t] = *reinterpret_cast<const bf16x8*>(sFT + t * (64 * 128) + swz(32 * ct + li, 4 * vb + 2 * s2 + lh));
            dw[ct] = mfma_split<TERMS>(a, b, dw[ct]);
          }
        }
      }
    }
    n = n_next;
    v0 = v_next;
    step(n_next, v_next);
    if (tile + 1 < t_end) {
      commit(tile + 1, v0);
      if (tile + 2 < t_end) feat_fetch<WVT, NT_>(pre, feat + (long long)n_next * V * Cin, v_next, V, Cin, tid);
    }
    __syncthreads();
  }
  float* ow = pw + (long long)slab * Cout * Cin;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = 32 * ct + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = grp * (32 * NWV) + 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (k < Cout && c < Cin) ow[(long long)k * Cin + c] = dw[ct][r] * desc_w;
    }
  }
  db += __shfl_xor(db, 32, 64);
  if (ROWS) db /= sDh;                                     // a power of two (1 for TERMS == 3)
  if (lh == 0 && co < Cout) pb[(long long)slab * Cout + co] = db;
}

// ---- dfeat: workgroup = 256 voxels (8 waves x 32), loops over 64-channel blocks of W --------------------
constexpr int BF_TPB = 512;
constexpr int WBLK = 64;                           // channels per staged weight block
template <int TERMS>
__global__ __launch_bounds__(BF_TPB, 2) void headcom_bwd_feat_bf_kernel(const float* __restrict__ feat,
                                                                        const __bf16* __restrict__ wkp,
                                                                        const __bf16* __restrict__ wt,
                                                                        const float* __restrict__ bias,
                                                                        const float* __restrict__ g,
                                                                        float* __restrict__ dfeat, long long V,
                                                                        int Cin, int Cout, int CoutP, Dims d,
                                                                        const float* __restrict__ hs, int mask_dfeat,
                                                                        unsigned* __restrict__ amax /* max |dfeat| bits | NULL */,
                                                                        const int* __restrict__ gate, int want) {
  if (gate && *gate != want) return;         // the other arithmetic runs this backward (head_dh_scale_kernel)
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  constexpr int IMG = TERMS * 64 * 128;            // one image of one block: [TERMS][64 rows][128 B]
  constexpr int BUF = 2 * IMG + WBLK * 32;         // wkp block + wt block + (g float4, bias) per channel
  constexpr int NLD = (2 * IMG / 16 + BF_TPB - 1) / BF_TPB;   // 16-byte items per thread per block
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y;
  const long long v = (long long)blockIdx.x * 256 + 32 * wv + li;      // this lane's voxel (column)
  const bool vok = v < V;
  const int nks = (Cin + 15) >> 4;
  const float* fn = feat + (long long)n * V * Cin;
  const float sFs = hs[0], desc = hs[1] * hs[3], sDh = hs[4], desc_f = hs[5] * hs[3];
  // coordinates of the lane's voxel
  float cz = 0.f, cy = 0.f, cx = 0.f;
  if (vok) {
    const int x = (int)(v % d.W), y = (int)((v / d.W) % d.H), z = (int)(v / ((long long)d.W * d.H));
    cz = d.D > 1 ? (float)z / (float)(d.D - 1) : 0.f;
    cy = d.H > 1 ? (float)y / (float)(d.H - 1) : 0.f;
    cx = d.W > 1 ? (float)x / (float)(d.W - 1) : 0.f;
  }
  // B fragments of the transposed logits GEMM: K step s, lane-half lh holds ci = 16 s + 8 (e >> 2) + 4 lh + (e & 3)
  bf16x8 fb[4][TERMS];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float x8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int c0 = 16 * s + 4 * lh;
    if (vok && c0 < Cin) {
      const float4 p = *reinterpret_cast<const float4*>(fn + v * Cin + c0);
      x8[0] = p.x; x8[1] = p.y; x8[2] = p.z; x8[3] = p.w;
    }
    if (vok && c0 + 8 < Cin) {
      const float4 p = *reinterpret_cast<const float4*>(fn + v * Cin + c0 + 8);
      x8[4] = p.x; x8[5] = p.y; x8[6] = p.z; x8[7] = p.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) x8[j] *= sFs;
    split8<TERMS>(x8, fb[s]);
  }
  f32x16 acc2[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;

  const int nblk = CoutP / WBLK;
  uint4 pre[NLD];
  auto fetch = [&](int blk) {                      // global -> registers (next block, in flight during the MFMAs)
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * BF_TPB;
      pre[k] = make_uint4(0u, 0u, 0u, 0u);
      if (i < 2 * IMG / 16) {
        const int chunk = i & 7, row = (i >> 3) & 63, t = (i >> 9) % TERMS, img = i / (TERMS * 512);
        const __bf16* src = img == 0 ? wkp + ((long long)t * CoutP + blk * WBLK + row) * 64 + chunk * 8
                                     : wt + ((long long)t * 64 + row) * CoutP + blk * WBLK + chunk * 8;
        pre[k] = *reinterpret_cast<const uint4*>(src);
      }
    }
  };
  auto commit = [&](int blk, unsigned char* buf) {  // registers -> LDS (+ g / bias of the block's channels)
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * BF_TPB;
      if (i < 2 * IMG / 16) {
        const int chunk = i & 7, row = (i >> 3) & 63, t = (i >> 9) % TERMS, img = i / (TERMS * 512);
        *reinterpret_cast<uint4*>(buf + img * IMG + t * (64 * 128) + swz(row, chunk)) = pre[k];
      }
    }
    if (tid < WBLK) {
      const int k = blk * WBLK + tid;
      float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
      float b = 0.f;
      if (k < Cout) {
        gq = *reinterpret_cast<const float4*>(g + ((long long)n * Cout + k) * 4);
        b = bias ? bias[k] : 0.f;
      }
      // staged pre-multiplied: the gradient coefficients by the operand's range scale, the bias by 1 / descale (the
      // logits accumulator starts there, so [h > 0] is the sign of the accumulator)
      float* sg = reinterpret_cast<float*>(buf + 2 * IMG) + tid * 8;
      sg[0] = gq.x * sDh; sg[1] = gq.y * sDh; sg[2] = gq.z * sDh; sg[3] = gq.w * sDh; sg[4] = b / desc;
    }
  };
  fetch(0);
  commit(0, hsm);
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    unsigned char* buf = hsm + (blk & 1) * BUF;
    if (blk + 1 < nblk) fetch(blk + 1);
    const unsigned char* sWk = buf;
    const unsigned char* sWt = buf + IMG;
    const float* sG = reinterpret_cast<const float*>(buf + 2 * IMG);
#pragma unroll
    for (int m = 0; m < 2; ++m) {                  // 32-channel M tile of the block
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = sG[(32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh) * 8 + 4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < nks) {
          bf16x8 a[TERMS];
#pragma unroll
          for (int t = 0; t < TERMS; ++t)
            a[t] = *reinterpret_cast<const bf16x8*>(sWk + t * (64 * 128) + swz(32 * m + li, 2 * s + lh));
          acc = mfma_split<TERMS>(a, fb[s], acc);
        }
      }
      // dh^T: register r = channel row, lane = voxel.  (A lane past the volume computes some dh of its own column only,
      // which is never stored.)
      float dh[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 gq = *reinterpret_cast<const float4*>(sG + (32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh) * 8);
        const float gd = fmaf(gq.w, cx, fmaf(gq.z, cy, fmaf(gq.y, cz, gq.x)));
        dh[r] = acc[r] > 0.f ? gd : 0.f;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 b[TERMS];
        split8<TERMS>(dh + 8 * s2, b);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (32 * mt < Cin) {                           // wave-uniform: rows ci >= Cin of W^T are zero padding
            bf16x8 a[TERMS];
#pragma unroll
            for (int t = 0; t < TERMS; ++t)
              a[t] = *reinterpret_cast<const bf16x8*>(sWt + t * (64 * 128) + swz(32 * mt + li, 4 * m + 2 * s2 + lh));
            acc2[mt] = mfma_split<TERMS>(a, b, acc2[mt]);
          }
        }
      }
    }
    if (blk + 1 < nblk) commit(blk + 1, hsm + ((blk + 1) & 1) * BUF);   // other buffer: its readers finished a block ago
    __syncthreads();
  }
  float mxo = 0.f;
  if (vok) {
    float* o = dfeat + ((long long)n * V + v) * Cin;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * mt + 8 * q + 4 * lh;
        if (c < Cin) {
          float4 r = make_float4(acc2[mt][4 * q] * desc_f, acc2[mt][4 * q + 1] * desc_f, acc2[mt][4 * q + 2] * desc_f,
                                 acc2[mt][4 * q + 3] * desc_f);
          if (mask_dfeat) {   // feat is a ReLU output: its producer's backward gets the gradient already masked
            const float4 f = *reinterpret_cast<const float4*>(fn + v * Cin + c);
            r.x = f.x > 0.f ? r.x : 0.f; r.y = f.y > 0.f ? r.y : 0.f; r.z = f.z > 0.f ? r.z : 0.f; r.w = f.w > 0.f ? r.w : 0.f;
          }
          *reinterpret_cast<float4*>(o + c) = r;
          mxo = fmaxf(fmaxf(mxo, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        }
      }
  }
  if (amax) kmh_absmax::publish(mxo, amax);      // the consumer convolution's range scale, for free
}

// =============================================================================================
// Mask variants of the two backward kernels (ROWS geometry: W % 32 == 0, V % FVT == 0).  The forward kernel stored
// [h > 0], so neither kernel recomputes the logits: no filter fragments / plain feature image, half the
// MFMAs, and dh is one fma on the block's row constants plus the mask.
//
// dW / db: lane = channel, register r = voxel row.  The lane's 32-bit word of the block holds its 16 voxels at bits
// (r & 3) + 8 (r >> 2) + 4 lh: v_bfe_i32 (0 / -1) + v_and per element.
// WIDE: Cin > 32 (compile time: a run-time test around the second accumulator tile's products makes the compiler shuffle
// whole accumulator tiles between registers).
template <int TERMS, int NWV, bool WIDE, bool AMP = false>
__global__ __launch_bounds__(64 * NWV, 2) void headcom_bwd_w_mask_kernel(
    const float* __restrict__ feat, const unsigned* __restrict__ mask, const float* __restrict__ g,
    float* __restrict__ pw /* (nslab, Cout, Cin) */, float* __restrict__ pb /* (nslab, Cout) */, int N, long long V,
    int Cin, int Cout, int CoutP, Dims d, int tiles_per_slab, int ngroups, const float* __restrict__ hs,
    const int* __restrict__ gate, int want) {
  if (gate && *gate != want) return;         // the other arithmetic runs this backward (head_dh_scale_kernel)
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  constexpr int WBUF = TERMS * 64 * 128;     // one buffer: sFT [TERMS][64 ci][128 B] (columns = sigma(voxel))
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = item % ngroups, slab = item / ngroups;
  constexpr int NT_ = 64 * NWV;
  const int co = grp * (32 * NWV) + 32 * wv + li;
  const float sFs = hs[0], sDh = hs[4], desc_w = hs[5] * hs[1];
  f32x16 dw[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[ct][r] = 0.f;
  kmh_f2 db2 = {0.f, 0.f};
  const long long tiles_per_n = V / WVT, ntiles = tiles_per_n * N;
  long long t_beg = (long long)slab * tiles_per_slab, t_end = t_beg + tiles_per_slab;
  if (t_end > ntiles) t_end = ntiles;
  FeatRegs<WVT, NT_> pre;                    // the next tile's features, in flight under this tile's MFMAs
  int n = t_beg < t_end ? (int)(t_beg / tiles_per_n) : 0;
  long long v0 = (t_beg - (long long)n * tiles_per_n) * WVT;
  const unsigned* mrow = mask + co;
  auto commit = [&](long long tile, long long vfirst) {
    unsigned char* b = hsm + (int)((tile - t_beg) & 1) * WBUF;
    feat_commit<TERMS, WVT, NT_, true, false>(pre, vfirst, V, d, nullptr, b, nullptr, tid, sFs, false);
  };
  auto mfetch = [&](int nn, long long vv, unsigned out[2]) {
    const long long nb = ((long long)nn * V + vv) >> 5;
    out[0] = mrow[nb * CoutP];
    out[1] = mrow[(nb + 1) * CoutP];
  };
  auto step = [&](int& nn, long long& vv) { vv += WVT; if (vv >= V) { vv = 0; ++nn; } };
  int n_next = n;
  long long v_next = v0;
  step(n_next, v_next);
  unsigned mcur[2] = {0u, 0u}, mnext[2] = {0u, 0u};
  if (t_beg < t_end) {
    feat_fetch<WVT, NT_>(pre, feat + (long long)n * V * Cin, v0, V, Cin, tid);
    mfetch(n, v0, mcur);
    commit(t_beg, v0);
    if (t_beg + 1 < t_end) {
      feat_fetch<WVT, NT_>(pre, feat + (long long)n_next * V * Cin, v_next, V, Cin, tid);
      mfetch(n_next, v_next, mnext);
    }
  }
  __syncthreads();
  RowCursor cur;
  cur.set(v0, d);
  const float iz = d.D > 1 ? 1.f / (float)(d.D - 1) : 0.f, iy = d.H > 1 ? 1.f / (float)(d.H - 1) : 0.f,
              ix = d.W > 1 ? 1.f / (float)(d.W - 1) : 0.f;
  for (long long tile = t_beg; tile < t_end; ++tile) {
    const unsigned char* sFT = hsm + (int)((tile - t_beg) & 1) * WBUF;
    const float4 gv = co < Cout ? *reinterpret_cast<const float4*>(g + ((long long)n * Cout + co) * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    const float gxs = gv.w * ix * sDh;                               // per x step, in the operand's range scale
#pragma unroll
    for (int vb = 0; vb < WVT / 32; ++vb) {
      const int xb = cur.x, yb = cur.y, zb = cur.z;                  // wave-uniform origin of block v0 + 32 vb
      cur.advance32(d);
      const float G = (gv.x + gv.y * ((float)zb * iz) + gv.z * ((float)yb * iy)) * sDh + gxs * (float)(xb + 4 * lh);
      const int w = (int)(mcur[vb] >> (4 * lh));
      float dh[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pos = (r & 3) + 8 * (r >> 2);
        unsigned m;                                                   // 0 or ~0 (v_bfe_i32; kept opaque: the compiler
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "n"(pos));     // would turn "& m" back into and + cmp + cndmask)
        dh[r] = __uint_as_float(__float_as_uint(fmaf(gxs, (float)pos, G)) & m);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) db2 += kmh_f2{dh[2 * j], dh[2 * j + 1]};     // scaled by sDh: undone when written
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 a[TERMS];
        split8<TERMS>(dh + 8 * s2, a);
#pragma unroll
        for (int ct = 0; ct < (WIDE ? 2 : 1); ++ct) {
          bf16x8 b[TERMS];
#pragma unroll
          for (int t = 0; t < TERMS; ++t)
            b[t] = *reinterpret_cast<const bf16x8*>(sFT + t * (64 * 128) + swz(32 * ct + li, 4 * vb + 2 * s2 + lh));
          dw[ct] = mfma_split<TERMS, AMP>(a, b, dw[ct]);
        }
      }
    }
    n = n_next;
    v0 = v_next;
    step(n_next, v_next);
    mcur[0] = mnext[0]; mcur[1] = mnext[1];
    if (tile + 1 < t_end) {
      commit(tile + 1, v0);
      if (tile + 2 < t_end) {
        feat_fetch<WVT, NT_>(pre, feat + (long long)n_next * V * Cin, v_next, V, Cin, tid);
        mfetch(n_next, v_next, mnext);
      }
    }
    __syncthreads();
  }
  float* ow = pw + (long long)slab * Cout * Cin;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = 32 * ct + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = grp * (32 * NWV) + 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (k < Cout && c < Cin) ow[(long long)k * Cin + c] = dw[ct][r] * desc_w;
    }
  }
  float db = db2.x + db2.y;
  db += __shfl_xor(db, 32, 64);
  db /= sDh;                                               // a power of two (1 for TERMS == 3)
  if (lh == 0 && co < Cout) pb[(long long)slab * Cout + co] = db;
}

// dfeat: workgroup = 256 voxels (8 waves x one 32-voxel block of an x row), loops over 64-channel blocks of W^T.
// lane = voxel, register r = channel row.  Per (wave, channel of the staged block) LDS holds {A, B, mask word}:
// the gradient coefficient is fma(B[k], cx, A[k]) with A[k] = (g0 + gz cz + gy cy) S_dh, the lane's bit of the word
// (v_bfe_i32: 0 / ~0) masks it -- one broadcast ds_read_b128 and three VALU per element.  The mask words are fetched
// with the filter block, one block ahead (registers), like the filter fragments themselves.
template <int TERMS, bool WIDE, bool AMP = false>
__global__ __launch_bounds__(BF_TPB, 2) void headcom_bwd_feat_mask_kernel(
    const float* __restrict__ feat, const __bf16* __restrict__ wt, const unsigned* __restrict__ mask,
    const float* __restrict__ g, float* __restrict__ dfeat, long long V, int Cin, int Cout, int CoutP, Dims d,
    const float* __restrict__ hs, int mask_dfeat, unsigned* __restrict__ amax, const int* __restrict__ gate, int want) {
  if (gate && *gate != want) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  constexpr int IMG = TERMS * 64 * 128;            // W^T block: [TERMS][64 ci][128 B]
  constexpr int BUF = IMG + 8 * WBLK * 16;         // + {A, B, mask word, -} per (wave, channel of the block)
  constexpr int NLD = (IMG / 16 + BF_TPB - 1) / BF_TPB;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y;
  const long long vw = (long long)blockIdx.x * 256 + 32 * wv;          // the wave's block (V % 32 == 0: whole waves)
  const bool wok = vw < V;
  const long long v = vw + li;
  const float* fn = feat + (long long)n * V * Cin;
  const float sDh = hs[4], desc_f = hs[5] * hs[3];
  float cz = 0.f, cy = 0.f, cx = 0.f;
  if (wok) {
    const unsigned u = (unsigned)vw, row = u / (unsigned)d.W, x0 = u - row * (unsigned)d.W;
    const unsigned z = row / (unsigned)d.H, y = row - z * (unsigned)d.H;
    cz = d.D > 1 ? (float)z / (float)(d.D - 1) : 0.f;
    cy = d.H > 1 ? (float)y / (float)(d.H - 1) : 0.f;
    cx = d.W > 1 ? (float)(x0 + li) / (float)(d.W - 1) : 0.f;
  }
  const unsigned* mrow = mask + ((((long long)n * V + (wok ? vw : 0)) >> 5) * CoutP) + lane;   // the wave's block
  f32x16 acc2[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;
  const int nblk = CoutP / WBLK;
  uint4 pre[NLD];
  unsigned mpre;
  auto fetch = [&](int blk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * BF_TPB;
      pre[k] = make_uint4(0u, 0u, 0u, 0u);
      if (i < IMG / 16) {
        const int chunk = i & 7, row = (i >> 3) & 63, t = i >> 9;
        pre[k] = *reinterpret_cast<const uint4*>(wt + ((long long)t * 64 + row) * CoutP + blk * WBLK + chunk * 8);
      }
    }
    mpre = mrow[blk * WBLK];                       // (a wave past the volume reads block 0's: never used)
  };
  auto commit = [&](int blk, unsigned char* buf) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * BF_TPB;
      if (i < IMG / 16) {
        const int chunk = i & 7, row = (i >> 3) & 63, t = i >> 9;
        *reinterpret_cast<uint4*>(buf + t * (64 * 128) + swz(row, chunk)) = pre[k];
      }
    }
    const int k = blk * WBLK + lane;
    float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < Cout) gq = *reinterpret_cast<const float4*>(g + ((long long)n * Cout + k) * 4);
    reinterpret_cast<float4*>(buf + IMG)[wv * WBLK + lane] =
        make_float4((gq.x + gq.y * cz + gq.z * cy) * sDh, gq.w * sDh, __uint_as_float(mpre), 0.f);
  };
  fetch(0);
  commit(0, hsm);
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    unsigned char* buf = hsm + (blk & 1) * BUF;
    if (blk + 1 < nblk) fetch(blk + 1);
    const unsigned char* sWt = buf;
    const float4* sAB = reinterpret_cast<const float4*>(buf + IMG) + wv * WBLK;
#pragma unroll
    for (int m = 0; m < 2; ++m) {                  // 32-channel tile of the block
      float dh[16];
      // the records in two batches of eight ds_read_b128, each batch issued before its first use: read by read (hipcc's
      // placement) the wave parks at ~12 LDS round trips per tile and the kernel runs at that latency.  (The empty asm
      // keeps each record one 128-bit read; the explicit v_fma keeps hipcc from packing pairs into v_pk_fma_f32 + moves.)
      typedef float kmh_f4 __attribute__((ext_vector_type(4)));
      const kmh_f4* sAB4 = reinterpret_cast<const kmh_f4*>(sAB);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        kmh_f4 ab[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) ab[r] = sAB4[32 * m + (r & 3) + 8 * ((8 * hb + r) >> 2) + 4 * lh];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          asm volatile("" : "+v"(ab[r]));
          unsigned mk;                             // the lane's voxel: bit li of the channel's word -> 0 / ~0
          float gd;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mk) : "v"(ab[r].z), "v"(li));
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(gd) : "v"(ab[r].y), "v"(cx), "v"(ab[r].x));
          dh[8 * hb + r] = __uint_as_float(__float_as_uint(gd) & mk);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 b[TERMS];
        split8<TERMS>(dh + 8 * s2, b);
#pragma unroll
        for (int mt = 0; mt < (WIDE ? 2 : 1); ++mt) {
          bf16x8 a[TERMS];
#pragma unroll
          for (int t = 0; t < TERMS; ++t)
            a[t] = *reinterpret_cast<const bf16x8*>(sWt + t * (64 * 128) + swz(32 * mt + li, 4 * m + 2 * s2 + lh));
          acc2[mt] = mfma_split<TERMS, AMP>(a, b, acc2[mt]);
        }
      }
    }
    if (blk + 1 < nblk) commit(blk + 1, hsm + ((blk + 1) & 1) * BUF);
    __syncthreads();
  }
  float mxo = 0.f;
  if (wok) {
    float* o = dfeat + ((long long)n * V + v) * Cin;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * mt + 8 * q + 4 * lh;
        if (c < Cin) {
          float4 r = make_float4(acc2[mt][4 * q] * desc_f, acc2[mt][4 * q + 1] * desc_f, acc2[mt][4 * q + 2] * desc_f,
                                 acc2[mt][4 * q + 3] * desc_f);
          if (mask_dfeat) {
            const float4 f = *reinterpret_cast<const float4*>(fn + v * Cin + c);
            r.x = f.x > 0.f ? r.x : 0.f; r.y = f.y > 0.f ? r.y : 0.f; r.z = f.z > 0.f ? r.z : 0.f; r.w = f.w > 0.f ? r.w : 0.f;
          }
          *reinterpret_cast<float4*>(o + c) = r;
          mxo = fmaxf(fmaxf(mxo, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        }
      }
  }
  if (amax) kmh_absmax::publish(mxo, amax);
}


static int fwd_slabs(long long V, int* tps) {
  const long long ntiles = (V + VT - 1) / VT;
  long long t = (ntiles + 255) / 256;
  if (t < 1) t = 1;
  *tps = (int)t;
  return (int)((ntiles + t - 1) / t);
}
static int bwdw_slabs(int N, long long V, int* tps) {
  const long long ntiles = ((V + VT - 1) / VT) * N;
  long long t = (ntiles + 255) / 256;
  if (t < 1) t = 1;
  *tps = (int)t;
  return (int)((ntiles + t - 1) / t);
}

}  // namespace

/* workspace sizes (bytes) */
KMH_API size_t kmh_headcom_fwd_ws_bytes(int N, long long V, int Cout) {
  int tps;
  return (size_t)N * Cout * fwd_slabs(V, &tps) * 5 * sizeof(double);
}
KMH_API size_t kmh_headcom_bwd_ws_bytes(int N, long long V, int Cin, int Cout) {
  int tps;
  const int ns = bwdw_slabs(N, V, &tps);
  return (size_t)N * Cout * 4 * sizeof(float) + (size_t)ns * 4 * ((size_t)Cout * Cin + Cout) * sizeof(float) + 256;
}

/* feat (N,V,Cin) NDHWC, w (Cout,Cin), bias (Cout)|NULL -> pts (N,Cout,3) (z,y,x) in [-1,1], sums (N,Cout,4) =
 * sum relu(h) (1, cz, cy, cx); sq (N,Cout)|NULL = sum relu(h)^2 (keypoint weighting, keymorph/model.py:75-109) */
KMH_API int kmh_headcom_fwd(const float* feat, const float* w, const float* bias, float* pts, float* sums, float* sq,
                            int N, int D, int H, int W, int Cin, int Cout, void* ws, void* stream) {
  if (Cin > 64) return -22;
  hipStream_t s = (hipStream_t)stream;
  const long long V = (long long)D * H * W;
  int tps;
  const int ns = fwd_slabs(V, &tps);
  Dims d{D, H, W};
  headcom_fwd_kernel<<<dim3(ns, ceil_div(Cout, GC), N), HTPB, 0, s>>>(feat, w, bias, (double*)ws, V, Cin, Cout, d, tps,
                                                                     ns);
  headcom_final_kernel<<<ceil_div(N * Cout, 64), 64, 0, s>>>((const double*)ws, ns, N * Cout, pts, sums, sq);
  return KMH_LAUNCH_CHECK();
}

/* dpts (N,Cout,3) [+ dpower (N,Cout)|NULL] -> dfeat (N,V,Cin), dw (Cout,Cin), dbias (Cout)|NULL; recomputes the logits */
KMH_API int kmh_headcom_bwd(const float* dpts, const float* dpower, const float* feat, const float* w, const float* bias,
                            const float* sums, float* dfeat, float* dw, float* dbias, int N, int D, int H, int W,
                            int Cin, int Cout, int mask_dfeat, void* ws, void* stream) {
  if (Cin > 64) return -22;
  hipStream_t s = (hipStream_t)stream;
  const long long V = (long long)D * H * W;
  Dims d{D, H, W};
  float* g = (float*)ws;
  int tps;
  const int ns = bwdw_slabs(N, V, &tps);
  float* pw = (float*)((char*)ws + (((size_t)N * Cout * 4 * sizeof(float) + 255) & ~(size_t)255));
  float* pb = pw + (size_t)ns * 4 * Cout * Cin;
  headcom_coef_kernel<<<ceil_div(N * Cout, 64), 64, 0, s>>>(dpts, dpower, sums, N * Cout, g);
  if (dfeat)
    headcom_bwd_feat_kernel<<<dim3(ceil_div(V, VT), N), HTPB, 0, s>>>(feat, w, bias, g, dfeat, V, Cin, Cout, d, mask_dfeat);
  if (dw) {
    headcom_bwd_w_kernel<<<dim3(ns, ceil_div(Cout, GC)), HTPB, 0, s>>>(feat, w, bias, g, pw, pb, N, V, Cin, Cout, d,
                                                                      tps);
    int nb = ceil_div((long long)Cout * Cin, 64);
    if (nb > 2048) nb = 2048;
    headcom_reduce_kernel<<<nb, 256, 0, s>>>(pw, ns * 4, (long long)Cout * Cin, dw);
    if (dbias) headcom_reduce_kernel<<<ceil_div(Cout, 64), 256, 0, s>>>(pb, ns * 4, Cout, dbias);
  }
  return KMH_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// split-bf16 entry points (terms = 3: fp32-class, the default; terms = 2: ~4e-6 relative)
namespace {
// hs[0..5] = 1 (TERMS == 3), or filled by the absmax launches / the dh bound below (TERMS == 2)
__global__ void head_scales_one_kernel(float* __restrict__ hs) {
  if (threadIdx.x < 8) hs[threadIdx.x] = 1.f;
}
// |dh| <= |g0| + |gz| + |gy| + |gx| (the normalised coordinates lie in [0, 1]): range scale of the head gradient
// Also decides which arithmetic runs the backward.  One power-of-two scale for all channels keeps a channel whose bound
// is b at a relative precision of max(2^-22, 2^-40 B / b) (B = the largest bound: below B 2^-18 the low fp16 term is
// denormal; measured on the worst rows it is ~2^6 worse than that, the values of a channel lying below its bound).  A
// live channel with a single faint voxel has b ~ 1 / mass: when any live channel sits more than 2^30 below B, gate = 1
// and the three-term bf16 kernels (8 exponent bits, no range scale) run instead of the split-fp16 ones -- both sets are
// launched, each returns at once unless the gate names it.  (2^30, not less: at a fresh initialisation with lambda = 0
// the bounds of the 2048 (sample, keypoint) channels already span 2^28 -- median 2^-20 of the largest -- because the
// keypoint gradients do; those weakest rows then carry ~1 % error on a gradient 2^-28 of the largest, which Adam's
// per-parameter normalisation does not see, and the step keeps its split-fp16 speed.)
__global__ __launch_bounds__(256) void head_dh_scale_kernel(const float* __restrict__ g, int NK, float* __restrict__ out2,
                                                            int* __restrict__ gate) {
  float m = 0.f;
  for (int i = threadIdx.x; i < NK; i += 256)
    m = fmaxf(m, fabsf(g[i * 4]) + fabsf(g[i * 4 + 1]) + fabsf(g[i * 4 + 2]) + fabsf(g[i * 4 + 3]));
  __shared__ float red[4];
  __shared__ int wide;
  if (threadIdx.x == 0) wide = 0;
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  const float B = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  if (threadIdx.x == 0) range_scale(B, out2);
  if (gate) {
    int w = 0;
    for (int i = threadIdx.x; i < NK; i += 256) {
      const float b = fabsf(g[i * 4]) + fabsf(g[i * 4 + 1]) + fabsf(g[i * 4 + 2]) + fabsf(g[i * 4 + 3]);
      w |= (b > 0.f && b < B * 9.3132257e-10f) ? 1 : 0;        // 2^-30
    }
    if (w) atomicOr(&wide, 1);
    __syncthreads();
    if (threadIdx.x == 0) *gate = wide;
  }
}
__global__ void head_scales_copy_kernel(const float* __restrict__ src, float* __restrict__ dst) {
  if (threadIdx.x < 4) dst[threadIdx.x] = src[threadIdx.x];
}
// cached (float[4] | NULL): the {S, 1/S} pairs of feat and w a forward call of the same graph measured
template <int TERMS>
static int head_scales(const float* feat, long long nfeat, const float* w, long long nw, float* hs, hipStream_t s,
                       const float* cached = nullptr) {
  head_scales_one_kernel<<<1, 64, 0, s>>>(hs);
  if (TERMS == 2 && cached) {
    head_scales_copy_kernel<<<1, 64, 0, s>>>(cached, hs);
    return KMH_LAUNCH_CHECK();
  }
  if (TERMS == 2) {
    int rc = kmh_absmax::launch(feat, nfeat, 0.f, hs, s);
    if (rc) return rc;
    rc = kmh_absmax::launch(w, nw, 0.f, hs + 2, s);
    if (rc) return rc;
  }
  return KMH_LAUNCH_CHECK();
}

// the ROWS geometry of the split-operand kernels: 32-voxel blocks are pieces of one x row, no padding voxels
static bool head_rows_ok(int D, int H, int W) {
  const long long V = (long long)D * H * W;
  return W % 32 == 0 && V % FVT == 0 && V < (1ll << 31);
}

struct HeadBfPlan {
  int CoutP, nwv, ngroups, nwv_w, ngroups_w, nslab_f, tps_f, nslab_w, tps_w;
  size_t img_bytes;     // the three pre-split weight images
};
static HeadBfPlan head_bf_plan(int N, long long V, int Cout, int terms) {
  HeadBfPlan p;
  p.CoutP = ceil_div(Cout, GC) * GC;
  p.nwv = (p.CoutP % 512 == 0) ? 16 : 4;            // waves (32-channel tiles) per workgroup
  p.ngroups = p.CoutP / (32 * p.nwv);
  p.nwv_w = (p.CoutP % 256 == 0) ? 8 : 4;           // dW kernel: its accumulators cap it at 8 waves per CU
  p.ngroups_w = p.CoutP / (32 * p.nwv_w);
  {
    const long long ntiles = (V + FVT - 1) / FVT;
    long long want = (p.nwv == 16 ? 256 : 1536) / ((long long)p.ngroups * (N > 0 ? N : 1));
    if (want < 1) want = 1;
    if (want > ntiles) want = ntiles;
    p.tps_f = (int)((ntiles + want - 1) / want);
    p.nslab_f = (int)((ntiles + p.tps_f - 1) / p.tps_f);
  }
  {
    const long long ntiles = ((V + WVT - 1) / WVT) * N;
    long long want = (p.nwv_w == 8 ? 512 : 1536) / p.ngroups_w;
    if (want < 1) want = 1;
    if (want > ntiles) want = ntiles;
    p.tps_w = (int)((ntiles + want - 1) / want);
    p.nslab_w = (int)((ntiles + p.tps_w - 1) / p.tps_w);
  }
  p.img_bytes = (size_t)3 * terms * p.CoutP * 64 * sizeof(__bf16);
  return p;
}
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <int TERMS>
static int head_pack(const float* w, int Cout, int Cin, const HeadBfPlan& p, void* img, const float* hs, hipStream_t s) {
  __bf16* wk = (__bf16*)img;
  __bf16* wkp = wk + (size_t)TERMS * p.CoutP * 64;
  __bf16* wt = wkp + (size_t)TERMS * p.CoutP * 64;
  headcom_pack_bf_kernel<TERMS><<<ceil_div(p.CoutP * 64, 256), 256, 0, s>>>(w, Cout, Cin, p.CoutP, wk, wkp, wt, hs);
  return KMH_LAUNCH_CHECK();
}

template <int TERMS>
static int head_fwd_bf(const float* feat, const float* w, const float* bias, float* pts, float* sums, float* sq, float* scales_out, int N,
                       int D, int H, int W, int Cin, int Cout, unsigned* mask, void* ws, hipStream_t s) {
  const long long V = (long long)D * H * W;
  const HeadBfPlan p = head_bf_plan(N, V, Cout, TERMS);
  double* partial = (double*)ws;
  void* img = (char*)ws + align256((size_t)N * Cout * p.nslab_f * 5 * sizeof(double));
  float* hs = (float*)((char*)img + align256(p.img_bytes));
  int rc = head_scales<TERMS>(feat, (long long)N * V * Cin, w, (long long)Cout * Cin, hs, s);
  if (rc) return rc;
  rc = head_pack<TERMS>(w, Cout, Cin, p, img, hs, s);
  if (rc) return rc;
  Dims d{D, H, W};
  const size_t lds = 2 * ((size_t)TERMS * FVT * 128 + FVT * sizeof(float4));   // double buffered
  const bool rows = head_rows_ok(D, H, W);
  if (mask && !rows) return -22;             // kmh_headcom_mask_words said 0 for this geometry
  const bool amp = TERMS == 2 && kmh_amp_enabled();      // use_amp: one product (the kernels that keep the sign mask)
  auto kern = p.nwv == 16 ? (rows ? (amp ? headcom_fwd_bf_kernel<TERMS, true, 16, true> : headcom_fwd_bf_kernel<TERMS, true, 16>)
                                  : headcom_fwd_bf_kernel<TERMS, false, 16>)
                          : (rows ? (amp ? headcom_fwd_bf_kernel<TERMS, true, 4, true> : headcom_fwd_bf_kernel<TERMS, true, 4>)
                                  : headcom_fwd_bf_kernel<TERMS, false, 4>);
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  kern<<<dim3(p.nslab_f * p.ngroups, N), 64 * p.nwv, lds, s>>>(feat, (const __bf16*)img, bias, partial, V, Cin, Cout, p.CoutP, d,
                                                         p.tps_f, p.nslab_f, p.ngroups, hs, sq != nullptr, mask);
  headcom_final_kernel<<<ceil_div(N * Cout, 64), 64, 0, s>>>(partial, p.nslab_f, N * Cout, pts, sums, sq);
  if (scales_out) head_scales_copy_kernel<<<1, 64, 0, s>>>(hs, scales_out);
  return KMH_LAUNCH_CHECK();
}

template <int TERMS>
static int head_bwd_bf(const float* dpts, const float* dpower, const float* feat, const float* w, const float* bias, const float* sums,
                       float* dfeat, float* dw, float* dbias, int N, int D, int H, int W, int Cin, int Cout,
                       int mask_dfeat, const float* scales_in, float* dfeat_scale2, const unsigned* mask, void* ws,
                       hipStream_t s) {
  const long long V = (long long)D * H * W;
  if (mask && !head_rows_ok(D, H, W)) return -22;
  const HeadBfPlan p = head_bf_plan(N, V, Cout, TERMS);
  char* base = (char*)ws;
  float* g = (float*)base;
  base += align256((size_t)N * Cout * 4 * sizeof(float));
  void* img = base;
  base += align256(p.img_bytes);
  float* pw = (float*)base;
  float* pb = pw + (size_t)p.nslab_w * Cout * Cin;
  float* hs = (float*)((char*)pb + align256((size_t)p.nslab_w * Cout * sizeof(float)));
  // TERMS == 2 only: the three-term images / unit scales / gate of the wide-dynamic-range fallback
  const HeadBfPlan p3 = head_bf_plan(N, V, Cout, 3);
  void* img3 = (char*)hs + 256;
  float* hs3 = (float*)((char*)img3 + align256(p3.img_bytes));
  int* gate = TERMS == 2 ? (int*)((char*)hs3 + 128) : nullptr;
  int rc = head_scales<TERMS>(feat, (long long)N * V * Cin, w, (long long)Cout * Cin, hs, s, scales_in);
  if (rc) return rc;
  rc = head_pack<TERMS>(w, Cout, Cin, p, img, hs, s);
  if (rc) return rc;
  Dims d{D, H, W};
  headcom_coef_kernel<<<ceil_div(N * Cout, 64), 64, 0, s>>>(dpts, dpower, sums, N * Cout, g);
  if (TERMS == 2) head_dh_scale_kernel<<<1, 256, 0, s>>>(g, N * Cout, hs + 4, gate);

  auto launch = [&](auto tag, const HeadBfPlan& pl, const void* img_, const float* hs_, int want) -> int {
    constexpr int T = decltype(tag)::value;
    const __bf16* wk = (const __bf16*)img_;
    const __bf16* wkp = wk + (size_t)T * pl.CoutP * 64;
    const __bf16* wt = wkp + (size_t)T * pl.CoutP * 64;
    if (dfeat && mask) {
      const size_t lds = 2 * ((size_t)T * 64 * 128 + 8 * WBLK * 16);
      const bool amp = T == 2 && kmh_amp_enabled();
      auto kern = Cin > 32 ? (amp ? headcom_bwd_feat_mask_kernel<T, true, true> : headcom_bwd_feat_mask_kernel<T, true>)
                           : (amp ? headcom_bwd_feat_mask_kernel<T, false, true> : headcom_bwd_feat_mask_kernel<T, false>);
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      kern<<<dim3(ceil_div(V, 256), N), BF_TPB, lds, s>>>(
          feat, wt, mask, g, dfeat, V, Cin, Cout, pl.CoutP, d, hs_, mask_dfeat, reinterpret_cast<unsigned*>(dfeat_scale2),
          gate, want);
    } else if (dfeat) {
      const size_t lds = 2 * ((size_t)2 * T * 64 * 128 + WBLK * 32);
      hipError_t e = hipFuncSetAttribute((const void*)headcom_bwd_feat_bf_kernel<T>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      headcom_bwd_feat_bf_kernel<T><<<dim3(ceil_div(V, 256), N), BF_TPB, lds, s>>>(
          feat, wkp, wt, bias, g, dfeat, V, Cin, Cout, pl.CoutP, d, hs_, mask_dfeat,
          reinterpret_cast<unsigned*>(dfeat_scale2), gate, want);
    }
    if (dw && mask) {
      const size_t lds = 2 * ((size_t)T * 64 * 128);                               // double buffered
      const bool amp = T == 2 && kmh_amp_enabled();
      auto kern = pl.nwv_w == 8 ? (Cin > 32 ? (amp ? headcom_bwd_w_mask_kernel<T, 8, true, true> : headcom_bwd_w_mask_kernel<T, 8, true>)
                                            : (amp ? headcom_bwd_w_mask_kernel<T, 8, false, true> : headcom_bwd_w_mask_kernel<T, 8, false>))
                                : (Cin > 32 ? (amp ? headcom_bwd_w_mask_kernel<T, 4, true, true> : headcom_bwd_w_mask_kernel<T, 4, true>)
                                            : (amp ? headcom_bwd_w_mask_kernel<T, 4, false, true> : headcom_bwd_w_mask_kernel<T, 4, false>));
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      kern<<<dim3(pl.nslab_w * pl.ngroups_w), 64 * pl.nwv_w, lds, s>>>(feat, mask, g, pw, pb, N, V, Cin, Cout, pl.CoutP, d,
                                                                      pl.tps_w, pl.ngroups_w, hs_, gate, want);
    } else if (dw) {
      const size_t lds = 2 * ((size_t)2 * T * WVT * 128 + WVT * sizeof(float4));   // double buffered
      const bool rows = W % 32 == 0 && V % WVT == 0 && V < (1ll << 31);
      auto kern = pl.nwv_w == 8 ? (rows ? headcom_bwd_w_bf_kernel<T, true, 8> : headcom_bwd_w_bf_kernel<T, false, 8>)
                                : (rows ? headcom_bwd_w_bf_kernel<T, true, 4> : headcom_bwd_w_bf_kernel<T, false, 4>);
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      kern<<<dim3(pl.nslab_w * pl.ngroups_w), 64 * pl.nwv_w, lds, s>>>(feat, wk, bias, g, pw, pb, N, V, Cin, Cout, pl.CoutP,
                                                                      d, pl.tps_w, pl.ngroups_w, hs_, gate, want);
    }
    return KMH_LAUNCH_CHECK();
  };
  rc = launch(std::integral_constant<int, TERMS>{}, p, img, hs, 0);
  if (rc) return rc;
  if (TERMS == 2) {
    head_scales_one_kernel<<<1, 64, 0, s>>>(hs3);
    rc = head_pack<3>(w, Cout, Cin, p3, img3, hs3, s);
    if (rc) return rc;
    rc = launch(std::integral_constant<int, 3>{}, p3, img3, hs3, 1);
    if (rc) return rc;
  }
  if (dfeat && dfeat_scale2) kmh_absmax::final_kernel<<<1, 1, 0, s>>>(dfeat_scale2, 0.f);
  if (dw) {
    int nb = ceil_div((long long)Cout * Cin, 64);
    if (nb > 2048) nb = 2048;
    headcom_reduce_kernel<<<nb, 256, 0, s>>>(pw, p.nslab_w, (long long)Cout * Cin, dw);
    if (dbias) headcom_reduce_kernel<<<ceil_div(Cout, 64), 256, 0, s>>>(pb, p.nslab_w, Cout, dbias);
  }
  return KMH_LAUNCH_CHECK();
}
}  // namespace

KMH_API size_t kmh_headcom_fwd_bf_ws_bytes(int N, long long V, int Cout, int terms) {
  const HeadBfPlan p = head_bf_plan(N, V, Cout, terms);
  return align256((size_t)N * Cout * p.nslab_f * 5 * sizeof(double)) + align256(p.img_bytes) + 256;
}
KMH_API size_t kmh_headcom_bwd_bf_ws_bytes(int N, long long V, int Cin, int Cout, int terms) {
  const HeadBfPlan p = head_bf_plan(N, V, Cout, terms);
  const HeadBfPlan p3 = head_bf_plan(N, V, Cout, 3);       // the fallback's images, unit scales and gate (terms == 2)
  return align256((size_t)N * Cout * 4 * sizeof(float)) + align256(p.img_bytes) +
         (size_t)p.nslab_w * (size_t)Cout * Cin * sizeof(float) + align256((size_t)p.nslab_w * Cout * sizeof(float)) + 512 +
         align256(p3.img_bytes) + 512;
}

/* same contracts as kmh_headcom_fwd / kmh_headcom_bwd; Cin % 4 == 0, Cin <= 64 */
KMH_API size_t kmh_headcom_mask_words(int N, int D, int H, int W, int Cout) {
  if (!head_rows_ok(D, H, W)) return 0;
  return (size_t)N * ((size_t)D * H * W / 32) * (size_t)(ceil_div(Cout, GC) * GC);
}
KMH_API int kmh_headcom_fwd_bf(const float* feat, const float* w, const float* bias, float* pts, float* sums, float* sq,
                               float* scales_out, int N, int D, int H, int W, int Cin, int Cout, int terms, unsigned* mask,
                               void* ws, void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: the fp16 kernels with hi x hi only (use_amp), for this call
  if (Cin > 64 || (Cin & 3) || (terms != 2 && terms != 3)) return -22;
  return terms == 3
             ? head_fwd_bf<3>(feat, w, bias, pts, sums, sq, scales_out, N, D, H, W, Cin, Cout, mask, ws, (hipStream_t)stream)
             : head_fwd_bf<2>(feat, w, bias, pts, sums, sq, scales_out, N, D, H, W, Cin, Cout, mask, ws, (hipStream_t)stream);
}
KMH_API int kmh_headcom_bwd_bf(const float* dpts, const float* dpower, const float* feat, const float* w, const float* bias,
                               const float* sums, float* dfeat, float* dw, float* dbias, int N, int D, int H, int W,
                               int Cin, int Cout, int terms, int mask_dfeat, const float* scales_in,
                               float* dfeat_scale2, const unsigned* mask, void* ws, void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: the fp16 kernels with hi x hi only (use_amp), for this call
  if (Cin > 64 || (Cin & 3) || (terms != 2 && terms != 3)) return -22;
  return terms == 3 ? head_bwd_bf<3>(dpts, dpower, feat, w, bias, sums, dfeat, dw, dbias, N, D, H, W, Cin, Cout,
                                     mask_dfeat, scales_in, dfeat_scale2, mask, ws, (hipStream_t)stream)
                    : head_bwd_bf<2>(dpts, dpower, feat, w, bias, sums, dfeat, dw, dbias, N, D, H, W, Cin, Cout,
                                     mask_dfeat, scales_in, dfeat_scale2, mask, ws, (hipStream_t)stream);
}
