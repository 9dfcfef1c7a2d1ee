// Trilinear / nearest 3-D sampler (ATen grid_sampler_3d semantics: padding_mode=border,
// align_corners=False) forward + backward, and the MSE / Dice reductions that follow it.
// Replaces keymorph/utils.py:14-21 (align_img) and keymorph/loss_ops.py:9-63.
//
// HBM-bound: per output voxel 12 B of grid + 4 B out (+ 8 gathers per channel that hit
// L2 / Infinity Cache because neighbouring voxels sample neighbouring texels).  Each thread
// owns VPT=4 consecutive output voxels so the grid is read as 3 x 16-B loads and the output
// written as one 16-B store per channel.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int VPT = 4;      // voxels per thread
constexpr int TPB = 256;    // threads per block

struct Tap {
  int x0, y0, z0;        // floor corner
  float fx, fy, fz;      // fractional offsets
  float mx, my, mz;      // d(ix)/d(gx) incl. clamp mask (W/2 or 0)
};

__device__ __forceinline__ float unnorm_clip(float g, int size, float& mult) {
  // ((g+1)*size-1)/2 then clip_coordinates_set_grad: borders count as out of bounds for the grad
  float v = ((g + 1.f) * (float)size - 1.f) * 0.5f;
  float hi = (float)(size - 1);
  if (v <= 0.f) { mult = 0.f; return 0.f; }
  if (v >= hi) { mult = 0.f; return hi; }
  mult = 0.5f * (float)size;
  return v;
}

__device__ __forceinline__ Tap make_tap(float gx, float gy, float gz, int D, int H, int W) {
  Tap t;
  float ix = unnorm_clip(gx, W, t.mx);
  float iy = unnorm_clip(gy, H, t.my);
  float iz = unnorm_clip(gz, D, t.mz);
  float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
  t.x0 = (int)fx0; t.y0 = (int)fy0; t.z0 = (int)fz0;
  t.fx = ix - fx0; t.fy = iy - fy0; t.fz = iz - fz0;
  return t;
}

// 8 corner values of one channel plane; corners past the far border contribute 0 (weight is 0 there)
__device__ __forceinline__ void gather8(const float* __restrict__ p, const Tap& t, int D, int H, int W,
                                        float v[8]) {
  const int x1 = t.x0 + 1 < W ? t.x0 + 1 : t.x0;
  const int y1 = t.y0 + 1 < H ? t.y0 + 1 : t.y0;
  const int z1 = t.z0 + 1 < D ? t.z0 + 1 : t.z0;
  const float ox = t.x0 + 1 < W ? 1.f : 0.f, oy = t.y0 + 1 < H ? 1.f : 0.f, oz = t.z0 + 1 < D ? 1.f : 0.f;
  const long long r00 = ((long long)t.z0 * H + t.y0) * W, r01 = ((long long)t.z0 * H + y1) * W;
  const long long r10 = ((long long)z1 * H + t.y0) * W, r11 = ((long long)z1 * H + y1) * W;
  v[0] = p[r00 + t.x0];
  v[1] = p[r00 + x1] * ox;
  v[2] = p[r01 + t.x0] * oy;
  v[3] = p[r01 + x1] * (ox * oy);
  v[4] = p[r10 + t.x0] * oz;
  v[5] = p[r10 + x1] * (ox * oz);
  v[6] = p[r11 + t.x0] * (oy * oz);
  v[7] = p[r11 + x1] * (ox * oy * oz);
}

__device__ __forceinline__ float blend8(const float v[8], const Tap& t) {
  const float ax = 1.f - t.fx, ay = 1.f - t.fy, az = 1.f - t.fz;
  // same association as ATen: value * (wx*wy*wz) summed corner by corner -- as ONE explicit fma chain, so that every
  // instantiation of every sampler kernel rounds identically (left to -ffp-contract=fast the fused and the plain warp
  // differed by 1 ulp in 13 % of the voxels)
  float o = v[0] * (ax * ay * az);
  o = fmaf(v[1], t.fx * ay * az, o);
  o = fmaf(v[2], ax * t.fy * az, o);
  o = fmaf(v[3], t.fx * t.fy * az, o);
  o = fmaf(v[4], ax * ay * t.fz, o);
  o = fmaf(v[5], t.fx * ay * t.fz, o);
  o = fmaf(v[6], ax * t.fy * t.fz, o);
  o = fmaf(v[7], t.fx * t.fy * t.fz, o);
  return o;
}

__device__ __forceinline__ void load_grid4(const float* __restrict__ grid, long long v0, long long nvox,
                                           bool full, float g[VPT][3]) {
  // 12 floats = 3 x float4 when the whole quad is in range and the sample base is 16-B aligned
  if (full) {
    const float4* gp = reinterpret_cast<const float4*>(grid + v0 * 3);
    float4 a = gp[0], b = gp[1], c = gp[2];
    g[0][0] = a.x; g[0][1] = a.y; g[0][2] = a.z;
    g[1][0] = a.w; g[1][1] = b.x; g[1][2] = b.y;
    g[2][0] = b.z; g[2][1] = b.w; g[2][2] = c.x;
    g[3][0] = c.y; g[3][1] = c.z; g[3][2] = c.w;
  } else {
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) g[i][k] = (v0 + i < nvox) ? grid[(v0 + i) * 3 + k] : 0.f;
  }
}

// ----------------------------------------------------------------------------------------------
template <int MODE, bool FUSE_MSE>
__global__ __launch_bounds__(TPB) void sample_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ out,
    const float* __restrict__ fixed, double* __restrict__ partial, int C, int D, int H, int W,
    long long ovox /* Do*Ho*Wo */) {
  const int n = blockIdx.y;
  const long long v0 = ((long long)blockIdx.x * TPB + threadIdx.x) * VPT;
  float acc = 0.f;
  if (v0 < ovox) {
    float g[VPT][3];
    const bool full = (v0 + VPT <= ovox) && ((ovox & 3) == 0);
    load_grid4(grid + (long long)n * ovox * 3, v0, ovox, full, g);
    Tap t[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) t[i] = make_tap(g[i][0], g[i][1], g[i][2], D, H, W);
    const long long plane = (long long)D * H * W;
    for (int c = 0; c < C; ++c) {
      const float* p = x + ((long long)n * C + c) * plane;
      float o[VPT];
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        if (MODE == 0) {
          float v[8];
          gather8(p, t[i], D, H, W, v);
          o[i] = blend8(v, t[i]);
        } else {
          // nearest: nearbyint (half to even) of the clipped coordinate
          int xn = (int)rintf((float)t[i].x0 + t[i].fx);
          int yn = (int)rintf((float)t[i].y0 + t[i].fy);
          int zn = (int)rintf((float)t[i].z0 + t[i].fz);
          o[i] = p[((long long)zn * H + yn) * W + xn];
        }
      }
      const long long ob = ((long long)n * C + c) * ovox + v0;
      if (FUSE_MSE) {
        if (full) {
          float4 f = *reinterpret_cast<const float4*>(fixed + ob);
          float d0 = o[0] - f.x, d1 = o[1] - f.y, d2 = o[2] - f.z, d3 = o[3] - f.w;
          acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        } else {
#pragma unroll
          for (int i = 0; i < VPT; ++i)
            if (v0 + i < ovox) { float d = o[i] - fixed[ob + i]; acc += d * d; }
        }
      }
      if (full) {
        *reinterpret_cast<float4*>(out + ob) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int i = 0; i < VPT; ++i)
          if (v0 + i < ovox) out[ob + i] = o[i];
      }
    }
  }
  if (FUSE_MSE) {
    __shared__ double red[TPB / kWave];
    double s = block_sum<double>((double)acc, red);
    if (threadIdx.x == 0) partial[(long long)blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(TPB) void sample_bwd_grid_kernel(
    const float* __restrict__ x, const float* __restrict__ grid, const float* __restrict__ gout,
    float* __restrict__ dgrid, int C, int D, int H, int W, long long ovox) {
  const int n = blockIdx.y;
  const long long v0 = ((long long)blockIdx.x * TPB + threadIdx.x) * VPT;
  if (v0 >= ovox) return;
  float g[VPT][3];
  const bool full = (v0 + VPT <= ovox) && ((ovox & 3) == 0);
  load_grid4(grid + (long long)n * ovox * 3, v0, ovox, full, g);
  Tap t[VPT];
  float gx[VPT], gy[VPT], gz[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    t[i] = make_tap(g[i][0], g[i][1], g[i][2], D, H, W);
    gx[i] = gy[i] = gz[i] = 0.f;
  }
  const long long plane = (long long)D * H * W;
  for (int c = 0; c < C; ++c) {
    const float* p = x + ((long long)n * C + c) * plane;
    const long long ob = ((long long)n * C + c) * ovox + v0;
    float go[VPT];
    if (full) {
      float4 q = *reinterpret_cast<const float4*>(gout + ob);
      go[0] = q.x; go[1] = q.y; go[2] = q.z; go[3] = q.w;
    } else {
#pragma unroll
      for (int i = 0; i < VPT; ++i) go[i] = (v0 + i < ovox) ? gout[ob + i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      float v[8];
      gather8(p, t[i], D, H, W, v);
      const float fx = t[i].fx, fy = t[i].fy, fz = t[i].fz;
      const float ax = 1.f - fx, ay = 1.f - fy, az = 1.f - fz;
      // d/dix, d/diy, d/diz of the trilinear blend (ATen grid_sampler_3d_backward)
      float dx = -v[0] * (ay * az) + v[1] * (ay * az) - v[2] * (fy * az) + v[3] * (fy * az)
                 - v[4] * (ay * fz) + v[5] * (ay * fz) - v[6] * (fy * fz) + v[7] * (fy * fz);
      float dy = -v[0] * (ax * az) - v[1] * (fx * az) + v[2] * (ax * az) + v[3] * (fx * az)
                 - v[4] * (ax * fz) - v[5] * (fx * fz) + v[6] * (ax * fz) + v[7] * (fx * fz);
      float dz = -v[0] * (ax * ay) - v[1] * (fx * ay) - v[2] * (ax * fy) - v[3] * (fx * fy)
                 + v[4] * (ax * ay) + v[5] * (fx * ay) + v[6] * (ax * fy) + v[7] * (fx * fy);
      gx[i] += dx * go[i];
      gy[i] += dy * go[i];
      gz[i] += dz * go[i];
    }
  }
  float* dg = dgrid + ((long long)n * ovox + v0) * 3;
  float r[VPT * 3];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    r[i * 3 + 0] = gx[i] * t[i].mx;
    r[i * 3 + 1] = gy[i] * t[i].my;
    r[i * 3 + 2] = gz[i] * t[i].mz;
  }
  if (full) {
    float4* d4 = reinterpret_cast<float4*>(dg);
    d4[0] = make_float4(r[0], r[1], r[2], r[3]);
    d4[1] = make_float4(r[4], r[5], r[6], r[7]);
    d4[2] = make_float4(r[8], r[9], r[10], r[11]);
  } else {
#pragma unroll
    for (int i = 0; i < VPT; ++i)
      if (v0 + i < ovox) { dg[i * 3] = r[i * 3]; dg[i * 3 + 1] = r[i * 3 + 1]; dg[i * 3 + 2] = r[i * 3 + 2]; }
  }
}

// ----------------------------------------------------------------------------------------------
// Lane-contiguous variants (the ones the launchers use whenever W >= 2 and a channel plane has < 2^31 voxels).
// One voxel per lane per pass, so one gather instruction covers 64 NEIGHBOURING voxels (2-4 cache lines for a
// smooth grid instead of 8+), the two x-corners of a row come from ONE 8-byte load, all in-plane offsets are
// 32-bit, and the AoS grid / grid-gradient rows go through LDS so their global accesses are 16-byte coalesced.
constexpr int PASSES = 4;   // 256-voxel passes per workgroup

struct Tap32 {
  int r00, r01, r10, r11;   // row offsets (z, y) inside one channel plane, + the pair base xb
  bool sel;                 // x0 is the last column: the pair was loaded one to the left
  float oy, oz;             // 0 when the +1 corner is past the far border
  float fx, fy, fz;
};

__device__ __forceinline__ Tap32 make_tap32(const Tap& t, int D, int H, int W) {
  Tap32 q;
  const int y1 = t.y0 + 1 < H ? t.y0 + 1 : t.y0, z1 = t.z0 + 1 < D ? t.z0 + 1 : t.z0;
  q.sel = t.x0 > W - 2;
  const int xb = q.sel ? W - 2 : t.x0;
  q.r00 = (t.z0 * H + t.y0) * W + xb; q.r01 = (t.z0 * H + y1) * W + xb;
  q.r10 = (z1 * H + t.y0) * W + xb;   q.r11 = (z1 * H + y1) * W + xb;
  q.oy = t.y0 + 1 < H ? 1.f : 0.f; q.oz = t.z0 + 1 < D ? 1.f : 0.f;
  q.fx = t.fx; q.fy = t.fy; q.fz = t.fz;
  return q;
}

__device__ __forceinline__ void load_pair(const float* __restrict__ p, bool sel, float& lo, float& hi) {
  float2 r;
  __builtin_memcpy(&r, p, sizeof(float2));      // 4-byte aligned 8-byte load (global_load_dwordx2)
  lo = sel ? r.y : r.x;
  hi = sel ? 0.f : r.y;
}

__device__ __forceinline__ void gather8_pairs(const float* __restrict__ p, const Tap32& q, float v[8]) {
  load_pair(p + q.r00, q.sel, v[0], v[1]);
  load_pair(p + q.r01, q.sel, v[2], v[3]);
  load_pair(p + q.r10, q.sel, v[4], v[5]);
  load_pair(p + q.r11, q.sel, v[6], v[7]);
  v[2] *= q.oy; v[3] *= q.oy; v[4] *= q.oz; v[5] *= q.oz;
  v[6] *= q.oy * q.oz; v[7] *= q.oy * q.oz;
}

// the workgroup's PASSES*256 grid rows (x, y, z) -> LDS, 16-byte coalesced
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int cnt, float* sg, int tid) {
  if (cnt == TPB * PASSES && ((reinterpret_cast<unsigned long long>(src) & 15) == 0)) {
#pragma unroll
    for (int k = 0; k < PASSES * 3 / 4; ++k)
      reinterpret_cast<float4*>(sg)[tid + k * TPB] = reinterpret_cast<const float4*>(src)[tid + k * TPB];
  } else {
    for (int e = tid; e < cnt * 3; e += TPB) sg[e] = src[e];
  }
}
__device__ __forceinline__ void unstage_rows(float* __restrict__ dst, int cnt, const float* sg, int tid) {
  if (cnt == TPB * PASSES && ((reinterpret_cast<unsigned long long>(dst) & 15) == 0)) {
#pragma unroll
    for (int k = 0; k < PASSES * 3 / 4; ++k)
      reinterpret_cast<float4*>(dst)[tid + k * TPB] = reinterpret_cast<const float4*>(sg)[tid + k * TPB];
  } else {
    for (int e = tid; e < cnt * 3; e += TPB) dst[e] = sg[e];
  }
}

// FUSE_GRAD (with FUSE_MSE): the loss is mean((out - fixed)^2), whose cotangent 2 (out - fixed) / count is known right
// here, so the same pass also produces d(loss)/d(grid) -- the rows of the staged grid are overwritten with it and written
// out like the grid came in.  One launch and 36 B per voxel instead of three (warp, MSE backward, grid backward) and 68.
template <int MODE, bool FUSE_MSE, bool FUSE_GRAD = false>
__global__ __launch_bounds__(TPB) void sample_fwd_lc_kernel(
    const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ out,
    const float* __restrict__ fixed, double* __restrict__ partial, int C, int D, int H, int W, long long ovox,
    float* __restrict__ dgrid = nullptr, float gcoef = 0.f /* 2 / (N C voxels) */) {
  __shared__ __attribute__((aligned(16))) float sg[TPB * PASSES * 3];
  const int n = blockIdx.y, tid = threadIdx.x;
  // (one contiguous chunk range per XCD -- xcd_remap of the block index -- measured slower: DESIGN.md section 8, round 4)
  const long long vb = (long long)blockIdx.x * (TPB * PASSES);
  const int cnt = ovox - vb < TPB * PASSES ? (int)(ovox - vb) : TPB * PASSES;
  stage_rows(grid + ((long long)n * ovox + vb) * 3, cnt, sg, tid);
  __syncthreads();
  const long long plane = (long long)D * H * W;
  constexpr int ILP = 2;       // voxels whose 4 pair-gathers are in flight together; 2 keeps the kernel at ~64 VGPRs
  float acc = 0.f;
#pragma unroll 1
  for (int j0 = 0; j0 < PASSES; j0 += ILP) {
    Tap t[ILP];
    Tap32 q[ILP];
    int near[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int l = tid + (j0 + u) * TPB;             // lane-contiguous: voxel vb + l
      t[u] = make_tap(sg[l * 3], sg[l * 3 + 1], sg[l * 3 + 2], D, H, W);
      q[u] = make_tap32(t[u], D, H, W);
      if (l >= cnt) { q[u].r00 = q[u].r01 = q[u].r10 = q[u].r11 = 0; q[u].sel = false; }   // any valid address
      near[u] = 0;
      if (MODE != 0 && l < cnt) {
        const int xn = (int)rintf((float)t[u].x0 + t[u].fx), yn = (int)rintf((float)t[u].y0 + t[u].fy),
                  zn = (int)rintf((float)t[u].z0 + t[u].fz);
        near[u] = (zn * H + yn) * W + xn;
      }
    }
    float ggx[ILP], ggy[ILP], ggz[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) ggx[u] = ggy[u] = ggz[u] = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* p = x + ((long long)n * C + c) * plane;
      const long long ob = ((long long)n * C + c) * ovox + vb;
      float o[ILP], fv[ILP];
      if (FUSE_MSE) {
#pragma unroll
        for (int u = 0; u < ILP; ++u) fv[u] = (tid + (j0 + u) * TPB < cnt) ? fixed[ob + tid + (j0 + u) * TPB] : 0.f;
      }
      if (MODE == 0) {
        float v[ILP][8];
#pragma unroll
        for (int u = 0; u < ILP; ++u) gather8_pairs(p, q[u], v[u]);
#pragma unroll
        for (int u = 0; u < ILP; ++u) o[u] = blend8(v[u], t[u]);
        if (FUSE_GRAD) {
#pragma unroll
          for (int u = 0; u < ILP; ++u) {
            const float fx = t[u].fx, fy = t[u].fy, fz = t[u].fz;
            const float ax = 1.f - fx, ay = 1.f - fy, az = 1.f - fz;
            const float* w = v[u];
            // d/dix, d/diy, d/diz of the trilinear blend (ATen grid_sampler_3d_backward), as sample_bwd_grid_lc_kernel
            const float dx = -w[0] * (ay * az) + w[1] * (ay * az) - w[2] * (fy * az) + w[3] * (fy * az)
                             - w[4] * (ay * fz) + w[5] * (ay * fz) - w[6] * (fy * fz) + w[7] * (fy * fz);
            const float dy = -w[0] * (ax * az) - w[1] * (fx * az) + w[2] * (ax * az) + w[3] * (fx * az)
                             - w[4] * (ax * fz) - w[5] * (fx * fz) + w[6] * (ax * fz) + w[7] * (fx * fz);
            const float dz = -w[0] * (ax * ay) - w[1] * (fx * ay) - w[2] * (ax * fy) - w[3] * (fx * fy)
                             + w[4] * (ax * ay) + w[5] * (fx * ay) + w[6] * (ax * fy) + w[7] * (fx * fy);
            const float go = (tid + (j0 + u) * TPB < cnt) ? (o[u] - fv[u]) * gcoef : 0.f;
            ggx[u] += dx * go; ggy[u] += dy * go; ggz[u] += dz * go;
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < ILP; ++u) o[u] = p[near[u]];
      }
#pragma unroll
      for (int u = 0; u < ILP; ++u) {
        const int l = tid + (j0 + u) * TPB;
        if (l < cnt) {
          if (FUSE_MSE) { const float d = o[u] - fv[u]; acc += d * d; }
          if (out) out[ob + l] = o[u];
        }
      }
    }
    if (FUSE_GRAD) {      // each lane owns its rows of sg: coordinates in, gradient out
#pragma unroll
      for (int u = 0; u < ILP; ++u) {
        const int l = tid + (j0 + u) * TPB;
        sg[l * 3] = ggx[u] * t[u].mx; sg[l * 3 + 1] = ggy[u] * t[u].my; sg[l * 3 + 2] = ggz[u] * t[u].mz;
      }
    }
  }
  if (FUSE_GRAD) {
    __syncthreads();
    unstage_rows(dgrid + ((long long)n * ovox + vb) * 3, cnt, sg, tid);
  }
  if (FUSE_MSE) {
    __shared__ double red[TPB / kWave];
    double s = block_sum<double>((double)acc, red);
    if (threadIdx.x == 0) partial[(long long)blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(TPB) void sample_bwd_grid_lc_kernel(
    const float* __restrict__ x, const float* __restrict__ grid, const float* __restrict__ gout,
    float* __restrict__ dgrid, int C, int D, int H, int W, long long ovox) {
  __shared__ __attribute__((aligned(16))) float sg[TPB * PASSES * 3];
  constexpr int ILP = 2;                   // voxels whose 4 pair-gathers are in flight together (ILP = 1: 151 us, 2: 138 us)
  const int n = blockIdx.y, tid = threadIdx.x;
  const long long vb = (long long)blockIdx.x * (TPB * PASSES);
  const int cnt = ovox - vb < TPB * PASSES ? (int)(ovox - vb) : TPB * PASSES;
  stage_rows(grid + ((long long)n * ovox + vb) * 3, cnt, sg, tid);
  __syncthreads();
  const long long plane = (long long)D * H * W;
  // each lane turns its own rows of sg from grid coordinates into grid gradients, ILP rows at a time
#pragma unroll 1
  for (int j0 = 0; j0 < PASSES; j0 += ILP) {
    Tap t[ILP];
    Tap32 q[ILP];
    float gx[ILP], gy[ILP], gz[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int l = tid + (j0 + u) * TPB;
      t[u] = make_tap(sg[l * 3], sg[l * 3 + 1], sg[l * 3 + 2], D, H, W);
      q[u] = make_tap32(t[u], D, H, W);
      if (l >= cnt) { q[u].r00 = q[u].r01 = q[u].r10 = q[u].r11 = 0; q[u].sel = false; }
      gx[u] = gy[u] = gz[u] = 0.f;
    }
    for (int c = 0; c < C; ++c) {
      const float* p = x + ((long long)n * C + c) * plane;
      const long long ob = ((long long)n * C + c) * ovox + vb;
      float v[ILP][8], go[ILP];
#pragma unroll
      for (int u = 0; u < ILP; ++u) {
        const int l = tid + (j0 + u) * TPB;
        go[u] = l < cnt ? gout[ob + l] : 0.f;
        gather8_pairs(p, q[u], v[u]);
      }
#pragma unroll
      for (int u = 0; u < ILP; ++u) {
        const float fx = t[u].fx, fy = t[u].fy, fz = t[u].fz;
        const float ax = 1.f - fx, ay = 1.f - fy, az = 1.f - fz;
        const float* w = v[u];
        // d/dix, d/diy, d/diz of the trilinear blend (ATen grid_sampler_3d_backward)
        const float dx = -w[0] * (ay * az) + w[1] * (ay * az) - w[2] * (fy * az) + w[3] * (fy * az)
                         - w[4] * (ay * fz) + w[5] * (ay * fz) - w[6] * (fy * fz) + w[7] * (fy * fz);
        const float dy = -w[0] * (ax * az) - w[1] * (fx * az) + w[2] * (ax * az) + w[3] * (fx * az)
                         - w[4] * (ax * fz) - w[5] * (fx * fz) + w[6] * (ax * fz) + w[7] * (fx * fz);
        const float dz = -w[0] * (ax * ay) - w[1] * (fx * ay) - w[2] * (ax * fy) - w[3] * (fx * fy)
                         + w[4] * (ax * ay) + w[5] * (fx * ay) + w[6] * (ax * fy) + w[7] * (fx * fy);
        gx[u] += dx * go[u]; gy[u] += dy * go[u]; gz[u] += dz * go[u];
      }
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int l = tid + (j0 + u) * TPB;
      sg[l * 3] = gx[u] * t[u].mx; sg[l * 3 + 1] = gy[u] * t[u].my; sg[l * 3 + 2] = gz[u] * t[u].mz;
    }
  }
  __syncthreads();
  unstage_rows(dgrid + ((long long)n * ovox + vb) * 3, cnt, sg, tid);
}

// ----------------------------------------------------------------------------------------------
// Fused align_img + soft DiceLoss (scripts/train.py:146-164 with loss_fn == "dice"; keymorph/utils.py:14-21,
// keymorph/loss_ops.py:16-63) WITHOUT the warped segmentation ever being stored.  Dice couples every voxel of a
// (sample, channel) row through its three sums, so the cotangent of the warp is only known after a full pass:
//   pass A (warp_dice_sums_kernel)  per (n, c): sum t p, sum p^2, sum t^2 with p = warp(x)[n, c] recomputed on the fly;
//   host: loss rows 1 - (2 I + 1) / (P + T + 1), and for the backward ca = -2 g / den, cb = 2 g num / den^2;
//   pass B (warp_dice_grad_kernel)  d(loss)/d(grid) = sum_c (ca[n,c] t + cb[n,c] p) * d p / d grid, p recomputed again.
// Per output voxel: A reads 12 + 8 C bytes, B reads 12 + 8 C and writes 12 -- the three-launch route (warp, Dice sums,
// axpby, grid backward) moves 24 + 32 C.  Both kernels are persistent over 1024-voxel chunks (lane-contiguous like
// sample_fwd_lc_kernel) and fetch the NEXT chunk's grid rows into registers before the current chunk's gathers.
// ILP = voxels of a lane whose gathers are in flight together (PASSES / ILP sub-passes per chunk)
constexpr int WD_MAXC = 128;

struct GridRows { float4 a, b, c; };      // PASSES * 3 / 4 = 3 float4 per thread (named members: an array went to scratch)
static_assert(PASSES * 3 / 4 == 3, "GridRows holds three float4 per thread");

__device__ __forceinline__ bool rows_fast(const float* src, int cnt) {
  return cnt == TPB * PASSES && ((reinterpret_cast<unsigned long long>(src) & 15) == 0);
}
__device__ __forceinline__ void fetch_rows(const float* __restrict__ src, bool fast, GridRows& g, int tid) {
  if (fast) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    g.a = s4[tid]; g.b = s4[tid + TPB]; g.c = s4[tid + 2 * TPB];
  }
}
__device__ __forceinline__ void commit_rows(const float* __restrict__ src, int cnt, bool fast, const GridRows& g, float* sg,
                                            int tid) {
  if (fast) {
    float4* d4 = reinterpret_cast<float4*>(sg);
    d4[tid] = g.a; d4[tid + TPB] = g.b; d4[tid + 2 * TPB] = g.c;
  } else {
    for (int e = tid; e < cnt * 3; e += TPB) sg[e] = src[e];
  }
}

__device__ __forceinline__ void blend_grads(const float w[8], const Tap& t, float& dx, float& dy, float& dz) {
  const float fx = t.fx, fy = t.fy, fz = t.fz;
  const float ax = 1.f - fx, ay = 1.f - fy, az = 1.f - fz;
  // d/dix, d/diy, d/diz of the trilinear blend (ATen grid_sampler_3d_backward), as sample_bwd_grid_lc_kernel
  dx = -w[0] * (ay * az) + w[1] * (ay * az) - w[2] * (fy * az) + w[3] * (fy * az)
       - w[4] * (ay * fz) + w[5] * (ay * fz) - w[6] * (fy * fz) + w[7] * (fy * fz);
  dy = -w[0] * (ax * az) - w[1] * (fx * az) + w[2] * (ax * az) + w[3] * (fx * az)
       - w[4] * (ax * fz) - w[5] * (fx * fz) + w[6] * (ax * fz) + w[7] * (fx * fz);
  dz = -w[0] * (ax * ay) - w[1] * (fx * ay) - w[2] * (ax * fy) - w[3] * (fx * fy)
       + w[4] * (ax * ay) + w[5] * (fx * ay) + w[6] * (ax * fy) + w[7] * (fx * fy);
}

// Gathers go through BUFFER loads: the channel plane's base lives in a scalar descriptor that the channel loop advances
// with two scalar adds, the per-voxel part is a 32-bit byte offset computed once per chunk -- no 64-bit VALU address
// arithmetic and no address registers per load (flat loads cost this loop 2 VALU + 2 VGPRs per gather), and lanes past
// the end of a chunk read zeros from the range check instead of needing clamped addresses.
typedef unsigned kmh_u2 __attribute__((vector_size(8)));      // the builtin's own return type (an ext_vector_type
                                                              // of the same size converts by SPLATTING element 0)
struct TapB {
  unsigned o00, o01, o10, o11;   // byte offsets of the four x-pairs inside one channel plane
  bool sel;                      // x0 is the last column: the pair was loaded one to the left
  float fx, fy, fz;
};
__device__ __forceinline__ TapB make_tapb(const Tap& t, int D, int H, int W) {
  TapB q;
  const int y1 = t.y0 + 1 < H ? t.y0 + 1 : t.y0, z1 = t.z0 + 1 < D ? t.z0 + 1 : t.z0;   // (fy = 0 / fz = 0 there)
  q.sel = t.x0 > W - 2;
  const int xb = q.sel ? W - 2 : t.x0;
  q.o00 = 4u * (unsigned)((t.z0 * H + t.y0) * W + xb); q.o01 = 4u * (unsigned)((t.z0 * H + y1) * W + xb);
  q.o10 = 4u * (unsigned)((z1 * H + t.y0) * W + xb);   q.o11 = 4u * (unsigned)((z1 * H + y1) * W + xb);
  q.fx = t.fx; q.fy = t.fy; q.fz = t.fz;
  return q;
}
__device__ __forceinline__ void pair_b(__amdgpu_buffer_rsrc_t r, unsigned off, bool sel, float& lo, float& hi) {
  const kmh_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
  // (scalars first: __builtin_bit_cast applied directly to a vector ELEMENT reads element 0 whatever the index -- hipcc 7.2)
  const unsigned ua = v[0], ub = v[1];
  const float a = __uint_as_float(ua), b = __uint_as_float(ub);
  lo = sel ? b : a;
  hi = sel ? 0.f : b;
}
__device__ __forceinline__ void gather8_b(__amdgpu_buffer_rsrc_t r, const TapB& q, float v[8]) {
  pair_b(r, q.o00, q.sel, v[0], v[1]);
  pair_b(r, q.o01, q.sel, v[2], v[3]);
  pair_b(r, q.o10, q.sel, v[4], v[5]);
  pair_b(r, q.o11, q.sel, v[6], v[7]);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float blend8f(const float v[8], float fx, float fy, float fz) {
  Tap t;
  t.fx = fx; t.fy = fy; t.fz = fz;
  return blend8(v, t);
}

// Walk over a sample's chunks for a PERSISTENT launch (gridDim.x a multiple of 8): XCD k (= blockIdx.x % 8: observed
// dispatch order, used for locality only -- any placement is correct) owns ONE contiguous range of chunks and its resident
// blocks sweep it side by side, so that chunks next to each other (a rotated grid makes a 4-row chunk touch ~50 source rows
// that its neighbours touch too) meet in one L2.  Measured: the same speed and the same FETCH_SIZE as the plain strided
// walk at 2 x 14 x 256^3 (the duplicate fetches are not cross-XCD duplicates: DESIGN.md section 8); kept because it is no
// slower and the kernels need a chunk loop for the prefetch of the next chunk's grid rows anyway.
struct ChunkWalk { int cur, end, step; };
__device__ __forceinline__ ChunkWalk chunk_walk(int b, int nb, int nchunk) {
  const int NX = nb < 8 ? nb : 8;            // fewer than 8 blocks: as many ranges as blocks (every range needs an owner)
  const int xcd = b % NX, idx = b / NX;
  const int q = nchunk / NX, r = nchunk % NX;
  ChunkWalk w;
  const int lo = xcd * q + (xcd < r ? xcd : r);
  w.end = lo + q + (xcd < r ? 1 : 0);
  w.step = (nb - xcd + NX - 1) / NX;          // blocks of this launch row that sit on this XCD
  w.cur = lo + idx;
  return w;
}

// LAB variants: both segmentations are exactly one-hot (what scripts/train.py:54-79 builds: one_hot of a label map,
// augmented with NEAREST sampling), so a voxel's C channel values are determined by ONE byte.  kmh_onehot_to_labels checks
// that on the device and writes the label maps; the kernels then gather 8 corner LABELS per voxel once instead of 8 corner
// values per channel (56 B of gathers and 56 B of fixed-segmentation reads per voxel become 8 + 1), and feed
// v_k = [label_k == c] into the SAME blend / derivative arithmetic: bit-identical results.  `gate` (device int): the LAB
// kernels return at once when it reads 0, the dense ones when it reads non-zero -- no host synchronisation decides.
struct LabTaps { unsigned c[8]; unsigned t; };      // 8 corner labels (255 = none) and the fixed label
__device__ __forceinline__ unsigned ld_lab(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, (int)off, 0, 0) & 255u;
}
// q's byte offsets are 4 * voxel index: the label map has one byte per voxel
__device__ __forceinline__ void gather_labels(__amdgpu_buffer_rsrc_t r, const TapB& q, bool live, LabTaps& L) {
  const unsigned o[4] = {q.o00 >> 2, q.o01 >> 2, q.o10 >> 2, q.o11 >> 2};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned a = ld_lab(r, o[k]), b = ld_lab(r, o[k] + 1u);
    L.c[2 * k] = live ? (q.sel ? b : a) : 255u;             // as pair_b: the last column's pair sits one to the left
    L.c[2 * k + 1] = (live && !q.sel) ? b : 255u;
  }
}

// partial: (N, gridDim.x, C, 3) doubles
template <int WD_ILP, bool LAB = false>
__global__ __launch_bounds__(TPB) void warp_dice_sums_kernel(
    const float* __restrict__ x, const float* __restrict__ grid, const float* __restrict__ fixed,
    double* __restrict__ partial, int C, int D, int H, int W, long long ovox, int nchunk,
    const unsigned char* __restrict__ labx = nullptr, const unsigned char* __restrict__ labf = nullptr,
    const int* __restrict__ gate = nullptr) {
  if (gate && ((*gate != 0) != LAB)) return;           // uniform: the other variant of this launch pair does the work
  __shared__ __attribute__((aligned(16))) float sg[TPB * PASSES * 3];
  __shared__ double racc[TPB / kWave][WD_MAXC][3];
  const int n = blockIdx.y, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  for (int e = tid; e < (TPB / kWave) * WD_MAXC * 3; e += TPB) (&racc[0][0][0])[e] = 0.0;
  const long long plane = (long long)D * H * W;
  const unsigned plane_bytes = (unsigned)(plane * 4);
  const float* gbase = grid + (long long)n * ovox * 3;
  const ChunkWalk cw = chunk_walk(blockIdx.x, gridDim.x, nchunk);
  int chunk = cw.cur;
  GridRows nxt = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  bool nfast = false;
  if (chunk < cw.end) {
    const long long vb = (long long)chunk * (TPB * PASSES);
    const int cnt = ovox - vb < TPB * PASSES ? (int)(ovox - vb) : TPB * PASSES;
    nfast = rows_fast(gbase + vb * 3, cnt);
    fetch_rows(gbase + vb * 3, nfast, nxt, tid);
  }
#pragma unroll 1
  for (; chunk < cw.end; chunk += cw.step) {
    const long long vb = (long long)chunk * (TPB * PASSES);
    const int cnt = ovox - vb < TPB * PASSES ? (int)(ovox - vb) : TPB * PASSES;
    __syncthreads();                                  // the previous chunk's readers of sg are done
    commit_rows(gbase + vb * 3, cnt, nfast, nxt, sg, tid);
    __syncthreads();
    {                                                 // the next chunk's rows: in flight under this chunk's gathers
      const int c2 = chunk + cw.step;
      if (c2 < cw.end) {
        const long long vb2 = (long long)c2 * (TPB * PASSES);
        const int cnt2 = ovox - vb2 < TPB * PASSES ? (int)(ovox - vb2) : TPB * PASSES;
        nfast = rows_fast(gbase + vb2 * 3, cnt2);
        fetch_rows(gbase + vb2 * 3, nfast, nxt, tid);
      }
    }
#pragma unroll 1
    for (int j0 = 0; j0 < PASSES; j0 += WD_ILP) {
      TapB q[WD_ILP];
#pragma unroll
      for (int u = 0; u < WD_ILP; ++u) {
        const int l = tid + (j0 + u) * TPB;
        q[u] = make_tapb(make_tap(sg[l * 3], sg[l * 3 + 1], sg[l * 3 + 2], D, H, W), D, H, W);
        if (l >= cnt) {     // past the chunk: every corner reads 0 through the range check, and the weights must be finite
          q[u].o00 = q[u].o01 = q[u].o10 = q[u].o11 = plane_bytes;      // (sg holds stale LDS there: 0 * NaN would poison
          q[u].fx = q[u].fy = q[u].fz = 0.f;                             //  the wave's sums)
        }
      }
      const int left = cnt - j0 * TPB;                 // voxels of the chunk from this sub-pass on (may be <= 0)
      const unsigned fbytes = left > 0 ? 4u * (unsigned)left : 0u;
      LabTaps lab[WD_ILP];
      if constexpr (LAB) {
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(labx + (long long)n * plane), 0, (int)plane, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(labf + (long long)n * ovox + vb + j0 * TPB), 0, left > 0 ? left : 0, 0x00020000);
#pragma unroll
        for (int u = 0; u < WD_ILP; ++u) {
          const bool live = tid + (j0 + u) * TPB < cnt;
          gather_labels(rl, q[u], live, lab[u]);
          const unsigned t = ld_lab(rt, (unsigned)(tid + u * TPB));
          lab[u].t = live ? t : 255u;
        }
      }
#pragma unroll 1
      for (int c = 0; c < C; ++c) {
        float v[WD_ILP][8], tv[WD_ILP];
        if constexpr (LAB) {
#pragma unroll
          for (int u = 0; u < WD_ILP; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[u][k] = lab[u].c[k] == (unsigned)c ? 1.f : 0.f;
            tv[u] = lab[u].t == (unsigned)c ? 1.f : 0.f;
          }
        } else {
          const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + ((long long)n * C + c) * plane, plane_bytes);
          const __amdgpu_buffer_rsrc_t rf = make_rsrc(fixed + ((long long)n * C + c) * ovox + vb + j0 * TPB, fbytes);
#pragma unroll
          for (int u = 0; u < WD_ILP; ++u) {
            gather8_b(rx, q[u], v[u]);
            tv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, 4 * (tid + u * TPB), 0, 0));
          }
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < WD_ILP; ++u) {
          const float o = blend8f(v[u], q[u].fx, q[u].fy, q[u].fz);      // 0 for lanes past the chunk (all corners read 0)
          s0 = fmaf(tv[u], o, s0); s1 = fmaf(o, o, s1); s2 = fmaf(tv[u], tv[u], s2);
        }
        s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { racc[wid][c][0] += (double)s0; racc[wid][c][1] += (double)s1; racc[wid][c][2] += (double)s2; }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < C * 3; e += TPB) {
    const int c = e / 3, k = e - c * 3;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < TPB / kWave; ++w) s += racc[w][c][k];
    partial[(((long long)n * gridDim.x + blockIdx.x) * C + c) * 3 + k] = s;
  }
}

// Multi-channel bilinear warp (align_img of a one-hot segmentation, keymorph/utils.py:14-21 under
// scripts/pairwise_register_eval.py).  sample_fwd_lc_kernel walks 4-row x 256 chunks of ONE output plane and pays one memory
// round trip per (sub-pass, channel): 1.8 ms at 14 x 256^3 (1.1 TB/s; PMC: L2 hit rate 37 %, FETCH_SIZE 3.4x the algorithmic
// reads).  Here a workgroup owns a compact 16 x 8 x 8 output TILE -- under a rotation about any axis its source box stays
// ~22 x 18 x 17, where a 32 x 8 x 4 tile's does not fit the LDS box below (measured: 1.06 vs 1.73 ms on bench.py's
// three-axis affine grid, 0.94 vs 0.88 ms on a one-axis rotation) -- tiles are walked x-fastest in one contiguous range per
// XCD, and the machinery of the Dice kernels does the rest: persistent blocks, the next tile's grid rows prefetched into
// registers, 32-bit corner offsets, buffer loads, range-checked stores.
// The blend is blend8 on the same corner values (a corner past the far border has weight 0 there and reads 0 here):
// bit-equal to the single-channel kernel (tests/test_ops_gpu.py::test_multichannel_sampler_equals_per_channel).
#ifndef KMH_MT_X
#define KMH_MT_X 16
#define KMH_MT_Y 8
#define KMH_MT_Z 8
#endif
constexpr int MT_X = KMH_MT_X, MT_Y = KMH_MT_Y, MT_Z = KMH_MT_Z;
static_assert(MT_X * MT_Y * MT_Z == TPB * PASSES && (MT_X & (MT_X - 1)) == 0 && (MT_Y & (MT_Y - 1)) == 0 && MT_X % 4 == 0,
              "1024-voxel tiles with power-of-two sides");
constexpr int MT_LX = __builtin_ctz(MT_X), MT_LXY = __builtin_ctz(MT_X * MT_Y);
constexpr int MT_Q4 = MT_X * 3 / 4;                  // 16-byte pieces of a tile row of the grid
// tile voxel l = tid + pass * 256  ->  (l & (MT_X-1), (l >> MT_LX) & (MT_Y-1), l >> MT_LXY)
// BOX path: the tile's corners live in a small source box (identity-like grid: 34 x 10 x 5); per channel the workgroup copies
// that box into LDS with coalesced 4-byte loads (prefetched into registers one channel ahead) and the 8 corners of a voxel
// come from four ds_read2_b32 instead of four 64-lane gathers of 4-byte-aligned 8-byte pairs (10 % faster than the global
// gathers of the same tile walk, 0.88 vs 0.97 ms at 14 x 256^3; with loads and stores compiled out the kernel still takes
// 0.41 ms: ~25 instructions per voxel and channel at 3 waves per SIMD are what bounds it, not the memory system).
// A tile whose box does not fit (strong zoom-out / shear: more than MB_PITCH columns or MB_ROWS rows) takes the global gathers.
constexpr int MB_PITCH = MT_X + 8, MB_CAP = 6144;

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m, 64); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
  return v;
}

struct BoxGeom { int x0, y0, z0, nx, ny, nrows; };
constexpr int MB_RPS = TPB / MB_PITCH;               // 6 box rows copied per step (240 of the 256 threads)
constexpr int MB_KMAX = 26;                          // steps: up to 156 rows
constexpr int MB_ROWS = MB_CAP / MB_PITCH;           // 153

// one channel loop of a tile through the LDS box; KR = box rows per thread (registers of the one-channel-ahead prefetch).
// rowoff[r]: byte offset (inside a channel plane) of box row r's first float, or >= plane_bytes for rows past the box.
// (16-byte copies -- 7 instead of 26 loads per thread and channel -- measured no faster: the kernel is bound by the ~25
// instructions per voxel and channel of the gather + blend + store, not by the copy.)
template <int KR>
__device__ __forceinline__ void mc_box_channels(const float* __restrict__ x, float* __restrict__ out, int n, int C,
                                                long long plane, long long ovox, unsigned plane_bytes, unsigned out_bytes,
                                                float* box, const unsigned* rowoff, int tid, const BoxGeom& g, int W,
                                                const unsigned (&lo)[PASSES][4], const float (&fr)[PASSES][3],
                                                const unsigned (&oo)[PASSES]) {
  const int r0 = tid / MB_PITCH, ix = tid - r0 * MB_PITCH;
  // idle lanes, columns past the box and the column past the volume's last one read 0 (x0 = W - 1: the pair's second value
  // has weight fx = 0, and the next row's first voxel there could turn 0 * Inf into a NaN the reference does not produce)
  const unsigned colb = (r0 < MB_RPS && ix < g.nx && g.x0 + ix < W) ? 4u * (unsigned)ix : plane_bytes;
  float R[KR];
  auto prefetch = [&](int c) {
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + ((long long)n * C + c) * plane, plane_bytes);
#pragma unroll
    for (int k = 0; k < KR; ++k)
      R[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(rowoff[r0 + k * MB_RPS] + colb), 0, 0));
  };
  prefetch(0);
#pragma unroll 1
  for (int c = 0; c < C; ++c) {
    __syncthreads();                                  // the previous channel's gathers are done
    if (r0 < MB_RPS) {
#pragma unroll
      for (int k = 0; k < KR; ++k) box[tid + k * (MB_RPS * MB_PITCH)] = R[k];
    }
    __syncthreads();
    if (c + 1 < C) prefetch(c + 1);                   // the next channel's box: in flight under this channel's gathers
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + ((long long)n * C + c) * ovox, out_bytes);
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float2 r;
        __builtin_memcpy(&r, box + lo[u][k], sizeof(float2));      // 4-byte aligned pair: ds_read2_b32
        v[2 * k] = r.x; v[2 * k + 1] = r.y;
      }
      const float o = blend8f(v, fr[u][0], fr[u][1], fr[u][2]);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ro, (int)oo[u], 0, 0);      // dropped past the volume
    }
  }
}

template <int MC_ILP>
__global__ __launch_bounds__(TPB, 3) void sample_fwd_mc_kernel(
    const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ out, int C, int D, int H, int W,
    int Do, int Ho, int Wo, int ntx, int nty, int ntile, int use_box) {
  __shared__ __attribute__((aligned(16))) float sg[TPB * PASSES * 3];
  __shared__ __attribute__((aligned(16))) float box[MB_KMAX * MB_RPS * MB_PITCH];
  __shared__ unsigned rowoff[MB_KMAX * MB_RPS + MB_RPS];
  __shared__ int sred[TPB / kWave][6];
  const int n = blockIdx.y, tid = threadIdx.x;
  const long long plane = (long long)D * H * W, ovox = (long long)Do * Ho * Wo;
  const unsigned plane_bytes = (unsigned)(plane * 4), out_bytes = (unsigned)(ovox * 4);
  const float* gbase = grid + (long long)n * ovox * 3;
  const int lx = tid & (MT_X - 1), ly = (tid >> MT_LX) & (MT_Y - 1), lz = tid >> MT_LXY, dzp = TPB >> MT_LXY;      // pass u: plane lz + u * dzp
  // the tile's 32 grid rows (96 floats each) as 768 float4: thread t owns numbers t, t + 256, t + 512
  auto row_src = [&](int tile, int idx, bool& ok) -> const float* {
    const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
    const int r = idx / MT_Q4, q4 = idx - r * MT_Q4;
    const int z = tz * MT_Z + r / MT_Y, y = ty * MT_Y + (r & (MT_Y - 1));
    ok = z < Do && y < Ho;
    return gbase + (((long long)z * Ho + y) * Wo + tx * MT_X) * 3 + q4 * 4;
  };
  auto tile_fast = [&](int tile) -> bool {       // whole 32-voxel rows, 16-byte aligned: Wo % 4 == 0 and the tile inside in x
    const int tx = tile % ntx;
    return (Wo & 3) == 0 && tx * MT_X + MT_X <= Wo && ((reinterpret_cast<unsigned long long>(gbase) & 15) == 0);
  };
  auto fetch = [&](int tile, GridRows& g) {
    bool ok;
    const float* p0 = row_src(tile, tid, ok);
    g.a = ok ? *reinterpret_cast<const float4*>(p0) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p1 = row_src(tile, tid + TPB, ok);
    g.b = ok ? *reinterpret_cast<const float4*>(p1) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p2 = row_src(tile, tid + 2 * TPB, ok);
    g.c = ok ? *reinterpret_cast<const float4*>(p2) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  const ChunkWalk cw = chunk_walk(blockIdx.x, gridDim.x, ntile);
  int tile = cw.cur;
  GridRows nxt = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  bool nfast = false;
  if (tile < cw.end) {
    nfast = tile_fast(tile);
    if (nfast) fetch(tile, nxt);
  }
#pragma unroll 1
  for (; tile < cw.end; tile += cw.step) {
    const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
    const int x0 = tx * MT_X, y0 = ty * MT_Y, z0 = tz * MT_Z;
    __syncthreads();                                  // the previous tile's readers of sg / box are done
    if (nfast) {
      float4* d4 = reinterpret_cast<float4*>(sg);
      d4[tid] = nxt.a; d4[tid + TPB] = nxt.b; d4[tid + 2 * TPB] = nxt.c;
    } else {                                          // edge tile / unaligned rows: element by element, zeros outside
      for (int e = tid; e < TPB * PASSES * 3; e += TPB) {
        const int l = e / 3, k = e - l * 3;
        const int xx = x0 + (l & (MT_X - 1)), yy = y0 + ((l >> MT_LX) & (MT_Y - 1)), zz = z0 + (l >> MT_LXY);
        sg[e] = (xx < Wo && yy < Ho && zz < Do) ? gbase[(((long long)zz * Ho + yy) * Wo + xx) * 3 + k] : 0.f;
      }
    }
    __syncthreads();
    {                                                 // the next tile's rows: in flight under this tile's gathers
      const int t2 = tile + cw.step;
      if (t2 < cw.end) {
        nfast = tile_fast(t2);
        if (nfast) fetch(t2, nxt);
      }
    }
    const bool in_xy = x0 + lx < Wo && y0 + ly < Ho;
    bool boxed = false;
    if (use_box) {
      // corners of the lane's 4 voxels (one per tile plane) and the box that holds every live corner of the tile
      int cx[PASSES], cy[PASSES], cz[PASSES], cy1[PASSES], cz1[PASSES];
      float fr[PASSES][3];
      unsigned oo[PASSES];
      int mn[3] = {1 << 30, 1 << 30, 1 << 30}, mx[3] = {-1, -1, -1};
#pragma unroll
      for (int u = 0; u < PASSES; ++u) {
        const int l = tid + u * TPB, z = z0 + lz + u * dzp;
        const Tap t = make_tap(sg[l * 3], sg[l * 3 + 1], sg[l * 3 + 2], D, H, W);
        const bool live = in_xy && z < Do;
        cx[u] = t.x0; cy[u] = t.y0; cz[u] = t.z0;      // (the pair of x0 = W - 1 takes a zero from past the box's last column)
        cy1[u] = t.y0 + 1 < H ? t.y0 + 1 : t.y0; cz1[u] = t.z0 + 1 < D ? t.z0 + 1 : t.z0;
        fr[u][0] = t.fx; fr[u][1] = t.fy; fr[u][2] = t.fz;
        oo[u] = live ? 4u * (unsigned)(((long long)z * Ho + (y0 + ly)) * Wo + (x0 + lx)) : 0xfffffffcu;
        if (live) {
          mn[0] = cx[u] < mn[0] ? cx[u] : mn[0]; mx[0] = cx[u] + 1 > mx[0] ? cx[u] + 1 : mx[0];
          mn[1] = cy[u] < mn[1] ? cy[u] : mn[1]; mx[1] = cy1[u] > mx[1] ? cy1[u] : mx[1];
          mn[2] = cz[u] < mn[2] ? cz[u] : mn[2]; mx[2] = cz1[u] > mx[2] ? cz1[u] : mx[2];
        } else {            // a dead lane's corners sit on the box origin (filled in below)
          cx[u] = cy[u] = cz[u] = cy1[u] = cz1[u] = -1;
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) { mn[k] = wave_min_i(mn[k]); mx[k] = wave_max_i(mx[k]); }
      if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { sred[tid >> 6][k] = mn[k]; sred[tid >> 6][3 + k] = mx[k]; }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int w = 0; w < TPB / kWave; ++w) {
          mn[k] = sred[w][k] < mn[k] ? sred[w][k] : mn[k];
          mx[k] = sred[w][3 + k] > mx[k] ? sred[w][3 + k] : mx[k];
        }
      }
      BoxGeom g;
      g.x0 = mn[0]; g.y0 = mn[1]; g.z0 = mn[2];
      g.nx = mx[0] - mn[0] + 1; g.ny = mx[1] - mn[1] + 1;
      g.nrows = g.ny * (mx[2] - mn[2] + 1);
      boxed = g.nx <= MB_PITCH && g.nrows <= MB_ROWS;      // uniform
      if (boxed) {
        unsigned lo[PASSES][4];
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
          const bool dead = cx[u] < 0;
          const int ax = dead ? 0 : cx[u] - g.x0, ay = dead ? 0 : cy[u] - g.y0, ay1 = dead ? 0 : cy1[u] - g.y0;
          const int az = dead ? 0 : cz[u] - g.z0, az1 = dead ? 0 : cz1[u] - g.z0;
          lo[u][0] = (unsigned)((az * g.ny + ay) * MB_PITCH + ax); lo[u][1] = (unsigned)((az * g.ny + ay1) * MB_PITCH + ax);
          lo[u][2] = (unsigned)((az1 * g.ny + ay) * MB_PITCH + ax); lo[u][3] = (unsigned)((az1 * g.ny + ay1) * MB_PITCH + ax);
        }
        // row offsets of the box (rows past it: out of range)
        for (int r = tid; r < MB_KMAX * MB_RPS + MB_RPS; r += TPB) {
          const int iz = (int)(((float)r + 0.5f) / (float)g.ny), iy = r - iz * g.ny;      // exact: r, ny < 2^10
          rowoff[r] = r < g.nrows ? 4u * (unsigned)(((g.z0 + iz) * H + (g.y0 + iy)) * W + g.x0) : plane_bytes;
        }
        __syncthreads();
        if (g.nrows <= 10 * MB_RPS)
          mc_box_channels<10>(x, out, n, C, plane, ovox, plane_bytes, out_bytes, box, rowoff, tid, g, W, lo, fr, oo);
        else if (g.nrows <= 18 * MB_RPS)
          mc_box_channels<18>(x, out, n, C, plane, ovox, plane_bytes, out_bytes, box, rowoff, tid, g, W, lo, fr, oo);
        else
          mc_box_channels<MB_KMAX>(x, out, n, C, plane, ovox, plane_bytes, out_bytes, box, rowoff, tid, g, W, lo, fr, oo);
      }
    }
    if (boxed) continue;
#pragma unroll 1
    for (int j0 = 0; j0 < PASSES; j0 += MC_ILP) {
      TapB q[MC_ILP];
      unsigned oo[MC_ILP];                            // byte offset of the lane's output voxel inside a channel plane
#pragma unroll
      for (int u = 0; u < MC_ILP; ++u) {
        const int l = tid + (j0 + u) * TPB;
        const int z = z0 + lz + (j0 + u) * dzp;
        q[u] = make_tapb(make_tap(sg[l * 3], sg[l * 3 + 1], sg[l * 3 + 2], D, H, W), D, H, W);
        const bool live = in_xy && z < Do;
        oo[u] = live ? 4u * (unsigned)(((long long)z * Ho + (y0 + ly)) * Wo + (x0 + lx)) : 0xfffffffcu;   // dropped by the range check
        if (!live) q[u].o00 = q[u].o01 = q[u].o10 = q[u].o11 = plane_bytes;      // reads zeros
      }
#pragma unroll 2
      for (int c = 0; c < C; ++c) {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + ((long long)n * C + c) * plane, plane_bytes);
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + ((long long)n * C + c) * ovox, out_bytes);
        float v[MC_ILP][8];
#pragma unroll
        for (int u = 0; u < MC_ILP; ++u) gather8_b(rx, q[u], v[u]);
#pragma unroll
        for (int u = 0; u < MC_ILP; ++u) {
          const float o = blend8f(v[u], q[u].fx, q[u].fy, q[u].fz);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ro, (int)oo[u], 0, 0);
        }
      }
    }
  }
}

// partial (N, nb, C, 3) -> sums (N*C, 3) floats: one wave per (n, c, k), fixed order
__global__ __launch_bounds__(TPB) void warp_dice_final_kernel(const double* __restrict__ partial, int nb, int C, int total,
                                                              float* __restrict__ sums) {
  const int e = blockIdx.x * (TPB / kWave) + (threadIdx.x >> 6);
  if (e >= total) return;
  const int lane = threadIdx.x & 63;
  const int n = e / (C * 3), r = e - n * (C * 3);
  const double* p = partial + (long long)n * nb * C * 3 + r;
  double s = 0.0;
  for (int b = lane; b < nb; b += kWave) s += p[(long long)b * C * 3];
  s = wave_sum(s);
  if (lane == 0) sums[e] = (float)s;
}

template <int WD_ILP, bool LAB = false>
__global__ __launch_bounds__(TPB) void warp_dice_grad_kernel(
    const float* __restrict__ x, const float* __restrict__ grid, const float* __restrict__ fixed,
    const float* __restrict__ ca, const float* __restrict__ cb, float* __restrict__ dgrid, int C, int D, int H, int W,
    long long ovox, int nchunk, const unsigned char* __restrict__ labx = nullptr,
    const unsigned char* __restrict__ labf = nullptr, const int* __restrict__ gate = nullptr) {
  if (gate && ((*gate != 0) != LAB)) return;
  __shared__ __attribute__((aligned(16))) float sg[TPB * PASSES * 3];
  const int n = blockIdx.y, tid = threadIdx.x;
  const long long plane = (long long)D * H * W;
  const unsigned plane_bytes = (unsigned)(plane * 4);
  const float* gbase = grid + (long long)n * ovox * 3;
  const ChunkWalk cw = chunk_walk(blockIdx.x, gridDim.x, nchunk);
  int chunk = cw.cur;
  GridRows nxt = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  bool nfast = false;
  if (chunk < cw.end) {
    const long long vb = (long long)chunk * (TPB * PASSES);
    const int cnt = ovox - vb < TPB * PASSES ? (int)(ovox - vb) : TPB * PASSES;
    nfast = rows_fast(gbase + vb * 3, cnt);
    fetch_rows(gbase + vb * 3, nfast, nxt, tid);
  }
#pragma unroll 1
  for (; chunk < cw.end; chunk += cw.step) {
    const long long vb = (long long)chunk * (TPB * PASSES);
    const int cnt = ovox - vb < TPB * PASSES ? (int)(ovox - vb) : TPB * PASSES;
    __syncthreads();                                  // the previous chunk's gradient rows have left sg
    commit_rows(gbase + vb * 3, cnt, nfast, nxt, sg, tid);
    __syncthreads();
    {
      const int c2 = chunk + cw.step;
      if (c2 < cw.end) {
        const long long vb2 = (long long)c2 * (TPB * PASSES);
        const int cnt2 = ovox - vb2 < TPB * PASSES ? (int)(ovox - vb2) : TPB * PASSES;
        nfast = rows_fast(gbase + vb2 * 3, cnt2);
        fetch_rows(gbase + vb2 * 3, nfast, nxt, tid);
      }
    }
#pragma unroll 1
    for (int j0 = 0; j0 < PASSES; j0 += WD_ILP) {
      TapB q[WD_ILP];
      float gx[WD_ILP], gy[WD_ILP], gz[WD_ILP];
#pragma unroll
      for (int u = 0; u < WD_ILP; ++u) {
        const int l = tid + (j0 + u) * TPB;
        q[u] = make_tapb(make_tap(sg[l * 3], sg[l * 3 + 1], sg[l * 3 + 2], D, H, W), D, H, W);
        if (l >= cnt) {     // reads 0 with finite weights: no contribution (and nothing of this row is stored)
          q[u].o00 = q[u].o01 = q[u].o10 = q[u].o11 = plane_bytes;
          q[u].fx = q[u].fy = q[u].fz = 0.f;
        }
        gx[u] = gy[u] = gz[u] = 0.f;
      }
      const int left = cnt - j0 * TPB;
      const unsigned fbytes = left > 0 ? 4u * (unsigned)left : 0u;
      LabTaps lab[WD_ILP];
      if constexpr (LAB) {
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(labx + (long long)n * plane), 0, (int)plane, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(labf + (long long)n * ovox + vb + j0 * TPB), 0, left > 0 ? left : 0, 0x00020000);
#pragma unroll
        for (int u = 0; u < WD_ILP; ++u) {
          const bool live = tid + (j0 + u) * TPB < cnt;
          gather_labels(rl, q[u], live, lab[u]);
          const unsigned t = ld_lab(rt, (unsigned)(tid + u * TPB));
          lab[u].t = live ? t : 255u;
        }
      }
#pragma unroll 1
      for (int c = 0; c < C; ++c) {
        const float a = ca[n * C + c], b = cb[n * C + c];
        float v[WD_ILP][8], tv[WD_ILP];
        if constexpr (LAB) {
#pragma unroll
          for (int u = 0; u < WD_ILP; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[u][k] = lab[u].c[k] == (unsigned)c ? 1.f : 0.f;
            tv[u] = lab[u].t == (unsigned)c ? 1.f : 0.f;
          }
        } else {
          const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + ((long long)n * C + c) * plane, plane_bytes);
          const __amdgpu_buffer_rsrc_t rf = make_rsrc(fixed + ((long long)n * C + c) * ovox + vb + j0 * TPB, fbytes);
#pragma unroll
          for (int u = 0; u < WD_ILP; ++u) {
            gather8_b(rx, q[u], v[u]);
            tv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, 4 * (tid + u * TPB), 0, 0));
          }
        }
#pragma unroll
        for (int u = 0; u < WD_ILP; ++u) {
          Tap t;
          t.fx = q[u].fx; t.fy = q[u].fy; t.fz = q[u].fz;
          const float o = blend8(v[u], t);
          const float go = fmaf(a, tv[u], b * o);          // 0 past the chunk: tv and every corner read 0
          float dx, dy, dz;
          blend_grads(v[u], t, dx, dy, dz);
          gx[u] = fmaf(dx, go, gx[u]); gy[u] = fmaf(dy, go, gy[u]); gz[u] = fmaf(dz, go, gz[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < WD_ILP; ++u) {              // each lane owns its rows of sg: coordinates in, gradient out
        const int l = tid + (j0 + u) * TPB;
        float mx, my, mz;                              // d(ix)/d(gx) incl. the clamp mask, from the coordinates still in sg
        unnorm_clip(sg[l * 3], W, mx); unnorm_clip(sg[l * 3 + 1], H, my); unnorm_clip(sg[l * 3 + 2], D, mz);
        sg[l * 3] = gx[u] * mx; sg[l * 3 + 1] = gy[u] * my; sg[l * 3 + 2] = gz[u] * mz;
      }
    }
    __syncthreads();
    unstage_rows(dgrid + ((long long)n * ovox + vb) * 3, cnt, sg, tid);
  }
}

// x (N, C, V) floats -> lab (N, V) bytes when every voxel is exactly one-hot (one channel == 1.0f, all others == 0.0f);
// any other voxel clears *ok (preset to 1 by the launcher; same-value stores from many threads)
__global__ __launch_bounds__(TPB) void onehot_to_labels_kernel(const float* __restrict__ x, int C, long long V,
                                                               unsigned char* __restrict__ lab, int* __restrict__ ok) {
  const int n = blockIdx.y;
  const float* xn = x + (long long)n * C * V;
  unsigned char* ln = lab + (long long)n * V;
  // 16-byte loads need every channel plane (and the byte map) aligned: V % 4 == 0 and aligned bases; else the scalar loop
  const bool vec = (V & 3) == 0 && ((reinterpret_cast<unsigned long long>(x) & 15) == 0) &&
                   ((reinterpret_cast<unsigned long long>(lab) & 3) == 0);
  const long long V4 = vec ? (V >> 2) : 0;
  bool good = true;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < V4; i += (long long)gridDim.x * TPB) {
    int ones[4] = {0, 0, 0, 0}, which[4] = {0, 0, 0, 0};
    bool clean = true;
    for (int c = 0; c < C; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(xn + (long long)c * V + 4 * i);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (vv[j] == 1.f) { ++ones[j]; which[j] = c; }
        else if (vv[j] != 0.f) clean = false;            // (NaN lands here too)
      }
    }
    good = good && clean && ones[0] == 1 && ones[1] == 1 && ones[2] == 1 && ones[3] == 1;
    *reinterpret_cast<unsigned*>(ln + 4 * i) = (unsigned)which[0] | ((unsigned)which[1] << 8) | ((unsigned)which[2] << 16) |
                                               ((unsigned)which[3] << 24);
  }
  if (!vec) {                                            // unaligned shapes: one voxel per thread
    for (long long v = (long long)blockIdx.x * TPB + threadIdx.x; v < V; v += (long long)gridDim.x * TPB) {
      int ones = 0, which = 0;
      for (int c = 0; c < C; ++c) {
        const float t = xn[(long long)c * V + v];
        if (t == 1.f) { ++ones; which = c; } else if (t != 0.f) good = false;
      }
      good = good && ones == 1;
      ln[v] = (unsigned char)which;
    }
  }
  if (!good) *ok = 0;
}

// scatter-add backward wrt the sampled volume (not on the training hot path: the volumes are data;
// used by augmentation-through-images and for completeness of align_img's autograd).
__global__ __launch_bounds__(TPB) void sample_bwd_input_kernel(
    const float* __restrict__ grid, const float* __restrict__ gout, float* __restrict__ dx, int C, int D,
    int H, int W, long long ovox) {
  const int n = blockIdx.y;
  const long long v = (long long)blockIdx.x * TPB + threadIdx.x;
  if (v >= ovox) return;
  const float* gp = grid + ((long long)n * ovox + v) * 3;
  Tap t = make_tap(gp[0], gp[1], gp[2], D, H, W);
  const long long plane = (long long)D * H * W;
  const float wx[2] = {1.f - t.fx, t.fx}, wy[2] = {1.f - t.fy, t.fy}, wz[2] = {1.f - t.fz, t.fz};
  for (int c = 0; c < C; ++c) {
    const float go = gout[((long long)n * C + c) * ovox + v];
    float* p = dx + ((long long)n * C + c) * plane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int xx = t.x0 + (k & 1), yy = t.y0 + ((k >> 1) & 1), zz = t.z0 + (k >> 2);
      if (xx < W && yy < H && zz < D)
        atomicAdd(p + ((long long)zz * H + yy) * W + xx, go * wx[k & 1] * wy[(k >> 1) & 1] * wz[k >> 2]);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// reductions
constexpr int RED_BLOCKS = 2048;

__global__ __launch_bounds__(TPB) void sqdiff_partial_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ b, long long n,
                                                             double* __restrict__ partial) {
  float acc = 0.f;
  double dacc = 0.0;
  const long long n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  int cnt = 0;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
    float4 p = a4[i], q = b4[i];
    float d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z, d3 = p.w - q.w;
    acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    if (++cnt == 64) { dacc += acc; acc = 0.f; cnt = 0; }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float d = a[n4 * 4 + threadIdx.x] - b[n4 * 4 + threadIdx.x];
    acc += d * d;
  }
  dacc += acc;
  __shared__ double red[TPB / kWave];
  double s = block_sum<double>(dacc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(TPB) void finalize_mean_kernel(const double* __restrict__ partial, int np,
                                                            double inv_n, float* __restrict__ out) {
  // fixed summation order (deterministic); 8 independent loads in flight per lane
  double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int i = threadIdx.x;
  for (; i + 7 * TPB < np; i += 8 * TPB) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s8[k] += partial[i + k * TPB];
  }
  for (int k = 0; i < np; i += TPB, ++k) s8[k & 7] += partial[i];
  double s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  __shared__ double red[TPB / kWave];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) out[0] = (float)(s * inv_n);
}

__global__ __launch_bounds__(TPB) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ gscale, long long n,
                                                      float* __restrict__ da) {
  const float s = gscale[0] * 2.f / (float)n;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB)
    da[i] = s * (a[i] - b[i]);
}

// Dice: per row r: {sum t*p, sum p*p, sum t*t}.  grid (bx, R); partial (R, bx, 3) doubles.
__global__ __launch_bounds__(TPB) void dice_partial_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ target, long long V,
                                                           double* __restrict__ partial) {
  const int r = blockIdx.y;
  const float* p = pred + (long long)r * V;
  const float* t = target + (long long)r * V;
  double s0 = 0, s1 = 0, s2 = 0;
  float a0 = 0, a1 = 0, a2 = 0;
  int cnt = 0;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < V; i += (long long)gridDim.x * TPB) {
    float pv = p[i], tv = t[i];
    a0 += tv * pv; a1 += pv * pv; a2 += tv * tv;
    if (++cnt == 256) { s0 += a0; s1 += a1; s2 += a2; a0 = a1 = a2 = 0.f; cnt = 0; }
  }
  s0 += a0; s1 += a1; s2 += a2;
  __shared__ double red[TPB / kWave];
  s0 = block_sum<double>(s0, red);
  s1 = block_sum<double>(s1, red);
  s2 = block_sum<double>(s2, red);
  if (threadIdx.x == 0) {
    double* o = partial + ((long long)r * gridDim.x + blockIdx.x) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}

__global__ __launch_bounds__(TPB) void dice_finalize_kernel(const double* __restrict__ partial, int nb,
                                                            float* __restrict__ sums) {
  const int r = blockIdx.x;
  double s0 = 0, s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < nb; i += TPB) {
    const double* o = partial + ((long long)r * nb + i) * 3;
    s0 += o[0]; s1 += o[1]; s2 += o[2];
  }
  __shared__ double red[TPB / kWave];
  s0 = block_sum<double>(s0, red);
  s1 = block_sum<double>(s1, red);
  s2 = block_sum<double>(s2, red);
  if (threadIdx.x == 0) { sums[r * 3] = (float)s0; sums[r * 3 + 1] = (float)s1; sums[r * 3 + 2] = (float)s2; }
}

__global__ __launch_bounds__(TPB) void rows_axpby_kernel(const float* __restrict__ t, const float* __restrict__ p,
                                                         const float* __restrict__ ca, const float* __restrict__ cb,
                                                         long long V, float* __restrict__ out) {
  const int r = blockIdx.y;
  const float a = ca[r], b = cb[r];
  const long long base = (long long)r * V;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < V; i += (long long)gridDim.x * TPB)
    out[base + i] = a * t[base + i] + b * p[base + i];
}

__global__ __launch_bounds__(TPB) void argmax_onehot_kernel(const float* __restrict__ pred, int C, long long V,
                                                            float* __restrict__ out) {
  const int n = blockIdx.y;
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= V) return;
  const float* p = pred + (long long)n * C * V + i;
  float best = p[0];
  int bi = 0;
  for (int c = 1; c < C; ++c) {
    float v = p[(long long)c * V];
    if (v > best) { best = v; bi = c; }
  }
  float* o = out + (long long)n * C * V + i;
  for (int c = 0; c < C; ++c) o[(long long)c * V] = (c == bi) ? 1.f : 0.f;
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------
// Jacobian determinant of a dense map (keymorph/loss_ops.py:161-247, eval metrics jdstd / jdlessthan0):
// J[a][c] = d disp_c / d axis_a by central differences (0.5 (f[i+1] - f[i-1]), zero outside the volume) + I, on
// the volume cropped by 2 voxels per side.  One pass: optional per-voxel determinant + {sum, sum^2, #(<= 0)}.
__global__ __launch_bounds__(TPB) void jacdet_kernel(const float* __restrict__ disp, long long cstride,
                                                     long long vstride, int D, int H, int W, float* __restrict__ jd,
                                                     double* __restrict__ partial /* (nblocks, 3) */) {
  const int Di = D - 4, Hi = H - 4, Wi = W - 4;
  const long long total = (long long)Di * Hi * Wi;
  double s = 0, ss = 0, neg = 0;
  for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
    const int x = (int)(e % Wi) + 2, y = (int)((e / Wi) % Hi) + 2, z = (int)(e / ((long long)Wi * Hi)) + 2;
    float J[3][3];   // [axis a][component c]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* p = disp + c * cstride;
      auto at = [&](int zz, int yy, int xx) { return p[(((long long)zz * H + yy) * W + xx) * vstride]; };
      J[0][c] = 0.5f * at(z + 1, y, x) - 0.5f * at(z - 1, y, x);
      J[1][c] = 0.5f * at(z, y + 1, x) - 0.5f * at(z, y - 1, x);
      J[2][c] = 0.5f * at(z, y, x + 1) - 0.5f * at(z, y, x - 1);
    }
    J[0][0] += 1.f; J[1][1] += 1.f; J[2][2] += 1.f;
    // same expansion (and association) as the reference
    const float det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) -
                      J[1][0] * (J[0][1] * J[2][2] - J[0][2] * J[2][1]) +
                      J[2][0] * (J[0][1] * J[1][2] - J[0][2] * J[1][1]);
    if (jd) jd[e] = det;
    s += det; ss += (double)det * det; neg += det <= 0.f ? 1.0 : 0.0;
  }
  __shared__ double red[TPB / kWave];
  s = block_sum<double>(s, red);
  ss = block_sum<double>(ss, red);
  neg = block_sum<double>(neg, red);
  if (threadIdx.x == 0) { partial[blockIdx.x * 3] = s; partial[blockIdx.x * 3 + 1] = ss; partial[blockIdx.x * 3 + 2] = neg; }
}

__global__ __launch_bounds__(TPB) void jacdet_final_kernel(const double* __restrict__ partial, int nb, double count,
                                                           double* __restrict__ out /* mean, std (ddof 0), #<=0, count */) {
  double s = 0, ss = 0, neg = 0;
  for (int i = threadIdx.x; i < nb; i += TPB) { s += partial[i * 3]; ss += partial[i * 3 + 1]; neg += partial[i * 3 + 2]; }
  __shared__ double red[TPB / kWave];
  s = block_sum<double>(s, red);
  ss = block_sum<double>(ss, red);
  neg = block_sum<double>(neg, red);
  if (threadIdx.x == 0) {
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0) var = 0;
    out[0] = mean; out[1] = sqrt(var); out[2] = neg; out[3] = count;
  }
}

static bool lane_contiguous_ok(int D, int H, int W) {
  static const bool force_old = getenv("KMH_SAMPLER_OLD") != nullptr;   // A/B switch for tools/bench_sampler.py
  return !force_old && W >= 2 && (long long)D * H * W < (1ll << 31);
}

KMH_API int kmh_abi_version(void) { return 1; }

KMH_API size_t kmh_reduce_ws_bytes(void) { return (size_t)65536 * sizeof(double) * 3; }

KMH_API int kmh_grid_sample3d_fwd(const float* x, const float* grid, float* out, int N, int C, int D, int H,
                                  int W, int Do, int Ho, int Wo, int mode, void* stream) {
  if (N <= 0 || C <= 0) return -22;
  const long long ovox = (long long)Do * Ho * Wo;
  dim3 g(ceil_div(ovox, (long long)TPB * VPT), N);
  hipStream_t s = (hipStream_t)stream;
  // C >= 2, bilinear: the persistent tiled multi-channel kernel (KMH_SAMPLER_MC=0: the lc kernel; KMH_SAMPLER_MC_MINC=1: also C = 1)
  static const int mc = getenv("KMH_SAMPLER_MC") ? atoi(getenv("KMH_SAMPLER_MC")) : 4;
  static const int minc = getenv("KMH_SAMPLER_MC_MINC") ? atoi(getenv("KMH_SAMPLER_MC_MINC")) : 2;
  if (mode == 0 && C >= minc && mc && lane_contiguous_ok(D, H, W) && (long long)D * H * W < (1ll << 30) && ovox < (1ll << 30)) {
    const int ntx = (Wo + MT_X - 1) / MT_X, nty = (Ho + MT_Y - 1) / MT_Y, ntz = (Do + MT_Z - 1) / MT_Z;
    const long long nt = (long long)ntx * nty * ntz;
    if (nt < (1ll << 30)) {
      static const int capa = getenv("KMH_MC_BLOCKS") ? atoi(getenv("KMH_MC_BLOCKS")) : 2048;   // ~ resident blocks of the chip
      long long nb = (capa / N) & ~7;                  // a multiple of 8 per sample row: blockIdx.x % 8 is the XCD
      if (nb < 8) nb = 8;
      if (nb > nt) nb = nt;
      const dim3 gm((unsigned)nb, N);
      static const int use_box = getenv("KMH_SAMPLER_BOX") ? atoi(getenv("KMH_SAMPLER_BOX")) : 1;
      if (mc == 2) sample_fwd_mc_kernel<2><<<gm, TPB, 0, s>>>(x, grid, out, C, D, H, W, Do, Ho, Wo, ntx, nty, (int)nt, use_box);
      else sample_fwd_mc_kernel<4><<<gm, TPB, 0, s>>>(x, grid, out, C, D, H, W, Do, Ho, Wo, ntx, nty, (int)nt, use_box);
      return KMH_LAUNCH_CHECK();
    }
  }
  if (lane_contiguous_ok(D, H, W)) {
    if (mode == 0)
      sample_fwd_lc_kernel<0, false><<<g, TPB, 0, s>>>(x, grid, out, nullptr, nullptr, C, D, H, W, ovox);
    else
      sample_fwd_lc_kernel<1, false><<<g, TPB, 0, s>>>(x, grid, out, nullptr, nullptr, C, D, H, W, ovox);
  } else if (mode == 0) {
    sample_fwd_kernel<0, false><<<g, TPB, 0, s>>>(x, grid, out, nullptr, nullptr, C, D, H, W, ovox);
  } else {
    sample_fwd_kernel<1, false><<<g, TPB, 0, s>>>(x, grid, out, nullptr, nullptr, C, D, H, W, ovox);
  }
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_warp_mse_fwd(const float* x, const float* grid, const float* fixed, float* out,
                             float* out_loss, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                             void* ws, void* stream) {
  const long long ovox = (long long)Do * Ho * Wo;
  dim3 g(ceil_div(ovox, (long long)TPB * VPT), N);
  if ((long long)g.x * g.y > 65536 * 3) return -22;
  hipStream_t s = (hipStream_t)stream;
  if (lane_contiguous_ok(D, H, W)) {
    sample_fwd_lc_kernel<0, true><<<g, TPB, 0, s>>>(x, grid, out, fixed, (double*)ws, C, D, H, W, ovox);
  } else {
    sample_fwd_kernel<0, true><<<g, TPB, 0, s>>>(x, grid, out, fixed, (double*)ws, C, D, H, W, ovox);
  }
  finalize_mean_kernel<<<1, TPB, 0, s>>>((const double*)ws, (int)(g.x * g.y),
                                         1.0 / ((double)N * C * (double)ovox), out_loss);
  return KMH_LAUNCH_CHECK();
}

namespace {
// dgrid *= g[0] unless g[0] == 1 (the usual loss.backward()): the fused pass already wrote d(loss)/d(grid)
__global__ __launch_bounds__(TPB) void scale_unless_one_kernel(float* __restrict__ a, long long n4,
                                                                const float* __restrict__ g) {
  const float s = g[0];
  if (s == 1.f) return;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
    float4 v = reinterpret_cast<float4*>(a)[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    reinterpret_cast<float4*>(a)[i] = v;
  }
}
}  // namespace

/* Fused align_img + MSELoss + their backward with respect to the grid, for a loss that IS the mean squared error
 * (keymorph/utils.py:14-21, loss_ops.py:9-13 and autograd of both; caller scripts/train.py:146-176): one pass writes
 * out (or nothing when out == NULL), out_loss[0] and dgrid = d(out_loss)/d(grid).  Returns KMH_EINVAL (-22) when the
 * lane-contiguous kernel does not apply to the shape: the caller then uses the separate entry points. */
KMH_API int kmh_warp_mse_fwd_grad(const float* x, const float* grid, const float* fixed, float* out, float* out_loss,
                                  float* dgrid, int N, int C, int D, int H, int W, int Do, int Ho, int Wo, void* ws,
                                  void* stream) {
  const long long ovox = (long long)Do * Ho * Wo;
  dim3 g(ceil_div(ovox, (long long)TPB * VPT), N);
  if ((long long)g.x * g.y > 65536 * 3 || !lane_contiguous_ok(D, H, W)) return -22;
  hipStream_t s = (hipStream_t)stream;
  const double cnt = (double)N * C * (double)ovox;
  sample_fwd_lc_kernel<0, true, true><<<g, TPB, 0, s>>>(x, grid, out, fixed, (double*)ws, C, D, H, W, ovox, dgrid,
                                                        (float)(2.0 / cnt));
  finalize_mean_kernel<<<1, TPB, 0, s>>>((const double*)ws, (int)(g.x * g.y), 1.0 / cnt, out_loss);
  return KMH_LAUNCH_CHECK();
}

/* a (n floats, n % 4 == 0, 16-byte aligned) *= g[0], skipped on the device when g[0] == 1 */
KMH_API int kmh_scale_unless_one(float* a, long long n, const float* g, void* stream) {
  if (n & 3) return -22;
  long long nb = (n / 4 + TPB - 1) / TPB;
  if (nb > 2048) nb = 2048;
  scale_unless_one_kernel<<<(int)nb, TPB, 0, (hipStream_t)stream>>>(a, n / 4, g);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_grid_sample3d_bwd_grid(const float* x, const float* grid, const float* gout, float* dgrid,
                                       int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                                       void* stream) {
  const long long ovox = (long long)Do * Ho * Wo;
  dim3 g(ceil_div(ovox, (long long)TPB * VPT), N);
  if (lane_contiguous_ok(D, H, W))
    sample_bwd_grid_lc_kernel<<<g, TPB, 0, (hipStream_t)stream>>>(x, grid, gout, dgrid, C, D, H, W, ovox);
  else
    sample_bwd_grid_kernel<<<g, TPB, 0, (hipStream_t)stream>>>(x, grid, gout, dgrid, C, D, H, W, ovox);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_grid_sample3d_bwd_input(const float* grid, const float* gout, float* dx, int N, int C, int D,
                                        int H, int W, int Do, int Ho, int Wo, void* stream) {
  const long long ovox = (long long)Do * Ho * Wo;
  dim3 g(ceil_div(ovox, TPB), N);
  sample_bwd_input_kernel<<<g, TPB, 0, (hipStream_t)stream>>>(grid, gout, dx, C, D, H, W, ovox);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_mse_fwd(const float* a, const float* b, long long n, float* out, void* ws, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int nb = (int)((n / 4 + TPB - 1) / TPB);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  if (nb < 1) nb = 1;
  sqdiff_partial_kernel<<<nb, TPB, 0, s>>>(a, b, n, (double*)ws);
  finalize_mean_kernel<<<1, TPB, 0, s>>>((const double*)ws, nb, 1.0 / (double)n, out);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_mse_bwd(const float* a, const float* b, const float* gscale, long long n, float* da,
                        void* stream) {
  int nb = (int)((n + TPB - 1) / TPB);
  if (nb > 4096) nb = 4096;
  mse_bwd_kernel<<<nb, TPB, 0, (hipStream_t)stream>>>(a, b, gscale, n, da);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_dice_sums(const float* pred, const float* target, int R, long long V, float* sums, void* ws,
                          void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int nb = (int)((V + TPB * 8 - 1) / (TPB * 8));
  int cap = 65536 / (R > 0 ? R : 1);
  if (cap < 1) return -22;
  if (nb > cap) nb = cap;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  dice_partial_kernel<<<dim3(nb, R), TPB, 0, s>>>(pred, target, V, (double*)ws);
  dice_finalize_kernel<<<R, TPB, 0, s>>>((const double*)ws, nb, sums);
  return KMH_LAUNCH_CHECK();
}

/* Fused align_img + soft Dice sums: sums[(n*C + c)*3 + {0,1,2}] = {sum t p, sum p^2, sum t^2} over the output voxels, with
 * p = grid_sample(x, grid)[n, c] (bilinear, border, align_corners = False) and t = fixed[n, c]; the warped tensor is never
 * written.  Replaces keymorph/utils.py:14-21 followed by the three reductions of keymorph/loss_ops.py:28-52 (caller
 * scripts/train.py:146-164).  ws: kmh_reduce_ws_bytes().  Returns KMH_EINVAL (-22) when the lane-contiguous kernel does
 * not apply (W < 2, a plane of >= 2^31 voxels, C > 128): the caller then uses the separate entry points. */
/* The ONE statement of when the fused warp + Dice kernels apply (both entry points return -22 otherwise, and
 * kmh_warp_dice_ok lets the host decide BEFORE it builds an autograd node: the unfused composition align_img + DiceLoss is
 * the documented fallback): the lane-contiguous sampler (W >= 2, not switched off by KMH_SAMPLER_OLD), < 2^30 voxels per
 * channel plane (32-bit byte offsets), <= 128 channels (the sums' LDS table), and N * C rows whose block partials --
 * (N, nb, C, 3) doubles with nb >= 1 -- fit the reduction workspace. */
static bool warp_dice_supported(int N, int C, int D, int H, int W) {
  return N > 0 && C > 0 && C <= WD_MAXC && lane_contiguous_ok(D, H, W) && (long long)D * H * W < (1ll << 30) &&
         (long long)N * C <= 65536;
}
KMH_API int kmh_warp_dice_ok(int N, int C, int D, int H, int W) { return warp_dice_supported(N, C, D, H, W) ? 1 : 0; }

KMH_API int kmh_warp_dice_sums(const float* x, const float* grid, const float* fixed, float* sums, int N, int C, int D,
                               int H, int W, int Do, int Ho, int Wo, const unsigned char* lab_x,
                               const unsigned char* lab_fixed, const int* gate, void* ws, void* stream) {
  if (!warp_dice_supported(N, C, D, H, W)) return -22;
  const bool labs = lab_x && lab_fixed && gate;
  if (!labs && (lab_x || lab_fixed || gate)) return -22;            // all three or none
  const long long ovox = (long long)Do * Ho * Wo;
  const int nchunk = ceil_div(ovox, (long long)TPB * PASSES);
  const long long nb_cap = 65536 / ((long long)N * C);   // partial (N, nb, C, 3) doubles inside the reduction workspace
  long long nb = nb_cap;                                 // (>= 1: warp_dice_supported)
  if (nb > nchunk) nb = nchunk;
  static const int capa = getenv("KMH_WD_BLOCKS") ? atoi(getenv("KMH_WD_BLOCKS")) : 768;   // ~ resident blocks of the chip
  const int per_n = (capa / N) & ~7;                 // a multiple of 8 per sample row: blockIdx.x % 8 is the XCD
  if (nb > per_n) nb = per_n < 8 ? 8 : per_n;
  if (nb > nb_cap) nb = nb_cap;                      // (per_n < 8 rounds UP to 8: never past the workspace)
  if (nb < 1) nb = 1;
  hipStream_t s = (hipStream_t)stream;
  static const int ilp = getenv("KMH_WD_ILP_A") ? atoi(getenv("KMH_WD_ILP_A")) : 4;        // A/B switch (tools/bench_warp_dice.py)
  const dim3 g((unsigned)nb, N);
  // with label maps: BOTH variants are launched with the same grid; the device flag lets exactly one of them work
  if (labs)
    warp_dice_sums_kernel<4, true><<<g, TPB, 0, s>>>(x, grid, fixed, (double*)ws, C, D, H, W, ovox, nchunk, lab_x,
                                                         lab_fixed, gate);
  if (ilp == 4)
    warp_dice_sums_kernel<4><<<g, TPB, 0, s>>>(x, grid, fixed, (double*)ws, C, D, H, W, ovox, nchunk, nullptr, nullptr,
                                                  labs ? gate : nullptr);
  else
    warp_dice_sums_kernel<2><<<g, TPB, 0, s>>>(x, grid, fixed, (double*)ws, C, D, H, W, ovox, nchunk, nullptr, nullptr,
                                                  labs ? gate : nullptr);
  const int total = N * C * 3;
  warp_dice_final_kernel<<<ceil_div(total, TPB / kWave), TPB, 0, s>>>((const double*)ws, (int)nb, C, total, sums);
  return KMH_LAUNCH_CHECK();
}

/* x (N, C, V) floats -> lab (N, V) bytes = the channel that holds the 1 when every voxel is exactly one-hot; ok[0] (device
 * int, this call only ever CLEARS it: preset it to 1, chain several tensors onto one flag) stays 1 iff that held
 * everywhere.  What keymorph/utils.py:200-240 (one_hot / one_hot_subsampled_pair) and nearest-sampled augmentation
 * (keymorph/augmentation.py:160-163) produce is exactly one-hot; a soft segmentation clears the flag and the Dice kernels
 * then read the float tensors.  C <= 255. */
KMH_API int kmh_onehot_to_labels(const float* x, int N, int C, long long V, unsigned char* lab, int* ok, void* stream) {
  if (N <= 0 || C <= 0 || C > 255 || V <= 0) return -22;
  long long nb = (V / 4 + TPB - 1) / TPB;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  onehot_to_labels_kernel<<<dim3((unsigned)nb, N), TPB, 0, (hipStream_t)stream>>>(x, C, V, lab, ok);
  return KMH_LAUNCH_CHECK();
}

/* d/d(grid) of sum_{n,c} g[n,c] * DiceRow[n,c] given ca[n*C+c] = -2 g / den and cb[n*C+c] = 2 g num / den^2 (num = 2 I + 1,
 * den = P + T + 1 from kmh_warp_dice_sums): dgrid[n, v, :] = sum_c (ca t + cb p) * d p / d grid, p recomputed from x.
 * Autograd of keymorph/loss_ops.py:16-63 through keymorph/utils.py:14-21 in one pass. */
KMH_API int kmh_warp_dice_bwd_grid(const float* x, const float* grid, const float* fixed, const float* ca, const float* cb,
                                   float* dgrid, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                                   const unsigned char* lab_x, const unsigned char* lab_fixed, const int* gate, void* stream) {
  if (!warp_dice_supported(N, C, D, H, W)) return -22;
  const bool labs = lab_x && lab_fixed && gate;
  if (!labs && (lab_x || lab_fixed || gate)) return -22;
  const long long ovox = (long long)Do * Ho * Wo;
  const int nchunk = ceil_div(ovox, (long long)TPB * PASSES);
  static const int cap = getenv("KMH_WD_BLOCKS") ? atoi(getenv("KMH_WD_BLOCKS")) : 768;
  static const int ilp = getenv("KMH_WD_ILP_B") ? atoi(getenv("KMH_WD_ILP_B")) : 4;      // 1.97 ms vs 2.58 (ILP 2) at 2 x 14 x 256^3
  int per_n = (cap / N) & ~7;
  if (per_n < 8) per_n = 8;
  int nb = nchunk < per_n ? nchunk : per_n;
  const dim3 g(nb, N);
  hipStream_t s = (hipStream_t)stream;
  if (labs)
    warp_dice_grad_kernel<2, true><<<g, TPB, 0, s>>>(x, grid, fixed, ca, cb, dgrid, C, D, H, W, ovox, nchunk, lab_x,
                                                         lab_fixed, gate);
  if (ilp == 4)
    warp_dice_grad_kernel<4><<<g, TPB, 0, s>>>(x, grid, fixed, ca, cb, dgrid, C, D, H, W, ovox, nchunk, nullptr, nullptr,
                                                  labs ? gate : nullptr);
  else
    warp_dice_grad_kernel<2><<<g, TPB, 0, s>>>(x, grid, fixed, ca, cb, dgrid, C, D, H, W, ovox, nchunk, nullptr, nullptr,
                                                  labs ? gate : nullptr);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_rows_axpby(const float* t, const float* p, const float* ca, const float* cb, int R,
                           long long V, float* out, void* stream) {
  int nb = (int)((V + TPB * 4 - 1) / (TPB * 4));
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  rows_axpby_kernel<<<dim3(nb, R), TPB, 0, (hipStream_t)stream>>>(t, p, ca, cb, V, out);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_argmax_onehot(const float* pred, int N, int C, long long V, float* out, void* stream) {
  argmax_onehot_kernel<<<dim3(ceil_div(V, TPB), N), TPB, 0, (hipStream_t)stream>>>(pred, C, V, out);
  return KMH_LAUNCH_CHECK();
}

/* disp: 3 components of a (D,H,W) map, component c at disp + c*cstride, voxel v at + v*vstride (NCDHW: cstride =
 * D*H*W, vstride = 1; a permuted (D,H,W,3) grid: cstride = 1, vstride = 3).  jd (D-4,H-4,W-4) or NULL;
 * stats[4] doubles = {mean, std (ddof 0), #(det <= 0), #voxels}.  keymorph/loss_ops.py:161-247 */
KMH_API int kmh_jacobian_det(const float* disp, long long cstride, long long vstride, int D, int H, int W, float* jd,
                             double* stats, void* ws, void* stream) {
  if (D < 5 || H < 5 || W < 5) return -22;
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)(D - 4) * (H - 4) * (W - 4);
  int nb = ceil_div(total, TPB);
  if (nb > 4096) nb = 4096;
  jacdet_kernel<<<nb, TPB, 0, s>>>(disp, cstride, vstride, D, H, W, jd, (double*)ws);
  jacdet_final_kernel<<<1, TPB, 0, s>>>((const double*)ws, nb, (double)total, stats);
  return KMH_LAUNCH_CHECK();
}
