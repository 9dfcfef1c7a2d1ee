// First U-Net convolution (Cin = 1): the correlations its backward needs, as a dedicated fp32 VALU kernel.
//   R[co][tap] = sum_v x[v + tap] dz[v][co]          S[co][tap] = sum_v [v + tap inside the volume] dz[v][co]
// (x = the RAW 1-channel input, dz = gradient wrt the conv output, optionally masked by dzmask > 0).  From R and S,
// kmh_conv3d_first_layer_fold forms dW = scale R + shift S and GroupNorm's (sum dxn, sum dxn x) without the
// 1-channel data gradient (reference: the autograd of keymorph/unet3d/buildingblocks.py:46-78 for encoders[0]).
//
// M = 54 rows x N = 16 columns x K = 16.7 M voxels is a hopeless shape for the MFMA weight-gradient kernel (its launch
// is 100 % staging); here one thread owns one (co, kz, ky, half of the x range) and streams along x with a 3-tap
// register window (two x halves: 9 instead of 4.5 compute waves per 65 KB of LDS -- a 4-way split adds nothing):
// per 4 voxels 2 x ds_read_b128 (the x row, and a TRANSPOSED dz row [co][x]) and 12 fp32 FMAs; S needs only the
// row sum of dz and two end corrections.  Exact fp32, independent of the convolution arithmetic mode.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int FL_XS = 2;         // the x range of a row is split over FL_XS thread groups (more waves per 65 KB of LDS)
constexpr int FL_TPB = 320;      // 2 x 144 compute threads (16 co x 9 (kz, ky) per x half) + staging helpers
constexpr int FL_YR = 8;         // output rows per workgroup
constexpr int FL_CO = 16;

// c123 (N, Cout, 3) | NULL: GroupNorm's backward of the NEXT layer applied while dz is staged -- dz then holds that layer's
// normalised-input gradient dxn and dzmask its input (this layer's output y), and the gradient that enters the
// correlations is [y > 0] (c1 dxn + c2 y + c3): the 256^3 x 16-channel gradient is never written by a separate pass
// (kmh_gn_bwd_apply: one read of dxn and y, one write) and read back here.
__global__ __launch_bounds__(FL_TPB) void first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                            const float* __restrict__ dzmask,
                                                            float* __restrict__ partial /* (nblk, Cout, 2, 27) */, int D,
                                                            int H, int W, int Cout, int ytiles,
                                                            const float* __restrict__ c123) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const int WP = (W + 3) & ~3;                 // row length rounded to 4
  const int XP = WP + 8;                       // x row pitch: [0..3] left halo (index 3 = x -1), data at 4.., right halo
  const int DP = WP + 4;                       // dz row pitch: +4 floats so that the 4 channel quads of a voxel and
                                               // consecutive voxels spread over the LDS banks (2-way instead of 16-way)
  float* xs = fsm;                             // [3][FL_YR + 2][XP]
  float* ds = fsm + 3 * (FL_YR + 2) * XP;      // [2][FL_CO][DP]   (double buffered, transposed)
  const int tid = threadIdx.x;
  const int n = blockIdx.z, z = blockIdx.y, y0 = blockIdx.x * FL_YR;
  const float* xn = x + (long long)n * D * H * W;
  const float* dn = dz + (long long)n * D * H * W * Cout;
  const float* mn = dzmask ? dzmask + (long long)n * D * H * W * Cout : nullptr;
  const bool v4 = ((W & 3) == 0) && ((Cout & 3) == 0);

  // ---- input window: planes z-1..z+1, rows y0-1..y0+FL_YR, zero outside the volume (and in the x halos)
  {
    const int q_per_row = XP / 4;              // float4 slots per row (first and last are the halos)
    for (int e = tid; e < 3 * (FL_YR + 2) * q_per_row; e += FL_TPB) {
      const int qi = e % q_per_row, r = e / q_per_row;
      const int ry = r % (FL_YR + 2), rz = r / (FL_YR + 2);
      const int gx = 4 * qi - 4, gy = y0 + ry - 1, gz = z + rz - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D && gx >= 0 && gx < W) {
        const float* src = xn + ((long long)gz * H + gy) * W + gx;
        if (v4) v = *reinterpret_cast<const float4*>(src);
        else {
          v.x = src[0];
          if (gx + 1 < W) v.y = src[1];
          if (gx + 2 < W) v.z = src[2];
          if (gx + 3 < W) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(xs + r * XP + 4 * qi) = v;
    }
  }
  // dz row (z, y0 + yy) -> ds[buf][co][x] (zero beyond W / H / Cout), split into "issue the global loads" and "write the
  // transposed LDS row": the loads of row yy+1 are in flight while row yy is multiplied (their latency used to be
  // exposed once per row)
  constexpr int NPRE = 4;                      // float4 items per thread: WP * 4 / FL_TPB <= 4 for W <= 320
  float4 pre[NPRE];
  // this thread's channel quad is the same for all its items (FL_TPB % 4 == 0): its GroupNorm-backward coefficients
  float gc[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * (tid & 3) + j;
#pragma unroll
    for (int k = 0; k < 3; ++k) gc[j][k] = (c123 && c < Cout) ? c123[((long long)n * Cout + c) * 3 + k] : 0.f;
  }
  auto load_dz = [&](int yy) {
    const int gy = y0 + yy;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = tid + k * FL_TPB;
      const int q = e & 3, xx = e >> 2;                    // lanes: the 4 channel quads of a voxel, then voxels
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < WP * (FL_CO / 4) && xx < W && gy < H && 4 * q < Cout) {
        const long long off = (((long long)z * H + gy) * W + xx) * Cout + 4 * q;
        v = *reinterpret_cast<const float4*>(dn + off);
        if (mn) {
          const float4 m = *reinterpret_cast<const float4*>(mn + off);
          if (c123) {            // the same expression as gn_bwd_apply_kernel: c1 dxn + c2 x + c3, then the ReLU mask
            v.x = gc[0][0] * v.x + gc[0][1] * m.x + gc[0][2];
            v.y = gc[1][0] * v.y + gc[1][1] * m.y + gc[1][2];
            v.z = gc[2][0] * v.z + gc[2][1] * m.z + gc[2][2];
            v.w = gc[3][0] * v.w + gc[3][1] * m.w + gc[3][2];
          }
          if (!(m.x > 0.f)) v.x = 0.f;
          if (!(m.y > 0.f)) v.y = 0.f;
          if (!(m.z > 0.f)) v.z = 0.f;
          if (!(m.w > 0.f)) v.w = 0.f;
        }
      }
      pre[k] = v;
    }
  };
  auto store_dz = [&](int buf) {
    float* dst = ds + buf * FL_CO * DP;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = tid + k * FL_TPB;
      const int q = e & 3, xx = e >> 2;
      if (e < WP * (FL_CO / 4)) {
        dst[(4 * q + 0) * DP + xx] = pre[k].x; dst[(4 * q + 1) * DP + xx] = pre[k].y;
        dst[(4 * q + 2) * DP + xx] = pre[k].z; dst[(4 * q + 3) * DP + xx] = pre[k].w;
      }
    }
  };
  auto stage_dz_scalar = [&](int yy, int buf) {
    float* dst = ds + buf * FL_CO * DP;
    const int gy = y0 + yy;
    for (int e = tid; e < WP * FL_CO; e += FL_TPB) {
      const int co = e % FL_CO, xx = e / FL_CO;
      float v = 0.f;
      if (xx < W && gy < H && co < Cout) {
        const long long off = (((long long)z * H + gy) * W + xx) * Cout + co;
        v = dn[off];
        if (mn && c123) {
          const float* cc = c123 + ((long long)n * Cout + co) * 3;
          v = cc[0] * v + cc[1] * mn[off] + cc[2];
        }
        if (mn && !(mn[off] > 0.f)) v = 0.f;
      }
      dst[co * DP + xx] = v;
    }
  };
  const bool pipelined = v4 && WP * (FL_CO / 4) <= NPRE * FL_TPB;
  if (pipelined) { load_dz(0); store_dz(0); } else stage_dz_scalar(0, 0);
  __syncthreads();

  // compute threads: (co, (kz, ky), x part); the split needs part boundaries on float4 slots
  const int nxs = (WP % (4 * FL_XS) == 0) ? FL_XS : 1;
  const int co = tid % FL_CO, kzky = (tid / FL_CO) % 9, part = tid / (FL_CO * 9);
  const bool computes = part < nxs;
  const int xbeg = part * (WP / nxs), xend = xbeg + WP / nxs;
  const int kz = kzky / 3, ky = kzky % 3;
  float R[3] = {0.f, 0.f, 0.f}, S[3] = {0.f, 0.f, 0.f};
  for (int yy = 0; yy < FL_YR; ++yy) {
    if (yy + 1 < FL_YR) {                                          // next row into the other buffer
      if (pipelined) load_dz(yy + 1); else stage_dz_scalar(yy + 1, (yy + 1) & 1);
    }
    if (computes && y0 + yy < H) {
      const float* xr = xs + (kz * (FL_YR + 2) + yy + ky) * XP + 4;   // x row (z + kz - 1, y + ky - 1), index 0 = x 0
      const float* dr = ds + ((yy & 1) * FL_CO + co) * DP;
      const int gy = y0 + yy + ky - 1, gz = z + kz - 1;
      const bool rowin = (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D;
      float prev = xr[xbeg - 1];                                   // x[xbeg - 1] (the zero left halo for part 0)
      float4 cur = *reinterpret_cast<const float4*>(xr + xbeg);
      float rowsum = 0.f;
      for (int xx = xbeg; xx < xend; xx += 4) {
        const float4 d = *reinterpret_cast<const float4*>(dr + xx);
        const float4 nxt = *reinterpret_cast<const float4*>(xr + xx + 4);   // right halo is zero padded
        R[0] += prev * d.x + cur.x * d.y + cur.y * d.z + cur.z * d.w;       // tap kx = -1: x[v - 1]
        R[1] += cur.x * d.x + cur.y * d.y + cur.z * d.z + cur.w * d.w;      // kx = 0
        R[2] += cur.y * d.x + cur.z * d.y + cur.w * d.z + nxt.x * d.w;      // kx = +1
        rowsum += (d.x + d.y) + (d.z + d.w);
        prev = cur.w;
        cur = nxt;
      }
      if (rowin) {
        S[1] += rowsum;
        S[0] += rowsum - (xbeg == 0 ? dr[0] : 0.f);                          // kx = -1 leaves the volume at x = 0
        S[2] += rowsum - ((W - 1 >= xbeg && W - 1 < xend) ? dr[W - 1] : 0.f); // kx = +1 leaves it at x = W - 1
      }
    }
    if (pipelined && yy + 1 < FL_YR) store_dz((yy + 1) & 1);
    __syncthreads();
  }
  if (part < FL_XS && co < Cout) {             // (parts beyond nxs wrote nothing into R / S: zeros)
    float* o = partial + (((((long long)n * gridDim.y + z) * ytiles + blockIdx.x) * FL_XS + part) * Cout + co) * 54;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      o[(kz * 3 + ky) * 3 + kx] = R[kx];
      o[27 + (kz * 3 + ky) * 3 + kx] = S[kx];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the same correlations on the fp32 MATRIX cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation --
// the arithmetic class of the VALU kernel above, another summation order).  R is a GEMM after all, just a thin one:
//   R[tap][co] = sum_v X[tap][v] dz[v][co],   X[tap][v] = x[v + tap]:  M = 27 taps (two 16-row tiles), N = 16, K = voxels.
// The VALU kernel reads 32 bytes of LDS per 12 multiply-adds and runs at 0.22 of the VALU peak (LDS-read bound); here a wave
// takes 4 consecutive voxels per step: A = 2 x ds_read_b32 (lane = (tap, voxel): a gather from the 3-plane input window),
// B = one ds_read_b32 of the [x][co] gradient row, 2 MFMAs = 1728 multiply-adds.  Row 27 of the A operand is all ones: its
// accumulator is the row sum of dz that the indicator correlations S need (reset after every row); S itself stays on the
// VALU (144 threads x 3 adds per row).  With the GroupNorm backward of the next layer applied while the gradient row is
// staged (c123), the kernel is bound by reading dxn and y once: 8.6 GB per step.
constexpr int FM_TPB = 256;      // 4 waves: each takes every 4th group of 4 voxels of a row
constexpr int FM_YR = 8;         // output rows per workgroup
typedef float fm_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(FM_TPB) void first_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                 const float* __restrict__ dzmask,
                                                                 float* __restrict__ partial /* (nblk, Cout, 2, 27) */,
                                                                 int D, int H, int W, int Cout, int ytiles,
                                                                 const float* __restrict__ c123) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const int WP = (W + 3) & ~3;
  const int XP = WP + 8;                          // x row pitch: data at index 4.., zero halos either side
  float* xs = fsm;                                // [3][FM_YR + 2][XP]
  float* ds = fsm + 3 * (FM_YR + 2) * XP;         // [2][WP][16]   gradient row, [x][co]
  float* srow = ds + (2 * WP * 16 > 4096 ? 2 * WP * 16 : 4096);   // [2][4 waves][16]  per-wave row sums of dz (by row parity)
  float* sfl = srow + 2 * 4 * 16;                 // [2][2][16]        first / last voxel of the row (by row parity)
  // (the final cross-wave reduction, [4 waves][2 tiles][64 lanes][4] doubles = 16 KB, re-uses the gradient-row buffers: with
  // its own 16 KB the image was 81.6 KB and only ONE workgroup fitted a CU -- 32 KB of loads in flight per CU)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = blockIdx.z, z = blockIdx.y, y0 = blockIdx.x * FM_YR;
  const float* xn = x + (long long)n * D * H * W;
  const float* dn = dz + (long long)n * D * H * W * Cout;
  const float* mn = dzmask ? dzmask + (long long)n * D * H * W * Cout : nullptr;
  const bool v4 = ((W & 3) == 0) && ((Cout & 3) == 0);

  // ---- input window: planes z-1..z+1, rows y0-1..y0+FM_YR, zero outside the volume (and in the x halos)
  {
    const int q_per_row = XP / 4;
    for (int e = tid; e < 3 * (FM_YR + 2) * q_per_row; e += FM_TPB) {
      const int qi = e % q_per_row, r = e / q_per_row;
      const int ry = r % (FM_YR + 2), rz = r / (FM_YR + 2);
      const int gx = 4 * qi - 4, gy = y0 + ry - 1, gz = z + rz - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D && gx >= 0 && gx < W) {
        const float* src = xn + ((long long)gz * H + gy) * W + gx;
        if (v4) v = *reinterpret_cast<const float4*>(src);
        else {
          v.x = src[0];
          if (gx + 1 < W) v.y = src[1];
          if (gx + 2 < W) v.z = src[2];
          if (gx + 3 < W) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(xs + r * XP + 4 * qi) = v;
    }
  }
  // ---- gradient row (z, y0 + yy) -> ds[buf][x][co]: global loads split from the LDS stores (the next row is in flight
  // under this row's MFMAs); the pending GroupNorm backward and the ReLU mask are applied on the way
  constexpr int NPRE = 4;                         // float4 items per thread: WP * 4 / FM_TPB <= 4 for W <= 256
  // two rows in flight (row yy + 1 and row yy + 2), RAW: gradient and mask operand as loaded.  The GroupNorm backward and the
  // ReLU mask are applied when the row is written to LDS, a row later -- applied at load time (round 3's first version) every
  // load_dz waited for its own loads, and the kernel ran at one HBM round trip per row (5.4 us; 3.1 TB/s)
  float4 preA[NPRE], preB[NPRE], mskA[NPRE], mskB[NPRE];
  float gc[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * (tid & 3) + j;
#pragma unroll
    for (int k = 0; k < 3; ++k) gc[j][k] = (c123 && c < Cout) ? c123[((long long)n * Cout + c) * 3 + k] : 0.f;
  }
  auto load_dz = [&](int yy, float4 (&pre)[NPRE], float4 (&msk)[NPRE]) {
    const int gy = y0 + yy;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = tid + k * FM_TPB;
      const int q = e & 3, xx = e >> 2;
      pre[k] = msk[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < WP * 4 && xx < W && gy < H && 4 * q < Cout) {
        const long long off = (((long long)z * H + gy) * W + xx) * Cout + 4 * q;
        pre[k] = *reinterpret_cast<const float4*>(dn + off);
        if (mn) msk[k] = *reinterpret_cast<const float4*>(mn + off);
      }
    }
  };
  auto store_dz = [&](int buf, const float4 (&pre)[NPRE], const float4 (&msk)[NPRE]) {
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = tid + k * FM_TPB;
      float4 v = pre[k];
      if (mn) {
        const float4 m = msk[k];
        if (c123) {            // the same expression as gn_bwd_apply_kernel
          v.x = gc[0][0] * v.x + gc[0][1] * m.x + gc[0][2];
          v.y = gc[1][0] * v.y + gc[1][1] * m.y + gc[1][2];
          v.z = gc[2][0] * v.z + gc[2][1] * m.z + gc[2][2];
          v.w = gc[3][0] * v.w + gc[3][1] * m.w + gc[3][2];
        }
        if (!(m.x > 0.f)) v.x = 0.f;          // (items outside the volume carry m = 0: they stay zero)
        if (!(m.y > 0.f)) v.y = 0.f;
        if (!(m.z > 0.f)) v.z = 0.f;
        if (!(m.w > 0.f)) v.w = 0.f;
      }
      if (e < WP * 4) *reinterpret_cast<float4*>(ds + (buf * WP + (e >> 2)) * 16 + 4 * (e & 3)) = v;
    }
  };
  auto stage_dz_scalar = [&](int yy, int buf) {
    const int gy = y0 + yy;
    for (int e = tid; e < WP * 16; e += FM_TPB) {
      const int co = e & 15, xx = e >> 4;
      float v = 0.f;
      if (xx < W && gy < H && co < Cout) {
        const long long off = (((long long)z * H + gy) * W + xx) * Cout + co;
        v = dn[off];
        if (mn && c123) {
          const float* cc = c123 + ((long long)n * Cout + co) * 3;
          v = cc[0] * v + cc[1] * mn[off] + cc[2];
        }
        if (mn && !(mn[off] > 0.f)) v = 0.f;
      }
      ds[(buf * WP + xx) * 16 + co] = v;
    }
  };
  const bool pipelined = v4 && WP * 4 <= NPRE * FM_TPB;
  if (pipelined) { load_dz(0, preA, mskA); store_dz(0, preA, mskA); load_dz(1, preB, mskB); } else stage_dz_scalar(0, 0);

  // ---- MFMA operands of this lane: A row i = tap (two tiles), K index k = voxel within the group of 4
  const int ai = lane & 15, ak = lane >> 4;
  const int t0 = ai, t1 = 16 + ai;                       // taps of the two M tiles (t1 >= 27: padding; t1 == 27: the ones row)
  const int offA0 = ((t0 / 9) * (FM_YR + 2) + (t0 / 3) % 3) * XP + 3 + t0 % 3 + ak;
  const int offA1 = t1 < 27 ? ((t1 / 9) * (FM_YR + 2) + (t1 / 3) % 3) * XP + 3 + t1 % 3 + ak : -1;
  const float a1_const = t1 == 27 ? 1.f : 0.f;
  const int offB = ak * 16 + ai;
  fm_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  double racc0[4] = {0.0, 0.0, 0.0, 0.0}, racc1[4] = {0.0, 0.0, 0.0, 0.0};
  // S: thread (co, (kz, ky)) as in the VALU kernel, fed by the row sums
  const int sco = tid & 15, skzky = tid >> 4;
  const int skz = skzky / 3, sky = skzky % 3;
  double S[3] = {0.0, 0.0, 0.0};
  auto s_update = [&](int yy) {                          // row yy's sums are complete (a barrier ago)
    if (tid < 144 && y0 + yy < H) {
      const float* sr = srow + (yy & 1) * 64;
      const double rowsum = ((double)sr[sco] + (double)sr[16 + sco]) + ((double)sr[32 + sco] + (double)sr[48 + sco]);
      const double first = sfl[(yy & 1) * 32 + sco], last = sfl[(yy & 1) * 32 + 16 + sco];
      const int gy = y0 + yy + sky - 1, gz = z + skz - 1;
      if ((unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D) {
        S[1] += rowsum;
        S[0] += rowsum - first;                          // kx = -1 leaves the volume at x = 0
        S[2] += rowsum - last;                           // kx = +1 leaves it at x = W - 1
      }
    }
  };
  __syncthreads();
  static_assert(FM_YR % 2 == 0, "the row loop is unrolled by two (register double-buffer of the prefetched rows)");
#pragma unroll
  for (int yy = 0; yy < FM_YR; ++yy) {
    // rows yy + 1 (issued an iteration ago) and yy + 2 (issued now) are in flight under this row's MFMAs
    if (pipelined) {
      if (yy + 2 < FM_YR) { if (yy & 1) load_dz(yy + 2, preB, mskB); else load_dz(yy + 2, preA, mskA); }
    } else if (yy + 1 < FM_YR) stage_dz_scalar(yy + 1, (yy + 1) & 1);
    if (yy > 0) s_update(yy - 1);
    const float* xr = xs + yy * XP;
    const float* dr = ds + (yy & 1) * WP * 16;
    for (int x0 = 4 * wv; x0 < WP; x0 += 16) {
      const float b = dr[x0 * 16 + offB];
      const float a0 = xr[offA0 + x0];
      const float a1 = offA1 >= 0 ? xr[offA1 + x0] : a1_const;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc1, 0, 0, 0);
    }
    // the ones row (tile 1, row 11 = lanes 32..47, register 3) holds this wave's share of the row sums: hand it over
    if (ak == 2) srow[(yy & 1) * 64 + wv * 16 + ai] = acc1[3];
    // fp32 accumulation chains end with the row (64 voxels per wave): the row's sums go into fp64 accumulators
#pragma unroll
    for (int r = 0; r < 4; ++r) { racc0[r] += (double)acc0[r]; racc1[r] += (double)acc1[r]; acc0[r] = 0.f; acc1[r] = 0.f; }
    if (tid < 16) { sfl[(yy & 1) * 32 + tid] = dr[tid]; sfl[(yy & 1) * 32 + 16 + tid] = dr[(W - 1) * 16 + tid]; }
    if (pipelined && yy + 1 < FM_YR) { if (yy & 1) store_dz((yy + 1) & 1, preA, mskA); else store_dz((yy + 1) & 1, preB, mskB); }
    __syncthreads();
  }
  s_update(FM_YR - 1);
  // ---- R: sum the 4 waves' accumulators; element (tap, co) sits in lane (row / 4) * 16 + co, register row % 4
  // (the cross-wave sum is formed in fp64 as well; every wave is past the last row's barrier: ds is free)
  double* dred = reinterpret_cast<double*>(ds);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dred[((wv * 2 + 0) * 64 + lane) * 4 + r] = racc0[r];
    dred[((wv * 2 + 1) * 64 + lane) * 4 + r] = racc1[r];
  }
  __syncthreads();
  float* o = partial + ((((long long)n * gridDim.y + z) * ytiles + blockIdx.x)) * Cout * 54;
  for (int e = tid; e < 27 * 16; e += FM_TPB) {
    const int co = e & 15, t = e >> 4;
    const int tile = t >> 4, row = t & 15;
    const int l = (row >> 2) * 16 + co, r = row & 3;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += dred[((w * 2 + tile) * 64 + l) * 4 + r];
    if (co < Cout) o[co * 54 + t] = (float)v;
  }
  if (tid < 144 && sco < Cout) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) o[sco * 54 + 27 + (skz * 3 + sky) * 3 + kx] = (float)S[kx];
  }
}

// rs[n][e] = sum over the sample's workgroups, fixed order in fp64
__global__ __launch_bounds__(256) void first_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int per,
                                                                double* __restrict__ rs) {
  const int n = blockIdx.y;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= per) return;
  const int lane = threadIdx.x & 63;
  const float* p = partial + (long long)n * nblk * per + e;
  double s4[4] = {0, 0, 0, 0};
  int b = lane;
  for (; b + 3 * 64 < nblk; b += 4 * 64) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s4[k] += p[(long long)(b + k * 64) * per];
  }
  for (; b < nblk; b += 64) s4[0] += p[(long long)b * per];
  const double s = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
  if (lane == 0) rs[(long long)n * per + e] = s;
}


// ---------------------------------------------------------------------------------------------------------------
// Forward of the same layer: y = relu(conv3(x * scale[n] + shift[n]))  (1 -> Cout <= 16 channels, zero padding AFTER
// the normalisation), exact fp32 on the VALU.  The MFMA kernel spends this launch padding one input channel to eight
// and is bound by its 4.3 GB of output anyway; here a thread owns (voxel column, quad of 4 output channels): its 108
// filter taps live in registers, a 3 x 3 x 3 input window slides down y (9 LDS reads + 54 packed FMAs per voxel), and
// a wave's 16-byte stores cover 1 KB of contiguous NDHWC output.  Also emits the per-block (sum y, sum y^2) pairs of
// the next GroupNorm (the conv kernels' epilogue statistics).
constexpr int FF_X = 32, FF_Y = 9;           // output tile: 32 voxels along x, 9 rows (3 blocks of 3: see the row loop)
constexpr int FF_Z = 16;                     // consecutive z planes per workgroup (filter taps loaded once)
constexpr int FF_CPT = 2;                    // channels per thread: 2 keeps the 27 packed taps + the 27-value window at
                                             // ~100 registers (4 needed 252: two workgroups per CU, stores and LDS
                                             // reads exposed)
constexpr int FF_SUB = 16 / FF_CPT;          // threads per voxel column
constexpr int FF_TPB = FF_X * FF_SUB;        // 32 voxel columns x 8 channel pairs = 256 threads
constexpr int FF_P = FF_X + 2;               // halo row pitch (floats)
constexpr int FF_TILE = 3 * (FF_Y + 2) * FF_P;
constexpr int FF_NLD = (FF_TILE + FF_TPB - 1) / FF_TPB;   // tile elements per thread

__global__ __launch_bounds__(FF_TPB) void first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          float* __restrict__ y, double* __restrict__ stats_partial,
                                                          int D, int H, int W, int Cout, int xt, int yt) {
  static_assert(FF_CPT == 2, "one packed pair of channels per thread");
  __shared__ float sx[2][FF_TILE];             // double-buffered [3 planes][FF_Y + 2][FF_P] window
  __shared__ double sred[FF_TPB / kWave][16][2];
  const int tid = threadIdx.x, q = tid % FF_SUB, vx = tid / FF_SUB;
  const int n = blockIdx.z, zc = blockIdx.y * FF_Z;
  const int bx = blockIdx.x % xt, by = blockIdx.x / xt;
  const int x0 = bx * FF_X, y0 = by * FF_Y;
  const float* xn = x + (long long)n * D * H * W;
  const float sc = scale ? scale[n] : 1.f, sh = scale ? shift[n] : 0.f;
  // tile element e -> (lz, ly, lx); the global load of plane window z is split from its LDS store so that the next
  // window is in flight while this one is multiplied
  // (the normalisation is applied when the window is written to LDS: applied to the loaded value here, the load would have
  // to land inside load_tile and every plane would start with an exposed memory round trip)
  float pre[FF_NLD];
  unsigned pre_in = 0u;                        // bit k: element k of the window lies inside the volume
  auto load_tile = [&](int z) {
    pre_in = 0u;
#pragma unroll
    for (int k = 0; k < FF_NLD; ++k) {
      const int e = tid + k * FF_TPB;
      const int lx = e % FF_P, r = e / FF_P, ly = r % (FF_Y + 2), lz = r / (FF_Y + 2);
      const int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = z + lz - 1;
      float v = 0.f;
      if (e < FF_TILE && (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D) {
        v = xn[((long long)gz * H + gy) * W + gx];
        pre_in |= 1u << k;
      }
      pre[k] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int k = 0; k < FF_NLD; ++k) {
      const int e = tid + k * FF_TPB;
      if (e < FF_TILE) sx[buf][e] = ((pre_in >> k) & 1u) ? pre[k] * sc + sh : 0.f;      // zero padding AFTER the normalisation
    }
  };
  load_tile(zc);
  // this thread's filter taps: channels 2q, 2q+1 as one packed pair
  kmh_f2 w01[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    const int c = FF_CPT * q;
    w01[t] = kmh_f2{c < Cout ? w[c * 27 + t] : 0.f, c + 1 < Cout ? w[(c + 1) * 27 + t] : 0.f};
  }
  store_tile(0);
  __syncthreads();
  const int gx = x0 + vx;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  float* yn = y + (long long)n * D * H * W * Cout;
  const int zend = (zc + FF_Z < D) ? zc + FF_Z : D;
  for (int z = zc; z < zend; ++z) {
    const float* t = sx[(z - zc) & 1];
    if (z + 1 < zend) load_tile(z + 1);
    float win[3][3][3];                       // [kz][ky][kx], rows ky = 1, 2 preloaded with tile rows 0, 1
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 2; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) win[kz][ky + 1][kx] = t[(kz * (FF_Y + 2) + ky) * FF_P + vx + kx];
    // rows in blocks of three: sliding the 3-row window three times brings every value back to its register, so the
    // block loop needs no register moves and is NOT unrolled (a fully unrolled row loop kept ~250 registers live)
    static_assert(FF_Y % 3 == 0, "row blocks of three");
#pragma unroll 1
    for (int yb = 0; yb < FF_Y; yb += 3)
#pragma unroll
    for (int yj = 0; yj < 3; ++yj) {
      const int yy = yb + yj;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          win[kz][0][kx] = win[kz][1][kx];
          win[kz][1][kx] = win[kz][2][kx];
          win[kz][2][kx] = t[(kz * (FF_Y + 2) + yy + 2) * FF_P + vx + kx];
        }
      kmh_f2 a01 = {0.f, 0.f};
#pragma unroll
      for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float v = win[kz][ky][kx];
            a01 = __builtin_elementwise_fma(kmh_f2{v, v}, w01[(kz * 3 + ky) * 3 + kx], a01);
          }
      const int gy = y0 + yy;
      if (gx < W && gy < H) {
        const float o[2] = {fmaxf(a01.x, 0.f), fmaxf(a01.y, 0.f)};
        float* dst = yn + (((long long)z * H + gy) * W + gx) * Cout + FF_CPT * q;
        if ((Cout & 1) == 0) {
          if (FF_CPT * q < Cout) *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (FF_CPT * q + j < Cout) dst[j] = o[j];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { s1[j] += o[j]; s2[j] += o[j] * o[j]; }   // channels >= Cout hold zeros
      }
    }
    if (z + 1 < zend) store_tile((z + 1 - zc) & 1);   // the other buffer: its readers finished a plane ago
    __syncthreads();
  }
  if (stats_partial) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      double d1 = (double)s1[j], d2 = (double)s2[j];
#pragma unroll
      for (int o = FF_SUB; o < kWave; o <<= 1) { d1 += __shfl_xor(d1, o, kWave); d2 += __shfl_xor(d2, o, kWave); }
      if (lane < FF_SUB) { sred[wv][FF_CPT * lane + j][0] = d1; sred[wv][FF_CPT * lane + j][1] = d2; }
    }
    __syncthreads();
    if (tid < 2 * Cout) {
      const int c = tid >> 1, k = tid & 1;
      const long long blk = ((long long)n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      double s = 0;
#pragma unroll
      for (int wq = 0; wq < FF_TPB / kWave; ++wq) s += sred[wq][c][k];
      stats_partial[(blk * Cout + c) * 2 + k] = s;
    }
  }
}

}  // namespace

KMH_API size_t kmh_conv3d_first_layer_fwd_ws_bytes(int N, int D, int H, int W, int Cout) {
  return (size_t)N * ceil_div(D, FF_Z) * ceil_div(W, FF_X) * ceil_div(H, FF_Y) * Cout * 2 * sizeof(double);
}

/* x (N,D,H,W) raw 1-channel input, scale / shift (N) GroupNorm coefficients of that channel (NULL: identity),
 * w (Cout,1,3,3,3) with Cout <= 16 -> y (N,D,H,W,Cout) = relu(conv3(x * scale + shift));
 * stats_out (N,Cout,2) doubles | NULL = (sum y, sum y^2) like kmh_conv3d_fwd_bf's; ws: ..._fwd_ws_bytes.
 * Replaces the first conv of keymorph/unet3d/buildingblocks.py:46-78 (encoders[0].SingleConv1), exact fp32. */
KMH_API int kmh_conv3d_first_layer_fwd(const float* x, const float* scale, const float* shift, const float* w, float* y,
                                       int N, int D, int H, int W, int Cout, void* ws, double* stats_out, void* stream) {
  if (Cout > 16 || Cout < 1) return -22;
  hipStream_t s = (hipStream_t)stream;
  const int xt = ceil_div(W, FF_X), yt = ceil_div(H, FF_Y);
  const int zt = ceil_div(D, FF_Z);
  first_fwd_kernel<<<dim3(xt * yt, zt, N), FF_TPB, 0, s>>>(x, scale, shift, w, y, stats_out ? (double*)ws : nullptr, D, H,
                                                          W, Cout, xt, yt);
  if (stats_out)
    kmh_stats::final_kernel<<<dim3(ceil_div(Cout * 2, 256 / kWave), N), 256, 0, s>>>((const double*)ws, zt * xt * yt,
                                                                                  Cout, stats_out);
  return KMH_LAUNCH_CHECK();
}

KMH_API size_t kmh_conv3d_first_layer_wgrad_ws_bytes(int N, int D, int H, int W, int Cout) {
  (void)W;
  return (size_t)N * D * ceil_div(H, FL_YR) * FL_XS * Cout * 54 * sizeof(float);
}

/* x (N,D,H,W) raw 1-channel input, dz (N,D,H,W,Cout) with Cout <= 16, dzmask like dz or NULL ->
 * rs (N,Cout,2,27) DOUBLES: rs[n][co][0][tap] = R, rs[n][co][1][tap] = S  (the input of kmh_conv3d_first_layer_fold).
 * c123 (N,Cout,3) | NULL (needs dzmask): the gradient is [dzmask > 0] (c1 dz + c2 dzmask + c3), i.e. the NEXT layer's
 * GroupNorm backward (kmh_gn_bwd_apply with relu_mask) applied on the fly to its normalised-input gradient dz. */
KMH_API int kmh_conv3d_first_layer_wgrad(const float* x, const float* dz, const float* dzmask, const float* c123, double* rs,
                                         int N, int D, int H, int W, int Cout, void* ws, void* stream) {
  if (Cout > FL_CO || Cout < 1 || W < 1 || (c123 && !dzmask)) return -22;
  hipStream_t s = (hipStream_t)stream;
  const int WP = (W + 3) & ~3, XP = WP + 8;
  const size_t lds = ((size_t)3 * (FL_YR + 2) * XP + (size_t)2 * FL_CO * (WP + 4)) * sizeof(float);
  if (lds > 160 * 1024) return -22;
  hipError_t e = hipFuncSetAttribute((const void*)first_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int yt = ceil_div(H, FL_YR);
  const int per = Cout * 54;
  // matrix-core kernel (round 3) whenever its LDS image fits; KEYMORPH_FIRST_WGRAD_VALU=1 keeps the VALU kernel (A/B runs)
  static const bool valu_only = getenv("KEYMORPH_FIRST_WGRAD_VALU") != nullptr;
  size_t ds_floats = (size_t)2 * WP * 16;
  if (ds_floats < 2 * 4 * 2 * 64 * 4) ds_floats = 2 * 4 * 2 * 64 * 4;          // room for the final reduction (16 KB of doubles)
  const size_t lds_m = ((size_t)3 * (FM_YR + 2) * XP + ds_floats + 2 * 4 * 16 + 2 * 2 * 16) * sizeof(float);
  if (!valu_only && lds_m <= 160 * 1024 && FM_YR == FL_YR) {
    hipError_t e2 = hipFuncSetAttribute((const void*)first_wgrad_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
    if (e2 != hipSuccess) return (int)e2;
    first_wgrad_mfma_kernel<<<dim3(yt, D, N), FM_TPB, lds_m, s>>>(x, dz, dzmask, (float*)ws, D, H, W, Cout, yt, c123);
    first_wgrad_reduce_kernel<<<dim3(ceil_div(per, 4), N), 256, 0, s>>>((const float*)ws, D * yt, per, rs);
    return KMH_LAUNCH_CHECK();
  }
  first_wgrad_kernel<<<dim3(yt, D, N), FL_TPB, lds, s>>>(x, dz, dzmask, (float*)ws, D, H, W, Cout, yt, c123);
  first_wgrad_reduce_kernel<<<dim3(ceil_div(per, 4), N), 256, 0, s>>>((const float*)ws, D * yt * FL_XS, per, rs);
  return KMH_LAUNCH_CHECK();
}
