// Dense sampling-grid generators and their backward passes.
//   affine : keymorph/transformations.py:37-79  (AffineTransform.affine_grid / get_flow_field)
//   TPS    : keymorph/keypoint_aligners.py:365-433 (TPS.get_flow_field / transform_points)
// The identity grid (keymorph/utils.py:387-398, linspace(-1,1,n) per axis, ij order) is
// generated from the voxel index -- it is never materialised -- and the result is written
// once, already flipped to the xyz order grid_sample expects.
//
// TPS forward is VALU/transcendental bound (8.6 G (voxel,keypoint) pairs at 256^3 x 512, each
// one v_log (common.h: tps_log2x2) + ~8 VALU; 201 MB written, 12 KB read): keypoints + weights live in
// LDS and are broadcast-read; each lane owns VPT consecutive voxels.
// TPS backward is the transposed reduction: each lane owns KPT keypoints (6 accumulators
// each), voxels + their incoming gradient are staged through LDS and broadcast.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int TPB = 256;
constexpr int VPT = 4;

// torch.linspace(-1, 1, n)[i] in fp32 (symmetric evaluation: start + i*step below the midpoint,
// end - (n-1-i)*step above)
__device__ __forceinline__ float lin(int i, int n, float step) {
  return (i < n / 2) ? (-1.f + step * (float)i) : (1.f - step * (float)(n - 1 - i));
}
__host__ __device__ __forceinline__ float lin_step(int n) { return n > 1 ? 2.f / (float)(n - 1) : 0.f; }

// ---------------------------------------------------------------------------------------------
// affine
__global__ __launch_bounds__(TPB) void affine_grid_fwd_kernel(const float* __restrict__ mat,
                                                              float* __restrict__ out, int D, int H, int W) {
  const int n = blockIdx.y;
  const long long nvox = (long long)D * H * W;
  const long long v0 = ((long long)blockIdx.x * TPB + threadIdx.x) * VPT;
  if (v0 >= nvox) return;
  const float* m = mat + n * 12;
  float M[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) M[i] = m[i];
  const float sz = lin_step(D), sy = lin_step(H), sx = lin_step(W);
  float r[VPT * 3];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    long long v = v0 + i;
    if (v >= nvox) v = nvox - 1;
    const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long long)W * H));
    const float gz = lin(z, D, sz), gy = lin(y, H, sy), gx = lin(x, W, sx);
    // rows of M are (z, y, x) outputs; flip -> (x, y, z)
    r[i * 3 + 2] = M[0] * gz + M[1] * gy + M[2] * gx + M[3];
    r[i * 3 + 1] = M[4] * gz + M[5] * gy + M[6] * gx + M[7];
    r[i * 3 + 0] = M[8] * gz + M[9] * gy + M[10] * gx + M[11];
  }
  float* o = out + ((long long)n * nvox + v0) * 3;
  if (v0 + VPT <= nvox && (nvox & 3) == 0) {
    float4* o4 = reinterpret_cast<float4*>(o);
    o4[0] = make_float4(r[0], r[1], r[2], r[3]);
    o4[1] = make_float4(r[4], r[5], r[6], r[7]);
    o4[2] = make_float4(r[8], r[9], r[10], r[11]);
  } else {
    for (int i = 0; i < VPT; ++i)
      if (v0 + i < nvox) { o[i * 3] = r[i * 3]; o[i * 3 + 1] = r[i * 3 + 1]; o[i * 3 + 2] = r[i * 3 + 2]; }
  }
}

// dM[r][k] = sum_v dgrid[v][2-r] * (gz, gy, gx, 1)[k].  partial (N, nblk, 12) doubles.
__global__ __launch_bounds__(TPB) void affine_grid_bwd_partial(const float* __restrict__ dgrid,
                                                               double* __restrict__ partial, int D, int H,
                                                               int W) {
  const int n = blockIdx.y;
  const long long nvox = (long long)D * H * W;
  const float sz = lin_step(D), sy = lin_step(H), sx = lin_step(W);
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  const float* dg = dgrid + (long long)n * nvox * 3;
  for (long long v = (long long)blockIdx.x * TPB + threadIdx.x; v < nvox; v += (long long)gridDim.x * TPB) {
    const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long long)W * H));
    const float p[4] = {lin(z, D, sz), lin(y, H, sy), lin(x, W, sx), 1.f};
    const float g[3] = {dg[v * 3 + 2], dg[v * 3 + 1], dg[v * 3 + 0]};  // (z, y, x) rows
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[r * 4 + k] += g[r] * p[k];
  }
  __shared__ double red[TPB / kWave];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    double s = block_sum<double>((double)acc[i], red);
    if (threadIdx.x == 0) partial[((long long)n * gridDim.x + blockIdx.x) * 12 + i] = s;
  }
}

// 12 outputs x 16 slices of the per-block partials, combined in a fixed order (12 threads walking 1024 partials: 0.2 ms)
__global__ __launch_bounds__(192) void affine_grid_bwd_final(const double* __restrict__ partial, int nblk, float* __restrict__ dmat) {
  __shared__ double red[16][12];
  const int n = blockIdx.x;
  const int i = threadIdx.x % 12, sl = threadIdx.x / 12;   // 192 threads
  double s = 0;
  for (int b = sl; b < nblk; b += 16) s += partial[((long long)n * nblk + b) * 12 + i];
  red[sl][i] = s;
  __syncthreads();
  if (sl) return;
  s = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k][i];
  dmat[n * 12 + i] = (float)s;
}

// ---------------------------------------------------------------------------------------------
// TPS forward on the implicit grid.  LDS: ctrl (T x float4: cz,cy,cx,0) + weights (T x float4).
// ROWQ (implicit grid with W % VPT == 0): the lane's VPT voxels are consecutive in x, so (cz - z)^2 + (cy - y)^2 + 1e-6 --
// the inner two links of tps_d2's fma chain -- is ONE scalar per keypoint instead of VPT/2 packed evaluations: 4 plain
// VALU replace 8 packed ones of the 22 per keypoint and voxel quad, bit-identical results (the chain's order is kept).
// TV voxels per lane: 8 on the implicit grid when W % 8 == 0 (the per-keypoint work -- two LDS reads, the row constant -- is
// shared by twice the voxels), else VPT
template <bool EXPLICIT_POINTS, bool ROWQ = false, int TV = VPT>
__global__ __launch_bounds__(TPB) void tps_eval_fwd_kernel(const float* __restrict__ theta,
                                                           const float* __restrict__ ctrl,
                                                           const float* __restrict__ pts,
                                                           float* __restrict__ out, int T, int D, int H,
                                                           int W, long long npts) {
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  float4* sc = smem4;      // [T]
  float4* sw = smem4 + T;  // [T]
  const int n = blockIdx.y;
  const float* th = theta + (long long)n * (T + 4) * 3;
  const float* cc = ctrl + (long long)n * T * 3;
  for (int t = threadIdx.x; t < T; t += TPB) {
    sc[t] = make_float4(cc[t * 3], cc[t * 3 + 1], cc[t * 3 + 2], 0.f);
    const float hl = kTpsHalfLn2;   // folded into the weights: the loop accumulates (w ln2 / 2) * (2 U / ln2)
    sw[t] = make_float4(th[t * 3] * hl, th[t * 3 + 1] * hl, th[t * 3 + 2] * hl, 0.f);
  }
  __syncthreads();
  const long long v0 = ((long long)blockIdx.x * TPB + threadIdx.x) * TV;
  if (v0 >= npts) return;
  const float* a = th + (long long)T * 3;  // rows: 1, z, y, x ; cols: (z, y, x) outputs
  float pz[TV], py[TV], px[TV], oz[TV], oy[TV], ox[TV];
  const float sz = lin_step(D), sy = lin_step(H), sx = lin_step(W);
#pragma unroll
  for (int i = 0; i < TV; ++i) {
    long long v = v0 + i;
    if (v >= npts) v = npts - 1;
    if (EXPLICIT_POINTS) {
      const float* p = pts + ((long long)n * npts + v) * 3;
      pz[i] = p[0]; py[i] = p[1]; px[i] = p[2];
    } else {
      const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long long)W * H));
      pz[i] = lin(z, D, sz); py[i] = lin(y, H, sy); px[i] = lin(x, W, sx);
    }
  }
  // packed fp32 (v_pk_*_f32): two voxels per instruction; the transcendentals stay scalar
  static_assert(TV % 2 == 0, "voxel pairs");
  kmh_f2 qz[TV / 2], qy[TV / 2], qx[TV / 2], az2[TV / 2], ay2[TV / 2], ax2[TV / 2];
#pragma unroll
  for (int h = 0; h < TV / 2; ++h) {
    qz[h] = kmh_f2{pz[2 * h], pz[2 * h + 1]}; qy[h] = kmh_f2{py[2 * h], py[2 * h + 1]}; qx[h] = kmh_f2{px[2 * h], px[2 * h + 1]};
    az2[h] = ay2[h] = ax2[h] = kmh_f2{0.f, 0.f};
  }
  if constexpr (ROWQ && !EXPLICIT_POINTS) {
    const float pz0 = pz[0], py0 = py[0];
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
      const float4 c = sc[t];
      const float4 w = sw[t];
      const float dzs = c.x - pz0, dys = c.y - py0;
      const float zy = fmaf(dys, dys, fmaf(dzs, dzs, 1e-6f));      // tps_d2's chain up to its last link
      const kmh_f2 zy2 = {zy, zy};
#pragma unroll
      for (int h = 0; h < TV / 2; ++h) {
        const kmh_f2 dx = c.z - qx[h];
        const kmh_f2 u = tps_u2_from_d2(__builtin_elementwise_fma(dx, dx, zy2));
        az2[h] += u * w.x; ay2[h] += u * w.y; ax2[h] += u * w.z;
      }
    }
  } else {
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
      const float4 c = sc[t];
      const float4 w = sw[t];
#pragma unroll
      for (int h = 0; h < TV / 2; ++h) {
        const kmh_f2 dz = c.x - qz[h], dy = c.y - qy[h], dx = c.z - qx[h];
        const kmh_f2 u = tps_u2_from_d2(tps_d2(dz, dy, dx));
        az2[h] += u * w.x; ay2[h] += u * w.y; ax2[h] += u * w.z;
      }
    }
  }
#pragma unroll
  for (int h = 0; h < TV / 2; ++h) {
    oz[2 * h] = az2[h].x; oz[2 * h + 1] = az2[h].y; oy[2 * h] = ay2[h].x; oy[2 * h + 1] = ay2[h].y;
    ox[2 * h] = ax2[h].x; ox[2 * h + 1] = ax2[h].y;
  }
  float r[TV * 3];
#pragma unroll
  for (int i = 0; i < TV; ++i) {
    const float az = a[0] + a[3] * pz[i] + a[6] * py[i] + a[9] * px[i];
    const float ay = a[1] + a[4] * pz[i] + a[7] * py[i] + a[10] * px[i];
    const float ax = a[2] + a[5] * pz[i] + a[8] * py[i] + a[11] * px[i];
    if (EXPLICIT_POINTS) {  // ij order out
      r[i * 3 + 0] = az + oz[i]; r[i * 3 + 1] = ay + oy[i]; r[i * 3 + 2] = ax + ox[i];
    } else {  // flipped to xyz
      r[i * 3 + 0] = ax + ox[i]; r[i * 3 + 1] = ay + oy[i]; r[i * 3 + 2] = az + oz[i];
    }
  }
  float* o = out + ((long long)n * npts + v0) * 3;
  if (v0 + TV <= npts && (npts & 3) == 0) {
    float4* o4 = reinterpret_cast<float4*>(o);
#pragma unroll
    for (int k = 0; k < TV * 3 / 4; ++k) o4[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
  } else {
    for (int i = 0; i < TV; ++i)
      if (v0 + i < npts) { o[i * 3] = r[i * 3]; o[i * 3 + 1] = r[i * 3 + 1]; o[i * 3 + 2] = r[i * 3 + 2]; }
  }
}

// ---------------------------------------------------------------------------------------------
// TPS backward wrt (theta_w, ctrl): lanes own keypoints, voxels are broadcast from LDS.
constexpr int KPT = 2;             // keypoints per lane
constexpr int BWD_TPB = 256;       // -> 512 keypoints per block pass
constexpr int VSTAGE = 256;        // voxels staged per LDS refill (one per thread)
constexpr int VCHUNK = 8192;       // voxels per block

// 2 dU/d(d2) = 2 ln(r + eps) + r / (r + eps), from L = 2 log2(r + eps) and the reciprocal-square-root estimate y:
// r / (r + eps) = 1 / (1 + t) = 1 - t + O(t^2), t = eps / r = 1e-6 * 0.98636 y <= 1e-3 (t^2 <= 1e-6 of a term of size one)
__device__ __forceinline__ kmh_f2 tps_du2(kmh_f2 L, kmh_f2 y) {
  const kmh_f2 ln2 = {0.6931471805599453f, 0.6931471805599453f}, one = {1.f, 1.f};
  const kmh_f2 mt = {-1.0e-6f * kTpsRsqCentre, -1.0e-6f * kTpsRsqCentre};
  return __builtin_elementwise_fma(y, mt, __builtin_elementwise_fma(L, ln2, one));
}

// ROWS (implicit grid with W % VSTAGE == 0): a stage's 256 voxels are one piece of ONE grid row, so dz, dy and the inner
// links of the distance chain are per-stage constants of the lane's keypoints, and sum f dz = dz sum f (same for dy).
template <bool EXPLICIT_POINTS, bool ROWS = false>
__global__ __launch_bounds__(BWD_TPB) void tps_eval_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ theta, const float* __restrict__ ctrl,
    const float* __restrict__ pts, float* __restrict__ partial /* (N, nchunk, T, 6) */, int T, int D,
    int H, int W, long long npts, int nchunk) {
  __shared__ __attribute__((aligned(16))) float4 sp[VSTAGE];  // (pz, py, px, 0)
  __shared__ __attribute__((aligned(16))) float4 sg[VSTAGE];  // (gz, gy, gx, 0)
  const int n = blockIdx.z;
  const int chunk = blockIdx.x;
  const int ktile = blockIdx.y;
  const float* th = theta + (long long)n * (T + 4) * 3;
  const float* cc = ctrl + (long long)n * T * 3;
  static_assert(KPT == 2, "the lane's two keypoints are the two halves of the packed-fp32 operands");
  kmh_f2 cz, cy, cx, wz, wy, wx;
  kmh_f2 aw[3], ac[3];
  int tk[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    tk[k] = ktile * (BWD_TPB * KPT) + k * BWD_TPB + threadIdx.x;
    const int t = tk[k] < T ? tk[k] : T - 1;
    cz[k] = cc[t * 3]; cy[k] = cc[t * 3 + 1]; cx[k] = cc[t * 3 + 2];
    wz[k] = th[t * 3]; wy[k] = th[t * 3 + 1]; wx[k] = th[t * 3 + 2];
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) aw[d] = ac[d] = kmh_f2{0.f, 0.f};
  const float sz = lin_step(D), sy = lin_step(H), sx = lin_step(W);
  const long long vbeg = (long long)chunk * VCHUNK;
  long long vend = vbeg + VCHUNK;
  if (vend > npts) vend = npts;
  for (long long vs = vbeg; vs < vend; vs += VSTAGE) {
    const long long v = vs + threadIdx.x;
    float4 p = make_float4(0, 0, 0, 0), g = make_float4(0, 0, 0, 0);
    if (v < vend) {
      const float* dg = dout + ((long long)n * npts + v) * 3;
      if (EXPLICIT_POINTS) {
        const float* q = pts + ((long long)n * npts + v) * 3;
        p = make_float4(q[0], q[1], q[2], 0.f);
        g = make_float4(dg[0], dg[1], dg[2], 0.f);
      } else {
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long long)W * H));
        p = make_float4(lin(z, D, sz), lin(y, H, sy), lin(x, W, sx), 0.f);
        g = make_float4(dg[2], dg[1], dg[0], 0.f);  // stored xyz -> (z, y, x)
      }
    }
    __syncthreads();
    sp[threadIdx.x] = p;
    sg[threadIdx.x] = g;  // zero gradient for out-of-range voxels => no contribution
    __syncthreads();
    const int cnt = (int)((vend - vs) < VSTAGE ? (vend - vs) : VSTAGE);
    if constexpr (ROWS && !EXPLICIT_POINTS) {
      const float4 p0 = sp[0];
      const kmh_f2 dz = cz - p0.x, dy = cy - p0.y;
      const kmh_f2 eps2 = {1e-6f, 1e-6f};
      const kmh_f2 zy = __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, eps2));
      kmh_f2 fs = {0.f, 0.f};
#pragma unroll 2
      for (int j = 0; j < cnt; ++j) {
        const float px = sp[j].z;
        const float4 gg = sg[j];
        const kmh_f2 dx = cx - px;
        const kmh_f2 d2 = __builtin_elementwise_fma(dx, dx, zy);      // == tps_d2(dz, dy, dx)
        const kmh_f2 y = tps_rsq_est2(d2);
        const kmh_f2 L = tps_log2x2(d2, y);                         // 2 log2(r + eps), common.h
        const kmh_f2 u = d2 * L;
        aw[0] += u * gg.x; aw[1] += u * gg.y; aw[2] += u * gg.z;
        const kmh_f2 s = wz * gg.x + wy * gg.y + wx * gg.z;
        const kmh_f2 f = s * tps_du2(L, y);
        fs += f;
        ac[2] += f * dx;
      }
      ac[0] += fs * dz; ac[1] += fs * dy;
      continue;
    }
#pragma unroll 2
    for (int j = 0; j < cnt; ++j) {
      const float4 pp = sp[j];
      const float4 gg = sg[j];
      // both keypoints of the lane at once (v_pk_*_f32); one transcendental per (voxel, keypoint): log2(d2)
      const kmh_f2 dz = cz - pp.x, dy = cy - pp.y, dx = cx - pp.z;
      const kmh_f2 d2 = tps_d2(dz, dy, dx);                       // includes the + 1e-6
      const kmh_f2 y = tps_rsq_est2(d2);
      const kmh_f2 L = tps_log2x2(d2, y);                           // 2 log2(r + eps): ln 2 / 2 is applied once, after the loop
      const kmh_f2 u = d2 * L;
      aw[0] += u * gg.x; aw[1] += u * gg.y; aw[2] += u * gg.z;
      const kmh_f2 s = wz * gg.x + wy * gg.y + wx * gg.z;
      const kmh_f2 f = s * tps_du2(L, y);
      ac[0] += f * dz; ac[1] += f * dy; ac[2] += f * dx;
    }
  }
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    if (tk[k] < T) {
      float* o = partial + (((long long)n * nchunk + chunk) * T + tk[k]) * 6;
      o[0] = aw[0][k] * kTpsHalfLn2; o[1] = aw[1][k] * kTpsHalfLn2; o[2] = aw[2][k] * kTpsHalfLn2;
      o[3] = ac[0][k]; o[4] = ac[1][k]; o[5] = ac[2][k];
    }
  }
}

// sum the per-chunk partials in fp64, fixed order: a 1024-thread workgroup owns 64 of the (t, 6) outputs of a sample and
// deals the chunks to 16 slices (one thread per output walking all 2048 chunks of a 256^3 grid was 0.67 ms of pure latency)
constexpr int FIN_O = 64, FIN_S = 16;
__global__ __launch_bounds__(FIN_O * FIN_S) void tps_bwd_final_kernel(const float* __restrict__ partial, int nchunk,
                                                                      int T, float* __restrict__ dtheta,
                                                                      float* __restrict__ dctrl, int accumulate_ctrl) {
  __shared__ double red[FIN_S][FIN_O];
  const int n = blockIdx.y, o = threadIdx.x % FIN_O, cs = threadIdx.x / FIN_O;
  const int idx = blockIdx.x * FIN_O + o;
  double s = 0;
  if (idx < T * 6) {
    const float* p = partial + (long long)n * nchunk * T * 6 + idx;
    for (int c = cs; c < nchunk; c += FIN_S) s += p[(long long)c * T * 6];
  }
  red[cs][o] = s;
  __syncthreads();
  if (cs != 0 || idx >= T * 6) return;
  s = 0;
#pragma unroll
  for (int k = 0; k < FIN_S; ++k) s += red[k][o];
  const int t = idx / 6, j = idx % 6;
  if (j < 3) dtheta[((long long)n * (T + 4) + t) * 3 + j] = (float)s;
  else if (accumulate_ctrl) dctrl[((long long)n * T + t) * 3 + (j - 3)] += (float)s;
  else dctrl[((long long)n * T + t) * 3 + (j - 3)] = (float)s;
}

// affine rows of dtheta from the (N,12) "dmat" computed by the affine-grid backward reduction:
// dmat[r][k] (r = output z,y,x; k = gz,gy,gx,1)  ->  dtheta[T + {0:1, 1:z, 2:y, 3:x}][r]
__global__ void tps_affine_rows_kernel(const float* __restrict__ dmat, int T, float* __restrict__ dtheta) {
  const int n = blockIdx.x, i = threadIdx.x;
  if (i >= 12) return;
  const int r = i / 4, k = i % 4;
  const int row = (k == 3) ? 0 : k + 1;
  dtheta[((long long)n * (T + 4) + T + row) * 3 + r] = dmat[n * 12 + i];
}

// explicit-point variants of the small reductions (P ~ T ~ 512): dpts and the affine rows.
__global__ __launch_bounds__(TPB) void tps_points_bwd_pts_kernel(
    const float* __restrict__ dout, const float* __restrict__ theta, const float* __restrict__ ctrl,
    const float* __restrict__ pts, float* __restrict__ dpts, int T, int P) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * TPB + threadIdx.x;
  if (p >= P) return;
  const float* th = theta + (long long)n * (T + 4) * 3;
  const float* cc = ctrl + (long long)n * T * 3;
  const float* q = pts + ((long long)n * P + p) * 3;
  const float* g = dout + ((long long)n * P + p) * 3;
  const float pz = q[0], py = q[1], px = q[2], gz = g[0], gy = g[1], gx = g[2];
  const float* a = th + (long long)T * 3;
  // affine part: out_d = a[0][d] + sum_k a[1+k][d] p_k  =>  dp_k = sum_d a[1+k][d] g_d
  float dz = a[3] * gz + a[4] * gy + a[5] * gx;
  float dy = a[6] * gz + a[7] * gy + a[8] * gx;
  float dx = a[9] * gz + a[10] * gy + a[11] * gx;
  for (int t = 0; t < T; ++t) {
    const float ez = cc[t * 3] - pz, ey = cc[t * 3 + 1] - py, ex = cc[t * 3 + 2] - px;
    const float d2 = ez * ez + ey * ey + ex * ex + 1e-6f;
    const float r = __builtin_amdgcn_sqrtf(d2), re = r + 1e-6f;
    const float L = __builtin_amdgcn_logf(re) * 0.6931471805599453f;
    const float s = th[t * 3] * gz + th[t * 3 + 1] * gy + th[t * 3 + 2] * gx;
    const float f = s * (2.f * L + r * __builtin_amdgcn_rcpf(re));
    dz -= f * ez; dy -= f * ey; dx -= f * ex;  // d(d2)/dp = -2 (c - p)
  }
  float* o = dpts + ((long long)n * P + p) * 3;
  o[0] = dz; o[1] = dy; o[2] = dx;
}

__global__ __launch_bounds__(TPB) void points_affine_rows_kernel(const float* __restrict__ dout,
                                                                 const float* __restrict__ pts, int T, int P,
                                                                 float* __restrict__ dtheta) {
  // 12 sums over P points, one block per sample
  const int n = blockIdx.x;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int p = threadIdx.x; p < P; p += TPB) {
    const float* q = pts + ((long long)n * P + p) * 3;
    const float* g = dout + ((long long)n * P + p) * 3;
    const float pk[4] = {1.f, q[0], q[1], q[2]};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[k * 3 + d] += pk[k] * g[d];
  }
  __shared__ double red[TPB / kWave];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    double s = block_sum<double>((double)acc[i], red);
    if (threadIdx.x == 0) dtheta[((long long)n * (T + 4) + T) * 3 + i] = (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// matrix applied to explicit points: out = M[:, :3, :] [p; 1]  (keymorph/transformations.py:81-114)
__global__ __launch_bounds__(TPB) void affine_points_fwd_kernel(const float* __restrict__ M,
                                                                const float* __restrict__ pts,
                                                                float* __restrict__ out, int P) {
  const int n = blockIdx.y, p = blockIdx.x * TPB + threadIdx.x;
  if (p >= P) return;
  const float* m = M + n * 12;
  const float* q = pts + ((long long)n * P + p) * 3;
  float* o = out + ((long long)n * P + p) * 3;
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = m[r * 4] * q[0] + m[r * 4 + 1] * q[1] + m[r * 4 + 2] * q[2] + m[r * 4 + 3];
}

__global__ __launch_bounds__(TPB) void affine_points_bwd_kernel(const float* __restrict__ dout,
                                                                const float* __restrict__ M,
                                                                const float* __restrict__ pts,
                                                                float* __restrict__ dM, float* __restrict__ dpts,
                                                                int P) {
  const int n = blockIdx.x;
  const float* m = M + n * 12;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int p = threadIdx.x; p < P; p += TPB) {
    const float* q = pts + ((long long)n * P + p) * 3;
    const float* g = dout + ((long long)n * P + p) * 3;
    const float pk[4] = {q[0], q[1], q[2], 1.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[r * 4 + k] += g[r] * pk[k];
    if (dpts) {
      float* o = dpts + ((long long)n * P + p) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) o[k] = m[k] * g[0] + m[4 + k] * g[1] + m[8 + k] * g[2];
    }
  }
  __shared__ double red[TPB / kWave];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    double s = block_sum<double>((double)acc[i], red);
    if (threadIdx.x == 0) dM[n * 12 + i] = (float)s;
  }
}

constexpr int AFF_BWD_BLOCKS = 1024;

}  // namespace

// ---------------------------------------------------------------------------------------------
KMH_API int kmh_affine_grid_fwd(const float* mat, float* out, int N, int D, int H, int W, void* stream) {
  const long long nvox = (long long)D * H * W;
  affine_grid_fwd_kernel<<<dim3(ceil_div(nvox, (long long)TPB * VPT), N), TPB, 0, (hipStream_t)stream>>>(
      mat, out, D, H, W);
  return KMH_LAUNCH_CHECK();
}

static int affine_bwd_blocks(long long nvox) {
  int nb = ceil_div(nvox, (long long)TPB * 16);
  if (nb > AFF_BWD_BLOCKS) nb = AFF_BWD_BLOCKS;
  return nb < 1 ? 1 : nb;
}

KMH_API int kmh_affine_grid_bwd(const float* dgrid, float* dmat, int N, int D, int H, int W, void* ws,
                                void* stream) {
  // ws: N * AFF_BWD_BLOCKS * 12 doubles (<= kmh_reduce_ws_bytes() for N <= 16)
  const long long nvox = (long long)D * H * W;
  const int nb = affine_bwd_blocks(nvox);
  if ((size_t)N * nb * 12 * sizeof(double) > (size_t)65536 * 8 * 3) return -22;
  hipStream_t s = (hipStream_t)stream;
  affine_grid_bwd_partial<<<dim3(nb, N), TPB, 0, s>>>(dgrid, (double*)ws, D, H, W);
  affine_grid_bwd_final<<<N, 192, 0, s>>>((const double*)ws, nb, dmat);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_tps_grid_fwd(const float* theta, const float* ctrl, float* out, int N, int T, int D, int H,
                             int W, void* stream) {
  const long long nvox = (long long)D * H * W;
  const size_t lds = (size_t)T * 2 * sizeof(float4);
  if (lds > 64 * 1024) return -22;
  static const bool no_rowq = getenv("KMH_TPS_NO_ROWQ") != nullptr;       // A/B switch (tools/prof_tps.py)
  static const int tv = getenv("KMH_TPS_TV") ? atoi(getenv("KMH_TPS_TV")) : 8;
  const dim3 g(ceil_div(nvox, (long long)TPB * VPT), N);
  if (W % 8 == 0 && !no_rowq && tv == 8)
    tps_eval_fwd_kernel<false, true, 8><<<dim3(ceil_div(nvox, (long long)TPB * 8), N), TPB, lds, (hipStream_t)stream>>>(
        theta, ctrl, nullptr, out, T, D, H, W, nvox);
  else if (W % VPT == 0 && !no_rowq)
    tps_eval_fwd_kernel<false, true><<<g, TPB, lds, (hipStream_t)stream>>>(theta, ctrl, nullptr, out, T, D, H, W, nvox);
  else
    tps_eval_fwd_kernel<false, false><<<g, TPB, lds, (hipStream_t)stream>>>(theta, ctrl, nullptr, out, T, D, H, W, nvox);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_tps_points_fwd(const float* theta, const float* ctrl, const float* pts, float* out, int N,
                               int T, int P, void* stream) {
  const size_t lds = (size_t)T * 2 * sizeof(float4);
  if (lds > 64 * 1024) return -22;
  tps_eval_fwd_kernel<true><<<dim3(ceil_div(P, (long long)TPB * VPT), N), TPB, lds, (hipStream_t)stream>>>(
      theta, ctrl, pts, out, T, 1, 1, 1, P);
  return KMH_LAUNCH_CHECK();
}

static size_t tps_bwd_ws(int N, int T, long long npts) {
  const long long nchunk = (npts + VCHUNK - 1) / VCHUNK;
  size_t partial = (size_t)N * nchunk * T * 6 * sizeof(float);
  size_t aff = (size_t)N * AFF_BWD_BLOCKS * 12 * sizeof(double) + (size_t)N * 12 * sizeof(float);
  return partial + aff + 256;
}

KMH_API size_t kmh_tps_grid_bwd_ws_bytes(int N, int T, int D, int H, int W) {
  return tps_bwd_ws(N, T, (long long)D * H * W);
}
KMH_API size_t kmh_tps_points_bwd_ws_bytes(int N, int T, int P) { return tps_bwd_ws(N, T, P); }

KMH_API int kmh_tps_grid_bwd(const float* dgrid, const float* theta, const float* ctrl, float* dtheta,
                             float* dctrl, int N, int T, int D, int H, int W, void* ws, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long long nvox = (long long)D * H * W;
  const int nchunk = (int)((nvox + VCHUNK - 1) / VCHUNK);
  const int ktiles = (T + BWD_TPB * KPT - 1) / (BWD_TPB * KPT);
  float* partial = (float*)ws;
  size_t off = ((size_t)N * nchunk * T * 6 * sizeof(float) + 255) & ~(size_t)255;
  double* affp = (double*)((char*)ws + off);
  float* dmat = (float*)((char*)affp + (size_t)N * AFF_BWD_BLOCKS * 12 * sizeof(double));
  static const bool no_rowq = getenv("KMH_TPS_NO_ROWQ") != nullptr;
  if (W % VSTAGE == 0 && !no_rowq)
    tps_eval_bwd_kernel<false, true><<<dim3(nchunk, ktiles, N), BWD_TPB, 0, s>>>(dgrid, theta, ctrl, nullptr, partial,
                                                                               T, D, H, W, nvox, nchunk);
  else
    tps_eval_bwd_kernel<false, false><<<dim3(nchunk, ktiles, N), BWD_TPB, 0, s>>>(dgrid, theta, ctrl, nullptr, partial,
                                                                                T, D, H, W, nvox, nchunk);
  tps_bwd_final_kernel<<<dim3(ceil_div(T * 6, FIN_O), N), FIN_O * FIN_S, 0, s>>>(partial, nchunk, T, dtheta, dctrl, 0);
  const int nb = affine_bwd_blocks(nvox);
  affine_grid_bwd_partial<<<dim3(nb, N), TPB, 0, s>>>(dgrid, affp, D, H, W);
  affine_grid_bwd_final<<<N, 192, 0, s>>>(affp, nb, dmat);
  tps_affine_rows_kernel<<<N, 64, 0, s>>>(dmat, T, dtheta);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_tps_points_bwd(const float* dout, const float* theta, const float* ctrl, const float* pts,
                               float* dtheta, float* dctrl, float* dpts, int N, int T, int P, void* ws,
                               void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = (P + VCHUNK - 1) / VCHUNK;
  const int ktiles = (T + BWD_TPB * KPT - 1) / (BWD_TPB * KPT);
  float* partial = (float*)ws;
  tps_eval_bwd_kernel<true><<<dim3(nchunk, ktiles, N), BWD_TPB, 0, s>>>(dout, theta, ctrl, pts, partial, T, 1,
                                                                      1, 1, P, nchunk);
  tps_bwd_final_kernel<<<dim3(ceil_div(T * 6, FIN_O), N), FIN_O * FIN_S, 0, s>>>(partial, nchunk, T, dtheta, dctrl, 0);
  points_affine_rows_kernel<<<N, TPB, 0, s>>>(dout, pts, T, P, dtheta);
  if (dpts)
    tps_points_bwd_pts_kernel<<<dim3(ceil_div(P, TPB), N), TPB, 0, s>>>(dout, theta, ctrl, pts, dpts, T, P);
  return KMH_LAUNCH_CHECK();
}

// keymorph/augmentation.py:85-158 (AffineDeformation3d.build_affine_matrix): M = Mz (Ms (Mt (R3 (R2 R1)))),
// one thread per sample, fp32 products in the reference's nesting order.
namespace {
__device__ __forceinline__ void mm4(const float* a, const float* b, float* c) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
      c[i * 4 + j] = s;
    }
}
__global__ void affine_build_kernel(const float* __restrict__ scale, const float* __restrict__ offset,
                                    const float* __restrict__ theta, const float* __restrict__ shear,
                                    float* __restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float c1 = cosf(theta[b * 3]), s1 = sinf(theta[b * 3]);
  const float c2 = cosf(theta[b * 3 + 1]), s2 = sinf(theta[b * 3 + 1]);
  const float c3 = cosf(theta[b * 3 + 2]), s3 = sinf(theta[b * 3 + 2]);
  const float R1[16] = {1, 0, 0, 0, 0, c1, -s1, 0, 0, s1, c1, 0, 0, 0, 0, 1};
  const float R2[16] = {c2, 0, s2, 0, 0, 1, 0, 0, -s2, 0, c2, 0, 0, 0, 0, 1};
  const float R3[16] = {c3, -s3, 0, 0, s3, c3, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float Mt[16] = {1, 0, 0, offset[b * 3], 0, 1, 0, offset[b * 3 + 1], 0, 0, 1, offset[b * 3 + 2], 0, 0, 0, 1};
  const float Ms[16] = {scale[b * 3], 0, 0, 0, 0, scale[b * 3 + 1], 0, 0, 0, 0, scale[b * 3 + 2], 0, 0, 0, 0, 1};
  const float* z = shear + b * 6;
  const float Mz[16] = {1, z[0], z[1], 0, z[2], 1, z[3], 0, z[4], z[5], 1, 0, 0, 0, 0, 1};
  float t0[16], t1[16];
  mm4(R2, R1, t0);
  mm4(R3, t0, t1);     // Mr
  mm4(Mt, t1, t0);
  mm4(Ms, t0, t1);
  mm4(Mz, t1, t0);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[b * 16 + i] = t0[i];
}
}  // namespace

KMH_API int kmh_affine_build_matrix(const float* scale, const float* offset, const float* theta, const float* shear,
                                    float* out, int B, void* stream) {
  affine_build_kernel<<<ceil_div(B, 64), 64, 0, (hipStream_t)stream>>>(scale, offset, theta, shear, out, B);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_affine_points_fwd(const float* M, const float* pts, float* out, int N, int P, void* stream) {
  affine_points_fwd_kernel<<<dim3(ceil_div(P, TPB), N), TPB, 0, (hipStream_t)stream>>>(M, pts, out, P);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_affine_points_bwd(const float* dout, const float* M, const float* pts, float* dM, float* dpts,
                                  int N, int P, void* stream) {
  affine_points_bwd_kernel<<<N, TPB, 0, (hipStream_t)stream>>>(dout, M, pts, dM, dpts, P);
  return KMH_LAUNCH_CHECK();
}
