// Closed-form keypoint aligner fits (forward + backward), one workgroup per sample.
//   affine : keymorph/keypoint_aligners.py:76-114   M = Y X^T (X X^T)^-1
//   rigid  : keymorph/keypoint_aligners.py:151-213  Kabsch via 3x3 SVD, row-scaled reflection fix
//   4x4 inverse : keymorph/transformations.py:23-35
//   TPS    : keymorph/keypoint_aligners.py:276-363  A = [[U + lambda I, P], [P^T, 0]], A theta = [tgt; 0]
// The reference runs the TPS solve on the HOST (three LAPACK gesv of the same (T+4)^2 matrix per
// fit).  Here the matrix is assembled on the device in fp32 with the reference's formula
// writes it, factorised ONCE in fp64 (blocked right-looking LU with partial pivoting: 16-column
// panel + U12 strip in LDS, wavefront-shuffle pivot search) and solved for the 3 right-hand
// sides together; the factors stay in the workspace so the backward (A is symmetric, so
// A^T g = dtheta is the same solve) costs one more substitution, not a factorisation.
#include <atomic>
#include "common.h"

namespace {

constexpr int TPB = 256;

// ------------------------------------------------------------------------------------------
// small dense helpers (thread-local, double)
__device__ inline bool inv4(const double* a, double* inv) {
  // Gauss-Jordan with partial pivoting on [a | I]
  double m[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { m[i][j] = a[i * 4 + j]; m[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  bool ok = true;
  for (int c = 0; c < 4; ++c) {
    int p = c;
    double best = fabs(m[c][c]);
    for (int r = c + 1; r < 4; ++r)
      if (fabs(m[r][c]) > best) { best = fabs(m[r][c]); p = r; }
    if (best == 0.0) ok = false;
    if (p != c)
      for (int j = 0; j < 8; ++j) { double t = m[c][j]; m[c][j] = m[p][j]; m[p][j] = t; }
    const double d = 1.0 / m[c][c];
    for (int j = 0; j < 8; ++j) m[c][j] *= d;
    for (int r = 0; r < 4; ++r)
      if (r != c) {
        const double f = m[r][c];
        for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
      }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) inv[i * 4 + j] = m[i][4 + j];
  return ok;
}

// moments of one sample: S = sum w X X^T (16), C = sum w y X^T (12), all threads of the block
// participate; results valid in thread 0 only.
__device__ void affine_moments(const float* x, const float* y, const float* w, int K, double* S, double* C,
                               double* red) {
  double s[16], c[12];
  for (int i = 0; i < 16; ++i) s[i] = 0;
  for (int i = 0; i < 12; ++i) c[i] = 0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const double X[4] = {x[k * 3], x[k * 3 + 1], x[k * 3 + 2], 1.0};
    const double Y[3] = {y[k * 3], y[k * 3 + 1], y[k * 3 + 2]};
    const double wk = w ? (double)w[k] : 1.0;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) s[i * 4 + j] += wk * X[i] * X[j];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) c[i * 4 + j] += wk * Y[i] * X[j];
  }
  for (int i = 0; i < 16; ++i) { double r = block_sum<double>(s[i], red); if (threadIdx.x == 0) S[i] = r; }
  for (int i = 0; i < 12; ++i) { double r = block_sum<double>(c[i], red); if (threadIdx.x == 0) C[i] = r; }
}

__global__ __launch_bounds__(TPB) void affine_fit_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ w, float* __restrict__ M,
                                                             int K) {
  __shared__ double red[TPB / kWave];
  __shared__ double S[16], C[12];
  const int n = blockIdx.x;
  x += (long long)n * K * 3; y += (long long)n * K * 3;
  if (w) w += (long long)n * K;
  affine_moments(x, y, w, K, S, C, red);
  if (threadIdx.x == 0) {
    double Si[16];
    inv4(S, Si);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += C[i * 4 + k] * Si[k * 4 + j];
        M[n * 12 + i * 4 + j] = (float)acc;
      }
  }
}

__global__ __launch_bounds__(TPB) void affine_fit_bwd_kernel(const float* __restrict__ dM,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ w,
                                                             float* __restrict__ dx, float* __restrict__ dy,
                                                             float* __restrict__ dwgt, int K) {
  __shared__ double red[TPB / kWave];
  __shared__ double S[16], C[12];
  __shared__ double dC[12], dSs[16];  // dSs = dS + dS^T
  const int n = blockIdx.x;
  x += (long long)n * K * 3; y += (long long)n * K * 3;
  dx += (long long)n * K * 3; dy += (long long)n * K * 3;
  if (w) w += (long long)n * K;
  if (dwgt) dwgt += (long long)n * K;
  affine_moments(x, y, w, K, S, C, red);
  if (threadIdx.x == 0) {
    double Si[16], Mm[12], g[12], dS[16];
    inv4(S, Si);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        double a = 0;
        for (int k = 0; k < 4; ++k) a += C[i * 4 + k] * Si[k * 4 + j];
        Mm[i * 4 + j] = a;
        g[i * 4 + j] = dM[n * 12 + i * 4 + j];
      }
    // dC = dM S^-1 (S symmetric);  dS = -M^T dC
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        double a = 0;
        for (int k = 0; k < 4; ++k) a += g[i * 4 + k] * Si[j * 4 + k];
        dC[i * 4 + j] = a;
      }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += Mm[k * 4 + i] * dC[k * 4 + j];
        dS[i * 4 + j] = -a;
      }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) dSs[i * 4 + j] = dS[i * 4 + j] + dS[j * 4 + i];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const double X[4] = {x[k * 3], x[k * 3 + 1], x[k * 3 + 2], 1.0};
    const double Y[3] = {y[k * 3], y[k * 3 + 1], y[k * 3 + 2]};
    const double wk = w ? (double)w[k] : 1.0;
    for (int i = 0; i < 3; ++i) {
      double a = 0;
      for (int j = 0; j < 4; ++j) a += dSs[i * 4 + j] * X[j];
      for (int r = 0; r < 3; ++r) a += dC[r * 4 + i] * Y[r];
      dx[k * 3 + i] = (float)(wk * a);
      double b = 0;
      for (int j = 0; j < 4; ++j) b += dC[i * 4 + j] * X[j];
      dy[k * 3 + i] = (float)(wk * b);
    }
    if (dwgt) {  // S = sum w X X^T, C = sum w Y X^T  =>  dw_k = X^T dS X + Y^T dC X
      double a = 0;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) a += 0.5 * dSs[i * 4 + j] * X[i] * X[j];
      for (int r = 0; r < 3; ++r)
        for (int j = 0; j < 4; ++j) a += dC[r * 4 + j] * Y[r] * X[j];
      dwgt[k] = (float)a;
    }
  }
}

// ------------------------------------------------------------------------------------------
// 3x3 SVD by one-sided Jacobi (double): H = U diag(s) V^T, det(U) = det(V) = +1 completions for
// null directions.  Returns R0 = V U^T (orthogonal polar factor of H^T) and (U, s, V).
__device__ inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ void svd3(const double* H, double* U, double* s, double* V) {
  double A[9], Vm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) A[i] = H[i];
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; ++i) {
          alpha += A[i * 3 + p] * A[i * 3 + p];
          beta += A[i * 3 + q] * A[i * 3 + q];
          gamma += A[i * 3 + p] * A[i * 3 + q];
        }
        off += gamma * gamma;
        if (fabs(gamma) <= 1e-300 || gamma * gamma <= 1e-32 * alpha * beta) continue;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < 3; ++i) {
          const double ap = A[i * 3 + p], aq = A[i * 3 + q];
          A[i * 3 + p] = c * ap - sn * aq; A[i * 3 + q] = sn * ap + c * aq;
          const double vp = Vm[i * 3 + p], vq = Vm[i * 3 + q];
          Vm[i * 3 + p] = c * vp - sn * vq; Vm[i * 3 + q] = sn * vp + c * vq;
        }
      }
    if (off < 1e-60) break;
  }
  // column norms = singular values; sort descending
  double nv[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) nv[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (nv[ord[b]] > nv[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
  double u[3][3], v[3][3];
  const double tol = 1e-13 * (nv[ord[0]] > 0 ? nv[ord[0]] : 1.0);
  int rank = 0;
  for (int j = 0; j < 3; ++j) {
    const int c = ord[j];
    s[j] = nv[c];
    for (int i = 0; i < 3; ++i) v[j][i] = Vm[i * 3 + c];
    if (nv[c] > tol) {
      for (int i = 0; i < 3; ++i) u[j][i] = A[i * 3 + c] / nv[c];
      rank = j + 1;
    }
  }
  // make V right-handed (flip the last column if needed, with its u partner)
  {
    double cr[3];
    cross3(v[0], v[1], cr);
    const double det = cr[0] * v[2][0] + cr[1] * v[2][1] + cr[2] * v[2][2];
    if (det < 0) { for (int i = 0; i < 3; ++i) { v[2][i] = -v[2][i]; if (rank == 3) u[2][i] = -u[2][i]; } }
  }
  if (rank <= 1) {
    if (rank == 0) { for (int i = 0; i < 3; ++i) u[0][i] = v[0][i]; }
    // complete u with the frame that rotates v0 -> u0 minimally: use v1 projected
    double t[3];
    double d = u[0][0] * v[1][0] + u[0][1] * v[1][1] + u[0][2] * v[1][2];
    for (int i = 0; i < 3; ++i) t[i] = v[1][i] - d * u[0][i];
    double nn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    if (nn < 1e-8) {
      d = u[0][0] * v[2][0] + u[0][1] * v[2][1] + u[0][2] * v[2][2];
      for (int i = 0; i < 3; ++i) t[i] = v[2][i] - d * u[0][i];
      nn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    }
    for (int i = 0; i < 3; ++i) u[1][i] = t[i] / nn;
  }
  if (rank <= 2) cross3(u[0], u[1], u[2]);  // right-handed completion => det(U) = +1 given det(V) = +1
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) { U[i * 3 + j] = u[j][i]; V[i * 3 + j] = v[j][i]; }
}

struct RigidState {
  double c1[3], c2[3], H[9], U[9], s[3], V[9], R0[9], R[9], dsign;
};

__device__ void rigid_moments(const float* p1, const float* p2, const float* w, int K, RigidState* st,
                              double* red) {
  // centroids
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const double wk = w ? (double)w[k] : 1.0 / (double)K;
    for (int i = 0; i < 3; ++i) { a[i] += wk * p1[k * 3 + i]; a[3 + i] += wk * p2[k * 3 + i]; }
  }
  __shared__ double cen[6];
  for (int i = 0; i < 6; ++i) { double r = block_sum<double>(a[i], red); if (threadIdx.x == 0) cen[i] = r; }
  __syncthreads();
  double h[9];
  for (int i = 0; i < 9; ++i) h[i] = 0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const double wk = w ? (double)w[k] : 1.0;
    double q1[3], q2[3];
    for (int i = 0; i < 3; ++i) { q1[i] = (p1[k * 3 + i] - cen[i]) * wk; q2[i] = (p2[k * 3 + i] - cen[3 + i]) * wk; }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) h[i * 3 + j] += q1[i] * q2[j];
  }
  for (int i = 0; i < 9; ++i) { double r = block_sum<double>(h[i], red); if (threadIdx.x == 0) st->H[i] = r; }
  if (threadIdx.x == 0) {
    for (int i = 0; i < 3; ++i) { st->c1[i] = cen[i]; st->c2[i] = cen[3 + i]; }
    svd3(st->H, st->U, st->s, st->V);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += st->V[i * 3 + k] * st->U[j * 3 + k];
        st->R0[i * 3 + j] = acc;
      }
    const double* r = st->R0;
    const double det = r[0] * (r[4] * r[8] - r[5] * r[7]) - r[1] * (r[3] * r[8] - r[5] * r[6]) +
                       r[2] * (r[3] * r[7] - r[4] * r[6]);
    st->dsign = det > 0 ? 1.0 : (det < 0 ? -1.0 : 0.0);
    for (int i = 0; i < 9; ++i) st->R[i] = st->R0[i] * (i >= 6 ? st->dsign : 1.0);  // scale LAST ROW
  }
  __syncthreads();
}

__global__ __launch_bounds__(TPB) void rigid_fit_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ w, float* __restrict__ M,
                                                            int K) {
  __shared__ double red[TPB / kWave];
  __shared__ RigidState st;
  const int n = blockIdx.x;
  x += (long long)n * K * 3; y += (long long)n * K * 3;
  if (w) w += (long long)n * K;
  rigid_moments(x, y, w, K, &st, red);
  if (threadIdx.x == 0) {
    for (int i = 0; i < 3; ++i) {
      double t = st.c2[i];
      for (int j = 0; j < 3; ++j) {
        M[n * 12 + i * 4 + j] = (float)st.R[i * 3 + j];
        t -= st.R[i * 3 + j] * st.c1[j];
      }
      M[n * 12 + i * 4 + 3] = (float)t;
    }
  }
}

__global__ __launch_bounds__(TPB) void rigid_fit_bwd_kernel(const float* __restrict__ dM,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ w, float* __restrict__ dx,
                                                            float* __restrict__ dy, float* __restrict__ dwgt,
                                                            int K) {
  __shared__ double red[TPB / kWave];
  __shared__ RigidState st;
  __shared__ double dH[9], dc1[3], dc2[3];
  const int n = blockIdx.x;
  x += (long long)n * K * 3; y += (long long)n * K * 3;
  dx += (long long)n * K * 3; dy += (long long)n * K * 3;
  if (w) w += (long long)n * K;
  if (dwgt) dwgt += (long long)n * K;
  rigid_moments(x, y, w, K, &st, red);
  if (threadIdx.x == 0) {
    double gR[9], gT[3];
    for (int i = 0; i < 3; ++i) {
      gT[i] = dM[n * 12 + i * 4 + 3];
      for (int j = 0; j < 3; ++j) gR[i * 3 + j] = dM[n * 12 + i * 4 + j];
    }
    // T = c2 - R c1
    for (int i = 0; i < 3; ++i) {
      dc2[i] = gT[i];
      double a = 0;
      for (int j = 0; j < 3; ++j) { a -= st.R[j * 3 + i] * gT[j]; gR[i * 3 + j] -= gT[i] * st.c1[j]; }
      dc1[i] = a;
    }
    // R = diag(1,1,d) R0  =>  G0 = diag(1,1,d) gR
    double G0[9];
    for (int i = 0; i < 9; ++i) G0[i] = gR[i] * (i >= 6 ? st.dsign : 1.0);
    // polar-factor adjoint: B = V^T G0 U ; E_ij = (B_ij - B_ji)/(s_i + s_j) ; dH = U E^T V^T
    double B[9], E[9], tmp[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += st.V[k * 3 + i] * G0[k * 3 + j];
        tmp[i * 3 + j] = a;
      }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += tmp[i * 3 + k] * st.U[k * 3 + j];
        B[i * 3 + j] = a;
      }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double den = st.s[i] + st.s[j];
        E[i * 3 + j] = den > 1e-300 ? (B[i * 3 + j] - B[j * 3 + i]) / den : 0.0;
      }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += st.U[i * 3 + k] * E[j * 3 + k];
        tmp[i * 3 + j] = a;  // U E^T
      }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += tmp[i * 3 + k] * st.V[j * 3 + k];
        dH[i * 3 + j] = a;
      }
  }
  __syncthreads();
  // dq1_k = dH q2_k, dq2_k = dH^T q1_k ; q = (p - c) w ; c = sum w p (or mean)
  double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const double wk = w ? (double)w[k] : 1.0;
    double q1[3], q2[3];
    for (int i = 0; i < 3; ++i) { q1[i] = (x[k * 3 + i] - st.c1[i]) * wk; q2[i] = (y[k * 3 + i] - st.c2[i]) * wk; }
    for (int i = 0; i < 3; ++i) {
      double a = 0, b = 0;
      for (int j = 0; j < 3; ++j) { a += dH[i * 3 + j] * q2[j]; b += dH[j * 3 + i] * q1[j]; }
      s1[i] += wk * a; s2[i] += wk * b;
    }
  }
  __shared__ double tot[6];
  for (int i = 0; i < 3; ++i) {
    double r = block_sum<double>(s1[i], red); if (threadIdx.x == 0) tot[i] = r;
    r = block_sum<double>(s2[i], red); if (threadIdx.x == 0) tot[3 + i] = r;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const double wk = w ? (double)w[k] : 1.0;
    const double cw = w ? (double)w[k] : 1.0 / (double)K;  // d c / d p_k
    double q1[3], q2[3];
    for (int i = 0; i < 3; ++i) { q1[i] = (x[k * 3 + i] - st.c1[i]) * wk; q2[i] = (y[k * 3 + i] - st.c2[i]) * wk; }
    for (int i = 0; i < 3; ++i) {
      double a = 0, b = 0;
      for (int j = 0; j < 3; ++j) { a += dH[i * 3 + j] * q2[j]; b += dH[j * 3 + i] * q1[j]; }
      dx[k * 3 + i] = (float)(wk * a + cw * (dc1[i] - tot[i]));
      dy[k * 3 + i] = (float)(wk * b + cw * (dc2[i] - tot[3 + i]));
    }
    if (dwgt && w) {
      // c = sum w p  =>  dc/dw_k = p_k ;  q_k = (p_k - c) w_k  =>  dq_k/dw_k = p_k - c
      double acc = 0;
      for (int i = 0; i < 3; ++i) {
        double a = 0, b = 0;
        for (int j = 0; j < 3; ++j) { a += dH[i * 3 + j] * q2[j]; b += dH[j * 3 + i] * q1[j]; }
        acc += a * (x[k * 3 + i] - st.c1[i]) + b * (y[k * 3 + i] - st.c2[i]);
        acc += (dc1[i] - tot[i]) * x[k * 3 + i] + (dc2[i] - tot[3 + i]) * y[k * 3 + i];
      }
      dwgt[k] = (float)acc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// homogeneous inverse of [M; 0 0 0 1] (general 4x4 inverse like torch.inverse), one thread per sample
__global__ void affine_inverse_fwd_kernel(const float* __restrict__ M, float* __restrict__ Minv, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double a[16], inv[16];
  for (int i = 0; i < 12; ++i) a[i] = M[n * 12 + i];
  a[12] = a[13] = a[14] = 0; a[15] = 1;
  inv4(a, inv);
  for (int i = 0; i < 12; ++i) Minv[n * 12 + i] = (float)inv[i];
}
// dA = -B^T dB B^T with B = A^-1, dB's last row = 0; keep the top 3 rows
__global__ void affine_inverse_bwd_kernel(const float* __restrict__ dMinv, const float* __restrict__ Minv,
                                          float* __restrict__ dM, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double B[16], G[16], T[16];
  for (int i = 0; i < 12; ++i) { B[i] = Minv[n * 12 + i]; G[i] = dMinv[n * 12 + i]; }
  B[12] = B[13] = B[14] = 0; B[15] = 1;
  G[12] = G[13] = G[14] = G[15] = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double a = 0;
      for (int k = 0; k < 4; ++k) a += B[k * 4 + i] * G[k * 4 + j];
      T[i * 4 + j] = a;  // B^T G
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      double a = 0;
      for (int k = 0; k < 4; ++k) a += T[i * 4 + k] * B[j * 4 + k];
      dM[n * 12 + i * 4 + j] = (float)(-a);
    }
}

// ------------------------------------------------------------------------------------------
// TPS: assemble + LU + solve
constexpr int LU_TPB = 1024;

// A (n x lda) row-major doubles, n = T + 4
__global__ __launch_bounds__(256) void tps_assemble_kernel(const float* __restrict__ ctrl,
                                                           const float* __restrict__ lmbda,
                                                           const float* __restrict__ w, double* __restrict__ A,
                                                           int T, int lda, size_t a_stride,
                                                           const int* __restrict__ retry_info = nullptr) {
  const int b = blockIdx.z;
  if (retry_info && retry_info[b] != 2) return;      // retry pass: only the systems whose cluster factorisation gave up
  const int n = T + 4;
  const int j = blockIdx.x * 16 + (threadIdx.x & 15);
  const int i = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= n || j >= n) return;
  const float* c = ctrl + (long long)b * T * 3;
  double v;
  if (i < T && j < T) {
    const float dz = c[i * 3] - c[j * 3], dy = c[i * 3 + 1] - c[j * 3 + 1], dx = c[i * 3 + 2] - c[j * 3 + 2];
    float u = tps_u_from_d2(tps_d2(dz, dy, dx));             // same d2 and U as the evaluators (common.h)
    const float lam = lmbda[b];
    if (w) {
      // reciprocal of the WHOLE diag-embedded matrix (+1e-6), keymorph/keypoint_aligners.py:298-302
      const float wv = (i == j) ? w[(long long)b * T + i] : 0.f;
      u = u + (1.f / (wv + 1e-6f)) * lam;
    } else if (i == j) {
      u = u + lam;
    }
    v = (double)u;
  } else if (i < T) {
    const int k = j - T;
    v = (k == 0) ? 1.0 : (double)c[i * 3 + k - 1];
  } else if (j < T) {
    const int k = i - T;
    v = (k == 0) ? 1.0 : (double)c[j * 3 + k - 1];
  } else {
    v = 0.0;
  }
  A[(size_t)b * a_stride + (size_t)i * lda + j] = v;
}

// (KMH_LU_EXP: timing experiments only, tools/build_exp_lib.sh -DKMH_LU_EXP=<bits>; results are garbage with any bit set.
//  1 = no trailing update, 2 = no U12 strip, 4 = no pivot search, 8 = no rank-1 update inside the panel)
#ifndef KMH_LU_EXP
#define KMH_LU_EXP 0
#endif
#ifndef KMH_LU_KEYED
#define KMH_LU_KEYED 1          // 0: round 3's four-barrier column step (A/B: tools/build_exp_lib.sh -DKMH_LU_KEYED=0)
#endif
// Blocked right-looking LU with partial pivoting, one workgroup per sample.
// LDS: sP[m][NB+1] (panel, rows k0..n) then sU[NB][ncols] (U12 strip).
template <int NB, bool MFMA64>
__global__ __launch_bounds__(LU_TPB) void tps_lu_kernel(double* __restrict__ Aall, int* __restrict__ ipiv_all,
                                                        int* __restrict__ info_all, int n, int lda,
                                                        size_t a_stride, int retry_only = 0) {
  if (retry_only && info_all[blockIdx.x] != 2) return;      // (uniform per workgroup: before any barrier)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int PS = NB + 1;
  double* sP = smem;                       // [n][PS]
  double* sU = smem + (size_t)n * PS;      // [NB][n]
  __shared__ double s_val[LU_TPB / kWave];
  __shared__ int s_idx[LU_TPB / kWave];
  __shared__ int s_piv[NB];
  __shared__ unsigned long long s_key[NB];      // KMH_LU_KEYED: (magnitude, row) key of every panel column's pivot
  // Row interchanges are NOT carried out in memory: rowmap[i] = the physical (= original) row that currently sits at
  // position i of the pivoted order.  Every access to a row of A goes through it, a pivot swaps two of its entries, and the
  // solve reads the same map (ipiv_all receives rowmap, not LAPACK's sequential interchanges).  Swapping the rows of the
  // columns outside the panel in global memory was 16 dependent load / store round trips per panel and thread.
  int* rowmap = reinterpret_cast<int*>(smem + (size_t)n * PS + (size_t)NB * n);     // [n]
  double* A = Aall + (size_t)blockIdx.x * a_stride;
  int* ipiv = ipiv_all + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wid = tid / kWave;
  int bad = 0;
  for (int e = tid; e < n; e += LU_TPB) rowmap[e] = e;
  __syncthreads();

  for (int k0 = 0; k0 < n; k0 += NB) {
    const int nb = (n - k0 < NB) ? (n - k0) : NB;
    const int m = n - k0;
    // 1. panel -> LDS
    for (int e = tid; e < m * nb; e += LU_TPB) {
      const int r = e / nb, c = e % nb;
      sP[r * PS + c] = A[(size_t)rowmap[k0 + r] * lda + k0 + c];
    }
    __syncthreads();
    // 2. unblocked LU of the panel
    if constexpr (KMH_LU_KEYED) {
      // Round 4: two barriers per column instead of four.  The pivot of column j+1 is found WHILE column j's rank-1 update
      // writes it: every thread turns |its new entry| into a 64-bit key (magnitude bits with the low 11 replaced by
      // 2047 - row, so that equal magnitudes resolve to the smallest row like LAPACK's idamax and a plain unsigned maximum is
      // order independent), a wave reduces with shuffles and ONE LDS atomicMax per wave publishes it -- no second reduction
      // stage on wave 0, no separate search phase re-reading the column.  (Magnitudes are compared to 42 mantissa bits: a
      // pivot within 2^-42 of the largest is as good as the largest.)
      auto row_key = [](double v, int r) -> unsigned long long {
        const double a = fabs(v);
        const unsigned long long b = (a == a) ? (unsigned long long)__double_as_longlong(a) : 0ull;     // NaN never wins
        return (b & ~0x7FFull) | (unsigned long long)(0x7FF - r);
      };
      auto publish = [&](unsigned long long key, int col) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
          const unsigned long long ok = __shfl_xor(key, o, kWave);
          key = ok > key ? ok : key;
        }
        if (lane == 0 && key) atomicMax(&s_key[col], key);
      };
      if (tid < NB) s_key[tid] = 0ull;
      __syncthreads();
      {
        unsigned long long key = 0ull;
        for (int r = tid; r < m; r += LU_TPB) { const unsigned long long k = row_key(sP[r * PS], r); key = k > key ? k : key; }
        publish(key, 0);
      }
      for (int j = 0; j < nb; ++j) {
        __syncthreads();               // column j's keys are in (and the previous column's update is visible)
        const unsigned long long kj = s_key[j];
        const int p = 0x7FF - (int)(kj & 0x7FFull);
        if (tid == 0 && (kj >> 11) == 0ull) bad = 1;       // the whole column is zero (or NaN): singular
        const bool have = (kj >> 11) != 0ull;
        if (have && p != j && tid < nb) {
          const double t = sP[j * PS + tid];
          sP[j * PS + tid] = sP[p * PS + tid];
          sP[p * PS + tid] = t;
        }
        if (have && p != j && tid == LU_TPB - 1) { const int t = rowmap[k0 + j]; rowmap[k0 + j] = rowmap[k0 + p]; rowmap[k0 + p] = t; }
        __syncthreads();
        const double pinv = 1.0 / sP[j * PS + j];
        unsigned long long key = 0ull;
        for (int r = j + 1 + tid; r < m; r += LU_TPB) {
          const double l = sP[r * PS + j] * pinv;
          sP[r * PS + j] = l;
          for (int c = j + 1; c < nb; ++c) sP[r * PS + c] -= l * sP[j * PS + c];
          if (j + 1 < nb) { const unsigned long long k = row_key(sP[r * PS + j + 1], r); key = k > key ? k : key; }
        }
        if (j + 1 < nb) publish(key, j + 1);
      }
      __syncthreads();
    } else {
    for (int j = 0; j < nb; ++j) {
      // pivot search over rows j..m-1 of column j
      double best = -1.0;
      int bi = j;
      if (KMH_LU_EXP & 4) { if (tid == 0) s_piv[j] = j; __syncthreads(); } else {
      for (int r = j + tid; r < m; r += LU_TPB) {
        const double v = fabs(sP[r * PS + j]);
        if (v > best) { best = v; bi = r; }
      }
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o, kWave);
        const int oi = __shfl_xor(bi, o, kWave);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (lane == 0) { s_val[wid] = best; s_idx[wid] = bi; }
      __syncthreads();
      if (wid == 0) {
        best = lane < LU_TPB / kWave ? s_val[lane] : -1.0;
        bi = lane < LU_TPB / kWave ? s_idx[lane] : 0x7fffffff;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
          const double ov = __shfl_xor(best, o, kWave);
          const int oi = __shfl_xor(bi, o, kWave);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { s_piv[j] = bi; if (!(best > 0.0)) bad = 1; }
      }
      __syncthreads();
      }
      const int p = s_piv[j];
      if (p != j && tid < nb) {
        const double t = sP[j * PS + tid];
        sP[j * PS + tid] = sP[p * PS + tid];
        sP[p * PS + tid] = t;
      }
      if (p != j && tid == LU_TPB - 1) { const int t = rowmap[k0 + j]; rowmap[k0 + j] = rowmap[k0 + p]; rowmap[k0 + p] = t; }
      __syncthreads();
      // scale + rank-1 update of the remaining panel columns: one thread per row (rows are
      // private to their thread; row j is read-only here), so no extra barrier is needed
      const double pinv = 1.0 / sP[j * PS + j];
      if (!(KMH_LU_EXP & 8))
      for (int r = j + 1 + tid; r < m; r += LU_TPB) {
        const double l = sP[r * PS + j] * pinv;
        sP[r * PS + j] = l;
        for (int c = j + 1; c < nb; ++c) sP[r * PS + c] -= l * sP[j * PS + c];
      }
      __syncthreads();
    }
    }
    // 3. (no interchanges in memory: rowmap)
    // 4. panel back to global
    for (int e = tid; e < m * nb; e += LU_TPB) {
      const int r = e / nb, c = e % nb;
      A[(size_t)rowmap[k0 + r] * lda + k0 + c] = sP[r * PS + c];
    }
    __syncthreads();
    const int ncols = n - k0 - nb;
    if (ncols > 0) {
      // 5. U12 = L11^-1 A12, one thread per column
      if (!(KMH_LU_EXP & 2))
      for (int c = tid; c < ncols; c += LU_TPB) {
        double col[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) col[r] = (r < nb) ? A[(size_t)rowmap[k0 + r] * lda + k0 + nb + c] : 0.0;
#pragma unroll
        for (int r = 1; r < NB; ++r) {
          double a = col[r];
#pragma unroll
          for (int k = 0; k < r; ++k) a -= sP[r * PS + k] * col[k];
          col[r] = a;
        }
#pragma unroll
        for (int r = 0; r < NB; ++r)
          if (r < nb) { sU[r * n + c] = col[r]; A[(size_t)rowmap[k0 + r] * lda + k0 + nb + c] = col[r]; }
      }
      __syncthreads();
      // 6. trailing update A22 -= L21 U12.
      // Round 3: on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), one 16 x 16 tile of A22 per wave and trip: the 4x4
      // register tiles below read 8 doubles from LDS per 16 multiply-adds and were bound by exactly that (the panel
      // factorisation, which the round-2 experiments went after, is the smaller part); a matrix-core tile reads 2
      // per 16.  A / B operands: lane l holds L[r0 + (l & 15)][4 s + (l >> 4)] and U[4 s + (l >> 4)][c0 + (l & 15)];
      // result register q of lane l is row (l >> 4) + 4 q, column l & 15.
      if (KMH_LU_EXP & 1) {
      } else if (MFMA64) {
        typedef double kmh_d4 __attribute__((ext_vector_type(4)));
        const int t16 = (ncols + 15) / 16;
        const int li = lane & 15, lk = lane >> 4;
        // TU tiles per trip: their A22 loads are issued together (one tile at a time, every tile started with an exposed
        // global round trip: ~60 of them per wave and panel at the start of the factorisation)
        constexpr int TU = 4;
        constexpr int NWV = LU_TPB / kWave;
        for (int tile0 = wid; tile0 < t16 * t16; tile0 += NWV * TU) {
          double oldv[TU][4];
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            const int tile = tile0 + u * NWV;
            const int r0 = (tile / t16) * 16, c0 = (tile % t16) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = r0 + lk + 4 * q, c = c0 + li;
              oldv[u][q] = (tile < t16 * t16 && r < ncols && c < ncols) ? A[(size_t)rowmap[k0 + nb + r] * lda + k0 + nb + c] : 0.0;
            }
          }
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            const int tile = tile0 + u * NWV;
            if (tile >= t16 * t16) break;                      // wave-uniform
            const int r0 = (tile / t16) * 16, c0 = (tile % t16) * 16;
            kmh_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < NB / 4; ++s4) {
              const int k = 4 * s4 + lk;
              const double a = (r0 + li < ncols && k < nb) ? sP[(nb + r0 + li) * PS + k] : 0.0;
              const double b = (c0 + li < ncols && k < nb) ? sU[k * n + c0 + li] : 0.0;
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = r0 + lk + 4 * q, c = c0 + li;
              if (r < ncols && c < ncols) A[(size_t)rowmap[k0 + nb + r] * lda + k0 + nb + c] = oldv[u][q] - acc[q];
            }
          }
        }
      } else {
      const int tr = (ncols + 3) / 4, tc = (ncols + 3) / 4;  // A22 is ncols x ncols (square)
      for (int tile = tid; tile < tr * tc; tile += LU_TPB) {
        const int r0 = (tile / tc) * 4, c0 = (tile % tc) * 4;
        double acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
#pragma unroll 4
        for (int k = 0; k < nb; ++k) {
          double l[4], u[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) l[i] = (r0 + i < ncols) ? sP[(nb + r0 + i) * PS + k] : 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) u[j] = (c0 + j < ncols) ? sU[k * n + c0 + j] : 0.0;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += l[i] * u[j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (r0 + i < ncols && c0 + j < ncols)
              A[(size_t)rowmap[k0 + nb + r0 + i] * lda + k0 + nb + c0 + j] -= acc[i][j];
      }
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < n; e += LU_TPB) ipiv[e] = rowmap[e];
  if (tid == 0) info_all[blockIdx.x] = bad;
}

// ---------------------------------------------------------------------------------------------------------------------
// Cluster variant (round 4): G workgroups per system.  The phase breakdown of the one-workgroup kernel
// put 45 % of a fit into the U12 strip + trailing update, which are not arithmetic but ONE CU's path to L2 (every panel reads
// and writes the whole trailing matrix).  Here workgroup 0 of a system factorises each panel exactly as above and publishes
// it (+ its pivots); all G workgroups then take a contiguous slice of the trailing COLUMNS each -- U12 strip and A22 update of
// a column slice need nothing from another slice -- and report back.  Two hand-offs per panel through two monotone counters
// in global memory: `ready` (panels published, written by workgroup 0) and `done` (slice updates finished, incremented by
// every workgroup), agent-scope release / acquire (the workgroups may sit on different XCDs whose L2s are not coherent).
// Every wait is bounded (a workgroup that gives up raises `abort_flag`, which ends every other wait too, and the fit
// reports info = 2): the launcher only uses this kernel when all N * G workgroups are resident at once.
constexpr int CL_SPIN_LIMIT = 1 << 20;      // x ~1 us per probe: about a second, then give up

__device__ __forceinline__ bool cl_wait_ge(int* counter, int value, int* abort_flag) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int it = 0, ok = 1;
    // probe with RELAXED loads (an acquire per probe invalidates this CU's caches every time), ONE acquire fence at the end
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
      if (++it > CL_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ok = ok;
  }
  __syncthreads();                       // the acquiring lane's invalidate covers this CU's L1: the others read after the barrier
  const bool ok = s_ok != 0;
  __syncthreads();                       // (s_ok may be rewritten by the next wait)
  return ok;
}
// Every thread's global stores of this phase -> visible to the other workgroups, then the counter moves.  The barrier orders all
// waves' stores before lane 0 (they have left for L2 when a wave passes it), and lane 0's agent-scope RELEASE is cumulative over
// that: its L2 write-back covers the whole workgroup's lines.  (A __threadfence() in every thread as well -- 16 waves each
// writing the L2 back -- cost 0.42 ms per fit and changed nothing.)
__device__ __forceinline__ void cl_publish(int* counter, bool add, int value) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (add) __hip_atomic_fetch_add(counter, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(counter, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int NB>
__global__ __launch_bounds__(LU_TPB) void tps_lu_cluster_kernel(double* __restrict__ Aall, int* __restrict__ ipiv_all,
                                                                int* __restrict__ info_all, int n, int lda, size_t a_stride,
                                                                int* __restrict__ pivs_all /* (N, n) */,
                                                                int* __restrict__ sync_all /* (N, 4): ready, done, abort */) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int PS = NB + 1;
  double* sP = smem;                       // [n][PS]
  double* sU = smem + (size_t)n * PS;      // [NB][n]
  int* rowmap = reinterpret_cast<int*>(smem + (size_t)n * PS + (size_t)NB * n);     // [n]
  __shared__ unsigned long long s_key[NB];
  __shared__ int s_pv[NB];
  const int G = gridDim.x, g = blockIdx.x, b = blockIdx.y;
  double* A = Aall + (size_t)b * a_stride;
  int* pivs = pivs_all + (size_t)b * n;
  int* ready = sync_all + 4 * b;
  int* done = ready + 1;
  int* abort_flag = ready + 2;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wid = tid / kWave;
  int bad = 0;
  for (int e = tid; e < n; e += LU_TPB) rowmap[e] = e;
  __syncthreads();
  auto row_key = [](double v, int r) -> unsigned long long {
    const double a = fabs(v);
    const unsigned long long bits = (a == a) ? (unsigned long long)__double_as_longlong(a) : 0ull;
    return (bits & ~0x7FFull) | (unsigned long long)(0x7FF - r);
  };
  auto publish_key = [&](unsigned long long key, int col) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
      const unsigned long long ok = __shfl_xor(key, o, kWave);
      key = ok > key ? ok : key;
    }
    if (lane == 0 && key) atomicMax(&s_key[col], key);
  };

  int panel = 0;
  for (int k0 = 0; k0 < n; k0 += NB, ++panel) {
    const int nb = (n - k0 < NB) ? (n - k0) : NB;
    const int m = n - k0;
    const int ncols = n - k0 - nb;
    if (g == 0) {
      // every slice update with the previous panel has been reported (and is visible after the acquire)
      if (!cl_wait_ge(done, panel * G, abort_flag)) { bad = 2; break; }
      for (int e = tid; e < m * nb; e += LU_TPB) {
        const int r = e / nb, c = e % nb;
        sP[r * PS + c] = A[(size_t)rowmap[k0 + r] * lda + k0 + c];
      }
      if (tid < NB) s_key[tid] = 0ull;
      __syncthreads();
      {
        unsigned long long key = 0ull;
        for (int r = tid; r < m; r += LU_TPB) { const unsigned long long k = row_key(sP[r * PS], r); key = k > key ? k : key; }
        publish_key(key, 0);
      }
      for (int j = 0; j < nb; ++j) {
        __syncthreads();
        const unsigned long long kj = s_key[j];
        const bool have = (kj >> 11) != 0ull;
        const int p = have ? 0x7FF - (int)(kj & 0x7FFull) : j;
        if (tid == 0) { if (!have) bad = 1; pivs[k0 + j] = p; }
        if (p != j && tid < nb) {
          const double t = sP[j * PS + tid];
          sP[j * PS + tid] = sP[p * PS + tid];
          sP[p * PS + tid] = t;
        }
        if (p != j && tid == LU_TPB - 1) { const int t = rowmap[k0 + j]; rowmap[k0 + j] = rowmap[k0 + p]; rowmap[k0 + p] = t; }
        __syncthreads();
        const double pinv = 1.0 / sP[j * PS + j];
        unsigned long long key = 0ull;
        for (int r = j + 1 + tid; r < m; r += LU_TPB) {
          const double l = sP[r * PS + j] * pinv;
          sP[r * PS + j] = l;
          for (int c = j + 1; c < nb; ++c) sP[r * PS + c] -= l * sP[j * PS + c];
          if (j + 1 < nb) { const unsigned long long k = row_key(sP[r * PS + j + 1], r); key = k > key ? k : key; }
        }
        if (j + 1 < nb) publish_key(key, j + 1);
      }
      __syncthreads();
      for (int e = tid; e < m * nb; e += LU_TPB) {
        const int r = e / nb, c = e % nb;
        A[(size_t)rowmap[k0 + r] * lda + k0 + c] = sP[r * PS + c];
      }
      cl_publish(ready, false, panel + 1);
    } else {
      if (!cl_wait_ge(ready, panel + 1, abort_flag)) { bad = 2; break; }
      // this workgroup's copy of the row map follows the published pivots; then the factorised panel comes in
      if (tid < nb) s_pv[tid] = pivs[k0 + tid];
      __syncthreads();
      if (tid == 0)
        for (int j = 0; j < nb; ++j) {
          const int p = s_pv[j];
          if (p != j) { const int t = rowmap[k0 + j]; rowmap[k0 + j] = rowmap[k0 + p]; rowmap[k0 + p] = t; }
        }
      __syncthreads();
      if (ncols > 0)
        for (int e = tid; e < m * nb; e += LU_TPB) {
          const int r = e / nb, c = e % nb;
          sP[r * PS + c] = A[(size_t)rowmap[k0 + r] * lda + k0 + c];
        }
    }
    __syncthreads();
    if (ncols <= 0) continue;            // the last panel has nothing behind it
    // ---- this workgroup's slice of the trailing columns: whole 16-column tiles, dealt contiguously
    const int t16 = (ncols + 15) / 16;
    const int per = (t16 + G - 1) / G;
    const int ct0 = g * per, ct1 = (ct0 + per < t16) ? ct0 + per : t16;       // column tiles [ct0, ct1)
    const int c_lo = ct0 * 16, c_hi = (ct1 * 16 < ncols) ? ct1 * 16 : ncols;
    if (ct0 < ct1) {
      // U12 = L11^-1 A12 for the slice, one thread per column
      for (int c = c_lo + tid; c < c_hi; c += LU_TPB) {
        double col[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) col[r] = (r < nb) ? A[(size_t)rowmap[k0 + r] * lda + k0 + nb + c] : 0.0;
#pragma unroll
        for (int r = 1; r < NB; ++r) {
          double a = col[r];
#pragma unroll
          for (int k = 0; k < r; ++k) a -= sP[r * PS + k] * col[k];
          col[r] = a;
        }
#pragma unroll
        for (int r = 0; r < NB; ++r)
          if (r < nb) { sU[r * n + c] = col[r]; A[(size_t)rowmap[k0 + r] * lda + k0 + nb + c] = col[r]; }
      }
      __syncthreads();
      // A22[:, slice] -= L21 U12[:, slice] on the fp64 matrix cores (as tps_lu_kernel)
      typedef double kmh_d4 __attribute__((ext_vector_type(4)));
      const int li = lane & 15, lk = lane >> 4;
      constexpr int TU = 4;
      constexpr int NWV = LU_TPB / kWave;
      const int nct = ct1 - ct0, ntile = t16 * nct;
      for (int tile0 = wid; tile0 < ntile; tile0 += NWV * TU) {
        double oldv[TU][4];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const int tile = tile0 + u * NWV;
          const int r0 = (tile / nct) * 16, c0 = (ct0 + tile % nct) * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = r0 + lk + 4 * q, c = c0 + li;
            oldv[u][q] = (tile < ntile && r < ncols && c < ncols) ? A[(size_t)rowmap[k0 + nb + r] * lda + k0 + nb + c] : 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const int tile = tile0 + u * NWV;
          if (tile >= ntile) break;                          // wave-uniform
          const int r0 = (tile / nct) * 16, c0 = (ct0 + tile % nct) * 16;
          kmh_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s4 = 0; s4 < NB / 4; ++s4) {
            const int k = 4 * s4 + lk;
            const double a = (r0 + li < ncols && k < nb) ? sP[(nb + r0 + li) * PS + k] : 0.0;
            const double bb = (c0 + li < ncols && k < nb) ? sU[k * n + c0 + li] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = r0 + lk + 4 * q, c = c0 + li;
            if (r < ncols && c < ncols) A[(size_t)rowmap[k0 + nb + r] * lda + k0 + nb + c] = oldv[u][q] - acc[q];
          }
        }
      }
    }
    cl_publish(done, true, 1);
  }
  if (g == 0) {
    int* ipiv = ipiv_all + (size_t)b * n;
    for (int e = tid; e < n; e += LU_TPB) ipiv[e] = rowmap[e];
    if (tid == 0) info_all[b] = bad;
  } else if (bad && tid == 0) {
    atomicMax(&info_all[b], bad);        // (a worker that gave up: make sure the fit reports it)
  }
}

// Solve (P A = L U) x = b for 3 right-hand sides.  b/x (n x 3) doubles in LDS.  One workgroup
// per sample; blocked substitution: each 16x16 diagonal block is staged in LDS and solved by
// three lanes (one per right-hand side), the off-diagonal strip is applied by all threads
// (each reads 128 contiguous bytes of its own row).
constexpr int SB = 16;
__global__ __launch_bounds__(LU_TPB) void tps_solve_kernel(const double* __restrict__ Aall,
                                                           const int* __restrict__ ipiv_all,
                                                           const float* __restrict__ rhs /* (N, rows, 3) */,
                                                           int rhs_rows, float* __restrict__ out /* (N,n,3) | null */,
                                                           double* __restrict__ out64 /* (N,n,3) | null */,
                                                           int n, int lda, size_t a_stride,
                                                           const int* __restrict__ info_all = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double sb[];  // [n][3] then int piv[n]
  __shared__ double sD[SB][SB + 1];
  int* spiv = reinterpret_cast<int*>(sb + (size_t)n * 3);
  const double* A = Aall + (size_t)blockIdx.x * a_stride;
  const int* ipiv = ipiv_all + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x;
  for (int e = tid; e < n; e += LU_TPB) spiv[e] = ipiv[e];          // the factorisation's row map (tps_lu_kernel)
  __syncthreads();
  for (int e = tid; e < n * 3; e += LU_TPB) {                      // P b: position i of the pivoted order takes row spiv[i]
    const int r = spiv[e / 3], j = e % 3;
    sb[e] = (r < rhs_rows) ? (double)rhs[((size_t)blockIdx.x * rhs_rows + r) * 3 + j] : 0.0;
  }
  __syncthreads();
  // forward: L y = b (unit lower)
  for (int k0 = 0; k0 < n; k0 += SB) {
    const int nb = (n - k0 < SB) ? (n - k0) : SB;
    if (tid < SB * SB) {
      const int r = tid / SB, c = tid % SB;
      sD[r][c] = (r < nb && c < nb) ? A[(size_t)spiv[k0 + r] * lda + k0 + c] : 0.0;
    }
    __syncthreads();
    // the block's triangular solve on ONE wave, lane = (row, right-hand side), columns eliminated in order with the solved
    // value passed by a shuffle (three lanes walking the rows one after the other took ~3.5 us per block: 120 dependent
    // LDS round trips)
    if (tid < 64) {
      const int r = tid / 3, j = tid - 3 * r;
      const bool act = r < nb;
      double val = act ? sb[(k0 + r) * 3 + j] : 0.0;
      for (int k = 0; k + 1 < nb; ++k) {
        const double xk = __shfl(val, 3 * k + j, 64);
        if (act && r > k) val -= sD[r][k] * xk;
      }
      if (act) sb[(k0 + r) * 3 + j] = val;
    }
    __syncthreads();
    for (int r = k0 + nb + tid; r < n; r += LU_TPB) {
      const double* Arow = A + (size_t)spiv[r] * lda;
      double a0 = 0, a1 = 0, a2 = 0;
      for (int k = 0; k < nb; ++k) {
        const double l = Arow[k0 + k];
        a0 += l * sb[(k0 + k) * 3]; a1 += l * sb[(k0 + k) * 3 + 1]; a2 += l * sb[(k0 + k) * 3 + 2];
      }
      sb[r * 3] -= a0; sb[r * 3 + 1] -= a1; sb[r * 3 + 2] -= a2;
    }
    __syncthreads();
  }
  // backward: U x = y
  const int nblk = (n + SB - 1) / SB;
  for (int bk = nblk - 1; bk >= 0; --bk) {
    const int k0 = bk * SB;
    const int nb = (n - k0 < SB) ? (n - k0) : SB;
    if (tid < SB * SB) {
      const int r = tid / SB, c = tid % SB;
      sD[r][c] = (r < nb && c < nb) ? A[(size_t)spiv[k0 + r] * lda + k0 + c] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {                                   // as above, from the last column up; the lane of row k divides
      const int r = tid / 3, j = tid - 3 * r;
      const bool act = r < nb;
      double val = act ? sb[(k0 + r) * 3 + j] : 0.0;
      const double ukk = act ? sD[r][r] : 1.0;
      for (int k = nb - 1; k >= 0; --k) {
        if (r == k) val = val / ukk;
        const double xk = __shfl(val, 3 * k + j, 64);
        if (act && r < k) val -= sD[r][k] * xk;
      }
      if (act) sb[(k0 + r) * 3 + j] = val;
    }
    __syncthreads();
    for (int r = tid; r < k0; r += LU_TPB) {
      const double* Urow = A + (size_t)spiv[r] * lda;
      double a0 = 0, a1 = 0, a2 = 0;
      for (int k = 0; k < nb; ++k) {
        const double u = Urow[k0 + k];
        a0 += u * sb[(k0 + k) * 3]; a1 += u * sb[(k0 + k) * 3 + 1]; a2 += u * sb[(k0 + k) * 3 + 2];
      }
      sb[r * 3] -= a0; sb[r * 3 + 1] -= a1; sb[r * 3 + 2] -= a2;
    }
    __syncthreads();
  }
  // info == 2: the factorisation gave up waiting for a workgroup of its cluster (tps_lu_cluster_kernel: its workgroups were not
  // all resident) AND the retry pass of kmh_tps_fit_fwd did not replace it (it always does: tps_lu_kernel reports 0 or 1).
  // Belt and braces: such a result must not pass as coefficients, so it becomes NaN.
  const bool poisoned = info_all && info_all[blockIdx.x] == 2;
  for (int e = tid; e < n * 3; e += LU_TPB) {
    const double v = poisoned ? __longlong_as_double(0x7ff8000000000000ll) : sb[e];
    if (out) out[(size_t)blockIdx.x * n * 3 + e] = (float)v;
    if (out64) out64[(size_t)blockIdx.x * n * 3 + e] = v;
  }
}

// Backward through the assembly: given g = A^-1 dtheta (fp64) and theta:
//   dtgt = g[:T];  dA = -g theta^T;  dctrl from K (pairwise) and from P.
__global__ __launch_bounds__(256) void tps_fit_bwd_kernel(const double* __restrict__ g64,
                                                          const float* __restrict__ theta,
                                                          const float* __restrict__ ctrl,
                                                          const float* __restrict__ lmbda,
                                                          const float* __restrict__ w,
                                                          float* __restrict__ dctrl, float* __restrict__ dtgt,
                                                          float* __restrict__ dwgt, int T) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // g[T][3], th[T][3], c[T][3] as floats
  const int b = blockIdx.y;
  const int n = T + 4;
  const double* g = g64 + (size_t)b * n * 3;
  const float* th = theta + (size_t)b * n * 3;
  const float* c = ctrl + (size_t)b * T * 3;
  float* sg = sh; float* st = sh + T * 3; float* sc = sh + 2 * T * 3;
  for (int e = threadIdx.x; e < T * 3; e += blockDim.x) { sg[e] = (float)g[e]; st[e] = th[e]; sc[e] = c[e]; }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T) return;
  const float ciz = sc[i * 3], ciy = sc[i * 3 + 1], cix = sc[i * 3 + 2];
  const float gi0 = sg[i * 3], gi1 = sg[i * 3 + 1], gi2 = sg[i * 3 + 2];
  const float ti0 = st[i * 3], ti1 = st[i * 3 + 1], ti2 = st[i * 3 + 2];
  double az = 0, ay = 0, ax = 0;
  for (int j = 0; j < T; ++j) {
    const float dz = ciz - sc[j * 3], dy = ciy - sc[j * 3 + 1], dx = cix - sc[j * 3 + 2];
    const float d2 = dz * dz + dy * dy + dx * dx + 1e-6f;
    const float r = __builtin_amdgcn_sqrtf(d2), re = r + 1e-6f;
    const float L = __builtin_amdgcn_logf(re) * 0.6931471805599453f;
    // dA_ij + dA_ji = -(g_i . th_j + g_j . th_i)
    const float s = -(gi0 * st[j * 3] + gi1 * st[j * 3 + 1] + gi2 * st[j * 3 + 2] +
                      sg[j * 3] * ti0 + sg[j * 3 + 1] * ti1 + sg[j * 3 + 2] * ti2);
    const float f = s * (2.f * L + r / re);  // * dU/d(d2) * 2
    az += (double)(f * dz); ay += (double)(f * dy); ax += (double)(f * dx);
  }
  // P blocks: A[i][T+1+k] = A[T+1+k][i] = c_ik  =>  dc_ik += dA[i][T+1+k] + dA[T+1+k][i]
  double pk[3];
  for (int k = 0; k < 3; ++k) {
    double a = 0;
    for (int d = 0; d < 3; ++d)
      a -= g[(size_t)i * 3 + d] * (double)th[(size_t)(T + 1 + k) * 3 + d] + g[(size_t)(T + 1 + k) * 3 + d] * (double)th[(size_t)i * 3 + d];
    pk[k] = a;
  }
  float* dc = dctrl + ((size_t)b * T + i) * 3;
  dc[0] = (float)(az + pk[0]); dc[1] = (float)(ay + pk[1]); dc[2] = (float)(ax + pk[2]);
  float* dt = dtgt + ((size_t)b * T + i) * 3;
  dt[0] = gi0; dt[1] = gi1; dt[2] = gi2;
  if (dwgt && w) {
    // A_ii = U(0) + lmbda / (w_i + 1e-6)  =>  dw_i = dA_ii * (-lmbda / (w_i + 1e-6)^2),  dA_ii = -g_i . th_i
    const double wi = (double)w[(size_t)b * T + i] + 1e-6;
    const double gt = (double)gi0 * ti0 + (double)gi1 * ti1 + (double)gi2 * ti2;
    dwgt[(size_t)b * T + i] = (float)((double)lmbda[b] * gt / (wi * wi));
  }
}

// ------------------------------------------------------------------------------------------
// center of mass on a materialised (N,K,D,H,W) heat-map
__global__ __launch_bounds__(TPB) void com3d_partial_kernel(const float* __restrict__ feat, int D, int H, int W,
                                                            double* __restrict__ partial /* (NK, nsplit, 4) */) {
  const int ch = blockIdx.y;
  const long long V = (long long)D * H * W;
  const float* p = feat + (long long)ch * V;
  const long long per = (V + gridDim.x - 1) / gridDim.x;
  const long long beg = per * blockIdx.x;
  long long end = beg + per;
  if (end > V) end = V;
  const float sz = D > 1 ? 1.f / (float)(D - 1) : 0.f, sy = H > 1 ? 1.f / (float)(H - 1) : 0.f,
              sx = W > 1 ? 1.f / (float)(W - 1) : 0.f;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  float f0 = 0, f1 = 0, f2 = 0, f3 = 0;
  int cnt = 0;
  for (long long v = beg + threadIdx.x; v < end; v += TPB) {
    const float val = fmaxf(p[v], 0.f);
    const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long long)W * H));
    f0 += val; f1 += val * (sz * (float)z); f2 += val * (sy * (float)y); f3 += val * (sx * (float)x);
    if (++cnt == 128) { a0 += f0; a1 += f1; a2 += f2; a3 += f3; f0 = f1 = f2 = f3 = 0.f; cnt = 0; }
  }
  a0 += f0; a1 += f1; a2 += f2; a3 += f3;
  __shared__ double red[TPB / kWave];
  a0 = block_sum<double>(a0, red); a1 = block_sum<double>(a1, red);
  a2 = block_sum<double>(a2, red); a3 = block_sum<double>(a3, red);
  if (threadIdx.x == 0) {
    double* o = partial + ((long long)ch * gridDim.x + blockIdx.x) * 4;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
}

__global__ void com3d_final_kernel(const double* __restrict__ partial, int nsplit, int NK,
                                   float* __restrict__ pts, float* __restrict__ sums) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= NK) return;
  double s[4] = {0, 0, 0, 0};
  for (int b = 0; b < nsplit; ++b)
    for (int k = 0; k < 4; ++k) s[k] += partial[((long long)ch * nsplit + b) * 4 + k];
  const double den = s[0] + 1e-8;
  for (int k = 0; k < 3; ++k) pts[ch * 3 + k] = (float)(s[1 + k] / den * 2.0 - 1.0);
  for (int k = 0; k < 4; ++k) sums[ch * 4 + k] = (float)s[k];
}

__global__ __launch_bounds__(TPB) void com3d_bwd_kernel(const float* __restrict__ dpts,
                                                        const float* __restrict__ feat,
                                                        const float* __restrict__ sums, float* __restrict__ dfeat,
                                                        int D, int H, int W) {
  const int ch = blockIdx.y;
  const long long V = (long long)D * H * W;
  const float den = sums[ch * 4] + 1e-8f;
  const float k2 = 2.f / den;
  const float gz = dpts[ch * 3] * k2, gy = dpts[ch * 3 + 1] * k2, gx = dpts[ch * 3 + 2] * k2;
  const float g0 = -(gz * (sums[ch * 4 + 1] / den) + gy * (sums[ch * 4 + 2] / den) + gx * (sums[ch * 4 + 3] / den));
  const float sz = D > 1 ? 1.f / (float)(D - 1) : 0.f, sy = H > 1 ? 1.f / (float)(H - 1) : 0.f,
              sx = W > 1 ? 1.f / (float)(W - 1) : 0.f;
  const float* p = feat + (long long)ch * V;
  float* o = dfeat + (long long)ch * V;
  for (long long v = (long long)blockIdx.x * TPB + threadIdx.x; v < V; v += (long long)gridDim.x * TPB) {
    const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long long)W * H));
    o[v] = p[v] > 0.f ? (g0 + gz * (sz * (float)z) + gy * (sy * (float)y) + gx * (sx * (float)x)) : 0.f;
  }
}

static inline int lda_of(int n) { return (n + 7) & ~7; }
static inline size_t solve_lds(int n) { return (size_t)n * 3 * sizeof(double) + (size_t)n * sizeof(int); }

struct FitWs {
  double* A; int* ipiv; int* info; double* g64; size_t a_stride; int lda; int* pivs; int* sync;
};
static FitWs carve(void* ws, int N, int T) {
  FitWs f;
  const int n = T + 4;
  f.lda = lda_of(n);
  f.a_stride = (size_t)n * f.lda;
  char* p = (char*)ws;
  f.A = (double*)p; p += (size_t)N * f.a_stride * sizeof(double);
  f.g64 = (double*)p; p += (size_t)N * n * 3 * sizeof(double);
  f.ipiv = (int*)p; p += (size_t)N * n * sizeof(int);
  f.info = (int*)p; p += (((size_t)N * sizeof(int)) + 15) & ~(size_t)15;
  f.pivs = (int*)p; p += (size_t)N * n * sizeof(int);
  f.sync = (int*)p;             // N x 4 ints: ready, done, abort, -
  return f;
}

}  // namespace

// ------------------------------------------------------------------------------------------
KMH_API int kmh_affine_fit_fwd(const float* x, const float* y, const float* w, float* M, int N, int K,
                               void* stream) {
  affine_fit_fwd_kernel<<<N, TPB, 0, (hipStream_t)stream>>>(x, y, w, M, K);
  return KMH_LAUNCH_CHECK();
}
KMH_API int kmh_affine_fit_bwd(const float* dM, const float* x, const float* y, const float* w, const float* M,
                               float* dx, float* dy, float* dw, int N, int K, void* stream) {
  (void)M;
  if (dw && !w) return -22;
  affine_fit_bwd_kernel<<<N, TPB, 0, (hipStream_t)stream>>>(dM, x, y, w, dx, dy, dw, K);
  return KMH_LAUNCH_CHECK();
}
KMH_API int kmh_rigid_fit_fwd(const float* x, const float* y, const float* w, float* M, int N, int K,
                              void* stream) {
  rigid_fit_fwd_kernel<<<N, TPB, 0, (hipStream_t)stream>>>(x, y, w, M, K);
  return KMH_LAUNCH_CHECK();
}
KMH_API int kmh_rigid_fit_bwd(const float* dM, const float* x, const float* y, const float* w, float* dx,
                              float* dy, float* dw, int N, int K, void* stream) {
  if (dw && !w) return -22;
  rigid_fit_bwd_kernel<<<N, TPB, 0, (hipStream_t)stream>>>(dM, x, y, w, dx, dy, dw, K);
  return KMH_LAUNCH_CHECK();
}
KMH_API int kmh_affine_inverse_fwd(const float* M, float* Minv, int N, void* stream) {
  affine_inverse_fwd_kernel<<<ceil_div(N, 64), 64, 0, (hipStream_t)stream>>>(M, Minv, N);
  return KMH_LAUNCH_CHECK();
}
KMH_API int kmh_affine_inverse_bwd(const float* dMinv, const float* Minv, float* dM, int N, void* stream) {
  affine_inverse_bwd_kernel<<<ceil_div(N, 64), 64, 0, (hipStream_t)stream>>>(dMinv, Minv, dM, N);
  return KMH_LAUNCH_CHECK();
}

KMH_API size_t kmh_tps_fit_ws_bytes(int N, int T) {
  const int n = T + 4;
  return (size_t)N * n * lda_of(n) * sizeof(double) + (size_t)N * n * 3 * sizeof(double) +
         (size_t)N * n * sizeof(int) * 2 + (size_t)N * sizeof(int) * 5 + 512;
}

// Compute units of the current device (one query per process and device; 0 on failure: the cluster kernel is then not used)
static int device_cus() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    v = cus > 0 ? cus : -1;
    cached[dev].store(v, std::memory_order_relaxed);
  }
  return v > 0 ? v : 0;
}

// *used_cluster = 1 when the cluster kernel was launched (the caller then queues the retry pass behind it)
template <int NB>
static int launch_lu(const FitWs& f, int N, int n, hipStream_t s, int* used_cluster) {
  *used_cluster = 0;
  const size_t lds = ((size_t)n * (NB + 1) + (size_t)NB * n) * sizeof(double) + (size_t)n * sizeof(int);   // + rowmap
  if (lds > 160 * 1024 - 1024) return -22;
  static const bool valu_trailing = getenv("KEYMORPH_TPS_LU_VALU") != nullptr;      // A/B measurements only
  if (valu_trailing) {
    hipError_t e = hipFuncSetAttribute((const void*)tps_lu_kernel<NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    tps_lu_kernel<NB, false><<<N, LU_TPB, lds, s>>>(f.A, f.ipiv, f.info, n, f.lda, f.a_stride);
    return KMH_LAUNCH_CHECK();
  }
  // cluster of G workgroups per system when the systems are large enough to pay for the hand-offs and every workgroup of the
  // launch can be resident at once (one per CU: 136 KB of LDS): KEYMORPH_TPS_LU_CLUSTER=<G> (0 / 1 = one workgroup per system).
  // The workgroups synchronise through global counters inside an ORDINARY launch, so residency is a matter of the grid size:
  // the grid stays within HALF the device's compute units (queried, not assumed) and the occupancy query must admit the kernel
  // at all.  That leaves room for a neighbour on another stream, but proves nothing about CUs held by other processes or a CU
  // mask -- which is why every wait in the kernel is bounded and a fit that gave up (info = 2) is REDONE on the one-workgroup
  // kernel by the retry pass queued behind it (kmh_tps_fit_fwd): the caller never sees a poisoned result.
  static const int genv = getenv("KEYMORPH_TPS_LU_CLUSTER") ? atoi(getenv("KEYMORPH_TPS_LU_CLUSTER")) : 8;
  const int cus = device_cus();
  int G = genv;
  while (G > 1 && (long long)N * G * 2 > cus) G >>= 1;
  if (G > 1 && n >= 128) {
    hipError_t e = hipFuncSetAttribute((const void*)tps_lu_cluster_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)tps_lu_cluster_kernel<NB>, LU_TPB, lds);
    if (e != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    if (per_cu >= 1) {
      e = hipMemsetAsync(f.sync, 0, (size_t)N * 4 * sizeof(int), s);
      if (e != hipSuccess) return (int)e;
      tps_lu_cluster_kernel<NB><<<dim3(G, N), LU_TPB, lds, s>>>(f.A, f.ipiv, f.info, n, f.lda, f.a_stride, f.pivs, f.sync);
      *used_cluster = 1;
      return KMH_LAUNCH_CHECK();
    }
  }
  hipError_t e = hipFuncSetAttribute((const void*)tps_lu_kernel<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  tps_lu_kernel<NB, true><<<N, LU_TPB, lds, s>>>(f.A, f.ipiv, f.info, n, f.lda, f.a_stride);
  return KMH_LAUNCH_CHECK();
}

// the systems a cluster factorisation gave up on (info = 2), once more on the one-workgroup kernel: both launches return at
// once for every other system (a few microseconds), no host round trip
template <int NB>
static int launch_lu_retry(const FitWs& f, const float* ctrl, const float* lmbda, const float* w, int N, int T,
                           hipStream_t s) {
  const int n = T + 4;
  const size_t lds = ((size_t)n * (NB + 1) + (size_t)NB * n) * sizeof(double) + (size_t)n * sizeof(int);
  hipError_t e = hipFuncSetAttribute((const void*)tps_lu_kernel<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  tps_assemble_kernel<<<dim3(ceil_div(n, 16), ceil_div(n, 16), N), 256, 0, s>>>(ctrl, lmbda, w, f.A, T, f.lda, f.a_stride,
                                                                             f.info);
  tps_lu_kernel<NB, true><<<N, LU_TPB, lds, s>>>(f.A, f.ipiv, f.info, n, f.lda, f.a_stride, 1);
  return KMH_LAUNCH_CHECK();
}

/* Test hook: the next `count` calls of kmh_tps_fit_fwd that use the cluster factorisation mark every system as "gave up"
 * (info = 2) before the retry pass, which must then reproduce the one-workgroup result.  Returns the previous count. */
static std::atomic<int> g_force_retry{0};
KMH_API int kmh_tps_fit_force_retry(int count) { return g_force_retry.exchange(count < 0 ? 0 : count); }

namespace {
__global__ void tps_mark_retry_kernel(int* __restrict__ info, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) info[i] = 2;
}
}  // namespace

KMH_API int kmh_tps_fit_fwd(const float* ctrl, const float* tgt, const float* lmbda, const float* w,
                            float* theta, int N, int T, void* ws, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int n = T + 4;
  FitWs f = carve(ws, N, T);
  tps_assemble_kernel<<<dim3(ceil_div(n, 16), ceil_div(n, 16), N), 256, 0, s>>>(ctrl, lmbda, w, f.A, T, f.lda,
                                                                             f.a_stride);
  int rc, clustered = 0;
  if (n <= 600) rc = launch_lu<16>(f, N, n, s, &clustered);
  else if (n <= 1150) rc = launch_lu<8>(f, N, n, s, &clustered);
  else return -22;
  if (rc) return rc;
  if (clustered) {
    int forced = g_force_retry.load(std::memory_order_relaxed);
    if (forced > 0 && g_force_retry.compare_exchange_strong(forced, forced - 1))
      tps_mark_retry_kernel<<<ceil_div(N, 64), 64, 0, s>>>(f.info, N);
    rc = n <= 600 ? launch_lu_retry<16>(f, ctrl, lmbda, w, N, T, s) : launch_lu_retry<8>(f, ctrl, lmbda, w, N, T, s);
    if (rc) return rc;
  }
  tps_solve_kernel<<<N, LU_TPB, solve_lds(n), s>>>(f.A, f.ipiv, tgt, T, theta, nullptr, n, f.lda, f.a_stride, f.info);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_tps_fit_bwd(const float* dtheta, const float* theta, const float* ctrl, const float* lmbda,
                            const float* w, float* dctrl, float* dtgt, float* dw, int N, int T, void* ws,
                            void* stream) {
  if (dw && !w) return -22;
  hipStream_t s = (hipStream_t)stream;
  const int n = T + 4;
  FitWs f = carve(ws, N, T);
  // A is symmetric, so A^T g = dtheta is the system the forward already factorised.
  tps_solve_kernel<<<N, LU_TPB, solve_lds(n), s>>>(f.A, f.ipiv, dtheta, n, nullptr, f.g64, n, f.lda, f.a_stride);
  const size_t lds = (size_t)T * 9 * sizeof(float);
  if (lds > 64 * 1024) return -22;
  tps_fit_bwd_kernel<<<dim3(ceil_div(T, 256), N), 256, lds, s>>>(f.g64, theta, ctrl, lmbda, w, dctrl, dtgt, dw, T);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_com3d_fwd(const float* feat, float* pts, float* sums, int N, int K, int D, int H, int W, void* ws,
                          void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long long V = (long long)D * H * W;
  const int NK = N * K;
  int nsplit = ceil_div(V, (long long)TPB * 64);
  int cap = 65536 * 3 / 4 / (NK > 0 ? NK : 1);
  if (cap < 1) return -22;
  if (nsplit > cap) nsplit = cap;
  if (nsplit > 256) nsplit = 256;
  if (nsplit < 1) nsplit = 1;
  com3d_partial_kernel<<<dim3(nsplit, NK), TPB, 0, s>>>(feat, D, H, W, (double*)ws);
  com3d_final_kernel<<<ceil_div(NK, 64), 64, 0, s>>>((const double*)ws, nsplit, NK, pts, sums);
  return KMH_LAUNCH_CHECK();
}

KMH_API int kmh_com3d_bwd(const float* dpts, const float* feat, const float* sums, float* dfeat, int N, int K,
                          int D, int H, int W, void* stream) {
  const long long V = (long long)D * H * W;
  int nb = ceil_div(V, (long long)TPB * 4);
  if (nb > 1024) nb = 1024;
  com3d_bwd_kernel<<<dim3(nb, N * K), TPB, 0, (hipStream_t)stream>>>(dpts, feat, sums, dfeat, D, H, W);
  return KMH_LAUNCH_CHECK();
}
