// Shared device helpers for the keymorph_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KMH_API extern "C" __attribute__((visibility("default")))

// Every launcher returns the hipError_t of the launch (0 = ok).
#define KMH_LAUNCH_CHECK() ((int)hipGetLastError())

constexpr int kWave = 64;  // CDNA wavefront width

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}

// Block-wide sum; result valid in thread 0.  `scratch` needs blockDim.x/64 slots.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  T r = T(0);
  if (wid == 0) {
    r = lane < nw ? scratch[lane] : T(0);
    r = wave_sum(r);
  }
  return r;
}

// TPS radial basis U(r) = r^2 log(r + 1e-6), r = sqrt(d2 + 1e-6)  (keymorph/keypoint_aligners.py:322-339).
// ONE definition for the system assembly (fits.hip) and for every evaluation (grids.hip): with lambda = 0 and
// hundreds of clustered keypoints the spline weights reach 1e3..1e4, and a 1-ulp mismatch between the U used
// to fit and the U used to evaluate shows up as 1e-3 interpolation error at the control points.
// v_log_f32 is a 1-ulp hardware approximation, i.e. the same accuracy class as libm's.
// squared distance as ONE explicit fma chain, identical (IEEE fma per component) in scalar and packed form
// tps_d2 INCLUDES the reference's + 1e-6 under the square root (one fma chain seeded with it)
__device__ __forceinline__ float tps_d2(float dz, float dy, float dx) { return fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, 1e-6f))); }

// log(r + e), r = sqrt(d2), e = 1e-6, with ONE transcendental instead of two (v_sqrt/v_rsq and v_log are quarter-rate and
// were 16 of the 24 issue slots per pair of kernel values):  2 log2(r + e) = log2(d2) + 2 log2(1 + e/r), and e/r <= 1e-3
// (d2 >= 1e-6), so log2(1 + e/r) = (e/r) / ln 2 to 5e-4 of ITSELF.  e/r comes from the integer reciprocal-square-root
// estimate (two integer instructions, +-3.4 % of e/r, centred by the 0.98636): the whole correction is <= 2e-6/r in
// natural-log units and its error <= 3.5e-8/r -- under the rounding of r + 1e-6 in fp32 (6e-8 relative) that the
// reference's own operation sequence carries, and the fp64 value of log(sqrt(d2) + 1e-6) is what the tests compare with.
__device__ __forceinline__ float tps_rsq_est(float d2) { return __uint_as_float(0x5f3759dfu - (__float_as_uint(d2) >> 1)); }
constexpr float kTpsRsqCentre = 0.98636f;                                    // zero-mean estimate
constexpr float kTpsEpsLog2x2 = 2.0e-6f * 0.98636f / 0.6931471805599453f;    // 2 log2(1 + e/r) = this * estimate
constexpr float kTpsHalfLn2 = 0.5f * 0.6931471805599453f;
// 2 log2(r + e)
__device__ __forceinline__ float tps_log2x2(float d2) { return fmaf(tps_rsq_est(d2), kTpsEpsLog2x2, __builtin_amdgcn_logf(d2)); }
__device__ __forceinline__ float tps_u_from_d2(float d2) { return (d2 * tps_log2x2(d2)) * kTpsHalfLn2; }

// two-lane version for the packed-fp32 evaluators (v_pk_add/mul/fma_f32): the SAME operation sequence per component
typedef float kmh_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ kmh_f2 tps_d2(kmh_f2 dz, kmh_f2 dy, kmh_f2 dx) {
  const kmh_f2 eps = {1e-6f, 1e-6f};
  return __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, eps)));
}
// (scalars first: hipcc 7.2 mis-reads element 0 when a bit cast is applied to a vector element directly)
__device__ __forceinline__ kmh_f2 tps_rsq_est2(kmh_f2 d2) {
  const float a = d2.x, b = d2.y;
  return kmh_f2{tps_rsq_est(a), tps_rsq_est(b)};
}
__device__ __forceinline__ kmh_f2 tps_log2x2(kmh_f2 d2, kmh_f2 y /* tps_rsq_est2(d2) */) {
  const float a = d2.x, b = d2.y;
  const kmh_f2 lg = {__builtin_amdgcn_logf(a), __builtin_amdgcn_logf(b)};
  return __builtin_elementwise_fma(y, kmh_f2{kTpsEpsLog2x2, kTpsEpsLog2x2}, lg);
}
// 2 U / ln 2 = d2 * 2 log2(r + 1e-6): the evaluators fold ln 2 / 2 into the (per-keypoint) weights instead of every value
__device__ __forceinline__ kmh_f2 tps_u2_from_d2(kmh_f2 d2) { return d2 * tps_log2x2(d2, tps_rsq_est2(d2)); }

// XCD-aware work remap (MI355X: 8 XCDs with private 4 MB L2s; the dispatcher places block b on XCD b % 8 --
// observed behaviour used for SPEED only, any placement is correct).  Returns the linear work item for
// hardware block id `b` such that each XCD walks one CONTIGUOUS range of work items: neighbouring bricks
// (shared halos) and the channel groups of one brick then hit the same L2 instead of 8 different ones.
// Bijective for any total (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int b, int total) {
  constexpr int NX = 8;
  const int q = total / NX, r = total % NX;
  const int xcd = b % NX, idx = b / NX;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// ---------------------------------------------------------------------------------------------
// Split-operand arithmetic shared by csrc/conv_bf.hip and csrc/headcom.hip.
// TERMS == 3: bf16 hi+mid+lo (24 bits), 6 products ("bf16x6").  TERMS == 2: FP16 hi+lo (22 bits: a 2^-23
// representation error, the size of fp32's own rounding), 3 products ("f16x3") -- half the MFMA work.  fp16's narrow
// exponent makes that accurate only if every operand tensor is first scaled by a power of two (exact) so that its
// largest magnitude sits just under 2^15: elements then keep 22 bits down to ~2^-18 of the maximum and lose only
// absolute accuracy below that (<= 2^-25 of a scaled unit).  Scales live in device memory as {S, 1/S} pairs;
// epilogues multiply by 1/(S_A S_B), also exact.
typedef float kmh_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 kmh_bf16x8 __attribute__((ext_vector_type(8)));     // also the raw 8 x 16-bit container of fp16 data
typedef _Float16 kmh_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short kmh_u16x8 __attribute__((ext_vector_type(8)));

template <int TERMS>
__device__ __forceinline__ unsigned short to16(float r, float& back) {
  if constexpr (TERMS == 2) {
    const _Float16 h = (_Float16)r;
    back = (float)h;
    return __builtin_bit_cast(unsigned short, h);
  } else {
    const __bf16 h = (__bf16)r;
    back = (float)h;
    return __builtin_bit_cast(unsigned short, h);
  }
}
template <int TERMS>
__device__ __forceinline__ kmh_f32x16 mfma16(kmh_bf16x8 a, kmh_bf16x8 b, kmh_f32x16 c) {
  if constexpr (TERMS == 2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(kmh_f16x8, a), __builtin_bit_cast(kmh_f16x8, b), c, 0,
                                                  0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// use_amp (keymorph/model.py:176-191 runs the backbone under fp16 autocast): the ONE-product arithmetic of the split-operand
// kernels -- operands range-scaled and split exactly as for f16x3, only hi x hi multiplied: fp16 inputs (11 significant bits),
// fp32 accumulation, a third of the MFMA work.  PER CALL, no process state: an entry point that takes `terms` accepts
// terms == 1 = "the fp16 kernels (terms 2), hi x hi only"; it opens a KmhAmpCall, which turns the 1 into 2 for everything below
// and makes kmh_amp_enabled() true on THIS thread until the entry point returns (launchers pick the AMP = true instances).
bool kmh_amp_enabled();
bool kmh_amp_call_begin(int* terms);          // returns the previous per-thread state
void kmh_amp_call_end(bool prev);
struct KmhAmpCall {
  bool prev;
  explicit KmhAmpCall(int& terms) : prev(kmh_amp_call_begin(&terms)) {}
  ~KmhAmpCall() { kmh_amp_call_end(prev); }
  KmhAmpCall(const KmhAmpCall&) = delete;
  KmhAmpCall& operator=(const KmhAmpCall&) = delete;
};
// 8 floats -> TERMS fragments (8 x 16 bit each)
// two values -> TERMS packed 16-bit pairs (lo half = first value): one packed conversion per term
template <int TERMS>
__device__ __forceinline__ void split_pair(float r0, float r1, unsigned out[TERMS]) {
  if constexpr (TERMS == 2) {
    // hi = (f16(r0), f16(r1)) in ONE v_cvt_pk_f16_f32; the residual r - f32(hi) as ONE v_fma_mix_f32 per value (its
    // operand selector converts the fp16 half of `hi`: op_sel_hi[0] = 1 says source 0 is fp16, op_sel[0] picks the high
    // half), exact in fp32; lo = ONE more packed conversion.  Four VALU of the 1.8 ns class per pair
    // (tools/ubench/valu_rates.hip; v_fma_mix{lo,hi}_f16, which would fold the last conversion, cost 3.4 ns each).
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t f = {r0, r1};
    const h2_t h = __builtin_convertvector(f, h2_t);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    out[0] = hb;
    float e0, e1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(e0) : "v"(hb), "v"(r0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(e1) : "v"(hb), "v"(r1));
    const f2_t g = {e0, e1};
    out[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(g, h2_t));
  } else {
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
      float b0, b1;
      const unsigned h0 = to16<TERMS>(r0, b0), h1 = to16<TERMS>(r1, b1);
      out[t] = h0 | (h1 << 16);
      r0 -= b0; r1 -= b1;
    }
  }
}

template <int TERMS>
__device__ __forceinline__ void split8(const float v[8], kmh_bf16x8 out[TERMS]) {
  if constexpr (TERMS == 2) {
    // fp16 hi + lo through packed conversions (v_cvt_pk_f16_f32): 4 + 4 conversions instead of 8 + 8
    unsigned w[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair<2>(v[2 * j], v[2 * j + 1], w[j]);
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const u4_t bits = {w[0][t], w[1][t], w[2][t], w[3][t]};
      out[t] = __builtin_bit_cast(kmh_bf16x8, bits);
    }
  } else {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
      kmh_u16x8 bits;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float back;
        bits[j] = to16<TERMS>(r[j], back);
        r[j] -= back;
      }
      out[t] = __builtin_bit_cast(kmh_bf16x8, bits);
    }
  }
}

// {S, 1/S} with S the power of two that puts `bound` in (2^14, 2^15]: fp16's largest finite value is 65504, and
// fp16 x fp16 products are exact in the fp32 MFMA accumulator
__device__ __forceinline__ void range_scale(float bound, float* out2) {
  float S = 1.f;
  if (bound > 0.f && bound < 3.0e38f) {        // finite, non-zero (NaN compares false)
    int e;
    frexpf(bound, &e);                         // bound = f * 2^e, f in [0.5, 1)  =>  bound <= 2^e
    int k = 15 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    S = ldexpf(1.f, k);
  }
  out2[0] = S;
  out2[1] = 1.f / S;
}

namespace kmh_absmax {
// wave maximum -> one atomic per wave, and only when it would raise the published value: thousands of same-address
// atomics serialise in L2 (~2.5 ns each), a plain load of the current maximum does not
__device__ __forceinline__ void publish(float m, unsigned* acc) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) {
    const unsigned bits = __float_as_uint(m);
    if (bits > __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(acc, bits);
  }
}
__global__ __launch_bounds__(256) static void partial_kernel(const float* __restrict__ x, long long n,
                                                             unsigned* __restrict__ acc) {
  float m = 0.f;
  // scalar head up to the first 16-byte boundary (parameters inside a flat bucket are only 4-byte aligned),
  // float4 body, scalar tail
  long long head = (long long)(((16 - (reinterpret_cast<unsigned long long>(x) & 15)) & 15) >> 2);
  if (head > n) head = n;
  const float* xb = x + head;
  const long long nb = n - head, n4 = nb >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(xb)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0) {
    for (long long i = threadIdx.x; i < head; i += 256) m = fmaxf(m, fabsf(x[i]));
    for (long long i = (n4 << 2) + threadIdx.x; i < nb; i += 256) m = fmaxf(m, fabsf(xb[i]));
  }
  // non-negative floats order like their bit patterns: an integer max is exact and order independent
  publish(m, acc);
}
__global__ static void final_kernel(float* __restrict__ out2, float min_abs) {
  const float m = fmaxf(__uint_as_float(reinterpret_cast<unsigned*>(out2)[0]), min_abs);
  range_scale(m, out2);
}
// out2[2] = {S, 1/S} for max(max|x|, min_abs); everything on `s`, no host sync
static inline int launch(const float* x, long long n, float min_abs, float* out2, hipStream_t s) {
  hipError_t e = hipMemsetAsync(out2, 0, 2 * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  long long nb = (n / 4 + 256) / 256;
  if (nb > 2048) nb = 2048;
  partial_kernel<<<(int)nb, 256, 0, s>>>(x, n, reinterpret_cast<unsigned*>(out2));
  final_kernel<<<1, 1, 0, s>>>(out2, min_abs);
  return (int)hipGetLastError();
}
}  // namespace kmh_absmax

namespace kmh_stats {
// partial (N, nblk, C, 2) doubles -> out (N, C, 2): one wave per output element, lanes stride over the partial
// blocks (4 independent loads in flight each), then a fixed-order wave reduction -- deterministic, and a ~64x
// shorter dependency chain than one thread.  Launch: grid (ceil(2C / 4), N), 256 threads.
__global__ __launch_bounds__(256) static void final_kernel(const double* __restrict__ partial, int nblk, int C,
                                                           double* __restrict__ out,
                                                           const int* __restrict__ only_if = nullptr) {
  if (only_if && *only_if == 0) return;        // device-side gate of a fallback path (no host synchronisation)
  const int n = blockIdx.y;
  const int e = blockIdx.x * (256 / kWave) + (threadIdx.x >> 6);
  if (e >= C * 2) return;
  const int lane = threadIdx.x & 63;
  const double* p = partial + (long long)n * nblk * C * 2 + e;
  double s4[4] = {0, 0, 0, 0};
  int b = lane;
  for (; b + 3 * kWave < nblk; b += 4 * kWave) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s4[k] += p[(long long)(b + k * kWave) * C * 2];
  }
  for (; b < nblk; b += kWave) s4[0] += p[(long long)b * C * 2];
  const double s = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
  if (lane == 0) out[(long long)n * C * 2 + e] = s;
}
}  // namespace kmh_stats

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
