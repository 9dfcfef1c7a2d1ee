// fp32-accurate 3x3x3 convolution on the 16-bit matrix cores: every fp32 operand is split exactly into 16-bit terms
// and each product is accumulated in fp32 from the significant term products --
//   TERMS = 2 ("f16x3", the default): operands range-scaled by a power of two, x S = hi + lo in fp16 (22 bits),
//              products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, exact descale in the epilogue;
//   TERMS = 3 ("bf16x6"): x = hi + mid + lo in bf16 (24 bits), six products on v_mfma_f32_32x32x16_bf16.
// Both are fp32-class (5e-7 against fp64, like the fp32 MFMA); on CDNA4 the 16-bit MFMA rate is 16x the fp32 MFMA rate,
// so this is 16/3 = 5.3x (16/6 = 2.7x) the fp32-MFMA roofline -- the 3xTF32 / BF16x9 idea on gfx950's 32x32x16 tile.
//
// Kernels in this file:
//   conv3_fwd_bf_kernel<NT, TERMS, MR, ZP, ZT>   forward and data gradient (same kernel on tap-mirrored weights);
//                                                 NT = 32-wide cout tiles per wave, MR = rows per wave, ZP = z-paired N
//                                                 tile for Cout <= 16, ZT = output-plane pairs per brick
//   conv3_wgrad_ws_kernel<NT, TERMS, MASK, PW>   weight gradient, producer / consumer waves (the default)
//   conv3_wgrad_bf_kernel<NT, TERMS>             weight gradient, single-role (bf16x6, odd channel counts, huge volumes)
//   pack_weight_bf_kernel, wgrad_bf_reduce_kernel, first_layer_fold_kernel
//
// Forward: same brick / wave decomposition as conv.hip (32x8x2 output voxels per 4-wave workgroup, 4 rows x NT
// channel tiles per wave), but:
//   * the halo brick is split while it is staged (GroupNorm scale/shift, ReLU, fused ReLU-backward mask first, then
//     packed 16-bit conversions) into TERMS voxel-major LDS images of 16-byte rows (8 channels), so an A fragment is
//     ONE conflict-free ds_read_b128 per term: 32 consecutive voxels x 8 channels;
//   * K = 16 of an MFMA = 2 taps x 8 channels: lanes 0-31 carry tap 2s, lanes 32-63 tap 2s+1 (27 taps =
//     13 pairs + one half-empty step whose weights are zero);
//   * the filter is pre-packed as [cin/8][term][step][half][cout][8] 16-bit values, so a B fragment is one
//     512-byte-per-half-wave global_load_dwordx4 from L2, prefetched through a register ring;
//   * the epilogue can emit the (sum, sum of squares) per channel that the next GroupNorm needs.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include <type_traits>

namespace {

typedef kmh_f32x16 f32x16;
typedef kmh_bf16x8 bf16x8;     // 8 x 16-bit fragment (bf16 or fp16 bits; see common.h for the split arithmetic)
}  // namespace (the typedefs every part of this file uses)

// Two translation units share this file: csrc/conv_wgrad.hip defines KMH_TU_WGRAD and compiles ONLY the weight-gradient section
// (with LLVM's max-ILP scheduling strategy, keymorph_amd/build.py: its wave-specialised kernels run 2-4 % faster with it),
// this file compiles everything else with the default strategy -- under max-ILP hipcc spills an in-flight destination of the
// inline-asm loads of conv3_fwd_g_kernel<2> right behind its load (found by tools/scan_asm_inflight.py).
#ifndef KMH_TU_WGRAD
#define KMH_TU_WGRAD 0
#endif
#if !KMH_TU_WGRAD
namespace {

constexpr int TX = 32, TZ = 2;
constexpr int HX = TX + 2, HZ = TZ + 2;
constexpr int KC = 8;              // channels per LDS refill (= half of the MFMA K)
// MR = output rows (M-tiles) per wave: brick height TY = 2*MR.  MR = 4: 32x8x2 brick, 1360-voxel halo;
// MR = 2: 32x4x2 brick, 816-voxel halo -> 39 KB of LDS (TERMS = 3) and 64 accumulator registers, i.e. 3-4
// resident workgroups per CU instead of 2.
constexpr int NSTEP = 14;          // tap pairs
constexpr int BF_TPB = 256;

// torch (Cout, Cin, 27) -> [nchunk][TERMS][nstep][2][CoutP][8] bf16 (zero padded); transposed = data gradient.
// zpair (logical Cout <= 16): the 32 columns are (co, pz) = (j & 15, j >> 4) -- the SAME 16 output channels for the
// two output planes z0, z0+1 of a brick, which share the 4-plane input window; taps run over kz' in 0..3 (36 taps,
// nstep = 18) and column (co, pz) carries w[kz' - pz] where that is a valid tap, 0 elsewhere.
constexpr int NSTEP_Z = 18;
template <int TERMS>
__global__ __launch_bounds__(256) void pack_weight_bf_kernel(const float* __restrict__ w, __bf16* __restrict__ out,
                                                             int Cout, int Cin, int CoutP, int nchunk,
                                                             int transposed, int zpair,
                                                             const float* __restrict__ wscale /* {S, 1/S} | NULL */) {
  // logical filter L[co][ci][tap] with (Co, Ci) = transposed ? (Cin, Cout) : (Cout, Cin)
  const int Co = transposed ? Cin : Cout, Ci = transposed ? Cout : Cin;
  const int nstep = zpair ? NSTEP_Z : NSTEP;
  const long long total = (long long)nchunk * nstep * 2 * CoutP * 8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e & 7);
    long long r = e >> 3;
    const int col = (int)(r % CoutP); r /= CoutP;
    const int h = (int)(r & 1); r >>= 1;
    const int s = (int)(r % nstep);
    const int chunk = (int)(r / nstep);
    const int ci = chunk * 8 + c;
    int tap = 2 * s + h, co = col;
    bool ok = tap < 27;
    if (zpair) {
      const int kz = tap / 9 - (col >> 4);          // tap = kz' * 9 + ky * 3 + kx
      co = col & 15;
      ok = (col < 32) && kz >= 0 && kz <= 2;
      tap = kz * 9 + tap % 9;
    }
    float v = 0.f;
    if (ok && ci < Ci && co < Co)
      v = transposed ? w[((long long)ci * Cin + co) * 27 + (26 - tap)] : w[((long long)co * Cin + ci) * 27 + tap];
    float rem = wscale ? v * wscale[0] : v;
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
      float back;
      const unsigned short hb = to16<TERMS>(rem, back);
      reinterpret_cast<unsigned short*>(out)[((((long long)chunk * TERMS + t) * nstep + s) * 2 + h) * CoutP * 8 +
                                             (long long)col * 8 + c] = hb;
      rem -= back;
    }
  }
}

// ZT = output-plane pairs per brick: 1 -> 32 x TY x 2 bricks (4 input planes per 2 output planes), 2 -> 32 x TY x 4
// bricks (6 per 4: the halo re-read factor drops from 2.66 to 1.99 and every B fragment feeds twice the MFMAs) for
// the low-channel layers of the full-resolution level, which are bound by input traffic, not by the matrix pipe.
template <int NT, int TERMS, int MR, bool ZP = false, int ZT = 1>
__global__ __launch_bounds__(BF_TPB, ((MR == 2 && NT == 2 && TERMS == 3) ? 3 : 2)) void conv3_fwd_bf_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mask, const bf16x8* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, int D, int H, int W, int Cin, int Cout, int CoutP, int relu_in, int relu_out,
    int tiles_x, int tiles_y, int tiles_z, int tiles_zp, const float* __restrict__ ascale /* {S, 1/S} of the input | NULL */,
    const float* __restrict__ wscale /* of the packed weights | NULL */,
    double* __restrict__ stats_partial /* (N, bricks, Cout, 2) per-brick (sum y, sum y^2) | NULL */,
    int in_blocked /* x is (N, Cin/8, D, H, W, 8): a chunk's voxels are contiguous 32-byte records */,
    const float* __restrict__ addend /* like y | NULL: added before the activation (not for the z-paired variant) */) {
  // ZP: the 4 waves split the brick's y rows (MR each) and every wave produces BOTH z planes in its N tile
  constexpr int TZv = 2 * ZT, HZv = TZv + 2;
  constexpr int TY = (ZP ? 4 : 2) * MR, HY = TY + 2, PL = HX * HY * HZv;
  constexpr int NST = ZP ? NSTEP_Z : NSTEP;
  static_assert(!ZP || NT == 1, "z-paired tiles are for Cout <= 16");
  __shared__ bf16x8 sIn[TERMS][PL];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n = blockIdx.z;
  // work item = (brick, cout group) with the cout group fastest, XCD-remapped (common.h)
  const int ncog = ZP ? 1 : (Cout + 32 * NT - 1) / (32 * NT);
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int cog = item % ncog, brick = item / ncog;
  // Brick order: the ~64 bricks an XCD has in flight form an 8 x 8 patch in (y, z) -- the directions in which
  // neighbouring halos overlap most (2 of 4 planes in z, 2 of 10 rows in y, only 2 of 34 columns in x) -- so the
  // shared planes are fetched into that XCD's L2 once instead of once per brick.  Patches are padded to 8 x 8;
  // workgroups of the padding exit here (before any barrier).
  const int tyz = (tiles_y + 7) >> 3;
  const int lz8 = brick & 7, ly8 = (brick >> 3) & 7, patch = brick >> 6;
  const int pyi = patch % tyz, rest = patch / tyz;
  const int pzi = rest % tiles_zp, bx = rest / tiles_zp;
  const int by = pyi * 8 + ly8, bz = pzi * 8 + lz8;
  if (by >= tiles_y || bz >= tiles_z) return;
  const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZv;
  const int co0 = cog * (32 * NT);
  const int wz = ZP ? 0 : wv >> 1, wy = (ZP ? wv : (wv & 1)) * MR;

  f32x16 acc[ZT][MR][NT];
#pragma unroll
  for (int p = 0; p < ZT; ++p)
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][m][t][r] = 0.f;

  const float sA = ascale ? ascale[0] : 1.f;                                   // power of two: folding it into the
  const float desc = (ascale ? ascale[1] : 1.f) * (wscale ? wscale[1] : 1.f);   // coefficients and the epilogue is exact
  const bool vec4 = (Cin & 3) == 0;
  const int nchunk = (Cin + KC - 1) / KC;
  const int vrow = (wz * HY + wy) * HX + li;    // this lane's voxel in the wave's first row, tap (0,0,0)

  // staging descriptors: this thread's (up to 6) halo voxels are the same for every channel chunk
  constexpr int NV = (PL + BF_TPB - 1) / BF_TPB;      // 6
  int sv_rel[NV];                                     // element offset of the voxel relative to the brick origin
  bool sv_in[NV];                                     // inside the volume?
  const long long origin = ((((long long)n * D + z0) * H + y0) * W + x0) * Cin;   // may address the halo "before" it
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * BF_TPB;
    const int lx = v % HX, ly = (v / HX) % HY, lz = v / (HX * HY);
    const int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = z0 + lz - 1;
    sv_in[i] = (v < PL) && ((unsigned)gx < (unsigned)W) && ((unsigned)gy < (unsigned)H) && ((unsigned)gz < (unsigned)D);
    sv_rel[i] = (((lz - 1) * H + (ly - 1)) * W + (lx - 1)) * (in_blocked ? KC : Cin);
  }
  // channel-blocked input: chunk ch of sample n starts at ((n * nchunk + ch) * D*H*W) * 8 floats
  const long long chunk_stride = in_blocked ? (long long)D * H * W * KC : KC;
  const float* xb = in_blocked ? x + ((long long)n * nchunk * D * H * W + (((long long)z0 * H + y0) * W + x0)) * KC : x + origin;
  const float* mb = mask ? mask + origin : nullptr;
  // per-lane B offset (in bf16x8 units) of step 0; step s adds 2*CoutP, term q adds NSTEP*2*CoutP
  const int boff = lh * CoutP + co0 + li;

  // staging = fetch (global -> registers, raw) + commit (mask, normalise, ReLU, zero padding, split, LDS).
  // PF (z-paired variant: 32 accumulator registers, short MFMA phase per chunk): the fetch of chunk ch+1 is in
  // flight during the MFMAs of chunk ch; otherwise fetch and commit run back to back, one voxel at a time.
  constexpr bool PF = false;   // measured: no gain for the z-paired variant (its exposed latency is the B loads)
  constexpr int NPRE = PF ? NV : 1;
  float pv[NPRE][8], pm[NPRE][8];
  auto fetch_one = [&](int ch, int i, float (&v8)[8], float (&m8)[8]) {
    const int c0 = ch * KC;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v8[j] = 0.f; m8[j] = 1.f; }
    if (sv_in[i]) {
      const float* p = xb + sv_rel[i] + ch * chunk_stride;
      if (vec4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (c0 + 4 * q < Cin) {
            const float4 t4 = *reinterpret_cast<const float4*>(p + 4 * q);
            v8[4 * q] = t4.x; v8[4 * q + 1] = t4.y; v8[4 * q + 2] = t4.z; v8[4 * q + 3] = t4.w;
            if (mb) {
              const float4 m4 = *reinterpret_cast<const float4*>(mb + sv_rel[i] + c0 + 4 * q);
              m8[4 * q] = m4.x; m8[4 * q + 1] = m4.y; m8[4 * q + 2] = m4.z; m8[4 * q + 3] = m4.w;
            }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (c0 + j < Cin) {
            v8[j] = p[j];
            if (mb) m8[j] = mb[sv_rel[i] + c0 + j];
          }
      }
    }
  };
  auto commit_one = [&](int ch, int i, const float (&v8)[8], const float (&m8)[8], const float (&csc)[8],
                        const float (&csh)[8]) {
    const int c0 = ch * KC;
    const int v = tid + i * BF_TPB;
    if (v < PL) {
      float val[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = (m8[j] > 0.f) ? v8[j] : 0.f;              // fused ReLU-backward mask
        t = t * csc[j] + csh[j];                            // identity when scale == NULL
        if (relu_in) t = fmaxf(t, 0.f);
        val[j] = (sv_in[i] && c0 + j < Cin) ? t : 0.f;      // zero padding AFTER the normalisation
      }
      bf16x8 parts[TERMS];
      split8<TERMS>(val, parts);
#pragma unroll
      for (int t = 0; t < TERMS; ++t) sIn[t][v] = parts[t];
    }
  };

  if (PF) {
#pragma unroll
    for (int i = 0; i < NV; ++i) fetch_one(0, i, pv[PF ? i : 0], pm[PF ? i : 0]);
  }
  for (int ch = 0; ch < nchunk; ++ch) {
    // normalisation coefficients of this chunk's 8 channels (wave-uniform)
    float csc[8], csh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = scale && (ch * KC + j < Cin);
      csc[j] = (ok ? scale[n * Cin + ch * KC + j] : 1.f) * sA;
      csh[j] = (ok ? shift[n * Cin + ch * KC + j] : 0.f) * sA;
    }
    // B fragments come straight from L2 through a register ring BD steps deep: a step of the big-layer variant
    // has 48 MFMAs (1536 cycles) of cover, a z-paired step only 12, so it looks 4 steps ahead
    // (measured for NT = 2 on f16x3: depth 4 is 1 % faster than 2, depth 6 is 3 % slower).  The ring is filled
    // BEFORE the staging phase (its registers are idle there), so the first MFMA of the chunk does not wait for L2.
    const bf16x8* wc = wp + (long long)ch * TERMS * NST * 2 * CoutP + boff;
    constexpr int BD = NT == 4 ? 1 : NT == 3 ? 2 : (ZP ? 4 : (TERMS == 2 ? 4 : (NT == 1 ? 2 : 1))) / ZT;
    constexpr int BPRE = BD >= 2 ? BD / 2 : BD;      // slots filled ahead of the staging (all of them would spill)
    bf16x8 bq[BD][NT][TERMS];
#pragma unroll
    for (int d = 0; d < BPRE; ++d)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < TERMS; ++q) bq[d][t][q] = wc[(q * NST + d) * 2 * CoutP + 32 * t];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (!PF) fetch_one(ch, i, pv[0], pm[0]);
      commit_one(ch, i, pv[PF ? i : 0], pm[PF ? i : 0], csc, csh);
    }
    __syncthreads();
    if (PF && ch + 1 < nchunk) {
#pragma unroll
      for (int i = 0; i < NV; ++i) fetch_one(ch + 1, i, pv[PF ? i : 0], pm[PF ? i : 0]);
    }
#pragma unroll
    for (int d = BPRE; d < BD; ++d)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < TERMS; ++q) bq[d][t][q] = wc[(q * NST + d) * 2 * CoutP + 32 * t];
    // ---- 14 tap-pair steps, fully unrolled (tap offsets are compile-time constants per lane half)
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      bf16x8 b[NT][TERMS];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < TERMS; ++q) b[t][q] = bq[s % BD][t][q];
      if (s + BD < NST) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < TERMS; ++q) bq[s % BD][t][q] = wc[(q * NST + s + BD) * 2 * CoutP + 32 * t];
      }
      constexpr int last_tap = ZP ? 35 : 26;
      const int tapA = 2 * s, tapB = (2 * s + 1 > last_tap) ? last_tap : 2 * s + 1;   // padded half-step: zero weights
      const int offA = ((tapA / 9) * HY + (tapA / 3) % 3) * HX + tapA % 3;
      const int offB = ((tapB / 9) * HY + (tapB / 3) % 3) * HX + tapB % 3;
      const int abase = vrow + (lh ? offB : offA);
#pragma unroll
      for (int p = 0; p < ZT; ++p) {     // the z-pairs of the brick share the B fragments of the step
        bf16x8 a[MR][TERMS];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int q = 0; q < TERMS; ++q) a[m][q] = sIn[q][abase + p * (2 * HY * HX) + m * HX];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            // smallest terms first
            if (TERMS == 3) {
              acc[p][m][t] = mfma16<TERMS>(a[m][2], b[t][0], acc[p][m][t]);
              acc[p][m][t] = mfma16<TERMS>(a[m][1], b[t][1], acc[p][m][t]);
              acc[p][m][t] = mfma16<TERMS>(a[m][0], b[t][2], acc[p][m][t]);
            }
            acc[p][m][t] = mfma16<TERMS>(a[m][1], b[t][0], acc[p][m][t]);
            acc[p][m][t] = mfma16<TERMS>(a[m][0], b[t][1], acc[p][m][t]);
            acc[p][m][t] = mfma16<TERMS>(a[m][0], b[t][0], acc[p][m][t]);
          }
      }
    }
  }
  // ---- epilogue (identical to the fp32 kernel): col = lane&31 (channel), row = voxel along x
  // GroupNorm statistics of the OUTPUT (the next layer's normalisation) ride in the epilogue: per-lane fp32 sums
  // of <= 64 values, then fp64 across the lanes / waves that share a channel, one (sum, sum^2) pair per brick.
  float st1[NT], st2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) st1[t] = st2[t] = 0.f;
  const long long sbrick = ((long long)n * tiles_z * tiles_y * tiles_x + ((long long)bz * tiles_y + by) * tiles_x + bx);
  if (ZP) {
    const int co = li & 15;                            // column = (channel, output plane)
#pragma unroll
    for (int p = 0; p < ZT; ++p) {
      const int gz = z0 + 2 * p + (li >> 4);
      if (gz < D && co < Cout) {
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const int gy = y0 + wy + m;
          if (gy >= H) continue;
          float* yp = y + ((((long long)n * D + gz) * H + gy) * W) * Cout + co;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int gx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (gx < W) {
              float v = acc[p][m][0][r] * desc + bv;
              if (relu_out) v = fmaxf(v, 0.f);
              yp[(long long)gx * Cout] = v;
              st1[0] += v; st2[0] += v * v;
            }
          }
        }
      }
    }
    if (stats_partial) {
      double d1 = (double)st1[0], d2 = (double)st2[0];
      d1 += __shfl_xor(d1, 16); d2 += __shfl_xor(d2, 16);
      d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
      __syncthreads();                                   // every wave is done with the LDS images
      double* sred = reinterpret_cast<double*>(&sIn[0][0]);
      if (lane < 16) { sred[(wv * 16 + lane) * 2] = d1; sred[(wv * 16 + lane) * 2 + 1] = d2; }
      __syncthreads();
      if (tid < 32 && (tid >> 1) < Cout) {
        const int c = tid >> 1, k = tid & 1;
        stats_partial[(sbrick * Cout + c) * 2 + k] =
            (sred[(0 * 16 + c) * 2 + k] + sred[(1 * 16 + c) * 2 + k]) + (sred[(2 * 16 + c) * 2 + k] + sred[(3 * 16 + c) * 2 + k]);
      }
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < ZT; ++p) {
    const int gz = z0 + 2 * p + wz;
    if (gz >= D) continue;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int gy = y0 + wy + m;
      if (gy >= H) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int co = co0 + 32 * t + li;
        if (co >= Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
        const long long rowoff = ((((long long)n * D + gz) * H + gy) * W) * Cout + co;
        float* yp = y + rowoff;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (gx < W) {
            float v = acc[p][m][t][r] * desc + bv;
            if (addend) v += addend[rowoff + (long long)gx * Cout];
            if (relu_out) v = fmaxf(v, 0.f);
            yp[(long long)gx * Cout] = v;
            st1[t] += v; st2[t] += v * v;
          }
        }
      }
    }
  }
  if (stats_partial) {
    __syncthreads();                                     // every wave is done with the LDS images
    double* sred = reinterpret_cast<double*>(&sIn[0][0]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double d1 = (double)st1[t], d2 = (double)st2[t];
      d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
      if (lh == 0) { sred[((wv * NT + t) * 32 + li) * 2] = d1; sred[((wv * NT + t) * 32 + li) * 2 + 1] = d2; }
    }
    __syncthreads();
    if (tid < 64 * NT) {
      const int k = tid & 1, c = tid >> 1;               // c = t * 32 + li
      const int co = co0 + c;
      if (co < Cout)
        stats_partial[(sbrick * Cout + co) * 2 + k] = (sred[((0 * NT) * 32 + c) * 2 + k] + sred[((1 * NT) * 32 + c) * 2 + k]) +
                                                      (sred[((2 * NT) * 32 + c) * 2 + k] + sred[((3 * NT) * 32 + c) * 2 + k]);
    }
  }
}

// =============================================================================================
// conv3_fwd_g_kernel -- the same implicit GEMM with the staging taken off the critical path (round 2).
// What bounded conv3_fwd_bf_kernel was its staging: per chunk every thread ran six serial load -> wait -> normalise ->
// split -> ds_write chains through registers, 0.7 of an MFMA phase long, while the matrix pipe idled (SQ: 46 % busy).
// Here ONE persistent 512-thread workgroup per CU walks a list of 32 x 8 x 4 output bricks (halo 34 x 10 x 6 = 2040
// voxels) as one pipeline of (brick, 8-channel chunk) stages:
//   * the raw fp32 halo of the NEXT stage -- the next chunk, or chunk 0 of the next brick -- is copied HBM -> LDS by
//     global_load_lds_dwordx4 (16 bytes per lane, no staging registers, no VALU) while the MFMAs of this stage run, so
//     neither a chunk nor a brick starts with an exposed HBM round trip, and a brick's output stores drain under the
//     next brick's first chunk;
//   * a short LDS -> LDS conversion phase (GroupNorm scale/shift, ReLU, zero padding, range scale, fp16 hi/lo split: the
//     same arithmetic, so results are bit-identical to conv3_fwd_bf_kernel) turns it into the two fragment images;
//   * eight waves = 4 output planes x 2 row halves, 4 rows x NT cout tiles each (128 accumulator registers for NT = 2):
//     two waves per SIMD keep the matrix pipe fed, every B fragment is reused by 4 rows, the halo re-read factor is
//     1.99 instead of 2.66.  ZP (Cout <= 16, z-paired weights): waves = 2 plane pairs x 4 row quarters, the N tile is
//     (16 couts x 2 planes) over the pair's 4-plane input window, 18 tap-pair steps.
// LDS: raw stage 2040 x 32 B + two fragment images 2040 x 16 B (reused as 8 x 8 KB epilogue tiles) + the DMA offset
// table 16 KB = 147 200 B.
constexpr int GTY = 8, GTZ = 4;
constexpr int GHY = GTY + 2, GHZ = GTZ + 2, GPL = HX * GHY * GHZ;      // 2040 halo voxels
constexpr int G_TPB = 512;
constexpr int G_SLOTS = 2 * GPL;                                       // 16-byte slots of the raw stage
constexpr int G_AUX = 0;     // (nt, aux = 2, measured 12 % slower: neighbouring bricks share halo lines through L2)
constexpr int G_NLD = (G_SLOTS + G_TPB - 1) / G_TPB;                   // 8 LDS-DMA instructions per thread and chunk
constexpr int G_NCV = (GPL + G_TPB - 1) / G_TPB;                       // 4 voxels converted per thread and chunk
constexpr int G_OFF_BYTES = G_NLD * G_TPB * 4;                         // the DMA source offsets live in LDS, not in VGPRs
constexpr int G_IMG_BYTES = 8 * 32 * 64 * 4;                           // fragment images (65 280 B) / epilogue tiles (8 x 8 KB)
constexpr int G_LDS_BYTES = G_SLOTS * 16 + G_IMG_BYTES + G_OFF_BYTES;  // 65 280 + 65 536 + 16 384 = 147 200

typedef __attribute__((address_space(3))) void* kmh_lds_ptr;
typedef const __attribute__((address_space(1))) void* kmh_glb_ptr;

// POOL (NT = 1, not z-paired: 16 < Cout <= 32): the epilogue applies MaxPool3d(2) (floor mode, ATen's first-max rule)
// to the brick it just computed -- a 32 x 8 x 4 brick at an even origin holds 16 x 4 x 2 whole windows -- and writes
// ONLY the pooled tensor (N, D/2, H/2, W/2, Cout), the winners' window indices (1 byte per pooled element, the format of
// kmh_maxpool3d_fwd) and the pooled tensor's (sum, sum^2) statistics: the full-resolution output of an encoder block that
// feeds nothing but the next level's pooling is never written (8.6 GB per step at 256^3) nor re-read by a pooling pass.
template <int NT, bool ZP, bool POOL = false>
__global__ __launch_bounds__(G_TPB, 2) void conv3_fwd_g_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const bf16x8* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, int D, int H, int W, int Cin,
    int Cout, int CoutP, int relu_in, int relu_out, int tiles_x, int tiles_y, int tiles_z, int tiles_zp,
    const float* __restrict__ ascale, const float* __restrict__ wscale, double* __restrict__ stats_partial,
    int in_blocked, const float* __restrict__ addend, int total_items, int N, long long* __restrict__ trace,
    unsigned* __restrict__ pool_arg = nullptr) {
  static_assert(!POOL || (NT == 1 && !ZP), "the pooling epilogue is built for the 32-wide tile");
  constexpr int TERMS = 2, MR = ZP ? 2 : 4;
  constexpr int NST = ZP ? NSTEP_Z : NSTEP;
  static_assert(!ZP || NT == 1, "z-paired tiles are for Cout <= 16");
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  float4* sRaw = reinterpret_cast<float4*>(gsm);                        // [GPL][2]: 8 fp32 channels per halo voxel
  bf16x8* sIn = reinterpret_cast<bf16x8*>(gsm + G_SLOTS * 16);          // [TERMS][GPL]
  int* sOff = reinterpret_cast<int*>(gsm + G_SLOTS * 16 + G_IMG_BYTES);     // [G_NLD][512] DMA source offsets (elements)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform by construction: keep it (and wz, wy, the
  const int li = lane & 31, lh = lane >> 5;                    // DMA's LDS bases, the per-wave step tests) in SGPRs
  const int ncog = ZP ? 1 : (Cout + 32 * NT - 1) / (32 * NT);
  const int tyz = (tiles_y + 7) >> 3;
  // work list of this workgroup: virtual block ids blockIdx.x, + gridDim.x, ... of a launch with N * total_items
  // blocks (sample-major), mapped like conv3_fwd_bf_kernel maps its blocks (cout groups adjacent, 8 x 8 (y, z) brick
  // patches per XCD; gridDim.x is a multiple of 8, so every id of the list lands on this workgroup's XCD).  The
  // workgroups of an XCD thus work on ~32 consecutive bricks of ONE sample at a time and share their halos through its
  // L2 (per-sample work lists, 8 consecutive bricks per XCD and sample, fetched 32 % more: PMC r2f vs r2c).
  struct Item { int n, cog, bx, by, bz; };
  const int total_all = N * total_items;
  auto decode = [&](int vb, Item& it) -> bool {
    const int gitem = xcd_remap(vb, total_all);
    it.n = gitem / total_items;
    const int item = gitem - it.n * total_items;
    it.cog = item % ncog;
    const int brick = item / ncog;
    const int lz8 = brick & 7, ly8 = (brick >> 3) & 7, patch = brick >> 6;
    const int pyi = patch % tyz, rest = patch / tyz;
    const int pzi = rest % tiles_zp;
    it.bx = rest / tiles_zp; it.by = pyi * 8 + ly8; it.bz = pzi * 8 + lz8;
    return it.by < tiles_y && it.bz < tiles_z;             // patches are padded to 8 x 8
  };
  auto next_item = [&](int& vb, Item& it) -> bool {        // advance to the next real brick of the list
    for (vb += gridDim.x; vb < total_all; vb += gridDim.x)
      if (decode(vb, it)) return true;
    return false;
  };
  int vb = (int)blockIdx.x - (int)gridDim.x;
  Item cur, nxt;
  if (!next_item(vb, cur)) return;

  const int wz = ZP ? 2 * (wv >> 2) : wv >> 1;             // first output plane of the wave
  const int wy = ZP ? (wv & 3) * MR : (wv & 1) * MR;
  const float sA = ascale ? ascale[0] : 1.f;
  const float desc = (ascale ? ascale[1] : 1.f) * (wscale ? wscale[1] : 1.f);
  const int nchunk = Cin / KC;
  const int vrow = (wz * GHY + wy) * HX + li;
  const long long vox = (long long)D * H * W;
  const long long chunk_stride = in_blocked ? vox * KC : KC;
  auto sample_base = [&](int n) { return in_blocked ? x + (long long)n * nchunk * vox * KC : x + (long long)n * vox * Cin; };

  // LDS-DMA descriptors of a brick: 16-byte slot e = r * 512 + tid holds half (e & 1) of halo voxel e >> 1; padding
  // voxels fetch the clamped in-volume voxel (any valid address: the conversion writes zeros for them)
  auto fill_offsets = [&](const Item& it) {
    const int x0 = it.bx * TX, y0 = it.by * GTY, z0 = it.bz * GTZ;
    int t_ = tid;                     // opaque: the 24 halo coordinates of this thread's slots are recomputed per brick
    asm volatile("" : "+v"(t_));      // instead of living in (spilled) registers across the whole pipeline
#pragma unroll
    for (int r = 0; r < G_NLD; ++r) {
      const int e = r * G_TPB + t_, v = (e >> 1) < GPL ? (e >> 1) : GPL - 1;
      const int lx = v % HX, ly = (v / HX) % GHY, lz = v / (HX * GHY);
      int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = z0 + lz - 1;
      gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
      gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
      gz = gz < 0 ? 0 : (gz > D - 1 ? D - 1 : gz);
      sOff[e] = ((gz * H + gy) * W + gx) * (in_blocked ? KC : Cin) + 4 * (e & 1);      // read back by this thread only
    }
  };
  auto inside_bits = [&](const Item& it) -> unsigned {     // bit i: voxel tid + 512 i of the halo is inside the volume
    const int x0 = it.bx * TX, y0 = it.by * GTY, z0 = it.bz * GTZ;
    unsigned bits = 0;
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int i = 0; i < G_NCV; ++i) {
      const int v = t_ + i * G_TPB;
      const int lx = v % HX, ly = (v / HX) % GHY, lz = v / (HX * GHY);
      const int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = z0 + lz - 1;
      const bool in = (v < GPL) && ((unsigned)gx < (unsigned)W) && ((unsigned)gy < (unsigned)H) && ((unsigned)gz < (unsigned)D);
      bits |= (in ? 1u : 0u) << i;
    }
    return bits;
  };
  auto dma_chunk = [&](int n, int ch) {             // this wave's 8 KB of a chunk's halo (offsets of the brick in sOff)
    const float* base = sample_base(n) + ch * chunk_stride;
#pragma unroll
    for (int r = 0; r < G_NLD; ++r) {
      if (r * G_TPB + tid < G_SLOTS)
        __builtin_amdgcn_global_load_lds((kmh_glb_ptr)(base + sOff[r * G_TPB + tid]),
                                         (kmh_lds_ptr)(sRaw + r * G_TPB + wv * 64), 16, 0, G_AUX);
    }
  };

  // ---- B fragments straight from L2 through a register ring BD tap-pair steps deep.  Measured facts that shape the loop:
  //   * loads and LDS-DMA of one wave share ONE vmcnt queue, and hipcc drains it completely at every use of a loaded
  //     register while a DMA is in flight: the B loads are therefore inline asm with hand-counted waits;
  //   * a count may only rely on the order of the B loads among themselves (an LDS-DMA can complete before an older
  //     load: counting DMAs as "younger, still in flight" gave wrong results): "vmcnt(number of B loads issued after
  //     the needed one)" is exact without a DMA in flight and merely stricter with one;
  //   * each wave issues its share of a DMA (8 x 1 KB) in ONE step, wave w in step w: its own next B loads queue
  //     behind 8 KB only, the two waves of a SIMD never stall in the same step, and the last piece has 6+ steps to land;
  //   * SQ counters put the matrix pipe at 55 % busy at an effective 1.93 GHz for the first (non-persistent) version of
  //     this kernel (46 % for conv3_fwd_bf_kernel), i.e. 1.12 PF of fp16 MFMA under the power cap.
  constexpr int BD = 4;
  constexpr int BL = 2 * NT;                                         // B loads per step
  const long long step_stride = 2ll * CoutP, term_stride = (long long)NST * step_stride;
  bf16x8 bq[BD][NT][TERMS];
  long long o0 = 0;
  auto b_issue = [&](int slot) {
    const bf16x8* p0 = wp + o0;
    const bf16x8* p1 = p0 + term_stride;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[slot][0][0]) : "v"(p0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[slot][0][1]) : "v"(p1) : "memory");
    if (NT == 2) {
      asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(bq[slot][NT - 1][0]) : "v"(p0) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(bq[slot][NT - 1][1]) : "v"(p1) : "memory");
    }
    o0 += step_stride;
  };

  // static priority for the second-dispatched half: the two waves of a SIMD (w, w + 4) are arbitrated by priority, then
  // AGE, and at equal priority the older wave finished every chunk ~7k cycles ahead of its partner and idled at the
  // barrier (cycle stamps, KMH_G_TRACE); MI355X_MICROARCH.md "Two waves per SIMD", item 4
  if (wv >= 4) __builtin_amdgcn_s_setprio(1);
  fill_offsets(cur);
  dma_chunk(cur.n, 0);
  unsigned cv_in = inside_bits(cur);
  int tr_n = 0;                                            // KMH_G_TRACE: s_memtime stamps of workgroup 0, wave 0
  auto stamp = [&]() {
    if (trace && blockIdx.x == 0 && tid == 0 && tr_n < 240) trace[tr_n++] = __builtin_readcyclecounter();
  };
  for (;;) {
    stamp();                                               // brick start
    const bool more = next_item(vb, nxt);
    const int co0 = cur.cog * (32 * NT);
    const int n = cur.n;
    const int boff = lh * CoutP + co0 + li;
    f32x16 acc[MR][NT];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    for (int ch = 0; ch < nchunk; ++ch) {
      float csc[8], csh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        csc[j] = (scale ? scale[n * Cin + ch * KC + j] : 1.f) * sA;
        csh[j] = (scale ? shift[n * Cin + ch * KC + j] : 0.f) * sA;
      }
      // this wave's DMAs of the stage have landed (and its output stores of the previous brick have drained); after the
      // barrier everybody's have -- and every wave is done with the MFMAs of the previous stage: the fragment images
      // may be overwritten
      stamp();                                             // chunk top
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp();                                             // own DMA landed / stores drained
      __builtin_amdgcn_s_barrier();
      stamp();                                             // barrier 1 passed
      // first BD steps of B fragments: in flight during the conversion
      o0 = (long long)ch * TERMS * term_stride + boff;
#pragma unroll
      for (int d = 0; d < BD; ++d) b_issue(d);
#pragma unroll
      for (int i = 0; i < G_NCV; ++i) {
        const int v = tid + i * G_TPB;
        if (v < GPL) {
          const float4 r0 = sRaw[2 * v], r1 = sRaw[2 * v + 1];
          const float raw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
          const bool in = (cv_in >> i) & 1u;
          float val[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float t = raw[j] * csc[j] + csh[j];
            if (relu_in) t = fmaxf(t, 0.f);
            val[j] = in ? t : 0.f;                         // zero padding AFTER the normalisation
          }
          bf16x8 parts[TERMS];
          split8<TERMS>(val, parts);
#pragma unroll
          for (int t = 0; t < TERMS; ++t) sIn[t * GPL + v] = parts[t];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp();                                             // conversion done
      __builtin_amdgcn_s_barrier();      // the fragment images are complete; the raw stage may be overwritten
      stamp();                                             // barrier 2 passed
      // the next stage's halo: the next chunk of this brick, or chunk 0 of the next brick (whose offsets replace this
      // brick's in the table: every DMA of this brick has been issued by now)
      const bool last_ch = ch + 1 == nchunk;
      const bool have_next = !last_ch || more;
      if (last_ch && more) fill_offsets(nxt);
      // (opaque copies: keep the per-step fragment addresses from being hoisted out of the loops into 14 + 14 VGPRs)
      int vr = vrow, lhv = lh;
      asm volatile("" : "+v"(vr), "+v"(lhv));
      // (Reading the A fragments of step s+1 into a second register set at the start of step s was measured and dropped:
      // NT = 2: 4 % slower (and 11 spilled registers); NT = 1 / z-paired tiles, 12 / 6 MFMAs per wave and step: the
      // tap-pair phase went from 14.5-17k to 16-18k cycles (KMH_G_TRACE).  hipcc's own placement -- each ds_read a few
      // MFMAs ahead of its first use -- stays.)
#pragma unroll
      for (int s = 0; s < NST; ++s) {
        if (s < 8 && s == wv && have_next) dma_chunk(last_ch ? nxt.n : n, last_ch ? 0 : ch + 1);
        {   // this step's B fragments: leave only the B loads issued after them in flight
          const int ahead = (s + BD - 1 < NST - 1 ? s + BD - 1 : NST - 1) - s;      // steps already issued beyond s
          switch (ahead * BL) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int last_tap = ZP ? 35 : 26;
        const int tapA = 2 * s, tapB = (2 * s + 1 > last_tap) ? last_tap : 2 * s + 1;      // padded half-step: zero weights
        const int offA = ((tapA / 9) * GHY + (tapA / 3) % 3) * HX + tapA % 3;
        const int offB = ((tapB / 9) * GHY + (tapB / 3) % 3) * HX + tapB % 3;
        const int abase = vr + (lhv ? offB : offA);
        bf16x8 a[MR][TERMS];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int q = 0; q < TERMS; ++q) a[m][q] = sIn[q * GPL + abase + m * HX];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[m][t] = mfma16<TERMS>(a[m][1], bq[s % BD][t][0], acc[m][t]);      // smallest terms first, as conv3_fwd_bf_kernel
            acc[m][t] = mfma16<TERMS>(a[m][0], bq[s % BD][t][1], acc[m][t]);
            acc[m][t] = mfma16<TERMS>(a[m][0], bq[s % BD][t][0], acc[m][t]);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (s + BD < NST) b_issue(s % BD);     // refill the slot just consumed
        __builtin_amdgcn_sched_barrier(0);
      }
      stamp();                                             // steps done
    }

    // ---- epilogue of the brick (the stores drain under the next brick's first stage).  The accumulators hold one
    // CHANNEL per lane (32 consecutive channels of a voxel across 32 lanes): stored as they are that is 128 four-byte
    // store instructions per lane and brick, which cost 56k cycles per brick -- two whole chunks (cycle stamps: the
    // store path is issue-bound, MI355X_MICROARCH.md "epilogue store tail").  So every wave transposes its tile row by
    // row through its own 8 KB of the (now idle) fragment-image region and stores 16 bytes per lane: 4x fewer
    // instructions, whole 64-byte channel runs per 4 lanes.
    const int x0 = cur.bx * TX, y0 = cur.by * GTY, z0 = cur.bz * GTZ;
    constexpr int CH = 32 * NT;                                // columns of the wave's tile
    constexpr int L4 = CH / 4;                                 // lanes per voxel in the transposed view (8 or 16)
    constexpr int VPI = 64 / L4;                               // voxels per read instruction (8 or 4)
    const long long sbrick = ((long long)n * tiles_z * tiles_y * tiles_x + ((long long)cur.bz * tiles_y + cur.by) * tiles_x + cur.bx);
    float* tile = reinterpret_cast<float*>(sIn) + wv * (32 * CH);
    const int c4 = lane % L4, vx = lane / L4;                  // this lane's column quad and first voxel in the view
    const int col = 4 * c4;
    const int pl = ZP ? col >> 4 : 0;                          // ZP: column = (channel, output plane of the pair)
    const int co = ZP ? (col & 15) : co0 + col;
    const bool co_ok = co < Cout;                              // Cout % 4 == 0 (launcher)
    float4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias && co_ok) bv = *reinterpret_cast<const float4*>(bias + co);
    float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                           // every wave is done with the fragment images
    const int gz = z0 + wz + pl;
    if constexpr (POOL) {
      // lane = (channel quad c4, x-pair group j): the wave's row tile is read back as voxel PAIRS (2 j + 16 kk, + 1), so
      // the x children of a window meet in one lane, its y children in consecutive rows of this wave, and its z
      // children in the wave two up (same rows, next plane): odd planes hand their (x, y)-pooled partials over through
      // LDS.  Scan order of the reference (z, y, x; a later value wins only if strictly greater, or NaN) is kept by
      // combining lower-index halves first.
      const int j = lane >> 3;
      float4 pm[2][2];
      unsigned pa[2][2];
      auto pick = [](float a, float b, unsigned ca, unsigned cb, float& m_, unsigned& c_) {
        const bool tb = (b > a) || (b != b);
        m_ = tb ? b : a; c_ = tb ? cb : ca;
      };
#pragma unroll
      for (int m = 0; m < MR; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * CH + li] = acc[m][0][r];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int xe = 2 * j + 16 * kk;
          const float4 a = *reinterpret_cast<const float4*>(tile + xe * CH + col);
          const float4 b = *reinterpret_cast<const float4*>(tile + (xe + 1) * CH + col);
          float va[4] = {a.x * desc + bv.x, a.y * desc + bv.y, a.z * desc + bv.z, a.w * desc + bv.w};
          float vb[4] = {b.x * desc + bv.x, b.y * desc + bv.y, b.z * desc + bv.z, b.w * desc + bv.w};
          float mx[4];
          unsigned cx[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (relu_out) { va[q] = fmaxf(va[q], 0.f); vb[q] = fmaxf(vb[q], 0.f); }
            pick(va[q], vb[q], 0u, 1u, mx[q], cx[q]);                      // x children: codes 0 / 1
          }
          if ((m & 1) == 0) {
            pm[m >> 1][kk] = float4{mx[0], mx[1], mx[2], mx[3]};
            pa[m >> 1][kk] = cx[0] | (cx[1] << 8) | (cx[2] << 16) | (cx[3] << 24);
          } else {                                                         // y children: + 2 for the second row
            float4& P = pm[m >> 1][kk];
            const unsigned A = pa[m >> 1][kk];
            float o0, o1, o2, o3;
            unsigned c0, c1, c2, c3;
            pick(P.x, mx[0], A & 255u, cx[0] + 2u, o0, c0);
            pick(P.y, mx[1], (A >> 8) & 255u, cx[1] + 2u, o1, c1);
            pick(P.z, mx[2], (A >> 16) & 255u, cx[2] + 2u, o2, c2);
            pick(P.w, mx[3], A >> 24, cx[3] + 2u, o3, c3);
            P = float4{o0, o1, o2, o3};
            pa[m >> 1][kk] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
          }
        }
      }
      // z children: waves 2, 3, 6, 7 (odd planes) publish, waves 0, 1, 4, 5 (even planes) combine and store
      float4* xv = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(sIn) + 8 * (32 * CH * 4));   // behind the 8 tiles
      unsigned* xa = reinterpret_cast<unsigned*>(xv + 4 * 4 * 64);
      const bool odd_plane = (wz & 1) != 0;
      const int slot = ((wv >> 2) * 2 + (wv & 1)) * 4 * 64;              // (plane pair, row half)
      if (odd_plane) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) { xv[slot + (2 * p + kk) * 64 + lane] = pm[p][kk]; xa[slot + (2 * p + kk) * 64 + lane] = pa[p][kk]; }
      }
      __syncthreads();
      if (!odd_plane) {
        const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
        const int oz = (z0 + wz) >> 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int oy = (y0 + wy + 2 * p) >> 1;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int ox = (x0 >> 1) + j + 8 * kk;
            const float4 Q = xv[slot + (2 * p + kk) * 64 + lane];
            const unsigned B = xa[slot + (2 * p + kk) * 64 + lane];
            const float4 P = pm[p][kk];
            const unsigned A = pa[p][kk];
            float o0, o1, o2, o3;
            unsigned c0, c1, c2, c3;
            pick(P.x, Q.x, A & 255u, (B & 255u) + 4u, o0, c0);
            pick(P.y, Q.y, (A >> 8) & 255u, ((B >> 8) & 255u) + 4u, o1, c1);
            pick(P.z, Q.z, (A >> 16) & 255u, ((B >> 16) & 255u) + 4u, o2, c2);
            pick(P.w, Q.w, A >> 24, (B >> 24) + 4u, o3, c3);
            if (oz < Do && oy < Ho && ox < Wo && co_ok) {
              const long long e = ((((long long)n * Do + oz) * Ho + oy) * Wo + ox) * Cout + co;
              *reinterpret_cast<float4*>(y + e) = float4{o0, o1, o2, o3};
              pool_arg[e >> 2] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
              st1[0] += o0; st2[0] += o0 * o0; st1[1] += o1; st2[1] += o1 * o1;
              st1[2] += o2; st2[2] += o2 * o2; st1[3] += o3; st2[3] += o3 * o3;
            }
          }
        }
      }
    } else {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * CH + 32 * t + li] = acc[m][t][r];
      const int gy = y0 + wy + m;
      const bool row_ok = gz < D && gy < H && co_ok;
      const long long rowoff = ((((long long)n * D + gz) * H + gy) * W) * Cout + co;
      float4 v4[32 / VPI], ad[32 / VPI];
#pragma unroll
      for (int k = 0; k < 32 / VPI; ++k) {
        const int xx = vx + VPI * k;
        v4[k] = *reinterpret_cast<const float4*>(tile + xx * CH + col);
        ad[k] = float4{0.f, 0.f, 0.f, 0.f};
        if (addend && row_ok && x0 + xx < W) ad[k] = *reinterpret_cast<const float4*>(addend + rowoff + (long long)(x0 + xx) * Cout);
      }
#pragma unroll
      for (int k = 0; k < 32 / VPI; ++k) {
        const int gx = x0 + vx + VPI * k;
        if (row_ok && gx < W) {
          float4 o;
          o.x = v4[k].x * desc + bv.x + ad[k].x; o.y = v4[k].y * desc + bv.y + ad[k].y;
          o.z = v4[k].z * desc + bv.z + ad[k].z; o.w = v4[k].w * desc + bv.w + ad[k].w;
          if (relu_out) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          *reinterpret_cast<float4*>(y + rowoff + (long long)gx * Cout) = o;
          st1[0] += o.x; st2[0] += o.x * o.x; st1[1] += o.y; st2[1] += o.y * o.y;
          st1[2] += o.z; st2[2] += o.z * o.z; st1[3] += o.w; st2[3] += o.w * o.w;
        }
      }
    }
    }   // !POOL
    if (stats_partial) {
      // per (wave, column) sums -> LDS -> one (sum, sum^2) pair per channel and brick, fixed order (deterministic)
      double d1[4], d2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d1[j] = (double)st1[j]; d2[j] = (double)st2[j];
#pragma unroll
        for (int o = L4; o < 64; o <<= 1) { d1[j] += __shfl_xor(d1[j], o); d2[j] += __shfl_xor(d2[j], o); }
      }
      __syncthreads();                                         // the tiles have been read back
      double* sred = reinterpret_cast<double*>(sIn);           // [wave][CH][2]  (NOT the raw stage: the next halo lands there)
      if (lane < L4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sred[((wv * CH) + col + j) * 2] = d1[j]; sred[((wv * CH) + col + j) * 2 + 1] = d2[j]; }
      }
      __syncthreads();
      const int ncol = ZP ? 16 : CH;
      if (tid < 2 * ncol) {
        const int k = tid & 1, c = tid >> 1;
        const int cch = ZP ? c : co0 + c;
        if (cch < Cout) {
          double sum = 0.0;
#pragma unroll
          for (int w8 = 0; w8 < 8; ++w8) {
            sum += sred[(w8 * CH + c) * 2 + k];
            if (ZP) sum += sred[(w8 * CH + 16 + c) * 2 + k];   // the second plane of the pair
          }
          stats_partial[(sbrick * Cout + cch) * 2 + k] = sum;
        }
      }
    }
    stamp();                                               // epilogue issued
    if (!more) break;
    cur = nxt;
    cv_in = inside_bits(cur);
  }
}


// =============================================================================================
// conv3_fwd_s_kernel -- ONE wave per SIMD (round 4).  What the cycle stamps of conv3_fwd_g_kernel say: a chunk is
// ~31.3k cycles = tap-pair steps 26.0k (two waves per SIMD share the matrix pipe: 77 cycles per MFMA and wave, 83 % of the
// pipe) + a conversion phase of 4.2k + two barriers, and with two waves per SIMD nothing can be moved into the MFMA shadow:
// their issue streams are zero-sum (MI355X_MICROARCH.md "Two waves per SIMD", item 3; measured here too: DESIGN.md section 8).
// A SINGLE wave per SIMD with the whole register file does hide up to ~5 single-issue instructions per MFMA gap (same
// guide, constants table).  So: 256 threads = 4 waves, wave = one plane of the 32 x 8 x 4 brick = 8 rows x NT cout tiles
// (256 accumulator registers for NT = 2, every B fragment reused by 8 rows), and per step the wave's stream is
//   wait for this step's B fragments | two LDS-DMA pieces of the NEXT stage's halo (steps 0-7) | the A fragments of the
//   NEXT step into a second register set | 48 MFMAs with, from step 8 on, the in-place conversion of the next stage's
//   voxels (the lane's own: it fetched both 16-byte halves itself) scheduled between them | the B loads of step s + 4.
// Two stage buffers of 2 x 2048 16-byte slots (fp32 halves in, fp16 hi / lo planes out, in place), ONE barrier per stage.
// Bit-identical to conv3_fwd_g_kernel / conv3_fwd_bf_kernel: per accumulator the three products keep their order (the loop
// runs term-major over the 16 accumulators, so back-to-back MFMAs never hit the same one).
constexpr int S_TPB = 256, S_MR = 8;
constexpr int S_PLANE = 2048;                                         // 16-byte slots per plane of a stage buffer (>= GPL)
constexpr int S_NLD = 2 * S_PLANE / S_TPB;                            // 16 LDS-DMA instructions per thread and chunk
constexpr int S_NCV = S_PLANE / S_TPB;                                // 8 voxels converted per thread and chunk
constexpr int S_BUF_BYTES = 2 * S_PLANE * 16;                         // 65 536
constexpr int S_OFF_BYTES = S_NLD * S_TPB * 4;                        // 16 384
constexpr int S_COEF = 1024;                                          // channels of one sample's (scale, shift) table
constexpr int S_LDS_BYTES = 2 * S_BUF_BYTES + S_OFF_BYTES + 2 * S_COEF * 4;      // 155 648
static_assert(GPL <= S_PLANE && GTZ == 4 && GTY == S_MR, "wave = plane, 8 rows");
#ifndef KMH_S_VPM
#define KMH_S_VPM 3
#endif
#ifndef KMH_S_BD
#define KMH_S_BD 2
#endif
#ifndef KMH_S_LEAD
#define KMH_S_LEAD 6
#endif
#ifndef KMH_S_RF
#define KMH_S_RF 1
#endif
#ifndef KMH_S_ADB
#define KMH_S_ADB 1
#endif
#ifndef KMH_SP_BA
#define KMH_SP_BA 9
#endif
#ifndef KMH_S_DEEP
#define KMH_S_DEEP 1
#endif
#ifndef KMH_S_POOLZ            // 1 = the pooling variant's waves own both planes of a pair (0: a plane each + LDS exchange, the A/B arm)
#define KMH_S_POOLZ 1
#endif
#ifndef KMH_S_CW               // 1 = hand-counted waits in the kernels that convert their operand (0: full drains, the A/B arm)
#define KMH_S_CW 1
#endif
#ifndef KMH_S_STAMPALL
#define KMH_S_STAMPALL 0
#endif
#ifndef KMH_S_IL               // 1 = LDS reads / conversion VALU dealt over the MFMA gaps of the whole step (0: reads first, the A/B arm)
#define KMH_S_IL 1
#endif
#ifndef KMH_S_ILR              // gaps that take LDS reads, reads per such gap
#define KMH_S_ILR 12
#endif
#ifndef KMH_S_ILRN
#define KMH_S_ILRN 2
#endif
#ifndef KMH_S_ILV0             // first gap that takes conversion VALU; VALU per gap on the 32-wide / 64-wide tile
#define KMH_S_ILV0 3
#endif
#ifndef KMH_S_ILV1
#define KMH_S_ILV1 3
#endif
#ifndef KMH_S_ILV2
#define KMH_S_ILV2 2
#endif
#ifndef KMH_S_UNCOND           // 1 = the step loop requests the "next stage's" fragments / voxels / pieces even when there is no next
#define KMH_S_UNCOND 1         // stage (addresses stay valid: stage 0 of the current pair): no branch inside a step, one scheduling
#endif                         // region per step (0 = the A/B arm)
#ifndef KMH_S_DEEP_RING_V      // 1 = the DEEP ring in "=v" registers: the variant that CRASHED (kept as the positive control of
#define KMH_S_DEEP_RING_V 0    // tests/test_asm_audit_cpu.py; never built into the library)
#endif
#ifndef KMH_SP_PSTEPS
#define KMH_SP_PSTEPS 5
#endif

// ZP (Cout <= 16, z-paired weights, NT = 1): wave = (plane pair, row half) -- 4 rows, the N tile is (16 couts x 2 planes) over the
// pair's 4-plane input window, 18 tap-pair steps of 12 MFMAs (the eight-wave kernel: 6 per wave and step, the most
// overhead-bound launch of the step).
// SPLIT (round 5): the input arrives ALREADY range-scaled and split -- x is (N, Cin/8, V + 1) records of 32 bytes, 8 fp16 hi
// then 8 fp16 lo of one voxel's chunk, record V of every (sample, chunk) plane all zeros (the source of every padding slot) --
// written so by its producer (kmh_maxpool3d_bwd_split: a pooling backward knows its output's range scale before it writes).
// A stage's fragment images are then 16 LDS-DMA pieces per lane (global_load_lds_dwordx4, hi -> plane 0, lo -> plane 1), one
// or two per step, and NO conversion: no staging registers, no VALU, no coefficient table.  What that buys where the
// conversion cannot hide: the z-paired 32 -> 16 data gradient at 256^3 has 18 x 12 MFMAs per chunk (6.9k cycles) against a
// conversion of the same length (cycle stamps, profiles/r5a): 1300-1800 cycles per conversion step against 384 of MFMA time.
// Bit-identical to the fp32 input: the producer applies the same fmaf(x, S, 0) and split8<2>.
// POOL (round 5; NT = 1, not z-paired: 16 < Cout <= 32): the pooling epilogue of conv3_fwd_g_kernel on this kernel's waves -- a
// wave holds one PLANE of the brick (8 rows), so the x children of a window meet in one lane of the row tile's read-back, its
// y children in consecutive rows of the same wave, and its z children in the wave next to it (odd planes publish their
// (x, y)-pooled partials through LDS); ATen's first-max rule, the same winners, the same outputs bit for bit.  What it buys:
// the plain one-wave kernel spends 38 % of a two-chunk 16 -> 32 brick in its epilogue storing 128 KB (cycle stamps, r5a); the
// pooled tensor is 16 KB.
template <int NT, bool ZP = false, bool SPLIT = false, bool POOL = false, bool AMP = false>
__global__ __launch_bounds__(S_TPB, 1) void conv3_fwd_s_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const bf16x8* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, int D, int H, int W, int Cin,
    int Cout, int CoutP, int relu_in, int relu_out, int tiles_x, int tiles_y, int tiles_z, int tiles_zp,
    const float* __restrict__ ascale, const float* __restrict__ wscale, double* __restrict__ stats_partial,
    int in_blocked, const float* __restrict__ addend, int total_items, int N, long long* __restrict__ trace,
    unsigned* __restrict__ pool_arg = nullptr) {
  constexpr int TERMS = 2, MR = ZP ? 4 : S_MR, NST = ZP ? NSTEP_Z : NSTEP;
  static_assert(!ZP || NT == 1, "z-paired tiles are for Cout <= 16");
  static_assert(!POOL || (NT == 1 && !ZP), "the pooling epilogue is built for the 32-wide tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  int* sOff = reinterpret_cast<int*>(gsm + 2 * S_BUF_BYTES);
  float* sCoef = reinterpret_cast<float*>(gsm + 2 * S_BUF_BYTES + S_OFF_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ncog = ZP ? 1 : (Cout + 32 * NT - 1) / (32 * NT);
  const int tyz = (tiles_y + 7) >> 3;
  struct Item { int n, cog, bx, by, bz; };
  const int total_all = N * total_items;
  auto decode = [&](int vb, Item& it) -> bool {      // the work list of conv3_fwd_g_kernel
    const int gitem = xcd_remap(vb, total_all);
    it.n = gitem / total_items;
    const int item = gitem - it.n * total_items;
    it.cog = item % ncog;
    const int brick = item / ncog;
    const int lz8 = brick & 7, ly8 = (brick >> 3) & 7, patch = brick >> 6;
    const int pyi = patch % tyz, rest = patch / tyz;
    const int pzi = rest % tiles_zp;
    it.bx = rest / tiles_zp; it.by = pyi * 8 + ly8; it.bz = pzi * 8 + lz8;
    return it.by < tiles_y && it.bz < tiles_z;
  };
  auto next_item = [&](int& vb, Item& it) -> bool {
    for (vb += gridDim.x; vb < total_all; vb += gridDim.x)
      if (decode(vb, it)) return true;
    return false;
  };
  int vb = (int)blockIdx.x - (int)gridDim.x;
  Item cur, nxt;
  if (!next_item(vb, cur)) return;

  const float sA = ascale ? ascale[0] : 1.f;
  const float desc = (ascale ? ascale[1] : 1.f) * (wscale ? wscale[1] : 1.f);
  const int nchunk = Cin / KC;
  // PZ (the pooling variant, round 5): a wave owns rows 4 (wv & 1) .. + 3 of BOTH planes of a plane pair (accumulator rows 0-3 /
  // 4-7) instead of 8 rows of one plane, so the 2 x 2 x 2 pooling window -- x: a register pair, y: two rows, z: rows m / m + 4 --
  // lies inside ONE wave: no exchange through LDS, no idle odd-plane waves, every wave transposes and stores 16 values per lane
  constexpr bool PZ = POOL && (KMH_S_POOLZ != 0);
  const int wz = (ZP || PZ) ? 2 * (wv >> 1) : wv, wy = ZP ? (wv & 1) * MR : (PZ ? (wv & 1) * (MR / 2) : 0);      // first output plane / row of the wave
  const int vrow = (wz * GHY + wy) * HX + li;
  auto arow = [](int m) -> int { return PZ ? (m & 3) * HX + (m >> 2) * (GHY * HX) : m * HX; };      // A-image offset of accumulator row m
  const long long vox = (long long)D * H * W;
  const long long plane = SPLIT ? (vox + 1) * KC : vox * KC;      // floats per (sample, chunk) plane of a channel-blocked input
  const long long chunk_stride = (in_blocked || SPLIT) ? plane : KC;
  auto sample_base = [&](int n) {
    return (in_blocked || SPLIT) ? x + (long long)n * nchunk * plane : x + (long long)n * vox * Cin;
  };

  // LDS-DMA descriptors of a brick: 16-byte slot e = r * 256 + tid holds half (r >> 3) of halo voxel (r & 7) * 256 + tid
  auto fill_offsets = [&](const Item& it) -> unsigned {      // -> bit i: voxel tid + 256 i of the halo is inside the volume
    const int x0 = it.bx * TX, y0 = it.by * GTY, z0 = it.bz * GTZ;
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if constexpr (SPLIT) {
      // one record offset (floats) per halo voxel of this lane: slot r * 256 + tid; padding voxels and the 8 slots past the
      // halo point at the plane's zero record
#pragma unroll
      for (int r = 0; r < S_NCV; ++r) {
        const int v = r * S_TPB + t_;
        const int lx = v % HX, ly = (v / HX) % GHY, lz = v / (HX * GHY);
        const int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = z0 + lz - 1;
        const bool in = (v < GPL) && ((unsigned)gx < (unsigned)W) && ((unsigned)gy < (unsigned)H) && ((unsigned)gz < (unsigned)D);
        sOff[r * S_TPB + t_] = in ? ((gz * H + gy) * W + gx) * KC : (int)vox * KC;      // read back by this thread only
      }
      return 0u;
    }
    // ONE coordinate decode per halo voxel of the lane: both 16-byte halves' offsets and the voxel's inside bit (round 5: the
    // table and inside_bits() used to decode the same voxels twice over, ~3k cycles of a brick's last stage and ~2k more
    // after its epilogue)
    unsigned bits = 0;
#pragma unroll
    for (int r = 0; r < S_NCV; ++r) {
      const int v0_ = r * S_TPB + t_, v = v0_ < GPL ? v0_ : GPL - 1;
      const int lx = v % HX, ly = (v / HX) % GHY, lz = v / (HX * GHY);
      int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = z0 + lz - 1;
      const bool in = (v0_ < GPL) && ((unsigned)gx < (unsigned)W) && ((unsigned)gy < (unsigned)H) && ((unsigned)gz < (unsigned)D);
      bits |= (in ? 1u : 0u) << r;
      gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
      gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
      gz = gz < 0 ? 0 : (gz > D - 1 ? D - 1 : gz);
      const int off = ((gz * H + gy) * W + gx) * (in_blocked ? KC : Cin);
      sOff[r * S_TPB + t_] = off;                 // read back by this thread only
      sOff[(8 + r) * S_TPB + t_] = off + 4;
    }
    return bits;
  };
  // the lane's voxel j of the next stage: its two 16-byte halves into a register ring (two voxels: loaded in step j, landed
  // by the drain at the head of step j + 1, converted there).  (LDS-DMA pieces -- no staging registers -- cost this single wave 100+ cycles of issue
  // each, 16 per chunk: steps with two pieces ran 1.8k cycles against the 1.54k of their 48 MFMAs.)
  typedef float kmh_f4 __attribute__((ext_vector_type(4)));      // (a native vector: the asm's "=v" operand)
  // DEEP (round 5, the 32-wide tile without a pre-split operand; KMH_S_DEEP=0: the A/B arm): its 24 MFMAs per step (768 cycles)
  // do not cover an HBM round trip, and a `vmcnt(0)` at every step head made every raw load issued in step s a wait at step
  // s + 1 (conversion steps 1200 cycles against 910 without a conversion).  As for SPLIT, the B ring holds a whole stage (in
  // AGPRs) and fragments are requested 7 steps ahead; the raw voxels are requested THREE steps before their conversion instead
  // of one.  (First with two drains per stage and compiler-visible raw loads: - 1-2 %; then with the counted waits below.)
  constexpr bool DEEP = (NT == 1) && !ZP && !SPLIT && (KMH_S_DEEP != 0);
  // CW (round 5, every kernel that converts its operand): COUNTED waits.  Loads return in issue order, so `vmcnt(K)` with K =
  // the number of loads issued after the one a step needs says exactly "that one has landed" -- and leaves the younger ones in
  // flight: a raw voxel requested CD steps before its conversion gets CD steps of MFMAs to come in from HBM (a full drain at the
  // next step head gave it one: 1.7-1.9k cycles against 0.9k idle and 2-4k loaded HBM latency; the conversion steps of the
  // 64-wide tile ran 2.1-2.6k cycles against 1.93k for plain ones).  The raw loads are inline asm as the fragment loads are
  // (the compiler's own waits for visible loads do not count the asm ones, i.e. wait for too much), every wait is followed by
  // empty asm statements that re-define the registers it covers (so no use can move above it), and tools/scan_asm_inflight.py
  // checks in the ISA that no destination is copied or spilled between its load and the wait that covers it -- the root of
  // round 4's "counted waits give run-to-run different results".  Output stores of an epilogue are OLDER than every load that
  // is waited for with a count (the fragments requested before an epilogue are drained / waited for with vmcnt(0)), so their
  // completion order does not matter: pending stores only make a counted wait stricter.
  constexpr bool CW = !SPLIT && (KMH_S_CW != 0);
  static_assert(!CW || KMH_S_UNCOND, "counted waits need the same loads in every stage");
  constexpr int CD = DEEP ? 3 : (CW ? 2 : 1);              // steps between a voxel's request and its conversion
  constexpr int RQ = CD + 1;                               // raw ring slots
  kmh_f4 rawq[RQ][2];
  auto raw_issue = [&](int n, int ch, int slot, int off0, int off1) {
    const float* base = sample_base(n) + ch * chunk_stride;
    const float* p0 = base + off0;
    const float* p1 = base + off1;
    if constexpr (CW) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rawq[slot][0]) : "v"(p0) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rawq[slot][1]) : "v"(p1) : "memory");
    } else {
      // (compiler-visible loads: its own waits for them are merely stricter than needed)
      rawq[slot][0] = *reinterpret_cast<const kmh_f4*>(p0);
      rawq[slot][1] = *reinterpret_cast<const kmh_f4*>(p1);
    }
  };
  auto raw_tie = [&](int slot) {                           // (after a wait: the slot's registers are defined HERE)
    asm volatile("" : "+v"(rawq[slot][0]));
    asm volatile("" : "+v"(rawq[slot][1]));
  };
  auto vm_wait = [](int K) {                               // s_waitcnt vmcnt(K), K a constant after unrolling
    switch (K) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  int coef_n = -1;
  auto fill_coef = [&](int n_) {                      // (the caller's barrier publishes it)
    for (int e = tid; e < Cin; e += S_TPB) {
      sCoef[e] = (scale ? scale[n_ * Cin + e] : 1.f) * sA;
      sCoef[S_COEF + e] = (scale ? shift[n_ * Cin + e] : 0.f) * sA;
    }
    coef_n = n_;
  };
  // conversion of the lane's voxel i (raw halves r0, r1 in registers) into the fragment images of stage buffer pb (sample
  // coef_n): branch-free, scheduled INTO a step's MFMA block; slots past the halo hold a clamped voxel's data and become zeros
  const float relu_floor = relu_in ? 0.f : -__builtin_inff();
  auto convert1 = [&](int ch, unsigned bits, int pb, int i, const kmh_f4 r0, const kmh_f4 r1) {
    const float4 a0 = *reinterpret_cast<const float4*>(sCoef + ch * KC), a1 = *reinterpret_cast<const float4*>(sCoef + ch * KC + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(sCoef + S_COEF + ch * KC), b1 = *reinterpret_cast<const float4*>(sCoef + S_COEF + ch * KC + 4);
    const float csc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float csh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    bf16x8* B8 = reinterpret_cast<bf16x8*>(gsm + pb * S_BUF_BYTES);
    const int v = tid + i * S_TPB;
    const float raw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    const bool in = (bits >> i) & 1u;
    // ReLU and the zero padding (AFTER the normalisation) as ONE median per value: inside the volume the bounds are
    // (0 or -inf, +inf), outside (0, 0) -- 2 selects + 8 v_med3 per voxel instead of 8 maxima + 16 selects
    const float lo = in ? relu_floor : 0.f, hi = in ? __builtin_inff() : 0.f;
    float val[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) val[j] = __builtin_amdgcn_fmed3f(raw[j] * csc[j] + csh[j], lo, hi);
    bf16x8 parts[TERMS];
    split8<TERMS>(val, parts);
#pragma unroll
    for (int t = 0; t < TERMS; ++t) B8[t * S_PLANE + v] = parts[t];
  };

  // SPLIT: piece p (0..15) of a stage = plane (p & 1) of the lane's voxel p >> 1: 16 bytes per lane straight into the
  // fragment image of stage buffer `buf` (wave-uniform LDS base + 16 x lane), source = the voxel's record (+ 16 bytes for lo)
  auto dma_piece = [&](int n, int ch, int buf, int p, int off) {
    const float* src = sample_base(n) + ch * chunk_stride + off + 4 * (p & 1);
    unsigned char* dst = gsm + buf * S_BUF_BYTES + ((p & 1) * S_PLANE + (p >> 1) * S_TPB + wv * 64) * 16;
    __builtin_amdgcn_global_load_lds((kmh_glb_ptr)src, (kmh_lds_ptr)dst, 16, 0, 0);
  };
  // B fragments straight from L2 through a two-slot register ring that runs THROUGH the stage boundaries: the fragments of step
  // s + 1 are requested at the HEAD of step s (into the slot step s - 1 has just finished issuing from), so they have a whole
  // step -- ~1.5k cycles, several L2 round trips -- to land.  Without CW (KMH_S_CW=0, round 4) every step opens with a plain
  // `s_waitcnt vmcnt(0)`; round 4's counted waits gave run-to-run different results here -- the compiler was moving in-flight
  // destinations (see CW above and b_issue below), which the tied waits and the ISA audit now exclude.
  // SPLIT: the ring holds a whole stage's fragments (one slot per step) and the fragments of step s + BA are requested in step
  // s, so that the wave drains its memory queue only at the head of every BA-th step: the LDS-DMA pieces of the next stage's
  // halo, issued in the steps right after a drain, then have BA - SP_PS + 1 or more steps (thousands of cycles) to come in
  // from HBM before anything waits for them -- a drain at every step head would expose that latency 18 times per chunk.
  constexpr int BD = (SPLIT || DEEP) ? NST : KMH_S_BD;   // ring slots
  constexpr int BA = SPLIT ? KMH_SP_BA : (DEEP ? 7 : 1); // request distance = drain period (steps)
  constexpr int SP_PS = KMH_SP_PSTEPS;                   // SPLIT: the stage's 16 DMA pieces go out in steps 0 .. SP_PS - 1
  static_assert(NST % BD == 0 && (SPLIT || DEEP || BD == 2), "the ring slot of a step must not depend on the stage");
  static_assert(!SPLIT || (NST % BA == 0 && SP_PS < BA && NT == 1), "drains at steps 0, BA, ...; pieces land before the next one");
  auto piece_beg = [](int s) -> int { return s >= SP_PS ? S_NLD : (S_NLD * s) / SP_PS; };      // pieces of steps 0 .. SP_PS - 1
  constexpr int BL = 2 * NT;
  const long long step_stride = 2ll * CoutP, term_stride = (long long)NST * step_stride;
  bf16x8 bq[BD][NT][TERMS];
  long long o0 = 0;
  // (AMP: the lo fragments are never multiplied and must not even be requested -- an asm load whose destination the compiler
  // sees as dead lands LATER in a register it has meanwhile given to something else)
  auto b_issue = [&](int slot) {
    const bf16x8* p0 = wp + o0;
    const bf16x8* p1 = p0 + term_stride;
    // The stage-deep ring of the 32-wide tile lives in the ACCUMULATOR half of the register file.  With "=v" destinations the
    // kernel needs 256 VGPRs + 218 AGPRs and the compiler, for which an asm load's destination is defined as soon as the
    // statement ends, moved 11 of the ring's registers into AGPRs (v_accvgpr_write) while their loads were still in flight: the
    // copies held stale data, the loads landed in registers since given to something else, the GPU faulted.  "=a" destinations
    // leave it nothing to move (MFMA reads its B operand from AGPRs directly); tools/scan_asm_inflight.py audits the ISA of
    // every instance for such copies and runs in the CPU test suite.
    if constexpr (DEEP && !KMH_S_DEEP_RING_V) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(bq[slot][0][0]) : "v"(p0) : "memory");
      if (!AMP) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(bq[slot][0][1]) : "v"(p1) : "memory");
      o0 += step_stride;
      return;
    }
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[slot][0][0]) : "v"(p0) : "memory");
    if (!AMP) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[slot][0][1]) : "v"(p1) : "memory");
    if (NT == 2) {
      asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(bq[slot][NT - 1][0]) : "v"(p0) : "memory");
      if (!AMP) asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(bq[slot][NT - 1][1]) : "v"(p1) : "memory");
    }
    o0 += step_stride;
  };
  auto b_tie = [&](int slot) {                           // (after a wait: the slot's registers are defined HERE)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = AMP ? 0 : 1; q >= 0; --q) {
        if constexpr (DEEP && !KMH_S_DEEP_RING_V) asm volatile("" : "+a"(bq[slot][t][q]));
        else asm volatile("" : "+v"(bq[slot][t][q]));
      }
  };
  // CW: loads issued after the youngest one step s needs, up to the wait at its head (a step issues its fragments, then, in
  // steps 1 .. 8, a raw voxel's two halves): the fragments of step s were requested BA steps earlier, the voxel it converts CD
  constexpr int NBL = NT * (AMP ? 1 : 2);                // loads of one b_issue
  auto cw_count = [&](int s_) -> int {
    auto nr = [&](int t) -> int { t = ((t % NST) + NST) % NST; return (t >= 1 && t < 9) ? 2 : 0; };
    int yr = 0, yb = nr(s_ - BA);
    for (int t = s_ - CD + 1; t < s_; ++t) yr += NBL + nr(t);
    for (int t = s_ - BA + 1; t < s_; ++t) yb += NBL + nr(t);
    return yr < yb ? yr : yb;
  };
  auto a_offset = [&](int s) -> int {                    // LDS slot offset of this lane's A fragment of step s, row 0
    constexpr int last_tap = ZP ? 35 : 26;
    const int tapA = 2 * s, tapB = (2 * s + 1 > last_tap) ? last_tap : 2 * s + 1;      // padded half-step: zero weights
    const int offA = ((tapA / 9) * GHY + (tapA / 3) % 3) * HX + tapA % 3;
    const int offB = ((tapB / 9) * GHY + (tapB / 3) % 3) * HX + tapB % 3;
    return lh ? offB : offA;
  };

  int tr_n = 0;                                            // KMH_G_TRACE: cycle stamps of workgroup 0, wave 0
  auto stamp = [&]() {
    if (trace && blockIdx.x == 0 && tid == 0 && tr_n < 240) trace[tr_n++] = __builtin_readcyclecounter();
  };
  // prologue: the first stage's halo, fetched and converted with nothing to hide behind
  unsigned cv_in = fill_offsets(cur);
  unsigned cv_brick_next = 0u;                             // the NEXT brick's bits, formed with its table in its predecessor's last stage
  if constexpr (SPLIT) {
#pragma unroll
    for (int p = 0; p < S_NLD; ++p) dma_piece(cur.n, 0, 0, p, sOff[(p >> 1) * S_TPB + tid]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    fill_coef(cur.n);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < S_NCV; ++i) {
      raw_issue(cur.n, 0, 0, sOff[i * S_TPB + tid], sOff[(8 + i) * S_TPB + tid]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (CW) raw_tie(0);
      convert1(0, cv_in, 0, i, rawq[0][0], rawq[0][1]);
    }
  }
  o0 = lh * CoutP + cur.cog * (32 * NT) + li;             // chunk 0 of the first brick: the fragments of steps 0 .. BA - 1
#pragma unroll
  for (int d = 0; d < BA; ++d) b_issue(d);
  if (SPLIT || DEEP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int pb = 0;                                              // stage buffer holding the CURRENT stage's fragment images
  for (;;) {
    const bool more = next_item(vb, nxt);
    const int co0 = cur.cog * (32 * NT);
    const int n = cur.n;
    const int boff = lh * CoutP + co0 + li;
    f32x16 acc[MR][NT];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    for (int ch = 0; ch < nchunk; ++ch) {
      // stage top: this wave's conversion writes are out; after the barrier everybody's are -- and every wave is done with
      // the MFMAs (and the epilogue tiles) of the previous stage: the other buffer may be overwritten by the next DMA
      stamp();                                             // stage top
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stamp();                                             // barrier passed
      const bf16x8* sIn = reinterpret_cast<const bf16x8*>(gsm + pb * S_BUF_BYTES);      // [TERMS][S_PLANE]
      const bool last_ch = ch + 1 == nchunk;
      const bool have_next = KMH_S_UNCOND || !last_ch || more;
      const int nn = (last_ch && more) ? nxt.n : n, nch = last_ch ? 0 : ch + 1;      // (no next stage: any valid pair)
      unsigned cv_next = last_ch ? 0u : cv_in;             // (the next BRICK's table and bits: inside step 0, below)
      if (!SPLIT && nn != coef_n) {                        // uniform, rare: the work list moves on to another sample
        fill_coef(nn);
        __syncthreads();
      }
      const int co0n = ((last_ch && more) ? nxt.cog : cur.cog) * (32 * NT);      // the next stage's cout group
      int vr = vrow;
      asm volatile("" : "+v"(vr));
      int dof0 = 0, dof1 = 0;                              // the next raw voxel's source offsets (read one step ahead)
      bf16x8 a[KMH_S_ADB ? 2 : 1][MR][TERMS];              // this step's A fragments (and the next step's)
      {
        const int ab = vr + a_offset(0);
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int q = 0; q < TERMS; ++q) a[0][m][q] = sIn[q * S_PLANE + ab + arow(m)];
      }
#pragma unroll
      for (int s = 0; s < NST; ++s) {
        // everything requested so far has landed: this step's B fragments (requested at the head of the last step), the raw
        // voxel s - 1, a previous brick's output stores
        // (SPLIT: only every BA-th step drains; step 0 of a brick's first stage does not either -- its fragments were waited
        // for ahead of the previous brick's epilogue, whose output stores thus stay in flight under BA steps of MFMAs)
        // (CW: counted -- the fragments of this step and the voxel it converts have landed, younger requests stay in flight;
        // DEEP: the first steps of a brick's first stage need nothing that was not waited for ahead of the previous epilogue)
        if constexpr (CW) {
          if (!(DEEP && ch == 0 && s <= CD)) vm_wait(cw_count(s));
          b_tie(s % BD);
          if (s >= 1 + CD && s < 9 + CD) raw_tie((s - 1 - CD) % RQ);
        } else if (!(SPLIT || DEEP) || (s % BA == 0 && !(s == 0 && ch == 0))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (s + BA < NST) b_issue((s + BA) % BD);             // the fragments of step s + BA ...
        else if (have_next) {                                 // ... or of step s + BA - NST of the next stage
          if (s + BA == NST) o0 = (long long)nch * TERMS * term_stride + lh * CoutP + co0n + li;
          b_issue((s + BA - NST) % BD);
        }
        if (CW && s >= 1 && s < 9) raw_issue(nn, nch, (s - 1) % RQ, dof0, dof1);      // (CW: a fixed place in the load order)
        __builtin_amdgcn_sched_barrier(0);
        // the next stage's voxel s - 1 is requested in step s (1..8) and converted in step s + 1.  Step 0 of a brick's last stage
        // first replaces the offset table and the inside bits by the next brick's (~400 VALU: fillers here, 3k cycles at the
        // stage top before)
        if (s == 0 && last_ch && more) { cv_brick_next = fill_offsets(nxt); cv_next = cv_brick_next; }
        if constexpr (SPLIT) {
          // this step's share of the next stage's 16 pieces (step 0: after the offset table has been replaced)
          if (have_next) {
#pragma unroll
            for (int p = piece_beg(s); p < piece_beg(s + 1); ++p) dma_piece(nn, nch, pb ^ 1, p, sOff[(p >> 1) * S_TPB + tid]);
          }
        } else {
        if (!CW && s >= 1 && s < 9 && have_next) raw_issue(nn, nch, (s - 1) % RQ, dof0, dof1);
        if (s < 8) { dof0 = sOff[s * S_TPB + tid]; dof1 = sOff[(8 + s) * S_TPB + tid]; }
        }
        if (KMH_S_ADB && s + 1 < NST) {
          const int ab = vr + a_offset(s + 1);
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int q = 0; q < TERMS; ++q) a[(s + 1) & 1][m][q] = sIn[q * S_PLANE + ab + arow(m)];
        }
        if (!KMH_S_ADB && s > 0) {                          // single set: read right here, the compiler places the reads
          const int ab = vr + a_offset(s);
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int q = 0; q < TERMS; ++q) a[0][m][q] = sIn[q * S_PLANE + ab + arow(m)];
        }
        // the next stage's voxel s - 2 (requested in the last step)
        if (!SPLIT && s >= 1 + CD && s < 9 + CD) convert1(nch, cv_next, pb ^ 1, s - 1 - CD, rawq[(s - 1 - CD) % RQ][0], rawq[(s - 1 - CD) % RQ][1]);
        // term-major over the 8 x NT accumulators: per accumulator the order of conv3_fwd_bf_kernel (smallest terms first)
#pragma unroll
        for (int q3 = AMP ? 2 : 0; q3 < 3; ++q3)       // (AMP: hi x hi only)
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[m][t] = mfma16<TERMS>(a[KMH_S_ADB ? (s & 1) : 0][m][q3 == 0 ? 1 : 0], bq[s % BD][t][q3 == 1 ? 1 : 0], acc[m][t]);
        constexpr int NMF = MR * NT * (AMP ? 1 : 3);          // MFMAs of a step
        if constexpr (KMH_S_IL != 0) {
          // ONE scheduling region per step (no branch inside: KMH_S_UNCOND), its single-issue instructions dealt over the MFMA
          // gaps: a wave alone on its SIMD hides about five of them beside a 32-cycle MFMA, and whatever sits in front of the
          // step's first MFMA runs with the matrix pipe idle (the "reads first" arrangement put 30-60 instructions there).
          const bool conv_step = !SPLIT && s >= 1 + CD && s < 9 + CD;
#pragma unroll
          for (int k = 0; k < NMF; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // one MFMA
            if (k < KMH_S_ILR) __builtin_amdgcn_sched_group_barrier(0x100, KMH_S_ILRN, 0);      // LDS reads: early gaps
            if (k == 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                       // the raw voxel's two loads
            if (conv_step && k >= KMH_S_ILV0) __builtin_amdgcn_sched_group_barrier(0x002, NT == 2 ? KMH_S_ILV2 : KMH_S_ILV1, 0);
          }
        } else {
        if ((SPLIT || !(s >= 1 + CD && s < 9 + CD)) && KMH_S_RF) __builtin_amdgcn_sched_group_barrier(0x100, 64, 0);      // plain steps: reads first too
        if (!SPLIT && s >= 1 + CD && s < 9 + CD) {
          // every LDS read of the block first (the next step's A fragments, the coefficients, the next offsets), then a few bare
          // MFMAs while they land -- a wait in the middle of the MFMA stream stalls it --, then the conversion's VALU a few per gap
          __builtin_amdgcn_sched_group_barrier(0x100, 64, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, KMH_S_LEAD, 0);
#pragma unroll
          for (int k = 0; k < MR * NT * (AMP ? 1 : 3) - KMH_S_LEAD; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x002, KMH_S_VPM, 0);     // a few of the conversion's VALU ...
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // ... one MFMA
          }
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        if (KMH_S_STAMPALL || s == 0 || s == 9 || s == NST - 1) stamp();      // steps 0 / .. 9 / .. NST - 1 done (debug builds: every step)
      }
      if (!last_ch) pb ^= 1;                               // (after a brick's last stage the epilogue still uses its buffer)
    }

    // ---- epilogue of the brick: the wave's rows go through its own 8 KB of the (now idle) stage buffer, row by row, and
    // are stored 16 bytes per lane (conv3_fwd_g_kernel's epilogue, 8 rows per wave).  (Forming the products transposed --
    // weights as the A operand, so that four accumulator registers are four channels of one voxel and no LDS transposition is
    // needed -- was measured: the 32-byte pieces those stores write cost 60-68k cycles per brick against 33k here.)
    // (SPLIT: the next stage's first fragments, requested in the last BA steps, land here -- L2 hits, long issued -- so that
    // step 0 of the next brick need not drain the queue the output stores below are about to fill)
    if (SPLIT || DEEP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int x0 = cur.bx * TX, y0 = cur.by * GTY, z0 = cur.bz * GTZ;
    constexpr int CH = 32 * NT;
    constexpr int L4 = CH / 4;
    constexpr int VPI = 64 / L4;
    const long long sbrick = ((long long)n * tiles_z * tiles_y * tiles_x + ((long long)cur.bz * tiles_y + cur.by) * tiles_x + cur.bx);
    unsigned char* sEp = gsm + pb * S_BUF_BYTES;
    float* tile0 = reinterpret_cast<float*>(sEp) + wv * (2 * 32 * CH);      // two tiles per wave: row m + 1 is written while
                                                                          // row m's read-back and stores are in flight
    const int c4 = lane % L4, vx = lane / L4;
    const int col = 4 * c4;
    const int pl = ZP ? col >> 4 : 0;                          // ZP: column = (channel, output plane of the pair)
    const int co = ZP ? (col & 15) : co0 + col;
    const bool co_ok = co < Cout;
    float4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias && co_ok) bv = *reinterpret_cast<const float4*>(bias + co);
    float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                           // every wave is done with the fragment images
    const int gz = z0 + wz + pl;
    if (POOL) stamp();
    if constexpr (POOL) {
      // (x, y) pooling IN REGISTERS: a lane's accumulator registers (2 q, 2 q + 1) of row m are the two x children of pooled
      // column X = (q & 1) + 4 (q >> 1) + 2 lh for ONE channel (column li of the tile), rows 2 p / 2 p + 1 of the same wave its
      // y children: no LDS round trip per row (the row-tile version: 16 stores, a wait, 4 loads, a wait, eight times over --
      // this single wave per SIMD is latency-bound there).  Planes wz (even) and wz + 1 hold the z children: odd planes publish
      // through LDS.  Scan order of the reference (z, y, x; a later value wins only if strictly greater, or NaN): lower-index
      // halves are combined first -- the same winners as kmh_maxpool3d_fwd.
      // (after a ReLU no value is NaN -- v_max_f32 returns its other operand -- and ATen's "a NaN wins" test, one unordered compare
      // and one mask OR per comparison, 274 of the epilogue's ~ 2000 instructions, is compiled out: two instances of the block)
      auto pool_block = [&](auto nan_wins) {
      constexpr bool NANW = decltype(nan_wins)::value;
      auto pick = [](float a, float b, unsigned ca, unsigned cb, float& m_, unsigned& c_) {
        const bool tb = (b > a) || (NANW && (b != b));
        m_ = tb ? b : a; c_ = tb ? cb : ca;
      };
      const int cl = co0 + li;
      const float bch = (bias && cl < Cout) ? bias[cl] : 0.f;
      float pvv[MR / 2][8];
      unsigned pcc[MR / 2];                                                 // 8 window codes of 4 bits
#pragma unroll
      for (int p = 0; p < MR / 2; ++p) {
        pcc[p] = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float a0 = acc[2 * p][0][2 * q] * desc + bch, a1 = acc[2 * p][0][2 * q + 1] * desc + bch;
          float b0 = acc[2 * p + 1][0][2 * q] * desc + bch, b1 = acc[2 * p + 1][0][2 * q + 1] * desc + bch;
          if (relu_out) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); b0 = fmaxf(b0, 0.f); b1 = fmaxf(b1, 0.f); }
          float t0, t1, t;
          unsigned c0, c1, c;
          pick(a0, a1, 0u, 1u, t0, c0);                                      // x children of the first row: codes 0 / 1
          pick(b0, b1, 0u, 1u, t1, c1);                                      // ... of the second row
          pick(t0, t1, c0, c1 + 2u, t, c);                                   // y children: + 2 for the second row
          pvv[p][q] = t;
          pcc[p] |= c << (4 * q);
        }
      }
      stamp();
      if constexpr (PZ) {
        // z children = rows m / m + 4 of THIS wave: pooled rows p = 0, 1 of the lower plane meet p + 2 of the upper one.  Then
        // the wave's pooled tile (2 rows x 16 columns x 32 channels) is transposed through its own 5 KB so that a lane stores 4
        // channels of one pooled voxel (16 bytes; a wave instruction = 8 voxels = 1 KB of contiguous output).
        float* tv = reinterpret_cast<float*>(sEp) + wv * (32 * 32 + 32 * 8);                                // [32 voxels][32]
        unsigned char* tc = reinterpret_cast<unsigned char*>(tv + 32 * 32);                                 // [32 voxels][32] bytes
#pragma unroll
        for (int p = 0; p < MR / 4; ++p) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float o;
            unsigned c;
            pick(pvv[p][q], pvv[p + MR / 4][q], (pcc[p] >> (4 * q)) & 15u, ((pcc[p + MR / 4] >> (4 * q)) & 15u) + 4u, o, c);
            const int X = (q & 1) + 4 * (q >> 1) + 2 * lh;
            tv[(p * 16 + X) * 32 + li] = o;
            tc[(p * 16 + X) * 32 + li] = (unsigned char)c;
          }
        }
        // (the wave's own LDS writes are ordered before its reads)
        stamp();
        const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
        const int oz = (z0 + wz) >> 1;
        const int jx = lane >> 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int vi = jx + 8 * k, pr = vi >> 4, X = vi & 15;
          const float4 o4 = *reinterpret_cast<const float4*>(tv + vi * 32 + col);
          const unsigned cw = *reinterpret_cast<const unsigned*>(tc + vi * 32 + col);
          const int oy = ((y0 + wy) >> 1) + pr, ox = (x0 >> 1) + X;
          if (oz < Do && oy < Ho && ox < Wo && co_ok) {
            const long long e = ((((long long)n * Do + oz) * Ho + oy) * Wo + ox) * Cout + co;
            *reinterpret_cast<float4*>(y + e) = o4;
            pool_arg[e >> 2] = cw;
            st1[0] += o4.x; st2[0] += o4.x * o4.x; st1[1] += o4.y; st2[1] += o4.y * o4.y;
            st1[2] += o4.z; st2[2] += o4.z * o4.z; st1[3] += o4.w; st2[3] += o4.w * o4.w;
          }
        }
      } else {
      // z children: waves 1, 3 (odd planes) publish, waves 0, 2 combine.  (Splitting the rest of the epilogue between the two
      // waves of a pair -- each finishing two of the four window rows -- was measured: 4.08 against 3.95-4.08 ms, the selects
      // that deal the halves cost what the idle partner would have saved.)
      float* xv = reinterpret_cast<float*>(sEp);                              // [pair][36][64]: 32 values + 4 code words per lane
      const bool odd_plane = (wv & 1) != 0;
      float* xs = xv + (wv >> 1) * (36 * 64) + lane;
      if (odd_plane) {
#pragma unroll
        for (int p = 0; p < MR / 2; ++p) {
#pragma unroll
          for (int q = 0; q < 8; ++q) xs[(p * 8 + q) * 64] = pvv[p][q];
          xs[(32 + p) * 64] = __uint_as_float(pcc[p]);
        }
      }
      __syncthreads();
      stamp();
      if (!odd_plane) {
        // ... and transpose the pooled tile (4 rows x 16 columns x 32 channels) through this wave's 10 KB so that a lane stores 4
        // channels of one pooled voxel (16 bytes; a wave instruction = 8 voxels = 1 KB of contiguous output)
        float* tv = reinterpret_cast<float*>(sEp + 2 * 36 * 64 * 4) + (wv >> 1) * (64 * 32 + 64 * 8);      // [64 voxels][32]
        unsigned char* tc = reinterpret_cast<unsigned char*>(tv + 64 * 32);                                 // [64 voxels][32] bytes
#pragma unroll
        for (int p = 0; p < MR / 2; ++p) {
          const unsigned B = __float_as_uint(xs[(32 + p) * 64]);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float Q = xs[(p * 8 + q) * 64];
            float o;
            unsigned c;
            pick(pvv[p][q], Q, (pcc[p] >> (4 * q)) & 15u, ((B >> (4 * q)) & 15u) + 4u, o, c);
            const int X = (q & 1) + 4 * (q >> 1) + 2 * lh;
            tv[(p * 16 + X) * 32 + li] = o;
            tc[(p * 16 + X) * 32 + li] = (unsigned char)c;
          }
        }
        // (the wave's own LDS writes are ordered before its reads)
        stamp();
        const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
        const int oz = (z0 + wz) >> 1;
        const int jx = lane >> 3;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int vi = jx + 8 * k, pr = vi >> 4, X = vi & 15;
          const float4 o4 = *reinterpret_cast<const float4*>(tv + vi * 32 + col);
          const unsigned cw = *reinterpret_cast<const unsigned*>(tc + vi * 32 + col);
          const int oy = (y0 >> 1) + pr, ox = (x0 >> 1) + X;
          if (oz < Do && oy < Ho && ox < Wo && co_ok) {
            const long long e = ((((long long)n * Do + oz) * Ho + oy) * Wo + ox) * Cout + co;
            *reinterpret_cast<float4*>(y + e) = o4;
            pool_arg[e >> 2] = cw;
            st1[0] += o4.x; st2[0] += o4.x * o4.x; st1[1] += o4.y; st2[1] += o4.y * o4.y;
            st1[2] += o4.z; st2[2] += o4.z * o4.z; st1[3] += o4.w; st2[3] += o4.w * o4.w;
          }
        }
      }
      }   // !PZ
      };
      if (relu_out) pool_block(std::false_type{});
      else pool_block(std::true_type{});
    } else {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float* tile = tile0 + (m & 1) * (32 * CH);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * CH + 32 * t + li] = acc[m][t][r];
      const int gy = y0 + wy + m;
      const bool row_ok = gz < D && gy < H && co_ok;
      const long long rowoff = ((((long long)n * D + gz) * H + gy) * W) * Cout + co;
      float4 v4[32 / VPI], ad[32 / VPI];
#pragma unroll
      for (int k = 0; k < 32 / VPI; ++k) {
        const int xx = vx + VPI * k;
        v4[k] = *reinterpret_cast<const float4*>(tile + xx * CH + col);
        ad[k] = float4{0.f, 0.f, 0.f, 0.f};
        if (addend && row_ok && x0 + xx < W) ad[k] = *reinterpret_cast<const float4*>(addend + rowoff + (long long)(x0 + xx) * Cout);
      }
#pragma unroll
      for (int k = 0; k < 32 / VPI; ++k) {
        const int gx = x0 + vx + VPI * k;
        if (row_ok && gx < W) {
          float4 o;
          o.x = v4[k].x * desc + bv.x + ad[k].x; o.y = v4[k].y * desc + bv.y + ad[k].y;
          o.z = v4[k].z * desc + bv.z + ad[k].z; o.w = v4[k].w * desc + bv.w + ad[k].w;
          if (relu_out) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          *reinterpret_cast<float4*>(y + rowoff + (long long)gx * Cout) = o;
          st1[0] += o.x; st2[0] += o.x * o.x; st1[1] += o.y; st2[1] += o.y * o.y;
          st1[2] += o.z; st2[2] += o.z * o.z; st1[3] += o.w; st2[3] += o.w * o.w;
        }
      }
    }
    }   // !POOL
    if (POOL) stamp();
    if (stats_partial) {
      double d1[4], d2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d1[j] = (double)st1[j]; d2[j] = (double)st2[j];
#pragma unroll
        for (int o = L4; o < 64; o <<= 1) { d1[j] += __shfl_xor(d1[j], o); d2[j] += __shfl_xor(d2[j], o); }
      }
      if (POOL) stamp();
      __syncthreads();                                         // the tiles have been read back
      double* sred = reinterpret_cast<double*>(sEp);           // [wave][CH][2]
      if (lane < L4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sred[((wv * CH) + col + j) * 2] = d1[j]; sred[((wv * CH) + col + j) * 2 + 1] = d2[j]; }
      }
      __syncthreads();
      const int ncol = ZP ? 16 : CH;
      if (tid < 2 * ncol) {
        const int k = tid & 1, c = tid >> 1;
        const int cch = ZP ? c : co0 + c;
        if (cch < Cout) {
          // (the outputs are bit-identical to conv3_fwd_g_kernel's; these sums group them by wave = plane (pair) instead of by
          // (plane, row half), so they agree with that kernel's to fp32 rounding of the per-lane partial sums, not bit for bit)
          double sum = 0.0;
#pragma unroll
          for (int w4 = 0; w4 < 4; ++w4) {
            sum += sred[(w4 * CH + c) * 2 + k];
            if (ZP) sum += sred[(w4 * CH + 16 + c) * 2 + k];   // the second plane of the pair
          }
          stats_partial[(sbrick * Cout + cch) * 2 + k] = sum;
        }
      }
    }
    stamp();                                               // epilogue issued
    if (!more) break;
    cur = nxt;
    cv_in = cv_brick_next;
    pb ^= 1;
  }
  // (KMH_S_UNCOND: the last stage requested a "next stage" that does not exist -- nothing may still be in flight to this wave's
  // registers or to the workgroup's LDS when they are handed to another workgroup)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace
// use_amp per call (common.h): the state is the calling thread's, set for the duration of ONE entry-point call.
static thread_local bool t_amp_call = false;
bool kmh_amp_enabled() { return t_amp_call; }
bool kmh_amp_call_begin(int* terms) {
  const bool prev = t_amp_call;
  t_amp_call = (*terms == 1);
  if (*terms == 1) *terms = 2;
  return prev;
}
void kmh_amp_call_end(bool prev) { t_amp_call = prev; }
namespace {
static inline int cout_pad(int Cout) { return Cout > 64 ? (Cout + 127) & ~127 : (Cout + 63) & ~63; }
static inline bool use_zpair(int Cout) { return Cout <= 16; }

// =============================================================================================
// 3x3x3 convolution over a NEAREST-UPSAMPLED (x2) tensor without the upsampled tensor: the decoder's first convolution
// reads cat(skip, up2(low)); for the `low` channels the 27 taps of an output voxel of parity p = (pz, py, px) fall on
// only 2 x 2 x 2 low-resolution voxels (per axis: parity 0 -> offsets {-1: tap -1; 0: taps 0, +1}, parity 1 ->
// {0: taps -1, 0; +1: tap +1}), so with the taps of one low voxel summed beforehand (pack_weight_up_kernel) every
// output costs 8 multiply-adds per channel instead of 27.  Zero padding is consistent: padded positions -1 / 2L map to
// the low voxels -1 / L, which are outside too.  The kernel writes the low channels' contribution (descaled, no bias
// / ReLU); kmh_conv3d_fwd_bf over the skip channels then adds it in its epilogue (`addend`).
// Workgroup = 32 x 4 x 1 low voxels (-> 64 x 8 x 2 outputs), 8 waves: wave = ((pz, py), 32-cout tile) and holds both
// px parities of 4 rows (8 accumulator tiles); K = 16 = (low tap jx = lane half) x 8 channels; 4 tap pairs per parity.
// Chunks are double-buffered in LDS: the loads of chunk c+1 are in flight during the MFMAs of chunk c.
constexpr int UX = 32, UY = 4;
constexpr int UHX = UX + 2, UHY = UY + 2, UPL = UHX * UHY * 3;      // 612 halo voxels of the low tensor
constexpr int UP_TPB = 512;
constexpr int UP_NST = 32;                                          // 8 parities x 4 tap pairs

template <int TERMS>
__global__ __launch_bounds__(256) void pack_weight_up_kernel(const float* __restrict__ w, __bf16* __restrict__ out,
                                                             int Cout, int Ctot, int cofs, int Cl, int CoutP, int nchunk,
                                                             const float* __restrict__ wscale) {
  const long long total = (long long)nchunk * UP_NST * 2 * CoutP * 8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e & 7);
    long long r = e >> 3;
    const int col = (int)(r % CoutP); r /= CoutP;
    const int h = (int)(r & 1); r >>= 1;
    const int su = (int)(r % UP_NST);
    const int chunk = (int)(r / UP_NST);
    const int ci = chunk * 8 + c, p = su >> 2, st = su & 3;
    const int par[3] = {p >> 2, (p >> 1) & 1, p & 1}, j[3] = {st >> 1, st & 1, h};
    int lo[3], hi[3];                                   // tap range (0..2) of each axis that lands on low offset j
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = par[a] == 0 ? (j[a] == 0 ? 0 : 1) : (j[a] == 0 ? 0 : 2);
      hi[a] = par[a] == 0 ? (j[a] == 0 ? 0 : 2) : (j[a] == 0 ? 1 : 2);
    }
    float v = 0.f;
    if (ci < Cl && col < Cout) {
      const float* wr = w + ((long long)col * Ctot + cofs + ci) * 27;
      for (int kz = lo[0]; kz <= hi[0]; ++kz)
        for (int ky = lo[1]; ky <= hi[1]; ++ky)
          for (int kx = lo[2]; kx <= hi[2]; ++kx) v += wr[kz * 9 + ky * 3 + kx];
    }
    float rem = wscale ? v * wscale[0] : v;
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
      float back;
      const unsigned short hb = to16<TERMS>(rem, back);
      reinterpret_cast<unsigned short*>(out)[((((long long)chunk * TERMS + t) * UP_NST + su) * 2 + h) * CoutP * 8 +
                                             (long long)col * 8 + c] = hb;
      rem -= back;
    }
  }
}

template <int TERMS, bool AMP = false>
__global__ __launch_bounds__(UP_TPB, 2) void conv3_up2_fwd_kernel(
    const float* __restrict__ xl, const float* __restrict__ scale, const float* __restrict__ shift, int Ctot, int cofs,
    const bf16x8* __restrict__ wp, float* __restrict__ y, int Dl, int Hl, int Wl, int Cl, int Cout, int CoutP,
    int tiles_x, int tiles_y, const float* __restrict__ ascale, const float* __restrict__ wscale) {
  __shared__ bf16x8 sIn[2][TERMS][UPL];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n = blockIdx.z;
  const int ncog = (Cout + 63) / 64;
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int cog = item % ncog, brick = item / ncog;
  const int bx = brick % tiles_x, by = (brick / tiles_x) % tiles_y, zl = brick / (tiles_x * tiles_y);
  const int x0 = bx * UX, y0 = by * UY;
  const int pz = (wv >> 1) & 1, py = wv & 1, nt = wv >> 2;
  const int co = cog * 64 + 32 * nt + li;

  f32x16 acc[2][UY];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int m = 0; m < UY; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pl][m][r] = 0.f;
  const float sA = ascale ? ascale[0] : 1.f;
  const float desc = (ascale ? ascale[1] : 1.f) * (wscale ? wscale[1] : 1.f);
  const int nchunk = Cl / KC;

  // staging descriptors: up to 2 halo voxels per thread, the same for every chunk
  constexpr int NV = (UPL + UP_TPB - 1) / UP_TPB;      // 2
  int sv_rel[NV];
  bool sv_in[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * UP_TPB;
    const int lx = v % UHX, ly = (v / UHX) % UHY, lz = v / (UHX * UHY);
    const int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = zl + lz - 1;
    sv_in[i] = (v < UPL) && ((unsigned)gx < (unsigned)Wl) && ((unsigned)gy < (unsigned)Hl) && ((unsigned)gz < (unsigned)Dl);
    sv_rel[i] = sv_in[i] ? ((gz * Hl + gy) * Wl + gx) * Cl : 0;
  }
  const float* xn = xl + (long long)n * Dl * Hl * Wl * Cl;
  float pv[NV][8];
  auto fetch = [&](int ch) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float* p = xn + sv_rel[i] + ch * KC;       // a valid address also for padding voxels (zeroed at commit)
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      pv[i][0] = a.x; pv[i][1] = a.y; pv[i][2] = a.z; pv[i][3] = a.w;
      pv[i][4] = b.x; pv[i][5] = b.y; pv[i][6] = b.z; pv[i][7] = b.w;
    }
  };
  auto commit = [&](int ch, int stage) {
    float csc[8], csh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      csc[j] = (scale ? scale[(long long)n * Ctot + cofs + ch * KC + j] : 1.f) * sA;
      csh[j] = (scale ? shift[(long long)n * Ctot + cofs + ch * KC + j] : 0.f) * sA;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * UP_TPB;
      if (v < UPL) {
        float val[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) val[j] = sv_in[i] ? pv[i][j] * csc[j] + csh[j] : 0.f;   // zero padding AFTER the norm
        bf16x8 parts[TERMS];
        split8<TERMS>(val, parts);
#pragma unroll
        for (int t = 0; t < TERMS; ++t) sIn[stage][t][v] = parts[t];
      }
    }
  };

  const int wbase = (pz * UHY + py) * UHX + li + lh;   // + (jz * UHY + jy + m) * UHX + px: this lane's A voxel
  const int subase = (pz * 4 + py * 2) * 4;            // first step of parity (pz, py, 0)
  fetch(0);
  commit(0, 0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int stage = ch & 1;
    if (ch + 1 < nchunk) fetch(ch + 1);
    const bf16x8* wc = wp + (long long)ch * TERMS * UP_NST * 2 * CoutP + lh * CoutP + co;
    constexpr int BD = 4;                               // B ring depth over the wave's 8 (px, tap pair) steps
    bf16x8 bq[BD][TERMS];
#pragma unroll
    for (int d = 0; d < BD; ++d)
#pragma unroll
      for (int q = 0; q < TERMS; ++q)
        bq[d][q] = wc[((long long)(q * UP_NST + subase + d)) * 2 * CoutP];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                       // k = px * 4 + tap pair
      const int pl = k >> 2, st = k & 3;
      bf16x8 b[TERMS];
#pragma unroll
      for (int q = 0; q < TERMS; ++q) b[q] = bq[k % BD][q];
      if (k + BD < 8) {
#pragma unroll
        for (int q = 0; q < TERMS; ++q)
          bq[k % BD][q] = wc[((long long)(q * UP_NST + subase + k + BD)) * 2 * CoutP];
      }
      const int off = wbase + ((st >> 1) * UHY + (st & 1)) * UHX + pl;
#pragma unroll
      for (int m = 0; m < UY; ++m) {
        bf16x8 a[TERMS];
#pragma unroll
        for (int q = 0; q < TERMS; ++q) a[q] = sIn[stage][q][off + m * UHX];
        if (TERMS == 3) {
          acc[pl][m] = mfma16<TERMS>(a[2], b[0], acc[pl][m]);
          acc[pl][m] = mfma16<TERMS>(a[1], b[1], acc[pl][m]);
          acc[pl][m] = mfma16<TERMS>(a[0], b[2], acc[pl][m]);
        }
        if constexpr (!AMP) {
          acc[pl][m] = mfma16<TERMS>(a[1], b[0], acc[pl][m]);
          acc[pl][m] = mfma16<TERMS>(a[0], b[1], acc[pl][m]);
        }
        acc[pl][m] = mfma16<TERMS>(a[0], b[0], acc[pl][m]);
      }
    }
    if (ch + 1 < nchunk) commit(ch + 1, stage ^ 1);    // the other stage: its readers finished a chunk ago
    __syncthreads();
  }
  // epilogue: one channel per lane in the accumulators -> 16 bytes per lane after a per-wave transposition through the
  // (now idle) fragment images: 32 store instructions per lane instead of 128 (the store path is issue-bound, see
  // conv3_fwd_g_kernel).  Cout % 4 == 0 is guaranteed by the caller (upcat_conv_ok: channels % 8 == 0).
  const int D = 2 * Dl, H = 2 * Hl, W = 2 * Wl;
  const int gz = 2 * zl + pz;
  float* tile = reinterpret_cast<float*>(&sIn[0][0][0]) + wv * (32 * 32);
  const int c4 = lane & 7, vx = lane >> 3;
  const int cq = cog * 64 + 32 * nt + 4 * c4;
  const bool cq_ok = cq < Cout;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int m = 0; m < UY; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[pl][m][r] * desc;
      const bool row_ok = cq_ok && y0 + m < Hl;
      const int gy = 2 * (y0 + m) + py;
      float* yp = y + ((((long long)n * D + gz) * H + gy) * W) * Cout + cq;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int xx = vx + 8 * k, xlw = x0 + xx;
        const float4 v = *reinterpret_cast<const float4*>(tile + xx * 32 + 4 * c4);
        if (row_ok && xlw < Wl) *reinterpret_cast<float4*>(yp + (long long)(2 * xlw + pl) * Cout) = v;
      }
    }
}

// ---- data gradient of the same operator: ds[m][ci] = sum over the 4 x 4 x 4 high-resolution positions u = 2m + t,
// t in {-1, 0, 1, 2} per axis, of Wt[t][co][ci] dz[u][co] -- the sum over a low voxel's 8 children of the gradient with
// respect to the upsampled tensor, computed at LOW resolution with 64 (pre-summed) taps instead of 8 x 27.  Per axis
// t <-> (output parity p, low offset index j) of the forward: -1 <-> (1, 1), 0 <-> (0, 1), 1 <-> (1, 0), 2 <-> (0, 0).
// Workgroup = 16 x 4 x 1 low voxels; LDS = the 34 x 10 x 4 high-resolution halo of dz (8 channels, hi + lo);
// wave = 32-channel tile of ci; K = 16 = (x tap pair) x 8 dz channels; 32 steps per chunk.
constexpr int DUX = 16, DUY = 4;                                           // low brick of the data gradient: 16 x 4 x 1
constexpr int DHX = 2 * DUX + 2, DHY = 2 * DUY + 2, DPL = DHX * DHY * 4;   // 34 x 10 x 4 = 1360 halo voxels of dz
constexpr int DUP_NST = 32;                                                // (tz, ty) x (x tap pair)
constexpr int DUP_TPB = 256;                                               // 4 waves = the 4 channel tiles of 128 ci

template <int TERMS>
__global__ __launch_bounds__(256) void pack_weight_upt_kernel(const float* __restrict__ w, __bf16* __restrict__ out,
                                                              int Cout, int Ctot, int cofs, int Cl, int CiP, int nchunk,
                                                              const float* __restrict__ wscale) {
  const long long total = (long long)nchunk * DUP_NST * 2 * CiP * 8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e & 7);                        // dz channel inside the chunk
    long long r = e >> 3;
    const int col = (int)(r % CiP); r /= CiP;          // input (low) channel
    const int h = (int)(r & 1); r >>= 1;
    const int s = (int)(r % DUP_NST);
    const int chunk = (int)(r / DUP_NST);
    const int co = chunk * 8 + c;
    const int idx[3] = {s >> 3, (s >> 1) & 3, 2 * (s & 1) + h};     // t + 1 per axis (z, y, x)
    int lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int par = (idx[a] + 1) & 1, j = idx[a] <= 1 ? 1 : 0;
      lo[a] = par == 0 ? (j == 0 ? 0 : 1) : (j == 0 ? 0 : 2);
      hi[a] = par == 0 ? (j == 0 ? 0 : 2) : (j == 0 ? 1 : 2);
    }
    float v = 0.f;
    if (co < Cout && col < Cl) {
      const float* wr = w + ((long long)co * Ctot + cofs + col) * 27;
      for (int kz = lo[0]; kz <= hi[0]; ++kz)
        for (int ky = lo[1]; ky <= hi[1]; ++ky)
          for (int kx = lo[2]; kx <= hi[2]; ++kx) v += wr[kz * 9 + ky * 3 + kx];
    }
    float rem = wscale ? v * wscale[0] : v;
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
      float back;
      const unsigned short hb = to16<TERMS>(rem, back);
      reinterpret_cast<unsigned short*>(out)[((((long long)chunk * TERMS + t) * DUP_NST + s) * 2 + h) * CiP * 8 +
                                             (long long)col * 8 + c] = hb;
      rem -= back;
    }
  }
}

template <int TERMS, bool AMP = false>
__global__ __launch_bounds__(DUP_TPB, TERMS == 2 ? 2 : 3) void conv3_up2_dgrad_kernel(
    const float* __restrict__ dz /* (N,2Dl,2Hl,2Wl,Cout) */, const bf16x8* __restrict__ wp,
    float* __restrict__ ds /* (N,Dl,Hl,Wl,Cl) */, int Dl, int Hl, int Wl, int Cl, int CiP, int Cout, int tiles_x,
    int tiles_y, const float* __restrict__ dscale, const float* __restrict__ wscale,
    double* __restrict__ stats_partial /* (N, bricks, Cl, 2) | NULL: per-brick (sum ds, sum ds^2) of every channel */,
    int in_blocked /* dz is channel-blocked (N, Cout/8, 2Dl, 2Hl, 2Wl, 8): a chunk's 32 bytes per voxel are contiguous ACROSS
                      voxels, whole lines per request instead of 32-byte pieces of 64-byte sectors */) {
  // Workgroup = 16 x 4 x 1 low voxels = two M tiles of (16 x, 2 y); wave = one 32-channel tile of ci, both M tiles.
  // 43.5 KB of LDS; two (f16x3, prefetching: 224 registers) or three (bf16x6) workgroups per CU, whose staging and MFMA
  // phases overlap.
  __shared__ bf16x8 sIn[TERMS][DPL];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n = blockIdx.z;
  const int ncig = (Cl + 127) / 128;
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int cig = item % ncig, brick = item / ncig;
  const int bx = brick % tiles_x, by = (brick / tiles_x) % tiles_y, zl = brick / (tiles_x * tiles_y);
  const int x0 = bx * DUX, y0 = by * DUY;
  const int ci = cig * 128 + 32 * wv + li;
  const int D = 2 * Dl, H = 2 * Hl, W = 2 * Wl;

  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const float sD = dscale ? dscale[0] : 1.f;
  const float desc = (dscale ? dscale[1] : 1.f) * (wscale ? wscale[1] : 1.f);
  const int nchunk = (Cout + KC - 1) / KC;
  const float* dn = dz + (long long)n * D * H * W * Cout;
  constexpr int NV = (DPL + DUP_TPB - 1) / DUP_TPB;    // 6
  // row li of an M tile = low voxel (x = li & 15, y = 2 mt + (li >> 4)); its halo origin is (2 y, 2 x)
  const int abase = (2 * (li >> 4)) * DHX + 2 * (li & 15) + lh;

  // the next chunk's halo is fetched into registers under this chunk's MFMAs (staging it at the top of its own chunk left
  // an HBM round trip exposed per chunk and workgroup)
  float4 pre[NV][2];
  auto fetch = [&](int ch) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * DUP_TPB;
      pre[i][0] = pre[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < DPL) {
        const int lx = v % DHX, ly = (v / DHX) % DHY, lz = v / (DHX * DHY);
        const int gx = 2 * x0 - 1 + lx, gy = 2 * y0 - 1 + ly, gz = 2 * zl - 1 + lz;
        if ((unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D) {
          const long long vox = ((long long)gz * H + gy) * W + gx;
          const float* p = in_blocked ? dn + ((long long)ch * D * H * W + vox) * KC : dn + vox * Cout + ch * KC;
          if ((Cout & 3) == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
              if (ch * KC + 4 * q < Cout) pre[i][q] = *reinterpret_cast<const float4*>(p + 4 * q);
          } else {
            float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (ch * KC + j < Cout) t8[j] = p[j];
            pre[i][0] = make_float4(t8[0], t8[1], t8[2], t8[3]);
            pre[i][1] = make_float4(t8[4], t8[5], t8[6], t8[7]);
          }
        }
      }
    }
  };
  constexpr bool PF = TERMS == 2;                       // (the three-term variant has no registers to spare: it fetches in place)
  if (PF) fetch(0);
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();                                    // the previous chunk's readers are done
    if (!PF) fetch(ch);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * DUP_TPB;
      if (v < DPL) {
        const float val[8] = {pre[i][0].x * sD, pre[i][0].y * sD, pre[i][0].z * sD, pre[i][0].w * sD,
                              pre[i][1].x * sD, pre[i][1].y * sD, pre[i][1].z * sD, pre[i][1].w * sD};
        bf16x8 parts[TERMS];
        split8<TERMS>(val, parts);
#pragma unroll
        for (int t = 0; t < TERMS; ++t) sIn[t][v] = parts[t];
      }
    }
    __syncthreads();
    if (PF && ch + 1 < nchunk) fetch(ch + 1);
    const bf16x8* wc = wp + (long long)ch * TERMS * DUP_NST * 2 * CiP + lh * CiP + ci;
    constexpr int BD = 4;
    bf16x8 bq[BD][TERMS];
#pragma unroll
    for (int d = 0; d < BD; ++d)
#pragma unroll
      for (int q = 0; q < TERMS; ++q) bq[d][q] = wc[((long long)(q * DUP_NST + d)) * 2 * CiP];
#pragma unroll 1
    for (int tz = 0; tz < 4; ++tz) {                    // 8 steps per z tap: the ring (depth 4) index stays constant
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int s = tz * 8 + s8;
        bf16x8 b[TERMS];
#pragma unroll
        for (int q = 0; q < TERMS; ++q) b[q] = bq[s8 % BD][q];
        if (s + BD < DUP_NST) {
#pragma unroll
          for (int q = 0; q < TERMS; ++q) bq[s8 % BD][q] = wc[((long long)(q * DUP_NST + s + BD)) * 2 * CiP];
        }
        const int off = abase + (tz * DHY + (s8 >> 1)) * DHX + 2 * (s8 & 1);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          bf16x8 a[TERMS];
#pragma unroll
          for (int q = 0; q < TERMS; ++q) a[q] = sIn[q][off + 4 * m * DHX];
          if (TERMS == 3) {
            acc[m] = mfma16<TERMS>(a[2], b[0], acc[m]);
            acc[m] = mfma16<TERMS>(a[1], b[1], acc[m]);
            acc[m] = mfma16<TERMS>(a[0], b[2], acc[m]);
          }
          if constexpr (!AMP) {
            acc[m] = mfma16<TERMS>(a[1], b[0], acc[m]);
            acc[m] = mfma16<TERMS>(a[0], b[1], acc[m]);
          }
          acc[m] = mfma16<TERMS>(a[0], b[0], acc[m]);
        }
      }
    }
  }
  if (ci >= Cl) return;
  float s1 = 0.f, s2 = 0.f;                             // <= 32 values per lane: fp32, then fp64 per brick
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;  // row of the M tile
      const int gx = x0 + (row & 15), gy = y0 + 2 * m + (row >> 4);
      if (gx < Wl && gy < Hl) {
        const float v = acc[m][r] * desc;
        ds[((((long long)n * Dl + zl) * Hl + gy) * Wl + gx) * Cl + ci] = v;
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
    }
  }
  if (stats_partial) {                                  // the consumer's GroupNorm backward wants sum ds per channel
    double d1 = (double)s1, d2 = (double)s2;
    d1 += __shfl_xor(d1, 32, 64);
    d2 += __shfl_xor(d2, 32, 64);
    if (lh == 0) {
      double* o = stats_partial + (((long long)n * gridDim.x / ncig + brick) * Cl + ci) * 2;
      o[0] = d1; o[1] = d2;
    }
  }
}

// ---- weight gradient of the same operator: C_n (Cl x J) = A_n^T B_n over the low-resolution voxels, A = the normalised
// low tensor (V x Cl), B = the box sums of dz (V x J, J = 27 Cout, norm.hip: up2_boxsum_kernel).  Both operands have the
// reduction index slowest, so both are transposed while they are staged (voxel pairs packed into 32-bit LDS words, like
// the 27-tap weight gradient's images).  Workgroup = 128 x 128 tile of C over one K slab, 32 voxels per step; wave = 64 x 64.
// Bound: a CU streams in ~10 B / cycle (256 CUs: 5.1 TB/s), this tile loads (128 + 128) x 4 B per 2 x 128 x 128 multiply-adds.  A
// 128 x 256 tile on 8 waves (1.33x the intensity) needs 176 registers = ONE workgroup per CU and is no faster (2.97 vs 2.86 ms).
constexpr int GK = 32;                        // voxels per staging step (64: two workgroups per CU, 5 % slower)
constexpr int GPITCH = GK * 2 + 16;           // bytes per LDS row (32 x 2 B + pad: 5 x 16 B, conflict-free b128 reads)
template <int TERMS, bool AMP = false>
__global__ __launch_bounds__(256, 3) void up2_wgrad_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                float* __restrict__ Cp, int V, int Cl, int J, int kslab,
                                                                int ntn, int ntm, const float* __restrict__ ascale,
                                                                const float* __restrict__ bscale,
                                                                const float* __restrict__ a_scale /* (N, Cl) | NULL */,
                                                                const float* __restrict__ a_shift, int xcd) {
  __shared__ __attribute__((aligned(16))) unsigned char sA[TERMS][128 * GPITCH];
  __shared__ __attribute__((aligned(16))) unsigned char sB[TERMS][128 * GPITCH];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n = blockIdx.z;
  // The column tiles of one (row tile, K slab) read the SAME rows of A.  Dealt round-robin over the XCDs, every XCD's L2 fetched
  // them for itself: 14.75 GB per launch for 7.25 GB of G at 64^3 x 128 x 1728 (PMC), and the launch ran at the HBM rate of THAT.
  // With one contiguous item range per XCD (xcd_remap) the tiles of a slab sit on one XCD and walk the slab together: A comes
  // from HBM once.  (KEYMORPH_UP2_GEMM_NO_XCD=1: the round-robin order, for A/B runs.)
  int item = xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int tn = item % ntn; item /= ntn;
  const int tm = item % ntm;
  const int slab = item / ntm;
  const int m0 = tm * 128, n0 = tn * 128;
  const int wm = wv & 1, wn = wv >> 1;
  const float sa = ascale ? ascale[0] : 1.f, sb = bscale ? bscale[0] : 1.f;
  const float desc = (ascale ? ascale[1] : 1.f) * (bscale ? bscale[1] : 1.f);
  const float* An = A + (long long)n * V * Cl;
  const float* Bn = B + (long long)n * V * J;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int k_beg = slab * kslab;
  int k_end = k_beg + kslab;
  if (k_end > V) k_end = V;
  // staging items: (voxel pair kp, column quad cq) -> 2 float4 loads, 4 packed words per term.  Eight consecutive lanes take the
  // eight quads of ONE 128-byte line of a voxel row (round 4; four lanes / 64-byte pieces before: 2.9 TB/s -> see DESIGN.md)
  constexpr int NKP = GK / 2, NIT = GK / 16;   // voxel pairs per step, staging items per thread
  float4 pa[NIT][2], pb[NIT][2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = tid + i * 256, cq = (e & 7) + 8 * (e / (8 * NKP)), kp = (e >> 3) & (NKP - 1);   // lanes: 8 quads (one line) x 8 voxel pairs
      const int k = k0 + 2 * kp;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const int ca = m0 + 4 * cq, cb = n0 + 4 * cq;
      pa[i][0] = (k < k_end && ca < Cl) ? *reinterpret_cast<const float4*>(An + (long long)k * Cl + ca) : z4;
      pa[i][1] = (k + 1 < k_end && ca < Cl) ? *reinterpret_cast<const float4*>(An + (long long)(k + 1) * Cl + ca) : z4;
      pb[i][0] = (k < k_end && cb < J) ? *reinterpret_cast<const float4*>(Bn + (long long)k * J + cb) : z4;
      pb[i][1] = (k + 1 < k_end && cb < J) ? *reinterpret_cast<const float4*>(Bn + (long long)(k + 1) * J + cb) : z4;
    }
  };
  // GroupNorm's per-(sample, channel) affine of the A operand, applied while it is staged (a_scale != NULL): the caller
  // hands over the RAW low tensor and no normalised copy of it is written and read back
  float4 csc[NIT], csh[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + i * 256, cq = (e & 7) + 8 * (e / (8 * NKP)), ca = m0 + 4 * cq;
    csc[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    csh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a_scale && ca < Cl) {
      csc[i] = *reinterpret_cast<const float4*>(a_scale + (long long)n * Cl + ca);
      csh[i] = *reinterpret_cast<const float4*>(a_shift + (long long)n * Cl + ca);
    }
  }
  auto commit = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = tid + i * 256, cq = (e & 7) + 8 * (e / (8 * NKP)), kp = (e >> 3) & (NKP - 1);   // (2-way LDS write conflicts at most)
      const bool v0 = k0 + 2 * kp < k_end, v1 = k0 + 2 * kp + 1 < k_end;     // rows past the slab stay zero (no shift)
      const float a0[4] = {v0 ? fmaf(pa[i][0].x, csc[i].x, csh[i].x) : 0.f, v0 ? fmaf(pa[i][0].y, csc[i].y, csh[i].y) : 0.f,
                           v0 ? fmaf(pa[i][0].z, csc[i].z, csh[i].z) : 0.f, v0 ? fmaf(pa[i][0].w, csc[i].w, csh[i].w) : 0.f};
      const float a1[4] = {v1 ? fmaf(pa[i][1].x, csc[i].x, csh[i].x) : 0.f, v1 ? fmaf(pa[i][1].y, csc[i].y, csh[i].y) : 0.f,
                           v1 ? fmaf(pa[i][1].z, csc[i].z, csh[i].z) : 0.f, v1 ? fmaf(pa[i][1].w, csc[i].w, csh[i].w) : 0.f};
      const float b0[4] = {pb[i][0].x, pb[i][0].y, pb[i][0].z, pb[i][0].w}, b1[4] = {pb[i][1].x, pb[i][1].y, pb[i][1].z, pb[i][1].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned w[TERMS];
        split_pair<TERMS>(a0[j] * sa, a1[j] * sa, w);
#pragma unroll
        for (int t = 0; t < TERMS; ++t) *reinterpret_cast<unsigned*>(sA[t] + (4 * cq + j) * GPITCH + 4 * kp) = w[t];
        split_pair<TERMS>(b0[j] * sb, b1[j] * sb, w);
#pragma unroll
        for (int t = 0; t < TERMS; ++t) *reinterpret_cast<unsigned*>(sB[t] + (4 * cq + j) * GPITCH + 4 * kp) = w[t];
      }
    }
  };
  fetch(k_beg);
  for (int k0 = k_beg; k0 < k_end; k0 += GK) {
    __syncthreads();                           // the previous step's fragment reads are done
    commit(k0);
    __syncthreads();
    if (k0 + GK < k_end) fetch(k0 + GK);       // in flight during the MFMAs
#pragma unroll
    for (int s = 0; s < GK / 16; ++s) {
      bf16x8 a[2][TERMS], b[2][TERMS];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < TERMS; ++t) {
          a[i][t] = *reinterpret_cast<const bf16x8*>(sA[t] + (64 * wm + 32 * i + li) * GPITCH + (16 * s + 8 * lh) * 2);
          b[i][t] = *reinterpret_cast<const bf16x8*>(sB[t] + (64 * wn + 32 * i + li) * GPITCH + (16 * s + 8 * lh) * 2);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (TERMS == 3) {
            acc[i][j] = mfma16<TERMS>(a[i][2], b[j][0], acc[i][j]);
            acc[i][j] = mfma16<TERMS>(a[i][1], b[j][1], acc[i][j]);
            acc[i][j] = mfma16<TERMS>(a[i][0], b[j][2], acc[i][j]);
          }
          if constexpr (!AMP) {
            acc[i][j] = mfma16<TERMS>(a[i][1], b[j][0], acc[i][j]);
            acc[i][j] = mfma16<TERMS>(a[i][0], b[j][1], acc[i][j]);
          }
          acc[i][j] = mfma16<TERMS>(a[i][0], b[j][0], acc[i][j]);
        }
    }
  }
  const int nslab = gridDim.x / (ntn * ntm);
  float* Cn = Cp + ((long long)n * nslab + slab) * Cl * J;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wn + 32 * j + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < Cl && col < J) Cn[(long long)row * J + col] = acc[i][j][r] * desc;
      }
    }
}

// C (N, Cl, J) = sum over the K slabs, fixed order, fp64
__global__ __launch_bounds__(256) void up2_wgrad_reduce_kernel(const float* __restrict__ Cp, int nslab, long long per,
                                                               float* __restrict__ C) {
  const int n = blockIdx.y;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long long)gridDim.x * 256) {
    double s = 0;
    for (int k = 0; k < nslab; ++k) s += Cp[((long long)n * nslab + k) * per + e];
    C[(long long)n * per + e] = (float)s;
  }
}


// ---- round 5: the same product with the box sums formed ON THE FLY (the 27 box-sum tensors -- 4.5 GB written by up2_boxsum and
// read back by the product above, which ran at the HBM rate of that -- never exist).  Reference: autograd of the decoder's
// interpolate(nearest x2) + cat + SingleConv (keymorph/unet3d/buildingblocks.py:471-475, :46-78).
// One K step = a 4 x 4 x 2 tile of low voxels (32).  Workgroup = 128 rows of Cl x (27 taps x 8 couts = 216 columns, 7 MFMA
// tiles) over a slab of K tiles; wave = one 32-row tile x all 7 column tiles (112 accumulators).  Per step: the tile's
// 10 x 10 x 6 window of dz (8 channels: 19 KB, prefetched during the previous step's MFMAs) goes to LDS as fp32, 192 threads form
// the 32 x 27 x 8 box sums from it with the additions of up2_boxsum_tiled_kernel in the same order (bit-identical sums), scale
// them by S_dz / 8, split them and write the B image; the A image (raw low tensor with GroupNorm's affine) as in the kernel above.
// Two workgroups per CU (76.5 KB of LDS each): one's box sums (VALU) run beside the other's MFMAs.
// Measured (profiles/r5r_up2_wgrad_fold.txt, N = 4, dz channel-blocked): 128 -> 64 at 128^3: 5.12 -> 2.95 ms, 256 -> 128 at 64^3:
// 1.91 -> 1.40 ms.  With the box sums AND the MFMAs compiled out a launch still takes 2.08 / 1.03 ms: the kernel is bound by what
// a CU can load (window 19.2 KB + A rows 16 KB per step: 9.2 GB per launch, mostly L2 hits, at ~ 10 B / cycle / CU); the box sums
// add 0.5 ms, the MFMAs 0.25.  Fetching the window before or after the box sums: no difference.
#ifndef WF_DMA                 // 1 = the dz window by LDS-DMA (0: through registers, the A/B arm)
#define WF_DMA 1
#endif
#ifndef KMH_WF_MAP
#define KMH_WF_MAP 1
#endif
#ifndef WF_EARLY_W
#define WF_EARLY_W 1
#endif
constexpr int WF_HX = 10, WF_HY = 10, WF_HZ = 6, WF_VOX = WF_HX * WF_HY * WF_HZ;      // window of a 4 x 4 x 2 low tile
constexpr int WF_NC = 224;                                                           // 216 columns, padded to 7 x 32
// MODE (round 5, last): the kernel is bound by what a CU can load, so two of its workgroups become the two halves of ONE
// 512-thread workgroup that share what they both read: MODE 1 = two cout octets over the same A rows (one A image, two windows
// and B images: 54.4 KB of loads per step instead of 70.4), MODE 2 = two 128-row tiles over the same window and box sums (one
// window and B image, two A images: 51.2 KB, and half the box-sum work).  MODE 0 = the 256-thread kernel, two per CU.
// Measured (profiles/r5y_up2_wgrad_fold_modes.txt): MODE 2 -5 % where it applies (256 -> 128 at 64^3: 1.38 -> 1.31 ms); MODE 1
// +3 % (128 -> 64 at 128^3: 2.79 -> 2.88 ms: 23 % fewer bytes, but one workgroup's phases no longer overlap another's) -- it is
// compiled, tested and selectable (KEYMORPH_UP2_FOLD_MODE=1), not chosen.
template <int MODE>
constexpr int wf_lds_bytes() {
  constexpr int NW = MODE == 1 ? 2 : 1, NA = MODE == 2 ? 2 : 1;
  return NW * (WF_VOX * 2 * 16 + 2 * WF_NC * GPITCH) + NA * (2 * 128 * GPITCH + 1024);
}
template <bool AMP, int MODE>
__global__ __launch_bounds__(MODE ? 512 : 256, MODE ? 1 : 2) void up2_wgrad_fold_kernel(
    const float* __restrict__ xl, const float* __restrict__ dz, float* __restrict__ Cp, int Dl, int Hl, int Wl, int Cl, int Cout,
    int tiles_x, int tiles_y, int ktiles, int tiles_per_slab, int ntm /* grid row tiles */, int nto /* grid column groups */,
    const float* __restrict__ ascale, const float* __restrict__ dscale, const float* __restrict__ a_scale,
    const float* __restrict__ a_shift, int dz_blocked, int xcd, const float* __restrict__ zero16) {
  constexpr int NW = MODE == 1 ? 2 : 1, NA = MODE == 2 ? 2 : 1;          // windows + B images, A images
  constexpr int TPBF = MODE ? 512 : 256;
  constexpr int W_BYTES = WF_VOX * 2 * 16, A_BYTES = 2 * 128 * GPITCH, B_BYTES = 2 * WF_NC * GPITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char wf_lds[];
  unsigned char* sW0 = wf_lds;                                                      // [NW][voxel][2 quads] fp32
  unsigned char* sA0 = sW0 + NW * W_BYTES;                                          // [NA][2 terms][128 rows][GPITCH]
  unsigned char* sB0 = sA0 + NA * A_BYTES;                                          // [NW][2 terms][224 rows][GPITCH]
  float* sC0 = reinterpret_cast<float*>(sB0 + NW * B_BYTES);                        // [NA][2][128]: GroupNorm's affine of the A rows
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int hf = MODE ? tid >> 8 : 0, t8 = tid & 255;                               // the thread's half, its index in it
  const int hw = MODE ? wv >> 2 : 0, wq = wv & 3;                                   // the wave's half, its 32-row tile
  const int n = blockIdx.z;
  int item = xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int tn = item % nto; item /= nto;                 // column group (the groups of one K slab read the same rows of xl)
  const int tm = item % ntm;
  const int slab = item / ntm;
  const int oct_t = MODE == 1 ? 2 * tn + hf : tn, oct_w = MODE == 1 ? 2 * tn + hw : tn;              // cout octet: staged / multiplied
  const int m0_t = (MODE == 2 ? 2 * tm + hf : tm) * 128, m0_w = (MODE == 2 ? 2 * tm + hw : tm) * 128;  // first row: staged / multiplied
  const int iw_t = MODE == 1 ? hf : 0, ia_t = MODE == 2 ? hf : 0;                   // the images this thread stages into
  const int D = 2 * Dl, H = 2 * Hl, W = 2 * Wl;
  const long long Vl = (long long)Dl * Hl * Wl, Vh = (long long)D * H * W;
  const float sa = ascale[0], sb = dscale[0] * 0.125f;                             // box sums: |sum of 8| <= 8 max|dz|
  const float desc = ascale[1] * dscale[1] * 8.f;
  const float* xn = xl + (long long)n * Vl * Cl;
  const float* dn = dz + (long long)n * Vh * Cout;
  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int e = tid; e < NW * 2 * 8 * GPITCH / 4; e += TPBF) {                       // columns 216 .. 223 stay zero
    const int im = e / (2 * 8 * GPITCH / 4), r = e % (2 * 8 * GPITCH / 4), t = r / (8 * GPITCH / 4), w = r % (8 * GPITCH / 4);
    reinterpret_cast<unsigned*>(sB0 + im * B_BYTES + t * WF_NC * GPITCH + 216 * GPITCH)[w] = 0u;
  }
  const int t_beg = slab * tiles_per_slab;
  int t_end = t_beg + tiles_per_slab;
  if (t_end > ktiles) t_end = ktiles;
  // staging: the window (600 voxels x 2 quads = 1200 float4) by LDS-DMA, the A rows (16 voxel pairs x 32 quads) through registers
  constexpr int NIW = MODE == 2 ? 3 : 5;                                            // window elements per thread
  constexpr int NIA = MODE == 1 ? 1 : 2;                                            // A items per thread
  float4 pw[WF_DMA ? 1 : NIW], pa[NIA][2];
  int x0 = 0, y0 = 0, z0 = 0;                                                       // the tile the registers hold
  auto fetch_w = [&](int t) {                             // the window of tile t
    const int bx = t % tiles_x, by = (t / tiles_x) % tiles_y, bz = t / (tiles_x * tiles_y);
    const int wx = 8 * bx - 1, wy = 8 * by - 1, wz = 4 * bz - 1;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* sWt = reinterpret_cast<float4*>(sW0 + iw_t * W_BYTES);
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int e = (MODE == 2 ? tid + i * 512 : t8 + i * 256), q = e & 1, v = e >> 1;
      const int lx = v % WF_HX, ly = (v / WF_HX) % WF_HY, lz = v / (WF_HX * WF_HY);
      const int ux = wx + lx, uy = wy + ly, uz = wz + lz;
      const bool in = e < 2 * WF_VOX && (unsigned)ux < (unsigned)W && (unsigned)uy < (unsigned)H && (unsigned)uz < (unsigned)D;
      const long long vox = in ? ((long long)uz * H + uy) * W + ux : 0;
      const float* src = dz_blocked ? dn + ((long long)oct_t * Vh + vox) * 8 + 4 * q : dn + vox * Cout + 8 * oct_t + 4 * q;
      if constexpr (WF_DMA != 0) {
        // straight into the window (element e = lane-linear: 16 bytes per lane behind a wave-uniform base), no staging
        // registers; voxels outside the volume copy 16 bytes of zeros.  (Past element 1199 a lane must not write: what
        // follows the window in LDS is another image.)
        if (e < 2 * WF_VOX)
          __builtin_amdgcn_global_load_lds((kmh_glb_ptr)(in ? src : zero16), (kmh_lds_ptr)(sWt + (e - lane)), 16, 0, 0);
      } else {
        pw[i] = in ? *reinterpret_cast<const float4*>(src) : z4;
      }
    }
  };
  auto fetch_a = [&](int t) {                             // the A rows of tile t (which becomes the tile the registers hold)
    const int bx = t % tiles_x, by = (t / tiles_x) % tiles_y, bz = t / (tiles_x * tiles_y);
    x0 = 4 * bx; y0 = 4 * by; z0 = 2 * bz;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int e = (MODE == 1 ? tid : t8 + i * 256), cq = (e & 7) + 8 * (e >> 7), kp = (e >> 3) & 15;      // 8 lanes: one 128-byte line of a voxel row
      const int k = 2 * kp, gx = x0 + (k & 3), gy = y0 + ((k >> 2) & 3), gz = z0 + (k >> 4), ca = m0_t + 4 * cq;
      const bool rowok = gy < Hl && gz < Dl && ca < Cl;
      const float* src = xn + (((long long)gz * Hl + gy) * Wl + gx) * Cl + ca;
      pa[i][0] = (rowok && gx < Wl) ? *reinterpret_cast<const float4*>(src) : z4;
      pa[i][1] = (rowok && gx + 1 < Wl) ? *reinterpret_cast<const float4*>(src + Cl) : z4;
    }
  };
  if ((MODE == 2 ? t8 : tid) < 128) {
    const int c = MODE == 2 ? t8 : tid, ca = m0_t + c;
    float* sC = sC0 + ia_t * 256;
    sC[c] = (a_scale && ca < Cl) ? a_scale[(long long)n * Cl + ca] : 1.f;
    sC[128 + c] = (a_scale && ca < Cl) ? a_shift[(long long)n * Cl + ca] : 0.f;
  }
  auto commit = [&]() {                                   // registers -> the window and the A image (tile x0, y0, z0)
    if constexpr (WF_DMA == 0) {
      float4* sWt = reinterpret_cast<float4*>(sW0 + iw_t * W_BYTES);
#pragma unroll
      for (int i = 0; i < NIW; ++i) {
        const int e = (MODE == 2 ? tid + i * 512 : t8 + i * 256);
        if (e < 2 * WF_VOX) sWt[e] = pw[i];
      }
    }
    unsigned char* sA = sA0 + ia_t * A_BYTES;
    const float* sC = sC0 + ia_t * 256;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int e = (MODE == 1 ? tid : t8 + i * 256), cq = (e & 7) + 8 * (e >> 7), kp = (e >> 3) & 15;
      const int k = 2 * kp, gx = x0 + (k & 3), gy = y0 + ((k >> 2) & 3), gz = z0 + (k >> 4);
      const bool rowok = gy < Hl && gz < Dl;
      const bool v0 = rowok && gx < Wl, v1 = rowok && gx + 1 < Wl;              // voxels past the volume: zero rows (no shift)
      const float4 sc = *reinterpret_cast<const float4*>(sC + 4 * cq), sh = *reinterpret_cast<const float4*>(sC + 128 + 4 * cq);
      const float a0[4] = {v0 ? fmaf(pa[i][0].x, sc.x, sh.x) : 0.f, v0 ? fmaf(pa[i][0].y, sc.y, sh.y) : 0.f,
                           v0 ? fmaf(pa[i][0].z, sc.z, sh.z) : 0.f, v0 ? fmaf(pa[i][0].w, sc.w, sh.w) : 0.f};
      const float a1[4] = {v1 ? fmaf(pa[i][1].x, sc.x, sh.x) : 0.f, v1 ? fmaf(pa[i][1].y, sc.y, sh.y) : 0.f,
                           v1 ? fmaf(pa[i][1].z, sc.z, sh.z) : 0.f, v1 ? fmaf(pa[i][1].w, sc.w, sh.w) : 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned w[2];
        split_pair<2>(a0[j] * sa, a1[j] * sa, w);
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<unsigned*>(sA + t * 128 * GPITCH + (4 * cq + j) * GPITCH + 4 * kp) = w[t];
      }
    }
  };
  auto boxes = [&]() {                                    // window -> the B image: thread = (low voxel m, kz, channel quad q)
    const int bt = MODE == 1 ? t8 : tid;                  // (MODE 1: 192 threads of each half; else the first 192 of the workgroup)
    if (bt >= 192) return;
    const float4* sW = reinterpret_cast<const float4*>(sW0 + iw_t * W_BYTES);
    unsigned char* sB = sB0 + iw_t * B_BYTES;
#if KMH_WF_MAP
    // a WAVE = one kz: (q, low voxel m) vary over its lanes.  With kz across the lanes (round 5) three lanes of every quad of
    // lanes wrote the same bank of the B image (72 columns x 80 bytes = 0 mod 128 bytes between the kz groups) and read window
    // planes 32 banks apart: 58 % of the kernel's LDS-active cycles were bank conflicts at 59 % LDS busy
    // (profiles/r6n_lds_by_kernel.txt).  Same sums per (m, kz, q), same order: bit-identical.
    const int q = bt & 1, kz = bt >> 6, m = (bt >> 1) & 31;
#else
    const int q = bt & 1, kz = (bt >> 1) % 3, m = bt / 6;
#endif
    const int lmx = m & 3, lmy = (m >> 2) & 3, lmz = m >> 4;
    float4 Y[3][3];
#pragma unroll
    for (int a = 0; a < 9; ++a) (&Y[0][0])[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    // window index i = u - (2m - 1) in 0..3 per axis; tap k (offset k - 1) sums i in {2 - k, 3 - k}
#pragma unroll
    for (int dzp = 0; dzp < 2; ++dzp) {
      const int lz = 2 * lmz + (2 - kz) + dzp;
#pragma unroll
      for (int iy = 0; iy < 4; ++iy) {
        const float4* row = sW + (((lz * WF_HY + 2 * lmy + iy) * WF_HX + 2 * lmx) * 2 + q);
        const float4 a4[4] = {row[0], row[2], row[4], row[6]};
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
        __builtin_amdgcn_sched_barrier(0);     // (one window row at a time: 32 rows hoisted together spill the accumulators)
      }
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      const float4 y = (&Y[0][0])[a];
      const int col = (kz * 9 + a) * 8 + 4 * q;            // column = tap x 8 + cout within the octet
      unsigned w01[2], w23[2];
      split_pair<2>(y.x * sb, y.y * sb, w01);
      split_pair<2>(y.z * sb, y.w * sb, w23);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned char* base = sB + t * WF_NC * GPITCH + col * GPITCH + 2 * m;
        *reinterpret_cast<unsigned short*>(base) = (unsigned short)(w01[t] & 0xffffu);
        *reinterpret_cast<unsigned short*>(base + GPITCH) = (unsigned short)(w01[t] >> 16);
        *reinterpret_cast<unsigned short*>(base + 2 * GPITCH) = (unsigned short)(w23[t] & 0xffffu);
        *reinterpret_cast<unsigned short*>(base + 3 * GPITCH) = (unsigned short)(w23[t] >> 16);
      }
    }
  };
  if (t_beg < t_end) { fetch_w(t_beg); fetch_a(t_beg); }
  const unsigned char* sAw = sA0 + (MODE == 2 ? hw : 0) * A_BYTES;                  // the images this wave multiplies
  const unsigned char* sBw = sB0 + (MODE == 1 ? hw : 0) * B_BYTES;
  for (int t = t_beg; t < t_end; ++t) {
    __syncthreads();                           // the previous step's fragment reads are done
    commit();
    if (WF_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this lane's pieces of the window are in LDS
    __syncthreads();
    if (!WF_DMA && WF_EARLY_W && t + 1 < t_end) fetch_w(t + 1);      // the next window: in flight during the box sums and the MFMAs
    boxes();
    __syncthreads();
    if (t + 1 < t_end) {                       // the next A rows: during the MFMAs (after the box sums: their registers are free again)
      if (WF_DMA || !WF_EARLY_W) fetch_w(t + 1);
      fetch_a(t + 1);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
        a[q] = *reinterpret_cast<const bf16x8*>(sAw + q * 128 * GPITCH + (32 * wq + li) * GPITCH + (16 * s + 8 * lh) * 2);
#pragma unroll
      for (int j = 0; j < 7; ++j) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
          b[q] = *reinterpret_cast<const bf16x8*>(sBw + q * WF_NC * GPITCH + (32 * j + li) * GPITCH + (16 * s + 8 * lh) * 2);
        if constexpr (!AMP) {
          acc[j] = mfma16<2>(a[1], b[0], acc[j]);
          acc[j] = mfma16<2>(a[0], b[1], acc[j]);
        }
        acc[j] = mfma16<2>(a[0], b[0], acc[j]);
      }
    }
  }
  const int nslab = gridDim.x / (nto * ntm);
  const int J = 27 * Cout;
  float* Cn = Cp + ((long long)n * nslab + slab) * Cl * J;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int col = 32 * j + li;
    const int jj = (col >> 3) * Cout + 8 * oct_w + (col & 7);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0_w + 32 * wq + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < Cl && col < 216) Cn[(long long)row * J + jj] = acc[j][r] * desc;
    }
  }
}

}  // namespace

static int up2_wgrad_slabs(int V, int Cl, int J, int N, int* kslab) {
  const int tiles = ceil_div(Cl, 128) * ceil_div(J, 128) * N;
  int want = 2048 / tiles;                    // ~2048 workgroups
  if (want < 1) want = 1;
  int ks = ceil_div(V, want);
  ks = (ks + GK - 1) / GK * GK;
  *kslab = ks;
  return ceil_div(V, ks);
}

KMH_API size_t kmh_up2_wgrad_gemm_ws_bytes(int N, int V, int Cl, int J) {
  int ks;
  const int ns = up2_wgrad_slabs(V, Cl, J, N, &ks);
  return (size_t)N * ns * Cl * J * sizeof(float);
}

/* C (N, Cl, J) = A^T B per sample: A (N, V, Cl) the normalised low tensor, B (N, V, J) the box sums (kmh_up2_boxsum);
 * Cl % 4 == 0, J % 4 == 0; ascale / bscale = {S, 1/S} range scales of A and B (terms == 2). */
KMH_API int kmh_up2_wgrad_gemm(const float* A, const float* B, float* C, int N, int V, int Cl, int J, int terms,
                               const float* ascale, const float* bscale, const float* a_scale, const float* a_shift,
                               void* ws, void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: the fp16 kernels with hi x hi only (use_amp), for this call
  if ((Cl & 3) || (J & 3) || (terms != 2 && terms != 3) || (terms == 2 && (!ascale || !bscale)) || (!a_scale != !a_shift))
    return -22;
  int ks;
  const int ns = up2_wgrad_slabs(V, Cl, J, N, &ks);
  const int ntn = ceil_div(J, 128), ntm = ceil_div(Cl, 128);
  hipStream_t s = (hipStream_t)stream;
  dim3 g(ntn * ntm * ns, 1, N);
  static const int xcd = getenv("KEYMORPH_UP2_GEMM_NO_XCD") ? 0 : 1;
  if (terms == 2)
    if (kmh_amp_enabled())
      up2_wgrad_gemm_kernel<2, true><<<g, 256, 0, s>>>(A, B, (float*)ws, V, Cl, J, ks, ntn, ntm, ascale, bscale, a_scale, a_shift, xcd);
    else
    up2_wgrad_gemm_kernel<2><<<g, 256, 0, s>>>(A, B, (float*)ws, V, Cl, J, ks, ntn, ntm, ascale, bscale, a_scale, a_shift, xcd);
  else
    up2_wgrad_gemm_kernel<3><<<g, 256, 0, s>>>(A, B, (float*)ws, V, Cl, J, ks, ntn, ntm, ascale, bscale, a_scale, a_shift, xcd);
  const long long per = (long long)Cl * J;
  int nb = ceil_div(per, 256);
  if (nb > 1024) nb = 1024;
  up2_wgrad_reduce_kernel<<<dim3(nb, N), 256, 0, s>>>((const float*)ws, ns, per, C);
  return KMH_LAUNCH_CHECK();
}

// which fold kernel: 2 = two row tiles per workgroup (Cl > 128 with an even tile count), 1 = two cout octets, 0 = the 256-thread one
// (KEYMORPH_UP2_FOLD_MODE=0|1|2 forces one where it applies: A/B runs)
static int up2_fold_mode(int Cl, int Cout) {
  const int ntm = ceil_div(Cl, 128), nto = Cout / 8;
  int mode = (ntm % 2 == 0) ? 2 : 0;      // (MODE 1 measured 3 % SLOWER than two 256-thread workgroups per CU: forced only)
  const char* env = getenv("KEYMORPH_UP2_FOLD_MODE");       // (read per call: the tests switch it)
  if (env) {
    const int want = atoi(env);
    if (want == 0 || (want == 1 && nto % 2 == 0) || (want == 2 && ntm % 2 == 0)) mode = want;
  }
  return mode;
}

static int up2_fold_slabs(int N, int Dl, int Hl, int Wl, int Cl, int Cout, int* tiles_per_slab, int* ktiles) {
  const int kt = ceil_div(Wl, 4) * ceil_div(Hl, 4) * ceil_div(Dl, 2);
  const int mode = up2_fold_mode(Cl, Cout);
  const int per = (Cout / 8) * ceil_div(Cl, 128) * N / (mode ? 2 : 1);       // workgroups per slab
  int want = ceil_div(mode ? 256 : 512, per);               // one 512-thread or two 256-thread workgroups per CU
  if (want < 1) want = 1;
  if (want > kt) want = kt;
  const int tps = ceil_div(kt, want);
  *tiles_per_slab = tps;
  *ktiles = kt;
  return ceil_div(kt, tps);
}

/* 1 if kmh_up2_wgrad_fold takes this configuration (fp16 split, whole cout octets), else 0 */
KMH_API int kmh_up2_wgrad_fold_ok(int Cl, int Cout, int terms) {
  return (terms == 2 && Cl > 0 && (Cl & 3) == 0 && Cout > 0 && (Cout & 7) == 0) ? 1 : 0;
}

static inline size_t up2_fold_slab_bytes(int N, int Dl, int Hl, int Wl, int Cl, int Cout) {
  int tps, kt;
  const int ns = up2_fold_slabs(N, Dl, Hl, Wl, Cl, Cout, &tps, &kt);
  return (((size_t)N * ns * Cl * 27 * Cout * sizeof(float)) + 255) & ~(size_t)255;
}
/* the partial slabs + 256 bytes of zeros (the source of window voxels outside the volume; written by every call on its stream) */
KMH_API size_t kmh_up2_wgrad_fold_ws_bytes(int N, int Dl, int Hl, int Wl, int Cl, int Cout) {
  return up2_fold_slab_bytes(N, Dl, Hl, Wl, Cl, Cout) + 256;
}

template <bool AMP, int MODE>
static int launch_up2_fold(dim3 g, hipStream_t s, const float* xl, const float* dz, float* ws, int Dl, int Hl, int Wl, int Cl, int Cout,
                           int kt, int tps, int ntm, int nto, const float* ascale, const float* dscale, const float* a_scale,
                           const float* a_shift, int dz_blocked, int xcd, const float* zero16) {
  constexpr int lds = wf_lds_bytes<MODE>();
  hipError_t e = hipFuncSetAttribute((const void*)up2_wgrad_fold_kernel<AMP, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  up2_wgrad_fold_kernel<AMP, MODE><<<g, MODE ? 512 : 256, lds, s>>>(xl, dz, ws, Dl, Hl, Wl, Cl, Cout, ceil_div(Wl, 4), ceil_div(Hl, 4),
                                                                  kt, tps, ntm, nto, ascale, dscale, a_scale, a_shift, dz_blocked,
                                                                  xcd, zero16);
  return 0;
}

/* C (N, Cl, 27 Cout) = kmh_up2_wgrad_gemm(xl, kmh_up2_boxsum(dz)) without the box-sum tensor: xl (N, Dl, Hl, Wl, Cl) the raw
 * low tensor (a_scale / a_shift: GroupNorm's affine, or both NULL), dz (N, 2Dl, 2Hl, 2Wl, Cout) or channel-blocked
 * (dz_blocked), ascale / dscale = {S, 1/S} range scales of the normalised low tensor and of dz; terms: 2, or 1 = hi x hi only
 * (use_amp); ws: kmh_up2_wgrad_fold_ws_bytes, 16-byte aligned. */
KMH_API int kmh_up2_wgrad_fold(const float* xl, const float* dz, float* C, int N, int Dl, int Hl, int Wl, int Cl, int Cout,
                               int terms, const float* ascale, const float* dscale, const float* a_scale, const float* a_shift,
                               int dz_blocked, void* ws, void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: hi x hi only (use_amp), for this call
  if (!ws || ((uintptr_t)ws & 15) || !kmh_up2_wgrad_fold_ok(Cl, Cout, terms) || !ascale || !dscale || (!a_scale != !a_shift) || N <= 0 || N > 65535) return -22;
  int tps, kt;
  const int ns = up2_fold_slabs(N, Dl, Hl, Wl, Cl, Cout, &tps, &kt);
  const int mode = up2_fold_mode(Cl, Cout);
  const int nto = (Cout / 8) / (mode == 1 ? 2 : 1), ntm = ceil_div(Cl, 128) / (mode == 2 ? 2 : 1);      // as the grid sees them
  hipStream_t s = (hipStream_t)stream;
  dim3 g(nto * ntm * ns, 1, N);
  static const int xcd = getenv("KEYMORPH_UP2_GEMM_NO_XCD") ? 0 : 1;
  // 256 bytes of zeros behind the slabs: the LDS-DMA source of window voxels outside the volume.  From the caller's workspace,
  // zeroed on the caller's stream: no allocation, no host synchronisation, nothing process-wide (stream capture stays legal).
  const float* zero16 = (const float*)((const char*)ws + up2_fold_slab_bytes(N, Dl, Hl, Wl, Cl, Cout));
  if (hipMemsetAsync((void*)zero16, 0, 256, s) != hipSuccess) return -12;
  const bool amp = kmh_amp_enabled();
  int rc;
#define KMH_FOLD(A, M) launch_up2_fold<A, M>(g, s, xl, dz, (float*)ws, Dl, Hl, Wl, Cl, Cout, kt, tps, ntm, nto, ascale, dscale, \
                                             a_scale, a_shift, dz_blocked, xcd, zero16)
  if (mode == 2) rc = amp ? KMH_FOLD(true, 2) : KMH_FOLD(false, 2);
  else if (mode == 1) rc = amp ? KMH_FOLD(true, 1) : KMH_FOLD(false, 1);
  else rc = amp ? KMH_FOLD(true, 0) : KMH_FOLD(false, 0);
#undef KMH_FOLD
  if (rc) return rc;
  const long long per = (long long)Cl * 27 * Cout;
  int nb = ceil_div(per, 256);
  if (nb > 1024) nb = 1024;
  up2_wgrad_reduce_kernel<<<dim3(nb, N), 256, 0, s>>>((const float*)ws, ns, per, C);
  return KMH_LAUNCH_CHECK();
}

KMH_API size_t kmh_conv3d_up2_dgrad_pack_bytes(int Cout, int Cl, int terms) {
  const int CiP = (Cl + 127) & ~127;
  return (size_t)((Cout + 7) / 8) * terms * DUP_NST * 2 * CiP * 8 * sizeof(__bf16);
}

KMH_API int kmh_conv3d_up2_dgrad_pack_weight(const float* w, void* packed, int Cout, int Ctot, int cofs, int Cl, int terms,
                                             const float* wscale, void* stream) {
  if (cofs < 0 || cofs + Cl > Ctot || (terms != 2 && terms != 3) || (terms == 2 && !wscale)) return -22;
  const int CiP = (Cl + 127) & ~127, nchunk = (Cout + 7) / 8;
  const long long total = (long long)nchunk * DUP_NST * 2 * CiP * 8;
  int nb = ceil_div(total, 256);
  if (nb > 2048) nb = 2048;
  hipStream_t s = (hipStream_t)stream;
  if (terms == 2) pack_weight_upt_kernel<2><<<nb, 256, 0, s>>>(w, (__bf16*)packed, Cout, Ctot, cofs, Cl, CiP, nchunk, wscale);
  else pack_weight_upt_kernel<3><<<nb, 256, 0, s>>>(w, (__bf16*)packed, Cout, Ctot, cofs, Cl, CiP, nchunk, wscale);
  return KMH_LAUNCH_CHECK();
}

/* ds (N,Dl,Hl,Wl,Cl) = for every low voxel, the sum over its 8 children of the gradient of conv3(up2(.), w[:, cofs:cofs+Cl])
 * with respect to the upsampled tensor, from dz (N,2Dl,2Hl,2Wl,Cout) (no ReLU mask operand: dz is already masked). */
KMH_API size_t kmh_conv3d_up2_dgrad_stats_ws_bytes(int N, int Dl, int Hl, int Wl, int Cl) {
  return (size_t)N * ceil_div(Wl, DUX) * ceil_div(Hl, DUY) * Dl * Cl * 2 * sizeof(double);
}
/* stats_out (N,Cl,2) doubles | NULL (then stats_ws may be NULL): per-channel (sum ds, sum ds^2), from the epilogue */
KMH_API int kmh_conv3d_up2_dgrad(const float* dz, const void* packed, float* ds, int N, int Dl, int Hl, int Wl, int Cl,
                                 int Cout, int terms, const float* dscale, const float* wscale, void* stats_ws,
                                 double* stats_out, int in_blocked, void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: the fp16 kernels with hi x hi only (use_amp), for this call
  if ((terms != 2 && terms != 3) || (terms == 2 && (!dscale || !wscale)) || (stats_out && !stats_ws)) return -22;
  if (in_blocked && (Cout & 7)) return -22;               // whole 8-channel chunks
  const int CiP = (Cl + 127) & ~127;
  const int tx = ceil_div(Wl, DUX), ty = ceil_div(Hl, DUY);
  dim3 g(tx * ty * Dl * ceil_div(Cl, 128), 1, N);
  hipStream_t s = (hipStream_t)stream;
  double* sp = stats_out ? (double*)stats_ws : nullptr;
  if (terms == 2)
    if (kmh_amp_enabled())
      conv3_up2_dgrad_kernel<2, true><<<g, DUP_TPB, 0, s>>>(dz, (const bf16x8*)packed, ds, Dl, Hl, Wl, Cl, CiP, Cout, tx, ty, dscale, wscale, sp, in_blocked);
    else
    conv3_up2_dgrad_kernel<2><<<g, DUP_TPB, 0, s>>>(dz, (const bf16x8*)packed, ds, Dl, Hl, Wl, Cl, CiP, Cout, tx, ty, dscale, wscale, sp, in_blocked);
  else
    conv3_up2_dgrad_kernel<3><<<g, DUP_TPB, 0, s>>>(dz, (const bf16x8*)packed, ds, Dl, Hl, Wl, Cl, CiP, Cout, tx, ty, dscale, wscale, sp, in_blocked);
  if (stats_out)
    kmh_stats::final_kernel<<<dim3(ceil_div(Cl * 2, 256 / kWave), N), 256, 0, s>>>(sp, tx * ty * Dl, Cl, stats_out);
  return KMH_LAUNCH_CHECK();
}

KMH_API size_t kmh_conv3d_up2_pack_bytes(int Cout, int Cl, int terms) {
  return (size_t)(Cl / 8) * terms * UP_NST * 2 * cout_pad(Cout) * 8 * sizeof(__bf16);
}

/* w (Cout, Ctot, 3,3,3): the channels [cofs, cofs + Cl) are the upsampled ones; wscale {S, 1/S} must leave room for the
 * sum of 8 taps (the host passes the 27-tap scale / 8). */
KMH_API int kmh_conv3d_up2_pack_weight(const float* w, void* packed, int Cout, int Ctot, int cofs, int Cl, int terms,
                                       const float* wscale, void* stream) {
  if ((Cl & 7) || cofs < 0 || cofs + Cl > Ctot || (terms != 2 && terms != 3) || (terms == 2 && !wscale)) return -22;
  const int CoutP = cout_pad(Cout), nchunk = Cl / 8;
  const long long total = (long long)nchunk * UP_NST * 2 * CoutP * 8;
  int nb = ceil_div(total, 256);
  if (nb > 2048) nb = 2048;
  hipStream_t s = (hipStream_t)stream;
  if (terms == 2) pack_weight_up_kernel<2><<<nb, 256, 0, s>>>(w, (__bf16*)packed, Cout, Ctot, cofs, Cl, CoutP, nchunk, wscale);
  else pack_weight_up_kernel<3><<<nb, 256, 0, s>>>(w, (__bf16*)packed, Cout, Ctot, cofs, Cl, CoutP, nchunk, wscale);
  return KMH_LAUNCH_CHECK();
}

/* y (N, 2Dl, 2Hl, 2Wl, Cout) = conv3(up2_nearest(norm(xl)), w[:, cofs:cofs+Cl]) with norm = the (N, Ctot) GroupNorm
 * coefficients at channel offset cofs -- the contribution of the upsampled half of a decoder's concatenated input
 * (keymorph/unet3d/buildingblocks.py:471-475 + 46-78), to be passed as `addend` to kmh_conv3d_fwd_bf over the skip
 * half.  No bias, no activation. */
KMH_API int kmh_conv3d_up2_fwd(const float* xl, const float* scale, const float* shift, int Ctot, int cofs,
                               const void* packed, float* y, int N, int Dl, int Hl, int Wl, int Cl, int Cout, int terms,
                               const float* ascale, const float* wscale, void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: the fp16 kernels with hi x hi only (use_amp), for this call
  if ((Cl & 7) || (terms != 2 && terms != 3) || (terms == 2 && (!ascale || !wscale))) return -22;
  if ((long long)Dl * Hl * Wl * Cl >= (1ll << 31)) return -22;
  const int CoutP = cout_pad(Cout);
  const int tx = ceil_div(Wl, UX), ty = ceil_div(Hl, UY);
  dim3 g(tx * ty * Dl * ceil_div(Cout, 64), 1, N);
  hipStream_t s = (hipStream_t)stream;
  if (terms == 2)
    if (kmh_amp_enabled())
      conv3_up2_fwd_kernel<2, true><<<g, UP_TPB, 0, s>>>(xl, scale, shift, Ctot, cofs, (const bf16x8*)packed, y, Dl, Hl, Wl, Cl,
                                                Cout, CoutP, tx, ty, ascale, wscale);
    else
    conv3_up2_fwd_kernel<2><<<g, UP_TPB, 0, s>>>(xl, scale, shift, Ctot, cofs, (const bf16x8*)packed, y, Dl, Hl, Wl, Cl,
                                                Cout, CoutP, tx, ty, ascale, wscale);
  else
    conv3_up2_fwd_kernel<3><<<g, UP_TPB, 0, s>>>(xl, scale, shift, Ctot, cofs, (const bf16x8*)packed, y, Dl, Hl, Wl, Cl,
                                                Cout, CoutP, tx, ty, ascale, wscale);
  return KMH_LAUNCH_CHECK();
}

KMH_API size_t kmh_conv3d_pack_bf_bytes(int Cout, int Cin, int transposed, int terms) {
  const int Co = transposed ? Cin : Cout, Ci = transposed ? Cout : Cin;
  return (size_t)((Ci + 7) / 8) * terms * (use_zpair(Co) ? NSTEP_Z : NSTEP) * 2 * cout_pad(Co) * 8 * sizeof(__bf16);
}

KMH_API int kmh_conv3d_pack_weight_bf(const float* w, void* packed, int Cout, int Cin, int transposed, int terms,
                                      const float* wscale, void* stream) {
  const int Co = transposed ? Cin : Cout, Ci = transposed ? Cout : Cin;
  const int nchunk = (Ci + 7) / 8, CoutP = cout_pad(Co), zp = use_zpair(Co);
  const long long total = (long long)nchunk * (zp ? NSTEP_Z : NSTEP) * 2 * CoutP * 8;
  int nb = ceil_div(total, 256);
  if (nb > 2048) nb = 2048;
  hipStream_t s = (hipStream_t)stream;
  if (terms == 2 && !wscale) return -22;
  if (terms == 2) pack_weight_bf_kernel<2><<<nb, 256, 0, s>>>(w, (__bf16*)packed, Cout, Cin, CoutP, nchunk, transposed, zp, wscale);
  else if (terms == 3) pack_weight_bf_kernel<3><<<nb, 256, 0, s>>>(w, (__bf16*)packed, Cout, Cin, CoutP, nchunk, transposed, zp, wscale);
  else return -22;
  return KMH_LAUNCH_CHECK();
}

/* x (N,D,H,W,Cin) -> y (N,D,H,W,Cout); `packed` from kmh_conv3d_pack_weight_bf for the SAME (Cin, Cout) view:
 * forward: pack(w, Cout, Cin, 0); data gradient: pack(w, Cout_w, Cin_w, 1) and call with Cin = Cout_w, Cout = Cin_w */
template <int NT, int TERMS, int MR, bool ZP = false, int ZT = 1>
static int launch_fwd_bf(const float* x, const float* scale, const float* shift, const float* mask, const bf16x8* wp,
                         const float* bias, float* y, int N, int D, int H, int W, int Cin, int Cout, int CoutP,
                         int relu_in, int relu_out, const float* ascale, const float* wscale, double* stats_ws,
                         double* stats_out, hipStream_t s, int in_blocked = 0, const float* addend = nullptr) {
  const int tx = ceil_div(W, TX), ty = ceil_div(H, (ZP ? 4 : 2) * MR), tz = ceil_div(D, 2 * ZT);
  const int typ = ceil_div(ty, 8), tzp = ceil_div(tz, 8);         // (y, z) patches of 8 x 8 bricks
  dim3 g(tx * typ * tzp * 64 * (ZP ? 1 : ceil_div(Cout, 32 * NT)), 1, N);
  conv3_fwd_bf_kernel<NT, TERMS, MR, ZP, ZT><<<g, BF_TPB, 0, s>>>(x, scale, shift, mask, wp, bias, y, D, H, W, Cin, Cout,
                                                             CoutP, relu_in, relu_out, tx, ty, tz, tzp, ascale, wscale,
                                                             stats_out ? stats_ws : nullptr, in_blocked, addend);
  if (stats_out)
    kmh_stats::final_kernel<<<dim3(ceil_div(Cout * 2, 256 / kWave), N), 256, 0, s>>>(stats_ws, tx * ty * tz, Cout,
                                                                                  stats_out);
  return KMH_LAUNCH_CHECK();
}

template <int NT, bool ZP, bool POOL = false>
static int launch_fwd_g(const float* x, const float* scale, const float* shift, const bf16x8* wp, const float* bias,
                        float* y, int N, int D, int H, int W, int Cin, int Cout, int CoutP, int relu_in, int relu_out,
                        const float* ascale, const float* wscale, double* stats_ws, double* stats_out, hipStream_t s,
                        int in_blocked, const float* addend, unsigned* pool_arg = nullptr) {
  hipError_t e = hipFuncSetAttribute((const void*)conv3_fwd_g_kernel<NT, ZP, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     G_LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  const int tx = ceil_div(W, TX), ty = ceil_div(H, GTY), tz = ceil_div(D, GTZ);
  const int typ = ceil_div(ty, 8), tzp = ceil_div(tz, 8);
  const int total = tx * typ * tzp * 64 * (ZP ? 1 : ceil_div(Cout, 32 * NT));      // virtual blocks per sample
  // persistent workgroups, one per CU, over ONE work list of N * total bricks (a multiple of 8 workgroups, so that every
  // id of a workgroup's list falls on its own XCD)
  long long all = (long long)N * total;
  int wgs = 256;
  if (wgs > ((all + 7) / 8) * 8) wgs = (int)(((all + 7) / 8) * 8);
  dim3 g(wgs, 1, 1);
  static long long* trace = nullptr;                       // KMH_G_TRACE=1 (debug): cycle stamps of workgroup 0 to stderr
  static const bool tracing = getenv("KMH_G_TRACE") != nullptr;
  if (tracing && !trace) { if (hipMalloc(&trace, 240 * sizeof(long long)) != hipSuccess) trace = nullptr; }
  if (tracing && trace) (void)hipMemsetAsync(trace, 0, 240 * sizeof(long long), s);
  conv3_fwd_g_kernel<NT, ZP, POOL><<<g, G_TPB, G_LDS_BYTES, s>>>(x, scale, shift, wp, bias, y, D, H, W, Cin, Cout, CoutP,
                                                                 relu_in, relu_out, tx, ty, tz, tzp, ascale, wscale,
                                                                 stats_out ? stats_ws : nullptr, in_blocked, addend, total, N,
                                                                 tracing ? trace : nullptr, pool_arg);
  if (tracing && trace) {
    long long h[240];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "KMH_G_TRACE NT=%d ZP=%d Cin=%d Cout=%d D=%d:", NT, (int)ZP, Cin, Cout, D);
    for (int i = 1; i < 240 && h[i]; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
    fprintf(stderr, "\n");
  }
  if (stats_out)
    kmh_stats::final_kernel<<<dim3(ceil_div(Cout * 2, 256 / kWave), N), 256, 0, s>>>(stats_ws, tx * ty * tz, Cout,
                                                                                  stats_out);
  return KMH_LAUNCH_CHECK();
}

template <int NT, bool ZP = false, bool SPLIT = false, bool POOL = false>
static int launch_fwd_s(const float* x, const float* scale, const float* shift, const bf16x8* wp, const float* bias, float* y,
                        int N, int D, int H, int W, int Cin, int Cout, int CoutP, int relu_in, int relu_out,
                        const float* ascale, const float* wscale, double* stats_ws, double* stats_out, hipStream_t s,
                        int in_blocked, const float* addend, unsigned* pool_arg = nullptr) {
  hipError_t e = hipFuncSetAttribute((const void*)conv3_fwd_s_kernel<NT, ZP, SPLIT, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     S_LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  const int tx = ceil_div(W, TX), ty = ceil_div(H, GTY), tz = ceil_div(D, GTZ);
  const int typ = ceil_div(ty, 8), tzp = ceil_div(tz, 8);
  const int total = tx * typ * tzp * 64 * (ZP ? 1 : ceil_div(Cout, 32 * NT));      // virtual blocks per sample
  long long all = (long long)N * total;
  int wgs = 256;                                          // persistent, one per CU
  if (wgs > ((all + 7) / 8) * 8) wgs = (int)(((all + 7) / 8) * 8);
  static long long* trace = nullptr;                       // KMH_G_TRACE=1 (debug): cycle stamps of workgroup 0 to stderr
  static const bool tracing = getenv("KMH_G_TRACE") != nullptr;
  if (tracing && !trace) { if (hipMalloc(&trace, 240 * sizeof(long long)) != hipSuccess) trace = nullptr; }
  if (tracing && trace) (void)hipMemsetAsync(trace, 0, 240 * sizeof(long long), s);
  if (kmh_amp_enabled()) {
    e = hipFuncSetAttribute((const void*)conv3_fwd_s_kernel<NT, ZP, SPLIT, POOL, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            S_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    conv3_fwd_s_kernel<NT, ZP, SPLIT, POOL, true><<<dim3(wgs), S_TPB, S_LDS_BYTES, s>>>(
        x, scale, shift, wp, bias, y, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, tx, ty, tz, tzp, ascale, wscale,
        stats_out ? stats_ws : nullptr, in_blocked, addend, total, N, tracing ? trace : nullptr, pool_arg);
  } else
  conv3_fwd_s_kernel<NT, ZP, SPLIT, POOL><<<dim3(wgs), S_TPB, S_LDS_BYTES, s>>>(x, scale, shift, wp, bias, y, D, H, W, Cin, Cout, CoutP, relu_in,
                                                               relu_out, tx, ty, tz, tzp, ascale, wscale,
                                                               stats_out ? stats_ws : nullptr, in_blocked, addend, total, N,
                                                               tracing ? trace : nullptr, pool_arg);
  if (tracing && trace) {
    long long h[240];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "KMH_G_TRACE fwd_s NT=%d ZP=%d SPLIT=%d POOL=%d Cin=%d Cout=%d D=%d:", NT, (int)ZP, (int)SPLIT, (int)POOL, Cin, Cout, D);
    for (int i = 1; i < 240 && h[i]; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
    fprintf(stderr, "\n");
  }
  if (stats_out)
    kmh_stats::final_kernel<<<dim3(ceil_div(Cout * 2, 256 / kWave), N), 256, 0, s>>>(stats_ws, tx * ty * tz, Cout,
                                                                                  stats_out);
  return KMH_LAUNCH_CHECK();
}

// the LDS-DMA kernel's preconditions: fp16 split, whole 8-channel chunks, no fused mask operand, and enough bricks to
// give every CU several of them
// 0: never, 1: when the launch has >= 512 bricks (default), 2: whenever the preconditions hold (parity tests force the
// kernel onto small / ragged volumes this way).  Initialised from KEYMORPH_FWD_G, changed by kmh_conv3d_fwd_bf_set_dispatch.
static std::atomic<int> g_fwd_g_mode{-1};
static int fwd_g_mode() {
  int m = g_fwd_g_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    m = getenv("KEYMORPH_FWD_G") ? atoi(getenv("KEYMORPH_FWD_G")) : 1;
    if (m < 0 || m > 2) m = 1;
    g_fwd_g_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}

static bool fwd_g_ok(bool mask, bool addend, int N, int D, int H, int W, int Cin, int Cout, int terms) {
  const int mode = fwd_g_mode();
  if (!mode || terms != 2 || mask || (Cin & 7) || (Cout & 3)) return false;
  if (use_zpair(Cout) && addend) return false;
  if ((long long)D * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31)) return false;          // 32-bit element offsets
  const long long wgs = (long long)N * ceil_div(W, TX) * ceil_div(H, GTY) * ceil_div(D, GTZ) *
                        (use_zpair(Cout) ? 1 : ceil_div(Cout, 64));
  return wgs >= (mode == 2 ? 1 : 512);
}

/* Kernel selection of kmh_conv3d_fwd_bf, settable at run time: mode 0 = conv3_fwd_bf_kernel always, 1 = the LDS-DMA
 * kernel (conv3_fwd_g_kernel) for launches of >= 512 bricks (default), 2 = conv3_fwd_g_kernel whenever its
 * preconditions hold, whatever the size.  Returns the previous mode (-22 for a bad argument). */
KMH_API int kmh_conv3d_fwd_bf_set_dispatch(int mode) {
  if (mode < 0 || mode > 2) return -22;
  const int old = fwd_g_mode();
  g_fwd_g_mode.store(mode, std::memory_order_relaxed);
  return old;
}

/* Which kernel kmh_conv3d_fwd_bf launches for this call under the current dispatch mode:
 * 0 conv3_fwd_bf_kernel, 1 conv3_fwd_g_kernel<1,false>, 2 conv3_fwd_g_kernel<2,false>, 3 conv3_fwd_g_kernel<1,true>
 * (z-paired, Cout <= 16). */
KMH_API int kmh_conv3d_fwd_bf_variant(int N, int D, int H, int W, int Cin, int Cout, int terms, int has_mask,
                                      int has_addend) {
  if (!fwd_g_ok(has_mask != 0, has_addend != 0, N, D, H, W, Cin, Cout, terms)) return 0;
  return use_zpair(Cout) ? 3 : (Cout > 32 ? 2 : 1);
}

/* in_blocked == 2 of kmh_conv3d_fwd_bf: x is the PRE-SPLIT channel-blocked tensor a producer such as kmh_maxpool3d_bwd_split
 * writes -- (N, Cin/8, D*H*W + 1) records of 32 bytes = the 8 fp16 "hi" then the 8 fp16 "lo" terms of fmaf(value, S, 0) with
 * S = ascale[0], record D*H*W of every (sample, chunk) plane all zeros -- so that the kernel copies fragments instead of
 * converting them (conv3_fwd_s_kernel<1, true, true>).  Served: the z-paired tile (Cout <= 16) of the one-wave kernel under its
 * usual preconditions, no scale / shift / mask / relu_in / addend (a gradient operand).  1 = served. */
KMH_API int kmh_conv3d_fwd_bf_split_ok(int N, int D, int H, int W, int Cin, int Cout, int terms) {
  if (!use_zpair(Cout) || Cin > S_COEF) return 0;
  if (((long long)D * H * W + 1) * KC >= (1ll << 31)) return 0;
  // The shape's own preconditions only -- NOT the tunable dispatch threshold (kmh_conv3d_fwd_bf_set_dispatch / KEYMORPH_FWD_G):
  // in_blocked == 2 always launches conv3_fwd_s_kernel<1, true, true>, whatever the mode, and a producer that asked this
  // question at forward time must get the same answer at backward time.
  if (N <= 0 || terms != 2 || (Cin & 7) || (Cout & 3)) return 0;
  if ((long long)D * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31)) return 0;          // 32-bit element offsets
  return 1;
}

/* Convolution + ReLU + MaxPool3d(2) in one launch, for an encoder block whose output feeds ONLY the next level's pooling
 * (keymorph/unet3d/buildingblocks.py:46-78 then :321-380 `self.pooling(x)`): yp (N, D/2, H/2, W/2, Cout) = the pooled
 * output, arg (same shape, 1 byte per element) = the winners' window indices exactly as kmh_maxpool3d_fwd records them
 * (first maximum in z, y, x order), stats_out (N, Cout, 2) = (sum, sum^2) of the POOLED tensor; the full-resolution
 * output is never written.  Same arguments otherwise as kmh_conv3d_fwd_bf (no mask, bias, addend).
 * kmh_conv3d_fwd_bf_pool_ok says whether a shape is served (split-fp16 mode, whole 8-channel input chunks, 16 < Cout <= 32,
 * and the LDS-DMA kernel selected by the current dispatch mode). */
KMH_API int kmh_conv3d_fwd_bf_pool_ok(int N, int D, int H, int W, int Cin, int Cout, int terms) {
  return (Cout > 16 && Cout <= 32 && D >= 2 && H >= 2 && W >= 2 && fwd_g_ok(false, false, N, D, H, W, Cin, Cout, terms)) ? 1 : 0;
}

KMH_API int kmh_conv3d_fwd_bf_pool(const float* x, const float* scale, const float* shift, const void* packed, float* yp,
                                   unsigned char* arg, int N, int D, int H, int W, int Cin, int Cout, int relu_in,
                                   int terms, const float* ascale, const float* wscale, void* stats_ws, double* stats_out,
                                   int in_blocked, void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: the fp16 kernels with hi x hi only (use_amp), for this call
  if (!kmh_conv3d_fwd_bf_pool_ok(N, D, H, W, Cin, Cout, terms) || !ascale || !wscale || !yp || !arg) return -22;
  if (((uintptr_t)yp & 15) || ((uintptr_t)arg & 3)) return -22;
  // the one-wave kernel's pooling epilogue (KEYMORPH_FWD_S=0 or KEYMORPH_POOL_G=1: the eight-wave kernel's, the A/B arm;
  // same pooled values and winners)
  static const bool pool_g = (getenv("KEYMORPH_FWD_S") && atoi(getenv("KEYMORPH_FWD_S")) == 0) || getenv("KEYMORPH_POOL_G") != nullptr;
  if (!pool_g && Cin <= S_COEF)
    return launch_fwd_s<1, false, false, true>(x, scale, shift, (const bf16x8*)packed, nullptr, yp, N, D, H, W, Cin, Cout,
                                               cout_pad(Cout), relu_in, 1, ascale, wscale, (double*)stats_ws, stats_out,
                                               (hipStream_t)stream, in_blocked, nullptr, (unsigned*)arg);
  return launch_fwd_g<1, false, true>(x, scale, shift, (const bf16x8*)packed, nullptr, yp, N, D, H, W, Cin, Cout,
                                      cout_pad(Cout), relu_in, 1, ascale, wscale, (double*)stats_ws, stats_out,
                                      (hipStream_t)stream, in_blocked, nullptr, (unsigned*)arg);
}

static inline int fwd_bf_rows(int Cout, int rows_per_wave) {   // smallest brick height in y of the variants that may run
  (void)rows_per_wave;
  return use_zpair(Cout) ? 8 : 4;
}

/* x (N,D,H,W,Cin) -> y (N,D,H,W,Cout); `packed` from kmh_conv3d_pack_weight_bf for the SAME (Cin, Cout) view:
 * forward: pack(w, Cout, Cin, 0); data gradient: pack(w, Cout_w, Cin_w, 1) and call with Cin = Cout_w, Cout = Cin_w.
 * rows_per_wave: 4 (32x8x2 brick) or 2 (32x4x2 brick, higher occupancy); 0 = library default. */
KMH_API size_t kmh_conv3d_fwd_bf_stats_ws_bytes(int N, int D, int H, int W, int Cout, int rows_per_wave) {
  return (size_t)N * ceil_div(W, TX) * ceil_div(H, fwd_bf_rows(Cout, rows_per_wave)) * ceil_div(D, TZ) * Cout * 2 *
         sizeof(double);
}

/* in_blocked != 0: x is stored channel-blocked, (N, Cin/8, D, H, W, 8) -- the 8 channels of a chunk of one voxel are
 * one 32-byte record and a chunk's voxels are contiguous, so the loader uses whole cache lines instead of a quarter of
 * each (Cin % 8 == 0, no mask; results are bit-identical; 8-13 % faster on the data-gradient launches).
 * stats_out (N,Cout,2) doubles | NULL: per-channel (sum y, sum y^2) of the OUTPUT, accumulated in the epilogue (what
 * kmh_channel_stats(y) would return: the next layer's GroupNorm statistics without another pass over y);
 * stats_ws: kmh_conv3d_fwd_bf_stats_ws_bytes. */
KMH_API int kmh_conv3d_fwd_bf(const float* x, const float* scale, const float* shift, const float* mask,
                              const void* packed, const float* bias, float* y, int N, int D, int H, int W, int Cin,
                              int Cout, int relu_in, int relu_out, int terms, int rows_per_wave, const float* ascale,
                              const float* wscale, void* stats_ws, double* stats_out, int in_blocked,
                              const float* addend, void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: the fp16 kernels with hi x hi only (use_amp), for this call
  const int CoutP = cout_pad(Cout);
  hipStream_t s = (hipStream_t)stream;
  const bf16x8* wp = (const bf16x8*)packed;
  const int mr = rows_per_wave == 4 ? 4 : 2;
#define KMH_BF_CALL(NT_, T_, MR_) \
  return launch_fwd_bf<NT_, T_, MR_>(x, scale, shift, mask, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked, addend)
  if (terms != 2 && terms != 3) return -22;
  if (addend && use_zpair(Cout)) return -22;
  if (in_blocked && ((Cin & 7) || mask)) return -22;
  if (in_blocked == 2) {      // pre-split input: see kmh_conv3d_fwd_bf_split_ok
    if (!kmh_conv3d_fwd_bf_split_ok(N, D, H, W, Cin, Cout, terms) || scale || shift || relu_in || addend || !ascale || !wscale)
      return -22;
    return launch_fwd_s<1, true, true>(x, nullptr, nullptr, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, 0, relu_out, ascale, wscale,
                                       (double*)stats_ws, stats_out, s, 2, nullptr);
  }
  if (terms == 2 && (!ascale || !wscale)) return -22;       // fp16 split without range scaling is not accurate
  // deep (32 x 8 x 4) bricks for the z-paired (Cout <= 16) launches on big volumes: less halo traffic, twice the B
  // reuse: +5 % on the 256^3 32->16 data gradient.  (The NT = 1, 4-rows-per-wave variant spills with 128 accumulator
  // registers plus 8 staging descriptors and is 4 % slower: not instantiated.)
  static const bool no_deep = getenv("KEYMORPH_FWD_NO_DEEP") != nullptr;     // A/B measurements only
  const bool deep = !no_deep && terms == 2 && Cout <= 32 && D >= 16 && (long long)D * H * W >= (1ll << 21);
  if (fwd_g_ok(mask != nullptr, addend != nullptr, N, D, H, W, Cin, Cout, terms)) {
    // the one-wave-per-SIMD kernel takes the 64-wide tile and the plain 32-wide one (KEYMORPH_FWD_S=1: only the 64-wide; 3: the
    // z-paired tile too -- bit-identical, 3.7 % faster alone at 2 x 256^3 and flat inside the step, so not the default;
    // 0: conv3_fwd_g_kernel for all of them, the A/B arm)
    static const int fwd_s = getenv("KEYMORPH_FWD_S") ? atoi(getenv("KEYMORPH_FWD_S")) : 2;
    if (fwd_s >= 3 && Cin <= S_COEF && use_zpair(Cout) && !addend)
      return launch_fwd_s<1, true>(x, scale, shift, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked, addend);
    if (use_zpair(Cout)) return launch_fwd_g<1, true>(x, scale, shift, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked, addend);
    if (fwd_s && Cin <= S_COEF && !use_zpair(Cout)) {
      if (Cout > 32) return launch_fwd_s<2>(x, scale, shift, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked, addend);
      if (fwd_s >= 2) return launch_fwd_s<1>(x, scale, shift, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked, addend);
    }
    if (Cout > 32) return launch_fwd_g<2, false>(x, scale, shift, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked, addend);
    return launch_fwd_g<1, false>(x, scale, shift, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked, addend);
  }
  if (use_zpair(Cout)) {   // weights were packed z-paired by kmh_conv3d_pack_weight_bf for this Cout
    if (terms == 2 && deep) return launch_fwd_bf<1, 2, 2, true, 2>(x, scale, shift, mask, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked);
    if (terms == 2) return launch_fwd_bf<1, 2, 2, true>(x, scale, shift, mask, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked);
    return launch_fwd_bf<1, 3, 2, true>(x, scale, shift, mask, wp, bias, y, N, D, H, W, Cin, Cout, CoutP, relu_in, relu_out, ascale, wscale, (double*)stats_ws, stats_out, s, in_blocked);
  }
  // Small volumes (the 32^3 level): the 32x8x2-brick grid has only ~512 workgroups for 512 slots, so half-height
  // bricks (twice the workgroups) run 1.9x faster there; with Cout % 128 == 0 the 128-wide N tile (NT = 4) adds
  // up to 10 % (the halo is staged once for twice the output channels).  Large grids prefer the tall bricks.
  const long long wgs4 = (long long)N * ceil_div(W, TX) * ceil_div(H, 8) * ceil_div(D, 2) * ceil_div(Cout, 64);
  const bool small_grid = wgs4 < 2048;
  if (small_grid && terms == 2 && Cout % 128 == 0) KMH_BF_CALL(4, 2, 2);
  // 64 < Cout <= 96 (the 32 -> 96 data gradient at full resolution): one 96-wide N tile on half-height bricks instead
  // of two 64-wide channel groups, the second of them half empty and both staging the same halo
  // (measured for Cout = 192 / 384 as 2 / 4 groups of 96: 9 % slower than 64-wide groups on the tall bricks)
  if (terms == 2 && Cout > 64 && Cout <= 96) KMH_BF_CALL(3, 2, 2);
  if (Cout > 32) {
    if (terms == 2) { if (mr == 4 && !small_grid) KMH_BF_CALL(2, 2, 4); else KMH_BF_CALL(2, 2, 2); }
    else { if (mr == 4) KMH_BF_CALL(2, 3, 4); else KMH_BF_CALL(2, 3, 2); }
  } else {
    if (terms == 2) { if (mr == 4) KMH_BF_CALL(1, 2, 4); else KMH_BF_CALL(1, 2, 2); }
    else { if (mr == 4) KMH_BF_CALL(1, 3, 4); else KMH_BF_CALL(1, 3, 2); }
  }
#undef KMH_BF_CALL
}

#endif   // !KMH_TU_WGRAD
#if KMH_TU_WGRAD
// =============================================================================================
// Split-bf16 weight gradient: dW[tap][ci][co] = sum_v xn[v + tap][ci] * dz[v][co].
// K of the MFMA = 16 consecutive voxels of one brick row, so BOTH operands need, per lane, 8 consecutive
// voxels of ONE channel: the brick is transposed while it is staged into channel-major bf16 LDS images
//   sXT[term][ci (+1 zero plane)][halo row][24]   plane pitch 1168 B (= 73 x 16 B: lanes = channels hit 16
//   sDT[term][co][128 voxels]                     plane pitch  272 B (= 17 x 16 B)   distinct 16-B slots)
// A fragment = aligned ds_read_b128 + ds_read_b32 around the window, then a funnel shift by the tap's x
// offset (0 / 2 / 4 bytes: v_alignbyte for kx = 1, register renaming for kx = 2); B fragment = one aligned
// ds_read_b128.  M rows are packed (tap, ci) with ci tiles of <= 16 channels (2 taps per 32-row tile),
// tiles dealt to the 8 waves exactly like the fp32 kernel; per-wave partial slabs, deterministic reduce.
namespace {

constexpr int WX = 16, WY = 4, WZ = 2;
constexpr int WHY = WY + 2, WHZ = WZ + 2;
constexpr int XROWS = WHY * WHZ;           // 24 halo rows
constexpr int XPITCH = 24;                 // elements per halo row (18 used)
constexpr int XPLANE = 1168;               // bytes per channel plane (24 rows x 48 B = 1152, padded)
constexpr int DPLANE = 272;                // bytes per cout plane (128 voxels x 2 B = 256, padded)
constexpr int WV = WX * WY * WZ;           // 128
// Round 6, the wave-specialised kernel: the x image is a RING of z planes.  Bricks are walked z-fastest, so a brick's 4-plane
// halo window shares 2 planes with its predecessor's: only the 2 new planes (12 of 24 halo rows) are fetched, normalised and
// split per brick -- the operand was 3.4 x redundant (432 halo voxels per 128-voxel brick), now 1.7 x.  8 ring planes: 4 being
// multiplied, up to 4 being filled (the first brick of a z column fills all 4 and starts 4 planes further on, so it never
// touches what the previous column's last brick is still being read from).
constexpr int RZ = 8;                                  // ring planes
// Plane pitch of the ring image: 2312 bytes = 578 words, i.e. 2 mod 32.  The consumers' A fragments are five dwords per lane that
// the compiler reads with 4-byte instructions (ds_read2_b32: it scalarises a 16-byte load whose dwords feed the funnel shifts one by
// one), whose 32-lane groups are 16 channels x 2 taps: with the pitch at 4 mod 32 words (2320 B = 145 x 16 B, chosen in round 2 for
// 16-byte reads that the compiler never emitted) 32 lanes hit 8 banks -- 57 % of the LDS-active cycles were bank conflicts and the LDS
// was 86 % busy (profiles/r6i_sq_counters_wgrad_ring.txt).  At 2 mod 32 the channels take 16 distinct banks and the producers'
// 4-byte stores (channel quads 8 banks apart) stay conflict-free: -6 ... -11 % on every launch; 6, 10, 18 mod 32 the same, an ODD
// pitch twice as slow (profiles/r6k_wgrad_ring_pitch_ab.txt).  KMH_WG_RPAD (A/B builds): bytes added to the 2304 of the rows.
#ifndef KMH_WG_RPAD
#define KMH_WG_RPAD 8
#endif
#ifndef KMH_WG_RROW
#define KMH_WG_RROW 24          // elements (2 bytes) per halo row of the ring image (18 used; a lane reads 5 dwords from byte 0 or 16)
#endif
#ifndef KMH_WG_RZPAD
#define KMH_WG_RZPAD 0          // bytes added to a ring plane
#endif
constexpr int XPITCH_R = KMH_WG_RROW;
constexpr int ZSLOT = WHY * XPITCH_R * 2 + KMH_WG_RZPAD;   // bytes per ring plane inside a channel plane
constexpr int XPLANE_R = RZ * ZSLOT + KMH_WG_RPAD;
static_assert(XPITCH_R * 2 >= 36 && (XPLANE_R & 3) == 0 && (ZSLOT & 3) == 0, "a row holds 18 elements; dword-aligned planes");
constexpr int WGB_TPB = 512;
constexpr int MTWB = 2;                    // M tiles per wave (14 tiles over 8 waves)

__device__ __forceinline__ unsigned pack2(__bf16 lo, __bf16 hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
}

// ---- pieces shared by the two weight-gradient kernels -----------------------------------------------------------
// Tile dealing.  M-tile m = (kx group, slot): all 32 rows of a tile share the tap's x offset kx = m / TPK (the funnel
// shift is then wave-uniform); within the kx group the 9 (kz, ky) taps are packed TPT = 32 / CP per tile.  The three
// kx tiles of one slot read the SAME 20 bytes per lane and differ only in the shift, so with 8 tile groups and 5
// slots (CP = 16: 15 tiles) waves 0-4 take (slot w, kx 0) and (slot w, kx 1) -- one LDS read feeds both fragments --
// and waves 5-7 share out the five kx = 2 tiles.  Other shapes: round robin.
struct WgradTiles {
  int TPT, TPK, tile[MTWB], abase[MTWB], akx[MTWB];
  int akz[MTWB];                                  // ring layout: the lane's tap kz (abase then holds no z term)
  bool share_a;                                   // wave-uniform: tile 1 reuses tile 0's LDS words
};
__device__ __forceinline__ WgradTiles wgrad_deal_tiles(int CP, int MT, int TG, int tg, int li, int lh, bool ring = false) {
  WgradTiles w;
  w.TPT = 32 / CP;
  w.TPK = (9 + w.TPT - 1) / w.TPT;
  const bool paired = (TG == 8 && MT == 15);
  w.share_a = paired && tg < 5;
#pragma unroll
  for (int j = 0; j < MTWB; ++j) {
    int m = tg + TG * j;
    if (paired) {
      if (tg < 5) m = j * w.TPK + tg;                        // (kx = j, slot = tg)
      else { const int k = (tg - 5) * 2 + j; m = k < 5 ? 2 * w.TPK + k : MT; }   // (kx = 2, slot = k); k = 5: none
    }
    w.tile[j] = m;
    const int kx = m / w.TPK, slot = m - kx * w.TPK;
    const int t9 = slot * w.TPT + li / CP, c = li % CP;
    const bool valid = (m < MT) && (t9 < 9);
    const int kz = t9 / 3, ky = t9 % 3;
    w.abase[j] = valid ? (c * XPLANE + (kz * WHY + ky) * (XPITCH * 2) + 16 * lh) : (CP * XPLANE + 16 * lh);
    w.akz[j] = 0;
    if (ring) {      // the z offset is added per brick: ((ring base + row plane + kz) mod RZ) planes (the zero plane: any)
      w.abase[j] = valid ? (c * XPLANE_R + ky * (XPITCH_R * 2) + 16 * lh) : (CP * XPLANE_R + 16 * lh);
      w.akz[j] = valid ? kz : 0;
    }
    w.akx[j] = __builtin_amdgcn_readfirstlane(m < MT ? kx : 0);
  }
  return w;
}

// MFMA phase of one brick: one K16 step per brick row (z, y), this wave's k-split share of the rows
// MODE (wave-uniform, fixed for the life of the wave) specialises the funnel shift of the A fragments:
//   0  generic: any kx per tile, branch-free selects (8 VALU per fragment)
//   1  the paired dealing's waves 0-4: tile 0 is kx = 0 (the words as read), tile 1 is kx = 1 of the SAME words
//      (4 alignbyte); one LDS read feeds both
//   2  the paired dealing's waves 5-7: both tiles are kx = 2, a pure register renaming (no VALU)
template <int NT, int TERMS, int MODE = 0, bool AMP = false, bool RING = false>
__device__ __forceinline__ void wgrad_mfma_brick(const unsigned char* sXT, const unsigned char* sDT, int xt_bytes,
                                                 const WgradTiles& w, int ks, int KS, int li, int lh,
                                                 f32x16 (&acc)[MTWB][NT], int ring_base = 0) {
  constexpr int CO = 32 * NT;
  // the paired dealing implies CP = 16 and no k-split (KS = 1): the row loop is unrolled and every LDS offset but the
  // per-lane base is an instruction immediate
  if (MODE != 0) xt_bytes = 17 * (RING ? XPLANE_R : XPLANE);
  const unsigned char* sDTl = sDT + li * DPLANE + 16 * lh;
  // RING: the byte offset of ring plane (base + zz + kz) mod RZ, per tile and output plane zz of the brick (per lane: kz is)
  int zo[MTWB][WZ];
#pragma unroll
  for (int j = 0; j < MTWB; ++j)
#pragma unroll
    for (int z = 0; z < WZ; ++z) zo[j][z] = RING ? ((ring_base + z + w.akz[j]) & (RZ - 1)) * ZSLOT : 0;
  auto one_row = [&](int row) {
    const int zz = row / WY, yy = row - zz * WY;
    const int arow = RING ? yy * (XPITCH_R * 2) : (zz * WHY + yy) * (XPITCH * 2);
    const int brow = row * WX * 2;
    bf16x8 b[NT][TERMS];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < TERMS; ++q)
        b[t][q] = *reinterpret_cast<const bf16x8*>(sDTl + (q * CO + 32 * t) * DPLANE + brow);
    bf16x8 a[MTWB][TERMS];
    uint4 wq[TERMS];
    unsigned w4q[TERMS];
#pragma unroll
    for (int j = 0; j < MTWB; ++j) {
#pragma unroll
      for (int q = 0; q < TERMS; ++q) {
        if (MODE == 1 ? j == 0 : (MODE == 2 || j == 0 || !w.share_a)) {
          const unsigned char* p = sXT + q * xt_bytes + w.abase[j] + (RING ? zo[j][zz] : 0) + arow;
          if constexpr (RING) {
            // five dwords, read AS dwords (the ring's plane pitch is a multiple of 8, not of 16 bytes: see XPLANE_R)
            const unsigned* p32 = reinterpret_cast<const unsigned*>(p);
            wq[q] = make_uint4(p32[0], p32[1], p32[2], p32[3]);
            w4q[q] = p32[4];
          } else {
            wq[q] = *reinterpret_cast<const uint4*>(p);
            w4q[q] = *reinterpret_cast<const unsigned*>(p + 16);
          }
        }
        const uint4 v = wq[q];
        const unsigned v4 = w4q[q];
        if (MODE == 1) {
          uint4 r = v;
          if (j == 1) {
            r.x = __builtin_amdgcn_alignbyte(v.y, v.x, 2u);
            r.y = __builtin_amdgcn_alignbyte(v.z, v.y, 2u);
            r.z = __builtin_amdgcn_alignbyte(v.w, v.z, 2u);
            r.w = __builtin_amdgcn_alignbyte(v4, v.w, 2u);
          }
          a[j][q] = __builtin_bit_cast(bf16x8, r);
          continue;
        }
        if (MODE == 2) {
          uint4 r;
          r.x = v.y; r.y = v.z; r.z = v.w; r.w = v4;
          a[j][q] = __builtin_bit_cast(bf16x8, r);
          continue;
        }
        // branch-free funnel shift by the tile's (wave-uniform) tap x offset kx in {0, 1, 2} elements: kx = 2
        // selects the next dword as source, kx = 1 shifts by two bytes -- no control flow between the LDS reads,
        // so all fragment loads of a row are in flight together
        const bool k2 = w.akx[j] == 2;
        const unsigned sh = w.akx[j] == 1 ? 2u : 0u;
        uint4 r;
        r.x = __builtin_amdgcn_alignbyte(v.y, k2 ? v.y : v.x, sh);
        r.y = __builtin_amdgcn_alignbyte(v.z, k2 ? v.z : v.y, sh);
        r.z = __builtin_amdgcn_alignbyte(v.w, k2 ? v.w : v.z, sh);
        r.w = __builtin_amdgcn_alignbyte(v4, k2 ? v4 : v.w, sh);
        a[j][q] = __builtin_bit_cast(bf16x8, r);
      }
    }
#pragma unroll
    for (int j = 0; j < MTWB; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TERMS == 3) {
          acc[j][t] = mfma16<TERMS>(a[j][2], b[t][0], acc[j][t]);
          acc[j][t] = mfma16<TERMS>(a[j][1], b[t][1], acc[j][t]);
          acc[j][t] = mfma16<TERMS>(a[j][0], b[t][2], acc[j][t]);
        }
        if constexpr (!AMP) {
          acc[j][t] = mfma16<TERMS>(a[j][1], b[t][0], acc[j][t]);
          acc[j][t] = mfma16<TERMS>(a[j][0], b[t][1], acc[j][t]);
        }
        acc[j][t] = mfma16<TERMS>(a[j][0], b[t][0], acc[j][t]);
      }
  };
  // (Round 6, measured and removed: the fragments of row r + 1 read into a second register set before row r's MFMAs -- the
  // compiler's own order is "rrLM rrrrLM ...", 33 lgkmcnt waits per 48 MFMAs; with the read-ahead 14, all counted (lgkmcnt(10..12))
  // -- changed no launch: 3.435 / 3.400 ms with / without at 16 -> 32, 2 x 256^3; N = 64 spills.  profiles/r6d_wgrad_readahead_ab.txt)
  if (MODE != 0) {
#pragma unroll
    for (int row = 0; row < WY * WZ; ++row) one_row(row);
  } else {
    for (int row = ks; row < WY * WZ; row += KS) one_row(row);
  }
}

// this wave's accumulators -> its partial slab (tap, ci, co)
template <int NT>
__device__ __forceinline__ void wgrad_store_partial(float* out, const WgradTiles& w, int MT, int CP, int ci0, int co0,
                                                    int Cin, int Cout, int li, int lh, const f32x16 (&acc)[MTWB][NT]) {
#pragma unroll
  for (int j = 0; j < MTWB; ++j) {
    const int m = w.tile[j];
    if (m >= MT) continue;
    const int kx = m / w.TPK, slot = m - kx * w.TPK;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = co0 + 32 * t + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int t9 = slot * w.TPT + rr / CP, c = ci0 + rr % CP;
        const int tap = t9 * 3 + kx;             // (kz*3 + ky)*3 + kx
        if (t9 < 9 && c < Cin && co < Cout) out[((long long)tap * Cin + c) * Cout + co] = acc[j][t][r];
      }
    }
  }
}

template <int NT, int TERMS>
__global__ __launch_bounds__(WGB_TPB, 2) void conv3_wgrad_bf_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ dz, const float* __restrict__ dzmask, float* __restrict__ partial, int N, int D,
    int H, int W, int Cin, int Cout, int relu_in, int CP, int MT, int TG, int KS, int ci_tiles, int tiles_x,
    int tiles_y, int tiles_z, int bricks_per_slab, int nslab_total, int Cmem /* channel stride of x in memory */,
    int ones_ch /* logical channel that reads as 1 inside the volume (-1: none) */,
    const float* __restrict__ xscale /* {S, 1/S} of x | NULL */, const float* __restrict__ dscale /* of dz | NULL */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
  constexpr int CO = 32 * NT;
  const int xt_bytes = (CP + 1) * XPLANE;                 // one term of sXT
  unsigned char* sXT = smemb;                             // [TERMS][(CP+1)][XPLANE]
  unsigned char* sDT = smemb + TERMS * xt_bytes;          // [TERMS][CO][DPLANE]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // work item = (slab, (ci tile, cout group)) with the tile index fastest, XCD-remapped: the workgroups that
  // re-read the same bricks for different channel tiles run on the same XCD at the same time
  const int ntile = gridDim.x / nslab_total;
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = item % ntile, slab = item / ntile;
  const int cit = tile % ci_tiles, cog = tile / ci_tiles;
  const int ci0 = cit * CP, co0 = cog * CO;
  const int tg = wv % TG, ks = wv / TG;

  const WgradTiles wt = wgrad_deal_tiles(CP, MT, TG, tg, li, lh);
  f32x16 acc[MTWB][NT];
#pragma unroll
  for (int j = 0; j < MTWB; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  // zero plane (padded M rows) of every term
  for (int e = tid; e < TERMS * (XPLANE / 4); e += WGB_TPB) {
    const int t = e / (XPLANE / 4), o = e - t * (XPLANE / 4);
    reinterpret_cast<unsigned*>(sXT + t * xt_bytes + CP * XPLANE)[o] = 0u;
  }
  // slabs never straddle samples (nslab_total = N * slabs per sample): the reduce kernel can then give per-sample sums
  const int slabs_per_n = nslab_total / N, bricks_in_n = tiles_x * tiles_y * tiles_z;
  const long long b_base = (long long)(slab / slabs_per_n) * bricks_in_n;
  const long long b_beg = b_base + (long long)(slab % slabs_per_n) * bricks_per_slab;
  long long b_end = b_beg + bricks_per_slab;
  if (b_end > b_base + bricks_in_n) b_end = b_base + bricks_in_n;
  const bool xvec = (CP >= 4) && ((Cin & 3) == 0) && (Cmem == Cin);
  const bool dvec = (Cout & 3) == 0;
  const int cq = CP >> 2;                 // channel quads per voxel (xvec)

  // ---- software pipeline over bricks: the global loads of brick b+1 are issued into registers before the
  //      MFMA phase of brick b and converted / transposed into LDS after it (1 workgroup per CU: nothing
  //      else would hide the HBM latency).  Item = 2 voxels x (4 channels | 1 channel).
  constexpr int XI = 2;                         // input-halo items per thread (24 rows x 9 pairs x <=4 quads = 864)
  constexpr int DI = (WV / 2) * (CO / 4) / WGB_TPB;   // dz items per thread (vector path): 2 (NT=2) or 1
  const int x_per_row = 9 * (xvec ? cq : CP);
  const int x_items = XROWS * x_per_row;
  float4 px[XI][2];
  float4 pd[DI > 0 ? DI : 1][2], pm[DI > 0 ? DI : 1][2];
  int pn = 0;                                   // sample index of the prefetched brick

  // per-thread staging descriptors (identical for every brick): computed once.  Global addresses are
  // (per-brick base pointer) + (precomputed 32-bit element offset relative to the brick origin).
  int xi_dz[XI], xi_dy[XI], xi_dx[XI], xi_cb[XI], xi_lds[XI], xi_rel[XI];
  bool xi_on[XI];
  const int HWs = H * W;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int e = tid + i * WGB_TPB;
    const int rowh = e / x_per_row, rem = e - rowh * x_per_row;
    const int cpart = rem / 9, pr = rem - cpart * 9;
    const int lz = rowh / WHY, ly = rowh - lz * WHY;
    xi_cb[i] = xvec ? 4 * cpart : cpart;
    xi_on[i] = (e < x_items) && (ci0 + xi_cb[i] < Cin);
    xi_dz[i] = lz - 1; xi_dy[i] = ly - 1; xi_dx[i] = 2 * pr - 1;
    xi_lds[i] = xi_cb[i] * XPLANE + (rowh * XPITCH + 2 * pr) * 2;
    xi_rel[i] = ((xi_dz[i] * H + xi_dy[i]) * W + xi_dx[i]) * Cmem + xi_cb[i];
  }
  constexpr int DIR = DI > 0 ? DI : 1;
  int di_dz[DIR], di_dy[DIR], di_dx[DIR], di_lds[DIR], di_rel[DIR];
  bool di_on[DIR];
#pragma unroll
  for (int i = 0; i < DI; ++i) {
    const int e = tid + i * WGB_TPB;
    // lanes: 4 consecutive cout quads (one 64-B global segment), then 64 voxel pairs, then quad groups
    const int q = (e & 3) + 4 * (e >> 8), pv = (e >> 2) & 63;
    const int lx = (pv % (WX / 2)) * 2, ly = (pv / (WX / 2)) % WY, lz = pv / ((WX / 2) * WY);
    di_dz[i] = lz; di_dy[i] = ly; di_dx[i] = lx;
    di_on[i] = co0 + 4 * q < Cout;
    di_lds[i] = (4 * q) * DPLANE + ((lz * WY + ly) * WX + lx) * 2;
    di_rel[i] = ((lz * H + ly) * W + lx) * Cout + 4 * q;
  }
  const float sX = xscale ? xscale[0] : 1.f, sD = dscale ? dscale[0] : 1.f;
  // normalisation coefficients of this thread's channels, reloaded only when the sample index changes
  float xsc[XI][4], xsh[XI][4];
  int coef_n = -1;
  auto load_coefs = [&](int n) {
    if (n == coef_n) return;
    coef_n = n;
#pragma unroll
    for (int i = 0; i < XI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = ci0 + xi_cb[i] + j;
        const bool ok = scale && xi_on[i] && c < Cin && (xvec || j == 0);
        xsc[i][j] = (ok ? scale[n * Cin + c] : 1.f) * sX;      // power-of-two range scale folded in (exact)
        xsh[i][j] = (ok ? shift[n * Cin + c] : 0.f) * sX;
      }
  };
  const int bricks_per_n = tiles_x * tiles_y * tiles_z, tiles_xy = tiles_x * tiles_y;
  auto brick_coords = [&](long long bi64, int& n, int& x0, int& y0, int& z0) {
    const int bi = (int)bi64;                    // < 2^31 bricks by construction
    n = bi / bricks_per_n;
    const int r = bi - n * bricks_per_n;
    const int bz = r / tiles_xy, r2 = r - bz * tiles_xy;
    const int by = r2 / tiles_x, bx = r2 - by * tiles_x;
    x0 = bx * WX; y0 = by * WY; z0 = bz * WZ;
  };
  auto prefetch = [&](long long bi) {
    int n, x0, y0, z0;
    brick_coords(bi, n, x0, y0, z0);
    pn = n;
    const long long origin = (((long long)n * D + z0) * H + y0) * W + x0;     // wave-uniform
    const float* xb = x + origin * Cmem + ci0;
    const float* db = dz + origin * Cout + co0;
    const float* mb = dzmask ? dzmask + origin * Cout + co0 : nullptr;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      px[i][0] = px[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int gy = y0 + xi_dy[i], gz = z0 + xi_dz[i];
      if (xi_on[i] && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int gx = x0 + xi_dx[i] + u;
          if ((unsigned)gx < (unsigned)W) {
            if (xvec) px[i][u] = *reinterpret_cast<const float4*>(xb + xi_rel[i] + u * Cin);
            else px[i][u].x = (ci0 + xi_cb[i] == ones_ch) ? 1.f : xb[xi_rel[i] + u * Cmem];
          }
        }
      }
    }
    if (dvec) {
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        const int gy = y0 + di_dy[i], gz = z0 + di_dz[i];
        const bool rok = di_on[i] && (gy < H) && (gz < D);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int gx = x0 + di_dx[i] + u;
          pd[i][u] = make_float4(0.f, 0.f, 0.f, 0.f);
          pm[i][u] = make_float4(1.f, 1.f, 1.f, 1.f);
          if (rok && gx < W) {
            pd[i][u] = *reinterpret_cast<const float4*>(db + di_rel[i] + u * Cout);
            if (mb) pm[i][u] = *reinterpret_cast<const float4*>(mb + di_rel[i] + u * Cout);
          }
        }
      }
    }
  };
  auto commit = [&](long long bi) {   // registers -> normalise / mask -> split -> transposed LDS images
    int n, x0, y0, z0;
    brick_coords(bi, n, x0, y0, z0);
    load_coefs(n);
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      if (xi_on[i]) {
        const int gy = y0 + xi_dy[i], gz = z0 + xi_dz[i];
        const bool rowok = (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D;
        const int nch = xvec ? 4 : 1;
        float v[2][4] = {{px[i][0].x, px[i][0].y, px[i][0].z, px[i][0].w}, {px[i][1].x, px[i][1].y, px[i][1].z, px[i][1].w}};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool ok = rowok && (unsigned)(x0 + xi_dx[i] + u) < (unsigned)W;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < nch) {
              float t = v[u][j] * xsc[i][j] + xsh[i][j];     // identity coefficients when scale == NULL
              if (relu_in) t = fmaxf(t, 0.f);
              v[u][j] = ok ? t : 0.f;                        // zero padding AFTER the normalisation
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < nch) {
            float r0 = v[0][j], r1 = v[1][j];
#pragma unroll
            for (int t = 0; t < TERMS; ++t) {
              float b0, b1;
              const unsigned h0 = to16<TERMS>(r0, b0), h1 = to16<TERMS>(r1, b1);
              *reinterpret_cast<unsigned*>(sXT + t * xt_bytes + xi_lds[i] + j * XPLANE) = h0 | (h1 << 16);
              r0 -= b0; r1 -= b1;
            }
          }
        }
      }
    }
    if (dvec) {
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        float v[2][4] = {{pd[i][0].x, pd[i][0].y, pd[i][0].z, pd[i][0].w}, {pd[i][1].x, pd[i][1].y, pd[i][1].z, pd[i][1].w}};
        const float m[2][4] = {{pm[i][0].x, pm[i][0].y, pm[i][0].z, pm[i][0].w}, {pm[i][1].x, pm[i][1].y, pm[i][1].z, pm[i][1].w}};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float r0 = (m[0][j] > 0.f) ? v[0][j] * sD : 0.f, r1 = (m[1][j] > 0.f) ? v[1][j] * sD : 0.f;
#pragma unroll
          for (int t = 0; t < TERMS; ++t) {
            float b0, b1;
            const unsigned h0 = to16<TERMS>(r0, b0), h1 = to16<TERMS>(r1, b1);
            *reinterpret_cast<unsigned*>(sDT + t * CO * DPLANE + di_lds[i] + j * DPLANE) = h0 | (h1 << 16);
            r0 -= b0; r1 -= b1;
          }
        }
      }
    } else {   // odd Cout: direct (unpipelined) scalar staging
      for (int e = tid; e < (WV / 2) * CO; e += WGB_TPB) {
        const int c = e % CO, pv = e / CO;
        const int lx = (pv % (WX / 2)) * 2, ly = (pv / (WX / 2)) % WY, lz = pv / ((WX / 2) * WY);
        const int gy = y0 + ly, gz = z0 + lz;
        float r[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int gx = x0 + lx + u;
          if ((gx < W) & (gy < H) & (gz < D) & (co0 + c < Cout)) {
            const long long off = ((((long long)n * D + gz) * H + gy) * W + gx) * Cout + co0 + c;
            r[u] = (dzmask && !(dzmask[off] > 0.f)) ? 0.f : dz[off] * sD;
          }
        }
        const int vox = (lz * WY + ly) * WX + lx;
#pragma unroll
        for (int t = 0; t < TERMS; ++t) {
          float b0, b1;
          const unsigned h0 = to16<TERMS>(r[0], b0), h1 = to16<TERMS>(r[1], b1);
          *reinterpret_cast<unsigned*>(sDT + (t * CO + c) * DPLANE + vox * 2) = h0 | (h1 << 16);
          r[0] -= b0; r[1] -= b1;
        }
      }
    }
  };

  if (b_beg < b_end) prefetch(b_beg);
  for (long long bi = b_beg; bi < b_end; ++bi) {
    __syncthreads();            // previous brick's MFMA phase is done with the LDS images
    commit(bi);
    __syncthreads();
    if (bi + 1 < b_end) prefetch(bi + 1);
    wgrad_mfma_brick<NT, TERMS>(sXT, sDT, xt_bytes, wt, ks, KS, li, lh, acc);
  }
  wgrad_store_partial<NT>(partial + (((long long)slab * KS + ks) * 27) * Cin * Cout, wt, MT, CP, ci0, co0, Cin, Cout, li,
                          lh, acc);
}

// =============================================================================================
// Wave-specialised weight gradient (the vector path: Cin % 4 == 0, Cout % 4 == 0).  The kernel above needs ~235
// registers per lane, i.e. ONE 512-thread workgroup per CU, so its staging (global -> normalise -> split -> transposed
// LDS images) and its MFMA phase run back to back.  Here a 768-thread workgroup has 8 CONSUMER waves (the same tile
// dealing and MFMA loop, no staging registers) and 4 PRODUCER waves (one per SIMD) that stage brick b+1 into the
// other half of a double-buffered LDS image while the consumers multiply brick b: one raw s_barrier per brick, the
// producers' global loads for brick b+2 stay in flight across it (only LDS traffic is drained at the barrier).
constexpr int WS_CONS = 8;
// PW producer waves: 4 (one per SIMD, 3 waves per SIMD in all: 168 registers) or 8 (two per SIMD, 128 registers: the
// consumers of the unmasked N = 64 variant fit, and the producers -- the pole with 4 -- get twice the issue slots)

__device__ __forceinline__ void ws_barrier() {
  // LDS writes / reads of this wave are complete, outstanding GLOBAL loads are not waited for
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DSPLIT (round 5): dz is the pre-split record tensor of kmh_maxpool3d_bwd_split -- (N, Cout/8, V + 1) records of 8 fp16 hi +
// 8 fp16 lo terms of fmaf(dz, S, 0) -- so a producer item (4 channels of two x neighbours) is four 8-byte loads and eight
// 16-bit packs instead of two 16-byte loads, eight multiplies and four split_pair sequences: the same words in the same
// transposed image, bit-identical sums.
template <int NT, int TERMS, bool MASK, int PW, bool DSPLIT = false, bool AMP = false>
__global__ __launch_bounds__(64 * (WS_CONS + PW), (PW == 8 ? 4 : 3)) void conv3_wgrad_ws_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ dz, const float* __restrict__ dzmask, float* __restrict__ partial, int N, int D,
    int H, int W, int Cin, int Cout, int relu_in, int CP, int MT, int TG, int KS, int ci_tiles, int tiles_x,
    int tiles_y, int tiles_z, int bricks_per_slab, int nslab_total, const float* __restrict__ xscale,
    const float* __restrict__ dscale, int dz_blocked /* dz is (N, Cout/8, D, H, W, 8) */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
  constexpr int CO = 32 * NT;
  // LDS: sXT[TERMS][CP+1][XPLANE_R] -- ONE ring image of RZ z planes (see XPLANE_R) -- then two stages of sDT[TERMS][CO][DPLANE]
  const int xt_bytes = (CP + 1) * XPLANE_R;               // one term of sXT
  constexpr int dt_bytes = TERMS * CO * DPLANE;           // one stage of sDT
  unsigned char* const sXTr = smemb;
  unsigned char* const sDT0 = smemb + TERMS * xt_bytes;
  constexpr int WS_TPB = 64 * (WS_CONS + PW), WS_PT = 64 * PW;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ntile = gridDim.x / nslab_total;
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = item % ntile, slab = item / ntile;
  const int cit = tile % ci_tiles, cog = tile / ci_tiles;
  const int ci0 = cit * CP, co0 = cog * CO;

  // zero plane (padded M rows) of every term: all RZ ring planes of it
  for (int e = tid; e < TERMS * (XPLANE_R / 4); e += WS_TPB) {
    const int t = e / (XPLANE_R / 4), o = e - t * (XPLANE_R / 4);
    reinterpret_cast<unsigned*>(sXTr + t * xt_bytes + CP * XPLANE_R)[o] = 0u;
  }
  // slabs never straddle samples (nslab_total = N * slabs per sample): the reduce kernel can then give per-sample sums
  const int bricks_per_n = tiles_x * tiles_y * tiles_z;
  const int slabs_per_n = nslab_total / N;
  const long long b_base = (long long)(slab / slabs_per_n) * bricks_per_n;
  const long long b_beg = b_base + (long long)(slab % slabs_per_n) * bricks_per_slab;
  long long b_end = b_beg + bricks_per_slab;
  if (b_end > b_base + bricks_per_n) b_end = b_base + bricks_per_n;
  if (b_beg >= b_end) return;                             // uniform over the workgroup
  auto brick_coords = [&](long long bi64, int& n, int& x0, int& y0, int& z0) {
    const int bi = (int)bi64;
    n = bi / bricks_per_n;
    const int r = bi - n * bricks_per_n;                   // z fastest: consecutive bricks share two halo planes
    const int col = r / tiles_z, bz = r - col * tiles_z;
    const int by = col / tiles_x, bx = col - by * tiles_x;
    x0 = bx * WX; y0 = by * WY; z0 = bz * WZ;
  };
  // ring base of a brick: + 2 planes per brick inside a z column, + 4 at the first brick of a column (which fills all four
  // planes of its window); producers and consumers advance it by the same rule
  const int cz_first = (int)((b_beg - b_base) % tiles_z);

  // (Round 6, measured and removed -- profiles/r6e_wgrad_prio_order_ab.txt: s_setprio 2 on the consumer waves: flat; on the
  // producer waves: 1-10 % slower (the consumers' stream is the critical path); term-major MFMA order over a row's
  // accumulators instead of three products of one accumulator back to back: flat.)
  if (wv >= WS_CONS) {
    // ------------------------------------------------------------------------------ producers
    const int pt = tid - 64 * WS_CONS;
    const int cq = CP >> 2;                               // channel quads per voxel
    const int x_per_row = 9 * cq;
    // x items of HALF a halo window (2 planes = 12 rows): set 0 = planes 2, 3 (the NEW planes of every brick), set 1 = planes
    // 0, 1 (fetched only by the first brick of a z column).  Same (ly, pair, channels) in both sets: lz differs by 2.
    const int x_items = 2 * WHY * x_per_row;              // <= 432
    constexpr int XH = (432 + WS_PT - 1) / WS_PT;         // items per thread and half: 1 (PW = 8) or 2
    constexpr int XI = 2 * XH;                            // register sets: [0, XH) = set 0, [XH, 2 XH) = set 1
    constexpr int DI = (WV / 2) * (CO / 4) / WS_PT;       // 4 (NT = 2) or 2
    int xi_pk[XI], xi_lds[XI];                     // pk = lz | ly << 4 | (2 pr) << 8 | cb << 16 | on << 30 (halo coords)
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = pt + (i % XH) * WS_PT;
      const int rowh = e / x_per_row, rem = e - rowh * x_per_row;
      const int cpart = rem / 9, pr = rem - cpart * 9;
      const int lzh = rowh / WHY, ly = rowh - lzh * WHY;
      const int lz = lzh + (i < XH ? 2 : 0);
      const int cb = 4 * cpart;
      const bool on = (e < x_items) && (ci0 + cb < Cin);
      xi_pk[i] = on ? (lz | (ly << 4) | ((2 * pr) << 8) | (cb << 16) | (1 << 30)) : 0;   // off: loads a valid dummy
      xi_lds[i] = cb * XPLANE_R + (ly * XPITCH_R + 2 * pr) * 2;                               // + the ring plane's ZSLOT, per brick
    }
    int di_pk[DI], di_lds[DI], di_q4[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int e = pt + i * WS_PT;
      const int q = (e & 3) + 4 * (e >> 8), pv = (e >> 2) & 63;
      const int lx = (pv % (WX / 2)) * 2, ly = (pv / (WX / 2)) % WY, lz = pv / ((WX / 2) * WY);
      const bool on = co0 + 4 * q < Cout;
      di_pk[i] = lz | (ly << 4) | (lx << 8) | (on ? (1 << 30) : 0);
      di_lds[i] = (4 * q) * DPLANE + ((lz * WY + ly) * WX + lx) * 2;
      di_q4[i] = on ? 4 * q : 0;                          // off: loads a valid dummy, writes zeros
      // channel-blocked dz: element offset of the quad inside the sample = (chunk plane) + voxel * 8 + (quad in chunk)
      if (dz_blocked) di_q4[i] = on ? ((co0 + 4 * q) >> 3) * (D * H * W * 8) + ((4 * q) & 7) : 0;
      // pre-split records: planes of V + 1 records of 8 floats; the quad's four fp16 hi terms are floats (quad in chunk) / 2 ..
      // + 1 of the record, its lo terms 4 floats further
      if (DSPLIT) di_q4[i] = on ? ((co0 + 4 * q) >> 3) * ((D * H * W + 1) * 8) + (((4 * q) & 7) >> 1) : 0;
    }
    const int dstride = (dz_blocked || DSPLIT) ? 8 : Cout;      // floats between x neighbours of one dz quad
    const float sX = xscale ? xscale[0] : 1.f, sD = dscale ? dscale[0] : 1.f;
    float4 px[XI][2], pd[DI][2], pm[MASK ? DI : 1][2];

    // Loads are unconditional: halo coordinates are clamped into the volume (the value is zeroed at conversion time
    // when the true coordinate was outside), so a brick's 16 (+8 mask) 16-byte loads per thread go out back to back.
    // Element offsets inside one sample are 24-bit multiply-adds (the launcher checks D*H*W*C < 2^31).
    auto issue = [&](int n, int x0, int y0, int z0, bool col_start) {
      const int xn_sets = col_start ? XI : XH;             // uniform: a column's first brick fetches all four planes
      const float* xn = x + (long long)n * D * H * W * Cin + ci0;
      const float* dn = DSPLIT ? dz + (long long)n * (Cout >> 3) * ((long long)D * H * W + 1) * 8
                               : dz + (long long)n * D * H * W * Cout + (dz_blocked ? 0 : co0);
      const float* mn = MASK ? dzmask + (long long)n * D * H * W * Cout + co0 : nullptr;
      // all element offsets first, then the loads back to back
      unsigned xo[XI][2], dO[DI][2];
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        if (i >= xn_sets) break;
        const int gz = min(max(z0 + (xi_pk[i] & 15) - 1, 0), D - 1), gy = min(max(y0 + ((xi_pk[i] >> 4) & 15) - 1, 0), H - 1);
        const int gx0 = x0 + ((xi_pk[i] >> 8) & 255) - 1, cb = (xi_pk[i] >> 16) & 255;
        const unsigned row = __umul24(__umul24(gz, H) + gy, W);
        xo[i][0] = __umul24(row + min(max(gx0, 0), W - 1), Cin) + cb;
        xo[i][1] = __umul24(row + min(max(gx0 + 1, 0), W - 1), Cin) + cb;
      }
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        const int gz = min(z0 + (di_pk[i] & 15), D - 1), gy = min(y0 + ((di_pk[i] >> 4) & 15), H - 1);
        const int gx0 = x0 + ((di_pk[i] >> 8) & 255);
        const unsigned row = __umul24(__umul24(gz, H) + gy, W);
        dO[i][0] = __umul24(row + min(gx0, W - 1), dstride) + di_q4[i];
        dO[i][1] = __umul24(row + min(gx0 + 1, W - 1), dstride) + di_q4[i];
      }
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        if (i >= xn_sets) break;
        px[i][0] = *reinterpret_cast<const float4*>(xn + xo[i][0]);
        px[i][1] = *reinterpret_cast<const float4*>(xn + xo[i][1]);
      }
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        if constexpr (DSPLIT) {        // (hi.x, hi.y, lo.x, lo.y): 4 + 4 fp16 terms of the voxel's channel quad
          const float2 h0 = *reinterpret_cast<const float2*>(dn + dO[i][0]), l0 = *reinterpret_cast<const float2*>(dn + dO[i][0] + 4);
          const float2 h1 = *reinterpret_cast<const float2*>(dn + dO[i][1]), l1 = *reinterpret_cast<const float2*>(dn + dO[i][1] + 4);
          pd[i][0] = make_float4(h0.x, h0.y, l0.x, l0.y);
          pd[i][1] = make_float4(h1.x, h1.y, l1.x, l1.y);
          continue;
        }
        pd[i][0] = *reinterpret_cast<const float4*>(dn + dO[i][0]);
        pd[i][1] = *reinterpret_cast<const float4*>(dn + dO[i][1]);
        if (MASK) {
          pm[i][0] = *reinterpret_cast<const float4*>(mn + dO[i][0]);
          pm[i][1] = *reinterpret_cast<const float4*>(mn + dO[i][1]);
        }
      }
    };
    // (n, channel) normalisation coefficients of this workgroup's CP channels, pre-multiplied by the range scale:
    // a small LDS table behind the two stages, rewritten (by every producer wave for itself: LDS operations of one
    // wave are ordered, and the waves write identical values) when the sample index changes
    float* ctab = reinterpret_cast<float*>(sDT0 + 2 * dt_bytes);
    int tab_n = -1;
    const float relu_lo = relu_in ? 0.f : -INFINITY;
    // Keeps every use of the staged registers behind the barrier: register-only work may otherwise be hoisted above
    // the (volatile, but not register-clobbering) barrier statement, and the wait for the loads with it.
    auto pin = [](float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
    auto convert = [&](int n, int x0, int y0, int z0, unsigned char* sXT, unsigned char* sDT, int ring_base, bool col_start) {
      const int xn_sets = col_start ? XI : XH;
      if (n != tab_n) {
        tab_n = n;
        if (lane < 2 * CP) {
          const int c = ci0 + (lane % CP);
          float v = lane < CP ? sX : 0.f;
          if (scale && c < Cin) v = (lane < CP ? scale[(long long)n * Cin + c] : shift[(long long)n * Cin + c]) * sX;
          ctab[lane] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        if (i >= xn_sets) break;
        const int zdst = ((ring_base + (xi_pk[i] & 15)) & (RZ - 1)) * ZSLOT;      // this halo plane's place in the ring
        const int gz = z0 + (xi_pk[i] & 15) - 1, gy = y0 + ((xi_pk[i] >> 4) & 15) - 1;
        const int gx0 = x0 + ((xi_pk[i] >> 8) & 255) - 1, cb = (xi_pk[i] >> 16) & 255;
        const bool rowok = (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D;
        const float4 sc4 = *reinterpret_cast<const float4*>(ctab + cb);
        const float4 sh4 = *reinterpret_cast<const float4*>(ctab + CP + cb);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        float v[2][4] = {{px[i][0].x, px[i][0].y, px[i][0].z, px[i][0].w}, {px[i][1].x, px[i][1].y, px[i][1].z, px[i][1].w}};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool ok = rowok && (unsigned)(gx0 + u) < (unsigned)W;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float t = fmaxf(v[u][j] * sc[j] + sh[j], relu_lo);
            v[u][j] = ok ? t : 0.f;                        // zero padding AFTER the normalisation
          }
        }
        if (xi_pk[i] >> 30) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned w[TERMS];
            split_pair<TERMS>(v[0][j], v[1][j], w);
#pragma unroll
            for (int t = 0; t < TERMS; ++t)
              *reinterpret_cast<unsigned*>(sXT + t * xt_bytes + xi_lds[i] + zdst + j * XPLANE_R) = w[t];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        const int gz = z0 + (di_pk[i] & 15), gy = y0 + ((di_pk[i] >> 4) & 15), gx0 = x0 + ((di_pk[i] >> 8) & 255);
        const bool rok = (di_pk[i] >> 30) && gy < H && gz < D;
        if constexpr (DSPLIT) {
          // words of the transposed image: (voxel 0 | voxel 1 << 16) per channel and term
          static_assert(TERMS == 2 && !MASK, "pre-split records are fp16 hi / lo, already masked");
          const bool k0 = rok && gx0 < W, k1 = rok && gx0 + 1 < W;
          unsigned a[4] = {__float_as_uint(pd[i][0].x), __float_as_uint(pd[i][0].y), __float_as_uint(pd[i][0].z), __float_as_uint(pd[i][0].w)};
          unsigned b[4] = {__float_as_uint(pd[i][1].x), __float_as_uint(pd[i][1].y), __float_as_uint(pd[i][1].z), __float_as_uint(pd[i][1].w)};
#pragma unroll
          for (int u = 0; u < 4; ++u) { a[u] = k0 ? a[u] : 0u; b[u] = k1 ? b[u] : 0u; }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned av = a[2 * t + (j >> 1)], bv = b[2 * t + (j >> 1)];
              const unsigned wd = (j & 1) ? ((av >> 16) | (bv & 0xffff0000u)) : ((av & 0xffffu) | (bv << 16));
              *reinterpret_cast<unsigned*>(sDT + t * CO * DPLANE + di_lds[i] + j * DPLANE) = wd;
            }
          continue;
        }
        float v[2][4] = {{pd[i][0].x, pd[i][0].y, pd[i][0].z, pd[i][0].w}, {pd[i][1].x, pd[i][1].y, pd[i][1].z, pd[i][1].w}};
        float m[2][4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
        if (MASK) {
          m[0][0] = pm[i][0].x; m[0][1] = pm[i][0].y; m[0][2] = pm[i][0].z; m[0][3] = pm[i][0].w;
          m[1][0] = pm[i][1].x; m[1][1] = pm[i][1].y; m[1][2] = pm[i][1].z; m[1][3] = pm[i][1].w;
        }
        const bool ok0 = rok && gx0 < W, ok1 = rok && gx0 + 1 < W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float r0 = (ok0 && m[0][j] > 0.f) ? v[0][j] * sD : 0.f, r1 = (ok1 && m[1][j] > 0.f) ? v[1][j] * sD : 0.f;
          unsigned w[TERMS];
          split_pair<TERMS>(r0, r1, w);
#pragma unroll
          for (int t = 0; t < TERMS; ++t)
            *reinterpret_cast<unsigned*>(sDT + t * CO * DPLANE + di_lds[i] + j * DPLANE) = w[t];
        }
      }
    };

    // brick coordinates advance incrementally (z fastest, then x, y; a slab stays inside one sample): no divisions in the loop
    int cn, cx, cy, cz;
    {
      int x0, y0, z0;
      brick_coords(b_beg, cn, x0, y0, z0);
      cx = x0 / WX; cy = y0 / WY; cz = z0 / WZ;
    }
    int rbase = 0;                                        // ring base of the brick whose loads are in flight
    bool cstart = true;                                   // ... and whether it is the first of its column (here: of the slab)
    issue(cn, cx * WX, cy * WY, cz * WZ, true);
    for (long long bi = b_beg; bi < b_end; ++bi) {
      unsigned char* sDTs = sDT0 + ((bi - b_beg) & 1) * dt_bytes;
      const int pn = cn, px0 = cx * WX, py0 = cy * WY, pz0 = cz * WZ;      // the brick whose loads are in flight
      const int pbase = rbase;
      const bool pstart = cstart;
      cstart = false;
      if (++cz == tiles_z) { cz = 0; cstart = true; if (++cx == tiles_x) { cx = 0; if (++cy == tiles_y) { cy = 0; ++cn; } } }
      rbase = (rbase + (cstart ? 4 : 2)) & (RZ - 1);
#pragma unroll
      for (int i = 0; i < XI; ++i) { pin(px[i][0]); pin(px[i][1]); }
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        pin(pd[i][0]); pin(pd[i][1]);
        if (MASK) { pin(pm[i][0]); pin(pm[i][1]); }
      }
      convert(pn, px0, py0, pz0, sXTr, sDTs, pbase, pstart);                 // waits for the loads of brick bi only
      if (bi + 1 < b_end) issue(cn, cx * WX, cy * WY, cz * WZ, cstart);      // in flight across the barrier
      ws_barrier();
    }
    return;
  }

  // -------------------------------------------------------------------------------- consumers
  const int tg = wv % TG, ks = wv / TG;
  const WgradTiles wt = wgrad_deal_tiles(CP, MT, TG, tg, li, lh, true);
  f32x16 acc[MTWB][NT];
#pragma unroll
  for (int j = 0; j < MTWB; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  ws_barrier();                                            // brick b_beg is staged (and the zero planes written)
  auto bricks = [&](auto mode) {
    int rbase = 0, cz = cz_first;                           // as the producers count them
    for (long long bi = b_beg; bi < b_end; ++bi) {
      const unsigned char* sDT = sDT0 + ((bi - b_beg) & 1) * dt_bytes;
      wgrad_mfma_brick<NT, TERMS, decltype(mode)::value, AMP, true>(sXTr, sDT, xt_bytes, wt, ks, KS, li, lh, acc, rbase);
      const bool nstart = ++cz == tiles_z;
      if (nstart) cz = 0;
      rbase = (rbase + (nstart ? 4 : 2)) & (RZ - 1);
      if (bi + 1 < b_end) ws_barrier();                    // brick bi+1 is staged: its ring planes and the other sDT stage
    }
  };
  // the paired dealing (CP = 16) fixes every wave's tap x offsets: waves 0-4 hold (kx 0, kx 1) of one slot, waves
  // 5-7 two kx = 2 tiles (an absent sixth one reads the zero plane, whatever its shift)
  if (TG == 8 && MT == 15) {
    if (tg < 5) bricks(std::integral_constant<int, 1>{});
    else bricks(std::integral_constant<int, 2>{});
  } else {
    bricks(std::integral_constant<int, 0>{});
  }
  wgrad_store_partial<NT>(partial + (((long long)slab * KS + ks) * 27) * Cin * Cout, wt, MT, CP, ci0, co0, Cin, Cout, li,
                          lh, acc);
}

__global__ __launch_bounds__(256) void wgrad_bf_reduce_kernel(const float* __restrict__ partial, int nslab, int Cin,
                                                              int Cout, float* __restrict__ dw, int accumulate,
                                                              const float* __restrict__ xscale,
                                                              const float* __restrict__ dscale) {
  const double desc = (double)(xscale ? xscale[1] : 1.f) * (double)(dscale ? dscale[1] : 1.f);
  const long long total = (long long)27 * Cin * Cout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    double s = 0;
    for (int k = 0; k < nslab; ++k) s += partial[(long long)k * total + e];
    const int co = (int)(e % Cout), ci = (int)((e / Cout) % Cin), tap = (int)(e / ((long long)Cout * Cin));
    const long long o = ((long long)co * Cin + ci) * 27 + tap;
    dw[o] = accumulate ? dw[o] + (float)(s * desc) : (float)(s * desc);
  }
}

// Reduce + fold: dw as above, and, from the PER-SAMPLE sums the slab order allows,
//   bhat[n][ci] = sum_{tap, co} w[co][ci][tap] * dWn[n][tap][ci][co]  =  sum_v dxn[n][v][ci] * xhat[n][v][ci]
// (dxn = the data gradient of the same dz, xhat = the convolution's input): GroupNorm's second backward statistic
// without a pass over dxn and x.  Block = (ci, tap triple); bhat must be zero on entry.
// 1024 threads = LP columns (the next power of two >= 3 Cout, capped at 1024) x S = 1024 / LP slices of the slabs: the
// slab sums are strided reads 27 Cin Cout floats apart, and one thread walking all of them ran at 1.1 TB/s.
__global__ __launch_bounds__(1024) void wgrad_bf_reduce_fold_kernel(const float* __restrict__ partial, int N, int per_n,
                                                                    int Cin, int Cout, float* __restrict__ dw,
                                                                    int accumulate, const float* __restrict__ xscale,
                                                                    const float* __restrict__ dscale,
                                                                    const float* __restrict__ w,
                                                                    double* __restrict__ bhat, int LP) {
  const double desc = (double)(xscale ? xscale[1] : 1.f) * (double)(dscale ? dscale[1] : 1.f);
  const long long total = (long long)27 * Cin * Cout;
  const int ci = blockIdx.x, t3 = blockIdx.y;
  __shared__ double red[1024];
  __shared__ double wred[1024 / kWave];
  const int S = 1024 / LP, lcol = threadIdx.x % LP, sl = threadIdx.x / LP;
  const int L = 3 * Cout;
  for (int l0 = 0; l0 < L; l0 += LP) {                      // (one pass unless 3 Cout > 1024)
    const int l = l0 + lcol;
    const bool act = l < L;
    const int tap = 3 * t3 + (act ? l / Cout : 0), co = act ? l % Cout : 0;
    const long long e = ((long long)tap * Cin + ci) * Cout + co;
    const long long o = ((long long)co * Cin + ci) * 27 + tap;
    const double wv = act ? (double)w[o] : 0.0;
    double tot = 0;
    for (int n = 0; n < N; ++n) {
      double sn = 0;
      if (act)
        for (int k = sl; k < per_n; k += S) sn += partial[((long long)n * per_n + k) * total + e];
      __syncthreads();
      red[threadIdx.x] = sn;
      __syncthreads();
      double bn = 0;
      if (sl == 0 && act) {
        sn = 0;
        for (int q = 0; q < S; ++q) sn += red[q * LP + lcol];      // fixed order
        tot += sn;
        bn = wv * sn * desc;
      }
      const double r = block_sum<double>(bn, wred);
      if (threadIdx.x == 0) atomicAdd(bhat + (long long)n * Cin + ci, r);
    }
    if (sl == 0 && act) dw[o] = accumulate ? dw[o] + (float)(tot * desc) : (float)(tot * desc);
  }
}

struct WgradBfPlan {
  int CP, MT, TG, KS, ci_tiles, co_groups, NT, tiles_x, tiles_y, tiles_z, nslab, bricks_per_slab;
  long long nbricks;
  size_t lds;        // one stage of conv3_wgrad_bf_kernel
  size_t lds_ws;     // conv3_wgrad_ws_kernel: the ring x image + two dz stages + the coefficient table
};

static WgradBfPlan wgrad_bf_plan(int N, int D, int H, int W, int Cin, int Cout, int terms) {
  WgradBfPlan p;
  p.CP = 1;
  while (p.CP < Cin && p.CP < 16) p.CP <<= 1;
  p.ci_tiles = (Cin + p.CP - 1) / p.CP;
  {
    const int tpt = 32 / p.CP;
    p.MT = 3 * ((9 + tpt - 1) / tpt);            // uniform-kx tiles: 15 (CP=16), 9, 6, 3, 3
  }
  p.TG = 1;
  while (p.TG < 8 && p.TG < p.MT) p.TG <<= 1;
  p.KS = 8 / p.TG;
  p.NT = Cout > 32 ? 2 : 1;
  p.co_groups = (Cout + 32 * p.NT - 1) / (32 * p.NT);
  p.tiles_x = (W + WX - 1) / WX; p.tiles_y = (H + WY - 1) / WY; p.tiles_z = (D + WZ - 1) / WZ;
  p.nbricks = (long long)N * p.tiles_x * p.tiles_y * p.tiles_z;
  // ~768 workgroups in all; a slab is a run of bricks of ONE sample
  const long long bricks_per_n = (long long)p.tiles_x * p.tiles_y * p.tiles_z;
  long long want = 768 / ((long long)p.ci_tiles * p.co_groups * N);
  if (want < 1) want = 1;
  if (want > bricks_per_n) want = bricks_per_n;
  p.bricks_per_slab = (int)((bricks_per_n + want - 1) / want);
  p.nslab = N * (int)((bricks_per_n + p.bricks_per_slab - 1) / p.bricks_per_slab);
  p.lds = (size_t)terms * ((size_t)(p.CP + 1) * XPLANE + (size_t)32 * p.NT * DPLANE);
  p.lds_ws = (size_t)terms * ((size_t)(p.CP + 1) * XPLANE_R + 2 * (size_t)32 * p.NT * DPLANE) + 256;
  return p;
}

template <int NT, int TERMS>
static int launch_wgrad_bf(const WgradBfPlan& p, const float* x, const float* scale, const float* shift,
                           const float* dz, const float* dzmask, float* ws, int N, int D, int H, int W, int Cin,
                           int Cout, int relu_in, int Cmem, int ones_ch, const float* xscale, const float* dscale,
                           hipStream_t s) {
  hipError_t e = hipFuncSetAttribute((const void*)conv3_wgrad_bf_kernel<NT, TERMS>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
  if (e != hipSuccess) return (int)e;
  dim3 g(p.ci_tiles * p.co_groups * p.nslab);
  conv3_wgrad_bf_kernel<NT, TERMS><<<g, WGB_TPB, p.lds, s>>>(x, scale, shift, dz, dzmask, ws, N, D, H, W, Cin, Cout,
                                                            relu_in, p.CP, p.MT, p.TG, p.KS, p.ci_tiles, p.tiles_x,
                                                            p.tiles_y, p.tiles_z, p.bricks_per_slab, p.nslab, Cmem,
                                                            ones_ch, xscale, dscale);
  return KMH_LAUNCH_CHECK();
}

template <int NT, int TERMS, bool MASK, int PW, bool DSPLIT = false>
static int launch_wgrad_ws(const WgradBfPlan& p, const float* x, const float* scale, const float* shift,
                           const float* dz, const float* dzmask, float* ws, int N, int D, int H, int W, int Cin,
                           int Cout, int relu_in, const float* xscale, const float* dscale, int dz_blocked, hipStream_t s) {
  const size_t lds = p.lds_ws;                             // ring x image, two dz stages, the coefficient table
  hipError_t e = hipFuncSetAttribute((const void*)conv3_wgrad_ws_kernel<NT, TERMS, MASK, PW, DSPLIT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 g(p.ci_tiles * p.co_groups * p.nslab);
  if constexpr (!MASK && PW == 8) {
    if (kmh_amp_enabled()) {
      e = hipFuncSetAttribute((const void*)conv3_wgrad_ws_kernel<NT, TERMS, MASK, PW, DSPLIT, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      conv3_wgrad_ws_kernel<NT, TERMS, MASK, PW, DSPLIT, true><<<g, 64 * (WS_CONS + PW), lds, s>>>(
          x, scale, shift, dz, dzmask, ws, N, D, H, W, Cin, Cout, relu_in, p.CP, p.MT, p.TG, p.KS, p.ci_tiles, p.tiles_x, p.tiles_y,
          p.tiles_z, p.bricks_per_slab, p.nslab, xscale, dscale, dz_blocked);
      return KMH_LAUNCH_CHECK();
    }
  }
  conv3_wgrad_ws_kernel<NT, TERMS, MASK, PW, DSPLIT><<<g, 64 * (WS_CONS + PW), lds, s>>>(x, scale, shift, dz, dzmask, ws, N, D, H, W, Cin, Cout,
                                                               relu_in, p.CP, p.MT, p.TG, p.KS, p.ci_tiles, p.tiles_x,
                                                               p.tiles_y, p.tiles_z, p.bricks_per_slab, p.nslab, xscale,
                                                               dscale, dz_blocked);
  return KMH_LAUNCH_CHECK();
}

}  // namespace

// the wave-specialised kernel's preconditions (vector path of the f16x3 mode)
static bool wgrad_ws_ok(const WgradBfPlan& p, int D, int H, int W, int Cin, int Cout, int terms) {
  static const bool no_ws = getenv("KEYMORPH_WGRAD_NO_WS") != nullptr;     // A/B measurements only
  return !no_ws && terms == 2 && p.CP >= 4 && (Cin & 3) == 0 && (Cout & 3) == 0 && p.lds_ws <= 160 * 1024 &&
         (long long)D * H * W * (Cin > Cout ? Cin : Cout) < (1ll << 31) &&
         (long long)D * H * W <= (1ll << 24);   // 24-bit multiply-adds index the voxels of one sample
}

/* 1 when kmh_conv3d_wgrad_bf accepts a channel-blocked dz, (N, Cout/8, D, H, W, 8), for this shape */
KMH_API int kmh_conv3d_wgrad_bf_blocked_ok(int N, int D, int H, int W, int Cin, int Cout, int terms) {
  const WgradBfPlan p = wgrad_bf_plan(N, D, H, W, Cin, Cout, terms);
  return (Cout & 7) == 0 && wgrad_ws_ok(p, D, H, W, Cin, Cout, terms) ? 1 : 0;
}

KMH_API size_t kmh_conv3d_wgrad_bf_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int terms) {
  const WgradBfPlan p = wgrad_bf_plan(N, D, H, W, Cin, Cout, terms);
  return (size_t)p.nslab * p.KS * 27 * Cin * Cout * sizeof(float);
}

/* append_ones != 0: x has Cin-1 real channels in memory and a virtual last channel that reads 1 inside the volume
 * (0 in the zero padding); dw then has Cin logical input channels.  With scale == NULL this yields, per output
 * channel and tap, R = sum_v x[v+tap] dz[v] and S = sum_v [v+tap inside] dz[v] in ONE pass. */
KMH_API int kmh_conv3d_wgrad_bf(const float* x, const float* scale, const float* shift, const float* dz,
                                const float* dzmask, float* dw, int N, int D, int H, int W, int Cin, int Cout,
                                int relu_in, int accumulate, int terms, int append_ones, const float* xscale,
                                const float* dscale, int dz_blocked, const float* w_fold, double* bhat, void* ws,
                                void* stream) {
  KmhAmpCall amp_call(terms);      // terms == 1: the fp16 kernels with hi x hi only (use_amp), for this call
  hipStream_t s = (hipStream_t)stream;
  if ((w_fold == nullptr) != (bhat == nullptr)) return -22;
  const WgradBfPlan p = wgrad_bf_plan(N, D, H, W, Cin, Cout, terms);
  if (dz_blocked && (dzmask || !kmh_conv3d_wgrad_bf_blocked_ok(N, D, H, W, Cin, Cout, terms))) return -22;
  if (dz_blocked == 2 && (terms != 2 || ((long long)D * H * W + 1) * (Cout > Cin ? Cout : Cin) >= (1ll << 31))) return -22;
  if (p.MT > p.TG * MTWB || (terms != 2 && terms != 3)) return -22;
  const int Cmem = append_ones ? Cin - 1 : Cin, ones_ch = append_ones ? Cin - 1 : -1;
  if (append_ones && (scale || Cin > 4)) return -22;
  int rc;
  if (terms == 2 && (!xscale || !dscale)) return -22;      // fp16 split without range scaling is not accurate
#define KMH_WG_CALL(NT_, T_) launch_wgrad_bf<NT_, T_>(p, x, scale, shift, dz, dzmask, (float*)ws, N, D, H, W, Cin, Cout, relu_in, Cmem, ones_ch, xscale, dscale, s)
  // wave-specialised kernel (producer / consumer waves, double-buffered LDS): vector path of the f16x3 mode
  const bool ws_ok = wgrad_ws_ok(p, D, H, W, Cin, Cout, terms) && !append_ones;
#define KMH_WS_CALL(NT_, M_, PW_) launch_wgrad_ws<NT_, 2, M_, PW_>(p, x, scale, shift, dz, dzmask, (float*)ws, N, D, H, W, Cin, Cout, relu_in, xscale, dscale, dz_blocked, s)
  static const int pw = getenv("KEYMORPH_WGRAD_PRODUCERS") ? atoi(getenv("KEYMORPH_WGRAD_PRODUCERS")) : 8;
  if (ws_ok && dz_blocked == 2) {      // pre-split dz records (kmh_maxpool3d_bwd_split)
    rc = p.NT == 2 ? launch_wgrad_ws<2, 2, false, 8, true>(p, x, scale, shift, dz, nullptr, (float*)ws, N, D, H, W, Cin, Cout, relu_in, xscale, dscale, 0, s)
                   : launch_wgrad_ws<1, 2, false, 8, true>(p, x, scale, shift, dz, nullptr, (float*)ws, N, D, H, W, Cin, Cout, relu_in, xscale, dscale, 0, s);
  } else if (ws_ok) {
    if (p.NT == 2) rc = dzmask ? KMH_WS_CALL(2, true, 4) : (pw == 8 ? KMH_WS_CALL(2, false, 8) : KMH_WS_CALL(2, false, 4));
    else rc = dzmask ? KMH_WS_CALL(1, true, 4) : (pw == 8 ? KMH_WS_CALL(1, false, 8) : KMH_WS_CALL(1, false, 4));
  } else if (p.NT == 2) rc = terms == 2 ? KMH_WG_CALL(2, 2) : KMH_WG_CALL(2, 3);
  else rc = terms == 2 ? KMH_WG_CALL(1, 2) : KMH_WG_CALL(1, 3);
#undef KMH_WS_CALL
#undef KMH_WG_CALL
  if (rc) return rc;
  const long long total = (long long)27 * Cin * Cout;
  int nb = ceil_div(total, 256);
  if (nb > 2048) nb = 2048;
  if (bhat)
  {
    int LP = 64;
    while (LP < 3 * Cout && LP < 1024) LP <<= 1;
    wgrad_bf_reduce_fold_kernel<<<dim3(Cin, 9), 1024, 0, s>>>((const float*)ws, N, (p.nslab / N) * p.KS, Cin, Cout, dw,
                                                              accumulate, xscale, dscale, w_fold, bhat, LP);
  }
  else
    wgrad_bf_reduce_kernel<<<nb, 256, 0, s>>>((const float*)ws, p.nslab * p.KS, Cin, Cout, dw, accumulate, xscale, dscale);
  return KMH_LAUNCH_CHECK();
}

#endif   // KMH_TU_WGRAD
#if !KMH_TU_WGRAD
// First-layer fold (Cin = 1, GroupNorm over the single input channel): from the raw correlations of ONE sample
//   rs (Cout, 2, 27): rs[co][0][tap] = R = sum_v x[v+tap] dz[v][co],  rs[co][1][tap] = S = sum_v [inside] dz[v][co]
// produce  dw (+)= scale*R + shift*S   (the gradient wrt the filter applied to the NORMALISED input) and
//          ab = (A, B) = (sum dxn, sum dxn*x) = (sum_{co,tap} W S, sum_{co,tap} W R)  without ever forming dxn.
namespace {
template <typename RS>
__global__ __launch_bounds__(256) void first_layer_fold_kernel(const RS* __restrict__ rs, const float* __restrict__ w,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int Cout,
                                                               float* __restrict__ dw, double* __restrict__ ab,
                                                               int accumulate) {
  const float sc = scale[0], sh = shift[0];
  double a = 0, b = 0;
  for (int e = threadIdx.x; e < Cout * 27; e += 256) {
    const int co = e / 27, tap = e % 27;
    const RS R = rs[(co * 2 + 0) * 27 + tap], S = rs[(co * 2 + 1) * 27 + tap];      // (fp64 from the dedicated kernel)
    const float g = (float)((RS)sc * R + (RS)sh * S);
    dw[e] = accumulate ? dw[e] + g : g;
    a += (double)w[e] * (double)S;
    b += (double)w[e] * (double)R;
  }
  __shared__ double red[4];
  a = block_sum<double>(a, red);
  b = block_sum<double>(b, red);
  if (threadIdx.x == 0) { ab[0] = a; ab[1] = b; }
}
}  // namespace

/* rs_f64 != 0: rs holds doubles (what kmh_conv3d_first_layer_wgrad writes: GroupNorm's sums over the whole volume cancel to
 * ~1e-3 of their terms, so the correlations are kept in fp64 until they are folded); 0: floats (the split-operand weight
 * gradient over the virtual 2-channel input, Cout > 16) */
KMH_API int kmh_conv3d_first_layer_fold(const void* rs, int rs_f64, const float* w, const float* scale_n, const float* shift_n,
                                        int Cout, float* dw, double* ab_n, int accumulate, void* stream) {
  if (rs_f64)
    first_layer_fold_kernel<double><<<1, 256, 0, (hipStream_t)stream>>>((const double*)rs, w, scale_n, shift_n, Cout, dw, ab_n, accumulate);
  else
    first_layer_fold_kernel<float><<<1, 256, 0, (hipStream_t)stream>>>((const float*)rs, w, scale_n, shift_n, Cout, dw, ab_n, accumulate);
  return KMH_LAUNCH_CHECK();
}

#endif   // !KMH_TU_WGRAD