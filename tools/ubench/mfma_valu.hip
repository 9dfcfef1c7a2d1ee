// Microbenchmark: do VALU instructions overlap with MFMA on a CDNA4 SIMD?
//   each wave: ITER x { NM dense f16 32x32x16 MFMAs (independent accumulators), NV v_fma_f32 (independent chains) }
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu.hip -o gpurun_out/mfma_valu ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NM, int NV, int NCH, bool PHASED>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  f16v acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const float m = 1.0001f, c = 0.5f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (PHASED) {          // all matrix instructions, then all vector instructions (program order)
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        acc[j % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j % NCH], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(m), "v"(c));
    } else {               // NV / NM vector instructions after every matrix instruction
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        acc[j % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j % NCH], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NV / NM; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(m), "v"(c));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NM, int NV, int NCH, bool PHASED>
void run(int wgs_per_cu, int threads, const char* tag) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
  hipMalloc(&cyc, 8);
  const int iters = 8000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NM, NV, NCH, PHASED><<<grid, threads>>>(out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NM, NV, NCH, PHASED><<<grid, threads>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double waves_per_simd = wgs_per_cu * (threads / 64) / 4.0;
  printf("%-34s NM=%2d NV=%3d chains=%d waves/SIMD=%.0f: %7.3f ms  ns/iter/wave-slot %.1f  counter ticks/iter %.1f\n", tag, NM, NV, NCH, waves_per_simd, ms,
         ms * 1e6 / iters, (double)c / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  // the fused head's shape: 12 matrix instructions on ONE accumulator chain + ~100 vector instructions per block
  run<12, 0, 1, true>(4, 256, "mfma only, 1 chain");
  run<12, 0, 4, true>(4, 256, "mfma only, 4 chains");
  run<12, 96, 1, true>(4, 256, "phased, 1 chain");
  run<12, 96, 1, false>(4, 256, "interleaved, 1 chain");
  run<12, 96, 4, true>(4, 256, "phased, 4 chains");
  run<12, 96, 4, false>(4, 256, "interleaved, 4 chains");
  run<12, 96, 1, true>(2, 256, "phased, 1 chain");
  run<12, 96, 1, false>(2, 256, "interleaved, 1 chain");
  run<12, 96, 1, true>(1, 256, "phased, 1 chain");
  run<12, 96, 1, false>(1, 256, "interleaved, 1 chain");
  run<0, 96, 1, true>(4, 256, "valu only");
  run<0, 96, 1, true>(1, 256, "valu only");
  run<12, 48, 1, false>(4, 256, "interleaved, 1 chain");
  run<12, 192, 1, false>(4, 256, "interleaved, 1 chain");
  return 0;
}
