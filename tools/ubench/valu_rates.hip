// Microbenchmark: issue rates of plain / packed / transcendental fp32 VALU instructions on a CDNA4 SIMD, and how many
// plain VALU hide beside an MFMA.  build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o <bin>
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// KIND 0: v_fma_f32  1: v_pk_fma_f32  2: v_sqrt_f32  3: v_log_f32  4: v_rsq_f32  5: v_cvt_pk_f16_f32  6: v_bfe_i32
// 7: v_cndmask (vcc)  8: v_pk_add_f32  9: v_mul_lo_u32  10: v_mad_u64_u32
template <int KIND, int NI>
__global__ __launch_bounds__(256) void kv(float* out, int iters) {
  float v[8];
  f2 p[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i + 1.5f; p[i] = f2{v[i], v[i] + 1.f}; }
  const float m = 1.0001f, c = 0.5f;
  const f2 m2 = {m, m}, c2 = {c, c};
  const unsigned long long smask = 0x5555aaaa3333ccccull + (unsigned long long)iters;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(m), "v"(c));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 7]) : "v"(m2), "v"(c2));
      if (KIND == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[j & 7]));
      if (KIND == 3) asm volatile("v_log_f32 %0, %0" : "+v"(v[j & 7]));
      if (KIND == 4) asm volatile("v_rsq_f32 %0, %0" : "+v"(v[j & 7]));
      if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 6) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(v[j & 7]));
      if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j & 7]) : "v"(m) : );
      if (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(m2));
      if (KIND == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[j & 7]) : "v"(m), "v"(c) : "vcc");
      if (KIND == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(m), "s"(smask));
      if (KIND == 12) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 13) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 14) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 15) asm volatile("v_mov_b32 %0, %1" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 16) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 17) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 18) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 19) asm volatile("v_cmp_lt_f32 vcc, 0, %0" : : "v"(v[j & 7]) : "vcc");
      if (KIND == 20) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(v[j & 7]) : : "vcc");
      if (KIND == 21) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(p[j & 7]) : "v"(m2));
      if (KIND == 22) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(v[j & 7]));
      if (KIND == 23) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 24) asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(v[j & 7]) : "v"(m), "s"(smask));
      if (KIND == 25) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j & 7]) : "v"(m), "v"(c));
      if (KIND == 26) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j & 7]) : "v"(m) : "vcc");
      if (KIND == 27) asm volatile("v_cmp_lt_f32 s[20:21], 0, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[j & 7]) : "v"(m) : "s20", "s21");
      if (KIND == 28) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j & 7]) : "v"(m) : "vcc");
      if (KIND == 29) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 30) asm volatile("v_max_f32_e64 %0, %0, %1" : "+v"(v[j & 7]) : "v"(m));
      if (KIND == 31) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %1, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j & 7]), "+v"(v[(j + 3) & 7]) : : "vcc");
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NI>
void runv(int wgs_per_cu, const char* tag) {
  float* out;
  hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
  const int iters = 4000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kv<KIND, NI><<<grid, 256>>>(out, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kv<KIND, NI><<<grid, 256>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // instructions per SIMD = waves/SIMD x iters x NI; report ns per wave-instruction per SIMD
  const double waves = wgs_per_cu;   // 256 threads = 4 waves = one per SIMD per workgroup
  printf("%-22s waves/SIMD=%.0f: %8.3f ms  -> %.2f ns per wave64 instruction on a SIMD\n", tag, waves, ms,
         ms * 1e6 / (waves * iters * NI));
  hipFree(out);
}

int main() {
  for (int w : {1, 4}) {
    if (w == 1) {
      runv<0, 64>(1, "v_fma_f32"); runv<1, 64>(1, "v_pk_fma_f32"); runv<8, 64>(1, "v_pk_add_f32"); runv<2, 64>(1, "v_sqrt_f32");
      runv<3, 64>(1, "v_log_f32"); runv<4, 64>(1, "v_rsq_f32"); runv<5, 64>(1, "v_cvt_pk_f16_f32"); runv<6, 64>(1, "v_bfe_i32");
      runv<7, 64>(1, "v_cndmask_b32"); runv<9, 64>(1, "v_mul_lo_u32"); runv<10, 64>(1, "v_mad_u64_u32");
    } else {
      runv<0, 64>(4, "v_fma_f32"); runv<1, 64>(4, "v_pk_fma_f32"); runv<8, 64>(4, "v_pk_add_f32"); runv<2, 64>(4, "v_sqrt_f32");
      runv<3, 64>(4, "v_log_f32"); runv<4, 64>(4, "v_rsq_f32"); runv<5, 64>(4, "v_cvt_pk_f16_f32"); runv<6, 64>(4, "v_bfe_i32");
      runv<7, 64>(4, "v_cndmask_b32"); runv<9, 64>(4, "v_mul_lo_u32"); runv<10, 64>(4, "v_mad_u64_u32");
      runv<11, 64>(4, "v_cndmask_e64 sgpr"); runv<24, 64>(4, "v_cndmask_e64 0,v,s"); runv<12, 64>(4, "v_max_f32"); runv<13, 64>(4, "v_and_b32");
      runv<14, 64>(4, "v_add_f32"); runv<23, 64>(4, "v_mul_f32"); runv<25, 64>(4, "v_fmac_f32"); runv<15, 64>(4, "v_mov_b32"); runv<16, 64>(4, "v_alignbit_b32");
      runv<17, 64>(4, "v_fma_mix_f32"); runv<18, 64>(4, "v_fma_mixlo_f16"); runv<19, 64>(4, "v_cmp_lt_f32 vcc"); runv<20, 64>(4, "v_cmp + v_addc (x2)");
      runv<21, 64>(4, "v_lshl_add_u64"); runv<22, 64>(4, "v_cvt_f32_i32");
      runv<26, 64>(4, "v_cmp vcc + cndmask e32 (x2)"); runv<27, 64>(4, "v_cmp sgpr + cndmask e64 (x2)"); runv<28, 64>(4, "cmp, s_nop, cndmask e32");
      runv<29, 64>(4, "v_cndmask_e64 .. vcc"); runv<30, 64>(4, "v_max_f32_e64"); runv<31, 64>(4, "cmp,mov,mov,cndmask (x4)");
    }
  }
  return 0;
}
