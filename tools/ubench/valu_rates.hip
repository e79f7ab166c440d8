// Development aid: issue-rate probe for the VALU ops the raster walk is built from (gfx950).
// Each kernel runs a long dependent-free unrolled stream of one op on 8 independent accumulators per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int n, float fa, double da) {
  float f[8]; double d[8]; int i8[8];
  for (int j = 0; j < 8; ++j) { f[j] = threadIdx.x * 0.001f + j; d[j] = f[j]; i8[j] = threadIdx.x + j; }
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) f[j] = __fmaf_rn(f[j], fa, 1.0f);
      if (OP == 1) d[j] = __fma_rn(d[j], da, 1.0);
      if (OP == 2) d[j] = (double)i8[j] + d[j] * 0.0, i8[j] += 1;          // cvt_f64_i32 (+ fma + add)
      if (OP == 3) f[j] = (float)d[j], d[j] = d[j] + 1.0;                   // cvt_f32_f64 + add_f64
      if (OP == 4) f[j] = (float)i8[j] * fa, i8[j] += 3;                    // cvt_f32_i32 + mul + iadd
      if (OP == 5) f[j] = __builtin_amdgcn_sqrtf(f[j] + 1.0f);             // sqrt + add
      if (OP == 6) i8[j] = (i8[j] > 100 ? i8[j] - 7 : i8[j] + 3);          // cmp+cndmask+2 adds
      if (OP == 7) i8[j] = i8[j] * 13 + 1;                                   // mul_lo_u32 + add
      if (OP == 8) d[j] = d[j] / da;                                         // f64 division
      if (OP == 9) d[j] = d[j] * da;                                         // mul_f64
      if (OP == 10) d[j] = d[j] + da;                                        // add_f64
      if (OP == 11) d[j] = floor(d[j]) + da;                                 // floor_f64 + add_f64
      if (OP == 12) i8[j] += (int)(unsigned)d[j], d[j] += 1.0;                // cvt_u32_f64 + iadd + add_f64
      if (OP == 13) d[j] = (double)f[j] + d[j], f[j] += 1.0f;                 // cvt_f64_f32 + add_f64 + add_f32
      if (OP == 14) d[j] = __builtin_amdgcn_fract(d[j]) + da;                // fract_f64 + add_f64
      if (OP == 15) d[j] = (double)(unsigned)i8[j] + d[j], i8[j] += 1;        // cvt_f64_u32 + add_f64 + iadd
    }
  }
  float s = 0; for (int j = 0; j < 8; ++j) s += f[j] + (float)d[j] + i8[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int ops_per) {
  float *out; hipMalloc(&out, 256 * 2048 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<2048, 256>>>(out, 16, 1.0001f, 1.0000001);
  hipEventRecord(a); k<OP><<<2048, 256>>>(out, ITER, 1.0001f, 1.0000001); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double wave_instr = 2048.0 * 4 * ITER * 8;  // per listed statement
  double clk = ms * 1e-3 * 2.4e9 * 1024;      // SIMD-clocks available (256 CU x 4 SIMD)
  printf("%-28s %8.3f ms  -> %6.2f SIMD-clocks per wave-statement (%d ops)\n", name, ms, clk / wave_instr, ops_per);
  hipFree(out);
}
int main() {
  run<0>("fma_f32", 1); run<1>("fma_f64", 1); run<9>("mul_f64", 1); run<10>("add_f64", 1);
  run<2>("cvt_f64_i32+fma64+iadd", 3); run<3>("cvt_f32_f64+add_f64", 2);
  run<4>("cvt_f32_i32+mul+iadd", 3); run<5>("sqrt_f32+add", 2); run<6>("cmp+cndmask+2iadd", 4); run<7>("mul_lo_u32+add", 2);
  run<8>("div_f64", 1);
  run<11>("floor_f64+add_f64", 2); run<12>("cvt_u32_f64+iadd+add_f64", 3); run<13>("cvt_f64_f32+add_f64+add_f32", 3); run<14>("fract_f64+add_f64", 2);
  run<15>("cvt_f64_u32+add_f64+iadd", 3);
  return 0;
}
