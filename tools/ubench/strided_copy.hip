// Development aid (gfx950): what the memory system gives an FFT pass's access pattern without the FFT.  N complex
// points are viewed as A x R x B (a, j, col); a workgroup copies one tile (all R rows j, T consecutive columns) the
// way k_fft_mix2's strided pass touches memory: every thread loads RA rows (stride RB*B) into registers, then stores
// them (to the same positions of `out`).  Variants: T columns per tile (8-byte lanes), V = complex values per lane
// (1: dwordx2, 2: dwordx4), and a plain contiguous copy.     ./strided_copy [N] [B] [R]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int RA, int RB, int T, int V>
__global__ __launch_bounds__(RB * T / V) void k_tile_copy(const float2 *__restrict__ in, float2 *__restrict__ out, unsigned B, unsigned tiles,
                                                          unsigned A) {
  constexpr int R = RA * RB;
  const unsigned bid = blockIdx.x;
  const unsigned tile = bid % tiles, a = bid / tiles;
  const unsigned col0 = tile * T;
  const int s = threadIdx.x;
  const int t = (s % (T / V)) * V, j0 = s / (T / V);
  if (col0 + t >= B) return;
  const size_t base = (size_t)a * R * B + (size_t)j0 * B + col0 + t;
  if (V == 1) {
    float2 v[RA];
#pragma unroll
    for (int m = 0; m < RA; ++m) v[m] = in[base + (size_t)(RB * m) * B];
#pragma unroll
    for (int m = 0; m < RA; ++m) out[base + (size_t)(RB * m) * B] = make_float2(v[m].y, v[m].x);
  } else {
    float4 v[RA];
#pragma unroll
    for (int m = 0; m < RA; ++m) v[m] = *reinterpret_cast<const float4 *>(in + base + (size_t)(RB * m) * B);
#pragma unroll
    for (int m = 0; m < RA; ++m) *reinterpret_cast<float4 *>(out + base + (size_t)(RB * m) * B) = make_float4(v[m].y, v[m].x, v[m].w, v[m].z);
  }
}

__global__ __launch_bounds__(256) void k_copy4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v = in[i];
    out[i] = make_float4(v.y, v.x, v.w, v.z);
  }
}
// one element per thread, no loop: the launch shape of an elementwise pass
__global__ __launch_bounds__(256) void k_copy2_flat(const float2 *__restrict__ in, float2 *__restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { float2 v = in[i]; out[i] = make_float2(v.y, v.x); }
}

template <typename F>
static float time_it(const char *name, size_t bytes, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  float best = 1e30f, tot = 0;
  for (int i = 0; i < 10; ++i) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best; tot += ms;
  }
  printf("%-46s %8.1f us (mean %8.1f)  %7.1f GB/s\n", name, best * 1e3f, tot * 100.f, bytes / (best * 1e-3) / 1e9);
  return best;
}

int main(int argc, char **argv) {
  const size_t N = argc > 1 ? (size_t)atof(argv[1]) : 20000000;
  const unsigned B = argc > 2 ? (unsigned)atof(argv[2]) : (unsigned)(N / 100);
  float2 *in, *out;
  hipMalloc(&in, N * 8 + 4096); hipMalloc(&out, N * 8 + 4096);
  hipMemset(in, 1, N * 8); hipMemset(out, 0, N * 8);
  const size_t bytes = 16 * N;
  printf("N = %zu complex, R = 100, B = %u (row stride %zu bytes), A = %zu\n", N, B, (size_t)B * 8, N / 100 / B);
  const unsigned A = (unsigned)(N / 100 / B);
  time_it("contiguous float4 copy, grid-stride 2048 WG", bytes, [&] { hipLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, 0, (const float4 *)in, (float4 *)out, N / 2); });
  time_it("contiguous float4 copy, grid-stride 8192 WG", bytes, [&] { hipLaunchKernelGGL(k_copy4, dim3(8192), dim3(256), 0, 0, (const float4 *)in, (float4 *)out, N / 2); });
  time_it("contiguous float2 copy, one element per thread", bytes, [&] { hipLaunchKernelGGL(k_copy2_flat, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, in, out, N); });
#define TILE(RA, RB, T, V)                                                                                         \
  {                                                                                                                 \
    const unsigned tiles = (B + T - 1) / T;                                                                         \
    char nm[96];                                                                                                    \
    snprintf(nm, sizeof nm, "tile %dx%d rows x %d cols, %d B/lane, %d thr", RA, RB, T, 8 * V, RB * T / V);          \
    time_it(nm, bytes, [&] { hipLaunchKernelGGL((k_tile_copy<RA, RB, T, V>), dim3(tiles * A), dim3(RB * T / V), 0, 0, in, out, B, tiles, A); }); \
  }
  TILE(10, 10, 16, 1) TILE(10, 10, 32, 1) TILE(10, 10, 64, 1) TILE(10, 10, 64, 2) TILE(10, 10, 128, 2)
  TILE(20, 5, 64, 1) TILE(5, 20, 16, 1) TILE(5, 20, 32, 1) TILE(25, 4, 64, 1) TILE(4, 25, 16, 1)
  return 0;
}
