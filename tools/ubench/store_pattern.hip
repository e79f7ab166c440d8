// Development aid (gfx950): what the raster's store stream costs by shape.  Every variant writes the same 30 x
// 2576 x Y x 4 bytes in column-major (line fastest) layout, tile by tile as the raster kernel does (workgroup = 8
// wavefronts = 2 vertical x 4 horizontal, 32 pixel columns per wavefront), differing only in
//   V = lines per lane (1: dword stores, 256 B per wave-instruction; 2: dwordx2, 512 B; 4: dwordx4, 1024 B)
//   Y = column height (1125: every segment starts on an arbitrary 4-byte boundary; 1152: line-aligned segments)
// and optionally in a little VALU work between stores (W fma per pixel) to see how the two overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int V, int W>
__global__ __launch_bounds__(512) void k_store(float *out, int y_t, int x_t, int tiles_l, int tiles_p, float seed) {
  // tile: (2 * 64 * V) lines x 128 pixels; wave (wv, wh): lines [wv*64*V, +64*V), pixels [wh*32, +32)
  const int bid = blockIdx.x;
  const int tl = bid % tiles_l, rest = bid / tiles_l, tp = rest % tiles_p, f = rest / tiles_p;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wv = wave >> 2, wh = wave & 3;
  const int l0 = tl * (128 * V) + wv * 64 * V + lane * V;
  const int p0 = tp * 128 + wh * 32;
  float acc = seed + lane;
  float *base = out + (size_t)f * y_t * x_t;
  for (int p = p0; p < p0 + 32 && p < x_t; ++p) {
#pragma unroll
    for (int w = 0; w < W; ++w) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    if (l0 + V <= y_t) {
      float *dst = base + (size_t)p * y_t + l0;
      // (inline asm: the compiler cannot prove 8/16-byte alignment -- there is none for y_t = 1125 -- and would split
      // the store; the hardware takes dword-aligned wide stores)
      typedef float v2f __attribute__((ext_vector_type(2)));
      typedef float v4f __attribute__((ext_vector_type(4)));
      if (V == 1) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(acc) : "memory");
      if (V == 2) { v2f d = {acc, acc + 1.0f}; asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(d) : "memory"); }
      if (V == 4) { v4f d = {acc, acc + 1.0f, acc + 2.0f, acc + 3.0f}; asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(d) : "memory"); }
    }
  }
}

// the same bytes with every wave-store 128-byte aligned although y_t is not a multiple of 32: for pixel column p the
// 64-line windows of the wavefronts are shifted by s(p) = (-(f*P + p*y_t)) mod 32 lines ("sheared" tiles); one more
// wavefront row covers the tail.
template <int W>
__global__ __launch_bounds__(512) void k_store_sheared(float *out, int y_t, int x_t, int tiles_l, int tiles_p, float seed) {
  const int bid = blockIdx.x;
  const int tl = bid % tiles_l, rest = bid / tiles_l, tp = rest % tiles_p, f = rest / tiles_p;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wv = wave >> 2, wh = wave & 3;
  const int p0 = tp * 128 + wh * 32;
  float acc = seed + lane;
  const size_t fbase = (size_t)f * y_t * x_t;
  for (int p = p0; p < p0 + 32 && p < x_t; ++p) {
#pragma unroll
    for (int w = 0; w < W; ++w) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    const size_t col = fbase + (size_t)p * y_t;
    const int s = (int)((32 - (col & 31)) & 31);          // first line of this column that starts a 128-byte line
    const int l = (tl * 2 + wv) * 64 + s - 32 + lane;      // windows [64k + s - 32, +64)
    if (l >= 0 && l < y_t) {
      float *dst = out + col + l;
      asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(acc) : "memory");
    }
  }
}

template <int W>
static void run_sheared(const char *name, int y_t) {
  const int x_t = 2576, frames = 30;
  const int tiles_l = (y_t + 32 + 127) / 128, tiles_p = (x_t + 127) / 128;
  const size_t bytes = (size_t)frames * y_t * x_t * 4;
  float *out;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int grid = tiles_l * tiles_p * frames;
  for (int i = 0; i < 3; ++i) k_store_sheared<W><<<grid, 512>>>(out, y_t, x_t, tiles_l, tiles_p, 1.0f);
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_store_sheared<W><<<grid, 512>>>(out, y_t, x_t, tiles_l, tiles_p, 1.0f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
  printf("%-34s y_t=%d  %7.1f us  %6.2f TB/s  (grid %d)\n", name, y_t, ms * 1e3, bytes / (ms * 1e-3) / 1e12, grid);
  (void)hipFree(out);
}

template <int V, int W>
static void run(const char *name, int y_t) {
  const int x_t = 2576, frames = 30;
  const int tiles_l = (y_t + 128 * V - 1) / (128 * V), tiles_p = (x_t + 127) / 128;
  const size_t bytes = (size_t)frames * y_t * x_t * 4;
  float *out;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = tiles_l * tiles_p * frames;
  for (int i = 0; i < 3; ++i) k_store<V, W><<<grid, 512>>>(out, y_t, x_t, tiles_l, tiles_p, 1.0f);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_store<V, W><<<grid, 512>>>(out, y_t, x_t, tiles_l, tiles_p, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  printf("%-34s y_t=%d  %7.1f us  %6.2f TB/s  (grid %d)\n", name, y_t, ms * 1e3, bytes / (ms * 1e-3) / 1e12, grid);
  hipFree(out);
}

int main() {
  for (int y : {1125, 1152}) {
    run<1, 0>("dword    (1 line/lane), no VALU", y);
    run<2, 0>("dwordx2  (2 lines/lane), no VALU", y);
    run<4, 0>("dwordx4  (4 lines/lane), no VALU", y);
    run<1, 12>("dword    + 12 fma/pixel", y);
    run<2, 24>("dwordx2  + 24 fma/pixel", y);
    run<4, 48>("dwordx4  + 48 fma/pixel", y);
    run_sheared<0>("dword sheared (aligned), no VALU", y);
    run_sheared<12>("dword sheared + 12 fma/pixel", y);
  }
  return 0;
}
