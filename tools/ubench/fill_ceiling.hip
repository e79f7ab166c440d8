// Development aid (gfx950): what HBM takes per direction.  The raster launch of the frame loop (k_raster_fast) is 405 MB of
// writes (348 MB raster + 57.6 MB images) against 80 MB of reads; the roofline fraction bench.py prints prices all of it
// against the 8 TB/s spec figure.  This program measures, with the simplest possible kernels on buffers larger than the 256 MiB
// Infinity Cache, the rates the memory system actually sustains for
//   fill      write-only (dwordx4, grid-stride, plain / non-temporal), hipMemsetAsync beside it
//   sum       read-only
//   copy      read + write
//   mix r:w   the raster launch's own ratio (1 byte read per 5 written), streaming
// so that the launch's time can be set against the WRITE ceiling, which is what bounds it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_fill(v4f *out, size_t n4, float seed) {
  const v4f v = {seed, seed + 1.f, seed + 2.f, seed + 3.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}
__global__ __launch_bounds__(256) void k_sum(const v4f *in, size_t n4, float *res) {
  v4f a = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) a += in[i];
  if (a.x + a.y + a.z + a.w == 12345.678f) *res = a.x;
}
__global__ __launch_bounds__(256) void k_copy(const v4f *in, v4f *out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// one 16-byte read per five 16-byte writes (80 MB : 405 MB)
__global__ __launch_bounds__(256) void k_mix(const v4f *in, v4f *out, size_t n4w) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4w / 5; i += (size_t)gridDim.x * 256) {
    const v4f v = in[i];
#pragma unroll
    for (int k = 0; k < 5; ++k) out[(size_t)k * (n4w / 5) + i] = v;
  }
}

template <class F>
static float timed(F launch, int reps = 20) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t wbytes = (size_t)405359920 / 16 * 16, rbytes = 80000000;
  float *w, *r, *res;
  if (hipMalloc(&w, wbytes + 4096) != hipSuccess || hipMalloc(&r, wbytes + 4096) != hipSuccess || hipMalloc(&res, 16) != hipSuccess) return 1;
  (void)hipMemset(r, 0, wbytes);
  const size_t n4 = wbytes / 16;
  for (int grid : {2048, 4096, 8192, 16384}) {
    float ms = timed([&] { k_fill<false><<<grid, 256>>>((v4f *)w, n4, 1.f); });
    printf("fill dwordx4           grid %5d  %7.1f us  %5.2f TB/s written\n", grid, ms * 1e3, wbytes / (ms * 1e-3) / 1e12);
    ms = timed([&] { k_fill<true><<<grid, 256>>>((v4f *)w, n4, 1.f); });
    printf("fill dwordx4 non-temp. grid %5d  %7.1f us  %5.2f TB/s written\n", grid, ms * 1e3, wbytes / (ms * 1e-3) / 1e12);
  }
  float ms = timed([&] { (void)hipMemsetAsync(w, 0, wbytes, 0); });
  printf("hipMemsetAsync                     %7.1f us  %5.2f TB/s written\n", ms * 1e3, wbytes / (ms * 1e-3) / 1e12);
  for (int grid : {4096, 16384}) {
    ms = timed([&] { k_sum<<<grid, 256>>>((const v4f *)r, n4, res); });
    printf("sum (read-only)        grid %5d  %7.1f us  %5.2f TB/s read\n", grid, ms * 1e3, wbytes / (ms * 1e-3) / 1e12);
    ms = timed([&] { k_copy<<<grid, 256>>>((const v4f *)r, (v4f *)w, n4); });
    printf("copy                   grid %5d  %7.1f us  %5.2f TB/s read + written\n", grid, ms * 1e3, 2.0 * wbytes / (ms * 1e-3) / 1e12);
    ms = timed([&] { k_mix<<<grid, 256>>>((const v4f *)r, (v4f *)w, n4 / 5 * 5); });
    printf("mix 1 read : 5 written grid %5d  %7.1f us  %5.2f TB/s read + written  <- the raster launch's bytes (%.0f MB)\n", grid, ms * 1e3,
           1.2 * wbytes / (ms * 1e-3) / 1e12, 1.2 * wbytes / 1e6);
  }
  (void)rbytes;
  return 0;
}
