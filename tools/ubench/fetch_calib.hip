// Development aid (gfx950): calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE counters for the access widths the frame
// path's kernels use (MI355X_MICROARCH.md establishes "FETCH_SIZE = half the bytes" for 16 B/lane streaming reads only).
// Each kernel reads (or writes) a known number of bytes; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and
// compare the counter with the byte count printed here (tools/calib_fetch.sh does both and tabulates the ratio).
//   read4 / read8 / read16   streaming reads, 4 / 8 / 16 bytes per lane (a wave-load = 256 / 512 / 1024 contiguous bytes)
//   gather8                  8-byte reads, lanes 2400 bytes apart (the shifted-image gathers of k_shift_iir read 4 bytes
//                            600 floats apart along a column -- lanes consecutive; this is the row-strided worst case)
//   stage8                   8-byte reads of 32-byte runs every 2370 bytes (k_down_fused / k_raster_fast staging: a few
//                            consecutive IQ samples per raster line)
//   write4 / write16         streaming stores, 4 / 16 bytes per lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ in, size_t n, float *__restrict__ sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = in[i];
    acc += reinterpret_cast<const float *>(&v)[0];
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather8(const float2 *__restrict__ in, size_t n, size_t stride, float *__restrict__ sink) {
  // thread i reads element (i % rows) * stride + i / rows: consecutive lanes are `stride` elements apart
  const size_t rows = n / stride;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < rows * stride; i += (size_t)gridDim.x * 256) {
    const size_t r = i % rows, c = i / rows;
    acc += in[r * stride + c].x;
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_stage8(const float2 *__restrict__ in, size_t n, size_t pitch, int run, float *__restrict__ sink) {
  // 4 lanes per line read `run` = 4 consecutive elements at the start of every `pitch`-element line
  const size_t lines = n / pitch;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines * run; i += (size_t)gridDim.x * 256) {
    const size_t l = i / run, j = i % run;
    acc += in[l * pitch + j].x;
  }
  if (acc == 12345.678f) sink[0] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void k_write(T *__restrict__ out, size_t n) {
  T v;
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) reinterpret_cast<float *>(&v)[k] = (float)k;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}

int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 1024) << 20;   // MiB
  const char *only = argc > 2 ? argv[2] : "all";
  void *buf = nullptr;
  float *sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  auto want = [&](const char *n) { return !strcmp(only, "all") || !strcmp(only, n); };
  const int grid = 256 * 16;
  if (want("read4")) { hipLaunchKernelGGL(k_read<float>, dim3(grid), dim3(256), 0, 0, (const float *)buf, bytes / 4, sink); printf("read4 bytes %zu\n", bytes); }
  if (want("read8")) { hipLaunchKernelGGL(k_read<float2>, dim3(grid), dim3(256), 0, 0, (const float2 *)buf, bytes / 8, sink); printf("read8 bytes %zu\n", bytes); }
  if (want("read16")) { hipLaunchKernelGGL(k_read<float4>, dim3(grid), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, sink); printf("read16 bytes %zu\n", bytes); }
  if (want("gather8")) { hipLaunchKernelGGL(k_gather8, dim3(grid), dim3(256), 0, 0, (const float2 *)buf, bytes / 8, (size_t)300, sink); printf("gather8 bytes %zu\n", (bytes / 8 / 300) * 300 * 8); }
  if (want("stage8")) { hipLaunchKernelGGL(k_stage8, dim3(grid), dim3(256), 0, 0, (const float2 *)buf, bytes / 8, (size_t)296, 4, sink); printf("stage8 bytes %zu (useful; 128-byte lines touched: %zu)\n", (bytes / 8 / 296) * 4 * 8, (bytes / 8 / 296) * 128); }
  if (want("write4")) { hipLaunchKernelGGL(k_write<float>, dim3(grid), dim3(256), 0, 0, (float *)buf, bytes / 4); printf("write4 bytes %zu\n", bytes); }
  if (want("write16")) { hipLaunchKernelGGL(k_write<float4>, dim3(grid), dim3(256), 0, 0, (float4 *)buf, bytes / 16); printf("write16 bytes %zu\n", bytes); }
  hipDeviceSynchronize();
  return 0;
}
