// fft64.hip -- complex f64 FFT on the device, for one-off set-up work: initLPF (Resampler.jl:83-99) multiplies the
// impulse response by a Float64 window, which promotes h and H = fft(h) to ComplexF64 (:93-97), so the filter the
// resampler applies has f64 precision in the reference and must have it here.
//
// Stockham autosort, decimation in frequency: N = r_1 r_2 ... r_p with prime r_i <= 13, one out-of-place launch per
// factor (a thread reads r values, forms the r-point DFT directly and writes r values; twiddles from sincospi in f64).
// Lengths with a larger prime factor go through Bluestein's chirp transform on a power-of-two length.  Nothing here is
// on a hot path: a 4e6-point transform is 14 streaming launches.
#include <vector>

#include "common.h"

namespace tsdr {

constexpr int kMaxRadix64 = 13;

__device__ inline double2 cmul64(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// one Stockham pass: n = current sub-transform length, s = N / n its stride, radix r | n, sign = -1 forward / +1 inverse
//   y[q + s (r p + u)] = W_n^(p u) * sum_t x[q + s (p + t n/r)] W_r^(t u)        p < n/r, q < s, u < r
__global__ __launch_bounds__(256) void k_fft64_pass(const double2 *__restrict__ x, double2 *__restrict__ y, size_t N, size_t n,
                                                    size_t s, unsigned r, double sign) {
  const size_t m = n / r;
  double2 wr[kMaxRadix64];
  for (unsigned j = 0; j < r; ++j) {
    double sn, cs;
    sincospi(2.0 * (double)j / (double)r, &sn, &cs);
    wr[j] = make_double2(cs, sign * sn);
  }
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < N / r; g += (size_t)gridDim.x * blockDim.x) {
    const size_t p = g / s, q = g - p * s;
    double2 a[kMaxRadix64];
    for (unsigned t = 0; t < r; ++t) a[t] = x[q + s * (p + (size_t)t * m)];
    double sn, cs;
    sincospi(2.0 * (double)p / (double)n, &sn, &cs);
    const double2 w = make_double2(cs, sign * sn);
    double2 wu = make_double2(1.0, 0.0);
    for (unsigned u = 0; u < r; ++u) {
      double2 acc = a[0];
      for (unsigned t = 1; t < r; ++t) {
        const double2 c = cmul64(a[t], wr[(t * u) % r]);
        acc.x += c.x; acc.y += c.y;
      }
      y[q + s * ((size_t)r * p + u)] = cmul64(acc, wu);
      wu = cmul64(wu, w);
    }
  }
}

__global__ __launch_bounds__(256) void k_scale64(double2 *__restrict__ x, size_t N, double g) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
    x[i].x *= g; x[i].y *= g;
  }
}

// Bluestein: a[k] = x[k] c[k] (k < n, zero beyond), b = conj(c) wrapped onto length L; after the convolution out[k] = conv[k] c[k];
// c[k] = exp(sign i pi k^2 / n), k^2 reduced modulo 2n exactly
__device__ inline double2 chirp64(size_t k, size_t n, double sign) {
  const unsigned long long e = (unsigned long long)(((unsigned __int128)k * k) % (2 * (unsigned __int128)n));
  double sn, cs;
  sincospi((double)e / (double)n, &sn, &cs);
  return make_double2(cs, sign * sn);
}
__global__ __launch_bounds__(256) void k_blue64_prep(const double2 *__restrict__ x, size_t n, size_t L, double sign,
                                                     double2 *__restrict__ a, double2 *__restrict__ b) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < L; k += (size_t)gridDim.x * blockDim.x) {
    double2 av = make_double2(0.0, 0.0), bv = make_double2(0.0, 0.0);
    if (k < n) {
      const double2 c = chirp64(k, n, sign);
      av = cmul64(x[k], c);
      bv = make_double2(c.x, -c.y);
    } else if (L - k < n) {
      const double2 c = chirp64(L - k, n, sign);
      bv = make_double2(c.x, -c.y);
    }
    a[k] = av; b[k] = bv;
  }
}
__global__ __launch_bounds__(256) void k_mul64(double2 *__restrict__ a, const double2 *__restrict__ b, size_t L) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < L; k += (size_t)gridDim.x * blockDim.x) a[k] = cmul64(a[k], b[k]);
}
__global__ __launch_bounds__(256) void k_blue64_post(const double2 *__restrict__ conv, size_t n, double sign, double g,
                                                     double2 *__restrict__ out) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
    const double2 v = cmul64(conv[k], chirp64(k, n, sign));
    out[k] = make_double2(v.x * g, v.y * g);
  }
}

static int smooth_passes(tsdr_ctx *ctx, double2 *data, double2 *scratch, size_t N, const std::vector<unsigned> &rad, double sign) {
  double2 *src = data, *dst = scratch;
  size_t n = N, s = 1;
  for (unsigned r : rad) {
    TSDR_LAUNCH(ctx, "fft64_pass", k_fft64_pass, dim3(stream_grid(ctx, N / r)), dim3(256), 0, (const double2 *)src, dst, N, n, s, r, sign);
    n /= r; s *= r;
    std::swap(src, dst);
  }
  if (src != data) TSDR_HIP(ctx, hipMemcpyAsync(data, src, N * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
  return TSDR_OK;
}

// in-place complex f64 FFT of N device values; scratch: N more.  dir < 0 forward (unnormalised), > 0 inverse (scaled 1/N):
// FFTW.jl fft / ifft.  Enqueues on the context's stream.
int fft64_d(tsdr_ctx *ctx, double2 *data, double2 *scratch, size_t N, int dir) {
  if (N <= 1) return TSDR_OK;
  const double sign = dir < 0 ? -1.0 : 1.0;
  std::vector<unsigned> rad;
  size_t m = N;
  for (unsigned p : {2u, 3u, 5u, 7u, 11u, 13u})
    while (m % p == 0) { rad.push_back(p); m /= p; }
  if (m == 1) {
    int rc = smooth_passes(ctx, data, scratch, N, rad, sign);
    if (rc) return rc;
    if (dir > 0) TSDR_LAUNCH(ctx, "fft64_scale", k_scale64, dim3(stream_grid(ctx, N)), dim3(256), 0, data, N, 1.0 / (double)N);
    return TSDR_OK;
  }
  size_t L = 1;
  while (L < 2 * N - 1) L <<= 1;
  double2 *a = nullptr, *b = nullptr, *t = nullptr;
  if (hipMalloc((void **)&a, L * sizeof(double2)) != hipSuccess || hipMalloc((void **)&b, L * sizeof(double2)) != hipSuccess ||
      hipMalloc((void **)&t, L * sizeof(double2)) != hipSuccess) {
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(t);
    return set_err(ctx, TSDR_ENOMEM, "fft64: Bluestein buffers");
  }
  std::vector<unsigned> two;
  for (size_t v = L; v > 1; v >>= 1) two.push_back(2u);
  auto checked = [&](const char *what) {   // a launch failure must surface here, not as the sticky error of a later, unrelated launch
    const hipError_t le = hipGetLastError();
    return le == hipSuccess ? (int)TSDR_OK : hip_fail(ctx, le, what);
  };
  hipLaunchKernelGGL(k_blue64_prep, dim3(stream_grid(ctx, L)), dim3(256), 0, ctx->launch_stream, (const double2 *)data, N, L, sign, a, b);
  int rc = checked("fft64: k_blue64_prep");
  if (!rc) rc = smooth_passes(ctx, a, t, L, two, -1.0);
  if (!rc) rc = smooth_passes(ctx, b, t, L, two, -1.0);
  if (!rc) {
    hipLaunchKernelGGL(k_mul64, dim3(stream_grid(ctx, L)), dim3(256), 0, ctx->launch_stream, a, (const double2 *)b, L);
    rc = checked("fft64: k_mul64");
    if (!rc) rc = smooth_passes(ctx, a, t, L, two, +1.0);
  }
  if (!rc) {
    const double g = (1.0 / (double)L) * (dir > 0 ? 1.0 / (double)N : 1.0);
    hipLaunchKernelGGL(k_blue64_post, dim3(stream_grid(ctx, N)), dim3(256), 0, ctx->launch_stream, (const double2 *)a, N, sign, g, data);
    rc = checked("fft64: k_blue64_post");
  }
  // (a stream that never completes keeps its scratch: hipFree would wait for the device without a bound)
  if (int w = tsdr::wait_stream(ctx, ctx->stream, "fft64")) return rc ? rc : w;
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(t);
  return rc;
}

}  // namespace tsdr
