// demod.hip -- Demodulation.jl on gfx950: streaming, HBM-bound kernels.
//   amDemod        Demodulation.jl:26-28   8 B in / 4 B out per sample
//   invert_amDemod Demodulation.jl:31-35   needs a global max first
//   fmDemod        Demodulation.jl:17-23
//   abs2           GUI.jl:70 (power fed to the configuration search)
// Each lane moves 16-byte vectors (4 complex samples = 2 x float4 in, 1 x float4 out);
// the grid is capped and grid-strided so a launch is a few thousand workgroups.
#include "common.h"

namespace tsdr {

enum { DM_ABS = 0, DM_ABS2 = 1 };

template <int MODE>
__device__ inline float demod1(float re, float im) {
  return MODE == DM_ABS ? abs_c(re, im) : abs2_c(re, im);
}

// out[i] = f(iq[i]); optionally tracks max(out) through ordered-uint atomics (values >= 0,
// NaN bit patterns sort above +Inf, so a NaN propagates like Julia's maximum()).
template <int MODE, bool TRACK_MAX>
__global__ __launch_bounds__(256) void k_demod(const float4 *__restrict__ iq, size_t n, float4 *__restrict__ out,
                                               unsigned *__restrict__ maxbits) {
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned local = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = iq[2 * i], b = iq[2 * i + 1];
    float4 r;
    r.x = demod1<MODE>(a.x, a.y);
    r.y = demod1<MODE>(a.z, a.w);
    r.z = demod1<MODE>(b.x, b.y);
    r.w = demod1<MODE>(b.z, b.w);
    out[i] = r;
    if (TRACK_MAX) {
      local = max(local, __float_as_uint(r.x));
      local = max(local, __float_as_uint(r.y));
      local = max(local, __float_as_uint(r.z));
      local = max(local, __float_as_uint(r.w));
    }
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    size_t i = (n4 << 2) + threadIdx.x;
    const float *s = reinterpret_cast<const float *>(iq);
    float v = demod1<MODE>(s[2 * i], s[2 * i + 1]);
    reinterpret_cast<float *>(out)[i] = v;
    if (TRACK_MAX) local = max(local, __float_as_uint(v));
  }
  if (TRACK_MAX) {
    for (int off = 32; off > 0; off >>= 1) local = max(local, (unsigned)__shfl_xor((int)local, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, local);
  }
}

// out = 1 - out/max   (f32, correctly rounded division, two roundings)
__global__ __launch_bounds__(256) void k_invert(float *__restrict__ out, size_t n, const unsigned *__restrict__ maxbits) {
  const float mx = __uint_as_float(*maxbits);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float d = __fdiv_rn(out[i], mx);
    out[i] = __fsub_rn(1.0f, d);
  }
}

__global__ __launch_bounds__(256) void k_fm(const float2 *__restrict__ iq, size_t n, float *__restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (i == 0) { out[0] = 0.0f; continue; }
    float2 s1 = iq[i], s0 = iq[i - 1];
    float c = s0.x, d = -s0.y;  // conj(sig[n])
    float re = __fsub_rn(__fmul_rn(s1.x, c), __fmul_rn(s1.y, d));
    float im = __fadd_rn(__fmul_rn(s1.x, d), __fmul_rn(s1.y, c));
    out[i] = atan2f(im, re);
  }
}

template <int MODE>
static int demod_d(tsdr_ctx *ctx, const char *kname, const float *iq, size_t n, float *out) {
  if (!ctx || (n && (!iq || !out))) return TSDR_EINVAL;
  if (n == 0) return TSDR_OK;
  if (((uintptr_t)iq & 15) || ((uintptr_t)out & 15)) return set_err(ctx, TSDR_EINVAL, "%s: buffers must be 16-byte aligned", kname);
  int grid = stream_grid(ctx, ceil_div(n, 4));
  TSDR_LAUNCH(ctx, MODE == DM_ABS ? "am_demod" : "abs2", (k_demod<MODE, false>), dim3(grid), dim3(256), 0,
              reinterpret_cast<const float4 *>(iq), n, reinterpret_cast<float4 *>(out), (unsigned *)nullptr);
  return TSDR_OK;
}

}  // namespace tsdr

using namespace tsdr;

extern "C" {

int tsdr_am_demod_d(tsdr_ctx *ctx, const float *iq, size_t n, float *out) { return demod_d<DM_ABS>(ctx, "am_demod", iq, n, out); }
int tsdr_abs2_d(tsdr_ctx *ctx, const float *iq, size_t n, float *out) { return demod_d<DM_ABS2>(ctx, "abs2", iq, n, out); }

int tsdr_invert_am_d(tsdr_ctx *ctx, const float *iq, size_t n, float *out) {
  if (!ctx || n == 0 || !iq || !out) return TSDR_EINVAL;  // maximum() of an empty collection throws
  if (((uintptr_t)iq & 15) || ((uintptr_t)out & 15)) return set_err(ctx, TSDR_EINVAL, "invert_am: buffers must be 16-byte aligned");
  unsigned *mx = (unsigned *)ctx->scratch(WS_MISC, 16);
  if (!mx) return TSDR_ENOMEM;
  TSDR_HIP(ctx, hipMemsetAsync(mx, 0, 4, ctx->stream));
  int grid = stream_grid(ctx, ceil_div(n, 4));
  TSDR_LAUNCH(ctx, "invert_am_abs", (k_demod<DM_ABS, true>), dim3(grid), dim3(256), 0,
              reinterpret_cast<const float4 *>(iq), n, reinterpret_cast<float4 *>(out), mx);
  TSDR_LAUNCH(ctx, "invert_am_scale", k_invert, dim3(stream_grid(ctx, n)), dim3(256), 0, out, n, (const unsigned *)mx);
  return TSDR_OK;
}

int tsdr_fm_demod_d(tsdr_ctx *ctx, const float *iq, size_t n, float *out) {
  if (!ctx || (n && (!iq || !out))) return TSDR_EINVAL;
  if (n == 0) return TSDR_OK;
  TSDR_LAUNCH(ctx, "fm_demod", k_fm, dim3(stream_grid(ctx, n)), dim3(256), 0, reinterpret_cast<const float2 *>(iq), n, out);
  return TSDR_OK;
}

int tsdr_am_demod(tsdr_ctx *ctx, const float *iq, size_t n, float *out) {
  return host_map(ctx, iq, n * 8, out, n * 4, [&](void *i, void *o) { return tsdr_am_demod_d(ctx, (const float *)i, n, (float *)o); });
}
int tsdr_abs2(tsdr_ctx *ctx, const float *iq, size_t n, float *out) {
  return host_map(ctx, iq, n * 8, out, n * 4, [&](void *i, void *o) { return tsdr_abs2_d(ctx, (const float *)i, n, (float *)o); });
}
int tsdr_invert_am(tsdr_ctx *ctx, const float *iq, size_t n, float *out) {
  if (n == 0) return TSDR_EINVAL;
  return host_map(ctx, iq, n * 8, out, n * 4, [&](void *i, void *o) { return tsdr_invert_am_d(ctx, (const float *)i, n, (float *)o); });
}
int tsdr_fm_demod(tsdr_ctx *ctx, const float *iq, size_t n, float *out) {
  return host_map(ctx, iq, n * 8, out, n * 4, [&](void *i, void *o) { return tsdr_fm_demod_d(ctx, (const float *)i, n, (float *)o); });
}

}  // extern "C"
