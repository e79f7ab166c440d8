// resample.hip -- Resampler.jl on gfx950.
//
//   sig_to_image (Resampler.jl:117-122)   k_raster      IQ/|IQ| -> column-major (y_t,x_t) raster
//   downgradeImage (:124-126)             k_resize2d    raster -> (h_out,w_out)
//   sig_to_image |> downgradeImage        k_down_fused  IQ/|IQ| -> (h_out,w_out) with no raster in HBM
//   imresize(sig,n)                       k_resize1d
//   naiveResampler (:103-110)             k_naive
//
// Arithmetic mirrors ImageTransformations.imresize! / Interpolations BSpline(Linear()) as
// restated in oracle/tempest_oracle.c: f64 source coordinate sf*i+off (two roundings),
// f64 weights, f32 samples, one rounding to f32 per interpolated value.  The fused kernel
// rounds each raster value to f32 before the 2-D blend, exactly as going through the
// materialised Float32 raster would.
//
// Layout: the raster is column-major (y_t,x_t): element (line l, pixel p) at p*y_t + l, so a
// wavefront owns 64 consecutive LINES of one pixel column and its store is one contiguous
// 256-byte segment.  Source samples of a tile (64 lines x TP pixels) are staged once in LDS
// ([line][sample], odd row pitch): HBM sees every IQ sample once per tile row, coalesced in
// 128-byte runs, and |IQ| is evaluated once per staged sample rather than once per pixel.
#include "common.h"

namespace tsdr {

template <bool CPLX>
__device__ inline float load_sample(const float *__restrict__ src, unsigned k) {
  if (CPLX) {
    float2 z = reinterpret_cast<const float2 *>(src)[k];
    return abs_c(z.x, z.y);
  }
  return src[k];
}

// ------------------------------------------------------------------------------------------
// k_raster: one frame-tile of 64 lines x TP pixels per workgroup.
// ------------------------------------------------------------------------------------------
template <bool CPLX>
__global__ __launch_bounds__(256) void k_raster(const float *__restrict__ in, size_t in_stride, unsigned S, int y_t,
                                                int x_t, float *__restrict__ out, size_t out_stride, int TP, int W,
                                                int tiles_p) {
  extern __shared__ float lds[];
  const int Wp = W | 1;
  int *kfirst = reinterpret_cast<int *>(lds + 64 * Wp);
  const int tl = blockIdx.x / tiles_p, tp = blockIdx.x - tl * tiles_p;
  const int l0 = tl * 64, p0 = tp * TP;
  const int f = blockIdx.y;
  const float *src = in + (size_t)f * in_stride * (CPLX ? 2 : 1);
  const unsigned P = (unsigned)y_t * (unsigned)x_t;
  const RsAxis ax = rs_axis(S, P);
  const bool same = (S == P);  // imresize copies when sizes match
  const int tid = threadIdx.x;

  if (tid < 64) {
    int l = min(l0 + tid, y_t - 1);
    double d;
    kfirst[tid] = (int)rs_pos(ax, (double)((unsigned)l * (unsigned)x_t + (unsigned)p0 + 1u), d);
  }
  __syncthreads();
  {  // stage: 16 lanes per line -> 128-byte runs of IQ
    const int sub = tid >> 4, j0 = tid & 15;
    for (int r = sub; r < 64; r += 16) {
      const unsigned kf = (unsigned)kfirst[r];
      for (int j = j0; j < W; j += 16) {
        unsigned k = min(kf + (unsigned)j, S - 1u);
        lds[r * Wp + j] = load_sample<CPLX>(src, k);
      }
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int l = l0 + lane;
  if (l < y_t) {
    const int kf = kfirst[lane];
    const int pw = TP >> 2;
    const int pbeg = p0 + wave * pw, pend = min(pbeg + pw, x_t);
    const float *row = lds + lane * Wp;
    float *o = out + (size_t)f * out_stride + (size_t)l;
    const unsigned base = (unsigned)l * (unsigned)x_t + 1u;
    for (int p = pbeg; p < pend; ++p) {
      double d;
      int j = (int)rs_pos(ax, (double)(base + (unsigned)p), d) - kf;
      float a = row[j], b = row[j + 1];
      o[(size_t)p * y_t] = same ? (d == 1.0 ? b : a) : rs_blend(a, b, d);
    }
  }
}

// direct variant (no LDS) for ratios the tiled kernel cannot stage; lanes along lines.
template <bool CPLX>
__global__ __launch_bounds__(256) void k_raster_direct(const float *__restrict__ in, size_t in_stride, unsigned S,
                                                       int y_t, int x_t, float *__restrict__ out, size_t out_stride) {
  const int f = blockIdx.y;
  const float *src = in + (size_t)f * in_stride * (CPLX ? 2 : 1);
  const unsigned P = (unsigned)y_t * (unsigned)x_t;
  const RsAxis ax = rs_axis(S, P);
  const bool same = (S == P);
  const int lblocks = (y_t + 63) >> 6;
  const size_t total = (size_t)lblocks * x_t * 64;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
    int lane = (int)(w & 63);
    size_t rest = w >> 6;
    int p = (int)(rest % (size_t)x_t);
    int l = (int)(rest / (size_t)x_t) * 64 + lane;
    if (l >= y_t) continue;
    double d;
    unsigned k = rs_pos(ax, (double)((unsigned)l * (unsigned)x_t + (unsigned)p + 1u), d);
    float a = load_sample<CPLX>(src, k), b = load_sample<CPLX>(src, k + 1u);
    out[(size_t)f * out_stride + (size_t)p * y_t + l] = same ? (d == 1.0 ? b : a) : rs_blend(a, b, d);
  }
}

// ------------------------------------------------------------------------------------------
// k_down_fused: sig_to_image |> downgradeImage without the raster.  Tile = 64 output rows x
// TC output columns; the source lines those rows touch are staged in LDS.
// ------------------------------------------------------------------------------------------
struct DownParams {
  unsigned S;
  int y_t, x_t, h_out, w_out;
  int TC, NL, W, tiles_c;
};

template <bool CPLX>
__global__ __launch_bounds__(256) void k_down_fused(const float *__restrict__ in, size_t in_stride, DownParams q,
                                                    float *__restrict__ out, size_t out_stride) {
  extern __shared__ float lds[];
  const int Wp = q.W | 1;
  int *kfirst = reinterpret_cast<int *>(lds + (size_t)q.NL * Wp);
  const int tr = blockIdx.x / q.tiles_c, tc = blockIdx.x - tr * q.tiles_c;
  const int r0 = tr * 64, c0 = tc * q.TC;
  const int f = blockIdx.y;
  const float *src = in + (size_t)f * in_stride * (CPLX ? 2 : 1);
  const unsigned P = (unsigned)q.y_t * (unsigned)q.x_t;
  const RsAxis ax1 = rs_axis(q.S, P);
  const RsAxis ay = rs_axis((size_t)q.y_t, (size_t)q.h_out);
  const RsAxis axx = rs_axis((size_t)q.x_t, (size_t)q.w_out);
  const bool same1 = (q.S == P);
  const int tid = threadIdx.x;
  double dtmp;
  // tile's source-line and raster-pixel ranges (uniform)
  const int ly0 = (int)rs_pos(ay, (double)(r0 + 1), dtmp);
  const int ly1 = (int)rs_pos(ay, (double)(min(r0 + 63, q.h_out - 1) + 1), dtmp) + 1;
  const int nl = ly1 - ly0 + 1;
  const int pxa = (int)rs_pos(axx, (double)(c0 + 1), dtmp);
  for (int i = tid; i < nl; i += 256)
    kfirst[i] = (int)rs_pos(ax1, (double)((unsigned)(ly0 + i) * (unsigned)q.x_t + (unsigned)pxa + 1u), dtmp);
  __syncthreads();
  {
    const int sub = tid >> 4, j0 = tid & 15;
    for (int i = sub; i < nl; i += 16) {
      const unsigned kf = (unsigned)kfirst[i];
      for (int j = j0; j < q.W; j += 16) {
        unsigned k = min(kf + (unsigned)j, q.S - 1u);
        lds[i * Wp + j] = load_sample<CPLX>(src, k);
      }
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int r = r0 + lane;
  if (r >= q.h_out) return;
  double dy;
  const int ky = (int)rs_pos(ay, (double)(r + 1), dy);
  const int i0 = ky - ly0;
  const float *row0 = lds + (size_t)i0 * Wp;
  const float *row1 = row0 + Wp;
  const int kf0 = kfirst[i0], kf1 = kfirst[i0 + 1];
  const unsigned b0 = (unsigned)ky * (unsigned)q.x_t + 1u, b1 = b0 + (unsigned)q.x_t;
  const int cend = min(c0 + q.TC, q.w_out);
  float *o = out + (size_t)f * out_stride + (size_t)r;
  for (int c = c0 + wave; c < cend; c += 4) {
    double dx, d;
    const unsigned kx = rs_pos(axx, (double)(c + 1), dx);
    int j;
    // the four raster values, each rounded to f32 as the materialised raster would hold them
    j = (int)rs_pos(ax1, (double)(b0 + kx), d) - kf0;
    const float R00 = same1 ? (d == 1.0 ? row0[j + 1] : row0[j]) : rs_blend(row0[j], row0[j + 1], d);
    j = (int)rs_pos(ax1, (double)(b0 + kx + 1u), d) - kf0;
    const float R01 = same1 ? (d == 1.0 ? row0[j + 1] : row0[j]) : rs_blend(row0[j], row0[j + 1], d);
    j = (int)rs_pos(ax1, (double)(b1 + kx), d) - kf1;
    const float R10 = same1 ? (d == 1.0 ? row1[j + 1] : row1[j]) : rs_blend(row1[j], row1[j + 1], d);
    j = (int)rs_pos(ax1, (double)(b1 + kx + 1u), d) - kf1;
    const float R11 = same1 ? (d == 1.0 ? row1[j + 1] : row1[j]) : rs_blend(row1[j], row1[j + 1], d);
    // first dimension (lines) outermost: wy0*(wx0*a00 + wx1*a01) + wy1*(wx0*a10 + wx1*a11)
    const double wx0 = 1.0 - dx, wy0 = 1.0 - dy;
    const double top = __dadd_rn(__dmul_rn(wx0, (double)R00), __dmul_rn(dx, (double)R01));
    const double bot = __dadd_rn(__dmul_rn(wx0, (double)R10), __dmul_rn(dx, (double)R11));
    o[(size_t)c * q.h_out] = (float)__dadd_rn(__dmul_rn(wy0, top), __dmul_rn(dy, bot));
  }
}

// ------------------------------------------------------------------------------------------
// generic helpers
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize1d(const float *__restrict__ in, size_t n_in, size_t n_out,
                                                  float *__restrict__ out) {
  const RsAxis ax = rs_axis(n_in, n_out);
  const bool same = (n_in == n_out);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (size_t)gridDim.x * blockDim.x) {
    double x = __dadd_rn(__dmul_rn(ax.sf, (double)(i + 1)), ax.off);
    x = fmin(fmax(x, 1.0), ax.n_in);
    double xf = floor(x);
    if (xf > ax.n_in - 1.0) xf -= 1.0;
    const double d = x - xf;
    const size_t k = (size_t)xf - 1;
    out[i] = same ? in[i] : rs_blend(in[k], in[k + 1], d);
  }
}

// imresize(image,(h_out,w_out)), column-major; lanes along output rows
__global__ __launch_bounds__(256) void k_resize2d(const float *__restrict__ in, int h_in, int w_in, int h_out,
                                                  int w_out, float *__restrict__ out) {
  const RsAxis ay = rs_axis((size_t)h_in, (size_t)h_out), ax = rs_axis((size_t)w_in, (size_t)w_out);
  const bool same = (h_in == h_out && w_in == w_out);
  const size_t total = (size_t)h_out * w_out;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i % (size_t)h_out), c = (int)(i / (size_t)h_out);
    if (same) { out[i] = in[i]; continue; }
    double dy, dx;
    const unsigned ky = rs_pos(ay, (double)(r + 1), dy), kx = rs_pos(ax, (double)(c + 1), dx);
    const float *p0 = in + (size_t)kx * h_in + ky, *p1 = p0 + h_in;
    const double a00 = p0[0], a10 = p0[1], a01 = p1[0], a11 = p1[1];
    const double wx0 = 1.0 - dx, wy0 = 1.0 - dy;
    const double top = __dadd_rn(__dmul_rn(wx0, a00), __dmul_rn(dx, a01));
    const double bot = __dadd_rn(__dmul_rn(wx0, a10), __dmul_rn(dx, a11));
    out[i] = (float)__dadd_rn(__dmul_rn(wy0, top), __dmul_rn(dy, bot));
  }
}

__global__ __launch_bounds__(256) void k_naive(const float *__restrict__ in, size_t n_out, unsigned up,
                                               float *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (size_t)gridDim.x * blockDim.x)
    out[i] = in[i / up];
}

// ------------------------------------------------------------------------------------------
// host-side tiling plans
// ------------------------------------------------------------------------------------------
struct RasterPlan { bool tiled; int TP, W; size_t lds; };

static RasterPlan plan_raster(size_t S, size_t P) {
  const double sf = (double)S / (double)P;
  RasterPlan pl{false, 0, 0, 0};
  for (int TP = 128; TP >= 4; TP >>= 1) {
    long W = (long)((double)(TP - 1) * sf) + 4;
    if (W <= 191) {
      pl.tiled = true; pl.TP = TP; pl.W = (int)W;
      pl.lds = (size_t)64 * (size_t)(W | 1) * 4 + 64 * 4;
      break;
    }
  }
  return pl;
}

struct DownPlan { bool fused; DownParams q; size_t lds; };

static DownPlan plan_down(size_t S, int y_t, int x_t, int h_out, int w_out) {
  DownPlan pl;
  pl.fused = false;
  pl.lds = 0;
  const double sf = (double)S / ((double)y_t * (double)x_t);
  const double sfy = (double)y_t / (double)h_out, sfx = (double)x_t / (double)w_out;
  const long NL = (long)(63.0 * sfy) + 3;
  static const int cand[] = {32, 16, 8, 4};
  for (int pass = 0; pass < 2 && !pl.fused; ++pass) {
    const size_t cap = pass == 0 ? 32 * 1024 : 60 * 1024;
    for (int TC : cand) {
      const long DPX = (long)((double)(TC - 1) * sfx) + 2;
      const long W = (long)((double)DPX * sf) + 4;
      const size_t lds = (size_t)NL * (size_t)(W | 1) * 4 + (size_t)NL * 4;
      if (lds <= cap && W < (1 << 20)) {
        pl.fused = true;
        pl.lds = lds;
        pl.q.S = (unsigned)S; pl.q.y_t = y_t; pl.q.x_t = x_t; pl.q.h_out = h_out; pl.q.w_out = w_out;
        pl.q.TC = TC; pl.q.NL = (int)NL; pl.q.W = (int)W; pl.q.tiles_c = (int)ceil_div((size_t)w_out, (size_t)TC);
        break;
      }
    }
  }
  return pl;
}

static int check_geom(tsdr_ctx *ctx, size_t S, int y_t, int x_t) {
  if (y_t <= 0 || x_t <= 0) return set_err(ctx, TSDR_EINVAL, "y_t and x_t must be positive");
  const size_t P = (size_t)y_t * (size_t)x_t;
  if (S >= (size_t(1) << 31) || P >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "frame larger than 2^31 samples/pixels");
  if (S != P && S < 2) return set_err(ctx, TSDR_EINVAL, "imresize needs at least 2 input samples");
  return TSDR_OK;
}

// sig_to_image for `frames` consecutive frames; in is real f32 (cplx=0) or interleaved IQ (cplx=1)
int raster_frames_d(tsdr_ctx *ctx, const float *in, int cplx, size_t in_stride, size_t S, int y_t, int x_t,
                    int frames, float *out, size_t out_stride) {
  int rc = check_geom(ctx, S, y_t, x_t);
  if (rc) return rc;
  if (frames <= 0) return TSDR_OK;
  const size_t P = (size_t)y_t * x_t;
  RasterPlan pl = plan_raster(S, P);
  if (pl.tiled) {
    const int tiles_p = (int)ceil_div((size_t)x_t, (size_t)pl.TP), tiles_l = (int)ceil_div((size_t)y_t, 64);
    dim3 grid((unsigned)(tiles_p * tiles_l), (unsigned)frames);
    if (cplx) {
      TSDR_LAUNCH(ctx, "raster_iq", (k_raster<true>), grid, dim3(256), pl.lds, in, in_stride, (unsigned)S, y_t, x_t, out,
                  out_stride, pl.TP, pl.W, tiles_p);
    } else {
      TSDR_LAUNCH(ctx, "raster_f32", (k_raster<false>), grid, dim3(256), pl.lds, in, in_stride, (unsigned)S, y_t, x_t, out,
                  out_stride, pl.TP, pl.W, tiles_p);
    }
  } else {
    dim3 grid((unsigned)stream_grid(ctx, ceil_div((size_t)y_t, 64) * 64 * (size_t)x_t), (unsigned)frames);
    if (cplx) {
      TSDR_LAUNCH(ctx, "raster_direct_iq", (k_raster_direct<true>), grid, dim3(256), 0, in, in_stride, (unsigned)S, y_t, x_t,
                  out, out_stride);
    } else {
      TSDR_LAUNCH(ctx, "raster_direct_f32", (k_raster_direct<false>), grid, dim3(256), 0, in, in_stride, (unsigned)S, y_t,
                  x_t, out, out_stride);
    }
  }
  return TSDR_OK;
}

int resize2d_d(tsdr_ctx *ctx, const float *img, int h_in, int w_in, int h_out, int w_out, float *out) {
  if (h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0) return set_err(ctx, TSDR_EINVAL, "resize2d: sizes must be positive");
  const bool same = (h_in == h_out && w_in == w_out);
  if (!same && (h_in < 2 || w_in < 2)) return set_err(ctx, TSDR_EINVAL, "resize2d: needs at least 2x2 input");
  TSDR_LAUNCH(ctx, "resize2d", k_resize2d, dim3(stream_grid(ctx, (size_t)h_out * w_out)), dim3(256), 0, img, h_in, w_in,
              h_out, w_out, out);
  return TSDR_OK;
}

// sig_to_image |> downgradeImage for `frames` frames, straight from the signal
int down_frames_d(tsdr_ctx *ctx, const float *in, int cplx, size_t in_stride, size_t S, int y_t, int x_t, int h_out,
                  int w_out, int frames, float *out, size_t out_stride) {
  int rc = check_geom(ctx, S, y_t, x_t);
  if (rc) return rc;
  if (h_out <= 0 || w_out <= 0) return set_err(ctx, TSDR_EINVAL, "output size must be positive");
  const bool same2 = (y_t == h_out && x_t == w_out);
  if (!same2 && (y_t < 2 || x_t < 2)) return set_err(ctx, TSDR_EINVAL, "imresize needs at least a 2x2 raster");
  if (frames <= 0) return TSDR_OK;
  // imresize returns a copy when the sizes already match: the raster IS the result
  if (same2) return raster_frames_d(ctx, in, cplx, in_stride, S, y_t, x_t, frames, out, out_stride);
  DownPlan pl = plan_down(S, y_t, x_t, h_out, w_out);
  if (pl.fused) {
    dim3 grid((unsigned)(ceil_div((size_t)h_out, 64) * (size_t)pl.q.tiles_c), (unsigned)frames);
    if (cplx) {
      TSDR_LAUNCH(ctx, "down_fused_iq", (k_down_fused<true>), grid, dim3(256), pl.lds, in, in_stride, pl.q, out, out_stride);
    } else {
      TSDR_LAUNCH(ctx, "down_fused_f32", (k_down_fused<false>), grid, dim3(256), pl.lds, in, in_stride, pl.q, out, out_stride);
    }
    return TSDR_OK;
  }
  // fallback: materialise each raster in workspace, then the generic 2-D resize
  const size_t P = (size_t)y_t * x_t;
  float *ras = (float *)ctx->scratch(WS_RASTER, P * 4);
  if (!ras) return TSDR_ENOMEM;
  for (int f = 0; f < frames; ++f) {
    rc = raster_frames_d(ctx, in + (size_t)f * in_stride * (cplx ? 2 : 1), cplx, in_stride, S, y_t, x_t, 1, ras, P);
    if (rc) return rc;
    rc = resize2d_d(ctx, ras, y_t, x_t, h_out, w_out, out + (size_t)f * out_stride);
    if (rc) return rc;
  }
  return TSDR_OK;
}

}  // namespace tsdr

using namespace tsdr;

extern "C" {

int tsdr_resize1d_d(tsdr_ctx *ctx, const float *sig, size_t n_in, size_t n_out, float *out) {
  if (!ctx || (n_out && (!sig || !out))) return TSDR_EINVAL;
  if (n_out == 0) return TSDR_OK;
  if (n_in != n_out && n_in < 2) return set_err(ctx, TSDR_EINVAL, "imresize needs at least 2 input samples");
  TSDR_LAUNCH(ctx, "resize1d", k_resize1d, dim3(stream_grid(ctx, n_out)), dim3(256), 0, sig, n_in, n_out, out);
  return TSDR_OK;
}

int tsdr_sig_to_image_d(tsdr_ctx *ctx, const float *sig, size_t S, int y_t, int x_t, float *img) {
  if (!ctx || !sig || !img) return TSDR_EINVAL;
  return raster_frames_d(ctx, sig, 0, S, S, y_t, x_t, 1, img, (size_t)y_t * x_t);
}

int tsdr_resize2d_d(tsdr_ctx *ctx, const float *img, int h_in, int w_in, int h_out, int w_out, float *out) {
  if (!ctx || !img || !out) return TSDR_EINVAL;
  return resize2d_d(ctx, img, h_in, w_in, h_out, w_out, out);
}

int tsdr_downgrade_d(tsdr_ctx *ctx, const float *img, int y_t, int x_t, float *out) {
  return tsdr_resize2d_d(ctx, img, y_t, x_t, TSDR_RENDER_H, TSDR_RENDER_W, out);
}

int tsdr_naive_resample_d(tsdr_ctx *ctx, const float *in, size_t n, int up, float *out) {
  if (!ctx || up < 1 || (n && (!in || !out))) return TSDR_EINVAL;
  if (n == 0) return TSDR_OK;
  TSDR_LAUNCH(ctx, "naive_resample", k_naive, dim3(stream_grid(ctx, n * (size_t)up)), dim3(256), 0, in, n * (size_t)up,
              (unsigned)up, out);
  return TSDR_OK;
}

int tsdr_resize1d(tsdr_ctx *ctx, const float *sig, size_t n_in, size_t n_out, float *out) {
  return host_map(ctx, sig, n_in * 4, out, n_out * 4,
                  [&](void *i, void *o) { return tsdr_resize1d_d(ctx, (const float *)i, n_in, n_out, (float *)o); });
}
int tsdr_sig_to_image(tsdr_ctx *ctx, const float *sig, size_t S, int y_t, int x_t, float *img) {
  if (y_t <= 0 || x_t <= 0) return TSDR_EINVAL;
  return host_map(ctx, sig, S * 4, img, (size_t)y_t * x_t * 4,
                  [&](void *i, void *o) { return tsdr_sig_to_image_d(ctx, (const float *)i, S, y_t, x_t, (float *)o); });
}
int tsdr_resize2d(tsdr_ctx *ctx, const float *img, int h_in, int w_in, int h_out, int w_out, float *out) {
  if (h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0) return TSDR_EINVAL;
  return host_map(ctx, img, (size_t)h_in * w_in * 4, out, (size_t)h_out * w_out * 4, [&](void *i, void *o) {
    return tsdr_resize2d_d(ctx, (const float *)i, h_in, w_in, h_out, w_out, (float *)o);
  });
}
int tsdr_downgrade(tsdr_ctx *ctx, const float *img, int y_t, int x_t, float *out) {
  return tsdr_resize2d(ctx, img, y_t, x_t, TSDR_RENDER_H, TSDR_RENDER_W, out);
}
int tsdr_naive_resample(tsdr_ctx *ctx, const float *in, size_t n, int up, float *out) {
  if (up < 1) return TSDR_EINVAL;
  return host_map(ctx, in, n * 4, out, n * (size_t)up * 4,
                  [&](void *i, void *o) { return tsdr_naive_resample_d(ctx, (const float *)i, n, up, (float *)o); });
}

}  // extern "C"
