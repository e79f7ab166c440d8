// fft.hip -- hand-written complex f32 FFT for gfx950 (FFTW.jl fft/ifft semantics: forward
// unnormalised, inverse scaled by 1/n).  No rocFFT/hipFFT.
//
// Power-of-two lengths: N = R_1 * R_2 * ... * R_p, every R_i <= 256, one kernel launch per factor
// ("pass").  With n = n_1*(R_2..R_p) + ... + n_p and k = k_1 + R_1*k_2 + R_1R_2*k_3 + ...:
//   pass i < p : for fixed (k_1..k_{i-1}, n_{i+1}..n_p) a length-R_i DFT over n_i (stride
//                B_i = R_{i+1}..R_p), then the twiddle W_{R_1..R_{i+1}}^{ n_{i+1} * (k_1 + R_1 k_2 + .. ) };
//                the result overwrites the same slots (layout [k_1]..[k_i][n_{i+1}]..[n_p]).
//   pass p     : length-R_p DFT over the contiguous n_p, written to natural order
//                k = K(k_1..k_{p-1}) + (R_1..R_{p-1}) * k_p.
// A workgroup owns a tile of R x T elements in LDS (T = 16..: T neighbouring columns are
// contiguous in HBM, so every global access is a >=128-byte run; the last pass tiles over k_1 so
// its transposed store is contiguous too).  Inside LDS the DFT is an in-place radix-4 DIF
// (one radix-2 stage first when log2 R is odd); the digit-reversed order is undone while storing.
// Twiddles come from f64-generated tables: W_4096 for the in-LDS stages, and a two-level table
// (W = hi[e >> h] * lo[e & mask]) for the inter-pass twiddles, so no sin/cos runs on the device
// and twiddle error stays at ~1.5 ulp for any N.
//
// Other lengths: Bluestein (chirp-z) on top of the power-of-two engine; chirp phases are reduced
// exactly (k^2 mod 2n in 64-bit integers) and evaluated in f64.
#include <cmath>

#include "common.h"

namespace tsdr {

__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ inline float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

enum { FFT_STRIDED = 0, FFT_LAST = 1, FFT_ROWS = 2 };

struct PassDesc {
  int mode, logR, logT, dir;
  float scale;
  unsigned long long N;  // elements per transform
  unsigned A, B, tiles;  // strided: outer count, inner size (= stride of the DFT index), B/T
  int logNtw, logBnext, logPprev, tw_h;
  int nprev;
  int logRprev[4];       // radices of the passes before this one (pass order)
  int logR1;
  unsigned Aprime, k1tiles;  // last pass: A / R_1, R_1 / T
  unsigned rows;             // rows mode: number of transforms
};

// position of frequency k after the in-place DIF stages
__device__ inline int fft_pos(int k, int logR) {
  int p = 0, rem = logR;
  if (logR & 1) { p |= (k & 1) << (rem - 1); k >>= 1; rem -= 1; }
  while (rem > 0) { p |= (k & 3) << (rem - 2); k >>= 2; rem -= 2; }
  return p;
}

// K(a): a = k_1*(R_2..R_m) + ... + k_m  ->  k_1 + R_1*k_2 + R_1R_2*k_3 + ...
__device__ inline unsigned digit_swap(unsigned a, int m, const int *logR) {
  unsigned K = 0;
  int wlog = 0;
  for (int j = 0; j < m; ++j) wlog += logR[j];
  for (int j = m - 1; j >= 0; --j) {
    wlog -= logR[j];                       // log2(P_{j}) with P_0 = 1 for j = 0
    const unsigned kj = a & ((1u << logR[j]) - 1u);
    a >>= logR[j];
    K += kj << wlog;
  }
  return K;
}

// in-place DIF on buf[idx*TP + t], idx < R, t < T; twR[e] = W_R^e (already conjugated for inverse)
__device__ inline void lds_fft(float2 *buf, const float2 *twR, int logR, int logT, int TP, int dir, int tid) {
  const int R = 1 << logR, T = 1 << logT;
  int L = R;
  if (logR & 1) {
    const int half = R >> 1;
    for (int w = tid; w < (half << logT); w += 256) {
      const int t = w & (T - 1), j = w >> logT;
      float2 a0 = buf[j * TP + t], a1 = buf[(j + half) * TP + t];
      buf[j * TP + t] = cadd(a0, a1);
      buf[(j + half) * TP + t] = cmul(csub(a0, a1), twR[j]);
    }
    L = half;
    __syncthreads();
  }
  while (L >= 4) {
    const int Q = L >> 2, logQ = 31 - __clz(Q), step = R / L;
    for (int w = tid; w < ((R >> 2) << logT); w += 256) {
      const int t = w & (T - 1), u = w >> logT;
      const int g = u >> logQ, j = u & (Q - 1);
      float2 *p0 = buf + (g * L + j) * TP + t;
      float2 *p1 = p0 + Q * TP, *p2 = p1 + Q * TP, *p3 = p2 + Q * TP;
      const float2 a0 = *p0, a1 = *p1, a2 = *p2, a3 = *p3;
      const float2 b0 = cadd(a0, a2), b1 = csub(a0, a2), b2 = cadd(a1, a3);
      const float2 d = csub(a1, a3);
      // forward: -i*d ; inverse: +i*d
      const float2 b3 = dir < 0 ? make_float2(d.y, -d.x) : make_float2(-d.y, d.x);
      const int e = j * step;
      *p0 = cadd(b0, b2);
      *p1 = cmul(cadd(b1, b3), twR[e]);
      *p2 = cmul(csub(b0, b2), twR[2 * e]);
      *p3 = cmul(csub(b1, b3), twR[3 * e]);
    }
    L = Q;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_fft_pass(const float2 *__restrict__ in, float2 *__restrict__ out, PassDesc d,
                                                  const float2 *__restrict__ tw_small, const float2 *__restrict__ tw_lo,
                                                  const float2 *__restrict__ tw_hi) {
  extern __shared__ float2 sm[];
  const int R = 1 << d.logR, T = 1 << d.logT, TP = T + 1;
  float2 *buf = sm;
  float2 *twR = sm + R * TP;
  const int tid = threadIdx.x;
  for (int e = tid; e < R; e += 256) {
    float2 w = tw_small[e << (12 - d.logR)];
    if (d.dir > 0) w.y = -w.y;
    twR[e] = w;
  }
  const unsigned bid = blockIdx.x;
  const int work = R << d.logT;
  if (d.mode == FFT_STRIDED) {
    const unsigned tile = bid % d.tiles, a = (bid / d.tiles) % d.A, b = bid / (d.tiles * d.A);
    const size_t base = (size_t)b * d.N + (size_t)a * R * d.B + (size_t)tile * T;
    for (int w = tid; w < work; w += 256) {
      const int t = w & (T - 1), j = w >> d.logT;
      buf[j * TP + t] = in[base + (size_t)j * d.B + t];
    }
    __syncthreads();
    lds_fft(buf, twR, d.logR, d.logT, TP, d.dir, tid);
    const unsigned Ka = digit_swap(a, d.nprev, d.logRprev);
    const unsigned mask = (d.logNtw >= 32) ? 0xFFFFFFFFu : ((1u << d.logNtw) - 1u);
    const unsigned lomask = (1u << d.tw_h) - 1u;
    for (int w = tid; w < work; w += 256) {
      const int t = w & (T - 1), k = w >> d.logT;
      float2 v = buf[fft_pos(k, d.logR) * TP + t];
      const unsigned nnext = (tile * T + (unsigned)t) >> d.logBnext;
      const unsigned e = (nnext * (Ka + ((unsigned)k << d.logPprev))) & mask;
      float2 tw = cmul(tw_hi[e >> d.tw_h], tw_lo[e & lomask]);
      if (d.dir > 0) tw.y = -tw.y;
      out[base + (size_t)k * d.B + t] = cmul(v, tw);
    }
  } else if (d.mode == FFT_LAST) {
    const unsigned kt = bid % d.k1tiles, arest = (bid / d.k1tiles) % d.Aprime, b = bid / (d.k1tiles * d.Aprime);
    const size_t tbase = (size_t)b * d.N;
    for (int w = tid; w < work; w += 256) {
      const int j = w & (R - 1), t = w >> d.logR;
      const size_t a = (size_t)(kt * T + (unsigned)t) * d.Aprime + arest;
      buf[j * TP + t] = in[tbase + a * R + j];
    }
    __syncthreads();
    lds_fft(buf, twR, d.logR, d.logT, TP, d.dir, tid);
    // digits k_2..k_{p-1} of arest -> their natural-order weight (already multiples of R_1)
    const unsigned Kp = digit_swap(arest, d.nprev - 1, d.logRprev + 1) << d.logR1;
    const size_t obase = tbase + (size_t)kt * T + Kp;
    for (int w = tid; w < work; w += 256) {
      const int t = w & (T - 1), k = w >> d.logT;
      float2 v = buf[fft_pos(k, d.logR) * TP + t];
      out[obase + ((size_t)k << d.logPprev) + t] = make_float2(v.x * d.scale, v.y * d.scale);
    }
  } else {
    const size_t row0 = (size_t)bid * T;
    for (int w = tid; w < work; w += 256) {
      const int j = w & (R - 1), t = w >> d.logR;
      const size_t row = row0 + t;
      buf[j * TP + t] = row < d.rows ? in[row * R + j] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    lds_fft(buf, twR, d.logR, d.logT, TP, d.dir, tid);
    for (int w = tid; w < work; w += 256) {
      const int k = w & (R - 1), t = w >> d.logR;
      const size_t row = row0 + t;
      if (row < d.rows) {
        float2 v = buf[fft_pos(k, d.logR) * TP + t];
        out[row * R + k] = make_float2(v.x * d.scale, v.y * d.scale);
      }
    }
  }
}

// ---- twiddle tables ------------------------------------------------------------------------
static int ensure_tw_small(tsdr_ctx *ctx) {
  if (ctx->tw_small) return TSDR_OK;
  std::vector<float2> h(4096);
  for (int e = 0; e < 4096; ++e) {
    const double ang = -2.0 * M_PI * (double)e / 4096.0;
    h[e] = make_float2((float)cos(ang), (float)sin(ang));
  }
  TSDR_HIP(ctx, hipMalloc((void **)&ctx->tw_small, 4096 * sizeof(float2)));
  TSDR_HIP(ctx, hipMemcpy(ctx->tw_small, h.data(), 4096 * sizeof(float2), hipMemcpyHostToDevice));
  return TSDR_OK;
}

int get_tw(tsdr_ctx *ctx, int logN, TwTable **out) {
  auto it = ctx->tw.find(logN);
  if (it != ctx->tw.end()) { *out = &it->second; return TSDR_OK; }
  TwTable t;
  t.logN = logN;
  t.h = (logN + 1) / 2;
  const size_t nlo = size_t(1) << t.h, nhi = size_t(1) << (logN - t.h);
  std::vector<float2> lo(nlo), hi(nhi);
  const long double N = ldexpl(1.0L, logN);
  for (size_t j = 0; j < nlo; ++j) {
    const long double ang = -2.0L * M_PIl * (long double)j / N;
    lo[j] = make_float2((float)cosl(ang), (float)sinl(ang));
  }
  for (size_t j = 0; j < nhi; ++j) {
    const long double ang = -2.0L * M_PIl * (long double)(j << t.h) / N;
    hi[j] = make_float2((float)cosl(ang), (float)sinl(ang));
  }
  TSDR_HIP(ctx, hipMalloc((void **)&t.lo, nlo * sizeof(float2)));
  TSDR_HIP(ctx, hipMalloc((void **)&t.hi, nhi * sizeof(float2)));
  TSDR_HIP(ctx, hipMemcpy(t.lo, lo.data(), nlo * sizeof(float2), hipMemcpyHostToDevice));
  TSDR_HIP(ctx, hipMemcpy(t.hi, hi.data(), nhi * sizeof(float2), hipMemcpyHostToDevice));
  auto ins = ctx->tw.emplace(logN, t);
  *out = &ins.first->second;
  return TSDR_OK;
}

// ---- power-of-two driver ---------------------------------------------------------------------
// in/out may alias.  Uses WS_FFT_B when more than one pass is needed: callers must not hand
// WS_FFT_B buffers to this function.
int fft_pow2(tsdr_ctx *ctx, const float2 *in, float2 *out, int logN, size_t batch, int dir, float scale) {
  if (logN < 0 || logN > 31) return set_err(ctx, TSDR_EINVAL, "fft: unsupported power-of-two length 2^%d", logN);
  if (batch == 0) return TSDR_OK;
  int rc = ensure_tw_small(ctx);
  if (rc) return rc;
  const size_t N = size_t(1) << logN;
  if (N * batch >= (size_t(1) << 40)) return set_err(ctx, TSDR_EINVAL, "fft: batch too large");
  if (logN == 0) {
    if (in != out) TSDR_HIP(ctx, hipMemcpyAsync(out, in, batch * sizeof(float2), hipMemcpyDeviceToDevice, ctx->stream));
    return TSDR_OK;
  }
  int p = logN <= 8 ? 1 : (logN + 7) / 8;
  int bits[4];
  for (int i = 0; i < p; ++i) bits[i] = logN / p + (i < logN % p ? 1 : 0);
  PassDesc d{};
  d.dir = dir < 0 ? -1 : 1;
  d.N = N;
  if (p == 1) {
    d.mode = FFT_ROWS;
    d.logR = logN;
    d.logT = 12 - logN;  // R*T = 4096
    d.scale = scale;
    d.rows = (unsigned)batch;
    if (batch >= (size_t(1) << 32)) return set_err(ctx, TSDR_EINVAL, "fft: too many rows");
    const int R = 1 << d.logR, T = 1 << d.logT;
    const size_t lds = ((size_t)R * (T + 1) + R) * sizeof(float2);
    const unsigned grid = (unsigned)ceil_div(batch, (size_t)T);
    TSDR_LAUNCH(ctx, "fft_rows", k_fft_pass, dim3(grid), dim3(256), lds, in, out, d, (const float2 *)ctx->tw_small,
                (const float2 *)nullptr, (const float2 *)nullptr);
    return TSDR_OK;
  }
  float2 *work = (float2 *)ctx->scratch(WS_FFT_B, N * batch * sizeof(float2));
  if (!work) return TSDR_ENOMEM;
  int logP = 0;  // log2(R_1..R_{i-1})
  const float2 *src = in;
  for (int i = 0; i < p - 1; ++i) {
    const int logB = logN - logP - bits[i];
    d.mode = FFT_STRIDED;
    d.logR = bits[i];
    d.logT = std::min(12 - bits[i], logB);
    d.scale = 1.0f;
    d.A = 1u << logP;
    d.B = 1u << logB;
    d.tiles = d.B >> d.logT;
    d.logNtw = logP + bits[i] + bits[i + 1];
    d.logBnext = logB - bits[i + 1];
    d.logPprev = logP;
    d.nprev = i;
    for (int j = 0; j < i; ++j) d.logRprev[j] = bits[j];
    TwTable *tw = nullptr;
    rc = get_tw(ctx, d.logNtw, &tw);
    if (rc) return rc;
    d.tw_h = tw->h;
    const int R = 1 << d.logR, T = 1 << d.logT;
    const size_t lds = ((size_t)R * (T + 1) + R) * sizeof(float2);
    const size_t grid = batch * d.A * d.tiles;
    if (grid >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "fft: grid too large");
    TSDR_LAUNCH(ctx, "fft_strided", k_fft_pass, dim3((unsigned)grid), dim3(256), lds, src, work, d,
                (const float2 *)ctx->tw_small, (const float2 *)tw->lo, (const float2 *)tw->hi);
    src = work;
    logP += bits[i];
  }
  d.mode = FFT_LAST;
  d.logR = bits[p - 1];
  d.logR1 = bits[0];
  d.logT = std::min(12 - bits[p - 1], bits[0]);
  d.scale = scale;
  d.logPprev = logP;
  d.nprev = p - 1;
  for (int j = 0; j < p - 1; ++j) d.logRprev[j] = bits[j];
  d.Aprime = 1u << (logP - bits[0]);
  d.k1tiles = 1u << (bits[0] - d.logT);
  {
    const int R = 1 << d.logR, T = 1 << d.logT;
    const size_t lds = ((size_t)R * (T + 1) + R) * sizeof(float2);
    const size_t grid = batch * d.Aprime * d.k1tiles;
    if (grid >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "fft: grid too large");
    TSDR_LAUNCH(ctx, "fft_last", k_fft_pass, dim3((unsigned)grid), dim3(256), lds, (const float2 *)work, out, d,
                (const float2 *)ctx->tw_small, (const float2 *)nullptr, (const float2 *)nullptr);
  }
  return TSDR_OK;
}

// ---- Bluestein ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chirp(float2 *__restrict__ chirp, unsigned long long n) {
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long e = (k * k) % (2ull * n);  // k < 2^32 so k*k fits
    const double ang = -M_PI * (double)e / (double)n;
    chirp[k] = make_float2((float)cos(ang), (float)sin(ang));
  }
}

__global__ __launch_bounds__(256) void k_blu_b(const float2 *__restrict__ chirp, size_t n, size_t L, float2 *__restrict__ b) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < L; k += (size_t)gridDim.x * blockDim.x) {
    float2 v = make_float2(0.f, 0.f);
    if (k < n) v = cconj(chirp[k]);
    else if (L - k < n) v = cconj(chirp[L - k]);
    b[k] = v;
  }
}

// a[b][k] = x[b][k]*chirp[k] (k<n) else 0.  conj_in: use conj(x) (inverse via conjugation)
__global__ __launch_bounds__(256) void k_blu_pre(const float *__restrict__ x, int is_complex, int conj_in, size_t n,
                                                 size_t L, size_t batch, const float2 *__restrict__ chirp,
                                                 float2 *__restrict__ a) {
  const size_t total = L * batch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / L, k = i - b * L;
    float2 v = make_float2(0.f, 0.f);
    if (k < n) {
      float2 s = is_complex ? reinterpret_cast<const float2 *>(x)[b * n + k] : make_float2(x[b * n + k], 0.f);
      if (conj_in) s.y = -s.y;
      v = cmul(s, chirp[k]);
    }
    a[i] = v;
  }
}

__global__ __launch_bounds__(256) void k_cmul_bcast(float2 *__restrict__ a, const float2 *__restrict__ b, size_t L,
                                                    size_t batch) {
  const size_t total = L * batch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    a[i] = cmul(a[i], b[i % L]);
}

__global__ __launch_bounds__(256) void k_blu_post(const float2 *__restrict__ a, size_t n, size_t L, size_t batch,
                                                  const float2 *__restrict__ chirp, int conj_out, float scale,
                                                  float2 *__restrict__ out) {
  const size_t total = n * batch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / n, k = i - b * n;
    float2 v = cmul(a[b * L + k], chirp[k]);
    if (conj_out) v.y = -v.y;
    out[i] = make_float2(v.x * scale, v.y * scale);
  }
}

__global__ __launch_bounds__(256) void k_r2c(const float *__restrict__ x, size_t n, float2 *__restrict__ z) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    z[i] = make_float2(x[i], 0.f);
}

static int get_bluestein(tsdr_ctx *ctx, size_t n, BluesteinPlan **out) {
  auto it = ctx->blu.find(n);
  if (it != ctx->blu.end()) { *out = &it->second; return TSDR_OK; }
  if (n >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "fft: length too large for the chirp-z path");
  BluesteinPlan pl;
  pl.n = n;
  pl.L = size_t(1) << ilog2(2 * n - 1);
  TSDR_HIP(ctx, hipMalloc((void **)&pl.chirp, n * sizeof(float2)));
  TSDR_HIP(ctx, hipMalloc((void **)&pl.bfft, pl.L * sizeof(float2)));
  TSDR_LAUNCH(ctx, "blu_chirp", k_chirp, dim3(stream_grid(ctx, n)), dim3(256), 0, pl.chirp, (unsigned long long)n);
  float2 *tmp = (float2 *)ctx->scratch(WS_FFT_D, pl.L * sizeof(float2));
  if (!tmp) return TSDR_ENOMEM;
  TSDR_LAUNCH(ctx, "blu_b", k_blu_b, dim3(stream_grid(ctx, pl.L)), dim3(256), 0, (const float2 *)pl.chirp, n, pl.L, tmp);
  int rc = fft_pow2(ctx, tmp, pl.bfft, ilog2(pl.L), 1, -1, 1.0f);
  if (rc) return rc;
  auto ins = ctx->blu.emplace(n, pl);
  *out = &ins.first->second;
  return TSDR_OK;
}

// General-length FFT of `batch` contiguous transforms.  x: real (is_complex=0) or interleaved
// complex; out: complex.  Uses WS_FFT_B (pow2 engine), WS_FFT_C, WS_FFT_D.
int fft_any(tsdr_ctx *ctx, const float *x, int is_complex, float2 *out, size_t n, size_t batch, int dir) {
  if (n == 0 || batch == 0) return TSDR_OK;
  const int d = dir < 0 ? -1 : 1;
  if (is_pow2(n)) {
    const float scale = d > 0 ? (float)(1.0 / (double)n) : 1.0f;
    const float2 *src = reinterpret_cast<const float2 *>(x);
    if (!is_complex) {
      TSDR_LAUNCH(ctx, "r2c", k_r2c, dim3(stream_grid(ctx, n * batch)), dim3(256), 0, x, n * batch, out);
      src = out;
    }
    return fft_pow2(ctx, src, out, ilog2(n), batch, d, scale);
  }
  BluesteinPlan *pl = nullptr;
  int rc = get_bluestein(ctx, n, &pl);
  if (rc) return rc;
  const size_t L = pl->L;
  float2 *a = (float2 *)ctx->scratch(WS_FFT_C, L * batch * sizeof(float2));
  float2 *a2 = (float2 *)ctx->scratch(WS_FFT_D, L * batch * sizeof(float2));
  if (!a || !a2) return TSDR_ENOMEM;
  const int inv = d > 0;
  TSDR_LAUNCH(ctx, "blu_pre", k_blu_pre, dim3(stream_grid(ctx, L * batch)), dim3(256), 0, x, is_complex, inv, n, L, batch,
              (const float2 *)pl->chirp, a);
  rc = fft_pow2(ctx, a, a2, ilog2(L), batch, -1, 1.0f);
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "blu_mul", k_cmul_bcast, dim3(stream_grid(ctx, L * batch)), dim3(256), 0, a2, (const float2 *)pl->bfft, L,
              batch);
  rc = fft_pow2(ctx, a2, a, ilog2(L), batch, +1, (float)(1.0 / (double)L));
  if (rc) return rc;
  const float scale = inv ? (float)(1.0 / (double)n) : 1.0f;
  TSDR_LAUNCH(ctx, "blu_post", k_blu_post, dim3(stream_grid(ctx, n * batch)), dim3(256), 0, (const float2 *)a, n, L, batch,
              (const float2 *)pl->chirp, inv, scale, out);
  return TSDR_OK;
}

}  // namespace tsdr

using namespace tsdr;

extern "C" {

int tsdr_fft_c2c_d(tsdr_ctx *ctx, const float *in, float *out, size_t n, size_t batch, int dir) {
  if (!ctx || ((n * batch) && (!in || !out))) return TSDR_EINVAL;
  return fft_any(ctx, in, 1, reinterpret_cast<float2 *>(out), n, batch, dir);
}

int tsdr_fft_c2c(tsdr_ctx *ctx, const float *in, float *out, size_t n, size_t batch, int dir) {
  if (!ctx) return TSDR_EINVAL;
  const size_t bytes = n * batch * 8;
  return host_map(ctx, in, bytes, out, bytes,
                  [&](void *i, void *o) { return tsdr_fft_c2c_d(ctx, (const float *)i, (float *)o, n, batch, dir); });
}

}  // extern "C"
