// autocorr.hip -- Autocorrelations.jl on gfx950.
//
// calculate_autocorrelation (Autocorrelations.jl:23-37) is the CIRCULAR autocorrelation of the
// first n samples, r[k] = sum_m x[m] x[(m+k) mod n] = ifft(fft(x) .* conj(fft(x)))[k].  n is
// arbitrary (4e6 = 2^8*5^6 at 20 MS/s), so instead of a length-n transform the kernels compute the
// LINEAR autocorrelation a[k] = sum_m x[m] x[m+k] with zero padding to M = 2^ceil(log2(2n)) and fold
//     r[0] = a[0],   r[k] = a[k] + a[n-k]   (0 < k < n)
// which is the same sum regrouped.  x is real, so the length-M real transforms run as length-M/2
// complex ones: pack z[j] = x[2j] + i x[2j+1], one pointwise pass turns Z = FFT(z) into the packed
// spectrum Y of the (real, even) power |X|^2, and IFFT(Y)/2 is a[2j] + i a[2j+1].
//
// Multi-GPU: tsdr_autocorr_partial_d computes the partial sum over a range of m on one
// GPU as a zero-padded cross-correlation of the segment with segment+halo; ranks all-reduce the
// partial vectors (linear domain) and only then apply 10log10(abs2) via tsdr_autocorr_finish_d.
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "fft_dev.h"

namespace tsdr {

int get_tw(tsdr_ctx *ctx, int logN, TwTable **out);

__device__ inline float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ inline float ld_power(const float *x, int is_iq, size_t i) {
  if (is_iq) {
    float2 z = reinterpret_cast<const float2 *>(x)[i];
    return abs2_c(z.x, z.y);
  }
  return x[i];
}

// z[j] = x[2j] + i x[2j+1], zero beyond n; Mc complex outputs
__global__ __launch_bounds__(256) void k_ac_pack(const float *__restrict__ x, int is_iq, size_t n, size_t Mc,
                                                 float2 *__restrict__ z) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < Mc; j += (size_t)gridDim.x * blockDim.x) {
    const size_t i0 = 2 * j;
    float a = i0 < n ? ld_power(x, is_iq, i0) : 0.f;
    float b = i0 + 1 < n ? ld_power(x, is_iq, i0 + 1) : 0.f;
    z[j] = make_float2(a, b);
  }
}

// In place: Z (FFT of the packed real sequence, length Mc) -> Y, the packed spectrum whose
// inverse transform (times 1/2) is the real sequence with spectrum P[k] = |X[k]|^2.
//   X[k]    = E + W O,  X[Mc-k] = conj(E - W O),  E = (Z[k]+conj Z[Mc-k])/2, O = -i (Z[k]-conj Z[Mc-k])/2
//   Y[k]    = (P + P') + i conj(W) (P - P'),   Y[Mc-k] = (P + P') + i W (P - P'),   W = W_M^k
// tw_lo == nullptr: M = 2*Mc is not a power of two and W comes from tw_frac(k, 8/M).
__global__ __launch_bounds__(256) void k_ac_power(float2 *__restrict__ Z, size_t Mc, const float2 *__restrict__ tw_lo,
                                                  const float2 *__restrict__ tw_hi, int tw_h, double inv_m8) {
  const size_t half = Mc >> 1;
  const unsigned lomask = (1u << tw_h) - 1u;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k <= half; k += (size_t)gridDim.x * blockDim.x) {
    const size_t kk = k == 0 ? 0 : Mc - k;
    const float2 a = Z[k], b = Z[kk];
    const float2 E = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));  // (Z[k]-conj Z[kk])/2
    const float2 O = make_float2(D.y, -D.x);                                // -i*D
    const float2 W = tw_lo ? cmulf(tw_hi[k >> tw_h], tw_lo[k & lomask]) : tw_frac((unsigned)k, inv_m8);
    const float2 WO = cmulf(W, O);
    const float2 X0 = make_float2(E.x + WO.x, E.y + WO.y), X1 = make_float2(E.x - WO.x, E.y - WO.y);
    const float P0 = X0.x * X0.x + X0.y * X0.y, P1 = X1.x * X1.x + X1.y * X1.y;
    const float s = P0 + P1, d = P0 - P1;
    // i*conj(W)*d = d*(W.y, W.x)... i*(Wx - iWy) = Wy + i Wx
    Z[k] = make_float2(s + d * W.y, d * W.x);
    if (k != 0 && kk != k) Z[kk] = make_float2(s - d * W.y, d * W.x);  // i*W*d = d*(-Wy + i Wx)
  }
}

// out[i] = f(r[k0+i]), r[k] = a[k] + a[n-k] (k>0), a = linear autocorrelation (float view of y)
__global__ __launch_bounds__(256) void k_ac_fold(const float *__restrict__ a, size_t n, size_t k0, size_t cnt,
                                                 int log_scale, float *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x) {
    const size_t k = k0 + i;
    float r = a[k];
    if (k > 0) r += a[n - k];
    const float p = r * r;
    out[i] = log_scale ? 10.0f * log10f(p) : p;
  }
}

__global__ __launch_bounds__(256) void k_ac_finish(const float *__restrict__ corr, size_t k0, size_t cnt, int log_scale,
                                                   float *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x) {
    const float r = corr[k0 + i];
    const float p = r * r;
    out[i] = log_scale ? 10.0f * log10f(p) : p;
  }
}

// ---- partial (sharded) correlation --------------------------------------------------------
// z[j] = u[j] + i v[j]; u = x[m0 .. m0+cnt), v = x[(m0+j) mod n], j < cnt+n_lags-1; zero padded to M
__global__ __launch_bounds__(256) void k_pc_pack(const float *__restrict__ x, int is_iq, size_t n, size_t m0,
                                                 size_t cnt, size_t vlen, size_t M, float2 *__restrict__ z) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += (size_t)gridDim.x * blockDim.x) {
    float u = 0.f, v = 0.f;
    if (j < vlen) {
      size_t idx = m0 + j;
      if (idx >= n) idx %= n;
      v = ld_power(x, is_iq, idx);
      if (j < cnt) u = v;
    }
    z[j] = make_float2(u, v);
  }
}

// in place: Z = FFT(u + i v) -> C = conj(U) * V with U=(Z[k]+conj Z[M-k])/2, V=-i(Z[k]-conj Z[M-k])/2
__global__ __launch_bounds__(256) void k_pc_cross(float2 *__restrict__ Z, size_t M) {
  const size_t half = M >> 1;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k <= half; k += (size_t)gridDim.x * blockDim.x) {
    const size_t kk = k == 0 ? 0 : M - k;
    const float2 a = Z[k], b = Z[kk];
    const float2 U = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));
    const float2 V = make_float2(D.y, -D.x);
    // conj(U)*V
    const float2 C = make_float2(U.x * V.x + U.y * V.y, U.x * V.y - U.y * V.x);
    Z[k] = C;
    // U[M-k] = conj U[k], V[M-k] = conj V[k]  ->  C[M-k] = conj(C[k])
    if (k != 0 && k != half) Z[kk] = make_float2(C.x, -C.y);
  }
}

__global__ __launch_bounds__(256) void k_real_part(const float2 *__restrict__ c, size_t cnt, float *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x)
    out[i] = c[i].x;
}

// ---- argmax (findmax: first maximum, NaN maximal; argmax_key: fft_dev.h) ----------------------

// `key` must be zero on entry; `clear` (the slot the next launch will use) is zeroed here, so no memset launch
// separates two searches.  The last workgroup to arrive writes the winning key, then a sequence number, into pinned
// host memory: no copy launch, and the host polls the sequence word instead of sleeping in hipStreamSynchronize (whose
// interrupt wake-up costs hundreds of microseconds once a search is longer than the runtime's spin window).
__global__ __launch_bounds__(256) void k_argmax(const float *__restrict__ v, size_t n, unsigned long long *__restrict__ key,
                                                unsigned long long *__restrict__ clear, unsigned *__restrict__ arrived,
                                                unsigned long long *__restrict__ host_out, unsigned long long seq) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *clear = 0ull;
  unsigned long long best = 0ull;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long k = argmax_key(v[i], (unsigned)i);
    best = k > best ? k : best;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off, 64);
    best = o > best ? o : best;
  }
  // one atomic per workgroup (thousands of wavefronts on one 64-bit word serialise at ~12 ns each)
  __shared__ unsigned long long wbest[4];
  if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long b = wbest[0];
    for (int i = 1; i < 4; ++i) b = wbest[i] > b ? wbest[i] : b;
    atomicMax(key, b);
    __threadfence();
    if (atomicAdd(arrived, 1u) == gridDim.x - 1u) {
      *arrived = 0u;
      host_out[0] = atomicMax(key, 0ull);  // (an atomic read: every workgroup's maximum is in)
      __threadfence_system();
      __hip_atomic_store(&host_out[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // the host polls this word
    }
  }
}

// behind a last pass with a fused findmax: fold the slot words (clearing them for the next search) and hand the winner
// to the host like k_argmax does
__global__ __launch_bounds__(64) void k_amax_publish(unsigned long long *__restrict__ slots, unsigned long long *__restrict__ host_out,
                                                     unsigned long long seq) {
  unsigned long long b = threadIdx.x < kAmaxSlots ? slots[threadIdx.x] : 0ull;
  if (threadIdx.x < kAmaxSlots) slots[threadIdx.x] = 0ull;
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(b, off, 64);
    b = o > b ? o : b;
  }
  if (threadIdx.x == 0) {
    host_out[0] = b;
    __threadfence_system();
    __hip_atomic_store(&host_out[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static inline double jl_round(double v) { return nearbyint(v); }  // Julia round(): ties to even

// findmax fused into the autocorrelation's last pass: the window and where the result goes (see amax_begin / amax_wait)
struct AmaxReq {
  size_t lo = 0, cnt = 0;  // window out[lo .. lo + cnt)
  unsigned long long *key = nullptr, *clear = nullptr, *slots = nullptr;
  unsigned *arrived = nullptr;
  unsigned long long *host = nullptr;
  unsigned long long seq = 0;
  bool fused = false;      // out: the last pass delivered the maximum (else the caller runs k_argmax)
};

// shared core: x (real f32, or IQ whose abs2 is taken on the fly), first n samples
static int autocorr_core(tsdr_ctx *ctx, const float *x, int is_iq, size_t n, size_t k0, size_t cnt, int log_scale,
                         float *out, AmaxReq *amax = nullptr) {
  // n = 2*Mc with Mc = 2^a 3^b 5^c (the usual case: decimal sample rates, or a power of two): the circular
  // correlation of length n is transformed natively -- no zero padding, no fold, half the bytes (or less) of the
  // padded route below, which remains for every other n.
  const bool half_pow2 = (n & 1) == 0 && is_pow2(n / 2);
  const bool mixed_on = ctx->opt_ac_mixed != 0;
  if ((n & 1) == 0 && n > 1024 && (half_pow2 || (mixed_on && fft_mixed_ok(n / 2))) &&
      (!is_iq || (reinterpret_cast<uintptr_t>(x) & 15u) == 0)) {
    const size_t Mc = n / 2;
    float2 *z = (float2 *)ctx->scratch(WS_FFT_A, Mc * sizeof(float2));
    float2 *Z = (float2 *)ctx->scratch(WS_FFT_C, Mc * sizeof(float2));
    if (!z || !Z) return TSDR_ENOMEM;
    if (!half_pow2) {
      // mixed-radix route, three fusions: the first forward pass forms abs2 and packs; the first inverse pass forms the
      // packed power spectrum from Z while loading (no k_ac_power round trip); the last inverse pass writes
      // abs2 / 10log10 of the wanted lags straight to `out` (no k_ac_finish round trip)
      FftEpilogue epi;
      epi.out = out;
      epi.k0 = k0;
      epi.cnt = cnt;
      epi.log_scale = log_scale;
      if (amax && amax->cnt && amax->cnt < (size_t(1) << 32)) {
        epi.amax_keys = amax->slots; epi.amax_lo = amax->lo; epi.amax_cnt = amax->cnt;
        amax->fused = true;
      }
      // one launch carries the last forward pass, the power spectrum and the first inverse pass when the split allows
      if (ctx->opt_ac_fuse_mid) {
        bool done = false;
        int rc = fft_mixed_autocorr(ctx, reinterpret_cast<const float2 *>(x), is_iq ? SRC_IQPOW : SRC_REAL, n, Mc, Z, z,
                                    (float)(0.5 / (double)Mc), (k0 + cnt + 1) / 2, &epi, &done);
        if (rc || done) return rc;
      }
      int rc = fft_mixed(ctx, reinterpret_cast<const float2 *>(x), Z, Mc, 1, -1, 1.0f, is_iq ? SRC_IQPOW : SRC_REAL, n, 0);
      if (rc) return rc;
      return fft_mixed(ctx, Z, z, Mc, 1, +1, (float)(0.5 / (double)Mc), SRC_POWER, Mc, (k0 + cnt + 1) / 2, &epi);
    }
    int rc = fft_pow2(ctx, reinterpret_cast<const float2 *>(x), Z, ilog2(Mc), 1, -1, 1.0f, is_iq ? SRC_IQPOW : SRC_REAL, n, 0);
    if (rc) return rc;
    TSDR_LAUNCH(ctx, "ac_power", k_ac_power, dim3(stream_grid(ctx, Mc / 2 + 1)), dim3(256), 0, Z, Mc, (const float2 *)nullptr,
                (const float2 *)nullptr, 0, 4.0 / (double)Mc);
    rc = fft_pow2(ctx, Z, z, ilog2(Mc), 1, +1, (float)(0.5 / (double)Mc), SRC_C2C, n, (k0 + cnt + 1) / 2);
    if (rc) return rc;
    TSDR_LAUNCH(ctx, "ac_finish", k_ac_finish, dim3(stream_grid(ctx, cnt)), dim3(256), 0, reinterpret_cast<const float *>(z),
                k0, cnt, log_scale, out);
    return TSDR_OK;
  }
  const int logM = ilog2(2 * n);
  const size_t M = size_t(1) << logM, Mc = M >> 1;
  if (logM > 31) return set_err(ctx, TSDR_EINVAL, "autocorr: window too long");
  float2 *z = (float2 *)ctx->scratch(WS_FFT_A, Mc * sizeof(float2));
  float2 *Z = (float2 *)ctx->scratch(WS_FFT_C, Mc * sizeof(float2));
  if (!z || !Z) return TSDR_ENOMEM;
  TwTable *tw = nullptr;
  int rc = get_tw(ctx, logM, &tw);
  if (rc) return rc;
  // forward transform straight from the samples: the first pass packs (x[2j], x[2j+1]) -- forming abs2.(iq) on
  // the fly when asked -- and never reads the zero padding; no separate pack pass, no 33 MB round trip
  const bool aligned = !is_iq || (reinterpret_cast<uintptr_t>(x) & 15u) == 0;  // the IQ loader reads float4 pairs
  if (logM - 1 > 8 && aligned) {
    rc = fft_pow2(ctx, reinterpret_cast<const float2 *>(x), Z, logM - 1, 1, -1, 1.0f, is_iq ? SRC_IQPOW : SRC_REAL, n, 0);
  } else {
    TSDR_LAUNCH(ctx, "ac_pack", k_ac_pack, dim3(stream_grid(ctx, Mc)), dim3(256), 0, x, is_iq, n, Mc, z);
    rc = fft_pow2(ctx, z, Z, logM - 1, 1, -1, 1.0f, SRC_C2C, 0, 0);
  }
  if (rc) return rc;
  // only lags < n are ever folded: the last inverse pass stores a[0 .. n] = n/2 + 1 complex values.  With more
  // than one pass the power spectrum is formed by the first pass's loader (SRC_POWER); the in-place kernel remains
  // for single-pass lengths.
  if (logM - 1 > 8) {
    rc = fft_pow2(ctx, Z, z, logM - 1, 1, +1, (float)(0.5 / (double)Mc), SRC_POWER, Mc, n / 2 + 1);
  } else {
    TSDR_LAUNCH(ctx, "ac_power", k_ac_power, dim3(stream_grid(ctx, Mc / 2 + 1)), dim3(256), 0, Z, Mc, (const float2 *)tw->lo,
                (const float2 *)tw->hi, tw->h, 0.0);
    rc = fft_pow2(ctx, Z, z, logM - 1, 1, +1, (float)(0.5 / (double)Mc), SRC_C2C, 0, n / 2 + 1);
  }
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "ac_fold", k_ac_fold, dim3(stream_grid(ctx, cnt)), dim3(256), 0, reinterpret_cast<const float *>(z), n, k0,
              cnt, log_scale, out);
  return TSDR_OK;
}

int autocorr_args(tsdr_ctx *ctx, size_t len, double Fs, double minDelay, double maxDelay, size_t *n, size_t *k0,
                         size_t *cnt) {
  const double dmin = jl_round(minDelay * Fs), dmax = jl_round(maxDelay * Fs);
  // (round(x) |> Int of a non-finite or huge value is an InexactError in the reference; a negative index a BoundsError)
  if (!(dmin >= 0) || !(dmax >= 1) || dmax > 1e15 || dmin > 1e15) return set_err(ctx, TSDR_EBOUNDS, "autocorr: delay window out of range");
  const size_t indexMin = 1 + (size_t)dmin, indexMax = (size_t)dmax;
  *n = 2 * indexMax < len ? 2 * indexMax : len;                     // :27
  if (indexMax > *n) return set_err(ctx, TSDR_EBOUNDS, "autocorr: signal shorter than maxDelay*Fs (BoundsError at :33)");
  *k0 = indexMin - 1;
  *cnt = indexMin <= indexMax ? indexMax - indexMin + 1 : 0;
  return TSDR_OK;
}

}  // namespace tsdr

using namespace tsdr;

extern "C" {

static int autocorr_any_d(tsdr_ctx *ctx, const float *x, int is_iq, size_t len, double Fs, double minDelay, double maxDelay,
                          int log_scale, float *out, size_t *n_out) {
  if (!ctx || !x || !out) return TSDR_EINVAL;
  size_t n, k0, cnt;
  int rc = autocorr_args(ctx, len, Fs, minDelay, maxDelay, &n, &k0, &cnt);
  if (rc) return rc;
  if (n_out) *n_out = cnt;
  if (cnt == 0) return TSDR_OK;
  return autocorr_core(ctx, x, is_iq, n, k0, cnt, log_scale, out);
}

int tsdr_autocorr_d(tsdr_ctx *ctx, const float *x, size_t len, double Fs, double minDelay, double maxDelay, int log_scale,
                    float *out, size_t *n_out) {
  return autocorr_any_d(ctx, x, 0, len, Fs, minDelay, maxDelay, log_scale, out, n_out);
}

int tsdr_autocorr_iq_d(tsdr_ctx *ctx, const float *iq, size_t len, double Fs, double minDelay, double maxDelay,
                       int log_scale, float *out, size_t *n_out) {
  return autocorr_any_d(ctx, iq, 1, len, Fs, minDelay, maxDelay, log_scale, out, n_out);
}

int tsdr_autocorr(tsdr_ctx *ctx, const float *x, size_t len, double Fs, double minDelay, double maxDelay, int log_scale,
                  float *out, size_t *n_out) {
  if (!ctx || !x || !out) return TSDR_EINVAL;
  size_t n, k0, cnt;
  int rc = autocorr_args(ctx, len, Fs, minDelay, maxDelay, &n, &k0, &cnt);
  if (rc) return rc;
  if (n_out) *n_out = cnt;
  if (cnt == 0) return TSDR_OK;
  return host_map(ctx, x, n * 4, out, cnt * 4,
                  [&](void *i, void *o) { return autocorr_core(ctx, (const float *)i, 0, n, k0, cnt, log_scale, (float *)o); });
}

int tsdr_autocorr_partial_d(tsdr_ctx *ctx, const float *x, int is_iq, size_t n, size_t m0, size_t cnt, size_t n_lags,
                            float *part) {
  if (!ctx || !x || !part) return TSDR_EINVAL;
  if (n == 0 || cnt == 0 || n_lags == 0 || m0 >= n || cnt > n || n_lags > n) return set_err(ctx, TSDR_EINVAL, "autocorr_partial: bad range");
  const size_t vlen = cnt + n_lags - 1;
  const int logM = ilog2(vlen);
  if (logM > 31) return set_err(ctx, TSDR_EINVAL, "autocorr_partial: window too long");
  const size_t M = size_t(1) << logM;
  float2 *z = (float2 *)ctx->scratch(WS_FFT_A, M * sizeof(float2));
  float2 *Z = (float2 *)ctx->scratch(WS_FFT_C, M * sizeof(float2));
  if (!z || !Z) return TSDR_ENOMEM;
  TSDR_LAUNCH(ctx, "pc_pack", k_pc_pack, dim3(stream_grid(ctx, M)), dim3(256), 0, x, is_iq, n, m0, cnt, vlen, M, z);
  int rc = fft_pow2(ctx, z, Z, logM, 1, -1, 1.0f, SRC_C2C, 0, 0);
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "pc_cross", k_pc_cross, dim3(stream_grid(ctx, M / 2 + 1)), dim3(256), 0, Z, M);
  rc = fft_pow2(ctx, Z, z, logM, 1, +1, (float)(1.0 / (double)M), SRC_C2C, 0, n_lags);
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "pc_real", k_real_part, dim3(stream_grid(ctx, n_lags)), dim3(256), 0, (const float2 *)z, n_lags, part);
  return TSDR_OK;
}

int tsdr_autocorr_finish_d(tsdr_ctx *ctx, const float *corr, size_t k0, size_t cnt, int log_scale, float *out) {
  if (!ctx || (cnt && (!corr || !out))) return TSDR_EINVAL;
  if (cnt == 0) return TSDR_OK;
  TSDR_LAUNCH(ctx, "ac_finish", k_ac_finish, dim3(stream_grid(ctx, cnt)), dim3(256), 0, corr, k0, cnt, log_scale, out);
  return TSDR_OK;
}

int tsdr_zoom_bounds(size_t N, double Fs, double rate_min, double rate_max, size_t *pmin, size_t *pmax) {
  if (!pmin || !pmax) return TSDR_EINVAL;
  const double a = jl_round(1.0 / rate_max * Fs), b = jl_round(1.0 / rate_min * Fs);  // :46-47
  if (!(a >= 0) || !(b >= 0)) return TSDR_EBOUNDS;
  *pmin = a < (double)N ? (size_t)a : N;
  *pmax = b < (double)N ? (size_t)b : N;
  if (*pmin < 1) return TSDR_EBOUNDS;  // G[0:...] is a BoundsError
  return TSDR_OK;
}

// one findmax request: the key slot for this search (zero), the slot to clear for the next one, the sequence number
static int amax_begin(tsdr_ctx *ctx, AmaxReq *r) {
  if (!ctx->amax_keys) {
    TSDR_HIP(ctx, hipMalloc((void **)&ctx->amax_keys, 32 + 8 * kAmaxSlots));  // two key slots + the arrival counter + fused-findmax slots
    TSDR_HIP(ctx, hipMemset(ctx->amax_keys, 0, 32 + 8 * kAmaxSlots));
    TSDR_HIP(ctx, hipHostMalloc((void **)&ctx->amax_host, 64, hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(ctx->amax_host, 0, 64);
    TSDR_HIP(ctx, hipHostGetDevicePointer((void **)&ctx->amax_host_dev, ctx->amax_host, 0));
    ctx->amax_slot = 0;
  }
  // (the two key slots change roles only when k_argmax runs -- argmax_launch: it is the kernel that clears the other one.  A
  // findmax fused into an FFT pass uses `slots` and leaves both alone; toggling here as well handed the next k_argmax a slot
  // that still held an older search's key)
  // a fused search that failed between its last FFT pass and the publish launch (the only kernel that clears the slot words)
  // left keys behind that the next fused search would take its maximum against
  if (ctx->amax_dirty) {
    TSDR_HIP(ctx, hipMemsetAsync(ctx->amax_keys + 4, 0, 8 * kAmaxSlots, ctx->launch_stream));
    ctx->amax_dirty = false;
  }
  r->key = ctx->amax_keys + ctx->amax_slot;
  r->clear = ctx->amax_keys + (ctx->amax_slot ^ 1);
  r->arrived = reinterpret_cast<unsigned *>(ctx->amax_keys + 2);
  r->slots = ctx->amax_keys + 4;
  r->host = ctx->amax_host_dev;
  r->seq = ++ctx->amax_seq;
  return TSDR_OK;
}

// the value rides in the key's upper half (NaN canonicalised); the kernel delivers key and sequence number to pinned
// memory and the host polls the sequence word
static int amax_wait(tsdr_ctx *ctx, unsigned long long seq, size_t *idx, float *val) {
  bool seen = false;
  std::chrono::steady_clock::time_point t0;
  for (unsigned it = 1; !seen; ++it) {
    seen = __atomic_load_n(&ctx->amax_host[1], __ATOMIC_ACQUIRE) == seq;
    if (seen || (it & 0xFFFu) != 0) continue;
    if (hipStreamQuery(ctx->launch_stream) != hipErrorNotReady) break;  // finished or failed
    (void)hipGetLastError();
    // bounded like every host-side wait of the library (ctx.hip:wait_event)
    const auto now = std::chrono::steady_clock::now();
    if (it == 0x1000u) { t0 = now; continue; }
    if (ctx->opt_wait_ms > 0 && now - t0 > std::chrono::milliseconds(ctx->opt_wait_ms)) {
      ++ctx->wait_timeouts;
      return set_err(ctx, TSDR_EHIP, "argmax: the stream did not deliver the result within %d ms (bounded host wait; option wait_ms)", ctx->opt_wait_ms);
    }
  }
  if (!seen) {
    { int _w = tsdr::wait_stream(ctx, ctx->launch_stream, "argmax"); if (_w) return _w; }
    if (__atomic_load_n(&ctx->amax_host[1], __ATOMIC_ACQUIRE) != seq) return set_err(ctx, TSDR_EHIP, "argmax: result not delivered");
  }
  const unsigned long long h = *ctx->amax_host;
  *idx = (size_t)(0xFFFFFFFFu - (unsigned)(h & 0xFFFFFFFFull));
  if (val) {
    const unsigned u = (unsigned)(h >> 32);
    const unsigned bits = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;  // inverse of argmax_key's order-preserving map
    std::memcpy(val, &bits, 4);
  }
  return TSDR_OK;
}

static int argmax_launch(tsdr_ctx *ctx, const float *v, size_t n, const AmaxReq &r) {
  const int ablocks = (int)std::min<size_t>(ceil_div(n, 2048), 256);
  TSDR_LAUNCH(ctx, "argmax", k_argmax, dim3(ablocks), dim3(256), 0, v, n, r.key, r.clear, r.arrived, r.host, r.seq);
  ctx->amax_slot ^= 1;
  return TSDR_OK;
}

int tsdr_argmax_d(tsdr_ctx *ctx, const float *v, size_t n, size_t *idx, float *val) {
  if (!ctx || !v || !idx || n == 0) return TSDR_EINVAL;  // findmax of an empty collection throws
  if (n >= (size_t(1) << 32)) return set_err(ctx, TSDR_EINVAL, "argmax: vector too long");
  AmaxReq r;
  int rc = amax_begin(ctx, &r);
  if (rc) return rc;
  rc = argmax_launch(ctx, v, n, r);
  if (rc) return rc;
  return amax_wait(ctx, r.seq, idx, val);
}

int tsdr_autocorr_search_d(tsdr_ctx *ctx, const float *x, int is_iq, size_t len, double Fs, double minDelay, double maxDelay,
                           int log_scale, float *out, size_t *n_out, size_t win_lo, size_t win_cnt, size_t *idx, float *val) {
  if (!ctx || !x || !out || !idx) return TSDR_EINVAL;
  size_t n, k0, cnt;
  int rc = autocorr_args(ctx, len, Fs, minDelay, maxDelay, &n, &k0, &cnt);
  if (rc) return rc;
  if (n_out) *n_out = cnt;
  if (win_cnt == 0 || win_lo >= cnt || win_cnt > cnt - win_lo) return set_err(ctx, TSDR_EBOUNDS, "autocorr_search: window outside the lag vector");
  if (win_cnt >= (size_t(1) << 32)) return set_err(ctx, TSDR_EINVAL, "argmax: vector too long");
  AmaxReq r;
  rc = amax_begin(ctx, &r);
  if (rc) return rc;
  r.lo = win_lo;
  r.cnt = win_cnt;
  ctx->amax_dirty = true;   // until the publish launch (or the route without an epilogue) is known to have been enqueued
  rc = autocorr_core(ctx, x, is_iq, n, k0, cnt, log_scale, out, &r);
  if (rc) return rc;
  if (!r.fused) {  // routes whose last pass has no epilogue: the separate kernel
    ctx->amax_dirty = false;
    rc = argmax_launch(ctx, out + win_lo, win_cnt, r);
    if (rc) return rc;
  } else {
    TSDR_LAUNCH(ctx, "amax_publish", k_amax_publish, dim3(1), dim3(64), 0, r.slots, r.host, r.seq);
    ctx->amax_dirty = false;
  }
  return amax_wait(ctx, r.seq, idx, val);
}

}  // extern "C"
