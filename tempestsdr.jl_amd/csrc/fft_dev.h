// fft_dev.h -- device helpers shared by the power-of-two (fft.hip) and mixed-radix (fft_mixed.hip) engines.
#pragma once
#include <utility>

#include "common.h"

namespace tsdr {

enum { FFT_STRIDED = 0, FFT_LAST = 1, FFT_ROWS = 2 };

// Complex arithmetic on the packed-f32 pipe (v_pk_*_f32: two f32 lanes per instruction at the rate of one): the FFT
// kernels issue VALU instructions 70 % of the time, and a complex product is two instructions this way instead of four
// (six without explicit FMAs -- the library is built with -ffp-contract=off for the bit-exact kernels).  The compiler
// finds the packed add on its own but not the product's source modifiers: t = (-a.y b.y, a.y b.x) takes the high half of
// a twice, b swapped with its new low half negated; r = (a.x, a.x) * b + t.  Same roundings as
// (fma(a.x, b.x, -(a.y b.y)), fma(a.x, b.y, a.y b.x)).
typedef float tsdr_v2f __attribute__((ext_vector_type(2)));
__device__ inline tsdr_v2f c2v(float2 a) { return tsdr_v2f{a.x, a.y}; }
__device__ inline float2 v2c(tsdr_v2f a) { return make_float2(a.x, a.y); }
__device__ inline float2 cmul(float2 a, float2 b) {
  const tsdr_v2f av = c2v(a), bv = c2v(b);
  tsdr_v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(av), "v"(bv));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
  return v2c(r);
}
__device__ inline float2 cadd(float2 a, float2 b) { return v2c(c2v(a) + c2v(b)); }
__device__ inline float2 csub(float2 a, float2 b) { return v2c(c2v(a) - c2v(b)); }
__device__ inline float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// p - i q = (p.x + q.y, p.y - q.x) and p + i q = (p.x - q.y, p.y + q.x): one packed add with q's halves swapped and one
// of them negated by source modifiers
__device__ inline float2 cadd_mi(float2 p, float2 q) {
  tsdr_v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(c2v(p)), "v"(c2v(q)));
  return v2c(r);
}
__device__ inline float2 cadd_pi(float2 p, float2 q) {
  tsdr_v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(c2v(p)), "v"(c2v(q)));
  return v2c(r);
}
// a * s and a * s + c for a real s (both halves by the same factor)
__device__ inline float2 cscale(float2 a, float s) { return v2c(c2v(a) * tsdr_v2f{s, s}); }
__device__ inline float2 cfma_s(float2 a, float s, float2 c) { return v2c(__builtin_elementwise_fma(c2v(a), tsdr_v2f{s, s}, c2v(c))); }
// a * (kr + i ki) for a compile-time constant: the constant sits in a scalar register pair (no VGPRs, no per-use moves)
__device__ inline float2 cmul_k(float2 a, float kr, float ki) {
  const tsdr_v2f av = c2v(a), kv = tsdr_v2f{kr, ki};
  tsdr_v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(av), "s"(kv));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(av), "s"(kv), "v"(t));
  return v2c(r);
}

// Forward DFT of N = 2, 4, 8 or 16 points held in registers: radix-2 decimation in frequency with the
// twiddles as literals (trivial ones special-cased), result in bit-reversed order (X[k] at v[brev<N>(k)]).
__device__ constexpr float kCos16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                        0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
__device__ constexpr float kSin16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                                        1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

template <int N, int I>
__device__ inline float2 mul_w(float2 d) {  // d * exp(-2*pi*i*I/N), 0 <= I < N/2
  constexpr int E = I * (16 / N);
  if constexpr (E == 0) {
    return d;
  } else if constexpr (E == 4) {
    return make_float2(d.y, -d.x);
  } else if constexpr (E == 2) {
    return cscale(cadd_mi(d, d), kCos16[2]);    // ((d.x + d.y), (d.y - d.x)) / sqrt 2
  } else if constexpr (E == 6) {
    return cscale(cadd_pi(d, d), -kCos16[2]);   // ((d.y - d.x), -(d.x + d.y)) / sqrt 2
  } else {
    return cmul_k(d, kCos16[E], -kSin16[E]);
  }
}

template <int N, int I>
__device__ inline void bfly(float2 *v) {
  const float2 a = v[I], b = v[I + N / 2];
  v[I] = cadd(a, b);
  v[I + N / 2] = mul_w<N, I>(csub(a, b));
}

template <int N, int... I>
__device__ inline void dif_stage(float2 *v, std::integer_sequence<int, I...>) {
  (bfly<N, I>(v), ...);
}

template <int N>
__device__ inline void reg_dft(float2 *v) {
  if constexpr (N >= 2) {
    dif_stage<N>(v, std::make_integer_sequence<int, N / 2>{});
    reg_dft<N / 2>(v);
    reg_dft<N / 2>(v + N / 2);
  }
}

template <int N>
__device__ constexpr int brev(int k) {
  int r = 0;
  for (int b = 1; b < N; b <<= 1) { r = (r << 1) | (k & 1); k >>= 1; }
  return r;
}

// Twiddles without tables.  A gathered table read costs one L1 tag cycle per distinct line per lane and made
// the passes gather-bound, so the phase is reduced to an octant o (0..7) plus x in [0, 1] -- exactly, in
// integers, for power-of-two N -- and sin/cos(pi/4 * x) come from degree-7/8 polynomials (Chebyshev
// interpolants in x^2; measured error <= 8.6e-8 absolute over all f32 x, i.e. <= 1.5 ulp).
// tw_octant: exp(-i*pi/4*(o + x)) for even o, exp(-i*pi/4*(o + 1 - x)) for odd o (the caller mirrors x).
__device__ inline float2 tw_octant(unsigned o, float x) {
  const float u = x * x;
  const float sn = fmaf(fmaf(fmaf(-3.595429006963968e-05f, u, 0.0024900068528950214f), u, -0.08074543625116348f), u,
                        0.7853981852531433f) * x;
  const float cs = fmaf(fmaf(fmaf(fmaf(3.5297971407999285e-06f, u, -0.0003259385994169861f), u, 0.015854325145483017f), u,
                             -0.3084251284599304f), u, 1.0f);
  const bool swap = ((o + 1u) >> 1) & 1u;
  float cr = swap ? sn : cs, sr = swap ? cs : sn;
  if ((o + 2u) & 4u) cr = -cr;
  if (!(o & 4u)) sr = -sr;
  return make_float2(cr, sr);
}

// exp(-2*pi*i*e / 2^L) for 0 <= e < 2^L, 2 <= L <= 31
__device__ inline float2 tw_unit(unsigned e, int L) {
  const unsigned q = e << (32 - L);
  const unsigned o = q >> 29;
  unsigned r = q & 0x1FFFFFFFu;
  if (o & 1u) r = 0x20000000u - r;
  return tw_octant(o, (float)r * 0x1p-29f);
}

// exp(-2*pi*i*q/2^32), and the phase q = floor(2^32 * e/N) (within 2 units) from inv = floor(2^64/N) = hi:lo
__device__ inline float2 tw_q32(unsigned q) {
  const unsigned o = q >> 29;
  unsigned r = q & 0x1FFFFFFFu;
  if (o & 1u) r = 0x20000000u - r;
  return tw_octant(o, (float)r * 0x1p-29f);
}
__device__ inline unsigned phase_q32(unsigned e, unsigned inv_hi, unsigned inv_lo) {
  return e * inv_hi + __umulhi(e, inv_lo);
}

__device__ inline float2 conj_if(float2 v, unsigned smask) {  // smask = 0x80000000 for the inverse transform
  return make_float2(v.x, __uint_as_float(__float_as_uint(v.y) ^ smask));
}

// element g of the (single) transform for the fused loaders of the first pass:
//   SRC_C2C   in[g]
//   SRC_REAL  (x[2g], x[2g+1])            real f32 sequence of src_n samples packed two per complex, zero beyond
//   SRC_IQPOW (|iq[2g]|^2, |iq[2g+1]|^2)  the same with x = abs2.(iq) formed on the fly (GUI.jl:70)
//   SRC_POWER Y[g] of the autocorrelation: `in` = Z, the length-src_n transform of a packed real
//             sequence; Y is the packed spectrum whose inverse transform (times 1/2) is the real sequence with
//             spectrum |X|^2 (see k_ac_power, which this loader replaces: one launch and a 2 x 8*Mc-byte round
//             trip less)
//   SRC_STUFF (x[g / up], 0) when up divides g, else 0: the zero-stuffed real input of the resampler (Resampler.jl:
//             upsampling by inserting up - 1 zeros), src_n = up -- k_stuff without its pass
//   SRC_RE0   (x[g], 0): a real f32 sequence as complex input (getSpectrum of a real signal) -- k_r2c without its pass
//   SRC_MULH  in[g] * aux[g], aux = ComplexF64 (double2 behind the float2 pointer): the resampler's frequency-domain filter
//             applied while the inverse transform loads -- evaluated in f64 and rounded to ComplexF32 (Resampler.jl:51-53)
enum { SRC_C2C = 0, SRC_REAL = 1, SRC_IQPOW = 2, SRC_POWER = 3, SRC_STUFF = 4, SRC_MULH = 5, SRC_RE0 = 6 };

// tw_frac below (needed by the SRC_POWER loader when M = 2*Mc is not a power of two)
__device__ inline float2 tw_frac(unsigned e, double inv_n8);

__device__ inline float2 fft_load(const float2 *__restrict__ in, int src_mode, unsigned long long src_n, size_t g,
                                  double inv_m8 = 0.0, const float2 *__restrict__ aux = nullptr) {
  if (src_mode == SRC_C2C) return in[g];
  if (src_mode == SRC_MULH) {
    const float2 x = in[g];
    const double2 h = reinterpret_cast<const double2 *>(aux)[g];
    return make_float2((float)((double)x.x * h.x - (double)x.y * h.y), (float)((double)x.x * h.y + (double)x.y * h.x));
  }
  if (src_mode == SRC_RE0) return make_float2(reinterpret_cast<const float *>(in)[g], 0.f);
  if (src_mode == SRC_STUFF) {
    const size_t q = g / (size_t)src_n;
    return q * (size_t)src_n == g ? make_float2(reinterpret_cast<const float *>(in)[q], 0.f) : make_float2(0.f, 0.f);
  }
  if (src_mode == SRC_POWER) {
    const size_t Mc = (size_t)src_n;
    const float2 a = in[g], b = in[g ? Mc - g : 0];
    const float2 E = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));  // (Z[g] - conj Z[Mc-g]) / 2
    const float2 O = make_float2(D.y, -D.x);                                // -i*D
    // W_M^g, M = 2*Mc: exact octant split for a power of two, f64 phase (inv_m8 = 8/M) otherwise
    const float2 W = inv_m8 == 0.0 ? tw_unit((unsigned)g, 64 - __clzll((unsigned long long)Mc)) : tw_frac((unsigned)g, inv_m8);
    const float2 WO = cmul(W, O);
    const float2 X0 = make_float2(E.x + WO.x, E.y + WO.y), X1 = make_float2(E.x - WO.x, E.y - WO.y);
    const float P0 = X0.x * X0.x + X0.y * X0.y, P1 = X1.x * X1.x + X1.y * X1.y;
    const float sum = P0 + P1, dif = P0 - P1;
    return make_float2(sum + dif * W.y, dif * W.x);  // (P + P') + i conj(W) (P - P')
  }
  const unsigned long long i0 = 2ull * g;
  if (i0 >= src_n) return make_float2(0.f, 0.f);  // zero padding is never read
  if (src_mode == SRC_REAL) {
    const float *x = reinterpret_cast<const float *>(in);
    return make_float2(x[i0], i0 + 1 < src_n ? x[i0 + 1] : 0.f);
  }
  const float4 z = reinterpret_cast<const float4 *>(in)[g];  // iq[2g], iq[2g+1]
  return make_float2(z.x * z.x + z.y * z.y, i0 + 1 < src_n ? z.z * z.z + z.w * z.w : 0.f);
}

// exp(-2*pi*i*e/N) for any N: inv_n8 = 8/N in f64, 0 <= e < N.  The phase e/N is formed in f64 (relative error
// 2^-52), so the octant split is exact to ~1e-16 of a turn.
__device__ inline float2 tw_frac(unsigned e, double inv_n8) {
  const double ph = (double)e * inv_n8;
  const unsigned o = (unsigned)ph;
  const double f = ph - (double)o;
  const float x = (float)((o & 1u) ? 1.0 - f : f);
  return tw_octant(o & 7u, x);
}

// Epilogues of the last pass (o = index of the complex output x within its transform):
//   EPI_AC     autocorrelation (Autocorrelations.jl:33-36): x holds the real lags (2o, 2o+1); lags k0 <= k < k0+cnt leave
//              as abs2 (and 10log10) in f32 -- k_ac_finish without the round trip
//   EPI_REAL   out[o] = gain * real(x) for o < cnt: the resampler's `2*upCoeff*real.(ifft)` (Resampler.jl) without k_real_scale
//   EPI_SPEC   getSpectrum (GetSpectrum.jl:21-30): out[fftshift position of o] = abs2(x) or 10log10(abs2(x)); cnt = N, k0 = N div 2
//              With amax_keys set the same pass also finds findmax over out[amax_lo .. amax_lo + amax_cnt) (GUI.jl:79 on
//              the zoom window, Autocorrelations.jl:42-53): every thread keeps the best packed key of what it stores and a
//              workgroup contributes ONE relaxed atomicMax to one of kAmaxSlots words (epi_argmax_finish) -- no fence, no
//              arrival counter: a release fence per workgroup writes its freshly stored lags back through L2 and tripled
//              the pass (measured 12 -> 38 us at C2).  A one-wavefront launch behind the pass folds the slots and hands
//              the result to the host (k_amax_publish); the separate pass over the lags is gone.
enum { EPI_NONE = 0, EPI_AC = 1, EPI_REAL = 2, EPI_SPEC = 3 };
struct FftEpilogue {
  float *out = nullptr;
  unsigned long long k0 = 0, cnt = 0;
  int log_scale = 0;
  int kind = EPI_AC;
  float gain = 1.0f;
  // fused findmax (EPI_AC only)
  unsigned long long *amax_keys = nullptr;  // kAmaxSlots device words, zero on entry
  unsigned long long amax_lo = 0, amax_cnt = 0;
};
constexpr int kAmaxSlots = 16;

// findmax: first maximum, NaN maximal -- value bits made order-preserving, index complemented so the smallest wins ties
__device__ inline unsigned long long argmax_key(float v, unsigned idx) {
  unsigned u = (v != v) ? 0x7FC00000u : __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

__device__ inline void epilogue_store(const FftEpilogue &e, size_t o, float2 x, unsigned long long &best) {
  if (e.kind == EPI_REAL) {
    if (o < e.cnt) e.out[o] = e.gain * x.x;
    return;
  }
  if (e.kind == EPI_SPEC) {
    if (o < e.cnt) {
      size_t j = o + e.k0;
      if (j >= e.cnt) j -= e.cnt;
      const float p = x.x * x.x + x.y * x.y;
      e.out[j] = e.log_scale ? 10.0f * log10f(p) : p;
    }
    return;
  }
  const unsigned long long i0 = 2ull * o - e.k0, i1 = i0 + 1ull;  // (wraps to huge when below k0)
  if (i0 < e.cnt) {
    const float p = x.x * x.x, v = e.log_scale ? 10.0f * log10f(p) : p;
    e.out[i0] = v;
    if (i0 - e.amax_lo < e.amax_cnt) { const unsigned long long k = argmax_key(v, (unsigned)(i0 - e.amax_lo)); best = k > best ? k : best; }
  }
  if (i1 < e.cnt) {
    const float p = x.y * x.y, v = e.log_scale ? 10.0f * log10f(p) : p;
    e.out[i1] = v;
    if (i1 - e.amax_lo < e.amax_cnt) { const unsigned long long k = argmax_key(v, (unsigned)(i1 - e.amax_lo)); best = k > best ? k : best; }
  }
}
__device__ inline void epilogue_store(const FftEpilogue &e, size_t o, float2 x) {
  unsigned long long best = 0ull;
  epilogue_store(e, o, x, best);
}

// End of a last pass with a fused findmax: called by every thread of a workgroup (wkeys: one LDS word per wavefront).
__device__ inline void epi_argmax_finish(const FftEpilogue &e, unsigned long long best, unsigned long long *wkeys, int nwaves) {
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off, 64);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63) == 0) wkeys[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long b = wkeys[0];
    for (int i = 1; i < nwaves; ++i) b = wkeys[i] > b ? wkeys[i] : b;
    if (b) __hip_atomic_fetch_max(e.amax_keys + (blockIdx.x & (kAmaxSlots - 1)), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// host entry points of the two engines (fft.hip, fft_mixed.hip).  src_mode / src_n / src_aux: fused loader of the first
// pass; keep: complex outputs per transform the caller will look at (0 = all); epi: epilogue of the last pass.  Loaders
// and epilogues need one transform (batch == 1) of more than one pass.
int fft_pow2(tsdr_ctx *ctx, const float2 *in, float2 *out, int logN, size_t batch, int dir, float scale, int src_mode,
             size_t src_n, size_t keep, const FftEpilogue *epi = nullptr, const float2 *src_aux = nullptr);
int fft_mixed(tsdr_ctx *ctx, const float2 *in, float2 *out, size_t N, size_t batch, int dir, float scale, int src_mode,
              size_t src_n, size_t keep, const FftEpilogue *epi = nullptr, const float2 *src_aux = nullptr);
int fft_mixed_autocorr(tsdr_ctx *ctx, const float2 *x, int src_mode, size_t src_n, size_t Mc, float2 *Zbuf, float2 *zbuf,
                       float scale, size_t keep, const FftEpilogue *epi, bool *done);
unsigned fft_rows_welch_parts(tsdr_ctx *ctx);
int fft_rows_welch(tsdr_ctx *ctx, const float *sig, int is_complex, size_t N, size_t nbSeg, float *part, unsigned *nparts, bool *did);
int fft_rows1024(tsdr_ctx *ctx, const float2 *in, float2 *out, size_t batch, int dir, float scale);   // spectrum.hip
int fft_rows_store(tsdr_ctx *ctx, const float2 *in, float2 *out, size_t N, size_t batch, int dir, float scale, bool *did);
int fft_rows_waterfall(tsdr_ctx *ctx, const float *sig, int is_complex, size_t N, size_t nbSeg, double *wf, bool *did);
bool fft_mixed_ok(size_t N);
int fft_passes(size_t N);
int ensure_tw_small(tsdr_ctx *ctx);  // builds ctx->tw_small: W_4096^e for e < 4096  // launches a length-N transform takes (0: not a 2^a 3^b 5^c length)

}  // namespace tsdr
