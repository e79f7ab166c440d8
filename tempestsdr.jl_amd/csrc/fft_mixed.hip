// fft_mixed.hip -- complex f32 FFT for lengths N = 2^a * 3^b * 5^c (FFTW.jl fft/ifft semantics).
//
// Sample rates in this domain are decimal (20 MS/s, 200 MS/s ...), so the transform lengths the reference asks
// for -- two refresh periods of samples for the autocorrelation (Autocorrelations.jl:27), spectrum windows,
// resampler blocks -- are of the form 2^a * 5^b far more often than 2^a (4e6 = 2^8 * 5^6).  Padding such a length
// to a power of two doubles the bytes every pass moves; this engine transforms it natively.
//
// Same multi-pass structure as fft.hip: N = R_1 * ... * R_p with every R_i <= 256, one launch per factor, pass
// i < p a length-R_i DFT at stride B_i = R_{i+1}..R_p followed by the twiddle W_{P_i R_i R_{i+1}}^(n_{i+1} K),
// the last pass contiguous and written in natural order.  The differences:
//   * the radices are arbitrary, so tile coordinates come from integer divisions by launch constants instead of
//     shifts, tiles are ragged at the end of a row, and twiddle phases e/N are 32-bit fixed-point fractions from a
//     64-bit reciprocal (phase_q32);
//   * inside a tile the length-R DFT is an in-place decimation-in-frequency over LDS with stage radices
//     from {2, 3, 4, 5, 8, 9, 10, 16, 25} (register DFTs), and the mixed-radix digit reversal is undone through a
//     position -> frequency map while storing.
// The factors are ordered so that the last one carries the factors of two: every stride B_i is then a multiple
// of 16 elements and the T-wide runs of a tile stay 128-byte aligned.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "fft_dev.h"

namespace tsdr {

enum { MIX_MAX_PASS = 6, MIX_MAX_STAGE = 8 };

struct MixDesc {
  int mode, dir, logT, nst;
  float scale;
  unsigned R;
  unsigned char rad[MIX_MAX_STAGE];  // stage radices, product R
  unsigned long long N;              // elements per transform
  unsigned A, B, tiles;              // strided: outer count, inner size (= stride of the DFT index), ceil(B/T)
  unsigned Bnext, Pprev;             // B / R_{i+1};  R_1..R_{i-1}
  unsigned ntw_hi, ntw_lo;           // floor(2^64 / (Pprev * R * R_{i+1})): twiddle phase e/Ntw as a 32-bit fraction
  unsigned r_hi, r_lo;               // floor(2^64 / R)
  int nprev;
  unsigned Rprev[MIX_MAX_PASS], Wprev[MIX_MAX_PASS];  // radices of the earlier passes and their weights in k
  unsigned R1, Aprime, k1tiles;      // last pass: first radix, A / R_1, ceil(R_1 / T)
  unsigned rows;                     // rows mode: number of transforms
  int src_mode;
  unsigned long long src_n, keep;
  int tw_sets;                       // strided, two-step kernels: inter-pass twiddle sets held in LDS (0: none)
  const float2 *twg;                 // strided, Bnext == 1: W_{R*Rnext}^(col*k) at [k * B + col] (else nullptr)
  const float2 *src_aux;             // SRC_MULH: the factor array
  double src_w8;                     // SRC_POWER with M = 2*src_n not a power of two: 8/M (else 0)
  FftEpilogue epi;                   // last pass: autocorrelation epilogue when epi.out != nullptr
  // rows mode as getWelch's accumulator (GetSpectrum.jl:36-52): nothing is stored per transform; every workgroup walks
  // tiles blockIdx.x, blockIdx.x + gridDim.x, ... of `rows` segments, adds abs2 of every spectrum it forms in registers and
  // leaves ONE partial power spectrum, acc[blockIdx.x * R + k] (natural frequency order).  rows_real: the rows are real f32.
  float *acc;
  int rows_real;
  // ... or getWaterfall's writer (GetSpectrum.jl:54-66; three-step kernels only): Float64(abs2) of every spectrum straight from
  // the registers to wf[segment * R + fftshift position], acc unused (non-null only to select the branch)
  double *wf;
  // ... or plain batched row transforms (tsdr_fft_c2c with batch > 1, three-step kernels only): rows_out[row * R + k] = scale * X[k],
  // either direction; may alias the input (a tile's rows are all loaded before any of them is stored)
  float2 *rows_out;
};

// a = k_1*(R_2..R_m) + ... + k_m  ->  k_1*W_1 + ... + k_m*W_m  (uniform per workgroup: scalar code)
__device__ inline unsigned digit_swap_g(unsigned a, int m, const unsigned *R, const unsigned *W) {
  unsigned K = 0;
  for (int j = m - 1; j >= 0; --j) {
    const unsigned q = a / R[j];
    K += (a - q * R[j]) * W[j];
    a = q;
  }
  return K;
}

// x * exp(-2*pi*i*K/N) with the constant taken from a literal table
__device__ constexpr float kCos25[17] = {1.f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899655f,
    0.30901699437494745f, 0.062790519529313527f, -0.1873813145857246f, -0.42577929156507272f, -0.63742398974868975f,
    -0.80901699437494734f, -0.92977648588825135f, -0.99211470131447776f, -0.99211470131447788f, -0.92977648588825146f,
    -0.80901699437494778f, -0.63742398974868952f};
__device__ constexpr float kSin25[17] = {0.f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868862f, 0.84432792550201508f,
    0.95105651629515353f, 0.99802672842827156f, 0.98228725072868872f, 0.90482705246601947f, 0.77051324277578925f,
    0.58778525229247325f, 0.36812455268467814f, 0.12533323356430454f, -0.12533323356430429f, -0.36812455268467792f,
    -0.58778525229247269f, -0.77051324277578936f};
__device__ constexpr float kCos10[5] = {1.f, 0.80901699437494745f, 0.30901699437494745f, -0.30901699437494734f, -0.80901699437494734f};
__device__ constexpr float kSin10[5] = {0.f, 0.58778525229247314f, 0.95105651629515353f, 0.95105651629515364f, 0.58778525229247325f};
__device__ constexpr float kCos20[13] = {1.0f, 0.9510565162951535f, 0.8090169943749475f, 0.5877852522924731f, 0.30901699437494745f, 6.123233995736766e-17f, -0.30901699437494734f, -0.587785252292473f, -0.8090169943749473f, -0.9510565162951535f, -1.0f, -0.9510565162951538f, -0.8090169943749476f};
__device__ constexpr float kSin20[13] = {0.0f, 0.3090169943749474f, 0.5877852522924731f, 0.8090169943749475f, 0.9510565162951535f, 1.0f, 0.9510565162951536f, 0.8090169943749475f, 0.5877852522924732f, 0.3090169943749475f, 1.2246467991473532e-16f, -0.3090169943749469f, -0.587785252292473f};
__device__ constexpr float kCos9[5] = {1.f, 0.76604444311897801f, 0.17364817766693041f, -0.5f, -0.93969262078590832f};
__device__ constexpr float kSin9[5] = {0.f, 0.64278760968653925f, 0.98480775301220802f, 0.86602540378443871f, 0.34202014332566888f};

template <int N, int K>
__device__ inline float2 mul_c(float2 x) {
  if constexpr (K == 0) {
    return x;
  } else {
    constexpr float c = N == 25 ? kCos25[K] : N == 20 ? kCos20[K] : N == 10 ? kCos10[K] : kCos9[K];
    constexpr float sn = N == 25 ? kSin25[K] : N == 20 ? kSin20[K] : N == 10 ? kSin10[K] : kSin9[K];
    return cmul_k(x, c, -sn);
  }
}

// forward DFT of r points, natural order, in place
template <int r>
__device__ inline void dft_nat(float2 *x);

// r = P*Q as two register steps: n = Q'*a + b ... X[c + P*d] = sum_b W_Q^(b d) W_r^(b c) sum_a W_P^(a c) x[Q*a + b]
template <int P, int Q, int B, int... C>
__device__ inline void two_step_col(const float2 *x, float2 *y, std::integer_sequence<int, C...>) {
  float2 t[P];
#pragma unroll
  for (int a = 0; a < P; ++a) t[a] = x[Q * a + B];
  dft_nat<P>(t);
  ((y[Q * C + B] = mul_c<P * Q, B * C>(t[C])), ...);
}
template <int P, int Q, int... B>
__device__ inline void two_step_cols(const float2 *x, float2 *y, std::integer_sequence<int, B...>) {
  (two_step_col<P, Q, B>(x, y, std::make_integer_sequence<int, P>{}), ...);
}
template <int P, int Q>
__device__ inline void dft_two_step(float2 *x) {
  float2 y[P * Q];
  two_step_cols<P, Q>(x, y, std::make_integer_sequence<int, Q>{});  // y[Q*c + b]
#pragma unroll
  for (int c = 0; c < P; ++c) {
    float2 t[Q];
#pragma unroll
    for (int b = 0; b < Q; ++b) t[b] = y[Q * c + b];
    dft_nat<Q>(t);
#pragma unroll
    for (int d = 0; d < Q; ++d) x[c + P * d] = t[d];
  }
}

template <int r>
__device__ inline void dft_nat(float2 *x) {
  if constexpr (r == 3) {
    const float s = 0.86602540378443865f;
    const float2 t = cadd(x[1], x[2]), d = csub(x[1], x[2]);
    const float2 m = cfma_s(t, -0.5f, x[0]);
    const float2 q = cscale(d, s);
    x[0] = cadd(x[0], t);
    x[1] = cadd_mi(m, q);  // m - i s d
    x[2] = cadd_pi(m, q);
  } else if constexpr (r == 5) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const float2 a1 = cadd(x[1], x[4]), a2 = cadd(x[2], x[3]), b1 = csub(x[1], x[4]), b2 = csub(x[2], x[3]);
    // (packed: every line below is one instruction per complex value)
    const float2 p1 = cadd(x[0], cfma_s(a1, c1, cscale(a2, c2)));
    const float2 p2 = cadd(x[0], cfma_s(a1, c2, cscale(a2, c1)));
    const float2 q1 = cfma_s(b1, s1, cscale(b2, s2));
    const float2 q2 = cfma_s(b1, s2, cscale(b2, -s1));
    x[0] = cadd(x[0], cadd(a1, a2));
    x[1] = cadd_mi(p1, q1);  // p1 - i q1
    x[4] = cadd_pi(p1, q1);
    x[2] = cadd_mi(p2, q2);
    x[3] = cadd_pi(p2, q2);
  } else if constexpr (r == 25) {
    dft_two_step<5, 5>(x);
  } else if constexpr (r == 20) {
    dft_two_step<5, 4>(x);
  } else if constexpr (r == 10) {
    dft_two_step<5, 2>(x);
  } else if constexpr (r == 9) {
    dft_two_step<3, 3>(x);
  } else {
    reg_dft<r>(x);
    float2 y[r];
#pragma unroll
    for (int c = 0; c < r; ++c) y[c] = x[brev<r>(c)];
#pragma unroll
    for (int c = 0; c < r; ++c) x[c] = y[c];
  }
}

// One DIF stage of radix r on sub-transforms of length L: for every group g and j < Q = L/r
//   y_c = sum_m x[g*L + j + m*Q] W_r^(m c),   x[g*L + j + c*Q] <- y_c * W_L^(j c)
template <int r>
__device__ inline void mix_stage(float2 *buf, const float2 *twR, unsigned R, unsigned L, int logT, int TP, int tid) {
  const unsigned Q = L / r, step = R / L;
  const unsigned nb = (R / r) << logT;
  const unsigned T = 1u << logT;
  const float invQ = 1.0f / (float)Q;
  for (unsigned w = tid; w < nb; w += 256) {
    const unsigned t = w & (T - 1u), u = w >> logT;
    const unsigned g = (unsigned)(((float)u + 0.5f) * invQ), j = u - g * Q;  // u < 4096: exact
    float2 *p = buf + (g * L + j) * TP + t;
    float2 x[r];
#pragma unroll
    for (int m = 0; m < r; ++m) x[m] = p[m * Q * TP];
    dft_nat<r>(x);
    p[0] = x[0];
    if (Q > 1) {
#pragma unroll
      for (int c = 1; c < r; ++c) p[c * Q * TP] = cmul(x[c], twR[j * c * step]);
    } else {
#pragma unroll
      for (int c = 1; c < r; ++c) p[c * TP] = x[c];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256, 4) void k_fft_mix(const float2 *__restrict__ in, float2 *__restrict__ out, MixDesc d) {
  extern __shared__ float2 sm[];
  const unsigned R = d.R;
  const int logT = d.logT, TP = (1 << logT) + 1;
  const unsigned T = 1u << logT;
  float2 *buf = sm;
  float2 *twR = sm + R * TP;
  unsigned short *kmap = reinterpret_cast<unsigned short *>(twR + R);  // position after the stages -> frequency
  const int tid = threadIdx.x;
  const unsigned smask = d.dir > 0 ? 0x80000000u : 0u;
  for (unsigned e = tid; e < R; e += 256) {
    twR[e] = tw_q32(phase_q32(e, d.r_hi, d.r_lo));
    unsigned rem = e, k = 0, mult = 1, L = R;
    for (int s = 0; s < d.nst; ++s) {
      const unsigned r = d.rad[s], Q = L / r, dig = rem / Q;
      rem -= dig * Q;
      k += dig * mult;
      mult *= r;
      L = Q;
    }
    kmap[e] = (unsigned short)k;
  }
  const unsigned work = R << logT;
  const float invR = 1.0f / (float)R;
  constexpr int NB = 8;
  if (d.mode == FFT_ROWS && d.acc) {
    // ---- segment spectra that never leave the chip: sum over this workgroup's rows of |X_row[k]|^2
    constexpr int NACC = 16;            // work = R << logT <= 4096 elements = 16 per thread
    float a[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) a[i] = 0.f;
    const unsigned ntiles = (d.rows + T - 1u) >> logT;
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const size_t row0 = (size_t)tile * T;
      __syncthreads();   // twR / kmap (first trip); the previous tile's spectra have been read (later trips)
      for (unsigned w0 = tid; w0 < work; w0 += 256 * NB) {
        float2 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const unsigned w = min(w0 + 256 * u, work - 1u);
          const unsigned t = (unsigned)(((float)w + 0.5f) * invR), j = w - t * R;
          const size_t row = row0 + t;
          if (d.rows_real) v[u] = row < d.rows ? make_float2(reinterpret_cast<const float *>(in)[row * R + j], 0.f) : make_float2(0.f, 0.f);
          else v[u] = row < d.rows ? in[row * R + j] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const unsigned w = w0 + 256 * u;
          if (w < work) {
            const unsigned t = (unsigned)(((float)w + 0.5f) * invR), j = w - t * R;
            buf[j * TP + t] = v[u];
          }
        }
      }
      __syncthreads();
      unsigned L = R;
      for (int s = 0; s < d.nst; ++s) {
        const unsigned r = d.rad[s];
        switch (r) {
          case 2: mix_stage<2>(buf, twR, R, L, logT, TP, tid); break;
          case 3: mix_stage<3>(buf, twR, R, L, logT, TP, tid); break;
          case 4: mix_stage<4>(buf, twR, R, L, logT, TP, tid); break;
          case 5: mix_stage<5>(buf, twR, R, L, logT, TP, tid); break;
          case 8: mix_stage<8>(buf, twR, R, L, logT, TP, tid); break;
          case 9: mix_stage<9>(buf, twR, R, L, logT, TP, tid); break;
          case 10: mix_stage<10>(buf, twR, R, L, logT, TP, tid); break;
          case 25: mix_stage<25>(buf, twR, R, L, logT, TP, tid); break;
          default: mix_stage<16>(buf, twR, R, L, logT, TP, tid); break;
        }
        L /= r;
      }
      // thread-fixed (row of the tile, position) slots: the same every trip, so the sums stay in registers
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        const unsigned w = tid + 256u * i;
        if (w < work) {
          const unsigned t = (unsigned)(((float)w + 0.5f) * invR), p = w - t * R;
          const float2 x = buf[p * TP + t];
          a[i] += x.x * x.x + x.y * x.y;
        }
      }
    }
    __syncthreads();
    float *F = reinterpret_cast<float *>(buf);   // [T][R] row-of-tile sums, then added over the T rows in order
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const unsigned w = tid + 256u * i;
      if (w < work) F[w] = a[i];
    }
    __syncthreads();
    for (unsigned p = tid; p < R; p += 256) {
      float S = F[p];
      for (unsigned t = 1; t < T; ++t) S += F[t * R + p];
      d.acc[(size_t)blockIdx.x * R + kmap[p]] = S;
    }
    return;
  }
  const unsigned bid = blockIdx.x;

  size_t base = 0, tbase = 0, row0 = 0;
  unsigned tile = 0, a = 0, kt = 0, arest = 0, col0 = 0;
  if (d.mode == FFT_STRIDED) {
    tile = bid % d.tiles;
    a = (bid / d.tiles) % d.A;
    const unsigned b = bid / (d.tiles * d.A);
    col0 = tile * T;
    base = (size_t)b * d.N + (size_t)a * R * d.B + col0;
    for (unsigned w0 = tid; w0 < work; w0 += 256 * NB) {
      float2 v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const unsigned w = w0 + 256 * u;
        const unsigned t = w & (T - 1u), j = w >> logT;
        v[u] = (w < work && col0 + t < d.B) ? fft_load(in, d.src_mode, d.src_n, base + (size_t)j * d.B + t, d.src_w8, d.src_aux)
                                             : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const unsigned w = w0 + 256 * u;
        if (w < work) buf[(w >> logT) * TP + (w & (T - 1u))] = conj_if(v[u], smask);
      }
    }
  } else {
    if (d.mode == FFT_LAST) {
      kt = bid % d.k1tiles;
      arest = (bid / d.k1tiles) % d.Aprime;
      tbase = (size_t)(bid / (d.k1tiles * d.Aprime)) * d.N;
    } else {
      row0 = (size_t)bid * T;
    }
    for (unsigned w0 = tid; w0 < work; w0 += 256 * NB) {
      float2 v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const unsigned w = min(w0 + 256 * u, work - 1u);
        const unsigned t = (unsigned)(((float)w + 0.5f) * invR), j = w - t * R;  // w < 2^13: exact
        if (d.mode == FFT_LAST) {
          const unsigned k1 = kt * T + t;
          v[u] = k1 < d.R1 ? in[tbase + ((size_t)k1 * d.Aprime + arest) * R + j] : make_float2(0.f, 0.f);
        } else {
          const size_t row = row0 + t;
          v[u] = row < d.rows ? in[row * R + j] : make_float2(0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const unsigned w = w0 + 256 * u;
        if (w < work) {
          const unsigned t = (unsigned)(((float)w + 0.5f) * invR), j = w - t * R;
          buf[j * TP + t] = conj_if(v[u], smask);
        }
      }
    }
  }
  __syncthreads();

  unsigned L = R;
  for (int s = 0; s < d.nst; ++s) {
    const unsigned r = d.rad[s];
    switch (r) {
      case 2: mix_stage<2>(buf, twR, R, L, logT, TP, tid); break;
      case 3: mix_stage<3>(buf, twR, R, L, logT, TP, tid); break;
      case 4: mix_stage<4>(buf, twR, R, L, logT, TP, tid); break;
      case 5: mix_stage<5>(buf, twR, R, L, logT, TP, tid); break;
      case 8: mix_stage<8>(buf, twR, R, L, logT, TP, tid); break;
      case 9: mix_stage<9>(buf, twR, R, L, logT, TP, tid); break;
      case 10: mix_stage<10>(buf, twR, R, L, logT, TP, tid); break;
      case 25: mix_stage<25>(buf, twR, R, L, logT, TP, tid); break;
      default: mix_stage<16>(buf, twR, R, L, logT, TP, tid); break;
    }
    L /= r;
  }

  if (d.mode == FFT_STRIDED) {
    const unsigned Ka = digit_swap_g(a, d.nprev, d.Rprev, d.Wprev);
    const unsigned t = tid & (T - 1u);  // T <= 256: fixed per thread
    const unsigned col = col0 + t;
    const unsigned nnext = col / d.Bnext;
    if (col < d.B) {
      for (unsigned w0 = tid; w0 < work; w0 += 256 * NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const unsigned w = w0 + 256 * u;
          if (w < work) {
            const unsigned p = w >> logT, k = kmap[p];
            const float2 tw = tw_q32(phase_q32(nnext * (Ka + k * d.Pprev), d.ntw_hi, d.ntw_lo));
            out[base + (size_t)k * d.B + t] = conj_if(cmul(buf[p * TP + t], tw), smask);
          }
        }
      }
    }
  } else if (d.mode == FFT_LAST) {
    // digits k_2..k_{p-1} of arest -> their natural-order weight
    const unsigned Kp = digit_swap_g(arest, d.nprev - 1, d.Rprev + 1, d.Wprev + 1);
    const unsigned t = tid & (T - 1u);
    const unsigned k1 = kt * T + t;
    unsigned long long best = 0ull;
    if (k1 < d.R1) {
      const size_t orel = (size_t)k1 + Kp;
      for (unsigned w = tid; w < work; w += 256) {
        const unsigned p = w >> logT, k = kmap[p];
        const size_t o = orel + (size_t)k * d.Pprev;
        if (o < d.keep) {
          const float2 x = buf[p * TP + t];
          const float2 y = conj_if(make_float2(x.x * d.scale, x.y * d.scale), smask);
          if (d.epi.out) epilogue_store(d.epi, o, y, best); else out[tbase + o] = y;
        }
      }
    }
    if (d.epi.amax_keys) {
      __syncthreads();  // the tile is dead: its first words carry the wavefronts' keys
      epi_argmax_finish(d.epi, best, reinterpret_cast<unsigned long long *>(sm), 4);
    }
  } else {
    for (unsigned w = tid; w < work; w += 256) {
      const unsigned t = (unsigned)(((float)w + 0.5f) * invR), p = w - t * R;
      const size_t row = row0 + t;
      if (row < d.rows) {
        const float2 x = buf[p * TP + t];
        out[row * R + kmap[p]] = conj_if(make_float2(x.x * d.scale, x.y * d.scale), smask);
      }
    }
  }
}

// ---- two register steps per pass (the structure of fft.hip's k_fft_pass with general radices) --------------------
// R = RA * RB, both taken from the register DFT sizes: with n = j0 + RB*m and k = ka + RA*kb
//   X[ka + RA*kb] = sum_j0 W_RB^(j0*kb) * ( W_R^(j0*ka) * sum_m W_RA^(m*ka) x[j0 + RB*m] ).
// Step 1: RA-point DFTs over m in registers (slot = (column t, residue j0)), twiddle, ONE exchange through LDS;
// step 2: RB-point DFTs over j0 in registers (slot = (column t, frequency ka)), stores straight from registers.
// The generic kernel above walks the tile through LDS once per stage and twice more for loading and storing.
// STRIDED loads go global -> registers; LAST stages its rows (contiguous over the DFT index) through LDS first.
template <int RA, int RB>
struct Mix2Geom {
  static constexpr int R = RA * RB;
  static constexpr int tmax() { int t = 1; while (2 * t * R <= 4096 && 2 * t <= 256) t *= 2; return t; }
  static constexpr int TM = tmax();                       // widest tile (columns, a power of two)
  // threads per workgroup: one DFT slot per thread in the larger step when that fits 384 threads (10 x 10 with 32
  // columns: 320 threads, every lane busy in both steps), else 256 threads with several slots each
  static constexpr int BIG = (RA > RB ? RA : RB) * TM;
  static constexpr int NT = BIG <= 384 ? (BIG + 63) / 64 * 64 : 256;
  static constexpr int CA = (RB * TM + NT - 1) / NT;      // step-1 slots per thread
  static constexpr int CB = (RA * TM + NT - 1) / NT;      // step-2 slots per thread
};

template <int RA, int RB, int MODE>
__global__ __launch_bounds__((Mix2Geom<RA, RB>::NT)) void k_fft_mix2(const float2 *__restrict__ in, float2 *__restrict__ out, MixDesc d) {
  using G = Mix2Geom<RA, RB>;
  constexpr int R = G::R, CA = G::CA, CB = G::CB, NT = G::NT;
  extern __shared__ float2 sm[];
  const int logT = d.logT, T = 1 << logT, TP = T + 1;
  const int SA = (RB << logT) + (T < 32 ? T : 0);  // pitch of one ka plane of the exchange buffer
  float2 *buf = sm;                                // staging tile [j][TP] (LAST) and exchange buffer [ka][SA], aliased
  float2 *twR = sm + (R * TP > RA * SA ? R * TP : RA * SA);
  float2 *twK = twR + R;  // STRIDED: tw_sets x R inter-pass twiddles
  const int tid = threadIdx.x;
  const unsigned smask = d.dir > 0 ? 0x80000000u : 0u;
  if (RB > 1)
    for (int e = tid; e < R; e += NT) twR[e] = tw_q32(phase_q32((unsigned)e, d.r_hi, d.r_lo));
  const unsigned bid = blockIdx.x;
  const int n1 = RB << logT, n2 = RA << logT;  // DFT slots of step 1 / step 2

  float2 v[CA * RA > CB * RB ? CA * RA : CB * RB];
  size_t base = 0, tbase = 0;
  unsigned col0 = 0, a = 0, kt = 0, arest = 0, Ka = 0, n0 = 0;
  if (MODE == FFT_STRIDED) {
    const unsigned tile = bid % d.tiles;
    a = (bid / d.tiles) % d.A;
    const unsigned b = bid / (d.tiles * d.A);
    col0 = tile << logT;
    base = (size_t)b * d.N + (size_t)a * R * d.B + col0;
    if (d.src_mode == SRC_C2C) {  // (decided once: the loader switch inside the unrolled loads costs code and SALU)
#pragma unroll
      for (int q = 0; q < CA; ++q) {
        const int s = tid + NT * q;
        const int t = s & (T - 1), j0 = s >> logT;
        const bool ok = s < n1 && col0 + (unsigned)t < d.B;
        const float2 *src = in + base + (size_t)j0 * d.B + t;
#pragma unroll
        for (int m = 0; m < RA; ++m) v[q * RA + m] = ok ? conj_if(src[(size_t)(RB * m) * d.B], smask) : make_float2(0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int q = 0; q < CA; ++q) {
        const int s = tid + NT * q;
        const int t = s & (T - 1), j0 = s >> logT;
        const bool ok = s < n1 && col0 + (unsigned)t < d.B;
#pragma unroll
        for (int m = 0; m < RA; ++m)
          v[q * RA + m] = ok ? conj_if(fft_load(in, d.src_mode, d.src_n, base + (size_t)(j0 + RB * m) * d.B + t, d.src_w8, d.src_aux), smask)
                             : make_float2(0.f, 0.f);
      }
    }
    // inter-pass twiddles W^(n_{i+1} (Ka + k Pprev)).  n_{i+1} = col / Bnext takes tw_sets consecutive values inside a
    // tile (1 when T divides Bnext, 2 when a tile can straddle one boundary ...): the workgroup evaluates those
    // tw_sets * R values once into LDS, under the latency of the loads just issued.  The last strided pass
    // (Bnext = 1: a set per column) factors them as W^(col Ka) -- one evaluation per thread -- times the table
    // W_{R Rnext}^(col k).  Evaluating every output's own twiddle (the fallback) costs ~30 VALU instructions each,
    // +45 % on a pass that is VALU-bound.
    Ka = digit_swap_g(a, d.nprev, d.Rprev, d.Wprev);
    n0 = col0 / d.Bnext;
    for (int e = tid; e < d.tw_sets * R; e += NT) {
      const unsigned set = (unsigned)e / (unsigned)R, k = (unsigned)e - set * (unsigned)R;
      twK[e] = tw_q32(phase_q32((n0 + set) * (Ka + k * d.Pprev), d.ntw_hi, d.ntw_lo));
    }
  } else {
    kt = bid % d.k1tiles;
    arest = (bid / d.k1tiles) % d.Aprime;
    tbase = (size_t)(bid / (d.k1tiles * d.Aprime)) * d.N;
    const int work = R << logT;
    const float invR = 1.0f / (float)R;
    constexpr int NL = (R * G::TM + NT - 1) / NT;  // staged elements per thread
    float2 w[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = min(tid + NT * u, work - 1);
      const int t = (int)(((float)e + 0.5f) * invR), j = e - t * R;  // e < 2^13: exact
      const unsigned k1 = (kt << logT) + (unsigned)t;
      w[u] = k1 < d.R1 ? in[tbase + ((size_t)k1 * d.Aprime + arest) * R + j] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = tid + NT * u;
      if (e < work) {
        const int t = (int)(((float)e + 0.5f) * invR), j = e - t * R;
        buf[j * TP + t] = conj_if(w[u], smask);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int s = min(tid + NT * q, n1 - 1);
      const int t = s & (T - 1), j0 = s >> logT;
#pragma unroll
      for (int m = 0; m < RA; ++m) v[q * RA + m] = buf[(j0 + RB * m) * TP + t];
    }
  }
  __syncthreads();  // twR visible; staging tile fully read before the exchange buffer overwrites it

  // ---- step 1: RA-point DFTs over m, twiddle, exchange ----
#pragma unroll
  for (int q = 0; q < CA; ++q) dft_nat<RA>(v + q * RA);
  if (RB > 1) {
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int s = tid + NT * q;
      const int j0 = s >> logT;
      if (s < n1) {
#pragma unroll
        for (int ka = 0; ka < RA; ++ka) {
          float2 x = v[q * RA + ka];
          if (ka) x = cmul(x, twR[j0 * ka]);
          buf[ka * SA + s] = x;  // (j0, t) is the slot number itself
        }
      }
    }
    __syncthreads();
    // ---- step 2: RB-point DFTs over j0 ----
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      const int s = min(tid + NT * q, n2 - 1);
      const int t = s & (T - 1), ka = s >> logT;
#pragma unroll
      for (int j0 = 0; j0 < RB; ++j0) v[q * RB + j0] = buf[ka * SA + (j0 << logT) + t];
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) dft_nat<RB>(v + q * RB);
  }
  // register i of output slot q now holds X[ka + RA*i] (RB > 1) or X[i] of column slot q (RB == 1)
  constexpr int CO = RB > 1 ? CB : CA, RO = RB > 1 ? RB : RA;
  const int no = RB > 1 ? n2 : n1;
  if (MODE == FFT_STRIDED) {
#pragma unroll
    for (int q = 0; q < CO; ++q) {
      const int s = tid + NT * q;
      const int t = s & (T - 1);
      if (s < no && col0 + (unsigned)t < d.B) {
        float2 *dst = out + base + t;
        const unsigned col = col0 + (unsigned)t;
        if (d.tw_sets) {
          const unsigned set = d.tw_sets == 1 ? 0u : d.tw_sets == 2 ? (col >= (n0 + 1u) * d.Bnext ? 1u : 0u) : col / d.Bnext - n0;
          const float2 *tws = twK + set * R;
#pragma unroll
          for (int i = 0; i < RO; ++i) {
            const int k = RB > 1 ? ((s >> logT) + RA * i) : i;
            dst[(size_t)k * d.B] = conj_if(cmul(v[q * RO + i], tws[k]), smask);
          }
        } else if (d.twg) {
          const float2 w1 = tw_q32(phase_q32(col * Ka, d.ntw_hi, d.ntw_lo));
          const float2 *g = d.twg + col;
#pragma unroll
          for (int i = 0; i < RO; ++i) {
            const int k = RB > 1 ? ((s >> logT) + RA * i) : i;
            dst[(size_t)k * d.B] = conj_if(cmul(v[q * RO + i], cmul(w1, g[(size_t)k * d.B])), smask);
          }
        } else {
          const unsigned nnext = col / d.Bnext;
          const unsigned e0 = nnext * Ka, est = nnext * d.Pprev;  // exponent of output k: nnext*(Ka + k*Pprev) < N
#pragma unroll
          for (int i = 0; i < RO; ++i) {
            const int k = RB > 1 ? ((s >> logT) + RA * i) : i;
            dst[(size_t)k * d.B] = conj_if(cmul(v[q * RO + i], tw_q32(phase_q32(e0 + (unsigned)k * est, d.ntw_hi, d.ntw_lo))), smask);
          }
        }
      }
    }
  } else {
    const unsigned Kp = digit_swap_g(arest, d.nprev - 1, d.Rprev + 1, d.Wprev + 1);
    unsigned long long best = 0ull;
#pragma unroll
    for (int q = 0; q < CO; ++q) {
      const int s = tid + NT * q;
      const int t = s & (T - 1);
      const unsigned k1 = (kt << logT) + (unsigned)t;
      if (s < no && k1 < d.R1) {
        const size_t orel = (size_t)k1 + Kp;
#pragma unroll
        for (int i = 0; i < RO; ++i) {
          const int k = RB > 1 ? ((s >> logT) + RA * i) : i;
          const size_t o = orel + (size_t)k * d.Pprev;
          const float2 x = v[q * RO + i];
          if (o < d.keep) {
            const float2 y = conj_if(make_float2(x.x * d.scale, x.y * d.scale), smask);
            if (d.epi.out) epilogue_store(d.epi, o, y, best); else out[tbase + o] = y;
          }
        }
      }
    }
    if (d.epi.amax_keys) {
      __syncthreads();  // the exchange buffer is dead: its first words carry the wavefronts' keys
      epi_argmax_finish(d.epi, best, reinterpret_cast<unsigned long long *>(sm), NT / 64);
    }
  }
}


// ---- the middle of an autocorrelation: last forward pass + power spectrum + first inverse pass in ONE launch --------
// The forward transform's last pass leaves, per workgroup, T columns c (= the low-order digits of the output index) x all
// R values of the top digit k: Z[c + Bc*k], Bc = N/R.  An inverse transform whose FIRST factor is the same R reads exactly
// that set for its strided DFTs (stride Bc).  In between, the packed power spectrum (fft_load's SRC_POWER) needs
// Z[g] and Z[N - g]: column c pairs with column Bc - c (top digit R-1-k; column 0 with itself, top digit R-k).  So a
// workgroup takes T/2 "direct" columns d and their mirrors Bc - d, transforms their rows forward (two register steps,
// as FFT_LAST), parks Z in LDS, forms Y from the pairs, transforms inverse (two register steps, as FFT_STRIDED) and
// stores with the inverse's inter-pass twiddles.  Z is never written to memory and never read twice: one launch, one
// 16-byte-per-point round trip and one doubled read less per search.
struct MidDesc {
  unsigned R, Bc, ndir;           // radix, columns N/R, direct columns Bc/2 + 1
  int logT, nprev;                // tile width (direct + mirrored halves), earlier forward factors
  unsigned Rprev[MIX_MAX_PASS];   // forward factors R_1 .. R_{p-1} (column digits, least significant first)
  unsigned r_hi, r_lo;            // floor(2^64 / R)
  unsigned Bnext, ntw_hi, ntw_lo; // inverse pass 2: Bc / R'_2 and floor(2^64 / (R * R'_2))
  int tw_sets;                    // inverse inter-pass twiddle sets per half held in LDS (0: evaluated per output)
  double w8;                      // 8 / (2N): phase unit of W_{2N}^g
};

template <int RA, int RB>
__global__ __launch_bounds__((Mix2Geom<RA, RB>::NT)) void k_fft_mid(const float2 *__restrict__ in, float2 *__restrict__ out, MidDesc d) {
  using G = Mix2Geom<RA, RB>;
  constexpr int R = G::R, CA = G::CA, CB = G::CB, NT = G::NT;
  static_assert(RB > 1, "two register steps");
  extern __shared__ float2 sm[];
  const int logT = d.logT, T = 1 << logT, TP = T + 1, Th = T >> 1;
  const int SA = (RB << logT) + (T < 32 ? T : 0);
  float2 *buf = sm;
  float2 *twR = sm + (R * TP > RA * SA ? R * TP : RA * SA);
  float2 *twK = twR + R;
  const int nsets = d.tw_sets > 0 ? d.tw_sets : 0;
  unsigned *colinfo = reinterpret_cast<unsigned *>(twK + 2 * nsets * R);  // [T] column (~0: none), [T] source row
  const int tid = threadIdx.x;
  const unsigned smask = 0x80000000u;  // the second half of the kernel is an inverse transform
  for (int e = tid; e < R; e += NT) twR[e] = tw_q32(phase_q32((unsigned)e, d.r_hi, d.r_lo));
  const unsigned d0 = blockIdx.x * (unsigned)Th;
  if (tid < T) {
    const bool mir = tid >= Th;
    const unsigned dcol = d0 + (unsigned)(tid & (Th - 1));
    unsigned c = 0xFFFFFFFFu;
    if (dcol < d.ndir && !(mir && dcol == 0)) c = mir ? d.Bc - dcol : dcol;  // (the mirror of column 0 is column 0 itself)
    unsigned row = 0;
    if (c != 0xFFFFFFFFu) {
      unsigned cc = c;
      for (int j = 0; j < d.nprev; ++j) { const unsigned q = cc / d.Rprev[j]; row = row * d.Rprev[j] + (cc - q * d.Rprev[j]); cc = q; }
    }
    colinfo[tid] = c;
    colinfo[T + tid] = row;
  }
  __syncthreads();
  const int n1 = RB << logT, n2 = RA << logT;
  float2 v[CA * RA > CB * RB ? CA * RA : CB * RB];
  {  // stage the rows (contiguous over the DFT index), then the step-1 registers
    const int work = R << logT;
    const float invR = 1.0f / (float)R;
    constexpr int NL = (R * G::TM + NT - 1) / NT;
    float2 w[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = min(tid + NT * u, work - 1);
      const int t = (int)(((float)e + 0.5f) * invR), j = e - t * R;
      w[u] = colinfo[t] != 0xFFFFFFFFu ? in[(size_t)colinfo[T + t] * R + j] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = tid + NT * u;
      if (e < work) {
        const int t = (int)(((float)e + 0.5f) * invR), j = e - t * R;
        buf[j * TP + t] = w[u];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int s = min(tid + NT * q, n1 - 1);
      const int t = s & (T - 1), j0 = s >> logT;
#pragma unroll
      for (int m = 0; m < RA; ++m) v[q * RA + m] = buf[(j0 + RB * m) * TP + t];
    }
  }
  __syncthreads();
  // two register steps of one length-R DFT per column; afterwards register i of slot q = X[ka + RA*i] of column t
  auto two_steps = [&]() {
#pragma unroll
    for (int q = 0; q < CA; ++q) dft_nat<RA>(v + q * RA);
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int s = tid + NT * q;
      const int j0 = s >> logT;
      if (s < n1) {
#pragma unroll
        for (int ka = 0; ka < RA; ++ka) {
          float2 x = v[q * RA + ka];
          if (ka) x = cmul(x, twR[j0 * ka]);
          buf[ka * SA + s] = x;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      const int s = min(tid + NT * q, n2 - 1);
      const int t = s & (T - 1), ka = s >> logT;
#pragma unroll
      for (int j0 = 0; j0 < RB; ++j0) v[q * RB + j0] = buf[ka * SA + (j0 << logT) + t];
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) dft_nat<RB>(v + q * RB);
  };
  two_steps();
  __syncthreads();  // exchange buffer fully read
#pragma unroll
  for (int q = 0; q < CB; ++q) {  // park Z[k][t]
    const int s = tid + NT * q;
    if (s < n2) {
      const int t = s & (T - 1), ka = s >> logT;
#pragma unroll
      for (int i = 0; i < RB; ++i) buf[(ka + RA * i) * TP + t] = v[q * RB + i];
    }
  }
  // the inverse's inter-pass twiddles W_{R R'}^(n k): the columns of a half are consecutive, so n = c / Bnext takes
  // tw_sets consecutive values there
  unsigned n0h[2] = {0u, 0u};
  if (nsets) {
    const unsigned dmax = min(d0 + (unsigned)Th - 1u, d.ndir - 1u);
    n0h[0] = d0 / d.Bnext;
    n0h[1] = (dmax ? d.Bc - dmax : 0u) / d.Bnext;
    for (int e = tid; e < 2 * nsets * R; e += NT) {
      const unsigned hs = (unsigned)e / (unsigned)R, k = (unsigned)e - hs * (unsigned)R;
      const unsigned h = hs / (unsigned)nsets, set = hs - h * (unsigned)nsets;
      twK[e] = tw_q32(phase_q32((n0h[h] + set) * k, d.ntw_hi, d.ntw_lo));
    }
  }
  __syncthreads();
  // Y = packed power spectrum from Z[g] and Z[N-g] (fft_load's SRC_POWER arithmetic), conjugated for the inverse
#pragma unroll
  for (int q = 0; q < CA; ++q) {
    const int s = min(tid + NT * q, n1 - 1);
    const int t = s & (T - 1), j0 = s >> logT;
    const unsigned c = colinfo[t];
    const bool ok = tid + NT * q < n1 && c != 0xFFFFFFFFu;
    const int tp = c == 0u ? t : (t ^ Th);
#pragma unroll
    for (int m = 0; m < RA; ++m) {
      const int j = j0 + RB * m;
      const int jm = c == 0u ? (j ? R - j : 0) : R - 1 - j;
      const float2 a = buf[j * TP + t], b = buf[jm * TP + tp];
      const float2 E = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
      const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));
      const float2 O = make_float2(D.y, -D.x);
      const float2 W = tw_frac(ok ? c + d.Bc * (unsigned)j : 0u, d.w8);
      const float2 WO = cmul(W, O);
      const float2 X0 = make_float2(E.x + WO.x, E.y + WO.y), X1 = make_float2(E.x - WO.x, E.y - WO.y);
      const float P0 = X0.x * X0.x + X0.y * X0.y, P1 = X1.x * X1.x + X1.y * X1.y;
      const float sum = P0 + P1, dif = P0 - P1;
      v[q * RA + m] = ok ? conj_if(make_float2(sum + dif * W.y, dif * W.x), smask) : make_float2(0.f, 0.f);
    }
  }
  __syncthreads();  // Z fully read before the exchange buffer overwrites it
  two_steps();
#pragma unroll
  for (int q = 0; q < CB; ++q) {
    const int s = tid + NT * q;
    const int t = s & (T - 1), ka = s >> logT;
    const unsigned c = s < n2 ? colinfo[t] : 0xFFFFFFFFu;
    if (c != 0xFFFFFFFFu) {
      float2 *dst = out + c;
      const unsigned nn = c / d.Bnext;
      if (nsets) {
        const unsigned h = t >= Th ? 1u : 0u;
        const float2 *tws = twK + (h * (unsigned)nsets + (nn - n0h[h])) * R;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int k = ka + RA * i;
          dst[(size_t)k * d.Bc] = conj_if(cmul(v[q * RB + i], tws[k]), smask);
        }
      } else {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int k = ka + RA * i;
          dst[(size_t)k * d.Bc] = conj_if(cmul(v[q * RB + i], tw_q32(phase_q32(nn * (unsigned)k, d.ntw_hi, d.ntw_lo))), smask);
        }
      }
    }
  }
}

typedef void (*mid_fn)(const float2 *, float2 *, MidDesc);
struct MidEntry { unsigned R, RA; int tm, nt; mid_fn fn; size_t lds3; };   // lds3: three-step kernels' dynamic LDS (0: two-step)
#define MID(RA_, RB_) { RA_ * RB_, RA_, Mix2Geom<RA_, RB_>::TM, Mix2Geom<RA_, RB_>::NT, k_fft_mid<RA_, RB_>, 0 }
// last forward factors that have the fused kernel (the planner puts the factor with the most twos last)
static const MidEntry kMid[] = {MID(10, 20), MID(10, 10), MID(16, 16), MID(16, 10), MID(16, 9), MID(16, 8), MID(16, 5), MID(8, 8), MID(8, 5)};
#undef MID
static const MidEntry *mid3_lookup(unsigned R);
static const MidEntry *mid_lookup(unsigned R) {
  for (const MidEntry &e : kMid)
    if (e.R == R) return &e;
  return mid3_lookup(R);
}


// ---- three register steps per pass: factors of 500 .. 2000 ---------------------------------------------------------------
// R = RA * RB * RC.  With n = (n1 RB + n2) RC + n3 and k = k1 + RA k2 + RA RB k3:
//   step 1  slot (n2, n3, t):  A[k1] = sum_n1 W_RA^(n1 k1) x[n1, n2, n3],       times W_{RA RB}^(n2 k1)
//   step 2  slot (k1, n3, t):  B[k2] = sum_n2 W_RB^(n2 k2) A[k1, n2, n3],       times W_R^(n3 (k1 + RA k2))
//   step 3  slot (k1, k2, t):  X[k1 + RA k2 + RA RB k3] = sum_n3 W_RC^(n3 k3) B[k1, k2, n3]
// One slot per thread in every step (the workgroup has as many threads as the largest step has slots), two exchanges
// through ONE LDS tile that every step overwrites in place (a slot reads and writes the same RA / RB positions), k1 planes
// padded by T elements so that the step-3 reads of a half-wavefront fall on distinct banks.  A 2e6-point transform is TWO
// such passes (1000 x 2000) instead of three of the two-step kernel's: the passes are latency-bound at that size (every
// workgroup resident at once), so their number is what counts.  STRIDED loads go global -> registers in T-element runs;
// LAST reads its rows contiguously (slot order with the column slowest).  Inter-pass twiddles W^(n (Ka + k P)) are linear
// in k3: one evaluation for the slot's first output, one for the step, then a product per output.
template <int RA, int RB, int RC, int LOGT>
struct Mix3Geom {
  static constexpr int R = RA * RB * RC, T = 1 << LOGT;
  static constexpr int S1 = RB * RC * T, S2 = RA * RC * T, S3 = RA * RB * T;
  static constexpr int SMAX = S1 > S2 ? (S1 > S3 ? S1 : S3) : (S2 > S3 ? S2 : S3);
  static constexpr int NT = (SMAX + 63) / 64 * 64;
  static_assert(NT <= 1024, "three-step kernel: a step has more slots than a workgroup has threads");
  static constexpr int PLANE = RB * RC * T + T;
  static constexpr int VMAX = RA > RB ? (RA > RC ? RA : RC) : (RB > RC ? RB : RC);
  static constexpr size_t LDS = ((size_t)RA * PLANE + R) * sizeof(float2);
  // Exchange-tile position of (second digit, third digit n3, column t) within a k1 plane: (n2 RC + n3) T + (t ^ swz(n3)).
  // The passes that read their rows contiguously (FFT_LAST, the fused middle's forward half, getWelch's accumulator) write
  // step 1 with the lanes of a wavefront running along n3: unswizzled that is a stride of T float2 -- 16 (8) lanes of a
  // 16-lane group on the same bank pair, 7.7 (3.8) LDS passes per ds_write_b64 for T = 8 (4), SQ_LDS_BANK_CONFLICT = 3400
  // cycles per workgroup.  The column is XORed with n3 / 2: 2.0 (1.9) passes there, and the other three access patterns
  // (step-1 writes with the column fastest, steps 2 and 3: lanes along t, then n3 or k1) stay conflict-free -- n3 is
  // constant per thread in step 2 and a compile-time constant in step 3, so the swizzle costs a handful of XORs per pass.
  static __device__ __forceinline__ int swz(int n3) { return (n3 >> 1) & (T - 1); }
};

// MODE: FFT_STRIDED / FFT_LAST = passes of a multi-pass transform; the three whole-row modes below = rows of R points, T per
// tile, walked by persistent workgroups (separate instantiations: as run-time branches of one kernel the row store's conjugations
// and the writers' extra live values cost the accumulator 15-60 %)
enum { M3_ACC = 10, M3_WF = 11, M3_ROWS = 12 };   // getWelch's accumulator | getWaterfall's writer | batched row transforms
template <int RA, int RB, int RC, int LOGT, int MODE>
__global__ __launch_bounds__((Mix3Geom<RA, RB, RC, LOGT>::NT)) void k_fft_mix3(const float2 *__restrict__ in, float2 *__restrict__ out, MixDesc d) {
  using G = Mix3Geom<RA, RB, RC, LOGT>;
  constexpr int R = G::R, T = G::T, NT = G::NT, PLANE = G::PLANE, R23 = RB * RC;
  extern __shared__ float2 sm[];
  float2 *buf = sm;                       // [RA][PLANE]: (k1 | n1, second digit, third digit, column)
  float2 *twR = sm + RA * PLANE;          // W_R^e, e < R
  const int tid = threadIdx.x;
  const unsigned smask = d.dir > 0 ? 0x80000000u : 0u;
  for (int e = tid; e < R; e += NT) twR[e] = tw_q32(phase_q32((unsigned)e, d.r_hi, d.r_lo));
  const unsigned bid = blockIdx.x;
  float2 v[G::VMAX];
  if (MODE >= M3_ACC) {
    // ---- getWelch's accumulator (see MixDesc::acc): rows = segments, T of them per tile, nothing stored per transform.
    // A thread's step-3 slot (kk, t3) is the same for every tile, so abs2 of its RC outputs accumulates in registers.
    float acc[RC];
#pragma unroll
    for (int k3 = 0; k3 < RC; ++k3) acc[k3] = 0.f;
    const unsigned ntiles = (d.rows + (unsigned)T - 1u) >> LOGT;
    const int t1 = tid / R23, r23 = tid - t1 * R23;
    const int t3 = tid & (T - 1), kk = tid >> LOGT;
    for (unsigned tile = bid; tile < ntiles; tile += gridDim.x) {
      const unsigned row1 = (tile << LOGT) + (unsigned)t1;
      const bool ok1 = tid < G::S1 && row1 < d.rows;
      if (d.rows_real) {
        const float *src = reinterpret_cast<const float *>(in) + (size_t)row1 * R + r23;
#pragma unroll
        for (int n1 = 0; n1 < RA; ++n1) v[n1] = ok1 ? make_float2(src[n1 * R23], 0.f) : make_float2(0.f, 0.f);
      } else {
        const float2 *src = in + (size_t)row1 * R + r23;
#pragma unroll
        for (int n1 = 0; n1 < RA; ++n1) v[n1] = ok1 ? (MODE == M3_ROWS ? conj_if(src[n1 * R23], smask) : src[n1 * R23]) : make_float2(0.f, 0.f);
      }
      __syncthreads();  // twR (first trip); the previous tile's step-3 reads (later trips)
      if (tid < G::S1) {
        dft_nat<RA>(v);
        const int n2 = r23 / RC;
#pragma unroll
        for (int k1 = 0; k1 < RA; ++k1) {
          float2 x = v[k1];
          if (k1) x = cmul(x, twR[(n2 * k1) * RC]);
          buf[k1 * PLANE + r23 * T + (t1 ^ G::swz(r23 - n2 * RC))] = x;
        }
      }
      __syncthreads();
      if (tid < G::S2) {
        const int t = tid & (T - 1), q = tid >> LOGT, k1 = q / RC, n3 = q - k1 * RC;
        float2 *p = buf + k1 * PLANE + n3 * T + (t ^ G::swz(n3));
#pragma unroll
        for (int n2 = 0; n2 < RB; ++n2) v[n2] = p[n2 * RC * T];
        dft_nat<RB>(v);
#pragma unroll
        for (int k2 = 0; k2 < RB; ++k2) {
          float2 x = v[k2];
          if (k2 || k1) x = cmul(x, twR[n3 * (k1 + RA * k2)]);
          p[k2 * RC * T] = x;
        }
      }
      __syncthreads();
      if (tid < G::S3 && (tile << LOGT) + (unsigned)t3 < d.rows) {
        const int k2 = kk / RA, k1 = kk - k2 * RA;
        const float2 *p = buf + k1 * PLANE + k2 * RC * T;
#pragma unroll
        for (int n3 = 0; n3 < RC; ++n3) v[n3] = p[n3 * T + (t3 ^ G::swz(n3))];
        dft_nat<RC>(v);
        if (MODE == M3_ROWS) {
          float2 *row = d.rows_out + (size_t)((tile << LOGT) + (unsigned)t3) * R;
#pragma unroll
          for (int k3 = 0; k3 < RC; ++k3)
            row[kk + RA * RB * k3] = conj_if(make_float2(v[k3].x * d.scale, v[k3].y * d.scale), smask);
        } else if (MODE == M3_WF) {
          double *row = d.wf + (size_t)((tile << LOGT) + (unsigned)t3) * R;
#pragma unroll
          for (int k3 = 0; k3 < RC; ++k3) {
            int j = kk + RA * RB * k3 + R / 2;   // fftshift: frequency k lands at (k + floor(R / 2)) mod R
            if (j >= R) j -= R;
            __builtin_nontemporal_store((double)(v[k3].x * v[k3].x + v[k3].y * v[k3].y), &row[j]);   // written once
          }
        } else {
#pragma unroll
          for (int k3 = 0; k3 < RC; ++k3) acc[k3] += v[k3].x * v[k3].x + v[k3].y * v[k3].y;
        }
      }
    }
    if (MODE != M3_ACC) return;
    // the T segments of a tile sit in T adjacent lanes: added by a fixed xor tree, lane t3 = 0 stores
#pragma unroll
    for (int k3 = 0; k3 < RC; ++k3) {
      float sres = acc[k3];
#pragma unroll
      for (int off = 1; off < T; off <<= 1) sres += __shfl_xor(sres, off, 64);
      if (tid < G::S3 && t3 == 0) d.acc[(size_t)bid * R + (unsigned)(kk + RA * RB * k3)] = sres;
    }
    return;
  }
  size_t base = 0, tbase = 0;
  unsigned col0 = 0, a = 0, kt = 0, arest = 0;
  // ---- step 1 inputs
  int t1, r23;
  bool ok1;
  if (MODE == FFT_STRIDED) {
    const unsigned tile = bid % d.tiles;
    a = (bid / d.tiles) % d.A;
    const unsigned b = bid / (d.tiles * d.A);
    col0 = tile << LOGT;
    base = (size_t)b * d.N + (size_t)a * R * d.B + col0;
    t1 = tid & (T - 1); r23 = tid >> LOGT;
    ok1 = tid < G::S1 && col0 + (unsigned)t1 < d.B;
    const size_t g0 = base + (size_t)r23 * d.B + t1;
    if (d.src_mode == SRC_C2C) {
#pragma unroll
      for (int n1 = 0; n1 < RA; ++n1) v[n1] = ok1 ? conj_if(in[g0 + (size_t)(n1 * R23) * d.B], smask) : make_float2(0.f, 0.f);
    } else {
#pragma unroll
      for (int n1 = 0; n1 < RA; ++n1)
        v[n1] = ok1 ? conj_if(fft_load(in, d.src_mode, d.src_n, g0 + (size_t)(n1 * R23) * d.B, d.src_w8, d.src_aux), smask) : make_float2(0.f, 0.f);
    }
  } else {
    kt = bid % d.k1tiles;
    arest = (bid / d.k1tiles) % d.Aprime;
    tbase = (size_t)(bid / (d.k1tiles * d.Aprime)) * d.N;
    t1 = tid / R23; r23 = tid - t1 * R23;  // the column slowest: a wavefront reads one row contiguously
    const unsigned k1g = (kt << LOGT) + (unsigned)t1;
    ok1 = tid < G::S1 && k1g < d.R1;
    const float2 *src = in + tbase + ((size_t)k1g * d.Aprime + arest) * R + r23;
#pragma unroll
    for (int n1 = 0; n1 < RA; ++n1) v[n1] = ok1 ? conj_if(src[n1 * R23], smask) : make_float2(0.f, 0.f);
  }
  __syncthreads();  // twR
  if (tid < G::S1) {
    dft_nat<RA>(v);
    const int n2 = r23 / RC;
#pragma unroll
    for (int k1 = 0; k1 < RA; ++k1) {
      float2 x = v[k1];
      if (k1) x = cmul(x, twR[(n2 * k1) * RC]);  // W_{RA RB}^(n2 k1)
      buf[k1 * PLANE + r23 * T + (t1 ^ G::swz(r23 - n2 * RC))] = x;
    }
  }
  __syncthreads();
  // ---- step 2: slot (k1, n3, t), in place
  if (tid < G::S2) {
    const int t = tid & (T - 1), q = tid >> LOGT, k1 = q / RC, n3 = q - k1 * RC;
    float2 *p = buf + k1 * PLANE + n3 * T + (t ^ G::swz(n3));
#pragma unroll
    for (int n2 = 0; n2 < RB; ++n2) v[n2] = p[n2 * RC * T];
    dft_nat<RB>(v);
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) {
      float2 x = v[k2];
      const int e = n3 * (k1 + RA * k2);
      if (k2 || k1) x = cmul(x, twR[e]);  // (e = 0 gives 1 anyway)
      p[k2 * RC * T] = x;
    }
  }
  __syncthreads();
  // ---- step 3: slot (kk = k1 + RA k2, t); outputs straight from registers
  const int t3 = tid & (T - 1), kk = tid >> LOGT;
  const bool ok3 = tid < G::S3;
  if (ok3) {
    const int k2 = kk / RA, k1 = kk - k2 * RA;
    const float2 *p = buf + k1 * PLANE + k2 * RC * T;
#pragma unroll
    for (int n3 = 0; n3 < RC; ++n3) v[n3] = p[n3 * T + (t3 ^ G::swz(n3))];
    dft_nat<RC>(v);
  }
  if (MODE == FFT_STRIDED) {
    const unsigned col = col0 + (unsigned)t3;
    if (ok3 && col < d.B) {
      const unsigned Ka = digit_swap_g(a, d.nprev, d.Rprev, d.Wprev);
      const unsigned nn = col / d.Bnext;
      float2 w = tw_q32(phase_q32(nn * (Ka + (unsigned)kk * d.Pprev), d.ntw_hi, d.ntw_lo));
      const float2 ws = tw_q32(phase_q32(nn * ((unsigned)(RA * RB) * d.Pprev), d.ntw_hi, d.ntw_lo));
      float2 *dst = out + base + (size_t)kk * d.B + t3;
#pragma unroll
      for (int k3 = 0; k3 < RC; ++k3) {
        dst[(size_t)(RA * RB * k3) * d.B] = conj_if(cmul(v[k3], w), smask);
        w = cmul(w, ws);
      }
    }
  } else {
    const unsigned Kp = digit_swap_g(arest, d.nprev - 1, d.Rprev + 1, d.Wprev + 1);
    const unsigned k1g = (kt << LOGT) + (unsigned)t3;
    unsigned long long best = 0ull;
    if (ok3 && k1g < d.R1) {
      const size_t orel = (size_t)k1g + Kp;
#pragma unroll
      for (int k3 = 0; k3 < RC; ++k3) {
        const size_t o = orel + (size_t)(kk + RA * RB * k3) * d.Pprev;
        if (o < d.keep) {
          const float2 y = conj_if(make_float2(v[k3].x * d.scale, v[k3].y * d.scale), smask);
          if (d.epi.out) epilogue_store(d.epi, o, y, best); else out[tbase + o] = y;
        }
      }
    }
    if (d.epi.amax_keys) {
      __syncthreads();
      epi_argmax_finish(d.epi, best, reinterpret_cast<unsigned long long *>(sm), NT / 64);
    }
  }
}

typedef void (*mix3_fn)(const float2 *, float2 *, MixDesc);
struct Mix3Entry { unsigned R; int logT, nt; size_t lds; mix3_fn strided, last, acc, wf, rows; };
#define MIX3(RA_, RB_, RC_, LT_)                                                                                            \
  { RA_ * RB_ * RC_, LT_, Mix3Geom<RA_, RB_, RC_, LT_>::NT, Mix3Geom<RA_, RB_, RC_, LT_>::LDS,                               \
    k_fft_mix3<RA_, RB_, RC_, LT_, FFT_STRIDED>, k_fft_mix3<RA_, RB_, RC_, LT_, FFT_LAST>, k_fft_mix3<RA_, RB_, RC_, LT_, M3_ACC>, \
    k_fft_mix3<RA_, RB_, RC_, LT_, M3_WF>, k_fft_mix3<RA_, RB_, RC_, LT_, M3_ROWS> }
// 8000-point tiles (64 KiB of LDS + the twiddle table): 1000 x 8 columns, 2000 x 4; 500 x 8 (4000 points)
// (tiles half as wide -- 32-byte runs -- measured 25.8 / 22.9 us per pass against 18.5 / 19.9 at 2e6 points)
static const Mix3Entry kMix3[] = {MIX3(10, 10, 10, 3), MIX3(20, 10, 10, 2), MIX3(5, 10, 10, 3)};
#undef MIX3
static const Mix3Entry *mix3_lookup(unsigned R) {
  for (const Mix3Entry &e : kMix3)
    if (e.R == R) return &e;
  return nullptr;
}
// getWelch's accumulator only (fft_rows_welch: the FFT_LAST kernel with MixDesc::acc; the pass planner does not see these):
// the power-of-two segment lengths next to the 1024 that k_seg1024 serves -- 2048 = 16 x 16 x 8 (two segments per tile),
// 4096 = 16 x 16 x 16 (one), 512 = 8 x 8 x 8 (eight), 256 = 8 x 8 x 4 (eight), 128 -- and the round lengths that split into
// three of the register DFT sizes (4000, 3200, 2500, 1600, 1280, 1200, 768; 960 = 20 x 16 x 3 measured slower than the
// generic kernel: 108 against 88 us); everything else: the generic LDS-stage kernel
#define WELCH3(RA_, RB_, RC_, LT_)                                                                                          \
  { RA_ * RB_ * RC_, LT_, Mix3Geom<RA_, RB_, RC_, LT_>::NT, Mix3Geom<RA_, RB_, RC_, LT_>::LDS, nullptr, nullptr,             \
    k_fft_mix3<RA_, RB_, RC_, LT_, M3_ACC>, k_fft_mix3<RA_, RB_, RC_, LT_, M3_WF>, k_fft_mix3<RA_, RB_, RC_, LT_, M3_ROWS> }
static const Mix3Entry kWelch3[] = {
    WELCH3(16, 16, 8, 1), WELCH3(16, 16, 16, 0), WELCH3(8, 8, 8, 3), WELCH3(8, 8, 4, 3), WELCH3(8, 4, 4, 4),   // 2048 4096 512 256 128
    WELCH3(10, 10, 10, 2), WELCH3(5, 10, 10, 2),                                                               // 1000 500 on half the pass kernels' tiles: row / waterfall modes
    // (1000: rows 50.5 -> 40.8 us, waterfall 57 -> 47 us, but the accumulator 44 -> 50 us; 2000 on 4000-point tiles lost everywhere)
    WELCH3(20, 20, 10, 0), WELCH3(25, 10, 10, 0), WELCH3(20, 16, 10, 0), WELCH3(20, 10, 8, 1),                  // 4000 2500 3200 1600
    WELCH3(16, 16, 5, 1), WELCH3(20, 20, 3, 1), WELCH3(16, 16, 3, 2),                                           // 1280 1200 768
};
#undef WELCH3
static const Mix3Entry *welch3_lookup(unsigned R, bool accumulator) {
  if (accumulator)
    if (const Mix3Entry *e = mix3_lookup(R)) return e;   // getWelch: the pass kernels' 8000-point tiles measured better
  for (const Mix3Entry &e : kWelch3)
    if (e.R == R) return &e;
  return mix3_lookup(R);
}
// kernels above 64 KiB of dynamic LDS have to be opted in once
static int mix3_prepare(tsdr_ctx *ctx, const Mix3Entry *e) {
  for (mix3_fn f : {e->strided, e->last, e->acc, e->wf, e->rows}) {
    if (!f) continue;
    int rc = lds_opt_in(ctx, (const void *)f, e->lds);
    if (rc) return rc;
  }
  return TSDR_OK;
}



// ---- the fused middle (k_fft_mid) with three register steps: last forward factor = first inverse factor = 1000 or 2000 ----
// Same pairing of direct and mirrored columns, same power-spectrum arithmetic; the two length-R DFTs are k_fft_mix3's
// three steps.  With it a 2e6-point autocorrelation is THREE launches (1000 | 2000 + power + 2000 | 1000) and Z is never
// written.
template <int RA, int RB, int RC, int LOGT>
__global__ __launch_bounds__((Mix3Geom<RA, RB, RC, LOGT>::NT)) void k_fft_mid3(const float2 *__restrict__ in, float2 *__restrict__ out, MidDesc d) {
  using G = Mix3Geom<RA, RB, RC, LOGT>;
  constexpr int R = G::R, T = G::T, NT = G::NT, PLANE = G::PLANE, R23 = RB * RC, Th = T / 2;
  extern __shared__ float2 sm[];
  float2 *buf = sm;                       // [RA][PLANE] exchange tile; also Z[k][t] between the two transforms
  float2 *twR = sm + RA * PLANE;
  unsigned *colinfo = reinterpret_cast<unsigned *>(twR + R);  // [T] column (~0: none), [T] source row
  const int tid = threadIdx.x;
  const unsigned smask = 0x80000000u;
  for (int e = tid; e < R; e += NT) twR[e] = tw_q32(phase_q32((unsigned)e, d.r_hi, d.r_lo));
  const unsigned d0 = blockIdx.x * (unsigned)Th;
  if (tid < T) {
    const bool mir = tid >= Th;
    const unsigned dcol = d0 + (unsigned)(tid & (Th - 1));
    unsigned c = 0xFFFFFFFFu;
    if (dcol < d.ndir && !(mir && dcol == 0)) c = mir ? d.Bc - dcol : dcol;
    unsigned row = 0;
    if (c != 0xFFFFFFFFu) {
      unsigned cc = c;
      for (int j = 0; j < d.nprev; ++j) { const unsigned q = cc / d.Rprev[j]; row = row * d.Rprev[j] + (cc - q * d.Rprev[j]); cc = q; }
    }
    colinfo[tid] = c;
    colinfo[T + tid] = row;
  }
  __syncthreads();
  float2 v[G::VMAX];
  // three register steps of one length-R DFT per column; on entry v = the slot's RA inputs (slot = (r23, t1)), on exit
  // v[k3] = X[kk + RA RB k3] of column t3 for the slot (kk, t3) = (tid >> LOGT, tid & (T-1))
  auto three_steps = [&](int t1, int r23) {
    if (tid < G::S1) {
      dft_nat<RA>(v);
      const int n2 = r23 / RC;
#pragma unroll
      for (int k1 = 0; k1 < RA; ++k1) {
        float2 x = v[k1];
        if (k1) x = cmul(x, twR[(n2 * k1) * RC]);
        buf[k1 * PLANE + r23 * T + (t1 ^ G::swz(r23 - n2 * RC))] = x;
      }
    }
    __syncthreads();
    if (tid < G::S2) {
      const int t = tid & (T - 1), q = tid >> LOGT, k1 = q / RC, n3 = q - k1 * RC;
      float2 *p = buf + k1 * PLANE + n3 * T + (t ^ G::swz(n3));
#pragma unroll
      for (int n2 = 0; n2 < RB; ++n2) v[n2] = p[n2 * RC * T];
      dft_nat<RB>(v);
#pragma unroll
      for (int k2 = 0; k2 < RB; ++k2) {
        float2 x = v[k2];
        if (k2 || k1) x = cmul(x, twR[n3 * (k1 + RA * k2)]);
        p[k2 * RC * T] = x;
      }
    }
    __syncthreads();
    if (tid < G::S3) {
      const int t = tid & (T - 1), kk = tid >> LOGT, k2 = kk / RA, k1 = kk - k2 * RA;
      const float2 *p = buf + k1 * PLANE + k2 * RC * T;
#pragma unroll
      for (int n3 = 0; n3 < RC; ++n3) v[n3] = p[n3 * T + (t ^ G::swz(n3))];
      dft_nat<RC>(v);
    }
  };
  {  // forward: rows are contiguous over the DFT index (slot order with the column slowest)
    const int t1 = tid / R23, r23 = tid - t1 * R23;
    const bool ok = tid < G::S1 && colinfo[t1 < T ? t1 : 0] != 0xFFFFFFFFu;
    const float2 *src = in + (size_t)colinfo[T + (t1 < T ? t1 : 0)] * R + r23;
#pragma unroll
    for (int n1 = 0; n1 < RA; ++n1) v[n1] = ok ? src[n1 * R23] : make_float2(0.f, 0.f);
    three_steps(t1, r23);
  }
  const int t3 = tid & (T - 1), kk = tid >> LOGT;
  __syncthreads();  // every step-3 read of the tile is done
  if (tid < G::S3) {
#pragma unroll
    for (int k3 = 0; k3 < RC; ++k3) buf[(kk + RA * RB * k3) * T + t3] = v[k3];  // park Z[k][t]
  }
  __syncthreads();
  {  // Y from Z[g] and Z[N-g], conjugated for the inverse; slot (r23, t) with the column fastest
    const int t = tid & (T - 1), r23 = tid >> LOGT;
    const unsigned c = tid < G::S1 ? colinfo[t] : 0xFFFFFFFFu;
    const bool ok = c != 0xFFFFFFFFu;
    const int tp = c == 0u ? t : (t ^ Th);
#pragma unroll
    for (int n1 = 0; n1 < RA; ++n1) {
      const int j = ok ? n1 * R23 + r23 : 0;
      const int jm = c == 0u ? (j ? R - j : 0) : R - 1 - j;
      const float2 a = buf[j * T + t], b = buf[jm * T + tp];
      const float2 E = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
      const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));
      const float2 O = make_float2(D.y, -D.x);
      const float2 W = tw_frac(ok ? c + d.Bc * (unsigned)j : 0u, d.w8);
      const float2 WO = cmul(W, O);
      const float2 X0 = make_float2(E.x + WO.x, E.y + WO.y), X1 = make_float2(E.x - WO.x, E.y - WO.y);
      const float P0 = X0.x * X0.x + X0.y * X0.y, P1 = X1.x * X1.x + X1.y * X1.y;
      const float sum = P0 + P1, dif = P0 - P1;
      v[n1] = ok ? conj_if(make_float2(sum + dif * W.y, dif * W.x), smask) : make_float2(0.f, 0.f);
    }
    __syncthreads();  // Z fully read before the exchange tile overwrites it
    three_steps(t, r23);
  }
  if (tid < G::S3) {
    const unsigned c = colinfo[t3];
    if (c != 0xFFFFFFFFu) {
      const unsigned nn = c / d.Bnext;
      float2 w = tw_q32(phase_q32(nn * (unsigned)kk, d.ntw_hi, d.ntw_lo));
      const float2 ws = tw_q32(phase_q32(nn * (unsigned)(RA * RB), d.ntw_hi, d.ntw_lo));
      float2 *dst = out + c + (size_t)kk * d.Bc;
#pragma unroll
      for (int k3 = 0; k3 < RC; ++k3) {
        dst[(size_t)(RA * RB * k3) * d.Bc] = conj_if(cmul(v[k3], w), smask);
        w = cmul(w, ws);
      }
    }
  }
}
#define MID3(RA_, RB_, RC_, LT_) { RA_ * RB_ * RC_, 0, 1 << LT_, Mix3Geom<RA_, RB_, RC_, LT_>::NT, k_fft_mid3<RA_, RB_, RC_, LT_>, \
                                   Mix3Geom<RA_, RB_, RC_, LT_>::LDS + 2 * (size_t)(1 << LT_) * 4 }
static const MidEntry kMid3[] = {MID3(20, 10, 10, 2), MID3(10, 10, 10, 3)};
#undef MID3
static const MidEntry *mid3_lookup(unsigned R) {
  for (const MidEntry &e : kMid3)
    if (e.R == R) return &e;
  return nullptr;
}
typedef void (*mix2_fn)(const float2 *, float2 *, MixDesc);
struct Mix2Entry { unsigned R, RA; int tm, nt; mix2_fn strided, last; };
#define MIX2(RA_, RB_)                                                                                     \
  { RA_ * RB_, RA_, Mix2Geom<RA_, RB_>::TM, Mix2Geom<RA_, RB_>::NT, k_fft_mix2<RA_, RB_, FFT_STRIDED>, k_fft_mix2<RA_, RB_, FFT_LAST> }
// the factor sizes that have a two-step kernel (RB > 1 everywhere); for a size listed twice the first entry wins.
// The balanced splits come first: with RA ~ RB one thread owns one DFT of each step and nobody idles, and 10- or
// 16-point register DFTs keep the kernel near 64 VGPRs; 25 x 5 leaves 3 of 8 lanes without a step-1 DFT at 160 VGPRs.
static const Mix2Entry kMix2[] = {
    MIX2(10, 10), MIX2(10, 20), MIX2(10, 5),
    MIX2(16, 16), MIX2(16, 10), MIX2(16, 9), MIX2(16, 8), MIX2(16, 5), MIX2(25, 10), MIX2(25, 9), MIX2(25, 8),
    MIX2(25, 5),  MIX2(25, 4),  MIX2(25, 3), MIX2(25, 2), MIX2(10, 9), MIX2(9, 9),   MIX2(9, 8),  MIX2(9, 5),
    MIX2(8, 8),   MIX2(8, 5),   MIX2(5, 5),
};
#undef MIX2
static const Mix2Entry *mix2_lookup(unsigned R) {
  for (const Mix2Entry &e : kMix2)
    if (e.R == R) return &e;
  return nullptr;
}
static size_t mix2_lds(unsigned R, unsigned RA, int logT, int tw_sets = 1) {
  const size_t T = (size_t)1 << logT, RB = R / RA;
  const size_t SA = (RB << logT) + (T < 32 ? T : 0);
  return (std::max((size_t)R * (T + 1), (size_t)RA * SA) + (size_t)(1 + std::max(tw_sets, 1)) * R) * sizeof(float2);
}

// ---- planning --------------------------------------------------------------------------------
struct MixPlan {
  int p = 0;
  unsigned R[MIX_MAX_PASS];
  std::vector<unsigned char> rad[MIX_MAX_PASS];
};

static void stage_radices(unsigned e2, unsigned e3, unsigned e5, std::vector<unsigned char> &out) {
  out.clear();
  while (e5 >= 2) { out.push_back(25); e5 -= 2; }
  if (e5) {
    if (e2) { out.push_back(10); --e2; } else out.push_back(5);
  }
  while (e3 >= 2) { out.push_back(9); e3 -= 2; }
  if (e3) out.push_back(3);
  while (e2 >= 4) { out.push_back(16); e2 -= 4; }
  if (e2 == 3) out.push_back(8);
  if (e2 == 2) out.push_back(4);
  if (e2 == 1) out.push_back(2);
}

// What one pass through a factor costs relative to the best kernels (every pass moves the same 16 bytes per point;
// measured on MI355X at 2e6..2e7 points): balanced two-step kernels 1, the 25 x n ones ~1.6, the generic LDS-stage
// kernel ~2.2.
static double factor_cost(unsigned R) {
  if (mix3_lookup(R)) return 1.35;  // one pass through a three-step kernel (measured against the balanced two-step ones)
  const Mix2Entry *e = mix2_lookup(R);
  if (!e) return 2.2;
  return e->RA == 25 ? 1.6 : 1.0;
}

struct PlanSearch {
  unsigned ex[3];
  bool allow_big = true;  // factors of 500 .. 2000 (three-step kernels)
  int best_p = 0;
  double best = 1e30;
  unsigned cur[MIX_MAX_PASS][3], out[MIX_MAX_PASS][3];
  static unsigned val(const unsigned *e) {
    unsigned v = 1;
    for (unsigned i = 0; i < e[0]; ++i) v *= 2;
    for (unsigned i = 0; i < e[1]; ++i) v *= 3;
    for (unsigned i = 0; i < e[2]; ++i) v *= 5;
    return v;
  }
  // factors in non-increasing order (the order is fixed afterwards), depth-first with a cost bound
  void go(int depth, unsigned cap, double cost) {
    if (!(ex[0] | ex[1] | ex[2])) {
      if (depth == 1 && val(cur[0]) > 256) return;  // the three-step kernels are passes of a multi-pass transform only
      // ties: prefer a factor carrying 2^4 (it goes last: every stride a multiple of 16 elements)
      unsigned m2 = 0;
      for (int i = 0; i < depth; ++i) m2 = std::max(m2, std::min(cur[i][0], 4u));
      const double c = cost - 0.01 * m2;
      if (c < best - 1e-9) {
        best = c;
        best_p = depth;
        for (int i = 0; i < depth; ++i) for (int j = 0; j < 3; ++j) out[i][j] = cur[i][j];
      }
      return;
    }
    if (depth == MIX_MAX_PASS) return;
    {
      double rem = 1.0;
      for (unsigned i = 0; i < ex[0]; ++i) rem *= 2;
      for (unsigned i = 0; i < ex[1]; ++i) rem *= 3;
      for (unsigned i = 0; i < ex[2]; ++i) rem *= 5;
      const double need = std::max(1.0, std::ceil(std::log(rem) / std::log((double)cap) - 1e-9));  // passes still to come
      if (depth + (int)need > MIX_MAX_PASS || cost + need >= best + 0.05) return;
    }
    for (unsigned a = 0; a <= ex[0]; ++a)
      for (unsigned b = 0; b <= ex[1]; ++b)
        for (unsigned c = 0; c <= ex[2]; ++c) {
          const unsigned e[3] = {a, b, c};
          if (a > 8 || b > 5 || c > 3) continue;
          const unsigned R = val(e);
          if (R < 2 || R > cap) continue;
          if (R > 256 && !(allow_big && mix3_lookup(R))) continue;
          std::vector<unsigned char> rad;
          stage_radices(a, b, c, rad);
          if (rad.size() > MIX_MAX_STAGE) continue;
          for (int j = 0; j < 3; ++j) { cur[depth][j] = e[j]; ex[j] -= e[j]; }
          go(depth + 1, R, cost + factor_cost(R));
          for (int j = 0; j < 3; ++j) ex[j] += e[j];
        }
  }
};

// true when N = 2^a 3^b 5^c (N >= 2) and a pass split with every factor <= 256 exists.  The split minimises the
// summed pass costs above; the factor with the most twos goes last, the others largest first.
static bool fft_mixed_plan_search(size_t N, MixPlan *plan, bool allow_big);
bool fft_mixed_plan(size_t N, MixPlan *plan, bool allow_big = true) {  // the search runs once per length
  static std::mutex mu;
  static std::unordered_map<size_t, std::pair<bool, MixPlan>> cache[2];
  std::lock_guard<std::mutex> g(mu);
  auto &c = cache[allow_big ? 1 : 0];
  auto it = c.find(N);
  if (it == c.end()) {
    if (c.size() > 4096) c.clear();
    MixPlan pl;
    const bool ok = fft_mixed_plan_search(N, &pl, allow_big);
    it = c.emplace(N, std::make_pair(ok, pl)).first;
  }
  if (it->second.first) *plan = it->second.second;
  return it->second.first;
}
static bool fft_mixed_plan_search(size_t N, MixPlan *plan, bool allow_big) {
  if (N < 2 || N >= (size_t(1) << 31)) return false;
  PlanSearch ps;
  ps.allow_big = allow_big;
  ps.ex[0] = ps.ex[1] = ps.ex[2] = 0;
  const unsigned pr[3] = {2, 3, 5};
  size_t m = N;
  for (int i = 0; i < 3; ++i)
    while (m % pr[i] == 0) { m /= pr[i]; ++ps.ex[i]; }
  if (m != 1) return false;
  ps.go(0, allow_big ? 2000 : 256, 0.0);
  if (!ps.best_p) return false;
  const int p = ps.best_p;
  int last = 0;
  for (int i = 1; i < p; ++i) {
    const unsigned ti = std::min(ps.out[i][0], 4u), tl = std::min(ps.out[last][0], 4u);
    if (ti > tl || (ti == tl && PlanSearch::val(ps.out[i]) > PlanSearch::val(ps.out[last]))) last = i;
  }
  plan->p = p;
  int o = 0;
  for (int i = 0; i < p; ++i) {
    if (i == last) continue;
    plan->R[o] = PlanSearch::val(ps.out[i]);
    stage_radices(ps.out[i][0], ps.out[i][1], ps.out[i][2], plan->rad[o]);
    ++o;
  }
  plan->R[o] = PlanSearch::val(ps.out[last]);
  stage_radices(ps.out[last][0], ps.out[last][1], ps.out[last][2], plan->rad[o]);
  return true;
}

int fft_passes(size_t N) {
  if (is_pow2(N)) { int l = 0; while (((size_t)1 << l) < N) ++l; return l <= 8 ? 1 : (l + 7) / 8; }
  MixPlan pl;
  return fft_mixed_plan(N, &pl) ? pl.p : 0;
}

bool fft_mixed_ok(size_t N) {
  MixPlan pl;
  return fft_mixed_plan(N, &pl);
}

static int floor_log2(unsigned v) { int l = 0; while ((2u << l) <= v) ++l; return l; }
static int ceil_log2(unsigned v) { int l = 0; while ((1u << l) < v) ++l; return l; }

// Tile width: as wide as 4096 elements allow, narrowed (not below 16 columns = 128-byte runs) until the launch has
// enough workgroups to keep several resident per CU -- a workgroup is a chain of dependent LDS stages, and with one
// or two of them per CU nothing hides that latency.
static int pick_logT(int maxlog, int minlog, size_t other, size_t span) {
  int logT = std::max(maxlog, 0);
  while (logT > minlog && other * ceil_div(span, (size_t)1 << logT) < 2048) --logT;
  return logT;
}

static size_t mix_lds(unsigned R, int logT) {
  return ((size_t)R * ((1u << logT) + 1) + R) * sizeof(float2) + ((size_t)R * 2 + 15) / 16 * 16;
}

// W_{R*Rn}^(col*k) at [k * Rn + col], k < R, col < Rn (built once per pair in extended precision, then resident)
static int get_twg(tsdr_ctx *ctx, unsigned R, unsigned Rn, const float2 **out) {
  const unsigned key = (R << 16) | Rn;
  auto it = ctx->twg.find(key);
  if (it != ctx->twg.end()) { *out = it->second; return TSDR_OK; }
  std::vector<float2> h((size_t)R * Rn);
  const unsigned long long M = (unsigned long long)R * Rn;
  for (unsigned k = 0; k < R; ++k)
    for (unsigned c = 0; c < Rn; ++c) {
      const long double ang = -2.0L * M_PIl * (long double)(((unsigned long long)k * c) % M) / (long double)M;
      h[(size_t)k * Rn + c] = make_float2((float)cosl(ang), (float)sinl(ang));
    }
  float2 *dev = nullptr;
  TSDR_HIP(ctx, hipMalloc((void **)&dev, h.size() * sizeof(float2)));
  TSDR_HIP(ctx, hipMemcpy(dev, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
  ctx->twg.emplace(key, dev);
  *out = dev;
  return TSDR_OK;
}

// The three-step kernels (factors of 500 .. 2000) trade pass count for narrow tiles -- 8000 points are 1000 x 8 columns,
// i.e. 64-byte runs.  That wins while a pass is latency-bound and its data cache-resident (2e6 points, 16 MB: two passes of
// 16 us instead of three of 12), and loses once passes stream from HBM (2e7 points: 128-174 us per pass against 75-85 us
// for the two-step kernels' 256-byte runs).  So: only for transforms of at most 2^22 points in all.
static bool fft_big_ok(tsdr_ctx *ctx, size_t total_points) {
  return ctx->opt_fft_big && !ctx->opt_fft_no_mix2 && total_points <= (size_t(1) << 22);
}

// in/out may alias.  Uses WS_FFT_B when more than one pass is needed (callers must not hand WS_FFT_B buffers in).
// src_mode/src_n: fused first-pass loader (fft_dev.h), batch == 1 and p > 1 only; keep: complex outputs per
// transform the caller will look at (0 = all).
//   force: run this pass split instead of the planner's.  first_pass > 0: `in` already holds the output of pass
//   first_pass - 1 of that split (the fused autocorrelation middle wrote it); the remaining strided passes then run in
//   place in `in`.  work_out != nullptr: stop before the last pass and hand back the buffer the strided passes left
//   their result in.
static int fft_mixed_ex(tsdr_ctx *ctx, const float2 *in, float2 *out, size_t N, size_t batch, int dir, float scale, int src_mode,
                        size_t src_n, size_t keep, const FftEpilogue *epi, const float2 *src_aux, const MixPlan *force,
                        int first_pass, float2 **work_out) {
  MixPlan pl;
  if (force) pl = *force;
  else if (!fft_mixed_plan(N, &pl, fft_big_ok(ctx, N * batch)))
    return set_err(ctx, TSDR_EINVAL, "fft_mixed: length %zu is not 2^a*3^b*5^c", N);
  if (batch == 0) return TSDR_OK;
  if (N * batch >= (size_t(1) << 40)) return set_err(ctx, TSDR_EINVAL, "fft: batch too large");
  const int p = pl.p;
  if (src_mode != SRC_C2C && (batch != 1 || p == 1)) return set_err(ctx, TSDR_EINVAL, "fft: fused loader needs one multi-pass transform");
  MixDesc d{};
  d.dir = dir < 0 ? -1 : 1;
  d.N = N;
  d.src_mode = SRC_C2C;
  d.keep = keep ? keep : N;
  d.src_w8 = src_mode == SRC_POWER && !is_pow2(src_n) ? 4.0 / (double)src_n : 0.0;
  d.src_aux = src_aux;
  if (epi && (batch != 1 || p == 1)) return set_err(ctx, TSDR_EINVAL, "fft: epilogue needs one multi-pass transform");
  auto set_radix = [&](int i) {
    d.R = pl.R[i];
    d.nst = (int)pl.rad[i].size();
    for (int s = 0; s < d.nst; ++s) d.rad[s] = pl.rad[i][s];
    const unsigned __int128 inv = ((unsigned __int128)1 << 64) / d.R;
    d.r_hi = (unsigned)(inv >> 32);
    d.r_lo = (unsigned)inv;
  };
  if (p == 1) {
    d.mode = FFT_ROWS;
    set_radix(0);
    d.logT = pick_logT(std::min(8, floor_log2(4096u / d.R)), 0, 1, batch);
    d.scale = scale;
    d.rows = (unsigned)batch;
    if (batch >= (size_t(1) << 32)) return set_err(ctx, TSDR_EINVAL, "fft: too many rows");
    const unsigned grid = (unsigned)ceil_div(batch, (size_t)1 << d.logT);
    TSDR_LAUNCH(ctx, "fftm_rows", k_fft_mix, dim3(grid), dim3(256), mix_lds(d.R, d.logT), in, out, d);
    return TSDR_OK;
  }
  float2 *work = first_pass > 0 ? const_cast<float2 *>(in) : (float2 *)ctx->scratch(WS_FFT_B, N * batch * sizeof(float2));
  if (!work) return TSDR_ENOMEM;
  size_t P = 1;  // R_1..R_{i-1}
  size_t B = N;
  const float2 *src = in;
  for (int i = 0; i < first_pass && i < p - 1; ++i) { B /= pl.R[i]; P *= pl.R[i]; }
  static const char *const kStridedName[MIX_MAX_PASS] = {"fftm_strided1", "fftm_strided2", "fftm_strided3",
                                                         "fftm_strided4", "fftm_strided5", "fftm_strided6"};
  for (int i = first_pass; i < p - 1; ++i) {
    set_radix(i);
    B /= d.R;
    d.mode = FFT_STRIDED;
    d.src_mode = i == 0 ? src_mode : SRC_C2C;
    d.src_n = src_n;
    const Mix3Entry *m3 = ctx->opt_fft_no_mix2 ? nullptr : mix3_lookup(d.R);
    const Mix2Entry *m2 = (ctx->opt_fft_no_mix2 || m3) ? nullptr : mix2_lookup(d.R);
    // (the two- and three-step kernels keep their full tile: a narrower one leaves most threads without a step-1 DFT)
    d.logT = m3 ? m3->logT
                : pick_logT(std::min({8, m2 ? floor_log2((unsigned)m2->tm) : floor_log2(4096u / d.R), ceil_log2((unsigned)B)}),
                            m2 ? 8 : 4, batch * P, B);
    d.scale = 1.0f;
    d.A = (unsigned)P;
    d.B = (unsigned)B;
    d.tiles = (unsigned)ceil_div(B, (size_t)1 << d.logT);
    d.Bnext = (unsigned)(B / pl.R[i + 1]);
    d.Pprev = (unsigned)P;
    {
      const unsigned __int128 inv = ((unsigned __int128)1 << 64) / ((unsigned __int128)P * d.R * pl.R[i + 1]);
      d.ntw_hi = (unsigned)(inv >> 32);
      d.ntw_lo = (unsigned)inv;
    }
    d.nprev = i;
    size_t wgt = 1;
    for (int j = 0; j < i; ++j) { d.Rprev[j] = pl.R[j]; d.Wprev[j] = (unsigned)wgt; wgt *= pl.R[j]; }
    const size_t grid = batch * d.A * d.tiles;
    if (grid >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "fft: grid too large");
    d.tw_sets = 0;
    d.twg = nullptr;
    if (m3) {
      int rc3 = mix3_prepare(ctx, m3);
      if (rc3) return rc3;
      TSDR_LAUNCH(ctx, kStridedName[i], m3->strided, dim3((unsigned)grid), dim3(m3->nt), m3->lds, src, work, d);
    } else if (m2) {
      // how the two-step kernel gets its inter-pass twiddles (see the kernel): sets in LDS, or the column table
      const unsigned T = 1u << d.logT;
      if (d.Bnext % T == 0) d.tw_sets = 1;
      else if (d.Bnext == 1) { const int rc = get_twg(ctx, d.R, d.B, &d.twg); if (rc) return rc; }
      else if ((T - 1) / d.Bnext + 2 <= 4) d.tw_sets = (int)((T - 1) / d.Bnext + 2);
      TSDR_LAUNCH(ctx, kStridedName[i], m2->strided, dim3((unsigned)grid), dim3(m2->nt), mix2_lds(d.R, m2->RA, d.logT, d.tw_sets), src, work, d);
    } else {
      TSDR_LAUNCH(ctx, kStridedName[i], k_fft_mix, dim3((unsigned)grid), dim3(256), mix_lds(d.R, d.logT), src, work, d);
    }
    src = work;
    P *= d.R;
  }
  if (work_out) { *work_out = work; return TSDR_OK; }
  set_radix(p - 1);
  d.mode = FFT_LAST;
  d.src_mode = SRC_C2C;
  if (epi) d.epi = *epi;
  d.R1 = pl.R[0];
  const Mix3Entry *m3 = ctx->opt_fft_no_mix2 ? nullptr : mix3_lookup(d.R);
  const Mix2Entry *m2 = (ctx->opt_fft_no_mix2 || m3) ? nullptr : mix2_lookup(d.R);
  d.logT = m3 ? m3->logT
              : pick_logT(std::min({8, m2 ? floor_log2((unsigned)m2->tm) : floor_log2(4096u / d.R), ceil_log2(d.R1)}), m2 ? 8 : 4,
                          batch * (P / pl.R[0]), d.R1);
  d.scale = scale;
  d.Pprev = (unsigned)P;
  d.nprev = p - 1;
  {
    size_t wgt = 1;
    for (int j = 0; j < p - 1; ++j) { d.Rprev[j] = pl.R[j]; d.Wprev[j] = (unsigned)wgt; wgt *= pl.R[j]; }
  }
  d.Aprime = (unsigned)(P / pl.R[0]);
  d.k1tiles = (unsigned)ceil_div((size_t)d.R1, (size_t)1 << d.logT);
  const size_t grid = batch * d.Aprime * d.k1tiles;
  if (grid >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "fft: grid too large");
  if (m3) {
    int rc3 = mix3_prepare(ctx, m3);
    if (rc3) return rc3;
    TSDR_LAUNCH(ctx, "fftm_last", m3->last, dim3((unsigned)grid), dim3(m3->nt), m3->lds, (const float2 *)work, out, d);
  } else if (m2) {
    TSDR_LAUNCH(ctx, "fftm_last", m2->last, dim3((unsigned)grid), dim3(m2->nt), mix2_lds(d.R, m2->RA, d.logT), (const float2 *)work, out, d);
  } else {
    TSDR_LAUNCH(ctx, "fftm_last", k_fft_mix, dim3((unsigned)grid), dim3(256), mix_lds(d.R, d.logT), (const float2 *)work, out, d);
  }
  return TSDR_OK;
}


// getWelch's accumulation for any 2^a 3^b 5^c segment length up to 4096 without writing a segment spectrum: *nparts partial
// power spectra of N floats each (natural order) land in `part` (room for fft_rows_welch_parts() of them)
unsigned fft_rows_welch_parts(tsdr_ctx *ctx) { return (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 3u; }
int fft_rows_welch(tsdr_ctx *ctx, const float *sig, int is_complex, size_t N, size_t nbSeg, float *part, unsigned *nparts, bool *did) {
  *did = false;
  if (N < 2 || N > 4096 || nbSeg == 0 || nbSeg >= (size_t(1) << 31)) return TSDR_OK;
  // the whole segment as ONE factor of the generic LDS-stage kernel (the pass planner caps factors at 256 / 2000: its costs
  // are those of HBM-sized passes)
  unsigned ex[3] = {0, 0, 0};
  {
    const unsigned pr[3] = {2, 3, 5};
    size_t m = N;
    for (int i = 0; i < 3; ++i)
      while (m % pr[i] == 0) { m /= pr[i]; ++ex[i]; }
    if (m != 1) return TSDR_OK;
  }
  std::vector<unsigned char> rad;
  stage_radices(ex[0], ex[1], ex[2], rad);
  if (rad.size() > MIX_MAX_STAGE) return TSDR_OK;
  MixDesc d{};
  d.dir = -1; d.N = N; d.src_mode = SRC_C2C; d.keep = N; d.mode = FFT_ROWS; d.scale = 1.0f;
  d.R = (unsigned)N;
  d.nst = (int)rad.size();
  for (int s = 0; s < d.nst; ++s) d.rad[s] = rad[s];
  const unsigned __int128 inv = ((unsigned __int128)1 << 64) / d.R;
  d.r_hi = (unsigned)(inv >> 32);
  d.r_lo = (unsigned)inv;
  d.logT = floor_log2(4096u / d.R);
  d.rows = (unsigned)nbSeg;
  d.acc = part;
  d.rows_real = is_complex ? 0 : 1;
  if (const Mix3Entry *m3 = ctx->opt_fft_no_mix2 ? nullptr : welch3_lookup(d.R, true)) {
    // 500 / 1000 / 2000 (and 256 / 512 / 2048 / 4096 / 4000): the three-register-step kernel, 8 (4, 2, 1) segments per workgroup
    int rc3 = mix3_prepare(ctx, m3);
    if (rc3) return rc3;
    d.logT = m3->logT;
    d.mode = FFT_LAST;
    const unsigned ntiles3 = (unsigned)ceil_div(nbSeg, (size_t)1 << d.logT);
    // (at most fft_rows_welch_parts() workgroups -- the caller's buffer holds that many partial spectra --, i.e. three per CU:
    // every partial is one more row for k_welch_sum to add, and the short lengths' small tiles would otherwise put eight
    // workgroups on a CU)
    const unsigned per_cu3 = (unsigned)std::max<size_t>(1, std::min<size_t>(3, (size_t)(160 * 1024) / m3->lds));
    const unsigned grid3 = std::min({ntiles3, (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * per_cu3, fft_rows_welch_parts(ctx)});
    TSDR_LAUNCH(ctx, "welch_rows_acc3", m3->acc, dim3(grid3), dim3(m3->nt), m3->lds, reinterpret_cast<const float2 *>(sig), (float2 *)nullptr, d);
    *nparts = grid3;
    *did = true;
    return TSDR_OK;
  }
  const size_t lds = mix_lds(d.R, d.logT);
  if (lds > 64 * 1024) {  // (4096-point tiles + tables: above what a kernel gets without opting in)
    int rco = lds_opt_in(ctx, (const void *)k_fft_mix, lds);
    if (rco) return rco;
  }
  const unsigned ntiles = (unsigned)ceil_div(nbSeg, (size_t)1 << d.logT);
  const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(3, (size_t)(150 * 1024) / lds));
  const unsigned grid = std::min(ntiles, (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * per_cu);
  TSDR_LAUNCH(ctx, "welch_rows_acc", k_fft_mix, dim3(grid), dim3(256), lds, reinterpret_cast<const float2 *>(sig), (float2 *)nullptr, d);
  *nparts = grid;
  *did = true;
  return TSDR_OK;
}

// Batched row transforms (tsdr_fft_c2c with batch > 1) of the lengths the three-step kernels serve, in ONE launch: a row never
// leaves the chip between its steps.  (The pass engines split a 512 .. 4096-point row into two passes whose strided one has only
// 16-64 columns to work on: 67-197 us for 1e7 points against 35-50 us here; rows up to 256 points are one pass there already.)
int fft_rows_store(tsdr_ctx *ctx, const float2 *in, float2 *out, size_t N, size_t batch, int dir, float scale, bool *did) {
  *did = false;
  if (N <= 256 || N > 4096 || batch < 2 || batch >= (size_t(1) << 31) || ctx->opt_fft_no_mix2) return TSDR_OK;
  const Mix3Entry *m3 = welch3_lookup((unsigned)N, false);
  if (!m3) return TSDR_OK;
  MixDesc d{};
  d.dir = dir < 0 ? -1 : 1; d.N = N; d.src_mode = SRC_C2C; d.keep = N; d.mode = FFT_LAST; d.scale = scale;
  d.R = (unsigned)N;
  const unsigned __int128 inv = ((unsigned __int128)1 << 64) / d.R;
  d.r_hi = (unsigned)(inv >> 32);
  d.r_lo = (unsigned)inv;
  d.logT = m3->logT;
  d.rows = (unsigned)batch;
  d.rows_out = out;
  int rc = mix3_prepare(ctx, m3);
  if (rc) return rc;
  const unsigned ntiles = (unsigned)ceil_div(batch, (size_t)1 << d.logT);
  const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(3, (size_t)(160 * 1024) / m3->lds));
  const unsigned grid = std::min(ntiles, (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * per_cu);
  TSDR_LAUNCH(ctx, "fft_rows3", m3->rows, dim3(grid), dim3(m3->nt), m3->lds, in, (float2 *)nullptr, d);
  *did = true;
  return TSDR_OK;
}

// getWaterfall for the segment lengths the three-step kernels serve (1024 has k_seg1024): segments -> Float64 power spectra,
// fftshifted, in ONE launch -- the segment spectra never reach HBM (the route through a batched FFT + k_waterfall writes and
// re-reads them: 109-250 us per C2 buffer at 512 .. 4096 against 40-60 us here)
int fft_rows_waterfall(tsdr_ctx *ctx, const float *sig, int is_complex, size_t N, size_t nbSeg, double *wf, bool *did) {
  *did = false;
  if (N < 2 || N > 4096 || nbSeg == 0 || nbSeg >= (size_t(1) << 31) || ctx->opt_fft_no_mix2) return TSDR_OK;
  const Mix3Entry *m3 = welch3_lookup((unsigned)N, false);
  if (!m3) return TSDR_OK;
  MixDesc d{};
  d.dir = -1; d.N = N; d.src_mode = SRC_C2C; d.keep = N; d.mode = FFT_LAST; d.scale = 1.0f;
  d.R = (unsigned)N;
  const unsigned __int128 inv = ((unsigned __int128)1 << 64) / d.R;
  d.r_hi = (unsigned)(inv >> 32);
  d.r_lo = (unsigned)inv;
  d.logT = m3->logT;
  d.rows = (unsigned)nbSeg;
  d.wf = wf;
  d.rows_real = is_complex ? 0 : 1;
  int rc = mix3_prepare(ctx, m3);
  if (rc) return rc;
  const unsigned ntiles = (unsigned)ceil_div(nbSeg, (size_t)1 << d.logT);
  const unsigned per_cu = (unsigned)std::max<size_t>(1, (size_t)(160 * 1024) / m3->lds);
  const unsigned grid = std::min(ntiles, (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * per_cu);
  TSDR_LAUNCH(ctx, "waterfall_rows3", m3->wf, dim3(grid), dim3(m3->nt), m3->lds, reinterpret_cast<const float2 *>(sig), (float2 *)nullptr, d);
  *did = true;
  return TSDR_OK;
}

int fft_mixed(tsdr_ctx *ctx, const float2 *in, float2 *out, size_t N, size_t batch, int dir, float scale, int src_mode,
              size_t src_n, size_t keep, const FftEpilogue *epi, const float2 *src_aux) {
  return fft_mixed_ex(ctx, in, out, N, batch, dir, scale, src_mode, src_n, keep, epi, src_aux, nullptr, 0, nullptr);
}

// The circular autocorrelation of n = 2*Mc real samples (x, or abs2 of IQ formed while loading) as
//   forward passes 1..p-1  ->  [last forward pass + power spectrum + first inverse pass] (k_fft_mid)  ->  inverse passes 2..p
// with the epilogue (abs2 / 10log10 of the wanted lags, optional findmax) on the last one.  *done = false (nothing
// launched) when this length has no fused middle: the caller then runs the two transforms separately.
int fft_mixed_autocorr(tsdr_ctx *ctx, const float2 *x, int src_mode, size_t src_n, size_t Mc, float2 *Zbuf, float2 *zbuf,
                       float scale, size_t keep, const FftEpilogue *epi, bool *done) {
  *done = false;
  MixPlan F;
  if (ctx->opt_fft_no_mix2 || !fft_mixed_plan(Mc, &F, fft_big_ok(ctx, Mc)) || F.p < 2 || Mc >= (size_t(1) << 31)) return TSDR_OK;
  const int p = F.p;
  // the factor that goes LAST in the forward split (and first in the inverse one) must have the fused kernel: of those
  // that do, the one with the most twos (as the planner's own rule); the others keep their order
  {
    auto twos = [&](int i) { unsigned v = F.R[i], t = 0; while (v % 2 == 0 && t < 4) { v /= 2; ++t; } return t; };
    int pick = -1;
    for (int i = 0; i < p; ++i)
      if (mid_lookup(F.R[i]) && (pick < 0 || twos(i) > twos(pick) || (twos(i) == twos(pick) && F.R[i] > F.R[pick]))) pick = i;
    // (2e6 points: 1000 | 2000-mid | 1000 and 2000 | 1000-mid | 2000 measured the same, 74-75 us per search)
    if (pick < 0) return TSDR_OK;
    if (pick != p - 1) {
      const unsigned r = F.R[pick];
      const std::vector<unsigned char> rd = F.rad[pick];
      for (int i = pick; i < p - 1; ++i) { F.R[i] = F.R[i + 1]; F.rad[i] = F.rad[i + 1]; }
      F.R[p - 1] = r;
      F.rad[p - 1] = rd;
    }
  }
  const MidEntry *me = mid_lookup(F.R[p - 1]);
  // inverse split: the forward's last factor first; of the others the one with the most twos last, the rest largest first
  MixPlan I;
  I.p = p;
  I.R[0] = F.R[p - 1];
  I.rad[0] = F.rad[p - 1];
  {
    std::vector<int> rest;
    for (int i = 0; i < p - 1; ++i) rest.push_back(i);
    auto twos = [&](int i) { unsigned v = F.R[i], t = 0; while (v % 2 == 0 && t < 4) { v /= 2; ++t; } return t; };
    int last = rest[0];
    for (int i : rest)
      if (twos(i) > twos(last) || (twos(i) == twos(last) && F.R[i] > F.R[last])) last = i;
    std::vector<int> order;
    for (int i : rest) if (i != last) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return F.R[a] > F.R[b]; });
    order.push_back(last);
    for (int o = 0; o < (int)order.size(); ++o) { I.R[o + 1] = F.R[order[o]]; I.rad[o + 1] = F.rad[order[o]]; }
  }
  MidDesc m{};
  m.R = F.R[p - 1];
  m.Bc = (unsigned)(Mc / m.R);
  m.ndir = m.Bc / 2 + 1;
  m.logT = floor_log2((unsigned)me->tm);
  if (m.logT < 1) return TSDR_OK;
  m.nprev = p - 1;
  for (int j = 0; j < p - 1; ++j) m.Rprev[j] = F.R[j];
  {
    const unsigned __int128 inv = ((unsigned __int128)1 << 64) / m.R;
    m.r_hi = (unsigned)(inv >> 32);
    m.r_lo = (unsigned)inv;
    const unsigned __int128 inv2 = ((unsigned __int128)1 << 64) / ((unsigned __int128)m.R * I.R[1]);
    m.ntw_hi = (unsigned)(inv2 >> 32);
    m.ntw_lo = (unsigned)inv2;
  }
  m.Bnext = m.Bc / I.R[1];
  const unsigned Th = 1u << (m.logT - 1);
  const unsigned sets = (Th - 1) / m.Bnext + 2;
  m.tw_sets = sets <= 4 ? (int)sets : 0;
  m.w8 = 4.0 / (double)Mc;
  float2 *w = nullptr;
  int rc = fft_mixed_ex(ctx, x, nullptr, Mc, 1, -1, 1.0f, src_mode, src_n, 0, nullptr, nullptr, &F, 0, &w);
  if (rc) return rc;
  if (me->RA == 0) {  // three-step kernel
    const size_t lds = me->lds3;
    rc = lds_opt_in(ctx, (const void *)me->fn, lds);
    if (rc) return rc;
    const unsigned grid = (unsigned)ceil_div((size_t)m.ndir, (size_t)Th);
    TSDR_LAUNCH(ctx, "fftm_mid", me->fn, dim3(grid), dim3(me->nt), lds, (const float2 *)w, Zbuf, m);
  } else {
    const size_t T = (size_t)1 << m.logT, RB = m.R / me->RA;
    const size_t SA = (RB << m.logT) + (T < 32 ? T : 0);
    const size_t lds = (std::max((size_t)m.R * (T + 1), (size_t)me->RA * SA) + (size_t)(1 + 2 * m.tw_sets) * m.R) * sizeof(float2) + 2 * T * 4;
    const unsigned grid = (unsigned)ceil_div((size_t)m.ndir, (size_t)Th);
    TSDR_LAUNCH(ctx, "fftm_mid", me->fn, dim3(grid), dim3(me->nt), lds, (const float2 *)w, Zbuf, m);
  }
  rc = fft_mixed_ex(ctx, Zbuf, zbuf, Mc, 1, +1, scale, SRC_C2C, 0, keep, epi, nullptr, &I, 1, nullptr);
  if (rc) return rc;
  *done = true;
  return TSDR_OK;
}

}  // namespace tsdr

extern "C" int tsdr_fft_plan(size_t n, unsigned *factors, int cap) {
  using namespace tsdr;
  if (n < 2) return 0;
  if (is_pow2(n)) {  // fft.hip's split: the bits dealt evenly over ceil(log2 n / 8) passes
    int l = 0;
    while (((size_t)1 << l) < n) ++l;
    const int p = l <= 8 ? 1 : (l + 7) / 8;
    for (int i = 0; i < p && i < cap && factors; ++i) factors[i] = 1u << (l / p + (i < l % p ? 1 : 0));
    return p;
  }
  MixPlan pl;
  if (!fft_mixed_plan(n, &pl, n <= (size_t(1) << 22))) return 0;  // (as fft_big_ok with the default options, batch 1)
  for (int i = 0; i < pl.p && i < cap && factors; ++i) factors[i] = pl.R[i];
  return pl.p;
}

