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
// A workgroup owns a tile of R x T elements (T = 16..: T neighbouring columns are contiguous in
// HBM, so every global access is a >=128-byte run; the last pass tiles over k_1 so its transposed
// store is contiguous too).  The length-R DFT is two register-resident radix-<=16 steps with one
// exchange through LDS between them (see k_fft_pass).
// In-tile twiddles come from an f64-generated W_4096 table; inter-pass twiddles are evaluated in
// registers from the exact integer phase (tw_unit, <= 1.5 ulp for any N).  The two-level tables
// (W = hi[e >> h] * lo[e & mask], get_tw) serve the real-FFT post-processing kernels.
//
// Other lengths: Bluestein (chirp-z) on top of the power-of-two engine; chirp phases are reduced
// exactly (k^2 mod 2n in 64-bit integers) and evaluated in f64.
#include <cmath>

#include "fft_dev.h"

namespace tsdr {

struct PassDesc {
  int mode, logR, logT, dir;
  float scale;
  unsigned long long N;  // elements per transform
  unsigned A, B, tiles;  // strided: outer count, inner size (= stride of the DFT index), B/T
  int logNtw, logBnext, logPprev;
  int nprev;
  int logRprev[4];       // radices of the passes before this one (pass order)
  int logR1;
  unsigned Aprime, k1tiles;  // last pass: A / R_1, R_1 / T
  unsigned rows;             // rows mode: number of transforms
  int src_mode;              // first pass loader (SRC_*); only with batch == 1
  unsigned long long src_n;  // real samples behind SRC_REAL / SRC_IQPOW
  unsigned long long keep;   // last pass: complex outputs >= keep (per transform) are not stored
  const float2 *src_aux;     // SRC_MULH: the factor array
  FftEpilogue epi;           // last pass: epilogue when epi.out != nullptr (fft_dev.h)
};

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

// One pass over a tile of R x T (<= 4096) elements per 256-thread workgroup, 16 elements per thread.
// R = RA * RB (RA <= 16): with n = j0 + RB*m and k = ka + RA*kb
//   X[ka + RA*kb] = sum_j0 W_RB^(j0*kb) * ( W_R^(j0*ka) * sum_m W_RA^(m*ka) x[j0 + RB*m] ).
// Step 1 runs the RA-point DFTs over m in registers (thread = column t, residue j0), multiplies by W_R^(j0*ka)
// and passes the tile through LDS once; step 2 runs the RB-point DFTs over j0 in registers (thread = column t,
// frequency ka) and stores straight from registers.  So the tile crosses LDS once (write + read) instead of
// once per radix-4 stage, and all 16 global loads of a thread are in flight together.
//   STRIDED: global -> registers (columns t are contiguous in HBM), ..., registers -> global with the
//            inter-pass twiddle.
//   LAST   : rows are contiguous over the DFT index, so the tile is first staged through LDS with a linear,
//            fully coalesced load; the store is contiguous over the k_1 tile.
//   ROWS   : as LAST, and the result is staged back through LDS for a linear store.
// The inverse transform is conj(F(conj(x))): the sign bit of the imaginary part is flipped on the way in and
// out, every twiddle is the forward one.
template <int LOGR>
struct Split {
  static constexpr int LA = LOGR <= 4 ? LOGR : (LOGR + 1) / 2;
  static constexpr int LB = LOGR - LA;
};

template <int LOGR, int MODE>
__global__ __launch_bounds__(256) void k_fft_pass(const float2 *__restrict__ in, float2 *__restrict__ out, PassDesc d,
                                                  const float2 *__restrict__ tw_small) {
  constexpr int LA = Split<LOGR>::LA, LB = Split<LOGR>::LB;
  constexpr int R = 1 << LOGR, RA = 1 << LA, RB = 1 << LB;
  constexpr int CA = 16 / RA, CB = 16 / RB;  // DFTs per thread in step 1 / step 2
  extern __shared__ float2 sm[];
  const int logT = d.logT, T = 1 << logT, TP = T + 1;
  const int SA = (RB << logT) + (T < 32 ? T : 0);  // pitch of one ka plane of the exchange buffer
  float2 *buf = sm;                                // staging tile [j][TP] and exchange buffer [ka][SA] (aliased)
  float2 *twR = sm + 4096 + 256 + 16;
  float2 *twK = twR + 256;  // STRIDED: per-frequency inter-pass twiddles when they are uniform over the tile
  bool tw_shared = false;
  const int tid = threadIdx.x;
  const unsigned smask = d.dir > 0 ? 0x80000000u : 0u;
  if (RB > 1)
    for (int e = tid; e < R; e += 256) twR[e] = tw_small[e << (12 - LOGR)];
  const unsigned bid = blockIdx.x;
  const int work = R << logT;
  const int n1 = RB << logT, n2 = RA << logT;  // DFT slots of step 1 / step 2

  constexpr int CO = RB > 1 ? CB : CA;  // output slots per thread
  constexpr int RO = RB > 1 ? RB : RA;  // outputs per slot
  const int no = RB > 1 ? n2 : n1;

  float2 v[16];
  float2 tw[MODE == FFT_STRIDED ? 16 : 1];
  size_t base = 0, tbase = 0, row0 = 0;
  unsigned tile = 0, a = 0, kt = 0, arest = 0;
  if (MODE == FFT_STRIDED) {
    tile = bid % d.tiles;
    a = (bid / d.tiles) % d.A;
    const unsigned b = bid / (d.tiles * d.A);
    base = (size_t)b * d.N + (size_t)a * R * d.B + (size_t)tile * T;
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int s = tid + 256 * q;
      const int t = s & (T - 1), j0 = s >> logT;
#pragma unroll
      for (int m = 0; m < RA; ++m)
        v[q * RA + m] = s < n1 ? conj_if(fft_load(in, d.src_mode, d.src_n, base + (size_t)(j0 + RB * m) * d.B + t, 0.0, d.src_aux), smask)
                               : make_float2(0.f, 0.f);
    }
    // inter-pass twiddles of this thread's 16 outputs: independent of the data, so they are evaluated here,
    // under the latency of the loads just issued.  When the whole tile lies inside one n_{i+1} (always in the
    // first of three or more passes) the twiddle depends on the output frequency only: the workgroup evaluates
    // its R values once into LDS (twK) instead of 16 per thread, and the threads pick theirs up after the barrier.
    const unsigned Ka = digit_swap(a, d.nprev, d.logRprev);
    const unsigned mask = (d.logNtw >= 32) ? 0xFFFFFFFFu : ((1u << d.logNtw) - 1u);
    tw_shared = d.logBnext >= logT;
    if (tw_shared) {
      const unsigned nnext = (tile * T) >> d.logBnext;
      for (int k = tid; k < R; k += 256) twK[k] = tw_unit((nnext * (Ka + ((unsigned)k << d.logPprev))) & mask, d.logNtw);
    } else {
#pragma unroll
      for (int q = 0; q < CO; ++q) {
        const int s = min(tid + 256 * q, no - 1);
        const unsigned nnext = (tile * T + (unsigned)(s & (T - 1))) >> d.logBnext;
#pragma unroll
        for (int i = 0; i < RO; ++i) {
          const int k = RB > 1 ? ((s >> logT) + RA * i) : i;
          tw[q * RO + i] = tw_unit((nnext * (Ka + ((unsigned)k << d.logPprev))) & mask, d.logNtw);
        }
      }
    }
  } else {
    if (MODE == FFT_LAST) {
      kt = bid % d.k1tiles;
      arest = (bid / d.k1tiles) % d.Aprime;
      tbase = (size_t)(bid / (d.k1tiles * d.Aprime)) * d.N;
    } else {
      row0 = (size_t)bid * T;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int w = min(tid + 256 * u, work - 1);
      const int j = w & (R - 1), t = w >> LOGR;
      if (MODE == FFT_LAST) {
        v[u] = in[tbase + ((size_t)(kt * T + (unsigned)t) * d.Aprime + arest) * R + j];
      } else {
        const size_t row = row0 + t;
        v[u] = row < d.rows ? in[row * R + j] : make_float2(0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int w = tid + 256 * u;
      if (w < work) buf[(w & (R - 1)) * TP + (w >> LOGR)] = conj_if(v[u], smask);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int s = min(tid + 256 * q, n1 - 1);
      const int t = s & (T - 1), j0 = s >> logT;
#pragma unroll
      for (int m = 0; m < RA; ++m) v[q * RA + m] = buf[(j0 + RB * m) * TP + t];
    }
  }
  __syncthreads();  // twR visible; staging tile fully read before the exchange buffer overwrites it

  // ---- step 1: RA-point DFTs over m, twiddle, exchange ----
#pragma unroll
  for (int q = 0; q < CA; ++q) reg_dft<RA>(v + q * RA);
  if (RB > 1) {
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int s = tid + 256 * q;
      const int j0 = s >> logT;
      if (s < n1) {
#pragma unroll
        for (int ka = 0; ka < RA; ++ka) {
          float2 x = v[q * RA + brev<RA>(ka)];
          if (ka) x = cmul(x, twR[j0 * ka]);
          buf[ka * SA + s] = x;  // (j0, t) is the slot number itself
        }
      }
    }
    __syncthreads();
    // ---- step 2: RB-point DFTs over j0 ----
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      const int s = min(tid + 256 * q, n2 - 1);
      const int t = s & (T - 1), ka = s >> logT;
#pragma unroll
      for (int j0 = 0; j0 < RB; ++j0) v[q * RB + j0] = buf[ka * SA + (j0 << logT) + t];
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) reg_dft<RB>(v + q * RB);
  } else {
    // single-step radix (R <= 16): put the result in natural order, one "DFT of one point" per slot below
    float2 n[16];
#pragma unroll
    for (int q = 0; q < CA; ++q)
#pragma unroll
      for (int ka = 0; ka < RA; ++ka) n[q * RA + ka] = v[q * RA + brev<RA>(ka)];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = n[i];
  }
  // Register i of the thread now holds, for RB > 1, slot q = i / RB (column t, frequency ka) and X[ka + RA*kb]
  // with kb = brev(i % RB); for RB == 1, slot q = i / RA (column t) and X[i % RA].

  if (MODE == FFT_STRIDED) {
#pragma unroll
    for (int q = 0; q < CO; ++q) {
      const int s = tid + 256 * q;
      const int t = s & (T - 1);
      if (s < no) {
#pragma unroll
        for (int i = 0; i < RO; ++i) {
          const int k = RB > 1 ? ((s >> logT) + RA * i) : i;
          const float2 x = RB > 1 ? v[q * RB + brev<RB>(i)] : v[q * RA + i];
          out[base + (size_t)k * d.B + t] = conj_if(cmul(x, tw_shared ? twK[k] : tw[q * RO + i]), smask);
        }
      }
    }
  } else if (MODE == FFT_LAST) {
    // digits k_2..k_{p-1} of arest -> their natural-order weight (already multiples of R_1)
    const unsigned Kp = digit_swap(arest, d.nprev - 1, d.logRprev + 1) << d.logR1;
    const size_t orel = (size_t)kt * T + Kp;
#pragma unroll
    for (int q = 0; q < CO; ++q) {
      const int s = tid + 256 * q;
      const int t = s & (T - 1);
      if (s < no) {
#pragma unroll
        for (int i = 0; i < RO; ++i) {
          const int k = RB > 1 ? ((s >> logT) + RA * i) : i;
          const float2 x = RB > 1 ? v[q * RB + brev<RB>(i)] : v[q * RA + i];
          const size_t o = orel + ((size_t)k << d.logPprev) + t;
          // outputs past `keep` are never looked at by the caller (e.g. lags beyond the window)
          if (o < d.keep) {
            const float2 y = conj_if(make_float2(x.x * d.scale, x.y * d.scale), smask);
            if (d.epi.out) epilogue_store(d.epi, o, y); else out[tbase + o] = y;
          }
        }
      }
    }
  } else {
    __syncthreads();  // exchange buffer fully read
#pragma unroll
    for (int q = 0; q < CO; ++q) {
      const int s = tid + 256 * q;
      const int t = s & (T - 1);
      if (s < no) {
#pragma unroll
        for (int i = 0; i < RO; ++i) {
          const int k = RB > 1 ? ((s >> logT) + RA * i) : i;
          buf[k * TP + t] = RB > 1 ? v[q * RB + brev<RB>(i)] : v[q * RA + i];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int w = tid + 256 * u;
      const int k = w & (R - 1), t = w >> LOGR;
      const size_t row = row0 + t;
      if (w < work && row < d.rows) {
        const float2 x = buf[k * TP + t];
        out[row * R + k] = conj_if(make_float2(x.x * d.scale, x.y * d.scale), smask);
      }
    }
  }
}

typedef void (*fft_pass_fn)(const float2 *, float2 *, PassDesc, const float2 *);

template <int MODE>
static fft_pass_fn pass_fn(int logR) {
  switch (logR) {
    case 1: return k_fft_pass<1, MODE>;
    case 2: return k_fft_pass<2, MODE>;
    case 3: return k_fft_pass<3, MODE>;
    case 4: return k_fft_pass<4, MODE>;
    case 5: return k_fft_pass<5, MODE>;
    case 6: return k_fft_pass<6, MODE>;
    case 7: return k_fft_pass<7, MODE>;
    default: return k_fft_pass<8, MODE>;
  }
}
static const size_t kPassLds = (4096 + 256 + 16 + 256 + 256) * sizeof(float2);

// ---- twiddle tables ------------------------------------------------------------------------
int ensure_tw_small(tsdr_ctx *ctx) {  // W_4096^e, e < 4096 (ctx->tw_small)
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
int fft_pow2(tsdr_ctx *ctx, const float2 *in, float2 *out, int logN, size_t batch, int dir, float scale, int src_mode,
             size_t src_n, size_t keep, const FftEpilogue *epi, const float2 *src_aux) {
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
  d.src_mode = SRC_C2C;
  d.src_n = 0;
  d.keep = keep ? keep : N;
  if ((src_mode != SRC_C2C || epi) && (batch != 1 || logN <= 8)) return set_err(ctx, TSDR_EINVAL, "fft: fused loader / epilogue needs one multi-pass transform");
  d.src_aux = src_aux;
  if (p == 1) {
    d.mode = FFT_ROWS;
    d.logR = logN;
    d.logT = 12 - logN;  // R*T = 4096
    d.scale = scale;
    d.rows = (unsigned)batch;
    if (batch >= (size_t(1) << 32)) return set_err(ctx, TSDR_EINVAL, "fft: too many rows");
    const int T = 1 << d.logT;
    const unsigned grid = (unsigned)ceil_div(batch, (size_t)T);
    TSDR_LAUNCH(ctx, "fft_rows", pass_fn<FFT_ROWS>(d.logR), dim3(grid), dim3(256), kPassLds, in, out, d, (const float2 *)ctx->tw_small);
    return TSDR_OK;
  }
  float2 *work = (float2 *)ctx->scratch(WS_FFT_B, N * batch * sizeof(float2));
  if (!work) return TSDR_ENOMEM;
  int logP = 0;  // log2(R_1..R_{i-1})
  const float2 *src = in;
  for (int i = 0; i < p - 1; ++i) {
    const int logB = logN - logP - bits[i];
    d.mode = FFT_STRIDED;
    d.src_mode = i == 0 ? src_mode : SRC_C2C;
    d.src_n = src_n;
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
    const size_t grid = batch * d.A * d.tiles;
    if (grid >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "fft: grid too large");
    static const char *const kStridedName[3] = {"fft_strided1", "fft_strided2", "fft_strided3"};
    TSDR_LAUNCH(ctx, kStridedName[i], pass_fn<FFT_STRIDED>(d.logR), dim3((unsigned)grid), dim3(256), kPassLds, src, work, d,
                (const float2 *)ctx->tw_small);
    src = work;
    logP += bits[i];
  }
  d.mode = FFT_LAST;
  d.src_mode = SRC_C2C;
  if (epi) d.epi = *epi;
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
    const size_t grid = batch * d.Aprime * d.k1tiles;
    if (grid >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "fft: grid too large");
    TSDR_LAUNCH(ctx, "fft_last", pass_fn<FFT_LAST>(d.logR), dim3((unsigned)grid), dim3(256), kPassLds, (const float2 *)work, out, d,
                (const float2 *)ctx->tw_small);
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
  int rc = fft_pow2(ctx, tmp, pl.bfft, ilog2(pl.L), 1, -1, 1.0f, SRC_C2C, 0, 0);
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
  if (is_complex && batch > 1 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0) {  // rows of 257 .. 4096 points: one launch, on chip
    if (n == 1024) return fft_rows1024(ctx, reinterpret_cast<const float2 *>(x), out, batch, d, d > 0 ? (float)(1.0 / 1024.0) : 1.0f);
    bool did = false;
    int rcr = fft_rows_store(ctx, reinterpret_cast<const float2 *>(x), out, n, batch, d, d > 0 ? (float)(1.0 / (double)n) : 1.0f, &did);
    if (rcr || did) return rcr;
  }
  if (is_pow2(n)) {
    const float scale = d > 0 ? (float)(1.0 / (double)n) : 1.0f;
    const float2 *src = reinterpret_cast<const float2 *>(x);
    if (!is_complex) {
      TSDR_LAUNCH(ctx, "r2c", k_r2c, dim3(stream_grid(ctx, n * batch)), dim3(256), 0, x, n * batch, out);
      src = out;
    }
    return fft_pow2(ctx, src, out, ilog2(n), batch, d, scale, SRC_C2C, 0, 0);
  }
  if (fft_mixed_ok(n)) {  // 2^a 3^b 5^c: native mixed-radix passes (fft_mixed.hip)
    const float scale = d > 0 ? (float)(1.0 / (double)n) : 1.0f;
    const float2 *src = reinterpret_cast<const float2 *>(x);
    if (!is_complex) {
      TSDR_LAUNCH(ctx, "r2c", k_r2c, dim3(stream_grid(ctx, n * batch)), dim3(256), 0, x, n * batch, out);
      src = out;
    }
    return fft_mixed(ctx, src, out, n, batch, d, scale, SRC_C2C, 0, 0);
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
  rc = fft_pow2(ctx, a, a2, ilog2(L), batch, -1, 1.0f, SRC_C2C, 0, 0);
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "blu_mul", k_cmul_bcast, dim3(stream_grid(ctx, L * batch)), dim3(256), 0, a2, (const float2 *)pl->bfft, L,
              batch);
  rc = fft_pow2(ctx, a2, a, ilog2(L), batch, +1, (float)(1.0 / (double)L), SRC_C2C, 0, 0);
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
