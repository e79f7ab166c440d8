// spectrum.hip -- GetSpectrum.jl on gfx950 (getSpectrum :21-30, getWelch :36-52,
// getWaterfall :54-66) and the init_resampler closure of Resampler.jl:26-99.
// All transforms go through the hand-written FFT engine (fft.hip).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "fft_dev.h"

struct tsdr_resampler {
  tsdr_ctx *ctx;
  size_t bufferSize, sizeFFT;
  int up;
  double2 *H = nullptr;    // initLPF's H: ComplexF64 as in the reference (the Float64 window promotes it, Resampler.jl:93-97)
  double2 *Hs = nullptr;   // (H[k] + conj H[N-k]) / 2 for k <= N/2: the filter of the real part (half-size route), or null
  float2 *work = nullptr;  // containerFFT / inFFT / outFFT, device
  double2 *tw = nullptr;   // half-size route: {cos, sin}(2 pi e / N) for e < 1024, then for e = 1024 h (h <= N / 2048 + 1)
};

namespace tsdr {

int fft64_d(tsdr_ctx *ctx, double2 *data, double2 *scratch, size_t N, int dir);

int fft_any(tsdr_ctx *ctx, const float *x, int is_complex, float2 *out, size_t n, size_t batch, int dir);

// fftshift: output j takes input (j + ceil(N/2)) mod N
__device__ inline size_t shift_src(size_t j, size_t N) {
  size_t s = j + (N - N / 2);
  return s >= N ? s - N : s;
}

__global__ __launch_bounds__(256) void k_spec_out(const float2 *__restrict__ X, size_t N, int lin, float *__restrict__ y) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (size_t)gridDim.x * blockDim.x) {
    const float2 c = X[shift_src(j, N)];
    const float p = c.x * c.x + c.y * c.y;
    y[j] = lin ? p : 10.0f * log10f(p);
  }
}

// S[k] = sum over segments of |X_seg[k]|^2, then fftshift and optional dB -- in two levels so that the launch fills
// the device: block (x, c) adds the segments of chunk c for 256 frequencies (ascending, eight loads in flight), the
// second kernel adds the chunk sums (sixteen interleaved partial sums, then those in order).  With one chunk
// (nbSeg <= chunk) this is the reference's strict segment-by-segment order (GetSpectrum.jl:44); with more it is a fixed
// blocked order like the 1024-point path's.
__global__ __launch_bounds__(256) void k_welch_part(const float2 *__restrict__ X, size_t sizeFFT, size_t nbSeg, size_t chunk,
                                                    float *__restrict__ part) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= sizeFFT) return;
  const size_t s0 = (size_t)blockIdx.y * chunk, s1 = s0 + chunk < nbSeg ? s0 + chunk : nbSeg;
  float S = 0.f;
  size_t s = s0;
  for (; s + 8 <= s1; s += 8) {
    float2 c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = X[(s + u) * sizeFFT + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) S += c[u].x * c[u].x + c[u].y * c[u].y;
  }
  for (; s < s1; ++s) {
    const float2 c = X[s * sizeFFT + k];
    S += c.x * c.x + c.y * c.y;
  }
  part[(size_t)blockIdx.y * sizeFFT + k] = S;
}

// grid = ceil(sizeFFT / 16) blocks of 256 threads: thread (kq, g) adds chunk sums g, g + 16, ... of frequency
// 16 * blockIdx.x + kq in ascending order, the 16 partial results are then added in order of g (one thread per
// frequency looping over 256 chunk sums alone took 62 us of the general path's 152)
__global__ __launch_bounds__(256) void k_welch_sum(const float *__restrict__ part, size_t sizeFFT, unsigned nparts, int lin,
                                                   float *__restrict__ y) {
  __shared__ float sm[16][17];
  const int kq = threadIdx.x & 15, g = threadIdx.x >> 4;
  const size_t j = (size_t)blockIdx.x * 16 + kq;
  const size_t k = j < sizeFFT ? shift_src(j, sizeFFT) : 0;
  float t = 0.f;
  if (j < sizeFFT)
    for (unsigned c = g; c < nparts; c += 16) t += part[(size_t)c * sizeFFT + k];
  sm[g][kq] = t;
  __syncthreads();
  if (g == 0 && j < sizeFFT) {
    float S = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) S += sm[i][kq];
    y[j] = lin ? S : 10.0f * log10f(S);
  }
}

__global__ __launch_bounds__(256) void k_waterfall(const float2 *__restrict__ X, size_t sizeFFT, size_t nbSeg,
                                                   double *__restrict__ m) {
  const size_t total = sizeFFT * nbSeg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t s = i / sizeFFT, j = i - s * sizeFFT;
    const float2 c = X[s * sizeFFT + shift_src(j, sizeFFT)];
    m[i] = (double)(c.x * c.x + c.y * c.y);
  }
}

// ---- 1024-point segments without leaving the chip (getWelch / getWaterfall at their default sizeFFT) -------------
// One WAVEFRONT transforms one segment: 64 lanes x 16 complex values in registers, two exchanges through a private
// 8.1 KiB LDS region, no workgroup barrier inside a transform.  With n = l + 64 m (l = lane, m < 16) and
// k = ka + 16 (kb1 + 16 kb0):
//   step 1   A[ka]   = sum_m x[l + 64 m] W_16^(m ka)                  16-point DFT in registers, then * W_1024^(l ka)
//   LDS      Z1[ka][l]  (row pitch 65 complex = 130 words: the 16 lanes the LDS serves per cycle of the transposed
//            8-byte read below fall into 32 distinct banks; 66 made them collide two by two)
//   step 2   lane = (ka, l0), l = l0 + 4 l1:  B[kb1] = sum_l1 Z1[ka][l0 + 4 l1] W_16^(l1 kb1),  then * W_64^(l0 kb1)
//   LDS      Z2[kb1][l0][ka]  (ka innermost: both the write and the read below touch consecutive words)
//   step 3   lane l' reads, for j < 4, the four l0 of (ka = l' & 15, kb1 = 4 j + (l' >> 4)): 4-point DFT over l0
//            -> X[l' + 64 j + 256 kb0]: for a fixed register the 64 lanes hold 64 consecutive frequencies, so
//            every global access of the kernel -- loads included -- is one contiguous run per wave-instruction.
// A wavefront walks a contiguous chunk of segments.  Welch: |X|^2 is accumulated in registers across the chunk
// (ascending segments), the wavefronts of a workgroup are added in order through LDS and each workgroup leaves one
// partial spectrum; k_welch_finish adds those in index order, 16 at a time (so the sum over segments is a fixed
// blocked order: chunk -> workgroup -> groups of 16 workgroups -> total).  The reference accumulates its f32 sum
// strictly segment by segment (GetSpectrum.jl:44); with f32 FFT outputs that already differ from FFTW's in the last
// bits the order cannot make the result bit-exact either way, and the blocked order has the smaller rounding error.
// Waterfall: Float64(|X|^2) goes straight from the registers to sMatrix[:, segment] (fftshift applied to the index).
constexpr int kSegN = 1024, kSegWaves = 4, kSegPitch = 65;
// wavefronts per SIMD the kernel is held to, and whether segment s+1 is requested before s is transformed (32 VGPRs).
// Measured at 9765 segments: (3, prefetch) 29.1 / 27.0 us Welch / waterfall, (4, no prefetch) 28.5 / 28.6 us -- a tie;
// the kernel issues VALU instructions 71 % of the time either way.
constexpr int kSegOcc = 3;
constexpr bool kSegPrefetch = true;

// KIND: what leaves the registers -- 0 getWelch's accumulation, 1 getWaterfall's Float64 power spectra, 2 the spectra themselves
// (batched 1024-point row transforms, tsdr_fft_c2c: rows[seg * 1024 + k] = scale * X[k]; inverse by conjugating on the way in
// and out; may alias the input: a wavefront stores segment s after it has loaded s and s + 1, and segments belong to one wavefront)
enum { SEG_WELCH = 0, SEG_WATERFALL = 1, SEG_ROWS = 2 };
template <int KIND, bool CPLX>
__global__ __launch_bounds__(64 * kSegWaves, kSegOcc) void k_seg1024(const float *__restrict__ sig, size_t nbSeg,
                                                            unsigned nwaves, float *__restrict__ part,
                                                            double *__restrict__ wf, float2 *__restrict__ rows = nullptr,
                                                            unsigned smask = 0u, float scale = 1.0f) {
  constexpr bool WATERFALL = KIND == SEG_WATERFALL;
  __shared__ float2 lds[kSegWaves][16 * kSegPitch];
  __shared__ float2 tw2t[64];  // W_64^(l0 kb1) at [l0 * 16 + kb1]: four distinct addresses per wave-instruction
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float2 *z = lds[wave];
  if (threadIdx.x < 64) tw2t[threadIdx.x] = tw_unit((unsigned)((threadIdx.x >> 4) * (threadIdx.x & 15)), 6);
  __syncthreads();
  // wavefront w of nwaves takes the segments [w * nbSeg / nwaves, (w + 1) * nbSeg / nwaves): contiguous, sizes differ by <= 1
  const size_t wglobal = (size_t)blockIdx.x * kSegWaves + wave;
  const size_t seg0 = wglobal < nwaves ? wglobal * nbSeg / nwaves : nbSeg;
  const size_t seg1 = wglobal < nwaves ? (wglobal + 1) * nbSeg / nwaves : nbSeg;
  // per-lane twiddles W_1024^(l ka), the same for every segment
  float2 tw1[16];
#pragma unroll
  for (int k = 1; k < 16; ++k) tw1[k] = tw_unit((unsigned)(lane * k), 10);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  const int ka2 = lane & 15, l0 = lane >> 4;
  // the next segment's samples are requested before the current one is transformed: HBM latency under the arithmetic
  float2 nx[16];
  auto fetch = [&](size_t seg) {
    if (CPLX) {
      const float2 *x = reinterpret_cast<const float2 *>(sig) + seg * kSegN + lane;
#pragma unroll
      for (int m = 0; m < 16; ++m) nx[m] = x[64 * m];
    } else {
      const float *x = sig + seg * kSegN + lane;
#pragma unroll
      for (int m = 0; m < 16; ++m) nx[m] = make_float2(x[64 * m], 0.0f);
    }
  };
  if (kSegPrefetch && seg0 < seg1) fetch(seg0);
  for (size_t seg = seg0; seg < seg1; ++seg) {
    float2 v[16];
    if (!kSegPrefetch) fetch(seg);
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = KIND == SEG_ROWS ? conj_if(nx[m], smask) : nx[m];
    if (kSegPrefetch && seg + 1 < seg1) fetch(seg + 1);
    reg_dft<16>(v);
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
      float2 t = v[brev<16>(ka)];
      if (ka) t = cmul(t, tw1[ka]);
      z[ka * kSegPitch + lane] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int l1 = 0; l1 < 16; ++l1) v[l1] = z[ka2 * kSegPitch + l0 + 4 * l1];
    reg_dft<16>(v);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every lane has its Z1 values before Z2 overwrites the region
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kb1 = 0; kb1 < 16; ++kb1) {
      float2 t = v[brev<16>(kb1)];
      if (kb1) t = cmul(t, tw2t[l0 * 16 + kb1]);
      z[kb1 * 64 + l0 * 16 + ka2] = t;  // [kb1][l0][ka2]: the 16 lanes of a bank cycle write consecutive words
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float2 d[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = z[(4 * j + l0) * 64 + q * 16 + ka2];  // element (kb1 = 4j + lane/16, l0 = q, ka2 = lane%16)
      reg_dft<4>(d);
#pragma unroll
      for (int kb0 = 0; kb0 < 4; ++kb0) {
        const float2 X = d[brev<4>(kb0)];
        if (KIND == SEG_ROWS) {
          rows[seg * kSegN + (size_t)(lane + 64 * j + 256 * kb0)] = conj_if(make_float2(X.x * scale, X.y * scale), smask);
          continue;
        }
        const float p = X.x * X.x + X.y * X.y;  // abs2 in f32, as the reference's abs2.(::ComplexF32)
        if (WATERFALL) {
          const int k = lane + 64 * j + 256 * kb0;
          __builtin_nontemporal_store((double)p, &wf[seg * kSegN + ((k + kSegN / 2) & (kSegN - 1))]);   // fftshift: output index of frequency k; written once
        } else {
          acc[j * 4 + kb0] += p;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // Z2 fully read before the next segment's Z1 lands in the region
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (KIND == SEG_WELCH) {
    // the workgroup's wavefronts, in wavefront order, through LDS: one partial spectrum per workgroup
    __syncthreads();
    float *sp = reinterpret_cast<float *>(&lds[0][0]);  // [kSegWaves][1024] floats fit the kSegWaves x 8448-byte regions
#pragma unroll
    for (int i = 0; i < 16; ++i) sp[wave * kSegN + lane + 64 * (i >> 2) + 256 * (i & 3)] = acc[i];
    __syncthreads();
    for (int k = threadIdx.x; k < kSegN; k += 64 * kSegWaves) {
      float t = sp[k];
#pragma unroll
      for (int w = 1; w < kSegWaves; ++w) t += sp[w * kSegN + k];
      part[(size_t)blockIdx.x * kSegN + k] = t;
    }
  }
}

// S[k] = sum over workgroup partials (index order, blocks of 16), then fftshift and optional dB.
// grid = 1024/16 blocks of 256 threads: thread (kq, g) adds partials [16 g, 16 g + 16) of frequency 16*blockIdx.x + kq.
__global__ __launch_bounds__(256) void k_welch_finish(const float *__restrict__ part, unsigned nparts, int lin,
                                                      float *__restrict__ y) {
  __shared__ float sm[16][17];
  const int kq = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + kq;
  float tot = 0.0f;
  for (unsigned base = 0; base < nparts; base += 256) {
    float t = 0.0f;
    const unsigned p0 = base + 16 * g;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = p0 + i < nparts ? part[(size_t)(p0 + i) * kSegN + k] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += v[i];
    sm[g][kq] = t;
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) tot += sm[i][kq];
    }
    __syncthreads();
  }
  if (g == 0) {
    const int j = (k + kSegN / 2) & (kSegN - 1);
    y[j] = lin ? tot : 10.0f * log10f(tot);
  }
}

// ---- init_resampler kernels ------------------------------------------------------------------
// h[n] = ComplexF32(ifft(H0)[n]) * blackman(n): the reference's ifft runs on ComplexF32 data, the window is Float64
// (DSP.blackman: 0.42 - 0.5cos(2pi n/(N-1)) + 0.08cos(4pi n/(N-1))), so h is ComplexF64 with f32-rounded factors
__global__ __launch_bounds__(256) void k_window64(double2 *__restrict__ h, size_t N) {
  for (size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (size_t)gridDim.x * blockDim.x) {
    double w = 1.0;
    if (N > 1) {
      const double t = (double)n / (double)(N - 1);
      w = 0.42 - 0.5 * cos(2.0 * M_PI * t) + 0.08 * cos(4.0 * M_PI * t);
    }
    const double2 v = h[n];
    h[n] = make_double2((double)(float)v.x * w, (double)(float)v.y * w);
  }
}

// H[k] *= (-1)^k
__global__ __launch_bounds__(256) void k_altsign64(double2 *__restrict__ H, size_t N) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (size_t)gridDim.x * blockDim.x)
    if (k & 1) { double2 v = H[k]; H[k] = make_double2(-v.x, -v.y); }
}

// Hs[k] = (H[k] + conj H[(N-k) mod N]) / 2, k <= N/2
__global__ __launch_bounds__(256) void k_herm_half(const double2 *__restrict__ H, size_t N, double2 *__restrict__ Hs) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k <= N / 2; k += (size_t)gridDim.x * blockDim.x) {
    const double2 a = H[k], b = H[k ? N - k : 0];
    Hs[k] = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
  }
}

__global__ __launch_bounds__(256) void k_c64_to_c32(const double2 *__restrict__ a, size_t N, float2 *__restrict__ o) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (size_t)gridDim.x * blockDim.x)
    o[k] = make_float2((float)a[k].x, (float)a[k].y);
}

// ---- resampler! at its algorithmic cost (bufferSize even) ------------------------------------------------------------
// The zero-stuffed input's spectrum is the bufferSize-point spectrum X of the input repeated upCoeff times, and only
// real(ifft(.)) is kept (Resampler.jl:48-59), i.e. the inverse transform of the Hermitian part S[k] = X[k mod Nb] Hs[k].
// So: P = FFT_{Nb/2}(in viewed as Nb/2 complex pairs)  ->  this kernel  ->  z = IFFT_{N/2}(Zc), out = z viewed as N reals:
//   X[j]  = Xe + W_Nb^j Xo,  Xe = (P[j] + conj P[Nh-j]) / 2,  Xo = -i (P[j] - conj P[Nh-j]) / 2     (j <= Nh = Nb/2)
//   S[k]  = X[k mod Nb] * Hs[k]                 -- ComplexF32 times ComplexF64 in f64, as inFFT[n] * H[n] (:51-53)
//   A = (S[k] + conj S[M-k]) / 2,  B = conj(W_N^k) (S[k] - conj S[M-k]) / 2,   M = N/2
//   Zc[k] = gain (A + i B),  Zc[M-k] = gain (conj A + i conj B)
// One thread per pair (k, M-k), k <= M/2.
// {cos, sin}(2 pi e / N), e <= N / 2, as the product of the two table entries e = 1024 h + l (tsdr_resampler::tw): two loads
// out of L2 and four multiply-adds instead of a 100-instruction f64 sincospi -- of which the kernel needs three per output
// pair and which was all of its time (16 us of f64 VALU work for 52 MB of traffic)
__device__ inline double2 resamp_tw(const double2 *__restrict__ tw, size_t e) {
  const double2 b = tw[e & 1023u], a = tw[1024u + (e >> 10)];
  return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.y, b.x, a.x * b.y));
}

__device__ inline double2 resamp_X(const float2 *__restrict__ P, size_t Nh, size_t Nb, size_t j, const double2 *__restrict__ tw, unsigned up) {
  const bool upper = j > Nh;  // X[j] = conj X[Nb - j]
  const size_t jj = upper ? Nb - j : j;
  const float2 a = P[jj == Nh ? 0 : jj], b = P[jj == 0 || jj == Nh ? 0 : Nh - jj];
  const double ex = 0.5 * ((double)a.x + (double)b.x), ey = 0.5 * ((double)a.y - (double)b.y);
  const double dx = 0.5 * ((double)a.x - (double)b.x), dy = 0.5 * ((double)a.y + (double)b.y);  // (P[j] - conj P[Nh-j]) / 2
  const double ox = dy, oy = -dx;                                                                  // times -i
  const double2 w = resamp_tw(tw, jj * up);  // W_Nb^jj = W_N^(up jj) = w.x - i w.y
  const double xr = ex + (w.x * ox + w.y * oy), xi = ey + (w.x * oy - w.y * ox);
  return make_double2(xr, upper ? -xi : xi);
}

__global__ __launch_bounds__(256) void k_resamp_mid(const float2 *__restrict__ P, const double2 *__restrict__ Hs, size_t Nb, size_t N,
                                                    double gain, const double2 *__restrict__ tw, unsigned up, float2 *__restrict__ Zc) {
  const size_t Nh = Nb / 2, M = N / 2;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k <= M / 2; k += (size_t)gridDim.x * blockDim.x) {
    const size_t k2 = M - k;
    // (N < 2^30: tsdr_resampler_init)
    const double2 x1 = resamp_X(P, Nh, Nb, (unsigned)k % (unsigned)Nb, tw, up), x2 = resamp_X(P, Nh, Nb, (unsigned)k2 % (unsigned)Nb, tw, up);
    const double2 h1 = Hs[k], h2 = Hs[k2];
    // the products are rounded to ComplexF32 like the reference's in-place inFFT[n] = inFFT[n] * H[n]
    const double s1x = (double)(float)(x1.x * h1.x - x1.y * h1.y), s1y = (double)(float)(x1.x * h1.y + x1.y * h1.x);
    const double s2x = (double)(float)(x2.x * h2.x - x2.y * h2.y), s2y = (double)(float)(x2.x * h2.y + x2.y * h2.x);
    const double ax = 0.5 * (s1x + s2x), ay = 0.5 * (s1y - s2y);
    const double dx = 0.5 * (s1x - s2x), dy = 0.5 * (s1y + s2y);
    const double2 w = resamp_tw(tw, k);  // conj(W_N^k) = w.x + i w.y
    const double bx = w.x * dx - w.y * dy, by = w.x * dy + w.y * dx;
    // A + iB = (ax - by) + i (ay + bx);   conj A + i conj B = (ax + by) + i (-ay + bx)
    Zc[k] = make_float2((float)(gain * (ax - by)), (float)(gain * (ay + bx)));
    if (k2 < M && k2 != k) Zc[k2] = make_float2((float)(gain * (ax + by)), (float)(gain * (bx - ay)));
  }
}

// containerFFT[1:up:end] .= in  (zero elsewhere)
__global__ __launch_bounds__(256) void k_stuff(const float *__restrict__ in, size_t N, unsigned up, float2 *__restrict__ c) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (size_t)gridDim.x * blockDim.x) {
    const size_t q = k / up;
    c[k] = (k - q * up == 0) ? make_float2(in[q], 0.f) : make_float2(0.f, 0.f);
  }
}

// inFFT[n] = inFFT[n] * H[n]: ComplexF32 times ComplexF64, evaluated in f64 and stored back as ComplexF32 (:51-53)
__global__ __launch_bounds__(256) void k_cmul(float2 *__restrict__ a, const double2 *__restrict__ b, size_t N) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (size_t)gridDim.x * blockDim.x) {
    const float2 x = a[k];
    const double2 h = b[k];
    a[k] = make_float2((float)((double)x.x * h.x - (double)x.y * h.y), (float)((double)x.x * h.y + (double)x.y * h.x));
  }
}

// out[n] = 2*upCoeff*real(outFFT[n])
__global__ __launch_bounds__(256) void k_real_scale(const float2 *__restrict__ c, size_t N, float g, float *__restrict__ out) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (size_t)gridDim.x * blockDim.x)
    out[k] = g * c[k].x;
}

// ---- resampler! at sizeFFT = 4096 (the size production/test_resampler.jl times: 1024 samples, upCoeff 4) in ONE
// workgroup: zero-stuffed load, 4096-point transform, filter, inverse transform and 2*up*real(.) without leaving the
// CU -- four launches of a few microseconds each become one.
// 4096 = 16 x 16 x 16 over 256 threads x 16 values.  With n = t0 + 16 t1 + 256 m (t = t0 + 16 t1 the thread) and
// k = ka + 16 kb + 256 kc:
//   A  thread t:        DFT16 over m,  * W_4096^(t ka)          -> Z1[ka][t]
//   B  thread (ka, t0): DFT16 over t1, * W_256^(t0 kb)          -> Z2[ka][kb][t0]   (rows padded to 17: conflict-free)
//   C  thread (ka, kb): DFT16 over t0  -> X[ka + 16 kb + 256 kc] -> natural order in LDS -> thread t takes X[t + 256 m]
// so input and output have the same distribution and the routine runs twice (the inverse as conj . forward . conj).
__device__ inline void wg_fft4096(float2 (&v)[16], float2 *z, const float2 (&twA)[16], const float2 (&twB)[16], int tid) {
  reg_dft<16>(v);
#pragma unroll
  for (int ka = 0; ka < 16; ++ka) {
    float2 x = v[brev<16>(ka)];
    if (ka) x = cmul(x, twA[ka]);
    z[ka * 256 + tid] = x;
  }
  __syncthreads();
  const int hi = tid >> 4, lo = tid & 15;  // stage B: (ka, t0)
#pragma unroll
  for (int t1 = 0; t1 < 16; ++t1) v[t1] = z[hi * 256 + lo + 16 * t1];
  reg_dft<16>(v);
  __syncthreads();
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) {
    float2 x = v[brev<16>(kb)];
    if (kb) x = cmul(x, twB[kb]);
    z[(hi * 16 + kb) * 17 + lo] = x;
  }
  __syncthreads();
#pragma unroll
  for (int t0 = 0; t0 < 16; ++t0) v[t0] = z[tid * 17 + t0];  // stage C: thread = (ka, kb) = (tid >> 4, tid & 15)
  reg_dft<16>(v);
  __syncthreads();
#pragma unroll
  for (int kc = 0; kc < 16; ++kc) z[hi + 16 * lo + 256 * kc] = v[brev<16>(kc)];  // k = ka + 16 kb + 256 kc
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = z[tid + 256 * m];
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_resample4096(const float *__restrict__ in, unsigned up, const double2 *__restrict__ H,
                                                      const float2 *__restrict__ tw4096, float gain, float *__restrict__ out) {
  __shared__ float2 z[256 * 17];
  const int tid = threadIdx.x;
  float2 twA[16], twB[16];
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    twA[k] = tw4096[tid * k];               // W_4096^(t ka)
    twB[k] = tw4096[((tid & 15) * k) << 4];  // W_256^(t0 kb)
  }
  float2 v[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {  // zero-stuffed input: x[n] = in[n / up] when up divides n
    const unsigned n = (unsigned)tid + 256u * (unsigned)m, q = n / up;
    v[m] = q * up == n ? make_float2(in[q], 0.0f) : make_float2(0.0f, 0.0f);
  }
  wg_fft4096(v, z, twA, twB, tid);
#pragma unroll
  for (int m = 0; m < 16; ++m) {  // filter (ComplexF32 x ComplexF64 in f64, stored as ComplexF32), and conj for the inverse transform
    const double2 h = H[tid + 256 * m];
    const double xr = v[m].x, xi = v[m].y;
    v[m] = make_float2((float)(xr * h.x - xi * h.y), -(float)(xr * h.y + xi * h.x));
  }
  wg_fft4096(v, z, twA, twB, tid);
  const float inv = 1.0f / 4096.0f;
#pragma unroll
  for (int m = 0; m < 16; ++m) out[tid + 256 * m] = gain * (v[m].x * inv);  // real(conj(.)) = real(.): ifft scale, then 2*up
}

static int spectrum_d(tsdr_ctx *ctx, const float *sig, int is_complex, size_t N, int lin, float *y) {
  if (N == 0) return TSDR_OK;
  // (Round 5, built and dropped: ONE launch for short signals -- N = R1 * R2, workgroup k1 forms sum_n1 x[R2 n1 + n2] W_R1^(n1 k1)
  // by direct summation and runs one R2-point LDS transform: no second pass, no grid barrier.  Correct on every size tried, and
  // 41-56 us at N = 80 000 against the two passes' 14.9: the work per THREAD is N / 1024 terms whatever R1 is (grid = R1), and
  // 16 wavefronts of ~2600 instructions share one CU -- 20 us of issue before any latency.  NOTEBOOK.md.)
  float2 *X = (float2 *)ctx->scratch(WS_FFT_A, N * sizeof(float2));
  if (!X) return TSDR_ENOMEM;
  if (fft_passes(N) >= 2 && (reinterpret_cast<uintptr_t>(sig) & (is_complex ? 7u : 3u)) == 0) {
    // the passes alone: a real signal enters through the first pass's loader, abs2 / 10log10 and the fftshift leave
    // through the last pass's epilogue
    FftEpilogue epi;
    epi.kind = EPI_SPEC;
    epi.out = y;
    epi.cnt = N;
    epi.k0 = N / 2;
    epi.log_scale = !lin;
    const float2 *x = reinterpret_cast<const float2 *>(sig);
    const int sm = is_complex ? SRC_C2C : SRC_RE0;
    return is_pow2(N) ? fft_pow2(ctx, x, X, ilog2(N), 1, -1, 1.0f, sm, 0, 0, &epi) : fft_mixed(ctx, x, X, N, 1, -1, 1.0f, sm, 0, 0, &epi);
  }
  int rc = fft_any(ctx, sig, is_complex, X, N, 1, -1);
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "spectrum_out", k_spec_out, dim3(stream_grid(ctx, N)), dim3(256), 0, (const float2 *)X, N, lin, y);
  return TSDR_OK;
}

static int segments_fft(tsdr_ctx *ctx, const float *sig, int is_complex, size_t len, size_t sizeFFT, float2 **X,
                        size_t *nbSeg) {
  if (sizeFFT == 0) return set_err(ctx, TSDR_EINVAL, "sizeFFT must be positive");
  *nbSeg = len / sizeFFT;
  *X = (float2 *)ctx->scratch(WS_FFT_A, (*nbSeg ? *nbSeg : 1) * sizeFFT * sizeof(float2));
  if (!*X) return TSDR_ENOMEM;
  if (*nbSeg == 0) return TSDR_OK;
  return fft_any(ctx, sig, is_complex, *X, sizeFFT, *nbSeg, -1);
}

}  // namespace tsdr

using namespace tsdr;

extern "C" {

int tsdr_spectrum_d(tsdr_ctx *ctx, const float *sig, int is_complex, size_t N, int lin, float *y) {
  if (!ctx || (N && (!sig || !y))) return TSDR_EINVAL;
  return spectrum_d(ctx, sig, is_complex, N, lin, y);
}

int tsdr_spectrum(tsdr_ctx *ctx, const float *sig, int is_complex, size_t N, int lin, float *y) {
  return host_map(ctx, sig, N * (is_complex ? 8 : 4), y, N * 4,
                  [&](void *i, void *o) { return spectrum_d(ctx, (const float *)i, is_complex, N, lin, (float *)o); });
}

// wavefronts of the 1024-point fast path: one per segment while they all fit the device at once (three 4-wavefront
// workgroups per CU: 154 VGPRs), else exactly that many, each walking a contiguous chunk of segments -- a launch of
// 4/3 of the resident workgroups ran its last third alone on a mostly idle device (Welch 31 -> 24 us at 9765 segments)
static unsigned seg_waves(tsdr_ctx *ctx, size_t nbSeg, unsigned *blocks) {
  const size_t resident = (size_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * kSegOcc * kSegWaves;
  const unsigned nwaves = (unsigned)std::min<size_t>(nbSeg, resident);
  *blocks = (unsigned)ceil_div((size_t)nwaves, (size_t)kSegWaves);
  return nwaves;
}

}  // extern "C"
namespace tsdr {
// batched 1024-point row transforms on the wavefront-per-segment kernel (fft.hip:fft_any): 1e7 points in 30 us
int fft_rows1024(tsdr_ctx *ctx, const float2 *in, float2 *out, size_t batch, int dir, float scale) {
  if (batch == 0) return TSDR_OK;
  if (batch >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "fft: too many rows");
  unsigned blocks = 0;
  const unsigned nwaves = seg_waves(ctx, batch, &blocks);
  TSDR_LAUNCH(ctx, "fft_rows1024", (k_seg1024<SEG_ROWS, true>), dim3(blocks), dim3(64 * kSegWaves), 0, reinterpret_cast<const float *>(in), batch,
              nwaves, (float *)nullptr, (double *)nullptr, out, dir > 0 ? 0x80000000u : 0u, scale);
  return TSDR_OK;
}
}  // namespace tsdr
extern "C" {

int tsdr_welch_d(tsdr_ctx *ctx, const float *sig, int is_complex, size_t len, size_t sizeFFT, int lin, float *y) {
  if (!ctx || !y || (len && !sig)) return TSDR_EINVAL;
  if (sizeFFT == (size_t)kSegN && len / sizeFFT > 0 && len / sizeFFT < (size_t(1) << 31) &&
      (reinterpret_cast<uintptr_t>(sig) & (is_complex ? 7u : 3u)) == 0) {
    const size_t nbSeg = len / sizeFFT;
    unsigned blocks = 0;
    const unsigned nwaves = seg_waves(ctx, nbSeg, &blocks);
    float *part = (float *)ctx->scratch(WS_FFT_A, (size_t)blocks * kSegN * 4);
    if (!part) return TSDR_ENOMEM;
    if (is_complex) {
      TSDR_LAUNCH(ctx, "welch_seg1024", (k_seg1024<SEG_WELCH, true>), dim3(blocks), dim3(64 * kSegWaves), 0, sig, nbSeg, nwaves, part,
                  (double *)nullptr);
    } else {
      TSDR_LAUNCH(ctx, "welch_seg1024", (k_seg1024<SEG_WELCH, false>), dim3(blocks), dim3(64 * kSegWaves), 0, sig, nbSeg, nwaves, part,
                  (double *)nullptr);
    }
    TSDR_LAUNCH(ctx, "welch_finish", k_welch_finish, dim3(kSegN / 16), dim3(256), 0, (const float *)part, blocks, lin, y);
    return TSDR_OK;
  }
  if (sizeFFT >= 2 && sizeFFT <= 4096 && len / sizeFFT > 0 && (reinterpret_cast<uintptr_t>(sig) & (is_complex ? 7u : 3u)) == 0) {
    // any other 2^a 3^b 5^c segment length that fits one workgroup's LDS: the segment transforms stay on the chip too --
    // every workgroup adds abs2 of the spectra it forms and leaves one partial sum (round 4: the segment spectra of this
    // route used to go through HBM, 5.1x the algorithmic traffic)
    float *part = (float *)ctx->scratch(WS_FFT_C, (size_t)fft_rows_welch_parts(ctx) * sizeFFT * sizeof(float));
    if (!part) return TSDR_ENOMEM;
    unsigned nparts = 0;
    bool did = false;
    int rc = fft_rows_welch(ctx, sig, is_complex, sizeFFT, len / sizeFFT, part, &nparts, &did);
    if (rc) return rc;
    if (did) {
      TSDR_LAUNCH(ctx, "welch_sum", k_welch_sum, dim3((unsigned)ceil_div(sizeFFT, 16)), dim3(256), 0, (const float *)part, sizeFFT, nparts, lin, y);
      return TSDR_OK;
    }
  }
  float2 *X;
  size_t nbSeg;
  int rc = segments_fft(ctx, sig, is_complex, len, sizeFFT, &X, &nbSeg);
  if (rc) return rc;
  if (nbSeg == 0) {  // sum over no segments: zeros (-Inf dB), as the reference's zero-initialised accumulator gives
    TSDR_LAUNCH(ctx, "welch_sum", k_welch_sum, dim3((unsigned)ceil_div(sizeFFT, 16)), dim3(256), 0, (const float *)nullptr, sizeFFT, 0u, lin,
                y);
    return TSDR_OK;
  }
  const size_t chunk = std::max<size_t>(16, ceil_div(nbSeg, (size_t)256));
  const unsigned nparts = (unsigned)ceil_div(nbSeg, chunk);
  float *part = (float *)ctx->scratch(WS_FFT_C, (size_t)nparts * sizeFFT * sizeof(float));
  if (!part) return TSDR_ENOMEM;
  if (ceil_div(sizeFFT, 256) >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "welch: sizeFFT too large");
  TSDR_LAUNCH(ctx, "welch_part", k_welch_part, dim3((unsigned)ceil_div(sizeFFT, 256), nparts), dim3(256), 0, (const float2 *)X, sizeFFT, nbSeg,
              chunk, part);
  TSDR_LAUNCH(ctx, "welch_sum", k_welch_sum, dim3((unsigned)ceil_div(sizeFFT, 16)), dim3(256), 0, (const float *)part, sizeFFT, nparts, lin, y);
  return TSDR_OK;
}

int tsdr_welch(tsdr_ctx *ctx, const float *sig, int is_complex, size_t len, size_t sizeFFT, int lin, float *y) {
  return host_map(ctx, sig, len * (is_complex ? 8 : 4), y, sizeFFT * 4, [&](void *i, void *o) {
    return tsdr_welch_d(ctx, (const float *)i, is_complex, len, sizeFFT, lin, (float *)o);
  });
}

int tsdr_waterfall_d(tsdr_ctx *ctx, const float *sig, int is_complex, size_t len, size_t sizeFFT, double *sMatrix) {
  if (!ctx || (len && !sig)) return TSDR_EINVAL;
  if (sizeFFT == (size_t)kSegN && len / sizeFFT > 0 && len / sizeFFT < (size_t(1) << 31) && sMatrix &&
      (reinterpret_cast<uintptr_t>(sig) & (is_complex ? 7u : 3u)) == 0) {
    const size_t nbSeg = len / sizeFFT;
    unsigned blocks = 0;
    const unsigned nwaves = seg_waves(ctx, nbSeg, &blocks);
    if (is_complex) {
      TSDR_LAUNCH(ctx, "waterfall_seg1024", (k_seg1024<SEG_WATERFALL, true>), dim3(blocks), dim3(64 * kSegWaves), 0, sig, nbSeg, nwaves,
                  (float *)nullptr, sMatrix);
    } else {
      TSDR_LAUNCH(ctx, "waterfall_seg1024", (k_seg1024<SEG_WATERFALL, false>), dim3(blocks), dim3(64 * kSegWaves), 0, sig, nbSeg, nwaves,
                  (float *)nullptr, sMatrix);
    }
    return TSDR_OK;
  }
  if (sizeFFT && sMatrix && len / sizeFFT > 0 && (reinterpret_cast<uintptr_t>(sig) & (is_complex ? 7u : 3u)) == 0) {
    bool did = false;
    int rcw = fft_rows_waterfall(ctx, sig, is_complex, sizeFFT, len / sizeFFT, sMatrix, &did);
    if (rcw || did) return rcw;
  }
  float2 *X;
  size_t nbSeg;
  int rc = segments_fft(ctx, sig, is_complex, len, sizeFFT, &X, &nbSeg);
  if (rc) return rc;
  if (nbSeg == 0) return TSDR_OK;
  if (!sMatrix) return TSDR_EINVAL;
  TSDR_LAUNCH(ctx, "waterfall_out", k_waterfall, dim3(stream_grid(ctx, sizeFFT * nbSeg)), dim3(256), 0, (const float2 *)X,
              sizeFFT, nbSeg, sMatrix);
  return TSDR_OK;
}

int tsdr_waterfall(tsdr_ctx *ctx, const float *sig, int is_complex, size_t len, size_t sizeFFT, double *sMatrix) {
  if (sizeFFT == 0) return TSDR_EINVAL;
  const size_t nb = len / sizeFFT;
  return host_map(ctx, sig, len * (is_complex ? 8 : 4), sMatrix, nb * sizeFFT * 8, [&](void *i, void *o) {
    return tsdr_waterfall_d(ctx, (const float *)i, is_complex, len, sizeFFT, (double *)o);
  });
}

// ---- init_resampler / initLPF (Resampler.jl:26-99) ------------------------------------------------
// The phase theta[k] of Resampler.jl:88-90 as Julia evaluates it.  `2*pi*(0:sizeFFT-1)/sizeFFT` is a TwicePrecision range (ref 0,
// step (2pi_d, 0) divided by sizeFFT in twice precision, hi word truncated by ceil(log2(sizeFFT-1)) bits), and the complex
// scalar 1im*groupDelay times it is again a range -- StepRangeLen{ComplexF64} with the TwicePrecision product as its step
// (broadcast.jl's StepRangeLen{T} method; no low word from the complex mul12) -- so element k is
//   fl( fl(k*hi2) + fl(k*lo2) ),  hi2 = fl(g*hi_t + fl(g*lo_t)),  lo2 = (g*hi_t - hi2) + fl(g*lo_t),
// an ulp away from fl(g*fl(2pi*k/N)) on a quarter of the k.  The ulp matters where 6 divides sizeFFT (3 when upCoeff = 1):
// round.(exp(im*theta)) (:90) then has entries whose sine or cosine is 0.5 -/+ 1e-13.  (A first version used the long-double
// pi here: 0.9 % off the oracle on such sizes.)  Needs -ffp-contract=off (build.py) like everything on this path.
struct LpfPhase { double hi, lo; };
static LpfPhase lpf_phase_step(size_t N) {
  auto canon = [](double big, double little) { LpfPhase r; r.hi = big + little; r.lo = (big - r.hi) + little; return r; };
  const double two_pi = 6.283185307179586, y = (double)N, g = -((double)N - 1.0) / 2.0;
  const double q = two_pi / y;
  const double uh = q * y, ul = std::fma(q, y, -uh);
  const LpfPhase d = canon(q, (((two_pi - uh) - ul) + 0.0) / y);
  int nb = 0;
  if (N >= 2) nb = std::min(27, (int)std::ceil(std::log2((double)(N - 1))));
  unsigned long long bits;
  std::memcpy(&bits, &d.hi, 8);
  bits &= ~((1ull << nb) - 1ull);
  double hi_t;
  std::memcpy(&hi_t, &bits, 8);
  const double lo_t = (d.hi - hi_t) + d.lo;
  return canon(hi_t * g, lo_t * g);
}
static inline double lpf_phase(size_t k, const LpfPhase &st) {
  const double a = (double)k * st.hi, b = (double)k * st.lo;
  return a + b;
}

int tsdr_resampler_init(tsdr_ctx *ctx, size_t bufferSize, int upCoeff, tsdr_resampler **out) {
  if (!ctx || !out) return TSDR_EINVAL;
  *out = nullptr;
  if (bufferSize == 0 || upCoeff < 1) return set_err(ctx, TSDR_EINVAL, "init_resampler: bufferSize and upCoeff must be positive");
  const size_t N = bufferSize * (size_t)upCoeff;
  if (N >= (size_t(1) << 30)) return set_err(ctx, TSDR_EINVAL, "init_resampler: sizeFFT too large");
  tsdr_resampler *r = new tsdr_resampler();
  r->ctx = ctx; r->bufferSize = bufferSize; r->up = upCoeff; r->sizeFFT = N;
  double2 *scratch = nullptr;
  const bool half = (bufferSize % 2 == 0) && bufferSize >= 4;
  if (hipMalloc((void **)&r->H, N * sizeof(double2)) != hipSuccess || hipMalloc((void **)&r->work, N * sizeof(float2)) != hipSuccess ||
      hipMalloc((void **)&scratch, N * sizeof(double2)) != hipSuccess ||
      (half && hipMalloc((void **)&r->Hs, (N / 2 + 1) * sizeof(double2)) != hipSuccess)) {
    (void)hipFree(scratch);
    tsdr_resampler_free(r);
    return set_err(ctx, TSDR_ENOMEM, "init_resampler: allocation failed");
  }
  // H0 = round.(H .* exp(im*groupDelay*pulsation)) with H[1:bound]=1  (:85-91): entries are
  // integers in {-1,0,1}; evaluated on the host in f64 like the reference's broadcast.
  std::vector<double2> H0(N, make_double2(0.0, 0.0));
  const double bound_d = nearbyint((double)N / (double)upCoeff / 2.0);
  const size_t bound = bound_d < (double)N ? (size_t)bound_d : N;
  const LpfPhase st = lpf_phase_step(N);
  for (size_t k = 0; k < bound; ++k) {
    const double th = lpf_phase(k, st);
    H0[k] = make_double2(nearbyint(cos(th)), nearbyint(sin(th)));
  }
  hipError_t e = hipMemcpyAsync(r->H, H0.data(), N * sizeof(double2), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) { if (int w = tsdr::wait_stream(ctx, ctx->stream, "init_resampler upload")) return w; }   // (a stuck stream keeps the tables)
  if (e != hipSuccess) { (void)hipFree(scratch); tsdr_resampler_free(r); return hip_fail(ctx, e, "init_resampler upload"); }
  // h = ifft(H0) .* blackman  (ifft on ComplexF32 data in the reference, the window and everything after it in f64);
  // H = fft(h) .* (-1)^k -- on the device in f64 (fft64.hip)
  // (every launch is checked where it is made: a failure here must not surface as the sticky error of a later, unrelated
  // launch and leave a silently wrong filter behind)
  auto finish = [&]() -> int {
    int rc2 = fft64_d(ctx, r->H, scratch, N, +1);
    if (rc2) return rc2;
    TSDR_LAUNCH(ctx, "lpf_window64", k_window64, dim3(stream_grid(ctx, N)), dim3(256), 0, r->H, N);
    rc2 = fft64_d(ctx, r->H, scratch, N, -1);
    if (rc2) return rc2;
    TSDR_LAUNCH(ctx, "lpf_altsign64", k_altsign64, dim3(stream_grid(ctx, N)), dim3(256), 0, r->H, N);
    if (r->Hs) TSDR_LAUNCH(ctx, "lpf_herm_half", k_herm_half, dim3(stream_grid(ctx, N / 2 + 1)), dim3(256), 0, (const double2 *)r->H, N, r->Hs);
    { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
    return TSDR_OK;
  };
  const unsigned long long wt = ctx->wait_timeouts;
  int rc = finish();
  if (ctx->wait_timeouts != wt) return rc;   // (a stream that never completed keeps scratch and tables: hipFree would wait without a bound)
  (void)hipFree(scratch);
  if (rc) { tsdr_resampler_free(r); return rc; }
  if (r->Hs) {  // twiddle tables of k_resamp_mid (resamp_tw): evaluated in long double, rounded once
    const size_t nhi = N / 2048 + 2;
    std::vector<double2> tw(1024 + nhi);
    const long double w0 = 6.283185307179586476925286766559005768L / (long double)N;
    for (size_t e = 0; e < 1024; ++e) tw[e] = make_double2((double)cosl(w0 * (long double)e), (double)sinl(w0 * (long double)e));
    for (size_t h = 0; h < nhi; ++h) tw[1024 + h] = make_double2((double)cosl(w0 * (long double)(1024 * h)), (double)sinl(w0 * (long double)(1024 * h)));
    if (hipMalloc((void **)&r->tw, tw.size() * sizeof(double2)) != hipSuccess) { tsdr_resampler_free(r); return set_err(ctx, TSDR_ENOMEM, "init_resampler: allocation failed"); }
    e = hipMemcpy(r->tw, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { tsdr_resampler_free(r); return hip_fail(ctx, e, "init_resampler upload"); }
  }
  *out = r;
  return TSDR_OK;
}

int tsdr_resampler_run_d(tsdr_resampler *r, const float *in, size_t n_in, float *out) {
  if (!r || !in || !out) return TSDR_EINVAL;
  tsdr_ctx *ctx = r->ctx;
  if (n_in != r->bufferSize) return set_err(ctx, TSDR_EINVAL, "Size of input %zu should match size used during init %zu", n_in, r->bufferSize);
  const size_t N = r->sizeFFT;
  if (N == 4096) {  // one workgroup, one launch
    int rc = ensure_tw_small(ctx);
    if (rc) return rc;
    TSDR_LAUNCH(ctx, "resampler_4096", k_resample4096, dim3(1), dim3(256), 0, in, (unsigned)r->up, (const double2 *)r->H,
                (const float2 *)ctx->tw_small, (float)(2 * r->up), out);
    return TSDR_OK;
  }
  if (r->Hs && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 7u) == 0) {
    // the algorithmic cost (see k_resamp_mid): a forward transform of bufferSize/2 complex points on the input as it
    // lies, one pointwise kernel, an inverse transform of N/2 complex points written straight into `out`
    const size_t Nh = r->bufferSize / 2, M = N / 2;
    float2 *P = (float2 *)ctx->scratch(WS_FFT_A, Nh * sizeof(float2));
    if (!P) return TSDR_ENOMEM;
    int rc = fft_any(ctx, in, 1, P, Nh, 1, -1);
    if (rc) return rc;
    TSDR_LAUNCH(ctx, "resampler_mid", k_resamp_mid, dim3(stream_grid(ctx, M / 2 + 1)), dim3(256), 0, (const float2 *)P,
                (const double2 *)r->Hs, r->bufferSize, N, (double)(2 * r->up), (const double2 *)r->tw, (unsigned)r->up, r->work);
    return fft_any(ctx, reinterpret_cast<const float *>(r->work), 1, reinterpret_cast<float2 *>(out), M, 1, +1);
  }
  float2 *tmp = (float2 *)ctx->scratch(WS_FFT_A, N * sizeof(float2));
  if (!tmp) return TSDR_ENOMEM;
  const int passes = fft_passes(N);
  if (passes >= 2) {
    // two transforms and nothing else: the zero-stuffing is the forward transform's loader, the filter the inverse
    // transform's, 2*upCoeff*real(.) the epilogue of its last pass
    const bool p2 = is_pow2(N);
    const float2 *x = reinterpret_cast<const float2 *>(in);
    int rc = p2 ? fft_pow2(ctx, x, tmp, ilog2(N), 1, -1, 1.0f, SRC_STUFF, r->up, 0)
                : fft_mixed(ctx, x, tmp, N, 1, -1, 1.0f, SRC_STUFF, r->up, 0);
    if (rc) return rc;
    FftEpilogue epi;
    epi.kind = EPI_REAL;
    epi.out = out;
    epi.cnt = N;
    epi.gain = (float)(2 * r->up);
    const float inv = (float)(1.0 / (double)N);
    return p2 ? fft_pow2(ctx, tmp, r->work, ilog2(N), 1, +1, inv, SRC_MULH, 0, 0, &epi, reinterpret_cast<const float2 *>(r->H))
              : fft_mixed(ctx, tmp, r->work, N, 1, +1, inv, SRC_MULH, 0, 0, &epi, reinterpret_cast<const float2 *>(r->H));
  }
  TSDR_LAUNCH(ctx, "resampler_stuff", k_stuff, dim3(stream_grid(ctx, N)), dim3(256), 0, in, N, (unsigned)r->up, r->work);
  int rc = fft_any(ctx, reinterpret_cast<const float *>(r->work), 1, tmp, N, 1, -1);
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "resampler_filter", k_cmul, dim3(stream_grid(ctx, N)), dim3(256), 0, tmp, (const double2 *)r->H, N);
  rc = fft_any(ctx, reinterpret_cast<const float *>(tmp), 1, r->work, N, 1, +1);
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "resampler_out", k_real_scale, dim3(stream_grid(ctx, N)), dim3(256), 0, (const float2 *)r->work, N,
              (float)(2 * r->up), out);
  return TSDR_OK;
}

int tsdr_resampler_run(tsdr_resampler *r, const float *in, size_t n_in, float *out) {
  if (!r) return TSDR_EINVAL;
  if (n_in != r->bufferSize) return set_err(r->ctx, TSDR_EINVAL, "Size of input %zu should match size used during init %zu", n_in, r->bufferSize);
  return host_map(r->ctx, in, n_in * 4, out, r->sizeFFT * 4,
                  [&](void *i, void *o) { return tsdr_resampler_run_d(r, (const float *)i, n_in, (float *)o); });
}

int tsdr_resampler_lpf(tsdr_resampler *r, float *H_host) {
  if (!r || !H_host) return TSDR_EINVAL;
  tsdr_ctx *ctx = r->ctx;
  float2 *tmp = (float2 *)ctx->scratch(WS_FFT_A, r->sizeFFT * sizeof(float2));
  if (!tmp) return TSDR_ENOMEM;
  TSDR_LAUNCH(ctx, "lpf_c32", k_c64_to_c32, dim3(stream_grid(ctx, r->sizeFFT)), dim3(256), 0, (const double2 *)r->H, r->sizeFFT, tmp);
  TSDR_HIP(ctx, hipMemcpyAsync(H_host, tmp, r->sizeFFT * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  return TSDR_OK;
}

int tsdr_resampler_lpf64(tsdr_resampler *r, double *H_host) {
  if (!r || !H_host) return TSDR_EINVAL;
  tsdr_ctx *ctx = r->ctx;
  TSDR_HIP(ctx, hipMemcpyAsync(H_host, r->H, r->sizeFFT * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  return TSDR_OK;
}

int tsdr_fft_z2z(tsdr_ctx *ctx, const double *in, double *out, size_t n, int dir) {
  if (!ctx || (n && (!in || !out))) return TSDR_EINVAL;
  if (n == 0) return TSDR_OK;
  if (n >= (size_t(1) << 30)) return set_err(ctx, TSDR_EINVAL, "fft_z2z: length too large");
  double2 *d = nullptr, *sc = nullptr;
  if (hipMalloc((void **)&d, n * sizeof(double2)) != hipSuccess || hipMalloc((void **)&sc, n * sizeof(double2)) != hipSuccess) {
    (void)hipFree(d); (void)hipFree(sc);
    return set_err(ctx, TSDR_ENOMEM, "fft_z2z: allocation failed");
  }
  int rc = TSDR_OK;
  hipError_t e = hipMemcpyAsync(d, in, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) rc = fft64_d(ctx, d, sc, n, dir);
  if (e == hipSuccess && !rc) e = hipMemcpyAsync(out, d, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) { if (int w = tsdr::wait_stream(ctx, ctx->stream, "fft_z2z")) return rc ? rc : w; }   // (a stuck stream keeps its scratch)
  (void)hipFree(d); (void)hipFree(sc);
  if (!rc && e != hipSuccess) rc = hip_fail(ctx, e, "fft_z2z");
  return rc;
}

void tsdr_resampler_free(tsdr_resampler *r) {
  if (!r) return;
  if (r->ctx && tsdr::wait_stream(r->ctx, r->ctx->stream, "tsdr_resampler_free")) return;   // (a stuck stream keeps its tables)
  if (r->H) (void)hipFree(r->H);
  if (r->Hs) (void)hipFree(r->Hs);
  if (r->work) (void)hipFree(r->work);
  if (r->tw) (void)hipFree(r->tw);
  delete r;
}

}  // extern "C"
