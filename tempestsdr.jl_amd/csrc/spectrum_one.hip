// spectrum_one.hip -- getSpectrum (GetSpectrum.jl:21-30) of a short signal in ONE launch.
//
// y = 10log10.(abs2.(fftshift(fft(sig[1:N])))) at N = 80 000 (production/investigate_data.jl:44) is 640 KB of complex
// points: too many for one workgroup's LDS, so the pass engines need two launches -- and at this size a launch is its chain of
// latencies (load, register steps, store: 5-6 us), not its bytes; a grid-wide barrier between the passes costs more than the
// kernel boundary it would replace.  This kernel needs neither: N = R1 * R2, n = R2 n1 + n2, k = k1 + R1 k2,
//
//   X[k1 + R1 k2] = sum_{n2 < R2} W_R2^(n2 k2) * [ W_N^(n2 k1) * sum_{n1 < R1} x[R2 n1 + n2] W_R1^(n1 k1) ]
//
// Workgroup k1 (grid = R1) forms the bracket for every n2 by DIRECT summation over n1 -- R1 multiply-adds per point, N * R1
// in all (3.2 M at N = 80 000, R1 = 40: nothing), every load a coalesced run of the input, which all workgroups read out of
// L2 -- and then runs ONE R2-point transform in LDS (Stockham autosort, radices 5 / 4 / 3 / 2), whose outputs are the
// spectrum's residues k = k1 mod R1.  abs2, 10log10 and the fftshift are the store.  R1 may be any integer (the direct sum
// does not care): only R2 has to be 2^a 3^b 5^c.
#include <algorithm>

#include "common.h"

namespace tsdr {

namespace {

constexpr int kOneNT = 1024;       // threads per workgroup: one point of the direct summation each (R2 <= 1024)
constexpr int kOneR2Max = 1024;    // points of the LDS transform (two float2 buffers: 16 KiB)
constexpr int kOneR1Max = 512;     // direct-summation factor (work N * R1)

struct OneArgs {
  const float *sig;
  float *y;
  unsigned N, R1, R2;
  int is_complex, lin;
  unsigned nrad;
  unsigned char rad[12];   // radices of the R2-point transform, in stage order
};

__device__ inline float2 cmulf1(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// exp(-2 pi i m / n), m < n <= 2^24 (both exact in f32)
__device__ inline float2 unit_root(unsigned m, unsigned n) {
  float s, c;
  sincospif(-2.0f * ((float)m / (float)n), &s, &c);
  return make_float2(c, s);
}

template <int P>
__device__ inline void dft_small(float2 (&v)[5]) {
  if (P == 2) {
    const float2 a = v[0], b = v[1];
    v[0] = make_float2(a.x + b.x, a.y + b.y); v[1] = make_float2(a.x - b.x, a.y - b.y);
  } else if (P == 4) {
    const float2 a = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), b = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
    const float2 c = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
    v[0] = make_float2(a.x + c.x, a.y + c.y);
    v[2] = make_float2(a.x - c.x, a.y - c.y);
    v[1] = make_float2(b.x + d.y, b.y - d.x);   // b - i d
    v[3] = make_float2(b.x - d.y, b.y + d.x);   // b + i d
  } else if (P == 3) {
    const float h = 0.86602540378443865f;
    const float2 s = make_float2(v[1].x + v[2].x, v[1].y + v[2].y), d = make_float2(v[1].x - v[2].x, v[1].y - v[2].y);
    const float2 m = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
    v[0] = make_float2(v[0].x + s.x, v[0].y + s.y);
    v[1] = make_float2(m.x + h * d.y, m.y - h * d.x);   // m - i h d
    v[2] = make_float2(m.x - h * d.y, m.y + h * d.x);
  } else {  // 5
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f, s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const float2 a1 = make_float2(v[1].x + v[4].x, v[1].y + v[4].y), b1 = make_float2(v[1].x - v[4].x, v[1].y - v[4].y);
    const float2 a2 = make_float2(v[2].x + v[3].x, v[2].y + v[3].y), b2 = make_float2(v[2].x - v[3].x, v[2].y - v[3].y);
    const float2 x0 = v[0];
    v[0] = make_float2(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
    const float2 p1 = make_float2(x0.x + c1 * a1.x + c2 * a2.x, x0.y + c1 * a1.y + c2 * a2.y);
    const float2 p2 = make_float2(x0.x + c2 * a1.x + c1 * a2.x, x0.y + c2 * a1.y + c1 * a2.y);
    const float2 q1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const float2 q2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    v[1] = make_float2(p1.x + q1.y, p1.y - q1.x);   // p1 - i q1
    v[4] = make_float2(p1.x - q1.y, p1.y + q1.x);
    v[2] = make_float2(p2.x + q2.y, p2.y - q2.x);
    v[3] = make_float2(p2.x - q2.y, p2.y + q2.x);
  }
}

// one Stockham stage of radix P over the R2 points of A -> B (L = product of the radices already done)
template <int P>
__device__ inline void stockham_stage(const float2 *A, float2 *B, unsigned R2, unsigned L, int tid) {
  const unsigned M = R2 / P;
  for (unsigned j = (unsigned)tid; j < M; j += kOneNT) {
    const unsigned k = j % L, blk = j / L;
    float2 v[5];
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = A[j + (unsigned)q * M];
    if (k) {
      const float2 w1 = unit_root(k, L * P);
      float2 w = w1;
#pragma unroll
      for (int q = 1; q < P; ++q) {
        v[q] = cmulf1(v[q], w);
        if (q + 1 < P) w = cmulf1(w, w1);
      }
    }
    dft_small<P>(v);
    float2 *o = B + blk * L * P + k;
#pragma unroll
    for (int s = 0; s < P; ++s) o[(unsigned)s * L] = v[s];
  }
}

__global__ __launch_bounds__(kOneNT) void k_spec_one(OneArgs a) {
  extern __shared__ float2 sh2[];
  float2 *A = sh2, *B = sh2 + a.R2, *tw1 = sh2 + 2 * a.R2;   // [R2] [R2] [R1]
  const int tid = threadIdx.x;
  const unsigned k1 = blockIdx.x, R1 = a.R1, R2 = a.R2, N = a.N;
  for (unsigned e = (unsigned)tid; e < R1; e += kOneNT) tw1[e] = unit_root(e, R1);
  __syncthreads();
  // the bracket: direct summation over n1.  A thread owns one n2; its R1 loads (stride R2: one coalesced run per n1 across the
  // workgroup) are independent of one another and go out kOneLD at a time -- a loop that waits for each load before the next
  // is R1 round trips to L2 (first version: 45 us per call)
  constexpr int kOneLD = 40;
  for (unsigned n2 = (unsigned)tid; n2 < R2; n2 += kOneNT) {
    float2 acc = make_float2(0.f, 0.f);
    unsigned e = 0;
    for (unsigned nb = 0; nb < R1; nb += kOneLD) {
      float2 xv[kOneLD];
      if (a.is_complex) {
        const float2 *x = reinterpret_cast<const float2 *>(a.sig) + n2;
#pragma unroll
        for (int u = 0; u < kOneLD; ++u) xv[u] = nb + u < R1 ? x[(size_t)R2 * (nb + u)] : make_float2(0.f, 0.f);
      } else {
        const float *x = a.sig + n2;
#pragma unroll
        for (int u = 0; u < kOneLD; ++u) xv[u] = make_float2(nb + u < R1 ? x[(size_t)R2 * (nb + u)] : 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < kOneLD; ++u) {
        const float2 w = tw1[e];
        acc.x = fmaf(xv[u].x, w.x, fmaf(-xv[u].y, w.y, acc.x));
        acc.y = fmaf(xv[u].x, w.y, fmaf(xv[u].y, w.x, acc.y));
        e += k1; if (e >= R1) e -= R1;
      }
    }
    A[n2] = cmulf1(acc, unit_root(n2 * k1, N));   // W_N^(n2 k1): n2 k1 < R2 R1 = N
  }
  __syncthreads();
  unsigned L = 1;
  for (unsigned st = 0; st < a.nrad; ++st) {
    const unsigned p = a.rad[st];
    if (p == 5) stockham_stage<5>(A, B, R2, L, tid);
    else if (p == 4) stockham_stage<4>(A, B, R2, L, tid);
    else if (p == 3) stockham_stage<3>(A, B, R2, L, tid);
    else stockham_stage<2>(A, B, R2, L, tid);
    __syncthreads();
    float2 *t = A; A = B; B = t;
    L *= p;
  }
  // X[k1 + R1 k2] -> fftshift position (k + N div 2) mod N (GetSpectrum.jl:27); abs2 / 10log10 as the pass engines' epilogue
  const unsigned half = N / 2;
  for (unsigned k2 = (unsigned)tid; k2 < R2; k2 += kOneNT) {
    const float2 X = A[k2];
    const float p = X.x * X.x + X.y * X.y;
    unsigned pos = k1 + R1 * k2 + half;
    if (pos >= N) pos -= N;
    a.y[pos] = a.lin ? p : 10.0f * log10f(p);
  }
}

}  // namespace

// *did = true when the one-launch route took the call (N = R1 * R2 with a 2-3-5-smooth R2 in [64, 1024] and R1 <= 512)
int spectrum_one_d(tsdr_ctx *ctx, const float *sig, int is_complex, size_t N, int lin, float *y, bool *did) {
  *did = false;
  if (N < 4096 || N > (size_t)kOneR1Max * kOneR2Max) return TSDR_OK;
  if ((reinterpret_cast<uintptr_t>(sig) & (is_complex ? 7u : 3u)) != 0) return TSDR_OK;
  // the largest smooth divisor R2 <= 2048 (the direct factor R1 = N / R2 then is the smallest possible)
  unsigned best = 0;
  for (unsigned r2 = kOneR2Max; r2 >= 64; --r2) {
    if (N % r2) continue;
    unsigned v = r2;
    for (unsigned f : {2u, 3u, 5u}) while (v % f == 0) v /= f;
    if (v == 1) { best = r2; break; }
  }
  if (!best || N / best > (size_t)kOneR1Max) return TSDR_OK;
  OneArgs a{};
  a.sig = sig; a.y = y; a.N = (unsigned)N; a.R2 = best; a.R1 = (unsigned)(N / best);
  a.is_complex = is_complex; a.lin = lin;
  unsigned v = best;
  while (v % 5 == 0) { a.rad[a.nrad++] = 5; v /= 5; }
  while (v % 3 == 0) { a.rad[a.nrad++] = 3; v /= 3; }
  while (v % 4 == 0) { a.rad[a.nrad++] = 4; v /= 4; }
  if (v == 2) a.rad[a.nrad++] = 2;
  const size_t lds = ((size_t)2 * a.R2 + a.R1) * sizeof(float2);
  TSDR_LAUNCH(ctx, "spectrum_one", k_spec_one, dim3(a.R1), dim3(kOneNT), lds, a);
  *did = true;
  return TSDR_OK;
}

}  // namespace tsdr
