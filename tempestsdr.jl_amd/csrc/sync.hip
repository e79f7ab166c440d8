// sync.hip -- FrameSynchronisation.jl on gfx950, plus the per-frame post-ops of the GUI loop.
//
//   vsync(image,sync)            FrameSynchronisation.jl:56-79
//   fill_beta!                   :94-112      averagePixel :84-90     modIndex :120-122
//   init_gaussian_filter(5)      :124-129
//   circshift + IIR              GUI.jl:172,175
//
// The arithmetic follows the oracle's evaluation ORDER, not only its formulas, so that for the same image
// the projections, beta values and therefore the argmax indices are bit-identical (oracle: "summation orders"):
//   * row sums  sum(image;dims=2): STRICTLY left to right from 0.0f -- the order Base's _mapreducedim! fixes when
//     the first dimension is kept.  One lane owns a row; the eight wavefronts of a workgroup take turns along the
//     columns and hand the running sums on through LDS, so the loads of all eight run in parallel while the adds
//     stay one chain (k_proj);
//   * column sums  sum(image;dims=1): Julia leaves the order to @simd; fixed as 64-row blocks, each accumulated in
//     order, block sums added top to bottom;  Sigma = sum(c_v) likewise open, fixed in sum64 order (lane m
//     accumulates elements m, m+64, ..., then the xor-butterfly 32,16,..,1 = the oracle's tree64);
//   * the 5-tap causal FIR is DSP.jl's transposed-direct-form muladd chain,
//       y[i] = fma(x[i],h0, fma(x[i-1],h1, fma(x[i-2],h2, fma(x[i-3],h3, h4*x[i-4]))))   (x[<0] = 0);
//   * each centre's running blank sum _Sigma is the reference's sequential recurrence over w.
// Launches per buffer: k_proj (one pass over every image) or, in TSDR_FAST mode, the partial sums the raster
// kernel forms on the fly; k_fold (partials -> raw projections -> FIR -> Sigma, once per frame and axis); k_beta
// (beta scan + argmax).  In k_beta LPC lanes share one blank-band centre: each replays the cheap running-sum
// prefix (adds only, same order => same bits) and evaluates its share of the widths, so the 2*(w_min+W) serial
// adds of a centre are the only serial part and the ~20 operations per (centre, w) run LPC-wide.  The two
// divisions per width are by small integers: a table of correctly rounded reciprocals and Markstein's two-FMA
// correction return the correctly rounded quotient (bit-identical to IEEE division) at a third of the
// instruction count.
// The argmax over (w,c) is a lane-local scan, a 64-wide shuffle reduction and one 64-bit atomicMax per
// wavefront on a packed key (beta bits << 32 | ~c): beta >= +0 so its bit pattern is order-preserving, NaN
// patterns sort above +Inf (Julia's findmax treats NaN as maximal), and ~c makes the smallest column win
// ties = first maximum in column-major order (only the column of the maximum is ever used, :66,:76).
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include <hip/hip_ext.h>

#include "common.h"
#include "down_fused.h"
#include "guard.h"
#include "sync_layout.h"

struct tsdr_sync {
  tsdr_ctx *ctx;
  int y_t, x_t;
  int wmin_y, wmax_y, wmin_x, wmax_x;
  float h[5];
  float *beta_x = nullptr;  // device, (1+wmax_x-wmin_x) x x_t   (the current set: one of bset[])
  float *beta_y = nullptr;  // device, (1+wmax_y-wmin_y) x y_t
  float *bset[4][2] = {};   // [pipeline lane][x / y]: sync_use_lane
  int *pending = nullptr;   // device: [cur] = s_y the next vsync call will return (argmax of beta_y); double-buffered
  int cur = 0;
};

namespace tsdr {

struct SyncGeom {
  int y_t, x_t, wmin_y, wmax_y, wmin_x, wmax_x;
  float h0, h1, h2, h3, h4;
};

__device__ inline float wave_tree64(float v) {  // oracle tree64: v[i] += v[i+off], off = 32..1 ; result in lane 0
  for (int off = 32; off > 0; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- projections: ONE pass over the image ---------------------------------------------------------------
#ifndef TSDR_ROWSUM_CHUNK8
// Workgroup = (frame, 64-row block), 8 wavefronts; lanes are rows.  Wavefront j owns the CH columns [CH*j, CH*(j+1)) of
// a super-round of 8*CH columns (one super-round covers an 800-column image): it loads them all (CH coalesced 256-byte
// loads per lane in flight -- the whole 64 x 800 slab of the workgroup is requested at once), passes them SB at a time
// through its own LDS tile so that lanes-as-columns can fold the 64 rows in order (-> colpart[block][column]), and then
// the eight wavefronts add their CH values to the row's running sum one after the other, wavefront 0 first: the row
// sum is ONE left-to-right chain (-> rowpart[0][row], already final), but only the adds are serial -- 8 hand-overs per
// super-round.  grid = (nrb, frames).
// (launch bound: 128 registers, so that two workgroups share a CU and the 300 of a C2 buffer are resident at once --
// at 136 registers they ran in two rounds)
constexpr size_t kProjLds = (size_t)(8 * 64 * 26 + 64) * sizeof(float);  // 8 tiles of 64 x 26 floats + the row chain: 52 KiB
// body of one k_proj workgroup (512 threads): frame f, 64-row block rb; uses the first kProjLds bytes of the launch's
// dynamic LDS (declared here, not passed in: through a pointer parameter the compiler lost the address space, issued flat
// accesses and moved the 100-value register array to scratch -- k_proj 27 -> 52 us)
__device__ __forceinline__ void proj_wg(const float *__restrict__ img, size_t img_stride, int y_t, int x_t, float *__restrict__ proj,
                               size_t proj_stride, unsigned long long *__restrict__ keys, int f, int rb) {
  constexpr int CH = 100, SB = 25, PITCH = SB + 1, RND = 8 * CH;
  extern __shared__ float lds[];
  float *chain = lds + 8 * 64 * PITCH;
  const int nrb = (y_t + 63) >> 6;
  const float *im = img + (size_t)f * img_stride;
  float *pr = proj + (size_t)f * proj_stride;
  if (rb == 0 && threadIdx.x < 2) keys[(size_t)f * 2 + threadIdx.x] = 0ull;  // this frame's argmax keys (k_beta's atomicMax)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = rb * 64 + lane;
  const bool rv = r < y_t;
  const int nval = min(64, y_t - rb * 64);
  const float *p = im + (rv ? r : 0);
  float *mytile = lds + wave * (64 * PITCH);
  const int rounds = (x_t + RND - 1) / RND;
  for (int round = 0; round < rounds; ++round) {
    const int cs = round * RND + wave * CH;
    const int nc = max(0, min(CH, x_t - cs));
    float v[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) v[u] = (rv && u < nc) ? p[(size_t)(cs + u) * y_t] : 0.0f;
    // column sums of this wavefront's chunk, SB columns at a time (the tile is private to the wavefront: LDS
    // operations of one wavefront complete in order, the workgroup barrier is not needed)
#pragma unroll
    for (int sb = 0; sb < CH / SB; ++sb) {
      if (sb * SB < nc) {
#pragma unroll
        for (int u = 0; u < SB; ++u) mytile[lane * PITCH + u] = v[sb * SB + u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < min(SB, nc - sb * SB)) {  // lanes become columns: the rows of the block in row order
          float t = 0.0f;
          const float *col = mytile + lane;
          int rr = 0;
          for (; rr + 8 <= nval; rr += 8) {  // reads first, then the adds
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = col[(rr + u) * PITCH];
#pragma unroll
            for (int u = 0; u < 8; ++u) t = __fadd_rn(t, w[u]);
          }
          for (; rr < nval; ++rr) t = __fadd_rn(t, col[rr * PITCH]);
          pr[(size_t)rb * x_t + cs + sb * SB + lane] = t;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    for (int k = 0; k < 8; ++k) {  // the row chain: wavefront k's turn
      if (wave == k && nc > 0) {
        float a = (round == 0 && k == 0) ? 0.0f : chain[lane];
        if (nc == CH) {
#pragma unroll
          for (int u = 0; u < CH; ++u) a = __fadd_rn(a, v[u]);
        } else {
#pragma unroll
          for (int u = 0; u < CH; ++u) if (u < nc) a = __fadd_rn(a, v[u]);
        }
        chain[lane] = a;
      }
      __syncthreads();
    }
  }
  if (wave == 0 && rv) pr[(size_t)nrb * x_t + r] = chain[lane];
}
__global__ __launch_bounds__(512, 4) void k_proj(const float *__restrict__ img, size_t img_stride, int y_t, int x_t,
                                                 float *__restrict__ proj, size_t proj_stride,
                                                 unsigned long long *__restrict__ keys) {
  proj_wg(img, img_stride, y_t, x_t, proj, proj_stride, keys, (int)blockIdx.y, (int)blockIdx.x);
}
constexpr int kProjRowParts = 1;
static inline dim3 proj_grid(int y_t, int frames) { return dim3((unsigned)((y_t + 63) >> 6), (unsigned)frames); }
static inline dim3 proj_block() { return dim3(512); }
static inline size_t proj_lds_bytes() { return kProjLds; }
#else
// Round-1 order (compile-time alternative, see the oracle's ORC_ROWSUM_CHUNK8): row sums in 8 column chunks, each
// accumulated in order from 0.0f, the chunk sums added left to right by k_fold.  One wavefront per (64-row block,
// column chunk); grid = (8 * nrb, frames).
__global__ __launch_bounds__(64) void k_proj(const float *__restrict__ img, size_t img_stride, int y_t, int x_t,
                                             float *__restrict__ proj, size_t proj_stride,
                                             unsigned long long *__restrict__ keys) {
  constexpr int SUB = 32, PITCH = SUB + 1;
  __shared__ float tile[64 * PITCH];
  const int f = blockIdx.y;
  const int nrb = (y_t + 63) >> 6, chunk = (x_t + 7) >> 3;
  const int j = blockIdx.x & 7, rb = blockIdx.x >> 3;
  const float *im = img + (size_t)f * img_stride;
  float *pr = proj + (size_t)f * proj_stride;
  const int lane = threadIdx.x;
  if (blockIdx.x == 0 && lane < 2) keys[(size_t)f * 2 + lane] = 0ull;
  const int r = rb * 64 + lane;
  const bool rv = r < y_t;
  const int c0 = j * chunk, c1 = min(c0 + chunk, x_t);
  const int nval = min(64, y_t - rb * 64);
  const float *p = im + (rv ? r : 0) + (size_t)c0 * y_t;
  float a = 0.0f;
  for (int cs = c0; cs < c1; cs += SUB) {
    const int nc = min(SUB, c1 - cs);
    for (int u = 0; u < nc; ++u) { const float v = p[(size_t)u * y_t]; a = __fadd_rn(a, v); tile[lane * PITCH + u] = v; }
    p += (size_t)SUB * y_t;
    __syncthreads();
    if (lane < nc) {
      float t = 0.0f;
      const float *col = tile + lane;
      for (int rr = 0; rr < nval; ++rr) t = __fadd_rn(t, col[rr * PITCH]);
      pr[(size_t)rb * x_t + cs + lane] = t;
    }
    __syncthreads();
  }
  if (rv) pr[(size_t)nrb * x_t + (size_t)j * y_t + r] = a;
}
constexpr int kProjRowParts = 8;
// (this measurement-only build has no guard kernel: FAST-mode frame loops run in TSDR_EXACT instead)
static inline dim3 proj_grid(int y_t, int frames) { return dim3((unsigned)(8 * ((y_t + 63) >> 6)), (unsigned)frames); }
static inline dim3 proj_block() { return dim3(64); }
static inline size_t proj_lds_bytes() { return 0; }
#endif

__device__ inline unsigned long long pack_key(float v, int c) {
  unsigned bits = (v != v) ? 0x7FC00000u : __float_as_uint(v);
  return ((unsigned long long)bits << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)c);
}

// x / d for a small positive integer d, given rd = RN(1/d): Markstein's correction returns RN(x/d) exactly
// (q = RN(x*rd) is within 1 ulp, the residual x - q*d is exact in one FMA, RN(q + res*rd) is the correctly
// rounded quotient).  Outside the safely normal range the IEEE division runs instead.
__device__ inline float div_small(float x, float d, float rd) {
  const float ax = fabsf(x);
  if (!(ax > 1e-30f && ax < 1e30f)) return __fdiv_rn(x, d);
  const float q = __fmul_rn(x, rd);
  const float res = __fmaf_rn(-q, d, x);
  return __fmaf_rn(res, rd, q);
}

// DSP.filt's short-FIR chain at output i (x[<0] = 0): see the file header / oracle fir_filt
__device__ inline float fir5(const float *raw, int i, float h0, float h1, float h2, float h3, float h4) {
  const float x4 = i >= 4 ? raw[i - 4] : 0.0f, x3 = i >= 3 ? raw[i - 3] : 0.0f, x2 = i >= 2 ? raw[i - 2] : 0.0f,
              x1 = i >= 1 ? raw[i - 1] : 0.0f, x0 = raw[i];
#ifdef TSDR_FIR_NOFMA
  float acc = __fmul_rn(h4, x4);
  acc = __fadd_rn(acc, __fmul_rn(h3, x3));
  acc = __fadd_rn(acc, __fmul_rn(h2, x2));
  acc = __fadd_rn(acc, __fmul_rn(h1, x1));
  return __fadd_rn(acc, __fmul_rn(h0, x0));
#else
  float acc = __fmul_rn(h4, x4);
  acc = __fmaf_rn(x3, h3, acc);
  acc = __fmaf_rn(x2, h2, acc);
  acc = __fmaf_rn(x1, h1, acc);
  return __fmaf_rn(x0, h0, acc);
#endif
}

// beta of one centre over widths [wa, wb] after replaying the running sum up to wa-1; returns the first
// maximum of that range (NaN maximal); writes beta values when bout != nullptr.  rtab == nullptr: IEEE division.
__device__ inline float beta_scan(const float *cv, int n, int c0, int w_min, int wa, int wb, float S, float *bout,
                                  const float2 *rtab) {
  float acc = 0.0f;
  int k = (c0 - (w_min - 1)) % n; if (k < 0) k += n;
  for (int t = 0; t < 2 * (w_min - 1) + 1; ++t) { acc = __fadd_rn(acc, cv[k]); if (++k == n) k = 0; }
  float s = __fmul_rn(2.0f, acc);
  int lo = (c0 - w_min) % n; if (lo < 0) lo += n;
  int hi = (c0 + w_min) % n;
  for (int w = w_min; w < wa; ++w) {  // prefix replay: adds only
    s = __fadd_rn(s, __fmul_rn(2.0f, cv[lo]));
    s = __fadd_rn(s, __fmul_rn(2.0f, cv[hi]));
    if (--lo < 0) lo = n - 1;
    if (++hi == n) hi = 0;
  }
  float bv = 0.0f;
  bool have = false;
  for (int w = wa; w <= wb; ++w) {
    s = __fadd_rn(s, __fmul_rn(2.0f, cv[lo]));
    s = __fadd_rn(s, __fmul_rn(2.0f, cv[hi]));
    const float d1 = (float)(2 * (n - w)), d2 = (float)(2 * w);
    float v;
    if (rtab) {
      const float2 rr = rtab[w - w_min];
      v = __fadd_rn(div_small(__fsub_rn(S, s), d1, rr.x), div_small(s, d2, rr.y));
    } else {
      v = __fadd_rn(__fdiv_rn(__fsub_rn(S, s), d1), __fdiv_rn(s, d2));
    }
    v = __fmul_rn(v, v);
    if (bout) bout[w - w_min] = v;
    if (!have) { bv = v; have = true; }
    else if (!(bv != bv) && (v != v || v > bv)) bv = v;
    if (--lo < 0) lo = n - 1;
    if (++hi == n) hi = 0;
  }
  return bv;
}

// ---- fold + FIR + Sigma + beta scan + argmax.  Workgroup = 64 blank-band centres of one (frame, axis) x NWV
// wavefronts; lane = centre, wavefront q = the q-th share of the widths, so the width -- and with it both divisors and
// their reciprocals -- is uniform over a wavefront and every LDS read is 64 consecutive words.
// Prologue (every workgroup of an axis repeats it; the inputs are a few KB out of L2): the projection partial sums
//   proj per frame: colpart[ncp][x_t] | rowpart[nrp][y_t]
// are added per element in index order starting from the first (the order the producer defines: k_proj's row blocks
// top to bottom, the raster kernel's tiles in tile order), filtered (fir5) and summed (Sigma, sum64 order).
// The centres' circular neighbourhood is then unwrapped into a linear LDS window: a lane's walk is plain descending /
// ascending addresses.  Wavefront q replays the running sum _Sigma(w) (FrameSynchronisation.jl:101-107) up to the
// first width of its share -- fma(2, c_v, s) is the reference's s + 2*c_v exactly, doubling being exact -- then
// evaluates beta over its share.  The replay is adds only and every LDS read is issued eight widths ahead of its
// use, so the ~24 operations per (centre, width) run NWV-wide while the serial part of a centre stays the
// 2*(w_min + W) adds the reference has.
// The two divisions per width are by small integers: RN(1/d) (one IEEE division per width and wavefront, kept in a
// lane and read back with v_readlane) and Markstein's two-FMA correction give the correctly rounded quotient.  The
// correction needs |x| comfortably normal: min/max of |x| are tracked and a lane that ever left [1e-30, 1e30] (or
// met an Inf) redoes its share with IEEE divisions.
// grid.x = ceil(x_t/64) + ceil(y_t/64), grid.y = frames.  write_frame: frame whose beta matrices are stored.
// The frame's two argmax keys must be zero on entry (the projection producer clears them).
struct BetaArgs {
  const float *proj;
  size_t proj_stride;
  int ncp, nrp;
  SyncGeom g;
  unsigned long long *keys;
  int write_frame;
  float *bx, *by;
  uint2 *top2;       // sync guard (or null): per (frame, workgroup of the frame) {largest column maximum, second largest among
                     // OTHER columns} of the workgroup's 64 centres, as order-preserving words
};

__device__ inline int gridDim_x_of_frame(const SyncGeom &g) { return ((g.x_t + 63) >> 6) + ((g.y_t + 63) >> 6); }

// body of one k_beta workgroup: blk = block index within the frame (x-axis blocks first), f = frame
template <int NWV>
__device__ __forceinline__ void beta_wg(const BetaArgs &A, int blk, int f, float *sh) {
  const float *__restrict__ proj = A.proj;
  const size_t proj_stride = A.proj_stride;
  const int ncp = A.ncp, nrp = A.nrp;
  const SyncGeom &g = A.g;
  unsigned long long *__restrict__ keys = A.keys;
  const int write_frame = A.write_frame;
  float *__restrict__ bx = A.bx, *__restrict__ by = A.by;
  __shared__ unsigned colk[NWV][64];
  __shared__ float Ssh;
  const int nbx = (g.x_t + 63) >> 6;
  const int axis = blk < nbx ? 0 : 1;
  const int n = axis == 0 ? g.x_t : g.y_t;
  const int w_min = axis == 0 ? g.wmin_x : g.wmin_y, w_max = axis == 0 ? g.wmax_x : g.wmax_y;
  const int W = w_max - w_min + 1;
  const int NU = 64 + 2 * w_max;
  float *raw = sh, *cv = sh + n, *cu = sh + 2 * n;  // [n] raw projection, [n] filtered, [NU] cv[(cbase - w_max + j) mod n]
  const int tid = threadIdx.x;
  const int cbase = (blk - (axis == 0 ? 0 : nbx)) * 64;
  {
    const int cnt = axis == 0 ? ncp : nrp;
    const float *pr = proj + (size_t)f * proj_stride + (axis == 0 ? 0 : (size_t)ncp * g.x_t);
    for (int i0 = tid; i0 < n; i0 += 2 * 64 * NWV) {  // two elements per thread and trip: their loads overlap
      const int i1 = i0 + 64 * NWV;
      const bool two = i1 < n;
      const float *qa = pr + i0, *qb = pr + (two ? i1 : i0);
      float ta = qa[0], tb = qb[0];
      int j = 1;
      for (; j + 8 <= cnt; j += 8) {  // loads batched, adds in order
        float va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { va[u] = qa[(size_t)(j + u) * n]; vb[u] = qb[(size_t)(j + u) * n]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { ta = __fadd_rn(ta, va[u]); tb = __fadd_rn(tb, vb[u]); }
      }
      for (; j < cnt; ++j) { ta = __fadd_rn(ta, qa[(size_t)j * n]); tb = __fadd_rn(tb, qb[(size_t)j * n]); }
      raw[i0] = ta;
      if (two) raw[i1] = tb;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 64 * NWV) cv[i] = fir5(raw, i, g.h0, g.h1, g.h2, g.h3, g.h4);
  __syncthreads();
  if (tid < 64) {  // Sigma in sum64 order
    float a = 0.0f;
    for (int i = tid; i < n; i += 64) a = __fadd_rn(a, cv[i]);
    a = wave_tree64(a);
    if (tid == 0) Ssh = a;
  } else {
    int k = (cbase - w_max) % n; if (k < 0) k += n;
    k += tid - 64;
    for (int j = tid - 64; j < NU; j += 64 * (NWV - 1)) {
      while (k >= n) k -= n;
      cu[j] = cv[k];
      k += 64 * (NWV - 1);
    }
  }
  __syncthreads();
  const float S = Ssh;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int c0 = cbase + lane;
  const int Wq = (W + NWV - 1) / NWV;
  const int ia = min(q * Wq, W), ib = min(ia + Wq, W);  // this wavefront's widths: w_min + [ia, ib)
  const float *ctr = cu + w_max + lane;
  float acc = 0.0f;
  {
    const float *pk = ctr - (w_min - 1);
    const int np = 2 * (w_min - 1) + 1;
    int t = 0;
    for (; t + 8 <= np; t += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pk[t + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __fadd_rn(acc, v[u]);
    }
    for (; t < np; ++t) acc = __fadd_rn(acc, pk[t]);
  }
  float s = __fmul_rn(2.0f, acc);
  const float *plo = ctr - w_min, *phi = ctr + w_min;
  int i = 0;
  for (; i + 8 <= ia; i += 8) {  // replay of the widths before this wavefront's share
    float lo[8], hi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { lo[u] = plo[-(i + u)]; hi[u] = phi[i + u]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s = __fmaf_rn(2.0f, lo[u], s); s = __fmaf_rn(2.0f, hi[u], s); }
  }
  for (; i < ia; ++i) { s = __fmaf_rn(2.0f, plo[-i], s); s = __fmaf_rn(2.0f, phi[i], s); }
  const float s_share = s;  // running sum at the start of the share (kept for the IEEE re-run)
  float *bout = (f == write_frame && c0 < n) ? (axis == 0 ? bx : by) + (size_t)c0 * W : nullptr;
  unsigned kb = 0u;          // largest beta of the share as an order-preserving word (beta >= +0; NaN above +Inf)
  float amin = 1.0f, amax = 1.0f;
  for (int i0 = ia; i0 < ib; i0 += 64) {
    const int i1 = min(i0 + 64, ib);
    // lane L holds the reciprocals of width i0 + L
    const int wl = w_min + min(i0 + lane, W - 1);
    const float r1l = __fdiv_rn(1.0f, (float)(2 * (n - wl))), r2l = __fdiv_rn(1.0f, (float)(2 * wl));
    for (i = i0; i < i1; i += 8) {
      float lo[8], hi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // (reads past the share stay inside the window: i + u < W + 8 <= w_max + 8)
        const int iu = min(i + u, W - 1);
        lo[u] = plo[-iu]; hi[u] = phi[iu];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (i + u < i1) {  // uniform
          s = __fmaf_rn(2.0f, lo[u], s);
          s = __fmaf_rn(2.0f, hi[u], s);
          const int w = w_min + i + u;
          const float d1 = (float)(2 * (n - w)), d2 = (float)(2 * w);
          const float rd1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r1l), i + u - i0));
          const float rd2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r2l), i + u - i0));
          const float x1 = __fsub_rn(S, s);
          const float q1 = __fmul_rn(x1, rd1), q2 = __fmul_rn(s, rd2);
          const float t1 = __fmaf_rn(__fmaf_rn(-q1, d1, x1), rd1, q1), t2 = __fmaf_rn(__fmaf_rn(-q2, d2, s), rd2, q2);
          float v = __fadd_rn(t1, t2);
          v = __fmul_rn(v, v);
          amin = fminf(amin, fminf(fabsf(x1), fabsf(s)));
          amax = fmaxf(amax, fmaxf(fabsf(x1), fabsf(s)));
          if (bout) bout[i + u] = v;
          const unsigned bits = (v != v) ? 0x7FC00000u : __float_as_uint(v);
          kb = max(kb, bits);
        }
      }
    }
  }
  // rare: a value outside Markstein's safe range (all-zero image, denormals, Inf).  The lane redoes its share with
  // IEEE divisions; everything else is unchanged.
  if (!(amin > 1e-30f && amax < 1e30f)) {
    s = s_share;
    kb = 0u;
    for (i = ia; i < ib; ++i) {
      s = __fmaf_rn(2.0f, plo[-i], s);
      s = __fmaf_rn(2.0f, phi[i], s);
      const int w = w_min + i;
      float v = __fadd_rn(__fdiv_rn(__fsub_rn(S, s), (float)(2 * (n - w))), __fdiv_rn(s, (float)(2 * w)));
      v = __fmul_rn(v, v);
      if (bout) bout[i] = v;
      const unsigned bits = (v != v) ? 0x7FC00000u : __float_as_uint(v);
      kb = max(kb, bits);
    }
  }
  // per-centre maximum over the wavefronts' width shares, then the workgroup's first maximum (smallest column on ties)
  // and -- for the sync guard -- the largest value any OTHER column of the workgroup reaches
  colk[q][lane] = (c0 < n && ia < ib) ? kb : 0u;
  __syncthreads();
  if (q == 0) {
    unsigned m = colk[0][lane];
#pragma unroll
    for (int j = 1; j < NWV; ++j) m = max(m, colk[j][lane]);
    unsigned long long key = c0 < n ? (((unsigned long long)m << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)c0)) : 0ull;
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(key, off, 64);
      key = o > key ? o : key;
    }
    if (lane == 0 && key) atomicMax(&keys[(size_t)f * 2 + axis], key);
    if (A.top2) {
      const unsigned wc = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      unsigned m2 = (c0 < n && (unsigned)c0 != wc) ? m : 0u;
      for (int off = 32; off > 0; off >>= 1) m2 = max(m2, (unsigned)__shfl_xor((int)m2, off, 64));
      if (lane == 0) A.top2[(size_t)f * gridDim_x_of_frame(g) + blk] = make_uint2((unsigned)(key >> 32), m2);
    }
  }
}

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void k_beta(BetaArgs A) {
  extern __shared__ float sh[];
  beta_wg<NWV>(A, (int)blockIdx.x, (int)blockIdx.y, sh);
}

__device__ inline int key_col1(unsigned long long key) {  // 1-based column of the packed argmax
  return (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) + 1;
}

// ---- shift + IIR over the frames of one buffer (GUI.jl:172,175) ----------------------------
// For frame f: s_x = argmax beta_x(f); s_y = argmax beta_y(f-1) (pend_in[0] for f = 0).
// out = alpha*out + (1-alpha)*img_shifted, f32, two products and one sum (no FMA).
// Block 0 also publishes (s_y,s_x) per frame and the s_y the NEXT call starts with (pend_out; the two
// pending slots alternate between calls, so no block can read a value this launch has overwritten).
struct IirArgs {
  const float *img;
  size_t img_stride;
  int h, w, frames;
  const unsigned long long *keys;
  const int *pend_in;
  int *pend_out, *sync_idx;
  int do_align;
  int sy_current;   // option "vsync_current_sy": s_y from THIS frame's beta_y instead of the previous call's (the reference's :66)
  float alpha;
  float *state, *frames_out;
};

// body of one shift + IIR workgroup: wg = workgroup index, nthr = its thread count
__device__ __forceinline__ void shift_iir_wg(const IirArgs &A, unsigned wg, unsigned nthr) {
  const float *__restrict__ img = A.img;
  const size_t img_stride = A.img_stride;
  const int h = A.h, w = A.w, frames = A.frames, do_align = A.do_align;
  const unsigned long long *__restrict__ keys = A.keys;
  const int *__restrict__ pend_in = A.pend_in;
  int *__restrict__ pend_out = A.pend_out, *__restrict__ sync_idx = A.sync_idx;
  const float alpha = A.alpha;
  float *__restrict__ state = A.state, *__restrict__ frames_out = A.frames_out;
  const size_t npx = (size_t)h * w;
  const size_t idx = (size_t)wg * nthr + threadIdx.x;
  if (do_align && wg == 0 && threadIdx.x == 0) {
    int sy = pend_in[0];
    for (int f = 0; f < frames; ++f) {
      const int sy_now = key_col1(keys[(size_t)f * 2 + 1]);
      if (sync_idx) { sync_idx[2 * f] = A.sy_current ? sy_now : sy; sync_idx[2 * f + 1] = key_col1(keys[(size_t)f * 2 + 0]); }
      sy = sy_now;
    }
    pend_out[0] = sy;   // (rolled in either mode: the option may change between calls)
  }
  if (idx >= npx) return;
  const int i = (int)(idx % (size_t)h), j = (int)(idx / (size_t)h);
  float acc = state[idx];
  const float oma = __fsub_rn(1.0f, alpha);
  auto src_of = [&](int f) -> size_t {
    if (!do_align) return idx;
    const int sy = A.sy_current ? key_col1(keys[(size_t)f * 2 + 1]) : f == 0 ? pend_in[0] : key_col1(keys[(size_t)(f - 1) * 2 + 1]);
    const int sx = key_col1(keys[(size_t)f * 2 + 0]);
    int si = i + sy; if (si >= h) si %= h;   // 1 <= s <= n: one conditional subtraction almost always
    int sj = j + sx; if (sj >= w) sj %= w;
    return (size_t)sj * h + si;
  };
  // The gathered pixels of FB frames are requested together (their addresses do not depend on acc); the
  // recurrence then runs over them in frame order.  One load in flight per lane made this kernel latency-bound.
  constexpr int FB = 6;
  int f = 0;
  for (; f + FB <= frames; f += FB) {
    float v[FB];
#pragma unroll
    for (int u = 0; u < FB; ++u) v[u] = img[(size_t)(f + u) * img_stride + src_of(f + u)];
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      acc = __fadd_rn(__fmul_rn(alpha, acc), __fmul_rn(oma, v[u]));
      if (frames_out) __builtin_nontemporal_store(acc, &frames_out[(size_t)(f + u) * npx + idx]);
    }
  }
  for (; f < frames; ++f) {
    const float v = img[(size_t)f * img_stride + src_of(f)];
    acc = __fadd_rn(__fmul_rn(alpha, acc), __fmul_rn(oma, v));
    if (frames_out) __builtin_nontemporal_store(acc, &frames_out[(size_t)f * npx + idx]);
  }
  state[idx] = acc;
}

__global__ __launch_bounds__(256) void k_shift_iir(IirArgs A) { shift_iir_wg(A, blockIdx.x, 256u); }

// (Round 4, measured and dropped: four pixels per thread -- state and frames_out as 16-byte vectors, the four shifted source
// pixels of a frame as one 4-byte-aligned global_load_dwordx4, six frames in flight: 32.4 us against 25.0 us for this kernel at
// C2, and the pipelined loop lost 20 %: the misaligned 16-byte gathers cost more than the narrower accesses they replace.)

// standalone vsync: publish (s_y,s_x) of one scanned image and roll the pending s_y
__global__ void k_publish(const unsigned long long *__restrict__ keys, const int *__restrict__ pend_in,
                          int *__restrict__ pend_out, int *__restrict__ s_yx, int sy_current) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (s_yx) { s_yx[0] = sy_current ? key_col1(keys[1]) : pend_in[0]; s_yx[1] = key_col1(keys[0]); }
  pend_out[0] = key_col1(keys[1]);
}

__global__ __launch_bounds__(256) void k_circshift(const float *__restrict__ in, int h, int w, int s_y, int s_x,
                                                   float *__restrict__ out) {
  const size_t npx = (size_t)h * w;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npx) return;
  const int i = (int)(idx % (size_t)h), j = (int)(idx / (size_t)h);
  int si = (i + s_y) % h; if (si < 0) si += h;
  int sj = (j + s_x) % w; if (sj < 0) sj += w;
  out[idx] = in[(size_t)sj * h + si];
}

// standalone fill_beta!: beta (W x n) from an already filtered projection (one lane per centre)
__global__ __launch_bounds__(64) void k_fill_beta(const float *__restrict__ cvin, int n, int w_min, int w_max,
                                                  float *__restrict__ beta) {
  extern __shared__ float cv[];
  const int lane = threadIdx.x;
  for (int i = lane; i < n; i += 64) cv[i] = cvin[i];
  __syncthreads();
  float a = 0.0f;
  for (int i = lane; i < n; i += 64) a = __fadd_rn(a, cv[i]);
  const float S = __shfl(wave_tree64(a), 0, 64);
  const int c0 = blockIdx.x * 64 + lane;
  if (c0 >= n) return;
  const int W = w_max - w_min + 1;
  (void)beta_scan(cv, n, c0, w_min, w_min, w_max, S, beta + (size_t)c0 * W, nullptr);
}

static SyncGeom geom_of(const tsdr_sync *s) {
  SyncGeom g;
  g.y_t = s->y_t; g.x_t = s->x_t;
  g.wmin_y = s->wmin_y; g.wmax_y = s->wmax_y; g.wmin_x = s->wmin_x; g.wmax_x = s->wmax_x;
  g.h0 = s->h[0]; g.h1 = s->h[1]; g.h2 = s->h[2]; g.h3 = s->h[3]; g.h4 = s->h[4];
  return g;
}

// layout k_proj produces for this state's image size
ProjLayout sync_proj_layout(const tsdr_sync *s) {
  ProjLayout pl;
  pl.ncp = (s->y_t + 63) >> 6;
  pl.nrp = kProjRowParts;
  return pl;
}

void sync_image_size(const tsdr_sync *s, int *y_t, int *x_t) { *y_t = s->y_t; *x_t = s->x_t; }

// vsync statistics for `frames` images already on the device: fills keys[2*frames] (x then y per frame); beta
// matrices of the LAST frame are materialised into the sync state.
//   proj: workspace of frames * proj_floats(layout) floats.  have == nullptr: the projections are formed here from
//   the images (k_proj); else *have describes partial sums some producer has already written to proj.
//   A producer other than k_proj must also have cleared keys[2*frames].
constexpr int kBetaWaves = 8;

static void beta_args(tsdr_sync *s, const float *proj, ProjLayout pl, unsigned long long *keys, int frames, BetaArgs *B,
                      unsigned *nbb, size_t *lds, uint2 *top2 = nullptr) {
  const int y = s->y_t, x = s->x_t;
  B->proj = proj; B->proj_stride = proj_floats(y, x, pl); B->ncp = pl.ncp; B->nrp = pl.nrp; B->g = geom_of(s);
  B->keys = keys; B->write_frame = frames - 1; B->bx = s->beta_x; B->by = s->beta_y;
  B->top2 = top2;
  const size_t nmax = (size_t)(x > y ? x : y), wmax = (size_t)std::max(s->wmax_x, s->wmax_y);
  *nbb = (unsigned)(ceil_div((size_t)x, 64) + ceil_div((size_t)y, 64));
  *lds = (2 * nmax + 64 + 2 * wmax + 8) * 4;
}

static void iir_args(tsdr_sync *s, const float *img, size_t img_stride, int h, int w, int frames, const unsigned long long *keys,
                     int do_align, float alpha, float *state, float *frames_out, int *sync_idx, IirArgs *I) {
  I->img = img; I->img_stride = img_stride; I->h = h; I->w = w; I->frames = frames; I->keys = keys;
  I->pend_in = do_align ? s->pending + s->cur : nullptr;
  I->pend_out = do_align ? s->pending + (s->cur ^ 1) : nullptr;
  I->sync_idx = sync_idx; I->do_align = do_align; I->alpha = alpha; I->state = state; I->frames_out = frames_out;
  I->sy_current = (do_align && s) ? s->ctx->opt_vsync_current_sy : 0;   // (s is null when do_align == 0)
}

//   top2 (sync guard, guard.h): k_beta also leaves every workgroup's {best, best other column} pair there.
int sync_scan_d(tsdr_sync *s, const float *img, size_t img_stride, int frames, unsigned long long *keys, float *proj,
                const ProjLayout *have, uint2 *top2) {
  tsdr_ctx *ctx = s->ctx;
  const int y = s->y_t, x = s->x_t;
  if (!proj || !keys) return TSDR_ENOMEM;
  ProjLayout pl;
  if (have) {
    pl = *have;
  } else {
    pl = sync_proj_layout(s);
    TSDR_LAUNCH(ctx, "sync_proj", k_proj, proj_grid(y, frames), proj_block(), proj_lds_bytes(), img, img_stride, y, x, proj,
                proj_floats(y, x, pl), keys);
  }
  BetaArgs B;
  size_t lds = 0;
  unsigned nbb = 0;
  beta_args(s, proj, pl, keys, frames, &B, &nbb, &lds, top2);
  if (ctx->opt_beta_waves == 4) {
    TSDR_LAUNCH(ctx, "sync_beta", k_beta<4>, dim3(nbb, (unsigned)frames), dim3(64 * 4), lds, B);
  } else {
    TSDR_LAUNCH(ctx, "sync_beta", k_beta<kBetaWaves>, dim3(nbb, (unsigned)frames), dim3(64 * kBetaWaves), lds, B);
  }
  return TSDR_OK;
}

// ---- the sync guard's ONE launch (guard.h) ----------------------------------------------------------------------------
// A small persistent grid.  Every workgroup works out the list of flagged frames itself (the first wavefront scans the
// top-2 records k_beta left: a few KB out of L2) -- when the list is empty, which is the rule, all of them leave at once
// and the launch costs one kernel boundary.  Otherwise the re-evaluation of the flagged frames runs as a queue of work
// items in dependency order,
//     A (flagged frame, image tile)   EXACT 600x800 tile from IQ (down_fused_body), written over the FAST image
//     B (flagged frame, row block)    its projections in the reference's order (proj_wg)      -- needs all A of the frame
//     C (flagged frame, centre block) beta scan + argmax key (beta_wg<8>)                      -- needs all B of the frame
// taken by ticket (one atomicAdd per item).  A workgroup that holds ticket t waits only for items with smaller tickets,
// all of which are held by workgroups that are running, so the queue cannot deadlock whatever the dispatch order or the
// number of resident workgroups.  Hand-over between items of different CUs: every storing wavefront drains its stores,
// workgroup barrier, one lane release-fences and bumps the frame's counter; the consumer polls the counter, acquire-
// fences, barrier (MI355X_MICROARCH.md, inter-workgroup visibility).  The last workgroup to leave zeroes the queue words.
constexpr size_t kGuardLdsMax = 128 * 1024;   // k_guard's dynamic LDS opt-in (one workgroup per CU)
struct GuardSync {          // device words, all zero between launches
  unsigned ticket, exited;
  unsigned done[1];         // [2 * frames]: items A / B finished per LIST position
};

struct GuardAllArgs {
  GuardArgs g;
  int frames;
  // A
  const float *iq; size_t in_stride; DownParams dq; float *img; size_t img_stride; int tilesA;
  // B, C
  int y_t, x_t; float *proj; size_t proj_stride; int ncp, nrp; BetaArgs B; int nB, nC;
  unsigned *sync;           // GuardSync
  size_t lds_bytes;         // A's LDS need
  size_t lds_total;         // dynamic LDS of the launch
};

__device__ inline void guard_wait(unsigned *ctr, unsigned target) {
  if (threadIdx.x == 0) {
    // (the queue cannot deadlock by construction; the bound turns a protocol bug into a failed launch instead of a hung GPU:
    // 2^24 polls are several seconds, four orders of magnitude beyond any real wait)
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 24)) __builtin_trap();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}
__device__ inline void guard_signal(unsigned *ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wavefront drains its own stores
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#ifndef TSDR_ROWSUM_CHUNK8
// (launch bound: one workgroup per CU is all the launch asks for, so the bodies may use up to 256 registers: no spills)
__global__ __launch_bounds__(512, 2) void k_guard(GuardAllArgs a) {
  extern __shared__ double glds[];
  __shared__ int list[kGuardChunk];
  __shared__ int cnt_s;
  __shared__ unsigned ticket_s;
  const int tid = threadIdx.x;
  {
    // all top-2 records of the launch come into LDS with ONE round trip (every thread a few), then one lane per frame
    // scans its frame's records there: a lane walking them in global memory pays the L2 latency once per record, which
    // made this -- the whole cost of the launch when nothing is flagged -- 6 us instead of 2
    const int nbb = a.g.nbx + a.g.nby, total = a.frames * nbb;
    uint2 *rec = reinterpret_cast<uint2 *>(glds);
    const bool staged = (size_t)total * sizeof(uint2) <= a.lds_total;
    if (staged) {
      for (int i = tid; i < total; i += 512) rec[i] = a.g.top2[i];
      __syncthreads();
    }
    GuardArgs gl = a.g;
    if (staged) gl.top2 = rec;
    if (tid < 64) {
      int n = 0;
      for (int base = 0; base < a.frames; base += 64) {
        const int f = base + tid;
        bool bad_y = false;
        const bool bad = f < a.frames && guard_eval(gl, f, &bad_y);
        const unsigned long long m = __ballot(bad);
        if (bad) list[n + (int)__builtin_popcountll(m & ((1ull << tid) - 1ull))] = f | (bad_y ? 0x40000000 : 0);
        n += (int)__builtin_popcountll(m);
        if (blockIdx.x == 0 && f < a.frames) a.g.flags[f] = bad ? 1 : 0;
      }
      if (tid == 0) {
        cnt_s = n;
        if (blockIdx.x == 0) {
          atomicAdd(&a.g.stats[0], (unsigned long long)a.frames);
          if (n) atomicAdd(&a.g.stats[1], (unsigned long long)n);
          // this call's own counts into its pinned ring entry (guard launches of two pipeline lanes may complete in either
          // order: the host folds the entries by sequence number, not by arrival)
          unsigned tf = (unsigned)a.frames, tn = (unsigned)n;
          if (a.g.acc) {   // (launches of one call are stream-ordered: plain read-modify-write)
            tf += a.g.acc[0]; tn += a.g.acc[1];
            a.g.acc[0] = a.g.host ? 0u : tf; a.g.acc[1] = a.g.host ? 0u : tn;
          }
          if (a.g.host)
            __hip_atomic_store(a.g.host, (a.g.host_tag << 48) | ((unsigned long long)tf << 24) | (unsigned long long)(tn & 0xFFFFFFu),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    __syncthreads();
  }
  const int cnt = cnt_s;
  if (cnt == 0 || a.g.count_only) return;
  unsigned *ticket = a.sync, *exited = a.sync + 1, *doneA = a.sync + 2, *doneB = a.sync + 2 + a.frames;
  const unsigned nA = (unsigned)cnt * (unsigned)a.tilesA, nBt = (unsigned)cnt * (unsigned)a.nB, nCt = (unsigned)cnt * (unsigned)a.nC;
  for (;;) {
    if (tid == 0) ticket_s = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned t = ticket_s;
    __syncthreads();
    if (t >= nA + nBt + nCt) break;
    // list[j]: frame, bit 30 = the y axis is too close to call as well.  Only then do the row sums -- one strict left-to-right
    // chain per row, the slow part -- have to be redone: when x alone is flagged (the usual case: frames drift along x), the
    // image tiles leave the exact column sums themselves (they are k_proj's 64-row blocks) and the x scan follows them directly
    if (t < nA) {
      const int j = (int)(t / (unsigned)a.tilesA), tile = (int)(t - (unsigned)j * (unsigned)a.tilesA);
      const int f = list[j] & 0x3FFFFFFF;
      if (tile == 0 && tid == 0) a.B.keys[(size_t)f * 2] = 0ull;  // the x key (proj_wg clears both when it runs)
      const int tr = tile / a.dq.tiles_c;
      down_fused_body<true, DM_EXACT, 512, DS_COLSUM>(a.iq, a.in_stride, a.dq, a.img, a.img_stride, tile, f, glds,
                                                  a.proj + (size_t)f * a.proj_stride + (size_t)tr * a.x_t,
                                                  reinterpret_cast<float *>(reinterpret_cast<char *>(glds) + a.lds_bytes));
      guard_signal(doneA + j);
    } else if (t < nA + nBt) {
      const unsigned u = t - nA;
      const int j = (int)(u / (unsigned)a.nB), rb = (int)(u - (unsigned)j * (unsigned)a.nB);
      if (list[j] & 0x40000000) {
        guard_wait(doneA + j, (unsigned)a.tilesA);
        proj_wg(a.img, a.img_stride, a.y_t, a.x_t, a.proj, a.proj_stride, a.B.keys, list[j] & 0x3FFFFFFF, rb);
        guard_signal(doneB + j);
      }
    } else {
      const unsigned u = t - nA - nBt;
      const int j = (int)(u / (unsigned)a.nC), blk = (int)(u - (unsigned)j * (unsigned)a.nC);
      const bool need_y = (list[j] & 0x40000000) != 0;
      const int nbx = (a.x_t + 63) >> 6;
      if (need_y) {
        guard_wait(doneB + j, (unsigned)a.nB);
        beta_wg<8>(a.B, blk, list[j] & 0x3FFFFFFF, reinterpret_cast<float *>(glds));
      } else if (blk < nbx) {
        guard_wait(doneA + j, (unsigned)a.tilesA);
        beta_wg<8>(a.B, blk, list[j] & 0x3FFFFFFF, reinterpret_cast<float *>(glds));
      }
    }
    __syncthreads();  // the item's LDS is free again
  }
  // leave: the last workgroup out zeroes the queue words for the next launch
  if (tid == 0) {
    if (__hip_atomic_fetch_add(exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
      for (int i = 0; i < 2 + 2 * a.frames; ++i) __hip_atomic_store(a.sync + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
#endif

// Sync guard: frames whose FAST-mode decision was closer than g.thr are re-evaluated in the exact sequence (image, projections
// in the reference's order, beta scan); img / keys of those frames are overwritten, everything else is left alone.
// plan_only: report whether this geometry can be guarded at all.
bool guard_image_plan(tsdr_ctx *ctx, size_t S, int y_t, int x_t, int h_out, int w_out, DownParams *q, size_t *lds);

int sync_guard_d(tsdr_sync *s, const float *iq, size_t S, int y_t, int x_t, int frames, float *img, size_t img_stride,
                 unsigned long long *keys, float *proj, const GuardArgs &g, bool *can, bool plan_only) {
  tsdr_ctx *ctx = s->ctx;
  if (can) *can = false;
#ifdef TSDR_ROWSUM_CHUNK8
  return TSDR_OK;
#else
  GuardAllArgs a{};
  if (!guard_image_plan(ctx, S, y_t, x_t, s->y_t, s->x_t, &a.dq, &a.lds_bytes)) return TSDR_OK;
  {  // the launch's dynamic LDS (exact tiles of up to 96 KiB + their column-sum scratch) plus k_guard's static arrays (list,
     // colk, ticket words: < 4 KiB) must fit what the kernel opted in to; otherwise the frame loop falls back to whole
     // buffers in TSDR_EXACT, as for geometries without a fused exact kernel
    const size_t lds_a = ((a.lds_bytes + 15) & ~(size_t)15) + (size_t)a.dq.TC * 65 * 4;
    if (std::max(lds_a, kProjLds) + 4096 > kGuardLdsMax) return TSDR_OK;
  }
  if (can) *can = true;
  if (plan_only || frames <= 0) return TSDR_OK;
  const int y = s->y_t, x = s->x_t;
  const ProjLayout pl = sync_proj_layout(s);
  size_t lds_beta = 0;
  unsigned nbb = 0;
  const int ncu = ctx->cu_count > 0 ? ctx->cu_count : 256;
  for (int f0 = 0; f0 < frames; f0 += kGuardChunk) {   // (the list of flagged frames lives in LDS)
    const int nf = std::min(kGuardChunk, frames - f0);
    // queue words: zero between launches (the kernel restores that itself)
    const size_t words = 2 + 2 * (size_t)kGuardChunk + 2;   // + the per-call accumulator of multi-launch calls (GuardArgs::acc)
    unsigned *&qwords = ctx->guard_sync[ctx->pipe_lane & 3];   // (guard launches of different pipeline lanes may run side by side)
    if (!qwords) {
      TSDR_HIP(ctx, hipMalloc((void **)&qwords, words * 4));
      TSDR_HIP(ctx, hipMemsetAsync(qwords, 0, words * 4, ctx->launch_stream));
    }
    a.g = g;
    if (frames > kGuardChunk) {   // the call's ring entry is written once, by its last launch, with the call's totals
      a.g.acc = qwords + 2 + 2 * (size_t)kGuardChunk;
      if (f0 + nf < frames) a.g.host = nullptr;
    }
    a.g.top2 = g.top2 + (size_t)f0 * (size_t)(g.nbx + g.nby);
    a.g.flags = g.flags + f0;
    a.frames = nf;
    // (an sc16 buffer has one float's worth of bytes per sample)
    a.iq = iq + (size_t)f0 * S * iq_floats(ctx->iq_fmt); a.in_stride = S; a.img = img + (size_t)f0 * img_stride; a.img_stride = img_stride;
    a.tilesA = (int)(ceil_div((size_t)y, 64) * (size_t)a.dq.tiles_c);
    a.y_t = y; a.x_t = x;
    a.proj_stride = proj_floats(y, x, pl); a.ncp = pl.ncp; a.nrp = pl.nrp;
    a.proj = proj + (size_t)f0 * a.proj_stride;
    beta_args(s, a.proj, pl, keys + (size_t)f0 * 2, nf, &a.B, &nbb, &lds_beta, nullptr);
    a.B.write_frame = (f0 + nf == frames) ? nf - 1 : -1;
    a.nB = (y + 63) >> 6; a.nC = (int)nbb;
    a.sync = qwords;
    a.lds_bytes = (a.lds_bytes + 15) & ~(size_t)15;
    const size_t lds = std::max(std::max(a.lds_bytes + (size_t)a.dq.TC * 65 * 4, kProjLds), lds_beta);
    a.lds_total = lds;
    // one workgroup per CU: 64 / 128 / 512 measured no better.  (Round 4, dropped: an eighth of the grid after a few launches
    // that flagged nothing -- the launch's cost beside the pipeline's image kernel is finding room for ONE 512-thread
    // workgroup with 52 KiB of LDS, not their number, so it bought nothing there and cost C3, where a frame is flagged every
    // other buffer, 0.11 instead of 0.057 ms per step.)
    const unsigned grid = g.count_only ? 1u : (unsigned)ncu;
    if (lds > 64 * 1024) {   // above what a kernel gets without opting in
      int rco = lds_opt_in(ctx, (const void *)k_guard, kGuardLdsMax);
      if (rco) return rco;
    }
    TSDR_LAUNCH(ctx, "sync_guard", k_guard, dim3(grid), dim3(512), lds, a);
  }
  return TSDR_OK;
#endif
}

void sync_beta_blocks(const tsdr_sync *s, int *nbx, int *nby) { *nbx = (s->x_t + 63) >> 6; *nby = (s->y_t + 63) >> 6; }

// workspace for sync_scan_d: the projection buffer (and, on request, a key buffer) for `frames` frames in slot `slot`
// (0/1: the two-stage pipeline keeps two buffers in flight) with room for `pl` (or k_proj's layout when pl == nullptr)
int sync_workspace(tsdr_sync *s, int frames, int slot, int nslots, const ProjLayout *pl_in, float **proj,
                   unsigned long long **keys) {
  tsdr_ctx *ctx = s->ctx;
  const ProjLayout k = sync_proj_layout(s);
  ProjLayout pl = pl_in ? *pl_in : k;
  pl.ncp = std::max(pl.ncp, k.ncp);
  pl.nrp = std::max(pl.nrp, k.nrp);
  const size_t pf = (size_t)frames * proj_floats(s->y_t, s->x_t, pl);
  float *p = (float *)ctx->scratch(WS_PROJ, (size_t)nslots * pf * 4);
  if (!p) return TSDR_ENOMEM;
  *proj = p + (size_t)slot * pf;
  if (keys) {  // callers that bring their own key buffer pass nullptr and WS_KEYS is left alone
    unsigned long long *kk = (unsigned long long *)ctx->scratch(WS_KEYS, (size_t)nslots * frames * 2 * 8);
    if (!kk) return TSDR_ENOMEM;
    *keys = kk + (size_t)slot * frames * 2;
  }
  return TSDR_OK;
}

int shift_iir_d(tsdr_ctx *ctx, tsdr_sync *s, const float *img, size_t img_stride, int h, int w, int frames,
                const unsigned long long *keys, int do_align, float alpha, float *state, float *frames_out,
                int *sync_idx) {
  const size_t npx = (size_t)h * w;
  IirArgs I;
  iir_args(s, img, img_stride, h, w, frames, keys, do_align, alpha, state, frames_out, sync_idx, &I);
  if (ctx->launch_stop_ev && !ctx->prof_on) {
    // the pipeline's "tail of this buffer done" event is this dispatch's own completion signal: no marker packet behind it
    hipEvent_t ev = ctx->launch_stop_ev;
    ctx->launch_stop_ev = nullptr;
    hipExtLaunchKernelGGL(k_shift_iir, dim3((unsigned)ceil_div(npx, 256)), dim3(256), 0, ctx->launch_stream, nullptr, ev, 0, I);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hip_fail(ctx, le, "shift_iir");
  } else {
    const bool rec = ctx->launch_stop_ev != nullptr;
    hipEvent_t ev = ctx->launch_stop_ev;
    ctx->launch_stop_ev = nullptr;
    TSDR_LAUNCH(ctx, "shift_iir", k_shift_iir, dim3((unsigned)ceil_div(npx, 256)), dim3(256), 0, I);
    if (rec) TSDR_HIP(ctx, hipEventRecord(ev, ctx->launch_stream));
  }
  if (do_align) s->cur ^= 1;
  return TSDR_OK;
}

// Pipelined frame loop, symmetric mode (frames.hip): the beta matrices k_beta / k_guard leave for the LAST frame of a buffer go
// to a set of their own per lane, since the statistics of two buffers may run side by side; the state's current matrices
// are those of the lane submitted last.
int sync_use_lane(tsdr_sync *s, int lane) {
  tsdr_ctx *ctx = s->ctx;
  lane &= 3;
  if (!s->bset[lane][0]) {
    const size_t nbx = (size_t)(1 + s->wmax_x - s->wmin_x) * s->x_t, nby = (size_t)(1 + s->wmax_y - s->wmin_y) * s->y_t;
    float *bx = nullptr, *by = nullptr;
    if (hipMalloc((void **)&bx, nbx * 4) != hipSuccess || hipMalloc((void **)&by, nby * 4) != hipSuccess) {
      if (bx) (void)hipFree(bx);
      return set_err(ctx, TSDR_ENOMEM, "sync state allocation failed");
    }
    s->bset[lane][0] = bx; s->bset[lane][1] = by;
  }
  s->beta_x = s->bset[lane][0];
  s->beta_y = s->bset[lane][1];
  return TSDR_OK;
}

}  // namespace tsdr

using namespace tsdr;

extern "C" {

int tsdr_sync_create(tsdr_ctx *ctx, int y_t, int x_t, tsdr_sync **out) {
  if (!ctx || !out) return TSDR_EINVAL;
  *out = nullptr;
  if (y_t < 8 || x_t < 20) return set_err(ctx, TSDR_EINVAL, "SyncXY needs an image of at least 8x20");
  // k_beta keeps one axis' raw and filtered projection plus a window of it in LDS (about 10 * max(x_t, y_t) bytes of the
  // 64 KiB a kernel gets without opting in): larger images are refused here, loudly, not at launch
  if (y_t > 6000 || x_t > 6000) return set_err(ctx, TSDR_EINVAL, "SyncXY supports images up to 6000x6000 (got %dx%d)", y_t, x_t);
  tsdr_sync *s = new tsdr_sync();
  s->ctx = ctx; s->y_t = y_t; s->x_t = x_t;
  // init_gaussian_filter(5): exp(-2k^2/25), k=-2..2, normalised in f64, stored as Float32
  double t[5], sum = 0.0;
  for (int k = -2; k <= 2; ++k) { t[k + 2] = exp(-2.0 * (double)(k * k) / 25.0); sum += t[k + 2]; }
  for (int i = 0; i < 5; ++i) s->h[i] = (float)(t[i] / sum);
  s->wmin_y = (int)ceil(1.0 / 100.0 * (double)y_t);
  s->wmax_y = (int)floor((double)y_t / 4.0);
  s->wmin_x = (int)ceil(5.0 / 100.0 * (double)x_t);
  s->wmax_x = (int)floor((double)x_t / 4.0);
  const size_t nbx = (size_t)(1 + s->wmax_x - s->wmin_x) * x_t, nby = (size_t)(1 + s->wmax_y - s->wmin_y) * y_t;
  if (hipMalloc((void **)&s->bset[0][0], nbx * 4) != hipSuccess || hipMalloc((void **)&s->bset[0][1], nby * 4) != hipSuccess ||
      hipMalloc((void **)&s->pending, 16) != hipSuccess) {
    tsdr_sync_free(s);
    return set_err(ctx, TSDR_ENOMEM, "sync state allocation failed");
  }
  s->beta_x = s->bset[0][0]; s->beta_y = s->bset[0][1];
  int rc = tsdr_sync_reset(s);
  if (rc) { tsdr_sync_free(s); return rc; }
  *out = s;
  return TSDR_OK;
}

int tsdr_sync_reset(tsdr_sync *s) {
  if (!s) return TSDR_EINVAL;
  tsdr_ctx *ctx = s->ctx;
  {  // a pipelined submission still owes this state its shift + IIR stage
    int rc = pipe_drain(ctx);
    if (rc) return rc;
  }
  const size_t nbx = (size_t)(1 + s->wmax_x - s->wmin_x) * s->x_t, nby = (size_t)(1 + s->wmax_y - s->wmin_y) * s->y_t;
  TSDR_HIP(ctx, hipMemsetAsync(s->beta_x, 0, nbx * 4, ctx->stream));
  TSDR_HIP(ctx, hipMemsetAsync(s->beta_y, 0, nby * 4, ctx->stream));
  const int one[4] = {1, 1, 0, 0};  // findmax of an all-zero beta_y is index (1,1)
  s->cur = 0;
  TSDR_HIP(ctx, hipMemcpyAsync(s->pending, one, 16, hipMemcpyHostToDevice, ctx->stream));
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  return TSDR_OK;
}

void tsdr_sync_free(tsdr_sync *s) {
  if (!s) return;
  if (s->ctx) {
    (void)pipe_drain(s->ctx);  // submitted buffers read this state's pending s_y: they come first
    if (tsdr::wait_stream(s->ctx, s->ctx->stream, "tsdr_sync_free")) return;   // (a stuck stream keeps the state's memory)
  }
  for (auto &b : s->bset) for (float *p : b) if (p) (void)hipFree(p);
  if (s->pending) (void)hipFree(s->pending);
  delete s;
}

int tsdr_sync_bounds(const tsdr_sync *s, int b[4]) {
  if (!s || !b) return TSDR_EINVAL;
  b[0] = s->wmin_y; b[1] = s->wmax_y; b[2] = s->wmin_x; b[3] = s->wmax_x;
  return TSDR_OK;
}

int tsdr_vsync_d(tsdr_sync *s, const float *img, int *s_yx_dev) {
  if (!s || !img) return TSDR_EINVAL;
  tsdr_ctx *ctx = s->ctx;
  {
    int rc = pipe_drain(ctx);
    if (rc) return rc;
  }
  float *proj = nullptr;
  unsigned long long *keys = nullptr;
  int rc = sync_workspace(s, 1, 0, 1, nullptr, &proj, &keys);
  if (rc) return rc;
  rc = sync_scan_d(s, img, (size_t)s->y_t * s->x_t, 1, keys, proj, nullptr, nullptr);
  if (rc) return rc;
  TSDR_LAUNCH(ctx, "sync_publish", k_publish, dim3(1), dim3(64), 0, (const unsigned long long *)keys,
              (const int *)(s->pending + s->cur), s->pending + (s->cur ^ 1), s_yx_dev, ctx->opt_vsync_current_sy);
  s->cur ^= 1;
  return TSDR_OK;
}

int tsdr_vsync(tsdr_sync *s, const float *img, int *s_y, int *s_x) {
  if (!s || !img || !s_y || !s_x) return TSDR_EINVAL;
  tsdr_ctx *ctx = s->ctx;
  const size_t bytes = (size_t)s->y_t * s->x_t * 4;
  float *d = (float *)ctx->scratch(WS_IN, bytes);
  int *didx = (int *)ctx->scratch(WS_OUT, 16);
  if (!d || !didx) return TSDR_ENOMEM;
  TSDR_HIP(ctx, hipMemcpyAsync(d, img, bytes, hipMemcpyHostToDevice, ctx->stream));
  int rc = tsdr_vsync_d(s, d, didx);
  if (rc) return rc;
  int hidx[2];
  TSDR_HIP(ctx, hipMemcpyAsync(hidx, didx, 8, hipMemcpyDeviceToHost, ctx->stream));
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  *s_y = hidx[0]; *s_x = hidx[1];
  return TSDR_OK;
}

int tsdr_sync_beta(tsdr_sync *s, int which, float *beta_host) {
  if (!s || !beta_host || (which != 0 && which != 1)) return TSDR_EINVAL;
  tsdr_ctx *ctx = s->ctx;
  const size_t n = which == 0 ? (size_t)(1 + s->wmax_x - s->wmin_x) * s->x_t : (size_t)(1 + s->wmax_y - s->wmin_y) * s->y_t;
  {  // submitted buffers fill the beta sets on the pipeline's internal streams: they come first
    int rc = pipe_drain(ctx);
    if (rc) return rc;
  }
  TSDR_HIP(ctx, hipMemcpyAsync(beta_host, which == 0 ? s->beta_x : s->beta_y, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  return TSDR_OK;
}

int tsdr_fill_beta(tsdr_ctx *ctx, const float *cv, int n, int w_min, int w_max, float *beta) {
  if (!ctx || !cv || !beta || n < 2 || w_min < 1 || w_max < w_min || w_max >= n) return TSDR_EINVAL;
  const size_t W = (size_t)(w_max - w_min + 1);
  return host_map(ctx, cv, (size_t)n * 4, beta, W * n * 4, [&](void *i, void *o) {
    TSDR_LAUNCH(ctx, "fill_beta", k_fill_beta, dim3((unsigned)ceil_div((size_t)n, 64)), dim3(64), (size_t)n * 4,
                (const float *)i, n, w_min, w_max, (float *)o);
    return (int)TSDR_OK;
  });
}

int tsdr_circshift_neg(tsdr_ctx *ctx, const float *img, int h, int w, int s_y, int s_x, float *out) {
  if (!ctx || !img || !out || h <= 0 || w <= 0) return TSDR_EINVAL;
  const size_t npx = (size_t)h * w;
  s_y %= h; s_x %= w;   // any shift is legal (circshift reduces it); the kernel adds it to an index in 32 bits
  return host_map(ctx, img, npx * 4, out, npx * 4, [&](void *i, void *o) {
    TSDR_LAUNCH(ctx, "circshift", k_circshift, dim3((unsigned)ceil_div(npx, 256)), dim3(256), 0, (const float *)i, h, w,
                s_y, s_x, (float *)o);
    return (int)TSDR_OK;
  });
}

}  // extern "C"
