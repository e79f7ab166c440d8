// down_fused.h -- sig_to_image |> downgradeImage without the raster, as a workgroup body: shared by resample.hip's
// k_down_fused (the raster-free frame path) and sync.hip's sync-guard kernel, which re-derives single frames in the
// exact operation sequence.
#pragma once
#include "common.h"

namespace tsdr {


// ---- |IQ| -----------------------------------------------------------------------------------------------
template <bool EXACT>
__device__ inline float abs_iq(float re, float im) {
  if (EXACT) return abs_c(re, im);
  const float m = fmaf(re, re, im * im);
  if (m > 1e-30f && m < 1e30f) return __builtin_amdgcn_sqrtf(m);  // v_sqrt_f32 (1 ulp); <= 1.5 ulp overall
  return abs_c(re, im);                                            // tiny / huge / non-finite: the f64 form
}

// FAST |IQ| of the raster-free kernel: the same m as abs_iq<false>, but a correctly rounded square root (HIP's sqrtf: v_sqrt_f32
// plus a fix-up, ~10 instructions on the ~460 k samples a frame's tiles stage) -- 0.75 ulp overall instead of 1.5.  Frames of
// this kernel are within 3.6e-7 of EXACT's over the fuzzers' cases.  Making the samples EXACT's own -- by the f64 square root
// (+7.7 us per C2 buffer) or by sqrtf plus an exact midpoint test (+4.4 us) -- was measured and brings 2.4e-7; not taken.
__device__ inline float abs_iq_rn(float re, float im) {
  const float m = fmaf(re, re, im * im);
  if (m > 1e-30f && m < 1e30f) return sqrtf(m);
  return abs_c(re, im);
}

// a + d*(b - a) with one f64 FMA and one rounding to f32: within 1 ulp of the value whatever |b-a| is
__device__ inline float fast_blend(float a, float b, double d) {
  return (float)fma(d, (double)b - (double)a, (double)a);
}

// sum over the 64 lanes by the fixed DPP tree (row_shr 1, 2, 4, 8, then row_bcast 15 and 31); the total lands in lane 63.
// Each step is ONE in-place v_add_f32_dpp -- v[i] += v[i - k] on the lanes the step reaches; lanes it does not reach (out of
// their row: bound_ctrl reads 0; masked by row / bank mask: not written) keep their value, i.e. add 0.  Written as assembly
// because the builtin form (update_dpp with old = 0, then an add) compiles to mov 0 + mov_dpp + add per step: 20 instructions
// instead of 6 in kernels that issue VALU instructions all the time.  s_nop 1: a DPP source written by the previous VALU
// instruction needs two wait states.  Must be called with all 64 lanes active.
__device__ __forceinline__ float wave_sum63(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// Sums over the 64 lanes of FOUR values at once (round 4): two v_permlane32_swap + adds fold lanes l and l + 32 of (a, b)
// and of (c, d) into one register each, one v_permlane16_swap + add folds rows of 16 lanes, four in-row DPP adds finish:
// 10 instructions for four sums instead of 4 x 6.  Totals land in lane 15 (a), 31 (c), 47 (b), 63 (d).  Fixed order.
// Must be called with all 64 lanes active.
__device__ __forceinline__ float wave_sum4(float a, float b, float c, float d) {
  auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const float ab = __fadd_rn(__uint_as_float(p[0]), __uint_as_float(p[1]));   // lanes < 32: a[l] + a[l+32]; lanes >= 32: b
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
  const float cd = __fadd_rn(__uint_as_float(q[0]), __uint_as_float(q[1]));
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ab), __float_as_uint(cd), false, false);
  float v = __fadd_rn(__uint_as_float(r[0]), __uint_as_float(r[1]));          // rows of 16 lanes: a | c | b | d
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// ------------------------------------------------------------------------------------------------------------
// k_down_fused: sig_to_image |> downgradeImage without the raster.  Tile = 64 output rows x TC output columns;
// the source lines those rows touch are staged in LDS; each output pixel evaluates its four raster taps (each
// rounded to f32 as the materialised raster would hold it) and blends them.
// ------------------------------------------------------------------------------------------------------------
struct DownParams {
  unsigned S;
  int y_t, x_t, h_out, w_out;
  int TC, NL, W, tiles_c;
  int lpl_log;
  int ld16 = 0;                     // staging: 16 loads in flight per lane instead of 4 (wide rows): selects the LD = 16 kernel
  int sparse = 0;                   // 1: only the two source lines of every output row are staged (row r: staged rows 2r, 2r+1) --
                                    // for vertical ratios above 2, where the lines BETWEEN them would be more than half the tile
  int xcd_tpx = 0, xcd_tiles = 0;   // k_down_fused's XCD-aware 1-D grid: tiles per XCD and frame (0: plain (tile, frame) grid), tiles per frame
  // PSUM (FAST frame loop without a raster): projection partial sums of the (h_out, w_out) image and the frame's argmax keys
  float *proj = nullptr;        // per frame: colpart[row blocks][w_out] | rowpart[tiles_c][h_out]  (sync_layout.h)
  size_t proj_stride = 0;
  unsigned long long *keys = nullptr;
  IqFmt iqf;                    // CPLX input: ComplexF32 or int16 pairs (common.h)
};

// DM_FAST_F32: f64 tap coordinates, f64 blends (any sample rate).  DM_FAST_FX (round 4; at most 0.5 samples per raster pixel):
// tap coordinates in 32.32 fixed point -- a tap's sample index and its fraction are the two halves of one 64-bit add --, the
// second tap of a line from the first one's three-sample window, every blend one f32 FMA: ~50 full-rate VALU instructions per
// output pixel instead of ~75, a third of them half-rate f64.
enum { DM_EXACT = 0, DM_FAST_F32 = 2, DM_FAST_FX = 3 };   // (a {a, slope} f64-pair staging, mode 1, lost to f32 staging with 64-column tiles: 65 vs 50 us)

// EXACT raster value at flat index `flat` (0-based) of a staged f32 line
__device__ inline float raster_tap_exact(const RsAxis &ax1, bool same1, unsigned flat, const float *row, int kf) {
  double d;
  const int j = (int)rs_pos(ax1, (double)(flat + 1u), d) - kf;
  return same1 ? (d == 1.0 ? row[j + 1] : row[j]) : rs_blend(row[j], row[j + 1], d);
}
// FAST raster value at 0-based source coordinate x >= 0 (f64) from the staged f32 samples
template <int MODE>
__device__ inline float raster_tap_fast(double x, const void *row, int kf) {
  const double xf = floor(x);
  const int j = (int)xf - kf;
  const float *r = reinterpret_cast<const float *>(row);
  return fast_blend(r[j], r[j + 1], x - xf);
}

// NT: threads of the workgroup (a multiple of 64); lds_dn: its dynamic LDS region (plan_down's `lds` bytes).
// COLSUM (sync guard): the tile's rows are one of k_proj's 64-row blocks, so the body also leaves that block's column
// sums of its TC columns -- rows added in row order from 0.0f, exactly k_proj's colpart -- in colpart[c] (c = image
// column); colT: TC * 65 floats of LDS beyond lds_dn's region.
// SUMS = 2 (PSUM, FAST only): the tile also leaves the partial projection sums of its 64 rows x TC columns -- per column the
// sum over the rows (one fixed DPP tree per column, -> colpart[row block][column]) and per row the sum over the columns (a
// wavefront adds its columns in ascending order, the wavefronts are added in order through LDS, -> rowpart[column tile][row]):
// every slot is written by exactly one lane, nothing is cleared; tile 0 clears the frame's two argmax keys.  psum_lds:
// (NT + TC) floats of LDS beyond lds_dn's region.
enum { DS_NONE = 0, DS_COLSUM = 1, DS_PSUM = 2 };
// LD: staging loads in flight per lane and trip (4; 16 for wide rows -- a template parameter, not a run-time branch: the
// kernel's register allocation is the maximum over its paths, and 24 more VGPRs cost the 4-load geometries 18 %)
// IQF: how IQ samples are read (common.h): the EXACT body (also the sync guard's) takes the format from the parameters, the FAST
// bodies have it fixed at compile time
template <bool CPLX, int MODE, int NT, int SUMS = DS_NONE, int LD = 4, int IQF = (MODE == DM_EXACT ? IQF_RT : IQF_CF32)>
__device__ __forceinline__ void down_fused_body(const float *__restrict__ in, size_t in_stride, const DownParams &q,
                                       float *__restrict__ out, size_t out_stride, int tile_idx, int f, double *lds_dn,
                                       float *__restrict__ colpart = nullptr, float *colT = nullptr) {
  constexpr bool EXACT = MODE == DM_EXACT, FX = MODE == DM_FAST_FX;
  constexpr bool COLSUM = SUMS == DS_COLSUM, PSUM = SUMS == DS_PSUM;
  constexpr int SB = 4;  // bytes per staged sample (f64 staging, which would save the taps' conversions: 58 vs 54 us)
  const int Wp = q.W | 1;
  char *base = reinterpret_cast<char *>(lds_dn);
  double *cdx = reinterpret_cast<double *>(base + (((size_t)q.NL * Wp * SB + 15) & ~(size_t)15));  // [TC] column weight
  double *cxs = cdx + q.TC;                                                                           // [TC] sf * kx
  int *ckx = reinterpret_cast<int *>(cxs + q.TC);                                                     // [TC] kx
  int *kfirst = ckx + q.TC;                                                                           // [NL]
  const int tr = tile_idx / q.tiles_c, tc = tile_idx - tr * q.tiles_c;
  const int r0 = tr * 64, c0 = tc * q.TC;
  const float *src = in + (size_t)f * in_stride * (CPLX ? iq_floats_as<IQF>(q.iqf) : 1);
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
  const int nl = q.sparse ? 2 * min(64, q.h_out - r0) : ly1 - ly0 + 1;
  const int pxa = (int)rs_pos(axx, (double)(c0 + 1), dtmp);
  for (int i = tid; i < nl; i += NT) {
    // source line of staged row i: dense -- ly0 + i; sparse -- the upper (i even) or lower tap line of output row r0 + i/2
    const int li = q.sparse ? (int)rs_pos(ay, (double)(r0 + (i >> 1) + 1), dtmp) + (i & 1) : ly0 + i;
    const unsigned flat = (unsigned)li * (unsigned)q.x_t + (unsigned)pxa;
    int k;
    if (EXACT) k = (int)rs_pos(ax1, (double)(flat + 1u), dtmp);
    else if (FX) k = (int)floor(fma(ax1.sf, (double)flat + 0.5, -0.5)) - 1;   // may be -1 / -2 on the frame's first line: the
                                                                               // staging clamps, so x < 0 reads s[0] twice
    else k = max((int)floor(fmax(fma(ax1.sf, (double)flat + 0.5, -0.5), 0.0)) - 1, 0);  // one spare sample on the
    kfirst[i] = k;                                                                        // left: tap coordinates
  }                                                                                       // are built by additions
  if (tid < q.TC) {  // per-column table: kx, weight, and the column's coordinate offset sf*kx
    const int c = min(c0 + tid, q.w_out - 1);
    double dx;
    const int kx = (int)rs_pos(axx, (double)(c + 1), dx);
    ckx[tid] = kx;
    if (FX) {  // the same table slots: weight as f32, sf*kx in 32.32 fixed point
      reinterpret_cast<float2 *>(cdx)[tid] = make_float2((float)dx, (float)(1.0 - dx));
      reinterpret_cast<long long *>(cxs)[tid] = __double2ll_rd(ax1.sf * (double)kx * 4294967296.0);
    } else { cdx[tid] = dx; cxs[tid] = ax1.sf * (double)kx; }
  }
  __syncthreads();
  {  // stage the source lines: a lane's loads of one trip (4, or 16 for wide rows: q.ld16) are issued before any |IQ| math --
     // with 4 in flight a wide tile (C3: 121 samples per line) is a chain of dependent round trips, 28 us per tile
    const int lpl = 1 << q.lpl_log;
    const int sub = tid >> q.lpl_log, j0 = tid & (lpl - 1), nsub = NT >> q.lpl_log;
    // LN lines per trip (two when a lane has few loads per line): the loads of both are in flight before either is consumed --
    // with 32 lines per pass a C2 tile (121 lines, 4 loads per lane and line) was four dependent round trips
    constexpr int LN = LD <= 4 ? 2 : 1;
    for (int i = sub; i < nl; i += LN * nsub) {
      int kf[LN];
#pragma unroll
      for (int n = 0; n < LN; ++n) kf[n] = kfirst[min(i + n * nsub, nl - 1)];
      for (int jb = j0; jb < q.W; jb += LD * lpl) {
        float re[LN][LD], im[LN][LD];
#pragma unroll
        for (int n = 0; n < LN; ++n) {
#pragma unroll
          for (int u = 0; u < LD; ++u) {
            const unsigned k = (unsigned)min(max(kf[n] + min(jb + u * lpl, q.W - 1), 0), (int)q.S - 1);
            if (CPLX) { const float2 z = ld_iq_as<IQF>(src, k, q.iqf); re[n][u] = z.x; im[n][u] = z.y; }
            else { re[n][u] = src[k]; im[n][u] = 0.f; }
          }
        }
#pragma unroll
        for (int n = 0; n < LN; ++n) {
          const int in = i + n * nsub;
#pragma unroll
          for (int u = 0; u < LD; ++u) {
            const int j = jb + u * lpl;
            if (j < q.W && in < nl) {
              const float a = !CPLX ? re[n][u] : EXACT ? abs_iq<true>(re[n][u], im[n][u]) : abs_iq_rn(re[n][u], im[n][u]);
              reinterpret_cast<float *>(base)[in * Wp + j] = a;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // (wave-uniform on purpose: the column loop's counters, table addresses and bounds then live in scalar registers)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = r0 + lane;
  if (SUMS == DS_NONE && r >= q.h_out) return;  // (no barrier follows in the body)
  float racc = 0.0f;   // PSUM: this wavefront's sum of row r over its columns
  float *prow = PSUM ? colT : nullptr;          // [NT/64][64] row sums per wavefront, then [TC] column sums
  float *pcol = PSUM ? colT + NT : nullptr;
  if (r < q.h_out || PSUM) {
  const bool live = r < q.h_out;
  const int rr_ = live ? r : q.h_out - 1;       // (PSUM: lanes past the image follow the last row and contribute 0)
  double dy;
  const int ky = (int)rs_pos(ay, (double)(rr_ + 1), dy);
  const int i0 = q.sparse ? 2 * (rr_ - r0) : ky - ly0;
  const char *row0 = base + (size_t)i0 * Wp * SB;
  const char *row1 = row0 + (size_t)Wp * SB;
  const int kf0 = kfirst[i0], kf1 = kfirst[i0 + 1];
  const unsigned b0 = (unsigned)ky * (unsigned)q.x_t, b1 = b0 + (unsigned)q.x_t;
  const int cend = min(c0 + q.TC, q.w_out);
  float *o = out + (size_t)f * out_stride + (size_t)rr_;
  // FAST: 0-based source coordinate of raster pixel (ky, 0) and the per-line / per-pixel increments
  const double xrow = fma(ax1.sf, (double)b0 + 0.5, -0.5), xline = ax1.sf * (double)q.x_t;
  // FX: 32.32 fixed-point coordinates of the two lines' first raster pixel, relative to the staged rows' first samples
  long long B0 = 0, B1 = 0;
  unsigned sf_lo = 0u;
  float dys = 0.f, uys = 0.f;
  if (FX) {
    B0 = __double2ll_rd(xrow * 4294967296.0) - ((long long)kf0 << 32);
    B1 = __double2ll_rd((xrow + xline) * 4294967296.0) - ((long long)kf1 << 32);
    sf_lo = (unsigned)__double2ll_rd(ax1.sf * 4294967296.0);   // sf < 1 on this route
    dys = (float)(dy * 0x1p-32); uys = (float)((1.0 - dy) * 0x1p-32);
  }
  if (FX) {
    // four columns per trip (this wavefront's columns are NT/64 apart): their table and sample reads are issued together,
    // and the four column sums share one reduction
    constexpr int CSTEP = NT / 64;
    for (int cb = c0 + wave; cb < cend; cb += 4 * CSTEP) {
      float vv[4];
      long long cx[4];
      float dxw[4], uxw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ct = min(cb + u * CSTEP, cend - 1) - c0;
        const float2 w2 = reinterpret_cast<const float2 *>(cdx)[ct];   // {dx, 1 - dx}, each rounded from f64
        dxw[u] = w2.x; uxw[u] = w2.y;
        cx[u] = reinterpret_cast<const long long *>(cxs)[ct];
      }
      float R[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int ln = 0; ln < 2; ++ln) {
          const unsigned long long X = (unsigned long long)((ln ? B1 : B0) + cx[u]);
          const unsigned j = (unsigned)(X >> 32), lo = (unsigned)X;
          const float *p = reinterpret_cast<const float *>(ln ? row1 : row0) + j;
          const float s0 = p[0], s1 = p[1], s2 = p[2];
          const unsigned lo2 = lo + sf_lo;        // the next raster pixel: same window, or one sample on (sf <= 0.5)
          const bool cy = lo2 < lo;
          const float a2 = cy ? s1 : s0, b2 = cy ? s2 : s1;
          // 2^32 ((1-t) a + t b) = U a + T b with T = lo and U = ~lo EACH converted from its own integer (complements to
          // 2^32 - 1): the samples are magnitudes (>= 0), so both terms are non-negative and carry a relative error of 2^-24
          // whatever t is.  fma(t, b - a, a) rounds b - a, and a - t a with the ROUNDED t is off by up to 2^-25 a: either is
          // many ulp of a pixel much smaller than one of its samples (white noise: 1.4e-6).  The factor 2^-32 rides in the
          // row weights of the last blend, so the precise weights cost one instruction per tap.
          const float T0 = (float)lo, U0 = (float)~lo, T1 = (float)lo2, U1 = (float)~lo2;
          R[u][2 * ln] = __fmaf_rn(T0, s1, __fmul_rn(U0, s0));
          R[u][2 * ln + 1] = __fmaf_rn(T1, b2, __fmul_rn(U1, a2));
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = cb + u * CSTEP;
        const float top = __fmaf_rn(dxw[u], R[u][1], __fmul_rn(uxw[u], R[u][0]));
        const float bot = __fmaf_rn(dxw[u], R[u][3], __fmul_rn(uxw[u], R[u][2]));
        const float v = __fmaf_rn(dys, bot, __fmul_rn(uys, top));   // (dys, uys carry the taps' 2^-32)
        const bool colok = c < cend;
        if (live && colok) o[(size_t)c * q.h_out] = v;
        vv[u] = (live && colok) ? v : 0.0f;
        if (PSUM) racc = __fadd_rn(racc, vv[u]);
      }
      if (PSUM) {
        const float tot = wave_sum4(vv[0], vv[1], vv[2], vv[3]);   // lane 15: column u = 0, 31: u = 2, 47: u = 1, 63: u = 3
        const int uu = lane == 15 ? 0 : lane == 31 ? 2 : lane == 47 ? 1 : 3;
        const int c = cb + uu * CSTEP;
        if ((lane & 15) == 15 && c < cend) pcol[c - c0] = tot;
      }
    }
  } else {
  for (int c = c0 + wave; c < cend; c += NT / 64) {
    const int ct = c - c0;
    float v;
    {
    const double dx = cdx[ct];
    float R00, R01, R10, R11;
    if (EXACT) {
      const unsigned kx = (unsigned)ckx[ct];
      const float *f0 = reinterpret_cast<const float *>(row0), *f1 = reinterpret_cast<const float *>(row1);
      R00 = raster_tap_exact(ax1, same1, b0 + kx, f0, kf0);
      R01 = raster_tap_exact(ax1, same1, b0 + kx + 1u, f0, kf0);
      R10 = raster_tap_exact(ax1, same1, b1 + kx, f1, kf1);
      R11 = raster_tap_exact(ax1, same1, b1 + kx + 1u, f1, kf1);
    } else {
      const double x00 = xrow + cxs[ct];
      R00 = raster_tap_fast<MODE>(fmax(x00, 0.0), row0, kf0);
      R01 = raster_tap_fast<MODE>(fmax(x00 + ax1.sf, 0.0), row0, kf0);
      R10 = raster_tap_fast<MODE>(x00 + xline, row1, kf1);           // line ky+1 >= 1: coordinate > 0
      R11 = raster_tap_fast<MODE>(x00 + xline + ax1.sf, row1, kf1);
    }
    if (EXACT) {
      // first dimension (lines) outermost: wy0*(wx0*a00 + wx1*a01) + wy1*(wx0*a10 + wx1*a11)
      const double wx0 = 1.0 - dx, wy0 = 1.0 - dy;
      const double top = __dadd_rn(__dmul_rn(wx0, (double)R00), __dmul_rn(dx, (double)R01));
      const double bot = __dadd_rn(__dmul_rn(wx0, (double)R10), __dmul_rn(dx, (double)R11));
      v = (float)__dadd_rn(__dmul_rn(wy0, top), __dmul_rn(dy, bot));
    } else {
      const double top = fma(dx, (double)R01 - (double)R00, (double)R00);
      const double bot = fma(dx, (double)R11 - (double)R10, (double)R10);
      v = (float)fma(dy, bot - top, top);
    }
    }
    if (live) o[(size_t)c * q.h_out] = v;
    if (COLSUM) colT[ct * 65 + lane] = v;
    if (PSUM) {
      const float m = live ? v : 0.0f;
      racc = __fadd_rn(racc, m);
      const float tot = wave_sum63(m);
      if (lane == 63) pcol[ct] = tot;
    }
  }
  }
  }
  if (PSUM) {
    prow[wave * 64 + lane] = racc;
    __syncthreads();
    float *pr = q.proj + (size_t)f * q.proj_stride;
    const int nrb = (q.h_out + 63) >> 6;
    if (tile_idx == 0 && tid < 2) q.keys[(size_t)f * 2 + tid] = 0ull;
    if (tid < 64 && r0 + tid < q.h_out) {
      float a = prow[tid];
#pragma unroll
      for (int w2 = 1; w2 < NT / 64; ++w2) a = __fadd_rn(a, prow[w2 * 64 + tid]);
      pr[(size_t)nrb * q.w_out + (size_t)tc * q.h_out + r0 + tid] = a;
    }
    if (tid < q.TC && c0 + tid < q.w_out) pr[(size_t)tr * q.w_out + c0 + tid] = pcol[tid];
  }
  if (COLSUM) {
    __syncthreads();
    const int nval = min(64, q.h_out - r0), c = c0 + tid;
    if (tid < q.TC && c < q.w_out) {
      float t = 0.0f;
      const float *col = colT + tid * 65;
      for (int rr = 0; rr < nval; ++rr) t = __fadd_rn(t, col[rr]);
      colpart[c] = t;
    }
  }
}

}  // namespace tsdr
