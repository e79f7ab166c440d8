// resample.hip -- Resampler.jl on gfx950.
//
//   sig_to_image (Resampler.jl:117-122)     k_raster_fast (FAST) / k_raster_tile (EXACT)   IQ/|IQ| -> column-major (y_t,x_t) raster
//   ... |> downgradeImage (:124-126)        <DOWN> of either: the same launch also emits the 600x800 image
//   sig_to_image |> downgradeImage          k_raster_fast with out == null (FAST) / k_down_fused: no raster in HBM
//   downgradeImage / imresize(image,size)   k_resize2d
//   imresize(sig,n)                         k_resize1d
//   naiveResampler (:103-110)               k_naive
//
// Two arithmetic modes (tsdr_precision):
//   EXACT  mirrors ImageTransformations.imresize! / Interpolations BSpline(Linear()) as restated in
//          oracle/tempest_oracle.c: f64 source coordinate sf*i+off (two roundings), f64 weights, one rounding
//          to f32 per value.  Bit-identical to the oracle.
//   FAST   evaluates the SAME coordinate exactly as a rational, x0(i) = ((2i+1)S - P) / 2P (0-based), carried
//          along a line with integer adds (quotient k, remainder r).  The blend is a + r*slope with the slope
//          (b - a)/2P staged per sample: as hi + lo f32 halves and two f32 FMAs when 2P < 2^24 (r is then an exact
//          f32 integer), else in f64 with one f64 FMA.  Within 1 ulp of EXACT either way.
//
// Layout: the raster is column-major (y_t,x_t): element (line l, pixel p) at p*y_t + l, so a wavefront owns 64
// consecutive LINES of one pixel column and its store is one contiguous 256-byte segment.  Source samples of a
// tile (64 lines x 128 pixels) are staged once in LDS ([line][sample], odd pitch): HBM sees every IQ sample once
// per tile row in 128-byte runs and |IQ| is evaluated once per staged sample, not once per pixel.  Workgroups are
// ordered so that the tiles stacked over one pixel strip run back to back on ONE XCD: the 256-byte segments of
// vertically adjacent tiles share 128-byte lines, which then merge in that XCD's L2 instead of reaching HBM as
// two partial writes.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "down_fused.h"
#include "sync_layout.h"

namespace tsdr {

template <bool CPLX, bool EXACT>
__device__ inline float load_sample(const float *__restrict__ src, unsigned k, const IqFmt &iqf) {
  if (CPLX) {
    const float2 z = ld_iq(src, k, iqf);
    return abs_iq<EXACT>(z.x, z.y);
  }
  return src[k];
}

// ---- FAST coordinate: exact rational, incremental ---------------------------------------------------------
struct FastAx {
  unsigned S, P, D;       // D = 2P
  unsigned qstep, rstep;  // 2S = qstep*D + rstep : advance of (k, r) per output sample
  double invDd;           // 1/D
};

static inline FastAx fast_axis(size_t S, size_t P) {
  FastAx f;
  f.S = (unsigned)S; f.P = (unsigned)P; f.D = (unsigned)(2 * P);
  f.qstep = (unsigned)((2 * S) / (2 * P));
  f.rstep = (unsigned)((2 * S) % (2 * P));
  f.invDd = 1.0 / (double)(2 * P);
  return f;
}

// floor division of num = (2*flat+1)*S - P by D (num may be negative for the first pixels of a frame)
__device__ inline void fast_pos(const FastAx &f, unsigned flat, int &k, unsigned &r) {
  const long long num = (long long)(2ull * flat + 1ull) * (long long)f.S - (long long)f.P;
  long long q = (long long)floor((double)num * f.invDd);
  long long rem = num - q * (long long)f.D;
  if (rem < 0) { rem += f.D; q -= 1; }
  else if (rem >= (long long)f.D) { rem -= f.D; q += 1; }
  k = (int)q;
  r = (unsigned)rem;
}

// ------------------------------------------------------------------------------------------------------------
// k_raster_tile (EXACT mode): one tile of 64 lines x TP pixels per workgroup (TP = 128 whenever the staged span fits).
// DOWN: tiles overlap by one line / one pixel (63 x TP-1 owned); the tile's raster values are kept in LDS and
// the 600x800 output pixels whose top-left tap falls in the owned area are produced by the same workgroup.
// ------------------------------------------------------------------------------------------------------------
struct TileParams {
  unsigned S;
  int y_t, x_t;
  int TP, W;            // pixels per tile, LDS row capacity (samples)
  int tiles_l, tiles_p; // tiles per frame
  int frames;
  int own_l, own_p;     // owned lines / pixels per tile (64/TP, or 63/TP-1 with DOWN)
  int h_out, w_out;     // DOWN only
  int NR, NC;           // DOWN: candidate output rows / columns per tile
  int lpl_log;          // log2(lanes per line) of the staging loop
  int cs;               // k_raster_fast: consecutive samples per staging lane (<= 4)
  int xcd_group;        // neighbouring pixel strips dealt to the same XCD
  int xcd_group_log;    // log2(xcd_group)
  float inv_tiles_p;    // 1/tiles_p (unit -> frame, strip without an integer division)
  RsAxis ax, ay, axx;   // sig->raster, raster lines->rows, raster pixels->columns (host-computed)
  double inv_sfy, inv_sfx;
  float *proj;          // k_raster_fast<DOWN>: projection partial sums of the (h_out, w_out) images, or null
  size_t proj_stride;   // floats per frame: colpart[tiles_l][w_out] | rowpart[tiles_p][h_out]
  unsigned long long *keys;  // with proj: two vsync argmax keys per frame, cleared here for k_beta's atomicMax
  IqFmt iqf;            // CPLX input: ComplexF32 or int16 pairs (common.h)
};

template <bool CPLX, bool DOWN>
__global__ __launch_bounds__(256) void k_raster_tile(const float *__restrict__ in, size_t in_stride, TileParams q,
                                                     float *__restrict__ out, size_t out_stride,
                                                     float *__restrict__ down, size_t down_stride) {
  extern __shared__ double lds_d[];
  const int Wp = q.W | 1;
  float *smp = reinterpret_cast<float *>(lds_d);  // staged samples [64][Wp] f32
  char *after = reinterpret_cast<char *>(lds_d) + (size_t)64 * Wp * 4;
  after += (16 - ((size_t)after & 15)) & 15;
  // candidate tables (DOWN): doubles first (alignment), then ints, then the raster tile
  double *rd = reinterpret_cast<double *>(after);
  double *cd = rd + (DOWN ? q.NR : 0);
  int *kfirst = reinterpret_cast<int *>(cd + (DOWN ? q.NC : 0));  // [64]
  int *vinfo = kfirst + 64;                                         // DOWN: {first valid row cand, #rows, first col cand, #cols}
  int *rk = vinfo + 4;
  int *ck = rk + (DOWN ? q.NR : 0);
  float *tile = reinterpret_cast<float *>(ck + (DOWN ? q.NC : 0));  // DOWN: [TP][65] raster values (pixel-major)

  // XCD-aware order (blocks are dealt round-robin over the 8 XCDs, so b & 7 labels the XCD group).  A unit is
  // (frame, pixel strip).  Each XCD gets a CONTIGUOUS range of units and walks it unit by unit, line tiles
  // fastest: (1) the tiles stacked over one strip run back to back on one XCD, so the partial 128-byte lines
  // at their seams merge in that L2; (2) neighbouring strips -- which read the same IQ lines -- run on the same
  // XCD close in time, so IQ is fetched into one L2 once instead of once per XCD (measured: FETCH_SIZE 3.5x
  // the IQ bytes with units dealt round-robin).
  // grid = (8, tiles_l, units per XCD): the linear block id is x + 8*(y + tiles_l*z), so x is the XCD slot, the
  // line tiles of a unit are consecutive on it, and no integer division is needed to decode the tile
  const unsigned xcd = blockIdx.x, ul = blockIdx.z;
  const int tl = (int)blockIdx.y;
  const unsigned U = (unsigned)(q.frames * q.tiles_p);
  const unsigned gl = (unsigned)q.xcd_group_log;
  const unsigned u = ((((ul >> gl) << 3) + xcd) << gl) + (ul & ((1u << gl) - 1u));
  if (u >= U) return;
  int f = (int)(((float)u + 0.5f) * q.inv_tiles_p);            // u / tiles_p (u < 2^20: exact)
  int tp = (int)u - f * q.tiles_p;
  if (tp < 0) { tp += q.tiles_p; --f; } else if (tp >= q.tiles_p) { tp -= q.tiles_p; ++f; }
  const int l0 = tl * q.own_l, p0 = tp * q.own_p;
  const float *src = in + (size_t)f * in_stride * (CPLX ? iq_floats(q.iqf) : 1);
  const unsigned P = (unsigned)q.y_t * (unsigned)q.x_t;
  const RsAxis ax = q.ax;
  const bool same = (q.S == P);
  const int tid = threadIdx.x;

  if (DOWN) {
    // candidate output rows/columns whose top-left tap may fall in this tile (monotone maps -> contiguous);
    // -1 marks "not owned by this tile"
    const RsAxis ay = q.ay, axx = q.axx;
    // wave 0 builds the row table (two rounds when NR > 64), wave 1 the column table; the valid entries of
    // each are contiguous, so one ballot per round yields first index and count
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    if (wv == 0) {
      const int rbase = max(0, (int)floor(((double)l0 + 0.5) * q.inv_sfy - 0.5) - 2);
      const bool last = (tl == q.tiles_l - 1);
      int first = 0x7fffffff, count = 0;
      for (int base = 0; base < q.NR; base += 64) {
        const int t = base + ln, r = rbase + t;
        int k = -1; double d = 0.0;
        if (t < q.NR && r < q.h_out) {
          k = (int)rs_pos(ay, (double)(r + 1), d);
          if (k < l0 || (!last && k >= l0 + q.own_l)) k = -1;
        }
        if (t < q.NR) { rk[t] = k < 0 ? -1 : ((r << 8) | (k - l0)); rd[t] = d; }  // row index, local line (< 64)
        const unsigned long long m = __ballot(k >= 0);
        if (m) { first = min(first, base + (int)__builtin_ctzll(m)); count += (int)__builtin_popcountll(m); }
      }
      if (ln == 0) { vinfo[0] = first; vinfo[1] = count; }
    } else if (wv == 1) {
      const int cbase = max(0, (int)floor(((double)p0 + 0.5) * q.inv_sfx - 0.5) - 2);
      const bool last = (tp == q.tiles_p - 1);
      int first = 0x7fffffff, count = 0;
      for (int base = 0; base < q.NC; base += 64) {
        const int t = base + ln, c = cbase + t;
        int k = -1; double d = 0.0;
        if (t < q.NC && c < q.w_out) {
          k = (int)rs_pos(axx, (double)(c + 1), d);
          if (k < p0 || (!last && k >= p0 + q.own_p)) k = -1;
        }
        if (t < q.NC) { ck[t] = k < 0 ? -1 : ((c << 8) | (k - p0)); cd[t] = d; }  // column index, local pixel (< 128)
        const unsigned long long m = __ballot(k >= 0);
        if (m) { first = min(first, base + (int)__builtin_ctzll(m)); count += (int)__builtin_popcountll(m); }
      }
      if (ln == 0) { vinfo[2] = first; vinfo[3] = count; }
    }
  }
  {  // stage: 2^lpl_log lanes per line (chosen on the host to waste the fewest lane slots); the loads of
     // up to four samples are issued back to back before any |IQ| math, so their latencies overlap.  Every
     // staging thread derives its line's first sample index itself (no barrier before staging; the candidate
     // tables above are only needed by the downgrade phase) and lane 0 of the line publishes it for the walk.
    const int lpl = 1 << q.lpl_log;
    const int sub = tid >> q.lpl_log, j0 = tid & (lpl - 1), nsub = 256 >> q.lpl_log;
    for (int r = sub; r < 64; r += nsub) {
      unsigned kf;
      {
        const int l = min(l0 + r, q.y_t - 1);
        const unsigned flat = (unsigned)l * (unsigned)q.x_t + (unsigned)p0;
        double d;
        const int k = (int)rs_pos(ax, (double)(flat + 1u), d);
        kf = (unsigned)k;
        if (j0 == 0) kfirst[r] = k;
      }
      for (int jb = j0; jb < q.W; jb += 4 * lpl) {
        float re[4], im[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = jb + u * lpl;
          const unsigned k = min(kf + (unsigned)min(j, q.W - 1), q.S - 1u);
          if (CPLX) { const float2 z = ld_iq(src, k, q.iqf); re[u] = z.x; im[u] = z.y; }
          else { re[u] = src[k]; im[u] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = jb + u * lpl;
          if (j < q.W) {
            smp[r * Wp + j] = CPLX ? abs_iq<true>(re[u], im[u]) : re[u];
          }
        }
      }
    }
  }
  __syncthreads();
  {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (uniform: the pixel loop's bounds are scalar)
    const int l = l0 + lane;
    const int pw = q.TP >> 2;
    const int pbeg = p0 + wave * pw, pend = min(pbeg + pw, q.x_t);
    // Overlapping (halo) lines/pixels of DOWN tiles are written by both neighbours: the values are
    // bit-identical (same flat index, exact integer walk), so the duplicate store is benign and the loop
    // needs no per-pixel predicate.  Lines past the frame are masked once, here.
    if (l < q.y_t && pbeg < pend) {
      const int kf = kfirst[lane];
      float *o = out ? out + (size_t)f * out_stride + (size_t)l + (size_t)pbeg * q.y_t : nullptr;
      float *trow = DOWN ? tile + (pbeg - p0) * 65 + lane : nullptr;
      const unsigned flat0 = (unsigned)l * (unsigned)q.x_t + (unsigned)pbeg;
      const float *row = smp + lane * Wp;
      // Clamps of the source coordinate can only act on the first and last ceil(1 / sf) pixels of a frame (x < 1 or x >= n_in):
      // every other tile -- all but two per frame -- evaluates the position without them (rs_pos_inner: same bits, four f64
      // instructions fewer of the ~17 a pixel costs; the kernel is f64-issue-bound)
      const double margin = 1.0 / ax.sf + 4.0;
      const double fmin = (double)l0 * (double)q.x_t + (double)p0;                                            // smallest flat index of the tile
      const double fmax = (double)min(l0 + 63, q.y_t - 1) * (double)q.x_t + (double)min(p0 + q.TP, q.x_t);   // one past its largest
      const bool edge = same || fmin < margin || fmax + margin >= (double)P;
      const bool has_out = out != nullptr;   // (uniform: `o` is a per-lane pointer, its null test a per-lane branch)
      if (!edge) {
        // UNR pixels per round: their positions first, then their LDS reads, then the blends (one pixel at a time the loop waits
        // out the LDS latency behind every position).  The 1-based flat index is carried as an f64 integer -- + 1.0 is exact,
        // and v_cvt_f64_u32 costs two f64 additions (tools/ubench/valu_rates.hip); the kernel is VALU-issue-bound
        constexpr int UNR = 2;   // (4: the same; 8: slower)
        int p = pbeg;
        double i1 = (double)(flat0 + 1u);
        for (; p + UNR <= pend; p += UNR, i1 += (double)UNR) {
          double d[UNR]; int j[UNR]; float a[UNR], bb[UNR];
#pragma unroll
          for (int u = 0; u < UNR; ++u) j[u] = (int)rs_pos_inner(ax, u == 0 ? i1 : i1 + (double)u, d[u]) - kf;
#pragma unroll
          for (int u = 0; u < UNR; ++u) { a[u] = row[j[u]]; bb[u] = row[j[u] + 1]; }
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const float v = rs_blend(a[u], bb[u], d[u]);
            if (has_out) o[(size_t)u * (size_t)q.y_t] = v;
            if (DOWN) trow[u * 65] = v;
          }
          if (has_out) o += (size_t)UNR * (size_t)q.y_t;
          if (DOWN) trow += UNR * 65;
        }
        for (; p < pend; ++p, i1 += 1.0) {
          double d;
          const int j = (int)rs_pos_inner(ax, i1, d) - kf;
          const float v = rs_blend(row[j], row[j + 1], d);
          if (has_out) { *o = v; o += q.y_t; }
          if (DOWN) { *trow = v; trow += 65; }
        }
      } else {
      for (int p = pbeg; p < pend; ++p) {
        double d;
        const int j = (int)rs_pos(ax, (double)(flat0 + (unsigned)(p - pbeg) + 1u), d) - kf;
        const float a = row[j], bb = row[j + 1];
        const float v = same ? (d == 1.0 ? bb : a) : rs_blend(a, bb, d);
        if (o) { *o = v; o += q.y_t; }
        if (DOWN) { *trow = v; trow += 65; }
      }
      }
    }
  }
  if (!DOWN) return;
  __syncthreads();
  {
    // output pixels owned by this tile: the valid candidates are contiguous (monotone maps), so the
    // nvr x nvc grid is walked flat by all 256 threads, rows fastest (coalesced stores)
    float *dn = down + (size_t)f * down_stride;
    const int rfirst = vinfo[0], nvr = vinfo[1], cfirst = vinfo[2], nvc = vinfo[3];
    const int total = nvr * nvc;
    const float inv = nvr > 0 ? 1.0f / (float)nvr : 0.0f;
    for (int idx = tid; idx < total; idx += 256) {
      const int ci = (int)(((float)idx + 0.5f) * inv);  // idx / nvr, exact for idx < 2^16
      const int ri = idx - ci * nvr;
      const int rkv = rk[rfirst + ri], ckv = ck[cfirst + ci];
      const int r = rkv >> 8, ly = rkv & 255, c = ckv >> 8, lx = ckv & 255;
      const double dy = rd[rfirst + ri], dx = cd[cfirst + ci];
      const float *t0 = tile + lx * 65 + ly;
      const float R00 = t0[0], R10 = t0[1], R01 = t0[65], R11 = t0[66];
      const double wx0 = 1.0 - dx, wy0 = 1.0 - dy;
      const double top = __dadd_rn(__dmul_rn(wx0, (double)R00), __dmul_rn(dx, (double)R01));
      const double bot = __dadd_rn(__dmul_rn(wx0, (double)R10), __dmul_rn(dx, (double)R11));
      const float v = (float)__dadd_rn(__dmul_rn(wy0, top), __dmul_rn(dy, bot));
      dn[(size_t)c * q.h_out + r] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_raster_fast: the FAST-mode tile kernel (same tiling, launch order and coordinate arithmetic as
// k_raster_tile, which keeps the EXACT mode).  What is different, and why: counters on the C2 buffer show the
// tile kernel issue-bound on VALU instructions (~550 per thread, only a third of them in the pixel walk) and
// held to five workgroups per CU by LDS.  So here
//   * the source position of a tile / line / wave segment is advanced from the frame origin with 32-bit
//     multiply-adds and a float-reciprocal quotient (adv32) instead of two 64-bit/f64 floor divisions per thread;
//   * a staging thread owns CONTIGUOUS samples of its line and forms the slopes in registers, so the staged
//     {a, slope} records are written once -- no second pass over LDS and one barrier less;
//   * the downgraded image is produced INSIDE the walk, from registers: a lane walks one raster line, so the four
//     taps of an output pixel are this lane's previous and current pixel and the same two of the lane below
//     (one ds_bpermute each).  Which pixel steps complete an output column is a wave-uniform bit mask; which
//     lanes own an output row is a per-lane flag.  The 16.6 KiB raster tile in LDS, its store per pixel, the
//     separate downgrade loop and its barrier are gone, and eight workgroups fit per CU.
// The inverse maps (raster line -> output row, raster pixel -> output column) need source/destination ratios
// strictly above 1 on both axes, so that a line (pixel) is the top-left tap of at most one row (column).
// ------------------------------------------------------------------------------------------------------------
struct FastInc {
  unsigned qL, rL;    // advance of (k, r) per raster line       : 2*x_t*S       = qL*D + rL
  unsigned qTL, rTL;  // per tile row (own_l lines)
  unsigned qTP, rTP;  // per tile column (own_p pixels)
  int k00;            // position of pixel 0 of a frame: floor((S - P)/D) ...
  unsigned r00;       // ... and remainder
  float invD;
};

static inline FastInc fast_inc(size_t S, size_t P, int x_t, int own_l, int own_p) {
  FastInc n;
  const long long D = 2 * (long long)P;
  auto split = [&](long long delta, unsigned &q, unsigned &r) { q = (unsigned)(delta / D); r = (unsigned)(delta % D); };
  split(2LL * x_t * (long long)S, n.qL, n.rL);
  split(2LL * x_t * (long long)S * own_l, n.qTL, n.rTL);
  split(2LL * (long long)S * own_p, n.qTP, n.rTP);
  const long long num = (long long)S - (long long)P;
  long long q = num / D;
  if (num % D < 0) --q;
  n.k00 = (int)q;
  n.r00 = (unsigned)(num - q * D);
  n.invD = 1.0f / (float)D;
  return n;
}

// (k, r) += m*(q, rr) with r kept in [0, D); needs r + m*rr < 2^32 (host-checked) and D < 2^24
__device__ inline void adv32(int &k, unsigned &r, unsigned m, unsigned q, unsigned rr, unsigned D, float invD) {
  const unsigned t = r + m * rr;
  unsigned c = (unsigned)((float)t * invD);  // floor(t/D) within +-1
  int rem = (int)(t - c * D);
  if (rem < 0) { rem += (int)D; --c; }
  else if ((unsigned)rem >= D) { rem -= (int)D; ++c; }
  k += (int)(m * q + c);
  r = (unsigned)rem;
}

// output row/column whose top-left tap is source index l (-1: none); sf > 1 makes it unique.  The first guess is
// the smallest c with sf*(c + 0.5) - 0.5 >= l; rs_pos decides, and a miss by rounding moves one step.
__device__ inline int inv_tap(const RsAxis &a, double inv_sf, int l, int n_out, double &d) {
  int c = (int)ceil(((double)l + 0.5) * inv_sf - 0.5);
  c = min(max(c, 0), n_out - 1);
  int k = (int)rs_pos(a, (double)(c + 1), d);
  if (k == l) return c;
  c += k > l ? -1 : 1;
  if (c < 0 || c >= n_out) return -1;
  k = (int)rs_pos(a, (double)(c + 1), d);
  return k == l ? c : -1;
}

struct DownInfo {   // per-lane / per-wave state of the in-walk downgrade
  unsigned long long colmask;  // bit j: the segment pixel j is the left tap of an output column
  int ccol;                    // lane j: that column
  float cdx; double cdxd;      // lane j: its weight (F32W / f64 walk)
  int rrow;                    // this lane's output row (-1: none)
  float rdy; double rdyd;
  int below;                   // ds_bpermute address of the lane below
  float *dn;                   // frame base of the (h_out, w_out) image
  int h_out;
  bool proj;                   // also accumulate the image's projection sums (wave-uniform)
  float racc;                  // sum of this lane's output row over the wave's output columns, in column order
  float csum;                  // lane j: sum over the wave's output rows of the column whose left tap is pixel j
  int lane;
};


// dword store with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (the compiler keeps
// emitting a 64-bit VGPR address, i.e. a 64-bit VALU add per store, for this pattern)
__device__ inline void store_saddr(float *base_uniform, unsigned lane_off_bytes, float v) {
  asm volatile("global_store_dword %0, %1, %2" : : "v"(lane_off_bytes), "v"(v), "s"(base_uniform) : "memory");
}

// One output column: this lane's line gives the upper blend along the pixel axis; the lower one is the same
// quantity of the lane below (one ds_bpermute), so it is bit-identical to what that lane computes for itself.
template <bool F32W>
__device__ inline void down_event(DownInfo &di, int j, float R00, float R01) {
  const int c = __builtin_amdgcn_readlane(di.ccol, j);
  // The blend runs in f64 in both walks (rounded once to f32, like the oracle's): f32 blends measured 5 ulp on
  // white-noise input, where neighbouring taps differ by their own size, against 3 ulp this way.
  float v;
  {
    const long long bits = __double_as_longlong(di.cdxd);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffLL), j), hi = __builtin_amdgcn_readlane((int)(bits >> 32), j);
    const double dx = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    const double top = fma(dx, (double)R01 - (double)R00, (double)R00);
    const long long tb = __double_as_longlong(top);
    const int blo = __builtin_amdgcn_ds_bpermute(di.below, (int)(tb & 0xffffffffLL));
    const int bhi = __builtin_amdgcn_ds_bpermute(di.below, (int)(tb >> 32));
    const double bot = __longlong_as_double(((long long)bhi << 32) | (unsigned)blo);
    v = (float)fma(di.rdyd, bot - top, top);
  }
  if (di.rrow >= 0) store_saddr(di.dn + (size_t)c * di.h_out, (unsigned)di.rrow * 4u, v);
  if (di.proj) {
    // projection sums of the (h_out, w_out) image, formed where its pixels are: this lane's row gains the pixel
    // (columns arrive in ascending order), and the column's sum over the wave's rows is one fixed-order DPP tree,
    // parked in lane j until the walk is over
    const float m = di.rrow >= 0 ? v : 0.0f;
    di.racc = __fadd_rn(di.racc, m);
    const float tot = wave_sum63(m);
    const float t63 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 63));
    di.csum = di.lane == j ? t63 : di.csum;
  }
}

// The common case of a lane's walk: a full segment of PW pixels in chunks of CH, each chunk unrolled -- the (k, r)
// advance, the LDS read one pixel ahead, two FMAs and the store, with no branch and no register shuffling inside
// a chunk.  The chunk's values stay in registers for the output columns whose left tap they are (the scan of
// the wave-uniform column mask is a handful of scalar bit tests per chunk).  Chunking rather than unrolling the
// whole segment keeps the kernel at 64 VGPRs, i.e. eight waves per SIMD.
// REC4 (F32W only): the row holds plain samples; a pixel is ((D - r) a + r b) / D with both weights exact integers in f32 --
// the convex form, so that each term carries a relative error whatever r is (see DM_FAST_FX in down_fused.h) -- four
// instructions instead of two, for a quarter of the LDS: what lets C3's wide tiles (39 samples x 127 lines) share a CU.
// (round 5: the division by D is in the staged samples -- a' = a / D once per sample instead of once per pixel, 0.115 samples per
// pixel at C2 -- so a pixel is three instructions: (D - r) a' + r b')
__device__ __forceinline__ float rec4_pixel(float ref, float Df, float2 s) {
  return __fmaf_rn(ref, s.y, __fmul_rn(__fsub_rn(Df, ref), s.x));
}
// (the integer walk, D >= 2^24: the weights are conversions of the two integers -- relative error 2^-24 each, which is all the
// convex form needs)
__device__ __forceinline__ float rec4_pixel_u(unsigned r, unsigned D, float2 s) {
  return __fmaf_rn((float)r, s.y, __fmul_rn((float)(D - r), s.x));
}
__device__ __forceinline__ float2 rec4_read(const float *row, int kk) { return make_float2(row[kk], row[kk + 1]); }

template <bool F32W, bool OUT, bool DOWNR, int PW, bool REC4 = false>
__device__ inline void fast_walk_full(const FastAx &fa, const void *__restrict__ rowv, int kk, unsigned r, bool extra,
                                      float *__restrict__ ob, int loff, size_t ostride, DownInfo &di, float invD = 0.f) {
  // ob is wave-uniform (first pixel of the segment, line 0 of the frame), loff the lane's line: the store then
  // takes a scalar base advanced by scalar adds and a fixed VGPR offset -- no per-pixel VALU address math
  // (eight per chunk where the f32 walk has 32-pixel segments -- C2: 118.6-121.5 -> 111.7-113.0 us for the launch on one box;
  // C3 (16-pixel segments) and C5 (integer walk) measured 2 % better with four, 16 was no better than 8)
  constexpr int CH = (F32W && PW >= 32) ? 8 : PW >= 4 ? 4 : PW;
  if (OUT) {
    // (ob IS wave-uniform; saying so once keeps it, and everything added to it below, in scalar registers whatever the
    // register allocator does with the values it was computed from -- store_saddr's "s" operand cannot take a VGPR pair)
    const unsigned long long b = reinterpret_cast<unsigned long long>(ob);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    ob = reinterpret_cast<float *>(((unsigned long long)hi << 32) | (unsigned long long)lo);
  }
  const float4 *row4 = reinterpret_cast<const float4 *>(rowv);
  const double2 *row2 = reinterpret_cast<const double2 *>(rowv);
  const float rstepf = (float)fa.rstep, Df = (float)fa.D;
  float rf = (float)r;
  const float *row1 = reinterpret_cast<const float *>(rowv);
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  double2 s2 = make_double2(0.0, 0.0);
  float2 s1 = make_float2(0.f, 0.f);
  if (REC4) s1 = rec4_read(row1, kk); else if (F32W) s4 = row4[kk]; else s2 = row2[kk];
  float last = 0.f;  // last pixel of the previous chunk
  unsigned long long mm = di.colmask << 1;  // column mask aligned to the chunks: bit j+1 = column j (wave-uniform)
  for (int cb = 0; cb < PW; cb += CH) {
    float v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const float ref = rf;
      const unsigned reu = r;
      if (F32W) {
        const float r2 = rf + rstepf;
        const bool c = r2 >= Df;
        kk += (int)fa.qstep + (c ? 1 : 0);
        rf = c ? r2 - Df : r2;
      } else {
        const unsigned r2 = r + fa.rstep;
        const bool c = r2 >= fa.D;
        kk += (int)fa.qstep + (c ? 1 : 0);
        r = c ? r2 - fa.D : r2;
      }
      float4 n4 = s4; double2 n2 = s2; float2 n1 = s1;
      if (REC4) n1 = rec4_read(row1, kk); else if (F32W) n4 = row4[kk]; else n2 = row2[kk];  // next pixel's record; in range by the W bound
      v[i] = REC4 ? (F32W ? rec4_pixel(ref, Df, s1) : rec4_pixel_u(reu, fa.D, s1))
                  : F32W ? fmaf(ref, s4.z, fmaf(ref, s4.y, s4.x)) : (float)fma((double)reu, s2.y, s2.x);
      if (OUT) { store_saddr(ob, (unsigned)loff * 4u, v[i]); ob += ostride; }
      s4 = n4; s2 = n2; s1 = n1;
    }
    if (DOWNR) {
      const unsigned bits = (unsigned)mm;  // bit j: column cb-1+j is completed by pixel cb+j
      mm >>= CH;
      if (bits & 1u) down_event<F32W>(di, cb - 1, last, v[0]);
#pragma unroll
      for (int j = 1; j < CH; ++j)
        if ((bits >> j) & 1u) down_event<F32W>(di, cb - 1 + j, v[j - 1], v[j]);
      last = v[CH - 1];
    }
  }
  if (DOWNR && extra && ((di.colmask >> (PW - 1)) & 1ull)) {
    const float ve = REC4 ? (F32W ? rec4_pixel(rf, Df, s1) : rec4_pixel_u(r, fa.D, s1))
                          : F32W ? fmaf(rf, s4.z, fmaf(rf, s4.y, s4.x)) : (float)fma((double)r, s2.y, s2.x);
    down_event<F32W>(di, PW - 1, last, ve);
  }
}

// Edge cases (segment cut by the frame end, first pixels of a frame where x0 < 0): the same walk as a plain loop.
// n_own pixels are stored; one more is evaluated (not stored) when `extra`.
template <bool F32W, bool CLAMP, bool OUT, bool DOWNR, bool REC4 = false>
__device__ inline void fast_walk2(const FastAx &fa, const void *__restrict__ rowv, int kk, unsigned r, int n_own, bool extra,
                                  float *__restrict__ o, size_t ostride, DownInfo &di, float invD = 0.f) {
  const float4 *row4 = reinterpret_cast<const float4 *>(rowv);
  const double2 *row2 = reinterpret_cast<const double2 *>(rowv);
  const int n = n_own + (extra ? 1 : 0);
  const float rstepf = (float)fa.rstep, Df = (float)fa.D;
  float rf = (float)r;
  float prev = 0.f;
  for (int i = 0; i < n; ++i) {
    const int ks = CLAMP ? max(kk, 0) : kk;
    const float ref = CLAMP ? (kk < 0 ? 0.f : rf) : rf;
    const unsigned reu = CLAMP ? (kk < 0 ? 0u : r) : r;
    float v;
    if (REC4) {
      const float2 s1 = rec4_read(reinterpret_cast<const float *>(rowv), ks);
      v = F32W ? rec4_pixel(ref, Df, s1) : rec4_pixel_u(reu, fa.D, s1);
    }
    else if (F32W) { const float4 s4 = row4[ks]; v = fmaf(ref, s4.z, fmaf(ref, s4.y, s4.x)); }
    else { const double2 s2 = row2[ks]; v = (float)fma((double)reu, s2.y, s2.x); }
    if (OUT) { if (i < n_own && (!CLAMP || o)) { *o = v; o += ostride; } }
    if (DOWNR) { if (i > 0 && ((di.colmask >> (i - 1)) & 1ull)) down_event<F32W>(di, i - 1, prev, v); }
    prev = v;
    if (F32W) {
      const float r2 = rf + rstepf;
      const bool c = r2 >= Df;
      kk += (int)fa.qstep + (c ? 1 : 0);
      rf = c ? r2 - Df : r2;
    } else {
      const unsigned r2 = r + fa.rstep;
      const bool c = r2 >= fa.D;
      kk += (int)fa.qstep + (c ? 1 : 0);
      r = c ? r2 - fa.D : r2;
    }
  }
}

// OUT: the raster is written (out != null); a template parameter so that the raster-free launch is a different
// kernel symbol and the two show up separately in profiler statistics
// VW: wavefronts stacked vertically in one workgroup (256*VW threads: VW x 4 wavefronts of 64 lines x PW pixels,
// neighbours sharing one line with DOWN).  A wave's 256-byte column segment starts wherever (p*y_t + line)*4 falls,
// so its first and last 128-byte lines are shared with the segments above and below; when those belong to other
// workgroups the halves reach L2 ~20 us apart, longer than a line survives there under this write stream, and are
// written back as partial lines twice.  Stacking VW waves makes VW-1 of every VW seams internal to a workgroup.
// REC4: the staged lines are plain f32 samples (fast_walk_full) instead of {a, slope hi, slope lo} records
// IQF: the REC4 instantiations (the hot ones) exist once per input format, the others read it from the parameters
template <bool CPLX, bool F32W, bool DOWN, int PW, bool OUT, int VW, bool REC4 = false, int IQF = (REC4 ? IQF_CF32 : IQF_RT)>
__global__ __launch_bounds__(256 * VW, 8) void k_raster_fast(const float *__restrict__ in, size_t in_stride, TileParams q,
                                                     FastAx fa, FastInc fi, float *__restrict__ out, size_t out_stride,
                                                     float *__restrict__ down, size_t down_stride) {
  extern __shared__ double lds_d[];
  const int Wp = q.W | 1;
  float4 *smp4 = reinterpret_cast<float4 *>(lds_d);      // F32W : [NL][Wp] {a, slope hi, slope lo, -}
  double2 *smp2 = reinterpret_cast<double2 *>(lds_d);    // !F32W: [NL][Wp] {a, slope} in f64
  float *smp1 = reinterpret_cast<float *>(lds_d);        // REC4 : [NL][Wp] a
  constexpr int LSTEP = DOWN ? 63 : 64;              // line pitch of vertically stacked waves
  constexpr int NL = LSTEP * (VW - 1) + 64;            // lines of the tile
  char *after = reinterpret_cast<char *>(lds_d) + (REC4 ? (((size_t)NL * Wp * 4 + 15) & ~(size_t)15) : (size_t)NL * Wp * 16);
  double *rdyd = reinterpret_cast<double *>(after);      // DOWN: per-line row weight, per-pixel column weight
  double *cdxd = rdyd + (DOWN ? NL : 0);
  int *rrow = reinterpret_cast<int *>(cdxd + (DOWN ? q.TP + 1 : 0));
  int *ccol = rrow + (DOWN ? NL : 0);

  // (frame, strip, line tile) of this workgroup: same XCD-aware order as k_raster_tile
  const unsigned xcd = blockIdx.x, ul = blockIdx.z;
  const int tl = (int)blockIdx.y;
  const unsigned U = (unsigned)(q.frames * q.tiles_p);
  const unsigned gl = (unsigned)q.xcd_group_log;
  const unsigned u = ((((ul >> gl) << 3) + xcd) << gl) + (ul & ((1u << gl) - 1u));
  if (u >= U) return;
  int f = (int)(((float)u + 0.5f) * q.inv_tiles_p);
  int tp = (int)u - f * q.tiles_p;
  if (tp < 0) { tp += q.tiles_p; --f; } else if (tp >= q.tiles_p) { tp -= q.tiles_p; ++f; }
  const int l0 = tl * q.own_l, p0 = tp * q.own_p;
  const float *src = in + (size_t)f * in_stride * (CPLX ? iq_floats_as<IQF>(q.iqf) : 1);
  const int tid = threadIdx.x;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wave = wave_id & 3, wv = wave_id >> 2;  // horizontal segment, vertical position
  const int lbase = wv * LSTEP;                       // first line of this wave in the tile

  // position of (line l0, pixel p0); lines past the frame replicate the last one (same values, same addresses)
  int kb = fi.k00; unsigned rb = fi.r00;
  if (F32W) {
    adv32(kb, rb, (unsigned)tl, fi.qTL, fi.rTL, fa.D, fi.invD);
    adv32(kb, rb, (unsigned)tp, fi.qTP, fi.rTP, fa.D, fi.invD);
  }
  auto line_pos = [&](int line_in_tile, int &k, unsigned &r) {
    const int l = min(l0 + line_in_tile, q.y_t - 1);
    if (F32W) { k = kb; r = rb; adv32(k, r, (unsigned)(l - l0), fi.qL, fi.rL, fa.D, fi.invD); }
    else fast_pos(fa, (unsigned)l * (unsigned)q.x_t + (unsigned)p0, k, r);
  };

  {  // stage: 2^lpl_log lanes per line, each owning `cs` consecutive samples (cs <= 4) plus the one after them.
     // The loads of a thread's first line are issued before the output row/column tables are worked out (f64
     // arithmetic that needs no memory), so their latency is covered.
    const int lpl = 1 << q.lpl_log;
    const int sub = tid >> q.lpl_log, j0 = tid & (lpl - 1), nsub = (256 * VW) >> q.lpl_log;
    const int cs = q.cs;
    const int jb = j0 * cs;
    auto issue = [&](int r, float (&re)[5], float (&im)[5]) {
      int k; unsigned rr;
      line_pos(r, k, rr);
      const unsigned kf = (unsigned)max(k, 0);
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        re[t] = 0.f; im[t] = 0.f;
        if (REC4 ? t < cs : t <= cs) {
          const unsigned ks = min(kf + (unsigned)min(jb + t, q.W - 1), q.S - 1u);
          if (CPLX) { const float2 z = ld_iq_as<IQF>(src, ks, q.iqf); re[t] = z.x; im[t] = z.y; }
          else re[t] = src[ks];
        }
      }
    };
    auto consume = [&](int r, const float (&re)[5], const float (&im)[5]) {
      float a[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) a[t] = (CPLX && (REC4 ? t < cs : t <= cs)) ? abs_iq<false>(re[t], im[t]) : re[t];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = jb + t;
        if (REC4) {
          if (t < cs && j < q.W) smp1[r * Wp + j] = __fmul_rn(a[t], fi.invD);   // a' = a / D (rec4_pixel)
        } else if (t < cs && j < q.W) {
          const double sl = ((double)a[t + 1] - (double)a[t]) * fa.invDd;
          if (F32W) {
            const float hi = (float)sl;
            smp4[r * Wp + j] = make_float4(a[t], hi, (float)(sl - (double)hi), 0.f);
          } else {
            smp2[r * Wp + j] = make_double2((double)a[t], sl);
          }
        }
      }
    };
    float re[5], im[5];
    if (sub < NL) issue(sub, re, im);
    if (DOWN) {
      // first NL threads: output row of every line; the rest: output column of every pixel of the tile
      if (tid < NL) {
        const int l = l0 + tid;
        const bool mine = tid < q.own_l || tl == q.tiles_l - 1;
        double d = 0.0;
        const int r = (mine && l < q.y_t) ? inv_tap(q.ay, q.inv_sfy, l, q.h_out, d) : -1;
        rrow[tid] = r; rdyd[tid] = d;
      } else {
        for (int j = tid - NL; j <= q.TP; j += 256 * VW - NL) {
          const int p = p0 + j;
          const bool mine = j < q.TP && (j < q.own_p || tp == q.tiles_p - 1);
          double d = 0.0;
          const int c = (mine && p < q.x_t) ? inv_tap(q.axx, q.inv_sfx, p, q.w_out, d) : -1;
          ccol[j] = c; cdxd[j] = d;
        }
      }
    }
    if (sub < NL) consume(sub, re, im);
    // (Round 4, tried: the remaining lines' loads three lines at a time -- C3's tiles are four dependent round trips per
    // thread.  The kernel's store addressing (scalar base, "s" asm constraint) does not survive the extra register
    // pressure: the build fails, and with the base forced through readfirstlane the kernel faults.  Left as it was.)
    for (int r = sub + nsub; r < NL; r += nsub) {
      issue(r, re, im);
      consume(r, re, im);
    }
  }
  __syncthreads();
  DownInfo di{};
  di.rrow = -1;
  int my_col = -1;  // lane j: the output column whose left tap is pixel j of this wave's segment
  {
    constexpr int pw = PW;  // = TP/4
    const int pbeg = p0 + wave * pw;
    const int n_own = min(pw, q.x_t - pbeg);
    if (n_own <= 0 && !(DOWN && q.proj)) return;
    if (n_own > 0) {
    // one pixel past the segment (evaluated, not stored) lets the last owned pixel be a left tap
    const bool extra = DOWN && wave < 3 && pbeg + pw < q.x_t;
    int k; unsigned r;
    line_pos(lbase + lane, k, r);
    const int kf = max(k, 0);
    const long long num0 = (long long)(2ull * ((unsigned long long)l0 * q.x_t + p0) + 1ull) * (long long)fa.S - (long long)fa.P;
    if (F32W) adv32(k, r, (unsigned)(wave * pw), fa.qstep, fa.rstep, fa.D, fi.invD);
    else fast_pos(fa, (unsigned)min(l0 + lbase + lane, q.y_t - 1) * (unsigned)q.x_t + (unsigned)pbeg, k, r);
    const int l = min(l0 + lbase + lane, q.y_t - 1);
    float *o = OUT ? out + (size_t)f * out_stride + (size_t)l + (size_t)pbeg * q.y_t : nullptr;
    if (DOWN) {
      const int cj = lane <= pw ? wave * pw + lane : q.TP;  // entry TP is never a column
      di.ccol = ccol[cj];
      di.cdxd = cdxd[cj];
      di.cdx = (float)di.cdxd;
      di.colmask = __ballot(lane < pw && di.ccol >= 0);
      my_col = lane < pw ? di.ccol : -1;
      // lane 63 is the next wave's lane 0 unless this is the bottom wave of the tile
      di.rrow = (lane < 63 || wv == VW - 1) ? rrow[lbase + lane] : -1;
      di.rdyd = rdyd[lbase + lane];
      di.rdy = (float)di.rdyd;
      di.below = ((lane + 1) & 63) << 2;
      di.dn = down + (size_t)f * down_stride;
      di.h_out = q.h_out;
      di.proj = q.proj != nullptr;
      di.lane = lane;
    }
    const void *row = REC4 ? (const void *)(smp1 + (lbase + lane) * Wp)
                           : F32W ? (const void *)(smp4 + (lbase + lane) * Wp) : (const void *)(smp2 + (lbase + lane) * Wp);
    const int kk = k - kf;
    const bool full = n_own == PW && (!DOWN || extra || wave == 3);
    if (num0 < 0) fast_walk2<F32W, true, OUT, DOWN, REC4>(fa, row, kk, r, n_own, extra, o, (size_t)q.y_t, di, fi.invD);
    else if (!full) fast_walk2<F32W, false, OUT, DOWN, REC4>(fa, row, kk, r, n_own, extra, o, (size_t)q.y_t, di, fi.invD);
    else {
      float *ob = OUT ? out + (size_t)f * out_stride + (size_t)pbeg * q.y_t : nullptr;
      fast_walk_full<F32W, OUT, DOWN, PW, REC4>(fa, row, kk, r, extra, ob, l, (size_t)q.y_t, di, fi.invD);
    }
    }
  }
  if (DOWN && q.proj) {
    // Projection partial sums of this tile: the four wavefronts side by side are added left to right into one row
    // partial per output row (-> rowpart[strip][row]), the VW stacked ones top to bottom into one column partial
    // per output column (-> colpart[line tile][column]).  Every (strip, row) and (line tile, column) slot is
    // written by exactly one lane of one workgroup, so the buffer needs no clearing; k_fold adds the partials of
    // a row strip by strip and those of a column line tile by line tile.
    // (the staged samples are dead once every wavefront has finished its walk: their LDS is reused, so the sums cost
    // no LDS of their own -- at C3's sampling ratio 4 KiB more would have halved the workgroups per CU)
    __syncthreads();
    float *rp = reinterpret_cast<float *>(lds_d);            // [4*VW][64] row sums per wavefront
    float *cp = rp + 256 * VW;                               // [4*VW][64] column sums per wavefront
    rp[wave_id * 64 + lane] = di.racc;
    cp[wave_id * 64 + lane] = di.csum;
    __syncthreads();
    float *pr = q.proj + (size_t)f * q.proj_stride;
    if (tl == 0 && tp == 0 && tid < 2) q.keys[(size_t)f * 2 + tid] = 0ull;
    if (wave == 0 && di.rrow >= 0) {
      const float *x = rp + (wv * 4) * 64 + lane;
      const float a = __fadd_rn(__fadd_rn(__fadd_rn(x[0], x[64]), x[128]), x[192]);
      pr[(size_t)q.tiles_l * q.w_out + (size_t)tp * q.h_out + di.rrow] = a;
    }
    if (wv == 0 && my_col >= 0) {
      float a = cp[wave * 64 + lane];
#pragma unroll
      for (int v2 = 1; v2 < VW; ++v2) a = __fadd_rn(a, cp[(v2 * 4 + wave) * 64 + lane]);
      pr[(size_t)tl * q.w_out + my_col] = a;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_raster_fast4 (round 6; option "raster_v4", OFF: built, correct, measured, loses): the FAST raster + image + sums launch with
// FOUR consecutive raster lines per lane.  Measured on C2 (30 frames): 126-137 us against 112-121 us for k_raster_fast on the
// same boxes (eight wavefronts of 16 pixel columns: 145 us; five wavefronts per SIMD: 145 us; XCD grouping 1 / 2 / 4 / 8: no
// difference; conflict-free LDS planes: no difference).  By deletion: everything but the raster stores 79 us (k_raster_fast:
// 77), the image stores 13.6 us of that, the raster stores +57 us (k_raster_fast: +42): what the store-only skeleton promised
// (tools/ubench/store_pattern.hip: 81 against 130 us with "free" arithmetic) does not carry over to a kernel whose stores are
// paced by its own arithmetic.  Kept as the A/B, tested for identical rasters / images / indices.
//
// Why: the launch is bound by the NUMBER of 128-byte write requests, not by their bytes (tools/ubench/store_pattern.hip:
// every store shape fits ~45 G line requests per second, partial or full).  With one line per lane a wave-store is 256 bytes
// on an arbitrary 4-byte boundary = 3 requests, and a column event's image rows leave as ~136-byte runs = 2 more; with four
// lines per lane the wave-store is ONE dwordx4 per lane = 1024 contiguous bytes = 9 requests instead of 12, and a wavefront's
// 136 image rows of a column are compacted through LDS into contiguous 256-byte runs (6 requests instead of 8-9).
//
// Same arithmetic as k_raster_fast<REC4, F32W> (position advance, |IQ| / D staged as f32, pixel = (D - r) a' + r b'; image =
// f64 blends of f32 pixels, one rounding): rasters and images are bit-identical to that kernel's.  What differs is the
// association of the projection partial sums (one row partial per 128-pixel strip as before; one column partial per 255-line
// tile instead of per 127-line tile), which FAST mode leaves open (the sync guard decides close calls exactly).
//
// Tile = 256 staged lines (255 owned) x 128 pixels (127 owned); 256 threads = 4 wavefronts side by side, 32 pixel columns
// each; lane j owns tile lines 4j .. 4j+3.  The vertical neighbour of lines 4j .. 4j+2 is in the lane itself, of line 4j+3
// it is lane j+1's first line (ds_bpermute of its horizontal blend).
// ------------------------------------------------------------------------------------------------------------
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float_t;   // (a pointer the compiler KNOWS is LDS: ds_* instructions, not flat_*)

__device__ inline void store4_saddr(float *base_uniform, unsigned lane_off_bytes, v4f_t v) {
  asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(lane_off_bytes), "v"(v), "s"(base_uniform) : "memory");
}

// a pointer that IS wave-uniform, said so (keeps it in scalar registers for store_saddr's "s" operand)
__device__ __forceinline__ float *uniform_ptr(float *p) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return reinterpret_cast<float *>(((unsigned long long)hi << 32) | (unsigned long long)lo);
}

struct Down4 {   // per-lane / per-wave state of the in-walk downgrade, four lines per lane
  unsigned long long colmask;  // bit j: segment pixel j is the left tap of an output column
  int ccol; double cdxd;       // lane j: that column and its weight
  int rrow[4]; double rdyd[4]; // this lane's lines: output row (-1: none) and weight
  int below;                   // ds_bpermute address of the lane below
  float *dn; int h_out;
  bool proj;
  float racc[4];               // row sums of this lane's rows over the wave's columns, in column order
  float csum;                  // lane j: sum over the wave's rows of the column whose left tap is pixel j
  int lane;
  int R0, nrows;               // the wave's output rows: [R0, R0 + nrows), contiguous
  lds_float_t *rowbuf;         // this wave's LDS scratch: one column's rows in row order
};

__device__ __forceinline__ void down_event4(Down4 &di, int j, const float (&p0)[4], const float (&p1)[4]) {
  const int c = __builtin_amdgcn_readlane(di.ccol, j);
  const long long bits = __double_as_longlong(di.cdxd);
  const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffLL), j), hi = __builtin_amdgcn_readlane((int)(bits >> 32), j);
  const double dx = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  double top[5];
#pragma unroll
  for (int i = 0; i < 4; ++i) top[i] = fma(dx, (double)p1[i] - (double)p0[i], (double)p0[i]);
  {
    const long long tb = __double_as_longlong(top[0]);
    const int blo = __builtin_amdgcn_ds_bpermute(di.below, (int)(tb & 0xffffffffLL));
    const int bhi = __builtin_amdgcn_ds_bpermute(di.below, (int)(tb >> 32));
    top[4] = __longlong_as_double(((long long)bhi << 32) | (unsigned)blo);
  }
  float m = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = (float)fma(di.rdyd[i], top[i + 1] - top[i], top[i]);
    if (di.rrow[i] >= 0) di.rowbuf[di.rrow[i] - di.R0] = v;
    if (di.proj) {
      const float mv = di.rrow[i] >= 0 ? v : 0.0f;
      di.racc[i] = __fadd_rn(di.racc[i], mv);
      m = __fadd_rn(m, mv);
    }
  }
  // the column's rows, compacted: lane t stores row R0 + t (+ 64, + 128, ...) -- contiguous runs whatever the lines' ownership
  // pattern is.  (LDS operations of one wavefront complete in order: the reads below see the writes above; the fence keeps the
  // compiler from moving them across each other.)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float *dcol = uniform_ptr(di.dn + (size_t)c * di.h_out + di.R0);
  for (int t = di.lane; t < di.nrows; t += 64) {
    const float v = di.rowbuf[t];
    store_saddr(dcol, (unsigned)t * 4u, v);
  }
  if (di.proj) {
    const float tot = wave_sum63(m);
    const float t63 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 63));
    di.csum = di.lane == j ? t63 : di.csum;
  }
}

// The walk of one wavefront: npix pixel columns (+ one evaluated, not stored, when `extra`), four lines per lane.
// CLAMP: first pixels of a frame (x0 < 0); TAILT: the tile reaches past the last raster line (per-line store predicates).
template <bool CLAMP, bool TAILT>
__device__ __forceinline__ void walk4(const FastAx &fa, const float *const (&row)[4], int (&kk)[4], float (&rf)[4], int npix, bool extra,
                                      float *ob, unsigned loff_bytes, size_t ostride, int nvalid, Down4 &di) {
  {
    const unsigned long long b = reinterpret_cast<unsigned long long>(ob);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    ob = reinterpret_cast<float *>(((unsigned long long)hi << 32) | (unsigned long long)lo);
  }
  const float rstepf = (float)fa.rstep, Df = (float)fa.D;
  float prev[4] = {0.f, 0.f, 0.f, 0.f};
  const int n = npix + (extra ? 1 : 0);
  unsigned long long mm = di.colmask << 1;   // bit i: column (left tap i - 1) is completed by pixel i
  float2 s1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s1[i] = rec4_read(row[i], CLAMP ? max(kk[i], 0) : kk[i]);
#pragma unroll 1
  for (int pi = 0; pi < n; ++pi) {   // (not unrolled: four independent lines per step are the instruction-level parallelism; 17 unrolled steps spill)
    float cur[4];
    float2 nx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ref = CLAMP ? (kk[i] < 0 ? 0.f : rf[i]) : rf[i];
      const float r2 = rf[i] + rstepf;
      const bool cy = r2 >= Df;
      kk[i] += (int)fa.qstep + (cy ? 1 : 0);
      rf[i] = cy ? r2 - Df : r2;
      // the next pixel's pair, one step ahead (after the last pixel: a read one or two floats past the row's samples, inside
      // the workgroup's LDS, never used)
      nx[i] = rec4_read(row[i], CLAMP ? max(kk[i], 0) : kk[i]);
      cur[i] = rec4_pixel(ref, Df, s1[i]);
    }
    if (pi < npix) {
      if (!TAILT) {
        v4f_t v = {cur[0], cur[1], cur[2], cur[3]};
        store4_saddr(ob, loff_bytes, v);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nvalid) store_saddr(uniform_ptr(ob), loff_bytes + 4u * (unsigned)i, cur[i]);
      }
      ob += ostride;
    }
    if ((mm >> pi) & 1ull) down_event4(di, pi - 1, prev, cur);
#pragma unroll
    for (int i = 0; i < 4; ++i) { prev[i] = cur[i]; s1[i] = nx[i]; }
  }
}

// PW: pixel columns per wavefront (32: four wavefronts per 128-pixel tile; 16: eight -- twice as many, half as long: the launch
// is ~3 rounds of workgroups at PW = 32, and its last, nearly empty round costs a quarter of it)
template <int IQF, int PW>
__global__ __launch_bounds__(64 * (128 / PW), 4) void k_raster_fast4(   // (HIP: second argument = wavefronts per SIMD -> at most 128 VGPRs)
   const float *__restrict__ in, size_t in_stride, TileParams q, FastAx fa, FastInc fi,
                                                        float *__restrict__ out, size_t out_stride, float *__restrict__ down, size_t down_stride) {
  extern __shared__ double lds_d[];
  constexpr int NL = 256, NWV = 128 / PW, NT = 64 * NWV;
  const int Wp = q.W | 1;
  float *smp1 = reinterpret_cast<float *>(lds_d);                       // [NL][Wp] |IQ| / D
  char *after = reinterpret_cast<char *>(lds_d) + (((size_t)NL * Wp * 4 + 15) & ~(size_t)15);
  double *rdyd = reinterpret_cast<double *>(after);                     // per line: row weight
  double *cdxd = rdyd + NL;                                             // per pixel: column weight
  int *rrow = reinterpret_cast<int *>(cdxd + q.TP + 1);
  int *ccol = rrow + NL;
  float *rowbuf = reinterpret_cast<float *>(ccol + q.TP + 1);           // [NWV][NL]

  const unsigned xcd = blockIdx.x, ul = blockIdx.z;
  const int tl = (int)blockIdx.y;
  const unsigned U = (unsigned)(q.frames * q.tiles_p);
  const unsigned gl = (unsigned)q.xcd_group_log;
  const unsigned u = ((((ul >> gl) << 3) + xcd) << gl) + (ul & ((1u << gl) - 1u));
  if (u >= U) return;
  int f = (int)(((float)u + 0.5f) * q.inv_tiles_p);
  int tp = (int)u - f * q.tiles_p;
  if (tp < 0) { tp += q.tiles_p; --f; } else if (tp >= q.tiles_p) { tp -= q.tiles_p; ++f; }
  const int l0 = tl * q.own_l, p0 = tp * q.own_p;
  const float *src = in + (size_t)f * in_stride * iq_floats_as<IQF>(q.iqf);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

  int kb = fi.k00; unsigned rb = fi.r00;
  adv32(kb, rb, (unsigned)tl, fi.qTL, fi.rTL, fa.D, fi.invD);
  adv32(kb, rb, (unsigned)tp, fi.qTP, fi.rTP, fa.D, fi.invD);
  auto line_pos = [&](int line_in_tile, int &k, unsigned &r) {
    const int l = min(l0 + line_in_tile, q.y_t - 1);
    k = kb; r = rb;
    adv32(k, r, (unsigned)(l - l0), fi.qL, fi.rL, fa.D, fi.invD);
  };

  {  // stage: 2^lpl_log lanes per line, `cs` consecutive samples each; four lines' loads in flight per lane, the output
     // row / column tables (f64 arithmetic that needs no memory) worked out under the first batch
    const int lpl = 1 << q.lpl_log;
    const int sub = tid >> q.lpl_log, j0 = tid & (lpl - 1), nsub = NT >> q.lpl_log;
    const int cs = q.cs, jb = j0 * cs;
    float re[4][4], im[4][4];
    auto issue = [&](int r, float (&rre)[4], float (&rim)[4]) {
      int k; unsigned rr;
      line_pos(min(r, NL - 1), k, rr);
      const unsigned kf = (unsigned)max(k, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        rre[t] = 0.f; rim[t] = 0.f;
        if (t < cs) {
          const unsigned ks = min(kf + (unsigned)min(jb + t, q.W - 1), q.S - 1u);
          const float2 z = ld_iq_as<IQF>(src, ks, q.iqf);
          rre[t] = z.x; rim[t] = z.y;
        }
      }
    };
    auto consume = [&](int r, const float (&rre)[4], const float (&rim)[4]) {
      if (r >= NL) return;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = jb + t;
        // a' = a / D (rec4_pixel).  Tile line r lies in LDS row (r & 3) * 64 + (r >> 2): the four lines of a lane in four planes,
        // so that the 64 lanes of a wave-read are Wp (odd) floats apart -- conflict-free -- instead of 4 Wp
        if (t < cs && j < q.W) smp1[((r & 3) * 64 + (r >> 2)) * Wp + j] = __fmul_rn(abs_iq<false>(rre[t], rim[t]), fi.invD);
      }
    };
#pragma unroll
    for (int b = 0; b < 4; ++b) issue(sub + b * nsub, re[b], im[b]);
    {
      // the first NL threads: output row of their line; the last TP + 1 (the first, with 256 threads): output column of every pixel
      if (tid < NL) {
        const int l = l0 + tid;
        const bool mine = tid < q.own_l || tl == q.tiles_l - 1;
        double d = 0.0;
        const int r = (mine && l < q.y_t) ? inv_tap(q.ay, q.inv_sfy, l, q.h_out, d) : -1;
        rrow[tid] = r; rdyd[tid] = d;
      }
      const int ct = NT > NL ? tid - NL : tid;
      if (ct >= 0 && ct <= q.TP) {
        const int pp = p0 + ct;
        const bool minec = ct < q.TP && (ct < q.own_p || tp == q.tiles_p - 1);
        double d2 = 0.0;
        const int c = (minec && pp < q.x_t) ? inv_tap(q.axx, q.inv_sfx, pp, q.w_out, d2) : -1;
        ccol[ct] = c; cdxd[ct] = d2;
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) consume(sub + b * nsub, re[b], im[b]);
    for (int r0 = sub + 4 * nsub; r0 < NL; r0 += 4 * nsub) {
#pragma unroll
      for (int b = 0; b < 4; ++b) issue(r0 + b * nsub, re[b], im[b]);
#pragma unroll
      for (int b = 0; b < 4; ++b) consume(r0 + b * nsub, re[b], im[b]);
    }
  }
  __syncthreads();

  Down4 di{};
  int my_col = -1;
  {
    const int pbeg = p0 + wave * PW;
    const int n_own = min(PW, q.x_t - pbeg);
    if (n_own > 0) {
      const bool extra = wave < NWV - 1 && pbeg + PW < q.x_t;
      int kk[4]; float rf[4]; const float *row[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int k; unsigned r;
        line_pos(4 * lane + i, k, r);
        const int kf = max(k, 0);
        adv32(k, r, (unsigned)(wave * PW), fa.qstep, fa.rstep, fa.D, fi.invD);
        kk[i] = k - kf; rf[i] = (float)r;
        row[i] = smp1 + (i * 64 + lane) * Wp;
        di.rrow[i] = rrow[4 * lane + i];
        di.rdyd[i] = rdyd[4 * lane + i];
        di.racc[i] = 0.f;
      }
      const int cj = lane <= PW ? wave * PW + lane : q.TP;   // entry TP is never a column
      di.ccol = ccol[cj];
      di.cdxd = cdxd[cj];
      di.colmask = __ballot(lane < PW && di.ccol >= 0);
      my_col = lane < PW ? di.ccol : -1;
      di.below = ((lane + 1) & 63) << 2;
      di.dn = down + (size_t)f * down_stride;
      di.h_out = q.h_out;
      di.proj = q.proj != nullptr;
      di.lane = lane;
      di.csum = 0.f;
      di.rowbuf = (lds_float_t *)(rowbuf + wave * NL);
      {  // the wave's rows: contiguous [R0, R0 + nrows) (every row between two owned ones has its top line in the tile)
        int mn = 0x7fffffff, mx = -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (di.rrow[i] >= 0) { mn = min(mn, di.rrow[i]); mx = max(mx, di.rrow[i]); }
        for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off, 64)); mx = max(mx, __shfl_xor(mx, off, 64)); }
        di.R0 = mx >= 0 ? mn : 0;
        di.nrows = mx >= 0 ? mx - mn + 1 : 0;
      }
      const long long num0 = (long long)(2ull * ((unsigned long long)l0 * q.x_t + p0) + 1ull) * (long long)fa.S - (long long)fa.P;
      float *ob = out + (size_t)f * out_stride + (size_t)pbeg * q.y_t;
      const unsigned loff = (unsigned)(l0 + 4 * lane) * 4u;
      const int nvalid = max(0, min(4, q.y_t - (l0 + 4 * lane)));   // lines of this lane inside the raster
      const bool tailt = l0 + NL > q.y_t;
      if (num0 < 0) {
        if (tailt) walk4<true, true>(fa, row, kk, rf, n_own, extra, ob, loff, (size_t)q.y_t, nvalid, di);
        else walk4<true, false>(fa, row, kk, rf, n_own, extra, ob, loff, (size_t)q.y_t, nvalid, di);
      } else if (tailt) walk4<false, true>(fa, row, kk, rf, n_own, extra, ob, loff, (size_t)q.y_t, nvalid, di);
      else walk4<false, false>(fa, row, kk, rf, n_own, extra, ob, loff, (size_t)q.y_t, nvalid, di);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { di.rrow[i] = -1; di.racc[i] = 0.f; }
    }
  }
  if (q.proj) {
    // row partials: the four wavefronts side by side added left to right -> rowpart[strip][row]; column partials: one
    // wavefront row per tile, so a column's sum over the tile's rows is already complete -> colpart[line tile][column]
    __syncthreads();   // (the staged samples are dead: their LDS carries the wavefronts' row sums)
    float *rp = reinterpret_cast<float *>(lds_d);   // [NWV][NL]
#pragma unroll
    for (int i = 0; i < 4; ++i) rp[wave * NL + 4 * lane + i] = di.racc[i];
    __syncthreads();
    float *pr = q.proj + (size_t)f * q.proj_stride;
    if (tl == 0 && tp == 0 && tid < 2) q.keys[(size_t)f * 2 + tid] = 0ull;
    const int r = tid < NL ? rrow[tid] : -1;
    if (r >= 0) {
      float a = rp[tid];
#pragma unroll
      for (int w = 1; w < NWV; ++w) a = __fadd_rn(a, rp[w * NL + tid]);
      pr[(size_t)q.tiles_l * q.w_out + (size_t)tp * q.h_out + r] = a;
    }
    if (my_col >= 0) pr[(size_t)tl * q.w_out + my_col] = di.csum;
  }
}

// direct variant (no LDS, EXACT arithmetic) for ratios the tiled kernel cannot stage; lanes along lines.
template <bool CPLX>
__global__ __launch_bounds__(256) void k_raster_direct(const float *__restrict__ in, size_t in_stride, unsigned S,
                                                       int y_t, int x_t, float *__restrict__ out, size_t out_stride, IqFmt iqf) {
  const int f = blockIdx.y;
  const float *src = in + (size_t)f * in_stride * (CPLX ? iq_floats(iqf) : 1);
  const unsigned P = (unsigned)y_t * (unsigned)x_t;
  const RsAxis ax = rs_axis(S, P);
  const bool same = (S == P);
  const int lblocks = (y_t + 63) >> 6;
  const size_t total = (size_t)lblocks * x_t * 64;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(w & 63);
    const size_t rest = w >> 6;
    const int p = (int)(rest % (size_t)x_t);
    const int l = (int)(rest / (size_t)x_t) * 64 + lane;
    if (l >= y_t) continue;
    double d;
    const unsigned k = rs_pos(ax, (double)((unsigned)l * (unsigned)x_t + (unsigned)p + 1u), d);
    const float a = load_sample<CPLX, true>(src, k, iqf), b = load_sample<CPLX, true>(src, k + 1u, iqf);
    out[(size_t)f * out_stride + (size_t)p * y_t + l] = same ? (d == 1.0 ? b : a) : rs_blend(a, b, d);
  }
}

// grid = (tiles, frames)
// (256 threads per 64 x 64-pixel tile: 512 threads measured 55.9 us, 512 threads on 64 x 128 58.5 us, 128 threads 70.4 us, against 54.3 us)
constexpr int kDownNT = 256;
template <bool CPLX, int MODE, int SUMS = DS_NONE, int LD = 4, int IQF = (MODE == DM_EXACT ? IQF_RT : IQF_CF32)>
__global__ __launch_bounds__(kDownNT) void k_down_fused(const float *__restrict__ in, size_t in_stride, DownParams q,
                                                    float *__restrict__ out, size_t out_stride, size_t lds_main) {
  extern __shared__ double lds_dn[];
  int tile = (int)blockIdx.x, f = (int)blockIdx.y;
  if (q.xcd_tpx) {
    // XCD-aware order (1-D grid): workgroup b runs on XCD b mod 8; each XCD takes a contiguous range of a frame's tiles --
    // neighbouring column tiles share the 128-byte lines at the ends of their staged sample runs, and tiles of one row block
    // re-read none of them from HBM when they meet in one L2
    const unsigned b = blockIdx.x, xcd = b & 7u, sidx = b >> 3;
    f = (int)(sidx / (unsigned)q.xcd_tpx);
    tile = (int)(xcd * (unsigned)q.xcd_tpx + (sidx - (unsigned)f * (unsigned)q.xcd_tpx));
    if (tile >= q.xcd_tiles) return;
  }
  down_fused_body<CPLX, MODE, kDownNT, SUMS, LD, IQF>(in, in_stride, q, out, out_stride, tile, f, lds_dn, nullptr,
                                         reinterpret_cast<float *>(reinterpret_cast<char *>(lds_dn) + lds_main));
}

// ------------------------------------------------------------------------------------------------------------
// generic helpers (EXACT arithmetic)
// ------------------------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------------------------
// host-side tiling plans
// ------------------------------------------------------------------------------------------------------------
static int check_geom(tsdr_ctx *ctx, size_t S, int y_t, int x_t) {
  if (y_t <= 0 || x_t <= 0) return set_err(ctx, TSDR_EINVAL, "y_t and x_t must be positive");
  const size_t P = (size_t)y_t * (size_t)x_t;
  if (S >= (size_t(1) << 31) || P >= (size_t(1) << 31)) return set_err(ctx, TSDR_EINVAL, "frame larger than 2^31 samples/pixels");
  if (S != P && S < 2) return set_err(ctx, TSDR_EINVAL, "imresize needs at least 2 input samples");
  return TSDR_OK;
}

template <bool CPLX, bool DOWN>
static int launch_tile(tsdr_ctx *ctx, const char *name, const float *in, size_t in_stride, const TileParams &q,
                       size_t lds, float *out, size_t out_stride, float *down, size_t down_stride) {
  const size_t units = (size_t)q.frames * q.tiles_p;
  const size_t G = (size_t)q.xcd_group;
  const size_t upx = ceil_div(units, 8 * G) * G;  // units per XCD slot
  if (upx > 65535 || (size_t)q.tiles_l > 65535 || units >= (size_t(1) << 20))
    return set_err(ctx, TSDR_EINVAL, "raster: too many tiles for one launch (split the buffer)");
  TSDR_LAUNCH(ctx, name, (k_raster_tile<CPLX, DOWN>), dim3(8, (unsigned)q.tiles_l, (unsigned)upx), dim3(256), lds, in,
              in_stride, q, out, out_stride, down, down_stride);
  return TSDR_OK;
}

// sig_to_image for `frames` consecutive frames (raster `out`, may be null when only `down` is wanted and the
// tile kernel applies) and, when `down` != null, the (h_out,w_out) image of each frame from the same launch.
// Returns TSDR_OK and sets *did_down when the down image was produced here.
// proj / got / plan_only: see raster_and_down_d.
int raster_frames_d(tsdr_ctx *ctx, const float *in, int cplx, size_t in_stride, size_t S, int y_t, int x_t, int frames,
                    float *out, size_t out_stride, float *down = nullptr, size_t down_stride = 0, int h_out = 0,
                    int w_out = 0, bool *did_down = nullptr, float *proj = nullptr, ProjLayout *got = nullptr,
                    bool plan_only = false, unsigned long long *keys = nullptr) {
  if (did_down) *did_down = false;
  int rc = check_geom(ctx, S, y_t, x_t);
  if (rc) return rc;
  if (frames <= 0) return TSDR_OK;
  const size_t P = (size_t)y_t * x_t;
  const double sf = (double)S / (double)P;
  // TSDR_FAST exists for the steady-state frame loop (tsdr_frames*), whose input is IQ; the per-function entry
  // points (real input) always run the oracle's operation sequence
  const bool exact = ctx->precision == TSDR_EXACT || !cplx || P >= (size_t(1) << 30);
  // fused downgrade in the raster launch: only when both axes shrink (<= 66 x 130 candidates per tile)
  const bool want_down = down && !(y_t == h_out && x_t == w_out) && y_t >= 2 * 64 && x_t >= 2 * 128 &&
                         (double)y_t / h_out >= 1.0 && (double)x_t / w_out >= 1.0;
  TileParams q{};
  q.S = (unsigned)S; q.y_t = y_t; q.x_t = x_t; q.frames = frames;
  if (cplx) q.iqf = ctx->iq_fmt;
  // pairs of strips.  Measured on C2 -- round 2 (16-byte sample records, 3 workgroups per CU): G=1 0.138 ms, G=4 0.132 ms, G=41 0.146 ms;
  // round 4 (f32 samples, 4 per CU), the launch alone on two boxes: G=1 115.3, G=2 110.9 / 114.9, G=4 113.7 / 117.4, G=8 118.4, G=16 120.5 us
  // (C3: no difference; C5: G=2 1 % behind G=4).  The EXACT tile kernel (two workgroups per CU fewer) keeps four: 146.5 against 148.3 us
  q.xcd_group_log = exact ? 2 : 1;
  q.xcd_group = 1 << q.xcd_group_log;
  q.ax = rs_axis(S, (size_t)y_t * x_t);
  if (h_out > 0 && w_out > 0) {
    q.ay = rs_axis((size_t)y_t, (size_t)h_out); q.axx = rs_axis((size_t)x_t, (size_t)w_out);
    q.inv_sfy = 1.0 / q.ay.sf; q.inv_sfx = 1.0 / q.axx.sf;
  }
  bool tiled = false;
  // EXACT with the downgrade fused in: 64-pixel tiles, because the raster tile kept in LDS then costs 16.6 KiB
  // instead of 33 KiB, which doubles the resident workgroups per CU.  FAST keeps no raster tile (k_raster_fast).
  int tp_max = (want_down && exact) ? 64 : 128;
  // staged-sample budget per tile: EXACT 4 B/sample (<= 48 KiB), FAST 16 B/sample (<= 47 samples per line = 47 KiB;
  // only down-sampling ratios get near it, up-sampling tiles stage ~11-18 samples per line)
  // (FAST with plain f32 samples -- REC4, k_raster_fast: the f32 walk with the in-walk downgrade -- 4 B/sample: <= 94 samples
  // per line, i.e. C3's 1.15 samples per raster pixel get 64-pixel tiles, and with them the in-walk projection sums)
  const bool rec4_ok = !exact && want_down && ctx->opt_raster_rec4 != 0 && 2 * P < (size_t(1) << 32) && y_t > h_out && x_t > w_out;
  auto pick_tp = [&](long w_cap) {
    tiled = false;
    for (int TP = tp_max; TP >= 4; TP >>= 1) {
      const long W = (long)((double)(TP - 1) * sf) + 4;
      if (W <= w_cap) { tiled = true; q.TP = TP; q.W = (int)W; break; }
    }
  };
  pick_tp(exact ? 191 : rec4_ok ? 94 : 47);
  if (rec4_ok && tiled && q.W > 47) {
    // the wider budget only holds for the plan REC4 serves: the in-walk downgrade (TP >= 32)
    if (q.TP < 32) pick_tp(47);
  }
  if (tiled && !exact) {  // FAST: k_raster_fast
    // the in-walk downgrade needs ratios strictly above 1 (a line / pixel is then the top-left tap of at most one
    // output row / column); otherwise the raster is produced here and the caller downgrades separately
    const bool dn = want_down && q.TP >= 32 && y_t > h_out && x_t > w_out;
    // wavefronts stacked vertically per workgroup (see k_raster_fast).  Measured on C2: 1 -> 0.121 ms, 2 -> 0.118 ms,
    // 4 (1024 threads, 77 KiB LDS) -> 0.131 ms; again with f32 samples (REC4, 12 KiB): 130 / 124 / 138 us for the launch
    // round 6: four lines per lane (k_raster_fast4) wherever the f32-sample walk with the in-walk image writes rasters from
    // 128-pixel tiles -- C2's route: a quarter fewer write requests for the same bytes (option "raster_v4", default on)
    const bool v4 = ctx->opt_raster_v4 > 0 && rec4_ok && dn && q.TP == 128 && out != nullptr && 2 * P < (size_t(1) << 24) && y_t >= 512 &&
                    x_t <= 127 * 128 && ctx->opt_raster_split == 0;
    int VW = v4 ? 1 : y_t >= 2 * 64 ? 2 : 1;
    const int lstep = dn ? 63 : 64, NL = v4 ? 256 : lstep * (VW - 1) + 64;
    q.own_l = v4 ? 255 : lstep * VW;
    q.own_p = dn ? q.TP - 1 : q.TP;
    q.tiles_l = dn ? (y_t - 2) / q.own_l + 1 : (int)ceil_div((size_t)y_t, (size_t)q.own_l);
    q.tiles_p = dn ? (x_t - 2) / q.own_p + 1 : (int)ceil_div((size_t)x_t, (size_t)q.TP);
    q.inv_tiles_p = 1.0f / (float)q.tiles_p;
    if (!dn && !out) return TSDR_OK;  // nothing to do here; caller falls back to k_down_fused
    // f32 walk and 32-bit position advance: D = 2P < 2^24 and few enough tiles that the advances stay below 2^32
    const bool w32 = 2 * P < (size_t(1) << 24) && q.tiles_l <= 128 && q.tiles_p <= 128;
    // Staged samples as plain f32 (REC4, k_raster_fast) instead of 16-byte records wherever the f32 walk with the in-walk
    // downgrade runs: C3's tiles (39 samples x 127 lines) were 79 KB of records, ONE 512-thread workgroup per CU with nothing
    // to cover its staging (357 -> 244 us per buffer with 20 KB of samples); C2's 42 KB -> 14 KB is worth 4-6 % of its
    // store-bound launch.  One more sample per line: a pixel reads (k, k + 1).
    const bool rec4 = rec4_ok && dn;
    if (!rec4 && q.W > 47) return set_err(ctx, TSDR_EINVAL, "raster: tile plan needs the f32-sample walk");  // (pick_tp above rules it out)
    if (rec4) q.W += 1;
    // staging lanes per line: the power of two that wastes the fewest lane slots with <= 4 samples per lane
    int best = -1; long best_slots = 1L << 60;
    for (int lg = 2; lg <= 6; ++lg) {
      const long lpl = 1L << lg, cs = (long)ceil_div((size_t)q.W, (size_t)lpl);
      if (cs > 4) continue;
      const long slots = cs * lpl;
      if (slots < best_slots || (slots == best_slots && lg > best)) { best = lg; best_slots = slots; }
    }
    q.lpl_log = best;
    q.cs = (int)ceil_div((size_t)q.W, (size_t)1 << best);
    size_t lds = rec4 ? (((size_t)NL * (size_t)(q.W | 1) * 4 + 15) & ~(size_t)15) + 16 : (size_t)NL * (size_t)(q.W | 1) * 16 + 16;
    const int v4pw = ctx->opt_raster_v4 == 32 ? 32 : 16;
    if (v4) lds += (size_t)(128 / v4pw) * NL * 4;   // the wavefronts' image-row scratch (down_event4)
    // the images' projection partial sums come out of the same walk when the caller has room for them
    // (with narrower tiles -- down-sampling ratios such as C3's -- the per-workgroup part of the sums is spread over
    // four times as many workgroups and costs more than the separate pass over the images: 0.382 vs 0.354 ms)
    const bool pj = dn && q.TP >= 64 && got != nullptr && ((proj != nullptr && keys != nullptr) || plan_only);
    if (dn) {
      q.h_out = h_out; q.w_out = w_out;
      lds += (size_t)(NL + q.TP + 1) * 12 + 16;
      if (pj) {
        lds = std::max(lds, (size_t)2 * 256 * VW * 4);  // the sums reuse the sample region after the walk
        got->ncp = q.tiles_l;
        got->nrp = q.tiles_p;
        q.proj = proj;
        q.keys = keys;
        q.proj_stride = proj_floats(h_out, w_out, *got);
      }
    }
    if (plan_only) { if (did_down) *did_down = dn; return TSDR_OK; }
    const FastAx fa = fast_axis(S, P);
    const FastInc fi = fast_inc(S, P, x_t, q.own_l, q.own_p);
    const size_t units = (size_t)q.frames * q.tiles_p;
    const size_t G = (size_t)q.xcd_group;
    const size_t upx = ceil_div(units, 8 * G) * G;  // units per XCD slot
    if (upx > 65535 || (size_t)q.tiles_l > 65535 || units >= (size_t(1) << 20))
      return set_err(ctx, TSDR_EINVAL, "raster: too many tiles for one launch (split the buffer)");
    const dim3 grid(8, (unsigned)q.tiles_l, (unsigned)upx);
    if (v4) {
      if (v4pw == 32) {
        if (q.iqf.sc16) {
          TSDR_LAUNCH(ctx, "raster_down_iq", (k_raster_fast4<IQF_SC16, 32>), grid, dim3(256), lds, in, in_stride, q, fa, fi, out, out_stride, down, down_stride);
        } else {
          TSDR_LAUNCH(ctx, "raster_down_iq", (k_raster_fast4<IQF_CF32, 32>), grid, dim3(256), lds, in, in_stride, q, fa, fi, out, out_stride, down, down_stride);
        }
      } else if (q.iqf.sc16) {
        TSDR_LAUNCH(ctx, "raster_down_iq", (k_raster_fast4<IQF_SC16, 16>), grid, dim3(512), lds, in, in_stride, q, fa, fi, out, out_stride, down, down_stride);
      } else {
        TSDR_LAUNCH(ctx, "raster_down_iq", (k_raster_fast4<IQF_CF32, 16>), grid, dim3(512), lds, in, in_stride, q, fa, fi, out, out_stride, down, down_stride);
      }
      if (did_down) *did_down = true;
      return TSDR_OK;
    }
#define FASTK2(C, W32, D, PW, VWK, NAME)                                                                              \
  do {                                                                                                                \
    if (out) {                                                                                                        \
      TSDR_LAUNCH(ctx, NAME, (k_raster_fast<C, W32, D, PW, true, VWK>), grid, dim3(256 * VWK), lds, in, in_stride, q, fa,   \
                  fi, out, out_stride, down, down_stride);                                                             \
    } else {                                                                                                          \
      TSDR_LAUNCH(ctx, "down_walk_iq", (k_raster_fast<C, W32, D, PW, false, VWK>), grid,                                \
                  dim3(256 * VWK), lds, in, in_stride, q, fa, fi, out, out_stride, down, down_stride);                   \
    }                                                                                                                 \
  } while (0)
#define FASTK2RF(W32, PW, VWK, IQFK)                                                                                  \
  do {                                                                                                                \
    if (out) {                                                                                                        \
      TSDR_LAUNCH(ctx, "raster_down_iq", (k_raster_fast<true, W32, true, PW, true, VWK, true, IQFK>), grid, dim3(256 * VWK), lds, in,   \
                  in_stride, q, fa, fi, out, out_stride, down, down_stride);                                          \
    } else {                                                                                                          \
      TSDR_LAUNCH(ctx, "down_walk_iq", (k_raster_fast<true, W32, true, PW, false, VWK, true, IQFK>), grid, dim3(256 * VWK), lds, in,    \
                  in_stride, q, fa, fi, out, out_stride, down, down_stride);                                          \
    }                                                                                                                 \
  } while (0)
#define FASTK2R(W32, PW, VWK)                                                                                         \
  do {                                                                                                                \
    if (q.iqf.sc16) FASTK2RF(W32, PW, VWK, IQF_SC16); else FASTK2RF(W32, PW, VWK, IQF_CF32);                          \
  } while (0)
#define FASTK1(C, W32, D, PW, NAME)                                                                                   \
  do {                                                                                                                \
    if (rec4 && D && (PW == 8 || PW == 16 || PW == 32)) { if (VW == 2) FASTK2R(W32, PW, 2); else FASTK2R(W32, PW, 1); } \
    else if (VW == 2) FASTK2(C, W32, D, PW, 2, NAME);                                                                 \
    else FASTK2(C, W32, D, PW, 1, NAME);                                                                              \
  } while (0)
#define FASTK(C, W32, D, NAME)                                                                                        \
  do {                                                                                                                \
    switch (q.TP) {                                                                                                   \
      case 128: FASTK1(C, W32, D, 32, NAME); break;                                                                   \
      case 64: FASTK1(C, W32, D, 16, NAME); break;                                                                    \
      case 32: FASTK1(C, W32, D, 8, NAME); break;                                                                     \
      case 16: FASTK1(C, W32, D, 4, NAME); break;                                                                     \
      case 8: FASTK1(C, W32, D, 2, NAME); break;                                                                      \
      default: FASTK1(C, W32, D, 1, NAME); break;                                                                     \
    }                                                                                                                 \
  } while (0)
    if (dn) {
      if (w32) { FASTK(true, true, true, "raster_down_iq"); } else { FASTK(true, false, true, "raster_down_iq"); }
      if (did_down) *did_down = true;
    } else {
      if (w32) { FASTK(true, true, false, "raster_iq"); } else { FASTK(true, false, false, "raster_iq"); }
    }
#undef FASTK1
#undef FASTK2
#undef FASTK2R
#undef FASTK2RF
#undef FASTK
    return TSDR_OK;
  }
  if (plan_only) return TSDR_OK;  // only k_raster_fast produces projection sums
  if (tiled) {
    const bool dn = want_down && q.TP >= 32;
    q.own_l = dn ? 63 : 64;
    q.own_p = dn ? q.TP - 1 : q.TP;
    q.tiles_l = dn ? (y_t - 2) / 63 + 1 : (int)ceil_div((size_t)y_t, 64);
    q.tiles_p = dn ? (x_t - 2) / q.own_p + 1 : (int)ceil_div((size_t)x_t, (size_t)q.TP);
    q.inv_tiles_p = 1.0f / (float)q.tiles_p;
    // staging lanes per line: a lane issues its loads four at a time, lpl samples apart -- the power of two that wastes the
    // fewest of the ceil(W / 4 lpl) * 4 lpl load slots of a line (see plan_down)
    int best = 0; long best_slots = 1L << 60;
    for (int lg = 2; lg <= 6; ++lg) {
      const long chunk = 4L << lg, slots = (long)ceil_div((size_t)q.W, (size_t)chunk) * chunk;
      if (slots < best_slots || (slots == best_slots && lg > best)) { best = lg; best_slots = slots; }
    }
    q.lpl_log = best;
    size_t lds = (size_t)64 * (size_t)(q.W | 1) * 4 + 16 + 64 * 4 + 16;
    if (dn) {
      q.h_out = h_out; q.w_out = w_out;
      q.NR = (int)ceil(64.0 / ((double)y_t / h_out)) + 5;
      q.NC = (int)ceil((double)q.TP / ((double)x_t / w_out)) + 5;
      if (q.NR > 128 || q.NC > 192) return set_err(ctx, TSDR_EINVAL, "raster: candidate table overflow");
      lds += (size_t)q.TP * 65 * 4 + (size_t)(q.NR + q.NC + 4) * 4 + (size_t)(q.NR + q.NC) * 8;
    }
#define TILE(C, D, NAME) launch_tile<C, D>(ctx, NAME, in, in_stride, q, lds, out, out_stride, down, down_stride)
    if (dn) {
      if (cplx) rc = TILE(true, true, "raster_down_iq_exact");
      else rc = TILE(false, true, "raster_down_f32_exact");
      if (!rc && did_down) *did_down = true;
    } else {
      if (!out) return TSDR_OK;  // nothing to do here; caller falls back to k_down_fused
      if (cplx) rc = TILE(true, false, "raster_iq_exact");
      else rc = TILE(false, false, "raster_f32_exact");
    }
#undef TILE
    return rc;
  }
  if (!out) return TSDR_OK;
  dim3 grid((unsigned)stream_grid(ctx, ceil_div((size_t)y_t, 64) * 64 * (size_t)x_t), (unsigned)frames);
  if (cplx) {
    TSDR_LAUNCH(ctx, "raster_direct_iq", (k_raster_direct<true>), grid, dim3(256), 0, in, in_stride, (unsigned)S, y_t, x_t,
                out, out_stride, ctx->iq_fmt);
  } else {
    TSDR_LAUNCH(ctx, "raster_direct_f32", (k_raster_direct<false>), grid, dim3(256), 0, in, in_stride, (unsigned)S, y_t, x_t,
                out, out_stride, IqFmt{});
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

struct DownPlan { bool fused; int mode; DownParams q; size_t lds; };
int raster_shear_d(tsdr_ctx *ctx, const float *in, size_t in_stride, size_t S, int y_t, int x_t, int frames, float *out,
                   size_t out_stride, bool shear, bool *did, bool plan_only);

static DownPlan plan_down(size_t S, int y_t, int x_t, int h_out, int w_out, bool exact, bool wide_exact = false, bool guard_tiles = false) {
  DownPlan pl;
  pl.fused = false;
  pl.lds = 0;
  pl.mode = DM_EXACT;
  const double sf = (double)S / ((double)y_t * (double)x_t);
  const double sfy = (double)y_t / (double)h_out, sfx = (double)x_t / (double)w_out;
  const long NLd = (long)(63.0 * sfy) + 3;
  // FAST: above a vertical ratio of ~2 the lines between an output row's two tap lines are more than half of the tile's
  // span: stage just the 2 x 64 tap lines (C5: 128 instead of 239)
  const bool sparse = !exact && NLd > 128;
  const long NL = sparse ? 128 : NLd;
  // 64-column tiles (4096 pixels per 256-thread workgroup) with f32 staging: measured at C2 against the former preference
  // (32 columns at most, {a, slope} f64 pairs when they fit 32 KiB -- which held the tile to 16 columns): 50 vs 65 us for the
  // FAST kernel; 128 columns: 58 us.  The EXACT tiling is the sync guard's as well and stays as it was.
  static const int cand_fast[] = {64, 32, 16, 8, 4}, cand_exact[] = {32, 16, 8, 4, 0};
  const int *cand = (exact && !wide_exact) ? cand_exact : cand_fast;   // wide_exact: the EXACT frame path (not the sync guard's tiles)
  for (int pass = 1; pass < 3 && !pl.fused; ++pass) {
    const int sb = 4;
    // (the sync guard's kernel runs one workgroup per CU and opts in to a large LDS: the widest tile that fits 96 KiB -- at C3
    // 32 columns instead of 16, i.e. one round of image tiles per flagged frame instead of two)
    const size_t cap = guard_tiles ? 96 * 1024 : pass == 2 ? 60 * 1024 : 32 * 1024;
    for (int ci = 0; ci < 5 && cand[ci] > 0; ++ci) {
      const int TC = cand[ci];
      const long DPX = (long)((double)(TC - 1) * sfx) + 2;
      const long W = (long)((double)DPX * sf) + 4 + (exact ? 0 : (sf <= 0.5 ? 2 : 1));
      const size_t lds = (((size_t)NL * (size_t)(W | 1) * sb + 15) & ~(size_t)15) + (size_t)TC * 20 + (size_t)NL * 4;
      if (lds <= cap && W < (1 << 20)) {
        pl.fused = true;
        pl.lds = lds;
        // FAST: fixed-point taps where a raster pixel spans at most half a sample (the second tap of a line then lies in the
        // first one's three-sample window); one more staged sample for that window
        pl.mode = exact ? DM_EXACT : (sf <= 0.5 ? DM_FAST_FX : DM_FAST_F32);
        pl.q.S = (unsigned)S; pl.q.y_t = y_t; pl.q.x_t = x_t; pl.q.h_out = h_out; pl.q.w_out = w_out;
        pl.q.sparse = sparse ? 1 : 0;
        pl.q.TC = TC; pl.q.NL = (int)NL; pl.q.W = (int)W; pl.q.tiles_c = (int)ceil_div((size_t)w_out, (size_t)TC);
        // staging lanes per line: a lane issues its loads four at a time, lpl samples apart, so a line costs
        // ceil(W / 4 lpl) * 4 lpl load slots -- the power of two that wastes the fewest (round 4: the former rule counted
        // ceil(W / lpl) * lpl and, for W = 29, chose 32 lanes per line: three of every four loads were clamped duplicates)
        // wide rows (many samples per raster pixel): 16 loads in flight per lane, 8 lanes per line
        const long LD = (!exact && W >= 48) ? 16 : 4;
        pl.q.ld16 = LD == 16 ? 1 : 0;
        int best = 2; long best_slots = 1L << 60;
        for (int lg = 2; lg <= 6; ++lg) {
          const long chunk = LD << lg, slots = (long)ceil_div((size_t)W, (size_t)chunk) * chunk;
          if (slots < best_slots || (slots == best_slots && lg > best)) { best = lg; best_slots = slots; }
        }
        pl.q.lpl_log = best;
        break;
      }
    }
  }
  return pl;
}

// sig_to_image |> downgradeImage for `frames` frames, straight from the signal (no raster in HBM)
// proj / got / keys (FAST, IQ input): the kernel also leaves the images' projection partial sums (layout in *got) and clears the
// frames' argmax keys; plan_only: nothing is launched, *got says what a real call would produce (ncp == 0: nothing).
int down_frames_d(tsdr_ctx *ctx, const float *in, int cplx, size_t in_stride, size_t S, int y_t, int x_t, int h_out,
                  int w_out, int frames, float *out, size_t out_stride, float *proj = nullptr, ProjLayout *got = nullptr,
                  bool plan_only = false, unsigned long long *keys = nullptr) {
  int rc = check_geom(ctx, S, y_t, x_t);
  if (rc) return rc;
  if (h_out <= 0 || w_out <= 0) return set_err(ctx, TSDR_EINVAL, "output size must be positive");
  const bool same2 = (y_t == h_out && x_t == w_out);
  if (!same2 && (y_t < 2 || x_t < 2)) return set_err(ctx, TSDR_EINVAL, "imresize needs at least a 2x2 raster");
  if (frames <= 0) return TSDR_OK;
  // imresize returns a copy when the sizes already match: the raster IS the result
  if (same2) return plan_only ? TSDR_OK : raster_frames_d(ctx, in, cplx, in_stride, S, y_t, x_t, frames, out, out_stride);
  const size_t P = (size_t)y_t * x_t;
  const bool exact = ctx->precision == TSDR_EXACT || !cplx;
  DownPlan pl = plan_down(S, y_t, x_t, h_out, w_out, exact, true);
  if (cplx) pl.q.iqf = ctx->iq_fmt;
  if (pl.fused) {
    const bool psum = !exact && got != nullptr && (plan_only || (proj != nullptr && keys != nullptr));
    if (psum) {
      got->ncp = (int)ceil_div((size_t)h_out, 64);
      got->nrp = pl.q.tiles_c;
    }
    if (plan_only) return TSDR_OK;
    dim3 grid((unsigned)(ceil_div((size_t)h_out, 64) * (size_t)pl.q.tiles_c), (unsigned)frames);
    if (ctx->opt_down_xcd && !exact) {
      pl.q.xcd_tiles = (int)grid.x;
      pl.q.xcd_tpx = (int)ceil_div((size_t)grid.x, 8);
      grid = dim3((unsigned)(8 * pl.q.xcd_tpx * frames), 1);
    }
    const size_t lds_main = (pl.lds + 15) & ~(size_t)15;
  // (the FAST kernels exist once per input format -- ComplexF32 or int16 pairs -- the EXACT one reads either)
#define DOWNKL(C, M, SUMS, LDN, NAME, LDS)                                                                                         \
  do {                                                                                                                             \
    if (M != DM_EXACT && pl.q.iqf.sc16) {                                                                                          \
      TSDR_LAUNCH(ctx, NAME, (k_down_fused<C, M, SUMS, LDN, (M == DM_EXACT ? IQF_RT : IQF_SC16)>), grid, dim3(kDownNT), LDS, in,   \
                  in_stride, pl.q, out, out_stride, lds_main);                                                                     \
    } else {                                                                                                                       \
      TSDR_LAUNCH(ctx, NAME, (k_down_fused<C, M, SUMS, LDN>), grid, dim3(kDownNT), LDS, in, in_stride, pl.q, out, out_stride,     \
                  lds_main);                                                                                                       \
    }                                                                                                                              \
  } while (0)
#define DOWNK(C, M, SUMS, NAME, LDS) DOWNKL(C, M, SUMS, 4, NAME, LDS)
#define DOWNK16(C, M, SUMS, NAME, LDS) DOWNKL(C, M, SUMS, 16, NAME, LDS)
    const size_t lds_ps = lds_main + (kDownNT + (size_t)pl.q.TC) * 4;
    if (cplx) {
      if (pl.mode == DM_EXACT) { DOWNK(true, DM_EXACT, DS_NONE, "down_fused_iq_exact", pl.lds); }
      else if (psum) {
        pl.q.proj = proj; pl.q.proj_stride = proj_floats(h_out, w_out, *got); pl.q.keys = keys;
        if (pl.mode == DM_FAST_FX) {
          if (pl.q.ld16) { DOWNK16(true, DM_FAST_FX, DS_PSUM, "down_fused_iq_sums", lds_ps); } else { DOWNK(true, DM_FAST_FX, DS_PSUM, "down_fused_iq_sums", lds_ps); }
        } else {
          if (pl.q.ld16) { DOWNK16(true, DM_FAST_F32, DS_PSUM, "down_fused_iq_sums", lds_ps); } else { DOWNK(true, DM_FAST_F32, DS_PSUM, "down_fused_iq_sums", lds_ps); }
        }
      }
      else if (pl.mode == DM_FAST_FX) {
        if (pl.q.ld16) { DOWNK16(true, DM_FAST_FX, DS_NONE, "down_fused_iq", pl.lds); } else { DOWNK(true, DM_FAST_FX, DS_NONE, "down_fused_iq", pl.lds); }
      }
      else if (pl.q.ld16) { DOWNK16(true, DM_FAST_F32, DS_NONE, "down_fused_iq", pl.lds); }
      else { DOWNK(true, DM_FAST_F32, DS_NONE, "down_fused_iq", pl.lds); }
    } else {
      DOWNK(false, DM_EXACT, DS_NONE, "down_fused_f32_exact", pl.lds);
    }
#undef DOWNK16
#undef DOWNK
#undef DOWNKL
    return TSDR_OK;
  }
  if (plan_only) return TSDR_OK;
  // fallback: materialise each raster in workspace, then the generic 2-D resize
  // (one raster per pipeline lane: two submissions of tsdr_frames_submit_d may be walking this loop side by side)
  float *ras = (float *)ctx->scratch(WS_RASTER, 4 * P * 4);
  if (!ras) return TSDR_ENOMEM;
  ras += (size_t)(ctx->pipe_lane & 3) * P;
  for (int f = 0; f < frames; ++f) {
    rc = raster_frames_d(ctx, in + (size_t)f * in_stride * (cplx ? iq_floats(ctx->iq_fmt) : 1), cplx, in_stride, S, y_t, x_t, 1, ras, P);
    if (rc) return rc;
    rc = resize2d_d(ctx, ras, y_t, x_t, h_out, w_out, out + (size_t)f * out_stride);
    if (rc) return rc;
  }
  return TSDR_OK;
}

// Sync guard (guard.h): the tiling of the EXACT raster-free kernel for this geometry, whose workgroup body the guard's
// kernel (sync.hip) runs on the frames it flags.  false: this geometry has no fused exact kernel.
bool guard_image_plan(tsdr_ctx *ctx, size_t S, int y_t, int x_t, int h_out, int w_out, DownParams *q, size_t *lds) {
  if (check_geom(ctx, S, y_t, x_t)) return false;
  if ((y_t == h_out && x_t == w_out) || y_t < 2 || x_t < 2) return false;
  DownPlan pl = plan_down(S, y_t, x_t, h_out, w_out, /*exact=*/true, false, /*guard_tiles=*/true);
  if (!pl.fused) return false;
  pl.q.iqf = ctx->iq_fmt;
  *q = pl.q;
  *lds = pl.lds;
  return true;
}

// raster (optional) + (h_out,w_out) image for every frame with as few passes over IQ as possible.
// proj != nullptr: room for the projection partial sums of every (h_out, w_out) image (layout: sync_layout.h); when the
// FAST tile kernel runs it leaves them there and describes them in *got (ncp == 0: not produced -- the caller then
// forms the projections from the images).  plan_only: nothing is launched, *got says what a real call would produce.
int raster_and_down_d(tsdr_ctx *ctx, const float *in, int cplx, size_t in_stride, size_t S, int y_t, int x_t, int h_out,
                      int w_out, int frames, float *raster, size_t raster_stride, float *down, size_t down_stride,
                      float *proj, ProjLayout *got, bool plan_only, unsigned long long *keys) {
  if (got) *got = ProjLayout{};
  // FAST without a raster to write: k_down_fused re-derives the four taps of every output pixel from 64 x 64-pixel tiles of
  // staged samples and leaves the projection partial sums itself (round 3: 54 us at C2 against the walk's 77 us with
  // out == null -- the walk evaluates all 2.9 M raster pixels of a frame for the 1.8 M that are taps).  Other geometries
  // fall through to the walk, then to the raster + resize fallback.
  if (!raster && ctx->precision == TSDR_FAST && cplx && !ctx->opt_fast_walk_only) {
    ProjLayout pl{};
    const DownPlan dp = plan_down(S, y_t, x_t, h_out, w_out, false);
    // ... where a tile of at least 32 columns fits (C2: 0.115 samples per raster pixel, 64 columns, 0.102 vs 0.123 ms per buffer
    // in round 3; C5: 0.084, 0.138 vs 0.279 ms), or, above 0.5 samples per raster pixel, one of 16 (C3: 1.15 samples per pixel --
    // the walk won there, 0.420 vs 0.461 ms, while the tap kernel staged its 121-sample rows four loads at a time: a chain of
    // dependent round trips, 28 us per tile.  With 16 loads in flight per lane: 186 us against the walk's 323 + k_proj's 24,
    // 0.243 vs 0.403 ms per buffer).  Option "down_spp_max_pct" (default 200) bounds the ratio; the walk keeps the rest.
    const double spp = (double)S / ((double)y_t * (double)x_t);   // samples per raster pixel
    if (dp.fused && dp.q.TC >= (spp > 0.5 ? 16 : 32) && spp <= (double)ctx->opt_down_spp_max_pct * 0.01 && !(y_t == h_out && x_t == w_out) &&
        check_geom(ctx, S, y_t, x_t) == TSDR_OK && y_t >= 2 && x_t >= 2) {
      int rc = down_frames_d(ctx, in, cplx, in_stride, S, y_t, x_t, h_out, w_out, frames, down, down_stride, proj, got ? &pl : nullptr,
                             plan_only, keys);
      if (rc) return rc;
      if (got) *got = pl;
      return TSDR_OK;
    }
  }
  // FAST with a raster (option "raster_split"; A/B of round 4): the rasters by the store-aligned ("sheared") raster-only
  // kernel of raster_shear.hip, the images + projection sums by the raster-free kernel -- two launches, IQ read twice,
  // instead of the one walk that produces raster, image and sums with misaligned column stores
  if (raster && ctx->precision == TSDR_FAST && cplx && ctx->opt_raster_split && !ctx->iq_fmt.sc16) {   // (the A/B kernel reads ComplexF32 only)
    const DownPlan dp = plan_down(S, y_t, x_t, h_out, w_out, false);
    const double spp = (double)S / ((double)y_t * (double)x_t);
    if (dp.fused && dp.q.TC >= 32 && spp <= 0.5 && !(y_t == h_out && x_t == w_out) && check_geom(ctx, S, y_t, x_t) == TSDR_OK && y_t >= 64 && x_t >= 128) {
      bool did = false;
      int rc = raster_shear_d(ctx, in, in_stride, S, y_t, x_t, frames, raster, raster_stride, ctx->opt_raster_split == 1, &did, plan_only);
      if (rc) return rc;
      if (did) {
        ProjLayout pl{};
        rc = down_frames_d(ctx, in, cplx, in_stride, S, y_t, x_t, h_out, w_out, frames, down, down_stride, proj, got ? &pl : nullptr, plan_only, keys);
        if (rc) return rc;
        if (got) *got = pl;
        return TSDR_OK;
      }
    }
  }
  if (raster || (ctx->precision == TSDR_FAST && cplx)) {
    bool did = false;
    ProjLayout pl{};
    int rc = raster_frames_d(ctx, in, cplx, in_stride, S, y_t, x_t, frames, raster, raster_stride, down, down_stride, h_out,
                             w_out, &did, proj, got ? &pl : nullptr, plan_only, keys);
    if (rc) return rc;
    if (did) { if (got) *got = pl; return TSDR_OK; }
  }
  if (plan_only) return TSDR_OK;
  return down_frames_d(ctx, in, cplx, in_stride, S, y_t, x_t, h_out, w_out, frames, down, down_stride);
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
