// raster_shear.hip -- sig_to_image (Resampler.jl:117-122) of every frame of a buffer, TSDR_FAST arithmetic, raster only,
// with every store on the 128-byte grid.
//
// The raster of a frame is the reference's column-major Matrix{Float32}(y_t, x_t): element (line l, pixel p) at
// p * y_t + l.  Lanes of a wavefront are consecutive LINES, so a wave-store is one 256-byte column segment -- which, with
// y_t = 1125, starts at an arbitrary 4-byte boundary and touches three 128-byte lines, two of them partially (the store
// pattern that holds k_raster_fast at 0.47 of the HBM peak: DESIGN.md section 4).  Here the 64-line window of a wavefront
// is SHEARED: for pixel column p it starts at the line whose address is a multiple of 128 bytes,
//     l_first(p) = 64 * tl - o(p),   o(p) = (element offset of (line 0, pixel p)) mod 32,
// so every wave-store is exactly two full 128-byte lines.  o(p+1) = (o(p) + y_t) mod 32: from one column to the next the
// whole window moves by the same delta = -(y_t mod 32) or 32 - (y_t mod 32) lines, wave-uniformly, and a lane follows a
// diagonal through the tile.  What that costs: the tile stages 64 + 31 lines of |IQ| instead of 64, a lane's staged row
// changes with every pixel (one more LDS read per pixel for the row's first sample index), and nothing can be carried
// along a line in a lane -- which is why this kernel does no in-walk downgrade and no projection sums; the frame loop
// pairs it with k_down_fused (the raster-free image kernel, which reads IQ again).
//
// Arithmetic: 32.32 fixed-point source coordinate x = sf (m + 1/2) - 1/2 of raster pixel m = l x_t + p (exact start per
// lane from f64, one 64-bit add per pixel), |IQ| by v_sqrt_f32 (abs_iq<false>), pixel = (1-t) a + t b as two f32 FMAs:
// within ~2 ulp of TSDR_EXACT (tests assert 6e-7).  SHEAR = false is the same kernel with o(p) = 0: the unsheared control
// of the A/B (identical instruction stream, misaligned stores).
#include "common.h"
#include "down_fused.h"

namespace tsdr {

struct ShearParams {
  unsigned S;
  int y_t, x_t, frames;
  int W, rows;            // staged samples per row, staged rows (64 + 31 with SHEAR)
  int tiles_p, tiles_l;
  int c;                  // y_t mod 32
  unsigned out_mis;       // (out address / 4) mod 32
  double sf;
  long long XA, XB;       // 32.32 increments per column: sf (1 + delta x_t) for delta = -c and 32 - c
  float inv_W;
};

constexpr int kShearTP = 128;   // pixel columns per workgroup: 4 wavefronts x 32

// dword store with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset
__device__ __forceinline__ void store_saddr_f(float *base_uniform, unsigned lane_off_bytes, float v) {
  asm volatile("global_store_dword %0, %1, %2" : : "v"(lane_off_bytes), "v"(v), "s"(base_uniform));   // (no "memory" clobber: the
}                                                                                                       // kernel never reads `out`, and LDS reads may move across it)

template <bool SHEAR>
__global__ __launch_bounds__(256) void k_raster_shear(const float2 *__restrict__ iq, size_t in_stride, ShearParams q,
                                                       float *__restrict__ out, size_t out_stride) {
  extern __shared__ float lds_sh[];
  const int Wp = q.W | 1;
  float *smp = lds_sh;                                    // [rows][Wp] |IQ|
  int *kfirst = reinterpret_cast<int *>(smp + (size_t)q.rows * Wp);   // [rows]
  const int tp = blockIdx.x, tl = blockIdx.y, f = blockIdx.z;
  const int tid = threadIdx.x;
  const int p0 = tp * kShearTP;
  const int l_tile0 = tl * 64 - (SHEAR ? 31 : 0);         // line of staged row 0
  const float2 *src = iq + (size_t)f * in_stride;
  for (int r = tid; r < q.rows; r += 256) {
    const int l = min(max(l_tile0 + r, 0), q.y_t - 1);
    const double m = (double)l * (double)q.x_t + (double)p0;
    kfirst[r] = (int)floor(fma(q.sf, m + 0.5, -0.5)) - 1;   // (may be -2 on the frame's first line: the staging clamps)
  }
  __syncthreads();
  {
    const int total = q.rows * q.W;
    for (int i0 = tid; i0 < total; i0 += 4 * 256) {   // four loads in flight per thread
      float2 z[4];
      int rr[4], jj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * 256, total - 1);
        int r = (int)(((float)i + 0.5f) * q.inv_W);
        int j = i - r * q.W;
        if (j < 0) { j += q.W; --r; } else if (j >= q.W) { j -= q.W; ++r; }
        rr[u] = r; jj[u] = j;
        const int k = min(max(kfirst[r] + j, 0), (int)q.S - 1);
        z[u] = src[k];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * 256 < total) smp[rr[u] * Wp + jj[u]] = abs_iq<false>(z[u].x, z[u].y);
    }
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int pbeg = p0 + wave * 32;
  const int pend = min(pbeg + 32, q.x_t);
  if (pbeg >= pend) return;
  // o(p): element offset of (line 0, pixel p) of this frame modulo 32 (wave-uniform)
  unsigned o = 0u;
  if (SHEAR) o = (unsigned)((q.out_mis + (unsigned long long)f * out_stride + (unsigned long long)pbeg * (unsigned)q.y_t) & 31ull);
  const int row0 = lane + (SHEAR ? 31 : 0) - (int)o;      // staged row of this lane's line; line = l_tile0 + row
  long long X;
  {
    const int l = l_tile0 + row0;
    const double m = (double)l * (double)q.x_t + (double)pbeg;   // (lines outside the frame are never stored)
    X = __double2ll_rd(fma(q.sf, m + 0.5, -0.5) * 4294967296.0);
  }
  // per-lane state as byte offsets: row4 = 4 row (kfirst entry, store offset), rowoff = 4 row Wp (staged row)
  int row4 = row0 * 4, rowoff = row0 * Wp * 4;
  const char *smpb = reinterpret_cast<const char *>(smp);
  const char *kfb = reinterpret_cast<const char *>(kfirst);
  // store: scalar base = element (line l_tile0, pixel p), per-lane offset 4 row
  float *ob = out + (size_t)f * out_stride + (size_t)pbeg * q.y_t + l_tile0;
  const unsigned cc = (unsigned)q.c;
  const int lim4 = q.y_t * 4, base4 = l_tile0 * 4, jmax = q.W - 2;
  const int Wp4 = Wp * 4;
  // one pixel of this lane: position X, staged row given by (row4, rowoff)
  auto pixel = [&](long long Xp, int kf, int roff) -> float {
    const int k = (int)(Xp >> 32);
    const unsigned lo = (unsigned)Xp;
    // weights t = frac, u = 1 - frac, EACH from its own 32-bit integer (lo + 1/2 and ~lo + 1/2 are complements to 2^32):
    // both carry a relative error of 2^-24, so u a + t b -- a sum of non-negatives -- does too.  (1 - t from the rounded t
    // is off by up to 2^-25 absolutely: 1e-6 of a pixel that sits just before a sample 100 times smaller than the previous one.)
    const float t = __fmaf_rn((float)lo, 0x1p-32f, 0x1p-33f), u = __fmaf_rn((float)~lo, 0x1p-32f, 0x1p-33f);
    // (lanes whose line lies outside the frame carry a coordinate outside the staged range: index clamped, pixel not stored)
    const int j = min(max(k - kf, 0), jmax);
    const float *ab = reinterpret_cast<const float *>(smpb + roff + j * 4);
    return __fmaf_rn(t, ab[1], __fmul_rn(u, ab[0]));
  };
  int p = pbeg;
  for (; p + 4 <= pend; p += 4) {   // four columns per trip: their LDS reads are issued together
    int r4[4], ro[4];
    long long Xs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r4[i] = row4; ro[i] = rowoff; Xs[i] = X;
      if (SHEAR) {
        const unsigned o2 = (o + cc) & 31u;
        const bool wrapped = o2 < o;                        // delta = o - o2 = 32 - c when the offset wrapped, else -c
        const int d = (int)o - (int)o2;                     // (wave-uniform)
        row4 += d * 4; rowoff += d * Wp4;
        X += wrapped ? q.XB : q.XA;
        o = o2;
      } else {
        X += q.XA;
      }
    }
    int kf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kf[i] = *reinterpret_cast<const int *>(kfb + r4[i]);
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pixel(Xs[i], kf[i], ro[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((unsigned)(r4[i] + base4) < (unsigned)lim4) store_saddr_f(ob, (unsigned)r4[i], v[i]);
      ob += q.y_t;
    }
  }
  for (; p < pend; ++p) {
    const int kf = *reinterpret_cast<const int *>(kfb + row4);
    const float v = pixel(X, kf, rowoff);
    if ((unsigned)(row4 + base4) < (unsigned)lim4) store_saddr_f(ob, (unsigned)row4, v);
    ob += q.y_t;
    if (SHEAR) {
      const unsigned o2 = (o + cc) & 31u;
      const bool wrapped = o2 < o;
      const int d = (int)o - (int)o2;
      row4 += d * 4; rowoff += d * Wp4;
      X += wrapped ? q.XB : q.XA;
      o = o2;
    } else {
      X += q.XA;
    }
  }
}

// plan_only: nothing is launched, *did says whether a real call would
int raster_shear_d(tsdr_ctx *ctx, const float *in, size_t in_stride, size_t S, int y_t, int x_t, int frames, float *out,
                   size_t out_stride, bool shear, bool *did, bool plan_only) {
  *did = false;
  const double sf = (double)S / ((double)y_t * (double)x_t);
  if (sf > 0.5 || y_t < 64 || x_t < kShearTP || frames <= 0 || frames > 65535) return TSDR_OK;
  if ((reinterpret_cast<uintptr_t>(out) & 3u) != 0) return TSDR_OK;
  ShearParams q{};
  q.S = (unsigned)S; q.y_t = y_t; q.x_t = x_t; q.frames = frames; q.sf = sf;
  q.c = y_t & 31;
  if (q.c == 0 && (out_stride & 31) == 0 && (reinterpret_cast<uintptr_t>(out) & 127u) == 0) shear = false;   // already on the grid
  q.rows = shear ? 95 : 64;
  q.W = (int)((double)(kShearTP - 1) * sf) + 5;
  q.inv_W = 1.0f / (float)q.W;
  q.tiles_p = (int)ceil_div((size_t)x_t, (size_t)kShearTP);
  q.tiles_l = (int)ceil_div((size_t)y_t + (shear ? 31 : 0), 64);
  q.out_mis = (unsigned)((reinterpret_cast<uintptr_t>(out) >> 2) & 31u);
  const double xt = (double)x_t;
  q.XA = (long long)floor(sf * (1.0 - (shear ? (double)q.c * xt : 0.0)) * 4294967296.0);
  q.XB = (long long)floor(sf * (1.0 + (double)(32 - q.c) * xt) * 4294967296.0);
  const size_t lds = ((size_t)q.rows * (size_t)(q.W | 1) + (size_t)q.rows) * 4;
  if (lds > 60 * 1024 || q.tiles_l > 65535) return TSDR_OK;
  if (plan_only) { *did = true; return TSDR_OK; }
  const dim3 grid((unsigned)q.tiles_p, (unsigned)q.tiles_l, (unsigned)frames);
  if (shear) {
    TSDR_LAUNCH(ctx, "raster_sheared_iq", k_raster_shear<true>, grid, dim3(256), lds, reinterpret_cast<const float2 *>(in), in_stride, q, out,
                out_stride);
  } else {
    TSDR_LAUNCH(ctx, "raster_unsheared_iq", k_raster_shear<false>, grid, dim3(256), lds, reinterpret_cast<const float2 *>(in), in_stride, q,
                out, out_stride);
  }
  *did = true;
  return TSDR_OK;
}

}  // namespace tsdr
