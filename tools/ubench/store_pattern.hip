// Development aid (gfx950): what the raster's store stream costs by shape.  Every variant writes the same 30 x
// 2576 x Y x 4 bytes in column-major (line fastest) layout, tile by tile as the raster kernel does (workgroup = 8
// wavefronts = 2 vertical x 4 horizontal, 32 pixel columns per wavefront), differing only in
//   V = lines per lane (1: dword stores, 256 B per wave-instruction; 2: dwordx2, 512 B; 4: dwordx4, 1024 B)
//   Y = column height (1125: every segment starts on an arbitrary 4-byte boundary; 1152: line-aligned segments)
// and optionally in a little VALU work between stores (W fma per pixel) to see how the two overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int V, int W>
__global__ __launch_bounds__(512) void k_store(float *out, int y_t, int x_t, int tiles_l, int tiles_p, float seed) {
  // tile: (2 * 64 * V) lines x 128 pixels; wave (wv, wh): lines [wv*64*V, +64*V), pixels [wh*32, +32)
  const int bid = blockIdx.x;
  const int tl = bid % tiles_l, rest = bid / tiles_l, tp = rest % tiles_p, f = rest / tiles_p;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wv = wave >> 2, wh = wave & 3;
  const int l0 = tl * (128 * V) + wv * 64 * V + lane * V;
  const int p0 = tp * 128 + wh * 32;
  float acc = seed + lane;
  float *base = out + (size_t)f * y_t * x_t;
  for (int p = p0; p < p0 + 32 && p < x_t; ++p) {
#pragma unroll
    for (int w = 0; w < W; ++w) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    if (l0 + V <= y_t) {
      float *dst = base + (size_t)p * y_t + l0;
      // (inline asm: the compiler cannot prove 8/16-byte alignment -- there is none for y_t = 1125 -- and would split
      // the store; the hardware takes dword-aligned wide stores)
      typedef float v2f __attribute__((ext_vector_type(2)));
      typedef float v4f __attribute__((ext_vector_type(4)));
      if (V == 1) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(acc) : "memory");
      if (V == 2) { v2f d = {acc, acc + 1.0f}; asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(d) : "memory"); }
      if (V == 4) { v4f d = {acc, acc + 1.0f, acc + 2.0f, acc + 3.0f}; asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(d) : "memory"); }
    }
  }
}

// the same bytes with every wave-store 128-byte aligned although y_t is not a multiple of 32: for pixel column p the
// 64-line windows of the wavefronts are shifted by s(p) = (-(f*P + p*y_t)) mod 32 lines ("sheared" tiles); one more
// wavefront row covers the tail.
template <int W>
__global__ __launch_bounds__(512) void k_store_sheared(float *out, int y_t, int x_t, int tiles_l, int tiles_p, float seed) {
  const int bid = blockIdx.x;
  const int tl = bid % tiles_l, rest = bid / tiles_l, tp = rest % tiles_p, f = rest / tiles_p;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wv = wave >> 2, wh = wave & 3;
  const int p0 = tp * 128 + wh * 32;
  float acc = seed + lane;
  const size_t fbase = (size_t)f * y_t * x_t;
  for (int p = p0; p < p0 + 32 && p < x_t; ++p) {
#pragma unroll
    for (int w = 0; w < W; ++w) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    const size_t col = fbase + (size_t)p * y_t;
    const int s = (int)((32 - (col & 31)) & 31);          // first line of this column that starts a 128-byte line
    const int l = (tl * 2 + wv) * 64 + s - 32 + lane;      // windows [64k + s - 32, +64)
    if (l >= 0 && l < y_t) {
      float *dst = out + col + l;
      asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(acc) : "memory");
    }
  }
}

template <int W>
static void run_sheared(const char *name, int y_t) {
  const int x_t = 2576, frames = 30;
  const int tiles_l = (y_t + 32 + 127) / 128, tiles_p = (x_t + 127) / 128;
  const size_t bytes = (size_t)frames * y_t * x_t * 4;
  float *out;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int grid = tiles_l * tiles_p * frames;
  for (int i = 0; i < 3; ++i) k_store_sheared<W><<<grid, 512>>>(out, y_t, x_t, tiles_l, tiles_p, 1.0f);
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_store_sheared<W><<<grid, 512>>>(out, y_t, x_t, tiles_l, tiles_p, 1.0f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
  printf("%-34s y_t=%d  %7.1f us  %6.2f TB/s  (grid %d)\n", name, y_t, ms * 1e3, bytes / (ms * 1e-3) / 1e12, grid);
  (void)hipFree(out);
}

template <int V, int W>
static void run(const char *name, int y_t) {
  const int x_t = 2576, frames = 30;
  const int tiles_l = (y_t + 128 * V - 1) / (128 * V), tiles_p = (x_t + 127) / 128;
  const size_t bytes = (size_t)frames * y_t * x_t * 4;
  float *out;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = tiles_l * tiles_p * frames;
  for (int i = 0; i < 3; ++i) k_store<V, W><<<grid, 512>>>(out, y_t, x_t, tiles_l, tiles_p, 1.0f);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_store<V, W><<<grid, 512>>>(out, y_t, x_t, tiles_l, tiles_p, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  printf("%-34s y_t=%d  %7.1f us  %6.2f TB/s  (grid %d)\n", name, y_t, ms * 1e3, bytes / (ms * 1e-3) / 1e12, grid);
  hipFree(out);
}

// ---- round 6: the raster launch's WHOLE traffic pattern with free arithmetic ("skeleton") -----------------------------------
// k_raster_fast (resample.hip) at C2, stripped of everything but its memory operations and a VALU filler:
//   tile = 127 lines x 127 pixel columns, 512 threads = 2 (vertical) x 4 (horizontal) wavefronts, lanes = lines, a wavefront
//   walks 32 pixel columns: one 256-byte column segment per pixel at (p * y_t + line) * 4 -- arbitrary 4-byte alignment;
//   staging: per line the 17 IQ samples (8 bytes each) the tile's pixels lie between, read at (line * x_t + p0) * S / P,
//   |IQ| to LDS, one barrier; per pixel one LDS read;
//   image: every column that is the left tap of an output column (31 %) stores ~34 floats (the wave's output rows) into the
//   column-major 600 x 800 image -- a 136-byte run at arbitrary alignment;
//   W fma per pixel of filler (the real walk: 9 + ~5 amortised event / staging instructions).
// Same grid order as the kernel (8 XCD slots x line tiles x units).  IMG / IQR switch the image stores and the IQ reads.
template <int W, bool IMG, bool IQR>
__global__ __launch_bounds__(512, 8) void k_skeleton(const float2 *__restrict__ iq, float *__restrict__ out, float *__restrict__ img, int y_t,
                                                    int x_t, int S, int tiles_l, int tiles_p, int frames, float seed) {
  __shared__ float smp[127 * 19];
  const unsigned xcd = blockIdx.x, ul = blockIdx.z;
  const int tl = (int)blockIdx.y;
  const unsigned U = (unsigned)(frames * tiles_p);
  const unsigned u = ((((ul >> 1) << 3) + xcd) << 1) + (ul & 1u);
  if (u >= U) return;
  const int f = (int)(u / (unsigned)tiles_p), tp = (int)(u % (unsigned)tiles_p);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wv = wave >> 2, wh = wave & 3;
  const int l0 = tl * 126, p0 = tp * 127;
  const double sf = (double)S / ((double)y_t * x_t);
  float acc = seed + lane;
  if (IQR) {
    // 4 lanes per line, up to 5 samples each
    for (int r = threadIdx.x >> 2; r < 127; r += 128) {
      const int l = min(l0 + r, y_t - 1);
      const unsigned k0 = (unsigned)(((double)l * x_t + p0) * sf);
      for (int t = 0; t < 5; ++t) {
        const int j = (threadIdx.x & 3) * 5 + t;
        if (j < 19) {
          const float2 z = iq[(size_t)f * S + min(k0 + (unsigned)j, (unsigned)S - 1u)];
          smp[r * 19 + j] = __fsqrt_rn(z.x * z.x + z.y * z.y);
        }
      }
    }
    __syncthreads();
  }
  const int line = wv * 63 + lane;
  const int l = min(l0 + line, y_t - 1);
  float *base = out + (size_t)f * y_t * x_t;
  float *ib = img + (size_t)f * 480000;
  const int rrow = (int)((l + 0.5) * (600.0 / y_t));          // this lane's output row
  const bool own = (int)((l - 0.5) * (600.0 / y_t)) != rrow && rrow < 600;   // ~ every 1.875th line owns one
  float pos = (float)lane * 0.37f;
  for (int i = 0; i < 32; ++i) {
    const int p = p0 + wh * 32 + i;
    if (p >= x_t) break;
    if (IQR) { pos += 0.115f; acc += smp[line * 19 + min((int)pos & 15, 17)]; }
#pragma unroll
    for (int w = 0; w < W; ++w) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    float *dst = base + (size_t)p * y_t + l;
    asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(acc) : "memory");
    if (IMG) {
      const int c = (p * 800) / x_t;
      if (c != ((p - 1) * 800) / x_t && own && c < 800) {     // (wave-uniform column test in integers, per-lane row ownership)
        float *d2 = ib + (size_t)c * 600 + rrow;
        asm volatile("global_store_dword %0, %1, off" ::"v"(d2), "v"(acc) : "memory");
      }
    }
  }
}

// ---- round 6, the alternative: whole-column strips through an LDS image of the output ---------------------------------------
// Workgroup = 576 threads = 9 wavefronts; strip = ALL y_t lines x 32 pixel columns of one frame, as 4 sub-tiles of 8 columns.
// Lanes = lines (two passes of 576 lines).  A sub-tile's pixels go to LDS laid out exactly like the output (line fastest): the
// sub-tile is one contiguous 8 * y_t * 4 = 36 000-byte range of the raster, drained with 16-byte stores -- every write request a
// full 128-byte line whatever y_t is.  IQ: per line the ~6 samples the strip's 32 columns lie between, staged once per strip.
// Image: the strip's ~10 output columns, each a contiguous 2400-byte run of the column-major 600 x 800 image.
template <int W, bool IMG, int IQR>
__global__ __launch_bounds__(576, 2) void k_strip(const float2 *__restrict__ iq, float *__restrict__ out, float *__restrict__ img, int y_t, int x_t,
                                                 int S, int strips, int frames, float seed) {
  extern __shared__ float lds[];
  float *tile = lds;                 // [8][y_t]
  float *smp = lds + 8 * y_t;        // [y_t][9] (odd pitch: lanes = lines read conflict-free)
  const int total = frames * strips;
  const int per = (total + 7) / 8;
  const int id = (int)(blockIdx.x % 8u) * per + (int)(blockIdx.x / 8u);    // consecutive strips on one XCD
  if (id >= total) return;
  const int f = id / strips, sp = id % strips;
  const int p0 = sp * 32;
  const int tid = threadIdx.x;
  const double sf = (double)S / ((double)y_t * x_t);
  float acc[2] = {seed + tid, seed - tid};
  if (IQR & 1) {
    // 8 lanes per line, one sample each (6 used): a wave-load is 8 lines x 64 contiguous bytes; four lines in flight per lane
    const int t = tid & 7;
    {
      // ALL of a thread's samples in flight at once (16 lines, 72 apart): one round trip per strip, not four
      const int l = tid >> 3;
      float2 z[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int ll = min(l + 72 * k, y_t - 1);
        const unsigned k0 = (unsigned)(((float)ll * (float)x_t + (float)p0) * (float)sf);
        z[k] = iq[(size_t)f * S + min(k0 + (unsigned)t, (unsigned)S - 1u)];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (l + 72 * k < y_t) smp[(l + 72 * k) * 9 + t] = __fsqrt_rn(z[k].x * z[k].x + z[k].y * z[k].y);
    }
    __syncthreads();
  }
  float *base = out + (size_t)f * y_t * x_t;
  float pos = (float)tid * 0.37f;
  for (int st = 0; st < 4; ++st) {
    const int pc = p0 + st * 8;
    if (pc >= x_t) break;
    const int ncol = min(8, x_t - pc);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int l = pass * 576 + tid;
      if (l < y_t) {
        for (int i = 0; i < ncol; ++i) {
          if (IQR & 2) { pos += 0.115f; acc[pass] += smp[l * 9 + ((int)pos & 3)]; }
#pragma unroll
          for (int w = 0; w < W; ++w) acc[pass] = __fmaf_rn(acc[pass], 1.0001f, 0.5f);
          tile[i * y_t + l] = acc[pass];
        }
      }
    }
    __syncthreads();
    // drain: the sub-tile is the byte range [pc * y_t * 4, + ncol * y_t * 4) of the frame's raster (16-byte aligned when pc % 4 == 0)
    const int n4 = ncol * y_t / 4;
    float4 *dst = reinterpret_cast<float4 *>(base + (size_t)pc * y_t);
    const float4 *src = reinterpret_cast<const float4 *>(tile);
    for (int i = tid; i < n4; i += 576) dst[i] = src[i];
    if (IMG) {
      // output columns whose left tap lies in this sub-tile: ~2.5, each 600 contiguous floats
      const int c0 = (int)((pc + 0.5) * (800.0 / x_t)), c1 = (int)((pc + ncol + 0.5) * (800.0 / x_t));
      for (int c = c0 + 1; c <= c1 && c < 800; ++c)
        for (int r = tid; r < 600; r += 576) {
          const int lt = (int)(r * 1.875f);
          const float v = tile[lt] + tile[lt + 1] + tile[y_t + lt] + tile[y_t + lt + 1];
          img[(size_t)f * 480000 + (size_t)c * 600 + r] = v;
        }
    }
    __syncthreads();
  }
}

template <int W, bool IMG, int IQR>
static void run_strip(const char *name) {
  const int y_t = 1125, x_t = 2576, frames = 30, S = 333333;
  const int strips = (x_t + 31) / 32;
  const size_t bytes = (size_t)frames * y_t * x_t * 4;
  float *out, *img; float2 *iq;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess || hipMalloc(&img, (size_t)frames * 480000 * 4) != hipSuccess ||
      hipMalloc(&iq, (size_t)frames * S * 8) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  (void)hipMemset(iq, 0, (size_t)frames * S * 8);
  const size_t lds = (size_t)(8 * y_t + 9 * y_t) * 4;
  (void)hipFuncSetAttribute((const void *)k_strip<W, IMG, IQR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = ((frames * strips + 7) / 8) * 8;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k_strip<W, IMG, IQR><<<grid, 576, lds>>>(iq, out, img, y_t, x_t, S, strips, frames, 1.0f);
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_strip<W, IMG, IQR><<<grid, 576, lds>>>(iq, out, img, y_t, x_t, S, strips, frames, 1.0f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double alg = (double)bytes + (IMG ? frames * 480000.0 * 4 : 0.0) + (IQR ? frames * (double)S * 8 : 0.0);
  printf("%-58s %7.1f us  %6.2f TB/s algorithmic (%.1f MB)  [%s]\n", name, ms * 1e3, alg / (ms * 1e-3) / 1e12, alg / 1e6, hipGetErrorString(hipGetLastError()));
  (void)hipFree(out); (void)hipFree(img); (void)hipFree(iq);
}

// ---- round 6, the cheaper alternative: the same tile, TWO consecutive lines per lane ---------------------------------------------
// 256 threads = 4 wavefronts side by side, each 128 lines (lane j: lines 2j, 2j + 1) x 32 pixel columns: a wave-store is one
// dwordx2 per lane = 512 contiguous bytes (5 line requests instead of the 6 of two stacked dword wave-stores), and a column's
// image rows -- 68 of them -- leave as one contiguous 272-byte run (rows compacted across lanes) instead of two 136-byte ones.
template <int W, bool IMG>
__global__ __launch_bounds__(256, 8) void k_skeleton2(float *__restrict__ out, float *__restrict__ img, int y_t, int x_t, int tiles_l, int tiles_p,
                                                     int frames, float seed) {
  const unsigned xcd = blockIdx.x, ul = blockIdx.z;
  const int tl = (int)blockIdx.y;
  const unsigned U = (unsigned)(frames * tiles_p);
  const unsigned u = ((((ul >> 1) << 3) + xcd) << 1) + (ul & 1u);
  if (u >= U) return;
  const int f = (int)(u / (unsigned)tiles_p), tp = (int)(u % (unsigned)tiles_p);
  const int wh = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l0 = tl * 127, p0 = tp * 127;
  typedef float v2f __attribute__((ext_vector_type(2)));
  float acc = seed + lane, acc2 = seed - lane;
  const int l = l0 + 2 * lane;
  float *base = out + (size_t)f * y_t * x_t;
  float *ib = img + (size_t)f * 480000;
  const int row0 = (int)((l0 + 0.5) * (600.0 / y_t));          // first output row of the tile; the wave's 68 rows are contiguous
  for (int i = 0; i < 32; ++i) {
    const int p = p0 + wh * 32 + i;
    if (p >= x_t) break;
#pragma unroll
    for (int w = 0; w < W; ++w) { acc = __fmaf_rn(acc, 1.0001f, 0.5f); acc2 = __fmaf_rn(acc2, 1.0001f, 0.25f); }
    if (l + 1 < y_t) {
      float *dst = base + (size_t)p * y_t + l;
      v2f d = {acc, acc2};
      asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(d) : "memory");
    }
    if (IMG) {
      const int c = (p * 800) / x_t;
      if (c != ((p - 1) * 800) / x_t && c < 800) {
        if (row0 + lane < 600) { float *d2 = ib + (size_t)c * 600 + row0 + lane; asm volatile("global_store_dword %0, %1, off" ::"v"(d2), "v"(acc) : "memory"); }
        if (lane < 4 && row0 + 64 + lane < 600) { float *d3 = ib + (size_t)c * 600 + row0 + 64 + lane; asm volatile("global_store_dword %0, %1, off" ::"v"(d3), "v"(acc2) : "memory"); }
      }
    }
  }
}

// FOUR consecutive lines per lane: dwordx4 raster stores (1024 contiguous bytes per wave-store: 9 line requests instead of the 12 of
// four dword wave-stores), tile = 255 lines x 127 columns, 256 threads; an event's 136 image rows leave as three contiguous runs.
template <int W, bool IMG, bool IQR = false>
__global__ __launch_bounds__(256, 4) void k_skeleton4(float *__restrict__ out, float *__restrict__ img, int y_t, int x_t, int tiles_l, int tiles_p,
                                                     int frames, float seed, const float2 *__restrict__ iq = nullptr, int S = 0) {
  __shared__ float smp[IQR ? 256 * 19 : 1];
  const unsigned xcd = blockIdx.x, ul = blockIdx.z;
  const int tl = (int)blockIdx.y;
  const unsigned U = (unsigned)(frames * tiles_p);
  const unsigned u = ((((ul >> 1) << 3) + xcd) << 1) + (ul & 1u);
  if (u >= U) return;
  const int f = (int)(u / (unsigned)tiles_p), tp = (int)(u % (unsigned)tiles_p);
  const int wh = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l0 = tl * 255, p0 = tp * 127;
  typedef float v4f __attribute__((ext_vector_type(4)));
  float a0 = seed + lane, a1 = seed - lane, a2 = seed * 2 + lane, a3 = seed * 3 - lane;
  if (IQR) {
    const double sf = (double)S / ((double)y_t * x_t);
    for (int r = threadIdx.x >> 2; r < 256; r += 64) {      // 4 lanes per line, up to 5 samples each
      const int ll = min(l0 + r, y_t - 1);
      const unsigned k0 = (unsigned)(((double)ll * x_t + p0) * sf);
      for (int t = 0; t < 5; ++t) {
        const int j = (threadIdx.x & 3) * 5 + t;
        if (j < 19) {
          const float2 z = iq[(size_t)f * S + min(k0 + (unsigned)j, (unsigned)S - 1u)];
          smp[r * 19 + j] = __fsqrt_rn(z.x * z.x + z.y * z.y);
        }
      }
    }
    __syncthreads();
  }
  float pos = (float)lane * 0.37f;
  const int l = l0 + 4 * lane;
  float *base = out + (size_t)f * y_t * x_t;
  float *ib = img + (size_t)f * 480000;
  const int row0 = (l0 * 600) / y_t;
  for (int i = 0; i < 32; ++i) {
    const int p = p0 + wh * 32 + i;
    if (p >= x_t) break;
#pragma unroll
    for (int w = 0; w < W; ++w) { a0 = __fmaf_rn(a0, 1.0001f, 0.5f); a1 = __fmaf_rn(a1, 1.0001f, 0.25f); a2 = __fmaf_rn(a2, 1.0001f, 0.125f); a3 = __fmaf_rn(a3, 1.0001f, 0.0625f); }
    if (IQR) {
      pos += 0.115f;
      const int kk = min((int)pos & 15, 17), rr = 4 * lane * 19;
      a0 += smp[rr + kk]; a1 += smp[rr + 19 + kk]; a2 += smp[rr + 38 + kk]; a3 += smp[rr + 57 + kk];
    }
    if (l + 3 < y_t) {
      float *dst = base + (size_t)p * y_t + l;
      v4f d = {a0, a1, a2, a3};
      asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(d) : "memory");
    } else {
      for (int k = 0; k < 4; ++k) if (l + k < y_t) base[(size_t)p * y_t + l + k] = a0;
    }
    if (IMG) {
      const int c = (p * 800) / x_t;
      if (c != ((p - 1) * 800) / x_t && c < 800) {
        float *d2 = ib + (size_t)c * 600 + row0 + lane;
        if (row0 + lane < 600) asm volatile("global_store_dword %0, %1, off" ::"v"(d2), "v"(a0) : "memory");
        if (row0 + 64 + lane < 600) asm volatile("global_store_dword %0, %1, off" ::"v"(d2 + 64), "v"(a1) : "memory");
        if (lane < 8 && row0 + 128 + lane < 600) asm volatile("global_store_dword %0, %1, off" ::"v"(d2 + 128), "v"(a2) : "memory");
      }
    }
  }
}

template <int W, bool IMG, bool IQR = false>
static void run_skeleton4(const char *name) {
  const int y_t = 1125, x_t = 2576, frames = 30, S = 333333;
  float2 *iq = nullptr;
  if (hipMalloc(&iq, (size_t)frames * S * 8) != hipSuccess) exit(1);
  (void)hipMemset(iq, 0, (size_t)frames * S * 8);
  const int tiles_l = (y_t - 2) / 255 + 1, tiles_p = (x_t - 2) / 127 + 1;
  const size_t bytes = (size_t)frames * y_t * x_t * 4;
  float *out, *img;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess || hipMalloc(&img, (size_t)frames * 480000 * 4 + 4096) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  const unsigned units = frames * tiles_p, upx = (units + 15) / 16 * 2;
  const dim3 grid(8, tiles_l, upx);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k_skeleton4<W, IMG, IQR><<<grid, 256>>>(out, img, y_t, x_t, tiles_l, tiles_p, frames, 1.0f, iq, S);
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_skeleton4<W, IMG, IQR><<<grid, 256>>>(out, img, y_t, x_t, tiles_l, tiles_p, frames, 1.0f, iq, S);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double alg = (double)bytes + (IMG ? frames * 480000.0 * 4 : 0.0) + (IQR ? frames * (double)S * 8 : 0.0);
  printf("%-58s %7.1f us  %6.2f TB/s algorithmic (%.1f MB)  [%s]\n", name, ms * 1e3, alg / (ms * 1e-3) / 1e12, alg / 1e6, hipGetErrorString(hipGetLastError()));
  (void)hipFree(out); (void)hipFree(img); (void)hipFree(iq);
}

template <int W, bool IMG>
static void run_skeleton2(const char *name) {
  const int y_t = 1125, x_t = 2576, frames = 30;
  const int tiles_l = (y_t - 2) / 127 + 1, tiles_p = (x_t - 2) / 127 + 1;
  const size_t bytes = (size_t)frames * y_t * x_t * 4;
  float *out, *img;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess || hipMalloc(&img, (size_t)frames * 480000 * 4) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  const unsigned units = frames * tiles_p, upx = (units + 15) / 16 * 2;
  const dim3 grid(8, tiles_l, upx);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k_skeleton2<W, IMG><<<grid, 256>>>(out, img, y_t, x_t, tiles_l, tiles_p, frames, 1.0f);
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_skeleton2<W, IMG><<<grid, 256>>>(out, img, y_t, x_t, tiles_l, tiles_p, frames, 1.0f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double alg = (double)bytes + (IMG ? frames * 480000.0 * 4 : 0.0);
  printf("%-58s %7.1f us  %6.2f TB/s algorithmic (%.1f MB)\n", name, ms * 1e3, alg / (ms * 1e-3) / 1e12, alg / 1e6);
  (void)hipFree(out); (void)hipFree(img);
}

template <int W, bool IMG, bool IQR>
static void run_skeleton(const char *name) {
  const int y_t = 1125, x_t = 2576, frames = 30, S = 333333;
  const int tiles_l = (y_t - 2) / 126 + 1, tiles_p = (x_t - 2) / 127 + 1;
  const size_t bytes = (size_t)frames * y_t * x_t * 4;
  float *out, *img; float2 *iq;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess || hipMalloc(&img, (size_t)frames * 480000 * 4) != hipSuccess ||
      hipMalloc(&iq, (size_t)frames * S * 8) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  (void)hipMemset(iq, 0, (size_t)frames * S * 8);
  const unsigned units = frames * tiles_p, upx = (units + 15) / 16 * 2;
  const dim3 grid(8, tiles_l, upx);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k_skeleton<W, IMG, IQR><<<grid, 512>>>(iq, out, img, y_t, x_t, S, tiles_l, tiles_p, frames, 1.0f);
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k_skeleton<W, IMG, IQR><<<grid, 512>>>(iq, out, img, y_t, x_t, S, tiles_l, tiles_p, frames, 1.0f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double alg = (double)bytes + (IMG ? frames * 480000.0 * 4 : 0.0) + (IQR ? frames * (double)S * 8 : 0.0);
  printf("%-58s %7.1f us  %6.2f TB/s algorithmic (%.1f MB)\n", name, ms * 1e3, alg / (ms * 1e-3) / 1e12, alg / 1e6);
  (void)hipFree(out); (void)hipFree(img); (void)hipFree(iq);
}

int main() {
  printf("---- skeleton of the raster launch (C2: 30 frames, 2576 x 1125, tiles of 127 x 127, 512 threads, kernel's grid order)\n");
  run_skeleton<0, false, false>("raster stores only");
  run_skeleton<12, false, false>("raster stores + 12 fma/pixel");
  run_skeleton<12, false, true>("raster stores + IQ staging + LDS read + 12 fma/pixel");
  run_skeleton<12, true, false>("raster + image stores + 12 fma/pixel");
  run_skeleton<0, true, true>("raster + image stores + IQ staging, no filler");
  run_skeleton<12, true, true>("raster + image stores + IQ staging + 12 fma/pixel");
  run_skeleton<24, true, true>("raster + image stores + IQ staging + 24 fma/pixel");
  printf("---- the same tile with two lines per lane (dwordx2 raster stores, image rows as one run)\n");
  run_skeleton2<0, false>("2 lines/lane: raster stores only");
  run_skeleton2<6, false>("2 lines/lane: raster stores + 12 fma/pixel");
  run_skeleton2<6, true>("2 lines/lane: raster + image stores + 12 fma/pixel");
  run_skeleton<12, true, false>("(1 line/lane again) raster + image stores + 12 fma/pixel");
  run_skeleton4<0, false>("4 lines/lane: raster stores only");
  run_skeleton4<3, false>("4 lines/lane: raster stores + 12 fma/pixel");
  run_skeleton4<3, true>("4 lines/lane: raster + image stores + 12 fma/pixel");
  run_skeleton2<6, true>("2 lines/lane: raster + image stores + 12 fma/pixel");
  run_skeleton<12, true, false>("(1 line/lane again) raster + image stores + 12 fma/pixel");
  run_skeleton4<3, true>("4 lines/lane: raster + image stores + 12 fma/pixel");
  run_skeleton4<3, true, true>("4 lines/lane: raster + image + IQ staging + 12 fma/pixel");
  run_skeleton<12, true, true>("(1 line/lane) raster + image + IQ staging + 12 fma/pixel");
  run_skeleton4<3, true, true>("4 lines/lane: raster + image + IQ staging + 12 fma/pixel");
  run_skeleton<12, true, true>("(1 line/lane) raster + image + IQ staging + 12 fma/pixel");
  run_skeleton4<6, true, true>("4 lines/lane: raster + image + IQ staging + 24 fma/pixel");
  printf("---- whole-column strips through an LDS image of the output (576 threads, 32 columns x all lines, 8-column sub-tiles)\n");
  run_strip<0, false, 0>("strip: raster stores only");
  run_strip<12, false, 0>("strip: raster stores + 12 fma/pixel");
  run_strip<12, false, 1>("strip: raster + IQ staging (no LDS read) + 12 fma/pixel");
  run_strip<12, false, 2>("strip: raster + LDS read per pixel (no staging) + 12 fma");
  run_strip<12, false, 3>("strip: raster + IQ staging + LDS read + 12 fma/pixel");
  run_strip<12, true, 3>("strip: raster + image + IQ staging + 12 fma/pixel");
  run_strip<24, true, 3>("strip: raster + image + IQ staging + 24 fma/pixel");
  printf("---- store shapes (round 2-5)\n");
  for (int y : {1125, 1152}) {
    run<1, 0>("dword    (1 line/lane), no VALU", y);
    run<2, 0>("dwordx2  (2 lines/lane), no VALU", y);
    run<4, 0>("dwordx4  (4 lines/lane), no VALU", y);
    run<1, 12>("dword    + 12 fma/pixel", y);
    run<2, 24>("dwordx2  + 24 fma/pixel", y);
    run<4, 48>("dwordx4  + 48 fma/pixel", y);
    run_sheared<0>("dword sheared (aligned), no VALU", y);
    run_sheared<12>("dword sheared + 12 fma/pixel", y);
  }
  return 0;
}
