// common.h -- context, workspace and launch helpers shared by every translation unit
// of libtempest_hip.so.  gfx950 only; no CPU fallback anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/tempest_hip.h"

namespace tsdr {

// workspace slots: one growable device buffer each, owned by the context
enum Slot {
  WS_IN = 0,      // host-API staging of inputs
  WS_OUT,         // host-API staging of outputs
  WS_AUX,         // host-API staging (second output / state)
  WS_ABS,         // |IQ| or power scratch
  WS_RASTER,      // fallback raster when the fused path cannot tile
  WS_IMG,         // per-frame 600x800 images of one buffer
  WS_PROJ,        // raw + filtered projections, sums
  WS_KEYS,        // packed argmax keys per frame
  WS_FFT_A,       // FFT ping
  WS_FFT_B,       // FFT pong
  WS_FFT_C,       // Bluestein / correlation scratch
  WS_FFT_D,
  WS_MISC,
  WS_GUARD,       // sync guard: flags + per-workgroup top-2 column maxima (guard.h)
  WS_COUNT
};

struct ProfRec {
  const char *name;
  hipEvent_t e0, e1;
};

// How the frame kernels' loaders read one IQ sample: interleaved ComplexF32 (the reference's recv! buffers), or interleaved
// int16 pairs as SDR hardware delivers them, turned into ComplexF32(re, im) * scale in the loader itself (tsdr_frames_sc16*:
// int16 slots are never expanded in HBM -- half the IQ bytes the image kernel reads)
struct IqFmt { int sc16 = 0; float scale = 1.0f; };

struct TwTable {   // two-level table of W_N^e = exp(-2*pi*i*e/N), N = 2^logN
  int logN = 0, h = 0;
  float2 *lo = nullptr;  // W_N^j, j < 2^h
  float2 *hi = nullptr;  // W_N^(j*2^h), j < 2^(logN-h)
};

struct BluesteinPlan {
  size_t n = 0, L = 0;
  float2 *chirp = nullptr;  // exp(-i*pi*k^2/n), k < n
  float2 *bfft = nullptr;   // FFT_L of the wrapped conj chirp
};

}  // namespace tsdr

struct tsdr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  hipStream_t launch_stream = nullptr;  // stream TSDR_LAUNCH targets (== stream)
  std::string err;
  int cu_count = 0;
  int precision = TSDR_FAST;  // tsdr_precision
  tsdr::IqFmt iq_fmt;         // set for the duration of a tsdr_frames_sc16* call
  // development switches (tsdr_set_option; environment variables of the same upper-case names are read ONCE, in tsdr_create)
  int opt_ac_mixed = 1;     // autocorrelation of n = 2*(2^a3^b5^c) samples: native mixed-radix route (0: zero-padded power of two)
  int opt_fft_no_mix2 = 0;  // 1: every mixed-radix factor through the generic LDS-stage kernel
  int opt_fft_big = 1;      // mixed-radix planner may use factors of 500 / 1000 / 2000 (three-register-step kernels)
  int opt_vsync_current_sy = 0;  // 1: vsync / the frame loop return s_y of the CURRENT image (the corrected ordering, off by default)
  int opt_fast_walk_only = 0;    // 1: the raster-free FAST frame path runs the raster walk with out == null (round 2's route; A/B)
  int opt_down_spp_max_pct = 200; // raster-free FAST route: k_down_fused up to this many samples per raster pixel (percent), the raster walk above
                                  // (round 4: 50 -> 200 once wide rows stage with 16 loads in flight: C3 0.40 -> 0.24 ms per buffer)
  int opt_down_xcd = 1;      // raster-free FAST kernel: XCD-aware tile order
  int opt_raster_rec4 = -1;  // FAST raster walk: staged |IQ| as plain f32 samples (4 bytes) instead of {a, slope hi, slope lo} records (16): 0 = never (the A/B), else wherever the f32 walk forms the images
  int opt_raster_v4 = 0;     // FAST frame loop with rasters at C2-like geometry: four lines per lane (k_raster_fast4: dwordx4 raster stores, compacted image
                             // rows) -- 32: four wavefronts of 32 pixel columns per tile, other > 0: eight of 16.  Round 6's A/B: loses by 12-20 %, off
  int opt_raster_split = 0;  // FAST frame loop with rasters: 1 = sheared raster-only kernel + raster-free image kernel, 2 = the same unsheared (A/B)
  int opt_ac_fuse_mid = 1;  // autocorrelation: last forward pass + power spectrum + first inverse pass as one launch
  // sync guard of the FAST frame loop (guard.h): relative top-2 margin below which a frame is re-evaluated exactly
  // (0: off); running totals {frames checked, frames re-evaluated} on the device
  float guard_thr = 2e-5f;
  unsigned *guard_sync[4] = {};             // work-queue words of the guard kernel (zero between launches), one set per pipeline lane
  unsigned long long *guard_stats = nullptr;
  // adaptive route (option "sync_guard_auto"): when more than guard_auto_hi of the recent frames were flagged, re-evaluating
  // them one by one costs more than running whole buffers in the exact sequence, so the FAST frame loop does that until the
  // share (still counted, on the exact statistics) falls below guard_auto_lo.  Decided on the host, DETERMINISTICALLY (round 5):
  // every guarded call's guard launch writes ITS OWN {frames, flagged} into a pinned ring entry tagged with the call's
  // sequence number; the decision for call k folds the entries of calls <= k - kGuardLag, in submission order, waiting for
  // them if need be (call k - 3 is long complete in any steady state: it is the pipeline's own slot-reuse distance).  The same
  // sequence of buffers therefore takes the same route on every run, whatever the host / GPU timing.
  int opt_guard_auto = 1;
  float guard_auto_hi = 0.15f, guard_auto_lo = 0.05f;
  int opt_guard_nowait = 0;                  // measurement switch (TSDR_GUARD_NOWAIT): the decision never waits for an entry
  static constexpr int kGuardRing = 16, kGuardLag = 3;
  unsigned long long *guard_ring = nullptr;  // pinned: tag (seq + 1, 16 bits) << 48 | frames << 24 | flagged
  unsigned long long guard_seq = 0;          // guarded calls issued so far (the next one's sequence number)
  unsigned long long guard_consumed = 0;     // entries folded into the window so far
  unsigned guard_win_c = 0, guard_win_f = 0; // frames checked / flagged since the last decision
  bool guard_exact_now = false;
  unsigned long long guard_auto_buffers = 0, guard_auto_switches = 0;
  size_t guard_last_off = (size_t)-1;       // byte offset inside WS_GUARD of the most recent guarded call's top-2 records (tsdr_sync_guard_margins)
  int guard_last_frames = 0, guard_last_nbx = 0, guard_last_nby = 0;
  struct Buf { void *p = nullptr; size_t cap = 0; } ws[tsdr::WS_COUNT];
  // profiling
  bool prof_on = false;
  std::vector<tsdr::ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  struct ProfAgg { std::string name; double ms; long long n; };
  std::vector<ProfAgg> prof_agg;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // FFT state
  float2 *tw_small = nullptr;  // W_4096^e, e < 4096
  std::map<int, tsdr::TwTable> tw;
  std::map<unsigned, float2 *> twg;  // mixed-radix FFT: W_{R*Rn}^(col*k) tables of the last strided pass, key R << 16 | Rn
  std::map<size_t, tsdr::BluesteinPlan> blu;
  // tsdr_argmax_d: two device key slots (each launch clears the other one) and a pinned host word for the readback
  unsigned long long *amax_keys = nullptr, *amax_host = nullptr, *amax_host_dev = nullptr;
  int amax_slot = 0;
  bool amax_dirty = false;   // a fused findmax may have left keys in the slot words (autocorr.hip:amax_begin clears them)
  unsigned long long amax_seq = 0;
  // frame loop pipelined across buffers (tsdr_frames_submit_d, frames.hip): successive buffers on internal HIP streams so that
  // the latency-bound tail of one (vsync statistics, sync guard, shift + IIR) runs beside the image launch of the next.
  // WHICH streams, and in which arrangement, is measured, not assumed (frames.hip:pipe_pick): how well two HIP streams of a
  // process overlap depends on the hardware queues they were mapped to, i.e. on what else the process created before.
  static constexpr int kPipeSlots = 3;   // image / key / projection / guard-record slots in rotation
  static constexpr int kPoolN = 6, kPoolH = 2;   // candidate streams: normal priority / highest priority
  static constexpr int kTuneCands = 8, kTrial = 15;
  unsigned long long pipe_n = 0;    // submissions since the last flush point
  unsigned long long pipe_seq = 0;  // submissions since the lanes were last run empty (slot = pipe_seq % kPipeSlots)
  // what the slot offsets of the submissions in flight were computed from: frames per buffer (the image slots are that many
  // frames apart), the tile plan of (S, y_t, x_t, raster or not, precision) for the projection sums, the SyncXY object's block
  // counts for the guard records.  A change of any of them runs the pipeline empty first.
  struct PipeKey {
    size_t nb = 0, S = 0; int y_t = 0, x_t = 0, raster = -1, prec = -1, align = -1, sc16 = 0; const void *sync = nullptr;
    bool operator==(const PipeKey &o) const {
      return nb == o.nb && S == o.S && y_t == o.y_t && x_t == o.x_t && raster == o.raster && prec == o.prec && align == o.align && sc16 == o.sc16 && sync == o.sync;
    }
    // the same loop at a slightly different geometry (GUI.jl:492-506: the interactive y_t / x_t corrections move one line at a
    // time): same frames per buffer, raster or not, precision, alignment, input format; S and the raster size within 10 %.
    // What was measured for one holds for the other (the launches have the same shapes and durations within a few percent).
    bool near(const PipeKey &o) const {
      if (!(nb == o.nb && raster == o.raster && prec == o.prec && align == o.align && sc16 == o.sc16)) return false;
      const double s = (double)S / (double)(o.S ? o.S : 1), p = ((double)y_t * x_t) / ((double)o.y_t * o.x_t > 0 ? (double)o.y_t * o.x_t : 1.0);
      return s > 0.9 && s < 1.1 && p > 0.9 && p < 1.1;
    }
  } pipe_key;
  hipStream_t pool[kPoolN + kPoolH] = {};  // created at the first submission (frames.hip:pipe_init)
  hipStream_t lane[4] = {};         // the streams of the arrangement in use (members of pool): [0], [1], [3] equal lanes / [0] image lane, [2] tail lane
  hipEvent_t ev_img[kPipeSlots] = {}, ev_tail[kPipeSlots] = {};  // recorded behind a slot's image launch / its shift + IIR
  bool ev_tail_used[kPipeSlots] = {};
  hipEvent_t lane_in = nullptr;     // "inputs ready" point of the context's stream
  int pipe_last_slot = -1;          // slot of the latest submission: its tail is behind everything submitted
  int pipe_cand_now = -1;           // arrangement (index into frames.hip:kCands) of the submissions in flight
  bool pipe_one_lane = false;       // ... is the one-stream arrangement (no event per buffer; pipe_drain records one when asked)
  int pipe_lane = 0;                // lane of the call being enqueued (per-lane guard queue words, workspace raster)
  int opt_pipe_mode = -1;           // -1: the measured choice; 0: image lane + tail lane; 1: whole buffers alternate between opt_pipe_lanes
                                    // equal lanes, only shift + IIR chained; 2: one internal stream (the sequential order)
  int opt_pipe_lanes = 2;           // equal lanes of the forced arrangement 1: 2 or 3
  int opt_pipe_priority = 1;        // forced arrangement 0: the tail lane is a stream of the highest priority
  int opt_pipe_ext_event = 1;       // 1: a buffer's tail event rides on its shift + IIR dispatch (hipExtLaunchKernelGGL's stop event) instead of
                                    // a marker packet of its own behind it
  hipEvent_t launch_stop_ev = nullptr;  // set by the pipeline for the next shift + IIR launch, cleared by it
  int opt_pipe_dev_events = 1;      // the pipeline's hand-over events release to device scope (TSDR_PIPE_DEV_EVENTS=0: the default system scope; A/B)
  int opt_pipe_tune = 1;            // 0: "pipe_mode" -1 means arrangement 0 with rasters, 1 without (rounds 1-4), nothing is measured
  struct PipeTune {                 // the measured choice for one PipeKey
    PipeKey key; int state = 0;     // 0: nothing measured; 1: trials running; 2: settled
    int cand = 0, pos = 0, chosen = -1, round = 0;   // round 0: warm-up trial of candidate 0; 1, 2: the two measured passes
    int interrupts = 0;             // times another configuration cut into this one's trials (4: settle on the sequential order)
    bool inherited = false;         // settled by taking over a neighbouring configuration's choice (PipeKey::near), not by trials
    float ms[kTuneCands] = {};      // mean interval between the tails of successive buffers, per arrangement
  } tune;
  int opt_pipe_pin = -1;            // "pipe_pin" k >= 0: arrangement k of frames.hip:kCands, nothing measured; -1: not pinned
  bool tune_force = false;          // "pipe_measure" 1: the next configuration is measured whatever is known about it or its neighbours
  unsigned long long tune_runs = 0; // measurements (full sets of trials) started on this context so far
  std::vector<PipeTune> tune_done;  // settled measurements of earlier configurations (a caller that goes back to one -- GUI.jl's
                                    // y_t / x_t corrections, a raster asked for now and then -- does not measure it again); <= 16
  hipEvent_t tune_ev[kTrial] = {};
  // Every host-side wait of this library is bounded (round 6): a stream that does not complete within opt_wait_ms returns
  // TSDR_EHIP with the waiting stage in tsdr_last_error instead of holding the caller's thread (a GUI task inside a ccall) for
  // good.  0: unbounded (plain hipStreamSynchronize).  Option "wait_ms" / TSDR_WAIT_MS.
  int opt_wait_ms = 30000;
  hipEvent_t wait_ev = nullptr;     // the marker a bounded wait polls (created on first use)
  unsigned long long wait_timeouts = 0;   // bounded waits that gave up (tsdr_wait_stats)
  unsigned long long guard_uncounted = 0; // guard ring entries that never arrived within their bound / were overwritten unread
  bool guard_late = false;          // the last awaited guard entry timed out: later ones get a short bound until one arrives
  int opt_beta_waves = 4;           // wavefronts per k_beta workgroup (4 or 8): alone the two tie; beside the pipeline's image kernel a 256-thread
                                    // workgroup fits the holes its retiring workgroups leave (raster-free 357 k vs 309 k frames/s)

  void *scratch(int slot, size_t bytes);  // nullptr on failure (err set)
};

namespace tsdr {

int set_err(tsdr_ctx *ctx, int status, const char *fmt, ...);
int hip_fail(tsdr_ctx *ctx, hipError_t e, const char *what);
// A kernel that asks for more than 64 KiB of dynamic LDS has to be opted in -- once per (device, function): the attribute
// belongs to the device's function object, and one process may hold contexts on several devices (tsdr_group_*).
int lds_opt_in(tsdr_ctx *ctx, const void *fn, size_t bytes);
// frames.hip: order the context's stream behind every buffer submitted to the pipeline (tsdr_frames_submit_d)
int pipe_drain(tsdr_ctx *ctx);
int pipe_sync_lanes(tsdr_ctx *ctx);    // bounded host-side wait for the pipeline's internal streams (TSDR_EHIP when one is stuck)
// bounded host-side waits (ctx.hip): poll a marker event, first spinning, then in short sleeps; TSDR_EHIP + message after
// ctx->opt_wait_ms.  `what` names the stage for tsdr_last_error.
int wait_stream(tsdr_ctx *ctx, hipStream_t s, const char *what);
int wait_event(tsdr_ctx *ctx, hipEvent_t e, const char *what);
void prof_begin(tsdr_ctx *ctx, const char *name);
void prof_end(tsdr_ctx *ctx);

#define TSDR_HIP(ctx, call)                                          \
  do {                                                               \
    hipError_t _e = (call);                                          \
    if (_e != hipSuccess) return tsdr::hip_fail((ctx), _e, #call);   \
  } while (0)

// Launch a kernel on the context's stream; when profiling is on the launch is
// bracketed by its own hipEvent pair so bench.py can read per-kernel durations live.
#define TSDR_LAUNCH(ctx, kname, kernel, grid, block, shmem, ...)                              \
  do {                                                                                        \
    if ((ctx)->prof_on) tsdr::prof_begin((ctx), kname);                                       \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->launch_stream, __VA_ARGS__);               \
    if ((ctx)->prof_on) tsdr::prof_end((ctx));                                                \
    hipError_t _le = hipGetLastError();                                                       \
    if (_le != hipSuccess) return tsdr::hip_fail((ctx), _le, kname);                          \
  } while (0)

static inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

// grid for a capped, grid-strided streaming kernel of 256-thread workgroups
static inline int stream_grid(tsdr_ctx *ctx, size_t work_items) {
  size_t blocks = ceil_div(work_items, 256);
  size_t cap = (size_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// Host-pointer wrapper: stage `in` to the device, run the device-pointer core on the
// context's stream, copy `out` back, synchronise.  run(din, dout) returns a tsdr_status.
template <typename F>
static inline int host_map(tsdr_ctx *ctx, const void *in, size_t in_bytes, void *out, size_t out_bytes, F run) {
  if (!ctx) return TSDR_EINVAL;
  if ((in_bytes && !in) || (out_bytes && !out)) return TSDR_EINVAL;
  void *din = ctx->scratch(WS_IN, in_bytes);
  void *dout = ctx->scratch(WS_OUT, out_bytes);
  if (!din || !dout) return TSDR_ENOMEM;
  if (in_bytes) TSDR_HIP(ctx, hipMemcpyAsync(din, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  int rc = run(din, dout);
  if (rc) return rc;
  if (out_bytes) TSDR_HIP(ctx, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  return wait_stream(ctx, ctx->stream, "host wrapper: results");
}
static inline int ilog2(size_t v) { int l = 0; while ((size_t(1) << l) < v) ++l; return l; }
static inline bool is_pow2(size_t v) { return v && !(v & (v - 1)); }

// ---- imresize coordinate helpers (shared host/device; mirrors oracle resize_axis /
// resize_coord / lin_pos, i.e. ImageTransformations.imresize! + Interpolations Linear) ----
struct RsAxis { double sf, off, n_in; };

__host__ __device__ inline RsAxis rs_axis(size_t n_in, size_t n_out) {
  RsAxis a;
  a.sf = (double)n_in / (double)n_out;
  a.off = (1.0 - 0.5) - a.sf * (1.0 - 0.5);
  a.n_in = (double)n_in;
  return a;
}

#ifdef __HIPCC__
// x = sf*i + off with two roundings (no FMA), clamped to [1, n_in]; returns the 0-based
// left sample index and the f64 weight of the right sample.
__device__ inline unsigned rs_pos(const RsAxis &a, double i1, double &delta) {
  double x = __dadd_rn(__dmul_rn(a.sf, i1), a.off);
  x = fmax(x, 1.0);
  x = fmin(x, a.n_in);
  double xf = floor(x);
  if (xf > a.n_in - 1.0) xf -= 1.0;
  delta = x - xf;
  return (unsigned)xf - 1u;
}
// The same for positions known to lie strictly inside (1, n_in): no clamps, no end-of-signal case -- the identical x, floor and
// delta with four f64 instructions fewer.  Callers establish the precondition per tile (k_raster_tile: every pixel of a tile
// that does not hold the first or last few pixels of the frame).
__device__ inline unsigned rs_pos_inner(const RsAxis &a, double i1, double &delta) {
  const double x = __dadd_rn(__dmul_rn(a.sf, i1), a.off);
  const double xf = floor(x);
  delta = x - xf;
  return (unsigned)xf - 1u;
}
// (1-d)*a + d*b in f64 (two products, one sum, no FMA), rounded once to f32
__device__ inline float rs_blend(float a, float b, double d) {
  double v = __dadd_rn(__dmul_rn(1.0 - d, (double)a), __dmul_rn(d, (double)b));
  return (float)v;
}
// |re + i*im| = f32( sqrt_f64( re^2 + im^2 ) ), the squares exact in f64
__device__ inline float abs_c(float re, float im) {
  double s = __dadd_rn(__dmul_rn((double)re, (double)re), __dmul_rn((double)im, (double)im));
  float r = (float)sqrt(s);
  if (isinf(re) || isinf(im)) r = INFINITY;
  return r;
}
// one IQ sample of a frame's buffer (k in samples); sc16: the same product the ring's expansion kernel forms
__device__ inline float2 ld_iq(const float *__restrict__ src, unsigned k, const IqFmt &f) {
  if (f.sc16) {
    const short2 v = reinterpret_cast<const short2 *>(src)[k];
    return make_float2(__fmul_rn((float)v.x, f.scale), __fmul_rn((float)v.y, f.scale));
  }
  return reinterpret_cast<const float2 *>(src)[k];
}
// The same with the format fixed at compile time (IQF_CF32 / IQF_SC16: the FAST image kernels, whose register allocation and
// instruction stream must not pay for the other format) or read from the launch's parameters (IQF_RT: every other reader)
enum { IQF_CF32 = 0, IQF_SC16 = 1, IQF_RT = 2 };
template <int IQF>
__device__ inline float2 ld_iq_as(const float *__restrict__ src, unsigned k, const IqFmt &f) {
  if (IQF == IQF_CF32) return reinterpret_cast<const float2 *>(src)[k];
  if (IQF == IQF_SC16) {
    const short2 v = reinterpret_cast<const short2 *>(src)[k];
    return make_float2(__fmul_rn((float)v.x, f.scale), __fmul_rn((float)v.y, f.scale));
  }
  return ld_iq(src, k, f);
}
template <int IQF>
__host__ __device__ inline size_t iq_floats_as(const IqFmt &f) { return IQF == IQF_CF32 ? 2 : IQF == IQF_SC16 ? 1 : (f.sc16 ? 1 : 2); }
// floats per IQ sample in a buffer of that format (frame strides are counted in samples)
__host__ __device__ inline size_t iq_floats(const IqFmt &f) { return f.sc16 ? 1 : 2; }
__device__ inline float abs2_c(float re, float im) { return __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)); }
#endif

}  // namespace tsdr
