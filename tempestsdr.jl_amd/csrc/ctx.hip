// ctx.hip -- context lifetime, workspace, error reporting, HIP-event measurement.
#include <cstdarg>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <utility>

#include "common.h"

namespace tsdr {

int set_err(tsdr_ctx *ctx, int status, const char *fmt, ...) {
  if (ctx) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx->err = buf;
  }
  // HIP keeps a failed call's status as the thread's "last error" until someone reads it; TSDR_LAUNCH reads it after every
  // launch, so an allocation that failed here (reported as TSDR_ENOMEM) would otherwise fail the NEXT, valid call too
  (void)hipGetLastError();
  return status;
}

int hip_fail(tsdr_ctx *ctx, hipError_t e, const char *what) {
  return set_err(ctx, TSDR_EHIP, "%s: %s", what, hipGetErrorString(e));
}

int lds_opt_in(tsdr_ctx *ctx, const void *fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, size_t> opted;   // (device, function) -> bytes opted in to so far
  std::lock_guard<std::mutex> g(mu);
  size_t &have = opted[{ctx->device, fn}];
  if (bytes <= have) return TSDR_OK;
  TSDR_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return TSDR_OK;
}

// ---- bounded host-side waits -----------------------------------------------------------------------------------------
// One marker event behind whatever the stream holds, then polls of that event: a tight spin for the first 25 ms (results of a
// 15 us launch, or the end of a bench region, must not pay a sleep), then 50 us sleeps.  hipStreamSynchronize has no time limit: a lane held behind something that
// never completes would hold the caller's thread with it.
int wait_event(tsdr_ctx *ctx, hipEvent_t e, const char *what) {
  if (ctx->opt_wait_ms <= 0) { TSDR_HIP(ctx, hipEventSynchronize(e)); return TSDR_OK; }
  using clk = std::chrono::steady_clock;
  clk::time_point t0;
  bool slow = false;   // past the first 25 ms: one look per 50 us sleep (a thread that waits long must not burn a core; a short wait --
                       // the end of a timed region among them -- is seen within a microsecond, like hipStreamSynchronize's own spin)
  for (unsigned it = 1;; ++it) {
    const hipError_t q = hipEventQuery(e);
    if (q == hipSuccess) return TSDR_OK;
    if (q != hipErrorNotReady) return hip_fail(ctx, q, what);
    (void)hipGetLastError();   // (hipErrorNotReady is a status, not a failure for the next launch check to find)
    if (!slow && (it & 0x3Fu) != 0) continue;
    const auto now = clk::now();
    if (it == 0x40u) { t0 = now; continue; }
    const auto waited = now - t0;
    if (waited > std::chrono::milliseconds(ctx->opt_wait_ms)) {
      ++ctx->wait_timeouts;
      return set_err(ctx, TSDR_EHIP, "%s: the stream did not complete within %d ms (bounded host wait; option wait_ms)", what, ctx->opt_wait_ms);
    }
    if (waited > std::chrono::milliseconds(25)) { slow = true; std::this_thread::sleep_for(std::chrono::microseconds(50)); }
  }
}

int wait_stream(tsdr_ctx *ctx, hipStream_t s, const char *what) {
  if (ctx->opt_wait_ms <= 0) { TSDR_HIP(ctx, hipStreamSynchronize(s)); return TSDR_OK; }
  if (!ctx->wait_ev) {
    // (an event belongs to the device that is current when it is created: the context's, whatever the caller's is)
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != ctx->device) TSDR_HIP(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipEventCreateWithFlags(&ctx->wait_ev, hipEventDisableTiming);
    if (cur >= 0 && cur != ctx->device) (void)hipSetDevice(cur);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipEventCreateWithFlags");
  }
  TSDR_HIP(ctx, hipEventRecord(ctx->wait_ev, s));
  return wait_event(ctx, ctx->wait_ev, what);
}

static hipEvent_t take_event(tsdr_ctx *ctx) {
  if (!ctx->ev_pool.empty()) {
    hipEvent_t e = ctx->ev_pool.back();
    ctx->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(tsdr_ctx *ctx, const char *name) {
  ProfRec r;
  r.name = name;
  r.e0 = take_event(ctx);
  r.e1 = take_event(ctx);
  (void)hipEventRecord(r.e0, ctx->launch_stream);
  ctx->prof.push_back(r);
}

void prof_end(tsdr_ctx *ctx) { (void)hipEventRecord(ctx->prof.back().e1, ctx->launch_stream); }

// fold finished event pairs into the per-kernel aggregate
static int prof_collect(tsdr_ctx *ctx) {
  if (ctx->prof.empty()) return TSDR_OK;
  {  // (records may sit on the pipeline's lanes)
    int rc = pipe_drain(ctx);
    if (rc) return rc;
  }
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  for (auto &r : ctx->prof) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    bool found = false;
    for (auto &a : ctx->prof_agg)
      if (a.name == r.name) { a.ms += ms; a.n += 1; found = true; break; }
    if (!found) ctx->prof_agg.push_back({r.name, (double)ms, 1});
    ctx->ev_pool.push_back(r.e0);
    ctx->ev_pool.push_back(r.e1);
  }
  ctx->prof.clear();
  return TSDR_OK;
}

}  // namespace tsdr

void *tsdr_ctx::scratch(int slot, size_t bytes) {
  Buf &b = ws[slot];
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return b.p;
  if (b.p) {
    // (the pipeline's lanes may be using the other half of this workspace; a stream that is stuck keeps it -- freeing would wait
    // for the device without a bound)
    if (tsdr::wait_stream(this, stream, "workspace growth") || tsdr::pipe_sync_lanes(this)) return nullptr;
    (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    if (slot == tsdr::WS_GUARD) guard_last_off = (size_t)-1;   // (its contents are gone)
  }
  size_t want = bytes + bytes / 8 + 4096;  // slack so slowly growing inputs do not realloc
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    tsdr::set_err(this, TSDR_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    return nullptr;
  }
  b.cap = want;
  return b.p;
}

extern "C" {

const char *tsdr_version(void) { return "tempest-hip 0.1 (gfx950)"; }

const char *tsdr_strerror(int s) {
  switch (s) {
    case TSDR_OK: return "ok";
    case TSDR_EINVAL: return "invalid argument (AssertionError/ArgumentError in the reference)";
    case TSDR_EBOUNDS: return "index out of bounds (BoundsError in the reference)";
    case TSDR_ENOMEM: return "out of device memory";
    case TSDR_EHIP: return "HIP runtime error";
    case TSDR_ENODEV: return "no usable HIP device";
    default: return "unknown status";
  }
}

const char *tsdr_last_error(tsdr_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

tsdr_ctx *tsdr_create(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    fprintf(stderr, "tempest_hip: no usable HIP device (requested %d, found %d); there is no CPU fallback\n",
            device, ndev);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  tsdr_ctx *ctx = new tsdr_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
  ctx->own_stream = true;
  ctx->launch_stream = ctx->stream;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
  (void)hipEventCreate(&ctx->t0);
  (void)hipEventCreate(&ctx->t1);
  // development switches: the environment is consulted here and nowhere else
  if (const char *e = getenv("TSDR_AC_MIXED")) ctx->opt_ac_mixed = atoi(e) != 0;
  if (const char *e = getenv("TSDR_FFT_NO_MIX2")) ctx->opt_fft_no_mix2 = atoi(e) != 0;
  if (const char *e = getenv("TSDR_AC_FUSE_MID")) ctx->opt_ac_fuse_mid = atoi(e) != 0;
  if (const char *e = getenv("TSDR_FFT_BIG")) ctx->opt_fft_big = atoi(e) != 0;
  if (const char *e = getenv("TSDR_FAST_WALK_ONLY")) ctx->opt_fast_walk_only = atoi(e) != 0;
  if (const char *e = getenv("TSDR_SYNC_GUARD_PPB")) ctx->guard_thr = (float)atoi(e) * 1e-9f;
  if (const char *e = getenv("TSDR_SYNC_GUARD_AUTO")) ctx->opt_guard_auto = atoi(e) != 0;
  if (const char *e = getenv("TSDR_PIPE_PRIORITY")) ctx->opt_pipe_priority = atoi(e);
  if (const char *e = getenv("TSDR_RASTER_SPLIT")) ctx->opt_raster_split = atoi(e);
  if (const char *e = getenv("TSDR_RASTER_REC4")) ctx->opt_raster_rec4 = atoi(e);
  if (const char *e = getenv("TSDR_RASTER_V4")) ctx->opt_raster_v4 = atoi(e);
  if (const char *e = getenv("TSDR_DOWN_XCD")) ctx->opt_down_xcd = atoi(e) != 0;
  if (const char *e = getenv("TSDR_DOWN_SPP_MAX_PCT")) ctx->opt_down_spp_max_pct = atoi(e);
  if (const char *e = getenv("TSDR_PIPE_MODE")) ctx->opt_pipe_mode = atoi(e) < 0 ? -1 : atoi(e) > 2 ? 2 : atoi(e);
  if (const char *e = getenv("TSDR_GUARD_NOWAIT")) ctx->opt_guard_nowait = atoi(e) != 0;
  if (const char *e = getenv("TSDR_PIPE_DEV_EVENTS")) ctx->opt_pipe_dev_events = atoi(e) != 0;
  if (const char *e = getenv("TSDR_PIPE_TUNE")) ctx->opt_pipe_tune = atoi(e) != 0;
  if (const char *e = getenv("TSDR_PIPE_PIN")) ctx->opt_pipe_pin = atoi(e) < 0 || atoi(e) >= tsdr_ctx::kTuneCands ? -1 : atoi(e);
  if (const char *e = getenv("TSDR_PIPE_EXT_EVENT")) ctx->opt_pipe_ext_event = atoi(e) != 0;
  if (const char *e = getenv("TSDR_PIPE_LANES")) ctx->opt_pipe_lanes = atoi(e) == 3 ? 3 : 2;
  if (const char *e = getenv("TSDR_WAIT_MS")) ctx->opt_wait_ms = atoi(e) < 0 ? 0 : atoi(e);
  if (const char *e = getenv("TSDR_BETA_WAVES")) ctx->opt_beta_waves = atoi(e) == 8 ? 8 : 4;
  return ctx;
}

void tsdr_destroy(tsdr_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)tsdr::pipe_drain(ctx);
  if (tsdr::wait_stream(ctx, ctx->stream, "tsdr_destroy") || tsdr::pipe_sync_lanes(ctx)) {
    // a stream of this context never completed: releasing its memory, streams and events would wait for the device without a
    // bound (hipFree synchronises) or pull them from under a launch.  The context is abandoned: the caller's thread returns.
    fprintf(stderr, "tempest_hip: tsdr_destroy: %s -- context abandoned, its device memory is not released\n", ctx->err.c_str());
    return;
  }
  if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
  for (auto &l : ctx->pool) if (l) (void)hipStreamDestroy(l);
  for (auto &e : ctx->tune_ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : ctx->ev_img) if (e) (void)hipEventDestroy(e);
  for (auto &e : ctx->ev_tail) if (e) (void)hipEventDestroy(e);
  if (ctx->lane_in) (void)hipEventDestroy(ctx->lane_in);
  for (auto &b : ctx->ws) if (b.p) (void)hipFree(b.p);
  for (auto &r : ctx->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  if (ctx->amax_keys) (void)hipFree(ctx->amax_keys);
  if (ctx->guard_stats) (void)hipFree(ctx->guard_stats);
  for (auto &q : ctx->guard_sync) if (q) (void)hipFree(q);
  if (ctx->guard_ring) (void)hipHostFree(ctx->guard_ring);
  if (ctx->amax_host) (void)hipHostFree(ctx->amax_host);
  if (ctx->tw_small) (void)hipFree(ctx->tw_small);
  for (auto &kv : ctx->tw) { (void)hipFree(kv.second.lo); (void)hipFree(kv.second.hi); }
  for (auto &kv : ctx->twg) (void)hipFree(kv.second);
  for (auto &kv : ctx->blu) { (void)hipFree(kv.second.chirp); (void)hipFree(kv.second.bfft); }
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int tsdr_set_stream(tsdr_ctx *ctx, void *hip_stream) {
  if (!ctx) return TSDR_EINVAL;
  {  // buffers submitted to the pipeline are ordered through the old stream
    int rc = tsdr::pipe_drain(ctx);
    if (rc) return rc;
  }
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  if (hip_stream) {
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
  } else if (!ctx->own_stream) {
    TSDR_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  ctx->launch_stream = ctx->stream;
  return TSDR_OK;
}

int tsdr_set_precision(tsdr_ctx *ctx, int mode) {
  if (!ctx || (mode != TSDR_EXACT && mode != TSDR_FAST)) return TSDR_EINVAL;
  ctx->precision = mode;
  return TSDR_OK;
}

int tsdr_get_precision(tsdr_ctx *ctx) { return ctx ? ctx->precision : TSDR_EINVAL; }

int tsdr_set_option(tsdr_ctx *ctx, const char *name, int value) {
  if (!ctx || !name) return TSDR_EINVAL;
  if (!strcmp(name, "ac_mixed")) ctx->opt_ac_mixed = value != 0;
  else if (!strcmp(name, "fft_no_mix2")) ctx->opt_fft_no_mix2 = value != 0;
  else if (!strcmp(name, "ac_fuse_mid")) ctx->opt_ac_fuse_mid = value != 0;
  else if (!strcmp(name, "fast_walk_only")) ctx->opt_fast_walk_only = value != 0;
  else if (!strcmp(name, "vsync_current_sy")) {
    int rc = tsdr::pipe_drain(ctx);   // (submitted buffers keep the old setting; later calls are ordered behind them)
    if (rc) return rc;
    ctx->opt_vsync_current_sy = value != 0;
  }
  else if (!strcmp(name, "fft_big")) ctx->opt_fft_big = value != 0;
  else if (!strcmp(name, "down_xcd")) ctx->opt_down_xcd = value != 0;
  else if (!strcmp(name, "down_spp_max_pct")) ctx->opt_down_spp_max_pct = value < 0 ? 0 : value;
  else if (!strcmp(name, "raster_split")) ctx->opt_raster_split = value < 0 || value > 2 ? 0 : value;
  else if (!strcmp(name, "raster_rec4")) ctx->opt_raster_rec4 = value < 0 ? -1 : value != 0;
  else if (!strcmp(name, "raster_v4")) {
    int rc = tsdr::pipe_drain(ctx);   // (the projection-sum layout of submitted buffers belongs to the old setting)
    if (rc) return rc;
    ctx->opt_raster_v4 = value < 0 ? 0 : value;   // 0 (default): k_raster_fast; 32: k_raster_fast4, four wavefronts of 32 pixel columns per tile; other > 0: eight of 16
  }
  else if (!strcmp(name, "beta_waves")) ctx->opt_beta_waves = value == 8 ? 8 : 4;
  else if (!strcmp(name, "pipe_mode") || !strcmp(name, "pipe_lanes") || !strcmp(name, "pipe_priority") || !strcmp(name, "pipe_tune")) {
    int rc = tsdr::pipe_drain(ctx);   // (the next submission sees another arrangement and runs the lanes empty itself)
    if (rc) return rc;
    if (name[5] == 'm') ctx->opt_pipe_mode = value < 0 ? -1 : value > 2 ? 2 : value;
    else if (name[5] == 'l') ctx->opt_pipe_lanes = value == 3 ? 3 : 2;
    else if (name[5] == 'p') ctx->opt_pipe_priority = value != 0;
    else { ctx->opt_pipe_tune = value != 0; ctx->tune = tsdr_ctx::PipeTune{}; ctx->tune_done.clear(); }   // (setting it also discards what was measured)
  }
  else if (!strcmp(name, "pipe_pin") || !strcmp(name, "pipe_measure")) {
    int rc = tsdr::pipe_drain(ctx);
    if (rc) return rc;
    if (name[5] == 'p') {   // the arrangement (index into the candidates tsdr_frames_pipeline_info lists), nothing measured; -1: back to the measured choice
      if (value >= tsdr_ctx::kTuneCands) return tsdr::set_err(ctx, TSDR_EINVAL, "pipe_pin must be -1 .. %d", tsdr_ctx::kTuneCands - 1);
      ctx->opt_pipe_pin = value < 0 ? -1 : value;
    } else if (value) {     // measure (again) at the next submission: what is known about the current configuration is dropped
      ctx->tune = tsdr_ctx::PipeTune{};
      ctx->tune_force = true;
    }
  }
  else if (!strcmp(name, "pipe_ext_event")) ctx->opt_pipe_ext_event = value != 0;
  else if (!strcmp(name, "wait_ms")) ctx->opt_wait_ms = value < 0 ? 0 : value;
  else if (!strcmp(name, "sync_guard_ppb")) {
    if (value < 0 || value > 100000000) return tsdr::set_err(ctx, TSDR_EINVAL, "sync_guard_ppb must be in [0, 1e8]");
    ctx->guard_thr = (float)value * 1e-9f;
    // the adaptive route's history belongs to the old threshold: start over on the fast route
    ctx->guard_exact_now = false;
    ctx->guard_consumed = ctx->guard_seq;   // (the calls in flight were counted against the old threshold: never folded)
    ctx->guard_win_c = ctx->guard_win_f = 0;
  }
  else if (!strcmp(name, "sync_guard_auto")) {
    ctx->opt_guard_auto = value != 0;
    if (!value) ctx->guard_exact_now = false;
  }
  else return tsdr::set_err(ctx, TSDR_EINVAL, "unknown option '%s'", name);
  return TSDR_OK;
}

int tsdr_sync_guard_auto(tsdr_ctx *ctx, int *exact_now, unsigned long long *buffers_exact, unsigned long long *switches) {
  if (!ctx) return TSDR_EINVAL;
  if (exact_now) *exact_now = ctx->guard_exact_now ? 1 : 0;
  if (buffers_exact) *buffers_exact = ctx->guard_auto_buffers;
  if (switches) *switches = ctx->guard_auto_switches;
  return TSDR_OK;
}

int tsdr_wait_stats(tsdr_ctx *ctx, unsigned long long *timeouts, unsigned long long *guard_uncounted) {
  if (!ctx) return TSDR_EINVAL;
  if (timeouts) *timeouts = ctx->wait_timeouts;
  if (guard_uncounted) *guard_uncounted = ctx->guard_uncounted;
  return TSDR_OK;
}

static void hold_cb(void *p) { std::this_thread::sleep_for(std::chrono::milliseconds((long)(intptr_t)p)); }

int tsdr_debug_hold_stream(tsdr_ctx *ctx, int ms) {
  if (!ctx || ms < 0 || ms > 10000) return TSDR_EINVAL;
  TSDR_HIP(ctx, hipLaunchHostFunc(ctx->stream, hold_cb, (void *)(intptr_t)ms));
  return TSDR_OK;
}

int tsdr_sync_guard_stats(tsdr_ctx *ctx, unsigned long long *frames_checked, unsigned long long *frames_reevaluated, int reset) {
  if (!ctx) return TSDR_EINVAL;
  unsigned long long h[2] = {0ull, 0ull};
  int rc = tsdr::pipe_drain(ctx);
  if (rc) return rc;
  if (ctx->guard_stats) {
    TSDR_HIP(ctx, hipMemcpyAsync(h, ctx->guard_stats, 16, hipMemcpyDeviceToHost, ctx->stream));
    if (reset) TSDR_HIP(ctx, hipMemsetAsync(ctx->guard_stats, 0, 16, ctx->stream));
    { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  }
  if (frames_checked) *frames_checked = h[0];
  if (frames_reevaluated) *frames_reevaluated = h[1];
  return TSDR_OK;
}

int tsdr_sync_guard_margins(tsdr_ctx *ctx, int max_frames, float *margins, int *n_frames) {
  if (!ctx || max_frames < 0 || (max_frames && !margins)) return TSDR_EINVAL;
  int rc = tsdr::pipe_drain(ctx);
  if (rc) return rc;
  const int nbx = ctx->guard_last_nbx, nby = ctx->guard_last_nby, nbb = nbx + nby;
  const tsdr_ctx::Buf &gb_ws = ctx->ws[tsdr::WS_GUARD];
  // the records are still there only if the workspace has not been reallocated smaller than where they lay
  const bool have = ctx->guard_last_off != (size_t)-1 && gb_ws.p &&
                    ctx->guard_last_off + (size_t)ctx->guard_last_frames * nbb * sizeof(uint2) <= gb_ws.cap;
  const int F = have ? ctx->guard_last_frames : 0;
  if (n_frames) *n_frames = F;
  const int nf = F < max_frames ? F : max_frames;
  if (nf == 0) return TSDR_OK;
  std::vector<uint2> h((size_t)nf * nbb);
  TSDR_HIP(ctx, hipMemcpyAsync(h.data(), (const char *)gb_ws.p + ctx->guard_last_off, h.size() * sizeof(uint2), hipMemcpyDeviceToHost,
                               ctx->stream));
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  for (int f = 0; f < nf; ++f)
    for (int axis = 0; axis < 2; ++axis) {  // the same top-2 merge as guard_eval (guard.h)
      const uint2 *e = h.data() + (size_t)f * nbb + (axis ? nbx : 0);
      const int nb = axis ? nby : nbx;
      unsigned gb = 0, gs = 0;
      int holders = 0;
      for (int i = 0; i < nb; ++i) gb = e[i].x > gb ? e[i].x : gb;
      for (int i = 0; i < nb; ++i) holders += e[i].x == gb;
      if (holders >= 2) gs = gb;
      else for (int i = 0; i < nb; ++i) { const unsigned c = e[i].x == gb ? e[i].y : e[i].x; gs = c > gs ? c : gs; }
      float b, s2;
      memcpy(&b, &gb, 4); memcpy(&s2, &gs, 4);
      margins[2 * f + axis] = b != 0.f ? (b - s2) / b : 0.f;
    }
  return TSDR_OK;
}

int tsdr_synchronize(tsdr_ctx *ctx) {
  if (!ctx) return TSDR_EINVAL;
  {  // outputs of every submitted buffer are complete after this call (header contract)
    int rc = tsdr::pipe_drain(ctx);
    if (rc) return rc;
  }
  { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  return TSDR_OK;
}

int tsdr_device_info(tsdr_ctx *ctx, char *name, size_t cap, int *cu_count, size_t *hbm_bytes) {
  if (!ctx) return TSDR_EINVAL;
  hipDeviceProp_t prop;
  TSDR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (name && cap) { snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName); }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return TSDR_OK;
}

void *tsdr_dev_alloc(tsdr_ctx *ctx, size_t bytes) {
  if (!ctx) return nullptr;
  void *p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
    tsdr::set_err(ctx, TSDR_ENOMEM, "hipMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
}

int tsdr_dev_free(tsdr_ctx *ctx, void *dev) {
  if (!ctx) return TSDR_EINVAL;
  if (dev) {
    int rc = tsdr::pipe_drain(ctx);  // a submitted buffer may still have to write into this allocation
    if (rc) return rc;
    { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
    TSDR_HIP(ctx, hipFree(dev));
  }
  return TSDR_OK;
}

int tsdr_upload(tsdr_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
  if (!ctx || (bytes && (!dst_dev || !src_host))) return TSDR_EINVAL;
  if (bytes) {
    TSDR_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  }
  return TSDR_OK;
}

int tsdr_download(tsdr_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
  if (!ctx || (bytes && (!dst_host || !src_dev))) return TSDR_EINVAL;
  if (bytes) {
    TSDR_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    { int _w = tsdr::wait_stream(ctx, ctx->stream, __func__); if (_w) return _w; }
  }
  return TSDR_OK;
}

int tsdr_timer_start(tsdr_ctx *ctx) {
  if (!ctx) return TSDR_EINVAL;
  TSDR_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
  return TSDR_OK;
}

int tsdr_timer_stop(tsdr_ctx *ctx, double *ms) {
  if (!ctx || !ms) return TSDR_EINVAL;
  TSDR_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
  { int _w = tsdr::wait_event(ctx, ctx->t1, __func__); if (_w) return _w; }
  float f = 0.f;
  TSDR_HIP(ctx, hipEventElapsedTime(&f, ctx->t0, ctx->t1));
  *ms = f;
  return TSDR_OK;
}

int tsdr_profile_enable(tsdr_ctx *ctx, int on) {
  if (!ctx) return TSDR_EINVAL;
  int rc = tsdr::prof_collect(ctx);
  ctx->prof_on = on != 0;
  return rc;
}

int tsdr_profile_reset(tsdr_ctx *ctx) {
  if (!ctx) return TSDR_EINVAL;
  int rc = tsdr::prof_collect(ctx);
  ctx->prof_agg.clear();
  return rc;
}

int tsdr_profile_count(tsdr_ctx *ctx) {
  if (!ctx) return TSDR_EINVAL;
  int rc = tsdr::prof_collect(ctx);
  if (rc) return rc;
  return (int)ctx->prof_agg.size();
}

int tsdr_profile_get(tsdr_ctx *ctx, int idx, char *name, size_t cap, double *total_ms, long long *launches) {
  if (!ctx || idx < 0 || idx >= (int)ctx->prof_agg.size()) return TSDR_EINVAL;
  const auto &a = ctx->prof_agg[idx];
  if (name && cap) snprintf(name, cap, "%s", a.name.c_str());
  if (total_ms) *total_ms = a.ms;
  if (launches) *launches = a.n;
  return TSDR_OK;
}

}  // extern "C"
