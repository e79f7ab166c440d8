// group.hip -- one process, several MI355X: the multi-GPU split of SURVEY.md 8e behind the C ABI.
//
// The reference runtime is ONE process (GUI.jl:380-382: a producer task, a consumer task, the GUI task).  A tsdr_group is
// what such a process holds to use every GPU of the node: one tsdr_ctx per device, one RCCL communicator per device
// (ncclCommInitAll: single-process ranks over xGMI), and one host thread that drives them -- every per-device stage below
// only enqueues, so the devices run side by side.
//
//   tsdr_group_search   extract_configuration's inner step (GUI.jl:73-81).  The circular autocorrelation
//                       r[k] = sum_m x[m] x[(m+k) mod n] (Autocorrelations.jl:27-29) is a sum over m: device g receives
//                       ITS range of m plus a halo of indexMax samples (H2D of that slice only), computes the partial
//                       sums (tsdr_autocorr_partial_d), ONE ncclAllReduce(sum, f32, indexMax) adds them over xGMI, and only
//                       then the non-linear step -- 10log10(abs2) (:33) and findmax (GUI.jl:79) -- runs, on the root device.
//   tsdr_group_frames   the loop body GUI.jl:163-178 for one received buffer: frames are independent through sig_to_image /
//                       downgradeImage / the vsync statistics, so device g scans its contiguous range of frames
//                       (tsdr_frames_scan_d); the 600x800 images and two argmax keys per frame are gathered to the rendering
//                       device (ncclSend / ncclRecv), which resolves the two sequential couplings -- the lagged s_y and the
//                       IIR recurrence -- over all frames in order (tsdr_frames_combine_d).  Rasters, when wanted, go from
//                       each device straight to the caller's host array.
//   tsdr_group_welch    getWelch (GetSpectrum.jl:36-52): the sum of abs2.(fft(seg)) over segments is a sum -- device g
//                       accumulates its range of segments, ONE all-reduce of sizeFFT f32, 10log10 after it.
//
// Per-frame / per-lag arithmetic is the single-context library's, unchanged: a group of one device returns the
// single-context results bit for bit (tests/test_group_gpu.py); N > 1 differs only by the f32 order of the partial sums.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded at the first group of distinct devices (rccl_api below)

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

struct tsdr_sync;

namespace tsdr {
int autocorr_args(tsdr_ctx *ctx, size_t len, double Fs, double minDelay, double maxDelay, size_t *n, size_t *k0, size_t *cnt);
}

// One host thread per member (round 6).  Every per-device stage "only enqueues" -- but an H2D copy from the caller's PAGEABLE
// array (a Julia Array) is staged through the runtime's bounce buffers by the calling thread and returns when the data has
// left the host array, and a D2H copy into one returns when it has arrived: issued by ONE thread, member i + 1's upload
// started after member i's had finished, and 8 PCIe links carried one link's rate.  Each member's stage now runs on that
// member's own thread (its device current there for good); the caller's thread runs member 0 and waits for the others.
struct GroupMember {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has = false, quit = false;
  int rc = 0;
};

struct tsdr_group {
  int n = 0;
  std::vector<std::unique_ptr<GroupMember>> mem;   // [1 .. n): started at the first call that has work for them
  int opt_threads = 1;             // "member_threads" 1: member threads when the members sit on distinct devices (members sharing a
                                   // device share its link: their staged copies only contend -- sharded search 4.1 against 2.3 ms
                                   // with four members on one device); 2: always (tests); 0: never (rounds 1-5; the A/B)
  // "pin_host" 1: the caller's arrays are page-locked (hipHostRegister, portable) for the duration of the call -- DMA straight
  // from / into the array at the link's rate instead of through bounce buffers.  Off by default: registering and releasing
  // 80 MB costs more than the staged copy it replaces (measured, NOTEBOOK round 6).  (Registrations that OUTLIVE the call were
  // built and withdrawn: a later copy -- the caller's own, or the single-context entry points' -- whose range only partly
  // overlaps a registered one fails with hipErrorInvalidValue.)
  int opt_pin_host = 0;
  struct Pin { const char *p; size_t bytes; };
  std::vector<Pin> pins;
  std::vector<int> dev;
  std::vector<tsdr_ctx *> ctx;
  std::vector<ncclComm_t> comm;
  std::vector<tsdr_sync *> sync;   // one 600x800 SyncXY per member (scan workspaces); the root's carries the lagged s_y
  std::vector<hipEvent_t> ev;      // per member: "this stage is enqueued up to here"
  hipEvent_t t[4] = {};            // root stream: stage boundaries of the last call
  std::string err;
  double stage_ms[3] = {0, 0, 0};
  int last_route = 0;
  bool virt = false;               // a device is listed more than once: no RCCL (see tsdr_group_create)
};

namespace {

using namespace tsdr;

// librccl is loaded on demand (dlopen), not linked: single-GPU users of libtempest_hip.so do not need it installed, and a
// process that already holds an RCCL (PyTorch's) gets that same copy.
struct RcclApi {
  decltype(&::ncclCommInitAll) CommInitAll = nullptr;
  decltype(&::ncclCommDestroy) CommDestroy = nullptr;
  decltype(&::ncclAllReduce) AllReduce = nullptr;
  decltype(&::ncclGroupStart) GroupStart = nullptr;
  decltype(&::ncclGroupEnd) GroupEnd = nullptr;
  decltype(&::ncclSend) Send = nullptr;
  decltype(&::ncclRecv) Recv = nullptr;
  decltype(&::ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  std::string why;
};
const RcclApi &rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) { const char *e = dlerror(); a.why = std::string("librccl not found: ") + (e ? e : "dlopen failed"); return a; }
    bool all = true;
    auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) { all = false; a.why = std::string("librccl lacks ") + n; } return p; };
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.ok = all;
    return a;
  }();
  return api;
}
#define ncclCommInitAll rccl_api().CommInitAll
#define ncclCommDestroy rccl_api().CommDestroy
#define ncclAllReduce rccl_api().AllReduce
#define ncclGroupStart rccl_api().GroupStart
#define ncclGroupEnd rccl_api().GroupEnd
#define ncclSend rccl_api().Send
#define ncclRecv rccl_api().Recv
#define ncclGetErrorString rccl_api().GetErrorString

int gerr(tsdr_group *g, int status, const std::string &what) {
  if (g) g->err = what;
  (void)hipGetLastError();
  return status;
}

int member_err(tsdr_group *g, int i, int rc, const char *what) {
  return gerr(g, rc, std::string(what) + " on device " + std::to_string(g->dev[i]) + ": " + tsdr_last_error(g->ctx[i]));
}

#define G_HIP(g, call)                                                                                          \
  do {                                                                                                          \
    hipError_t _e = (call);                                                                                     \
    if (_e != hipSuccess) return gerr((g), TSDR_EHIP, std::string(#call) + ": " + hipGetErrorString(_e));       \
  } while (0)
#define G_NCCL(g, call)                                                                                         \
  do {                                                                                                          \
    ncclResult_t _r = (call);                                                                                   \
    if (_r != ncclSuccess) return gerr((g), TSDR_EHIP, std::string(#call) + ": " + ncclGetErrorString(_r));     \
  } while (0)

// contiguous, near-equal split of range(n) over `world` members (parallel.py:shard_range)
void shard_range(size_t n, int world, int rank, size_t *start, size_t *count) {
  const size_t base = n / (size_t)world, rem = n % (size_t)world;
  *start = (size_t)rank * base + std::min<size_t>((size_t)rank, rem);
  *count = base + ((size_t)rank < rem ? 1 : 0);
}

size_t pow2_at_least(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }
bool smooth235(size_t v) {
  for (size_t f : {2, 3, 5}) while (v > 1 && v % f == 0) v /= f;
  return v == 1;
}
// complex points per transform: the single-context route for n samples / the segment + halo cross-correlation of one member
size_t single_points(size_t n) { return (n % 2 == 0 && n > 1024 && smooth235(n / 2)) ? n / 2 : pow2_at_least(2 * n) / 2; }
size_t sharded_points(size_t n, size_t n_lags, int world) { return pow2_at_least((n + world - 1) / world + n_lags - 1); }

__global__ __launch_bounds__(256) void k_acc(float *__restrict__ acc, const float *__restrict__ x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[i] = acc[i] + x[i];
}

__global__ __launch_bounds__(256) void k_db(float *__restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = 10.0f * log10f(y[i]);
}

void member_loop(tsdr_group *g, int i) {
  GroupMember &w = *g->mem[i];
  (void)hipSetDevice(g->dev[i]);
  std::unique_lock<std::mutex> lk(w.m);
  for (;;) {
    w.cv.wait(lk, [&] { return w.has || w.quit; });
    if (w.quit) return;
    std::function<int()> job = std::move(w.job);
    lk.unlock();
    const int rc = job();
    lk.lock();
    w.rc = rc;
    w.has = false;
    w.cv.notify_all();
  }
}

// fn(i) for every member i < world: member 0 on the calling thread, the others on their own threads, side by side; returns
// when all have returned.  fn touches only member i's context (and reads the call's arguments); errors are reported from
// the calling thread afterwards.
template <class F>
int run_members(tsdr_group *g, int world, const char *what, F fn) {
  std::vector<int> rcs((size_t)world, TSDR_OK);
  const bool threads = world > 1 && (g->opt_threads == 2 || (g->opt_threads == 1 && !g->virt));
  if (threads) {
    if (g->mem.empty()) g->mem.resize((size_t)g->n);
    for (int i = 1; i < world; ++i) {
      if (!g->mem[i]) { g->mem[i] = std::make_unique<GroupMember>(); g->mem[i]->th = std::thread(member_loop, g, i); }
      GroupMember &w = *g->mem[i];
      { std::lock_guard<std::mutex> lk(w.m); w.job = [&fn, i] { return fn(i); }; w.has = true; }
      w.cv.notify_all();
    }
  }
  for (int i = 0; i < (threads ? 1 : world); ++i) {
    if (hipSetDevice(g->dev[i]) != hipSuccess) { rcs[i] = hip_fail(g->ctx[i], hipGetLastError(), "hipSetDevice"); continue; }
    rcs[i] = fn(i);
    if (rcs[i] && !threads) break;
  }
  if (threads)
    for (int i = 1; i < world; ++i) {
      GroupMember &w = *g->mem[i];
      std::unique_lock<std::mutex> lk(w.m);
      w.cv.wait(lk, [&] { return !w.has; });
      rcs[i] = w.rc;
    }
  for (int i = 0; i < world; ++i)
    if (rcs[i]) return member_err(g, i, rcs[i], what);
  return TSDR_OK;
}

void stop_members(tsdr_group *g) {
  for (auto &w : g->mem) {
    if (!w) continue;
    { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
  }
  g->mem.clear();
}

// "pin_host": [p, p + bytes) page-locked for every device of the node until the call returns (CallScope)
void pin_host(tsdr_group *g, const void *ptr, size_t bytes) {
  if (!g->opt_pin_host || !ptr || bytes < (1u << 20)) return;
  if (hipHostRegister((void *)ptr, bytes, hipHostRegisterPortable) == hipSuccess) g->pins.push_back({(const char *)ptr, bytes});
  (void)hipGetLastError();   // (an array that cannot be registered is copied through the bounce buffers as before)
}
void unpin_all(tsdr_group *g) {
  for (auto &e : g->pins) (void)hipHostUnregister((void *)e.p);
  g->pins.clear();
  (void)hipGetLastError();
}

#define M_HIP(c, call)                                               \
  do {                                                               \
    hipError_t _e = (call);                                          \
    if (_e != hipSuccess) return hip_fail((c), _e, #call);           \
  } while (0)

int sync_all(tsdr_group *g) {
  for (int i = 0; i < g->n; ++i) {
    G_HIP(g, hipSetDevice(g->dev[i]));
    if (int w = tsdr::wait_stream(g->ctx[i], g->ctx[i]->stream, "group: member stream")) return member_err(g, i, w, "sync_all");
  }
  return TSDR_OK;
}

int ensure_syncs(tsdr_group *g) {
  for (int i = 0; i < g->n; ++i) {
    if (g->sync[i]) continue;
    G_HIP(g, hipSetDevice(g->dev[i]));
    int rc = tsdr_sync_create(g->ctx[i], TSDR_RENDER_H, TSDR_RENDER_W, &g->sync[i]);
    if (rc) return member_err(g, i, rc, "SyncXY");
  }
  return TSDR_OK;
}

// the root's stream is ordered behind what every member has enqueued so far (members sharing one device: HIP events)
int root_after_members(tsdr_group *g) {
  for (int i = 1; i < g->n; ++i) {
    G_HIP(g, hipEventRecord(g->ev[i], g->ctx[i]->stream));
    G_HIP(g, hipStreamWaitEvent(g->ctx[0]->stream, g->ev[i], 0));
  }
  return TSDR_OK;
}

// sum of the members' vectors: ONE ncclAllReduce over xGMI -- or, members sharing a device, adds on the root in member order
int all_reduce_sum(tsdr_group *g, const std::vector<float *> &buf, size_t count) {
  if (!g->virt) {
    G_NCCL(g, ncclGroupStart());
    for (int i = 0; i < g->n; ++i) {
      ncclResult_t r = ncclAllReduce(buf[i], buf[i], count, ncclFloat, ncclSum, g->comm[i], g->ctx[i]->stream);
      if (r != ncclSuccess) { (void)ncclGroupEnd(); return gerr(g, TSDR_EHIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r)); }
    }
    G_NCCL(g, ncclGroupEnd());
    return TSDR_OK;
  }
  int rc = root_after_members(g);
  if (rc) return rc;
  for (int i = 1; i < g->n; ++i) {
    hipLaunchKernelGGL(k_acc, dim3((unsigned)std::min<size_t>(ceil_div(count, 256), 2048)), dim3(256), 0, g->ctx[0]->stream, buf[0], (const float *)buf[i], count);
    G_HIP(g, hipGetLastError());
  }
  return TSDR_OK;
}

// Every entry point: the calling thread's current device is put back on return (the stages below select each member's device
// in turn), and a call that fails half-way waits for whatever it has enqueued -- copies that read the caller's arrays among it --
// before it hands control back.
struct CallScope {
  tsdr_group *g;
  int prev = -1;
  bool ok = false;
  explicit CallScope(tsdr_group *g_) : g(g_) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~CallScope() {
    if (!ok && g)
      for (int i = 0; i < g->n; ++i)
        if (g->ctx[i] && hipSetDevice(g->dev[i]) == hipSuccess) (void)tsdr::wait_stream(g->ctx[i], g->ctx[i]->stream, "group: failed call");
    if (g) unpin_all(g);   // (every copy of the call is complete here: sync_all on success, the waits above otherwise)
    if (prev >= 0) (void)hipSetDevice(prev);
    (void)hipGetLastError();
  }
  int done() { ok = true; return TSDR_OK; }
};

void stage_times(tsdr_group *g) {
  (void)hipSetDevice(g->dev[0]);   // (the events are the root's)
  for (int k = 0; k < 3; ++k) {
    float ms = 0.f;
    g->stage_ms[k] = hipEventElapsedTime(&ms, g->t[k], g->t[k + 1]) == hipSuccess ? (double)ms : 0.0;
  }
}

}  // namespace

extern "C" {

int tsdr_group_create(const int *devices, int n, tsdr_group **out) {
  if (!out) return TSDR_EINVAL;
  *out = nullptr;
  if (n <= 0 || n > 64) return TSDR_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TSDR_ENODEV;
  CallScope scope(nullptr);   // (the caller's current device comes back)
  tsdr_group *g = new tsdr_group();
  g->n = n;
  for (int i = 0; i < n; ++i) {
    const int d = devices ? devices[i] : i;
    if (d < 0 || d >= ndev) { delete g; return TSDR_EINVAL; }
    if (std::find(g->dev.begin(), g->dev.end(), d) != g->dev.end()) g->virt = true;
    g->dev.push_back(d);
  }
  g->ctx.assign(n, nullptr); g->sync.assign(n, nullptr); g->ev.assign(n, nullptr);
  g->comm.assign(n, nullptr);
  bool ok = true;
  for (int i = 0; i < n && ok; ++i) {
    g->ctx[i] = tsdr_create(g->dev[i]);   // (makes its device current)
    ok = g->ctx[i] != nullptr && hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming) == hipSuccess;
  }
  // single-process ranks: rank i = member i on devices[i]; the collectives below run inside ncclGroupStart / End
  // (members sharing a device: RCCL takes one rank per device, so such a group exchanges by copies and adds on the device)
  bool comm_failed = false;
  if (ok && !g->virt) {
    if (!rccl_api().ok) {
      fprintf(stderr, "tempest_hip: tsdr_group_create: %s (a group of distinct devices needs RCCL)\n", rccl_api().why.c_str());
      ok = false;
    } else {
      ok = ncclCommInitAll(g->comm.data(), n, g->dev.data()) == ncclSuccess;
      comm_failed = !ok;   // (what a failed ncclCommInitAll left in comm[] is not ours to destroy)
    }
  }
  if (ok) ok = hipSetDevice(g->dev[0]) == hipSuccess;
  for (int k = 0; k < 4 && ok; ++k) ok = hipEventCreate(&g->t[k]) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    if (!comm_failed) for (auto &c : g->comm) c = c ? (ncclCommDestroy(c), nullptr) : nullptr;
    for (int i = 0; i < n; ++i) {
      if (g->ev[i]) { (void)hipSetDevice(g->dev[i]); (void)hipEventDestroy(g->ev[i]); }
      if (g->ctx[i]) tsdr_destroy(g->ctx[i]);
    }
    for (auto e : g->t) if (e) (void)hipEventDestroy(e);
    delete g;
    return TSDR_EHIP;
  }
  *out = g;
  return TSDR_OK;
}

void tsdr_group_destroy(tsdr_group *g) {
  if (!g) return;
  CallScope scope(nullptr);
  for (int i = 0; i < g->n; ++i) {
    (void)hipSetDevice(g->dev[i]);
    if (tsdr_synchronize(g->ctx[i])) {
      // a member's stream never completed (bounded wait): destroying communicators and contexts would wait with it
      fprintf(stderr, "tempest_hip: tsdr_group_destroy: %s -- group abandoned, its device memory is not released\n", tsdr_last_error(g->ctx[i]));
      return;
    }
  }
  stop_members(g);
  for (auto c : g->comm) if (c) (void)ncclCommDestroy(c);
  (void)hipSetDevice(g->dev[0]);
  for (auto e : g->t) if (e) (void)hipEventDestroy(e);
  for (int i = 0; i < g->n; ++i) {
    (void)hipSetDevice(g->dev[i]);
    if (g->sync[i]) tsdr_sync_free(g->sync[i]);
    if (g->ev[i]) (void)hipEventDestroy(g->ev[i]);
    tsdr_destroy(g->ctx[i]);
  }
  delete g;
}

int tsdr_group_size(const tsdr_group *g) { return g ? g->n : TSDR_EINVAL; }
tsdr_ctx *tsdr_group_ctx(tsdr_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->ctx[i] : nullptr; }
const char *tsdr_group_last_error(tsdr_group *g) { return g ? g->err.c_str() : "null group"; }

int tsdr_group_set_precision(tsdr_group *g, int mode) {
  if (!g) return TSDR_EINVAL;
  CallScope scope(nullptr);
  for (int i = 0; i < g->n; ++i) {
    G_HIP(g, hipSetDevice(g->dev[i]));
    int rc = tsdr_set_precision(g->ctx[i], mode);
    if (rc) return member_err(g, i, rc, "set_precision");
  }
  return TSDR_OK;
}

int tsdr_group_set_option(tsdr_group *g, const char *name, int value) {
  if (!g || !name) return TSDR_EINVAL;
  // the group's own switches: "member_threads" (1: one host thread per member drives its stage, default; 0: the caller's
  // thread drives them all in turn) and "pin_host" (1: the caller's arrays are page-locked at first use and stay so until the
  // group is destroyed; the caller keeps them alive that long)
  if (!strcmp(name, "member_threads")) { g->opt_threads = value < 0 ? 0 : value > 2 ? 2 : value; return TSDR_OK; }
  if (!strcmp(name, "pin_host")) {
    g->opt_pin_host = value != 0;
    return TSDR_OK;
  }
  CallScope scope(nullptr);
  for (int i = 0; i < g->n; ++i) {
    G_HIP(g, hipSetDevice(g->dev[i]));
    int rc = tsdr_set_option(g->ctx[i], name, value);
    if (rc) return member_err(g, i, rc, "set_option");
  }
  return TSDR_OK;
}

int tsdr_group_sync_reset(tsdr_group *g) {
  if (!g) return TSDR_EINVAL;
  CallScope scope(nullptr);
  for (int i = 0; i < g->n; ++i) {
    if (!g->sync[i]) continue;
    G_HIP(g, hipSetDevice(g->dev[i]));
    int rc = tsdr_sync_reset(g->sync[i]);
    if (rc) return member_err(g, i, rc, "sync_reset");
  }
  return TSDR_OK;
}

int tsdr_group_timing(tsdr_group *g, int *route, double *ms) {
  if (!g) return TSDR_EINVAL;
  if (route) *route = g->last_route;
  if (ms) for (int k = 0; k < 3; ++k) ms[k] = g->stage_ms[k];
  return TSDR_OK;
}

int tsdr_group_search(tsdr_group *g, const float *x, int is_iq, size_t len, double Fs, double minDelay, double maxDelay,
                      int log_scale, float *out, size_t *n_out, size_t win_lo, size_t win_cnt, size_t *idx, float *val,
                      int route) {
  if (!g || !x || !idx || route < 0 || route > 2) return TSDR_EINVAL;
  CallScope scope(g);
  const int N = g->n;
  tsdr_ctx *c0 = g->ctx[0];
  size_t n, k0, cnt;
  G_HIP(g, hipSetDevice(g->dev[0]));
  int rc = autocorr_args(c0, len, Fs, minDelay, maxDelay, &n, &k0, &cnt);
  if (rc) return member_err(g, 0, rc, "group_search");
  if (n_out) *n_out = cnt;
  if (win_cnt == 0 || win_lo >= cnt || win_cnt > cnt - win_lo) return gerr(g, TSDR_EBOUNDS, "group_search: window outside the lag vector");
  const size_t n_lags = k0 + cnt;   // = indexMax
  const size_t esz = is_iq ? 8 : 4;
  // "sharded" when asked for, or (auto) when a member's segment + halo transform is smaller than the one a single device
  // runs: the halo puts a floor of indexMax points under every member's transform, so with the reference's own window
  // (n = 2 indexMax, Autocorrelations.jl:27) one device is the faster route and the group says so (tsdr_group_timing)
  bool sharded = route == 1;
  if (route == 0 && N > 1) {
    const size_t one = n <= 2 * n_lags ? single_points(n) : pow2_at_least(n + n_lags - 1);
    sharded = sharded_points(n, n_lags, N) < one;
  }
  g->last_route = sharded ? 1 : 2;
  G_HIP(g, hipEventRecord(g->t[0], c0->stream));
  if (!sharded && n <= 2 * n_lags) {  // the single-context search on the root device
    float *dx = (float *)c0->scratch(WS_IN, n * esz);
    float *dout = (float *)c0->scratch(WS_OUT, cnt * 4);
    if (!dx || !dout) return member_err(g, 0, TSDR_ENOMEM, "group_search");
    pin_host(g, x, n * esz);
    G_HIP(g, hipMemcpyAsync(dx, x, n * esz, hipMemcpyHostToDevice, c0->stream));
    G_HIP(g, hipEventRecord(g->t[1], c0->stream));
    G_HIP(g, hipEventRecord(g->t[2], c0->stream));
    rc = tsdr_autocorr_search_d(c0, dx, is_iq, n, Fs, minDelay, maxDelay, log_scale, dout, nullptr, win_lo, win_cnt, idx, val);
    if (rc) return member_err(g, 0, rc, "group_search");
    G_HIP(g, hipEventRecord(g->t[3], c0->stream));
    if (out) G_HIP(g, hipMemcpyAsync(out, dout, cnt * 4, hipMemcpyDeviceToHost, c0->stream));
    if (int w = tsdr::wait_stream(c0, c0->stream, "group_search")) return member_err(g, 0, w, "group_search");
    stage_times(g);
    return scope.done();
  }
  // stage 1, per member (each on its own host thread): H2D of its range of m plus the halo (wrapping at n), partial sums
  // over that range
  const int world = sharded ? N : 1;
  std::vector<float *> part(world, nullptr);
  pin_host(g, x, n * esz);
  rc = run_members(g, world, "group_search", [&](int i) -> int {
    tsdr_ctx *c = g->ctx[i];
    size_t m0, mc;
    shard_range(n, world, i, &m0, &mc);
    part[i] = (float *)c->scratch(WS_OUT, n_lags * 4);
    if (!part[i]) return TSDR_ENOMEM;
    if (mc == 0) { M_HIP(c, hipMemsetAsync(part[i], 0, n_lags * 4, c->stream)); return TSDR_OK; }
    const size_t vlen = mc + n_lags - 1;
    char *dx = (char *)c->scratch(WS_IN, vlen * esz);
    if (!dx) return TSDR_ENOMEM;
    for (size_t done = 0; done < vlen;) {   // x[(m0 + j) mod n], j < vlen: at most a few contiguous pieces
      const size_t src = (m0 + done) % n, run = std::min(vlen - done, n - src);
      M_HIP(c, hipMemcpyAsync(dx + done * esz, (const char *)x + src * esz, run * esz, hipMemcpyHostToDevice, c->stream));
      done += run;
    }
    // the slice as a sequence of its own: sum_{m < mc} xs[m] xs[m + k], no wrap inside (m + k <= vlen - 1)
    return tsdr_autocorr_partial_d(c, (const float *)dx, is_iq, vlen, 0, mc, n_lags, part[i]);
  });
  if (rc) return rc;
  G_HIP(g, hipSetDevice(g->dev[0]));
  G_HIP(g, hipEventRecord(g->t[1], c0->stream));
  // stage 2: ONE all-reduce of the accumulators (linear domain) over xGMI
  if (sharded) {
    rc = all_reduce_sum(g, part, n_lags);
    if (rc) return rc;
  }
  G_HIP(g, hipSetDevice(g->dev[0]));
  G_HIP(g, hipEventRecord(g->t[2], c0->stream));
  // stage 3, root: the non-linear step comes after the reduce -- 10log10(abs2) (:33) and findmax over the zoom window
  float *dout = (float *)c0->scratch(WS_AUX, cnt * 4);
  if (!dout) return member_err(g, 0, TSDR_ENOMEM, "group_search");
  rc = tsdr_autocorr_finish_d(c0, part[0], k0, cnt, log_scale, dout);
  if (!rc) rc = tsdr_argmax_d(c0, dout + win_lo, win_cnt, idx, val);
  if (rc) return member_err(g, 0, rc, "group_search");
  G_HIP(g, hipEventRecord(g->t[3], c0->stream));
  if (out) G_HIP(g, hipMemcpyAsync(out, dout, cnt * 4, hipMemcpyDeviceToHost, c0->stream));
  rc = sync_all(g);
  if (rc) return rc;
  stage_times(g);
  return scope.done();
}

int tsdr_group_frames(tsdr_group *g, const float *iq, size_t nEch, size_t S, int y_t, int x_t, float alpha, int do_align,
                      float *imageOut_state, float *frames_out, float *raster_out, int *sync_idx, int *n_frames) {
  if (!g || !imageOut_state || S == 0 || y_t <= 0 || x_t <= 0 || (nEch && !iq)) return TSDR_EINVAL;
  const int N = g->n;
  const size_t nb = nEch / S, npx = (size_t)TSDR_RENDER_H * TSDR_RENDER_W, P = (size_t)y_t * x_t;
  if (nb > (size_t)1 << 20) return gerr(g, TSDR_EINVAL, "too many frames in one buffer");
  if (n_frames) *n_frames = (int)nb;
  if (nb == 0) return TSDR_OK;
  CallScope scope(g);
  int rc = ensure_syncs(g);
  if (rc) return rc;
  tsdr_ctx *c0 = g->ctx[0];
  std::vector<float *> img(N, nullptr);
  std::vector<unsigned long long *> keys(N, nullptr);
  std::vector<size_t> f0(N), fc(N);
  G_HIP(g, hipSetDevice(g->dev[0]));
  G_HIP(g, hipEventRecord(g->t[0], c0->stream));
  // stage 1, per member (each on its own host thread): H2D of its frames' samples, IQ -> 600x800 image + two argmax keys per
  // frame (+ raster -> the caller's host array, from that member over its own link)
  pin_host(g, iq, nb * S * 8);
  if (raster_out) pin_host(g, raster_out, nb * P * 4);
  rc = run_members(g, N, "group_frames", [&](int i) -> int {
    tsdr_ctx *c = g->ctx[i];
    shard_range(nb, N, i, &f0[i], &fc[i]);
    // (the root's buffers hold every frame of the buffer: its own range lies where the gather puts the others')
    const size_t hold = i == 0 ? nb : fc[i];
    img[i] = (float *)c->scratch(WS_IMG, (hold ? hold : 1) * npx * 4);
    keys[i] = (unsigned long long *)c->scratch(WS_KEYS, (hold ? hold : 1) * 2 * 8);
    if (!img[i] || !keys[i]) return TSDR_ENOMEM;
    if (fc[i] == 0) return TSDR_OK;
    float *d_iq = (float *)c->scratch(WS_IN, fc[i] * S * 8);
    float *d_ra = raster_out ? (float *)c->scratch(WS_FFT_A, fc[i] * P * 4) : nullptr;
    if (!d_iq || (raster_out && !d_ra)) return TSDR_ENOMEM;
    M_HIP(c, hipMemcpyAsync(d_iq, iq + 2 * f0[i] * S, fc[i] * S * 8, hipMemcpyHostToDevice, c->stream));
    int nf = 0;
    float *my_img = img[i] + (i == 0 ? f0[0] * npx : 0);
    unsigned long long *my_keys = keys[i] + (i == 0 ? f0[0] * 2 : 0);
    int rcm = tsdr_frames_scan_d(c, g->sync[i], d_iq, fc[i] * S, S, y_t, x_t, do_align, my_img, d_ra, do_align ? my_keys : nullptr, &nf);
    if (rcm) return rcm;
    if (d_ra) M_HIP(c, hipMemcpyAsync(raster_out + f0[i] * P, d_ra, fc[i] * P * 4, hipMemcpyDeviceToHost, c->stream));
    return TSDR_OK;
  });
  if (rc) return rc;
  G_HIP(g, hipSetDevice(g->dev[0]));
  G_HIP(g, hipEventRecord(g->t[1], c0->stream));
  // stage 2: gather to the rendering device (GUI.jl:177 hands the frames to ONE renderer): every other member sends its
  // share once -- 1.92 MB + 16 B per frame
  if (N > 1 && !g->virt) {
    G_NCCL(g, ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    for (int i = 1; i < N && r == ncclSuccess; ++i) {
      if (fc[i] == 0) continue;
      r = ncclSend(img[i], fc[i] * npx, ncclFloat, 0, g->comm[i], g->ctx[i]->stream);
      if (r == ncclSuccess) r = ncclRecv(img[0] + f0[i] * npx, fc[i] * npx, ncclFloat, i, g->comm[0], c0->stream);
      if (do_align && r == ncclSuccess) r = ncclSend(keys[i], fc[i] * 2, ncclUint64, 0, g->comm[i], g->ctx[i]->stream);
      if (do_align && r == ncclSuccess) r = ncclRecv(keys[0] + f0[i] * 2, fc[i] * 2, ncclUint64, i, g->comm[0], c0->stream);
    }
    if (r != ncclSuccess) { (void)ncclGroupEnd(); return gerr(g, TSDR_EHIP, std::string("gather: ") + ncclGetErrorString(r)); }
    G_NCCL(g, ncclGroupEnd());
  } else if (N > 1) {   // members sharing a device: device-to-device copies behind the members' scans
    rc = root_after_members(g);
    if (rc) return rc;
    for (int i = 1; i < N; ++i) {
      if (fc[i] == 0) continue;
      G_HIP(g, hipMemcpyAsync(img[0] + f0[i] * npx, img[i], fc[i] * npx * 4, hipMemcpyDeviceToDevice, c0->stream));
      if (do_align) G_HIP(g, hipMemcpyAsync(keys[0] + f0[i] * 2, keys[i], fc[i] * 2 * 8, hipMemcpyDeviceToDevice, c0->stream));
    }
  }
  G_HIP(g, hipSetDevice(g->dev[0]));
  G_HIP(g, hipEventRecord(g->t[2], c0->stream));
  // stage 3, root: lagged s_y (FrameSynchronisation.jl:66) + circshift + IIR (GUI.jl:172,175) over all frames in order.
  // (frames_out leaves from the root: frame k of the output is alpha * frame k - 1 + (1 - alpha) * shifted image k, an f32
  // recurrence whose rounding depends on the order -- a member could form its range's outputs only by a re-associated scan,
  // and the group's results are the single-context ones bit for bit.  Rasters, which have no such coupling, do leave from
  // each member over its own link, above.)
  if (frames_out) pin_host(g, frames_out, nb * npx * 4);
  float *d_state = (float *)c0->scratch(WS_AUX, npx * 4);
  float *d_frames = frames_out ? (float *)c0->scratch(WS_OUT, nb * npx * 4) : nullptr;
  int *d_idx = (sync_idx && do_align) ? (int *)c0->scratch(WS_MISC, nb * 8 + 16) : nullptr;
  if (!d_state || (frames_out && !d_frames) || (sync_idx && do_align && !d_idx)) return member_err(g, 0, TSDR_ENOMEM, "group_frames");
  G_HIP(g, hipMemcpyAsync(d_state, imageOut_state, npx * 4, hipMemcpyHostToDevice, c0->stream));
  rc = tsdr_frames_combine_d(c0, g->sync[0], img[0], keys[0], (int)nb, alpha, do_align, d_state, d_frames, d_idx);
  if (rc) return member_err(g, 0, rc, "frames_combine");
  G_HIP(g, hipEventRecord(g->t[3], c0->stream));
  G_HIP(g, hipMemcpyAsync(imageOut_state, d_state, npx * 4, hipMemcpyDeviceToHost, c0->stream));
  if (d_frames) G_HIP(g, hipMemcpyAsync(frames_out, d_frames, nb * npx * 4, hipMemcpyDeviceToHost, c0->stream));
  if (d_idx) G_HIP(g, hipMemcpyAsync(sync_idx, d_idx, nb * 8, hipMemcpyDeviceToHost, c0->stream));
  rc = sync_all(g);
  if (rc) return rc;
  g->last_route = 1;
  stage_times(g);
  return scope.done();
}

int tsdr_group_welch(tsdr_group *g, const float *sig, int is_complex, size_t len, size_t sizeFFT, int lin, float *y) {
  if (!g || !y || sizeFFT < 2 || (len && !sig)) return TSDR_EINVAL;
  const int N = g->n;
  const size_t nbSeg = len / sizeFFT, esz = is_complex ? 8 : 4;
  tsdr_ctx *c0 = g->ctx[0];
  CallScope scope(g);
  if (nbSeg == 0) {  // (the single-context call defines what an empty sum returns)
    G_HIP(g, hipSetDevice(g->dev[0]));
    int rc = tsdr_welch(c0, sig, is_complex, len, sizeFFT, lin, y);
    return rc ? member_err(g, 0, rc, "group_welch") : scope.done();
  }
  std::vector<float *> part(N, nullptr);
  G_HIP(g, hipSetDevice(g->dev[0]));
  G_HIP(g, hipEventRecord(g->t[0], c0->stream));
  pin_host(g, sig, nbSeg * sizeFFT * esz);
  {
    int rcs = run_members(g, N, "group_welch", [&](int i) -> int {
      tsdr_ctx *c = g->ctx[i];
      size_t s0, sc;
      shard_range(nbSeg, N, i, &s0, &sc);
      part[i] = (float *)c->scratch(WS_OUT, sizeFFT * 4);
      if (!part[i]) return TSDR_ENOMEM;
      if (sc == 0) { M_HIP(c, hipMemsetAsync(part[i], 0, sizeFFT * 4, c->stream)); return TSDR_OK; }
      char *dx = (char *)c->scratch(WS_IN, sc * sizeFFT * esz);
      if (!dx) return TSDR_ENOMEM;
      M_HIP(c, hipMemcpyAsync(dx, (const char *)sig + s0 * sizeFFT * esz, sc * sizeFFT * esz, hipMemcpyHostToDevice, c->stream));
      return tsdr_welch_d(c, (const float *)dx, is_complex, sc * sizeFFT, sizeFFT, /*lin=*/1, part[i]);   // fftshifted linear sums
    });
    if (rcs) return rcs;
  }
  G_HIP(g, hipSetDevice(g->dev[0]));
  G_HIP(g, hipEventRecord(g->t[1], c0->stream));
  if (N > 1) {
    int rca = all_reduce_sum(g, part, sizeFFT);
    if (rca) return rca;
  }
  G_HIP(g, hipSetDevice(g->dev[0]));
  G_HIP(g, hipEventRecord(g->t[2], c0->stream));
  if (!lin) hipLaunchKernelGGL(k_db, dim3((unsigned)ceil_div(sizeFFT, 256)), dim3(256), 0, c0->stream, part[0], sizeFFT);
  G_HIP(g, hipGetLastError());
  G_HIP(g, hipEventRecord(g->t[3], c0->stream));
  G_HIP(g, hipMemcpyAsync(y, part[0], sizeFFT * 4, hipMemcpyDeviceToHost, c0->stream));
  int rc = sync_all(g);
  if (rc) return rc;
  g->last_route = N > 1 ? 1 : 2;
  stage_times(g);
  return scope.done();
}

}  // extern "C"
