// ring.hip -- the consumer side of AtomicCircularBuffer (AtomicAbstractSDRs.jl:64-190) as a pinned-host staging
// ring feeding the GPU.
//
// Reference semantics kept: `depth` slots of nEch samples; the producer (the SDR thread, `circ_put!` :161-173)
// never blocks -- it writes slot ptr_write, advances it modulo depth and raises the "new data" count, saturating at
// depth (:125-129), so a slow consumer loses the oldest buffers; the consumer (`circ_take!` :177-190 = recv!) blocks
// until the count is positive, reads slot ptr_read, advances it, lowers the count.
//
// What is different, and why: the slots live in pinned host memory, so the copy to the GPU is an asynchronous DMA
// on its own stream; the ring owns two device buffers and, when a buffer is handed out, already starts the DMA of
// the next committed slot into the other one -- the H2D transfer of buffer k+1 runs under the kernels of buffer k.
// Slots may hold ComplexF32 (what recv! returns in the reference) or interleaved int16 I/Q as SDR hardware
// delivers it (half the PCIe bytes; expanded to ComplexF32 on the device with a caller-given scale).
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

struct tsdr_ring {
  tsdr_ctx *ctx = nullptr;
  size_t nEch = 0;
  int depth = 0, fmt = 0;          // fmt: 0 = ComplexF32, 1 = int16 I/Q
  float scale = 1.0f;              // int16 -> float factor
  size_t slot_bytes = 0;
  char *host = nullptr;            // depth pinned slots
  float *dev[2] = {nullptr, nullptr};
  void *raw[2] = {nullptr, nullptr};  // int16 landing buffers (fmt 1)
  hipStream_t copy = nullptr;
  hipEvent_t ready[2] = {nullptr, nullptr};  // DMA (+ conversion) of dev[i] complete
  hipEvent_t freed = nullptr;                 // consumer's stream has passed the previous hand-out
  std::mutex m;
  std::condition_variable cv;
  int ptr_write = 0, ptr_read = 0, t_new = 0;  // AtomicCircularBuffer state
  bool stop = false;
  // which slot sequence number sits (or is arriving) in dev[i]; -1 = none
  long long staged_seq[2] = {-1, -1};
  // generation of every host slot (bumped each time the producer publishes it) and the generation the slot had
  // when the DMA into dev[i] was issued: a prefetch is good only while the producer has not rewritten that slot
  std::vector<unsigned long long> slot_gen;
  unsigned long long staged_gen[2] = {0, 0};
  int next_dev = 0;
  int dma_slot[2] = {-1, -1};   // host slot the last DMA into dev[i] reads (guards against a lapping producer)
  int writing_slot = -1;         // slot the producer is filling right now
  bool staging[2] = {false, false};  // a DMA into dev[i] is being enqueued (its `ready` event is not recorded yet)
  long long seq_written = 0, seq_read = 0;  // counts of put / take
  unsigned long long produced = 0, consumed = 0, overflow = 0;
  unsigned long long prefetch_hits = 0, prefetch_misses = 0;  // takes served by the DMA sent ahead / staged at take time
  std::chrono::steady_clock::time_point t0;
};

namespace tsdr {

__global__ __launch_bounds__(256) void k_sc16_to_cf32(const short2 *__restrict__ in, size_t n, float scale,
                                                      float2 *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const short2 v = in[i];
    out[i] = make_float2((float)v.x * scale, (float)v.y * scale);
  }
}

// start the DMA of host slot `slot` into device buffer d (caller holds no lock; slot content is stable because the
// producer only overwrites slots the consumer has not reserved -- see ring_take)
static int ring_stage(tsdr_ring *r, int slot, int d) {
  tsdr_ctx *ctx = r->ctx;
  const char *src = r->host + (size_t)slot * r->slot_bytes;
  if (r->fmt == 0 || r->fmt == 2) {   // (2: the int16 pairs stay int16 on the device -- tsdr_frames_sc16_d reads them as they are)
    TSDR_HIP(ctx, hipMemcpyAsync(r->dev[d], src, r->slot_bytes, hipMemcpyHostToDevice, r->copy));
  } else {
    TSDR_HIP(ctx, hipMemcpyAsync(r->raw[d], src, r->slot_bytes, hipMemcpyHostToDevice, r->copy));
    hipLaunchKernelGGL(k_sc16_to_cf32, dim3((unsigned)stream_grid(ctx, r->nEch)), dim3(256), 0, r->copy,
                       (const short2 *)r->raw[d], r->nEch, r->scale, (float2 *)r->dev[d]);
    TSDR_HIP(ctx, hipGetLastError());
  }
  TSDR_HIP(ctx, hipEventRecord(r->ready[d], r->copy));
  return TSDR_OK;
}

}  // namespace tsdr

using namespace tsdr;

extern "C" {

int tsdr_ring_create(tsdr_ctx *ctx, size_t nEch, int depth, int fmt, float scale, tsdr_ring **out) {
  if (!ctx || !out || nEch == 0 || depth < 2 || fmt < 0 || fmt > 2) return TSDR_EINVAL;
  *out = nullptr;
  tsdr_ring *r = new tsdr_ring();
  r->ctx = ctx; r->nEch = nEch; r->depth = depth; r->fmt = fmt; r->scale = scale;
  r->slot_bytes = nEch * (fmt == 0 ? 8 : 4);
  r->slot_gen.assign((size_t)depth, 0ull);
  bool ok = hipHostMalloc((void **)&r->host, r->slot_bytes * depth, hipHostMallocDefault) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i) {
    ok = hipMalloc((void **)&r->dev[i], nEch * 8) == hipSuccess;
    if (ok && fmt == 1) ok = hipMalloc(&r->raw[i], r->slot_bytes) == hipSuccess;
    if (ok) ok = hipEventCreateWithFlags(&r->ready[i], hipEventDisableTiming) == hipSuccess;
  }
  if (ok) ok = hipEventCreateWithFlags(&r->freed, hipEventDisableTiming) == hipSuccess;
  if (ok) ok = hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking) == hipSuccess;
  if (!ok) {
    tsdr_ring_free(r);
    return set_err(ctx, TSDR_ENOMEM, "ring allocation failed (%zu B pinned, 2 x %zu B device)", r->slot_bytes * depth, nEch * 8);
  }
  r->t0 = std::chrono::steady_clock::now();
  *out = r;
  return TSDR_OK;
}

void tsdr_ring_free(tsdr_ring *r) {
  if (!r) return;
  // (bounded like every host-side wait of the library: a copy or consumer stream that never completes keeps the ring's memory)
  if (r->ctx && ((r->copy && tsdr::wait_stream(r->ctx, r->copy, "tsdr_ring_free")) || tsdr::wait_stream(r->ctx, r->ctx->stream, "tsdr_ring_free"))) return;
  if (r->copy) (void)hipStreamDestroy(r->copy);
  for (int i = 0; i < 2; ++i) {
    if (r->dev[i]) (void)hipFree(r->dev[i]);
    if (r->raw[i]) (void)hipFree(r->raw[i]);
    if (r->ready[i]) (void)hipEventDestroy(r->ready[i]);
  }
  if (r->freed) (void)hipEventDestroy(r->freed);
  if (r->host) (void)hipHostFree(r->host);
  delete r;
}

// A slot must not change under a DMA that is still reading it (the reference takes a per-slot lock around its
// copyto!, :183-185).  Called by the producer before it touches slot `pos`.
static void ring_wait_slot_idle(tsdr_ring *r, int pos) {
  for (int i = 0; i < 2; ++i) {
    bool pending;
    {
      std::unique_lock<std::mutex> lk(r->m);
      r->cv.wait(lk, [&] { return !(r->staging[i] && r->dma_slot[i] == pos); });  // until its event is recorded
      pending = r->dma_slot[i] == pos;
    }
    if (pending) (void)tsdr::wait_event(r->ctx, r->ready[i], "ring: DMA out of the slot about to be rewritten");
  }
}

static void ring_publish(tsdr_ring *r) {
  {
    std::lock_guard<std::mutex> g(r->m);
    r->writing_slot = -1;
    ++r->slot_gen[(size_t)r->ptr_write];                           // this slot now holds a newer buffer
    r->ptr_write = (r->ptr_write + 1) % r->depth;                  // atomic_update
    if (r->t_new == r->depth) ++r->overflow;                        // the oldest unread buffer was just overwritten
    r->t_new = r->t_new + 1 < r->depth ? r->t_new + 1 : r->depth;  // atomic_prodData: min(ptr+1, depth)
    ++r->produced;
    ++r->seq_written;
  }
  r->cv.notify_all();
}

/* the slot a zero-copy producer should fill next (e.g. recv!(slot, sdr)); tsdr_ring_commit publishes it */
void *tsdr_ring_write_ptr(tsdr_ring *r) {
  if (!r) return nullptr;
  int pos;
  {
    std::lock_guard<std::mutex> g(r->m);
    pos = r->ptr_write;
    r->writing_slot = pos;
  }
  ring_wait_slot_idle(r, pos);
  return r->host + (size_t)pos * r->slot_bytes;
}

int tsdr_ring_commit(tsdr_ring *r) {
  if (!r) return TSDR_EINVAL;
  ring_publish(r);
  return TSDR_OK;
}

/* circ_put! (:161-173): copy one buffer into slot ptr_write; does not wait for the consumer; overwrites the oldest
 * unread buffer when the ring is full */
int tsdr_ring_put(tsdr_ring *r, const void *data) {
  if (!r || !data) return TSDR_EINVAL;
  void *dst = tsdr_ring_write_ptr(r);
  std::memcpy(dst, data, r->slot_bytes);
  ring_publish(r);
  return TSDR_OK;
}

/* circ_take! (:177-190) with the copy landing on the device: blocks (up to timeout_ms, < 0 = forever) until a buffer
 * is available, returns a device pointer to its nEch ComplexF32 samples, valid until the second next take.  The
 * context's stream is ordered after the transfer; work the caller enqueues on it afterwards may use the buffer.
 * Returns TSDR_EBOUNDS on timeout / stop. */
int tsdr_ring_take_d(tsdr_ring *r, int timeout_ms, float **dev_iq) {
  if (!r || !dev_iq) return TSDR_EINVAL;
  tsdr_ctx *ctx = r->ctx;
  *dev_iq = nullptr;
  int slot, d;
  bool staged;
  {
    std::unique_lock<std::mutex> lk(r->m);
    auto pred = [&] { return r->t_new > 0 || r->stop; };
    if (timeout_ms < 0) r->cv.wait(lk, pred);
    else if (!r->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), pred)) return set_err(ctx, TSDR_EBOUNDS, "ring: no buffer within %d ms", timeout_ms);
    if (r->t_new == 0) return set_err(ctx, TSDR_EBOUNDS, "ring: stopped");
    r->cv.wait(lk, [&] { return r->writing_slot != r->ptr_read; });  // full ring: the producer is inside this very slot
    slot = r->ptr_read;
    r->ptr_read = (r->ptr_read + 1) % r->depth;   // atomic_update
    r->t_new -= 1;                                 // atomic_consData
    ++r->consumed;
    // was this slot already sent ahead?  A prefetch is valid only if the producer has not rewritten the slot since
    // the DMA was issued (a full ring overwrites slot ptr_read itself, :125-129).  The test is per slot: lifetime
    // counters would include every overflow ever recorded and never recover after the first `depth` drops.
    d = r->next_dev;
    staged = r->staged_seq[d] == r->seq_read && r->staged_gen[d] == r->slot_gen[(size_t)slot];
    ++r->seq_read;
    if (staged) ++r->prefetch_hits; else ++r->prefetch_misses;
    if (!staged) { r->dma_slot[d] = slot; r->staging[d] = true; }
  }
  // the device buffer about to be refilled further down (the other one) was handed out two takes ago: the
  // caller's stream has to be past its consumers before the DMA may overwrite it
  TSDR_HIP(ctx, hipEventRecord(r->freed, ctx->stream));
  TSDR_HIP(ctx, hipStreamWaitEvent(r->copy, r->freed, 0));
  if (!staged) {
    int rc = ring_stage(r, slot, d);
    { std::lock_guard<std::mutex> g(r->m); r->staging[d] = false; }
    r->cv.notify_all();
    if (rc) return rc;
  }
  TSDR_HIP(ctx, hipStreamWaitEvent(ctx->stream, r->ready[d], 0));
  *dev_iq = r->dev[d];
  // send the next committed slot ahead into the other device buffer
  int nslot = -1;
  {
    std::lock_guard<std::mutex> g(r->m);
    r->next_dev = d ^ 1;
    r->staged_seq[d ^ 1] = -1;
    r->dma_slot[d ^ 1] = -1;
    if (r->t_new > 0 && r->writing_slot != r->ptr_read) {
      nslot = r->ptr_read;
      r->staged_seq[d ^ 1] = r->seq_read;
      r->staged_gen[d ^ 1] = r->slot_gen[(size_t)nslot];
      r->dma_slot[d ^ 1] = nslot;
      r->staging[d ^ 1] = true;
    }
  }
  if (nslot >= 0) {
    int rc = ring_stage(r, nslot, d ^ 1);
    { std::lock_guard<std::mutex> g(r->m); r->staging[d ^ 1] = false; }
    r->cv.notify_all();
    if (rc) return rc;
  }
  return TSDR_OK;
}

int tsdr_ring_stop(tsdr_ring *r) {
  if (!r) return TSDR_EINVAL;
  { std::lock_guard<std::mutex> g(r->m); r->stop = true; }
  r->cv.notify_all();
  return TSDR_OK;
}

/* counters of print_summary (:333-341): produced / consumed buffers, overflows (buffers overwritten unread), and the
 * two rates in MS/s since creation */
int tsdr_ring_stats(tsdr_ring *r, unsigned long long *produced, unsigned long long *consumed, unsigned long long *overflow,
                    double *producer_msps, double *consumer_msps) {
  if (!r) return TSDR_EINVAL;
  std::lock_guard<std::mutex> g(r->m);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - r->t0).count();
  if (produced) *produced = r->produced;
  if (consumed) *consumed = r->consumed;
  if (overflow) *overflow = r->overflow;
  if (producer_msps) *producer_msps = dt > 0 ? (double)r->produced * (double)r->nEch / dt / 1e6 : 0.0;
  if (consumer_msps) *consumer_msps = dt > 0 ? (double)r->consumed * (double)r->nEch / dt / 1e6 : 0.0;
  return TSDR_OK;
}

/* how many takes found their buffer already sent ahead (H2D overlapped with the previous buffer's kernels) and how
 * many had to stage it at take time (first take, empty ring, or a slot the producer rewrote after the prefetch) */
int tsdr_ring_prefetch_stats(tsdr_ring *r, unsigned long long *hits, unsigned long long *misses) {
  if (!r) return TSDR_EINVAL;
  std::lock_guard<std::mutex> g(r->m);
  if (hits) *hits = r->prefetch_hits;
  if (misses) *misses = r->prefetch_misses;
  return TSDR_OK;
}

}  // extern "C"
