"""Host -> GPU ingest through the pinned staging ring: the headless stand-in for the
`recv!(buffer, csdr)` + coreProcessing loop of GUI.jl:150-178 when the IQ buffers start in host memory.

`stream_frames` is the loop itself; `bench_ingest` times it with a producer thread that publishes pre-filled pinned
slots as fast as the ring accepts them (a zero-copy SDR driver), so the measured rate is what PCIe + the kernels
sustain together -- the PCIe-inclusive number next to bench.py's HBM-resident `value`."""
import threading
import time

import numpy as np

from . import api


def stream_frames(ctx, ring, sync, n_buffers, nEch, S, y_t, x_t, alpha, state, frames_out, sync_idx, timeout_ms=10000,
                  sc16_scale=None):
    """Consumer loop: take a device buffer from the ring, run the frame path on it, repeat.  Returns frames done.
    sc16_scale: the ring hands out int16 pairs (fmt "sc16raw"), which the frame kernels convert in their loaders."""
    done = 0
    for _ in range(n_buffers):
        d_iq = ring.take_d(timeout_ms)
        if sc16_scale is None:
            done += api.frames_d(ctx, sync, d_iq, nEch, S, y_t, x_t, alpha, True, state, frames_out, None, sync_idx)
        else:
            done += api.frames_sc16_d(ctx, sync, d_iq, sc16_scale, nEch, S, y_t, x_t, alpha, True, state, frames_out, None, sync_idx)
    return done


def bench_ingest(ctx, tsdr, iq_host, S, y_t, x_t, seconds=1.0, depth=4, fmt="cf32"):
    """frames/s with every buffer crossing PCIe: H2D DMA of buffer k+1 under the kernels of buffer k."""
    import torch
    nEch = iq_host.size
    npx = tsdr.RENDER_H * tsdr.RENDER_W
    nb = nEch // S
    scale = 1.0
    if fmt in ("sc16", "sc16raw"):
        peak = float(np.max(np.abs(iq_host.view(np.float32)))) or 1.0
        scale = peak / 2047.0
        src = np.round(iq_host.view(np.float32) / scale).astype(np.int16)
    else:
        src = iq_host.view(np.float32)
    ring = tsdr.StagingRing(ctx, nEch, depth, fmt=fmt, scale=scale)
    dev = torch.device("cuda", torch.cuda.current_device())
    state = torch.zeros(npx, dtype=torch.float32, device=dev)
    frames_out = torch.empty(nb * npx, dtype=torch.float32, device=dev)
    sync_idx = torch.zeros(2 * nb, dtype=torch.int32, device=dev)
    sync = tsdr.SyncXY(ctx, tsdr.RENDER_H, tsdr.RENDER_W)
    for _ in range(depth):  # fill every pinned slot once; the timed producer only publishes them
        ring.write_view()[:] = src
        ring.commit()
    for _ in range(depth):
        ring.take_d(10000)
    ctx.synchronize()
    stop = threading.Event()
    lib, h = ctx.lib, ring.h

    def producer():
        while not stop.is_set():
            st = ring.stats()
            if st["produced"] - st["consumed"] < depth - 1:  # keep the ring fed without lapping the consumer
                lib.tsdr_ring_write_ptr(h)
                lib.tsdr_ring_commit(h)
            else:
                time.sleep(0)

    th = threading.Thread(target=producer, daemon=True)
    th.start()
    n_done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        n_done += stream_frames(ctx, ring, sync, 4, nEch, S, y_t, x_t, np.float32(0.1), state, frames_out, sync_idx,
                                sc16_scale=scale if fmt == "sc16raw" else None)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    ring.stop()
    th.join()
    st = ring.stats()
    ring.close()
    bytes_per_buf = nEch * (8 if fmt == "cf32" else 4)
    return {"fmt": fmt, "frames_per_s": round(n_done / dt, 1), "msps": round(n_done / nb * nEch / dt / 1e6, 1),
            "pcie_GBs": round(n_done / nb * bytes_per_buf / dt / 1e9, 2), "buffers": n_done // nb, "seconds": round(dt, 3),
            "overflow": st["overflow"], "depth": depth}


# ---- the small host-side pieces of the runtime loop -----------------------------------------------
class FrameChannel:
    """Drop-oldest bounded channel of GUI.jl:111-118 (`non_blocking_put!`): the producer never blocks; when the
    channel is full the oldest frame is taken out to make room."""

    def __init__(self, sz_max):
        import collections
        if sz_max < 1:
            raise ValueError("channel size must be >= 1")
        self._q = collections.deque()
        self.sz_max = int(sz_max)
        self._cv = threading.Condition()
        self.dropped = 0

    def put(self, image):
        with self._cv:
            if len(self._q) == self.sz_max:  # "This is full": take!(channelImage)
                self._q.popleft()
                self.dropped += 1
            self._q.append(image)
            self._cv.notify()

    def take(self, timeout=None):
        with self._cv:
            if not self._cv.wait_for(lambda: len(self._q) > 0, timeout):
                raise IndexError("FrameChannel.take: timed out")
            return self._q.popleft()

    def __len__(self):
        with self._cv:
            return len(self._q)


def record_buffers(take_host_buffer, nb_buffer, nEch, path, fmt="single"):
    """The record task of GUI.jl:181-190: nbBuffer consecutive recv! results concatenated and written with
    writeComplexBinary (DatBinaryFiles.jl:15-31).  `take_host_buffer()` returns one complex64[nEch] buffer."""
    from .dat_files import writeComplexBinary
    rec = np.empty(nb_buffer * nEch, np.complex64)
    for n in range(nb_buffer):
        rec[n * nEch:(n + 1) * nEch] = take_host_buffer()
    writeComplexBinary(rec, path, fmt)
    return rec.size
