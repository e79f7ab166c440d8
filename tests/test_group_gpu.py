"""GPU tests of the multi-GPU split behind the C ABI (tsdr_group_*: one process, one context + one RCCL communicator per
device, include/tempest_hip.h): what the reference's single-process runtime (GUI.jl:380-382) would hold to use several GPUs.

A group of ONE device must return the single-context results bit for bit -- per-frame and per-lag arithmetic is unchanged
and the 1-rank all-reduce is the identity; this runs on every GPU box and executes the RCCL path (communicator creation,
ncclAllReduce on the library's stream) on real hardware.  The N > 1 split -- frame ranges, H2D slices, halos with their wrap,
gather offsets -- runs on every box too, with 2 / 3 / 5 members SHARING the one device (a group may list a device more than
once; it then exchanges by copies and adds instead of RCCL, which takes one rank per device).  Groups of 2 / 4 / 8 distinct
devices over RCCL: tests/test_zz_group_devices_gpu.py (a child process per group, last in the suite)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _ndev():
    import torch
    return torch.cuda.device_count()


@pytest.fixture(scope="module")
def capture(synth):
    # 800x600@60Hz mode (VideoConfigurations.jl:27) at 2 MS/s: 7 frames + a ragged tail
    Fs, x_t, y_t, fv = 2.0e6, 1056, 628, 60.0
    S = synth.samples_per_frame(Fs, fv)
    return Fs, x_t, y_t, S, synth.synth_leak(Fs, x_t, y_t, fv, 7 * S + 123)


def _frames_equal(a, b):
    assert a["n_frames"] == b["n_frames"]
    assert np.array_equal(a["sync_idx"], b["sync_idx"]), (a["sync_idx"].tolist(), b["sync_idx"].tolist())
    for f, (x, y) in enumerate(zip(a["frames"], b["frames"])):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"frame {f}"
    for f, (x, y) in enumerate(zip(a.get("raster", []), b.get("raster", []))):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"raster {f}"


@pytest.mark.parametrize("precision", ["fast", "exact"])
def test_group_of_one_frames_equal_single_context(ctx, tsdr, capture, precision):
    """tsdr_group_frames on a one-device group = tsdr_frames, bit for bit, over two successive buffers (the lagged s_y and the
    IIR state cross the call boundary inside the group's SyncXY / the caller's imageOut)."""
    Fs, x_t, y_t, S, iq = capture
    g = tsdr.Group([0])
    try:
        assert len(g) == 1
        g.set_precision(precision)
        ctx.set_precision(precision)
        sync = tsdr.SyncXY(ctx, 600, 800)
        s1 = np.zeros((600, 800), np.float32, order="F")
        s2 = np.zeros((600, 800), np.float32, order="F")
        for part in (iq[: 4 * S + 50], iq[4 * S:]):
            a = ctx.frames(sync, part, S, y_t, x_t, np.float32(0.1), s1, want_raster=True)
            b = g.frames(part, S, y_t, x_t, np.float32(0.1), s2, want_raster=True)
            _frames_equal(a, b)
            assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
        # a fresh SyncXY: the first frame's s_y is 1 again (FrameSynchronisation.jl:66 on all-zero beta_y)
        g.sync_reset()
        b = g.frames(iq[: 2 * S], S, y_t, x_t, np.float32(0.1), s2)
        assert b["sync_idx"][0][0] == 1
        sync.close()
    finally:
        ctx.set_precision("fast")
        g.close()


def test_group_of_one_search_and_welch_equal_single_context(ctx, tsdr, capture):
    Fs, x_t, y_t, S, iq = capture
    g = tsdr.Group([0])
    try:
        G1, p1, v1 = ctx.autocorr_search(iq, Fs, 0.0, 0.04, 50, 90)
        G2, p2, v2 = g.autocorr_search(iq, Fs, 0.0, 0.04, 50, 90)            # auto: the root alone
        assert g.timing()[0] == "root"
        assert np.array_equal(G1.view(np.uint32), G2.view(np.uint32)) and p1 == p2 and v1 == v2
        # forced through the sharded route: partial sums over the whole range, a ONE-rank ncclAllReduce, the non-linear step
        G3, p3, v3 = g.autocorr_search(iq, Fs, 0.0, 0.04, 50, 90, route="sharded")
        route, ms = g.timing()
        assert route == "sharded" and all(m >= 0 for m in ms)
        x = (iq.real.astype(np.float32) ** 2 + iq.imag.astype(np.float32) ** 2).astype(np.float32)
        Go, _ = O.calculate_autocorrelation(x, Fs, 0.0, 0.04)
        assert np.max(np.abs(G3 - Go)) < 2e-4 and p3 == p1
        # real input, linear scale, a window that does not start at lag 0
        G4, p4, _ = g.autocorr_search(x, Fs, 0.002, 0.03, 50, 90, scale="lin", route="sharded")
        G5, p5, _ = ctx.autocorr_search(x, Fs, 0.002, 0.03, 50, 90, scale="lin")
        assert np.max(np.abs(G4 - G5)) <= 2e-5 * np.max(np.abs(G5)) and p4 == p5
        for size in (1024, 1000):
            f1, y1 = ctx.getWelch(Fs, iq, sizeFFT=size)
            f2, y2 = g.getWelch(Fs, iq, sizeFFT=size)
            assert np.array_equal(f1, f2) and np.array_equal(y1.view(np.uint32), y2.view(np.uint32)), size
    finally:
        g.close()


def test_group_misuse(tsdr):
    with pytest.raises(tsdr.TempestHIPError):
        tsdr.Group([])
    with pytest.raises(tsdr.TempestHIPError):
        tsdr.Group([_ndev() + 3])
    g = tsdr.Group([0])
    try:
        with pytest.raises(IndexError):  # BoundsError of Autocorrelations.jl:33: the capture is shorter than maxDelay * Fs
            g.autocorr_search(np.ones(1000, np.complex64), 2.0e6, 0.0, 0.04)
        with pytest.raises(AssertionError):
            g.set_option("no_such_option", 1)
        s = np.zeros((600, 800), np.float32, order="F")
        assert g.frames(np.ones(10, np.complex64), 33333, 628, 1056, np.float32(0.1), s)["n_frames"] == 0
    finally:
        g.close()


@pytest.mark.parametrize("n", [2, 3, 5])
def test_members_sharing_one_device_run_the_split_logic(ctx, tsdr, capture, n):
    """N > 1 members on ONE device (the device listed n times: exchanges by device-to-device copies and adds, no RCCL -- which
    wants one rank per device): everything else is the code a group of n distinct devices runs -- frame ranges (7 frames dealt
    raggedly over 2 / 3 / 5 members), per-member H2D slices, the gather offsets of images, keys and rasters, the search's
    segment + halo slices with the wrap at n, the Welch segment ranges.  Frames bit for bit the single-context result; the
    sharded search within 2e-4 dB with the same argmax (its partial sums are added in member order, not n's own order)."""
    Fs, x_t, y_t, S, iq = capture
    g = tsdr.Group([0] * n)
    try:
        assert len(g) == n
        for precision in ("fast", "exact"):
            g.set_precision(precision)
            g.set_option("sync_guard_auto", 0)
            g.sync_reset()
            ctx.set_precision(precision)
            sync = tsdr.SyncXY(ctx, 600, 800)
            s1 = np.zeros((600, 800), np.float32, order="F")
            s2 = np.zeros((600, 800), np.float32, order="F")
            for part in (iq[: 4 * S + 50], iq[4 * S:], iq[: S + 7]):     # 4, 3 and 1 frames: fewer frames than members too
                _frames_equal(ctx.frames(sync, part, S, y_t, x_t, np.float32(0.1), s1, want_raster=True),
                              g.frames(part, S, y_t, x_t, np.float32(0.1), s2, want_raster=True))
                assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
            sync.close()
        G1, p1, _ = ctx.autocorr_search(iq, Fs, 0.0, 0.04, 50, 90)
        G2, p2, _ = g.autocorr_search(iq, Fs, 0.0, 0.04, 50, 90, route="sharded")
        assert g.timing()[0] == "sharded"
        assert np.max(np.abs(G1 - G2)) < 2e-4 and p1 == p2
        # real input, linear scale, another window (n = 2 indexMax = 40 000 of the 160 000 samples: Autocorrelations.jl:27)
        x = (iq.real.astype(np.float32) ** 2 + iq.imag.astype(np.float32) ** 2).astype(np.float32)[: 8 * 20_000]
        G3, p3, _ = g.autocorr_search(x, Fs, 0.0, 0.01, 150, 400, scale="lin", route="sharded")
        G4, p4, _ = ctx.autocorr_search(x, Fs, 0.0, 0.01, 150, 400, scale="lin")
        assert np.max(np.abs(G3 - G4)) <= 2e-5 * np.max(np.abs(G4)) and p3 == p4
        for size in (1024, 1000):
            _, y1 = ctx.getWelch(Fs, iq, sizeFFT=size)
            _, y2 = g.getWelch(Fs, iq, sizeFFT=size)
            assert np.max(np.abs(y1 - y2)) < 2e-4, size
    finally:
        ctx.set_precision("fast")
        g.close()


@pytest.mark.parametrize("threads,pin", [(0, 0), (2, 0), (2, 1), (0, 1)])
def test_member_threads_and_pinned_arrays_change_nothing_but_the_transfers(ctx, tsdr, capture, threads, pin):
    """Round 6: each member's stage (its H2D slice, its launches, its raster D2H) runs on that member's own host thread, so
    that copies from / into the caller's pageable arrays overlap across members ("member_threads" 0: the caller's thread
    drives them all in turn, rounds 1-5; 2: threads even for members sharing a device, as here); "pin_host" 1 page-locks the caller's arrays for the duration of each call.  Four members on
    the one device: results are the single-context ones bit for bit in every combination, call after call on the same arrays
    and on overlapping ones (nothing stays registered between calls: the single-context copies in between would fail on a
    range that only partly overlaps a registered one)."""
    Fs, x_t, y_t, S, iq = capture
    g = tsdr.Group([0] * 4)
    try:
        g.set_option("member_threads", threads)
        g.set_option("pin_host", pin)
        g.set_option("sync_guard_auto", 0)
        sync = tsdr.SyncXY(ctx, 600, 800)
        s1 = np.zeros((600, 800), np.float32, order="F")
        s2 = np.zeros((600, 800), np.float32, order="F")
        big = np.concatenate([iq, iq, iq])         # > 1 MB per array: what "pin_host" registers
        for part in (big[: 9 * S], big[: 9 * S], big[2 * S: 14 * S + 5], iq[: 3 * S]):
            _frames_equal(ctx.frames(sync, part, S, y_t, x_t, np.float32(0.1), s1, want_raster=True),
                          g.frames(part, S, y_t, x_t, np.float32(0.1), s2, want_raster=True))
            assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
        sync.close()
        G1, p1, _ = ctx.autocorr_search(iq, Fs, 0.0, 0.04, 50, 90)
        for _ in range(2):
            G2, p2, _ = g.autocorr_search(iq, Fs, 0.0, 0.04, 50, 90, route="sharded")
            assert np.max(np.abs(G1 - G2)) < 2e-4 and p1 == p2
        _, y1 = ctx.getWelch(Fs, big, sizeFFT=1024)
        _, y2 = g.getWelch(Fs, big, sizeFFT=1024)
        assert np.max(np.abs(y1 - y2)) < 2e-4
        g.set_option("pin_host", 0)
        _, y3 = g.getWelch(Fs, big, sizeFFT=1024)
        assert np.array_equal(y2, y3)
    finally:
        g.close()
