"""GPU tests of the int16 I/Q input of the frame loop (tsdr_frames_sc16_d / tsdr_frames_submit_sc16_d, ring fmt "sc16raw"): what
SDR hardware delivers (AtomicAbstractSDRs.jl:284-306 before its conversion) goes into the image kernels as it is, and every
sample becomes ComplexF32(re, im) * scale in the kernels' loaders.  The bar: bit-identical -- rasters, frames, IIR state, sync
indices -- to the ComplexF32 entry points on the samples converted on the host with the same product, on every route that
reads IQ: the FAST raster walk, the FAST raster-free tap kernel, the sync guard's exact re-evaluation, whole exact buffers
(TSDR_EXACT), narrow geometries (the run-time-format kernels) and the pipelined submission."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _capture(synth, Fs, x_t, y_t, fv, n, card="box"):
    z = synth.synth_leak(Fs, x_t, y_t, fv, n, card=card)
    peak = float(np.max(np.abs(z.view(np.float32))))
    scale = np.float32(peak / 2047.0)
    q = np.round(z.view(np.float32) / scale).astype(np.int16)
    # the ComplexF32 samples the loaders form: float(int16) * scale, one rounding
    cf = (q.astype(np.float32) * scale).view(np.complex64)
    return q, scale, cf


def _run(ctx, tsdr, cf, q, scale, S, y_t, x_t, want_raster, pipelined=False, nsplit=1):
    from tempestsdr_jl_amd import api
    npx, P = 600 * 800, x_t * y_t
    nb = cf.size // S
    out = {}
    for name in ("cf32", "sc16"):
        sync = tsdr.SyncXY(ctx, 600, 800)
        d_state = ctx.upload(np.zeros(npx, np.float32))
        d_in = ctx.upload(cf.view(np.float32)) if name == "cf32" else ctx.upload(q)
        d_fr, d_ix = ctx.dev_alloc(nb * npx * 4), ctx.dev_alloc(nb * 8)
        d_ra = ctx.dev_alloc(nb * P * 4) if want_raster else None
        try:
            per = nb // nsplit
            for c in range(nsplit):
                cnt = per if c < nsplit - 1 else nb - per * (nsplit - 1)
                o_in = d_in + c * per * S * (8 if name == "cf32" else 4)
                fr, ix = d_fr + c * per * npx * 4, d_ix + c * per * 8
                ra = d_ra + c * per * P * 4 if want_raster else None
                if name == "cf32":
                    f = api.frames_submit_d if pipelined else api.frames_d
                    n = f(ctx, sync, o_in, cnt * S, S, y_t, x_t, np.float32(0.1), True, d_state, fr, ra, ix)
                else:
                    n = api.frames_sc16_d(ctx, sync, o_in, scale, cnt * S, S, y_t, x_t, np.float32(0.1), True, d_state, fr, ra, ix,
                                          submit=pipelined)
                assert n == cnt
            ctx.synchronize()
            out[name] = (ctx.download(d_fr, (nb * npx,), np.uint32), ctx.download(d_ix, (nb * 2,), np.int32),
                         ctx.download(d_state, (npx,), np.uint32),
                         ctx.download(d_ra, (nb * P,), np.uint32) if want_raster else np.zeros(0, np.uint32))
        finally:
            sync.close()
            for p in (d_state, d_in, d_fr, d_ix, d_ra):
                if p is not None:
                    ctx.dev_free(p)
    for a, b in zip(out["cf32"], out["sc16"]):
        assert np.array_equal(a, b)
    return out["sc16"]


@pytest.mark.parametrize("precision", ["fast", "exact"])
@pytest.mark.parametrize("want_raster", [True, False])
@pytest.mark.parametrize("geom", [(2.0e6, 1056, 628, 5), (20e6, 2576, 1125, 3), (2.0e6, 900, 590, 2)])
def test_sc16_frames_equal_cf32_frames(ctx, tsdr, synth, precision, want_raster, geom):
    Fs, x_t, y_t, nfr = geom                      # (590 lines < 600: no in-walk downgrade -- the run-time-format kernels)
    S = synth.samples_per_frame(Fs, 60.0)
    q, scale, cf = _capture(synth, Fs, x_t, y_t, 60.0, S * nfr + 11)
    ctx.set_precision(precision)
    try:
        _run(ctx, tsdr, cf, q, scale, S, y_t, x_t, want_raster)
    finally:
        ctx.set_precision("fast")


def test_sc16_through_the_sync_guard_and_the_pipeline(ctx, tsdr, synth):
    """The plateau leak flags most frames: the guard's exact re-evaluation reads the int16 samples too (one by one with
    "sync_guard_auto" 0, whole exact buffers with it on), and the pipelined submission takes the same loaders."""
    Fs, x_t, y_t, nfr = 2.0e6, 1056, 628, 12
    S = synth.samples_per_frame(Fs, 60.0)
    q, scale, cf = _capture(synth, Fs, x_t, y_t, 60.0, S * nfr, card="plateau")
    for auto in (0, 1):
        ctx.set_option("sync_guard_auto", auto)
        ctx.sync_guard_stats(reset=True)
        try:
            _run(ctx, tsdr, cf, q, scale, S, y_t, x_t, False, pipelined=True, nsplit=4)
            checked, flagged = ctx.sync_guard_stats()
            assert flagged > 0, "the plateau leak should have flagged frames"
        finally:
            ctx.set_option("sync_guard_auto", 1)


def test_sc16_guard_beyond_one_guard_launch(ctx, tsdr, synth):
    """More than kGuardChunk = 256 frames in one call: the guard runs as several launches, and the later ones must find their
    frames' int16 samples at f0 * S * 4 bytes (they were offset as ComplexF32, i.e. read frame 2 * f0 and past the
    buffer).  The plateau leak flags frames in every launch's range; one-by-one re-evaluation ("sync_guard_auto" 0: the adaptive
    route depends on the calls before, which differ between the two formats' runs here)."""
    Fs, x_t, y_t, nfr = 2.0e6, 1056, 628, 300
    S = synth.samples_per_frame(Fs, 60.0)
    q, scale, cf = _capture(synth, Fs, x_t, y_t, 60.0, S * nfr, card="plateau")
    ctx.set_option("sync_guard_auto", 0)
    ctx.sync_guard_stats(reset=True)
    try:
        _run(ctx, tsdr, cf, q, scale, S, y_t, x_t, False)
        checked, flagged = ctx.sync_guard_stats()
        assert checked == 2 * nfr and flagged > 2      # (two runs: cf32 and sc16)
    finally:
        ctx.set_option("sync_guard_auto", 1)


def test_guard_ring_entry_carries_the_calls_totals(tsdr, synth):
    """One pinned ring entry per CALL, written by its last guard launch with the call's totals (a call of 300 frames is two
    launches; round 5 let each launch overwrite the entry with its own counts, so the adaptive route saw 44 frames of 300).
    300 plateau frames per call, > 15 % flagged: the entry of call 1 is folded at call 4 (the lag is 3 calls) and, being a
    full window (>= 60 frames) by itself, switches the route at once -- call 4 is the first whole buffer run exactly."""
    from tempestsdr_jl_amd import api
    Fs, x_t, y_t, nfr = 2.0e6, 1056, 628, 300
    S = synth.samples_per_frame(Fs, 60.0)
    q, scale, cf = _capture(synth, Fs, x_t, y_t, 60.0, S * nfr, card="plateau")
    npx = 600 * 800
    ctx = tsdr.Context(0)
    sync = tsdr.SyncXY(ctx, 600, 800)
    d_state, d_in = ctx.upload(np.zeros(npx, np.float32)), ctx.upload(q)
    d_fr, d_ix = ctx.dev_alloc(nfr * npx * 4), ctx.dev_alloc(nfr * 8)
    try:
        seen = []
        for k in range(5):
            api.frames_sc16_d(ctx, sync, d_in, scale, nfr * S, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, None, d_ix)
            ctx.synchronize()
            seen.append(ctx.sync_guard_auto())
        checked, flagged = ctx.sync_guard_stats()
        assert flagged > 0.15 * checked, (checked, flagged)
        assert [s[1] for s in seen] == [0, 0, 0, 1, 2], seen      # whole buffers exact so far, after calls 1 .. 5
        assert ctx.wait_stats() == (0, 0)
    finally:
        sync.close()
        for p in (d_state, d_in, d_fr, d_ix):
            ctx.dev_free(p)
        ctx.close()


def test_ring_sc16raw_hands_out_the_int16_pairs(ctx, tsdr, synth):
    """ring fmt "sc16raw": the H2D DMA moves 4 bytes per sample and nothing expands them; frames_sc16_d on the buffer the
    ring hands out equals frames_d on the expanded ring's."""
    from tempestsdr_jl_amd import api
    Fs, x_t, y_t, nfr = 2.0e6, 1056, 628, 3
    S = synth.samples_per_frame(Fs, 60.0)
    q, scale, cf = _capture(synth, Fs, x_t, y_t, 60.0, S * nfr)
    npx = 600 * 800
    res = []
    for fmt in ("sc16raw", "sc16"):
        ring = tsdr.StagingRing(ctx, cf.size, 3, fmt=fmt, scale=float(scale))
        sync = tsdr.SyncXY(ctx, 600, 800)
        d_state, d_fr, d_ix = ctx.upload(np.zeros(npx, np.float32)), ctx.dev_alloc(nfr * npx * 4), ctx.dev_alloc(nfr * 8)
        try:
            ring.put(q)
            d = ring.take_d(1000)
            if fmt == "sc16raw":
                assert np.array_equal(ctx.download(d, (2 * cf.size,), np.int16), q)
                api.frames_sc16_d(ctx, sync, d, scale, cf.size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, None, d_ix)
            else:
                api.frames_d(ctx, sync, d, cf.size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, None, d_ix)
            ctx.synchronize()
            res.append((ctx.download(d_fr, (nfr * npx,), np.uint32), ctx.download(d_ix, (nfr * 2,), np.int32)))
        finally:
            sync.close()
            ring.close()
            for p in (d_state, d_fr, d_ix):
                ctx.dev_free(p)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
