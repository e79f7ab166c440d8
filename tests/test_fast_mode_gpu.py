"""GPU parity, TSDR_FAST mode (the default of the frame loop): exact-rational coordinates carried in integers, one
f64 FMA per blend, projection sums formed inside the raster kernel.

Bar (north_star): image pixels within 1e-5 relative, identical frame-sync indices.  FAST is designed
so that each blend stays within 1 ulp of the f64-faithful evaluation and |IQ| within 1.5 ulp (hardware
v_sqrt_f32); the tests assert 6e-7 relative (about 5 ulp; measured over 360 random cases: 4.7e-7 at ~1.2 samples per raster pixel, below 3.6e-7 at the ratios of C2 / C5) against the CPU oracle and IDENTICAL sync indices on
every frame: the library's sync guard (csrc/guard.h) re-evaluates, in the exact operation sequence, every frame whose
top-2 beta margin is below 2e-5, so no tie escape is needed (sync_margin.fast_vs_oracle keeps one only for the
guard-off comparison in test_sync_guard_is_what_makes_indices_identical)."""
import numpy as np
import pytest

import oracle_lib as O
from sync_margin import fast_vs_oracle

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(99)
RTOL = 6e-7
RTOL_TAPS = 6e-7   # the raster-free tap kernel (32.32 fixed-point coordinates, f32 blends) on white noise: worst seen 4.1e-7
                   # (1.43e-6 with one rounded blend weight instead of two exact ones: DESIGN section 2)


def relerr(got, want):
    want = np.asarray(want, np.float64)
    return float(np.max(np.abs(np.asarray(got, np.float64) - want) / np.maximum(np.abs(want), 1e-30)))


def test_default_mode_is_fast(ctx):
    assert ctx.precision == "fast"


@pytest.mark.parametrize("S,y_t,x_t", [(1200, 30, 40), (137, 30, 40), (3333, 70, 130), (26001, 125, 161),
                                       (333333, 1125, 2576), (3333333, 1125, 2576), (833333, 2250, 4400),
                                       (800000, 100, 128), (40001, 65, 300)])
def test_per_function_api_is_exact_whatever_the_mode(ctx, S, y_t, x_t):
    """TSDR_FAST applies to the frame loop only.  sig_to_image / imresize called on their own run the oracle's
    operation sequence even while the context is in its default FAST mode -- bit for bit, on sign-changing input
    too (the worst case for any approximate blend: next to a zero crossing a pixel is a difference of O(1) terms)."""
    assert ctx.precision == "fast"
    sig = rng.standard_normal(S).astype(np.float32)
    got, want = ctx.sig_to_image(sig, y_t, x_t), O.sig_to_image(sig, y_t, x_t)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    small = np.asfortranarray(rng.standard_normal((y_t, min(x_t, 300))).astype(np.float32))
    assert np.array_equal(ctx.downgradeImage(small).view(np.uint32), O.downgradeImage(small).view(np.uint32))


@pytest.mark.parametrize("case", [
    dict(Fs=1.0e6, x_t=160, y_t=125, fv=50.0, nfr=3),     # S == P copy path, upscale to 600x800 (separate kernels)
    dict(Fs=2.0e6, x_t=1056, y_t=628, fv=60.0, nfr=3),    # fused raster+downgrade launch
    dict(Fs=20e6, x_t=2576, y_t=1125, fv=60.0, nfr=3),    # C2
    dict(Fs=200e6, x_t=2576, y_t=1125, fv=60.0, nfr=2),   # C3 (downsampling)
    dict(Fs=50e6, x_t=4400, y_t=2250, fv=60.0, nfr=2),    # C5 (4K60 total raster): 2P >= 2^24, f64 walk
    dict(Fs=4.0e6, x_t=1000, y_t=600, fv=60.0, nfr=2),    # y_t == 600: no in-walk downgrade (ratio 1 on one axis)
    dict(Fs=4.0e6, x_t=800, y_t=700, fv=60.0, nfr=2),     # x_t == 800
    dict(Fs=3.0e6, x_t=801, y_t=601, fv=50.0, nfr=2),     # ratios barely above 1: nearly every line/pixel owns a row/column
    dict(Fs=6.0e6, x_t=300, y_t=9000, fv=50.0, nfr=1),    # 143 line tiles: 32-bit position advance not applicable
    dict(Fs=1.0e6, x_t=2000, y_t=130, fv=50.0, nfr=2),    # two line tiles, ragged last one
])
@pytest.mark.parametrize("want_raster", [True, False])
def test_frames_fast(ctx, tsdr, synth, case, want_raster):
    S = synth.samples_per_frame(case["Fs"], case["fv"])
    iq = synth.synth_leak(case["Fs"], case["x_t"], case["y_t"], case["fv"], S * case["nfr"] + 321)
    r = fast_vs_oracle(ctx, tsdr, O, iq, S, case["y_t"], case["x_t"], 0.1, want_raster, RTOL)
    assert r["n_frames"] == case["nfr"] and not r["ties"]
    print("sync guard (checked, re-evaluated):", r["guard"])


def test_fast_and_exact_agree_on_extreme_samples(ctx):
    # |IQ| guard: tiny / huge / non-finite samples take the scaled path in FAST mode too
    vals = np.array([0.0, 1e-30, 1e-22, 1e-3, 1.0, 1e18, 3e38], np.float32)
    re, im = np.meshgrid(vals, vals)
    z = np.tile((re.ravel() + 1j * im.ravel()).astype(np.complex64), 40)  # 1960 samples
    st_f = np.zeros((600, 800), np.float32, order="F")
    st_e = np.zeros((600, 800), np.float32, order="F")
    f = ctx.frames(None, z, 980, 70, 130, np.float32(0.5), st_f, do_align=False, want_raster=True)
    ctx.set_precision("exact")
    try:
        e = ctx.frames(None, z, 980, 70, 130, np.float32(0.5), st_e, do_align=False, want_raster=True)
    finally:
        ctx.set_precision("fast")
    for a, b in zip(f["raster"], e["raster"]):
        ok = np.isfinite(b) & (b > 0)
        assert np.max(np.abs(a[ok] - b[ok]) / b[ok]) < RTOL
        assert np.array_equal(np.isfinite(a), np.isfinite(b))


@pytest.mark.parametrize("precision", ["fast", "exact"])
def test_frames_many_frames(ctx, tsdr, synth, precision):
    """A long buffer (66 frames of the 800x600@60 mode at 2 MS/s: > 1000 (frame, strip) units in one launch).
    EXACT must stay bit-identical, FAST within tolerance."""
    Fs, x_t, y_t, fv, nfr = 2.0e6, 1056, 628, 60.0, 66
    S = synth.samples_per_frame(Fs, fv)
    iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 11)
    if precision == "fast":
        # (frame 57 of this buffer is an exact f32 tie between beta_x columns 367 and 368 in the oracle: the sync guard
        # re-evaluates it, and every other frame within 2e-5 of a tie, in the exact sequence)
        r = fast_vs_oracle(ctx, tsdr, O, iq, S, y_t, x_t, 0.1, True, RTOL)
        assert r["n_frames"] == nfr and not r["ties"]
        print("sync guard (checked, re-evaluated):", r["guard"])
        assert r["guard"][1] >= 1
        return
    gs = np.zeros((600, 800), np.float32, order="F")
    os_ = np.zeros((600, 800), np.float32, order="F")
    ctx.set_precision("exact")
    try:
        g = ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), gs, want_raster=True)
    finally:
        ctx.set_precision("fast")
    o = O.frames(O.SyncXY(600, 800), iq, S, y_t, x_t, np.float32(0.1), os_, want_raster=True)
    assert g["n_frames"] == o["n_frames"] == nfr
    assert np.array_equal(g["sync_idx"], o["sync_idx"])
    for f in (0, 1, nfr // 2, nfr - 1):
        assert np.array_equal(g["raster"][f].view(np.uint32), o["raster"][f].view(np.uint32)), f
        assert np.array_equal(g["frames"][f].view(np.uint32), o["frames"][f].view(np.uint32)), f
    assert np.array_equal(gs.view(np.uint32), os_.view(np.uint32))


@pytest.mark.parametrize("want_raster", [True, False])
@pytest.mark.parametrize("mode", [-1, 0, 1, 2])
def test_frames_pipeline_matches_sequential(ctx, tsdr, synth, want_raster, mode):
    """tsdr_frames_submit_d / tsdr_frames_flush (the tail of a buffer in flight beside the image launch of the next, on the
    library's internal streams) must return, bit for bit, what one tsdr_frames_d call per buffer returns: same kernels,
    and the SyncXY / imageOut state threaded through the buffers.  Both arrangements ("pipe_mode" 0: image lane + tail
    lane; 1: whole buffers on alternating equal lanes; 2: one internal stream; -1: the measured choice, here still in its
    first trial), with and without rasters, five buffers so that the three slots wrap."""
    from tempestsdr_jl_amd import api
    Fs, x_t, y_t, fv, nfr, nbuf = 2.0e6, 1056, 628, 60.0, 4, 5
    S = synth.samples_per_frame(Fs, fv)
    P, npx = x_t * y_t, 600 * 800
    bufs = [synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 7, n0=b * (S * nfr + 7)) for b in range(nbuf)]

    def run(pipelined, flush=True):
        sync = tsdr.SyncXY(ctx, 600, 800)
        d_state = ctx.upload(np.zeros(npx, np.float32))
        d_iq = [ctx.upload(b.view(np.float32)) for b in bufs]
        d_fr = [ctx.dev_alloc(nfr * npx * 4) for _ in bufs]
        d_ra = [ctx.dev_alloc(nfr * P * 4) if want_raster else None for _ in bufs]
        d_ix = [ctx.dev_alloc(nfr * 8) for _ in bufs]
        try:
            for b in range(nbuf):
                f = api.frames_submit_d if pipelined else api.frames_d
                n = f(ctx, sync, d_iq[b], bufs[b].size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr[b], d_ra[b], d_ix[b])
                assert n == nfr
            if pipelined and flush:
                api.frames_flush(ctx)
            ctx.synchronize()  # header contract: complete after tsdr_synchronize even without a flush
            return ([ctx.download(p, (nfr * npx,), np.uint32) for p in d_fr],
                    [ctx.download(p, (nfr * P,), np.uint32) for p in d_ra] if want_raster else [],
                    [ctx.download(p, (nfr * 2,), np.int32) for p in d_ix], ctx.download(d_state, (npx,), np.uint32))
        finally:
            for p in [d_state] + d_iq + d_fr + [r for r in d_ra if r is not None] + d_ix:
                ctx.dev_free(p)

    a = run(False)
    ctx.set_option("pipe_mode", mode)
    try:
        for b in (run(True), run(True, flush=False)):
            for k in range(3):
                for x, y in zip(a[k], b[k]):
                    assert np.array_equal(x, y)
            assert np.array_equal(a[3], b[3])
    finally:
        ctx.set_option("pipe_mode", -1)


@pytest.mark.parametrize("want_raster", [False, True])
def test_pipeline_measured_choice_walks_every_arrangement(tsdr, synth, want_raster):
    """"pipe_mode" -1 (the default): the first 255 submissions of a configuration go through the eight candidate arrangements,
    fifteen buffers each, twice after a warm-up trial, with the pipeline run empty at every trial boundary, and the rest use the one measured fastest.  Results
    must be those of one tsdr_frames_d per buffer throughout -- bit for bit, the SyncXY / imageOut state threaded through all
    270 buffers -- and tsdr_frames_pipeline_info must report a settled choice with every candidate timed."""
    from tempestsdr_jl_amd import api
    ctx = tsdr.Context(0)   # (a context of its own: the measurement is per context and configuration)
    ctx.set_option("pipe_mode", -1)   # (whatever TSDR_PIPE_* presets the environment carries)
    ctx.set_option("pipe_tune", 1)
    Fs, x_t, y_t, fv, nfr, nbuf, ndist = 2.0e6, 1056, 628, 60.0, 1, 270, 5
    S = synth.samples_per_frame(Fs, fv)
    P, npx = x_t * y_t, 600 * 800
    bufs = [synth.synth_leak(Fs, x_t, y_t, fv, S * nfr, n0=b * S * nfr) for b in range(ndist)]

    def run(pipelined):
        sync = tsdr.SyncXY(ctx, 600, 800)
        d_state = ctx.upload(np.zeros(npx, np.float32))
        d_iq = [ctx.upload(b.view(np.float32)) for b in bufs]
        d_fr = [ctx.dev_alloc(nfr * npx * 4) for _ in range(nbuf)]
        d_ra = [ctx.dev_alloc(nfr * P * 4) if want_raster else None for _ in range(nbuf)]
        d_ix = [ctx.dev_alloc(nfr * 8) for _ in range(nbuf)]
        try:
            for b in range(nbuf):
                f = api.frames_submit_d if pipelined else api.frames_d
                assert f(ctx, sync, d_iq[b % ndist], bufs[0].size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr[b], d_ra[b], d_ix[b]) == nfr
            ctx.synchronize()
            import zlib
            return ([zlib.crc32(ctx.download(p, (nfr * npx,), np.uint32).tobytes()) for p in d_fr],
                    [zlib.crc32(ctx.download(p, (nfr * P,), np.uint32).tobytes()) for p in d_ra] if want_raster else [],
                    [ctx.download(p, (nfr * 2,), np.int32).tolist() for p in d_ix], ctx.download(d_state, (npx,), np.uint32))
        finally:
            sync.close()
            for p in [d_state] + d_iq + d_fr + [r for r in d_ra if r is not None] + d_ix:
                ctx.dev_free(p)

    try:
        a = run(False)
        assert ctx.pipeline_info()["trials_left"] == 17
        b = run(True)
        info = ctx.pipeline_info()
        print("\n", info["text"])
        assert info["trials_left"] == 0 and 0 <= info["chosen"] < 8 and all(v > 0 for v in info["ms_per_buffer"])
        # never an arrangement measured slower than the sequential order
        assert info["ms_per_buffer"][info["chosen"]] <= info["ms_per_buffer"][0]
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
        assert np.array_equal(a[3], b[3])
    finally:
        ctx.close()


def test_pipeline_measurement_is_kept_per_configuration(tsdr, synth):
    """A caller that leaves a configuration and comes back to it (GUI.jl's y_t / x_t corrections and their undo, a raster asked
    for now and then) finds the arrangement measured for it the first time: no second round of trials."""
    from tempestsdr_jl_amd import api
    ctx = tsdr.Context(0)
    ctx.set_option("pipe_mode", -1)
    ctx.set_option("pipe_tune", 1)
    Fs, x_t, y_t, fv = 2.0e6, 1056, 628, 60.0
    S = synth.samples_per_frame(Fs, fv)
    P, npx = x_t * y_t, 600 * 800
    buf = synth.synth_leak(Fs, x_t, y_t, fv, S)
    sync = tsdr.SyncXY(ctx, 600, 800)
    d_state = ctx.upload(np.zeros(npx, np.float32))
    d_iq = ctx.upload(buf.view(np.float32))
    d_fr, d_ra, d_ix = ctx.dev_alloc(npx * 4), ctx.dev_alloc(P * 4), ctx.dev_alloc(8)
    try:
        def submit(n, raster):
            for _ in range(n):
                assert api.frames_submit_d(ctx, sync, d_iq, buf.size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, d_ra if raster else None, d_ix) == 1
        submit(260, False)
        first = ctx.pipeline_info()
        assert first["trials_left"] == 0 and first["chosen"] >= 0
        submit(3, True)                      # another configuration: its own trials begin
        assert ctx.pipeline_info()["trials_left"] > 0
        submit(1, False)                     # back: settled at once, the same table
        again = ctx.pipeline_info()
        assert again["trials_left"] == 0 and again["chosen"] == first["chosen"] and again["ms_per_buffer"] == first["ms_per_buffer"]
        submit(260, True)                    # the other configuration is measured from the start (its 3 buffers were dropped)
        assert ctx.pipeline_info()["trials_left"] == 0
        submit(1, False)
        assert ctx.pipeline_info()["ms_per_buffer"] == first["ms_per_buffer"]
        api.frames_flush(ctx)
        ctx.synchronize()
    finally:
        sync.close()
        for p in (d_state, d_iq, d_fr, d_ra, d_ix):
            ctx.dev_free(p)
        ctx.close()


def test_pipeline_choice_is_taken_over_by_neighbouring_geometries_and_can_be_pinned(tsdr, synth):
    """GUI.jl:492-506: the user corrects y_t one line at a time.  Each new y_t is a new configuration for the pipeline; it must
    not pay 255 trial submissions again -- a configuration within 10 % of a measured one takes over its choice.  50 values of
    y_t: ONE measurement.  Two configurations in strict alternation from the start settle (on the sequential order) instead of
    cutting into each other's trials for ever.  "pipe_pin" k runs arrangement k with nothing measured; "pipe_measure" 1
    measures the current configuration again.  Results equal one tsdr_frames_d per buffer throughout."""
    from tempestsdr_jl_amd import api
    Fs, x_t, y0, fv = 2.0e6, 1056, 628, 60.0
    S = synth.samples_per_frame(Fs, fv)
    npx = 600 * 800
    buf = synth.synth_leak(Fs, x_t, y0, fv, S)

    def session():
        c = tsdr.Context(0)
        c.set_option("pipe_mode", -1)
        c.set_option("pipe_tune", 1)
        return c, tsdr.SyncXY(c, 600, 800), c.upload(np.zeros(npx, np.float32)), c.upload(buf.view(np.float32)), c.dev_alloc(npx * 4), c.dev_alloc(8)

    def close(c, sync, *ptrs):
        sync.close()
        for p in ptrs:
            c.dev_free(p)
        c.close()

    ctx, sync, d_state, d_iq, d_fr, d_ix = session()
    ref, rsync, r_state, r_iq, r_fr, r_ix = session()
    try:
        def submit(n, y_t, check=False):
            for _ in range(n):
                assert api.frames_submit_d(ctx, sync, d_iq, buf.size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, None, d_ix) == 1
                if check:
                    assert api.frames_d(ref, rsync, r_iq, buf.size, S, y_t, x_t, np.float32(0.1), True, r_state, r_fr, None, r_ix) == 1
            if check:
                ctx.synchronize()
                ref.synchronize()
                assert np.array_equal(ctx.download(d_fr, (npx,), np.uint32), ref.download(r_fr, (npx,), np.uint32))
                assert np.array_equal(ctx.download(d_ix, (2,), np.int32), ref.download(r_ix, (2,), np.int32))
        submit(256, y0, check=True)
        first = ctx.pipeline_info()
        assert first["trials_left"] == 0 and first["measurements_started"] == 1
        for k in range(1, 51):                 # y_t walks 50 lines down and up again
            y_t = y0 + (k if k <= 25 else 50 - k)
            submit(2, y_t, check=True)
            info = ctx.pipeline_info()
            assert info["trials_left"] == 0 and info["chosen"] == first["chosen"], (y_t, info)
            assert y_t == y0 or "taken over" in info["text"], info["text"]
        assert ctx.pipeline_info()["measurements_started"] == 1
        # pinned: arrangement k, no trials even for a configuration far from everything measured
        ctx.set_option("pipe_pin", 4)
        submit(3, y0 // 2, check=True)
        info = ctx.pipeline_info()
        assert info["chosen"] == 4 and info["trials_left"] == 0 and info["measurements_started"] == 1 and "pinned" in info["text"]
        ctx.set_option("pipe_pin", -1)
        # measure now: the current configuration is measured again
        submit(1, y0)
        ctx.set_option("pipe_measure", 1)
        submit(1, y0)
        assert ctx.pipeline_info()["trials_left"] > 0 and ctx.pipeline_info()["measurements_started"] == 2
        with pytest.raises(AssertionError):      # (TSDR_EINVAL: an ArgumentError in the reference's terms)
            ctx.set_option("pipe_pin", 8)
    finally:
        close(ctx, sync, d_state, d_iq, d_fr, d_ix)
        close(ref, rsync, r_state, r_iq, r_fr, r_ix)
    # two configurations in strict alternation from the first submission on
    ctx, sync, d_state, d_iq, d_fr, d_ix = session()
    try:
        for i in range(40):
            for y_t in (y0, y0 // 2):
                assert api.frames_submit_d(ctx, sync, d_iq, buf.size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, None, d_ix) == 1
        info = ctx.pipeline_info()
        assert info["trials_left"] == 0 and info["chosen"] == 0, info     # settled: the sequential order
        ctx.synchronize()
    finally:
        close(ctx, sync, d_state, d_iq, d_fr, d_ix)


def test_pipeline_pending_stage_is_drained_by_other_entry_points(ctx, tsdr, synth):
    """Submitted buffers run on the pipeline's internal streams.  Entry points that use the same SyncXY / IIR state
    outside the pipeline (tsdr_vsync_d here, tsdr_frames_d) and tsdr_sync_free must order themselves behind them first: the
    results equal the strictly sequential ones and nothing runs on freed memory."""
    from tempestsdr_jl_amd import api
    Fs, x_t, y_t, fv, nfr = 2.0e6, 1056, 628, 60.0, 3
    S = synth.samples_per_frame(Fs, fv)
    npx = 600 * 800
    buf = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr)
    img = np.asfortranarray(rng.random((600, 800)).astype(np.float32))

    def run(pipelined):
        sync = tsdr.SyncXY(ctx, 600, 800)
        d_state = ctx.upload(np.zeros(npx, np.float32))
        d_iq = ctx.upload(buf.view(np.float32))
        d_fr, d_ix = ctx.dev_alloc(nfr * npx * 4), ctx.dev_alloc(nfr * 8)
        try:
            f = api.frames_submit_d if pipelined else api.frames_d
            f(ctx, sync, d_iq, buf.size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, None, d_ix)
            s_yx = sync.vsync(img)          # consumes the pending s_y of the submitted buffer: must come after its stage
            ctx.synchronize()
            out = (ctx.download(d_fr, (nfr * npx,), np.uint32), ctx.download(d_ix, (nfr * 2,), np.int32),
                   ctx.download(d_state, (npx,), np.uint32), tuple(int(v) for v in s_yx))
            f(ctx, sync, d_iq, buf.size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, None, d_ix)
            sync.close()                    # frees the state with submitted work possibly still in flight
            ctx.synchronize()
            return out
        finally:
            for p in (d_state, d_iq, d_fr, d_ix):
                ctx.dev_free(p)

    a, b = run(False), run(True)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert a[3] == b[3]


def test_sync_guard_is_what_makes_indices_identical(ctx, tsdr, synth):
    """The plateau leak (constant blanking level: neighbouring blank-band centres tie to 1e-6 and below in the oracle's
    own beta) through the FAST loop with the guard on -- identical indices on all 66 frames, no escape -- and with it
    off, where the run may take the tie escape; the guard's counters say how many frames it re-evaluated.  With the
    threshold raised to 1 (every frame flagged) FAST returns EXACT's frames bit for bit."""
    Fs, x_t, y_t, fv, nfr = 2.0e6, 1056, 628, 60.0, 66
    S = synth.samples_per_frame(Fs, fv)
    iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 11, card="plateau")
    on = fast_vs_oracle(ctx, tsdr, O, iq, S, y_t, x_t, 0.1, True, RTOL, guard=True)
    off = fast_vs_oracle(ctx, tsdr, O, iq, S, y_t, x_t, 0.1, True, RTOL, guard=False)
    assert on["guard"][0] == nfr and not on["ties"]
    print(f"guard on: {on['guard'][1]} of {nfr} frames re-evaluated; guard off: {len(off['ties'])} tie escapes")
    # every frame flagged -> the whole buffer is the exact sequence
    gs = [np.zeros((600, 800), np.float32, order="F") for _ in range(2)]
    ctx.set_option("sync_guard_ppb", 100000000)
    try:
        a = ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq[:S * 5], S, y_t, x_t, np.float32(0.1), gs[0])
        assert ctx.sync_guard_stats(reset=True)[1] >= 5
    finally:
        ctx.set_option("sync_guard_ppb", 20000)
    ctx.set_precision("exact")
    try:
        b = ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq[:S * 5], S, y_t, x_t, np.float32(0.1), gs[1])
    finally:
        ctx.set_precision("fast")
    assert np.array_equal(a["sync_idx"], b["sync_idx"])
    for fa, fb in zip(a["frames"], b["frames"]):
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
    assert np.array_equal(gs[0].view(np.uint32), gs[1].view(np.uint32))


@pytest.mark.parametrize("ppb", [100000000, 5000000])
@pytest.mark.parametrize("wl,nfr", [("C2", 7), ("C3", 3), ("C5", 3)])
def test_sync_guard_queue_under_full_load(ctx, tsdr, synth, wl, nfr, ppb):
    """Every frame flagged at the BASELINE geometries -- threshold 1: both axes, the full path (image tiles -> row blocks ->
    centre blocks); threshold 5e-3: the x axis only (this leak's y margins are 1e-2), the short path where the image tiles
    leave the exact column sums themselves and the row-sum pass is skipped.  The guard kernel's ticket queue then carries
    thousands of dependent work items (C2: 7 x (250 image tiles + 10 row blocks + 23 centre blocks)) across all CUs, and
    the result must be TSDR_EXACT's, bit for bit -- frames, sync indices, IIR state -- twice in a row on the same context
    (the queue words are back at zero after a launch)."""
    w = synth.WORKLOADS[wl]
    Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
    S = synth.samples_per_frame(Fs, fv)
    iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 13)
    ctx.set_precision("exact")
    try:
        se = np.zeros((600, 800), np.float32, order="F")
        e = ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), se)
    finally:
        ctx.set_precision("fast")
    ctx.set_option("sync_guard_ppb", ppb)
    try:
        for _ in range(2):
            ctx.sync_guard_stats(reset=True)
            sf = np.zeros((600, 800), np.float32, order="F")
            g = ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), sf)
            assert ctx.sync_guard_stats() == (nfr, nfr)
            assert np.array_equal(g["sync_idx"], e["sync_idx"])
            for a, b in zip(g["frames"], e["frames"]):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert np.array_equal(sf.view(np.uint32), se.view(np.uint32))
    finally:
        ctx.set_option("sync_guard_ppb", 20000)


def test_sync_guard_many_frames_in_chunks(ctx, tsdr, synth):
    """More frames per buffer than one guard launch lists (256): 300 frames of a small raster with the plateau leak, whose
    flagged frames fall into both launches; indices identical to the oracle on all of them."""
    Fs, x_t, y_t, fv, nfr = 1.0e6, 300, 130, 50.0, 300
    S = synth.samples_per_frame(Fs, fv)
    iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 5, card="plateau")
    r = fast_vs_oracle(ctx, tsdr, O, iq, S, y_t, x_t, 0.1, False, RTOL)
    assert r["n_frames"] == nfr and not r["ties"] and r["guard"][0] == nfr
    print("sync guard (checked, re-evaluated):", r["guard"])


@pytest.mark.parametrize("wl,nfr", [("C2", 4), ("C5", 2)])
def test_raster_free_routes_agree(ctx, tsdr, synth, wl, nfr):
    """Without a raster the FAST loop has two routes: the tap-based kernel with its own projection partial sums (default where
    its 64-column tile fits: C2, C5) and the raster walk with out == NULL (rounds 1-2; option "fast_walk_only").  Both against
    the oracle: identical sync indices, frames and IIR state within RTOL; and the two routes' own sync indices identical."""
    w = synth.WORKLOADS[wl]
    Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
    S = synth.samples_per_frame(Fs, fv)
    iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 9)
    res = {}
    for walk in (0, 1):
        ctx.set_option("fast_walk_only", walk)
        try:
            res[walk] = fast_vs_oracle(ctx, tsdr, O, iq, S, y_t, x_t, 0.1, False, RTOL)
        finally:
            ctx.set_option("fast_walk_only", 0)
        assert res[walk]["n_frames"] == nfr and not res[walk]["ties"], (walk, res[walk]["ties"])
    print("worst relative frame error: tap-based", res[0]["worst"], "walk", res[1]["worst"])


def test_sync_guard_adaptive_route(ctx, tsdr, synth):
    """At 1080p60 / 20 MS/s the plateau leak flags ~70 % of its frames: re-evaluating them one by one costs more than the
    exact sequence for everything, so after a window of 60 frames the FAST loop runs whole buffers exactly (bit-identical to
    TSDR_EXACT), keeps counting, and returns to the fast route once a leak with a defined sync answer (the box profile: < 1 %
    flagged) has filled a window.  The decision for call k folds the counts of calls <= k - 3, in submission order (a fixed
    lag, so that the route is a function of the buffers and not of host / GPU timing): the routes below are exact
    predictions.  Indices equal TSDR_EXACT's on every buffer on either route; "sync_guard_auto" = 0 pins the one-by-one route."""
    w = synth.WORKLOADS["C2"]
    Fs, x_t, y_t, fv, nfr = w["Fs"], w["x_t"], w["y_t"], w["fv"], 30
    S = synth.samples_per_frame(Fs, fv)
    tie = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 3, card="plateau")
    box = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 3, card="box")

    def run(iq, precision):
        ctx.set_precision(precision)
        try:
            st = np.zeros((600, 800), np.float32, order="F")
            r = ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), st)
        finally:
            ctx.set_precision("fast")
        return r, st

    e_tie, se_tie = run(tie, "exact")
    e_box, _ = run(box, "exact")
    ctx.set_option("sync_guard_ppb", 20000)  # (also restarts the adaptive route's window)
    base = ctx.sync_guard_auto()
    assert base[0] is False
    ctx.sync_guard_stats(reset=True)
    routes = []
    for i in range(6):
        g, sg = run(tie, "fast")
        routes.append(ctx.sync_guard_auto()[1] - base[1])
        assert np.array_equal(g["sync_idx"], e_tie["sync_idx"]), i
        if routes[-1] > (routes[-2] if i else 0):  # this call ran as a whole exact buffer
            for a, b in zip(g["frames"], e_tie["frames"]):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert np.array_equal(sg.view(np.uint32), se_tie.view(np.uint32))
    checked, flagged = ctx.sync_guard_stats()
    assert checked == 6 * nfr and flagged > 0.5 * checked, (checked, flagged)
    # calls 0 and 1 fill the window; call 4 is the first whose decision sees both (4 - 3 = 1): it and call 5 run exactly
    assert routes == [0, 0, 0, 0, 1, 2], routes
    assert ctx.sync_guard_auto()[0] is True
    for i in range(5):
        g, _ = run(box, "fast")
        assert np.array_equal(g["sync_idx"], e_box["sync_idx"]), i
    now, n_exact, switches = ctx.sync_guard_auto()
    # calls 6 .. 9 (box) still run exactly -- their decisions fold the tie calls 3 .. 5 and the box calls 6, 7, which complete
    # a quiet window at call 10: the fifth box buffer is back on the fast route
    assert now is False and n_exact - base[1] == 6 and switches - base[2] == 2, (now, n_exact, switches)
    # pinned to the one-by-one route
    ctx.set_option("sync_guard_auto", 0)
    try:
        for i in range(3):
            g, _ = run(tie, "fast")
            assert np.array_equal(g["sync_idx"], e_tie["sync_idx"])
        assert ctx.sync_guard_auto()[:2] == (False, n_exact)
    finally:
        ctx.set_option("sync_guard_auto", 1)
        ctx.set_option("sync_guard_ppb", 20000)


def test_adaptive_route_is_reproducible_run_to_run(tsdr, synth):
    """The same plateau buffers three times, each time through a fresh context and the pipelined entry point, "sync_guard_auto"
    on: WHICH buffers run as whole exact buffers is decided from per-call counts folded in submission order with a fixed lag
    (common.h: kGuardLag), so frames, indices and the route taken are bit-identical on every run -- not a function of how
    far the host happened to be ahead of the GPU."""
    from tempestsdr_jl_amd import api
    Fs, x_t, y_t, fv, nfr, nbuf = 2.0e6, 1056, 628, 60.0, 20, 8
    S = synth.samples_per_frame(Fs, fv)
    npx = 600 * 800
    bufs = [synth.synth_leak(Fs, x_t, y_t, fv, S * nfr, n0=b * S * nfr, card="plateau") for b in range(nbuf)]
    runs = []
    for rep in range(3):
        ctx = tsdr.Context(0)
        try:
            ctx.set_option("pipe_mode", rep % 3)    # (the arrangement changes the timing, not the route)
            sync = tsdr.SyncXY(ctx, 600, 800)
            d_state = ctx.upload(np.zeros(npx, np.float32))
            d_iq = [ctx.upload(b.view(np.float32)) for b in bufs]
            d_fr = [ctx.dev_alloc(nfr * npx * 4) for _ in bufs]
            d_ix = [ctx.dev_alloc(nfr * 8) for _ in bufs]
            route = []
            for b in range(nbuf):
                if rep == 1 and b % 3 == 2:
                    ctx.synchronize()               # a host that sometimes waits, sometimes runs ahead
                api.frames_submit_d(ctx, sync, d_iq[b], bufs[b].size, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr[b], None, d_ix[b])
                route.append(ctx.sync_guard_auto()[1])
            ctx.synchronize()
            runs.append((route, [ctx.download(p, (nfr * npx,), np.uint32) for p in d_fr],
                         [ctx.download(p, (nfr * 2,), np.int32) for p in d_ix], ctx.sync_guard_stats()))
            sync.close()
        finally:
            ctx.close()
    print("whole exact buffers after each submission:", runs[0][0], "guard counters", runs[0][3])
    assert runs[0][0][-1] > 0, "the plateau leak should have switched the route"
    for r in runs[1:]:
        assert r[0] == runs[0][0] and r[3] == runs[0][3]
        for x, y in zip(r[1] + r[2], runs[0][1] + runs[0][2]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_frames_fast_random_geometries_raster_free(ctx, tsdr, seed):
    """The same random geometries and white-noise IQ WITHOUT rasters: the tap kernel (k_down_fused: fixed-point taps and f32
    blends up to 0.5 samples per raster pixel, f64 taps above) forms the images.  White noise is its worst input -- seven f32
    roundings on taps that differ by their own size -- and the bar here is 1e-6 (north_star: 1e-5); sync indices identical."""
    r = np.random.default_rng(seed)
    for _ in range(5):
        y_t, x_t, nfr = int(r.integers(610, 1300)), int(r.integers(820, 2800)), int(r.integers(1, 4))
        S = max(2, int(y_t * x_t * float(np.exp(r.uniform(np.log(0.08), np.log(1.6))))))
        iq = ((r.standard_normal(S * nfr + 3) + 1j * r.standard_normal(S * nfr + 3)) * 1e-3).astype(np.complex64)
        res = fast_vs_oracle(ctx, tsdr, O, iq, S, y_t, x_t, 0.1, False, RTOL_TAPS)
        assert res["n_frames"] == nfr, (S, y_t, x_t)
        print(f"{y_t}x{x_t} S={S} ({S / (y_t * x_t):.3f} samples per pixel): worst frame pixel {res['worst']:.3e}")


@pytest.mark.parametrize("seed", [11, 12])
def test_frames_fast_random_geometries(ctx, tsdr, seed):
    """Random raster sizes and sampling ratios (0.08 .. 1.6 samples per pixel), white-noise IQ -- the hardest input
    for the blends, neighbouring taps differing by their own size: sync indices identical, frames and rasters within
    the bar.  (tools/fuzz_raster.py runs the same loop over many more cases.)"""
    r = np.random.default_rng(seed)
    for _ in range(5):
        y_t, x_t, nfr = int(r.integers(130, 1300)), int(r.integers(260, 2800)), int(r.integers(1, 4))
        S = max(2, int(y_t * x_t * float(np.exp(r.uniform(np.log(0.08), np.log(1.6))))))
        iq = ((r.standard_normal(S * nfr + 3) + 1j * r.standard_normal(S * nfr + 3)) * 1e-3).astype(np.complex64)
        res = fast_vs_oracle(ctx, tsdr, O, iq, S, y_t, x_t, 0.1, True, RTOL)
        assert res["n_frames"] == nfr, (S, y_t, x_t)


@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("case", [
    dict(Fs=20e6, x_t=2576, y_t=1125, fv=60.0, nfr=3),    # C2: y_t mod 32 = 5
    dict(Fs=50e6, x_t=4400, y_t=2250, fv=60.0, nfr=2),    # C5: y_t mod 32 = 10
    dict(Fs=2.0e6, x_t=1056, y_t=628, fv=60.0, nfr=3),    # y_t mod 32 = 20, ragged tiles on both axes
    dict(Fs=3.0e6, x_t=1200, y_t=640, fv=50.0, nfr=2),    # y_t mod 32 = 0: columns already on the 128-byte grid
])
def test_frames_fast_sheared_raster_route(ctx, tsdr, synth, case, split):
    """Option "raster_split" (round 4's A/B of the raster-writing kernel; off by default because it loses to the one-launch
    walk: DESIGN.md section 4): the rasters by the store-aligned raster-only kernel (raster_shear.hip: every wave-store two
    full 128-byte lines; 2 = the same kernel unsheared), images and projection sums by the raster-free kernel.  Same bar as
    every FAST route: identical sync indices; rasters and frames within the tap kernels' 1e-6 of the oracle."""
    S = synth.samples_per_frame(case["Fs"], case["fv"])
    iq = synth.synth_leak(case["Fs"], case["x_t"], case["y_t"], case["fv"], S * case["nfr"] + 123)
    ctx.set_option("raster_split", split)
    try:
        r = fast_vs_oracle(ctx, tsdr, O, iq, S, case["y_t"], case["x_t"], 0.1, True, RTOL_TAPS)
    finally:
        ctx.set_option("raster_split", 0)
    assert r["n_frames"] == case["nfr"] and not r["ties"]


@pytest.mark.parametrize("case", [
    dict(Fs=20e6, x_t=2576, y_t=1125, fv=60.0, nfr=3),     # C2: f32 walk, 128-pixel tiles
    dict(Fs=200e6, x_t=2576, y_t=1125, fv=60.0, nfr=2),    # C3: 1.15 samples per raster pixel -- 64-pixel tiles only with f32 samples
    dict(Fs=50e6, x_t=4400, y_t=2250, fv=60.0, nfr=2),     # C5: 2P >= 2^24, the integer walk
])
def test_frames_fast_raster_sample_formats(ctx, tsdr, synth, case):
    """Option "raster_rec4": the raster walk on plain f32 samples (default since round 4; the pixel in the convex form with the
    two integer weights) and, with 0, on the {a, slope hi, slope lo} records of rounds 1-3 (kept as the A/B): both against the
    oracle -- identical sync indices, rasters and frames within the FAST tolerance."""
    S = synth.samples_per_frame(case["Fs"], case["fv"])
    iq = synth.synth_leak(case["Fs"], case["x_t"], case["y_t"], case["fv"], S * case["nfr"] + 77)
    for rec4 in (0, 1):
        ctx.set_option("raster_rec4", rec4)
        try:
            r = fast_vs_oracle(ctx, tsdr, O, iq, S, case["y_t"], case["x_t"], 0.1, True, RTOL)
        finally:
            ctx.set_option("raster_rec4", 1)
        assert r["n_frames"] == case["nfr"] and not r["ties"], rec4


@pytest.mark.parametrize("mode", [0, 1])
def test_frames_pipeline_without_alignment_and_mixed_with_one_call(ctx, tsdr, synth, mode):
    """tsdr_frames_submit_d with do_align = 0 (no statistics, no guard: image launch + IIR only), and submissions
    interleaved with plain tsdr_frames_d calls on the same IIR state: the context's stream and the pipeline's internal
    streams must hand the state over in call order."""
    from tempestsdr_jl_amd import api
    Fs, x_t, y_t, fv, nfr, nbuf = 2.0e6, 1056, 628, 60.0, 3, 6
    S = synth.samples_per_frame(Fs, fv)
    npx = 600 * 800
    bufs = [synth.synth_leak(Fs, x_t, y_t, fv, S * nfr, n0=b * S * nfr) for b in range(nbuf)]

    def run(kinds, do_align):
        sync = tsdr.SyncXY(ctx, 600, 800) if do_align else None
        d_state = ctx.upload(np.zeros(npx, np.float32))
        d_iq = [ctx.upload(b.view(np.float32)) for b in bufs]
        d_fr = [ctx.dev_alloc(nfr * npx * 4) for _ in bufs]
        d_ix = [ctx.dev_alloc(nfr * 8) for _ in bufs]
        try:
            for b, kind in enumerate(kinds):
                f = api.frames_submit_d if kind == "s" else api.frames_d
                assert f(ctx, sync, d_iq[b], bufs[b].size, S, y_t, x_t, np.float32(0.25), do_align, d_state, d_fr[b], None,
                         d_ix[b] if do_align else None) == nfr
            ctx.synchronize()
            return [ctx.download(p, (nfr * npx,), np.uint32) for p in d_fr], ctx.download(d_state, (npx,), np.uint32)
        finally:
            if sync is not None:
                sync.close()
            for p in [d_state] + d_iq + d_fr + d_ix:
                ctx.dev_free(p)

    ctx.set_option("pipe_mode", mode)
    try:
        for do_align in (False, True):
            want = run("dddddd", do_align)
            for kinds in ("ssssss", "sdsdds", "ddssss"):
                got = run(kinds, do_align)
                for x, y in zip(want[0], got[0]):
                    assert np.array_equal(x, y), (kinds, do_align)
                assert np.array_equal(want[1], got[1]), (kinds, do_align)
    finally:
        ctx.set_option("pipe_mode", -1)


@pytest.mark.parametrize("pw", [32, 16])
@pytest.mark.parametrize("fmt", ["cf32", "sc16"])
def test_raster_v4_option_gives_the_same_rasters_images_and_indices(tsdr, synth, pw, fmt):
    """Option "raster_v4" (round 6's A/B of the raster launch's store pattern: four raster lines per lane, 1024-byte wave-stores,
    image rows compacted through LDS; off by default because it measures slower): same arithmetic per pixel, so rasters, frames
    and frame-sync indices are those of the default kernel bit for bit -- C2 geometry (five 255-line tiles, the last one cut by
    the raster's end; 21 strips, the last one 36 pixels wide; the frame's first pixels before the first sample), two buffers."""
    from tempestsdr_jl_amd import api
    w = synth.WORKLOADS["C2"]
    Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
    S = synth.samples_per_frame(Fs, fv)
    nfr, npx, P = 4, 600 * 800, x_t * y_t
    z = synth.synth_leak(Fs, x_t, y_t, fv, 2 * nfr * S)
    if fmt == "sc16":
        scale = np.float32(float(np.max(np.abs(z.view(np.float32)))) / 2047.0)
        q = np.round(z.view(np.float32) / scale).astype(np.int16)
    res = {}
    for v4 in (0, pw):
        ctx = tsdr.Context(0)
        ctx.set_option("raster_v4", v4)
        sync = tsdr.SyncXY(ctx, 600, 800)
        d_state = ctx.upload(np.zeros(npx, np.float32))
        d_in = ctx.upload(q) if fmt == "sc16" else ctx.upload(z.view(np.float32))
        d_fr, d_ra, d_ix = ctx.dev_alloc(nfr * npx * 4), ctx.dev_alloc(nfr * P * 4), ctx.dev_alloc(nfr * 8)
        out = []
        try:
            for b in range(2):
                o_in = d_in + b * nfr * S * (4 if fmt == "sc16" else 8)
                if fmt == "sc16":
                    api.frames_sc16_d(ctx, sync, o_in, scale, nfr * S, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, d_ra, d_ix)
                else:
                    api.frames_d(ctx, sync, o_in, nfr * S, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr, d_ra, d_ix)
                ctx.synchronize()
                out.append((ctx.download(d_fr, (nfr * npx,), np.uint32), ctx.download(d_ra, (nfr * P,), np.uint32),
                            ctx.download(d_ix, (nfr * 2,), np.int32)))
            res[v4] = out
        finally:
            sync.close()
            for p in (d_state, d_in, d_fr, d_ra, d_ix):
                ctx.dev_free(p)
            ctx.close()
    for a, b in zip(res[0], res[pw]):
        assert np.array_equal(a[1], b[1]), "rasters differ"
        assert np.array_equal(a[2], b[2]), "sync indices differ"
        assert np.array_equal(a[0], b[0]), "frames differ"
