"""GPU tests of the pinned-host staging ring: AtomicCircularBuffer semantics on the consumer side
(AtomicAbstractSDRs.jl:64-190) with the buffer landing on the device."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(5)


def _buf(n, k):
    z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    z[0] = k  # tag
    return z


def test_ring_fifo_and_data_integrity(ctx, tsdr):
    n, depth = 100_003, 4
    ring = tsdr.StagingRing(ctx, n, depth)
    bufs = [_buf(n, k) for k in range(3)]
    for b in bufs:
        ring.put(b)
    for k in range(3):
        d = ring.take_d(1000)
        got = ctx.download(d, (n,), np.complex64)
        assert np.array_equal(got.view(np.uint32), bufs[k].view(np.uint32)), k
    with pytest.raises(IndexError):  # nothing left: circ_take! would block, the timeout turns that into an error
        ring.take_d(50)
    st = ring.stats()
    assert (st["produced"], st["consumed"], st["overflow"]) == (3, 3, 0)
    ring.close()


def test_ring_overflow_overwrites_like_the_reference(ctx, tsdr):
    """depth 3, five puts before the first take: t_new saturates at depth (:125-129); ptr_write has wrapped, so the
    slots hold buffers 3, 4, 2 and ptr_read still points at slot 0 -> the consumer sees 3, 4, 2 (the reference's
    order under overflow), and two overflows are counted."""
    n, depth = 1000, 3
    ring = tsdr.StagingRing(ctx, n, depth)
    bufs = [_buf(n, k) for k in range(5)]
    for b in bufs:
        ring.put(b)
    order = [int(ctx.download(ring.take_d(1000), (n,), np.complex64)[0].real) for _ in range(3)]
    assert order == [3, 4, 2], order
    assert ring.stats()["overflow"] == 2
    with pytest.raises(IndexError):
        ring.take_d(20)
    ring.close()


def test_ring_prefetch_survives_overflow(ctx, tsdr):
    """After the ring has dropped more than `depth` buffers the consumer must go back to finding its next buffer
    already sent ahead (the H2D of buffer k+1 under the kernels of buffer k).  The validity test of a prefetch is
    per slot (was the slot rewritten since the DMA was issued?), not a lifetime counter that never recovers."""
    n, depth = 4096, 3
    ring = tsdr.StagingRing(ctx, n, depth)
    for k in range(9):                      # six overflows before the first take
        ring.put(_buf(n, k))
    assert ring.stats()["overflow"] == 6
    seen = [int(ctx.download(ring.take_d(1000), (n,), np.complex64)[0].real) for _ in range(3)]
    assert seen == [6, 7, 8], seen          # slots hold 6,7,8 and ptr_read sits on slot 0
    for k in range(9, 15):                  # steady state: one in, one out, never full again
        ring.put(_buf(n, k))
        ring.put(_buf(n, 100 + k))
        a = int(ctx.download(ring.take_d(1000), (n,), np.complex64)[0].real)
        b = int(ctx.download(ring.take_d(1000), (n,), np.complex64)[0].real)
        assert (a, b) == (k, 100 + k)
    st = ring.stats()
    # of the 15 takes: the first of each burst is staged at take time; every second one was sent ahead
    assert st["prefetch_hits"] >= 8, st
    # a slot rewritten after its prefetch must NOT be served from the stale device copy
    ring.put(_buf(n, 200)); ring.put(_buf(n, 201))
    assert int(ctx.download(ring.take_d(1000), (n,), np.complex64)[0].real) == 200   # prefetches slot of 201
    ring.put(_buf(n, 202)); ring.put(_buf(n, 203)); ring.put(_buf(n, 204))           # laps: slot of 201 now holds 204
    got = [int(ctx.download(ring.take_d(1000), (n,), np.complex64)[0].real) for _ in range(3)]
    assert got == [204, 202, 203], got
    ring.close()


def test_ring_sc16_expansion(ctx, tsdr):
    n = 50_000
    scale = 1.0 / 2048.0
    ring = tsdr.StagingRing(ctx, n, 4, fmt="sc16", scale=scale)
    iq = rng.integers(-2048, 2048, size=2 * n, dtype=np.int16)
    v = ring.write_view()          # zero-copy producer path
    v[:] = iq
    ring.commit()
    got = ctx.download(ring.take_d(1000), (2 * n,), np.float32)
    want = iq.astype(np.float32) * np.float32(scale)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ring.close()


def test_ring_threaded_stream_matches_direct_frames(ctx, tsdr, synth):
    """A producer thread streams six buffers through the ring while the consumer runs tsdr_frames_d on each device
    buffer as it arrives (the next transfer overlapping those kernels): same frames, indices and state as feeding
    the same buffers directly."""
    from tempestsdr_jl_amd import api
    Fs, x_t, y_t, fv, nfr, nbuf = 2.0e6, 1056, 628, 60.0, 3, 6
    S = synth.samples_per_frame(Fs, fv)
    nEch = S * nfr + 5
    npx = 600 * 800
    bufs = [synth.synth_leak(Fs, x_t, y_t, fv, nEch, n0=b * nEch) for b in range(nbuf)]

    def run(use_ring):
        sync = tsdr.SyncXY(ctx, 600, 800)
        d_state = ctx.upload(np.zeros(npx, np.float32))
        d_fr = [ctx.dev_alloc(nfr * npx * 4) for _ in bufs]
        d_ix = [ctx.dev_alloc(nfr * 8) for _ in bufs]
        d_direct = []
        ring = tsdr.StagingRing(ctx, nEch, depth=8) if use_ring else None
        try:
            if use_ring:
                th = threading.Thread(target=lambda: [ring.put(b) for b in bufs])
                th.start()
            for b in range(nbuf):
                if use_ring:
                    d_iq = ring.take_d(5000)
                else:
                    d_iq = ctx.upload(bufs[b].view(np.float32))
                    d_direct.append(d_iq)
                assert api.frames_d(ctx, sync, d_iq, nEch, S, y_t, x_t, np.float32(0.1), True, d_state, d_fr[b], None, d_ix[b]) == nfr
            if use_ring:
                th.join()
            ctx.synchronize()
            return ([ctx.download(p, (nfr * npx,), np.uint32) for p in d_fr], [ctx.download(p, (nfr * 2,), np.int32) for p in d_ix],
                    ctx.download(d_state, (npx,), np.uint32))
        finally:
            if ring is not None:
                ring.close()
            for p in [d_state] + d_fr + d_ix + d_direct:
                ctx.dev_free(p)

    a, b = run(False), run(True)
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[2], b[2])
