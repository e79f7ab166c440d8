"""Host-side logic around the hot path (CPU only): mode table, .dat files, sharding, and the
multi-GPU compositions driven over gloo with world_size 2 (numpy / oracle stand-ins compute)."""
import importlib
import os
import warnings

import numpy as np
import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def vc(tsdr):
    return importlib.import_module("tempestsdr_jl_amd.video_configurations")


@pytest.fixture(scope="module")
def dat(tsdr):
    return importlib.import_module("tempestsdr_jl_amd.dat_files")


@pytest.fixture(scope="module")
def par(tsdr):
    return importlib.import_module("tempestsdr_jl_amd.parallel")


# ---------------------------------------------------------------- VideoConfigurations.jl
def test_mode_table(vc):
    t = vc.allVideoConfigurations
    assert len(t) == 80 and isinstance(t, dict)  # test/runtests.jl:33-37 analogue
    m = t["1920x1080 @ 60Hz"]
    assert (m.width, m.height, m.refresh) == (2576, 1125, 60.0)
    assert t["2048x1536 @ 60Hz"].width == 2800 and t["PAL TV"].refresh == 25.0


def test_find_closest_configuration(vc):
    # documented operating point: 60.14 Hz, y_t = 1235 -> "1920x1200 @ 60Hz" (docs/src/gui.md:29)
    d = vc.find_closest_configuration(1235, 60.14)
    assert list(d) == ["1920x1200 @ 60Hz"]
    assert list(vc.find_closest_configuration(1125, 60.05)) == ["1920x1080 @ 60Hz"]
    # nearest RATE first, then nearest height: 100.2 Hz picks among the 100 Hz modes only
    d = vc.find_closest_configuration(900, 100.2)
    assert all(vc.allVideoConfigurations[k].refresh == 100 for k in d)
    # several modes share a height at one rate -> all returned, with a warning (:104-106)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        d = vc.find_closest_configuration(1250, 60)
        assert len(d) >= 1
    # every table entry is found from its own height and rate (the loop test/runtests.jl:40-49 meant to assert)
    for name, m in vc.allVideoConfigurations.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert name in vc.find_closest_configuration(m.height, m.refresh)


def test_find_configuration_identity_semantics(vc):
    m = vc.allVideoConfigurations["1920x1200 @ 60Hz"]
    assert vc.find_configuration(m) == "1920x1200 @ 60Hz"
    assert vc.find_configuration(vc.VideoMode(2592, 1242, 60.0)) is None  # mutable struct without ==
    assert vc.delay2yt(1 / (60.0 * 1125), 60.0) == 1125 and vc.yt2index(1125, 20e6, 60.0) == 296


# ---------------------------------------------------------------- DatBinaryFiles.jl
def test_dat_roundtrip_like_reference_tests(dat, tmp_path):
    rng = np.random.default_rng(0)
    x32 = (rng.standard_normal(32) + 1j * rng.standard_normal(32)).astype(np.complex64)
    x64 = rng.standard_normal(32) + 1j * rng.standard_normal(32)
    p32, p64 = str(tmp_path / "test32.dat"), str(tmp_path / "test64.dat")
    dat.writeComplexBinary(x32, p32)
    dat.writeComplexBinary(x64, p64)  # default :single, as in test/runtests.jl:13
    assert os.path.getsize(p32) == 32 * 8 and os.path.getsize(p64) == 32 * 8
    assert np.allclose(dat.readComplexBinary(p32), x32)
    assert np.allclose(dat.readComplexBinary(p64), x64, rtol=1e-6)
    # raw layout is interleaved I,Q float32 (GNU Radio file sink)
    raw = np.fromfile(p32, np.float32)
    assert np.array_equal(raw[0::2], x32.real) and np.array_equal(raw[1::2], x32.imag)
    pd = str(tmp_path / "d.dat")
    dat.writeComplexBinary(x64, pd, "double")
    assert np.array_equal(dat.readComplexBinary(pd, "double"), x64)
    ps = str(tmp_path / "s.dat")
    dat.writeComplexBinary(x64, ps, "short")
    z = dat.readComplexBinary(ps, "short")
    assert z.real.max() == 1 << 14 and z.imag.max() == 1 << 14  # per-component max-normalised (:17-20)
    assert dat.readComplexBinary(p32, "single", 10).size == 5
    with pytest.raises(ValueError):
        dat.readComplexBinary(p32, "int8")


# ---------------------------------------------------------------- sharding
def test_shard_range(par):
    for n in (0, 1, 7, 30, 4_000_000):
        for world in (1, 2, 3, 8):
            parts = [par.shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (a, ca), (b, _) in zip(parts, parts[1:]):
                assert a + ca == b
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def _np_partial(x, m0, cnt, n_lags):
    """sum_{m in [m0,m0+cnt)} x[m] x[(m+k) mod n], f64 -- numpy stand-in for tsdr_autocorr_partial_d"""
    n = x.size
    xd = x.astype(np.float64)
    ext = np.concatenate([xd, xd])
    seg = xd[m0:m0 + cnt]
    return np.array([np.dot(seg, ext[m0 + k:m0 + k + cnt]) for k in range(n_lags)])


def _worker_autocorr(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tempest_loader import load_package
    load_package()
    par = importlib.import_module("tempestsdr_jl_amd.parallel")
    rng = np.random.default_rng(5)
    n, n_lags, Fs = 3000, 1500, 30_000.0
    x = (rng.random(n) ** 2).astype(np.float32)

    def partial(m0, cnt):
        return torch.from_numpy(_np_partial(x, m0, cnt, n_lags))

    def all_reduce(buf):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    def finish(buf):
        r = buf.numpy()
        return 10 * np.log10(r * r)

    out = par.autocorr_sharded(partial, all_reduce, finish, n, n_lags, world, rank)
    ref, _ = O.calculate_autocorrelation(x, Fs, 0, n_lags / Fs)
    q.put((rank, float(np.max(np.abs(out - ref))), int(np.argmax(out[100:])), int(np.argmax(ref[100:]))))
    dist.destroy_process_group()


def _worker_frames(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tempest_loader import load_package
    load_package()
    par = importlib.import_module("tempestsdr_jl_amd.parallel")
    synth = importlib.import_module("tempestsdr_jl_amd.synth")
    Fs, x_t, y_t, fv, nfr = 1.0e6, 160, 125, 50.0, 5
    S = synth.samples_per_frame(Fs, fv)
    iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 99)
    alpha = np.float32(0.1)

    def scan(f0, cnt):  # oracle stand-in for tsdr_frames_scan_d: per-frame image + (s_x, next s_y)
        imgs, keys = [], []
        for f in range(f0, f0 + cnt):
            img = O.downgradeImage(O.sig_to_image(O.amDemod(iq[f * S:(f + 1) * S]), y_t, x_t))
            s = O.SyncXY(600, 800)
            s.vsync(img)
            sy_next, sx = s.vsync(img)  # second call on a fresh state exposes argmax(beta_y(img))
            imgs.append(img)
            keys.append((sx, sy_next))
        return imgs, keys

    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    def combine(imgs, keys):  # stand-in for tsdr_frames_combine_d
        acc = np.zeros((600, 800), np.float32)
        sy, idx = 1, []
        for img, (sx, sy_next) in zip(imgs, keys):
            idx.append((sy, sx))
            sh = O.circshift_neg(img, sy, sx)
            acc = (alpha * acc + (np.float32(1) - alpha) * sh).astype(np.float32)
            sy = sy_next
        return acc, idx

    acc, idx = par.frames_sharded(scan, all_gather, combine, nfr, world, rank)
    st = np.zeros((600, 800), np.float32, order="F")
    ref = O.frames(O.SyncXY(600, 800), iq, S, y_t, x_t, alpha, st)
    ok = bool(np.array_equal(acc, st)) and [tuple(r) for r in ref["sync_idx"].tolist()] == idx
    q.put((rank, ok))
    dist.destroy_process_group()


def _worker_welch(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tempest_loader import load_package
    load_package()
    par = importlib.import_module("tempestsdr_jl_amd.parallel")
    rng = np.random.default_rng(9)
    size, nb = 64, 37   # 37 segments: ragged over the ranks
    z = (rng.standard_normal(size * nb + 5) + 1j * rng.standard_normal(size * nb + 5)).astype(np.complex64)

    def partial(s0, cnt):  # numpy stand-in for tsdr_welch_d(lin = 1) on this rank's segments
        seg = z[s0 * size:(s0 + cnt) * size].reshape(cnt, size).astype(np.complex128)
        return torch.from_numpy(np.fft.fftshift(np.sum(np.abs(np.fft.fft(seg, axis=1)) ** 2, axis=0)))

    def all_reduce(buf):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    out = par.welch_sharded(partial, all_reduce, lambda b: 10 * np.log10(b.numpy()), nb, world, rank)
    ref = O.getWelch(z, size)
    q.put((rank, float(np.max(np.abs(out - ref)))))
    dist.destroy_process_group()


def _spawn(fn, world=2):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res)


def test_sharded_autocorr_gloo_world2():
    res = _spawn(_worker_autocorr)
    for rank, err_db, am, am_ref in res:
        assert err_db < 1e-3 and am == am_ref, (rank, err_db, am, am_ref)


def test_sharded_welch_gloo_world2():
    """getWelch of one capture with its segments sharded: partial sums, ONE all-reduce of sizeFFT floats, 10log10 after"""
    for rank, err_db in _spawn(_worker_welch):
        assert err_db < 1e-4, (rank, err_db)


def test_sharded_frames_gloo_world2():
    res = _spawn(_worker_frames)
    assert all(ok for _, ok in res), res


# ---------------------------------------------------------------- configuration search (host picks)
def test_search_flow_on_oracle(tsdr):
    """search.py's peak picks (GUI.jl:56-81, :491-506, investigate_data.jl:92) driven by the ORACLE's
    autocorrelation: the host logic is backend-agnostic, so this covers it without a GPU."""
    search = importlib.import_module("tempestsdr_jl_amd.search")
    synth = importlib.import_module("tempestsdr_jl_amd.synth")
    vcm = importlib.import_module("tempestsdr_jl_amd.video_configurations")

    class OracleCtx:
        abs2 = staticmethod(O.abs2)
        calculate_autocorrelation = staticmethod(O.calculate_autocorrelation)
        zoom_autocorr = staticmethod(O.zoom_autocorr)

    mode = vcm.allVideoConfigurations["800x600 @ 60Hz"]  # 1056 x 628 total
    Fs = 2.0e6
    iq = synth.synth_leak(Fs, mode.width, mode.height, mode.refresh, int(0.2 * Fs))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = search.search(OracleCtx, iq, Fs)
    assert abs(got["fv"] - 60.0) < 0.2
    assert abs(got["y_t"] - 628) <= 0.03 * 628
    assert vcm.allVideoConfigurations[got["name"]].refresh == 60.0
    with pytest.raises(IndexError):  # capture shorter than the 0.1 s window
        search.extract_configuration(OracleCtx, iq[:1000], Fs)


# ---- runtime-loop extras: drop-oldest frame channel, record task --------------------------------
def test_frame_channel_drops_oldest():
    """GUI.jl:111-118: when the channel is full the oldest image is removed before the new one goes in."""
    import importlib
    from tempest_loader import load_package
    load_package()
    ing = importlib.import_module("tempestsdr_jl_amd.ingest")
    ch = ing.FrameChannel(3)
    for k in range(5):
        ch.put(k)
    assert len(ch) == 3 and ch.dropped == 2
    assert [ch.take(0.1) for _ in range(3)] == [2, 3, 4]
    with pytest.raises(IndexError):
        ch.take(0.01)


def test_record_buffers_round_trip(tmp_path):
    """GUI.jl:181-190: nbBuffer recv! results concatenated and written as a .dat; read back identical (:single)."""
    import importlib
    from tempest_loader import load_package
    load_package()
    ing = importlib.import_module("tempestsdr_jl_amd.ingest")
    dat = importlib.import_module("tempestsdr_jl_amd.dat_files")
    rng = np.random.default_rng(3)
    nEch, nb = 1000, 4
    bufs = [(rng.standard_normal(nEch) + 1j * rng.standard_normal(nEch)).astype(np.complex64) for _ in range(nb)]
    it = iter(bufs)
    path = str(tmp_path / "dumpIQ_0.dat")
    assert ing.record_buffers(lambda: next(it), nb, nEch, path) == nb * nEch
    back = dat.readComplexBinary(path, "single")
    assert np.array_equal(np.asarray(back, np.complex64), np.concatenate(bufs))


def test_mode_table_order_matters_only_for_three_known_pairs():
    """find_closest_configuration returns every entry at the nearest height; the caller takes the first.  Which entry is
    first depends on the Dict's iteration order (Julia: hash order; here: table order) only when two modes share
    (height, refresh) -- exactly the three pairs video_configurations.AMBIGUOUS_HEIGHT_REFRESH lists; for every other
    (y_t, fv) in and around the table the sub-dict has ONE entry whatever the order, and a tie between two refresh
    rates needs fv exactly half-way between them."""
    import warnings
    from collections import defaultdict
    from tempestsdr_jl_amd import video_configurations as vc
    groups = defaultdict(list)
    for name, m in vc.allVideoConfigurations.items():
        groups[(m.height, m.refresh)].append((name, m.width))
    shared = sorted(k for k, v in groups.items() if len(v) > 1)
    assert shared == sorted(vc.AMBIGUOUS_HEIGHT_REFRESH)
    assert all(len({w for _, w in groups[k]}) == 2 for k in shared)   # and there the widths differ: the pick matters
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, m in vc.allVideoConfigurations.items():
            for dy in (-0.4, 0.0, 0.4):
                sub = vc.find_closest_configuration(m.height + dy, m.refresh)
                assert name in sub
                assert (len(sub) > 1) == ((m.height, m.refresh) in vc.AMBIGUOUS_HEIGHT_REFRESH), (name, sub)


def test_search_route_choice():
    """The sharded search is used only when a rank's segment+halo transform is smaller than the single-GPU one; with
    the reference's window (n = 2 * n_lags) the halo puts the same floor under both and every rank runs the
    single-GPU route (no collective)."""
    from tempestsdr_jl_amd import parallel as par
    assert par.single_route_points(4_000_000) == 2_000_000          # C2: native 2^7 5^6 transform
    assert par.single_route_points(4_000_002) == 4_194_304          # not 5-smooth: zero-padded 2^22
    for world in (2, 4, 8):
        assert par.search_route(4_000_000, 2_000_000, world) == "replicated"
        assert par.search_route(40_000_000, 20_000_000, world) == "replicated"
    assert par.search_route(4_000_000, 2_000_000, 1) == "single"
    # a window much longer than the lag range does shard
    assert par.search_route(16_000_000, 500_000, 8) == "sharded"
    assert par.sharded_route_points(16_000_000, 500_000, 8) == 4_194_304
    # the model follows the route HipSearch.run actually takes: a window longer than 2 * n_lags runs the partial-sum
    # cross-correlation over the whole range on one GPU (pow2(n + n_lags - 1) points), so sharding it does pay
    assert par.single_route_points(240_000, 60_000) == 524_288
    assert par.sharded_route_points(240_000, 60_000, 2) == 262_144
    for world in (2, 4, 8):
        assert par.search_route(240_000, 60_000, world) == "sharded"


def test_bench_group_child_failure_is_reported_not_raised():
    """bench.py's several-devices leg runs tools/group_devices.py in a child process: a child that cannot create its contexts
    (here: no GPU in the container) or that overruns its time must come back as an {"error": ...} entry of the JSON line."""
    import importlib.util
    import torch
    if torch.cuda.is_available():
        pytest.skip("the failure path needs a box without a GPU")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.run_group_child("0,1", "C2", timeout=240)
    assert r["devices"] == "0,1" and "error" in r and "exit code" in r["error"]
    r = bench.run_group_child("0,1", "C2", timeout=0.2)
    assert "error" in r and "no result within" in r["error"]


def test_bench_child_leg_that_fails_becomes_an_error_entry_not_an_exception():
    """bench.py runs every leg beside the headline in a child process (round 6).  Without a GPU the child cannot even create its
    context: the parent's run_child must turn that into an {"error": ...} entry that names the exit code and the last stage the
    child reached -- never raise, never hang -- and the line's packing (finish_line) must keep error / skip texts in place."""
    import argparse
    import importlib.util
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the child would succeed")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = argparse.Namespace(steps=2, warmup=1, repeats=2, workload="C2", precision="fast", card="box", search_steps=1, no_raster=False)
    r = bench.run_child("search", args, None, timeout=120)
    assert isinstance(r, dict) and "error" in r and "exit code" in r["error"] and "last stage" in r["error"], r
    line = {"value": 1.0, "ms_per_step": 2.0, "roofline": {"kernel": "k", "frac": 0.5}, "search": r,
            "c3": {"skipped": "the run's time budget (280 s) was used up before this leg " + "x" * 120},
            "config": {"workload": "w" * 150}, "legs": {"failed_or_skipped": ["search", "c3"], "wall_s": 1.0}}
    out = bench.finish_line(line)
    assert out["search"]["error"] == r["error"] and out["c3"]["skipped"].startswith("the run's time budget")
    assert out["summary"]["legs_failed_or_skipped"] == ["search", "c3"] and out["config"]["workload"] == "w" * 150
    assert set(bench.LEGS) == set(bench.LINE_KEY) and "group" == list(bench.LEGS)[-1]     # (the never-run N > 1 path last)
