"""GPU test: bench.py with the DRIVER'S exact arguments prints its one JSON line -- and keeps printing it when a leg hangs.

Round 5's line was lost: `python3 bench.py --gpus 1 --steps 20 --warmup 5` ran 1800 s on the driver's box and printed nothing
(VERDICT r5).  Since round 6 the headline legs run first, every other leg runs in a time-boxed child process, and a watchdog
prints the line with what has been measured if the parent itself stops.  (File name: sorts last -- a full bench run.)"""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}


def _one_line(r):
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), f"stdout must be ONE JSON line:\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    return json.loads(lines[0])


def test_the_driver_command_prints_one_complete_line():
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                       env=CLEAN, cwd=ROOT, capture_output=True, text=True, timeout=600)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r)
    assert wall < 300, f"the default bench took {wall:.0f} s"
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["dtype"] == "f32" and d["value"] > 0
    assert d["config"]["workload"].startswith("C2:") and "10000000 IQ/buffer" in d["config"]["workload"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "raster_down_iq" and 0.2 < rf["frac"] < 1.0 and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0) < 2e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert d["index_parity"]["sync_idx_equal_exact"] == "90/90"
    assert d["legs"]["failed_or_skipped"] == [], d["legs"]
    for leg in ("fused", "pipeline", "two_streams", "search", "group", "host_ingest", "spectra", "exact", "sc16_resident", "c5", "c3"):
        assert isinstance(d[leg], dict) and "error" not in d[leg], (leg, d[leg])
    assert "[bench]" in r.stderr and "leg c3" in r.stderr      # stage lines: a tail of stderr names the last leg reached


def test_a_leg_that_hangs_costs_only_itself():
    env = dict(CLEAN, TSDR_BENCH_TEST_HANG="search", TSDR_BENCH_LEG_LIMIT="20")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--repeats", "3",
                        "--cpu-buffers", "1", "--legs", "search,fused"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r)
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    assert "error" in d["search"] and "no result within 20 s" in d["search"]["error"] and "last stage" in d["search"]["error"]
    assert d["fused"]["value"] > 0 and d["legs"]["failed_or_skipped"] == ["search"]


def test_a_parent_that_hangs_still_prints_what_it_measured():
    env = dict(CLEAN, TSDR_BENCH_TEST_HANG="parent")
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--repeats", "3",
                        "--no-cpu", "--legs", "none", "--budget", "25"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert time.perf_counter() - t0 < 90
    d = _one_line(r)
    assert r.returncode == 3 and "watchdog" in d["incomplete"] and "last stage reached" in d["incomplete"]
    assert d["value"] > 0 and d["roofline"]["frac"] > 0        # the headline had been measured: it is in the line


def test_a_parent_that_fails_still_prints_what_it_measured():
    env = dict(CLEAN, TSDR_BENCH_TEST_HANG="raise")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--repeats", "3",
                        "--no-cpu", "--legs", "none"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
    d = _one_line(r)
    assert r.returncode == 1 and "RuntimeError" in d["incomplete"] and d["value"] > 0 and d["roofline"]["frac"] > 0
