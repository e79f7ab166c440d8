"""Multi-GPU (RCCL) execution of the two sharded paths, on machines that have more than one GPU.

The parent process never touches the GPU (torch.cuda.device_count() does not initialise it on this image): it launches
fresh one-process-per-GPU children through torch.distributed.run and checks their exit status.  On a 1-GPU box the
test skips; the same compositions run at world_size 2 over gloo on CPU in tests/test_host_logic.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ngpu():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_hipframes_and_hipsearch_over_rccl(world):
    n = _ngpu()
    if n < world:
        pytest.skip(f"needs {world} GPUs, this box has {n}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "mgpu_child.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"world {world} failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    assert "mgpu_child: OK" in r.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_paths_with_several_ranks_on_one_gpu(world):
    """The same child with every rank on cuda:0 and gloo (host-staged) collectives: scan / all-gather / combine and
    segment+halo partial sums / all-reduce / finish run through the product kernels at world size > 1 on any GPU box.
    What it cannot show is RCCL itself -- that is the test above."""
    if _ngpu() < 1:
        pytest.skip("needs a GPU")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGPU_SHARE_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "mgpu_child.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"world {world} (one GPU) failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    assert "mgpu_child: OK" in r.stdout


def test_bench_multi_rank_code_path_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it, except that both ranks share cuda:0 (gloo): the weak-scaling leg,
    the max-over-ranks timing, the strong leg and the search route choice must run to a well-formed JSON line."""
    import json
    if _ngpu() < 1:
        pytest.skip("needs a GPU")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TSDR_BENCH_SHARE_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--repeats", "2", "--search-steps", "2", "--no-cpu", "--no-ingest", "--no-extra"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"bench --gpus 2 failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and "shared_gpu_test_mode" in d
    assert d["strong"] and "error" not in d["strong"] and d["strong"]["value"] > 0, d["strong"]
    assert d["search"] and "error" not in d["search"], d["search"]
    assert "replicated" in d["search"]["mode"]  # the reference's window: sharding cannot pay


def test_bench_gpus_n_launched_bare_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` with no torch.distributed.run around it and no WORLD_SIZE in the environment: the
    parent must not touch the GPU, start the two ranks as a child process, relay rank 0's single JSON line and exit 0.
    Both ranks share cuda:0 over gloo here (TSDR_BENCH_SHARE_ONE_GPU=1); the line must carry the forced-sharded search
    leg with the size and the time of its all-reduce (SURVEY 8e / Autocorrelations.jl:27-29)."""
    import json
    if _ngpu() < 1:
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["TSDR_BENCH_SHARE_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--repeats", "2", "--search-steps", "2", "--no-cpu", "--no-ingest", "--no-extra"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"bare bench --gpus 2 failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    ss = d["search"]["search_sharded"]
    assert ss["all_reduce_bytes"] == 4 * d["search"]["lags"] and ss["ms_all_reduce"] > 0 and ss["ms_per_search"] > 0
    assert ss["same_argmax_as_route_above"] is True
    assert d["pipeline"] and d["pipeline"]["raster"]["value"] > 0 and d["pipeline"]["fused"]["value"] > 0
