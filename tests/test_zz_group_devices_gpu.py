"""GPU tests of tsdr_group_* on SEVERAL DISTINCT devices of one process -- RCCL over xGMI inside the library (ncclCommInitAll,
one ncclAllReduce of the autocorrelation accumulators, ncclSend / ncclRecv gather of the frames): skipped on boxes with fewer
GPUs.  No multi-GPU box was available while this was written, so these are the first runs of that path wherever they run:
each group runs in a CHILD process with a time limit (tools/group_devices.py, the program bench.py's `group.several_devices`
leg uses), so that a communicator that never comes back fails one test instead of stalling the suite, and the file sorts last.
What is asserted: frames bit for bit the single-context result (two precisions, two successive buffers), the sharded search
within 2e-4 dB with the same argmax, route "auto" = the root alone and bit-identical, getWelch within 2e-4 dB."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ndev():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("n", [2, 4, 8])
def test_groups_of_several_devices_equal_single_context(n):
    if _ndev() < n:
        pytest.skip(f"needs {n} GPUs in one process (this box has {_ndev()})")
    devs = ",".join(str(d) for d in range(n))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "group_devices.py"), devs, "C2"], capture_output=True, text=True,
                       timeout=240, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, f"rc {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
    out = json.loads(lines[-1])
    assert out["rccl"] is True and out["devices"] == list(range(n))
    assert out["frames_bit_identical_to_single_context"] is True
    s = out["search"]
    assert s["sharded"]["route_taken"] == "sharded" and s["sharded"]["same_argmax_as_single_context"]
    assert s["sharded"]["max_abs_dB_diff_vs_single_context"] < 2e-4
    assert s["root"]["route_taken"] == "root" and s["root"]["max_abs_dB_diff_vs_single_context"] == 0.0
    assert out["welch_max_abs_dB_diff_vs_single_context"] < 2e-4
