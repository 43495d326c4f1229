"""bench.py's launch contract without a GPU: `--gpus N` (N > 1) re-executes itself under torch.distributed.run, one rank per
"GPU", and prints ONE JSON line from rank 0; with no device visible the ranks run the gloo / NumPy dry run of the
scatter -> solve -> gather plumbing (reported as such: value null, dry_run true) — a check of the contract, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

from cvxopt_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(_capi.device_count() > 0, reason="with a GPU the same command is the real multi-GPU benchmark")
def test_gpus_2_self_spawns_two_ranks_and_prints_one_json_line():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0
    assert d["dry_run"] is True and d["value"] is None           # never a number without a GPU
    assert d["scaling"] == "strong" and d["config"]["problems_per_rank"] == [3, 3] and d["config"]["all_optimal"] is True
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert key in d
