"""bench.py as the driver calls it. `python bench.py --gpus N` (N > 1, no launcher on the command line) must start its
own N rank processes before touching the GPU, relay rank 0's JSON line and fail loudly when a rank fails.
The GPU test runs the 2-rank path on ONE device over gloo (NSVD_FORCE_DEVICE / NSVD_DIST_BACKEND: the GPU box has a
single GPU and RCCL refuses two ranks per device); the CPU test checks the failure propagation of the launcher."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(kw)
    return env


def test_launcher_propagates_a_failing_rank():
    """no GPU here: every rank dies in torch.cuda.set_device; the launcher must exit non-zero and say which rank"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       env=_env(NSVD_FORCE_DEVICE="0", NSVD_DIST_BACKEND="gloo"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "exited with code" in r.stderr and "rank" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launcher_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(max(n, 2)), "--steps", "3"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_bench_two_ranks_exactly_as_the_driver_calls_it():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=_env(NSVD_FORCE_DEVICE="0", NSVD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 1024
    assert d["value"] > 0 and d["params_finite"] and d["scaling"] == "weak"
    assert d["rccl_ranks"] == 2
    c = d["comm"]
    assert c["backend"] == "gloo" and c["exchange"] in ("allreduce", "rs_ag", "a2a")
    assert c["compute_only_ms"] > 0 and c["step_ms"] == d["ms_per_step"]
    w = c["exposed_wait_us_per_step"]
    assert "moments_allreduce" in w and any(k.startswith("grad_bucket0") for k in w)
    assert abs(sum(w.values()) - c["exposed_wait_us_total"]) < 0.1
    assert sum(c["grad_bucket_bytes"]) >= 4 * d["config"]["params"]
    # every exchange candidate was timed (or says why not) and the winner is the one the headline ran with
    cands = c["candidates"]
    assert all(any(k.startswith(ex + "/") for k in cands) for ex in ("allreduce", "rs_ag", "a2a"))
    ok = {k: v for k, v in cands.items() if "steps_per_s" in v}
    assert ok, cands
    best = max(ok, key=lambda k: ok[k]["steps_per_s"])
    assert best.startswith(c["exchange"] + "/")
    # the side lines of the same JSON: the head-sharded split and configs[2] under both shardings
    for k in ("sharding_hp", "cfg3_dp", "cfg3_hp"):
        assert k in d and "error" not in d[k] and d[k]["value"] > 0 and d[k]["params_finite"], (k, d.get(k))
    assert d["cfg3_dp"]["global_batch"] == 1024
