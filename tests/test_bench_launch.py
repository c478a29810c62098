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


def test_degraded_run_keeps_its_line_and_exits_non_zero():
    """every attempt of the launcher's ladder dies after its first measured split (tests/_bench_degraded_rank.py): the
    rank that holds the provisional line prints it marked `degraded` and exits with bench.DEGRADED_RC; the launcher
    tries its whole ladder, then relays that ONE line (of the first attempt, with the failed attempts listed) and
    exits with the same non-zero code - a failed run never reports rc 0, and its measurement is not thrown away"""
    wrapped = os.path.join(ROOT, "tests", "_bench_degraded_rank.py")
    # as a rank under a launcher (torchrun's view): the line, then exit code 3
    r = subprocess.run([sys.executable, wrapped, "--gpus", "2"], env=_env(RANK="0", WORLD_SIZE="2"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    d = json.loads(r.stdout.strip())
    assert d["value"] == 123.0 and "died after its first measured split" in d["degraded"]
    r = subprocess.run([sys.executable, wrapped, "--gpus", "2"], env=_env(RANK="1", WORLD_SIZE="2"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode not in (0, 3) and not r.stdout.strip()  # no provisional line on this rank: the plain traceback
    # as the driver calls it: python bench.py --gpus 2
    r = subprocess.run([sys.executable, wrapped, "--gpus", "2", "--steps", "3"],
                       env=_env(NSVD_FORCE_DEVICE="0", NSVD_DIST_BACKEND="nccl"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] == 123.0 and "degraded" in d
    lr = d["launcher_retry"]
    assert len(lr["failed_attempts"]) == 3 and lr["this_line"].startswith("as given")
    assert "--dp-exchange" not in d["argv"]  # the first attempt's line, not a retry's
    assert all("exited with code" in a["failed_with"] for a in lr["failed_attempts"])


def test_launcher_worst_case_wall_time_is_inside_the_drivers_window():
    """`python bench.py --gpus N` from the argument DEFAULTS: the launcher's ladder (auto-tuned run, then the plain
    all-reduce, then heads sharded over gloo) is bounded by --launch-timeout in total, whatever the tuner
    (--tune-seconds) or a hung collective (--collective-timeout) do inside an attempt - under 1500 s, inside the
    driver's 1800 s; and the tuner's own cap leaves room for the headline run behind it."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    src = open(BENCH).read()
    lt = float(re.search(r'"--launch-timeout", type=float, default=([0-9.]+)', src).group(1))
    tune = float(re.search(r'"--tune-seconds", type=float, default=([0-9.]+)', src).group(1))
    coll = float(re.search(r'"--collective-timeout", type=float, default=([0-9.]+)', src).group(1))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for n_attempts in (1, 2, 3):
        assert mod.launcher_worst_case_seconds(lt, n_attempts) < 1500.0
    # inside the first attempt (60 % of the window at worst): both shardings tuned to their cap + one hung collective
    assert 2 * tune + coll < 0.6 * lt


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
    par = d["config"]["parallelism"]
    assert par in ("dp2", "hp2") and d["config"]["global_batch"] == 1024
    tuned = d["config"]["parallelism_tuned"]  # --parallelism auto: both shardings timed, the faster is the headline
    assert set(tuned) == {"dp2", "hp2"} and par == max(tuned, key=tuned.get)
    assert d["value"] > 0 and d["params_finite"] and d["scaling"] == "weak"
    assert d["rccl_ranks"] == 2 and "launcher_retry" not in d
    # what `value` is at N > 1, spelled out, with the two unambiguous rates beside it
    assert d["global_batch"] == 1024 and "per-GPU batches" in d["value_is"]
    assert abs(d["optimizer_steps_per_s"] * 2 - d["value"]) < 1e-2 * d["value"]
    assert abs(d["samples_per_s"] - d["optimizer_steps_per_s"] * 1024) < 1e-2 * d["samples_per_s"]
    # north_star's literal split (dp, one moments all-reduce + one gradient all-reduce, blocking) is always on the line
    ns = d["north_star_split"]
    assert ns["parallelism"] == "dp2" and ns["value"] > 0 and ns["params_finite"] and "--sync" in ns["flags"]
    assert ns["global_batch"] == 1024 and abs(ns["optimizer_steps_per_s"] * 2 - ns["value"]) < 1e-2 * ns["value"]
    c = d["comm"]
    assert c["backend"] == "gloo"
    assert c["compute_only_ms"] > 0 and c["step_ms"] == d["ms_per_step"]
    w = c["exposed_wait_us_per_step"]
    assert abs(sum(w.values()) - c["exposed_wait_us_total"]) < 0.1
    # every candidate of both shardings was timed (or says why not) and each winner is the one that ran
    other = "hp" if par == "dp2" else "dp"
    side = d[f"sharding_{other}"]
    dp_c, hp_c = (c["candidates"], side["candidates"]) if par == "dp2" else (side["candidates"], c["candidates"])
    assert all(any(k.startswith(ex + "/") for k in dp_c) for ex in ("allreduce", "rs_ag", "a2a"))
    ok = {k: v for k, v in dp_c.items() if isinstance(v, dict) and "steps_per_s" in v}
    assert ok and any(k.endswith("/blocking") for k in ok) and any(k.endswith("/1_bucket") for k in ok)
    assert dp_c["chosen"] == max(ok, key=lambda k: ok[k]["steps_per_s"])
    assert set(hp_c) == {"all_gather/async", "all_gather/blocking", "chosen"}
    if par == "dp2":
        assert c["exchange"] in ("allreduce", "rs_ag", "a2a") and dp_c["chosen"].startswith(c["exchange"] + "/")
        assert "moments_allreduce" in w and any(k.startswith("grad_bucket0") for k in w)
        assert sum(c["grad_bucket_bytes"]) >= 4 * d["config"]["params"]
        assert c["collectives"].startswith("blocking" if dp_c["chosen"].endswith("/blocking") else "asynchronous")
    else:
        assert list(w) == ["f_Tf_all_gather_wait"]
        assert c["collectives"].startswith("blocking" if hp_c["chosen"].endswith("/blocking") else "asynchronous")
    # the side lines of the same JSON: the other sharding and configs[2] under both
    for k in (f"sharding_{other}", "cfg3_dp", "cfg3_hp"):
        assert k in d and "error" not in d[k] and d[k]["value"] > 0 and d[k]["params_finite"], (k, d.get(k))
    assert d["cfg3_dp"]["global_batch"] == 1024
    # `metric` itself says what the N > 1 value is; BASELINE configs[2] at its stated GLOBAL batch rides on every N > 1 line
    assert "weak scaling: value = n_gpus x optimiser steps/s of the global batch of 1024 rows" in d["metric"]
    ss = d["strong_scaling_configs2"]
    assert ss["global_batch"] == 4096 and ss["n_gpus"] == 2 and ss["scaling"] == "strong"
    assert ss["dp"]["optimizer_steps_per_s"] > 0 and ss["hp"]["optimizer_steps_per_s"] > 0 and ss["dp"]["params_finite"]
    assert ss["optimizer_steps_per_s"] == max(ss["dp"]["optimizer_steps_per_s"], ss["hp"]["optimizer_steps_per_s"])
    assert "not_measured_in_this_run" not in json.dumps(d)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_launcher_retries_with_the_plain_exchange_when_the_tuned_run_dies():
    """the auto-tuned attempt is made to die on one rank (tests/_bench_failing_rank.py wraps bench.py: the launcher
    starts its ranks as the script it was started as): the launcher must start the ranks again with the plainest
    sequence and still deliver ONE line, marked with what failed"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_bench_failing_rank.py"), "--gpus", "2", "--steps",
                        "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=_env(NSVD_FORCE_DEVICE="0", NSVD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    lr = d["launcher_retry"]
    assert len(lr["failed_attempts"]) == 1 and "rank 1 exited with code 3" in lr["failed_attempts"][0]["failed_with"]
    assert "--dp-exchange allreduce" in lr["this_line"]
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["value"] > 0 and d["params_finite"]
    c = d["comm"]
    assert c["exchange"] == "allreduce" and c["collectives"].startswith("blocking") and c["candidates"] is None
    assert len(c["grad_bucket_bytes"]) == 1 and "sharding_hp" not in d


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_as_a_rank_under_torch_distributed_run():
    """the other way a driver may start it: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`
    (RANK / WORLD_SIZE from the launcher: bench.py is then a rank, not a launcher) - one JSON line from rank 0"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", BENCH, "--gpus", "2", "--steps", "20",
                        "--warmup", "5", "--no-cpu-baseline", "--no-extras"],
                       env=_env(NSVD_FORCE_DEVICE="0", NSVD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines, r.stdout[-2000:]  # nothing else on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] in ("dp2", "hp2") and d["value"] > 0 and d["params_finite"]
    assert d["rccl_ranks"] == 2 and "launcher_retry" not in d


def test_claimed_stdout_carries_only_the_emitted_line():
    """bench._claim_stdout / _emit on the CPU: after the claim, whatever anything writes to file descriptor 1 - C stdio
    of a native library included - lands on stderr; the emitted JSON line is the whole of stdout."""
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; "
            "bench._claim_stdout(); os.write(1, b'banner of a native library\\n'); print('python chatter'); "
            "bench._emit({'metric': 'x', 'value': 1})") % os.path.dirname(BENCH)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == json.dumps({"metric": "x", "value": 1}) + "\n", r.stdout
    assert "banner of a native library" in r.stderr and "python chatter" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_stdout_carries_the_json_line_only_with_rccl_loaded():
    """RCCL prints a version banner to stdout through C stdio when its first communicator comes up (seen on the GPU
    box: five lines, behind the JSON line in a pipe). A computing bench.py process points file descriptor 1 at stderr and
    writes its line through a private duplicate of the original stdout: a driver that parses stdout finds one line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-exchange", "--steps", "20", "--warmup", "2",
                        "--accuracy", "off", "--no-extras", "--no-cpu-baseline"],
                       env=_env(HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-4000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and out[0].startswith("{"), r.stdout[-2000:]
    assert json.loads(out[0])["n_gpus"] == 1


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
def test_bench_widened_rows_on_two_ranks(cfg):
    """`bench.py --config cfg4 | cfg5 --gpus 2`: the kernel-operator step with heads sharded (weak scaling) and the CDK
    step with the towers' hidden width sharded (strong scaling), self-launched, one line with its `comm` block"""
    r = subprocess.run([sys.executable, BENCH, "--config", cfg, "--gpus", "2", "--steps", "10", "--warmup", "2",
                        "--collective-timeout", "120"],
                       env=_env(NSVD_FORCE_DEVICE="0", NSVD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["final_loss"] == d["final_loss"]
    assert d["scaling"] == ("weak" if cfg == "cfg4" else "strong")
    assert d["config"]["parallelism"] == ("hp2" if cfg == "cfg4" else "tp2")
    c = d["comm"]
    assert c["rccl_ranks"] == 2 and c["compute_only_ms"] > 0 and len(c["exposed_wait_us_per_step"]) == (1 if cfg == "cfg4" else 2)
