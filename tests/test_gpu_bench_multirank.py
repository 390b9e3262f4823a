"""bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), exercised on the ONE GPU
a test box has: DANA_BENCH_BACKEND=gloo puts both ranks on cuda:0 and exchanges over gloo, so the whole multi-rank
control flow runs -- process-group start-up, the launch-mode vote across ranks, barriers, max-over-ranks timing, the
training iteration with its bucketed gradient all-reduce (between graph replays / from the launch program's host
callbacks), rank-0 JSON -- before the first real
multi-GPU run (where the same code talks RCCL). train.py:104-105,138-139 is what the N > 1 path stands for."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _finite(x):
    if isinstance(x, dict):
        return all(_finite(v) for v in x.values())
    if isinstance(x, (list, tuple)):
        return all(_finite(v) for v in x)
    return not isinstance(x, float) or math.isfinite(x)


def test_bench_two_ranks_prints_one_line(dev):
    env = dict(os.environ, DANA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29733", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-pmc",
           "--no-secondary"]
    pr = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert pr.returncode == 0, pr.stderr.decode()[-3000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and j["collective_backend"] == "gloo"
    assert j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert j["config"]["global_batch"] == 8 and j["config"]["workload"].startswith("BASELINE configs[2]")
    assert j["value"] > 0 and j["ms_per_step"] > 0 and _finite(j)
    pr_ = j["ms_per_step_per_rank"]
    assert len(pr_["all"]) == 2 and pr_["min"] <= pr_["max"] and abs(pr_["max"] - j["ms_per_step"]) < 1e-6
    ts = j["train_step"]
    assert ts["value"] > 0 and ts["ms_per_step"] > 0 and "gloo, 2 ranks" in ts["what"] and ts["buckets"] >= 2
    assert j["roofline"]["frac"] > 0 and "roofline" in ts  # rank 0's per-launch passes, forward and training iteration
    # every rank times the three launch modes; the trial times are max-reduced over the ranks (N ranks share the host) and
    # all ranks take the fastest: the timed mode is the argmin of what the line reports
    lt = j["launch_trial"]
    assert set(lt) >= {"eager_ms_per_step", "graph_ms_per_step", "program_ms_per_step"}
    best = min((v, k.split("_")[0]) for k, v in lt.items() if k.endswith("_ms_per_step"))[1]
    assert {"graph": "hipGraph", "program": "launch-program", "eager": "eager"}[best] in j["launch"]
    assert set(j["host_enqueue_ms_per_step"]) == {"eager", "graph", "program"}
    assert ts["launch"] in ("eager", "hipGraph replay", "launch-program replay")


def test_bench_gpus_flag_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` with NO launcher around it (the shape of the driver's N = 1 command with another N):
    bench.py re-executes itself under torch.distributed.run and rank 0 prints ONE line with n_gpus 2."""
    env = dict(os.environ, DANA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-pmc",
           "--no-secondary", "--no-train-step", "--no-cpu-baseline"]
    pr = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert pr.returncode == 0, pr.stderr.decode()[-3000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and j["config"]["global_batch"] == 8
    assert j["value"] > 0 and _finite(j)
