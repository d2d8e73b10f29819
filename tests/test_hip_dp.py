"""Data-parallel path of the REAL SP model on the GPU: 2 ranks share the one GPU of the test box and
all-reduce over gloo -- split-half kernels, HIP side streams, FusedAdam, dp.GradReducer with several buckets.
  (a) the reduced gradient equals the sum of the two single-process gradients, bit for bit (2-rank sum is commutative
      and every kernel is deterministic), so the bucket hooks fired after every producer stream had finished;
  (b) replicas hold bit-identical parameters after two Adam steps, and they moved;
  (c) `bench.py --gpus 2` launches itself and reports n_gpus == 2."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.update(EGAZE_SINGLE_DEVICE="1", EGAZE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("EGAZE_PRECISION", None)
    env.pop("EGAZE_STREAMS", None)
    return env


def test_two_ranks_one_gpu_real_model(tmp_path):
    prefix = str(tmp_path / "obs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_gpu_worker.py"), prefix,
           "64", "2"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    obs = [torch.load(f"{prefix}.{k}") for k in range(2)]
    g0, g1 = obs[0]["g_local"]
    assert not torch.equal(g0, g1)                                      # different minibatches per rank
    assert g0.abs().max() > 0 and g1.abs().max() > 0
    for o in obs:                                                       # (a)
        assert o["n_buckets"] >= 4 and o["grad_scale"] == 0.5
        assert torch.equal(o["g_sum"], g0 + g1), (o["g_sum"] - (g0 + g1)).abs().max()
    assert torch.equal(obs[0]["flat_p"], obs[1]["flat_p"])              # (b)
    assert all(l == l for o in obs for l in o["losses"])                # finite
    assert obs[0]["losses"] != obs[1]["losses"]                         # per-rank batches, per-rank losses


@pytest.mark.parametrize("kind,size,batch", [("at", 16, 32), ("lf", 224, 4)])
def test_two_ranks_one_gpu_at_and_lf(kind, size, batch, tmp_path):
    """The multi-GPU forms of BASELINE configs 4 and 5 (AT lstmnet at T = 16 / B = 32 per rank -- the persistent recurrence
    launches; the late-fusion stage at 224 x 224) through the same data-parallel machinery as the SP model: 2 ranks on the one GPU
    over gloo, gradient buckets reduced from the sink hooks.  (a) reduced gradient = the sum of the two local gradients bit for bit,
    (b) bit-identical replicas after two Adam steps, finite per-rank losses that differ."""
    prefix = str(tmp_path / "obs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_gpu_worker.py"), prefix,
           str(size), str(batch), kind]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    obs = [torch.load(f"{prefix}.{k}") for k in range(2)]
    g0, g1 = obs[0]["g_local"]
    assert not torch.equal(g0, g1) and g0.abs().max() > 0 and g1.abs().max() > 0
    for o in obs:
        assert o["n_buckets"] >= 2 and o["grad_scale"] == 0.5
        assert torch.equal(o["g_sum"], g0 + g1), (o["g_sum"] - (g0 + g1)).abs().max()
    assert torch.equal(obs[0]["flat_p"], obs[1]["flat_p"])
    assert all(l == l for o in obs for l in o["losses"]) and obs[0]["losses"] != obs[1]["losses"]


def test_bench_gpus2_self_launch():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--size", "64", "--no-cpu-baseline", "--no-roofline"]
    env = _env()
    env["EGAZE_DP_CHECK"] = "1"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert "ranks: 2" in r.stderr
    assert r.stdout.count("parameters identical to rank 0: True") == 2


def test_bench_world4_dry_run_every_leg(tmp_path):
    """`bench.py --gpus 4` as the driver launches it at N > 1, WITH every leg the default run takes (bare-MFMA
    probe on rank 0 only, per-kernel profiled step, the opt-in-arithmetic leg with the headline's protocol, the exact-f32 leg;
    the cpu_baseline / LF / PCIe / RCCL-world-1 legs are world-1 only) -- four ranks on the one GPU of the test box over gloo.
    Every rank has to reach the same collectives in the same order or this hangs (timeout) / dies: one JSON line on rank 0,
    n_gpus 4, identical replicas afterwards, both arithmetics timed, and a weak_scaling block computed from a saved N = 1 line."""
    env = _env()
    env["EGAZE_DP_CHECK"] = "1"
    env["TMPDIR"] = str(tmp_path)
    common = ["--steps", "2", "--warmup", "1", "--repeats", "2", "--batch", "2", "--size", "64", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-roofline"] + common, env=env,
                        capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-3000:]
    n1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert n1["n_gpus"] == 1 and n1["extra"]["weak_scaling"] is None
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4"] + common
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                              # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp4"
    assert out["scaling"] == "weak" and out["roofline"]["mfma_macs_per_algorithmic_mac"] == 3.0
    assert out["extra"]["bwd2"]["bwd_products"] == 2 and len(out["extra"]["bwd2"]["regions_ms_per_step"]) == 2
    assert out["extra"]["f32_ms_per_step"] > 0
    ws = out["extra"]["weak_scaling"]
    assert ws["n1_ms"] == n1["ms_per_step"] and ws["this_ms"] == out["ms_per_step"]
    assert abs(ws["efficiency"] - ws["n1_ms"] / ws["this_ms"]) < 1e-12
    assert r.stdout.count("parameters identical to rank 0: True") == 4
