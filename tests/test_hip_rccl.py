"""The RCCL (`nccl` backend) path of the data-parallel step, executed on the one GPU of the test box at world size 1
(north_star: "RCCL all-reduce of gradients ... overlapped with the backward pass"; the reference is
single-GPU, gaze_full.py:37).  Real model_SP + FusedAdam, streams on, 8 MB buckets:
  * gradients with the reducer == gradients without it, bit for bit (a 1-rank sum is the identity; a bucket issued before
    its last producer kernel finished would show up as a stale / partial gradient);
  * at least 4 buckets are ISSUED from hooks inside backward() and at least 3 collectives have COMPLETED on the device
    before the last backward kernel (HIP events on the comm stream vs an event behind backward);
  * two optimizer steps land on bit-identical parameters (Adam ordered after the handles' wait());
  * no AccumulateGrad stream-mismatch warning on stderr."""
import json
import os
import socket
import subprocess
import sys

import pytest

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
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("EGAZE_PRECISION", "EGAZE_STREAMS", "EGAZE_DIST_BACKEND", "EGAZE_SINGLE_DEVICE"):
        env.pop(k, None)
    return env


def test_rccl_world1_real_model(tmp_path):
    out = str(tmp_path / "obs.json")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_gpu_worker.py"), out, "96", "4", str(_free_port())]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    obs = json.load(open(out))
    assert obs["backend"] == "nccl" and obs["world"] == 1
    assert obs["n_buckets"] >= 5                                      # 186 MB, halving buckets down to 8 MB: 93 + 47 + 23 + 12 + 8 + ...
    assert obs["grad_absmax"] > 0 and obs["grad_bit_exact"], obs
    assert obs["launched_in_backward"] >= 4, obs                      # issued from the sinks' hooks, not from wait()
    assert obs["launched_in_backward"] + obs["launched_in_wait"] == obs["n_buckets"]
    assert obs["buckets_complete_before_backward_end"] >= 3, obs      # overlapped on the device, not just issued early
    assert obs["params_bit_exact"] and obs["params_moved"], obs
    assert obs["steps_joined"] == 3
    assert "AccumulateGrad" not in r.stderr, r.stderr[-2000:]


def test_bench_rccl_world1_leg():
    """`bench.py` at N = 1 with EGAZE_DP_FORCE=1: the timed step runs with the RCCL reducer attached (process group of
    one rank), and says so in its JSON line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "64",
           "--no-cpu-baseline", "--no-roofline"]
    env = _env()
    env["EGAZE_DP_FORCE"] = "1"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and "rccl" in (out["config"]["collective"] or "")
    assert "RCCL ranks: 1" in r.stderr
