"""bench.py --gpus N started as a PLAIN python command (no torchrun): the script
must start its N ranks itself, as child processes, before anything touches the
GPU, relay rank 0's single JSON line and the ranks' exit code.

* without a GPU (this container): the ranks are started, form their process group
  and then fail LOUDLY at ba_create — the product has no CPU path — and the parent
  exits non-zero;
* on the GPU box: two ranks rehearse on the one card (BA_BENCH_BACKEND=gloo: RCCL
  refuses two ranks on one device; the exchange is staged through the host) and
  the line must name 2 ranks seen by the communicator; the strong-scaling run
  must reproduce the one-rank trajectory, the weak run doubles the landmarks.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, extra_env=None, timeout=900):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


def json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def _has_gpu():
    import torch
    return torch.cuda.device_count() > 0


def test_self_launch_without_gpu_fails_loudly(built):
    if _has_gpu():
        pytest.skip("a GPU is present: covered by the gpu-marked tests below")
    r = run_bench(["--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline"], {"BA_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0
    assert not json_lines(r.stdout)              # no result line without a GPU
    # both ranks were started and both refused to run without the HIP device
    assert r.stderr.count("no HIP device available (the HIP path has no CPU fallback)") >= 2
    assert "[rank0]" in r.stderr and "[rank1]" in r.stderr


@pytest.mark.gpu
def test_plain_python_gpus2_rehearsal_prints_one_json_line(built):
    common = ["--scale", "0.02", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
              "--no-roofline", "--config", "C3"]
    one = run_bench(["--gpus", "1"] + common)
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = json_lines(one.stdout)
    assert len(l1) == 1
    two = run_bench(["--gpus", "2"] + common, {"BA_BENCH_BACKEND": "gloo"})
    assert two.returncode == 0, two.stderr[-2000:]
    l2 = json_lines(two.stdout)
    assert len(l2) == 1, two.stdout
    a, b = l1[0], l2[0]
    assert b["n_gpus"] == 2 and b["n_ranks_seen"] == 2 and b["scaling"] == "strong"
    assert "REHEARSAL" in b["config"]["parallelism"]
    assert a["n_gpus"] == 1 and a["n_ranks_seen"] == 1
    assert b["config"]["n_observations"] == a["config"]["n_observations"]
    # the same problem, the same LM trajectory: sharding only permutes summation order
    assert abs(a["final_cost"] - b["final_cost"]) <= 1e-9 * abs(a["final_cost"])
    weak = run_bench(["--gpus", "2", "--weak"] + common, {"BA_BENCH_BACKEND": "gloo"})
    assert weak.returncode == 0, weak.stderr[-2000:]
    lw = json_lines(weak.stdout)
    assert len(lw) == 1
    w = lw[0]
    assert w["scaling"] == "weak" and w["n_ranks_seen"] == 2
    assert w["config"]["n_observations"] == 2 * a["config"]["n_observations"]
    assert w["config"]["n_opt_poses"] == a["config"]["n_opt_poses"]


@pytest.mark.gpu
def test_one_rank_rccl_exchange_path_of_the_bench(built):
    """What a one-GPU box can run of the multi-GPU bench with REAL RCCL: one rank,
    `nccl` process group, the library's own RCCL communicator (ba_rccl_*, hook called
    from C++), the sharded iteration (separate control kernel, non-direct
    k_schur_final + k_scatter) and the final landmark gather — same trajectory as the
    plain single-GPU run of the same problem."""
    common = ["--gpus", "1", "--scale", "0.05", "--config", "C3", "--steps", "6", "--warmup", "2",
              "--no-cpu-baseline", "--no-roofline"]
    plain = run_bench(common)
    assert plain.returncode == 0, plain.stderr[-2000:]
    forced = run_bench(common + ["--force-exchange"])
    assert forced.returncode == 0, forced.stderr[-3000:]
    a, b = json_lines(plain.stdout), json_lines(forced.stdout)
    assert len(a) == 1 and len(b) == 1
    a, b = a[0], b[0]
    assert "RCCL all-reduce issued by libba_hip.so" in b["config"]["exchange"]
    assert b["n_ranks_seen"] == 1 and "gather_points_ms" in b
    assert abs(a["final_cost"] - b["final_cost"]) <= 1e-9 * abs(a["final_cost"])
