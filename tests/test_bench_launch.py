"""bench.py's launcher: `--gpus N` without a torch.distributed launcher must become N ranks by itself, and a line that
says n_gpus: N must come from N ranks (it used to run one rank silently and print n_gpus: 1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_needs_spawn_only_without_a_launcher():
    a2, a1 = bench.parse(["--gpus", "2"]), bench.parse([])
    assert bench.needs_spawn(a2, {})
    assert not bench.needs_spawn(a2, {"WORLD_SIZE": "2"})  # the driver's torch.distributed.run: already a rank
    assert not bench.needs_spawn(a1, {})


def test_launch_command_is_one_rank_per_gpu_on_loopback():
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--config", "cfg4"]
    cmd = bench.launch_command(argv, 4, 29123)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29123"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv  # the ranks see the very same flags


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=240)


def test_gpus_2_without_env_runs_two_ranks():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist"]["world_size"] == 2


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_cfg4_is_a_known_config_with_both_models():
    from dp_timeautodiff_amd import workloads
    c = workloads.CONFIGS["cfg4"]
    assert c["parts"] == ("cfg4_vae", "cfg4_den") and c["B"] == 2048
    assert workloads.CONFIGS["cfg4_vae"]["B"] == workloads.CONFIGS["cfg4_den"]["B"] == 2048
    with pytest.raises(ValueError):
        workloads.build("cfg4", "cpu")
