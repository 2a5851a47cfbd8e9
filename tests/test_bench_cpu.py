"""bench.py's launcher contract, the part that needs no GPU: `--gpus N` without a launcher must never turn into a silent 1-GPU run
(round 4: `python bench.py --gpus 8` printed `n_gpus: 1`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, drop=()):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NIDX_BENCH_SAME_DEVICE") + tuple(drop):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_more_ranks_than_gpus_is_an_error_not_a_one_gpu_line():
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("this box has the GPUs the command asks for")
    p = run_bench(["--gpus", "2", "--n-vectors", "1000", "--steps", "1"])
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert "--gpus 2" in p.stderr and "GPU(s) visible" in p.stderr
    assert p.stdout.strip() == ""   # no JSON line of a smaller job


def test_cpu_quota_helper_reads_cgroup_v2(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib

    bench = importlib.import_module("bench")
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return real_open(tmp_path / "cpu.max", *a, **k)
        return real_open(path, *a, **k)

    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    monkeypatch.setattr("builtins.open", fake_open)
    assert bench.cpu_quota_cores() == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench.cpu_quota_cores() is None
