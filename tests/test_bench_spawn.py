"""`python bench.py --gpus N` typed without a launcher must start its N ranks as child processes BEFORE the parent makes any
GPU call (a process that has initialised the GPU must never be re-exec'ed; the parent here only waits)."""
import importlib.util
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_parent_spawns_before_any_gpu_call(monkeypatch):
    bench = _load_bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 0

    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)

    def forbidden(*a, **k):
        raise AssertionError("the parent touched torch.cuda before spawning its ranks")

    for name in ("is_available", "set_device", "current_device", "synchronize", "init", "device_count", "current_stream"):
        monkeypatch.setattr(torch.cuda, name, forbidden)
    monkeypatch.setattr(bench, "load_pkg", forbidden)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert e.value.code == 0
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(os.path.join(REPO, "bench.py")) + 1:] == ["--gpus", "2", "--steps", "5", "--warmup", "1"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_single_gpu_and_launched_ranks_do_not_spawn(monkeypatch):
    bench = _load_bench()

    def no_spawn(*a, **k):
        raise AssertionError("spawned although a launcher already set WORLD_SIZE")

    monkeypatch.setattr(bench, "spawn_ranks", no_spawn)
    monkeypatch.setenv("WORLD_SIZE", "2")

    class Stop(Exception):
        pass

    def stop(*a, **k):
        raise Stop()

    monkeypatch.setattr(bench, "load_pkg", stop)          # first thing main() does after the spawn decision
    with pytest.raises(Stop):
        bench.main(["--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE")
    with pytest.raises(Stop):
        bench.main(["--gpus", "1"])
