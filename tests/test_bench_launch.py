"""
bench.py's launcher logic on the CPU: a plain `python bench.py --gpus N` (N > 1, no torch.distributed.run around it) must
start its own ranks as a CHILD process before anything touches the GPU, relay rank 0's single JSON line and leave
with the child's exit code; `--gpus 1` and a run under torch.distributed.run must never spawn.
"""
import importlib
import io
import json
import os
import sys

import pytest

from tests.conftest import REPO


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(REPO)
    sys.modules.pop("bench", None)
    return importlib.import_module("bench")


class FakeChild:
    def __init__(self, cmd, env, lines, rc):
        self.cmd, self.env, self.stdout, self.rc = cmd, env, io.StringIO("".join(lines)), rc

    def wait(self, timeout=None):
        return self.rc

    def terminate(self):
        pass

    def kill(self):
        pass


def test_plain_multi_gpu_invocation_spawns_torchrun_child(bench, monkeypatch, capsys):
    import subprocess

    started = []

    def fake_popen(cmd, env=None, **kwargs):
        child = FakeChild(cmd, env, ["some rank chatter\n", json.dumps({"n_gpus": 4, "value": 1.0}) + "\n"], 0)
        started.append(child)
        return child

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(name, raising=False)
    torch_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as exit_info:
        bench.main(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert exit_info.value.code == 0
    assert ("torch" in sys.modules) == torch_before  # the parent never imports torch, let alone touches the GPU
    (child,) = started
    cmd = child.cmd
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]  # the ranks see the same arguments
    assert child.env["MASTER_ADDR"] == "127.0.0.1" and child.env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.count("\n") == 1 and json.loads(out.out)["n_gpus"] == 4  # ONE line on stdout
    assert "some rank chatter" in out.err


def test_child_failure_is_the_parents_exit_code(bench, monkeypatch):
    import subprocess

    monkeypatch.setattr(subprocess, "Popen", lambda cmd, env=None, **kw: FakeChild(cmd, env, [], 7))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as exit_info:
        bench.main(["--gpus", "2"])
    assert exit_info.value.code == 7


def test_no_spawn_under_a_launcher_or_with_one_gpu(bench, monkeypatch):
    import subprocess

    def no_popen(*a, **k):
        raise AssertionError("must not spawn")

    monkeypatch.setattr(subprocess, "Popen", no_popen)
    # under torch.distributed.run with a mismatching world size: refused, not re-launched
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as exit_info:
        bench.main(["--gpus", "4", "--no-cpu-baseline"])
    assert "does not match WORLD_SIZE" in str(exit_info.value.code)


def test_host_cpu_share_and_description(bench):
    cores, why = bench.host_cpu_share()
    assert 1 <= len(cores) <= 16 and len(set(cores)) == len(cores) and why
    assert set(cores) <= set(os.sched_getaffinity(0))
    info = bench.host_description()
    assert "numpy" in info and "cpu_model" in info
