"""
The host runtime of libdlagp_hip.so under ThreadSanitizer and AddressSanitizer + UBSan, on the CPU.

Round 3's advisor found real thread bugs in the host side of the library (per-thread stream / arena pools, deferred
set-up events, spare workspaces, batch membership), and the catalogue driver runs five threads over it; GPU sanitizers
do not exist on the pool.  So the WHOLE host side of csrc/dlagp_hip.hip is compiled with g++ over a fake device
(csrc/fake_device.h: host memory, streams as ticket counters, launches that mark the buffers they touch) and driven through
the C ABI by eight threads in the catalogue driver's call pattern (csrc/runtime_replay.cpp): contexts created with
deferred set-up on one thread, batched and swept on another, destroyed on a third in both orders, one deferred context
first-used by two batches at once, the two-phase call finished / cancelled, device-pointer calls left in flight at
destroy, allocation failures injected into whichever call is running, the pools flushed under the threads' feet.

Pass = exit code 0 (no unexpected status, no buffer recycled while a launch that touches it is in flight, nothing
leaked) and no sanitizer report.  Nothing numerical is checked here (the fake kernels do nothing): that is the GPU
tests' job.  ("host-side ASan/UBSan build of the shim".)
"""
import json
import os
import subprocess

import pytest

from tests.conftest import REPO

CSRC = os.path.join(REPO, "gpy_dla_detection_amd", "csrc")
REPORTS = ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer")


@pytest.fixture(scope="module")
def replay_binaries():
    res = subprocess.run(["make", "-C", CSRC, "runtime_replay_tsan", "runtime_replay_asan"], capture_output=True, text=True)
    if res.returncode != 0:
        if "cannot find" in res.stderr and ("tsan" in res.stderr or "asan" in res.stderr):
            pytest.skip("no sanitizer runtimes beside g++")
        raise AssertionError(res.stderr[-3000:])
    return {kind: os.path.join(CSRC, "runtime_replay_" + kind) for kind in ("tsan", "asan")}


def run(binary, *args):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run([binary, *args], capture_output=True, text=True, timeout=240, env=env)
    text = res.stdout + res.stderr
    for marker in REPORTS:
        assert marker not in text, text[-4000:]
    return res, text


@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_the_checker_has_teeth(replay_binaries, kind):
    """a buffer recycled under a launch in flight and a double free ARE reported by the fake device"""
    res, text = run(replay_binaries[kind], "--selftest")
    assert res.returncode == 0, text[-2000:]
    verdict = json.loads(res.stdout.strip().splitlines()[-1])
    assert verdict == {"selftest": True, "in_flight_recycle_flagged": True, "idle_after_sync": True, "double_free_flagged": True,
                       "stream_wait_covers_event": True, "stream_wait_stops_at_event": True}


@pytest.mark.parametrize("kind,seed", [("tsan", 1), ("tsan", 2), ("asan", 1), ("asan", 2)])
def test_catalogue_call_pattern_over_the_fake_device(replay_binaries, kind, seed):
    """odd seeds run with the one shared kernel stream per device (the default), even seeds with per-thread streams"""
    res, text = run(replay_binaries[kind], "250", str(seed))
    assert res.returncode == 0, text[-4000:]
    summary = json.loads(res.stdout.strip().splitlines()[-1])
    assert summary["unexpected_errors"] == 0 and summary["fake_device_errors"] == 0 and summary["leaked_allocations"] == 0
    assert summary["device_mallocs"] == summary["device_frees"]
    assert summary["calls"] > 2000 and summary["launches"] > 5000
    assert summary["injected_failures"] > 0  # the failure paths ran
