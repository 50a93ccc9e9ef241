"""
The launch planner of the MFMA path (csrc/plan_host.h) under AddressSanitizer + UBSan on the CPU.

The planner decides every index the accumulate kernel dereferences: the slot order of a sightline's kept pixels, the
passes (windows of 64 entries), the work items and the piece counts the finalisation adds up.  It is pure host code, so
it is built alone with `g++ -fsanitize=address,undefined` (`make -C gpy_dla_detection_amd/csrc sanitize`) and driven
through dlagp_plan_debug with random masks (long gaps, one kept pixel, everything but the ends masked ...), sample
counts 1 .. 10^5, forced cuts and slot counts, checking the invariants the kernels rely on.  The sanitized library runs
in a child interpreter with libasan preloaded; a sanitizer report aborts the child and fails the test.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import REPO

CSRC = os.path.join(REPO, "gpy_dla_detection_amd", "csrc")
CHUNK, ENTRIES, MAX_PASS_CHUNKS, GROUP = 8, 64, 7, 16


def random_mask(rng, kind, n_u):
    keep = np.ones(n_u, dtype=np.uint8)
    if kind == "random":
        keep = (rng.random(n_u) > rng.choice([0.02, 0.3, 0.9])).astype(np.uint8)
    elif kind == "gaps":  # long masked runs, as SDSS masks have (read_spec.py:38-71)
        for _ in range(rng.integers(1, 6)):
            start = rng.integers(0, n_u)
            keep[start:start + rng.integers(1, max(2, n_u // 2))] = 0
    elif kind == "ends":  # everything but the two ends masked
        keep[:] = 0
        keep[0] = keep[-1] = 1
    elif kind == "one":
        keep[:] = 0
        keep[rng.integers(0, n_u)] = 1
    elif kind == "sparse":  # isolated pixels 40-70 entries apart: every chunk closes early
        keep[:] = 0
        keep[::int(rng.integers(40, 70))] = 1
    elif kind == "alternate":
        keep[1::2] = 0
    if not keep.any():
        keep[rng.integers(0, n_u)] = 1
    return keep


def check_case(lib, ct, rng, case):
    nb = int(rng.integers(1, 13)) if case % 5 else int(rng.integers(16, 41))  # tapering needs at least two rounds of eight
    ntaps = int(rng.choice([1, 7]))
    S = int(10 ** rng.uniform(0, 5)) if case % 7 else int(rng.integers(1, 40))
    if nb * S > 400000:
        nb = max(1, 400000 // S)
    kinds = ["none", "random", "gaps", "ends", "one", "sparse", "alternate"]
    masks = []
    for _ in range(nb):
        n_u = int(rng.choice([1, 2, 7, 8, 9, 57, 64, 65, 300, 1250, 3061, 4000]))
        masks.append(random_mask(rng, kinds[int(rng.integers(0, len(kinds)))], n_u))
    options = None
    if case % 3:
        options = np.array([rng.integers(0, 11), rng.integers(0, 101), rng.integers(0, 2),
                            rng.choice([1, 7, 64, 2048, 100000]), rng.integers(0, 2), rng.choice([1, 1, 3, 10]),
                            rng.choice([0, 1, 2, 3, 8])], dtype=np.int64)
    n_u = (ct.c_int * nb)(*[len(m) for m in masks])
    keep = (ct.POINTER(ct.c_ubyte) * nb)(*[m.ctypes.data_as(ct.POINTER(ct.c_ubyte)) for m in masks])
    opt_ptr = options.ctypes.data_as(ct.POINTER(ct.c_int64)) if options is not None else None
    summary = np.zeros(8, dtype=np.int64)
    sp = summary.ctypes.data_as(ct.POINTER(ct.c_int64))
    i32, u8 = ct.POINTER(ct.c_int32), ct.POINTER(ct.c_ubyte)
    rc = lib.dlagp_plan_debug(nb, n_u, keep, ntaps, S, opt_ptr, sp, None, None, 0, None, 0, None, 0, None, 0)
    assert rc == 0, lib.dlagp_last_error()
    nslots, npasses, nitems, nblocks, max_parts, full, remainder, _ = (int(x) for x in summary)
    slot_pixel = np.full(nslots, -7, dtype=np.int32)
    slot_pos = np.full(nslots, -7, dtype=np.int32)
    passes = np.full((npasses, 3), -7, dtype=np.int32)
    items = np.full((nitems, 6), -7, dtype=np.int32)
    counts = np.zeros(nblocks, dtype=np.uint8)
    rc = lib.dlagp_plan_debug(nb, n_u, keep, ntaps, S, opt_ptr, sp, slot_pixel.ctypes.data_as(i32),
                              slot_pos.ctypes.data_as(i32), nslots, passes.ctypes.data_as(i32), npasses,
                              items.ctypes.data_as(i32), nitems, counts.ctypes.data_as(u8), nblocks)
    assert rc == 0, lib.dlagp_last_error()

    ngroups = -(-S // GROUP)
    assert nblocks == nb * ngroups and nslots % CHUNK == 0
    # -- slots: every kept pixel in exactly one slot, in order; every chunk's windows inside 64 consecutive entries
    chunks_of, slot_base, at = [], [], 0
    for mask in masks:
        pos = np.flatnonzero(mask).astype(np.int32)  # window start of kept pixel i (plan_host.h window_starts)
        n = len(pos)
        end = at
        seen = 0
        while seen < n:  # the sightline's slots end where its n-th kept pixel has been placed, rounded up to a chunk
            seen += int(np.count_nonzero(slot_pixel[end:end + CHUNK] >= 0))
            end += CHUNK
        sp_, ss_ = slot_pixel[at:end], slot_pos[at:end]
        real = sp_ >= 0
        np.testing.assert_array_equal(sp_[real], np.arange(n))
        np.testing.assert_array_equal(ss_[real], pos)
        # a padding slot repeats the window of the slot before it; a chunk never starts with one
        assert real[::CHUNK].all()
        pad = np.flatnonzero(~real)
        np.testing.assert_array_equal(ss_[pad], ss_[pad - 1])
        by_chunk = ss_.reshape(-1, CHUNK)
        assert np.all(by_chunk[:, -1] + ntaps <= by_chunk[:, 0] + ENTRIES)
        assert np.all(np.diff(ss_) >= 0) and ss_[-1] + ntaps <= len(mask) + ntaps - 1  # inside the entry grid
        chunks_of.append((end - at) // CHUNK)
        slot_base.append(at)
        at = end
    assert at == nslots
    # -- passes: 1..7 chunks, windows inside the 64 entries that start at the first slot's window
    assert np.all((passes[:, 2] >= 1) & (passes[:, 2] <= MAX_PASS_CHUNKS)) and np.all(passes[:, 1] >= 0)
    # -- items: b / group in range, every block's pieces tile its sightline's chunks exactly once, in piece order
    colgroups = int(options[5]) if options is not None else 1
    assert nitems == colgroups * int(counts.sum()) and max_parts == int(counts.max()) and counts.min() >= 1 and max_parts <= 255
    b, grp, part, p0, p1, cg = items.T
    # every (block, piece) once per column group, the groups one after the other
    per_group = nitems // colgroups
    np.testing.assert_array_equal(cg, np.repeat(np.arange(colgroups), per_group))
    for g in range(1, colgroups):
        np.testing.assert_array_equal(items[g * per_group:(g + 1) * per_group, :5], items[:per_group, :5])
    items, nitems = items[:per_group], per_group
    b, grp, part, p0, p1, cg = items.T
    assert np.all((b >= 0) & (b < nb) & (grp >= 0) & (grp < ngroups) & (part >= 0))
    assert np.all((p0 >= 0) & (p0 < p1) & (p1 <= npasses))
    block = b.astype(np.int64) * ngroups + grp
    assert np.all(part < counts[block])
    key = block * 256 + part
    assert len(np.unique(key)) == nitems  # no (block, piece) twice; with nitems == sum(counts): each exactly once
    # the pass list of a piece is shared by all blocks of its sightline: check one representative per (b, piece)
    rep = {}
    for i in np.unique(b.astype(np.int64) * 256 + part, return_index=True)[1]:
        rep[(int(b[i]), int(part[i]))] = (int(p0[i]), int(p1[i]))
    same = {(int(bb), int(pp)): (int(x), int(y)) for bb, pp, x, y in zip(b[::max(1, nitems // 997)], part[::max(1, nitems // 997)],
                                                                          p0[::max(1, nitems // 997)], p1[::max(1, nitems // 997)])}
    for k, v in same.items():
        assert rep[k] == v
    for bb in range(nb):
        nparts = int(counts[bb * ngroups])
        assert np.all(counts[bb * ngroups:(bb + 1) * ngroups] == nparts)
        chunk = 0
        ss_ = slot_pos[slot_base[bb]:slot_base[bb] + chunks_of[bb] * CHUNK]
        for pp in range(nparts):
            lo, hi = rep[(bb, pp)]
            for entry_start, chunk_begin, nchunks in passes[lo:hi]:
                assert chunk_begin == chunk and chunk_begin + nchunks <= chunks_of[bb]  # contiguous, inside the operand
                assert entry_start == ss_[chunk_begin * CHUNK]
                assert ss_[(chunk_begin + nchunks) * CHUNK - 1] + ntaps <= entry_start + ENTRIES
                chunk += nchunks
        assert chunk == chunks_of[bb]  # exactly once, nothing left over
    if options is not None and options[0] > 0:
        assert not remainder
    return nitems


def run_cases(lib_path, seed, budget_s):
    import ctypes as ct
    import time

    lib = ct.CDLL(lib_path)
    lib.dlagp_last_error.restype = ct.c_char_p
    lib.dlagp_plan_debug.restype = ct.c_int
    lib.dlagp_plan_debug.argtypes = [ct.c_int, ct.POINTER(ct.c_int), ct.POINTER(ct.POINTER(ct.c_ubyte)), ct.c_int, ct.c_int64,
                                     ct.POINTER(ct.c_int64), ct.POINTER(ct.c_int64), ct.POINTER(ct.c_int32),
                                     ct.POINTER(ct.c_int32), ct.c_int64, ct.POINTER(ct.c_int32), ct.c_int64,
                                     ct.POINTER(ct.c_int32), ct.c_int64, ct.POINTER(ct.c_ubyte), ct.c_int64]
    rng = np.random.default_rng(seed)
    t0, cases, items = time.time(), 0, 0
    while cases < 40 or time.time() - t0 < budget_s:
        items += check_case(lib, ct, rng, cases)
        cases += 1
    # argument errors come back as codes, not as crashes
    summary = np.zeros(8, dtype=np.int64)
    sp = summary.ctypes.data_as(ct.POINTER(ct.c_int64))
    assert lib.dlagp_plan_debug(0, None, None, 7, 10, None, sp, None, None, 0, None, 0, None, 0, None, 0) == -1
    empty = np.zeros(5, dtype=np.uint8)
    n_u = (ct.c_int * 1)(5)
    keep = (ct.POINTER(ct.c_ubyte) * 1)(empty.ctypes.data_as(ct.POINTER(ct.c_ubyte)))
    assert lib.dlagp_plan_debug(1, n_u, keep, 7, 10, None, sp, None, None, 0, None, 0, None, 0, None, 0) == -1
    assert b"keeps no pixel" in lib.dlagp_last_error()
    return cases, items


def test_planner_invariants_under_asan_and_ubsan():
    res = subprocess.run(["make", "-C", CSRC, "sanitize"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan.so beside g++")
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(libasan),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",  # the interpreter itself leaks by design
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    code = ("import sys; sys.path.insert(0, %r); from tests.test_planner_fuzz import run_cases; "
            "print('FUZZ-OK %%d cases %%d items' %% run_cases(%r, 20241024, 25.0))"
            % (REPO, os.path.join(CSRC, "libdlagp_plan_host.so")))
    res = subprocess.run([sys.executable, "-c", code], env=env, cwd=REPO, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "FUZZ-OK" in res.stdout, (res.stdout[-2000:], res.stderr[-6000:])
    assert "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-6000:]
    cases = int(res.stdout.split("FUZZ-OK")[1].split()[0])
    assert cases >= 40


def test_production_library_exports_the_same_planner():
    """libdlagp_hip.so carries the identical code path (plan_host.h): same plan for the bench workload"""
    import ctypes as ct

    from gpy_dla_detection_amd import _lib
    from gpy_dla_detection_amd.synthetic import make_sightline

    sight = make_sightline(0, n_unmasked=3061, n_masked=61, k=20)
    mask = np.ascontiguousarray(sight.keep, dtype=np.uint8)
    lib = _lib.load()
    n_u = (ct.c_int * 1)(len(mask))
    keep = (ct.POINTER(ct.c_ubyte) * 1)(mask.ctypes.data_as(ct.POINTER(ct.c_ubyte)))
    summary = np.zeros(8, dtype=np.int64)
    _lib.check(lib.dlagp_plan_debug(1, n_u, keep, 7, 10000, None, summary.ctypes.data_as(ct.POINTER(ct.c_int64)),
                                    None, None, 0, None, 0, None, 0, None, 0))
    nslots, npasses, nitems, nblocks, max_parts, full, remainder, frac = (int(x) for x in summary)
    # configs[1]: 625 blocks x (3 full pieces + remainder) = 2500 work items (DESIGN "Work decomposition")
    assert (nblocks, full, remainder, max_parts, nitems) == (625, 3, 1, 4, 2500)
    assert nslots >= 3000 and nslots % 8 == 0 and 0.30 < frac / 1e9 < 0.33
