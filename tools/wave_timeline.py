#!/usr/bin/env python3
"""
Schedule of the accumulate kernel's waves, from a diagnostic build of the library (-DDLAGP_TRACE_BUILD,
DLAGP_LIB=...): per work item start/end in 100 MHz real time and the (XCC, SE, CU, SIMD) it ran on.  Prints how full
the wave slots were over the launch and where the time between first start and last end went.
usage (GPU box): DLAGP_LIB=$PWD/ab/lib_trace.so python3 tools/wave_timeline.py
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gpy_dla_detection_amd import _lib
from gpy_dla_detection_amd import sightline as sl
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline

sight = make_sightline(0)
dev = sl.DeviceSightline.from_state(sight)
z, _, nhi = make_qmc_samples(sight, 10000)
for _ in range(300):  # sustained clocks
    dev.eval(z[:10000], nhi[:10000])
items = sl.get_option("last_launch_items")
buf = (ctypes.c_uint64 * (4 * 8192))()
_lib.check(_lib.load().dlagp_debug_read(buf, 4 * 8192, 0))
t = np.array(buf[: 4 * items], dtype=np.uint64).reshape(items, 4)
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
hw = t[:, 2]
passes, part = (t[:, 3] & 0xFFFFFFFF).astype(int), (t[:, 3] >> 32).astype(int)
t0 = start.min()
start, end = (start - t0) * 10.0, (end - t0) * 10.0  # ns
span = end.max()
simd = ((hw >> 4) & 3).astype(int)
cu = ((hw >> 8) & 15).astype(int)
sh = ((hw >> 12) & 1).astype(int)
se = ((hw >> 13) & 7).astype(int)
xcc = ((hw >> 32) & 15).astype(int)
print("items %d, span %.1f us (first start to last end)" % (items, span / 1e3))
print("starts: median %.1f us, 90%% %.1f us, max %.1f us" % tuple(np.percentile(start, [50, 90, 100]) / 1e3))
for p in sorted(set(part)):
    m = part == p
    print("  piece %d: %4d items, %2d passes, duration mean %.1f us (min %.1f, max %.1f), start mean %.1f us, end mean %.1f us"
          % (p, m.sum(), int(np.median(passes[m])), (end - start)[m].mean() / 1e3, (end - start)[m].min() / 1e3,
             (end - start)[m].max() / 1e3, start[m].mean() / 1e3, end[m].mean() / 1e3))
busy = (end - start).sum()
print("wave-slot occupancy: %.1f %% of 2048 slots x span" % (100.0 * busy / (2048 * span)))
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
per_simd = {}
for k, s_, a, b in zip(key, simd, start, end):
    per_simd.setdefault((int(k), int(s_)), []).append((a, b))
loads = np.array([sum(b - a for a, b in v) for v in per_simd.values()])
print("SIMDs used %d; wave-time per SIMD: mean %.1f us, min %.1f, max %.1f (2 x span = %.1f)" %
      (len(per_simd), loads.mean() / 1e3, loads.min() / 1e3, loads.max() / 1e3, 2 * span / 1e3))
counts = np.array([len(v) for v in per_simd.values()])
print("items per SIMD: " + ", ".join("%d x %d" % (c, (counts == c).sum()) for c in sorted(set(counts))))
ends = np.sort(end)
print("ends: 10%% %.1f us, 50%% %.1f, 90%% %.1f, 99%% %.1f, last %.1f" % tuple(np.percentile(ends, [10, 50, 90, 99, 100]) / 1e3))
# full-length items: how their duration depends on where they ran
full = passes == np.max(passes)
dur = (end - start) / 1e3
print("full items: duration percentiles 1/10/50/90/99: %.1f %.1f %.1f %.1f %.1f us" % tuple(np.percentile(dur[full], [1, 10, 50, 90, 99])))
for name, arr in (("xcc", xcc), ("se", se), ("simd", simd)):
    print("  by %s: " % name + "  ".join("%d: %.1f (n=%d)" % (v, dur[full & (arr == v)].mean(), (full & (arr == v)).sum()) for v in sorted(set(arr[full]))))
# does a SIMD's other slot matter?  full items whose SIMD also ran remainder items against those paired with another full item
count_on = {k: len(v) for k, v in per_simd.items()}
paired = np.array([count_on[(int(k), int(s_))] for k, s_ in zip(key, simd)])
print("  full item beside another full item: %.1f us; beside remainder items: %.1f us" % (dur[full & (paired == 2)].mean(), dur[full & (paired > 2)].mean()))
blk = np.arange(items)
print("  by dispatch order (blockIdx quartiles): " + "  ".join("%.1f" % dur[full & (blk * 4 // items == q)].mean() for q in range(4) if (full & (blk * 4 // items == q)).any()))
