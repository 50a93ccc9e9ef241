#!/usr/bin/env python3
"""
How the work items of a RAGGED batch (sightlines of different lengths: a catalogue's) reach the eight XCDs: from a trace
build of the library (-DDLAGP_TRACE_BUILD, DLAGP_LIB=...), the first 8192 work items of one sub-DLA sweep of 32 sightlines
with n_u uniform in [264, 1246] -- start / end (100 MHz real time) and XCC of every item.  Prints: whether workgroup i
runs on XCD i mod 8, whether starts are in dispatch order, and per XCD the busy wave-slot time over the window in which
all XCDs still have items coming.
usage (GPU box): DLAGP_LIB=$PWD/ab/r5_trace.so python3 tools/trace_ragged.py [library option name=value ...]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gpy_dla_detection_amd import _lib
from gpy_dla_detection_amd import sightline as sl
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline

nb, S = 32, 10000
rng = np.random.default_rng(5)
sights = []
for i in range(nb):
    n_u = int(round(755 * (1 + 0.65 * (2 * rng.random() - 1))))
    sights.append(make_sightline(i, n_unmasked=n_u, n_masked=max(1, round(n_u * 61 / 3061)), k=20))
devs = [sl.DeviceSightline.from_state(s, device=0) for s in sights]
batch = sl.DeviceBatch(devs)
for item in sys.argv[1:]:
    name, value = item.split("=")
    sl.set_option(name, int(value))
dev = torch.device("cuda", 0)
z = torch.from_numpy(np.stack([make_qmc_samples(s, S)[0] for s in sights])).to(dev)
nhi = torch.from_numpy(make_qmc_samples(sights[0], S, sub_dla=True)[2]).to(dev)
ev = torch.zeros(nb, dtype=torch.float64, device=dev)
ll = torch.empty((nb, 1, S), dtype=torch.float64, device=dev)
for _ in range(30):  # sustained clocks; the last launch is the one traced
    batch.sweep_device(S, 1, z.data_ptr(), nhi.data_ptr(), 0, 0.01, 0, ev.data_ptr(), ll.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
items = min(sl.get_option("last_launch_items"), 8192)
buf = (ctypes.c_uint64 * (4 * 8192))()
_lib.check(_lib.load().dlagp_debug_read(buf, 4 * 8192, 0))
t = np.array(buf[: 4 * items], dtype=np.uint64).reshape(items, 4)
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
xcc = ((t[:, 2] >> 32) & 15).astype(int)
passes = (t[:, 3] & 0xFFFFFFFF).astype(int)
t0 = start.min()
start, end = (start - t0) / 100.0, (end - t0) / 100.0  # us
idx = np.arange(items)
print("items traced %d of %d; lengths n = %s" % (items, sl.get_option("last_launch_items"), sorted(s.n for s in sights)))
print("workgroup i on XCD i mod 8: %.1f %% of the items" % (100.0 * np.mean(xcc == idx % 8)))
print("XCDs seen: %s" % sorted(set(xcc.tolist())))
order_breaks = np.mean(np.diff(start) < -0.5)
print("starts out of dispatch order by more than 0.5 us: %.2f %% of consecutive pairs" % (100 * order_breaks))
# window: from the moment every slot is filled (2048-th start) to the last traced start: every XCD still has items queued
w0, w1 = np.sort(start)[min(2047, items - 1)], start.max()
print("window %.1f .. %.1f us" % (w0, w1))
for x in range(8):
    m = xcc == x
    busy = np.clip(np.minimum(end[m], w1) - np.maximum(start[m], w0), 0, None).sum()
    print("  XCD %d: %4d items, mean passes %.1f, mean duration %.1f us, slot occupancy in the window %.1f %% of 256 slots"
          % (x, m.sum(), passes[m].mean(), (end - start)[m].mean(), 100.0 * busy / (256 * (w1 - w0))))
gaps = np.diff(np.sort(start[2048:])) if items > 2048 else np.array([0.0])
print("start-to-start gaps after the first fill: median %.3f us, 99%% %.3f us, max %.1f us" % (np.median(gaps), np.percentile(gaps, 99), gaps.max()))
