#!/usr/bin/env python3
"""
Ranks above 20: accumulate-kernel throughput of one n = 3000 sightline x 10 000 samples on the MFMA kernel (column groups)
and on the generic kernel, per rank.  One JSON line.   usage: tools/measure_ranks.py [ranks ...]   (needs a GPU)
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpy_dla_detection_amd import sightline as sl  # noqa: E402
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline  # noqa: E402

S = 10000
ranks = [int(a) for a in sys.argv[1:]] or [20, 21, 32, 40, 64]
out = {}
for k in ranks:
    sight = make_sightline(0, n_unmasked=3061, n_masked=61, k=k)
    dev = sl.DeviceSightline.from_state(sight, device=0)
    z, _, nhi = make_qmc_samples(sight, S)
    d_z, d_nhi = torch.from_numpy(z).cuda(), torch.from_numpy(nhi).cuda()
    d_out = torch.empty(S, dtype=torch.float64, device="cuda")
    flops = sight.n * k * (k + 1) + 2 * sight.n * k
    row = {"flop_per_eval": flops}
    for path, name, steps in ((2, "mfma", 20), (1, "generic", 3)):
        sl.set_option("path", path)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            dev.eval_device(S, 1, d_z.data_ptr(), d_nhi.data_ptr(), d_out.data_ptr(), stream)
        torch.cuda.synchronize()
        sl.set_option("timing", 1)
        sl.timing_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            dev.eval_device(S, 1, d_z.data_ptr(), d_nhi.data_ptr(), d_out.data_ptr(), stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        sl.set_option("timing", 0)
        kernel_ms, launches = sl.timing_read(reset=True)
        kernel = kernel_ms / 1e3 / launches
        row[name] = {"ms_per_step": wall * 1e3, "accumulate_ms": kernel * 1e3, "evals_per_s": S / wall,
                     "accumulate_tflops": S * flops / kernel / 1e12, "frac_of_78.6": S * flops / kernel / 1e12 / 78.6,
                     "items": sl.get_option("last_launch_items") if path == 2 else None, "checksum": float(d_out.sum().item())}
    sl.set_option("path", 0)
    row["mfma_over_generic"] = row["generic"]["ms_per_step"] / row["mfma"]["ms_per_step"]
    out[str(k)] = row
    dev.close()
print(json.dumps(out))
