#!/usr/bin/env python3
"""
Emulation of "a batch's levels as two halves on two streams" (profiles/r03_experiments.txt item 21): the DLA(1..4) sweep
of 32 resident sightlines as ONE batch on one stream against 2 / 4 sub-batches on streams of their own.
usage: tools/half_batches.py [n_unmasked]   (needs a GPU)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from gpy_dla_detection_amd import parallel, sightline as sl
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline
dev = torch.device('cuda', 0)
S, nb = 10000, 32
kw = dict(n_unmasked=int(sys.argv[1]) if len(sys.argv) > 1 else 3061)
sights = [make_sightline(i, **kw) for i in range(nb)]
devs = [sl.DeviceSightline.from_state(s, device=0) for s in sights]
z = torch.from_numpy(np.stack([make_qmc_samples(s, S)[0] for s in sights])).to(dev)
nhi = torch.from_numpy(make_qmc_samples(sights[0], S)[2]).to(dev)
uni = torch.from_numpy(parallel.resampling_uniforms(nb, 4, S, seed=0)).to(dev)
f64 = dict(dtype=torch.float64, device=dev)
MIN = 3000.0 * 1000.0 / 299792458.0
def run(groups, steps):
    """groups: list of (batch, rows slice, stream, buffers)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for b, rows, st, ev, ll, base in groups:
            b.sweep_device(S, 4, z[rows].data_ptr(), nhi.data_ptr(), 0, MIN, uni[rows].data_ptr(), ev.data_ptr(), ll.data_ptr(), base.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
def make(parts):
    out = []
    per = nb // parts
    for p in range(parts):
        rows = slice(p * per, (p + 1) * per)
        out.append((sl.DeviceBatch(devs[rows]), rows, torch.cuda.Stream(), torch.zeros(per * 4, **f64), torch.empty((per, 4, S), **f64), torch.zeros((per, 3, S), dtype=torch.int32, device=dev)))
    return out
for parts in (1, 2, 4, 1, 2, 4):
    g = make(parts)
    run(g, 2)
    print('parts', parts, 'ms/step %.3f' % run(g, 8), flush=True)
    evs = torch.cat([x[3] for x in g]).cpu().numpy()
    print('  checksum', float(np.nansum(evs)))
    for x in g: x[0].close()
