#!/usr/bin/env python3
"""
Throughput of the BASELINE.json configurations other than the bench line (configs[1]) on one GPU:
device-resident timings of batches and of sweeps with 1..4 absorbers, and the end-to-end model
selection rate through the drop-in classes (host resampling included).  Prints one JSON object.
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

torch.cuda.init()
from gpy_dla_detection_amd import parallel, sightline as sl
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline

S = 10000
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
out = {}


def timed(fn, reps=10, warm=2, ramp_s=0.15):
    # keep the device busy before timing: it needs ~20 ms of load to leave its idle power state
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < ramp_s:
        fn()
        torch.cuda.synchronize()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# one sightline, 1..4 absorbers per sample (config 5's levels)
sight = make_sightline(1)
ds = sl.DeviceSightline.from_state(sight)
z, _, nhi = make_qmc_samples(sight, S)
rng = np.random.default_rng(0)
for kd in (1, 2, 3, 4):
    zz = np.stack([z] + [z[rng.permutation(S)] for _ in range(kd - 1)], axis=1)
    nn = np.stack([nhi] + [nhi[rng.permutation(S)] for _ in range(kd - 1)], axis=1)
    d_z, d_n = torch.from_numpy(zz).to(dev), torch.from_numpy(nn).to(dev)
    d_o = torch.empty(S, dtype=torch.float64, device=dev)
    t = timed(lambda: ds.eval_device(S, kd, d_z.data_ptr(), d_n.data_ptr(), d_o.data_ptr(), stream), reps=100)
    out["one_sightline_kd%d_evals_per_s" % kd] = S / t

# the host-pointer entry (dlagp_eval: H2D of z / nhi, launch, D2H of the log-likelihoods, synchronous)
t = timed(lambda: ds.eval(z, nhi), reps=100)
out["one_sightline_kd1_host_pointers_evals_per_s"] = S / t

# config 3: 64 sightlines x 10 000 samples, DLA(1) + sub-DLA
nb = 64
sights = [make_sightline(i) for i in range(nb)]
devs = [sl.DeviceSightline.from_state(s) for s in sights]
batch = sl.DeviceBatch(devs)
zb = np.stack([make_qmc_samples(s, S)[0] for s in sights])
d_z = torch.from_numpy(zb).to(dev)
d_dla = torch.from_numpy(np.broadcast_to(make_qmc_samples(sights[0], S)[2], zb.shape).copy()).to(dev)
d_sub = torch.from_numpy(np.broadcast_to(make_qmc_samples(sights[0], S, sub_dla=True)[2], zb.shape).copy()).to(dev)
d_o = torch.empty((nb, S), dtype=torch.float64, device=dev)


def config3():
    batch.eval_device(S, 1, d_z.data_ptr(), d_dla.data_ptr(), d_o.data_ptr(), stream)
    batch.eval_device(S, 1, d_z.data_ptr(), d_sub.data_ptr(), d_o.data_ptr(), stream)


t = timed(config3, reps=5, warm=1)
out["config3_64x10000_dla_plus_subdla_evals_per_s"] = 2 * nb * S / t
out["config3_ms"] = t * 1e3

# end-to-end model selection [null, sub-DLA, DLA(1..4)] through the batch driver; contexts and samples
# are built once (as a catalogue run would), only the sweeps are timed
nsel = 16


def selection_rate(runner, reps=3):
    runner.run()
    t0 = time.perf_counter()
    for _ in range(reps):
        ev = runner.run()
    return nsel * reps / (time.perf_counter() - t0), ev


class HostLevels:  # evaluator without .sweep: ShardRunner does the per-level host half (exact reference RNG path)
    def __init__(self, inner):
        self.eval, self.nulls = inner.eval, inner.nulls


rate, ev_host = selection_rate(parallel.ShardRunner(
    sights[:nsel], num_samples=S, max_dlas=4, make_batch=lambda s, d: HostLevels(parallel._device_batch(s, d))))
out["model_selection_host_levels_spectra_per_s"] = rate
rate, ev = selection_rate(parallel.ShardRunner(sights[:nsel], num_samples=S, max_dlas=4))
out["model_selection_spectra_per_s"] = rate
out["model_selection_evals_per_s"] = rate * (1 + S + 4 * S)
out["model_selection_argmax"] = np.argmax(ev, axis=1).tolist()
out["model_selection_device_equals_host_levels"] = bool(np.allclose(ev, ev_host, rtol=1e-12))

# the same sweep with everything resident in HBM (dlagp_batch_sweep_device): 64 sightlines, DLA(1..4)
max_dlas = 4
d_nhi = torch.from_numpy(make_qmc_samples(sights[0], S)[2]).to(dev)
d_u = torch.from_numpy(parallel.resampling_uniforms(nb, max_dlas, S, 0)).to(dev)
d_ev = torch.empty((nb, max_dlas), dtype=torch.float64, device=dev)
d_ll = torch.empty((nb, max_dlas, S), dtype=torch.float64, device=dev)
d_base = torch.empty((nb, max_dlas - 1, S), dtype=torch.int32, device=dev)
t = timed(lambda: batch.sweep_device(S, max_dlas, d_z.data_ptr(), d_nhi.data_ptr(), 0, 3000.0e3 / 299792458.0,
                                     d_u.data_ptr(), d_ev.data_ptr(), d_ll.data_ptr(), d_base.data_ptr(), stream),
          reps=3, warm=1)
out["device_sweep_64x10000_dla1to4_ms"] = t * 1e3
out["device_sweep_64x10000_dla1to4_evals_per_s"] = nb * S * max_dlas / t
out["device_sweep_64x10000_dla1to4_spectra_per_s"] = nb / t
print(json.dumps(out))
