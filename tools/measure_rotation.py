#!/usr/bin/env python3
"""
Does a device-resident model selection run slower when consecutive steps work on DIFFERENT batches (each with its own
contexts, sample tables and multi-gigabyte sweep workspace, as in a catalogue run) than when one batch is swept again and
again (tools/measure_frac_vs_n.py)?  K batches of the synthetic catalogue (tools/measure_catalogue.py), 5-level model
selection each, accumulate time by event pairs.   usage: tools/measure_rotation.py [K] [rounds] [mix]
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpy_dla_detection_amd import parallel  # noqa: E402
from gpy_dla_detection_amd import sightline as sl  # noqa: E402
from gpy_dla_detection_amd.null_gp import prepare_data  # noqa: E402
from gpy_dla_detection_amd.run_bayes_select import DeviceEngine  # noqa: E402
from gpy_dla_detection_amd.samples import DLASamples, halton  # noqa: E402
from gpy_dla_detection_amd.set_parameters import Parameters  # noqa: E402
from tools.measure_catalogue import make_inputs  # noqa: E402

PEAK = 78.6


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    mix = sys.argv[3] if len(sys.argv) > 3 else "dr12q"
    nb, S = 32, 10000
    dev = torch.device("cuda", 0)
    params = Parameters(num_dla_samples=S)
    learned, spectra, z_qsos = make_inputs(K * nb, mix=mix)
    engine = DeviceEngine(learned, params)
    holder = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * halton(S, 3))
    f64 = dict(dtype=torch.float64, device=dev)
    nhi_dla = torch.from_numpy(10.0 ** (20.0 + 3.0 * halton(S, 3))).to(dev)
    nhi_sub = torch.from_numpy(10.0 ** (19.5 + 0.5 * halton(S, 3))).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    sets = []
    for k in range(K):
        devs, rows, kept = [], [], []
        for (w, f, v, m), zq in list(zip(spectra, z_qsos))[k * nb:(k + 1) * nb]:
            st = prepare_data(params, params.emitted_wavelengths(w, zq), f, v, m, zq, normalize=True)
            devs.append(engine.sightline(st, zq, True))
            rows.append(holder.sample_z_dlas(st["this_wavelengths"], zq))
            kept.append(len(st["x"]))
        sets.append(dict(
            devs=devs, batch=sl.DeviceBatch(devs), flop=float(np.sum(kept)) * 460.0 * 5 * S,
            z=torch.from_numpy(np.stack(rows)).to(dev), uniforms=torch.from_numpy(parallel.resampling_uniforms(nb, 4, S, seed=0)).to(dev),
            evid=torch.zeros(nb * 4, **f64), ll=torch.empty((nb, 4, S), **f64), base=torch.zeros((nb, 3, S), dtype=torch.int32, device=dev),
            sub_ev=torch.zeros(nb, **f64), sub_ll=torch.empty((nb, 1, S), **f64)))

    def step(s):
        s["batch"].two_model_sweep_device(S, 4, s["z"].data_ptr(), nhi_sub.data_ptr(), 0, nhi_dla.data_ptr(), 0, 3000.0 / 299792.458,
                                          s["uniforms"].data_ptr(), s["sub_ev"].data_ptr(), s["sub_ll"].data_ptr(),
                                          s["evid"].data_ptr(), s["ll"].data_ptr(), s["base"].data_ptr(), stream)

    def run(order):
        for s in order[:len(sets)]:  # every workspace allocated, clock up
            step(s)
        torch.cuda.synchronize()
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < 0.2:
            step(order[0])
            torch.cuda.synchronize()
        sl.set_option("timing", 1)
        sl.timing_read(reset=True)
        t0 = time.perf_counter()
        for s in order:
            step(s)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        sl.set_option("timing", 0)
        ms, launches = sl.timing_read(reset=True)
        flop = sum(s["flop"] for s in order)
        return {"steps": len(order), "wall_ms_per_step": 1e3 * wall / len(order), "accumulate_ms_per_step": ms / len(order),
                "launches_per_step": launches / len(order), "accumulate_frac": flop / (ms * 1e-3) / 1e12 / PEAK,
                "step_frac": flop / wall / 1e12 / PEAK}

    def run_threads(nthreads):
        """the rotation's steps issued by `nthreads` host threads onto the ONE stream, each thread its own batches"""
        import threading

        for s in sets:
            step(s)
        torch.cuda.synchronize()
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < 0.2:
            step(sets[0])
            torch.cuda.synchronize()
        sl.set_option("timing", 1)
        sl.timing_read(reset=True)

        def worker(t):
            torch.cuda.set_device(0)
            for _ in range(rounds):
                for s in sets[t::nthreads]:
                    step(s)

        t0 = time.perf_counter()
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        sl.set_option("timing", 0)
        ms, launches = sl.timing_read(reset=True)
        flop = rounds * sum(s["flop"] for s in sets)
        return {"threads": nthreads, "accumulate_frac": flop / (ms * 1e-3) / 1e12 / PEAK, "step_frac": flop / wall / 1e12 / PEAK,
                "launches": launches}

    out = {"batches": K, "rounds": rounds, "mix": mix,
           "each_batch_again_and_again": [run([s] * rounds) for s in sets],
           "batches_in_rotation": run(sets * rounds)}
    if K >= 3:
        out["rotation_issued_by_three_threads"] = run_threads(3)
    out["mean_frac_again_and_again"] = float(np.mean([r["accumulate_frac"] for r in out["each_batch_again_and_again"]]))
    print(json.dumps(out))
    for s in sets:
        s["batch"].close()
        for d in s["devs"]:
            d.close()
    engine.close()


if __name__ == "__main__":
    main()
