#!/usr/bin/env python3
"""
The catalogue driver's pipeline run twice (and once in order on the calling thread) over the same synthetic catalogue: the
three runs' outputs must be equal to the last bit -- thread timing decides who launches when, never what is computed --
and the library's device memory must not grow from run to run.   usage: tools/soak_pipeline.py [spectra] [S] [rounds]
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.init()
from gpy_dla_detection_amd.run_bayes_select import process_qso  # noqa: E402
from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples, halton  # noqa: E402
from gpy_dla_detection_amd.set_parameters import Parameters  # noqa: E402
from gpy_dla_detection_amd.sightline import get_option  # noqa: E402
from tools.measure_catalogue import FixedPrior, make_inputs  # noqa: E402


def main():
    num = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000  # (the recorded runs' size; batches of few samples are multi-piece)
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    params = Parameters(num_dla_samples=S)
    u = halton(S, 3)
    dla = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u)
    sub = SubDLASamples(params, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    learned, spectra, z_qsos = make_inputs(num, mix="dr12q")
    first, report = None, {"spectra": num, "S": S, "rounds": []}
    for r in range(rounds):
        free0 = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter()
        res = process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, batch_size=32)
        t = time.perf_counter() - t0
        differing = []
        if first is None:
            first = res
        else:
            differing = [k for k in res if not np.array_equal(np.asarray(res[k]), np.asarray(first[k]),
                                                              equal_nan=np.asarray(res[k]).dtype.kind == "f")]
        report["rounds"].append({"seconds": round(t, 3), "arrays_differing_from_round_0": differing,
                                 "hbm_growth_gb": round((free0 - torch.cuda.mem_get_info()[0]) / 2 ** 30, 3),
                                 "big_allocs": get_option("big_allocs"), "big_frees": get_option("big_frees")})
    # In order on the calling thread the batches are 0-32, 32-64, ...; the pipeline's are 0-16, 16-48, 48-80, ...: every
    # spectrum rides in a different batch.  On batches a block of 16 samples is one wave's work from the first pixel to the
    # log-likelihood (single-piece plans), so its result does not depend on what else is in the launch: equal to the last bit.
    ordered = process_qso(spectra[:256], z_qsos[:256], params, FixedPrior(), dla, sub, learned, batch_size=32, prefetch=False)
    report["in_order_run_equals_pipeline_rows_0_to_256"] = {
        k: bool(np.array_equal(np.asarray(ordered[k])[:256], np.asarray(first[k])[:256], equal_nan=True))
        for k in ("base_sample_inds", "sample_log_likelihoods_dla", "sample_log_likelihoods_lls", "log_posteriors_dla",
                  "model_posteriors", "MAP_z_dlas")}
    report["ok"] = (all(not r["arrays_differing_from_round_0"] for r in report["rounds"])
                    and all(report["in_order_run_equals_pipeline_rows_0_to_256"].values()))
    print(json.dumps(report))
    return 0 if report["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
