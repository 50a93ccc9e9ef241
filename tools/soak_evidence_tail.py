#!/usr/bin/env python3
"""
Soak of the level evidence in finalize_rank20's tail (dlagp_*eval_evidence_device): random sightlines, batch sizes, sample
counts, absorber counts and masks; every call against the two launches it replaces (dlagp_*eval_device +
dlagp_level_evidence_device): log-likelihoods bit for bit, evidences to 1e-13 and equal to NumPy's log-mean-exp of the
returned log-likelihoods.  Consecutive calls use DIFFERENT inputs and alternate between two workspaces' worth of shapes, so a
block's sums left over from an earlier launch (a stale read across compute units: the hand-off uses write-through stores and a
ticket, no agent-scope fence) would show as a wrong evidence.  One JSON line.   usage: tools/soak_evidence_tail.py [seed] [trials]
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpy_dla_detection_amd import sightline as sl  # noqa: E402
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline  # noqa: E402


def numpy_evidence(ll, level, S):
    out = np.full(ll.shape[0], np.nan)
    for b, row in enumerate(ll):
        if np.isfinite(row).any():
            top = np.nanmax(row)
            out[b] = top + np.log(np.nanmean(np.exp(row - top))) - level * np.log(S)
    return out


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(seed)
    device = torch.device("cuda", 0)
    sights = [make_sightline(100 + i, n_unmasked=int(rng.integers(120, 1500)), n_masked=int(rng.integers(0, 60))) for i in range(8)]
    devs = [sl.DeviceSightline.from_state(s) for s in sights]
    batches = {}
    stats = {"trials": trials, "folded": 0, "not_folded": 0, "ll_differs": 0, "evidence_beyond_1e-13": 0, "evidence_not_numpys": 0,
             "worst_evidence_rel": 0.0, "all_nan_levels": 0}
    for trial in range(trials):
        nb = int(rng.choice([1, 1, 1, 2, 3, 5, 8]))
        members = sorted(rng.choice(8, size=nb, replace=False).tolist())
        S = int(rng.choice([1, 15, 16, 17, 255, 1000, 4097, 10000, 16400, 20000])) if rng.random() < 0.5 else int(rng.integers(1, 12000))
        kd = int(rng.integers(1, 4))
        level = kd - 1
        if nb == 1:
            target = devs[members[0]]
        else:
            key = tuple(members)
            if key not in batches:
                batches[key] = sl.DeviceBatch([devs[m] for m in members])
            target = batches[key]
        z = np.stack([rng.permutation(make_qmc_samples(sights[m], S * kd)[0]).reshape(S, kd) for m in members])
        nhi = np.stack([rng.permutation(make_qmc_samples(sights[m], S * kd)[2]).reshape(S, kd) for m in members])
        share = rng.choice([0.0, 0.1, 0.9, 1.0])
        mask = (rng.random((nb, S)) < share).astype(np.uint8)
        d_z, d_n, d_mask = (torch.from_numpy(a).to(device) for a in (z, nhi, mask))
        want = torch.empty((nb, S), dtype=torch.float64, device=device)
        want_ev = torch.zeros(nb, dtype=torch.float64, device=device)
        got = torch.full((nb, S), 3.0, dtype=torch.float64, device=device)
        got_ev = torch.full((nb,), 3.0, dtype=torch.float64, device=device)
        target.eval_evidence_device(S, kd, d_z.data_ptr(), d_n.data_ptr(), got.data_ptr(), level, got_ev.data_ptr(),
                                    d_too_close=d_mask.data_ptr())
        folded = sl.get_option("last_evidence_folded")
        target.eval_device(S, kd, d_z.data_ptr(), d_n.data_ptr(), want.data_ptr())
        sl.level_evidence_device(want.data_ptr(), nb, S, level, want_ev.data_ptr(), d_too_close=d_mask.data_ptr())
        torch.cuda.synchronize()
        stats["folded" if folded else "not_folded"] += 1
        g, w, ge, we = got.cpu().numpy(), want.cpu().numpy(), got_ev.cpu().numpy(), want_ev.cpu().numpy()
        if not np.array_equal(g, w, equal_nan=True):
            stats["ll_differs"] += 1
        ref = numpy_evidence(g, level, S)
        stats["all_nan_levels"] += int(np.isnan(ref).sum())
        for a, b, name in ((ge, we, "evidence_beyond_1e-13"), (ge, ref, "evidence_not_numpys")):
            if not np.array_equal(np.isnan(a), np.isnan(b)):
                stats[name] += 1
                continue
            ok = ~np.isnan(a)
            if ok.any():
                rel = float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), 1.0)))
                stats["worst_evidence_rel"] = max(stats["worst_evidence_rel"], rel)
                if rel > 1e-13:
                    stats[name] += 1
    for batch in batches.values():
        batch.close()
    for d in devs:
        d.close()
    stats["ok"] = not (stats["ll_differs"] or stats["evidence_beyond_1e-13"] or stats["evidence_not_numpys"])
    print(json.dumps(stats))
    sys.exit(0 if stats["ok"] else 1)


if __name__ == "__main__":
    main()
