#!/usr/bin/env python3
"""
Where the resampled base indices of a multi-level sweep point (profiles/r03_experiments.txt item 16): per spectrum and
level the number of distinct values among the S draws, the shares of the most frequent ones, and the share of the
blocks of 16 consecutive samples (one wave of the accumulate kernel) that agree.
usage: tools/base_index_stats.py [catalogue|bench]   (needs a GPU)
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

S = 10000


def report(tag, base):
    """base: (levels, S) indices of one spectrum"""
    for level, x in enumerate(base):
        blocks = x[: S // 16 * 16].reshape(-1, 16)
        agree = np.mean((blocks == blocks[:, :1]).all(axis=1))
        counts = np.sort(np.unique(x, return_counts=True)[1])[::-1]
        print("%s level %d: %2d distinct, top shares %s, waves that agree %.3f"
              % (tag, level + 2, len(counts), np.round(counts[:4] / S, 4).tolist(), agree))


def wave_report(tag, base):
    """
    base: (num, levels, S).  Per wave of the cached-level kernel (16 consecutive samples): how many DISTINCT cached
    rows its samples stack at each level, and in total over the levels -- what a kernel that reads every distinct row
    once per pass would have to hold.
    """
    num, levels, _ = base.shape
    blocks = base[:, :, : S // 16 * 16].reshape(num, levels, -1, 16)
    srt = np.sort(blocks, axis=3)
    distinct = 1 + (np.diff(srt, axis=3) != 0).sum(axis=3)  # (num, levels, waves)
    for level in range(levels):
        hist = np.bincount(distinct[:, level].ravel(), minlength=17)[1:]
        print("%s level %d: distinct rows per wave: mean %.2f, share with <= 1/2/4/6/8: %s"
              % (tag, level + 2, distinct[:, level].mean(),
                 np.round([hist[:k].sum() / hist.sum() for k in (1, 2, 4, 6, 8)], 4).tolist()))
    for upto in range(2, levels + 1):
        total = distinct[:, :upto].sum(axis=1).ravel()
        print("%s level %d kernel (rows of levels 2..%d): total distinct per wave mean %.2f, share <= 4/6/8/12/16: %s"
              % (tag, upto + 1, upto + 1, total.mean(), np.round([(total <= k).mean() for k in (4, 6, 8, 12, 16)], 4).tolist()))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "catalogue"
    if what == "catalogue":
        from gpy_dla_detection_amd.run_bayes_select import process_qso
        from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples, halton
        from gpy_dla_detection_amd.set_parameters import Parameters
        from tools.measure_catalogue import FixedPrior, make_inputs

        params = Parameters(num_dla_samples=S)
        u = halton(S, 3)
        dla = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u)
        sub = SubDLASamples(params, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
        learned, spectra, z_qsos = make_inputs(64)
        res = process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, batch_size=32)
        base = res["base_sample_inds"].transpose(0, 2, 1)  # (num, levels, S)
        for q in range(12):
            report("spectrum %2d (%s)" % (q, "absorber" if q % 3 else "none"), base[q])
        distinct = np.array([[len(np.unique(row)) for row in b] for b in base])
        print("mean distinct values per level:", distinct.mean(axis=0), "max", distinct.max())
        wave_report("catalogue", base)
    else:
        from gpy_dla_detection_amd import parallel
        from gpy_dla_detection_amd.sightline import DeviceBatch, DeviceSightline
        from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline

        sights = [make_sightline(i) for i in range(12)]  # the bench's n = 3000 sightlines
        batch = DeviceBatch([DeviceSightline.from_state(s, device=0) for s in sights])
        zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
        nhi = make_qmc_samples(sights[0], S)[2]
        base = parallel.sweep_batch_device(batch, zs, nhi, 4, 3000.0 * 1000.0 / 299792458.0, seed=0)[2]
        for b in range(4):
            report("sightline %d" % b, base[b])
        wave_report("bench", base)


if __name__ == "__main__":
    main()
