"""
Test infrastructure: the oracle (oracle/ref_like.py) over a whole population of parameter sets, fanned over the host's
cores.  The GPU tests compare EVERY sample of the BASELINE configurations with the oracle, not a spot check: one
process does 2-3 k evaluations a second at n = 3000, sixteen finish 10 000 in under a second.

The workers are fresh interpreters (this file run as a script): the pytest process has initialised the GPU by the time
these tests run, and a forked copy of it must not be the one doing the work.  A job travels as one .npz in a temporary
directory: the sightline's post-`set_data` state (reference null_gp.py:117-177, :240-242) and the (m, kd) parameter
sets; worker w of W evaluates rows w, w + W, ... (absorber counts and costs interleave evenly) and writes its values.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_STATE_FIELDS = ("y", "v", "this_mu", "this_M", "this_omega2", "padded_wavelengths", "unmasked_wavelengths", "keep")


def host_workers(limit: int = 16) -> int:
    try:
        return max(1, min(limit, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(limit, os.cpu_count() or 1))


def oracle_log_likelihoods(jobs, workers: int = 0, num_lines: int = 3, broadening: bool = True):
    """
    :param jobs: [(sightline, z (m, kd), nhi (m, kd)), ...] -- `sightline` any object with the post-set_data attribute
        names (`synthetic.SyntheticSightline`); every row one `sample_log_likelihood_k_dlas` (reference dla_gp.py:311-329)
    :return: [values (m,), ...] in the order of `jobs`
    """
    workers = workers or host_workers()
    with tempfile.TemporaryDirectory(prefix="oracle_fanout_") as tmp:
        for j, (sight, z, nhi) in enumerate(jobs):
            z, nhi = np.asarray(z, dtype=np.float64), np.asarray(nhi, dtype=np.float64)
            if z.ndim == 1:
                z, nhi = z[:, None], nhi[:, None]
            assert z.shape == nhi.shape and z.ndim == 2
            state = {name: np.asarray(getattr(sight, name)) for name in _STATE_FIELDS}
            np.savez(os.path.join(tmp, "job%d.npz" % j), z=z, nhi=nhi, num_lines=num_lines, broadening=broadening, **state)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1",
                   PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), tmp, str(len(jobs)), str(w), str(workers)],
                                  env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for w in range(workers)]
        for w, proc in enumerate(procs):
            out, _ = proc.communicate(timeout=900)
            if proc.returncode != 0:
                raise RuntimeError("oracle worker %d failed:\n%s" % (w, out[-2000:]))
        results = []
        for j, (_, z, _) in enumerate(jobs):
            m = np.asarray(z).shape[0]
            values = np.empty(m)
            for w in range(workers):
                values[w::workers] = np.load(os.path.join(tmp, "out%d_%d.npy" % (j, w)))
            results.append(values)
        return results


def _worker(tmp: str, num_jobs: int, w: int, workers: int) -> None:
    from oracle import ref_like as R

    for j in range(num_jobs):
        with np.load(os.path.join(tmp, "job%d.npz" % j)) as job:
            st = R.SpectrumState(*(job[name] for name in _STATE_FIELDS), num_lines=int(job["num_lines"]),
                                 broadening=bool(job["broadening"]))
            z, nhi = job["z"][w::workers], job["nhi"][w::workers]
        values = np.array([R.sample_log_likelihood_k_dlas(st, z[i], nhi[i]) for i in range(z.shape[0])])
        np.save(os.path.join(tmp, "out%d_%d.npy" % (j, w)), values)


if __name__ == "__main__":
    _worker(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
