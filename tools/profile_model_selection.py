import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gpy_dla_detection_amd import parallel
from gpy_dla_detection_amd.synthetic import make_sightline
sights = [make_sightline(i) for i in range(8)]
parallel.model_evidences_sharded(sights[:2], num_samples=10000, max_dlas=4)  # warm
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
parallel.model_evidences_sharded(sights, num_samples=10000, max_dlas=4)
print("total", time.perf_counter() - t0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
