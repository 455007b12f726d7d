import numpy as np, time, os
from threadpoolctl import threadpool_limits, threadpool_info
import scipy.linalg
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
print([ (i['internal_api'], i['num_threads']) for i in threadpool_info()])
rs = np.random.RandomState(0)
a = rs.normal(size=(4000, 512)); cov = a.T @ a / 4000
for nt in (1, 2, 4, 8, 16, None):
    with threadpool_limits(limits=nt):
        ts = []
        for _ in range(5):
            t = time.perf_counter(); w, v = scipy.linalg.eigh(cov, subset_by_index=[508, 511]); ts.append(time.perf_counter() - t)
        t2 = []
        for _ in range(3):
            t = time.perf_counter(); w, v = np.linalg.eigh(cov); t2.append(time.perf_counter() - t)
    print("threads", nt, "scipy subset ms", [round(x * 1e3, 1) for x in ts], "numpy full ms", [round(x * 1e3, 1) for x in t2])
