import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from bodge_amd.solver import DeviceSolver
for shape, model in (([1000, 1000, 1], "swave"), ([100, 100, 100], "dwave")):
    system = bench.build_system(shape, model)
    t0 = time.perf_counter(); indptr, indices, data = system.bsr_arrays(); t1 = time.perf_counter()
    for rep in range(2):
        t2 = time.perf_counter(); solver = DeviceSolver(indptr, indices, data); t3 = time.perf_counter(); solver.close()
        print(f"{shape}: bsr_arrays {t1 - t0:.2f} s, bdg_create {t3 - t2:.2f} s", flush=True)
