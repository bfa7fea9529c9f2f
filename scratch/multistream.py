"""Experiment: exact-trace moments of a mid-size lattice with K device handles driven by K host threads."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from concurrent.futures import ThreadPoolExecutor
import numpy as np, bench
from bodge_amd import chebyshev, observables
from bodge_amd.solver import DeviceSolver

for L in (32, 64, 100):
    system = bench.build_system([L, L, 1])
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    rows = observables._electron_rows(system.shape[0])
    M = 600
    for K in (1, 2, 4, 8):
        solvers = [DeviceSolver(indptr, indices, data) for _ in range(K)]
        parts = [rows[k::K] for k in range(K)]
        def work(k):
            return solvers[k].moments_unit(scale, M, parts[k]).sum(axis=1)
        with ThreadPoolExecutor(K) as ex:
            list(ex.map(work, range(K)))  # warm-up
            t0 = time.perf_counter()
            mu = sum(ex.map(work, range(K)))
            dt = time.perf_counter() - t0
        print(f"{L}x{L}: K={K}: {dt*1e3:.1f} ms, F = {chebyshev.free_energy_series(2 * mu, scale, 0.1):.10f}", flush=True)
        for s in solvers: s.close()
