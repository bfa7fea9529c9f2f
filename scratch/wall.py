import os, sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver
system = bench.build_system([1000,1000,1]); indptr, indices, data = system.bsr_arrays()
scale = chebyshev.spectral_bound(indptr, data)
solver = DeviceSolver(indptr, indices, data); solver.set_lattice_shape([1000,1000,1])
solver.moments_random(scale, 16, 8)
for rep in range(4):
    for env in ({}, {"BODGE_AMD_DICT": "0"}):
        os.environ.update(env)
        t0 = time.perf_counter(); solver.moments_random(scale, 512, 8); dt = time.perf_counter() - t0
        for k in env: del os.environ[k]
        p = solver.perf()
        print(rep, env, f"wall {dt*1e3:.2f} ms  kernels {p['kernel_ms']:.2f} ms  diff {dt*1e3-p['kernel_ms']:.2f}")
