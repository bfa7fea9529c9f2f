import os, sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver
for shape, vectors, steps in (([200,200,1], 8, 1024), ([64,64,1], 4, 4096), ([20,20,1], 64, 512)):
    system = bench.build_system(shape); indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    solver = DeviceSolver(indptr, indices, data); solver.set_lattice_shape(shape)
    solver.dots_random(scale, 64, vectors)
    for rep in range(2):
        for env in ({"BODGE_AMD_GRAPH": "0"}, {"BODGE_AMD_GRAPH": "1"}):
            os.environ.update(env)
            t0 = time.perf_counter(); d, e = solver.dots_random(scale, steps, vectors); dt = time.perf_counter() - t0
            for k in env: del os.environ[k]
            p = solver.perf()
            print(shape, vectors, env, f"wall {dt*1e3:.2f} ms -> {dt/steps*1e6:.2f} us/step; kernels {p['kernel_ms']/steps*1e3:.2f} us/step; sum {d.sum():.6f}")
