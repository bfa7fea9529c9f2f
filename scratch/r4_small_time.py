"""Launch time of the one-step dictionary kernel on small lattices (64 / 8 vectors, 126-step runs) and the calls users make there."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver
for shape in ([32, 32, 1], [64, 64, 1], [128, 128, 1], [200, 200, 1], [300, 300, 1], [20, 20, 20], [40, 40, 40]):
    system = bench.build_system(shape, "swave" if shape[2] == 1 else "dwave")
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    with DeviceSolver(indptr, indices, data) as solver:
        solver.set_lattice_shape(shape)
        line = f"{str(shape):16s}"
        for vectors in (64, 8):
            best = None
            for rnd in range(4):
                solver.dots_random(scale, 126, vectors, seed=rnd)
                p = solver.perf()
                us = p["kernel_ms"] / p["launches"] * 1e3
                best = us if best is None else min(best, us)
            line += f"  {vectors:2d} vectors: {best:7.2f} us/launch ({p['launches']} launches, {p['vector_steps'] / p['window_ms']:8.1f} k vector-steps/s, steps/launch {p['steps_per_launch']})"
        t0 = time.perf_counter(); system.free_energy(0.1, method="chebyshev", moments=512, vectors=64, trace="stochastic"); t1 = time.perf_counter()
        t0 = time.perf_counter(); system.free_energy(0.1, method="chebyshev", moments=512, vectors=64, trace="stochastic"); t1 = time.perf_counter()
        print(line + f"  free_energy(512 moments, 64 vectors) {1e3 * (t1 - t0):7.1f} ms", flush=True)
