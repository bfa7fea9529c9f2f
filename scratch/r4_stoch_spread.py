"""Spread of the 8-vector stochastic-trace estimates of F/N on the 1000x1000 bench system (sets of disjoint vector ids, two seeds)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver
system = bench.build_system([1000, 1000, 1], "swave")
indptr, indices, data = system.bsr_arrays()
scale = chebyshev.spectral_bound(indptr, data)
with DeviceSolver(indptr, indices, data) as solver:
    solver.set_lattice_shape([1000, 1000, 1])
    for seed in (0, 1):
        f = [chebyshev.free_energy_series(solver.moments_random(scale, 512, 8, seed=seed, first_id=first) / 8, scale, 0.5) / 1e6 for first in range(0, 128, 8)]
        f = np.array(f)
        print(f"seed {seed}: mean {f.mean():.6f}  std of an 8-vector estimate {f.std(ddof=1):.2e}  max |f_i - f_j| {np.ptp(f):.2e}  first two {f[0]:.6f} {f[1]:.6f}")
