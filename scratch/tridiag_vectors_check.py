"""Eigenvectors through the tridiagonal route (inverse iteration + back-transformation): residual, orthogonality, time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import bodge_amd as ba
import systems
from bodge_amd import backend
from bodge_amd.solver import DeviceSolver

names = sys.argv[1:] or ["complex235", "random357", "snf", "chain128", "swave20", "dwave8", "chain300", "swave30_zeeman", "peierls30", "swave50_zeeman"]
for name in names:
    spec = systems.CATALOG[name]
    system = spec["build"](ba, **spec["kwargs"])
    n = system.shape[0]
    bsr = system.matrix("bsr")
    with DeviceSolver.from_hamiltonian(system) as dev, backend.options(BODGE_AMD_EIGH="tridiagonal"):
        t0 = time.perf_counter(); w, z = dev.eigh_above(0.0); dt = time.perf_counter() - t0
    vals = w[w > 0]
    res = np.abs(bsr @ z - z * vals).max()
    idx = np.arange(0, vals.size, max(1, vals.size // 256))
    gram = z[:, idx].conj().T @ z
    gram[np.arange(idx.size), idx] -= 1.0
    print(f"{name:16s} n = {n:6d}  vectors {z.shape[1]:6d}  {dt * 1e3:9.1f} ms   residual {res:.2e}   orthonormality {np.abs(gram).max():.2e}"
          f"   finite {np.isfinite(z).all()}", flush=True)
