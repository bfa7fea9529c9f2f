"""Eigenvalues-only route (Householder tridiagonalisation + bisection on the GPU) against numpy / goldens, with timings."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import bodge_amd as ba
import systems
from bodge_amd import backend
from bodge_amd.solver import DeviceSolver

gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_arrays.npz"))
for name in ["swave20", "complex235", "random357", "snf", "chain128", "swave30_zeeman", "peierls30", "chain300", "swave50_zeeman"]:
    spec = systems.CATALOG[name]
    system = spec["build"](ba, **spec["kwargs"])
    n = system.shape[0]
    with DeviceSolver.from_hamiltonian(system) as dev, backend.options(BODGE_AMD_EIGH="tridiagonal"):
        dev.eigh(vectors=False)
        t0 = time.perf_counter(); w, _ = dev.eigh(vectors=False); dt = time.perf_counter() - t0
    ref = gold[f"{name}/eigenvalues"] if f"{name}/eigenvalues" in gold.files else None
    positive = w[w > 0]
    if n <= 4000:
        exact = np.linalg.eigvalsh(np.asarray(system.matrix("dense")))
        err_all = np.abs(w - exact).max()
    else:
        err_all = float("nan")
    err_ref = np.abs(positive - ref).max() if ref is not None and len(ref) == len(positive) else float("nan")
    print(f"{name:16s} n = {n:6d}  {dt * 1e3:9.1f} ms   max |w - numpy| = {err_all:.2e}   vs reference (positive) = {err_ref:.2e}"
          f"   sum w = {w.sum():.2e}", flush=True)
