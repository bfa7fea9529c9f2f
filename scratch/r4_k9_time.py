"""K9 (one-stage dense route) timings, real and complex matrices: eigenvalues and eigenpairs above 0."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import backend
for L, model in ((30, "swave"), (30, "peierls"), (30, "texture"), (50, "swave"), (50, "peierls")):
    system = bench.build_system((L, L, 1), model)
    solver = system._solver()
    with backend.options(BODGE_AMD_EIGH="tridiagonal", BODGE_AMD_EIGH_STAGES="1"):
        solver.eigh(vectors=False)
        t0 = time.perf_counter(); w, _ = solver.eigh(vectors=False); t_val = time.perf_counter() - t0
        solver.eigh_above(0.0)
        t0 = time.perf_counter(); w2, z = solver.eigh_above(0.0); t_vec = time.perf_counter() - t0
    bsr = system.matrix("bsr")
    vals = w2[w2 > 0]
    res = np.abs(bsr @ z - z * vals).max()
    print(f"L={L} {model:8s} n={4*L*L:6d} eigenvalues {t_val*1e3:8.1f} ms  eigenpairs {t_vec*1e3:8.1f} ms  residual {res:.1e}  trace check {abs(w.sum()):.1e}", flush=True)
