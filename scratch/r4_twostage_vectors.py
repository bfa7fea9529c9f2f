"""diagonalize()-style eigenpairs by the two-stage route (band inverse iteration + stage-1 block reflectors) against the
one-stage route: wall time, eigen-equation residual, orthonormality.  usage: python scratch/r4_twostage_vectors.py [L ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import backend

sizes = [int(v) for v in sys.argv[1:]] or [9, 18, 30]
for L in sizes:
    for model in ("swave", "potential"):
        system = bench.build_system((L, L, 1), model)
        n = 4 * L * L
        solver = system._solver()
        bsr = system.matrix("bsr")
        line = f"L={L:3d} n={n:6d} {model:9s}"
        for stages in ("1", "2"):
            with backend.options(BODGE_AMD_EIGH="tridiagonal", BODGE_AMD_EIGH_STAGES=stages):
                if n <= 4000:
                    solver.eigh_above(0.0)
                t0 = time.perf_counter()
                w, z = solver.eigh_above(0.0)
                dt = time.perf_counter() - t0
            vals = w[w > 0]
            z = np.asarray(z)
            idx = np.arange(0, vals.size, max(1, vals.size // 48))
            res = np.abs(bsr @ z[:, idx] - z[:, idx] * vals[idx]).max()
            gram = z[:, idx].conj().T @ z
            gram[np.arange(idx.size), idx] -= 1.0
            line += f"  | {stages}-stage {dt*1e3:8.1f} ms  residual {res:.1e} orthonormality {np.abs(gram).max():.1e} finite {np.isfinite(z).all()}"
        print(line, flush=True)
