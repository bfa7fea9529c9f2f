"""Eigenvalues by the two-stage route (K10: dense -> band by MFMA panels -> tridiagonal by bulge chasing) against numpy and
against the one-stage route, with wall times.  usage: python scratch/r4_twostage_check.py [L ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import backend

sizes = [int(v) for v in sys.argv[1:]] or [5, 9, 18, 30]
for L in sizes:
    for model in ("swave", "potential") if L < 70 else ("swave",):
        system = bench.build_system((L, L, 1), model)
        n = 4 * L * L
        solver = system._solver()
        out = {}
        for stages in ("1", "2"):
            if stages == "1" and n > 20000 and not os.environ.get("CHECK_ONE_STAGE"):
                continue  # (50 s per solve at n = 4e4)
            with backend.options(BODGE_AMD_EIGH="tridiagonal", BODGE_AMD_EIGH_STAGES=stages):
                if n <= 20000:
                    solver.eigh(vectors=False)
                t0 = time.perf_counter()
                w, _ = solver.eigh(vectors=False)
                out[stages] = (w, time.perf_counter() - t0)
        if "1" not in out:
            out["1"] = (out["2"][0], float("nan"))
        line = f"L={L:3d} n={n:6d} {model:9s} one-stage {out['1'][1]*1e3:8.1f} ms  two-stage {out['2'][1]*1e3:8.1f} ms  |two - one| {np.abs(out['2'][0] - out['1'][0]).max():.2e}"
        if n <= 4000:
            ref = np.linalg.eigvalsh(np.asarray(system.matrix("dense")))
            line += f"  |two - numpy| {np.abs(out['2'][0] - ref).max():.2e}  |one - numpy| {np.abs(out['1'][0] - ref).max():.2e}"
        else:
            w = out["2"][0]
            line += f"  sum {w.sum():.2e}  sum of squares / ||H||_F^2 - 1 = {np.sum(w * w) / np.sum(np.abs(system.bsr_arrays()[2]) ** 2) - 1:.2e}"
        print(line, flush=True)
