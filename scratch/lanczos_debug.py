import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bodge_amd as ba
from scipy.linalg import eigh_tridiagonal
warnings.simplefilter("ignore")
rng = np.random.default_rng(0)
shape = (8, 11, 1)
for trial in range(40):
    lat = ba.CubicLattice(shape)
    system = ba.Hamiltonian(lat)
    mu = float(rng.uniform(0.5, 3.5)); h = float(rng.uniform(0.0, 0.4)); gap = float(rng.uniform(0.05, 1.0))
    periodic = rng.random() < 0.3
    with system as (H, D):
        H.set_sites(mu * ba.σ0 - h * ba.σ3)
        H.set_bonds(-1.0 * ba.σ0)
        pairs = lat.bond_array(coords=True)
        D.set_bonds(-gap * ba.dwave()(pairs[:, 0], pairs[:, 1]))
        if periodic: H.set_edges(-0.6 * ba.σ0)
    dense = np.asarray(system.matrix("dense"))
    w = np.linalg.eigvalsh(dense); wpos = w[w > 0]
    k = 3
    vals, vecs = system.lowest_eigenpairs(k, format="raw", method="lanczos")
    bad = np.abs(vals - wpos[:k]).max() > 1e-8
    print(f"trial {trial} mu={mu:.3f} h={h:.3f} gap={gap:.3f} periodic={periodic}: {'BAD' if bad else 'ok'} {vals} vs {wpos[:k]}", flush=True)
    if bad:
        solver = system._solver()
        solver.lanczos_begin(8, seed=0, max_iter=2000)
        a, b = solver.lanczos_advance(400)
        scale2 = a[:8].max()
        print("scale2", scale2, "residual of returned pairs", np.abs(dense @ vecs - vecs * vals).max(axis=0))
        for c in range(0):
            print(f"column {c}: beta/scale2 at 10-step intervals:", np.array2string(b[::10, c] / scale2, precision=2, max_line_width=200))
            for m in (50, 100, 150, 200, 300):
                th = eigh_tridiagonal(a[:m, c], b[:m - 1, c], eigvals_only=True, select="i", select_range=(0, 5))
                print(f"   m={m}: sqrt(theta) = {np.sqrt(np.clip(th, 0, None))}")
        print("distinct eigenvalues of H^2:", len(np.unique(np.round(wpos**2, 9))), " lowest eps:", wpos[:8])
        break
