"""BASELINE config 5 ladder: dense diagonalisation of (L, L, 1) s-wave + Zeeman lattices on the GPU.

Eigenvalues are compared with numpy.linalg.eigvalsh of the same dense matrix (the reference's
route) where the host can do it in reasonable time, otherwise with trace identities.
usage: python3 scratch/dense_ladder.py 30 50 100 [--cpu-limit 10000]
"""
import sys, time, threading, os
sys.path.insert(0, ".")
import numpy as np
import bench
from bodge_amd.solver import DeviceSolver

opts = dict(zip(sys.argv[1:], sys.argv[2:]))
sizes = [int(a) for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and not sys.argv[i].startswith("--")]
cpu_limit = int(opts.get("--cpu-limit", "10000"))
stop = False
def heartbeat():
    t0 = time.time()
    while not stop:
        time.sleep(30)
        print(f"  ... {time.time() - t0:.0f}s", flush=True)
threading.Thread(target=heartbeat, daemon=True).start()

for L in sizes:
    system = bench.build_system([L, L, 1])
    n = 4 * L * L
    solver = system._solver()
    print(f"== ({L},{L},1): n = {n}, dense {'real' if np.abs(system._matrix.data.imag).max() == 0 else 'complex'} "
          f"{n * n * 8 / 1e9:.2f} GB (float64)", flush=True)
    for vectors in (False, True):
        if vectors and n > 20000:
            continue
        t0 = time.time()
        w, z = solver.eigh(vectors=vectors)
        dt = time.time() - t0
        print(f"  GPU eigh vectors={vectors}: {dt:.2f} s", flush=True)
        if vectors:
            h = system._matrix.tocsr()
            k = np.linspace(0, n - 1, 16).astype(int)
            cols = np.ascontiguousarray(z[:, k])  # eigenvectors are the columns
            res = np.abs(h @ cols - cols * w[k]).max()
            print(f"  residual max |H z - w z| over 16 sampled pairs: {res:.2e}", flush=True)
    bsr = system._matrix
    fro = float((np.abs(bsr.data) ** 2).sum())
    print(f"  trace identities: sum w = {w.sum():.3e} (exact 0 by particle-hole symmetry), "
          f"sum w^2 - |H|_F^2 = {np.dot(w, w) - fro:.3e} (rel {abs(np.dot(w, w) - fro) / fro:.1e})", flush=True)
    if n <= cpu_limit:
        t0 = time.time()
        ref = np.linalg.eigvalsh(np.asarray(system.matrix("dense")))
        print(f"  CPU numpy.linalg.eigvalsh: {time.time() - t0:.2f} s; max |w - ref| = {np.abs(np.sort(w) - ref).max():.2e}", flush=True)
stop = True
