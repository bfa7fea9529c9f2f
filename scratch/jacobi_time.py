import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, bodge_amd as ba, systems, bench
for L in (3, 8, 13, 20, 22):  # 4N <= 2048: the own kernels (larger sizes would load rocSOLVER)
    s = bench.build_system([L, L, 1]) if L != 13 else systems.random_periodic(ba)
    n = s.shape[0]
    dense = np.asarray(s.matrix("dense"))
    ref = np.linalg.eigvalsh(dense)
    sol = s._solver()
    for vectors in (False, True):
        t0 = time.time(); w, z = sol.eigh(vectors=vectors); dt = time.time() - t0
        msg = f"n={n} vectors={vectors}: {dt:.3f} s, max|w-ref| = {np.abs(w - ref).max():.2e}"
        if vectors:
            msg += f", residual {np.abs(dense @ z - z * w).max():.2e}, orth {np.abs(z.conj().T @ z - np.eye(n)).max():.2e}"
        print(msg, flush=True)
