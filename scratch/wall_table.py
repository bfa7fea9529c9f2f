"""End-to-end wall time of system.free_energy() / diagonalize() / ldos() through the Python API
(second call of each: device copy and host-side scalars cached, as in a parameter sweep), next to
the reference's route (dense LAPACK / SuperLU on the host) where that is feasible."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from oracle import dense_ref

def timed(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    return best, out

rows = []
for L, T in ((20, 0.1), (22, 0.1), (64, 0.1), (200, 0.5), (1000, 0.5)):  # 22x22: 4N = 1936, the largest the own eigensolver takes
    system = bench.build_system([L, L, 1])
    n = system.shape[0]
    t_first, f = timed(lambda: system.free_energy(T), reps=1)
    t_again, f = timed(lambda: system.free_energy(T))
    line = f"free_energy({T}) {L}x{L} (4N={n}): first call {t_first:.3f} s, repeated {t_again*1e3:.1f} ms, F = {f:.10f}"
    if n <= 2048:
        dense = np.asarray(system.matrix("dense"))
        t_ref, f_ref = timed(lambda: dense_ref.free_energy(dense, T), reps=1)
        line += f" | host dense: {t_ref:.2f} s, |dF|/|F| = {abs(f - f_ref) / abs(f_ref):.1e}"
    print(line, flush=True)
    if n <= 2048:
        t_diag, (e, v) = timed(lambda: system.diagonalize(), reps=1)
        t_ref, _ = timed(lambda: dense_ref.diagonalize(dense), reps=1)
        print(f"diagonalize() {L}x{L}: {t_diag:.3f} s (host dense: {t_ref:.2f} s), {e.size} eigenpairs", flush=True)
    if L <= 200:
        site = (L // 2, L // 2, 0)
        energies = list(np.linspace(-0.3, 0.3, 13))
        t_l, rho = timed(lambda: system.ldos(site, energies), reps=1)
        line = f"ldos(13 energies) {L}x{L}: {t_l:.3f} s"
        if n <= 20000:
            t_ref, rho_ref = timed(lambda: dense_ref.ldos(system.matrix("csc"), system.lattice[site], energies), reps=1)
            line += f" (host SuperLU: {t_ref:.2f} s), max |d rho| = {np.abs(rho - rho_ref).max():.1e}"
        print(line, flush=True)
