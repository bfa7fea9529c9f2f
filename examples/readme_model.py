"""The Bodge README model (s-wave superconductor with a Zeeman field on a square lattice) through
bodge_amd: the calls are the reference's, the arithmetic runs on the GPU.

    python3 examples/readme_model.py [L]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from bodge_amd import *  # noqa: F401,F403  (same public names as `from bodge import *`)

L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lattice = CubicLattice((L, L, 1))
system = Hamiltonian(lattice)
t0 = time.perf_counter()
with system as (H, Δ):
    for i in lattice.sites():
        H[i, i] = 3.0 * σ0 - 0.05 * σ3
        Δ[i, i] = -0.1 * jσ2
    for i, j in lattice.bonds():
        H[i, j] = -1.0 * σ0
print(f"{L}x{L} lattice assembled through the dict API in {time.perf_counter() - t0:.2f} s")

for temperature in (0.0, 0.1, 0.5):
    t0 = time.perf_counter()
    value = system.free_energy(temperature)
    print(f"free_energy({temperature}) = {value:.10f}   [{time.perf_counter() - t0:.2f} s]")

t0 = time.perf_counter()
energies = list(np.linspace(-0.3, 0.3, 13))
rho = system.ldos((L // 2, L // 2, 0), energies)
print(f"ldos at the centre, 13 energies [{time.perf_counter() - t0:.2f} s]:", np.round(rho, 4))

t0 = time.perf_counter()
print(f"excitation gap = {system.lowest_eigenvalues(1)[0]:.6f}   [{time.perf_counter() - t0:.2f} s]")

if 4 * L * L <= 2048:
    t0 = time.perf_counter()
    E, v = system.diagonalize()
    print(f"diagonalize(): {E.size} positive eigenvalues, min {E.min():.6f}, eigenvectors {v.shape}   [{time.perf_counter() - t0:.2f} s]")
