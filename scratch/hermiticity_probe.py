"""Device Hermiticity test of a 10^6-site matrix: wall time of the `with` exit and (under rocprofv3 --pmc) its traffic."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bodge_amd as ba

lattice = ba.CubicLattice((1000, 1000, 1))
system = ba.Hamiltonian(lattice)
system.hermiticity_check = "device"
t0 = time.perf_counter()
with system as (H, D):
    H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3)
    D.set_sites(-0.1 * ba.jσ2)
    H.set_bonds(-1.0 * ba.σ0)
print(f"with-block incl. upload and device check: {time.perf_counter() - t0:.3f} s", flush=True)
solver = system._solver()
for _ in range(3):
    t0 = time.perf_counter(); d = solver.hermiticity_defect(); dt = time.perf_counter() - t0
    print(f"hermiticity_defect = {d:.1e} in {dt * 1e3:.2f} ms (matrix {solver.dim // 4 * 5 * 256 / 1e9:.2f} GB)", flush=True)
