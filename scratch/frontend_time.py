"""Where a from-scratch free_energy on a 10^6-site lattice spends its wall time (host front end vs device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bodge_amd as ba
from bodge_amd.solver import DeviceSolver

def stamp(label, t0):
    t1 = time.perf_counter()
    print(f"{label:46s} {t1 - t0:7.3f} s", flush=True)
    return t1

shape = tuple(int(v) for v in os.environ.get("FT_LATTICE", "1000,1000,1").split(","))
import bodge_amd.backend as backend
backend.load()
t = time.perf_counter()
for rep in range(2):
    print(f"--- pass {rep} {shape}")
    t = time.perf_counter(); t_all = t
    lat = ba.CubicLattice(shape)
    sysm = ba.Hamiltonian(lat); t = stamp("Hamiltonian(lattice)  [skeleton]", t)
    with sysm as (H, D):
        H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3); D.set_sites(-0.1 * ba.jσ2); H.set_bonds(-1.0 * ba.σ0)
        t = stamp("  queue terms (bond_array)", t)
    t = stamp("with-block exit [fill + Hermiticity]", t)
    f = sysm.free_energy(0.5, moments=64, vectors=8); t = stamp("first free_energy(0.5) [upload+dict+stencil+run]", t)
    f = sysm.free_energy(0.5, moments=64, vectors=8); t = stamp("second free_energy(0.5)", t)
    print(f"total {time.perf_counter() - t_all:.3f} s   F = {f:.6f}")
    import cProfile, pstats
    if rep == 1 and os.environ.get("FT_PROFILE"):
        pr = cProfile.Profile(); pr.enable()
        sysm2 = ba.Hamiltonian(lat)
        with sysm2 as (H, D):
            H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3); D.set_sites(-0.1 * ba.jσ2); H.set_bonds(-1.0 * ba.σ0)
        sysm2.free_energy(0.5, moments=64, vectors=8)
        pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(22)
