import os, sys, time, cProfile, pstats, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bodge_amd as ba
warnings.simplefilter("ignore")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lat = ba.CubicLattice((L, L, 1))
s = ba.Hamiltonian(lat)
with s as (H, D):
    H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3); D.set_sites(-0.1 * ba.jσ2); H.set_bonds(-1.0 * ba.σ0)
if os.environ.get("WARM"): s.free_energy(0.5)
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter(); f = s.free_energy(0.0); dt = time.perf_counter() - t
pr.disable()
print(f"free_energy(0.0) = {f:.10f} in {dt:.2f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
