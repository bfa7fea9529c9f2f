"""Wall time of one short dots_random call (the driver's --steps 20) against its kernel time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import bodge_amd as ba
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver

lat = ba.CubicLattice((1000, 1000, 1))
sysm = ba.Hamiltonian(lat)
with sysm as (H, D):
    H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3); D.set_sites(-0.1 * ba.jσ2); H.set_bonds(-1.0 * ba.σ0)
dev = sysm._solver()
scale = 1.01 * sysm.gershgorin_bound()
steps = int(os.environ.get("CO_STEPS", "20"))
for _ in range(30):
    dev.dots_random(scale, 64, 8, seed=0)   # settle the clock
walls, kernels = [], []
for rep in range(20):
    t = time.perf_counter()
    dev.dots_random(scale, steps, 8, seed=0)
    walls.append(time.perf_counter() - t)
    kernels.append(dev.perf()["kernel_ms"] * 1e-3)
w, k = np.median(walls), np.median(kernels)
print(f"steps={steps}: wall {w*1e3:.3f} ms, recurrence kernels {k*1e3:.3f} ms, other {1e3*(w-k):.3f} ms ({100*(w-k)/w:.1f} %), {8*steps/w/1e3:.1f} k vector-steps/s")
