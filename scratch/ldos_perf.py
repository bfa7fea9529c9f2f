import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from bodge_amd import observables
system = bench.build_system([1000, 1000, 1])
solver = system._solver()
scale = observables._scale_of(system)
rows = 4 * system.lattice[(500, 500, 0)] + np.arange(4)
for M in (512, 4000):
    solver.moments_unit(scale, 64, rows)
    t0 = time.perf_counter(); mu = solver.moments_unit(scale, M, rows); dt = time.perf_counter() - t0
    p = solver.perf()
    print(M, f"{dt:.3f} s", f"{dt / (M / 2) * 1e3:.3f} ms/step", {k: p[k] for k in ("kernel_ms", "launches", "lanes_per_row", "real_arithmetic", "dict_blocks", "grid", "strip_rows")}, flush=True)
t0 = time.perf_counter(); solver.dots_random(scale, 256, 8); print("random 8:", (time.perf_counter() - t0) / 256 * 1e3, "ms/step")
