import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from bodge_amd import observables
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
system = bench.build_system([L, L, 1])
solver = system._solver()
scale = observables._scale_of(system)
rows = observables._electron_rows(system.shape[0])[2048:2048 + 64]
os.environ["BODGE_AMD_NO_BAND"] = "1"
solver.moments_unit(scale, 64, rows)
t0 = time.perf_counter(); solver.moments_unit(scale, 2048, rows); dt = time.perf_counter() - t0
p = solver.perf()
print(f"{L}x{L}: wall {dt / 1024 * 1e6:.2f} us per launch, kernel (events) {p['kernel_ms'] / p['launches'] * 1e3:.2f} us, rl {p['lanes_per_row']} grid {p['grid']}", flush=True)
