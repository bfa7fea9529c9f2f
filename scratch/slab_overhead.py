"""Cost of the slab machinery on one GPU: 100^3 d-wave as one matrix vs 8 slabs (same-process group)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver, SlabGroup
system = bench.build_system([100, 100, 100], "dwave")
indptr, indices, data = system.bsr_arrays()
scale = chebyshev.spectral_bound(indptr, data)
steps, vectors = 128, 8
with DeviceSolver.from_hamiltonian(system) as whole:
    whole.dots_random(scale, 8, vectors)
    t0 = time.perf_counter(); whole.dots_random(scale, steps, vectors); t_whole = time.perf_counter() - t0
print(f"whole matrix: {t_whole / steps * 1e6:.1f} us per step", flush=True)
for n_slabs in (2, 4, 8):
    with SlabGroup.from_hamiltonian(system, n_slabs) as group:
        group.dots_random(scale, 8, vectors)
        t0 = time.perf_counter(); group.dots_random(scale, steps, vectors); t = time.perf_counter() - t0
    print(f"{n_slabs} slabs on one GPU: {t / steps * 1e6:.1f} us per step ({t / t_whole:.2f}x)", flush=True)
