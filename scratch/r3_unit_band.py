"""Unit start vectors (LDOS) on the lattice-stencil kernels with a band of planes against the one-step kernels
with their band of rows: wall time of dots_unit on 1000x1000 s-wave and 100^3 d-wave."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import backend, chebyshev
from bodge_amd.solver import DeviceSolver

for shape, model in (((1000, 1000, 1), "swave"), ((100, 100, 100), "dwave")):
    system = bench.build_system(list(shape), model)
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    dev = DeviceSolver(indptr, indices, data)
    dev.set_lattice_shape(list(shape))
    lx, ly, lz = shape
    centre = 4 * ((lx // 2 * ly + ly // 2) * lz + lz // 2)
    line = np.array([4 * (((lx // 2 - 4 + i) * ly + ly // 2) * lz + lz // 2) for i in range(8)])
    spread = np.array([4 * ((int(lx * (i + 0.5) / 8) * ly + ly // 3) * lz + lz // 2) + (i % 4) for i in range(8)])
    for label, rows in (("1 site, centre", np.array([centre])), ("8 sites, adjacent planes", line), ("8 sites, spread over x", spread)):
        for steps in (64, 256, 1000):
            out = {}
            for name, env in (("one-step", {"BODGE_AMD_SWEEP": "0"}), ("stencil", {})):
                with backend.options(**env):
                    dev.dots_unit(scale, 8, rows)
                    t0 = time.perf_counter()
                    d, e = dev.dots_unit(scale, steps, rows)
                    out[name] = (time.perf_counter() - t0, dev.perf(), d, e)
            a, b = out["one-step"], out["stencil"]
            err = max(np.abs(a[2] - b[2]).max(), np.abs(a[3] - b[3]).max())
            print(f"{shape} {label:26s} {steps:5d} steps: one-step {a[0]*1e3:8.2f} ms ({a[1]['bytes_moved']/1e9:7.2f} GB)  "
                  f"stencil {b[0]*1e3:8.2f} ms ({b[1]['bytes_moved']/1e9:7.2f} GB, steps/launch {b[1]['steps_per_launch']}, rolling {b[1]['rolling']}, "
                  f"lanes {b[1]['lanes_per_row']})  x{a[0]/b[0]:.2f}  max diff {err:.1e}", flush=True)
    dev.close()
