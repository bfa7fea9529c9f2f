"""Wall time per vector-step of dots_random(64 vectors) as a function of the batch width."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver

model = sys.argv[1] if len(sys.argv) > 1 else "swave"
shapes = [[100,100,1],[200,200,1],[300,300,1],[400,400,1],[600,600,1],[800,800,1],[1000,1000,1]] if model == "swave" else [[20,20,20],[40,40,40],[64,64,64],[100,100,100]]
steps, vectors = 64, 64
for shape in shapes:
    system = bench.build_system(shape, model)
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    solver = DeviceSolver(indptr, indices, data)
    solver.set_lattice_shape(shape)
    row = []
    for width in ("auto", 8, 16, 32, 64):
        if width == "auto":
            os.environ.pop("BODGE_AMD_BATCH", None)
        else:
            os.environ["BODGE_AMD_BATCH"] = str(width)
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            solver.dots_random(scale, steps, vectors, seed=rep)
            best = min(best, time.perf_counter() - t0)
        row.append(f"{width}: {best / (steps * vectors) * 1e6:6.2f} us")
    os.environ.pop("BODGE_AMD_BATCH", None)
    print(f"{'x'.join(map(str, shape)):>14s}  " + "   ".join(row), flush=True)
    solver.close()
