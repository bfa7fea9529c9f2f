"""Why are the streamed forms slower inside bench.py than in scratch/kbench.py?  Same call, different context."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import backend, chebyshev
from bodge_amd.solver import DeviceSolver, VEC_RADEMACHER, VEC_Z4

shape = [1000, 1000, 1]
def make(model):
    system = bench.build_system(shape, model)
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    s = DeviceSolver(indptr, indices, data)
    s.set_lattice_shape(shape)
    return s, scale

def show(tag, dev, fn):
    fn(); fn()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    p = dev.perf()
    print(f"{tag:40s} wall {dt*1e3:7.3f} ms  window {p['window_ms']:7.3f}  launches {p['launches']} kernel_ms/launch {p['kernel_ms']/p['launches']:.4f} grid {p['grid']} streams {p['streams']} lanes {p['lanes_per_row']}", flush=True)

order = sys.argv[1] if len(sys.argv) > 1 else "other_first"
if order == "main_first":
    main, mscale = make("swave")
    show("main swave moments", main, lambda: main.moments_random(mscale, 40, 8, seed=0))
dev, scale = make("potential")
for steps in (20, 63):
    show(f"potential dots_random {steps}", dev, lambda: dev.dots_random(scale, steps, 8, seed=0))
    show(f"potential moments_random {steps}", dev, lambda: dev.moments_random(scale, 2 * steps, 8, seed=0))
    with backend.options(BODGE_AMD_STREAMED_SHARE="0"):
        show(f"potential moments_random {steps} noshare", dev, lambda: dev.moments_random(scale, 2 * steps, 8, seed=0))
