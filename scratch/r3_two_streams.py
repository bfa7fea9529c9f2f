"""Do two lane groups run faster side by side (two handles = two streams and buffer sets on one GPU, one host
thread each) than back to back on one stream?  1000x1000 s-wave, 4 real vectors per lane group, K7b."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver

shape = [1000, 1000, 1]
system = bench.build_system(shape, "swave")
indptr, indices, data = system.bsr_arrays()
scale = chebyshev.spectral_bound(indptr, data)
devs = [DeviceSolver(indptr, indices, data) for _ in range(2)]
for d in devs:
    d.set_lattice_shape(shape)
    d.dots_random(scale, 64, 8, seed=0)

def one(steps, reps):
    t0 = time.perf_counter()
    for r in range(reps):
        devs[0].dots_random(scale, steps, 8, seed=r)
    return (time.perf_counter() - t0) / reps

def two(steps, reps):
    def work(i):
        for r in range(reps):
            devs[i].dots_random(scale, steps, 4, seed=r, first_id=4 * i)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t0 = time.perf_counter()
    for t in threads: t.start()
    for t in threads: t.join()
    return (time.perf_counter() - t0) / reps

for steps in (20, 63, 256):
    for rnd in range(3):
        a = one(steps, 20); b = two(steps, 20)
        print(f"steps {steps:4d}: one stream, 2 lane groups back to back {a*1e3:8.3f} ms ({8*steps/a/1e3:6.1f} k vsteps/s)   "
              f"two streams side by side {b*1e3:8.3f} ms ({8*steps/b/1e3:6.1f} k)   x{a/b:.3f}", flush=True)
