import sys, time
sys.path.insert(0, ".")
import numpy as np, bench
for shape in ([200,200,1],[1000,1000,1]):
    system = bench.build_system(shape)
    t=time.time(); gap = system.lowest_eigenvalues(1, tol=1e-5); dt=time.time()-t
    print(shape, gap, f"{dt:.2f}s", flush=True)
