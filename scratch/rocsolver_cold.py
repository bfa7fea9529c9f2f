"""Where the first rocSOLVER use on a fresh box spends its time (run once per box):
file read, dlopen, handle creation, first / second dsyevd and zheevd at the ladder sizes."""
import ctypes, os, sys, time

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
t00 = time.time()


def stamp(msg):
    print(f"[{time.time() - t00:7.1f}s] {msg}", flush=True)


os.environ["BODGE_AMD_TRACE"] = "1"
from bodge_amd import solver as _solver

_solver.prefetch_dense_library()  # background read by 16 threads inside the library; eigh waits for it
stamp("prefetch started")

import numpy as np
import bodge_amd as ba
import systems

stamp("imports done")
for name, env in [("chain300", "rocsolver"), ("chain300", "rocsolver"), ("swave30_zeeman", ""), ("swave30_zeeman", ""),
                  ("peierls30", ""), ("peierls30", "")]:
    if env:
        os.environ["BODGE_AMD_EIGH"] = env
    else:
        os.environ.pop("BODGE_AMD_EIGH", None)
    spec = systems.CATALOG[name]
    system = spec["build"](ba, **spec["kwargs"])
    stamp(f"{name}: built")
    solver = system._solver()
    stamp(f"{name}: uploaded")
    for vectors in (False, True):
        t0 = time.time()
        w, z = solver.eigh(vectors=vectors)
        stamp(f"{name}: eigh(vectors={vectors}) {time.time() - t0:.2f} s  finite={np.isfinite(w).all() and (z is None or np.isfinite(z).all())}")
