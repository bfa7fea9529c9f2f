import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bodge_amd as ba, systems
for algo in ("jacobi", "evd", "evj"):
    os.environ["BODGE_AMD_EIGH"] = algo
    for name in ("complex235", "barrier", "random357", "snf", "swave20"):
        s = systems.CATALOG[name]["build"](ba, **systems.CATALOG[name]["kwargs"])
        dense = np.asarray(s.matrix("dense"))
        t = time.time()
        w, z = s._solver().eigh(vectors=True)
        dt = time.time() - t
        nan = int(np.isnan(z).sum())
        res = np.abs(dense @ np.nan_to_num(z) - np.nan_to_num(z) * w).max()
        ref = np.linalg.eigvalsh(dense)
        nanrows = np.where(np.isnan(z).any(axis=1))[0]
        print(algo, name, dense.shape[0], f"{dt:.2f}s nan={nan} rows={nanrows[:3]}..{nanrows[-3:] if nan else ''} res={res:.2e} dw={np.abs(w-ref).max():.2e}", flush=True)
