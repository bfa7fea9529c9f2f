"""Single-site LDOS on large lattices with and without the band-limited sweep."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
for shape, site in (([1000, 1000, 1], (500, 500, 0)), ([100000, 1, 1], (50000, 0, 0)), ([100, 100, 100], (50, 50, 50))):
    system = bench.build_system(shape)
    energies = list(np.linspace(-0.3, 0.3, 13))
    system.ldos(site, [0.0, 0.5])
    out = {}
    for env in ({}, {"BODGE_AMD_NO_BAND": "1"}):
        os.environ.update(env)
        t0 = time.perf_counter(); rho = system.ldos(site, energies); dt = time.perf_counter() - t0
        for k in env: del os.environ[k]
        out[bool(env)] = rho
        print(f"{shape} {'full sweep' if env else 'band      '}: {dt:.3f} s", flush=True)
    print("   max |difference| =", np.abs(out[False] - out[True]).max(), flush=True)
