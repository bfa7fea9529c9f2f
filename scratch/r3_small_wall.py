"""Wall time of API calls on small lattices with the batches of a call on one stream and side by side on two."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bodge_amd as ba
from bodge_amd import backend

for shape in ((32, 32, 1), (64, 64, 1), (100, 100, 1)):
    lat = ba.CubicLattice(shape)
    s = ba.Hamiltonian(lat)
    with s as (H, D):
        H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3)
        D.set_sites(-0.1 * ba.jσ2)
        H.set_bonds(-1.0 * ba.σ0)
    site = (shape[0] // 2, shape[1] // 2, 0)
    energies = np.linspace(-0.3, 0.3, 13)
    for label, call in (("free_energy(0.1) [Chebyshev, exact trace]", lambda: s.free_energy(0.1, method="chebyshev", trace="exact", moments=256)),
                        ("ldos at 13 energies", lambda: s.ldos(site, energies))):
        out = {}
        for name, env in (("one stream", {"BODGE_AMD_STREAMS": "1"}), ("default", {})):
            with backend.options(**env):
                call()
                t0 = time.perf_counter()
                value = call()
                out[name] = (time.perf_counter() - t0, value)
        a, b = out["one stream"], out["default"]
        same = np.allclose(a[1], b[1], rtol=0, atol=0)
        print(f"{shape} {label:44s} one stream {a[0]*1e3:9.2f} ms   side by side {b[0]*1e3:9.2f} ms   x{a[0]/b[0]:.2f}   identical {same}", flush=True)
