"""Interleaved A/B timing of sweep-kernel variants selected by environment variables
(1000x1000 s-wave+Zeeman, 8 vectors, 256 steps per call; median kernel time per launch)."""
import os, sys, statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bodge_amd as ba
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver

shape = tuple(int(v) for v in os.environ.get("AB_LATTICE", "1000,1000,1").split(","))
lat = ba.CubicLattice(shape)
sysm = ba.Hamiltonian(lat)
with sysm as (H, D):
    if os.environ.get("AB_MODEL", "swave") == "dwave":
        pairs = lat.bond_array(coords=True)
        H.set_sites(3.0 * ba.σ0); H.set_bonds(-1.0 * ba.σ0)
        D.set_bonds(-0.1 * ba.dwave()(pairs[:, 0], pairs[:, 1]))
    else:
        H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3); D.set_sites(-0.1 * ba.jσ2); H.set_bonds(-1.0 * ba.σ0)
indptr, indices, data = sysm.bsr_arrays()
scale = chebyshev.spectral_bound(indptr, data)
NV = int(os.environ.get("AB_VECTORS", "8"))
variants = []
for spec in sys.argv[1:]:
    label, _, envs = spec.partition(":")
    variants.append((label, dict(kv.split("=") for kv in envs.split(",") if kv)))
with DeviceSolver(indptr, indices, data) as dev:
    dev.set_lattice_shape(shape)
    ref = None
    times = {label: [] for label, _ in variants}
    info = {}
    for rep in range(5):
        for label, env in variants:
            os.environ.update(env)
            d, e = dev.dots_random(scale, 256, NV, seed=0)
            p = dev.perf()
            for k in env:
                del os.environ[k]
            if ref is None:
                ref = (d, e)
            err = max(np.abs(d - ref[0]).max(), np.abs(e - ref[1]).max()) / (4.0 * lat.size)
            assert err < 1e-12, (label, err)
            times[label].append(p["kernel_ms"] / p["launches"] * 1e3)
            info[label] = p
    for label, _ in variants:
        t = statistics.median(times[label])
        p = info[label]
        total_us = p["kernel_ms"] * 1e3
        print(f"{label:34s} [{NV * 256 / total_us * 1e3:7.1f} k vector-steps/s over the whole call] {t:7.1f} us/launch (min {min(times[label]):6.1f})  {p['bytes_per_launch'] / t / 1e6:5.2f} TB/s alg  "
              f"{p['vectors_per_launch'] * p['steps_per_launch'] / t * 1e3:6.1f} k vector-steps/s  grid={p['grid']} rolling={p['rolling']} lanes={p['lanes_per_row']} "
              f"steps/launch={p['steps_per_launch']}", flush=True)
