"""Default kernel routing at lattice sizes around its thresholds: stencil kernels (as chosen) against the
one-step kernels on the same vectors, real and complex models, ragged shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import bodge_amd as ba
from bodge_amd.solver import DeviceSolver, VEC_RADEMACHER, VEC_Z4

cases = [((388, 387, 1), "swave"), ((387, 1, 389), "complex"), ((671, 673, 1), "swave"), ((300, 1501, 1), "periodic"),
         ((1400, 330, 1), "complex"), ((84, 84, 84), "dwave"), ((86, 85, 84), "dwave"), ((61, 101, 99), "swave"), ((2003, 301, 1), "swave")]
bad = 0
for shape, model in cases:
    lat = ba.CubicLattice(shape)
    s = ba.Hamiltonian(lat)
    with s as (H, D):
        H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3)
        H.set_bonds(-1.0 * ba.σ0)
        if model == "dwave":
            pairs = lat.bond_array(coords=True)
            D.set_bonds(-0.1 * ba.dwave()(pairs[:, 0], pairs[:, 1]))
        else:
            D.set_sites(-0.1 * ba.jσ2)
        if model == "complex":
            pairs = lat.bond_array(axis=0, coords=True)
            phase = np.where(pairs[:, 1, 0] > pairs[:, 0, 0], np.exp(0.3j), np.exp(-0.3j))
            H.set_bonds(-phase[:, None, None] * ba.σ0, axis=0)
        if model == "periodic":
            H.set_edges(-0.8 * ba.σ0)
    dev = s._solver()
    scale = 1.01 * s.gershgorin_bound()
    for kind, vectors in ((VEC_RADEMACHER, 11), (VEC_Z4, 5)):
        got = dev.dots_random(scale, 10, vectors, seed=4, kind=kind)
        p = dev.perf()
        os.environ["BODGE_AMD_SWEEP"] = "0"
        one = dev.dots_random(scale, 10, vectors, seed=4, kind=kind)
        del os.environ["BODGE_AMD_SWEEP"]
        err = max(np.abs(got[0] - one[0]).max(), np.abs(got[1] - one[1]).max()) / (4 * lat.size)
        flag = "ok " if err <= 1e-12 else "BAD"
        bad += flag == "BAD"
        print(f"{flag} {shape} {model} kind={kind} vectors={vectors}: steps/launch {p['steps_per_launch']} lanes {p['lanes_per_row']} rolling {p['rolling']} err {err:.1e}", flush=True)
    s._solver().close()
print("failures:", bad)
sys.exit(1 if bad else 0)
