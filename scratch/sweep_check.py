"""Two-steps-per-sweep kernel (sweep.hpp): agreement with the one-step kernels and the oracle on
small and full-size lattices, then launch times for the hint / segment variants."""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bodge_amd as ba
from bodge_amd import chebyshev
from bodge_amd.solver import DeviceSolver
from oracle import cheb_ref


def env(**kw):
    class E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            os.environ.update({k: str(v) for k, v in kw.items()})
        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    return E()


def build(shape, kind):
    lat = ba.CubicLattice(shape)
    sysm = ba.Hamiltonian(lat)
    with sysm as (H, D):
        if kind == "swave":
            H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3); D.set_sites(-0.1 * ba.jσ2); H.set_bonds(-1.0 * ba.σ0)
        elif kind == "peierls":
            pairs = lat.bond_array(axis=0, coords=True)
            ph = np.where(pairs[:, 1, 0] > pairs[:, 0, 0], np.exp(0.3j), np.exp(-0.3j))
            H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3); D.set_sites(-0.1 * ba.jσ2)
            H.set_bonds(-ph[:, None, None] * ba.σ0, axis=0); H.set_bonds(-1.0 * ba.σ0, axis=1)
        elif kind == "dwave":
            pairs = lat.bond_array(coords=True)
            H.set_sites(3.0 * ba.σ0); H.set_bonds(-1.0 * ba.σ0)
            D.set_bonds(-0.1 * ba.dwave()(pairs[:, 0], pairs[:, 1]))
        elif kind == "junction":  # position-dependent blocks: S / F / S along x, Zeeman in the middle
            x = np.arange(lat.size) // (shape[1] * shape[2])
            onsite = np.where((x > shape[0] // 3)[:, None, None] & (x < 2 * shape[0] // 3)[:, None, None],
                              0.5 * ba.σ0 + 1.5 * ba.σ3, -0.5 * ba.σ0)
            gap = np.where(((x <= shape[0] // 3) | (x >= 2 * shape[0] // 3))[:, None, None], -1.0 * ba.jσ2, 0 * ba.jσ2)
            H.set_sites(onsite); D.set_sites(gap); H.set_bonds(-1.0 * ba.σ0)
        elif kind == "periodic":
            H.set_sites(3.0 * ba.σ0); D.set_sites(-0.1 * ba.jσ2); H.set_bonds(-1.0 * ba.σ0); H.set_edges(-1.0 * ba.σ0)
    return sysm


ok = True
for shape, kind, vk in [((64, 48, 1), "swave", 0), ((40, 100, 1), "peierls", 1), ((33, 61, 1), "dwave", 0),
                        ((48, 50, 1), "junction", 0), ((30, 30, 1), "periodic", 0), ((16, 1, 40), "swave", 0)]:
    sysm = build(shape, kind)
    bsr = sysm.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    n = bsr.shape[0]
    for steps, vectors in [(8, 8 if vk == 0 else 4), (7, 3)]:
        ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(n, 5, range(vectors), vk))
        with DeviceSolver.from_hamiltonian(sysm) as dev:
            with env(BODGE_AMD_SWEEP=1):
                got = dev.dots_random(scale, steps, vectors, seed=5, kind=vk)
                perf = dev.perf()
            with env(BODGE_AMD_SWEEP=0):
                one = dev.dots_random(scale, steps, vectors, seed=5, kind=vk)
        err = max(np.abs(got[0] - ref[0]).max(), np.abs(got[1] - ref[1]).max()) / n
        err1 = max(np.abs(got[0] - one[0]).max(), np.abs(got[1] - one[1]).max()) / n
        good = err < 1e-12 and err1 < 1e-12 and (perf["steps_per_launch"] == 2) == (kind != "periodic")
        ok &= good
        print(f"{shape} {kind:9s} steps={steps} R={vectors}: sweep={perf['steps_per_launch']} launches={perf['launches']} "
              f"|sweep-oracle|/4N={err:.1e} |sweep-onestep|/4N={err1:.1e} {'ok' if good else 'FAIL'}", flush=True)

sysm = build((1000, 1000, 1), "swave")
indptr, indices, data = sysm.bsr_arrays()
scale = chebyshev.spectral_bound(indptr, data)
with DeviceSolver(indptr, indices, data) as dev:
    dev.set_lattice_shape((1000, 1000, 1))
    with env(BODGE_AMD_SWEEP=0):
        one = dev.dots_random(scale, 33, 8, seed=0)
    got = dev.dots_random(scale, 33, 8, seed=0)
    perf = dev.perf()
    n = 4e6
    err = max(np.abs(got[0] - one[0]).max(), np.abs(got[1] - one[1]).max()) / n
    print(f"1000x1000 33 steps: sweep={perf['steps_per_launch']} launches={perf['launches']} grid={perf['grid']} |diff|/4N={err:.1e}", flush=True)
    ok &= err < 1e-12 and perf["steps_per_launch"] == 2

    def timeit(label, **kw):
        with env(**kw):
            dev.dots_random(scale, 16, 8, seed=0)
            dev.dots_random(scale, 256, 8, seed=0)
            p = dev.perf()
        per_launch = p["kernel_ms"] / p["launches"] * 1e3
        steps_s = 8 * 256 / (p["kernel_ms"] * 1e-3)
        print(f"  {label:40s} {per_launch:7.1f} us/launch  {p['bytes_per_launch'] / per_launch / 1e6:6.2f} TB/s  {steps_s / 1e3:6.1f} k vector-steps/s (kernel time)", flush=True)

    timeit("one-step dictionary kernel", BODGE_AMD_SWEEP=0)
    for alt in (0, 1):
        for stream in range(8):
            timeit(f"sweep alternate={alt} stream={stream}", BODGE_AMD_SWEEP_STREAM=stream, BODGE_AMD_ALTERNATE=alt)
    if "--segments" in sys.argv:
        for segs in (8, 12, 16, 20, 24, 32, 48):
            timeit(f"sweep segments={segs}", BODGE_AMD_SWEEP_SEGMENTS=segs)
        timeit("sweep 1 workgroup/CU", BODGE_AMD_BLOCKS_PER_CU=1)
print("ALL OK" if ok else "FAILED")
