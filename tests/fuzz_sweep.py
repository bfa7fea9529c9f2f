"""Random small lattices / step counts / vector counts / knobs: stencil kernels (forced) against the one-step
kernels on the same vectors.  Prints one line per failure; exit code 1 if any."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]  # (also importable outside pytest)
import numpy as np
import bodge_amd as ba
from bodge_amd import backend, chebyshev
from bodge_amd.solver import DeviceSolver, VEC_RADEMACHER, VEC_Z4

def run(seed: int = 0, n_cases: int = 100, size: str | None = None, lanczos: bool = True) -> int:
    """Returns the number of failing cases (each printed).  Library switches are set through
    backend.options (bdg_set_option) and removed again, never through os.environ."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return _run(seed, n_cases, size, lanczos)


def _run(seed, n_cases, size, lanczos) -> int:
    rng = np.random.default_rng(seed)
    failures = 0
    t_start = time.time()
    for case in range(n_cases):
        three_d = rng.random() < 0.25
        if three_d:
            big3 = size == "large"
            shape = (int(rng.integers(8, 30 if big3 else 14)), int(rng.integers(5, 30 if big3 else 9)), int(rng.integers(5, 30 if big3 else 9)))
        else:
            big = size == "large"
            shape = (int(rng.integers(8, 120 if big else 40)), int(rng.integers(24, 500 if big else 90)), 1)
            if rng.random() < 0.3:
                shape = (shape[0], 1, shape[1])
        lat = ba.CubicLattice(shape)
        s = ba.Hamiltonian(lat)
        model = rng.choice(["uniform", "disorder", "texture", "complex", "periodic", "bond_disorder", "bond_phases"])
        with s as (H, D):
            if model == "disorder":  # > 256 distinct real diagonal blocks: the sweep that streams the on-site blocks
                H.set_sites(rng.normal(size=(lat.size, 1, 1)) * ba.σ0 + 0.1 * rng.normal(size=(lat.size, 1, 1)) * ba.σ3)
            elif model == "texture":  # the same with complex diagonal blocks (σ2 component)
                th, ph = rng.uniform(0, np.pi, (lat.size, 1, 1)), rng.uniform(0, 2 * np.pi, (lat.size, 1, 1))
                H.set_sites(3.0 * ba.σ0 - 0.3 * (np.sin(th) * np.cos(ph) * ba.σ1 + np.sin(th) * np.sin(ph) * ba.σ2 + np.cos(th) * ba.σ3))
            else:
                H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3)
            D.set_sites(-0.1 * ba.jσ2)
            H.set_bonds(-1.0 * ba.σ0)
            if model == "complex":
                pairs = lat.bond_array(axis=0, coords=True)
                phase = np.where(pairs[:, 1, 0] > pairs[:, 0, 0], np.exp(0.3j), np.exp(-0.3j))
                H.set_bonds(-phase[:, None, None] * ba.σ0, axis=0)
            if model in ("bond_disorder", "bond_phases"):
                # a different spin-diagonal hopping block on every bond (round 4: with a Peierls phase of its own on every bond,
                # on top of a texture - complex site records); the on-site terms position dependent too
                idx = lat.bond_array()
                lo, hi = idx.min(axis=1), idx.max(axis=1)  # the same amplitude both ways
                t = (0.8 + 0.4 * ((lo * 7919 + hi * 104729) % 1009) / 1009.0)[:, None, None]
                dt = (0.1 * ((lo * 31 + hi * 17) % 101) / 101.0)[:, None, None]
                if model == "bond_phases":
                    θ = 2 * np.pi * ((lo * 271 + hi * 65537) % 997) / 997.0
                    t = t * np.exp(1j * np.where(idx[:, 1] > idx[:, 0], θ, -θ))[:, None, None]
                    th, ph = rng.uniform(0, np.pi, (lat.size, 1, 1)), rng.uniform(0, 2 * np.pi, (lat.size, 1, 1))
                    H.set_sites(3.0 * ba.σ0 - 0.3 * (np.sin(th) * np.cos(ph) * ba.σ1 + np.sin(th) * np.sin(ph) * ba.σ2 + np.cos(th) * ba.σ3))
                else:
                    H.set_sites((3.0 + rng.uniform(-0.5, 0.5, lat.size))[:, None, None] * ba.σ0)
                H.set_bonds(-t * ba.σ0 + dt * ba.σ3)
            if (model == "periodic" or (model in ("disorder", "texture", "bond_disorder", "bond_phases") and rng.random() < 0.3)) and not three_d:
                H.set_edges(-0.7 * ba.σ0, axis=0)
                H.set_edges(-0.7 * ba.σ0, axis=1 if shape[1] > 1 else 2)
        indptr, indices, data = s.bsr_arrays()
        scale = chebyshev.spectral_bound(indptr, data)
        steps = int(rng.integers(1, 14))
        vectors = int(rng.integers(1, 23))
        kind = VEC_Z4 if rng.random() < 0.3 else VEC_RADEMACHER
        env = {}
        if rng.random() < 0.4: env["BODGE_AMD_SWEEP_LANES"] = str(rng.choice([2, 4]))
        if rng.random() < 0.2: env["BODGE_AMD_SWEEP_STEPS"] = "2"
        if rng.random() < 0.2: env["BODGE_AMD_SWEEP_SEGMENTS"] = str(int(rng.integers(1, 5)))
        if rng.random() < 0.2: env["BODGE_AMD_SWEEP_ZIGZAG"] = "0"
        if rng.random() < 0.2: env["BODGE_AMD_ALTERNATE"] = "0"
        if rng.random() < 0.3: env["BODGE_AMD_STREAMS"] = str(rng.choice([1, 2, 3, 4]))  # batches side by side on that many streams
        if rng.random() < 0.15: env["BODGE_AMD_KEEP_LAST"] = "1"
        # the sweeps of a chunk in one launch (cheb_march3): tickets, fixed units, or one launch per sweep for all lane groups
        if rng.random() < 0.3: env["BODGE_AMD_MARCH"] = str(rng.choice([1, 2, 3]))
        # a third of the cases start from unit vectors (LDOS): the stencil kernels then advance a band of planes only
        unit = rng.random() < 0.33
        if unit:
            steps = int(rng.integers(1, 40))
            near = rng.random() < 0.5  # start sites close together (narrow band) or anywhere
            x0 = int(rng.integers(0, shape[0]))
            xs = np.clip(x0 + rng.integers(-2, 3, vectors), 0, shape[0] - 1) if near else rng.integers(0, shape[0], vectors)
            rows = np.array([4 * ((int(x) * shape[1] + int(rng.integers(0, shape[1]))) * shape[2] + int(rng.integers(0, shape[2])))
                             + int(rng.integers(0, 4)) for x in xs])
        with DeviceSolver(indptr, indices, data) as dev:
            dev.set_lattice_shape(shape)
            call = (lambda: dev.dots_unit(scale, steps, rows)) if unit else (lambda: dev.dots_random(scale, steps, vectors, seed=case, kind=kind))
            with backend.options(BODGE_AMD_SWEEP="0"):
                one = call()
            if rng.random() < 0.3:  # (leave the buffers full of another run's vectors)
                with backend.options(BODGE_AMD_SWEEP="1"):
                    dev.dots_random(scale, 4, 9, seed=1)
            with backend.options(BODGE_AMD_SWEEP="1", **env):
                got = call()
                perf = dev.perf()
        n = 4 * lat.size
        err = max(np.abs(got[0] - one[0]).max(), np.abs(got[1] - one[1]).max()) / (1.0 if unit else n)
        tag = (f"case {case}: {shape} {model} steps={steps} vectors={vectors} {'unit' if unit else f'kind={kind}'} {env} -> steps/launch "
               f"{perf['steps_per_launch']} rolling {perf['rolling']} onsite-streamed {perf['onsite_streamed']} streams {perf['streams']} persistent {perf['persistent']}")
        if not err <= 1e-12:
            failures += 1
            print("FAIL", tag, "err", err, flush=True)
        elif case % 25 == 0:
            print("ok  ", tag, f"err {err:.1e}  [{time.time() - t_start:.0f} s]", flush=True)
    print(f"{n_cases} cases, {failures} failures, {time.time() - t_start:.0f} s")
    return failures


if __name__ == "__main__":
    sys.exit(1 if run(int(os.environ.get("FUZZ_SEED", "0")), int(os.environ.get("FUZZ_CASES", "100")),
                      os.environ.get("FUZZ_SIZE"), os.environ.get("FUZZ_LANCZOS", "1") == "1") else 0)
