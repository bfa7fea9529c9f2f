"""Random small systems through the public API against the CPU oracle (dense LAPACK / SuperLU routes of
the reference, restated in oracle/dense_ref.py): free_energy (dense and Chebyshev, several T),
diagonalize, ldos, slab groups against the whole matrix, devices=[0, 0].  One line per failure."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]  # (also importable outside pytest)
import numpy as np
import bodge_amd as ba
import systems
from oracle import dense_ref, cheb_ref
from bodge_amd import backend
from bodge_amd.solver import DeviceSolver

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

    def check(ok, what, case, detail):
        nonlocal failures
        if not ok:
            failures += 1
            print(f"FAIL case {case} {what}: {detail}", flush=True)

    for case in range(n_cases):
        dims = rng.choice([1, 2, 3], p=[0.2, 0.5, 0.3])
        if size == "mid":  # 4N between 2300 and 4000: beyond the small-matrix shortcuts
            shape = (int(rng.integers(24, 32)), int(rng.integers(24, 32)), 1) if dims != 3 else (int(rng.integers(8, 11)), int(rng.integers(8, 11)), int(rng.integers(8, 10)))
        elif dims == 1:
            shape = (int(rng.integers(3, 120)), 1, 1)
        elif dims == 2:
            shape = (int(rng.integers(2, 16)), int(rng.integers(2, 16)), 1)
        else:
            shape = (int(rng.integers(2, 7)), int(rng.integers(2, 7)), int(rng.integers(2, 6)))
        kind = rng.choice(["random_periodic", "swave", "dwave", "junction"])
        if kind == "random_periodic":
            system = systems.random_periodic(ba, shape=shape, seed=int(rng.integers(1 << 30)))
        else:
            lat = ba.CubicLattice(shape)
            system = ba.Hamiltonian(lat)
            with system as (H, D):
                mu = float(rng.uniform(0.5, 3.5)); h = float(rng.uniform(0.0, 0.4)); gap = float(rng.uniform(0.05, 1.0))
                if kind == "junction":
                    x = np.arange(lat.size) // (shape[1] * shape[2])
                    mid = ((x > shape[0] // 3) & (x < 2 * shape[0] // 3))[:, None, None]
                    H.set_sites(np.where(mid, 0.5 * ba.σ0 + 1.5 * ba.σ3, -mu * ba.σ0))
                    D.set_sites(np.where(mid, 0 * ba.jσ2, -gap * ba.jσ2))
                else:
                    H.set_sites(mu * ba.σ0 - h * ba.σ3)
                    if kind == "swave":
                        D.set_sites(-gap * ba.jσ2)
                H.set_bonds(-1.0 * ba.σ0)
                if kind == "dwave" and len(lat.bond_array()):
                    pairs = lat.bond_array(coords=True)
                    D.set_bonds(-gap * ba.dwave()(pairs[:, 0], pairs[:, 1]))
                if rng.random() < 0.3 and len(lat.edge_array()):
                    H.set_edges(-0.6 * ba.σ0)
        n = system.shape[0]
        dense = np.asarray(system.matrix("dense"))
        tag = f"{shape} {kind} n={n}"
        # free energy.  Zero modes: the reference sums over the eigenvalues LAPACK returns in (0, inf] (hamiltonian.py:228-231,
        # 296-302), so each zero mode is in or out by the sign of its round-off and moves F by T ln 2; this package counts the
        # upper half of a symmetric spectrum (observables.py).  The restatement below counts by numpy's round-off: a difference
        # of k T ln 2 with |k| <= the number of zero modes is the same answer.
        w_all = np.linalg.eigvalsh(dense)
        zero_modes = int((np.abs(w_all) <= 1e-10 * max(1.0, np.abs(w_all).max())).sum())

        def same_free_energy(got, ref, T, tol):
            return any(abs(got - ref - k * T * np.log(2.0)) <= tol * max(1.0, abs(ref)) for k in range(-zero_modes, zero_modes + 1))

        for T in (0.0, float(rng.choice([0.05, 0.3, 1.0]))):
            ref = dense_ref.free_energy(dense, T)
            got = system.free_energy(T, method="dense")
            check(same_free_energy(got, ref, T, 1e-10), "free_energy dense", case, f"{tag} T={T} {got} vs {ref} ({zero_modes} zero modes)")
        if system.has_symmetric_spectrum(1e-12):
            T = 0.5
            ref = dense_ref.free_energy(dense, T)
            got = system.free_energy(T, method="chebyshev", trace="exact")
            check(same_free_energy(got, ref, T, 1e-9), "free_energy chebyshev", case, f"{tag} T={T} {got} vs {ref} ({zero_modes} zero modes)")
        # stochastic trace on the oracle's vectors (same counter-based generator), both vector kinds
        if system.has_symmetric_spectrum(1e-12):
            from bodge_amd.observables import free_energy_stochastic
            bsr = system.matrix("bsr")
            scale = cheb_ref.spectral_bound(bsr)
            R, M, sd = int(rng.integers(1, 40)), 2 * int(rng.integers(4, 40)), int(rng.integers(0, 1000))
            for vk, ok_kind in (("rademacher", cheb_ref.VEC_RADEMACHER), ("z4", cheb_ref.VEC_Z4)):
                ref = cheb_ref.free_energy_stochastic(bsr, 0.7, M, R, seed=sd, kind=ok_kind, scale=scale)
                got, _ = free_energy_stochastic(system, 0.7, moments=M, vectors=R, seed=sd, vector_kind=vk, scale=scale)
                check(abs(got - ref) <= 1e-10 * max(1.0, abs(ref)), "free_energy stochastic", case, f"{tag} {vk} R={R} M={M} seed={sd}: {got} vs {ref}")
            two = system.free_energy(0.7, method="chebyshev", trace="stochastic", moments=M, vectors=R, seed=sd, devices=[0, 0])
            one = system.free_energy(0.7, method="chebyshev", trace="stochastic", moments=M, vectors=R, seed=sd)
            check(abs(one - two) <= 1e-11 * max(1.0, abs(one)), "stochastic devices=[0,0]", case, f"{tag} {one} vs {two}")
        # unit start vectors (the LDOS / exact-trace building block): random rows with repeats, several batches
        if n <= 2500:
            bsr_u = system.matrix("bsr")
            scale_u = cheb_ref.spectral_bound(bsr_u)
            rows = rng.integers(0, n, size=int(rng.integers(1, 150)))
            steps_u = int(rng.integers(1, 12))
            with DeviceSolver.from_hamiltonian(system) as dev:
                got_u = dev.dots_unit(scale_u, steps_u, rows)
            ref_u = cheb_ref.recurrence_dots(bsr_u, scale_u, 2 * steps_u, cheb_ref.unit_block(n, rows))
            err_u = max(np.abs(got_u[0] - ref_u[0]).max(), np.abs(got_u[1] - ref_u[1]).max())
            check(err_u <= 1e-12, "dots_unit", case, f"{tag} rows={len(rows)} steps={steps_u} err {err_u}")
        # diagonalize
        E, X = system.diagonalize(format="raw")
        w = np.linalg.eigvalsh(dense)
        # (the reference keeps the eigenvalues in (0, inf] - hamiltonian.py:228-231 - so a zero mode is in or out by the
        # sign of its round-off, there as here: the levels above round-off must agree one by one, zero modes only in number)
        tol = 1e-10 * max(1.0, np.abs(w).max())
        Es = np.sort(E)
        wpos, zero_modes = w[w > tol], int((np.abs(w) <= tol).sum())
        check(len(Es[Es > tol]) == len(wpos) and np.abs(Es[Es > tol] - wpos).max(initial=0) <= tol and len(Es[Es <= tol]) <= zero_modes,
              "diagonalize values", case, f"{tag} {len(E)} vs {len(wpos)} + at most {zero_modes} zero modes")
        if len(E):
            res = np.abs(dense @ X - X * E[None, :]).max()
            check(res <= 1e-9 * max(1.0, np.abs(w).max()) and np.isfinite(X).all(), "diagonalize residual", case, f"{tag} {res}")
        # ldos
        if n <= 1200:
            site = tuple(int(rng.integers(0, s)) for s in shape)
            energies = np.linspace(-0.9, 0.9, 7) * float(rng.uniform(0.3, 1.0))
            ref = dense_ref.ldos(system.matrix("csc"), system.lattice[site], energies)
            got = system.ldos(site, energies)
            check(np.allclose(got, ref, rtol=1e-8, atol=1e-10), "ldos", case, f"{tag} site {site} max diff {np.abs(got - ref).max()}")
            other = tuple(int(rng.integers(0, s)) for s in shape)
            both = system.ldos([site, other, site], energies)  # several sites share the recurrence launches
            alone = system.ldos(other, energies)
            check(both.shape == (3, len(energies)) and np.allclose(both[0], got, rtol=1e-10, atol=1e-12) and np.allclose(both[2], got, rtol=1e-10, atol=1e-12)
                  and np.allclose(both[1], alone, rtol=1e-10, atol=1e-12), "ldos multi-site", case, f"{tag} {site} {other}")
            with backend.options(BODGE_AMD_NO_BAND="1"):
                whole = system.ldos(site, energies)
            check(np.allclose(whole, got, rtol=1e-10, atol=1e-12), "ldos band-limited vs whole", case, f"{tag} {np.abs(whole - got).max()}")
        # slab group vs whole
        if shape[0] >= 4:
            n_slabs = int(rng.integers(2, min(4, shape[0] // 2) + 1))
            indptr, indices, data = system.bsr_arrays()
            scale = 1.01 * system.gershgorin_bound()
            vectors, steps = int(rng.integers(1, 12)), int(rng.integers(1, 9))
            with DeviceSolver.from_hamiltonian(system) as dev:
                whole = dev.dots_random(scale, steps, vectors, seed=case)
            from bodge_amd.solver import SlabGroup
            with SlabGroup.from_hamiltonian(system, n_slabs) as group:
                parts = group.dots_random(scale, steps, vectors, seed=case)
            err = max(np.abs(parts[0] - whole[0]).max(), np.abs(parts[1] - whole[1]).max()) / n
            check(err <= 1e-12, "slab group", case, f"{tag} slabs={n_slabs} err {err}")
        # lowest eigenpairs (Lanczos on H^2, two passes) against the dense spectrum; gapped, PH-symmetric systems
        if lanczos and system.has_symmetric_spectrum(1e-12) and 40 <= n <= 4200 and len(wpos) >= 6 and wpos[0] > 1e-3:
            k = int(rng.integers(1, 5))
            try:
                vals, vecs = system.lowest_eigenpairs(k, format="raw", method="lanczos" if n > 300 else "auto")
                ok = vals.shape == (k,) and np.abs(vals - wpos[:k]).max() <= 1e-8 and np.abs(dense @ vecs - vecs * vals).max() <= 1e-7
                check(ok, "lowest_eigenpairs", case, f"{tag} k={k} {vals} vs {wpos[:k]}")
            except Exception as exc:  # a documented refusal (e.g. no convergence within max_iter) is reported, not fatal
                print(f"note case {case} lowest_eigenpairs raised {type(exc).__name__}: {str(exc)[:120]} ({tag}, k={k}, levels {wpos[:k+2]})", flush=True)
        # the same free energy from two mirrors on one GPU
        if system.has_symmetric_spectrum(1e-12) and case % 5 == 0:
            one = system.free_energy(0.5, method="chebyshev", trace="exact")
            two = system.free_energy(0.5, method="chebyshev", trace="exact", devices=[0, 0])
            check(abs(one - two) <= 1e-11 * max(1.0, abs(one)), "devices=[0,0]", case, f"{tag} {one} vs {two}")
        # change a few terms (the device mirror must be rebuilt) and ask again; three mirrors with ragged shares
        if case % 3 == 0:
            sites = list(system.lattice.sites())
            with system as (H, D):
                for _ in range(int(rng.integers(1, 4))):
                    i = sites[int(rng.integers(len(sites)))]
                    H[i, i] = float(rng.normal()) * ba.σ0 + float(rng.normal()) * ba.σ3
            dense2 = np.asarray(system.matrix("dense"))
            ref2 = dense_ref.free_energy(dense2, 0.4)
            got2 = system.free_energy(0.4, method="dense")
            check(abs(got2 - ref2) <= 1e-10 * max(1.0, abs(ref2)), "free_energy after a second with-block", case, f"{tag} {got2} vs {ref2}")
            if system.has_symmetric_spectrum(1e-12):
                got3 = system.free_energy(0.4, method="chebyshev", trace="exact")
                check(abs(got3 - ref2) <= 1e-9 * max(1.0, abs(ref2)), "chebyshev after a second with-block", case, f"{tag} {got3} vs {ref2}")
                got4 = system.free_energy(0.4, method="chebyshev", trace="exact", devices=[0, 0, 0])
                check(abs(got4 - got3) <= 1e-11 * max(1.0, abs(got3)), "devices=[0,0,0]", case, f"{tag} {got4} vs {got3}")
        if case % 10 == 0:
            print(f"ok through case {case} ({tag}) [{time.time() - t_start:.0f} s]", flush=True)
    print(f"{n_cases} cases, {failures} failures, {time.time() - t_start:.0f} s")
    return failures


if __name__ == "__main__":
    sys.exit(1 if run(int(os.environ.get("FUZZ_SEED", "0")), int(os.environ.get("FUZZ_CASES", "100")),
                      os.environ.get("FUZZ_SIZE"), os.environ.get("FUZZ_LANCZOS", "1") == "1") else 0)
