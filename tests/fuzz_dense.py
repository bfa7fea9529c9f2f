"""Random Hermitian BdG matrices through the library's own dense route (csrc/tridiag.hpp, forced): eigenvalues
against numpy, eigen-equation residual and orthonormality of the eigenvectors of the positive half.  Prints one
line per failure; exit code 1 if any.  Cases lean on what is numerically delicate: decoupled sub-lattices (exact
zeros on the sub-diagonal: the tau = 0 path, split tridiagonal matrices), highly degenerate spectra (clusters),
zero modes, tiny matrices, complex blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bodge_amd as ba
from bodge_amd import backend
from bodge_amd.solver import DeviceSolver


def _system(rng, case):
    kind = rng.choice(["uniform", "disorder", "texture", "decoupled", "dwave", "flat", "periodic"])
    if rng.random() < 0.25:
        shape = (int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(1, 4)))  # tiny: n = 4 .. 192
    else:
        shape = (int(rng.integers(2, 26)), int(rng.integers(1, 26)), 1 if rng.random() < 0.7 else int(rng.integers(1, 4)))
    lat = ba.CubicLattice(shape)
    s = ba.Hamiltonian(lat)
    n = lat.size
    with s as (H, D):
        if kind == "disorder":
            H.set_sites(rng.normal(size=(n, 1, 1)) * ba.σ0 + 0.2 * rng.normal(size=(n, 1, 1)) * ba.σ3)
        elif kind == "texture":
            th, ph = rng.uniform(0, np.pi, (n, 1, 1)), rng.uniform(0, 2 * np.pi, (n, 1, 1))
            H.set_sites(2.0 * ba.σ0 - 0.5 * (np.sin(th) * np.cos(ph) * ba.σ1 + np.sin(th) * np.sin(ph) * ba.σ2 + np.cos(th) * ba.σ3))
        elif kind == "flat":
            H.set_sites(1.0 * ba.σ0)  # no hopping below: a few eigenvalues with huge multiplicity
        else:
            H.set_sites(float(rng.uniform(0, 4)) * ba.σ0 - float(rng.uniform(0, 0.5)) * ba.σ3)
        if kind != "dwave":
            D.set_sites(-float(rng.uniform(0, 1)) * ba.jσ2)
        if kind not in ("flat",) and n > 1:
            pairs = lat.bond_array(coords=True)
            if len(pairs):
                if kind == "decoupled":  # only the x bonds: Ly*Lz independent chains
                    if shape[0] > 1:
                        H.set_bonds(-1.0 * ba.σ0, axis=0)
                else:
                    H.set_bonds(-1.0 * ba.σ0)
                if kind == "dwave":
                    D.set_bonds(-float(rng.uniform(0.1, 1)) * ba.dwave()(pairs[:, 0], pairs[:, 1]))
                if kind == "periodic":
                    for axis in range(3):
                        if shape[axis] > 2:
                            H.set_edges(-1.0 * ba.σ0, axis=axis)
    return s, f"case {case}: {shape} {kind}"


def run(seed: int = 0, n_cases: int = 100, stages: str | None = None) -> int:
    """`stages="2"`: the eigenvalues through the two-stage route (csrc/twostage.hpp: band by MFMA panels, bulge chasing) for every
    real matrix large enough for a band (4N > 66) - the eigenvectors keep the one-stage route either way."""
    rng = np.random.default_rng(seed)
    failures, t_start = 0, time.time()
    for case in range(n_cases):
        system, tag = _system(rng, case)
        dim = system.shape[0]
        dense = np.asarray(system.matrix("dense"))
        exact = np.linalg.eigvalsh(dense)
        scale = max(1.0, np.abs(exact).max())
        extra = {"BODGE_AMD_EIGH_STAGES": stages} if stages else {}
        with DeviceSolver.from_hamiltonian(system) as dev, backend.options(BODGE_AMD_EIGH="tridiagonal", **extra):
            w, _ = dev.eigh(vectors=False)
            w2, z = dev.eigh_above(0.0)
        problems = []
        # (real matrices from 3000 rows: eigenpairs through the band, eigenvalues alone by the one-stage route up to 5000 - two
        # routes, rounding apart; below, and for complex matrices, the very same tridiagonal matrix)
        one_route = not stages and (dim < 3000 or np.abs(dense.imag).max() > 0)
        same = np.array_equal(w, w2) if one_route else np.abs(w - w2).max() <= 1e-11 * scale
        if not (np.abs(w - exact).max() <= 1e-11 * scale and same):
            problems.append(f"eigenvalues off by {np.abs(w - exact).max():.1e}")
        vals = w2[w2 > 0]
        if z.shape != (dim, vals.size) or not np.isfinite(z).all():
            problems.append(f"vectors {z.shape}, finite {np.isfinite(z).all()}")
        elif vals.size:
            residual = np.abs(dense @ z - z * vals).max()
            gram = np.abs(z.conj().T @ z - np.eye(vals.size)).max()
            if not (residual <= 1e-9 * scale and gram <= 1e-9):
                problems.append(f"residual {residual:.1e}, orthonormality {gram:.1e}")
        if problems:
            failures += 1
            print("FAIL", tag, f"n = {dim}:", "; ".join(problems), flush=True)
        elif case % 20 == 0:
            print("ok  ", tag, f"n = {dim}, {vals.size} vectors  [{time.time() - t_start:.0f} s]", flush=True)
    print(f"{n_cases} cases, {failures} failures, {time.time() - t_start:.0f} s")
    return failures


if __name__ == "__main__":
    sys.exit(1 if run(int(os.environ.get("FUZZ_SEED", "0")), int(os.environ.get("FUZZ_CASES", "100")), os.environ.get("FUZZ_STAGES")) else 0)
