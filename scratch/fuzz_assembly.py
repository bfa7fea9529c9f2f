import os, sys, itertools
sys.path.insert(0, '/root/repo')
import numpy as np
import bodge_amd as ba

def run(seed, native):
    os.environ["BODGE_AMD_HOST_NATIVE"] = "1" if native else "0"
    rng = np.random.default_rng(seed)
    shape = tuple(int(v) for v in rng.integers(1, 6, size=3))
    lat = ba.CubicLattice(shape)
    s = ba.Hamiltonian(lat)
    sites = list(lat.sites()); bonds = list(lat.bonds()); edges = list(lat.edges())
    out = []
    for block in range(int(rng.integers(1, 5))):
        err = None
        try:
            with s as (H, D):
                for op in range(int(rng.integers(1, 6))):
                    kind = rng.choice(["sites", "bonds", "edges", "keyed", "bad"], p=[0.25, 0.25, 0.15, 0.3, 0.05])
                    if kind == "sites":
                        v = rng.normal(size=(lat.size, 1, 1)) * ba.σ0 + rng.normal(size=(lat.size, 1, 1)) * ba.σ3
                        (H if rng.random() < 0.7 else D).set_sites(v if rng.random() < 0.5 else rng.normal() * ba.σ0)
                    elif kind == "bonds" and bonds:
                        ax = rng.choice([None, 0, 1, 2])
                        n = len(lat.bond_array(None if ax is None else int(ax)))
                        if n:
                            t = rng.normal(size=(n // 2, 1, 1)) * ba.σ0 + 1j * rng.normal(size=(n // 2, 1, 1)) * ba.σ2
                            both = np.empty((n, 2, 2), complex); both[0::2] = t; both[1::2] = t.conj().transpose(0, 2, 1)
                            H.set_bonds(both, axis=None if ax is None else int(ax))
                            if rng.random() < 0.5: D.set_bonds(rng.normal() * ba.jσ2, axis=None if ax is None else int(ax))
                    elif kind == "edges" and edges:
                        n = len(lat.edge_array())
                        t = rng.normal(size=(n // 2, 1, 1)) * ba.σ0
                        both = np.empty((n, 2, 2), complex); both[0::2] = t; both[1::2] = t
                        H.set_edges(both)
                    elif kind == "keyed":
                        for _ in range(int(rng.integers(1, 8))):
                            i = sites[int(rng.integers(len(sites)))]
                            H[i, i] = rng.normal() * ba.σ0 + rng.normal() * ba.σ1
                            if bonds and rng.random() < 0.5:
                                a, b = bonds[int(rng.integers(len(bonds)))]
                                t = rng.normal() * ba.σ0 + 1j * rng.normal() * ba.σ3
                                H[a, b] = t; H[b, a] = t.conj().T
                                D[a, b] = rng.normal() * ba.jσ2
                    elif kind == "bad":
                        i = sites[int(rng.integers(len(sites)))]
                        H[i, i] = 1j * ba.σ1 * rng.normal()  # not Hermitian
        except RuntimeError as exc:
            err = str(exc)
        out.append((err, s._data.copy(), s.bsr_arrays(), s.has_symmetric_spectrum(), s.gershgorin_bound()))
    return shape, out

bad = 0
for seed in range(int(__import__("os").environ.get("FUZZ_FIRST", "0")), int(__import__("os").environ.get("FUZZ_FIRST", "0")) + 400):
    sa, a = run(seed, True); sb, b = run(seed, False)
    assert sa == sb and len(a) == len(b)
    for (ea, da, ta, pa, ga), (eb, db, tb, pb, gb) in zip(a, b):
        ok = ea == eb and np.array_equal(da.view(np.uint64), db.view(np.uint64)) and pa == pb and abs(ga - gb) <= 4e-16 * abs(gb)
        ok = ok and all(np.array_equal(x, y) for x, y in zip(ta, tb))
        if not ok:
            bad += 1; print("MISMATCH seed", seed, sa, ea, eb, pa, pb, ga, gb)
print("400 seeds,", bad, "mismatches")
