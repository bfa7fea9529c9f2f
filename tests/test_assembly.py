"""Host-side Hamiltonian assembly and helpers (no GPU).

Mirrors the reference's tests/test_hamiltonian.py:17-315 and tests/test_common.py
with this package's API, and pins the produced BSR triple to the arrays the
reference itself produced (tests/golden).
"""

import numpy as np
import pytest
from pytest import raises

import systems
from bodge_amd import *  # noqa: F401,F403
from bodge_amd import CubicLattice, Hamiltonian, dwave, pwave, ssd, swave
from bodge_amd.common import jσ0, jσ1, jσ2, jσ3, σ0, σ1, σ2, σ3


def test_pauli_algebra():
    for s in (σ1, σ2, σ3):
        assert np.allclose(s @ s, σ0)
    assert np.allclose(σ1 @ σ2, jσ3)
    assert np.allclose(σ2 @ σ3, jσ1)
    assert np.allclose(σ3 @ σ1, jσ2)
    assert np.allclose(σ1 @ σ2 @ σ3, jσ0)


@pytest.mark.parametrize("name", sorted(systems.CATALOG))
def test_bsr_triple_matches_reference(api, golden, name):
    spec = systems.CATALOG[name]
    system = spec["build"](api, **spec["kwargs"])
    mat, ref = system._matrix, golden.values[name]
    assert [str(mat.indptr.dtype), str(mat.indices.dtype), str(mat.data.dtype)] == ref["dtypes"]
    assert mat.blocksize == (4, 4)
    assert np.array_equal(mat.indptr, golden.arrays[f"{name}/indptr"])
    assert np.array_equal(mat.indices, golden.arrays[f"{name}/indices"])
    if f"{name}/data" in golden.arrays:
        assert np.array_equal(mat.data, golden.arrays[f"{name}/data"])
    else:
        assert np.isclose(np.abs(mat.data).sum(), ref["data_abs_sum"], rtol=1e-13)
        assert np.allclose([mat.data.sum().real, mat.data.sum().imag], ref["data_sum"], atol=1e-9)
    trimmed = system.matrix("bsr")
    assert trimmed.indices.size == ref["trimmed_nnzb"]
    assert np.bincount(np.diff(trimmed.indptr)).tolist() == ref["row_blocks_hist"]
    indptr, indices, data = system.bsr_arrays()
    assert np.array_equal(indptr, trimmed.indptr) and np.array_equal(indices, trimmed.indices)
    assert np.array_equal(data, trimmed.data)


def test_random_fill_is_hermitian_and_bad_fill_raises(api):
    system = systems.random_periodic(api, seed=7)
    dense = system._matrix.todense()
    assert np.allclose(dense, dense.T.conj())
    with raises(RuntimeError):
        with system as (H, Δ):
            H[(1, 1, 1), (1, 1, 1)] = 1j * σ1


def test_particle_hole_block_structure(api):
    system = systems.random_periodic(api, seed=3)
    lat = system.lattice
    i, j = (0, 1, 2), (0, 1, 3)
    k_ij, k_ji = system.index(i, j), system.index(j, i)
    blk_ij, blk_ji = system._data[k_ij], system._data[k_ji]
    assert np.allclose(blk_ij[2:4, 2:4], -blk_ij[0:2, 0:2].conj())
    assert np.allclose(blk_ji[2:4, 0:2], blk_ij[0:2, 2:4].conj().T)
    assert system._matrix.indices[k_ij] == lat[j]
    with raises(IndexError):
        system.index((0, 0, 0), (2, 4, 6))  # not neighbours, not opposite faces along one axis


def test_matrix_export(api):
    lattice = CubicLattice((3, 5, 7))
    system = Hamiltonian(lattice)
    with system as (H, Δ):
        for i, j in lattice:
            H[i, j] = 3 * σ0 - 4 * σ2
            Δ[i, j] = 2 * σ3 + 5 * σ2
    dense, bsr, csr, csc = (system.matrix(format=f) for f in ("dense", "bsr", "csr", "csc"))
    assert isinstance(dense, np.ndarray)
    assert (bsr.getformat(), csr.getformat(), csc.getformat()) == ("bsr", "csr", "csc")
    assert np.real(dense[0, 0]) == 3 and np.imag(dense[0, 1]) == 4
    assert np.real(dense[0, 2]) == 2 and np.imag(dense[0, 3]) == -5
    for sparse in (bsr, csr, csc):
        assert np.max(np.abs(sparse - dense)) < 1e-6
    assert bsr.blocksize == (4, 4)
    with raises(RuntimeError):
        system.matrix(format="blah")
    with raises(Exception):
        system.matrix(format=1)


def test_reentering_with_block_updates_terms(api):
    """tests/test_physics.py:47-49 pattern: a second `with` overwrites only what it names."""
    system = systems.swave_square(api, L=4, gap=0.0)
    before = system._data.copy()
    rev = system._revision
    with system as (H, Δ):
        for i in system.lattice.sites():
            Δ[i, i] = 0.5 * jσ2
    assert system._revision == rev + 1
    k = system.index((1, 1, 0), (1, 1, 0))
    assert np.allclose(system._data[k, 0:2, 2:4], 0.5 * jσ2)
    assert np.allclose(system._data[k, 0:2, 0:2], before[k, 0:2, 0:2])


def test_bulk_setters_equal_dict_fill(api):
    lattice = CubicLattice((5, 4, 3))
    by_dict, by_bulk = Hamiltonian(lattice), Hamiltonian(lattice)
    rng = np.random.default_rng(5)
    onsite = rng.random((lattice.size, 4))
    d = dwave()
    with by_dict as (H, Δ):
        for n, i in enumerate(lattice.sites()):
            H[i, i] = onsite[n, 0] * σ0 + onsite[n, 3] * σ3
            Δ[i, i] = -0.2 * jσ2
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * σ0
            Δ[i, j] = -0.1 * d(i, j)
        for i, j in lattice.edges(axis=1):
            H[i, j] = -0.5 * σ0
    with by_bulk as (H, Δ):
        H.set_sites(onsite[:, 0, None, None] * σ0 + onsite[:, 3, None, None] * σ3)
        Δ.set_sites(-0.2 * jσ2)
        H.set_bonds(-1.0 * σ0)
        pairs = lattice.bond_array(coords=True)
        Δ.set_bonds(-0.1 * d(pairs[:, 0], pairs[:, 1]))
        H.set_edges(-0.5 * σ0, axis=1)
    assert np.array_equal(by_dict._data, by_bulk._data)


def test_generic_lattice_subclass_uses_iteration_path():
    class Ring(CubicLattice.__mro__[1]):  # plain Lattice subclass
        def index(self, coord):
            return coord[0]

        def sites(self):
            return ((x, 0, 0) for x in range(self.shape[0]))

        def bonds(self):
            n = self.shape[0]
            for x in range(n):
                yield (x, 0, 0), ((x + 1) % n, 0, 0)
                yield ((x + 1) % n, 0, 0), (x, 0, 0)

        def edges(self):
            return iter(())

    system = Hamiltonian(Ring((6, 1, 1)))
    assert np.array_equal(np.diff(system._matrix.indptr), np.full(6, 3))
    with system as (H, Δ):
        for i, j in system.lattice.bonds():
            H[i, j] = -1.0 * σ0
    dense = np.asarray(system.matrix())
    assert np.allclose(dense, dense.conj().T) and dense[0, 20] == -1


# ---- pairing helpers (reference tests/test_hamiltonian.py:110-315) ---------
def test_swave_returns_isigma2():
    assert np.array_equal(swave()((0, 0, 0), (1, 0, 0)), jσ2)


@pytest.mark.parametrize("spin,mat", [("e_x", σ1), ("e_y", σ2), ("e_z", σ3)])
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_pwave_basic(spin, mat, axis):
    gap = pwave(f"{spin} * p_{'xyz'[axis]}")
    for step_axis in range(3):
        j = [0, 0, 0]
        j[step_axis] = 1
        expect = mat @ jσ2 / 2 if step_axis == axis else 0 * σ0
        assert np.allclose(gap((0, 0, 0), tuple(j)), expect)


@pytest.mark.parametrize(
    "desc", ["e_x * p_x", "e_z * p_y", "e_y * jp_z", "e_z * (p_x + jp_y)", "(e_x + je_y) * (p_y + jp_z)"]
)
def test_pwave_odd_parity_and_hermitian(api, desc):
    gap = pwave(desc)
    for x in range(2):
        for axis in range(3):
            i = (x, x + 1, 2)
            j = tuple(c + (a == axis) for a, c in enumerate(i))
            assert np.allclose(gap(i, j), -gap(j, i))
    lattice = CubicLattice((6, 6, 1))
    system = Hamiltonian(lattice)
    with system as (H, Δ):
        for i, j in lattice.bonds():
            H[i, j] = -1 * σ0
            Δ[i, j] = -0.1 * gap(i, j)
    dense = system._matrix.todense()
    assert np.allclose(dense, dense.T.conj())


def test_dwave_symmetries():
    d = dwave()
    zero = 0 * σ0
    assert np.allclose(d((0, 0, 0), (0, 0, 0)), zero) and np.allclose(d((1, 2, 3), (1, 2, 3)), zero)
    assert np.allclose(d((0, 0, 0), (0, 0, 1)), zero) and np.allclose(d((0, 0, 1), (0, 0, 0)), zero)
    for a, b in [((0, 0, 0), (1, 0, 0)), ((0, 0, 0), (9, 0, 0)), ((1, 0, 0), (0, 0, 0))]:
        assert np.allclose(d(a, b), +1 * jσ2)
    for a, b in [((0, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 9, 0)), ((0, 1, 0), (0, 0, 0))]:
        assert np.allclose(d(a, b), -1 * jσ2)
    for a, b in [((1, 1, 0), (0, 0, 0)), ((1, -1, 0), (0, 0, 0)), ((0, 0, 0), (1, 1, 0))]:
        assert np.allclose(d(a, b), zero)


def test_ssd_profile():
    system = Hamiltonian(CubicLattice((31, 137, 1)))
    φ = ssd(system)
    assert np.allclose(φ((0, 0, 0), (0, 0, 0)), 0, atol=0.001)
    assert np.allclose(φ((15, 68, 0), (15, 68, 0)), 1, atol=0.001)
    assert φ((0, 0, 0), (0, 0, 0)) == φ((30, 136, 0), (30, 136, 0))
    assert φ((1, 21, 0), (11, 1, 0)) == φ((6, 11, 0), (6, 11, 0))


def test_argument_validation(api):
    system = systems.swave_square(api, L=3)
    with raises(TypeError):
        system.free_energy(1)  # temperature must be a float (beartype would reject an int too)
    with raises(ValueError):
        system.free_energy(-1.0, method="chebyshev")
    with raises(TypeError):
        Hamiltonian("not a lattice")


# ------------------------------------------------------------- property test
def _random_terms(rng, lattice, density):
    """Random Hermitian-compatible terms: H_ii Hermitian, H_ji = H_ij^†, Δ on a random subset."""
    hopping, pairing = {}, {}
    for i in lattice.sites():
        if rng.random() < density:
            m = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
            hopping[(i, i)] = m + m.conj().T
        if rng.random() < density:
            pairing[(i, i)] = rng.standard_normal() * np.array([[0, 1], [-1, 0]], dtype=complex)
    links = list(lattice.bonds()) + list(lattice.edges())
    for i, j in links:
        if i < j and rng.random() < density:  # one draw per undirected link, written both ways
            m = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
            hopping[(i, j)], hopping[(j, i)] = m, m.conj().T
            if rng.random() < 0.5:
                d = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
                pairing[(i, j)], pairing[(j, i)] = d, -d.T  # fermionic antisymmetry
    return hopping, pairing


@pytest.mark.parametrize("seed", range(12))
def test_random_lattices_assemble_like_the_reference_restatement(api, seed):
    """Random shapes (degenerate axes included), random sparse terms written through the dict API:
    the BSR triple must equal the loop-based restatement of hamiltonian.py:37-67, :102-118
    entry for entry, and `matrix("bsr")` must drop exactly the all-zero blocks."""
    from oracle import dense_ref

    rng = np.random.default_rng(100 + seed)
    shape = tuple(int(v) for v in rng.integers(1, 6, size=3))
    lattice = api.CubicLattice(shape)
    hopping, pairing = _random_terms(rng, lattice, density=float(rng.uniform(0.2, 1.0)))
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        for key, val in hopping.items():
            H[key] = val
        for key, val in pairing.items():
            Δ[key] = val
    index = lattice.__getitem__
    pairs = [(index(i), index(j)) for i, j in lattice]
    ref = dense_ref.assemble_bsr(
        lattice.size, pairs,
        {(index(i), index(j)): v for (i, j), v in hopping.items()},
        {(index(i), index(j)): v for (i, j), v in pairing.items()})
    ref.sort_indices()
    mine = system._matrix
    assert np.array_equal(mine.indptr, ref.indptr) and np.array_equal(mine.indices, ref.indices)
    assert np.array_equal(mine.data, ref.data)
    trimmed = system.matrix("bsr")
    keep = np.any(ref.data.reshape(len(ref.data), 16) != 0, axis=1)
    assert trimmed.indices.size == int(keep.sum())
    assert np.array_equal(np.asarray(trimmed.todense()), np.asarray(ref.todense()))
    ptr, idx, dat = system.bsr_arrays()
    assert np.array_equal(idx, trimmed.indices) and np.array_equal(ptr, trimmed.indptr) and np.array_equal(dat, trimmed.data)


def test_partial_updates_are_checked_and_a_failed_check_is_not_forgotten(api):
    """Re-entering `with` checks Hermiticity of the blocks it wrote; after a failed check the next
    one covers the whole matrix again (the bad block must not slip through)."""
    lattice = api.CubicLattice((4, 3, 1))
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = 1.0 * api.σ0
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * api.σ0
    with pytest.raises(RuntimeError):
        with system as (H, Δ):
            H[(0, 0, 0), (1, 0, 0)] = 2.0 * api.σ0  # its partner (1,0,0)->(0,0,0) still holds -1
    with pytest.raises(RuntimeError):
        with system as (H, Δ):
            H[(2, 2, 0), (2, 2, 0)] = 0.5 * api.σ3  # fine by itself: the earlier damage must still be seen
    with system as (H, Δ):
        H[(1, 0, 0), (0, 0, 0)] = 2.0 * api.σ0      # repaired
    assert system._hermiticity_defect() == 0.0
    with system as (H, Δ):
        H[(3, 1, 0), (3, 1, 0)] = 0.25 * api.σ1      # partial update on a healthy matrix
    dense = np.asarray(system.matrix("dense"))
    assert np.array_equal(dense, dense.conj().T)


# ---------------------------------------------------------------------------
# The threaded host helpers (bdg_host_*, csrc/host_assembly.hpp) and the written-down cubic skeleton
# against the numpy forms they replace: same arrays, bit for bit, signs of zeros included.

def _bits(array):
    return np.ascontiguousarray(array).view(np.uint64)


def _random_bulk_terms(api, shape, seed, native, monkeypatch):
    monkeypatch.setenv("BODGE_AMD_HOST_NATIVE", "1" if native else "0")
    rng = np.random.default_rng(seed)
    lattice = api.CubicLattice(shape)
    system = api.Hamiltonian(lattice)
    n_bonds, n_edges = len(lattice.bond_array()), len(lattice.edge_array())
    with system as (H, Δ):
        H.set_sites(rng.normal(size=(lattice.size, 1, 1)) * σ0 + rng.normal(size=(lattice.size, 1, 1)) * σ3)
        Δ.set_sites(-0.1 * jσ2)  # one matrix for every site
        if n_bonds:
            hop = rng.normal(size=(n_bonds // 2, 1, 1)) * σ0 + 1j * rng.normal(size=(n_bonds // 2, 1, 1)) * σ2
            both = np.empty((n_bonds, 2, 2), complex)
            both[0::2], both[1::2] = hop, hop.conj().transpose(0, 2, 1)
            H.set_bonds(both)
            Δ.set_bonds(0.05 * jσ2)  # singlet pairing on the bonds: Δ_ji = Δ_ij
        if n_edges:  # on an axis of extent 1 these are self-pairs (two terms per diagonal block: the later wins)
            hop = rng.normal(size=(n_edges // 2, 1, 1)) * σ0
            both = np.empty((n_edges, 2, 2), complex)
            both[0::2], both[1::2] = hop, hop
            H.set_edges(both)
        sites = list(lattice.sites())
        H[sites[0], sites[0]] = 2.5 * σ0 - 0.0 * σ3  # keyed entries go last and carry a negative zero
        Δ[sites[-1], sites[-1]] = -0.3 * jσ2
    return system


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 1, 3), (2, 1, 1), (2, 2, 2), (1, 2, 3), (3, 3, 1), (4, 1, 2), (2, 3, 4),
                                   (3, 1, 3), (5, 4, 3), (17, 9, 1), (6, 5, 4)])
def test_host_helpers_and_cubic_skeleton_match_the_numpy_forms(api, monkeypatch, shape):
    from bodge_amd import chebyshev

    native = _random_bulk_terms(api, shape, 3, True, monkeypatch)
    plain = _random_bulk_terms(api, shape, 3, False, monkeypatch)
    # skeleton: the written-down one against the sorted pair list of a generic lattice
    rows, cols = native._skeleton_pairs()
    n = native.lattice.size
    keys = np.unique(np.concatenate([rows * n + cols, cols * n + rows]))
    assert np.array_equal(native._keys, keys)
    assert np.array_equal(native._matrix.indices, (keys % n).astype(np.int32))
    assert np.array_equal(native._matrix.indptr[1:], np.cumsum(np.bincount(keys // n, minlength=n)))
    assert np.array_equal(native._mirror, np.searchsorted(keys, (keys % n) * n + keys // n))
    # fills
    assert np.array_equal(_bits(native._data), _bits(plain._data))
    # scan + export
    monkeypatch.setenv("BODGE_AMD_HOST_NATIVE", "1")
    scan_n, triple_n, bound_n = native._block_scan(), native.bsr_arrays(), native.gershgorin_bound()
    monkeypatch.setenv("BODGE_AMD_HOST_NATIVE", "0")
    scan_p, triple_p, bound_p = plain._block_scan(), plain.bsr_arrays(), plain.gershgorin_bound()
    assert np.array_equal(scan_n["nonzero"], scan_p["nonzero"]) and scan_n["n_nonzero"] == scan_p["n_nonzero"]
    assert scan_n["ph_defect"] == scan_p["ph_defect"]
    assert bound_n == bound_p == chebyshev.spectral_bound(native._matrix.indptr, native._data, pad=1.0)
    for a, b in zip(triple_n, triple_p):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(_bits(a) if a.dtype == complex else a, _bits(b) if b.dtype == complex else b)
    trimmed = native.matrix("bsr")
    assert np.array_equal(triple_n[0], trimmed.indptr) and np.array_equal(triple_n[1], trimmed.indices)
    assert np.array_equal(triple_n[2], trimmed.data)


def test_host_fill_applies_terms_in_order_and_checks_its_arguments(api):
    import ctypes as C

    from bodge_amd import backend

    lib = backend.load()
    data = np.zeros((5, 4, 4), complex)
    touched = np.zeros(5, np.uint8)
    ids = np.array([3, 1, 3, 3], np.int64)  # block 3 named three times: the last term stays
    values = (np.arange(4)[:, None, None] + 1) * (σ0 + 1j * σ1)
    backend.check(lib.bdg_host_fill_terms(backend.as_f64p(data), 5, backend.as_i64p(ids), 4,
                                          backend.as_f64p(np.ascontiguousarray(values)), 1, 0, backend.as_u8p(touched)))
    assert touched.tolist() == [0, 1, 0, 1, 0]
    assert np.array_equal(data[3, 0:2, 0:2], values[3]) and np.array_equal(data[3, 2:4, 2:4], -values[3].conj())
    assert np.array_equal(data[1, 0:2, 0:2], values[1]) and not data[[0, 2, 4]].any()
    pair = np.ascontiguousarray(np.array([[1 + 2j, 3 - 1j], [0.5j, -2.0]]))
    backend.check(lib.bdg_host_fill_terms(backend.as_f64p(data), 5, backend.as_i64p(ids), 4, backend.as_f64p(pair), 0, 1, None))
    backend.check(lib.bdg_host_fill_terms(backend.as_f64p(data), 5, backend.as_i64p(ids[:1]), 1, backend.as_f64p(pair), 0, 2, None))
    assert np.array_equal(data[1, 0:2, 2:4], pair) and np.array_equal(data[3, 2:4, 0:2], pair.conj().T)
    with raises(ValueError):
        backend.check(lib.bdg_host_fill_terms(backend.as_f64p(data), 5, backend.as_i64p(np.array([5], np.int64)), 1,
                                              backend.as_f64p(pair), 0, 0, None))
    with raises(ValueError):
        backend.check(lib.bdg_host_fill_terms(backend.as_f64p(data), 5, backend.as_i64p(ids), 4, backend.as_f64p(pair), 0, 7, None))
    with raises(ValueError):
        backend.check(lib.bdg_host_scan_blocks(backend.as_f64p(data), backend.as_i32p(np.array([0, 3, 2], np.int32)), 2,
                                               None, None, None, None, None))
    # a NaN anywhere makes the bound and the defect NaN (numpy's max does the same)
    data[2, 1, 1] = np.nan
    defect, bound = C.c_double(0.0), C.c_double(0.0)
    backend.check(lib.bdg_host_scan_blocks(backend.as_f64p(data), backend.as_i32p(np.array([0, 2, 5], np.int32)), 2,
                                           None, None, C.byref(defect), C.byref(bound), None))
    assert np.isnan(defect.value) and np.isnan(bound.value)


def test_long_block_rows_sum_like_numpy(api, monkeypatch):
    """Rows of 8 and more blocks take the unrolled branches of numpy's pairwise sum: for real and purely
    imaginary entries (every BASELINE configuration) the bound is numpy's double exactly."""
    import ctypes as C

    from bodge_amd import backend, chebyshev

    rng = np.random.default_rng(0)
    lengths = np.array([1, 2, 7, 8, 9, 10, 16, 17, 24, 129, 130, 300, 3])
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    for real in (True, False):
        data = rng.normal(size=(indptr[-1], 4, 4)) + (0 if real else 1j) * rng.normal(size=(indptr[-1], 4, 4))
        data = np.ascontiguousarray(data, dtype=complex)
        bound = C.c_double(0.0)
        backend.check(backend.load().bdg_host_scan_blocks(backend.as_f64p(data), backend.as_i32p(indptr), len(lengths),
                                                         None, None, None, C.byref(bound), None))
        monkeypatch.setenv("BODGE_AMD_HOST_NATIVE", "0")  # the numpy form of the same function
        expected = chebyshev.spectral_bound(indptr, data, pad=1.0)
        monkeypatch.delenv("BODGE_AMD_HOST_NATIVE")
        assert chebyshev.spectral_bound(indptr, data, pad=1.0) == bound.value  # (default: the library's scan)
        # (general complex entries: numpy's vectorised |z| and libm's hypot differ in the last bit)
        assert bound.value == expected if real else abs(bound.value - expected) <= 4e-16 * expected


def test_reference_style_loops_take_the_array_route_and_a_subclass_the_per_key_route(api):
    """`H[i, j] = ...` in Python loops (the reference's way, ref README.md:73-86): for a CubicLattice
    the keys become site indices in one array operation, for any other Lattice one `lattice[coord]`
    at a time - and the skeleton is written down for the one, sorted from the pair list for the other.
    Same matrix either way; errors of the keyed route unchanged."""
    class Relabelled(CubicLattice):  # not `type(...) is CubicLattice`: generic skeleton, per-key conversion
        pass

    rng = np.random.default_rng(5)
    built = []
    for cls in (CubicLattice, Relabelled):
        lattice = cls((5, 4, 3))
        system = Hamiltonian(lattice)
        values = np.random.default_rng(5)
        with system as (H, Δ):
            for i in lattice.sites():
                H[i, i] = values.normal() * σ0 + values.normal() * σ3
                Δ[i, i] = values.normal() * jσ2
            for i, j in lattice.bonds():
                H[i, j] = -1.0 * σ0 + (0.3j if i < j else -0.3j) * σ2
            for i, j in lattice.edges(axis=0):
                H[i, j] = -0.5 * σ0
        built.append(system)
    a, b = built
    assert np.array_equal(a._matrix.indptr, b._matrix.indptr) and np.array_equal(a._matrix.indices, b._matrix.indices)
    assert np.array_equal(_bits(a._data), _bits(b._data))
    for bad_key, error in [(((0, 0, 0), (5, 0, 0)), ValueError), (((0, 0, -1), (0, 0, 0)), ValueError),
                           (((0, 0, 0), (2, 2, 1)), IndexError)]:
        for system in built:
            with raises(error):
                with system as (H, Δ):
                    H[bad_key] = σ0
    for system in built:
        with raises(Exception):
            with system as (H, Δ):
                H[(0.5, 0, 0), (0, 0, 0)] = σ0
        with raises(ValueError):
            with system as (H, Δ):
                H[(0, 0, 0), (0, 0, 0)] = np.eye(3)
