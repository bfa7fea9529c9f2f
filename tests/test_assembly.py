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
