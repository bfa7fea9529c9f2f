"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the goldens.

Tolerances (fp64 throughout):
* SpMV and start vectors        bit-level / 1e-13 relative
* recurrence dots, per moment   1e-12 relative to μ_0
* free energy                   1e-10 relative, vs goldens where (T, M) supports it
* eigenvalues                   1e-10 absolute (pattern of ref tests/test_hamiltonian.py:411-413)
"""

import os

import numpy as np
import pytest

import systems
from oracle import cheb_ref, dense_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["dictionary", "streamed"])
def block_storage(request, knobs):
    """Every test runs twice: with the block dictionary (used whenever a matrix has few distinct
    blocks, i.e. for most lattice systems here) and with it disabled, so that the streamed
    (pipelined / generic) kernels keep their coverage."""
    if request.param == "streamed":
        knobs.set("BODGE_AMD_DICT", "0")
    return request.param


@pytest.fixture(scope="module")
def solver_cls(hip_library):
    from bodge_amd.solver import DeviceSolver

    return DeviceSolver


def _dense_once(block_storage, dim, limit=2000):
    """The dense routes scatter the uploaded BSR triple to a dense array: how the recurrence kernels keep the blocks
    (dictionary or streamed) plays no part.  Large systems therefore run in one of the two module-wide variants."""
    if block_storage != "dictionary" and dim > limit:
        pytest.skip("dense route: independent of the block storage of the recurrence kernels; large systems run once")


def _build(api, name):
    spec = systems.CATALOG[name]
    return spec["build"](api, **spec["kwargs"])


# ----------------------------------------------------------------------- SpMV
@pytest.mark.parametrize("name", ["random357", "complex235", "dwave8", "swave20", "chain128"])
def test_spmv_matches_scipy(api, solver_cls, name):
    system = _build(api, name)
    bsr = system.matrix("bsr")
    rng = np.random.default_rng(0)
    x = rng.standard_normal(bsr.shape[0]) + 1j * rng.standard_normal(bsr.shape[0])
    with solver_cls.from_hamiltonian(system) as dev:
        y = dev.spmv(x)
    ref = bsr @ x
    assert np.allclose(y, ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())


def test_spmv_keeps_explicit_zero_blocks_and_empty_rows(api, solver_cls):
    """Skeleton upload (zero blocks kept) and a matrix with empty block rows."""
    system = _build(api, "swave20")
    x = np.arange(system.shape[0], dtype=float) * (1 + 0.5j)
    with solver_cls.from_hamiltonian(system, drop_zero_blocks=False) as dev:
        assert dev.n_blocks == system._matrix.indices.size
        assert np.allclose(dev.spmv(x), system._matrix @ x, rtol=1e-13)
    indptr = np.array([0, 0, 1, 1, 3], dtype=np.int32)  # rows 0 and 2 are empty
    indices = np.array([3, 0, 1], dtype=np.int32)
    rng = np.random.default_rng(1)
    data = rng.standard_normal((3, 4, 4)) + 1j * rng.standard_normal((3, 4, 4))
    import scipy.sparse as sp

    mat = sp.bsr_matrix((data, indices, indptr), shape=(16, 16))
    x = rng.standard_normal(16) + 0j
    with solver_cls(indptr, indices, data) as dev:
        assert np.allclose(dev.spmv(x), mat @ x, rtol=1e-13)


@pytest.mark.parametrize("seed", range(8))
def test_random_periodic_lattices_of_random_shape(api, solver_cls, seed):
    """Seeded random shapes (1..6 per axis, so degenerate axes and their self-edges occur) with dense
    complex on-site, bond and periodic-edge terms: SpMV, recurrence dots, spectrum and the dense
    free energy against the oracle.  (These matrices are Hermitian but not particle-hole
    symmetric, so they run the full-block complex kernels and the dense route.)"""
    rng = np.random.default_rng(500 + seed)
    shape = tuple(int(v) for v in rng.integers(1, 7, size=3))
    system = systems.random_periodic(api, shape=shape, seed=900 + seed)
    bsr = system.matrix("bsr")
    n = bsr.shape[0]
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    scale = cheb_ref.spectral_bound(bsr)
    n_vectors = int(rng.integers(1, 20))
    ref = cheb_ref.recurrence_dots(bsr, scale, 24, cheb_ref.random_block(n, seed, range(n_vectors), cheb_ref.VEC_Z4))
    with solver_cls.from_hamiltonian(system) as dev:
        y = dev.spmv(x)
        got = dev.dots_random(scale, 12, n_vectors, seed=seed, kind=cheb_ref.VEC_Z4)
    assert np.allclose(y, bsr @ x, rtol=1e-12, atol=1e-12 * np.abs(bsr @ x).max())
    assert np.allclose(got[0], ref[0], rtol=0, atol=1e-12 * n) and np.allclose(got[1], ref[1], rtol=0, atol=1e-12 * n)
    dense = np.asarray(system.matrix("dense"))
    vals, vecs = system.diagonalize(format="raw")
    ref_vals, _ = dense_ref.diagonalize(dense, format="raw")
    assert np.allclose(vals, ref_vals, rtol=0, atol=1e-10) and np.allclose(dense @ vecs, vecs * vals, atol=1e-9)
    for temperature in (0.0, 0.3):
        assert np.isclose(system.free_energy(temperature), dense_ref.free_energy(dense, temperature), rtol=1e-10)


def test_device_start_vectors_equal_oracle(api, solver_cls):
    system = _build(api, "barrier")
    with solver_cls.from_hamiltonian(system) as dev:
        for kind in (cheb_ref.VEC_RADEMACHER, cheb_ref.VEC_Z4):
            for seed, vec in [(0, 0), (12345, 7), (2**63 + 5, 2**40)]:
                assert np.array_equal(dev.random_vector(seed, vec, kind),
                                      cheb_ref.random_vector(dev.dim, seed, vec, kind))


# ----------------------------------------------------------------- recurrence
@pytest.mark.parametrize("name,n_vectors,kind", [
    ("random357", 1, cheb_ref.VEC_Z4),
    ("random357", 3, cheb_ref.VEC_Z4),
    ("random357", 8, cheb_ref.VEC_RADEMACHER),
    ("dwave8", 16, cheb_ref.VEC_RADEMACHER),
    ("swave20", 64, cheb_ref.VEC_Z4),
    ("complex235", 70, cheb_ref.VEC_Z4),  # more than one batch of 64
])
def test_recurrence_dots_match_oracle(api, solver_cls, name, n_vectors, kind):
    system = _build(api, name)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    steps = 40
    start = cheb_ref.random_block(bsr.shape[0], 9, range(5, 5 + n_vectors), kind)
    d_ref, e_ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, start)
    with solver_cls.from_hamiltonian(system) as dev:
        d, e = dev.dots_random(scale, steps, n_vectors, seed=9, first_id=5, kind=kind)
        d2, e2 = dev.dots_random(scale, steps, n_vectors, seed=9, first_id=5, kind=kind)
    mu0 = bsr.shape[0]
    assert np.allclose(d, d_ref, rtol=0, atol=1e-12 * mu0)
    assert np.allclose(e, e_ref, rtol=0, atol=1e-12 * mu0)
    assert np.array_equal(d, d2) and np.array_equal(e, e2), "reductions must be bit-reproducible"


def test_unit_vector_moments_match_oracle(api, solver_cls):
    system = _build(api, "random357")
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    rows = np.array([0, 5, 17, 418, 419])
    ref = cheb_ref.moments(bsr, scale, 64, cheb_ref.unit_block(bsr.shape[0], rows))
    with solver_cls.from_hamiltonian(system) as dev:
        mu = dev.moments_unit(scale, 64, rows)
    assert np.allclose(mu, ref, rtol=0, atol=1e-13)


@pytest.mark.parametrize("name,n_vectors", [("dwave8", 5), ("swave20", 8), ("swave20", 64), ("chain128", 3)])
def test_real_arithmetic_equals_complex_arithmetic(api, solver_cls, knobs, name, n_vectors):
    """imag(H) == 0 and ±1 start vectors select the real-valued kernels; they must agree with
    the complex kernels (forced with BODGE_AMD_REAL=0) and with the oracle."""
    system = _build(api, name)
    bsr = system.matrix("bsr")
    real_h = not np.any(bsr.data.imag)
    scale = cheb_ref.spectral_bound(bsr)
    steps = 24
    with solver_cls.from_hamiltonian(system) as dev:
        fast = dev.dots_random(scale, steps, n_vectors, seed=4)
        assert dev.perf()["real_arithmetic"] == int(real_h)
        unit_fast = dev.moments_unit(scale, 2 * steps, np.array([1, 6, 11]))
        knobs.set("BODGE_AMD_REAL", "0")
        slow = dev.dots_random(scale, steps, n_vectors, seed=4)
        assert dev.perf()["real_arithmetic"] == 0
        unit_slow = dev.moments_unit(scale, 2 * steps, np.array([1, 6, 11]))
    start = cheb_ref.random_block(bsr.shape[0], 4, range(n_vectors))
    ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, start)
    for got in (fast, slow):
        assert np.allclose(got[0], ref[0], rtol=0, atol=1e-12 * bsr.shape[0])
        assert np.allclose(got[1], ref[1], rtol=0, atol=1e-12 * bsr.shape[0])
    assert np.allclose(unit_fast, unit_slow, rtol=0, atol=1e-13)


def _plain_lattice_system(api, shape, complex_terms):
    lattice = api.CubicLattice(shape)
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        H.set_sites(2.5 * api.σ0 - 0.1 * api.σ3 + (0.2 * api.σ2 if complex_terms else 0))
        Δ.set_sites(-0.3 * api.jσ2)
        H.set_bonds(-1.0 * api.σ0)
        H.set_edges(-0.4 * api.σ0, axis=1)
    return system


@pytest.mark.parametrize("shape,n_vectors,kind,complex_terms", [
    ((6, 150, 1), 8, cheb_ref.VEC_Z4, True),            # 900 rows, 32-row tiles: ragged tile mid-order
    ((5, 10, 31), 16, cheb_ref.VEC_RADEMACHER, False),  # 3-D real, plane of 310 rows, ragged
    ((3, 500, 1), 64, cheb_ref.VEC_Z4, False),          # 4-row tiles, many strips
    ((7, 90, 1), 5, cheb_ref.VEC_RADEMACHER, False),    # generic (RL=4) kernel, 64-row tiles
])
def test_strip_ordered_tiles_do_not_change_results(api, solver_cls, knobs, shape, n_vectors, kind,
                                                    complex_terms):
    """The geometry hint only permutes the order in which row tiles are processed."""
    system = _plain_lattice_system(api, shape, complex_terms)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    ref = cheb_ref.recurrence_dots(bsr, scale, 24, cheb_ref.random_block(bsr.shape[0], 2, range(n_vectors), kind))
    with solver_cls.from_hamiltonian(system) as dev:
        dev.set_lattice_shape((0, 0, 0))
        natural = dev.dots_random(scale, 12, n_vectors, seed=2, kind=kind)
        assert dev.perf()["strip_rows"] == 0
        dev.set_lattice_shape(shape)
        knobs.set("BODGE_AMD_L2_BUDGET", "4096")  # absurdly small: forces the narrowest strips
        strips = dev.dots_random(scale, 12, n_vectors, seed=2, kind=kind)
        assert 0 < dev.perf()["strip_rows"] < shape[1] * shape[2]
        with pytest.raises(ValueError):
            dev.set_lattice_shape((3, 3, 3))
    mu0 = system.shape[0]
    for got in (natural, strips):
        assert np.allclose(got[0], ref[0], rtol=0, atol=1e-12 * mu0)
        assert np.allclose(got[1], ref[1], rtol=0, atol=1e-12 * mu0)


def _sweep_system(api, shape, kind):
    """Lattice models for the stencil kernels: uniform, position dependent, complex, d-wave bonds,
    and periodic ones (wrap-around blocks in one or both directions)."""
    lattice = api.CubicLattice(shape)
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        if kind == "swave":
            H.set_sites(3.0 * api.σ0 - 0.05 * api.σ3)
            Δ.set_sites(-0.1 * api.jσ2)
            H.set_bonds(-1.0 * api.σ0)
        elif kind == "peierls":
            pairs = lattice.bond_array(axis=0, coords=True)
            phase = np.where(pairs[:, 1, 0] > pairs[:, 0, 0], np.exp(0.3j), np.exp(-0.3j))
            H.set_sites(3.0 * api.σ0 - 0.05 * api.σ3)
            Δ.set_sites(-0.1 * api.jσ2)
            H.set_bonds(-phase[:, None, None] * api.σ0, axis=0)
            H.set_bonds(-1.0 * api.σ0, axis=1 if shape[1] > 1 else 2)
            if shape[1] > 1 and shape[2] > 1:
                H.set_bonds(-0.7 * api.σ0, axis=2)
        elif kind == "dwave":
            pairs = lattice.bond_array(coords=True)
            H.set_sites(3.0 * api.σ0)
            H.set_bonds(-1.0 * api.σ0)
            Δ.set_bonds(-0.1 * api.dwave()(pairs[:, 0], pairs[:, 1]))
        elif kind == "junction":  # S / F / S along x (pattern of ref tests/test_hamiltonian.py:431-443)
            x = np.arange(lattice.size) // (shape[1] * shape[2])
            middle = ((x > shape[0] // 3) & (x < 2 * shape[0] // 3))[:, None, None]
            H.set_sites(np.where(middle, 0.5 * api.σ0 + 1.5 * api.σ3, -0.5 * api.σ0))
            Δ.set_sites(np.where(middle, 0 * api.jσ2, -1.0 * api.jσ2))
            H.set_bonds(-1.0 * api.σ0)
        elif kind in ("periodic", "periodic_x", "periodic_y"):
            H.set_sites(3.0 * api.σ0 - 0.05 * api.σ3)
            Δ.set_sites(-0.1 * api.jσ2)
            H.set_bonds(-1.0 * api.σ0)
            if kind == "periodic":
                H.set_edges(-1.0 * api.σ0)
            else:  # a ring in one direction only, with a different hopping across the seam
                H.set_edges(-0.8 * api.σ0 + 0.1 * api.σ3, axis=0 if kind == "periodic_x" else (1 if shape[1] > 1 else 2))
        else:
            raise ValueError(kind)
    return system


@pytest.mark.parametrize("shape,kind,vec_kind", [
    ((64, 48, 1), "swave", cheb_ref.VEC_RADEMACHER),     # 4 full windows of 12
    ((40, 100, 1), "peierls", cheb_ref.VEC_Z4),          # complex blocks and vectors, ragged last window
    ((33, 61, 1), "dwave", cheb_ref.VEC_RADEMACHER),     # pairing on the bonds, odd sizes
    ((48, 50, 1), "junction", cheb_ref.VEC_RADEMACHER),  # position-dependent blocks
    ((16, 1, 40), "swave", cheb_ref.VEC_RADEMACHER),     # (Lx, 1, Lz): the plane is a z-line
    ((9, 25, 1), "swave", cheb_ref.VEC_Z4),              # complex vectors on a real matrix, one short segment
    ((30, 30, 1), "periodic", cheb_ref.VEC_RADEMACHER),  # torus: wrap blocks close planes and their stack into rings
    ((9, 61, 1), "periodic_y", cheb_ref.VEC_RADEMACHER),  # rings inside the planes only, ragged last window
    ((37, 30, 1), "periodic_x", cheb_ref.VEC_Z4),        # ring of planes only
    ((16, 1, 40), "periodic", cheb_ref.VEC_RADEMACHER),  # (Lx, 1, Lz) torus
    ((12, 30, 4), "swave", cheb_ref.VEC_RADEMACHER),     # 3-D: one step per launch, x-neighbours in registers (K8)
    ((9, 7, 13), "dwave", cheb_ref.VEC_RADEMACHER),      # 3-D d-wave (zero z-bond pairing blocks), odd sizes
    ((10, 6, 8), "peierls", cheb_ref.VEC_Z4),            # 3-D complex
    ((8, 5, 6), "periodic", cheb_ref.VEC_RADEMACHER),    # 3-D with wrap blocks: falls back
])
def test_multi_step_sweep_kernels_match_oracle_and_one_step_kernels(api, solver_cls, knobs, block_storage,
                                                                        shape, kind, vec_kind):
    """K7 / K8 (sweep.hpp) forced on small lattices: every d_n, e_n against the CPU oracle and against
    the one-step kernels on the same vectors, for even and odd step counts (the odd tail of a
    two-step run goes alone), full and partial lane groups, both marching directions, zigzag on
    and off.  2-D stencils take two steps per sweep, 3-D stencils the rolling one-step kernel, and
    matrices that are no lattice stencil must take the generic kernels without being told."""
    system = _sweep_system(api, shape, kind)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    n = bsr.shape[0]
    three_d = shape[1] > 1 and shape[2] > 1
    is_stencil = not (kind == "periodic" and three_d)  # (the 3-D kernel K8 has no rings: periodic 3-D falls back)
    per_group = 8 if (vec_kind == cheb_ref.VEC_RADEMACHER and kind != "peierls") else 4
    with solver_cls.from_hamiltonian(system) as dev:
        for steps, vectors, extra in [(8, per_group, {}), (7, 3, {}), (5, per_group + 3, {"BODGE_AMD_SWEEP_ZIGZAG": "0"}),
                                      (6, 2, {"BODGE_AMD_ALTERNATE": "0", "BODGE_AMD_SWEEP_SEGMENTS": "3"}),
                                      (8, per_group, {"BODGE_AMD_SWEEP_STEPS": "2"}),  # cheb_sweep (two steps) instead of cheb_sweep3
                                      (7, 3, {"BODGE_AMD_SWEEP_STEPS": "2", "BODGE_AMD_SWEEP_ZIGZAG": "0"}),
                                      (6, 5, {"BODGE_AMD_SWEEP_LANES": "1"}),  # 60-position windows, 2 real / 1 complex vector per launch
                                      (5, 3, {"BODGE_AMD_SWEEP_LANES": "2"}),  # 26-position windows (3 steps), the default from 2.5e5 sites
                                      (7, 5, {"BODGE_AMD_SWEEP_LANES": "2", "BODGE_AMD_SWEEP_SEGMENTS": "2"}),
                                      (5, 5, {"BODGE_AMD_SWEEP_LANES": "2", "BODGE_AMD_SWEEP_STEPS": "2"}),  # 28-position windows
                                      (6, 7, {"BODGE_AMD_SWEEP_LANES": "4"}),  # (K8: 14-position windows, 8 real vectors per launch)
                                      (8, per_group, {"BODGE_AMD_SWEEP_GEN": "0"}),  # start block written by the fill kernel and read back
                                      (1, 3, {}), (2, per_group, {}),  # runs shorter than one sweep
                                      (4, 3, {"BODGE_AMD_NO_DIAGONAL_BLOCKS": "1"})]:  # (read at upload: no effect here, see below)
            ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(n, 5, range(vectors), vec_kind))
            knobs.set("BODGE_AMD_SWEEP", "0")
            one = dev.dots_random(scale, steps, vectors, seed=5, kind=vec_kind)
            assert dev.perf()["steps_per_launch"] == 1
            knobs.set("BODGE_AMD_SWEEP", "1")
            for key, value in extra.items():
                knobs.set(key, value)
            got = dev.dots_random(scale, steps, vectors, seed=5, kind=vec_kind)
            perf = dev.perf()
            again = dev.dots_random(scale, steps, vectors, seed=5, kind=vec_kind)
            for key in extra:
                knobs.unset(key)
            stencil = is_stencil and block_storage == "dictionary"  # (the stencil forms read the block dictionary)
            swept, rolled = stencil and not three_d, stencil and three_d
            lanes = int(extra.get("BODGE_AMD_SWEEP_LANES", 4)) if swept else 4  # (default below 1.5e5 sites, and below 2.5e5 for calls of more than one lane group: 4 lanes per site)
            if rolled:  # K8: 4 lanes per site, or 2 (30-position windows) for calls of at most 4 real / 2 complex vectors
                lanes = int(extra["BODGE_AMD_SWEEP_LANES"]) if extra.get("BODGE_AMD_SWEEP_LANES") in ("2", "4") else (2 if vectors <= per_group // 2 else 4)
                assert perf["lanes_per_row"] in (2, lanes)  # (a ragged last batch of a wide call takes 2)
            depth = int(extra.get("BODGE_AMD_SWEEP_STEPS", 3 if lanes >= 2 else 2))  # steps per sweep
            assert perf["steps_per_launch"] == (depth if swept else 1) and (perf["rolling"] == 1) == rolled, perf
            batches = -(-vectors // (per_group * lanes // 4))
            assert not swept or perf["lanes_per_row"] == lanes
            if swept:
                assert perf["sweeps"] == batches * -(-steps // depth)
                if not perf["persistent"]:
                    assert perf["launches"] == perf["sweeps"]
            if rolled:
                assert perf["launches"] == batches * steps
            assert np.array_equal(got[0], again[0]) and np.array_equal(got[1], again[1])  # bit reproducible
            if "BODGE_AMD_SWEEP_GEN" in extra:  # the first sweep making t_0 in registers changes no bit
                assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1])
            if not extra and steps == 8:
                first = got
            for other in (ref, one):
                assert np.abs(got[0] - other[0]).max() <= 1e-12 * n and np.abs(got[1] - other[1]).max() <= 1e-12 * n
    # blocks that are diagonal as 4x4 matrices (plain hopping) take a 4-MAC path in K7/K8; with the
    # flag withheld at upload every block takes the general 16-MAC path: same numbers
    knobs.set("BODGE_AMD_NO_DIAGONAL_BLOCKS", "1")
    knobs.set("BODGE_AMD_SWEEP", "1")
    with solver_cls.from_hamiltonian(system) as dev:
        general = dev.dots_random(scale, 6, per_group, seed=5, kind=vec_kind)
    knobs.unset("BODGE_AMD_NO_DIAGONAL_BLOCKS")
    with solver_cls.from_hamiltonian(system) as dev:
        flagged = dev.dots_random(scale, 6, per_group, seed=5, kind=vec_kind)
    assert np.abs(general[0] - flagged[0]).max() <= 1e-13 * n and np.abs(general[1] - flagged[1]).max() <= 1e-13 * n


def _position_dependent_system(api, shape, kind, seed=0, periodic=False):
    """Lattices whose ON-SITE terms differ from site to site (> 256 distinct diagonal blocks) over a
    few distinct bond blocks - what a disorder potential, a self-consistent gap Δ(r) or a magnetic
    texture makes of the reference's per-site fills (ref hamiltonian.py:102-118, tests/test_physics.py:
    342-387): "potential" real, "texture" complex (σ2 component), "gap" real with d-wave bonds."""
    lattice = api.CubicLattice(shape)
    system = api.Hamiltonian(lattice)
    rng = np.random.default_rng(seed)
    n = lattice.size
    with system as (H, Δ):
        if kind == "potential":
            v = rng.uniform(-0.5, 0.5, n)[:, None, None]
            g = rng.uniform(0.05, 0.15, n)[:, None, None]
            H.set_sites((3.0 + v) * api.σ0 - 0.05 * api.σ3)
            Δ.set_sites(-g * api.jσ2)
            H.set_bonds(-1.0 * api.σ0)
        elif kind == "texture":  # exchange field rotating in space: σ1, σ2, σ3 components
            th, ph = rng.uniform(0, np.pi, n)[:, None, None], rng.uniform(0, 2 * np.pi, n)[:, None, None]
            H.set_sites(3.0 * api.σ0 - 0.3 * (np.sin(th) * np.cos(ph) * api.σ1 + np.sin(th) * np.sin(ph) * api.σ2 + np.cos(th) * api.σ3))
            Δ.set_sites(-0.1 * api.jσ2)
            H.set_bonds(-1.0 * api.σ0, axis=0)
            axis = 1 if shape[1] > 1 else 2
            pairs = lattice.bond_array(axis=axis, coords=True)  # directed: a phase one way, its conjugate back
            hop = np.where(pairs[:, 1, axis] > pairs[:, 0, axis], -0.9 + 0.1j, -0.9 - 0.1j)[:, None, None]
            H.set_bonds(hop * api.σ0 + 0.05 * api.σ3, axis=axis)
        elif kind == "gap":
            pairs = lattice.bond_array(coords=True)
            v = rng.uniform(-0.5, 0.5, n)[:, None, None]
            H.set_sites((3.0 + v) * api.σ0)
            H.set_bonds(-1.0 * api.σ0)
            Δ.set_bonds(-0.1 * api.dwave()(pairs[:, 0], pairs[:, 1]))
        else:
            raise ValueError(kind)
        if periodic:  # (an axis of extent 1 has its "edge" coincide with the site itself: left alone)
            for axis in range(3):
                if shape[axis] > 1:
                    H.set_edges(-0.8 * api.σ0 + 0.1 * api.σ3, axis=axis)
    return system


@pytest.mark.parametrize("shape,kind,vec_kind,periodic,lanes", [
    ((48, 50, 1), "potential", cheb_ref.VEC_RADEMACHER, False, 2),  # real arithmetic, 4 vectors per launch (the default)
    ((48, 50, 1), "potential", cheb_ref.VEC_RADEMACHER, False, 4),  # ... 8 vectors per launch
    ((33, 61, 1), "texture", cheb_ref.VEC_Z4, False, 4),            # complex blocks and vectors, odd sizes, ragged window (the default)
    ((33, 61, 1), "texture", cheb_ref.VEC_Z4, False, 2),            # ... workgroups of seven waves
    ((40, 37, 1), "gap", cheb_ref.VEC_RADEMACHER, False, 2),        # pairing on the bonds
    ((30, 1, 44), "potential", cheb_ref.VEC_Z4, False, 4),          # complex vectors on a real matrix, (Lx, 1, Lz)
    ((30, 32, 1), "potential", cheb_ref.VEC_RADEMACHER, True, 2),   # torus: halo slots wrap, on-site records fetched per piece
    ((21, 30, 1), "texture", cheb_ref.VEC_Z4, True, 4),
    ((21, 30, 1), "texture", cheb_ref.VEC_Z4, True, 2),
])
def test_three_step_sweep_with_streamed_onsite_blocks(api, solver_cls, knobs, block_storage, shape, kind, vec_kind, periodic, lanes):
    """cheb_sweep3<..., OS> (sweep.hpp): matrices with more than 256 distinct blocks whose bonds repeat a
    few - the bond blocks sit in the LDS table, the diagonal block of every site is streamed once per
    launch into a three-plane LDS ring.  Every d_n, e_n against the CPU oracle and against the
    one-step (streamed-blocks) kernels on the same vectors; step counts that end a run on 1, 2 and 3
    steps, partial lane groups, several batches, both marching directions, GEN on and off.
    Round 4: with 2 lanes per site (32-slot windows; the default in real arithmetic - in complex arithmetic one workgroup
    of seven waves per CU, since the ring leaves no room for eight) as well as with 4."""
    system = _position_dependent_system(api, shape, kind, periodic=periodic)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    n = bsr.shape[0]
    complex_run = vec_kind == cheb_ref.VEC_Z4 or kind == "texture"
    per_group = lanes * (1 if complex_run else 2)
    with solver_cls.from_hamiltonian(system) as dev:
        if block_storage == "dictionary":  # the default: 2 lanes in real arithmetic, 4 in complex
            knobs.set("BODGE_AMD_SWEEP", "1")
            dev.dots_random(scale, 3, 2, seed=5, kind=vec_kind)
            assert dev.perf()["onsite_streamed"] == 1 and dev.perf()["lanes_per_row"] == (4 if complex_run else 2), dev.perf()
        knobs.set("BODGE_AMD_SWEEP_LANES", str(lanes))
        for steps, vectors, extra in [(9, per_group, {}), (7, 3, {}), (8, per_group + 3, {"BODGE_AMD_SWEEP_ZIGZAG": "0"}),
                                      (6, 2, {"BODGE_AMD_ALTERNATE": "0", "BODGE_AMD_SWEEP_SEGMENTS": "3"}),
                                      (9, per_group, {"BODGE_AMD_SWEEP_GEN": "0"}),
                                      (1, 3, {}), (2, per_group, {}), (13, 2 * per_group, {"BODGE_AMD_SWEEP_SEGMENTS": "2"})]:
            ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(n, 5, range(vectors), vec_kind))
            knobs.set("BODGE_AMD_SWEEP", "0")
            one = dev.dots_random(scale, steps, vectors, seed=5, kind=vec_kind)
            perf1 = dev.perf()
            assert perf1["steps_per_launch"] == 1 and perf1["dict_blocks"] == 0 and perf1["onsite_streamed"] == 0
            knobs.set("BODGE_AMD_SWEEP", "1")
            knobs.update(extra)
            got = dev.dots_random(scale, steps, vectors, seed=5, kind=vec_kind)
            perf = dev.perf()
            again = dev.dots_random(scale, steps, vectors, seed=5, kind=vec_kind)
            for key in extra:
                knobs.unset(key)
            if block_storage == "dictionary":
                assert perf["onsite_streamed"] == 1 and perf["steps_per_launch"] == 3 and perf["lanes_per_row"] == lanes, perf
                assert 0 < perf["dict_blocks"] <= 16 and perf["real_arithmetic"] == (0 if complex_run else 1)
                assert perf["launches"] == -(-vectors // per_group) * -(-steps // 3)
            else:  # BODGE_AMD_DICT=0: no dictionary of any kind, one step per launch
                assert perf["onsite_streamed"] == 0 and perf["steps_per_launch"] == 1
            assert np.array_equal(got[0], again[0]) and np.array_equal(got[1], again[1])
            if "BODGE_AMD_SWEEP_GEN" in extra:
                assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1])
            if not extra and steps == 9:
                first = got
            for other in (ref, one):
                assert np.abs(got[0] - other[0]).max() <= 1e-12 * n and np.abs(got[1] - other[1]).max() <= 1e-12 * n


@pytest.mark.parametrize("shape,kind,vec_kind", [
    ((64, 48, 1), "swave", cheb_ref.VEC_RADEMACHER),
    ((40, 100, 1), "peierls", cheb_ref.VEC_Z4),           # complex blocks and vectors, ragged last window
    ((30, 30, 1), "periodic", cheb_ref.VEC_RADEMACHER),   # torus: the units' neighbours wrap around in both directions
    ((37, 30, 1), "periodic_x", cheb_ref.VEC_Z4),         # ring of planes only
    ((48, 50, 1), "potential", cheb_ref.VEC_RADEMACHER),  # streamed on-site blocks (OS = 1), real
    ((33, 61, 1), "texture", cheb_ref.VEC_Z4),            # ... complex
    ((40, 36, 1), "ssd", cheb_ref.VEC_RADEMACHER),        # bond blocks streamed too (OS = 2)
])
def test_sweeps_of_a_chunk_in_one_launch_match_one_launch_per_sweep(api, solver_cls, knobs, block_storage, shape, kind, vec_kind):
    """cheb_march3 (sweep.hpp K7c; opt-in through BODGE_AMD_MARCH): all the sweeps of a 63-step reduction chunk in one
    launch, the waves claiming (sweep, lane group, unit) tasks and waiting on their neighbours' flags - by ticket (1),
    with a fixed unit per wave (3), or one sweep per launch for all lane groups (2).  The vectors are the bits
    cheb_sweep3 makes, so every d_n, e_n agrees with one launch per sweep to the rounding of the partial sums, and with
    the CPU oracle; runs that cross a chunk boundary, end on 1, 2 or 3 steps, one and two lane groups, several pairs."""
    if block_storage != "dictionary":
        pytest.skip("the stencil kernels read the block dictionary")
    if kind in ("potential", "texture"):
        system = _position_dependent_system(api, shape, kind)
    elif kind == "ssd":
        system = _ssd_system(api, shape, "ssd")
    else:
        system = _sweep_system(api, shape, kind)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    n = bsr.shape[0]
    complex_run = vec_kind == cheb_ref.VEC_Z4 or kind in ("peierls", "texture")
    per_group = 4 if complex_run else 8
    knobs.set("BODGE_AMD_SWEEP", "1")
    if kind in ("potential", "texture", "ssd"):
        knobs.set("BODGE_AMD_SWEEP_LANES", "4")  # (the chunk kernel exists for the 4-lane streamed forms only)
    with solver_cls.from_hamiltonian(system) as dev:
        for steps, vectors, extra in [(9, per_group, {}), (70, 2 * per_group, {}), (8, 3, {}), (64, per_group + 1, {"BODGE_AMD_SWEEP_SEGMENTS": "3"}),
                                      (13, 5 * per_group, {}), (7, 2 * per_group, {"BODGE_AMD_SWEEP_GEN": "0"}),
                                      (10, 2 * per_group, {"BODGE_AMD_SWEEP_LANES": "2"})]:
            if kind in ("potential", "texture", "ssd") and "BODGE_AMD_SWEEP_LANES" in extra:
                continue  # (streamed blocks: 4 lanes per site only)
            knobs.update(extra)
            knobs.set("BODGE_AMD_MARCH", "0")
            classic = dev.dots_random(scale, steps, vectors, seed=11, kind=vec_kind)
            base = dev.perf()
            assert base["steps_per_launch"] == 3 and base["persistent"] == 0 and base["sweeps"] == base["launches"], base
            ref = None
            if steps <= 13:
                ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(n, 11, range(vectors), vec_kind))
            for mode in ("1", "3", "2"):
                knobs.set("BODGE_AMD_MARCH", mode)
                got = dev.dots_random(scale, steps, vectors, seed=11, kind=vec_kind)
                perf = dev.perf()
                again = dev.dots_random(scale, steps, vectors, seed=11, kind=vec_kind)
                assert perf["persistent"] == 1 and perf["steps_per_launch"] == 3 and perf["sweeps"] == base["sweeps"], (mode, perf)
                assert perf["lanes_per_row"] == base["lanes_per_row"] and perf["onsite_streamed"] == base["onsite_streamed"]
                if mode != "2" and steps >= 6:
                    assert perf["launches"] < perf["sweeps"], (mode, perf)
                assert abs(perf["bytes_moved"] - base["bytes_moved"]) <= 1e-9 * base["bytes_moved"]
                assert np.array_equal(got[0], again[0]) and np.array_equal(got[1], again[1])  # whichever wave ran which task
                assert np.abs(got[0] - classic[0]).max() <= 1e-13 * n and np.abs(got[1] - classic[1]).max() <= 1e-13 * n, mode
                if ref is not None:
                    assert np.abs(got[0] - ref[0]).max() <= 1e-12 * n and np.abs(got[1] - ref[1]).max() <= 1e-12 * n, mode
            for key in extra:
                knobs.unset(key)


def test_a_persistent_launch_that_gives_up_waiting_is_repeated_sweep_by_sweep(api, solver_cls, knobs, block_storage):
    """A wave of cheb_march3 that cannot get its neighbours' flags within the timeout (a foreign kernel holding the GPU, a
    grid that is not resident with the fixed assignment) raises the abort word; every wave leaves, the handle goes back to
    one launch per sweep and the call is repeated.  Forced here by a debug bit: the result is the classic one, bit for bit."""
    if block_storage != "dictionary":
        pytest.skip("the stencil kernels read the block dictionary")
    system = _sweep_system(api, (64, 48, 1), "swave")
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    knobs.set("BODGE_AMD_SWEEP", "1")
    with solver_cls.from_hamiltonian(system) as dev:
        knobs.set("BODGE_AMD_MARCH", "0")
        classic = dev.dots_random(scale, 20, 16, seed=3)
        knobs.set("BODGE_AMD_MARCH", "1")
        knobs.set("BODGE_AMD_MARCH_DEBUG", "32")
        got = dev.dots_random(scale, 20, 16, seed=3)
        perf = dev.perf()
        assert perf["persistent"] == 0 and perf["sweeps"] == perf["launches"], perf
        assert np.array_equal(got[0], classic[0]) and np.array_equal(got[1], classic[1])
        knobs.unset("BODGE_AMD_MARCH_DEBUG")
        later = dev.dots_random(scale, 20, 16, seed=3)  # the handle keeps to one launch per sweep
        assert dev.perf()["persistent"] == 0 and np.array_equal(later[0], classic[0])


def _ssd_system(api, shape, kind, seed=0):
    """Every term of the s-wave model scaled by a position-dependent factor: "ssd" = the reference's sine-squared
    deformation (ref hamiltonian.py:488-531: φ at the site for on-site terms, at the bond midpoint for hopping),
    "bond_disorder" = random hopping amplitudes and spin splitting on every bond, "bond_phases" = the same with a Peierls phase on
    every bond and an exchange field of random direction on every site (complex blocks), "ssd_dwave" = ssd on a model with
    pairing on the bonds (bond blocks no longer diagonal)."""
    lattice = api.CubicLattice(shape)
    system = api.Hamiltonian(lattice)
    rng = np.random.default_rng(seed)
    sites = np.stack(np.unravel_index(np.arange(lattice.size), shape), axis=-1)
    pairs = lattice.bond_array(coords=True)
    φ = api.ssd(system)
    with system as (H, Δ):
        if kind in ("ssd", "ssd_dwave"):
            on_site, on_bond = φ(sites, sites)[:, None, None], φ(pairs[:, 0], pairs[:, 1])[:, None, None]
            H.set_sites(on_site * (3.0 * api.σ0 - 0.05 * api.σ3))
            H.set_bonds(-on_bond * api.σ0)
            if kind == "ssd":
                Δ.set_sites(-0.1 * on_site * api.jσ2)
            else:
                Δ.set_bonds(-0.1 * on_bond * api.dwave()(pairs[:, 0], pairs[:, 1]))
        else:
            idx = lattice.bond_array()
            lo, hi = idx.min(axis=1), idx.max(axis=1)  # the same amplitude both ways
            t = (0.8 + 0.4 * ((lo * 7919 + hi * 104729) % 1009) / 1009.0)[:, None, None]
            dt = (0.1 * ((lo * 31 + hi * 17) % 101) / 101.0)[:, None, None]
            if kind == "bond_phases":  # a Peierls phase of its own on every bond: a phase one way, its conjugate back
                θ = 2 * np.pi * ((lo * 271 + hi * 65537) % 997) / 997.0
                t = t * np.exp(1j * np.where(idx[:, 1] > idx[:, 0], θ, -θ))[:, None, None]
                th, ph = rng.uniform(0, np.pi, lattice.size)[:, None, None], rng.uniform(0, 2 * np.pi, lattice.size)[:, None, None]
                H.set_sites(3.0 * api.σ0 - 0.3 * (np.sin(th) * np.cos(ph) * api.σ1 + np.sin(th) * np.sin(ph) * api.σ2 + np.cos(th) * api.σ3))
            else:
                H.set_sites((3.0 + rng.uniform(-0.5, 0.5, lattice.size))[:, None, None] * api.σ0)
            Δ.set_sites(-0.1 * api.jσ2)
            H.set_bonds(-t * api.σ0 + dt * api.σ3)
    return system


@pytest.mark.parametrize("shape,kind", [((48, 50, 1), "ssd"), ((33, 61, 1), "bond_disorder"), ((30, 1, 44), "ssd"),
                                        ((33, 61, 1), "bond_phases"), ((40, 1, 30), "bond_phases")])
def test_three_step_sweep_with_streamed_bond_blocks(api, solver_cls, knobs, block_storage, shape, kind):
    """cheb_sweep3<., 4, ., ., OS = 2>: matrices in which the BOND blocks differ from bond to bond as well - what the
    reference's `ssd()` makes of a model, bond disorder, the Peierls phases of a position-dependent gauge - as long as they are
    diagonal as 4x4 matrices (spin-diagonal hopping, no bond pairing).  No table: every site streams one record (on-site block
    + its four bond blocks: 128 bytes in real arithmetic, 224 in complex - round 4).  Against the oracle and the one-step
    kernels; complex start vectors on a real matrix take the complex records; bond pairing must fall back by itself."""
    system = _ssd_system(api, shape, kind)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    n = bsr.shape[0]
    complex_matrix = kind == "bond_phases"
    per_group = 4 if complex_matrix else 8
    with solver_cls.from_hamiltonian(system) as dev:
        for steps, vectors, extra in [(9, per_group, {}), (7, 3, {}), (8, per_group + 3, {"BODGE_AMD_SWEEP_ZIGZAG": "0"}),
                                      (6, 2, {"BODGE_AMD_ALTERNATE": "0", "BODGE_AMD_SWEEP_SEGMENTS": "3"}),
                                      (9, per_group, {"BODGE_AMD_SWEEP_GEN": "0"}), (1, 3, {}), (2, per_group, {})]:
            ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(n, 5, range(vectors)))
            knobs.set("BODGE_AMD_SWEEP", "0")
            one = dev.dots_random(scale, steps, vectors, seed=5)
            assert dev.perf()["steps_per_launch"] == 1 and dev.perf()["onsite_streamed"] == 0
            knobs.set("BODGE_AMD_SWEEP", "1")
            knobs.update(extra)
            got = dev.dots_random(scale, steps, vectors, seed=5)
            perf = dev.perf()
            again = dev.dots_random(scale, steps, vectors, seed=5)
            for key in extra:
                knobs.unset(key)
            if block_storage == "dictionary":
                # (the envelope is symmetric: on a small lattice its bond blocks may still number <= 254 and fit a table)
                assert perf["onsite_streamed"] in ((2,) if kind != "ssd" or shape == (48, 50, 1) else (1, 2)), perf
                assert perf["steps_per_launch"] == 3 and perf["lanes_per_row"] == (4 if perf["onsite_streamed"] == 2 else 2), perf
                assert perf["real_arithmetic"] == (0 if complex_matrix else 1)
                if perf["onsite_streamed"] == 2:
                    assert perf["launches"] == -(-vectors // per_group) * -(-steps // 3)
            else:
                assert perf["onsite_streamed"] == 0 and perf["steps_per_launch"] == 1
            assert np.array_equal(got[0], again[0]) and np.array_equal(got[1], again[1])
            for other in (ref, one):
                assert np.abs(got[0] - other[0]).max() <= 1e-12 * n and np.abs(got[1] - other[1]).max() <= 1e-12 * n
        # complex start vectors: the complex records (round 3: one step per launch)
        ref = cheb_ref.recurrence_dots(bsr, scale, 12, cheb_ref.random_block(n, 5, range(3), cheb_ref.VEC_Z4))
        got = dev.dots_random(scale, 6, 3, seed=5, kind=cheb_ref.VEC_Z4)
        perf = dev.perf()
        assert perf["real_arithmetic"] == 0 and perf["steps_per_launch"] == (3 if block_storage == "dictionary" else 1), perf
        assert np.abs(got[0] - ref[0]).max() <= 1e-12 * n and np.abs(got[1] - ref[1]).max() <= 1e-12 * n
    # pairing on the bonds: the bond blocks are not diagonal, every block is distinct - one step per launch
    other = _ssd_system(api, (24, 30, 1), "ssd_dwave")
    bsr = other.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    ref = cheb_ref.recurrence_dots(bsr, scale, 10, cheb_ref.random_block(bsr.shape[0], 5, range(4)))
    with solver_cls.from_hamiltonian(other) as dev:
        got = dev.dots_random(scale, 5, 4, seed=5)
        assert dev.perf()["steps_per_launch"] == 1 and dev.perf()["onsite_streamed"] == 0
    assert np.abs(got[0] - ref[0]).max() <= 1e-12 * bsr.shape[0] and np.abs(got[1] - ref[1]).max() <= 1e-12 * bsr.shape[0]


def test_streamed_onsite_blocks_need_exactly_hermitian_diagonal_blocks_and_few_bond_blocks(api, solver_cls, knobs, block_storage):
    """The packed on-site record assumes A = A^†, C = B^† to the last bit.  A diagonal block that is
    Hermitian only to 1e-9 (it passes the reference's 1e-6 test, ref hamiltonian.py:121-122), a missing
    diagonal block, and bonds with more than 254 distinct blocks: results stay right, and only the
    middle case may still take the streamed-on-site sweep."""
    shape = (40, 36, 1)
    rng = np.random.default_rng(3)
    n_sites = shape[0] * shape[1]
    knobs.set("BODGE_AMD_SWEEP", "1")

    def run(system):
        bsr = system.matrix("bsr")
        scale = cheb_ref.spectral_bound(bsr)
        ref = cheb_ref.recurrence_dots(bsr, scale, 14, cheb_ref.random_block(bsr.shape[0], 2, range(4), cheb_ref.VEC_Z4))
        with solver_cls.from_hamiltonian(system) as dev:
            got = dev.dots_random(scale, 7, 4, seed=2, kind=cheb_ref.VEC_Z4)
            perf = dev.perf()
        assert np.abs(got[0] - ref[0]).max() <= 1e-12 * bsr.shape[0] and np.abs(got[1] - ref[1]).max() <= 1e-12 * bsr.shape[0]
        return perf

    lattice = api.CubicLattice(shape)
    v = rng.uniform(-0.5, 0.5, n_sites)[:, None, None]
    # (a) anti-Hermitian dust on the diagonal blocks: not packable, one-step kernels
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        H.set_sites((3.0 + v) * api.σ0 + 1e-9j * api.σ1)
        Δ.set_sites(-0.1 * api.jσ2)
        H.set_bonds(-1.0 * api.σ0)
    perf = run(system)
    assert perf["onsite_streamed"] == 0 and perf["steps_per_launch"] == 1
    # (b) a few sites without any on-site term: their diagonal block is not stored at all
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        keep = (rng.random(n_sites) > 0.02)[:, None, None]
        H.set_sites(np.where(keep, (3.0 + v) * api.σ0, 0.0 * api.σ0))
        Δ.set_sites(np.where(keep, -0.1 * api.jσ2, 0.0 * api.jσ2))
        H.set_bonds(-1.0 * api.σ0)
    assert system.bsr_arrays()[1].size < 5 * n_sites - 2 * (shape[0] + shape[1])
    perf = run(system)
    assert perf["onsite_streamed"] == (1 if block_storage == "dictionary" else 0)
    # (c) every bond different as well and not spin-diagonal: nothing to put in a table, nothing to stream per bond
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        H.set_sites((3.0 + v) * api.σ0)
        Δ.set_sites(-0.1 * api.jσ2)
        pairs = lattice.bond_array()  # directed pairs: the hopping must be the same both ways
        lo, hi = pairs.min(axis=1), pairs.max(axis=1)
        t = (0.8 + 0.4 * ((lo * 7919 + hi * 104729) % 1009) / 1009.0)[:, None, None]
        H.set_bonds(-t * api.σ0 + 0.1 * t * api.σ1)
    perf = run(system)
    assert perf["onsite_streamed"] == 0 and perf["steps_per_launch"] == 1


def test_stencil_kernels_step_aside_when_the_block_table_exceeds_lds(api, solver_cls, knobs, block_storage):
    """192 sites with random on-site terms: 193 distinct blocks - a dictionary, but in complex arithmetic
    its table (13 LDS slots per block) is beyond the 32 KB the kernels give it.  Asked for the stencil
    kernels, such a matrix must take the one-step kernels (found by scratch/fuzz_sweep.py: it raised)."""
    rng = np.random.default_rng(11)
    lattice = api.CubicLattice((8, 24, 1))
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        H.set_sites(rng.normal(size=(lattice.size, 1, 1)) * api.σ0 + 0.1 * rng.normal(size=(lattice.size, 1, 1)) * api.σ3)
        Δ.set_sites(-0.1 * api.jσ2)
        H.set_bonds(-1.0 * api.σ0)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    n = bsr.shape[0]
    knobs.set("BODGE_AMD_SWEEP", "1")
    with solver_cls.from_hamiltonian(system) as dev:
        for kind, expect_sweep in ((cheb_ref.VEC_Z4, False), (cheb_ref.VEC_RADEMACHER, block_storage == "dictionary")):
            got = dev.dots_random(scale, 7, 5, seed=2, kind=kind)
            assert (dev.perf()["steps_per_launch"] == 3) == expect_sweep
            ref = cheb_ref.recurrence_dots(bsr, scale, 14, cheb_ref.random_block(n, 2, range(5), kind))
            assert np.abs(got[0] - ref[0]).max() <= 1e-12 * n and np.abs(got[1] - ref[1]).max() <= 1e-12 * n


@pytest.mark.parametrize("sweep", ["0", "1"])
def test_batches_of_one_call_enqueued_back_to_back_change_no_bit(api, solver_cls, knobs, block_storage, sweep):
    """A call with more vectors than one launch carries is cut into batches; whole matrices enqueue
    them back to back (own timing events, own piece of the pinned result buffer) and wait once.
    Same numbers as one batch at a time, for random and for unit starts, ragged last batch included."""
    system = _sweep_system(api, (64, 48, 1), "swave")
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    rows = np.arange(0, 4 * 64 * 48, 97)[:70]  # 70 unit vectors: two batches of the one-step kernels
    knobs.set("BODGE_AMD_SWEEP", sweep)
    if sweep == "0":
        knobs.set("BODGE_AMD_BATCH", "32")  # (the stencil kernels cut their own batches: one lane group each)
    with solver_cls.from_hamiltonian(system) as dev:
        dev.set_lattice_shape((64, 48, 1))
        queued_random = dev.dots_random(scale, 7, 27, seed=3)
        side_by_side = dev.perf()
        queued = queued_random, dev.dots_unit(scale, 6, rows)
        launches = dev.perf()["launches"]
        # the marching kernels run two batches at a time on two streams (their launches fill each other's idle ends),
        # the one-step kernels stay on one; the window both streams share is shorter than their summed times
        marching = sweep == "1" and block_storage == "dictionary"
        assert (side_by_side["steps_per_launch"] == 3) == marching
        assert side_by_side["streams"] == (2 if marching else 1), side_by_side
        assert 0 < side_by_side["window_ms"] and (not marching or side_by_side["window_ms"] < side_by_side["kernel_ms"])
        knobs.set("BODGE_AMD_STREAMS", "3")
        three = dev.dots_random(scale, 7, 27, seed=3)
        assert dev.perf()["streams"] == (3 if marching else 1)  # (the 27 vectors are one batch of the one-step kernels)
        assert np.array_equal(three[0], queued_random[0]) and np.array_equal(three[1], queued_random[1])
        knobs.unset("BODGE_AMD_STREAMS")
        knobs.set("BODGE_AMD_NO_BATCH_PIPELINE", "1")
        single = dev.dots_random(scale, 7, 27, seed=3), dev.dots_unit(scale, 6, rows)
        assert dev.perf()["launches"] == launches and dev.perf()["streams"] == 1
    for a, b in zip(queued, single):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ref = cheb_ref.recurrence_dots(bsr, scale, 14, cheb_ref.random_block(bsr.shape[0], 3, range(27), cheb_ref.VEC_RADEMACHER))
    assert np.abs(queued[0][0] - ref[0]).max() <= 1e-12 * bsr.shape[0] and np.abs(queued[0][1] - ref[1]).max() <= 1e-12 * bsr.shape[0]


@pytest.mark.parametrize("shape,model,knob_set,sites,steps", [
    ((64, 48, 1), "swave", {}, [(30, 20, 0)], 50),                                   # K7b: the band reaches both ends of x
    ((64, 48, 1), "swave", {}, [(0, 0, 0), (63, 47, 0), (31, 5, 0)], 10),            # corners: clipped bands, several sites
    ((64, 48, 1), "swave", {"BODGE_AMD_SWEEP_LANES": "4"}, [(12, 40, 0), (50, 3, 0)], 17),
    ((64, 48, 1), "swave", {}, [(31, 40, 0), (33, 3, 0)], 11),                        # a narrow band all the way
    ((64, 48, 1), "peierls", {}, [(40, 11, 0)], 23),                                 # complex blocks: ComplexPHMode
    ((64, 48, 1), "swave", {"BODGE_AMD_SWEEP_STEPS": "2"}, [(20, 20, 0)], 21),        # K7
    ((30, 30, 1), "periodic", {}, [(3, 3, 0)], 12),                                  # ring of planes: no band
    ((40, 12, 14), "dwave", {}, [(20, 6, 7)], 30),                                   # K8
    ((40, 12, 14), "dwave", {}, [(0, 0, 0), (39, 11, 13)], 9),
])
def test_unit_start_vectors_on_the_stencil_kernels_with_a_band_of_planes(api, solver_cls, knobs, block_storage, shape, model, knob_set, sites, steps):
    """Unit start vectors (LDOS: ref hamiltonian.py:341-387 restated as a Chebyshev resolvent) take the
    lattice-stencil kernels as well.  t_n of a unit vector at plane x_s is zero outside planes x_s +- n, so
    a launch advances only that band of planes and the four rotating buffers hold zeros outside it.  Same
    dot products as the one-step kernels to round-off and as the oracle; fewer bytes than whole launches
    until the band has reached both ends."""
    system = _sweep_system(api, shape, model)
    bsr = system.matrix("bsr")
    n = bsr.shape[0]
    scale = cheb_ref.spectral_bound(bsr)
    rows = np.array([4 * ((x * shape[1] + y) * shape[2] + z) + (i % 4) for i, (x, y, z) in enumerate(sites)])
    for key, value in knob_set.items():
        knobs.set(key, value)
    with solver_cls.from_hamiltonian(system) as dev:
        dev.set_lattice_shape(shape)
        knobs.set("BODGE_AMD_SWEEP", "0")
        plain = dev.dots_unit(scale, steps, rows)
        assert dev.perf()["steps_per_launch"] == 1 and dev.perf()["rolling"] == 0
        knobs.set("BODGE_AMD_SWEEP", "1")
        dev.dots_random(scale, 5, 8, seed=1)  # (leaves all four buffers full of another run's vectors)
        banded = dev.dots_unit(scale, steps, rows)
        perf = dev.perf()
        again = dev.dots_unit(scale, steps, rows)
    three_d = shape[1] > 1 and shape[2] > 1
    stencil = block_storage == "dictionary"  # (the stencil forms read the block dictionary; without it: one-step kernels, row band)
    assert perf["rolling"] == (1 if three_d and stencil else 0)
    assert perf["steps_per_launch"] == (1 if three_d or not stencil else int(knob_set.get("BODGE_AMD_SWEEP_STEPS", 3)))
    whole = perf["launches"] * perf["bytes_per_launch"]
    if not stencil:
        pass
    elif model == "periodic":
        assert perf["bytes_moved"] > 0.8 * whole
    elif max(x for x, _, _ in sites) - min(x for x, _, _ in sites) + 2 * steps < shape[0] // 2:  # (the last band is half the planes)
        assert perf["bytes_moved"] < 0.6 * whole, (perf["bytes_moved"], whole)
    for a, b, c in zip(plain, banded, again):
        assert np.array_equal(b, c)
        assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(a).max())
    ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.unit_block(n, rows))
    assert np.abs(banded[0] - ref[0]).max() <= 1e-13 and np.abs(banded[1] - ref[1]).max() <= 1e-13


@pytest.mark.parametrize("shape,model,knob_set,steps", [
    ((64, 48, 1), "swave", {"BODGE_AMD_SWEEP": "1"}, 7),                                 # K7b: 3 + 3 + 1
    ((64, 48, 1), "swave", {"BODGE_AMD_SWEEP": "1"}, 6),                                 # K7b: 3 + 3
    ((64, 48, 1), "swave", {"BODGE_AMD_SWEEP": "1", "BODGE_AMD_SWEEP_STEPS": "2"}, 5),   # K7: 2 + 2 + 1
    ((64, 48, 1), "swave", {"BODGE_AMD_SWEEP": "0"}, 5),                                 # K1 dictionary
    ((64, 48, 1), "swave", {"BODGE_AMD_SWEEP": "0", "BODGE_AMD_DICT": "0"}, 5),          # K1 streamed
    ((12, 14, 16), "dwave", {"BODGE_AMD_SWEEP": "1"}, 5),                                # K8
])
def test_last_launch_of_a_run_stores_no_vectors_and_changes_no_bit(api, solver_cls, knobs, shape, model, knob_set, steps):
    """Nothing reads the vectors of the last step of a run (the calls return dot products), so the last
    launch does not store them.  Same bits as with the stores (BODGE_AMD_KEEP_LAST=1), also in a second
    call on the same handle (whose buffers the first one left half written), and the perf record counts
    the bytes that were not moved."""
    system = _sweep_system(api, shape, model)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    for key, value in knob_set.items():
        knobs.set(key, value)
    with solver_cls.from_hamiltonian(system) as dev:
        dev.set_lattice_shape(shape)
        lean = dev.dots_random(scale, steps, 8, seed=5)
        lean_perf = dev.perf()
        again = dev.dots_random(scale, steps, 8, seed=5)
        knobs.set("BODGE_AMD_KEEP_LAST", "1")
        kept = dev.dots_random(scale, steps, 8, seed=5)
        kept_perf = dev.perf()
    for a, b, c in zip(lean, again, kept):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert lean_perf["launches"] == kept_perf["launches"]
    assert 0 < lean_perf["bytes_moved"] < kept_perf["bytes_moved"] <= kept_perf["launches"] * kept_perf["bytes_per_launch"]
    n = bsr.shape[0]
    ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(n, 5, range(8), cheb_ref.VEC_RADEMACHER))
    assert np.abs(lean[0] - ref[0]).max() <= 1e-12 * n and np.abs(lean[1] - ref[1]).max() <= 1e-12 * n


@pytest.mark.parametrize("name,n_vectors,kind", [
    ("random357", 6, cheb_ref.VEC_Z4),          # complex blocks, periodic
    ("complex235", 3, cheb_ref.VEC_RADEMACHER),  # complex H with real start vectors
    ("swave20", 9, cheb_ref.VEC_RADEMACHER),    # real
    ("dwave8", 64, cheb_ref.VEC_RADEMACHER),    # real, 3-D
])
def test_particle_hole_packed_storage_is_exact(api, solver_cls, knobs, name, n_vectors, kind):
    """Blocks built by the Hamiltonian have the form [[A,B],[C,-A*]]; storing 12 of 16 entries
    must give bit-identical dot products (same products, same order)."""
    system = _build(api, name)
    scale = cheb_ref.spectral_bound(system.matrix("bsr"))
    with solver_cls.from_hamiltonian(system) as dev:
        packed = dev.dots_random(scale, 16, n_vectors, seed=3, kind=kind)
        assert dev.perf()["ph_packed"] == 1
        knobs.set("BODGE_AMD_PH", "0")
        full = dev.dots_random(scale, 16, n_vectors, seed=3, kind=kind)
        assert dev.perf()["ph_packed"] == 0
    assert np.array_equal(packed[0], full[0]) and np.array_equal(packed[1], full[1])


def test_dictionary_form_matches_streamed_form(api, solver_cls, knobs, block_storage):
    """Few distinct blocks -> table in LDS + 8 B per stored block; must agree with the streamed
    kernels to round-off (same products per row; only the cross-workgroup sum order differs)."""
    if block_storage == "streamed":
        pytest.skip("compares both forms itself")
    for name, n_vectors, kind in [("swave20", 8, cheb_ref.VEC_RADEMACHER), ("dwave8", 16, cheb_ref.VEC_RADEMACHER),
                                  ("chain128", 5, cheb_ref.VEC_Z4), ("snf", 64, cheb_ref.VEC_Z4)]:
        system = _build(api, name)
        scale = cheb_ref.spectral_bound(system.matrix("bsr"))
        with solver_cls.from_hamiltonian(system) as dev:
            table = dev.dots_random(scale, 16, n_vectors, seed=3, kind=kind)
            n_unique = dev.perf()["dict_blocks"]
            assert 0 < n_unique < 64, name
            knobs.set("BODGE_AMD_DICT", "0")
            streamed = dev.dots_random(scale, 16, n_vectors, seed=3, kind=kind)
            assert dev.perf()["dict_blocks"] == 0
            knobs.unset("BODGE_AMD_DICT")
        assert np.allclose(table[0], streamed[0], rtol=1e-13, atol=0)
        assert np.allclose(table[1], streamed[1], rtol=1e-12, atol=1e-12 * system.shape[0])
    # every block distinct: the dictionary is not built and the streamed kernels run
    system = _build(api, "random357")
    with solver_cls.from_hamiltonian(system) as dev:
        dev.dots_random(1.0 + cheb_ref.spectral_bound(system.matrix("bsr")), 2, 8, kind=cheb_ref.VEC_Z4)
        assert dev.perf()["dict_blocks"] == 0 and dev.perf()["pipelined"] == 1


def test_matrix_without_particle_hole_form_uses_full_storage(api, solver_cls):
    """A hand-made BSR matrix (Hermitian, but blocks not of Nambu form) must not be packed."""
    import scipy.sparse as sp

    rng = np.random.default_rng(3)
    blocks = rng.standard_normal((3, 4, 4)) + 1j * rng.standard_normal((3, 4, 4))
    diag0, diag1 = blocks[0] + blocks[0].conj().T, blocks[1] + blocks[1].conj().T
    data = np.array([diag0, blocks[2], blocks[2].conj().T, diag1])
    indptr, indices = np.array([0, 2, 4], dtype=np.int32), np.array([0, 1, 0, 1], dtype=np.int32)
    mat = sp.bsr_matrix((data, indices, indptr), shape=(8, 8))
    scale = 1.01 * np.abs(mat.toarray()).sum(axis=1).max()
    ref = cheb_ref.recurrence_dots(mat, scale, 20, cheb_ref.random_block(8, 1, range(4), cheb_ref.VEC_Z4))
    with solver_cls(indptr, indices, data) as dev:
        got = dev.dots_random(scale, 10, 4, seed=1, kind=cheb_ref.VEC_Z4)
        assert dev.perf()["ph_packed"] == 0 and dev.perf()["real_arithmetic"] == 0
    assert np.allclose(got[0], ref[0], atol=1e-12) and np.allclose(got[1], ref[1], atol=1e-12)


@pytest.mark.parametrize("n_vectors,kind,real", [
    (3, cheb_ref.VEC_Z4, False),         # complex, 4 lanes per row: 16 rows x ~60 blocks per wave tile
    (8, cheb_ref.VEC_RADEMACHER, True),  # real matrix, real arithmetic
    (40, cheb_ref.VEC_Z4, False),        # wide batch, 64 lanes per row
])
def test_general_matrix_with_long_rows_is_staged_in_chunks(solver_cls, n_vectors, kind, real):
    """A non-lattice Hermitian BSR matrix whose block rows hold 40-80 distinct blocks: more than
    a wave's LDS staging region takes at once, so the generic kernel passes them through in chunks."""
    import scipy.sparse as sp

    rng = np.random.default_rng(11)
    nb = 150
    mask = rng.random((nb, nb)) < 0.22
    mask |= mask.T
    np.fill_diagonal(mask, True)
    dense = rng.standard_normal((4 * nb, 4 * nb)) + (0 if real else 1j) * rng.standard_normal((4 * nb, 4 * nb))
    dense = (dense + dense.conj().T) * np.kron(mask, np.ones((4, 4)))
    mat = sp.bsr_matrix(dense.astype(np.complex128), blocksize=(4, 4))
    mat.sort_indices()
    assert np.diff(mat.indptr).max() > 40
    scale = 1.01 * np.abs(dense).sum(axis=1).max()
    start = cheb_ref.random_block(4 * nb, 2, range(n_vectors), kind)
    ref = cheb_ref.recurrence_dots(mat, scale, 16, start)
    x = rng.standard_normal(4 * nb) + 1j * rng.standard_normal(4 * nb)
    with solver_cls(mat.indptr, mat.indices, mat.data) as dev:
        assert np.allclose(dev.spmv(x), dense @ x, rtol=1e-12, atol=1e-12 * np.abs(dense @ x).max())
        got = dev.dots_random(scale, 8, n_vectors, seed=2, kind=kind)
        perf = dev.perf()
        assert perf["dict_blocks"] == 0 and perf["pipelined"] == 0 and perf["real_arithmetic"] == int(real)
    assert np.allclose(got[0], ref[0], rtol=0, atol=1e-12 * 4 * nb)
    assert np.allclose(got[1], ref[1], rtol=0, atol=1e-12 * 4 * nb)


@pytest.mark.parametrize("name,kind,lane_options", [
    ("dwave8", cheb_ref.VEC_RADEMACHER, (4, 8, 16, 32)),  # real arithmetic: two vectors per lane
    ("random357", cheb_ref.VEC_Z4, (4, 8, 16, 32, 64)),   # complex arithmetic
])
def test_lanes_override_gives_same_numbers(api, solver_cls, name, kind, lane_options):
    system = _build(api, name)
    scale = cheb_ref.spectral_bound(system.matrix("bsr"))
    with solver_cls.from_hamiltonian(system) as dev:
        base = dev.dots_random(scale, 10, 4, seed=1, kind=kind)
        for lanes in lane_options:
            dev.set_lanes_per_row(lanes)
            other = dev.dots_random(scale, 10, 4, seed=1, kind=kind)
            assert dev.perf()["lanes_per_row"] == lanes
            assert np.allclose(other[0], base[0], rtol=1e-13) and np.allclose(other[1], base[1], rtol=1e-13, atol=1e-9)


# ------------------------------------------------------------ free energy / F
@pytest.mark.parametrize("name", ["swave20", "swave20_zeeman", "snf", "complex235", "random357", "chain128"])
def test_free_energy_dense_matches_reference(api, golden, name):
    """Mirrors ref tests/test_hamiltonian.py:421-424 (accelerator vs CPU) with recorded CPU values."""
    system = _build(api, name)
    for temperature in systems.CATALOG[name]["temps"]:
        value = system.free_energy(float(temperature), method="dense")
        assert np.isclose(value, golden.free_energy(name, temperature), rtol=1e-10, atol=0)
    with pytest.raises(ValueError):
        system.free_energy(-1.0)


@pytest.mark.parametrize("name,temperature,moments", [
    ("swave20", 0.5, 128),
    ("swave20", 1.0, 64),
    ("swave20", 0.1, 512),
    ("snf", 1.0, 64),
    ("complex235", 1.0, 192),
    ("barrier", 0.1, 1400),
])
def test_free_energy_chebyshev_exact_trace_matches_reference(api, golden, name, temperature, moments):
    system = _build(api, name)
    value = system.free_energy(temperature, method="chebyshev", trace="exact", moments=moments)
    assert np.isclose(value, golden.free_energy(name, temperature), rtol=1e-10, atol=0)


def test_free_energy_default_moment_rule_reaches_1e10(api, golden):
    system = _build(api, "snf")
    for temperature in (0.1, 1.0):
        value = system.free_energy(temperature, method="chebyshev", trace="exact")
        assert np.isclose(value, golden.free_energy("snf", temperature), rtol=1e-10, atol=0)


@pytest.mark.parametrize("name", ["swave20_zeeman", "snf"])
def test_zero_temperature_chebyshev_on_a_gapped_spectrum(api, golden, name):
    """T = 0 beyond dense reach: |ε| is not analytic, but on a gapped spectrum the expansion of
    -(ε/4)·erf(ε/δ), δ = gap/5 (gap from Lanczos / the dense solver), reproduces the dense T = 0 value
    - to 1e-12 with 8·a/δ moments, a quarter of what the finite-temperature surrogate of round 1
    needed for 1e-10; the plain T = 0 coefficients with the same number of moments do not."""
    system = _build(api, name)
    exact = golden.free_energy(name, 0.0)
    value = system.free_energy(0.0, method="chebyshev")
    assert np.isclose(value, exact, rtol=1e-12, atol=0)
    plain = system.free_energy(0.0, method="chebyshev", gap_surrogate=False)
    assert abs(plain - exact) > abs(value - exact) and np.isclose(plain, exact, rtol=1e-6)


def test_free_energy_stochastic_matches_oracle_on_same_vectors(api):
    system = systems.swave_square(api, L=40)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    for kind, label in ((cheb_ref.VEC_RADEMACHER, "rademacher"), (cheb_ref.VEC_Z4, "z4")):
        ref = cheb_ref.free_energy_stochastic(bsr, 0.5, 128, n_vectors=8, seed=3, kind=kind, scale=scale)
        value = system.free_energy(0.5, method="chebyshev", trace="stochastic", moments=128, vectors=8,
                                   seed=3, vector_kind=label, scale=scale)
        assert np.isclose(value, ref, rtol=1e-10, atol=0)


def test_stochastic_free_energy_reports_a_meaningful_error(api, golden):
    """F from 64 random vectors with its standard error: the exact value lies within a few σ, the
    mean equals what `free_energy(trace="stochastic")` returns for the same seed, and more
    vectors shrink σ like 1/√R."""
    from bodge_amd.observables import free_energy_stochastic

    system = _build(api, "swave20_zeeman")
    exact = golden.free_energy("swave20_zeeman", 0.5)
    value, sigma = free_energy_stochastic(system, 0.5, vectors=64, seed=3)
    assert sigma > 0 and abs(value - exact) < 5 * sigma and sigma < 5e-3 * abs(exact)
    same = system.free_energy(0.5, method="chebyshev", trace="stochastic", vectors=64, seed=3)
    assert np.isclose(value, same, rtol=1e-12)
    _, sigma_few = free_energy_stochastic(system, 0.5, vectors=16, seed=3)
    assert 1.2 < sigma_few / sigma < 3.5  # ~2 = sqrt(64/16), up to the noise of the noise


def test_free_energy_like_reference_test(api):
    """ref tests/test_hamiltonian.py:428-464: F against the closed form over ±ε, T = 0, T < 0."""
    system = _build(api, "snf")
    eps, _ = system.diagonalize()
    both = np.hstack([-eps, +eps])
    for temperature in [0.01, 0.1, 1.0]:
        closed = -(temperature / 2) * np.sum(np.log(1 + np.exp(-both / temperature)))
        assert np.allclose(system.free_energy(temperature), closed)
    assert np.allclose(system.free_energy(0.0), 0.5 * np.sum(both[both < 0]))
    with pytest.raises(Exception):
        system.free_energy(-1.0)


# ----------------------------------------------------------------- diagonalize
@pytest.mark.parametrize("name", ["barrier", "complex235", "random357", "snf"])
def test_diagonalize_matches_reference(api, golden, name):
    system = _build(api, name)
    dense = np.asarray(system.matrix("dense"))
    vals, vecs = system.diagonalize(format="raw")
    ref = golden.eigenvalues(name)
    assert vals.shape == ref.shape and vecs.shape == (dense.shape[0], ref.size)
    assert np.all(vals > 0) and np.all(np.diff(vals) >= 0)
    assert np.allclose(vals, ref, rtol=0, atol=1e-10)
    assert np.allclose(dense @ vecs, vecs * vals, atol=1e-9)
    assert np.allclose(vecs.conj().T @ vecs, np.eye(ref.size), atol=1e-9)
    vals2, shaped = system.diagonalize()
    assert shaped.shape == (ref.size, system.lattice.size, 4)
    for n in (0, ref.size // 2, ref.size - 1):
        for site in (0, system.lattice.size - 1):
            assert np.allclose(shaped[n, site, :], vecs[4 * site : 4 * site + 4, n])
    with pytest.raises(Exception):
        system.diagonalize(format="foo")


def test_single_site_and_single_line_systems(api):
    """Smallest inputs: a 1x1x1 lattice (one 4x4 block; every axis degenerate, so the self-"edges"
    merge into the diagonal block, ref lattice.py:190-195) and a 1x1xL chain, through every observable."""
    for shape in [(1, 1, 1), (1, 1, 5)]:
        lattice = api.CubicLattice(shape)
        system = api.Hamiltonian(lattice)
        with system as (H, Δ):
            for i in lattice.sites():
                H[i, i] = 0.7 * api.σ0 + 0.2 * api.σ3
                Δ[i, i] = -0.3 * api.jσ2
            for i, j in lattice.bonds():
                H[i, j] = -1.0 * api.σ0
        dense = np.asarray(system.matrix("dense"))
        vals, vecs = system.diagonalize(format="raw")
        ref_vals, _ = dense_ref.diagonalize(dense, format="raw")
        assert np.allclose(vals, ref_vals, atol=1e-12) and np.allclose(dense @ vecs, vecs * vals, atol=1e-12)
        for temperature in (0.0, 0.3):
            for method in ("dense", "auto") + (("chebyshev",) if temperature > 0 else ()):
                assert np.isclose(system.free_energy(temperature, method=method),
                                  dense_ref.free_energy(dense, temperature), rtol=1e-10)
        site = tuple(s - 1 for s in shape)
        energies = [0.0, 0.5, 1.0]
        assert np.allclose(system.ldos(site, energies),
                           dense_ref.ldos(system.matrix("csc"), lattice[site], energies), rtol=1e-9, atol=1e-12)
        assert np.allclose(system.lowest_eigenvalues(1), ref_vals.min(), atol=1e-6)


def test_diagonalize_counts_2n_states(api):
    system = _build(api, "barrier")  # ref tests/test_hamiltonian.py:353
    vals, _ = system.diagonalize()
    assert vals.size == 2 * system.lattice.size


# ------------------------------------------------------------------------ LDOS
@pytest.mark.parametrize("name", ["ldos16", "random357", "chain128", "pwave31"])
def test_ldos_matches_reference(api, golden, name):
    system = _build(api, name)
    for n, (site, energies) in enumerate(systems.CATALOG[name]["ldos"]):
        rho = system.ldos(tuple(site), list(energies))
        assert np.allclose(rho, golden.ldos(name, n), rtol=1e-9, atol=1e-12)


def test_ldos_of_many_sites_in_one_call(api, golden):
    """A list of sites shares the launches (16 sites per batch); rows must equal per-site calls."""
    system = _build(api, "ldos16")
    sites = [(8, 8, 0), (0, 3, 0)] + [(x, 5, 0) for x in range(16)]
    energies = list(np.linspace(-0.3, 0.3, 7))
    table = system.ldos(sites, energies)
    assert table.shape == (18, 7)
    assert np.allclose(table[0], golden.ldos("ldos16", 0), rtol=1e-9, atol=1e-12)
    assert np.allclose(table[9], system.ldos((7, 5, 0), energies), rtol=1e-12)


def test_ldos_map_of_a_whole_lattice_uses_two_device_mirrors(api, golden):
    """All 256 sites in one call (1024 unit vectors): the rows are split over two device mirrors
    driven by two host threads; every row must still equal the single-site call."""
    system = _build(api, "ldos16")
    sites = [(x, y, 0) for x in range(16) for y in range(16)]
    energies = list(np.linspace(-0.3, 0.3, 7))
    table = system.ldos(sites, energies)
    assert table.shape == (256, 7) and len(system._devices) == 2
    assert np.allclose(table[sites.index((8, 8, 0))], golden.ldos("ldos16", 0), rtol=1e-9, atol=1e-12)
    for site in [(0, 0, 0), (7, 15, 0), (15, 15, 0)]:
        assert np.allclose(table[sites.index(site)], system.ldos(site, energies), rtol=1e-12)
    # the model is uniform: the map has the lattice's mirror symmetry
    grid = table.reshape(16, 16, 7)
    assert np.allclose(grid, grid[::-1], rtol=1e-9) and np.allclose(grid, grid.transpose(1, 0, 2), rtol=1e-9)


def test_ldos_is_positive_everywhere(api):
    """ref tests/test_hamiltonian.py:467-500 on a seeded random periodic metal."""
    system = systems.random_periodic(api, shape=(5, 5, 2), seed=21)
    energies = [0.0, 0.01, 0.10, 0.50, 1.00, 2.00, 4.00]
    for site in [(0, 0, 0), (2, 3, 1), (4, 4, 1), (1, 0, 1)]:
        assert np.all(system.ldos(site, energies) >= 0)


def test_c_abi_rejects_malformed_input(api, solver_cls, hip_library):
    """Error behaviour at the boundary: every malformed matrix or argument is refused with a
    message (ValueError for caller mistakes), nothing reaches a kernel."""
    rng = np.random.default_rng(0)
    good_ptr = np.array([0, 2, 4], dtype=np.int32)
    good_idx = np.array([0, 1, 0, 1], dtype=np.int32)
    data = rng.standard_normal((4, 4, 4)) + 0j
    cases = {
        "indptr does not span": (np.array([0, 2, 3], dtype=np.int32), good_idx),
        "not monotone": (np.array([0, 3, 2, 4], dtype=np.int32)[:3], good_idx),
        "out of range": (good_ptr, np.array([0, 1, 0, 2], dtype=np.int32)),
        "duplicate": (good_ptr, np.array([0, 0, 0, 1], dtype=np.int32)),
        "not sorted": (good_ptr, np.array([1, 0, 0, 1], dtype=np.int32)),
    }
    for fragment, (ptr, idx) in cases.items():
        with pytest.raises(ValueError) as err:
            solver_cls(ptr, idx, data)
        assert "bodge_hip" in str(err.value), fragment
    with solver_cls(good_ptr, good_idx, data) as dev:
        with pytest.raises(ValueError):
            dev.dots_random(-1.0, 4, 2)          # scale must be positive
        with pytest.raises(ValueError):
            dev.dots_random(1.0, 0, 2)           # at least one step
        with pytest.raises(ValueError):
            dev.dots_random(1.0, 4, 0)           # at least one vector
        with pytest.raises(ValueError):
            dev.moments_unit(1.0, 8, np.array([99], dtype=np.int64))  # row outside the matrix
        with pytest.raises(ValueError):
            dev.lanczos_advance(4)               # begin was not called
        assert np.isfinite(dev.dots_random(10.0, 4, 2)[0]).all()  # the handle is still usable


def test_handles_release_their_device_memory(api, solver_cls):
    """Create / use / destroy in a loop (recurrence, unit moments, Lanczos, dense eigensolver, slab
    group): free device memory must return to where it started."""
    import ctypes

    from bodge_amd.solver import SlabGroup

    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    system = _build(api, "swave20")
    scale = cheb_ref.spectral_bound(system.matrix("bsr"))

    def cycle():
        with solver_cls.from_hamiltonian(system) as dev:
            dev.dots_random(scale, 8, 24)
            dev.moments_unit(scale, 16, np.arange(40, dtype=np.int64))
            dev.lanczos_begin(4, seed=1, max_iter=64)
            dev.lanczos_advance(16)
            dev.eigh(vectors=True)
        with SlabGroup.from_hamiltonian(system, 3) as group:
            group.dots_random(scale, 4, 8)

    cycle()  # first use loads code objects and creates the runtime's own pools
    before = free_bytes()
    for _ in range(10):
        cycle()
    after = free_bytes()
    assert before - after < 8 << 20, f"{(before - after) >> 20} MiB of device memory not returned"


# ----------------------------------------------------- revision / re-upload
def test_device_copy_follows_with_block_updates(api):
    system = systems.swave_square(api, L=6, gap=0.0)
    before = system.free_energy(0.2)
    with system as (H, Δ):
        for i in system.lattice.sites():
            Δ[i, i] = 0.8 * api.jσ2
    after = system.free_energy(0.2)
    assert after < before - 1e-3
    dense = np.asarray(system.matrix("dense"))
    assert np.isclose(after, dense_ref.free_energy(dense, 0.2), rtol=1e-10)


# ------------------------------------------------------------ slab decomposition
@pytest.mark.parametrize("name,n_slabs,n_vectors,kind", [
    ("random357", 3, 4, cheb_ref.VEC_Z4),           # periodic: every slab has wrap-around halos
    ("dwave8", 4, 8, cheb_ref.VEC_RADEMACHER),      # 3-D real, 2 planes per slab
    ("dwave8", 8, 3, cheb_ref.VEC_Z4),              # one plane per slab
    ("swave20", 5, 64, cheb_ref.VEC_RADEMACHER),    # wide batch, real
    ("chain128", 7, 2, cheb_ref.VEC_Z4),            # complex chain, uneven slabs
])
def test_slab_group_matches_whole_matrix(api, solver_cls, name, n_slabs, n_vectors, kind):
    """Row slabs with halo exchange (same-process transport) against the undivided matrix
    and the oracle: the decomposition must not change any dot product."""
    from bodge_amd.solver import SlabGroup

    system = _build(api, name)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    steps = 20
    ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps,
                                   cheb_ref.random_block(bsr.shape[0], 6, range(2, 2 + n_vectors), kind))
    with solver_cls.from_hamiltonian(system) as whole:
        mono = whole.dots_random(scale, steps, n_vectors, seed=6, first_id=2, kind=kind)
    with SlabGroup.from_hamiltonian(system, n_slabs) as group:
        assert sum(p.n_own for p in group.plans) == system.lattice.size
        split = group.dots_random(scale, steps, n_vectors, seed=6, first_id=2, kind=kind)
        rows = np.array([3, 4 * (system.lattice.size // 2) + 1, system.shape[0] - 2])
        unit_split = group.dots_unit(scale, steps, rows)
    mu0 = bsr.shape[0]
    for got in (mono, split):
        assert np.allclose(got[0], ref[0], rtol=0, atol=1e-12 * mu0)
        assert np.allclose(got[1], ref[1], rtol=0, atol=1e-12 * mu0)
    unit_ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.unit_block(bsr.shape[0], rows))
    assert np.allclose(unit_split[0], unit_ref[0], rtol=0, atol=1e-13)
    assert np.allclose(unit_split[1], unit_ref[1], rtol=0, atol=1e-13)


@pytest.mark.parametrize("shape,kind,n_slabs,n_vectors,vec_kind,periodic_x", [
    ((32, 6, 8), "dwave", 3, 8, cheb_ref.VEC_RADEMACHER, False),   # slabs of 11 / 11 / 10 planes, 4 lanes per site
    ((27, 5, 7), "dwave", 2, 3, cheb_ref.VEC_RADEMACHER, False),   # 14 / 13 planes, 2 lanes per site (<= 4 real vectors)
    ((26, 6, 6), "peierls", 3, 6, cheb_ref.VEC_Z4, False),         # complex blocks, two batches of 4 and 2 vectors
    ((24, 6, 8), "swave", 3, 8, cheb_ref.VEC_RADEMACHER, True),    # ring of planes: slab 0 and slab 2 are neighbours too
])
def test_stencil_slabs_read_their_neighbours_planes_in_place(api, solver_cls, knobs, block_storage, shape, kind, n_slabs,
                                                             n_vectors, vec_kind, periodic_x):
    """Same-process slab groups of a 3-D lattice whose slabs are stacks of >= 8 whole x-planes run the
    rolling stencil kernel (cheb_roll3) and read the planes just outside a slab directly from the
    neighbouring member's buffer: no pack, copy or unpack.  Against the undivided matrix, the oracle,
    and the exchange-based form of the same group (stencil kernels off)."""
    from bodge_amd.solver import SlabGroup

    system = _sweep_system(api, shape, kind)
    if periodic_x:
        with system as (H, Δ):
            H.set_edges(-0.8 * api.σ0 + 0.1 * api.σ3, axis=0)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    n, steps = bsr.shape[0], 9
    ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(n, 6, range(n_vectors), vec_kind))
    with SlabGroup.from_hamiltonian(system, n_slabs) as group:
        assert all(plan.n_own // (shape[1] * shape[2]) >= 8 for plan in group.plans)
        knobs.set("BODGE_AMD_SWEEP", "0")
        exchanged = group.dots_random(scale, steps, n_vectors, seed=6, kind=vec_kind)
        assert all(member.perf()["rolling"] == 0 for member in group.members)
        knobs.set("BODGE_AMD_SWEEP", "1")
        in_place = group.dots_random(scale, steps, n_vectors, seed=6, kind=vec_kind)
        rolled = [member.perf()["rolling"] for member in group.members]
        again = group.dots_random(scale, steps, n_vectors, seed=6, kind=vec_kind)
    assert rolled == [1 if block_storage == "dictionary" else 0] * n_slabs  # (the stencil forms read the block dictionary)
    assert np.array_equal(in_place[0], again[0]) and np.array_equal(in_place[1], again[1])
    for got in (exchanged, in_place):
        assert np.abs(got[0] - ref[0]).max() <= 1e-12 * n and np.abs(got[1] - ref[1]).max() <= 1e-12 * n


# ------------------------------------------------------------------ Lanczos / gap
@pytest.mark.parametrize("name,k", [("swave20", 3), ("complex235", 2), ("snf", 2), ("chain128", 3), ("dwave8", 2)])
def test_lowest_eigenvalues_match_dense_spectrum(api, golden, name, k):
    """Gap probe beyond dense reach, checked where dense is still available: the k
    smallest distinct positive eigenvalues against the reference's spectrum."""
    system = _build(api, name)
    ref = golden.eigenvalues(name)
    distinct = [ref[0]]
    for value in ref[1:]:
        if value - distinct[-1] > 1e-6:
            distinct.append(value)
    got = system.lowest_eigenvalues(k, tol=1e-9, check_every=50, method="lanczos")
    assert got.shape == (k,)
    assert np.allclose(got, distinct[:k], rtol=1e-6, atol=1e-9)
    # (default: matrices this small are answered from the dense solver - same values, exactly)
    assert np.allclose(system.lowest_eigenvalues(k), distinct[:k], rtol=0, atol=1e-9)


def test_lanczos_tridiagonal_matches_oracle(api, solver_cls):
    """alpha_j, beta_j of the device Lanczos process on H^2 against a numpy restatement."""
    system = _build(api, "complex235")
    bsr = system.matrix("bsr")
    v = cheb_ref.random_vector(bsr.shape[0], 2, 0, cheb_ref.VEC_Z4)
    with solver_cls.from_hamiltonian(system) as dev:
        dev.lanczos_begin(1, seed=2, kind=cheb_ref.VEC_Z4, max_iter=20)
        alpha, beta = dev.lanczos_advance(12)
    v_prev, v_cur, b_prev = np.zeros_like(v), v / np.linalg.norm(v), 0.0
    for j in range(12):
        u = bsr @ v_cur
        w = bsr @ u - b_prev * v_prev
        a = np.vdot(u, u).real
        w = w - a * v_cur
        b = np.linalg.norm(w)
        assert np.isclose(alpha[j, 0], a, rtol=1e-11) and np.isclose(beta[j, 0], b, rtol=1e-9)
        v_prev, v_cur, b_prev = v_cur, w / b, b


def test_lanczos_run_is_ended_by_any_other_use_of_the_handle(api, solver_cls):
    """A Lanczos run keeps pointers into the handle's vector buffers.  Any other call that refills
    or reallocates them (here a wide recurrence call) must end the run: the next advance is refused
    with an error instead of iterating on stale or freed device memory; a fresh begin works."""
    system = _build(api, "snf")
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    with solver_cls.from_hamiltonian(system) as dev:
        dev.lanczos_begin(2, seed=1, max_iter=40)
        first = dev.lanczos_advance(5)
        dev.dots_random(scale, 4, 64, seed=3)  # 64 vectors: the vector buffers grow
        with pytest.raises(ValueError):
            dev.lanczos_advance(5)
        dev._lanczos_vectors = 2  # get past the Python-side guard: the library must refuse on its own
        with pytest.raises(ValueError):
            dev.lanczos_advance(5)
        dev.lanczos_begin(2, seed=1, max_iter=40)
        again = dev.lanczos_advance(5)
    assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])


def test_hermiticity_check_on_the_device(api, solver_cls, monkeypatch):
    """f3: the Hermiticity test of a closing `with` block (ref hamiltonian.py:121-122) made on the
    GPU.  Same criterion, same exception: Hermitian fills pass with the host's defect value,
    the reference's non-Hermitian cases (ref tests/test_hamiltonian.py:17-57: a hopping term
    without its partner, a non-Hermitian on-site term) raise RuntimeError, a repaired matrix
    passes again, and a block whose partner is an all-zero (dropped) block is caught."""
    from bodge_amd import hamiltonian

    monkeypatch.setattr(hamiltonian, "DEVICE_HERMITICITY_MIN_BLOCKS", 1)  # every `with` goes to the device
    system = systems.random_periodic(api)  # dense complex terms on sites, bonds and periodic edges
    host = system._hermiticity_defect()
    device = system._solver().hermiticity_defect()
    assert device < 1e-12 and abs(device - host) < 1e-15
    rng = np.random.default_rng(3)
    lattice = system.lattice
    with pytest.raises(RuntimeError, match="not Hermitian"):
        with system as (H, Δ):
            H[(0, 1, 2), (0, 1, 3)] = rng.random((2, 2)) + 1j * rng.random((2, 2))  # partner keeps its old value
    assert system._solver().hermiticity_defect() > 1e-3
    with pytest.raises(RuntimeError):
        with system as (H, Δ):
            H[(2, 2, 2), (2, 2, 2)] = np.array([[1.0, 2.0], [0.5j, 1.0]])  # non-Hermitian on-site term
    with system as (H, Δ):  # repair both
        t = 0.3 * api.σ0 + 0.1 * api.σ1
        H[(0, 1, 2), (0, 1, 3)] = t
        H[(0, 1, 3), (0, 1, 2)] = t
        H[(2, 2, 2), (2, 2, 2)] = 0.5 * api.σ3
    assert system._solver().hermiticity_defect() < 1e-12
    # a lone block: the open chain has no stored (j, i) partner once that block is all zero
    chain = api.Hamiltonian(api.CubicLattice((6, 1, 1)))
    with pytest.raises(RuntimeError):
        with chain as (H, Δ):
            for i in chain.lattice.sites():
                H[i, i] = 1.0 * api.σ0
            H[(1, 0, 0), (2, 0, 0)] = -1.0 * api.σ0
            H[(2, 0, 0), (1, 0, 0)] = 0.0 * api.σ0


def test_free_energy_over_several_devices_of_one_process(api, golden):
    """`free_energy(..., devices=[...])`: H replicated per listed GPU, start vectors shared out, one
    host thread per GPU, moments summed on the host (SURVEY §8b export 1, §5 `devices=`).  On a
    one-GPU box the list repeats ordinal 0 - two, then three, independent mirrors of the matrix.
    Per-vector results must be bit-identical to the single-mirror call (start vectors depend on
    (seed, id, element) only), F equal to round-off, exact traces match the reference's goldens."""
    from bodge_amd import observables

    system = _build(api, "swave20_zeeman")
    scale = observables._scale_of(system)
    single = system._solver().dots_random(scale, 24, 8, seed=11)
    for devices in ([0, 0], [0, 0, 0], [0]):
        split = observables.dots_random_devices(system, scale, 24, 8, devices, seed=11)
        assert np.array_equal(split[0], single[0]) and np.array_equal(split[1], single[1]), devices
    f_single = system.free_energy(0.5, method="chebyshev", trace="stochastic", vectors=8, moments=128, seed=11)
    f_split = system.free_energy(0.5, method="chebyshev", trace="stochastic", vectors=8, moments=128, seed=11,
                                 devices=[0, 0])
    assert np.isclose(f_split, f_single, rtol=1e-13, atol=0)
    exact = system.free_energy(0.5, method="chebyshev", trace="exact", devices=[0, 0, 0])
    assert np.isclose(exact, golden.free_energy("swave20_zeeman", 0.5), rtol=1e-10, atol=0)
    dense = system.free_energy(0.1, method="dense", devices=[0])
    assert np.isclose(dense, golden.free_energy("swave20_zeeman", 0.1), rtol=1e-10, atol=0)
    with pytest.raises(ValueError):
        system.free_energy(0.5, devices=[])
    with pytest.raises(ValueError):
        system.free_energy(0.5, devices=[0], comm=object())
    with pytest.raises(ValueError):
        system.free_energy(0.5, method="chebyshev", decomposition="slab")  # slabs need a communicator
    with pytest.raises((ValueError, RuntimeError)):
        system.free_energy(0.5, method="chebyshev", devices=[99])  # no such GPU


def test_default_call_warns_when_it_is_not_the_reference_computation(api, monkeypatch):
    """A caller of plain `free_energy(T)` must be told when the answer is not the reference's
    dense computation: a stochastic-trace estimate (with its standard error) beyond the exact
    trace, or the surrogate-temperature expansion at T = 0 beyond the dense eigensolver.  Explicit
    requests for those routes stay silent."""
    import warnings

    from bodge_amd import observables

    system = _build(api, "swave20_zeeman")
    monkeypatch.setattr(observables, "EXACT_TRACE_LIMIT", 1000)  # make 4N = 1600 "large"
    monkeypatch.setattr(observables, "DENSE_AUTO_LIMIT_T0", 1000)
    with pytest.warns(RuntimeWarning, match="stochastic-trace estimate from 64 random vectors, standard error"):
        value = system.free_energy(0.5)
    assert abs(value / system.free_energy(0.5, method="dense") - 1) < 5e-3
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        system.free_energy(0.5, method="chebyshev", trace="stochastic")
        system.free_energy(0.5, method="dense")
    monkeypatch.setattr(observables, "DENSE_VALUES_LIMIT_T0", 1000)
    with pytest.warns(RuntimeWarning, match="beyond the dense eigensolver: evaluated by the Chebyshev"):
        zero = system.free_energy(0.0, trace="exact")
    assert abs(zero / system.free_energy(0.0, method="dense") - 1) < 1e-10


def test_exact_zero_modes_count_once_per_pair(api):
    """ADVICE r3 / DESIGN §6: a site with no terms at all has four eigenvalues that are exactly zero.  The reference's
    `ε > 0` (ref hamiltonian.py:305) drops all four; the dense route here evaluates Tr f(H) - every ± pair once, so each
    of the two exact pairs adds T ln 2, the number the Chebyshev route gives as well."""
    lattice = api.CubicLattice((5, 4, 1))
    system = api.Hamiltonian(lattice)
    lonely = (0, 0, 0)
    with system as (H, Δ):
        for i in lattice.sites():
            if i != lonely:
                H[i, i] = 3.0 * api.σ0 - 0.05 * api.σ3
                Δ[i, i] = -0.1 * api.jσ2
        for i, j in lattice.bonds():
            if lonely not in (i, j):
                H[i, j] = -1.0 * api.σ0
    dense = np.asarray(system.matrix("dense"))
    eps = np.linalg.eigvalsh(dense)
    assert np.sum(np.abs(eps) < 1e-12) == 4
    kept = eps[eps > 1e-9]
    for T in (0.1, 0.5):
        strict = -0.5 * kept.sum() - T * np.log1p(np.exp(-kept / T)).sum()  # the reference's sum without the zero modes
        for method in ("dense", "chebyshev"):
            value = system.free_energy(T, method=method, **({"moments": 2048, "trace": "exact"} if method == "chebyshev" else {}))
            assert abs(value - (strict - 2 * T * np.log(2.0))) <= 1e-9 * abs(strict), (T, method, value, strict)


@pytest.mark.parametrize("name", ["swave20", "complex235", "random357", "snf", "chain128", "dwave8", "swave30_zeeman",
                                  "peierls30", "chain300", "swave50_zeeman"])
def test_eigenvalues_by_tridiagonalisation_match_the_reference(api, golden, knobs, block_storage, name):
    """The eigenvalues-only dense route (csrc/tridiag.hpp: Householder tridiagonalisation with lazily
    applied rank-2 updates + Sturm bisection, real or complex arithmetic by imag(H)) needs no rocSOLVER:
    all 4N eigenvalues against numpy (4N <= 4000) and the positive half against the reference's
    diagonalize() - including BASELINE config 5's ladder n = 3600 (real and complex), 1200 and 10^4 -
    to 1e-10 (measured 1e-13); ± symmetry of the BdG spectrum; F(T) through `free_energy(method="dense")`,
    which takes this route from 4N > 512 on, against the reference's values."""
    system = _build(api, name)
    dim = system.shape[0]
    _dense_once(block_storage, dim)
    ref = golden.eigenvalues(name)
    knobs.set("BODGE_AMD_EIGH", "tridiagonal")
    w, vectors = system._solver().eigh(vectors=False)
    assert vectors is None and w.shape == (dim,) and np.all(np.diff(w) >= 0)
    if len(ref) == dim // 2:  # (the reference keeps the positive half)
        assert np.abs(w[dim // 2:] - ref).max() <= 1e-10
        assert np.abs(w[: dim // 2][::-1] + ref).max() <= 1e-10 or name in ("random357",)  # ± symmetric unless triplet on-site pairing
    if dim <= 4000:
        assert np.abs(w - np.linalg.eigvalsh(np.asarray(system.matrix("dense")))).max() <= 1e-10
    with pytest.raises(ValueError):
        system._solver().eigh(vectors=True)  # this route has no eigenvectors
    knobs.unset("BODGE_AMD_EIGH")
    w_default, _ = system._solver().eigh(vectors=False)  # default routing: the same route above 512 rows, Jacobi below
    assert np.abs(w_default - w).max() <= 1e-10
    zero_modes = int(np.count_nonzero(np.abs(w) < 1e-12))
    for temperature in systems.CATALOG[name]["temps"]:
        value = system.free_energy(temperature, method="dense")
        expect = golden.free_energy(name, temperature)
        if zero_modes:
            # dwave8 has four zero eigenvalues.  The reference counts those that round-off happened to make
            # positive (one of four in its eigvalsh run, ref :302-305), here every ± pair counts once (two):
            # the values differ by a whole number of T ln 2
            missing = (expect - value) / (temperature * np.log(2.0))
            assert abs(missing - round(missing)) <= 1e-9 and 0 <= round(missing) <= zero_modes // 2, missing
        else:
            assert abs(value - expect) <= 1e-10 * abs(value)


@pytest.mark.parametrize("name,options", [
    ("swave30_zeeman", {"BODGE_AMD_EIGH_STAGES": "2"}),                                   # n = 3600: forced (default from 6000 rows)
    ("swave30_zeeman", {"BODGE_AMD_EIGH_STAGES": "2", "BODGE_AMD_EIGH_GRAM_QR": "0"}),      # every panel with a grid barrier per column
    ("swave30_zeeman", {"BODGE_AMD_EIGH_STAGES": "2", "BODGE_AMD_EIGH_LOOKAHEAD": "1"}),    # next panel factorised beside the update
    ("chain300", {"BODGE_AMD_EIGH_STAGES": "2"}),                                         # n = 1200, a chain: panels of few non-zeros
    ("dwave8", {"BODGE_AMD_EIGH_STAGES": "2"}),                                           # zero modes, 3-D
    ("swave50_zeeman", {}),                                                               # n = 10^4: the default route
    ("swave60_zeeman", {}),                                                               # n = 14 400 (golden recorded in round 4)
])
def test_eigenvalues_by_the_two_stage_route_match_the_reference(api, golden, knobs, block_storage, name, options):
    """K10 (csrc/twostage.hpp): real symmetric matrix -> band of half-width 32 by block Householder panels (Gram-matrix QR,
    fp64 MFMA products) -> tridiagonal by pipelined bulge chasing -> bisection.  Eigenvalues against the reference's
    diagonalize() (1e-10; measured 2e-13), against numpy up to 4000 rows, and against the one-stage route."""
    system = _build(api, name)
    dim = system.shape[0]
    _dense_once(block_storage, dim, limit=0)
    ref = golden.eigenvalues(name)
    knobs.set("BODGE_AMD_EIGH", "tridiagonal")
    knobs.set("BODGE_AMD_EIGH_STAGES", "1")
    one, _ = system._solver().eigh(vectors=False)
    knobs.unset("BODGE_AMD_EIGH_STAGES")
    knobs.update(options)
    two, _ = system._solver().eigh(vectors=False)
    assert two.shape == (dim,) and np.all(np.diff(two) >= 0)
    assert np.abs(two - one).max() <= 1e-11 * max(1.0, np.abs(one).max())
    assert np.abs(two[dim // 2:] - ref).max() <= 1e-10
    if dim <= 4000:
        assert np.abs(two - np.linalg.eigvalsh(np.asarray(system.matrix("dense")))).max() <= 1e-10
    for temperature in systems.CATALOG[name]["temps"]:
        if not np.count_nonzero(np.abs(two) < 1e-12):
            value = system.free_energy(temperature, method="dense")
            assert abs(value - golden.free_energy(name, temperature)) <= 1e-10 * abs(value)


@pytest.mark.parametrize("name", ["swave30_zeeman", "peierls30", "chain300", "swave50_zeeman", "swave60_zeeman", "dwave8", "snf"])
def test_dense_ladder_without_a_library(api, golden, block_storage, name):
    """BASELINE config 5's feasible ladder through the DEFAULT route of `diagonalize()` - the library's own
    Householder tridiagonalisation, bisection, inverse iteration and back-transformation (csrc/tridiag.hpp),
    no rocSOLVER: n = 3600 real and complex, the literal "300" chain (n = 1200), n = 10^4, and two smaller
    systems with highly degenerate spectra (dwave8: zero modes and 4-fold levels; snf).  Eigenvalues within
    1e-10 of the reference's own diagonalize(), eigen-equation residual <= 1e-9, orthonormal finite
    vectors (every vector against a sample of 64), the reference's (k, N, 4) layout, and no rocSOLVER
    object mapped into the process afterwards."""
    system = _build(api, name)
    dim = system.shape[0]
    _dense_once(block_storage, dim)
    vals, vecs = system.diagonalize(format="raw")
    ref = golden.eigenvalues(name)
    zero_modes = int(np.count_nonzero(ref < 1e-12))
    assert vecs.shape == (dim, vals.size) and np.isfinite(vecs).all() and np.all(np.diff(vals) >= 0)
    # (zero modes: which of a ± pair at 1e-16 counts as "positive" is round-off, here as in the reference)
    assert abs(vals.size - ref.size) <= zero_modes
    assert np.abs(vals[-(ref.size - zero_modes):] - ref[zero_modes:]).max() <= 1e-10
    bsr = system.matrix("bsr")
    assert np.abs(bsr @ vecs - vecs * vals).max() <= 1e-9
    idx = np.arange(0, vals.size, max(1, vals.size // 64))
    gram = vecs[:, idx].conj().T @ vecs
    gram[np.arange(idx.size), idx] -= 1.0
    assert np.abs(gram).max() <= 1e-9
    _, shaped = system.diagonalize()
    assert shaped.shape == (vals.size, dim // 4, 4) and np.array_equal(shaped[3, 7, :], vecs[28:32, 3])
    with open("/proc/self/maps") as fh:
        assert "librocsolver" not in fh.read() or os.environ.get("BODGE_AMD_EIGH")


@pytest.mark.parametrize("rayleigh_ritz", ["device", "host"])
@pytest.mark.parametrize("name,k", [("swave20", 6), ("snf", 5), ("complex235", 4), ("swave20_zeeman", 3)])
def test_lowest_eigenpairs_with_multiplicities_and_vectors(api, golden, name, k, rayleigh_ritz):
    """f4 completed: the k lowest positive eigenvalues WITH multiplicities (swave20's lowest level is
    four-fold: spin x the k_x <-> k_y symmetry of the square) and orthonormal eigenvectors in the
    reference's layouts (ref hamiltonian.py:235-248), from two passes of the device Lanczos process.
    Eigenvalues against the reference's spectrum, residual |Hv - εv| <= 1e-8."""
    system = _build(api, name)
    ref = golden.eigenvalues(name)
    # rayleigh_ritz="device": Gram matrices, projections and H products of the second pass on the GPU
    # (bdg_lanczos_ritz_pairs); "host": the numpy form on (levels, vectors, 4N) arrays
    vals, vecs = system.lowest_eigenpairs(k, format="raw", method="lanczos", rayleigh_ritz=rayleigh_ritz)
    dense = np.asarray(system.matrix("dense"))
    assert vals.shape == (k,) and vecs.shape == (dense.shape[0], k)
    assert np.all(np.diff(vals) >= -1e-12) and np.abs(vals - ref[:k]).max() <= 1e-9
    assert np.abs(dense @ vecs - vecs * vals).max() <= 1e-8
    assert np.abs(vecs.conj().T @ vecs - np.eye(k)).max() <= 1e-8
    vals2, shaped = system.lowest_eigenpairs(k, method="lanczos", rayleigh_ritz=rayleigh_ritz)
    assert shaped.shape == (k, system.lattice.size, 4) and np.allclose(vals2, vals, rtol=0, atol=1e-12)
    for n in range(k):  # same layout rule as diagonalize(): v[n, site, α] = X[4 site + α, n]
        assert np.abs(dense @ shaped[n].reshape(-1) - vals2[n] * shaped[n].reshape(-1)).max() <= 1e-8
    with pytest.raises(Exception):
        system.lowest_eigenpairs(k, format="foo")
    # default (method="auto"): a matrix within the own dense solver's reach is answered by it
    vals3, shaped3 = system.lowest_eigenpairs(k)
    assert np.abs(vals3 - ref[:k]).max() <= 1e-10 and shaped3.shape == shaped.shape
    for n in range(k):
        assert np.abs(dense @ shaped3[n].reshape(-1) - vals3[n] * shaped3[n].reshape(-1)).max() <= 1e-9


def test_bound_states_of_a_large_lattice_without_host_copies_of_the_ritz_block(api, solver_cls):
    """A 300 x 300 s-wave lattice (3.6e5 x 3.6e5, far beyond the dense solver) with two identical magnetic
    impurities 100 sites apart: their in-gap (Yu-Shiba-Rusinov) states are degenerate to ~1e-9, i.e. ONE
    level of multiplicity two for the Lanczos process.  `lowest_eigenpairs(2)` must return both states,
    orthonormal, residual |Hv - εv| <= 1e-8, with the Rayleigh-Ritz step on the device - and the same
    values from the host form of that step."""
    lattice = api.CubicLattice((300, 300, 1))
    system = api.Hamiltonian(lattice)
    spots = [lattice[(100, 150, 0)], lattice[(200, 150, 0)]]
    field = np.zeros(lattice.size)
    field[spots] = 2.0
    with system as (H, Δ):
        H.set_sites(3.0 * api.σ0 - field[:, None, None] * api.σ3)
        Δ.set_sites(-0.5 * api.jσ2)
        H.set_bonds(-1.0 * api.σ0)
    vals, vecs = system.lowest_eigenpairs(2, format="raw", tol=1e-8, max_iter=4000)
    assert vals.shape == (2,) and vecs.shape == (4 * lattice.size, 2)
    assert 0.0 < vals[0] < 0.45 and abs(vals[1] - vals[0]) < 1e-6  # both below the clean gap 0.5, degenerate
    bsr = system.matrix("bsr")
    assert np.abs(bsr @ vecs - vecs * vals).max() <= 1e-8
    assert np.abs(vecs.conj().T @ vecs - np.eye(2)).max() <= 1e-8
    weight = (np.abs(vecs.reshape(lattice.size, 4, 2)) ** 2).sum(axis=(1, 2))
    assert weight[spots].min() > 100 * weight.mean()  # localised at the two impurities
    host_vals, _ = system.lowest_eigenpairs(2, format="raw", tol=1e-8, max_iter=4000, rayleigh_ritz="host")
    assert np.abs(host_vals - vals).max() <= 1e-10


def test_lowest_eigenpairs_when_a_ritz_vector_has_no_positive_energy_part(api):
    """8x11 d-wave lattice (parameters found by scratch/fuzz_api.py): with seed 0 the Ritz vector of one of
    the eight start vectors for the lowest level lies wholly in the -ε eigenspace, so its projection
    (H + ε)y is round-off noise.  Normalised, it used to enter the Rayleigh-Ritz step as a full
    candidate and came back as a 'state' at 0.1378 with residual 0.4."""
    lattice = api.CubicLattice((8, 11, 1))
    system = api.Hamiltonian(lattice)
    pairs = lattice.bond_array(coords=True)
    with system as (H, Δ):
        H.set_sites(2.3461553344437616 * api.σ0 - 0.15347102170475338 * api.σ3)
        H.set_bonds(-1.0 * api.σ0)
        Δ.set_bonds(-0.9973494389997505 * api.dwave()(pairs[:, 0], pairs[:, 1]))
    dense = np.asarray(system.matrix("dense"))
    positive = np.linalg.eigvalsh(dense)
    positive = positive[positive > 0]
    vals, vecs = system.lowest_eigenpairs(3, format="raw", method="lanczos")
    assert np.abs(vals - positive[:3]).max() <= 1e-9
    assert np.abs(dense @ vecs - vecs * vals).max() <= 1e-8 and np.abs(vecs.conj().T @ vecs - np.eye(3)).max() <= 1e-8


def test_lowest_eigenpairs_of_a_tiny_matrix(api):
    """72x72 (3x3x2 d-wave): fifty Lanczos iterations exhaust the space and the process breaks down
    (scratch/fuzz_api.py found a negative 'eigenvalue'); the default route answers from diagonalize()."""
    lattice = api.CubicLattice((3, 3, 2))
    system = api.Hamiltonian(lattice)
    pairs = lattice.bond_array(coords=True)
    with system as (H, Δ):
        H.set_sites(1.7 * api.σ0 - 0.1 * api.σ3)
        H.set_bonds(-1.0 * api.σ0)
        Δ.set_bonds(-0.4 * api.dwave()(pairs[:, 0], pairs[:, 1]))
    dense = np.asarray(system.matrix("dense"))
    positive = np.linalg.eigvalsh(dense)
    positive = positive[positive > 0]
    vals, vecs = system.lowest_eigenpairs(4, format="raw")
    assert np.abs(vals - positive[:4]).max() <= 1e-10 and np.abs(dense @ vecs - vecs * vals).max() <= 1e-9
    assert np.allclose(system.lowest_eigenvalues(1), positive[:1], atol=1e-10)
    with pytest.raises(RuntimeError):
        system.lowest_eigenpairs(len(positive) + 1)
    with pytest.raises(ValueError):
        system.lowest_eigenpairs(2, method="krylov")


@pytest.mark.parametrize("name", ["swave30_zeeman", "peierls30"])
def test_own_jacobi_kernels_reach_4096_rows(api, golden, knobs, block_storage, name):
    """Between 4N = 2048 and 4096 the own one-sided Jacobi kernels (16 elements per thread) serve
    for as long as the rocSOLVER object has not arrived from cold storage; forced here.  n = 3600,
    real and complex, against the reference's own spectra: eigenvalues 1e-10, residual 1e-9."""
    _dense_once(block_storage, 3600)
    knobs.set("BODGE_AMD_EIGH", "jacobi")
    system = _build(api, name)
    vals, vecs = system.diagonalize(format="raw")
    ref = golden.eigenvalues(name)
    assert vals.shape == ref.shape and np.abs(vals - ref).max() <= 1e-10
    bsr = system.matrix("bsr")
    assert np.isfinite(vecs).all() and np.abs(bsr @ vecs - vecs * vals).max() <= 1e-9
    idx = np.arange(0, vals.size, 97)
    gram = vecs[:, idx].conj().T @ vecs
    gram[np.arange(idx.size), idx] -= 1.0
    assert np.abs(gram).max() <= 1e-9
