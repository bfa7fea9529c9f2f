"""Pin the CPU oracle to the reference's recorded outputs (no GPU).

`oracle.dense_ref` must reproduce the goldens to round-off; `oracle.cheb_ref`
(the algorithm the HIP library implements) must reproduce them to the
truncation level its moment count supports, which is stated per test.
"""

import numpy as np
import pytest
import scipy.sparse as sp

import systems
from oracle import cheb_ref, dense_ref

DENSE_CASES = ["swave20", "swave20_zeeman", "snf", "barrier", "complex235", "random357", "chain128", "ldos16"]


@pytest.mark.parametrize("name", DENSE_CASES)
def test_dense_free_energy_and_spectrum(api, golden, name):
    spec = systems.CATALOG[name]
    system = spec["build"](api, **spec["kwargs"])
    dense = np.asarray(system.matrix("dense"))
    for temperature in spec["temps"]:
        assert np.isclose(dense_ref.free_energy(dense, temperature), golden.free_energy(name, temperature),
                          rtol=1e-12, atol=0)
    vals, vecs = dense_ref.diagonalize(dense, format="raw")
    ref = golden.eigenvalues(name)
    assert vals.size == ref.size == golden.values[name]["n_eigenvalues"]
    assert np.allclose(vals, ref, rtol=0, atol=1e-10)
    assert np.allclose(dense @ vecs, vecs * vals, atol=1e-9)
    with pytest.raises(ValueError):
        dense_ref.free_energy(dense, -1.0)


@pytest.mark.parametrize("name", ["ldos16", "random357", "chain128", "pwave31"])
def test_dense_ldos(api, golden, name):
    spec = systems.CATALOG[name]
    system = spec["build"](api, **spec["kwargs"])
    csc = system.matrix("csc")
    for n, (site, energies) in enumerate(spec["ldos"]):
        rho = dense_ref.ldos(csc, system.lattice[site], energies)
        assert np.allclose(rho, golden.ldos(name, n), rtol=1e-9, atol=1e-12)


def test_oracle_assembly_equals_vectorised_assembly(api):
    system = systems.random_periodic(api, shape=(2, 3, 4), seed=11)
    lat = system.lattice
    rng = np.random.default_rng(11)  # replay the same draws through the loop-based restatement
    hop, pair = {}, {}
    r = rng.random
    mix = lambda c: c[0] * api.σ0 + c[1] * api.σ1 + c[2] * api.σ2 + c[3] * api.σ3
    trip = lambda: (r() * api.σ1 + r() * api.σ2 + r() * api.σ3) @ api.jσ2
    for i in lat.sites():
        hop[lat[i], lat[i]] = mix(r(4))
        pair[lat[i], lat[i]] = trip()
    for gen in (lat.bonds(), lat.edges()):
        for i, j in gen:
            t = mix(r(4))
            hop[lat[i], lat[j]] = t
            hop[lat[j], lat[i]] = t
            pair[lat[i], lat[j]] = trip()
    ref = dense_ref.assemble_bsr(lat.size, [(lat[a], lat[b]) for a, b in lat], hop, pair)
    assert np.array_equal(ref.indptr, system._matrix.indptr)
    assert np.array_equal(ref.indices, system._matrix.indices)
    assert np.array_equal(ref.data, system._data)


# ------------------------------------------------------------------ Chebyshev
def test_start_vectors_are_counter_based():
    full = cheb_ref.random_vector(1000, seed=3, vec_id=5, kind=cheb_ref.VEC_Z4)
    part = cheb_ref.random_vector(300, seed=3, vec_id=5, kind=cheb_ref.VEC_Z4, row0=700)
    assert np.array_equal(full[700:], part)
    assert np.allclose(np.abs(full), 1.0)
    rad = cheb_ref.random_vector(4096, seed=0, vec_id=0)
    assert set(np.unique(rad.real)) == {-1.0, 1.0} and not rad.imag.any()
    assert abs(rad.real.mean()) < 0.1
    assert not np.array_equal(rad, cheb_ref.random_vector(4096, seed=0, vec_id=1))
    # known answers (also produced by an independent C++ SplitMix64): pins the
    # generator that the numpy oracle and the HIP kernel must share
    assert int(cheb_ref.vector_key(0, 0)) == 0xA706DD2F4D197E6F
    assert int(cheb_ref.vector_key(12345, 7)) == 0x19BD65D14C45ECF7
    # (round 4: one hash per site, bits 63 - component / 62 - 2 component .. : values from a pure-Python integer restatement)
    assert rad.real[:12].astype(int).tolist() == [1, 1, -1, 1, 1, -1, 1, 1, -1, 1, 1, -1]
    z4 = cheb_ref.random_vector(12, 12345, 7, cheb_ref.VEC_Z4)
    quarter_turns = [0, 3, 1, 3, 3, 1, 2, 0, 0, 1, 3, 3]
    assert np.array_equal(z4, np.array([1, 1j, -1, -1j])[quarter_turns])


def test_doubling_identities_against_direct_moments(api):
    system = systems.complex_hopping(api)
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    start = cheb_ref.random_block(bsr.shape[0], 1, range(3), cheb_ref.VEC_Z4)
    mu = cheb_ref.moments(bsr, scale, 24, start)
    dense = np.asarray(system.matrix("dense")) / scale
    t_prev, t_cur = start, dense @ start
    direct = [np.einsum("ir,ir->r", start.conj(), start).real, np.einsum("ir,ir->r", start.conj(), t_cur).real]
    for _ in range(22):
        t_prev, t_cur = t_cur, 2 * dense @ t_cur - t_prev
        direct.append(np.einsum("ir,ir->r", start.conj(), t_cur).real)
    assert np.allclose(mu, np.array(direct), atol=1e-10)


@pytest.mark.parametrize(
    "name,temperature,moments,rtol",
    [
        ("barrier", 0.1, 700, 1e-10),  # a/πT ~ 29: 700 moments reach round-off
        ("complex235", 1.0, 192, 1e-11),  # a = 20.2: needs ~23 a/(πT) moments
        ("complex235", 0.1, 1400, 1e-11),
        ("snf", 1.0, 64, 1e-11),
        ("swave20", 0.5, 128, 1e-11),  # 3e-15 at M=128
    ],
)
def test_chebyshev_exact_trace_free_energy(api, golden, name, temperature, moments, rtol):
    spec = systems.CATALOG[name]
    system = spec["build"](api, **spec["kwargs"])
    value = cheb_ref.free_energy_exact_trace(system.matrix("bsr"), temperature, moments)
    assert np.isclose(value, golden.free_energy(name, temperature), rtol=rtol, atol=0)


def test_chebyshev_truncation_is_visible_when_moments_are_too_few(api, golden):
    system = systems.swave_square(api)
    value = cheb_ref.free_energy_exact_trace(system.matrix("bsr"), 0.1, 64)
    err = abs(value / golden.free_energy("swave20", 0.1) - 1)
    assert 1e-7 < err < 1e-3  # 3.2e-5 at (T=0.1, M=64)


@pytest.mark.parametrize("name", ["ldos16", "random357", "chain128"])
def test_chebyshev_ldos(api, golden, name):
    spec = systems.CATALOG[name]
    system = spec["build"](api, **spec["kwargs"])
    bsr = system.matrix("bsr")
    for n, (site, energies) in enumerate(spec["ldos"]):
        scale = cheb_ref.spectral_bound(bsr)
        gam = dense_ref.ldos_broadening(energies)[1].min()
        rho = cheb_ref.ldos(bsr, system.lattice[site], energies,
                            n_moments=cheb_ref.ldos_moment_count(scale, gam, digits=12))
        assert np.allclose(rho, golden.ldos(name, n), rtol=1e-9, atol=1e-12)


def test_stochastic_trace_converges_to_exact(api, golden):
    system = systems.swave_square(api)
    bsr = system.matrix("bsr")
    est = cheb_ref.free_energy_stochastic(bsr, 0.5, 96, n_vectors=32, seed=0)
    # 32 vectors x 1600 rows: relative noise ~ 1/sqrt(5e4) on the fluctuating part
    assert abs(est / golden.free_energy("swave20", 0.5) - 1) < 5e-3


def test_spectral_bound_encloses_spectrum(api, golden):
    for name in ["swave20", "random357", "dwave8"]:
        spec = systems.CATALOG[name]
        bound = cheb_ref.spectral_bound(spec["build"](api, **spec["kwargs"]).matrix("bsr"))
        assert bound > golden.values[name]["e_max"]
        assert bound < 2.5 * golden.values[name]["e_max"]


@pytest.mark.parametrize("name", ["random357", "complex235", "swave20_zeeman"])
def test_c_restatement_matches_numpy_oracle(api, name):
    """oracle/cheb_c.c (the multi-threaded CPU baseline) reproduces the numpy recurrence."""
    from oracle import cheb_c

    spec = systems.CATALOG[name]
    bsr = spec["build"](api, **spec["kwargs"]).matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    start = cheb_ref.random_block(bsr.shape[0], 5, range(6), cheb_ref.VEC_Z4)
    keep = start.copy()
    d_ref, e_ref = cheb_ref.recurrence_dots(bsr, scale, 48, start)
    for threads in (1, 3):
        cheb_c.set_threads(threads)
        d, e = cheb_c.recurrence_dots(bsr, scale, 48, start)
        assert np.array_equal(start, keep)
        assert np.abs(d - d_ref).max() < 1e-11 and np.abs(e - e_ref).max() < 1e-11
    if not np.iscomplexobj(bsr.data) or np.abs(bsr.data.imag).max() == 0:
        real_start = cheb_ref.random_block(bsr.shape[0], 5, range(6))
        d_ref, e_ref = cheb_ref.recurrence_dots(bsr, scale, 48, real_start)
        d, e = cheb_c.recurrence_dots(bsr, scale, 48, real_start, real=True)
        assert np.abs(d - d_ref).max() < 1e-11 and np.abs(e - e_ref).max() < 1e-11
