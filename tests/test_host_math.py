"""Host-side scalar work of the product (`bodge_amd.chebyshev`, no GPU, no oracle import for the
thing under test): fed with *exact* moments built from a dense spectrum, the series must return
the reference observables.  Exact moments: μ_m = Σ_k T_m(ε_k / a) = Σ_k cos(m arccos(ε_k / a))."""

import numpy as np
import pytest

import systems
from bodge_amd import chebyshev


def _spectrum(api, name):
    spec = systems.CATALOG[name]
    system = spec["build"](api, **spec["kwargs"])
    dense = np.asarray(system.matrix("dense"))
    vals, vecs = np.linalg.eigh(dense)
    indptr, _, data = system.bsr_arrays()
    return system, vals, vecs, chebyshev.spectral_bound(indptr, data)


def _trace_moments(vals, scale, n):
    theta = np.arccos(vals / scale)
    return np.cos(np.arange(n)[:, None] * theta[None, :]).sum(axis=1)


@pytest.mark.parametrize("name,temperature", [("barrier", 0.1), ("complex235", 1.0), ("complex235", 0.01),
                                              ("snf", 0.1), ("chain128", 0.01), ("swave20_zeeman", 0.1)])
def test_free_energy_series_from_exact_moments(api, golden, name, temperature):
    _, vals, _, scale = _spectrum(api, name)
    assert scale > np.abs(vals).max()
    moments = chebyshev.moments_for_free_energy(scale, temperature)
    value = chebyshev.free_energy_series(_trace_moments(vals, scale, moments), scale, temperature)
    assert np.isclose(value, golden.free_energy(name, temperature), rtol=1e-11, atol=0)


def test_default_moment_rule_is_sufficient(api, golden):
    _, vals, _, scale = _spectrum(api, "snf")
    for temperature in (0.1, 1.0):
        m = chebyshev.moments_for_free_energy(scale, temperature)
        assert m % 2 == 0 and 32 <= m <= 4096
        value = chebyshev.free_energy_series(_trace_moments(vals, scale, m), scale, temperature)
        assert np.isclose(value, golden.free_energy("snf", temperature), rtol=1e-10, atol=0)
    assert chebyshev.moments_for_free_energy(scale, 0.0) == 4096


def test_zero_temperature_series_converges_algebraically(api, golden):
    """f = -|ε|/4 is not analytic: the error falls like 1/M (no 1e-10 at T = 0 from Chebyshev)."""
    _, vals, _, scale = _spectrum(api, "barrier")
    exact = -0.5 * vals[vals > 0].sum()
    errs = [abs(chebyshev.free_energy_series(_trace_moments(vals, scale, m), scale, 0.0) / exact - 1)
            for m in (256, 1024, 4096)]
    assert errs[0] > errs[1] > errs[2] and errs[2] < 1e-4
    damped = chebyshev.free_energy_series(_trace_moments(vals, scale, 1024), scale, 0.0, damping=True)
    assert abs(damped / exact - 1) < 1e-2


@pytest.mark.parametrize("name", ["swave20_zeeman", "swave20", "snf", "complex235", "chain128"])
def test_gapped_ground_state_series_from_exact_moments(api, golden, name):
    """T = 0 on a gapped spectrum: the expansion of -(ε/4)·erf(ε/δ), δ = gap/5, with the rule's
    8·a/δ moments returns -½ Σ_{ε>0} ε to 1e-12 - the finite-temperature surrogate f_T at T = gap/20
    (round 1) is two orders less accurate with four times the moments, the plain |ε| series eight."""
    _, vals, _, scale = _spectrum(api, name)
    positive = vals[vals > 0]
    exact, gap = -0.5 * positive.sum(), float(positive.min())
    width = gap / 5.0
    m = chebyshev.moments_for_gapped_ground_state(scale, width)
    assert m % 2 == 0 and m >= 8 * scale / width
    mu = _trace_moments(vals, scale, 4 * m)
    value = chebyshev.free_energy_series(mu[:m], scale, 0.0, density=chebyshev.gapped_ground_state_density(width))
    assert abs(value / exact - 1) <= 1e-12
    old_surrogate = chebyshev.free_energy_series(mu[:m], scale, gap / 20.0)
    plain = chebyshev.free_energy_series(mu[:m], scale, 0.0)
    assert abs(old_surrogate / exact - 1) > 10 * abs(value / exact - 1) and abs(plain / exact - 1) > 1e-9
    if name == "swave20_zeeman":
        assert np.isclose(exact, golden.free_energy(name, 0.0), rtol=1e-12)


def test_resolvent_series_from_exact_moments(api, golden):
    system, vals, vecs, scale = _spectrum(api, "ldos16")
    site, energies = systems.CATALOG["ldos16"]["ldos"][0]
    i = system.lattice[site]
    eps = np.unique(np.abs(np.array(energies, dtype=float)))
    gam = np.gradient(eps)
    m = chebyshev.moments_for_resolvent(scale, gam.min())
    theta = np.arccos(vals / scale)
    cosines = np.cos(np.arange(m)[:, None] * theta[None, :])
    rho = {}
    for e, g in zip(eps, gam):
        diag = []
        for a in range(4):
            weights = np.abs(vecs[4 * i + a, :]) ** 2
            diag.append(chebyshev.resolvent_series(cosines @ weights, scale, e + 1j * g))
        rho[+e] = -np.imag(diag[0] + diag[1]) / np.pi
        rho[-e] = -np.imag(diag[2] + diag[3]) / np.pi
    got = np.array([rho[e] for e in np.array(energies, dtype=float)])
    assert np.allclose(got, golden.ldos("ldos16", 0), rtol=1e-10, atol=1e-13)


def test_dots_to_moments_and_coefficients():
    rng = np.random.default_rng(0)
    d, e = rng.random((6, 3)), rng.random((6, 3))
    mu = chebyshev.dots_to_moments(d, e)
    assert mu.shape == (12, 3)
    assert np.array_equal(mu[0], d[0]) and np.array_equal(mu[1], e[0])
    assert np.allclose(mu[6], 2 * d[3] - d[0]) and np.allclose(mu[7], 2 * e[3] - e[0])
    # coefficients of a polynomial are recovered exactly: 3 T_0 - 2 T_2 + 0.5 T_5
    coeff = chebyshev.chebyshev_coefficients(lambda x: 3 - 2 * (2 * x**2 - 1) + 0.5 * (16 * x**5 - 20 * x**3 + 5 * x), 8)
    assert np.allclose(coeff, [3, 0, -2, 0, 0, 0.5, 0, 0], atol=1e-14)
    kernel = chebyshev.jackson_kernel(64)
    assert np.isclose(kernel[0], 1.0) and np.all(np.diff(kernel) < 0) and kernel[-1] > 0


def test_spectral_bound_on_index_arrays(api, golden):
    for name in ("swave20", "random357", "dwave8", "complex235"):
        spec = systems.CATALOG[name]
        indptr, _, data = spec["build"](api, **spec["kwargs"]).bsr_arrays()
        bound = chebyshev.spectral_bound(indptr, data)
        assert golden.values[name]["e_max"] < bound < 2.5 * golden.values[name]["e_max"]
    assert chebyshev.spectral_bound(np.array([0, 0]), np.zeros((0, 4, 4))) == 1.0


def test_trace_form_needs_a_symmetric_spectrum(api, golden):
    """The random test matrix has on-site triplet pairing (Δ_ii ≠ -Δ_ii^T): Hermitian, but its
    spectrum is not ±-symmetric, so Σ_{ε>0} g(ε) is not the trace of the even function f."""
    system, vals, _, scale = _spectrum(api, "random357")
    assert not system.has_symmetric_spectrum(1e-12)
    assert not np.allclose(np.sort(vals), np.sort(-vals), atol=1e-8)
    with pytest.raises(RuntimeError, match="particle-hole"):
        system.free_energy(0.5, method="chebyshev")
    for name in ("swave20", "complex235", "dwave8", "pwave31", "chain128"):
        spec = systems.CATALOG[name]
        assert spec["build"](api, **spec["kwargs"]).has_symmetric_spectrum(1e-12), name


def test_smoothed_density_and_krylov_cut():
    """Two small host helpers: the smoothed T = 0 density against scipy's erf, and the cut of a Lanczos
    tridiagonal matrix at the first round-off-sized β (exhausted Krylov space)."""
    from scipy.special import erf

    from bodge_amd.observables import _krylov_length

    x = np.linspace(-7.0, 7.0, 20001)
    density = chebyshev.gapped_ground_state_density(0.013)
    assert np.abs(density(x) - (-(x / 4) * erf(x / 0.013))).max() <= 1e-15
    assert np.abs(density(x[np.abs(x) >= 0.065]) + np.abs(x[np.abs(x) >= 0.065]) / 4).max() <= 1e-13  # = -|ε|/4 beyond 5 widths
    assert chebyshev.moments_for_gapped_ground_state(7.2, 0.0105) == 6858 and chebyshev.moments_for_gapped_ground_state(1.0, 10.0) == 32
    beta = np.array([3.0, 2.5, 2.9, 1e-9, 4.0, 3.0])
    assert _krylov_length(beta, scale2=25.0) == 4 and _krylov_length(beta[:3], scale2=25.0) == 3


def test_bench_assembly_leaves_the_gpu_alone(api, monkeypatch):
    """bench.py takes its CPU baseline (which forks worker processes) before the first GPU call; the
    closing `with` block of its assembly must therefore make its Hermiticity test on the host even
    for large fills.  `hermiticity_check = "host"` never asks whether a GPU is there."""
    import bench
    from bodge_amd import hamiltonian

    def forbidden():
        raise AssertionError("the assembly asked for a GPU")

    monkeypatch.setattr(hamiltonian, "_gpu_present", forbidden)
    monkeypatch.setattr(hamiltonian, "DEVICE_HERMITICITY_MIN_BLOCKS", 1)
    for model in ("swave", "dwave", "potential", "texture", "peierls"):
        system = bench.build_system([12, 10, 1] if model != "dwave" else [6, 5, 4], model)
        assert system.hermiticity_check == "host" and system._devices == {}
        assert system._hermiticity_defect() < 1e-12
    system = api.Hamiltonian(api.CubicLattice((4, 4, 1)))
    system.hermiticity_check = "somewhere"
    with pytest.raises(ValueError):
        with system as (H, Δ):
            H.set_sites(api.σ0)


@pytest.mark.parametrize("scale,gap", [(8.0, 0.2), (8.0, 0.05), (20.0, 0.1)])
def test_gapped_ground_state_series_pointwise_bound(scale, gap):
    """The stated accuracy of the T = 0 route is a pointwise bound, not a cancellation in the trace
    (ADVICE r2): at M = moments_for_gapped_ground_state the truncated series of -(ε/4)·erf(5ε/gap) is
    within 1e-13 of -|ε|/4 for every gap ≤ |ε| < a; at 8·a/δ, round 2's order, it is 1e-11 .. 6e-11."""
    width = gap / 5
    density = chebyshev.gapped_ground_state_density(width)
    eps = np.concatenate([np.linspace(-0.999 * scale, -gap, 2001), np.linspace(gap, 0.999 * scale, 2001)])

    def worst(order):
        coeff = chebyshev.chebyshev_coefficients(lambda x: density(scale * x), order)
        return np.abs(np.polynomial.chebyshev.chebval(eps / scale, coeff) + np.abs(eps) / 4).max()

    order = chebyshev.moments_for_gapped_ground_state(scale, width)
    assert order == int(np.ceil(10 * scale / width)) + (int(np.ceil(10 * scale / width)) & 1)
    assert worst(order) <= 1e-13
    assert 1e-12 < worst(int(8 * scale / width)) < 1e-10
