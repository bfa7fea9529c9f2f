"""Parity at BASELINE.json's full sizes, where no CPU reference finishes in test time.

The CPU oracle needs ~0.6 s per block-step at 10^6 sites, so here it checks a few steps of a
few vectors directly, and the rest is covered by properties that do not depend on size:
exact norms of the start vectors, additivity over start vectors, equality of the kernel forms
(dictionary / streamed, real / complex arithmetic, packed / full blocks, strip order), bit
reproducibility, and the closed-form free energy of a diagonal Hamiltonian.
"""

import os

import numpy as np
import pytest

from oracle import cheb_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(api, hip_library):
    """configs[2]: CubicLattice((1000,1000,1)), s-wave + Zeeman, open boundaries (4M x 4M)."""
    from bodge_amd import chebyshev
    from bodge_amd.solver import DeviceSolver

    lattice = api.CubicLattice((1000, 1000, 1))
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        H.set_sites(3.0 * api.σ0 - 0.05 * api.σ3)
        Δ.set_sites(-0.1 * api.jσ2)
        H.set_bonds(-1.0 * api.σ0)
    indptr, indices, data = system.bsr_arrays()
    assert indices.size == 4_996_000  # 5 L^2 - 4 L
    scale = chebyshev.spectral_bound(indptr, data)
    solver = DeviceSolver(indptr, indices, data)
    solver.set_lattice_shape(lattice.shape)
    yield system, solver, scale
    solver.close()


def _with_env(solver, env, *args, **kwargs):
    """dots_random under library switches (set through bdg_set_option for the call, not os.environ)."""
    from bodge_amd import backend

    with backend.options(**env):
        out = solver.dots_random(*args, **kwargs)
        return out, solver.perf()


def test_first_steps_match_oracle_at_full_size(big):
    system, solver, scale = big
    bsr = system.matrix("bsr")
    steps, vectors = 3, 2
    ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(bsr.shape[0], 1, range(vectors)))
    got = solver.dots_random(scale, steps, vectors, seed=1)
    mu0 = bsr.shape[0]
    assert np.allclose(got[0], ref[0], rtol=0, atol=1e-12 * mu0) and np.allclose(got[1], ref[1], rtol=0, atol=1e-12 * mu0)


def test_size_independent_properties(big):
    system, solver, scale = big
    n = system.shape[0]
    steps = 24
    (d, e), perf = _with_env(solver, {}, scale, steps, 8, seed=0)
    assert perf["dict_blocks"] > 0 and perf["real_arithmetic"] == 1 and perf["ph_packed"] == 1
    assert perf["steps_per_launch"] == 3  # the three-step sweep is the default at this size
    # |v|^2 = 4N exactly for ±1 vectors, and <t_n|t_n> <= |v|^2 because |T_n| <= 1 on the spectrum
    assert np.array_equal(d[0], np.full(8, float(n)))
    assert np.all(d <= n * (1 + 1e-12)) and np.all(d > 0)
    # bit reproducible
    (d2, e2), _ = _with_env(solver, {}, scale, steps, 8, seed=0)
    assert np.array_equal(d, d2) and np.array_equal(e, e2)
    # additive over start vectors: columns do not interact (any batch split gives the same columns)
    (da, ea), _ = _with_env(solver, {}, scale, steps, 3, seed=0, first_id=0)
    (db, eb), _ = _with_env(solver, {}, scale, steps, 5, seed=0, first_id=3)
    assert np.allclose(np.hstack([da, db]), d, rtol=1e-13) and np.allclose(np.hstack([ea, eb]), e, rtol=1e-12, atol=1e-6)
    # every kernel form computes the same numbers
    for env, check in [
        ({"BODGE_AMD_SWEEP": "0"}, lambda p: p["steps_per_launch"] == 1 and p["dict_blocks"] > 0),
        ({"BODGE_AMD_ALTERNATE": "0", "BODGE_AMD_SWEEP_ZIGZAG": "0"}, lambda p: p["steps_per_launch"] == 3),
        ({"BODGE_AMD_SWEEP_SEGMENTS": "7"}, lambda p: p["steps_per_launch"] == 3),
        ({"BODGE_AMD_SWEEP_STEPS": "2"}, lambda p: p["steps_per_launch"] == 2),
        ({"BODGE_AMD_SWEEP_STEPS": "2", "BODGE_AMD_SWEEP_LANES": "4"}, lambda p: p["steps_per_launch"] == 2 and p["lanes_per_row"] == 4),
        ({"BODGE_AMD_SWEEP_LANES": "4"}, lambda p: p["steps_per_launch"] == 3 and p["lanes_per_row"] == 4),
        ({"BODGE_AMD_SWEEP_STEPS": "2", "BODGE_AMD_SWEEP_LANES": "1"}, lambda p: p["steps_per_launch"] == 2 and p["lanes_per_row"] == 1),
        ({"BODGE_AMD_DICT": "0"}, lambda p: p["dict_blocks"] == 0 and p["pipelined"] == 1),
        ({"BODGE_AMD_DICT": "0", "BODGE_AMD_PH": "0"}, lambda p: p["ph_packed"] == 0),
        ({"BODGE_AMD_DICT": "0", "BODGE_AMD_REAL": "0"}, lambda p: p["real_arithmetic"] == 0),
        ({"BODGE_AMD_DICT": "0", "BODGE_AMD_KERNEL": "generic"}, lambda p: p["pipelined"] == 0),
        ({"BODGE_AMD_SWEEP": "0", "BODGE_AMD_L2_BUDGET": "65536"}, lambda p: 0 < p["strip_rows"] < 1000),
    ]:
        (dx, ex), perf = _with_env(solver, env, scale, steps, 8, seed=0)
        assert check(perf), (env, perf)
        assert np.allclose(dx, d, rtol=1e-13), env
        assert np.allclose(ex, e, rtol=1e-12, atol=1e-6), env


def test_config3_1000x1000_512_moments_full_length_against_cpu_on_the_same_vectors(big):
    """configs[2] at full size AND full length (north_star: "free_energy matching SciPy to 1e-10"):
    (1000,1000,1) s-wave+Zeeman, M = 512 (256 launches), the 8 vectors of rank 0 (seed 0, ids 0..7).
    The C/OpenMP restatement runs the same vectors; it is pinned here to the numpy/scipy.sparse
    restatement on 2 vectors x 8 steps.  Every d_n, e_n within 1e-12 * 4N, F within 1e-10 relative
    (F defined by ref hamiltonian.py:305-321, evaluated as the trace of f(H))."""
    from bodge_amd import chebyshev
    from oracle import cheb_c

    system, solver, scale = big
    bsr = system.matrix("bsr")
    n, moments, vectors, temperature = bsr.shape[0], 512, 8, 0.5
    start = cheb_ref.random_block(n, 0, range(vectors))
    d2, e2 = cheb_ref.recurrence_dots(bsr, scale, 16, start[:, :2])
    cheb_c.set_threads(min(16, os.cpu_count() or 1))
    d_ref, e_ref = cheb_c.recurrence_dots(bsr, scale, moments, start, real=True)
    assert np.allclose(d_ref[:8, :2], d2, rtol=0, atol=1e-12 * n) and np.allclose(e_ref[:8, :2], e2, rtol=0, atol=1e-12 * n)
    # default route at this size: two recurrence steps per sweep (sweep.hpp) ...
    (d, e), perf = _with_env(solver, {}, scale, moments // 2, vectors, seed=0)
    # (two batches of 4 vectors; per batch 4 chunks of 63 steps = 21 sweeps each, and 3 + 1 steps at the end)
    assert perf["steps_per_launch"] == 3 and perf["lanes_per_row"] == 2 and perf["launches"] == 2 * (4 * 21 + 2)
    # ... the two batches side by side on two streams: they share a window shorter than their summed launch times,
    # and the launches moved less than full launches would (no reads in the first sweep, no stores in the last)
    assert perf["streams"] == 2 and 0 < perf["window_ms"] < perf["kernel_ms"]
    assert 0.9 * perf["launches"] * perf["bytes_per_launch"] < perf["bytes_moved"] < perf["launches"] * perf["bytes_per_launch"]
    (d1s, e1s), perf1 = _with_env(solver, {"BODGE_AMD_STREAMS": "1"}, scale, moments // 2, vectors, seed=0)
    # (one stream cuts 52 x-segments instead of 26: the workgroups' partial sums group differently, round-off only)
    assert perf1["streams"] == 1 and np.abs(d1s - d).max() <= 1e-13 * n and np.abs(e1s - e).max() <= 1e-13 * n
    # ... its two-step sibling ...
    (d2s, e2s), perf2 = _with_env(solver, {"BODGE_AMD_SWEEP_STEPS": "2"}, scale, moments // 2, vectors, seed=0)
    assert perf2["steps_per_launch"] == 2 and perf2["launches"] == 2 * (moments // 4)
    assert np.abs(d2s - d_ref).max() <= 1e-12 * n and np.abs(e2s - e_ref).max() <= 1e-12 * n
    assert np.abs(d - d_ref).max() <= 1e-12 * n and np.abs(e - e_ref).max() <= 1e-12 * n
    # ... and the one-step dictionary kernel, same vectors, full length as well
    (d1, e1), perf1 = _with_env(solver, {"BODGE_AMD_SWEEP": "0"}, scale, moments // 2, vectors, seed=0)
    assert perf1["steps_per_launch"] == 1 and perf1["launches"] == moments // 2 and perf1["dict_blocks"] > 0
    assert np.abs(d1 - d_ref).max() <= 1e-12 * n and np.abs(e1 - e_ref).max() <= 1e-12 * n
    f_ref = chebyshev.free_energy_series(chebyshev.dots_to_moments(d_ref, e_ref).mean(axis=1), scale, temperature)
    f_gpu = chebyshev.free_energy_series(solver.moments_random(scale, moments, vectors, seed=0) / vectors, scale, temperature)
    assert abs(f_gpu - f_ref) <= 1e-10 * abs(f_ref)
    # the same call through the reference-shaped API (own device mirror of the same matrix)
    f_api = system.free_energy(temperature, method="chebyshev", moments=moments, vectors=vectors, seed=0,
                               trace="stochastic", scale=scale)
    assert abs(f_api - f_ref) <= 1e-10 * abs(f_ref)
    # streamed (non-dictionary) complex128 kernels, the reference's own dtype: same vectors, 64 launches
    (dc, ec), perf_c = _with_env(solver, {"BODGE_AMD_DICT": "0", "BODGE_AMD_REAL": "0"}, scale, 64, vectors, seed=0)
    assert perf_c["real_arithmetic"] == 0 and perf_c["dict_blocks"] == 0 and perf_c["steps_per_launch"] == 1
    assert np.abs(dc - d_ref[:64]).max() <= 1e-12 * n and np.abs(ec - e_ref[:64]).max() <= 1e-12 * n


def test_stochastic_free_energy_is_stable_and_in_range(big, api):
    """512-moment stochastic-trace F of configs[2]: two disjoint sets of vectors must agree to the
    stochastic error, and F/N must sit near the 20x20 reference value per site."""
    system, solver, scale = big
    from bodge_amd import chebyshev

    n_sites = system.lattice.size
    f = []
    for first in (0, 8):
        mu = solver.moments_random(scale, 512, 8, seed=0, first_id=first) / 8
        f.append(chebyshev.free_energy_series(mu, scale, 0.5) / n_sites)
    # (an 8-vector estimate of F/N scatters by 3.7-4.5e-4 over sixteen disjoint vector sets and two seeds -
    # scratch/r4_stoch_spread.py; the difference of two by sqrt 2 more: four standard deviations)
    assert abs(f[0] - f[1]) < 2.5e-3
    assert -3.20 < f[0] < -3.10     # golden 20x20+Zeeman: F(0.5)/400 = -3.1433 (finite-size edges)


def test_diagonal_hamiltonian_has_closed_form_free_energy(api, hip_library):
    """No hopping: every site carries the same 4x4 block with eigenvalues ±(sqrt(μ²+Δ²) ± m), so F
    is known in closed form at any size.  The four unit vectors of one site give that site's
    trace exactly; the stochastic estimate over the whole lattice must agree to its noise level."""
    from bodge_amd import chebyshev

    lattice = api.CubicLattice((300, 300, 1))
    system = api.Hamiltonian(lattice)
    mu_, gap, m, temperature = 1.3, 0.4, 0.2, 0.35
    with system as (H, Δ):
        H.set_sites(mu_ * api.σ0 - m * api.σ3)
        Δ.set_sites(-gap * api.jσ2)
    eps = np.array([np.hypot(mu_, gap) + m, np.hypot(mu_, gap) - m])
    exact = lattice.size * np.sum(-eps / 2 - temperature * np.log1p(np.exp(-eps / temperature)))

    solver = system._solver()
    indptr, _, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    site = lattice[(150, 7, 0)]
    mu_site = solver.moments_unit(scale, 256, np.arange(4 * site, 4 * site + 4)).sum(axis=1)
    assert np.isclose(lattice.size * chebyshev.free_energy_series(mu_site, scale, temperature), exact, rtol=1e-12)

    estimate = system.free_energy(temperature, method="chebyshev", trace="stochastic", vectors=8, moments=256)
    assert np.isclose(estimate, exact, rtol=5e-3)


def test_config2_200x200_256_moments_against_cpu_on_the_same_vectors(api, hip_library):
    """configs[1] as SURVEY §8d C2 states it: (200,200,1) s-wave, M = 256, 64 Rademacher vectors,
    seed 0.  The CPU restatement runs the same vectors: every moment within 1e-12 (relative to
    mu_0 = 4N per vector), F within 1e-10 relative."""
    from bodge_amd import chebyshev
    from oracle import cheb_c

    lattice = api.CubicLattice((200, 200, 1))
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        H.set_sites(3.0 * api.σ0)
        Δ.set_sites(-0.1 * api.jσ2)
        H.set_bonds(-1.0 * api.σ0)
    bsr = system.matrix("bsr")
    assert bsr.indices.size == 199_200
    n, moments, vectors, temperature = bsr.shape[0], 256, 64, 0.5
    scale = cheb_ref.spectral_bound(bsr)
    start = cheb_ref.random_block(n, 0, range(vectors))
    # numpy/scipy oracle on the first 8 vectors pins the C restatement, which then covers all 64
    d8, e8 = cheb_ref.recurrence_dots(bsr, scale, moments, start[:, :8])
    d_ref, e_ref = cheb_c.recurrence_dots(bsr, scale, moments, start, real=True)
    assert np.allclose(d_ref[:, :8], d8, rtol=0, atol=1e-12 * n) and np.allclose(e_ref[:, :8], e8, rtol=0, atol=1e-12 * n)
    d, e = system._solver().dots_random(scale, moments // 2, vectors, seed=0)  # cached solver, reused below
    assert np.abs(d - d_ref).max() <= 1e-12 * n and np.abs(e - e_ref).max() <= 1e-12 * n
    f_ref = chebyshev.free_energy_series(chebyshev.dots_to_moments(d_ref, e_ref).mean(axis=1), scale, temperature)
    f_gpu = system.free_energy(temperature, method="chebyshev", moments=moments, vectors=vectors, seed=0,
                               trace="stochastic")
    assert abs(f_gpu - f_ref) <= 1e-10 * abs(f_ref)


def test_config4_100cubed_dwave_slabs_match_the_whole_matrix(api, hip_library):
    """configs[3]: (100,100,100) d-wave on all bonds, 8 row slabs of 12/13 x-planes with a halo
    exchange per step (same-process transport on one GPU) against the undivided matrix, and the
    first steps against the CPU oracle."""
    from bodge_amd.solver import DeviceSolver, SlabGroup

    lattice = api.CubicLattice((100, 100, 100))
    system = api.Hamiltonian(lattice)
    pairs = lattice.bond_array(coords=True)
    with system as (H, Δ):
        H.set_sites(3.0 * api.σ0)
        H.set_bonds(-1.0 * api.σ0)
        Δ.set_bonds(-0.1 * api.dwave()(pairs[:, 0], pairs[:, 1]))  # z bonds give 0 (ref tests/test_hamiltonian.py:249-251)
    indptr, indices, data = system.bsr_arrays()
    assert indices.size == 6_940_000  # 7 L^3 - 6 L^2
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    steps, vectors = 12, 4
    with DeviceSolver.from_hamiltonian(system) as whole:
        mono = whole.dots_random(scale, steps, vectors, seed=4)
    with SlabGroup.from_hamiltonian(system, 8) as group:
        sizes = sorted(p.n_own // 10_000 for p in group.plans)
        assert sizes == [12, 12, 12, 12, 13, 13, 13, 13]
        assert all(p.halo_rows in (10_000, 20_000) for p in group.plans)
        split = group.dots_random(scale, steps, vectors, seed=4)
    n = bsr.shape[0]
    assert np.allclose(split[0], mono[0], rtol=0, atol=1e-12 * n) and np.allclose(split[1], mono[1], rtol=0, atol=1e-12 * n)
    ref = cheb_ref.recurrence_dots(bsr, scale, 4, cheb_ref.random_block(n, 4, range(vectors)))
    assert np.allclose(mono[0][:2], ref[0], rtol=0, atol=1e-12 * n) and np.allclose(mono[1][:2], ref[1], rtol=0, atol=1e-12 * n)


def test_config4_100cubed_dwave_256_moments_full_length_whole_and_slabs_against_cpu(api, hip_library):
    """configs[3] at full length: (100,100,100) d-wave, M = 256 (128 launches), 4 vectors (seed 4),
    the undivided matrix AND the 8-slab group with a halo exchange per step, both against the
    C/OpenMP restatement on the same vectors (pinned to the numpy/scipy restatement on the first 4
    steps): every d_n, e_n within 1e-12 * 4N, F within 1e-10 relative."""
    from bodge_amd import chebyshev
    from bodge_amd.solver import DeviceSolver, SlabGroup
    from oracle import cheb_c

    lattice = api.CubicLattice((100, 100, 100))
    system = api.Hamiltonian(lattice)
    pairs = lattice.bond_array(coords=True)
    with system as (H, Δ):
        H.set_sites(3.0 * api.σ0)
        H.set_bonds(-1.0 * api.σ0)
        Δ.set_bonds(-0.1 * api.dwave()(pairs[:, 0], pairs[:, 1]))
    bsr = system.matrix("bsr")
    scale = cheb_ref.spectral_bound(bsr)
    n, moments, vectors, temperature = bsr.shape[0], 256, 4, 0.5
    start = cheb_ref.random_block(n, 4, range(vectors))
    d4, e4 = cheb_ref.recurrence_dots(bsr, scale, 8, start[:, :2])
    cheb_c.set_threads(min(16, os.cpu_count() or 1))
    d_ref, e_ref = cheb_c.recurrence_dots(bsr, scale, moments, start, real=True)
    assert np.allclose(d_ref[:4, :2], d4, rtol=0, atol=1e-12 * n) and np.allclose(e_ref[:4, :2], e4, rtol=0, atol=1e-12 * n)
    f_ref = chebyshev.free_energy_series(chebyshev.dots_to_moments(d_ref, e_ref).mean(axis=1), scale, temperature)
    with DeviceSolver.from_hamiltonian(system) as whole:
        mono = whole.dots_random(scale, moments // 2, vectors, seed=4)
        # two lane groups of 8 vectors side by side on two streams (K8 with 4 lanes per site): the first four are the same vectors
        wide = whole.dots_random(scale, moments // 2, 16, seed=4)
        wide_perf = whole.perf()
    assert wide_perf["rolling"] == 1 and wide_perf["streams"] == 2 and wide_perf["lanes_per_row"] == 4 and wide_perf["launches"] == moments
    assert np.abs(wide[0][:, :vectors] - d_ref).max() <= 1e-12 * n and np.abs(wide[1][:, :vectors] - e_ref).max() <= 1e-12 * n
    with SlabGroup.from_hamiltonian(system, 8) as group:
        split = group.dots_random(scale, moments // 2, vectors, seed=4)
        # slabs of 12 / 13 whole planes: the rolling stencil kernel, neighbours' boundary planes read in place
        assert all(member.perf()["rolling"] == 1 and member.perf()["launches"] == moments // 2 for member in group.members)
    for d, e in (mono, split):
        assert np.abs(d - d_ref).max() <= 1e-12 * n and np.abs(e - e_ref).max() <= 1e-12 * n
        f = chebyshev.free_energy_series(chebyshev.dots_to_moments(d, e).mean(axis=1), scale, temperature)
        assert abs(f - f_ref) <= 1e-10 * abs(f_ref)


def test_complex_hamiltonian_at_full_size(api, hip_library):
    """A genuinely complex matrix at 10^6 sites (Peierls phase on the x bonds: uniform flux-free
    gauge field, H_ij = -e^{±iφ} σ0): the complex dictionary kernel against the CPU oracle on the
    first steps, its streamed form against it bit-compatibly to round-off, and norm conservation
    of the start vectors."""
    from bodge_amd import chebyshev
    from bodge_amd.solver import DeviceSolver

    lattice = api.CubicLattice((1000, 1000, 1))
    system = api.Hamiltonian(lattice)
    pairs = lattice.bond_array(axis=0, coords=True)          # directed x bonds, both directions
    forward = pairs[:, 1, 0] > pairs[:, 0, 0]
    phase = np.where(forward, np.exp(0.3j), np.exp(-0.3j))
    with system as (H, Δ):
        H.set_sites(3.0 * api.σ0 - 0.05 * api.σ3)
        Δ.set_sites(-0.1 * api.jσ2)
        H.set_bonds(-phase[:, None, None] * api.σ0, axis=0)
        H.set_bonds(-1.0 * api.σ0, axis=1)
    indptr, indices, data = system.bsr_arrays()
    assert np.abs(data.imag).max() > 0.2 and indices.size == 4_996_000
    scale = chebyshev.spectral_bound(indptr, data)
    bsr = system.matrix("bsr")
    n, steps, vectors = bsr.shape[0], 3, 2
    ref = cheb_ref.recurrence_dots(bsr, scale, 2 * steps, cheb_ref.random_block(n, 7, range(vectors), cheb_ref.VEC_Z4))
    with DeviceSolver(indptr, indices, data) as solver:
        solver.set_lattice_shape(lattice.shape)
        got, perf = _with_env(solver, {}, scale, steps, vectors, seed=7, kind=cheb_ref.VEC_Z4)
        assert perf["real_arithmetic"] == 0 and perf["dict_blocks"] > 0
        streamed, perf_s = _with_env(solver, {"BODGE_AMD_DICT": "0"}, scale, steps, vectors, seed=7, kind=cheb_ref.VEC_Z4)
        assert perf_s["dict_blocks"] == 0
    for out in (got, streamed):
        assert np.allclose(out[0], ref[0], rtol=0, atol=1e-12 * n) and np.allclose(out[1], ref[1], rtol=0, atol=1e-12 * n)
    assert np.allclose(got[0][0], n, rtol=1e-15)  # |v|^2 = 4N exactly for Z4 vectors


def test_complex_hamiltonian_full_length_at_full_size(api, hip_library):
    """The reference's own dtype on the headline kernel, at full size AND full length: the Peierls-phase
    matrix of the test above (1000 x 1000, H_ij = -e^{±iφ} σ0 on the x bonds), M = 512, 4 Z4 vectors.
    Default route = cheb_sweep3<ComplexPHMode> (three steps per sweep, complex arithmetic), then the
    one-step dictionary and streamed kernels; every d_n, e_n against the C/OpenMP restatement in
    complex arithmetic on the same vectors (itself pinned to the numpy/scipy restatement on 2 vectors
    x 8 steps) within 1e-12 * 4N, F within 1e-10 relative (F as ref hamiltonian.py:305-321)."""
    from bodge_amd import chebyshev
    from bodge_amd.solver import DeviceSolver
    from oracle import cheb_c

    lattice = api.CubicLattice((1000, 1000, 1))
    system = api.Hamiltonian(lattice)
    pairs = lattice.bond_array(axis=0, coords=True)
    phase = np.where(pairs[:, 1, 0] > pairs[:, 0, 0], np.exp(0.3j), np.exp(-0.3j))
    with system as (H, Δ):
        H.set_sites(3.0 * api.σ0 - 0.05 * api.σ3)
        Δ.set_sites(-0.1 * api.jσ2)
        H.set_bonds(-phase[:, None, None] * api.σ0, axis=0)
        H.set_bonds(-1.0 * api.σ0, axis=1)
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    bsr = system.matrix("bsr")
    n, moments, vectors, temperature = bsr.shape[0], 512, 4, 0.5
    start = cheb_ref.random_block(n, 3, range(vectors), cheb_ref.VEC_Z4)
    d2, e2 = cheb_ref.recurrence_dots(bsr, scale, 16, start[:, :2])
    cheb_c.set_threads(min(16, os.cpu_count() or 1))
    d_ref, e_ref = cheb_c.recurrence_dots(bsr, scale, moments, start, real=False)
    assert np.allclose(d_ref[:8, :2], d2, rtol=0, atol=1e-12 * n) and np.allclose(e_ref[:8, :2], e2, rtol=0, atol=1e-12 * n)
    f_ref = chebyshev.free_energy_series(chebyshev.dots_to_moments(d_ref, e_ref).mean(axis=1), scale, temperature)
    with DeviceSolver(indptr, indices, data) as solver:
        solver.set_lattice_shape(lattice.shape)
        for env, expect in [({}, {"steps_per_launch": 3}),
                            ({"BODGE_AMD_SWEEP": "0"}, {"steps_per_launch": 1, "pipelined": 0}),
                            ({"BODGE_AMD_SWEEP": "0", "BODGE_AMD_DICT": "0"}, {"steps_per_launch": 1, "dict_blocks": 0})]:
            (d, e), perf = _with_env(solver, env, scale, moments // 2, vectors, seed=3, kind=cheb_ref.VEC_Z4)
            assert perf["real_arithmetic"] == 0 and all(perf[k] == v for k, v in expect.items()), (env, perf)
            assert (perf["dict_blocks"] > 0) == ("BODGE_AMD_DICT" not in env)
            assert np.abs(d - d_ref).max() <= 1e-12 * n and np.abs(e - e_ref).max() <= 1e-12 * n, env
            f_gpu = chebyshev.free_energy_series(chebyshev.dots_to_moments(d, e).mean(axis=1), scale, temperature)
            assert abs(f_gpu - f_ref) <= 1e-10 * abs(f_ref), env


@pytest.mark.parametrize("kind", ["potential", "texture", "ssd", "landau"])
def test_position_dependent_onsite_terms_full_length_at_full_size(api, hip_library, kind):
    """1000 x 1000 with a different on-site block at every site (10^6 distinct blocks - no dictionary),
    M = 512: the three-step sweep that streams the on-site blocks (cheb_sweep3 OS, sweep.hpp) against
    the C/OpenMP restatement on the same vectors, every d_n, e_n within 1e-12 * 4N and F within 1e-10
    relative; then the one-step streamed-blocks kernel, same gate.  "potential": random potential and
    gap amplitude, real arithmetic, 8 Rademacher vectors; "texture": an exchange field whose direction
    varies from site to site (σ1, σ2, σ3 components: complex blocks), 4 Z4 vectors; "ssd": every term -
    hopping included - scaled by the reference's sine-squared envelope (ref hamiltonian.py:488-531):
    the bond blocks are streamed as well (cheb_sweep3 OS = 2); "landau" (round 4): the texture model in a magnetic
    field - Peierls phases exp(±i B y) of the Landau gauge on the x bonds, 2000 distinct complex bond blocks: complex
    site records (224 B per site), 4 Z4 vectors."""
    from bodge_amd import chebyshev
    from bodge_amd.solver import DeviceSolver
    from oracle import cheb_c

    lattice = api.CubicLattice((1000, 1000, 1))
    system = api.Hamiltonian(lattice)
    rng = np.random.default_rng(11)
    sites = lattice.size
    with system as (H, Δ):
        if kind == "potential":
            H.set_sites((3.0 + rng.uniform(-0.5, 0.5, sites))[:, None, None] * api.σ0 - 0.05 * api.σ3)
            Δ.set_sites(-rng.uniform(0.05, 0.15, sites)[:, None, None] * api.jσ2)
        elif kind == "ssd":
            φ = api.ssd(system)
            coords = np.stack(np.unravel_index(np.arange(sites), lattice.shape), axis=-1)
            pairs = lattice.bond_array(coords=True)
            on_site, on_bond = φ(coords, coords)[:, None, None], φ(pairs[:, 0], pairs[:, 1])[:, None, None]
            H.set_sites(on_site * (3.0 * api.σ0 - 0.05 * api.σ3))
            Δ.set_sites(-0.1 * on_site * api.jσ2)
            H.set_bonds(-on_bond * api.σ0)
        else:
            th, ph = rng.uniform(0, np.pi, sites)[:, None, None], rng.uniform(0, 2 * np.pi, sites)[:, None, None]
            H.set_sites(3.0 * api.σ0 - 0.3 * (np.sin(th) * np.cos(ph) * api.σ1 + np.sin(th) * np.sin(ph) * api.σ2 + np.cos(th) * api.σ3))
            Δ.set_sites(-0.1 * api.jσ2)
        if kind == "landau":
            pairs = lattice.bond_array(axis=0, coords=True)  # directed x bonds: a phase one way, its conjugate back
            flux = 0.0123 * pairs[:, 0, 1]
            phase = np.exp(1j * np.where(pairs[:, 1, 0] > pairs[:, 0, 0], flux, -flux))
            H.set_bonds(-phase[:, None, None] * api.σ0, axis=0)
            H.set_bonds(-1.0 * api.σ0, axis=1)
        elif kind != "ssd":
            H.set_bonds(-1.0 * api.σ0)
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    bsr = system.matrix("bsr")
    real = kind not in ("texture", "landau")
    assert (np.abs(data.imag).max() == 0) == real
    n, moments, temperature = bsr.shape[0], 512, 0.5
    vectors, vec_kind = (8, cheb_ref.VEC_RADEMACHER) if real else (4, cheb_ref.VEC_Z4)
    start = cheb_ref.random_block(n, 4, range(vectors), vec_kind)
    d2, e2 = cheb_ref.recurrence_dots(bsr, scale, 16, start[:, :2])
    cheb_c.set_threads(min(16, os.cpu_count() or 1))
    d_ref, e_ref = cheb_c.recurrence_dots(bsr, scale, moments, start, real=real)
    assert np.allclose(d_ref[:8, :2], d2, rtol=0, atol=1e-12 * n) and np.allclose(e_ref[:8, :2], e2, rtol=0, atol=1e-12 * n)
    f_ref = chebyshev.free_energy_series(chebyshev.dots_to_moments(d_ref, e_ref).mean(axis=1), scale, temperature)
    with DeviceSolver(indptr, indices, data) as solver:
        solver.set_lattice_shape(lattice.shape)
        (d, e), perf = _with_env(solver, {}, scale, moments // 2, vectors, seed=4, kind=vec_kind)
        bonds_too = kind in ("ssd", "landau")
        assert perf["onsite_streamed"] == (2 if bonds_too else 1) and perf["steps_per_launch"] == 3, perf
        two_lanes = kind == "potential"  # (real on-site records: 32-slot windows since round 4, two lane groups of 4 vectors)
        assert perf["lanes_per_row"] == (2 if two_lanes else 4), perf
        assert perf["real_arithmetic"] == (1 if real else 0) and perf["dict_blocks"] == 1 and perf["dict_skipped"] == 1
        # per lane group four chunks of 63 steps, then 3 + 1
        assert perf["launches"] == (2 if two_lanes else 1) * (4 * 21 + 2)
        (d1, e1), perf1 = _with_env(solver, {"BODGE_AMD_SWEEP": "0"}, scale, moments // 2, vectors, seed=4, kind=vec_kind)
        assert perf1["onsite_streamed"] == 0 and perf1["steps_per_launch"] == 1 and perf1["dict_blocks"] == 0
    for got_d, got_e in ((d, e), (d1, e1)):
        assert np.abs(got_d - d_ref).max() <= 1e-12 * n and np.abs(got_e - e_ref).max() <= 1e-12 * n
        f_gpu = chebyshev.free_energy_series(chebyshev.dots_to_moments(got_d, got_e).mean(axis=1), scale, temperature)
        assert abs(f_gpu - f_ref) <= 1e-10 * abs(f_ref)


def test_periodic_lattice_at_full_size_sweeps_match_one_step_kernels(api, hip_library):
    """A 10^6-site torus (the reference's `lattice.edges()` terms set on both axes): the wrap-around
    blocks close the planes and their stack into rings for the multi-step sweep kernels; 10 steps
    (three sweeps and a lone step) against the one-step kernels on the same vectors, which take
    the wrap blocks as ordinary BSR blocks, and the first steps against the CPU oracle."""
    from bodge_amd import chebyshev
    from bodge_amd.solver import DeviceSolver

    lattice = api.CubicLattice((1000, 1000, 1))
    system = api.Hamiltonian(lattice)
    with system as (H, Δ):
        H.set_sites(3.0 * api.σ0 - 0.05 * api.σ3)
        Δ.set_sites(-0.1 * api.jσ2)
        H.set_bonds(-1.0 * api.σ0)
        H.set_edges(-1.0 * api.σ0)
    indptr, indices, data = system.bsr_arrays()
    assert indices.size == 5_000_000  # every row has its five blocks
    scale = chebyshev.spectral_bound(indptr, data)
    n = 4 * lattice.size
    with DeviceSolver(indptr, indices, data) as solver:
        solver.set_lattice_shape(lattice.shape)
        (d, e), perf = _with_env(solver, {}, scale, 10, 8, seed=2)
        assert perf["steps_per_launch"] == 3
        (d1, e1), perf1 = _with_env(solver, {"BODGE_AMD_SWEEP": "0"}, scale, 10, 8, seed=2)
        assert perf1["steps_per_launch"] == 1
    assert np.abs(d - d1).max() <= 1e-12 * n and np.abs(e - e1).max() <= 1e-12 * n
    bsr = system.matrix("bsr")
    ref = cheb_ref.recurrence_dots(bsr, scale, 4, cheb_ref.random_block(bsr.shape[0], 2, range(2)))
    assert np.allclose(d[:2, :2], ref[0], rtol=0, atol=1e-12 * n) and np.allclose(e[:2, :2], ref[1], rtol=0, atol=1e-12 * n)
