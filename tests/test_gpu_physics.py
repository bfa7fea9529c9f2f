"""Physics integration tests on the GPU path.

Each test mirrors one test of the reference's tests/test_physics.py (cited per
test) but goes through bodge_amd's device path.  Random angles are seeded.
"""

import numpy as np
import pytest

from bodge_amd import CubicLattice, Hamiltonian, pwave
from bodge_amd.common import jσ2, π, σ0, σ1, σ2, σ3

pytestmark = pytest.mark.gpu


def test_superconducting_gap_existence():
    """ref tests/test_physics.py:16-67."""
    lattice = CubicLattice((16, 16, 1))
    system = Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = -1.5 * σ0
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * σ0
    gap, site = 0.5, (8, 8, 0)
    ω = np.array([-1.2 * gap, -0.8 * gap, +0.8 * gap, 1.2 * gap])
    ρ1 = system.ldos(site, ω)
    ε1 = np.min(system.diagonalize()[0])
    with system as (H, Δ):
        for i in lattice.sites():
            Δ[i, i] = gap * jσ2
    ρ2 = system.ldos(site, ω)
    ε2 = np.min(system.diagonalize()[0])
    assert ρ2[1] < ρ1[1] and ρ2[2] < ρ1[2]
    assert ρ2[0] > ρ1[0] and ρ2[3] > ρ1[3]
    assert ε2 > ε1


def test_superconducting_gap_scaling():
    """ref tests/test_physics.py:70-112."""
    lattice = CubicLattice((32, 1, 1))
    system = Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = -1.5 * σ0
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * σ0
    gaps = []
    for Δ0 in [0.0, 0.01, 0.03, 0.1, 0.3, 1.0]:
        with system as (H, Δ):
            for i in lattice.sites():
                Δ[i, i] = Δ0 * jσ2
        gaps.append(np.min(system.diagonalize()[0]))
    assert all(a < b for a, b in zip(gaps[:-1], gaps[1:]))


def test_magnetic_isotropy():
    """ref tests/test_physics.py:115-172: F and LDOS invariant under field rotation to rtol 1e-10."""
    lattice = CubicLattice((128, 1, 1))
    system = Hamiltonian(lattice)
    i0, E0, T = (64, 0, 0), [0.0, 0.01], 0.01
    with system as (H, Δ):
        for i in lattice.sites():
            Δ[i, i] = -0.1 * jσ2
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * σ0
    F0 = system.free_energy(T)
    ρ0 = system.ldos(i0, E0)[0]
    rng = np.random.default_rng(42)
    Fs, ρs = [], []
    for _ in range(4):
        θ, ϕ = 2 * π * rng.random(2)
        axis = np.cos(θ) * σ1 + np.sin(θ) * np.cos(ϕ) * σ2 + np.sin(θ) * np.sin(ϕ) * σ3
        with system as (H, Δ):
            for i in lattice.sites():
                H[i, i] = -0.05 * axis
        Fs.append(system.free_energy(T))
        ρs.append(system.ldos(i0, E0)[0])
    assert all(not np.allclose(F0, F, rtol=1e-10) for F in Fs)
    assert all(not np.allclose(ρ0, ρ, rtol=1e-10) for ρ in ρs)
    assert all(np.allclose(a, b, rtol=1e-10) for a, b in zip(Fs[:-1], Fs[1:]))
    assert all(np.allclose(a, b, rtol=1e-10) for a, b in zip(ρs[:-1], ρs[1:]))


def test_superconducting_spinvalve():
    """ref tests/test_physics.py:175-228."""
    lattice = CubicLattice((128, 1, 1))
    system = Hamiltonian(lattice)
    left = lambda i: i[0] < 32
    right = lambda i: i[0] >= 128 - 32
    with system as (H, Δ):
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * σ0
        for i in lattice.sites():
            if not left(i) and not right(i):
                Δ[i, i] = -0.3 * jσ2
            else:
                H[i, i] = -0.7 * σ3
    parallel = system.free_energy(0.001)
    with system as (H, Δ):
        for i in lattice.sites():
            if right(i):
                H[i, i] = +0.7 * σ3
    assert system.free_energy(0.001) < parallel


def test_odd_frequency_peak():
    """ref tests/test_physics.py:231-269."""
    lattice = CubicLattice((128, 1, 1))
    system = Hamiltonian(lattice)
    Δ0 = 0.3
    with system as (H, Δ):
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * σ0
        for i in lattice.sites():
            Δ[i, i] = -Δ0 * jσ2
    E0 = [0.0, 0.05 * Δ0]
    z1 = system.ldos((63, 0, 0), E0)[0]
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = -0.5 * Δ0 * σ2
    z2 = system.ldos((63, 0, 0), E0)[0]
    assert z1 >= 0 and z2 >= z1


def test_energy_decreases_with_temperature():
    """ref tests/test_physics.py:272-297, on both device algorithms."""
    lattice = CubicLattice((10, 10, 1))
    system = Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = -2.0 * σ0
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * σ0
    for method in ("dense", "chebyshev"):
        values = [system.free_energy(T, method=method) for T in [0.01, 0.1, 0.5, 1.0]]
        assert all(a > b for a, b in zip(values[:-1], values[1:]))


def test_pwave_edge_states():
    """ref tests/test_physics.py:300-339."""
    lattice = CubicLattice((31, 31, 1))
    system = Hamiltonian(lattice)
    σp = pwave("e_z * p_x")
    with system as (H, Δ):
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * σ0
            Δ[i, j] = -0.1 * σp(i, j)
    energies = [0.0, 0.025]
    ρ = {site: system.ldos(site, energies)[0] for site in [(15, 15, 0), (15, 0, 0), (0, 15, 0), (0, 0, 0)]}
    assert ρ[(0, 15, 0)] > ρ[(15, 15, 0)] and ρ[(0, 15, 0)] > ρ[(15, 0, 0)]
    assert ρ[(0, 0, 0)] > ρ[(15, 15, 0)] and ρ[(0, 0, 0)] > ρ[(15, 0, 0)]


def test_josephson_minigap():
    """ref tests/test_physics.py:342-387 (complex order parameter: exercises the complex path)."""
    lattice = CubicLattice((128, 1, 1))

    def minigap(ϕ):
        system = Hamiltonian(lattice)
        with system as (H, Δ):
            for i in lattice.sites():
                if i[0] < 32:
                    Δ[i, i] = -3.0 * jσ2 * np.exp(-1j * ϕ / 2)
                if i[0] >= 128 - 32:
                    Δ[i, i] = -3.0 * jσ2 * np.exp(+1j * ϕ / 2)
            for i, j in lattice.bonds():
                H[i, j] = -1.0 * σ0
        return np.min(system.diagonalize()[0])

    g = [minigap(x * π) for x in (0.0, 0.5, 1.0, 1.5, 2.0)]
    assert g[0] > g[1] > g[2]
    assert np.allclose(g[0], g[4]) and np.allclose(g[1], g[3])
