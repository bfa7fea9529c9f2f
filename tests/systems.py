"""Named test systems, written against the public API only.

Every builder takes a namespace `ns` that provides CubicLattice, Hamiltonian,
the Pauli constants and the pairing helpers.  `tests/golden/make_golden.py`
runs them with the *reference* package to record golden values; the test-suite
runs the same builders with `bodge_amd`.  Parameters follow the reference's own
tests and README where cited.
"""

from __future__ import annotations

import numpy as np


def _pauli_mix(ns, c):
    return c[0] * ns.σ0 + c[1] * ns.σ1 + c[2] * ns.σ2 + c[3] * ns.σ3


def swave_square(ns, L=20, zeeman=0.0, gap=0.1, mu=3.0):
    """README.md:73-86 model: H_ii = μσ0 - mσ3, Δ_ii = -Δ0 iσ2, bonds -σ0, open edges."""
    lattice = ns.CubicLattice((L, L, 1))
    system = ns.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = mu * ns.σ0 - zeeman * ns.σ3
            Δ[i, i] = -gap * ns.jσ2
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * ns.σ0
    return system


def snf_trilayer(ns):
    """tests/test_hamiltonian.py:431-443: S/N/F stack on (10,7,3)."""
    lattice = ns.CubicLattice((10, 7, 3))
    system = ns.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            if i[0] <= 3:
                H[i, i] = -0.5 * ns.σ0
                Δ[i, i] = -1.0 * ns.jσ2
            if i[0] >= 7:
                H[i, i] = +0.5 * ns.σ0 + 1.5 * ns.σ3
        for i, j in lattice.bonds():
            H[i, j] = -1 * ns.σ0
    return system


def magnetic_barrier(ns):
    """tests/test_hamiltonian.py:329-340: (10,3,2) with a magnetic barrier."""
    lattice = ns.CubicLattice((10, 3, 2))
    system = ns.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = 4 * ns.σ0
            if i[0] > 5:
                Δ[i, i] = 1 * ns.jσ2
            elif i[0] > 3:
                H[i, i] = 6 * ns.σ0 + 2 * ns.σ3
        for i, j in lattice.bonds():
            H[i, j] = -1 * ns.σ0
    return system


def complex_hopping(ns):
    """tests/test_hamiltonian.py:394-406: (2,3,5) with σ2 on-site and σ1 hopping terms."""
    lattice = ns.CubicLattice((2, 3, 5))
    system = ns.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = 4 * ns.σ0
            if i[0] == 0:
                Δ[i, i] = 1 * ns.jσ2
            else:
                H[i, i] = 4 * ns.σ0 + 1 * ns.σ2
        for i, j in lattice.bonds():
            H[i, j] = -1 * ns.σ0 + 2 * ns.σ1
    return system


def dwave_cube(ns, L=8):
    """Pattern of tests/test_hamiltonian.py:276-284 in 3-D: d-wave pairing on every bond."""
    lattice = ns.CubicLattice((L, L, L))
    system = ns.Hamiltonian(lattice)
    d = ns.dwave()
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = 3 * ns.σ0
        for i, j in lattice.bonds():
            H[i, j] = -1 * ns.σ0
            Δ[i, j] = -0.1 * d(i, j)
    return system


def swave_ldos_square(ns):
    """(16,16,1) gapped s-wave with Zeeman splitting, probed at the centre site."""
    lattice = ns.CubicLattice((16, 16, 1))
    system = ns.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = 3 * ns.σ0 - 0.05 * ns.σ3
            Δ[i, i] = -0.2 * ns.jσ2
        for i, j in lattice.bonds():
            H[i, j] = -1 * ns.σ0
    return system


def random_periodic(ns, shape=(3, 5, 7), seed=1234):
    """tests/test_hamiltonian.py:28-47 pattern with a seeded generator: dense complex
    on-site, bond and periodic-edge terms (hopping symmetric so the result is Hermitian)."""
    rng = np.random.default_rng(seed)
    r = rng.random
    lattice = ns.CubicLattice(shape)
    system = ns.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = _pauli_mix(ns, r(4))
            Δ[i, i] = (r() * ns.σ1 + r() * ns.σ2 + r() * ns.σ3) @ ns.jσ2
        for i, j in lattice.bonds():
            t = _pauli_mix(ns, r(4))
            H[i, j] = t
            H[j, i] = t
            Δ[i, j] = (r() * ns.σ1 + r() * ns.σ2 + r() * ns.σ3) @ ns.jσ2
        for i, j in lattice.edges():
            t = _pauli_mix(ns, r(4))
            H[i, j] = t
            H[j, i] = t
            Δ[i, j] = (r() * ns.σ1 + r() * ns.σ2 + r() * ns.σ3) @ ns.jσ2
    return system


def pwave_square(ns, L=31):
    """tests/test_physics.py:306-316: p_x-wave triplet pairing on a square lattice."""
    lattice = ns.CubicLattice((L, L, 1))
    system = ns.Hamiltonian(lattice)
    sp = ns.pwave("e_z * p_x")
    with system as (H, Δ):
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * ns.σ0
            Δ[i, j] = -0.1 * sp(i, j)
    return system


def field_chain(ns, theta=0.7, phi=2.1, L=128):
    """tests/test_physics.py:125-157: s-wave chain in a tilted exchange field."""
    lattice = ns.CubicLattice((L, 1, 1))
    system = ns.Hamiltonian(lattice)
    axis = (
        np.cos(theta) * ns.σ1
        + np.sin(theta) * np.cos(phi) * ns.σ2
        + np.sin(theta) * np.sin(phi) * ns.σ3
    )
    with system as (H, Δ):
        for i in lattice.sites():
            Δ[i, i] = -0.1 * ns.jσ2
            H[i, i] = -0.05 * axis
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * ns.σ0
    return system


def peierls_square(ns, L=30, phase=0.3):
    """Genuinely complex matrix for the Hermitian (zheevd) dense route: the README model with a
    Peierls phase e^{±iφ} on the x bonds (BASELINE config 5 ladder rung n = 3600)."""
    lattice = ns.CubicLattice((L, L, 1))
    system = ns.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = 3.0 * ns.σ0 - 0.05 * ns.σ3
            Δ[i, i] = -0.1 * ns.jσ2
        for i, j in lattice.bonds():
            step = j[0] - i[0]
            H[i, j] = -np.exp(1j * phase * step) * ns.σ0
    return system


def swave_chain(ns, L=300, zeeman=0.05, gap=0.1, mu=1.0):
    """(L,1,1) s-wave chain with Zeeman splitting: the literal "300" of BASELINE config 5 at a
    size the dense path can hold."""
    lattice = ns.CubicLattice((L, 1, 1))
    system = ns.Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            H[i, i] = mu * ns.σ0 - zeeman * ns.σ3
            Δ[i, i] = -gap * ns.jσ2
        for i, j in lattice.bonds():
            H[i, j] = -1.0 * ns.σ0
    return system


# name -> (builder, kwargs, what to record)
CATALOG = {
    "swave20": dict(build=swave_square, kwargs={}, temps=[0.0, 0.01, 0.1, 0.5, 1.0], spectrum=True),
    "swave20_zeeman": dict(
        build=swave_square, kwargs=dict(zeeman=0.05), temps=[0.0, 0.1, 0.5], spectrum=True
    ),
    "snf": dict(build=snf_trilayer, kwargs={}, temps=[0.0, 0.01, 0.1, 1.0], spectrum=True),
    "barrier": dict(build=magnetic_barrier, kwargs={}, temps=[0.1], spectrum=True, triple=True),
    "complex235": dict(
        build=complex_hopping, kwargs={}, temps=[0.001, 0.01, 0.1, 1.0], spectrum=True, triple=True
    ),
    "dwave8": dict(build=dwave_cube, kwargs={}, temps=[0.5], spectrum=True),
    "ldos16": dict(
        build=swave_ldos_square,
        kwargs={},
        temps=[0.5],
        spectrum=True,
        ldos=[((8, 8, 0), list(np.linspace(-0.3, 0.3, 7))), ((0, 3, 0), [0.0, 0.25, 0.5, 1.0])],
    ),
    "random357": dict(
        build=random_periodic,
        kwargs={},
        temps=[0.0, 0.05, 0.5],
        spectrum=True,
        triple=True,
        ldos=[((1, 2, 3), [0.0, 0.01, 0.10, 0.50, 1.00, 2.00, 4.00])],
    ),
    "pwave31": dict(
        build=pwave_square,
        kwargs={},
        temps=[],
        spectrum=False,
        ldos=[((15, 15, 0), [0.0, 0.025]), ((0, 15, 0), [0.0, 0.025])],
    ),
    # dense-route ladder of BASELINE config 5: n = 3600 real, n = 3600 complex, n = 1200
    "swave30_zeeman": dict(build=swave_square, kwargs=dict(L=30, zeeman=0.05), temps=[0.0, 0.1, 0.5], spectrum=True),
    "peierls30": dict(build=peierls_square, kwargs={}, temps=[0.0, 0.1, 0.5], spectrum=True),
    "chain300": dict(build=swave_chain, kwargs={}, temps=[0.0, 0.1, 0.5], spectrum=True),
    # next rung: n = 10^4 (the reference's dense eigh takes minutes per call on the build host: one temperature)
    "swave50_zeeman": dict(build=swave_square, kwargs=dict(L=50, zeeman=0.05), temps=[0.5], spectrum=True),
    # n = 14 400: about an hour through the reference on the build host
    "swave60_zeeman": dict(build=swave_square, kwargs=dict(L=60, zeeman=0.05), temps=[0.5], spectrum=True),
    "chain128": dict(
        build=field_chain,
        kwargs={},
        temps=[0.01],
        spectrum=True,
        triple=True,
        ldos=[((64, 0, 0), [0.0, 0.01])],
    ),
}
