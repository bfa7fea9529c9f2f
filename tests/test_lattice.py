"""Lattice geometry: mirrors the reference's tests/test_lattice.py with this package's API."""

import numpy as np
from pytest import raises

from bodge_amd import CubicLattice, Lattice


def test_abstract_base():
    with raises(ValueError):
        Lattice((1, 1, 1))

    class Bare(Lattice):
        pass

    lat = Bare((1, 2, 3))
    for call in (lambda: lat[(0, 0, 0)], lat.sites, lat.bonds, lat.edges):
        with raises(NotImplementedError):
            call()
    assert str(lat) == "Bare(1, 2, 3)"
    assert (lat.size, lat.dim) == (6, 2)


def test_sites_are_contiguous_and_bounded():
    lat = CubicLattice((3, 5, 7))
    seen = 0
    for expected, site in enumerate(lat.sites()):
        assert lat[site] == expected
        assert all(0 <= c < L for c, L in zip(site, lat.shape))
        assert lat.coord(expected) == site
        seen += 1
    assert seen == 3 * 5 * 7
    for bad in [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (3, 0, 0), (0, 5, 0), (0, 0, 7)]:
        with raises(ValueError):
            lat[bad]
    with raises(TypeError):
        lat[(0.5, 0, 0)]


def test_bonds_per_axis():
    lat = CubicLattice((2, 3, 5))
    total = 0
    for axis in range(3):
        for a, b in lat.bonds(axis=axis):
            delta = np.subtract(b, a)
            assert abs(delta[axis]) == 1 and np.count_nonzero(delta) == 1
            total += 1
    assert total == 2 * ((2 - 1) * 3 * 5 + 2 * (3 - 1) * 5 + 2 * 3 * (5 - 1))
    assert total == len(list(lat.bonds()))
    with raises(ValueError):
        for _ in lat.bonds(axis=3):
            pass


def test_edges_per_axis():
    lat = CubicLattice((2, 3, 5))
    total = 0
    for axis in range(3):
        for a, b in lat.edges(axis=axis):
            assert {a[axis], b[axis]} == {0, lat.shape[axis] - 1}
            total += 1
    assert total == 2 * (2 * 3 + 3 * 5 + 5 * 2)
    with raises(ValueError):
        for _ in lat.edges(axis=3):
            pass


def test_generator_order_matches_nested_loops():
    """Order is part of the contract: it fixes which dict entry wins and the COO order."""
    lat = CubicLattice((3, 2, 4))
    Lx, Ly, Lz = lat.shape
    expect = []
    for x in range(Lx):
        for y in range(Ly):
            for z in range(Lz - 1):
                expect += [((x, y, z), (x, y, z + 1)), ((x, y, z + 1), (x, y, z))]
    for x in range(Lx):
        for y in range(Ly - 1):
            for z in range(Lz):
                expect += [((x, y, z), (x, y + 1, z)), ((x, y + 1, z), (x, y, z))]
    for x in range(Lx - 1):
        for y in range(Ly):
            for z in range(Lz):
                expect += [((x, y, z), (x + 1, y, z)), ((x + 1, y, z), (x, y, z))]
    assert list(lat.bonds()) == expect
    pairs = lat.bond_array()
    assert [(lat[a], lat[b]) for a, b in expect] == [tuple(p) for p in pairs.tolist()]


def test_degenerate_axis_gives_self_edges():
    lat = CubicLattice((4, 4, 1))
    z_edges = list(lat.edges(axis=2))
    assert len(z_edges) == 2 * 16 and all(a == b for a, b in z_edges)
    assert list(lat.bonds(axis=2)) == []


def test_iteration_covers_sites_bonds_edges():
    lat = CubicLattice((2, 3, 2))
    items = list(lat)
    assert items[: lat.size] == [(s, s) for s in lat.sites()]
    assert len(items) == lat.size + len(list(lat.bonds())) + len(list(lat.edges()))
