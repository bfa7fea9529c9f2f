"""Lattice geometry: site numbering and neighbour enumeration.

Behavioural contract (what has to agree with the reference so that the BSR
block-row order, and therefore every device buffer, is the same):

* abstract `Lattice` with `shape/size/dim`, `lattice[coord]`, iteration =
  on-site pairs, then bonds, then edges          (reference lattice.py:23-50)
* `CubicLattice.index`: z fastest, x slowest      (reference lattice.py:101-108)
* `bonds(axis)`: both directions of each nearest-neighbour pair; axis order
  2, 1, 0 when no axis is given                   (reference lattice.py:121-159)
* `edges(axis)`: opposite-face pairs, same order  (reference lattice.py:161-197)

Unlike the reference, which only has Python generators, the cubic lattice here
is array-first: `site_array`, `bond_array`, `edge_array` return whole index
tables built with numpy, and the generators are thin views over those tables.
The array forms feed the vectorised Hamiltonian assembly so a
10^6-site system is laid out in well under a second.
"""

from __future__ import annotations

from typing import Iterator

import numpy as np

from .common import Coord, Coords, Index, typecheck


class Lattice:
    """Abstract graph of sites (nodes), bonds (links) and periodic edges."""

    @typecheck
    def __init__(self, shape: Coord):
        if type(self).__name__ == "Lattice":
            raise ValueError("Lattice is an abstract base; instantiate a subclass.")
        self.shape: Coord = shape
        self.size: Index = int(np.prod(shape))
        self.dim: int = sum(1 for extent in shape if extent > 1)

    @typecheck
    def __getitem__(self, coord: Coord) -> Index:
        return self.index(coord)

    def __iter__(self) -> Iterator[Coords]:
        for site in self.sites():
            yield (site, site)
        yield from self.bonds()
        yield from self.edges()

    def __repr__(self) -> str:
        return f"{type(self).__name__}{self.shape}"

    # -- to be provided by concrete lattices ---------------------------------
    def index(self, coord: Coord) -> Index:
        raise NotImplementedError

    def sites(self):
        raise NotImplementedError

    def bonds(self):
        raise NotImplementedError

    def edges(self):
        raise NotImplementedError


def _check_axis(axis) -> None:
    if axis is not None and axis not in (0, 1, 2):
        raise ValueError("No such axis")


class CubicLattice(Lattice):
    """Primitive cubic (or rectangular / chain, with unit extents) lattice."""

    # ------------------------------------------------------------------ index
    @typecheck
    def index(self, coord: Coord) -> Index:
        x, y, z = coord
        Lx, Ly, Lz = self.shape
        if not (0 <= x < Lx and 0 <= y < Ly and 0 <= z < Lz):
            raise ValueError(f"Coordinate {coord} out of bounds")
        return int(z + Lz * (y + Ly * x))

    def coord(self, index: int) -> Coord:
        """Inverse of `index` (not in the reference; used by host-side tiling)."""
        _, Ly, Lz = self.shape
        if not 0 <= index < self.size:
            raise ValueError(f"Index {index} out of bounds")
        return (int(index // (Ly * Lz)), int((index // Lz) % Ly), int(index % Lz))

    # ------------------------------------------------------------ array forms
    def site_array(self) -> np.ndarray:
        """(N, 3) int64 coordinates, row k being the site whose index is k."""
        Lx, Ly, Lz = self.shape
        grid = np.indices((Lx, Ly, Lz), dtype=np.int64)
        return grid.reshape(3, -1).T.copy()

    def _pair_table(self, axis: int, wrap: bool) -> np.ndarray:
        """(P, 2, 3) coordinates of unordered partner pairs along `axis`.

        wrap=False: nearest neighbours (c, c + e_axis).
        wrap=True : opposite faces      (c with c_axis = 0, c with c_axis = L-1).
        Rows come in the nesting order x, y, z (slowest to fastest), which is
        the order the generators must reproduce.
        """
        extents = list(self.shape)
        extents[axis] = 1 if wrap else self.shape[axis] - 1
        if min(extents) <= 0:
            return np.zeros((0, 2, 3), dtype=np.int64)
        first = np.indices(tuple(extents), dtype=np.int64).reshape(3, -1).T
        second = first.copy()
        second[:, axis] += (self.shape[axis] - 1) if wrap else 1
        return np.stack([first, second], axis=1)

    def _directed(self, axis, wrap: bool) -> np.ndarray:
        """(2P, 2, 3): each pair followed immediately by its reverse."""
        _check_axis(axis)
        axes = (2, 1, 0) if axis is None else (axis,)
        chunks = []
        for ax in axes:
            pairs = self._pair_table(ax, wrap)
            both = np.empty((2 * len(pairs), 2, 3), dtype=np.int64)
            both[0::2] = pairs
            both[1::2] = pairs[:, ::-1]
            chunks.append(both)
        return np.concatenate(chunks, axis=0)

    def _directed_indices(self, axis, wrap: bool) -> np.ndarray:
        """(2P, 2) site indices of `_directed(axis, wrap)`, by index arithmetic (no coordinate tables)."""
        _check_axis(axis)
        Lx, Ly, Lz = self.shape
        stride = (Ly * Lz, Lz, 1)
        chunks = []
        for ax in (2, 1, 0) if axis is None else (axis,):
            extents = list(self.shape)
            extents[ax] = 1 if wrap else self.shape[ax] - 1
            if min(extents) <= 0:
                continue
            x, y, z = (np.arange(e, dtype=np.int64) for e in extents)
            first = ((x[:, None, None] * Ly + y[None, :, None]) * Lz + z[None, None, :]).reshape(-1)
            second = first + ((self.shape[ax] - 1) if wrap else 1) * stride[ax]
            both = np.empty((2 * len(first), 2), dtype=np.int64)
            both[0::2, 0], both[0::2, 1] = first, second
            both[1::2, 0], both[1::2, 1] = second, first
            chunks.append(both)
        return np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 2), dtype=np.int64)

    def _flatten(self, coords: np.ndarray) -> np.ndarray:
        _, Ly, Lz = self.shape
        return coords[..., 2] + Lz * (coords[..., 1] + Ly * coords[..., 0])

    def bond_array(self, axis=None, coords: bool = False) -> np.ndarray:
        """Directed nearest-neighbour pairs in generator order.

        Returns (B, 2) site indices, or (B, 2, 3) coordinates when coords=True.
        """
        return self._directed(axis, wrap=False) if coords else self._directed_indices(axis, wrap=False)

    def edge_array(self, axis=None, coords: bool = False) -> np.ndarray:
        """Directed opposite-face pairs in generator order (see `bond_array`)."""
        return self._directed(axis, wrap=True) if coords else self._directed_indices(axis, wrap=True)

    # -------------------------------------------------------------- generators
    def sites(self) -> Iterator[Coord]:
        Lx, Ly, Lz = self.shape
        return ((x, y, z) for x in range(Lx) for y in range(Ly) for z in range(Lz))

    @staticmethod
    def _walk(table: np.ndarray) -> Iterator[Coords]:
        for (a, b) in table.tolist():
            yield tuple(a), tuple(b)

    def bonds(self, axis=None) -> Iterator[Coords]:
        return self._walk(self._directed(axis, wrap=False))

    def edges(self, axis=None) -> Iterator[Coords]:
        return self._walk(self._directed(axis, wrap=True))
