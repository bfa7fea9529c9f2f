"""bodge_amd: MI355X-native solver behind the Bodge `Lattice` / `Hamiltonian` API.

`from bodge_amd import *` gives the same public names as the reference package
(`bodge/__init__.py:13-51`).  Assembly is host-side numpy; `free_energy`,
`diagonalize` and `ldos` run in a HIP shared library loaded through ctypes
(`bodge_amd.backend`) and raise if that library or a GPU is unavailable.
"""

from .common import (
    Coord,
    Coords,
    Index,
    Indices,
    jsigma,
    jsigma0,
    jsigma1,
    jsigma2,
    jsigma3,
    jσ,
    jσ0,
    jσ1,
    jσ2,
    jσ3,
    pi,
    sigma,
    sigma0,
    sigma1,
    sigma2,
    sigma3,
    π,
    σ,
    σ0,
    σ1,
    σ2,
    σ3,
)
from .hamiltonian import Hamiltonian, dwave, pwave, ssd, swave
from .lattice import CubicLattice, Lattice

__version__ = "0.1.0"
__all__ = [
    "Lattice", "CubicLattice", "Hamiltonian", "Coord", "Coords", "Index", "Indices",
    "ssd", "swave", "pwave", "dwave",
    "π", "σ", "σ0", "σ1", "σ2", "σ3", "jσ", "jσ0", "jσ1", "jσ2", "jσ3",
    "pi", "sigma", "sigma0", "sigma1", "sigma2", "sigma3",
    "jsigma", "jsigma0", "jsigma1", "jsigma2", "jsigma3",
]
