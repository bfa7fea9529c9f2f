"""Shared names for the package: numeric aliases, Pauli algebra, argument checks.

API surface mirrors the reference's `bodge/common.py:12-61` (type aliases, π,
σ0..σ3, jσ0..jσ3 and their ASCII spellings) because those constants are part of
what user scripts star-import.  The reference delegates runtime type checking to
`beartype` (common.py:9); that package is not a dependency here, so `typecheck`
below is a small annotation-driven validator covering the argument kinds the
public methods actually take (Coord tuples, float, bool, str).
"""

from __future__ import annotations

import functools
import inspect
import numbers

import numpy as np
import numpy.typing as npt
import scipy.sparse as sp

# ---------------------------------------------------------------------------
# Type aliases (reference: common.py:12-25).
Index = int
Coord = tuple[int, int, int]
Indices = tuple[Index, Index]
Coords = tuple[Coord, Coord]

Matrix = npt.NDArray[np.float64] | npt.NDArray[np.complex128]
CooMatrix = sp.coo_matrix
DiaMatrix = sp.dia_matrix
BsrMatrix = sp.bsr_matrix
CsrMatrix = sp.csr_matrix
CscMatrix = sp.csc_matrix
SpMatrix = sp.spmatrix

# ---------------------------------------------------------------------------
# Constants (reference: common.py:27-61).
π = np.pi


def _pauli(entries) -> np.ndarray:
    m = np.array(entries, dtype=np.complex128)
    m.setflags(write=True)
    return m


σ0: Matrix = _pauli([[1, 0], [0, 1]])
σ1: Matrix = _pauli([[0, 1], [1, 0]])
σ2: Matrix = _pauli([[0, -1j], [1j, 0]])
σ3: Matrix = _pauli([[1, 0], [0, -1]])
σ = np.stack([σ1, σ2, σ3])

jσ0: Matrix = 1j * σ0
jσ1: Matrix = 1j * σ1
jσ2: Matrix = 1j * σ2
jσ3: Matrix = 1j * σ3
jσ = np.stack([jσ1, jσ2, jσ3])

pi = π
sigma0, sigma1, sigma2, sigma3, sigma = σ0, σ1, σ2, σ3, σ
jsigma0, jsigma1, jsigma2, jsigma3, jsigma = jσ0, jσ1, jσ2, jσ3, jσ


# ---------------------------------------------------------------------------
# Runtime argument validation.
class TypeCheckError(TypeError):
    """Raised when a public method receives an argument of the wrong kind."""


def _is_coord(value) -> bool:
    return (
        isinstance(value, tuple)
        and len(value) == 3
        and all(isinstance(c, numbers.Integral) and not isinstance(c, bool) for c in value)
    )


_CHECKS = {
    "Coord": _is_coord,
    "float": lambda v: isinstance(v, (float, np.floating)),
    "bool": lambda v: isinstance(v, (bool, np.bool_)),
    "str": lambda v: isinstance(v, str),
    "int": lambda v: isinstance(v, numbers.Integral) and not isinstance(v, bool),
    "int | None": lambda v: v is None
    or (isinstance(v, numbers.Integral) and not isinstance(v, bool)),
}


def typecheck(func):
    """Validate annotated arguments of `func` on every call.

    Only annotations whose *source text* appears in `_CHECKS` are enforced
    (the module uses postponed evaluation, so annotations are strings); anything
    else is passed through unchecked.
    """
    sig = inspect.signature(func)
    checked = [
        (name, _CHECKS[p.annotation])
        for name, p in sig.parameters.items()
        if isinstance(p.annotation, str) and p.annotation in _CHECKS
    ]
    if not checked:
        return func

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        bound = sig.bind(*args, **kwargs)
        for name, ok in checked:
            if name in bound.arguments and not ok(bound.arguments[name]):
                raise TypeCheckError(
                    f"{func.__qualname__}(): argument '{name}' has unsupported value "
                    f"{bound.arguments[name]!r}"
                )
        return func(*args, **kwargs)

    return wrapper
