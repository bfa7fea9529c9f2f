"""Dense SciPy restatement of the reference observables (CPU, small lattices).

Each function follows the cited lines of `/root/reference/bodge/hamiltonian.py`
and takes plain arrays / scipy matrices so it has no dependency on the product
package.  TEST INFRASTRUCTURE ONLY - see `oracle/__init__.py`.
"""

from __future__ import annotations

import numpy as np
import scipy.linalg as la
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def assemble_bsr(n_sites, pair_list, hopping, pairing):
    """Loop-based BSR assembly.

    Follows hamiltonian.py:37-67 (COO skeleton of ones over every lattice pair
    and its transpose, converted to 4x4 BSR, zeroed) and :102-118 (H_ij goes to
    the electron block and -H_ij* to the hole block; Δ_ij to the upper right of
    block (i,j) and Δ_ij^† to the lower left of block (j,i)).

    pair_list : iterable of (i, j) site-index pairs (sites, bonds, edges)
    hopping / pairing : dict {(i, j): 2x2 complex}
    """
    rows, cols = [], []
    for i, j in pair_list:
        rows.append(4 * i)
        cols.append(4 * j)
        if i != j:
            rows.append(4 * j)
            cols.append(4 * i)
    ones = np.ones(len(rows), dtype=np.int8)
    shape = (4 * n_sites, 4 * n_sites)
    skeleton = sp.coo_matrix((ones, (rows, cols)), shape=shape).tobsr((4, 4))
    mat = sp.bsr_matrix(skeleton, dtype=np.complex128)
    mat.data[...] = 0

    def block(i, j):
        lo, hi = mat.indptr[i], mat.indptr[i + 1]
        return lo + int(np.where(mat.indices[lo:hi] == j)[0][0])

    for (i, j), val in hopping.items():
        k = block(i, j)
        mat.data[k, 0:2, 0:2] = val
        mat.data[k, 2:4, 2:4] = -np.conj(val)
    for (i, j), val in pairing.items():
        mat.data[block(i, j), 0:2, 2:4] = val
        mat.data[block(j, i), 2:4, 0:2] = np.conj(val).T
    return mat


def free_energy(h_dense, temperature):
    """hamiltonian.py:282-321: eigvalsh, keep ε > 0, F = -Σε/2 - T Σ log(1+e^{-ε/T})."""
    eps = la.eigvalsh(np.asarray(h_dense))
    eps = eps[eps > 0]
    internal = -0.5 * np.sum(eps)
    if temperature == 0:
        entropy = 0.0
    elif temperature > 0:
        entropy = np.sum(np.log(1 + np.exp(-eps / temperature)))
    else:
        raise ValueError("Expected non-negative temperature!")
    return internal - temperature * entropy


def diagonalize(h_dense, format="reshape"):
    """hamiltonian.py:203-251: eigh restricted to (0, inf), optional (k, N, 4) reshape."""
    vals, vecs = la.eigh(np.array(h_dense), subset_by_value=(0.0, np.inf), driver="evr")
    if format == "raw":
        return vals, vecs
    if format == "reshape":
        return vals, vecs.T.reshape((vals.size, -1, 4))
    raise RuntimeError(f"Eigenstate format '{format}' is not yet supported.")


def ldos_broadening(energies):
    """hamiltonian.py:346-352: ε = unique(|E|), Γ = gradient(ε)."""
    eps = np.unique(np.abs(np.array(energies, dtype=float)))
    return eps, np.gradient(eps)


def ldos(h_csc, site_index, energies):
    """hamiltonian.py:341-387: one sparse LU solve per distinct |ε| with η = local spacing."""
    h_csc = sp.csc_matrix(h_csc)
    n = h_csc.shape[0]
    identity = sp.identity(n, format="csc")
    energies = np.array(energies, dtype=float)
    eps, gam = ldos_broadening(energies)
    rows = np.array([4 * site_index + a for a in range(4)])
    rhs = sp.csc_matrix((np.ones(4), (rows, np.arange(4))), shape=(n, 4))
    rho = {}
    for e, g in zip(eps, gam):
        sol = spla.spsolve(sp.csc_matrix((e + 1j * g) * identity - h_csc), rhs)
        diag = np.asarray(sol.multiply(rhs).sum(axis=0)).ravel()
        rho[+e] = -np.imag(diag[0] + diag[1]) / np.pi
        rho[-e] = -np.imag(diag[2] + diag[3]) / np.pi
    return np.array([rho[e] for e in energies])
