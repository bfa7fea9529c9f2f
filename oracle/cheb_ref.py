"""CPU restatement of the Chebyshev (kernel-polynomial) path in numpy/scipy.sparse.

TEST INFRASTRUCTURE ONLY - see `oracle/__init__.py`.  The reference has no
Chebyshev code; what it fixes is the *definition* of the observables
(hamiltonian.py:305-321 for F, :349-382 for the LDOS).  This file restates, on
the CPU, exactly the algorithm the HIP library runs, so that device results can
be compared moment by moment on identical start vectors:

  t_0 = v,  t_1 = (1/a) H t_0,  t_{n+1} = (2/a) H t_n - t_{n-1}
  d_n = <t_n|t_n>,  e_n = Re <t_{n+1}|t_n>
  μ_0 = d_0, μ_1 = e_0, μ_{2n} = 2 d_n - μ_0, μ_{2n+1} = 2 e_n - μ_1
  F  = Σ_m c_m μ_m,   c_m = Chebyshev-Gauss coefficients of f(aε~),
       f(ε) = -(T/2) ln(2 cosh(ε/2T))  (sum of f over all 4N eigenvalues equals
       the reference's -Σ_{ε>0} ε/2 - T Σ_{ε>0} log(1+e^{-ε/T}) by ± symmetry)

and is itself pinned by reproducing the reference's dense results on small
lattices (tests/test_oracle.py).
"""

from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp

MASK64 = (1 << 64) - 1
VEC_RADEMACHER = 0  # entries ±1
VEC_Z4 = 1  # entries in {1, i, -1, -i}


# ---------------------------------------------------------------- random vectors
def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def vector_key(seed: int, vec_id: int) -> np.uint64:
    inner = _splitmix64(np.array([vec_id & MASK64], dtype=np.uint64))[0]
    return _splitmix64(np.array([(seed & MASK64) ^ int(inner)], dtype=np.uint64))[0]


def random_vector(n: int, seed: int, vec_id: int, kind: int = VEC_RADEMACHER, row0: int = 0) -> np.ndarray:
    """Counter-based start vector: element `row` = 4 * site + component depends only on (seed, vec_id, row) - one hash
    per site, its top bits shared out to the four components (kernels.hpp `start_entry`)."""
    key = vector_key(seed, vec_id)
    rows = np.arange(row0, row0 + n, dtype=np.uint64)
    component = rows & np.uint64(3)
    with np.errstate(over="ignore"):
        h = _splitmix64(key + (rows >> np.uint64(2)))
    if kind == VEC_RADEMACHER:
        return np.where((h >> (np.uint64(63) - component)) & np.uint64(1) == 0, 1.0, -1.0).astype(np.complex128)
    if kind == VEC_Z4:
        return np.array([1, 1j, -1, -1j], dtype=np.complex128)[((h >> (np.uint64(62) - np.uint64(2) * component)) & np.uint64(3)).astype(np.int64)]
    raise ValueError("unknown vector kind")


def random_block(n: int, seed: int, vec_ids, kind: int = VEC_RADEMACHER) -> np.ndarray:
    return np.stack([random_vector(n, seed, v, kind) for v in vec_ids], axis=1)


def unit_block(n: int, rows) -> np.ndarray:
    out = np.zeros((n, len(rows)), dtype=np.complex128)
    out[np.asarray(rows), np.arange(len(rows))] = 1.0
    return out


# ------------------------------------------------------------------ spectral bound
def spectral_bound(bsr: sp.bsr_matrix, pad: float = 1.01) -> float:
    """Gershgorin radius over scalar rows, padded so the spectrum is strictly inside."""
    absrow = np.abs(bsr.data).sum(axis=2)  # (nnzb, 4): Σ_β |a_{αβ}| per block
    nb = bsr.shape[0] // 4
    block_rows = np.repeat(np.arange(nb), np.diff(bsr.indptr))
    sums = np.zeros((nb, 4))
    np.add.at(sums, block_rows, absrow)
    return float(pad * sums.max())


# ------------------------------------------------------------------------ moments
def _column_dots(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Re <a_r|b_r> per column with pairwise (not sequential) summation: at 4·10^6 rows a
    running sum loses ~1e-11 relative, which is the size of the tolerances used here."""
    return np.sum(np.ascontiguousarray((a.conj() * b).real.T), axis=1)


def recurrence_dots(bsr, scale: float, n_moments: int, start: np.ndarray):
    """Run M/2 recurrence steps on the columns of `start`; return (d, e) of shape (M/2, R)."""
    if n_moments % 2 or n_moments < 2:
        raise ValueError("number of moments must be even and >= 2")
    steps = n_moments // 2
    t_prev = np.array(start, dtype=np.complex128, order="C")
    t_cur = (bsr @ t_prev) * (1.0 / scale)
    d = np.empty((steps, t_prev.shape[1]))
    e = np.empty_like(d)
    d[0] = _column_dots(t_prev, t_prev)
    e[0] = _column_dots(t_cur, t_prev)
    for n in range(1, steps):
        t_next = (bsr @ t_cur) * (2.0 / scale) - t_prev
        d[n] = _column_dots(t_cur, t_cur)
        e[n] = _column_dots(t_next, t_cur)
        t_prev, t_cur = t_cur, t_next
    return d, e


def dots_to_moments(d: np.ndarray, e: np.ndarray) -> np.ndarray:
    """(M/2, R) dot products -> (M, R) Chebyshev moments via the doubling identities."""
    steps = d.shape[0]
    mu = np.empty((2 * steps,) + d.shape[1:])
    mu[0::2] = 2 * d - d[0]
    mu[1::2] = 2 * e - e[0]
    mu[0] = d[0]
    mu[1] = e[0]
    return mu


def moments(bsr, scale, n_moments, start):
    return dots_to_moments(*recurrence_dots(bsr, scale, n_moments, start))


def trace_moments_exact(bsr, scale, n_moments, chunk: int = 256) -> np.ndarray:
    """Tr T_m(H/a) for m < M using every unit vector as a start vector."""
    n = bsr.shape[0]
    total = np.zeros(n_moments)
    for lo in range(0, n, chunk):
        rows = np.arange(lo, min(n, lo + chunk))
        total += moments(bsr, scale, n_moments, unit_block(n, rows)).sum(axis=1)
    return total


def trace_moments_stochastic(bsr, scale, n_moments, n_vectors, seed=0, kind=VEC_RADEMACHER, first_id=0):
    """(1/R) Σ_r <v_r|T_m|v_r> with counter-based vectors first_id .. first_id+R-1."""
    start = random_block(bsr.shape[0], seed, range(first_id, first_id + n_vectors), kind)
    return moments(bsr, scale, n_moments, start).sum(axis=1) / n_vectors


# ------------------------------------------------------------------- free energy
def free_energy_density(eps: np.ndarray, temperature: float) -> np.ndarray:
    """f(ε) with Σ_{all 4N ε} f(ε) = F of hamiltonian.py:305-321."""
    mag = np.abs(eps)
    if temperature == 0:
        return -mag / 4
    if temperature < 0:
        raise ValueError("Expected non-negative temperature!")
    return -mag / 4 - (temperature / 2) * np.log1p(np.exp(-mag / temperature))


def chebyshev_coefficients(func, n_moments: int, oversample: int = 4) -> np.ndarray:
    """c_m (m < M) of func on [-1, 1] by Chebyshev-Gauss quadrature with oversample*M nodes."""
    nodes = oversample * n_moments
    theta = np.pi * (np.arange(nodes) + 0.5) / nodes
    values = func(np.cos(theta))
    m = np.arange(n_moments)[:, None]
    coeff = (2.0 / nodes) * (np.cos(m * theta[None, :]) @ values)
    coeff[0] *= 0.5
    return coeff


def free_energy_from_moments(mu_trace: np.ndarray, scale: float, temperature: float) -> float:
    coeff = chebyshev_coefficients(
        lambda x: free_energy_density(scale * x, temperature), len(mu_trace)
    )
    return float(np.dot(coeff, mu_trace))


def free_energy_exact_trace(bsr, temperature, n_moments, scale=None) -> float:
    scale = spectral_bound(bsr) if scale is None else scale
    return free_energy_from_moments(trace_moments_exact(bsr, scale, n_moments), scale, temperature)


def free_energy_stochastic(bsr, temperature, n_moments, n_vectors, seed=0, kind=VEC_RADEMACHER, scale=None):
    scale = spectral_bound(bsr) if scale is None else scale
    mu = trace_moments_stochastic(bsr, scale, n_moments, n_vectors, seed, kind)
    return free_energy_from_moments(mu, scale, temperature)


# -------------------------------------------------------------------------- LDOS
def resolvent_diagonal(mu: np.ndarray, scale: float, z: complex) -> complex:
    """<e|(z - H)^{-1}|e> from the moments μ_n = <e|T_n(H/a)|e>, Im z > 0."""
    zt = complex(z) / scale
    phase = np.arccos(zt)
    n = np.arange(len(mu))
    weights = np.exp(-1j * n * phase) * np.where(n == 0, 1.0, 2.0)
    return complex((-1j / np.sqrt(1 - zt * zt)) * np.dot(weights, mu) / scale)


def ldos_moment_count(scale: float, gamma_min: float, digits: float = 10.0) -> int:
    """Even M such that exp(-M Γ/a) ~ 10^-digits."""
    m = int(np.ceil(digits * np.log(10.0) * scale / gamma_min))
    return max(64, m + (m % 2))


def ldos(bsr, site_index: int, energies, n_moments=None, scale=None) -> np.ndarray:
    """Same quantity as hamiltonian.py:341-387, via the Chebyshev resolvent series."""
    energies = np.array(energies, dtype=float)
    eps = np.unique(np.abs(energies))
    gam = np.gradient(eps)
    scale = spectral_bound(bsr) if scale is None else scale
    if n_moments is None:
        n_moments = ldos_moment_count(scale, float(np.min(gam)))
    rows = [4 * site_index + a for a in range(4)]
    mu = moments(bsr, scale, n_moments, unit_block(bsr.shape[0], rows))  # (M, 4)
    rho = {}
    for e, g in zip(eps, gam):
        diag = [resolvent_diagonal(mu[:, a], scale, e + 1j * g) for a in range(4)]
        rho[+e] = -np.imag(diag[0] + diag[1]) / np.pi
        rho[-e] = -np.imag(diag[2] + diag[3]) / np.pi
    return np.array([rho[e] for e in energies])


# ------------------------------------------------------------------ CPU baseline
def _recurrence_loop(mat, scale, t_prev, seconds, warmup):
    t_cur = (mat @ t_prev) * (1.0 / scale)
    done, t0, acc = 0, None, 0.0
    while True:
        if done == warmup:
            t0 = time.perf_counter()
        t_next = (mat @ t_cur) * (2.0 / scale) - t_prev
        acc += np.einsum("ir,ir->", t_cur.conj(), t_cur).real
        acc += np.einsum("ir,ir->", t_next.conj(), t_cur).real
        t_prev, t_cur = t_cur, t_next
        done += 1
        if t0 is not None and time.perf_counter() - t0 >= seconds:
            break
    return done - warmup, time.perf_counter() - t0


def time_recurrence(bsr, scale, n_vectors, seconds=10.0, seed=0, kind=VEC_RADEMACHER, warmup=2,
                    fmt="bsr", real=False):
    """Wall-clock the recurrence (SpMV + axpy + two dots) on this host, single thread.

    fmt  "bsr" (the reference's storage) or "csr" (skips in-block zeros, ~3x faster on a CPU)
    real drop the (all-zero) imaginary parts, as the GPU's real-arithmetic mode does
    Returns (vector_steps_per_second, block_steps_done).  Used only by bench.py's cpu_baseline.
    """
    mat = bsr if fmt == "bsr" else bsr.tocsr()
    start = random_block(bsr.shape[0], seed, range(n_vectors), kind)
    if real:
        mat, start = mat.real.copy(), np.ascontiguousarray(start.real)
        if fmt == "bsr":
            mat = sp.bsr_matrix(mat, blocksize=(4, 4))
    timed, elapsed = _recurrence_loop(mat, scale, start, seconds, warmup)
    return timed * n_vectors / elapsed, timed


_shared_matrix = None  # set before the pool forks: the workers inherit it (copy-on-write), nothing is pickled


def _worker(args):
    scale, n_vectors, seconds, seed = args
    mat = _shared_matrix
    start = random_block(mat.shape[0], seed, range(n_vectors), VEC_RADEMACHER)
    timed, elapsed = _recurrence_loop(mat, scale, start, seconds, 1)
    return timed * n_vectors / elapsed


def time_recurrence_processes(bsr, scale, n_vectors, n_processes, seconds=8.0):
    """Whole-host throughput: `n_processes` forked workers, each advancing its own vectors on
    the shared (copy-on-write) matrix.  Returns summed vector-steps per second."""
    import multiprocessing as mp

    global _shared_matrix
    _shared_matrix = bsr
    try:
        ctx = mp.get_context("fork")
        with ctx.Pool(n_processes) as pool:
            rates = pool.map(_worker, [(scale, n_vectors, seconds, 100 + p) for p in range(n_processes)], chunksize=1)
    finally:
        _shared_matrix = None
    return float(sum(rates))
