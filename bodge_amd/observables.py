"""`diagonalize`, `free_energy`, `ldos` on the GPU, behind the reference signatures.

Routing (all device-side; there is no CPU path):

* dense    BSR -> dense scatter kernel + Hermitian eigensolver on the GPU (own
           one-sided Jacobi kernels up to 4N = 2048, rocSOLVER `dsyevd` / `zheevd` above).
           Exact, O((4N)^3); what the reference's `cuda=True` branch does with
           CuPy (ref hamiltonian.py:206-221, :287-295).
* chebyshev  kernel-polynomial expansion on the BSR matrix, O(N·M): the
           recurrence kernel advances R start vectors together; exact trace
           (unit vectors of the electron rows, x2 by particle-hole symmetry) up to
           4N = EXACT_TRACE_LIMIT, stochastic trace beyond.  T = 0 on a gapped spectrum expands
           the smoothed density -(ε/4)·erf(5ε/gap).

`method="auto"` picks, by estimated run time, between the dense eigensolver
(4N <= DENSE_AUTO_LIMIT: the library-free Jacobi kernels; rocSOLVER up to
DENSE_AUTO_LIMIT_T0) and the Chebyshev expansion (`_auto_method`).
"""

from __future__ import annotations

import os

import numpy as np

from . import chebyshev as cheb
from .backend import VEC_RADEMACHER, VEC_Z4

DENSE_AUTO_LIMIT = 2048  # largest 4N served by the own Jacobi kernels (kJacobiLimit in the library)
DENSE_AUTO_LIMIT_T0 = 16384  # largest 4N method="auto" ever sends to a dense eigensolver
DENSE_VALUES_LIMIT_T0 = 8192  # T = 0: up to here the eigenvalues-only dense route (tridiagonalisation, ~1 s) is simply taken
GAP_SURROGATE_RATIO = 5.0  # T = 0: the kink of -|ε|/4 is smoothed over gap / this (see free_energy)
EXACT_TRACE_LIMIT = 65536  # largest 4N for which trace="auto" is exact (128x128 sites: ~2 s at T = 0.1)


def _warn(message: str) -> None:
    """RuntimeWarning for results that are not what the reference's call would have computed."""
    import warnings

    warnings.warn(message, RuntimeWarning, stacklevel=4)


def _scale_of(system, pad: float = 1.01) -> float:
    # zero blocks add nothing to the row sums: the skeleton arrays give the same bound as the
    # trimmed ones, without the copy; cached until the next `with` block
    bound = system.gershgorin_bound()  # (= cheb.spectral_bound(indptr, data, pad=1.0), same summation order)
    return pad * bound if bound > 0 else 1.0


def _electron_rows(dim: int) -> np.ndarray:
    """Scalar rows of the electron components (e↑, e↓) of every site.

    The Chebyshev route requires H = -τx H* τx (checked by the caller).  Then T_m(H) = ±τx T_m(H)* τx
    and only even m enter an even f, so the diagonal entry of f(H) on a hole row equals the one on
    the electron row of the same site and spin: the exact trace is twice the sum over these rows.
    """
    rows = np.arange(dim, dtype=np.int64)
    return rows[(rows & 3) < 2]


CONCURRENT_TRACE_MAX_SITES = 200_000  # below this a launch leaves the GPU part idle: run two at a time


def _unit_moments(system, scale: float, moments: int, rows: np.ndarray) -> np.ndarray:
    """Unit-vector moments <e_r|T_m(H/scale)|e_r> for every r in `rows`, shape (moments, len(rows)).

    Small lattices are launch-latency bound (one launch ≈ 7-10 µs whatever it holds), so two
    device mirrors, each with its own stream and buffers, are driven from two host threads (the
    library is thread safe per handle, ctypes releases the GIL): 1.5-1.9x at 32x32 … 100x100.
    """
    lanes = 2 if (len(rows) >= 128 and system.lattice.size <= CONCURRENT_TRACE_MAX_SITES) else 1
    if lanes == 1:
        return system._solver().moments_unit(scale, moments, rows)
    from concurrent.futures import ThreadPoolExecutor

    solvers = [system._solver(lane) for lane in range(lanes)]  # created here, on one thread
    half = (len(rows) // (64 * lanes)) * 64 or len(rows) // lanes
    parts = [rows[:half], rows[half:]]
    with ThreadPoolExecutor(lanes) as pool:
        blocks = pool.map(lambda job: job[0].moments_unit(scale, moments, job[1]), zip(solvers, parts))
        return np.concatenate(list(blocks), axis=1)


def _unit_moment_sum(system, scale: float, moments: int, rows: np.ndarray) -> np.ndarray:
    """Σ over `rows` of the unit-vector moments, shape (moments,).  Rows are fed in slices so that
    the (moments x rows) table never exceeds ~128 MB on the host (low temperatures: 10^4-10^5 moments)."""
    step = max(64, ((1 << 24) // max(1, moments)) // 64 * 64)
    total = np.zeros(moments)
    for lo in range(0, len(rows), 2 * step):  # two device mirrors take `step` rows each
        total += _unit_moments(system, scale, moments, rows[lo : lo + 2 * step]).sum(axis=1)
    return total


def _gap_estimate(system) -> float:
    """Smallest positive eigenvalue from a short Lanczos run (an upper bound, good to ~1e-3);
    cached until the matrix changes."""
    return system._memoized("gap_estimate", lambda: float(lowest_eigenvalues(system, 1, tol=1e-3)[0]))


def _auto_method(system, temperature: float, moments, scale) -> str:
    """Dense or Chebyshev for `free_energy(method="auto")`, by estimated run time.

    The dense route needs eigenvalues only: the library's own Householder tridiagonalisation +
    bisection (csrc/tridiag.hpp; no rocSOLVER), 8 ms at 4N = 512, 0.09 s at 3600, i.e. ≈ 2.0e-12·(4N)³ s above a floor of
    ≈ 18 µs per row (two launches); from 6000 rows on real matrices go through a band (csrc/twostage.hpp: 0.36 s at 10^4,
    7.8 s at 4·10^4), complex ones stay one-stage (1.1e-12·(4N)³ s); the Jacobi kernels below 4N = 512.  T = 0: dense up to 4N = 8192; above, the Chebyshev expansion of the smoothed
    density -(ε/4)·erf(5ε/gap) (see `free_energy`) when the spectrum is gapped enough, else dense.
    Matrices without the particle-hole form must go dense.  Otherwise the two routes are priced:
    an exact-trace Chebyshev run costs M/2 launches per batch of 64 unit vectors, each ≥ 7 µs or its
    HBM time.  Both routes meet the 1e-10 relative accuracy the tests ask for.
    """
    dim = system.shape[0]
    if dim > DENSE_AUTO_LIMIT_T0:
        return "chebyshev"
    if not system.has_symmetric_spectrum(1e-12):
        return "dense"
    a = _scale_of(system) if scale is None else float(scale)
    if temperature == 0:
        # beyond the own Jacobi kernels: Chebyshev on the smoothed density if the spectrum is gapped enough
        # for that expansion to converge within the moment cap, the dense library otherwise
        if dim <= DENSE_VALUES_LIMIT_T0 or moments is not None:
            return "dense"
        gap = _gap_estimate(system)
        gapped = gap > 0 and cheb.moments_for_gapped_ground_state(a, gap / GAP_SURROGATE_RATIO) <= cheb.MAX_MOMENTS
        return "chebyshev" if gapped else "dense"
    m = cheb.moments_for_free_energy(a, temperature) if moments is None else int(moments)
    batches = -(-(dim // 2) // 64) if dim <= EXACT_TRACE_LIMIT else 1  # electron rows only
    launch = max(7e-6, (dim // 4) * 64 * 192 / 5e12)
    chebyshev_seconds = 0.5 * m * batches * launch
    # (tridiagonalisation + bisection: ~18 us per row, 2.0e-12 dim^3 s below 6000 rows; from there on real matrices take the
    # two-stage route - band by MFMA panels, bulge chasing: 0.36 s at 10^4, 7.8 s at 4e4 - complex ones 1.1e-12 dim^3)
    if dim <= 512:
        dense_seconds = 6e-11 * dim**3
    elif dim < 6000:
        dense_seconds = max(1.8e-5 * dim, 2.0e-12 * dim**3)
    elif system._memoized("imag_free", lambda: not bool(system._data.imag.any())):
        dense_seconds = 1.5e-5 * dim + 1.2e-13 * dim**3 + 1.0e-9 * dim**2
    else:
        dense_seconds = 1.1e-12 * dim**3
    return "dense" if dense_seconds <= chebyshev_seconds else "chebyshev"


# --------------------------------------------------------------------------- F
def free_energy(
    system,
    temperature: float = 0.0,
    *,
    method: str = "auto",
    moments: int | None = None,
    vectors: int | None = None,
    seed: int = 0,
    vector_kind: str = "rademacher",
    trace: str = "auto",
    scale: float | None = None,
    gap_surrogate: bool = True,
    damping: bool = False,
    comm=None,
    decomposition: str = "vectors",
    devices=None,
) -> float:
    """Free energy of `system` at `temperature` (reference hamiltonian.py:254-321).

    method   "auto" | "dense" | "chebyshev"
    moments  Chebyshev order M (even); default from the analyticity strip of f at T
    trace    "exact" (all 4N unit vectors), "stochastic", or "auto"
    vectors  number of random vectors for the stochastic trace (default 64)
    gap_surrogate  at T = 0 with the Chebyshev method and no explicit `moments`: expand
             -(ε/4)·erf(ε/δ), δ = gap/5 (gap from `lowest_eigenvalues`), which differs from -|ε|/4 by
             1e-13 per level on a gapped spectrum and converges faster than geometrically (10·a/δ moments);
             False keeps the plain T = 0 coefficients
    devices  optional list of GPU ordinals of THIS process, e.g. `devices=[0, 1, 2, 3]` or
             `range(8)`: H is replicated on each, the start vectors (or the unit vectors of an exact
             trace) are shared out over them, one host thread drives each GPU, and the moments are
             summed on the host in device order - the single-process way to use a whole node from the
             reference's call site `system.free_energy(T)` (no launcher, no communicator).  An ordinal
             may repeat (`[0, 0]`: two mirrors on one GPU).  Not combined with `comm`.
    comm     optional `Communicator` (one process per GPU)
    decomposition  how the ranks of `comm` share the work:
             "vectors" - H replicated, each rank owns a contiguous share of the start
                         vectors, one all-reduce of the moments (default);
             "slab"    - each rank owns a slab of lattice planes, every rank advances all
                         vectors on its rows, halo rows exchanged per step (RCCL send/recv),
                         one all-reduce of the moments.  For matrices too large to replicate.
    """
    if temperature < 0:
        raise ValueError("Expected non-negative temperature!")
    dim = system.shape[0]
    if method == "auto":
        method = _auto_method(system, temperature, moments, scale)
    if method == "chebyshev" and not system.has_symmetric_spectrum(1e-12):
        # Σ_{ε>0} g(ε) equals a trace of a smooth function only for a ±-symmetric spectrum
        raise RuntimeError(
            "The Chebyshev free energy needs a particle-hole symmetric Hamiltonian "
            "(pairing terms with Δ_ij = -Δ_ji^T); use method='dense' for this matrix."
        )

    if devices is not None:
        devices = [int(d) for d in devices]
        if comm is not None:
            raise ValueError("free_energy: pass either devices=[...] (one process) or comm= (one process per GPU)")
        if not devices:
            raise ValueError("free_energy: devices=[] names no GPU")

    if method == "dense":
        eps, _ = system._solver(device=None if devices is None else devices[0]).eigh(vectors=False)
        if system.has_symmetric_spectrum(1e-12):
            # ±-symmetric spectrum: the reference's "ε > 0" (ref :305) keeps one member of every pair, and of a
            # pair of zero modes whichever round-off made positive - the upper half of the sorted spectrum is the
            # same set without depending on the sign of 1e-17 (each such pair then contributes T ln 2, as there).
            # Deliberate deviation: a pair of EXACT zeros (a decoupled site, a zero block) counts once
            # here as well - this is Tr f(H), what the Chebyshev route gives - where the reference's strict "> 0"
            # drops both members; F then differs by T ln 2 per exact pair (test_exact_zero_modes_count_once_per_pair)
            eps = np.maximum(eps[len(eps) // 2:], 0.0)
        else:
            eps = eps[eps > 0]
        internal = -0.5 * np.sum(eps)
        entropy = 0.0 if temperature == 0 else np.sum(np.log1p(np.exp(-eps / temperature)))
        return float(internal - temperature * entropy)

    if method != "chebyshev":
        raise RuntimeError(f"Free-energy method '{method}' is not supported")

    if decomposition not in ("vectors", "slab"):
        raise RuntimeError(f"Decomposition '{decomposition}' is not supported")
    if decomposition == "slab" and comm is None:
        raise ValueError("decomposition='slab' shares the lattice planes out over the ranks of a communicator: pass comm=")
    scale = _scale_of(system) if scale is None else float(scale)
    series_temperature = temperature
    density = None  # (None: f_T at `temperature`)
    if temperature == 0 and moments is None and gap_surrogate and decomposition != "slab":
        # f_0(ε) = -|ε|/4 has a kink at ε = 0, but a gapped spectrum never samples it: expand
        # -(ε/4)·erf(ε/δ) with δ = gap/5 instead, which differs from f_0 by ≤ (gap/4)·erfc(5) = 4e-13·gap
        # per level on |ε| ≥ gap and is entire - 10·a/δ moments are within 1e-13 per level (20x20 README model:
        # 1e-15 of the dense value at 6144 moments, where f_T at T = gap/20, the surrogate of round 1,
        # still has 2e-10 and needs 22 000 for 1e-11; cheb.gapped_ground_state_density).  The Lanczos
        # estimate approaches the gap from above, to 1e-3.  A gapless spectrum keeps the plain T = 0 series.
        gap = _gap_estimate(system)
        if gap > 1e-9 * scale:
            width = gap / GAP_SURROGATE_RATIO
            density = cheb.gapped_ground_state_density(width)
            moments = cheb.moments_for_gapped_ground_state(scale, width)
            if moments > cheb.MAX_MOMENTS:
                _warn(f"free_energy(0.0): gap/scale = {gap / scale:.1e} asks for {moments} Chebyshev moments; capped at "
                      f"{cheb.MAX_MOMENTS} (pass moments=... to override, or method='dense')")
                moments = cheb.MAX_MOMENTS
            _warn(f"free_energy(0.0) on a {dim}x{dim} matrix is beyond the dense eigensolver: evaluated by the Chebyshev "
                  f"expansion of -(ε/4)·erf(ε/δ), δ = gap/{GAP_SURROGATE_RATIO:g} = {width:.3g} "
                  "(within 1e-13 per level of -|ε|/4 on a gapped spectrum; method='dense' forces the reference's algorithm)")
    if moments is None:
        moments = cheb.moments_for_free_energy(scale, series_temperature)
    moments += moments & 1
    chosen_for_the_caller = trace == "auto"
    if trace == "auto":
        trace = "exact" if dim <= EXACT_TRACE_LIMIT else "stochastic"

    if decomposition == "slab" and comm is not None:
        solver = _slab_solver(system, comm)
        steps = moments // 2
        if temperature == 0 and series_temperature == 0:
            _warn("free_energy(0.0, decomposition='slab') uses the plain T = 0 series (|ε| is not analytic at 0: "
                  "~1e-8 at 4096 moments): the gap-sized surrogate temperature needs the whole matrix on one GPU; "
                  "pass moments= or a small positive temperature for more digits")
        if trace == "exact":
            # rows in slices, as _unit_moment_sum does: the (steps x rows) tables stay within ~128 MB
            rows = _electron_rows(dim)
            width = max(64, ((1 << 23) // max(1, steps)) // 64 * 64)
            mu_local = np.zeros(moments)
            for lo in range(0, len(rows), width):
                d, e = solver.dots_unit(scale, steps, rows[lo : lo + width])
                mu_local += cheb.dots_to_moments(d, e).sum(axis=1)
            mu = comm.allreduce_sum(mu_local) / 0.5  # the hole rows contribute the same again (see _electron_rows)
            return cheb.free_energy_series(mu, scale, series_temperature, damping=damping, density=density)
        if trace == "stochastic":
            kind = {"rademacher": VEC_RADEMACHER, "z4": VEC_Z4}[vector_kind]
            total = 64 if vectors is None else int(vectors)
            d, e = solver.dots_random(scale, steps, total, seed=seed, first_id=0, kind=kind)
        else:
            raise RuntimeError(f"Trace mode '{trace}' is not supported")
        mu = comm.allreduce_sum(cheb.dots_to_moments(d, e).sum(axis=1)) / total
        return cheb.free_energy_series(mu, scale, series_temperature, damping=damping, density=density)

    if devices is not None:
        if trace == "exact":
            mu = 2.0 * _unit_moment_sum_devices(system, scale, moments, _electron_rows(dim), devices)
        elif trace == "stochastic":
            kind = {"rademacher": VEC_RADEMACHER, "z4": VEC_Z4}[vector_kind]
            total = 64 if vectors is None else int(vectors)
            d, e = dots_random_devices(system, scale, moments // 2, total, devices, seed=seed, kind=kind)
            mu = cheb.dots_to_moments(d, e).sum(axis=1) / total
            if chosen_for_the_caller:
                _warn(f"free_energy() on a {dim}x{dim} matrix is a stochastic-trace estimate from {total} random "
                      "vectors (4N beyond the exact trace); free_energy_stochastic() reports its standard error")
        else:
            raise RuntimeError(f"Trace mode '{trace}' is not supported")
        return cheb.free_energy_series(mu, scale, series_temperature, damping=damping, density=density)

    solver = system._solver()

    if trace == "exact":
        rows = _electron_rows(dim)
        if comm is not None:
            rows = rows[comm.rank :: comm.n_ranks]
        mu = 2.0 * _unit_moment_sum(system, scale, moments, rows)
        if comm is not None:
            mu = comm.allreduce_sum(mu)
    elif trace == "stochastic":
        kind = {"rademacher": VEC_RADEMACHER, "z4": VEC_Z4}[vector_kind]
        total = 64 if vectors is None else int(vectors)
        first, count = shard_vectors(total, comm)
        if comm is None and chosen_for_the_caller:
            # the caller asked for "the free energy" and gets an estimate: say so, with its standard error
            d, e = solver.dots_random(scale, moments // 2, count, seed=seed, first_id=first, kind=kind)
            per_vector_mu = cheb.dots_to_moments(d, e)
            coeff = cheb.chebyshev_coefficients((lambda x: density(scale * x)) if density is not None else
                                                (lambda x: cheb._f_density(scale * x, series_temperature)), moments)
            per_vector = coeff @ per_vector_mu
            sigma = float(np.std(per_vector, ddof=1) / np.sqrt(total)) if total > 1 else float("nan")
            _warn(f"free_energy() on a {dim}x{dim} matrix is beyond the exact trace (4N <= {EXACT_TRACE_LIMIT}): this is a "
                  f"stochastic-trace estimate from {total} random vectors, standard error {sigma:.3g} "
                  f"({abs(sigma / np.mean(per_vector)):.1e} relative; seed={seed}).  Pass trace='stochastic' to "
                  "accept silently, vectors=... for more, or trace='exact'.")
            mu = per_vector_mu.sum(axis=1) / total
        else:
            if chosen_for_the_caller:
                _warn(f"free_energy() on a {dim}x{dim} matrix is a stochastic-trace estimate from {total} random "
                      "vectors (4N beyond the exact trace); free_energy_stochastic() reports its standard error")
            mu = solver.moments_random(scale, moments, count, seed=seed, first_id=first, kind=kind, comm=comm)
            mu = mu / total
    else:
        raise RuntimeError(f"Trace mode '{trace}' is not supported")
    return cheb.free_energy_series(mu, scale, series_temperature, damping=damping, density=density)


def _device_mirrors(system, devices):
    """One device mirror of `system` per entry of `devices` (created here, on the calling thread)."""
    return [system._solver(("devices", slot), device=d) for slot, d in enumerate(devices)]


def _shares(total: int, parts: int):
    """Contiguous shares [(first, count), ...] of `total` items over `parts` owners, empty ones dropped."""
    base, extra = divmod(total, parts)
    out, first = [], 0
    for k in range(parts):
        count = base + (1 if k < extra else 0)
        if count:
            out.append((k, first, count))
        first += count
    return out


def dots_random_devices(system, scale: float, steps: int, vectors: int, devices, *, seed: int = 0,
                        kind: int = VEC_RADEMACHER):
    """Recurrence dots (d, e), each (steps, vectors), of start vectors 0..vectors-1 computed on
    several GPUs of this process: vector ids are shared out contiguously, one host thread per GPU
    (the library is thread safe per handle and ctypes releases the GIL), columns put back in id
    order.  Start vectors are functions of (seed, id, element) alone, so the columns are
    bit-identical to those of a single-GPU call."""
    from concurrent.futures import ThreadPoolExecutor

    mirrors = _device_mirrors(system, devices)
    jobs = _shares(vectors, len(mirrors))
    with ThreadPoolExecutor(len(jobs)) as pool:
        parts = list(pool.map(lambda job: mirrors[job[0]].dots_random(scale, steps, job[2], seed=seed, first_id=job[1], kind=kind), jobs))
    return np.concatenate([p[0] for p in parts], axis=1), np.concatenate([p[1] for p in parts], axis=1)


def _unit_moment_sum_devices(system, scale: float, moments: int, rows: np.ndarray, devices) -> np.ndarray:
    """Σ over `rows` of the unit-vector moments with the rows shared out over several GPUs."""
    from concurrent.futures import ThreadPoolExecutor

    mirrors = _device_mirrors(system, devices)
    jobs = _shares(len(rows), len(mirrors))
    step = max(64, ((1 << 24) // max(1, moments)) // 64 * 64)  # <= ~128 MB of moments per slice on the host

    def run(job):
        slot, first, count = job
        total = np.zeros(moments)
        for lo in range(first, first + count, step):
            total += mirrors[slot].moments_unit(scale, moments, rows[lo : min(lo + step, first + count)]).sum(axis=1)
        return total

    with ThreadPoolExecutor(len(jobs)) as pool:
        return np.sum(list(pool.map(run, jobs)), axis=0)


def free_energy_stochastic(system, temperature: float, *, moments: int | None = None, vectors: int = 64,
                           seed: int = 0, vector_kind: str = "rademacher", scale: float | None = None):
    """(F, σ): stochastic-trace free energy and its standard error, from the spread of the
    per-vector estimates F_r = Σ_m c_m μ_m^{(r)} (F is their mean; σ = std(F_r)/√R).

    For lattices beyond the exact trace this tells how many vectors a wanted accuracy needs;
    differences between configurations computed with the same `seed` are far more accurate than σ
    suggests, because the same vectors are used and most of the noise cancels.
    """
    if temperature <= 0:
        raise ValueError("Expected positive temperature (T = 0 goes through free_energy)")
    if not system.has_symmetric_spectrum(1e-12):
        raise RuntimeError("The Chebyshev free energy needs a particle-hole symmetric Hamiltonian")
    scale = _scale_of(system) if scale is None else float(scale)
    if moments is None:
        moments = cheb.moments_for_free_energy(scale, temperature)
    moments += moments & 1
    kind = {"rademacher": VEC_RADEMACHER, "z4": VEC_Z4}[vector_kind]
    d, e = system._solver().dots_random(scale, moments // 2, int(vectors), seed=seed, kind=kind)
    mu = cheb.dots_to_moments(d, e)  # (M, R)
    coeff = cheb.chebyshev_coefficients(lambda x: cheb._f_density(scale * x, temperature), moments)
    per_vector = coeff @ mu
    sigma = float(np.std(per_vector, ddof=1) / np.sqrt(len(per_vector))) if len(per_vector) > 1 else float("nan")
    return float(np.mean(per_vector)), sigma


def _slab_solver(system, comm):
    """This rank's slab of `system` on its GPU (rebuilt after every `with` block)."""
    from . import slab
    from .solver import DeviceSolver

    cached = getattr(system, "_slab_device", None)
    if cached is not None and cached[0] == system._revision and cached[1] is comm:
        return cached[2]
    if cached is not None:
        cached[2].close()
    indptr, indices, data = system.bsr_arrays()
    bounds = slab.partition_rows(system.lattice.size, comm.n_ranks, slab.lattice_granule(system.lattice))
    plan = slab.build_plan(indptr, indices, data, bounds, comm.rank)
    solver = DeviceSolver.from_slab_plan(plan, comm=comm, device=comm.device)
    system._slab_device = (system._revision, comm, solver)
    return solver


def shard_vectors(total: int, comm) -> tuple[int, int]:
    """Contiguous share [first, first+count) of `total` start vectors for this rank."""
    if comm is None:
        return 0, total
    base, extra = divmod(total, comm.n_ranks)
    count = base + (1 if comm.rank < extra else 0)
    first = comm.rank * base + min(comm.rank, extra)
    if count == 0:
        raise ValueError("fewer start vectors than ranks")
    return first, count


# ----------------------------------------------------------------- diagonalize
def diagonalize(system, format: str = "reshape"):
    """Positive eigenpairs, ascending; same shapes as reference hamiltonian.py:228-248."""
    if format not in ("raw", "reshape"):
        raise RuntimeError(f"Eigenstate format '{format}' is not yet supported.")
    # (only the eigenvectors of the positive eigenvalues are computed and copied: half the work for a BdG matrix)
    vals, vecs = system._solver().eigh_above(0.0)
    vals = np.ascontiguousarray(vals[vals > 0])
    vecs = np.ascontiguousarray(vecs)
    assert vecs.shape[1] == vals.size
    if format == "raw":
        return vals, vecs
    return vals, vecs.T.reshape((vals.size, -1, 4))


# --------------------------------------------------- gap beyond dense reach
def _krylov_length(beta_column: np.ndarray, scale2: float) -> int:
    """Number of Lanczos steps of one start vector that are meaningful.  When the Krylov space of
    the vector is exhausted (a matrix with few distinct eigenvalues: small, or highly symmetric)
    β drops to round-off level and everything the process produces afterwards is noise - Ritz
    values below the spectrum included.  The tridiagonal matrix is cut at the first such β: its
    eigenvalues are then exact eigenvalues of H² (the space is invariant)."""
    tiny = np.flatnonzero(beta_column <= 1e-7 * scale2)
    return int(tiny[0]) + 1 if tiny.size else len(beta_column)



def lowest_eigenvalues(system, k: int = 1, *, tol: float = 1e-6, vectors: int = 4, seed: int = 0,
                       max_iter: int = 50000, check_every: int = 250, method: str = "auto") -> np.ndarray:
    """The k smallest *distinct* positive eigenvalues of H, ascending, for systems where the dense
    `diagonalize()` is out of reach (the reference uses `min(E)` as a gap probe,
    e.g. tests/test_physics.py:43-56, :370).

    Lanczos on H^2 with the recurrence kernel as the matrix-vector product: the lowest Ritz values
    θ converge to the squares of the eigenvalues nearest zero, ε = sqrt(θ).  `vectors` independent
    start vectors run side by side; the result is accepted when the k lowest distinct values
    changed by less than `tol` (relative) since the previous check and the processes agree with
    each other to 5·tol.  Isolated levels (finite systems, in-gap states) converge geometrically
    and reach 1e-9 in a few hundred iterations; the edge of a quasi-continuous band (a bulk gap
    on a large lattice) converges like 1/m², so there the remaining error is a few times `tol`
    and tolerances below ~1e-6 cost tens of thousands of iterations.  If `max_iter` is reached a
    warning is issued and the current estimate returned.
    Degenerate eigenvalues (e.g. spin-degenerate pairs) are reported once; repeated copies that
    plain Lanczos produces after convergence are merged the same way.
    method="auto" answers matrices up to 4N = 2048 from the dense solver instead (exact; a Krylov
    process exhausts so small a space); method="lanczos" always runs the process.
    """
    import warnings

    from scipy.linalg import eigh_tridiagonal

    if k < 1:
        raise ValueError("k must be at least 1")
    if method not in ("auto", "lanczos"):
        raise ValueError("method must be 'auto' or 'lanczos'")
    if method == "auto" and system.shape[0] <= DENSE_AUTO_LIMIT:
        # a Krylov process on a matrix this small exhausts its space within a few checks and breaks
        # down; the library-free dense solve takes a fraction of a second and is exact
        values = diagonalize(system, format="raw")[0]
        distinct = values[np.concatenate([[True], np.diff(values) > 1e-9 * max(1.0, float(values[-1]))])] if len(values) else values
        if len(distinct) < k:
            warnings.warn(f"lowest_eigenvalues: the matrix has only {len(distinct)} distinct positive eigenvalues", RuntimeWarning)
        return distinct[:k]
    solver = system._solver()
    solver.lanczos_begin(vectors, seed=seed, max_iter=max_iter)
    alpha = np.zeros((0, vectors))
    beta = np.zeros((0, vectors))
    previous = None
    scale2 = None
    while alpha.shape[0] < max_iter:
        a, b = solver.lanczos_advance(min(check_every, max_iter - alpha.shape[0]))
        alpha, beta = np.vstack([alpha, a]), np.vstack([beta, b])
        m = alpha.shape[0]
        if scale2 is None:
            scale2 = float(alpha[: min(m, 8)].max())  # ~ |H|^2, sets the merge resolution
        estimates = []
        for c in range(vectors):
            mc = _krylov_length(beta[:, c], scale2)
            want = min(mc, 4 * k + 8)
            theta = eigh_tridiagonal(alpha[:mc, c], beta[: mc - 1, c], select="i", select_range=(0, want - 1),
                                     eigvals_only=True) if mc > 1 else alpha[:1, c]
            eps = np.sqrt(np.clip(theta, 0.0, None))
            distinct = [eps[0]]
            for value in eps[1:]:
                if value - distinct[-1] > 1e-7 * np.sqrt(scale2):
                    distinct.append(value)
            estimates.append(distinct[:k])
        if all(len(e) == k for e in estimates):
            current = np.array(estimates)
            spread = np.max(np.abs(current - current[0]) / np.maximum(current[0], 1e-300))
            if previous is not None and previous.shape == current.shape:
                drift = np.max(np.abs(current - previous) / np.maximum(current, 1e-300))
                if spread < 5 * tol and drift < tol:
                    return np.min(current, axis=0)  # Ritz values approach eigenvalues from above
            previous = current
    if previous is None:
        raise RuntimeError(f"Lanczos found fewer than {k} distinct levels in {max_iter} iterations")
    warnings.warn(f"lowest_eigenvalues: not converged to {tol:g} after {max_iter} iterations", RuntimeWarning)
    return np.min(previous, axis=0)


def lowest_eigenpairs(system, k: int = 1, *, tol: float = 1e-8, vectors: int = 8, seed: int = 0,
                      max_iter: int = 20000, check_every: int = 50, format: str = "reshape", method: str = "auto",
                      rayleigh_ritz: str = "device"):
    """The k lowest positive eigenvalues of H **with their multiplicities** and orthonormal
    eigenvectors, for systems where the dense `diagonalize()` is out of reach: what
    `E, v = system.diagonalize()` followed by `E[:k], v[:k]` gives the reference's callers
    (e.g. ref tests/test_physics.py:105), same layouts (ref hamiltonian.py:235-248: "raw" ->
    (E (k,), X (4N, k)), "reshape" -> (E, v[n, site, α])).

    Two passes of the device Lanczos process on H² with `vectors` independent start vectors:
      1. α, β as in `lowest_eigenvalues`; at every checkpoint the host diagonalises the tridiagonal
         matrices and records, per start vector and level, the Ritz coordinates at the FIRST
         checkpoint where the residual estimate β_m |s_m| has dropped below tol·ε (before plain
         Lanczos starts to produce ghost copies of a converged level);
      2. the process is repeated (same seed: the Lanczos vectors are reproduced bit for bit) and
         the Ritz vectors y are accumulated on the device (`bdg_lanczos_ritz_vectors`).
    y lies in the eigenspace of H² for ε², i.e. in span(eigenvectors of H for +ε and −ε);
    (H + ε) y projects onto the +ε part.  The `vectors` candidates of a level span its whole
    eigenspace (up to `vectors` dimensions): their Gram matrix gives the multiplicity, and a
    Rayleigh-Ritz step with H inside that span the final pairs.  Residuals ‖Hv − εv‖ come out at
    about `tol`; eigenvalues, being Rayleigh quotients, at about tol².
    method="auto" answers matrices up to 4N = 2048 from `diagonalize()` itself; method="lanczos"
    always runs the two passes.  rayleigh_ritz="device" (default, up to 16 start vectors) keeps the
    block of Ritz vectors on the GPU and does the Gram / projection steps there
    (`bdg_lanczos_ritz_pairs`: only r x r matrices and the k results cross PCIe); "host" is the
    numpy form of the same steps on (levels, vectors, 4N) arrays, kept for comparison.
    """
    import warnings

    from scipy.linalg import eigh_tridiagonal

    if k < 1:
        raise ValueError("k must be at least 1")
    if format not in ("raw", "reshape"):
        raise RuntimeError(f"Eigenstate format '{format}' is not yet supported.")
    if method not in ("auto", "lanczos"):
        raise ValueError("method must be 'auto' or 'lanczos'")
    if rayleigh_ritz not in ("device", "host"):
        raise ValueError("rayleigh_ritz must be 'device' or 'host'")
    if method == "auto" and system.shape[0] <= DENSE_AUTO_LIMIT:
        # small matrices: the dense solve is exact and cheaper than a Krylov process that would
        # exhaust its space (found by scratch/fuzz_api.py on a 72x72 matrix)
        values, states = diagonalize(system, format="raw")
        if len(values) < k:
            raise RuntimeError(f"lowest_eigenpairs: the matrix has only {len(values)} positive eigenvalues, {k} asked for")
        values, states = values[:k], np.asarray(states)[:, :k]
        return (values, states) if format == "raw" else (values, states.T.reshape(k, -1, 4))
    solver = system._solver()
    dim = system.shape[0]
    levels = k  # k distinct levels hold at least k states
    solver.lanczos_begin(vectors, seed=seed, max_iter=max_iter)
    alpha = np.zeros((0, vectors))
    beta = np.zeros((0, vectors))
    # per start vector: converged Ritz pairs (θ, m, s) in the order they converged, identified by
    # VALUE (the position of a Ritz value in the sorted list changes while lower levels are still
    # emerging); later ghost copies of a recorded value are ignored
    records: list[list[tuple[float, int, np.ndarray]]] = [[] for _ in range(vectors)]
    targets = None
    while alpha.shape[0] < max_iter:
        a, b = solver.lanczos_advance(min(check_every, max_iter - alpha.shape[0]))
        alpha, beta = np.vstack([alpha, a]), np.vstack([beta, b])
        m = alpha.shape[0]
        scale2 = float(alpha[: min(m, 8)].max())  # ~ |H|^2 (early rows: noise after a breakdown can be large)
        same = 1e-9 * scale2         # two Ritz values this close are one level (or a ghost of it)
        pending = []                 # per start vector: lowest Ritz value that is neither converged nor a ghost
        for c in range(vectors):
            mc = _krylov_length(beta[:, c], scale2)  # (< m: this vector's Krylov space is exhausted, its Ritz pairs exact)
            want = min(mc, 4 * levels + 8)
            if mc > 1:
                theta, s = eigh_tridiagonal(alpha[:mc, c], beta[: mc - 1, c], select="i", select_range=(0, want - 1))
            else:
                theta, s = alpha[:1, c].copy(), np.ones((1, 1))
            residual = (beta[mc - 1, c] if mc == m else 0.0) * np.abs(s[-1, :])
            lowest_open = np.inf
            for idx in range(want):
                known = any(abs(theta[idx] - t) <= same for t, _, _ in records[c])
                eps = np.sqrt(max(theta[idx], 0.0))
                if not known and residual[idx] <= tol * max(eps, 1e-6 * np.sqrt(scale2)):
                    records[c].append((float(theta[idx]), mc, s[:, idx].copy()))
                    known = True
                if not known:
                    lowest_open = min(lowest_open, theta[idx])
            pending.append(lowest_open)
        # distinct converged levels over all start vectors, lowest first
        merged: list[float] = []
        for t in sorted(t for rec in records for t, _, _ in rec):
            if not merged or t - merged[-1] > same:
                merged.append(t)
        if len(merged) >= levels:
            wanted = merged[:levels]
            complete = all(any(abs(t - w) <= same for t, _, _ in rec) for rec in records for w in wanted)
            nothing_below = all(p > wanted[-1] + same for p in pending)  # no lower level still emerging anywhere
            if complete and nothing_below:
                targets = wanted
                break
    if targets is None:
        merged = []
        for t in sorted(t for rec in records for t, _, _ in rec):
            if not merged or t - merged[-1] > 1e-9 * scale2:
                merged.append(t)
        targets = merged[:levels]
        if not targets:
            raise RuntimeError(f"Lanczos converged no level to {tol:g} in {max_iter} iterations")
        warnings.warn(f"lowest_eigenpairs: only {len(targets)} of {levels} levels converged to {tol:g} in every start "
                      f"vector after {alpha.shape[0]} iterations; returning what was found", RuntimeWarning)
    levels = len(targets)
    same = 1e-9 * scale2
    chosen = [[next(((m, s) for t, m, s in rec if abs(t - w) <= same), None) for w in targets] for rec in records]
    n_iter = max(entry[0] for row in chosen for entry in row if entry is not None)
    coef = np.zeros((n_iter, levels, vectors))
    for c, row in enumerate(chosen):
        for l, entry in enumerate(row):
            if entry is not None:
                coef[: entry[0], l, c] = entry[1]
    solver.lanczos_begin(vectors, seed=seed, max_iter=max_iter)
    if rayleigh_ritz == "device" and vectors <= 16:
        # second pass and Rayleigh-Ritz inside the library: the (levels, vectors, 4N) block of Ritz vectors stays on
        # the GPU, H acts there, r x r matrices go to the host and only the k states asked for come back
        eps = np.sqrt(np.maximum(np.asarray(targets[:levels], dtype=float), 0.0))
        vals, vecs, _ = solver.lanczos_ritz_pairs(coef, eps, k)
        if format == "raw":
            return vals, np.ascontiguousarray(vecs.T)
        return vals, vecs.reshape(len(vals), dim // 4, 4)
    ritz = solver.lanczos_ritz_vectors(coef)  # (levels, vectors, 4N)

    values, states = [], []
    for l in range(levels):
        eps = float(np.sqrt(max(targets[l], 0.0)))
        # candidates in the +eps eigenspace of H, then an orthonormal basis of what they span
        have = [c for c in range(vectors) if chosen[c][l] is not None]
        cand = np.stack([solver.spmv(ritz[l, c]) + eps * ritz[l, c] for c in have])
        # a Ritz vector may happen to lie (almost) wholly in the -eps eigenspace: its projection is
        # round-off noise and must not be blown up to a unit vector (scratch/fuzz_api.py found one)
        length = np.linalg.norm(cand, axis=1)
        cand = cand[length > 1e-6 * length.max()]
        cand /= np.linalg.norm(cand, axis=1, keepdims=True)
        gram = cand.conj() @ cand.T
        weight, mix = np.linalg.eigh(gram)
        keep = weight > 1e-4 * weight[-1]
        # gram[i, j] = <c_i|c_j>, so b_m = Σ_i c_i mix[i, m] / sqrt(w_m) are orthonormal (no conjugate on
        # mix: with it the rows are orthonormal only when the Gram matrix is real - ADVICE r2)
        basis = (mix[:, keep] / np.sqrt(weight[keep])).T @ cand  # (rank = multiplicity)
        basis = np.linalg.qr(basis.T)[0].T  # tidy up round-off
        h_basis = np.stack([solver.spmv(v) for v in basis])
        small = basis.conj() @ h_basis.T
        energy, rot = np.linalg.eigh((small + small.conj().T) / 2)
        final = rot.T @ basis  # eigenvectors of H restricted to the span (rows)
        for e, v in zip(energy, final):
            values.append(float(e))
            states.append(v / np.linalg.norm(v))
    order = np.argsort(values)[:k]
    vals = np.array([values[i] for i in order])
    vecs = np.stack([states[i] for i in order])  # (k, 4N)
    if format == "raw":
        return vals, np.ascontiguousarray(vecs.T)
    return vals, vecs.reshape(len(order), dim // 4, 4)


# ------------------------------------------------------------------------ LDOS
def ldos(system, site, energies, *, moments: int | None = None, scale: float | None = None,
         digits: float = 12.0) -> np.ndarray:
    """LDOS at `site` (reference hamiltonian.py:324-387) from the Chebyshev resolvent.

    The broadening follows the reference exactly: ε = unique(|E|), Γ = gradient(ε),
    evaluated with the same numpy calls so that its one-ulp quirks carry over.
    `site` may also be a list of coordinates: all sites share the recurrence launches (16
    sites = 64 unit start vectors per batch) and the result has one row per site.
    """
    single = len(site) == 3 and all(isinstance(c, (int, np.integer)) for c in site)
    sites = [tuple(site)] if single else [tuple(s) for s in site]
    energies = np.array(energies, dtype=float)
    eps = np.unique(np.abs(energies))
    gam = np.gradient(eps)
    scale = _scale_of(system) if scale is None else float(scale)
    if moments is None:
        moments = cheb.moments_for_resolvent(scale, float(np.min(gam)), digits)
    moments += moments & 1

    index = np.array([system.lattice[s] for s in sites], dtype=np.int64)
    rows = (4 * index[:, None] + np.arange(4)[None, :]).reshape(-1)
    mu = _unit_moments(system, scale, moments, rows)  # (M, 4 * n_sites)

    out = np.empty((len(sites), energies.size))
    for n in range(len(sites)):
        rho = {}
        for e, g in zip(eps, gam):
            diag = [cheb.resolvent_series(mu[:, 4 * n + a], scale, e + 1j * g) for a in range(4)]
            rho[+e] = -np.imag(diag[0] + diag[1]) / np.pi
            rho[-e] = -np.imag(diag[2] + diag[3]) / np.pi
        out[n] = [rho[e] for e in energies]
    return out[0] if single else out
