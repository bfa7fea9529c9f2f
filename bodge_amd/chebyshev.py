"""Host-side scalar work of the Chebyshev path (O(M) and O(nnz) numpy, no matvecs).

The device produces moments μ_m = Tr-like sums of T_m(H/a); everything here is
the bookkeeping around them:

* `spectral_bound`      a ≥ ‖H‖ from Gershgorin discs over the BSR blocks
* `free_energy_series`  F = Σ_m c_m μ_m with c_m the Chebyshev coefficients of
                        f(ε) = -(T/2) ln(2 cosh(ε/2T)), whose sum over the full
                        ±-symmetric spectrum is the reference's
                        -½Σ_{ε>0}ε - TΣ_{ε>0}log(1+e^{-ε/T})  (ref hamiltonian.py:305-321)
* `resolvent_series`    <e|(z-H)^{-1}|e> from per-vector moments (LDOS, ref :367-382)
* `moments_for_*`       how many moments a requested accuracy needs
"""

from __future__ import annotations

import numpy as np


MAX_MOMENTS = 1 << 17  # default cap of the expansion order (low temperatures)


def _native_row_sum_max(indptr: np.ndarray, data: np.ndarray):
    """The row-sum maximum from the library's threaded block scan (`bdg_host_scan_blocks`: numpy's
    summation order, so the same double for real or purely imaginary entries), or None when the
    arrays are not the library's layout or the library is absent (BODGE_AMD_HOST_NATIVE=0: never)."""
    import ctypes as C
    import os

    if os.environ.get("BODGE_AMD_HOST_NATIVE", "1") == "0":
        return None
    if not (isinstance(data, np.ndarray) and data.dtype == np.complex128 and data.ndim == 3 and data.shape[1:] == (4, 4)
            and data.flags.c_contiguous and isinstance(indptr, np.ndarray) and indptr.dtype == np.int32
            and indptr.flags.c_contiguous and len(indptr) >= 1 and int(indptr[-1]) == len(data)):
        return None
    try:
        from . import backend

        lib = backend.load()
    except (RuntimeError, OSError):
        return None
    bound = C.c_double(0.0)
    if lib.bdg_host_scan_blocks(backend.as_f64p(data), backend.as_i32p(indptr), len(indptr) - 1, None, None, None,
                                C.byref(bound), None) != 0:
        return None
    return float(bound.value)


def spectral_bound(indptr: np.ndarray, data: np.ndarray, pad: float = 1.01) -> float:
    """max over scalar rows of Σ|H_rc|, times `pad` (> 1 keeps the spectrum strictly inside)."""
    n_sites = len(indptr) - 1
    if len(data) == 0:
        return 1.0
    bound = _native_row_sum_max(indptr, data)
    if bound is not None:
        return pad * bound if (bound > 0 or np.isnan(bound)) else 1.0  # (NaN entries give NaN, as the numpy form below does)
    if np.iscomplexobj(data) and not data.imag.any():
        per_block = np.abs(data.real).sum(axis=2)  # same numbers as |z|, without the hypot pass
    else:
        per_block = np.abs(data).sum(axis=2)  # (nnzb, 4)
    lengths = np.diff(indptr)
    if lengths.min(initial=1) > 0:
        radius = np.add.reduceat(per_block, indptr[:-1].astype(np.intp), axis=0)
    else:  # empty rows break reduceat's segment convention
        owner = np.repeat(np.arange(n_sites), lengths)
        radius = np.zeros((n_sites, 4))
        np.add.at(radius, owner, per_block)
    bound = float(radius.max())
    return pad * bound if bound > 0 else 1.0


def _f_density(eps: np.ndarray, temperature: float) -> np.ndarray:
    mag = np.abs(eps)
    if temperature == 0:
        return -mag / 4
    return -mag / 4 - (temperature / 2) * np.log1p(np.exp(-mag / temperature))


def chebyshev_coefficients(func, n_moments: int, oversample: int = 4) -> np.ndarray:
    """Coefficients c_m, m < M, of `func` on [-1, 1] (c_0 already halved).

    Chebyshev-Gauss quadrature on `oversample * M` nodes, c_m = (2/N) Σ_k f(cos θ_k) cos(m θ_k) with
    θ_k = π (k + ½) / N: a type-II discrete cosine transform, O(N log N) (low temperatures need
    M ~ 10^5 moments, where the direct cosine sums would take minutes).
    """
    from scipy.fft import dct

    nodes = oversample * n_moments
    theta = np.pi * (np.arange(nodes) + 0.5) / nodes
    coeff = dct(func(np.cos(theta)), type=2)[:n_moments] / nodes  # dct-II = 2 Σ_k v_k cos(m θ_k)
    coeff[0] *= 0.5
    return coeff


def dots_to_moments(d: np.ndarray, e: np.ndarray) -> np.ndarray:
    """(steps, R) recurrence dots -> (2*steps, R) moments: μ_2n = 2 d_n - μ_0, μ_2n+1 = 2 e_n - μ_1."""
    mu = np.empty((2 * d.shape[0],) + d.shape[1:])
    mu[0::2] = 2 * d - d[0]
    mu[1::2] = 2 * e - e[0]
    mu[0], mu[1] = d[0], e[0]
    return mu


def jackson_kernel(n_moments: int) -> np.ndarray:
    m = np.arange(n_moments)
    q = np.pi / (n_moments + 1)
    return ((n_moments - m + 1) * np.cos(q * m) + np.sin(q * m) / np.tan(q)) / (n_moments + 1)


def gapped_ground_state_density(width: float):
    """ε -> -(ε/4)·erf(ε/width): the T = 0 density -|ε|/4 with its kink at ε = 0 smoothed over `width`.

    On a spectrum with |ε| ≥ gap the two differ by (|ε|/4)·erfc(|ε|/width) ≤ (gap/4)·erfc(gap/width),
    4e-13·gap for width = gap/5, and the smoothed function is entire: its Chebyshev series converges
    faster than geometrically.  Pointwise on the gapped part of [-a, a] the truncated series is within
    1e-13 (absolute, per level) of -|ε|/4 at M = 10·a/width - 6e-11 at 8·a/width, 8e-13 at 9·a/width,
    tests/test_host_math.py - a third of what f_T at T = gap/20 needs for 1e-11; the error of F itself
    is usually smaller still, because the four measured spectra cancel part of it."""
    import math

    erf = np.vectorize(math.erf, otypes=[float])  # (scipy.special costs seconds to import from a cold disk)
    return lambda eps: -(np.asarray(eps, dtype=float) / 4.0) * erf(np.asarray(eps, dtype=float) / width)


def moments_for_gapped_ground_state(scale: float, width: float) -> int:
    """Even M = 10·a/width for the series of `gapped_ground_state_density(width)` (may exceed MAX_MOMENTS)."""
    m = max(32, int(np.ceil(10.0 * scale / width)))
    return m + (m & 1)


def free_energy_series(mu_trace: np.ndarray, scale: float, temperature: float, damping: bool = False,
                       density=None) -> float:
    """F from trace moments μ_m ≈ Tr T_m(H/scale); `density` replaces f_T(ε) (see gapped_ground_state_density)."""
    n = len(mu_trace)
    coeff = chebyshev_coefficients((lambda x: density(scale * x)) if density is not None
                                   else (lambda x: _f_density(scale * x, temperature)), n)
    if damping:
        coeff = coeff * jackson_kernel(n)
    return float(np.dot(coeff, mu_trace))


def moments_for_free_energy(scale: float, temperature: float, digits: float = 11.0) -> int:
    """Even M for ~10^-digits truncation error of f on [-a, a].

    f is analytic in the strip |Im ε| < πT, i.e. inside the Bernstein ellipse
    with ρ = 1 + πT/a (small T/a), so the coefficients decay like ρ^-m.
    T = 0 is only algebraic (|ε|); a fixed large default is returned.
    """
    if temperature <= 0:
        return 4096
    rate = np.log1p(np.pi * temperature / scale)
    m = int(np.ceil(digits * np.log(10.0) / rate))
    if m > MAX_MOMENTS:
        import warnings

        warnings.warn(
            f"T/scale = {temperature / scale:.1e} asks for {m} Chebyshev moments; capped at {MAX_MOMENTS}, "
            f"expect a truncation error of about 1e-{digits * MAX_MOMENTS / m:.0f} instead of 1e-{digits:.0f} "
            "(pass moments=... to override)", RuntimeWarning, stacklevel=3)
    m = int(np.clip(m, 32, MAX_MOMENTS))
    return m + (m & 1)


def moments_for_resolvent(scale: float, gamma: float, digits: float = 12.0) -> int:
    """Even M with exp(-M Γ/a) ≲ 10^-digits for the resolvent series at broadening Γ."""
    m = int(np.ceil(digits * np.log(10.0) * scale / gamma))
    m = int(np.clip(m, 64, 1 << 20))
    return m + (m & 1)


def resolvent_series(mu: np.ndarray, scale: float, z: complex) -> complex:
    """<e|(z - H)^{-1}|e> given μ_n = <e|T_n(H/scale)|e>, for Im z > 0."""
    zt = complex(z) / scale
    angle = np.arccos(zt)
    n = np.arange(len(mu))
    weights = np.exp(-1j * n * angle)
    weights[1:] *= 2.0
    return complex(-1j / np.sqrt(1 - zt * zt) * np.dot(weights, mu) / scale)
