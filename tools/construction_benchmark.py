#!/usr/bin/env python3
"""Construction benchmark: the reference's only published timing, run against this front end.

The reference publishes one table (misc/benchmark.csv, method in misc/benchmark.py:91-160): wall
time to build the BdG Hamiltonian of an S/F junction on an L x W square lattice and export it as
CSR, for L = 2 .. 2048 and W = L, 2L.  This script builds the same physical system through
`bodge_amd.Hamiltonian` and prints its seconds beside the published ones:

    x <  L/2   superconductor: gap Δ0 e^{iχx/L} (phase winding along x), H_ii = -μ σ0
    x >= L/2   ferromagnet:    H_ii = -μ σ0 - M0 σ3
    hopping -t along x, -2t along y;  t = 1, μ = -3t, M0 = 1.5t, Δ0 = 0.1t, χ = 0.5

in two styles:

    loops   one assignment per site and per bond in Python loops over lattice.sites() and
            lattice.bonds(axis) - how a user of the reference writes it (hamiltonian.py:102-118);
    arrays  the array-valued setters (set_sites / set_bonds), SURVEY §8 row f1.

Both produce the same matrix (checked below for the sizes where both run).  Host code only: no GPU.

    python3 tools/construction_benchmark.py [--max-atoms 4194304] [--loops-max-atoms 1048576]
"""

from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from bodge_amd import CubicLattice, Hamiltonian  # noqa: E402
from bodge_amd.common import jσ2, σ0, σ3  # noqa: E402

T_HOP, MU, M0, DELTA0, CHI = 1.0, -3.0, 1.5, 0.1, 0.5

# Seconds published by the reference for its own package ("Bodge" rows of misc/benchmark.csv, one
# machine, unspecified), by number of atoms.  16 384 atoms appears twice there (128 x 128 and 64 x 256).
PUBLISHED = {
    4: 0.00119, 8: 0.00173, 16: 0.00286, 32: 0.00454, 64: 0.0107, 128: 0.0157, 256: 0.0312, 512: 0.0621,
    1024: 0.125, 2048: 0.249, 4096: 0.491, 8192: 0.989, 16384: 2.11, 32768: 4.07, 65536: 8.29, 131072: 16.6,
    262144: 32.7, 524288: 65.6, 1048576: 133.9, 2097152: 278.7, 4194304: 581.0,
}


def build_loops(L: int, W: int):
    lattice = CubicLattice((L, W, 1))
    system = Hamiltonian(lattice)
    with system as (H, Δ):
        for i in lattice.sites():
            if i[0] < L // 2:
                H[i, i] = -MU * σ0
                Δ[i, i] = -DELTA0 * np.exp(1j * CHI * i[0] / L) * jσ2
            else:
                H[i, i] = -MU * σ0 - M0 * σ3
        for i, j in lattice.bonds(axis=0):
            H[i, j] = -T_HOP * σ0
        for i, j in lattice.bonds(axis=1):
            H[i, j] = -2 * T_HOP * σ0
    return system.matrix(format="csr")


def build_arrays(L: int, W: int):
    lattice = CubicLattice((L, W, 1))
    system = Hamiltonian(lattice)
    x = np.arange(lattice.size) // W  # site index = y + W * x  (z-extent 1)
    superconducting = (x < L // 2)[:, None, None]
    with system as (H, Δ):
        H.set_sites(np.where(superconducting, -MU * σ0, -MU * σ0 - M0 * σ3))
        Δ.set_sites(np.where(superconducting, -DELTA0 * np.exp(1j * CHI * x / L)[:, None, None] * jσ2, 0 * jσ2))
        H.set_bonds(-T_HOP * σ0, axis=0)
        if W > 1:
            H.set_bonds(-2 * T_HOP * σ0, axis=1)
    return system.matrix(format="csr")


def timed(build, L, W, repeats):
    best, out = float("inf"), None
    for _ in range(repeats):
        t0 = time.perf_counter()
        out = build(L, W)
        best = min(best, time.perf_counter() - t0)
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-atoms", type=int, default=4194304)
    ap.add_argument("--loops-max-atoms", type=int, default=1048576, help="largest system built with per-site Python loops")
    args = ap.parse_args()
    print(f"# host: {os.cpu_count()} logical cores; seconds = min over repeats of build + CSR export")
    print(f"{'atoms':>9s} {'L x W':>12s} {'reference (published)':>22s} {'loops':>10s} {'arrays':>10s} {'published / loops':>18s} {'published / arrays':>19s}")
    for n in range(1, 12):
        L = 2 ** n
        for W in (L, 2 * L):
            atoms = L * W
            if atoms > args.max_atoms:
                continue
            repeats = 30 if atoms < 1e4 else 3 if atoms < 2e5 else 1  # (the reference: 30 below 10^4 atoms, else 1)
            t_arrays, m_arrays = timed(build_arrays, L, W, repeats)
            t_loops = None
            if atoms <= args.loops_max_atoms:
                t_loops, m_loops = timed(build_loops, L, W, repeats)
                diff = abs(m_loops - m_arrays)
                assert diff.nnz == 0 or diff.max() == 0.0, "the two styles built different matrices"
            ref = PUBLISHED.get(atoms)
            print(f"{atoms:9d} {f'{L} x {W}':>12s} {ref if ref is not None else float('nan'):22.4g} "
                  f"{t_loops if t_loops is not None else float('nan'):10.4g} {t_arrays:10.4g} "
                  f"{(ref / t_loops) if ref and t_loops else float('nan'):18.1f} {(ref / t_arrays) if ref else float('nan'):19.1f}", flush=True)


if __name__ == "__main__":
    main()
