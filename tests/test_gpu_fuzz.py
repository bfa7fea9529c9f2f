"""Randomised parity on the GPU (fixed seeds): tests/fuzz_sweep.py and tests/fuzz_api.py (also runnable at
length through scratch/fuzz_*.py) with a bounded number of cases.  They found three defects when they were written (profiles/r02_fuzz_api.log, the fuzz section of
profiles/r02_sweep_experiments.log); here they keep watch.

* fuzz_sweep.py  random lattice shapes / models / step and vector counts / tuning knobs: the
                         lattice-stencil kernels (K7, K7b, K8, forced) against the one-step kernels, 1e-12·4N
* fuzz_dense.py  random BdG matrices (decoupled chains, flat bands, zero modes, tiny lattices, complex blocks) through the
                         library's own dense route: eigenvalues vs numpy 1e-11, eigenvectors' residual / orthonormality 1e-9
* fuzz_api.py    random small systems through the public API against the CPU oracle: free energy
                         (dense, Chebyshev exact and stochastic trace), diagonalize, ldos (single, several
                         sites, band-limited), unit-start recurrences, slab groups, lowest_eigenpairs,
                         several mirrors on one GPU, a second `with` block
"""

import pytest

import fuzz_api
import fuzz_dense
import fuzz_sweep

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,size,cases", [(101, None, 400), (102, "large", 150)])
def test_random_lattices_stencil_kernels_against_one_step_kernels(hip_library, seed, size, cases):
    assert fuzz_sweep.run(seed=seed, n_cases=cases, size=size) == 0


def test_random_systems_through_the_api_against_the_oracle(hip_library):
    assert fuzz_api.run(seed=103, n_cases=24) == 0


def test_random_matrices_through_the_own_dense_route(hip_library):
    assert fuzz_dense.run(seed=104, n_cases=18) == 0


def test_random_matrices_through_the_two_stage_dense_route(hip_library):
    """K10 (csrc/twostage.hpp) forced on random lattices of every kind, tiny to 2500 rows: degenerate spectra, decoupled chains
    (panels that are exactly rank deficient: the Gram route's fallback), zero modes."""
    assert fuzz_dense.run(seed=105, n_cases=16, stages="2") == 0
