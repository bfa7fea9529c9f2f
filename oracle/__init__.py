"""CPU oracle for the BdG hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in `bodge_amd/` imports this package.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may use it,
and only as the checker: the product path is the HIP library and it fails
loudly when that library is missing.

Parity status: PINNED.  `oracle.dense_ref` restates the reference's dense
SciPy algorithms line by line and is checked against golden values produced by
importing the reference itself (`tests/golden/make_golden.py`, run in the build
container where `/root/reference` exists; outputs in `tests/golden/*.json|npz`).
`oracle.cheb_ref` (the Chebyshev path, which the reference does not contain) is
pinned transitively: its exact-trace free energy and its resolvent LDOS must
reproduce those same goldens to the tolerances stated in `tests/test_oracle.py`.
`oracle.cheb_c` (C + OpenMP recurrence step, the multi-threaded CPU baseline) is
checked against `oracle.cheb_ref` there as well.
"""
