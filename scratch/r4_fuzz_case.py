"""one case of tests/fuzz_dense.py (seed, case) under several switches of the two-stage route"""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import numpy as np
import fuzz_dense
from bodge_amd import backend
from bodge_amd.solver import DeviceSolver

seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    system, tag = fuzz_dense._system(rng, case)
print(tag)
dense = np.asarray(system.matrix("dense"))
exact = np.linalg.eigvalsh(dense)
for label, opts in (("one stage", {"BODGE_AMD_EIGH_STAGES": "1"}), ("two stages", {"BODGE_AMD_EIGH_STAGES": "2"}),
                    ("two stages, QR with barriers", {"BODGE_AMD_EIGH_STAGES": "2", "BODGE_AMD_EIGH_GRAM_QR": "0"}),
                    ("two stages, floor 0.1", {"BODGE_AMD_EIGH_STAGES": "2", "BODGE_AMD_EIGH_GRAM_FLOOR": "0.1"}),
                    ("two stages, floor 0.3", {"BODGE_AMD_EIGH_STAGES": "2", "BODGE_AMD_EIGH_GRAM_FLOOR": "0.3"}),
                    ("two stages, floor 0.6", {"BODGE_AMD_EIGH_STAGES": "2", "BODGE_AMD_EIGH_GRAM_FLOOR": "0.6"}),
                    ("two stages, floor 0.9", {"BODGE_AMD_EIGH_STAGES": "2", "BODGE_AMD_EIGH_GRAM_FLOOR": "0.9"})):
    with DeviceSolver.from_hamiltonian(system) as dev, backend.options(BODGE_AMD_EIGH="tridiagonal", BODGE_AMD_TRACE="1", **opts):
        w, _ = dev.eigh(vectors=False)
        w2, z = dev.eigh_above(0.0)
    vals = w2[w2 > 0]
    res = np.abs(dense @ z - z * vals).max() if vals.size else 0.0
    print(f"{label:34s} eigenvalues off by {np.abs(w - exact).max():.2e}  (vectors call: {np.abs(w2 - exact).max():.2e})  residual {res:.2e}")
