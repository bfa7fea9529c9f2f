"""Which pair of eigenvectors loses orthogonality in fuzz_dense seed 333 case 58 with BODGE_AMD_EIGH_DEFER=3?"""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import numpy as np
import fuzz_dense
from bodge_amd import backend
from bodge_amd.solver import DeviceSolver

seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    system, tag = fuzz_dense._system(rng, case)
print(tag, system.shape)
dense = np.asarray(system.matrix("dense"))
exact = np.linalg.eigvalsh(dense)
span = np.abs(exact).max()
for defer in ("1", "3", "4"):
    with DeviceSolver.from_hamiltonian(system) as dev, backend.options(BODGE_AMD_EIGH="tridiagonal", BODGE_AMD_EIGH_DEFER=defer):
        w2, z = dev.eigh_above(0.0)
    vals = w2[w2 > 0]
    gram = np.abs(z.conj().T @ z - np.eye(vals.size))
    i, j = np.unravel_index(np.argmax(gram), gram.shape)
    i, j = min(i, j), max(i, j)
    gaps = np.diff(vals)
    print(f"defer {defer}: worst overlap {gram.max():.2e} between vectors {i} and {j}: values {vals[i]:.15f} {vals[j]:.15f}, difference / span {(vals[j]-vals[i])/span:.2e}")
    lo, hi = max(0, i - 3), min(len(vals) - 1, j + 3)
    print("   gaps / span around them:", " ".join(f"{g/span:.1e}" for g in gaps[lo:hi]))
    # cluster (chain of gaps < 1e-5 span) that contains i
    a = i
    while a > 0 and gaps[a - 1] < 1e-5 * span: a -= 1
    b = i
    while b < len(vals) - 1 and gaps[b] < 1e-5 * span: b += 1
    print(f"   chain of gaps < 1e-5 span around vector {i}: {a} .. {b} ({b - a + 1} members); vector {j} inside: {a <= j <= b}")
    print(f"   residual {np.abs(dense @ z - z * vals).max():.2e}; norms of the two: {np.linalg.norm(z[:, i]):.15f} {np.linalg.norm(z[:, j]):.15f}")
    sub = gram[a:b + 1, a:b + 1]
    print(f"   worst overlap inside that chain {sub.max():.2e}; rows with overlaps > 1e-10: {np.unique(np.argwhere(gram > 1e-10)[:, 0])[:20]}")

print("---- per-vector detail with defer 3")
with DeviceSolver.from_hamiltonian(system) as dev, backend.options(BODGE_AMD_EIGH="tridiagonal", BODGE_AMD_EIGH_DEFER="3"):
    w, _ = dev.eigh(vectors=False)
    w2, z = dev.eigh_above(0.0)
print("max |w - exact| =", np.abs(w - exact).max(), " span", span)
vals = w2[w2 > 0]
res = np.abs(dense @ z - z * vals).max(axis=0)
bad = np.argsort(res)[::-1][:8]
for k in bad:
    print(f"   vector {k}: value {vals[k]:.15f} residual {res[k]:.2e}")
# Rayleigh quotients and the exact eigenvalues nearest to them
rq = np.real(np.einsum("ij,ij->j", z.conj(), dense @ z))
print("   max |Rayleigh quotient - value| =", np.abs(rq - vals).max())
