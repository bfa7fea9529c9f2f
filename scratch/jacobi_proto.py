import sys; sys.path.insert(0, ".")
import numpy as np, bench
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mode = sys.argv[2] if len(sys.argv) > 2 else "eigh"
s = bench.build_system([L, L, 1]); H = np.asarray(s.matrix("dense")); n = H.shape[0]
g = np.abs(H).sum(axis=1).max(); shift = 1.5 * g + 1
b = 8; npad = -(-n // 16) * 16
G = np.zeros((npad, npad), complex); G[:n, :n] = H + shift * np.eye(n)
for k in range(n, npad): G[k, k] = 3 * g + 2
nb = npad // b
def pair(m_, rnd, k):
    m = m_ - 1
    return (m, rnd % m) if k == 0 else ((rnd + k) % m, (rnd - k + m) % m)
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-15
for sweep in range(30):
    active = 0
    for rnd in range(nb - 1):
        for k in range(nb // 2):
            P, Q = pair(nb, rnd, k)
            cols = np.r_[P * b:(P + 1) * b, Q * b:(Q + 1) * b]
            W = G[:, cols]; M = W.conj().T @ W
            d = np.sqrt(np.real(np.diag(M)))
            off = np.abs(M) / np.outer(d, d); np.fill_diagonal(off, 0)
            if off.max() <= tol: continue
            active += 1
            w, U = np.linalg.eigh(M)
            if mode == "eigh_sorted":
                # order eigenvectors so that U is as close to identity as possible (greedy on |U|)
                A = np.abs(U).copy(); perm = -np.ones(16, int)
                for _ in range(16):
                    i, j = np.unravel_index(np.argmax(A), A.shape); perm[i] = j; A[i, :] = -1; A[:, j] = -1
                U = U[:, perm]
                U = U * np.exp(-1j * np.angle(np.diag(U)))
            elif mode == "eigh_desc":
                U = U[:, ::-1]
            G[:, cols] = W @ U
    print("sweep", sweep, "active", active, flush=True)
    if active == 0: break
vals = np.sort(np.linalg.norm(G, axis=0)[:n] - shift) if mode != "x" else None
cn = np.linalg.norm(G, axis=0); idx = np.argsort(cn)[:n]
print("max err", np.abs(np.sort(cn[idx] - shift) - np.linalg.eigvalsh(H)).max())
