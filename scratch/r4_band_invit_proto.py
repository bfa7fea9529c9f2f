"""numpy model of the banded inverse iteration planned for the device (round 4): LU of B - lambda with partial pivoting in a
window of b + 1 rows x 2b + 1 columns (circular column frame, one slot per candidate row), forward solve over the slots,
backward solve over rows of U; eigenvectors of a symmetric band matrix from given eigenvalues."""
import sys
import numpy as np


def band_rows(a, b):
    n = a.shape[0]
    full = np.zeros((n + b + 2, 2 * b + 1))
    for i in range(n):
        for k in range(2 * b + 1):
            c = i - b + k
            if 0 <= c < n:
                full[i, k] = a[i, c]
    return full


def factor(full, n, b, lam, tiny):
    cw = 2 * b + 1
    W = np.zeros((b + 1, cw))
    rowid = np.arange(b + 1)
    for s in range(b + 1):
        i = s
        for col in range(0, 2 * b + 1):
            k = col - i + b
            if i < n and 0 <= k <= 2 * b:
                W[s, col % cw] = full[i, k] - (lam if col == i else 0.0)
    U = np.zeros((n, cw))
    Lm = np.zeros((n, b + 1))
    piv = np.zeros(n, dtype=int)
    for j in range(n):
        pj = j % cw
        v = np.where(rowid < n, W[:, pj], 0.0)
        p = int(np.argmax(np.abs(v)))
        pv = v[p]
        if abs(pv) < tiny:
            pv = -tiny if pv < 0 else tiny
        piv[j] = p
        lm = v / pv
        lm[p] = 0.0
        Lm[j] = lm
        cols = [(j + 1 + l) % cw for l in range(2 * b)]
        u = W[p, cols].copy()
        U[j, 0] = pv
        U[j, 1:] = u
        for s in range(b + 1):
            if s != p:
                W[s, cols] -= lm[s] * u
                W[s, pj] = 0.0  # (the position of column j is that of column j + 2b + 1 of the next frame)
        # the pivot's slot takes row j + b + 1 (frame j+1 .. j+2b+1)
        i_new = j + b + 1
        rowid[p] = i_new
        if i_new < n:
            for l in range(2 * b + 1):
                col = j + 1 + l
                W[p, col % cw] = full[i_new, l] - (lam if col == i_new else 0.0)
        else:
            W[p, :] = 0.0
    return U, Lm, piv


def solve(U, Lm, piv, n, b, rhs, tiny):
    cw = 2 * b + 1
    xs = np.zeros(b + 1)
    xs[: min(b + 1, n)] = rhs[: min(b + 1, n)]
    y = np.zeros(n)
    for j in range(n):
        p = piv[j]
        y[j] = xs[p]
        xs = xs - Lm[j] * y[j]
        xs[p] = rhs[j + b + 1] if j + b + 1 < n else 0.0
    x = np.zeros(n + cw)
    for j in range(n - 1, -1, -1):
        x[j] = (y[j] - U[j, 1:] @ x[j + 1 : j + 2 * b + 1]) / U[j, 0]
    return x[:n]


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (90, 5)
    a = rng.standard_normal((n, n))
    a = a + a.T
    a[np.abs(np.subtract.outer(np.arange(n), np.arange(n))) > b] = 0.0
    w, vref = np.linalg.eigh(a)
    full = band_rows(a, b)
    norm = np.abs(a).sum(axis=1).max()
    tiny = 2.2e-16 * norm
    worst_res = worst_ovl = 0.0
    for k in range(n):
        U, Lm, piv = factor(full, n, b, w[k], tiny)
        x = rng.uniform(-1, 1, n)
        for it in range(3):
            x /= np.linalg.norm(x)
            x = solve(U, Lm, piv, n, b, x, tiny)
        x /= np.linalg.norm(x)
        worst_res = max(worst_res, np.abs(a @ x - w[k] * x).max())
        worst_ovl = max(worst_ovl, 1 - abs(x @ vref[:, k]))
    print(f"n={n} b={b}: worst residual {worst_res:.2e}, worst 1 - |overlap| {worst_ovl:.2e}")


def _selftest():
    rng = np.random.default_rng(1)
    n, b = 40, 4
    a = rng.standard_normal((n, n)); a = a + a.T
    a[np.abs(np.subtract.outer(np.arange(n), np.arange(n))) > b] = 0.0
    lam = 0.3
    full = band_rows(a, b)
    U, Lm, piv = factor(full, n, b, lam, 1e-300)
    rhs = rng.standard_normal(n)
    x = solve(U, Lm, piv, n, b, rhs, 1e-300)
    print("solve error", np.abs((a - lam * np.eye(n)) @ x - rhs).max())
