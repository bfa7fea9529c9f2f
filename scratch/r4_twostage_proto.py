"""numpy model of the two-stage tridiagonalisation (round 4): dense -> band by block Householder panels, band ->
tridiagonal by bulge chasing in tasks (sweep j, step k); the task order of the device kernel (one workgroup per sweep,
sweep j+1 trailing sweep j by LAG tasks) is simulated with random legal interleavings."""
import sys
import numpy as np


def house(x):
    """(v, tau, beta): (I - tau v v^T) x = beta e_1, v[0] = 1 (LAPACK dlarfg)."""
    alpha = x[0]
    xnorm = np.linalg.norm(x[1:])
    v = x.copy()
    if xnorm == 0.0:
        v[:] = 0.0
        v[0] = 1.0
        return v, 0.0, alpha
    beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def to_band(a, b):
    """A -> Q^T A Q with bandwidth b (lower + upper), panels of b columns."""
    a = a.copy()
    n = a.shape[0]
    for j0 in range(0, n - b - 1, b):
        r0 = j0 + b
        m = n - r0
        w = min(b, m - 1) if m > 1 else 0
        pan = a[r0:, j0:j0 + b]
        V = np.zeros((m, b))
        taus = np.zeros(b)
        for i in range(min(b, m - 1)):
            v, tau, beta = house(pan[i:, i].copy())
            taus[i] = tau
            V[i:, i] = v
            pan[i:, i:] -= tau * np.outer(v, v @ pan[i:, i:])
            pan[i + 1:, i] = 0.0
        a[j0:j0 + b, r0:] = pan.T
        # T: Q = I - V T V^T
        T = np.zeros((b, b))
        for i in range(b):
            T[i, i] = taus[i]
            if i > 0:
                T[:i, i] = -taus[i] * T[:i, :i] @ (V[:, :i].T @ V[:, i])
        A22 = a[r0:, r0:]
        X = A22 @ V @ T
        W = X - 0.5 * V @ (T.T @ (V.T @ X))
        a[r0:, r0:] = A22 - V @ W.T - W @ V.T
    return a


class Band:
    """lower band storage with room for the bulge: ab[c, d] = A[c + d, c], 0 <= d <= 2b"""

    def __init__(self, a, b):
        n = a.shape[0]
        self.n, self.b = n, b
        self.ab = np.zeros((n + 3 * b + 2, 2 * b + 1))
        for c in range(n):
            for d in range(0, min(b, n - 1 - c) + 1):
                self.ab[c, d] = a[c + d, c]

    def get(self, i, c):
        if i < c:
            i, c = c, i
        return self.ab[c, i - c] if i - c <= 2 * self.b else 0.0

    def block(self, r, c, h, w):
        return np.array([[self.get(r + i, c + k) for k in range(w)] for i in range(h)])

    def put_block(self, r, c, blk, sym=False):
        h, w = blk.shape
        for i in range(h):
            for k in range(w):
                ii, cc = r + i, c + k
                if sym and ii < cc:
                    continue
                assert ii >= cc and ii - cc <= 2 * self.b, (ii, cc)
                self.ab[cc, ii - cc] = blk[i, k]


def task(B, j, k):
    """step k of sweep j; returns False when the sweep has run off the matrix"""
    n, b = B.n, B.b
    if k == 0:
        q = j           # column whose entries below the first subdiagonal go
        r = j + 1       # first row of the reflector
        h = min(b, n - r)
        if h <= 1:
            return False
        x = np.array([B.get(r + i, q) for i in range(h)])
        v, tau, beta = house(x)
        B.put_block(r, q, np.concatenate(([beta], np.zeros(h - 1)))[:, None])
    else:
        q = j + 1 + (k - 1) * b   # first column of the bulge block
        r = q + b
        h = min(b, n - r)
        if h <= 1:
            return False
        w = min(b, n - q)
        G = B.block(r, q, h, w)
        v, tau, beta = house(G[:, 0].copy())
        G[:, 1:] -= tau * np.outer(v, v @ G[:, 1:])
        G[0, 0] = beta
        G[1:, 0] = 0.0
        B.put_block(r, q, G)
    # two-sided on the symmetric block of rows / columns r .. r+h-1
    S = B.block(r, r, h, h)
    p = tau * S @ v
    wv = p - 0.5 * tau * (p @ v) * v
    S = S - np.outer(v, wv) - np.outer(wv, v)
    B.put_block(r, r, S, sym=True)
    # from the right on the block below: rows r+h .. r+h+h2-1
    h2 = min(b, n - (r + h))
    if h2 > 0:
        G2 = B.block(r + h, r, h2, h)
        G2 -= tau * np.outer(G2 @ v, v)
        B.put_block(r + h, r, G2)
    return True


def tasks_of_sweep(n, b, j):
    count = 0
    if min(b, n - (j + 1)) > 1:
        count = 1
        k = 1
        while min(b, n - (j + 1 + (k - 1) * b + b)) > 1:
            count += 1
            k += 1
    return count


def chase(B, lag=None, rng=None):
    n, b = B.n, B.b
    counts = [tasks_of_sweep(n, b, j) for j in range(n - 2)]
    if lag is None:
        for j in range(n - 2):
            for k in range(counts[j]):
                assert task(B, j, k)
        return
    done = [0] * (n - 2)
    live = [j for j in range(n - 2) if counts[j] > 0]
    while live:
        ready = [j for j in live if j == 0 or done[j - 1] >= min(counts[j - 1], done[j] + lag)]
        j = ready[rng.integers(len(ready))]
        assert task(B, j, done[j])
        done[j] += 1
        if done[j] == counts[j]:
            live.remove(j)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60, 4)
    a = rng.standard_normal((n, n))
    a = a + a.T
    ref = np.linalg.eigvalsh(a)
    band = to_band(a, b)
    mask = np.abs(np.subtract.outer(np.arange(n), np.arange(n))) > b
    print("outside the band:", np.abs(band[mask]).max(), " eigenvalues of the band:", np.abs(np.linalg.eigvalsh(band) - ref).max())
    for lag in (None, 4, 3, 2, 1):
        B = Band(band, b)
        try:
            chase(B, lag, np.random.default_rng(1))
        except AssertionError as exc:
            print("lag", lag, "assertion", exc)
            continue
        d = B.ab[:n, 0]
        e = B.ab[:n - 1, 1]
        rest = np.abs(B.ab[:n, 2:]).max()
        t = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        print("lag", lag, "left outside the tridiagonal:", rest, " eigenvalues:", np.abs(np.linalg.eigvalsh(t) - ref).max())
