// tridiag.hpp - all eigenvalues of the dense matrix without an external library: Householder
// tridiagonalisation on the GPU, then bisection on the tridiagonal matrix (Sturm counts).
// Part of the single translation unit bodge_hip.hip (included after dense.hpp).
//
// Why: free_energy's dense route (reference hamiltonian.py:282-321) and BASELINE config 5's check
// ("eigenvalues vs numpy.linalg.eigh to 1e-10") need eigenvalues only.  The Jacobi kernels (dense.hpp)
// stop at 4N = 4096 and rocSOLVER's 931 MB object takes minutes to arrive on a fresh machine; this
// path has no such wait and no size limit short of the n^2 matrix.  Eigenvectors (round 3): inverse iteration
// on the tridiagonal matrix and back-transformation through the stored reflectors, further down in this file
// (`bdg_eigh_dense_above`: the positive half only); rocSOLVER serves `bdg_eigh_dense` when ALL eigenvectors of a
// matrix above 2048 rows are asked for, and `BODGE_AMD_EIGH`.  Round 4: real matrices from 3000 / 5000 rows go
// through a band instead (twostage.hpp), entered from `eig_tridiagonal_typed` below.
//
// Algorithm (LAPACK's zhetd2, lower variant, restated for a row-major matrix that is updated lazily):
// the dense array the scatter kernels write is column-major H, read here as row-major B = H^T = conj(H),
// which is Hermitian with the same (real) spectrum.  For j = 0 .. n-2
//     x   = B'(j+1:n, j)                       = conj of row j right of the diagonal (contiguous)
//     v, tau, beta = Householder(x)            (I - tau v v^H)^H x = beta e_1;  e_j = beta, d_j = B'(j, j)
//     q   = B'(j+1:n, j+1:n) v                 one pass over the trailing block
//     p   = tau q;  w = p - (tau/2)(p^H v) v
//     B'' = B' - v w^H - w v^H                 NOT applied now: it is folded into the pass of step j+1,
// so every step reads and writes the trailing block exactly once ("fused pass": apply the pending
// rank-2 update of the previous step, store, and multiply by the new v) - 16 B (real) / 32 B (complex)
// per element and step, n^3/3 elements in total: HBM-bound, ~0.1 s at n = 3600 and 1-3 s at n = 10^4.
// Per step two launches: `td_vector_step` (one workgroup: finishes w of the previous step, builds the
// next Householder vector from the corrected row) and `td_fused_pass` (one wave per row).
#pragma once

namespace bdg {

__device__ inline double td_conj(double a) { return a; }
__device__ inline double2 td_conj(double2 a) { return make_double2(a.x, -a.y); }
__device__ inline double td_mul(double a, double b) { return a * b; }
__device__ inline double2 td_mul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
// (a real scalar times a pair of reals: the back-transformation of a real matrix walks two columns of Z per thread)
__device__ inline double2 td_mul(double a, double2 b) { return make_double2(a * b.x, a * b.y); }
__device__ inline double td_add(double a, double b) { return a + b; }
__device__ inline double2 td_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline double td_sub(double a, double b) { return a - b; }
__device__ inline double2 td_sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline double td_abs2(double a) { return a * a; }
__device__ inline double td_abs2(double2 a) { return fma(a.x, a.x, a.y * a.y); }
__device__ inline double td_re(double a) { return a; }
__device__ inline double td_re(double2 a) { return a.x; }
__device__ inline double td_im(double) { return 0.0; }
__device__ inline double td_im(double2 a) { return a.y; }
__device__ inline void td_set(double& out, double re, double) { out = re; }
__device__ inline void td_set(double2& out, double re, double im) { out = make_double2(re, im); }
__device__ inline double td_shfl(double a, int off) { return __shfl_xor(a, off); }
__device__ inline double2 td_shfl(double2 a, int off) { return make_double2(__shfl_xor(a.x, off), __shfl_xor(a.y, off)); }

// scalars of the step in flight (device memory): tau of the current Householder reflector
template <typename T>
struct TdScalars {
    T tau;
};

// workgroup-wide sum of a T over 1024 threads (fixed order: bit reproducible)
template <typename T>
__device__ inline T td_block_sum(T value, T* scratch) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int off = kWave / 2; off >= 1; off >>= 1) value = td_add(value, td_shfl(value, off));
    __syncthreads();  // (scratch may still be read from the previous sum)
    if (lane == 0) scratch[wave] = value;
    __syncthreads();
    T total;
    td_set(total, 0.0, 0.0);
    for (int k = 0; k < (int)blockDim.x / kWave; ++k) total = td_add(total, scratch[k]);
    return total;
}

// Reflector pairs (v_i, w_i) whose rank-2 update B - v w^H - w v^H has not been applied to the stored matrix yet.
// Applying every update at once (one read + one write of the trailing block per step) moves 16 B per real element and
// step; kept pending, a step only READS the block (its product with the new reflector, corrected by 2 P short dot
// products), and every kTdDefer-th step applies the pending pairs together: (kTdDefer + 1) / kTdDefer reads and 1 / kTdDefer
// writes per step - LAPACK's blocked sytrd / hetrd (latrd panels), restated for the lazily updated row-major matrix.
constexpr int kTdDefer = 4;

template <typename T>
struct TdPending {
    T* v[kTdDefer];
    T* w[kTdDefer];
    int count;  // finished pairs in slots 0 .. count-1
};

// One workgroup.  Step j: (i) finish w of step j-1 from q = (stored B) v_{j-1}, corrected for the pairs that were
// pending during that product, and file (v_{j-1}, w_{j-1}) as pending pair `pend.count`  (ii) row j of the matrix with
// every pending pair applied: d_j and x  (iii) the Householder vector of step j.  v_unf / q / the pending vectors are
// indexed by absolute row (>= j), v_new by absolute row (>= j+1).
// The reflector is also kept for the back-transformation of eigenvectors: v_j(j+2:) in row j of the matrix
// (dead from here on: later passes touch rows > j only), v_j(j+1) = 1 implied, tau_j in taus[j].
template <typename T>
__global__ __launch_bounds__(1024) void td_vector_step(T* __restrict__ a, int n, int j, const T* __restrict__ v_unf,
                                                       const T* __restrict__ q, TdPending<T> pend, T* __restrict__ v_new,
                                                       double* __restrict__ d, double* __restrict__ e, TdScalars<T>* scal,
                                                       T* __restrict__ taus, T* __restrict__ pend_dots) {
    __shared__ T scratch[16 * (2 * kTdDefer + 1)];
    __shared__ T row_v[kTdDefer], row_w[kTdDefer];
    __shared__ T alpha_slot;
    T zero;
    td_set(zero, 0.0, 0.0);
    const int np = pend.count;          // pairs pending while q was formed
    const int now = j > 0 ? np + 1 : 0;  // ... and once (v_{j-1}, w_{j-1}) has joined them
    // Round 4: the step is a chain of dependent round trips through L2 (it is one workgroup with a few entries per thread:
    // 25 us per column, 72 of the 92 ms of an n = 3600 solve).  While a thread's share of a vector fits kTdKeep registers
    // (n - j <= kTdKeep x 1024) what it has written stays in registers instead of being read back, row j of the matrix -
    // which depends on nothing this kernel computes - is asked for first of all, and alpha reaches the other threads through
    // LDS with the reduction's barrier instead of by a load of its own.
    constexpr bool kKeepable = sizeof(T) == sizeof(double);  // (complex entries: the kept values spill - 68-132 B of scratch; real only)
    constexpr int kTdKeep = kKeepable ? 4 : 1;
    const bool keep = kKeepable && n - j <= kTdKeep * (int)blockDim.x;  // (uniform)
    T row_keep[kTdKeep];
    if (keep) {
#pragma unroll
        for (int k = 0; k < kTdKeep; ++k) {
            const int c = j + 2 + (int)threadIdx.x + k * (int)blockDim.x;
            row_keep[k] = c < n ? a[(size_t)j * n + c] : zero;
        }
    }
    if (j > 0) {
        // (i) q_true = q - sum_i [ v_i (w_i^H v) + w_i (v_i^H v) ];  p = tau q_true;  w = p - (tau / 2) (p^H v) v   over rows j .. n-1
        const T tau = scal->tau;
        // (w_i^H v and v_i^H v for the pairs i < np: left behind by the step that made v - see (iii))
        T dots[2 * kTdDefer];
#pragma unroll
        for (int k = 0; k < 2 * kTdDefer; ++k) dots[k] = k < 2 * np ? pend_dots[k] : zero;
        T* w_out = pend.w[np];
        T* v_out = pend.v[np];
        T dot = zero;
        T p_keep[kTdKeep], v_keep[kTdKeep];
        int slot = 0;
        for (int r = j + threadIdx.x; r < n; r += blockDim.x, ++slot) {
            T qt = q[r];
#pragma unroll
            for (int i = 0; i < kTdDefer; ++i)
                if (i < np) qt = td_sub(qt, td_add(td_mul(pend.v[i][r], dots[2 * i]), td_mul(pend.w[i][r], dots[2 * i + 1])));
            const T p = td_mul(tau, qt);
            const T x = v_unf[r];
            if (keep) {
#pragma unroll
                for (int k = 0; k < kTdKeep; ++k)
                    if (k == slot) p_keep[k] = p, v_keep[k] = x;
            } else {
                w_out[r] = p;  // (finished below)
            }
            dot = td_add(dot, td_mul(td_conj(p), x));
        }
        dot = td_block_sum(dot, scratch);
        T half_tau;
        td_set(half_tau, -0.5 * td_re(tau), -0.5 * td_im(tau));
        const T alpha2 = td_mul(half_tau, dot);
        if (keep) {
#pragma unroll
            for (int k = 0; k < kTdKeep; ++k) {
                const int r = j + (int)threadIdx.x + k * (int)blockDim.x;
                if (r < n) {
                    w_out[r] = td_add(p_keep[k], td_mul(alpha2, v_keep[k]));
                    v_out[r] = v_keep[k];
                }
            }
        } else {
            for (int r = j + threadIdx.x; r < n; r += blockDim.x) {
                const T x = v_unf[r];
                w_out[r] = td_add(w_out[r], td_mul(alpha2, x));
                v_out[r] = x;
            }
        }
        __syncthreads();
    }
    // (ii) row j with every pending update: B'(j, c) = B(j, c) - sum_i [ v_i[j] conj(w_i[c]) + w_i[j] conj(v_i[c]) ]
    if (threadIdx.x < kTdDefer && (int)threadIdx.x < now) {
        row_v[threadIdx.x] = pend.v[threadIdx.x][j];
        row_w[threadIdx.x] = pend.w[threadIdx.x][j];
    }
    __syncthreads();
    auto entry = [&](int c) {
        T value = a[(size_t)j * n + c];
#pragma unroll
        for (int i = 0; i < kTdDefer; ++i)
            if (i < now) value = td_sub(value, td_add(td_mul(row_v[i], td_conj(pend.w[i][c])), td_mul(row_w[i], td_conj(pend.v[i][c]))));
        return value;
    };
    if (threadIdx.x == 0) d[j] = td_re(entry(j));
    const int m = n - j - 1;
    if (m == 0) return;
    // (iii) x_c = conj(B'(j, c)), c = j+1 .. n-1;  alpha = x_{j+1}.  The same loop takes w_i^H x and v_i^H x of the
    // pending pairs (their entries are loaded for the row anyway): what step j+1 corrects q with.
    T sums[2 * kTdDefer + 1];  // [2 i] = w_i^H x, [2 i + 1] = v_i^H x over c >= j+2, [2 K] = |x|^2
#pragma unroll
    for (int k = 0; k < 2 * kTdDefer + 1; ++k) sums[k] = zero;
    double norm2 = 0.0;
    T x_keep[kTdKeep];
    int xslot = 0;
    for (int c = j + 2 + threadIdx.x; c < n; c += blockDim.x, ++xslot) {
        T value = zero;
        if (keep) {
#pragma unroll
            for (int k = 0; k < kTdKeep; ++k)
                if (k == xslot) value = row_keep[k];
        } else {
            value = a[(size_t)j * n + c];
        }
        T wc[kTdDefer], vc[kTdDefer];
#pragma unroll
        for (int i = 0; i < kTdDefer; ++i)
            if (i < now) {
                wc[i] = td_conj(pend.w[i][c]);
                vc[i] = td_conj(pend.v[i][c]);
                value = td_sub(value, td_add(td_mul(row_v[i], wc[i]), td_mul(row_w[i], vc[i])));
            }
        const T x = td_conj(value);
        if (keep) {
#pragma unroll
            for (int k = 0; k < kTdKeep; ++k)
                if (k == xslot) x_keep[k] = x;
        } else {
            v_new[c] = x;  // (scaled below)
        }
        norm2 += td_abs2(x);
#pragma unroll
        for (int i = 0; i < kTdDefer; ++i)
            if (i < now) {
                sums[2 * i] = td_add(sums[2 * i], td_mul(wc[i], x));
                sums[2 * i + 1] = td_add(sums[2 * i + 1], td_mul(vc[i], x));
            }
    }
    td_set(sums[2 * kTdDefer], norm2, 0.0);
    if (threadIdx.x == 0) alpha_slot = td_conj(entry(j + 1));  // (read by everyone after the barriers of the reduction below)
    {   // wave sums of the 2 `now` + 1 values in use, one row of `scratch` per wave; every thread adds up |x|^2, thread k
        // (below) the k-th dot product: fixed order, bit reproducible
        constexpr int N = 2 * kTdDefer + 1;
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (k < 2 * now || k == N - 1)
                for (int off = kWave / 2; off >= 1; off >>= 1) sums[k] = td_add(sums[k], td_shfl(sums[k], off));
        __syncthreads();  // (scratch may still be read from the sum in (i))
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < N; ++k)
                if (k < 2 * now || k == N - 1) scratch[wave * N + k] = sums[k];
        __syncthreads();
        double total = 0.0;
        for (int w = 0; w < (int)blockDim.x / kWave; ++w) total += td_re(scratch[w * N + N - 1]);
        norm2 = total;
    }
    const T alpha = alpha_slot;
    if (norm2 == 0.0 && td_im(alpha) == 0.0) {  // nothing to annihilate: H = I
        if (threadIdx.x == 0) {
            e[j] = td_re(alpha);
            scal->tau = zero;
            taus[j] = zero;
            td_set(v_new[j + 1], 1.0, 0.0);
        }
        if ((int)threadIdx.x < 2 * now)  // v = e_{j+1}; tau = 0 makes the next w vanish whatever these are
            pend_dots[threadIdx.x] = td_conj((threadIdx.x & 1) ? pend.v[threadIdx.x >> 1][j + 1] : pend.w[threadIdx.x >> 1][j + 1]);
        __syncthreads();  // (every thread has read row j by now)
        for (int c = j + 2 + threadIdx.x; c < n; c += blockDim.x) {
            a[(size_t)j * n + c] = zero;
            v_new[c] = zero;  // (x, all zeros here, may have stayed in registers)
        }
        return;
    }
    const double ar = td_re(alpha), ai = td_im(alpha);
    const double beta = -copysign(sqrt(fma(ar, ar, fma(ai, ai, norm2))), ar);
    // scale = 1 / (alpha - beta)
    const double sr = ar - beta, si = ai, den = fma(sr, sr, si * si);
    T scale;
    td_set(scale, sr / den, -si / den);
    __syncthreads();  // (v_new written above by the same threads that rescale it: same index set, no hazard; keep order explicit)
    if (keep) {
#pragma unroll
        for (int k = 0; k < kTdKeep; ++k) {
            const int c = j + 2 + (int)threadIdx.x + k * (int)blockDim.x;
            if (c < n) {
                const T scaled = td_mul(x_keep[k], scale);
                v_new[c] = scaled;
                a[(size_t)j * n + c] = scaled;
            }
        }
    } else {
        for (int c = j + 2 + threadIdx.x; c < n; c += blockDim.x) {
            const T scaled = td_mul(v_new[c], scale);
            v_new[c] = scaled;
            a[(size_t)j * n + c] = scaled;  // (row j: every thread read its entries before the barrier above)
        }
    }
    if (threadIdx.x == 0) {
        e[j] = beta;
        T tau;
        td_set(tau, (beta - ar) / beta, -ai / beta);
        scal->tau = tau;
        taus[j] = tau;
        td_set(v_new[j + 1], 1.0, 0.0);
    }
    // v = (1, scale x): w_i^H v = conj(w_i[j+1]) + scale (w_i^H x)
    if ((int)threadIdx.x < 2 * now) {
        const int k = threadIdx.x;
        T total = zero;
        for (int w = 0; w < (int)blockDim.x / kWave; ++w) total = td_add(total, scratch[w * (2 * kTdDefer + 1) + k]);
        const T head = td_conj((k & 1) ? pend.v[k >> 1][j + 1] : pend.w[k >> 1][j + 1]);
        pend_dots[k] = td_add(head, td_mul(scale, total));
    }
}

// Trailing block rows / columns j+1 .. n-1: q[r] = sum_c B(r, c) v_new[c] with the stored B.  P = 0: read only.
// P > 0: the P pending pairs are applied first and the block stored (they then count as applied).  One wave per row;
// 16 bytes per lane and access: a complex entry, or two real ones (n = 4 nb is even, so rows start 16-byte aligned
// and only the first column of an odd j+1 stands alone).
// A wave takes ROWS consecutive rows at a time: the chunk of the pending vectors (2 P + 1 loads of 16 B per lane)
// then serves that many rows - with one row per wave the pass that applies four pairs moved four times the matrix
// bytes through L2 for the vectors alone (196 us per pass at n = 10^4 against 168 with four rows).  The read-only and
// one-pair passes keep one row per wave: more waves in flight (53 against 79 us).
constexpr int td_rows_for(int pending) { return pending >= 4 ? 4 : pending >= 2 ? 2 : 1; }

template <typename T, int P, int ROWS = td_rows_for(P)>
__global__ __launch_bounds__(256) void td_fused_pass(T* __restrict__ a, int n, int j, TdPending<T> pend, const T* __restrict__ v_new,
                                                     T* __restrict__ q) {
    const int lane = threadIdx.x & (kWave - 1);
    const int waves = gridDim.x * (blockDim.x / kWave);
    const int wave_id = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    for (int r0 = j + 1 + wave_id * ROWS; r0 < n; r0 += waves * ROWS) {
        T vr[ROWS][kTdDefer], wr[ROWS][kTdDefer], acc[ROWS];
        T* row[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const int r = min(r0 + k, n - 1);  // (rows past the end repeat the last one and are not stored)
            row[k] = a + (size_t)r * n;
            td_set(acc[k], 0.0, 0.0);
#pragma unroll
            for (int i = 0; i < kTdDefer; ++i) {
                td_set(vr[k][i], 0.0, 0.0);
                td_set(wr[k][i], 0.0, 0.0);
                if (i < P) vr[k][i] = pend.v[i][r], wr[k][i] = pend.w[i][r];
            }
        }
        const int live = min(ROWS, n - r0);
        if constexpr (sizeof(T) == sizeof(double)) {
            int c0 = j + 1;
            if (c0 & 1) {  // lone first column
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < ROWS; ++k)
                        if (k < live) {
                            T value = row[k][c0];
#pragma unroll
                            for (int i = 0; i < P; ++i) value -= vr[k][i] * pend.w[i][c0] + wr[k][i] * pend.v[i][c0];
                            if (P > 0) row[k][c0] = value;
                            acc[k] = value * v_new[c0];
                        }
                }
                ++c0;
            }
            for (int c = c0 + 2 * lane; c < n; c += 2 * kWave) {  // (n even: c + 1 < n)
                double2 wc[kTdDefer], vc[kTdDefer];
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    wc[i] = *reinterpret_cast<const double2*>(pend.w[i] + c);
                    vc[i] = *reinterpret_cast<const double2*>(pend.v[i] + c);
                }
                const double2 vn = *reinterpret_cast<const double2*>(v_new + c);
                // (rows past the end repeat the last one: their loads are safe and sit on every path - no branch around a load,
                // so the compiler keeps the other rows' loads in flight while it waits for one (counted vmcnt); only stores are guarded)
                double2 pair[ROWS];
#pragma unroll
                for (int k = 0; k < ROWS; ++k) pair[k] = *reinterpret_cast<const double2*>(row[k] + c);
#pragma unroll
                for (int k = 0; k < ROWS; ++k) {
#pragma unroll
                    for (int i = 0; i < P; ++i) {
                        pair[k].x -= vr[k][i] * wc[i].x + wr[k][i] * vc[i].x;
                        pair[k].y -= vr[k][i] * wc[i].y + wr[k][i] * vc[i].y;
                    }
                    if (P > 0 && k < live) *reinterpret_cast<double2*>(row[k] + c) = pair[k];
                    acc[k] = fma(pair[k].x, vn.x, fma(pair[k].y, vn.y, acc[k]));
                }
            }
        } else {
            for (int c = j + 1 + lane; c < n; c += kWave) {
                T wc[kTdDefer], vc[kTdDefer];
#pragma unroll
                for (int i = 0; i < P; ++i) wc[i] = td_conj(pend.w[i][c]), vc[i] = td_conj(pend.v[i][c]);
                const T vn = v_new[c];
                T value[ROWS];
#pragma unroll
                for (int k = 0; k < ROWS; ++k) value[k] = row[k][c];
#pragma unroll
                for (int k = 0; k < ROWS; ++k) {
#pragma unroll
                    for (int i = 0; i < P; ++i) value[k] = td_sub(value[k], td_add(td_mul(vr[k][i], wc[i]), td_mul(wr[k][i], vc[i])));
                    if (P > 0 && k < live) row[k][c] = value[k];
                    acc[k] = td_add(acc[k], td_mul(value[k], vn));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            for (int off = kWave / 2; off >= 1; off >>= 1) acc[k] = td_add(acc[k], td_shfl(acc[k], off));
            if (lane == 0 && k < live) q[r0 + k] = acc[k];
        }
    }
}

// k-th smallest eigenvalue of the symmetric tridiagonal matrix (d, e) by multisection on the Sturm count
// (number of negative pivots of T - x I), ascending output.  The Sturm recurrence is one dependent division per row:
// a lone thread per eigenvalue is a chain of 53 x n of them, with most SIMDs of the device idle at n = 10^4.  So
// kBisectPoints = 16 adjacent lanes share an eigenvalue: each evaluates the count at its own point of the current
// interval, which then shrinks to one of seventeen parts - 13 rounds instead of 53 (58 -> 15 ms at n = 10^4).
constexpr int kBisectPoints = 16;
__global__ __launch_bounds__(64) void td_bisect(const double* __restrict__ d, const double* __restrict__ e2, int n, double lo, double hi,
                                                double pivmin, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, point = lane % kBisectPoints, group = lane / kBisectPoints;
    const int k = blockIdx.x * (64 / kBisectPoints) + group;  // (k >= n: the lanes work on the last eigenvalue again and store nothing)
    const int kk = min(k, n - 1);
    double a = lo, b = hi;
    for (int it = 0; it < 64; ++it) {
        const double width = b - a;
        const double x = a + width * ((point + 1) * (1.0 / (kBisectPoints + 1)));
        // (an interval of a few ulps: some points coincide with its ends, which is harmless)
        double piv = d[0] - x;
        int count = piv < 0.0 ? 1 : 0;
        for (int i = 1; i < n; ++i) {
            if (fabs(piv) < pivmin) piv = -pivmin;
            piv = d[i] - x - e2[i - 1] / piv;
            count += piv < 0.0 ? 1 : 0;
        }
        // the part between the last point with count <= k and the first with count > k
        const bool above = count > kk;
        const unsigned long long votes = __ballot(above) >> (group * kBisectPoints) & ((1ull << kBisectPoints) - 1);
        const int first_above = votes ? __builtin_ctzll(votes) : kBisectPoints;  // points are ascending, counts monotone
        const double new_a = first_above == 0 ? a : __shfl(x, group * kBisectPoints + first_above - 1);
        const double new_b = first_above == kBisectPoints ? b : __shfl(x, group * kBisectPoints + first_above);
        if (!(new_b - new_a < width)) break;  // nothing gained any more: the interval is down to rounding
        a = new_a;
        b = new_b;
    }
    if (point == 0 && k < n) out[k] = 0.5 * (a + b);
}

// ---- eigenvectors of the tridiagonal matrix by inverse iteration (the scheme of LAPACK's dstein).
// One wave per cluster of close eigenvalues; its members are treated one after the other: LU of T - λ
// with partial pivoting (lane 0; scratch in global memory), three solves from a pseudo-random start,
// each followed by an orthogonalisation against the members already done and a normalisation (all
// lanes).  Singletons - almost every eigenvalue - need no orthogonalisation: eigenvalues further apart
// than the cluster tolerance give vectors orthogonal to ε|T| / gap.  Output z[i * ld + (k - first_index)].
struct TdCluster {
    int first, count;  // positions in the ascending eigenvalue array
};

__device__ inline double td_wave_sum(double v) {
    for (int off = kWave / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// The recurrences (LU, forward and backward substitution) are sequential in the row index.  They run
// REDUNDANTLY ON ALL 64 LANES with wave-uniform values: a tile of 64 rows is loaded with one coalesced
// load per array, row i of the tile is fetched from lane i's register with v_readlane (no memory access
// on the dependent chain - the first version, lane 0 walking global memory, spent 1.6 s of a 6 s solve
// at n = 10^4 here), lane i keeps the outputs of step i, and the tile is stored with coalesced stores.
__device__ inline double td_from_lane(double v, int src) {  // src wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// One wave per eigenvalue (`shift[k]`: the eigenvalue, members of a cluster a few ulp |T| apart so that their
// iterations decorrelate).  No orthogonalisation here: clusters are orthonormalised afterwards by
// td_cluster_orthonormalise - the members of a degenerate level (50 of them on a 50 x 50 lattice) need not
// wait for each other.
__global__ __launch_bounds__(64) void td_inverse_iteration(const double* __restrict__ d, const double* __restrict__ e,
                                                           int n, const double* __restrict__ shift, int n_vec,
                                                           double norm_t, double* __restrict__ scratch,
                                                           double* __restrict__ z, int ld) {
    const int lane = threadIdx.x;
    double* lu_a = scratch + (size_t)blockIdx.x * 6 * n;  // pivots
    double* lu_b = lu_a + n;                              // first super-diagonal of U
    double* lu_c = lu_b + n;                              // multipliers
    double* lu_d = lu_c + n;                              // second super-diagonal of U
    double* lu_in = lu_d + n;                             // 1.0 where rows k, k+1 were interchanged
    double* x = lu_in + n;
    const double eps = 2.220446049250313e-16;
    const double tiny = eps * norm_t;
    {
        for (int k_eig = blockIdx.x; k_eig < n_vec; k_eig += gridDim.x) {
            const double lambda = shift[k_eig];
            // ---- LU factorisation of T - lambda with partial pivoting (LAPACK dlagtf): a = pivots, b / d = first and
            // second super-diagonal of U, c = multipliers, in = 1 where rows k, k+1 were interchanged
            {
                double a_k = d[0] - lambda;           // pivot candidate of the current row
                double b_k = n > 1 ? e[0] : 0.0;      // its super-diagonal entry (changed by an interchange above it)
                double scale1 = fabs(a_k) + fabs(b_k);
                for (int t = 0; t < n; t += kWave) {
                    const int row = t + lane;
                    const double sub_l = row < n - 1 ? e[row] : 0.0;                    // c(row): sub-diagonal below row
                    const double a1_l = row + 1 < n ? d[row + 1] - lambda : 0.0;        // diagonal of row + 1
                    const double b1_l = row + 1 < n - 1 ? e[row + 1] : 0.0;             // super-diagonal of row + 1
                    double out_a = 0.0, out_b = 0.0, out_c = 0.0, out_d = 0.0, out_in = 0.0;
                    const int steps = min(kWave, n - t);
                    for (int i = 0; i < steps; ++i) {
                        const int k = t + i;
                        double oa = a_k, ob = b_k, oc = 0.0, od = 0.0, oin = 0.0;
                        if (k < n - 1) {
                            const double sub = td_from_lane(sub_l, i), a1 = td_from_lane(a1_l, i), b1 = td_from_lane(b1_l, i);
                            const double scale2 = fabs(sub) + fabs(a1) + fabs(b1);
                            const double piv1 = a_k == 0.0 ? 0.0 : fabs(a_k) / scale1;
                            if (sub == 0.0) {
                                scale1 = scale2;
                                a_k = a1;
                                b_k = b1;
                            } else if (fabs(sub) / scale2 <= piv1) {  // no interchange
                                scale1 = scale2;
                                oc = sub / a_k;
                                a_k = a1 - oc * b_k;
                                b_k = b1;
                            } else {  // rows k and k+1 interchanged
                                oin = 1.0;
                                oc = a_k / sub;
                                oa = sub;
                                ob = a1;
                                od = b1;
                                a_k = b_k - oc * a1;
                                b_k = -oc * b1;
                            }
                        }
                        if (lane == i) out_a = oa, out_b = ob, out_c = oc, out_d = od, out_in = oin;
                    }
                    if (row < n) lu_a[row] = out_a, lu_b[row] = out_b, lu_c[row] = out_c, lu_d[row] = out_d, lu_in[row] = out_in;
                }
            }
            for (int i = lane; i < n; i += kWave) {  // pseudo-random start in (-1, 1)
                const uint64_t h = splitmix64(((uint64_t)(k_eig + 1) << 32) ^ (uint64_t)i);
                x[i] = (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
            }
            __syncthreads();
            // (five solves, dstein's MAXITS: three left one vector in ~250 random periodic lattices with 1e-7 of a level
            // 4e-5 |T| away - residual 4e-11, overlap 4e-8; scratch/r3_cluster_diag.py)
            for (int iteration = 0; iteration < 5; ++iteration) {
                double norm2 = 0.0;
                for (int i = lane; i < n; i += kWave) norm2 += x[i] * x[i];
                norm2 = td_wave_sum(norm2);
                const double scale = norm2 > 0.0 ? 1.0 / sqrt(norm2) : 1.0;
                // ---- forward: the row operations of the factorisation (dlagts); the scaling is applied on the way
                {
                    double r = x[0] * scale;  // x(k) as modified so far
                    for (int t = 0; t < n; t += kWave) {
                        const int row = t + lane;
                        const double next_l = row + 1 < n ? x[row + 1] * scale : 0.0;
                        const double c_l = row < n ? lu_c[row] : 0.0, in_l = row < n ? lu_in[row] : 0.0;
                        double out = 0.0;
                        const int steps = min(kWave, n - t);
                        for (int i = 0; i < steps; ++i) {
                            const int k = t + i;
                            double o = r;
                            if (k < n - 1) {
                                const double next = td_from_lane(next_l, i), c = td_from_lane(c_l, i);
                                if (td_from_lane(in_l, i) == 0.0) r = next - c * r;
                                else {
                                    o = next;
                                    r = r - c * next;
                                }
                            }
                            if (lane == i) out = o;
                        }
                        __syncthreads();  // (every lane has read x(row + 1) of this tile before x(row) of the next is ... x(t + 64) is rewritten)
                        if (row < n) x[row] = out;
                    }
                }
                __syncthreads();
                // ---- backward: U y = x, tiny pivots perturbed (dlagts, job = -1)
                {
                    double y1 = 0.0, y2 = 0.0;  // y(k+1), y(k+2)
                    for (int t = ((n - 1) / kWave) * kWave; t >= 0; t -= kWave) {
                        const int row = t + lane;
                        const double x_l = row < n ? x[row] : 0.0, a_l = row < n ? lu_a[row] : 1.0;
                        const double b_l = row < n ? lu_b[row] : 0.0, d_l = row < n ? lu_d[row] : 0.0;
                        double out = 0.0;
                        for (int i = min(kWave, n - t) - 1; i >= 0; --i) {
                            const int k = t + i;
                            double value = td_from_lane(x_l, i);
                            if (k + 1 < n) value -= td_from_lane(b_l, i) * y1;
                            if (k + 2 < n) value -= td_from_lane(d_l, i) * y2;
                            double pivot = td_from_lane(a_l, i);
                            if (fabs(pivot) < tiny) pivot = pivot < 0.0 ? -tiny : tiny;
                            const double yk = value / pivot;
                            y2 = y1;
                            y1 = yk;
                            if (lane == i) out = yk;
                        }
                        if (row < n) x[row] = out;
                    }
                }
                __syncthreads();
            }
            double norm2 = 0.0;
            for (int i = lane; i < n; i += kWave) norm2 += x[i] * x[i];
            norm2 = td_wave_sum(norm2);
            const double scale = norm2 > 0.0 ? 1.0 / sqrt(norm2) : 1.0;
            for (int i = lane; i < n; i += kWave) z[(size_t)i * ld + k_eig] = x[i] * scale;
            __syncthreads();
        }
    }
}

// Gram-Schmidt with re-orthogonalisation over the columns first .. first+count-1 of z (one workgroup per cluster of
// close eigenvalues): an orthonormal basis of the cluster's invariant subspace.  Distinct-but-close members overlap by
// eps |T| / gap only and are hardly changed; the members of a degenerate level come out as some orthonormal basis of it.
// Member m is orthogonalised against its predecessors eight at a time (one pass over the rows and one workgroup
// reduction per eight: a flat band of a thousand-fold level stays affordable), twice, then normalised.
__global__ __launch_bounds__(256) void td_cluster_orthonormalise(const TdCluster* __restrict__ clusters, int n,
                                                                 double* __restrict__ z, int ld) {
    constexpr int kAtOnce = 8;
    __shared__ double red[4][kAtOnce];
    __shared__ double dots[kAtOnce];
    const TdCluster cluster = clusters[blockIdx.x];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int m = 0; m < cluster.count; ++m) {
        double* zm = z + cluster.first + m;
        for (int pass = 0; pass < 2; ++pass)
            for (int p0 = 0; p0 < m; p0 += kAtOnce) {
                const int np = min(kAtOnce, m - p0);
                const double* zp = z + cluster.first + p0;
                double acc[kAtOnce];
#pragma unroll
                for (int q = 0; q < kAtOnce; ++q) acc[q] = 0.0;
                for (int i = threadIdx.x; i < n; i += blockDim.x) {
                    const double value = zm[(size_t)i * ld];
#pragma unroll
                    for (int q = 0; q < kAtOnce; ++q)
                        if (q < np) acc[q] = fma(value, zp[(size_t)i * ld + q], acc[q]);
                }
#pragma unroll
                for (int q = 0; q < kAtOnce; ++q) {
                    const double total = td_wave_sum(acc[q]);
                    if (lane == 0) red[wave][q] = total;
                }
                __syncthreads();
                if ((int)threadIdx.x < kAtOnce) dots[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
                __syncthreads();
                for (int i = threadIdx.x; i < n; i += blockDim.x) {
                    double value = zm[(size_t)i * ld];
#pragma unroll
                    for (int q = 0; q < kAtOnce; ++q)
                        if (q < np) value = fma(-dots[q], zp[(size_t)i * ld + q], value);
                    zm[(size_t)i * ld] = value;
                }
                __syncthreads();
            }
        double norm2 = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) norm2 = fma(zm[(size_t)i * ld], zm[(size_t)i * ld], norm2);
        norm2 = td_wave_sum(norm2);
        if (lane == 0) red[wave][0] = norm2;
        __syncthreads();
        norm2 = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        __syncthreads();
        const double scale = norm2 > 0.0 ? 1.0 / sqrt(norm2) : 1.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) zm[(size_t)i * ld] *= scale;
        __syncthreads();
    }
}

// ---- back-transformation: eigenvectors of B = Q T Q^H are Q z, Q = H_0 H_1 ... H_{n-2}, H_l = I - tau_l v_l v_l^H
// (v_l(r) = 0 for r <= l, 1 at r = l+1, stored in row l of the matrix beyond that).  The reflectors are applied
// last first, kTdGroup at a time per pass over Z: for the group l_0 > l_1 > ... (l_i = hi - i) with column sums
// s_i = v_{l_i}^H Z taken on Z as it is BEFORE the group,
//     g_0 = tau_{l_0} s_0,   g_i = tau_{l_i} (s_i - sum_{m<i} (v_{l_i}^H v_{l_m}) g_m),   Z <- Z - sum_i v_{l_i} g_i
// (the compact WY form written out), and the same pass takes, on the values it has just written, the sums the
// NEXT group needs: one read and one write of Z per kTdGroup reflectors.
constexpr int kTdGroup = 4;

template <typename T>
__device__ inline T td_reflector_entry(const T* __restrict__ a, int n, int l, int r) {
    T v;
    td_set(v, r == l + 1 ? 1.0 : 0.0, 0.0);
    if (r > l + 1) v = a[(size_t)l * n + r];
    return v;
}

// cross[i][m] = v_{l_i}^H v_{l_m} for the reflectors l_i = hi - i, i < count (one workgroup)
template <typename T>
__global__ __launch_bounds__(256) void td_reflector_gram(const T* __restrict__ a, int n, int hi, int count, T* __restrict__ cross) {
    __shared__ T scratch[4];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int i = 1; i < count; ++i)
        for (int m = 0; m < i; ++m) {
            T acc;
            td_set(acc, 0.0, 0.0);
            for (int r = hi - m + 1 + threadIdx.x; r < n; r += blockDim.x)  // support of the higher reflector l_m
                acc = td_add(acc, td_mul(td_conj(td_reflector_entry(a, n, hi - i, r)), td_reflector_entry(a, n, hi - m, r)));
            for (int off = kWave / 2; off >= 1; off >>= 1) acc = td_add(acc, td_shfl(acc, off));
            __syncthreads();
            if (lane == 0) scratch[wave] = acc;
            __syncthreads();
            if (threadIdx.x == 0) cross[i * kTdGroup + m] = td_add(td_add(scratch[0], scratch[1]), td_add(scratch[2], scratch[3]));
        }
}

// column sums of the first group, on Z as it comes from the inverse iteration: partial[chunk][i][c]
// Z = the type a thread handles of a row of Z: T itself, or for real matrices double2 = two adjacent real columns
// (16-byte accesses; n_vec and ld then count pairs).
template <typename T, typename Z>
__global__ __launch_bounds__(256) void td_reflect_sums(const T* __restrict__ a, int n, int hi, int count, const Z* __restrict__ z,
                                                       int ld, int n_vec, int row0, int rows_per_chunk, Z* __restrict__ partial) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_vec) return;
    const int r0 = row0 + blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    Z acc[kTdGroup];
#pragma unroll
    for (int i = 0; i < kTdGroup; ++i) td_set(acc[i], 0.0, 0.0);
    for (int r = r0; r < r1; ++r) {
        const Z value = z[(size_t)r * ld + c];
#pragma unroll
        for (int i = 0; i < kTdGroup; ++i)
            if (i < count) acc[i] = td_add(acc[i], td_mul(td_conj(td_reflector_entry(a, n, hi - i, r)), value));
    }
#pragma unroll
    for (int i = 0; i < kTdGroup; ++i) partial[((size_t)blockIdx.y * kTdGroup + i) * n_vec + c] = acc[i];
}

// The group hi, hi-1, ... (count of them) applied to rows row0 .. n-1 (row0 = hi - count + 2) and the sums of the next
// group (next_hi = hi - count, next_count reflectors) taken on the result; the rows next_row0 .. row0 - 1 that only the
// next group touches are added by chunk 0.
template <typename T, typename Z>
__global__ __launch_bounds__(256) void td_reflect_group(const T* __restrict__ a, int n, int hi, int count, const T* __restrict__ taus,
                                                        const T* __restrict__ cross, Z* __restrict__ z, int ld, int n_vec,
                                                        int row0, int rows_per_chunk, int summed_chunks, const Z* __restrict__ partial,
                                                        int next_count, int next_row0, Z* __restrict__ next_partial) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_vec) return;
    Z g[kTdGroup], acc[kTdGroup];
#pragma unroll
    for (int i = 0; i < kTdGroup; ++i) {
        td_set(g[i], 0.0, 0.0);
        td_set(acc[i], 0.0, 0.0);
        if (i < count) {
            Z s;
            td_set(s, 0.0, 0.0);
            for (int chunk = 0; chunk < summed_chunks; ++chunk) s = td_add(s, partial[((size_t)chunk * kTdGroup + i) * n_vec + c]);
#pragma unroll
            for (int m = 0; m < kTdGroup; ++m)
                if (m < i) s = td_sub(s, td_mul(cross[i * kTdGroup + m], g[m]));
            g[i] = td_mul(taus[hi - i], s);
        }
    }
    const int next_hi = hi - count;
    if (next_count > 0 && blockIdx.y == 0)
        for (int r = next_row0; r < row0; ++r) {  // rows below this group's reach
            const Z value = z[(size_t)r * ld + c];
#pragma unroll
            for (int i = 0; i < kTdGroup; ++i)
                if (i < next_count) acc[i] = td_add(acc[i], td_mul(td_conj(td_reflector_entry(a, n, next_hi - i, r)), value));
        }
    const int r0 = row0 + blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    // four rows at a time: their loads of Z are in flight together (a thread walks one column: with one load per
    // iteration the pass ran at 1.3 TB/s)
    constexpr int kRows = 8;
    int r = r0;
    for (; r + kRows <= r1; r += kRows) {
        Z value[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) value[k] = z[(size_t)(r + k) * ld + c];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
#pragma unroll
            for (int i = 0; i < kTdGroup; ++i)
                if (i < count) value[k] = td_sub(value[k], td_mul(td_reflector_entry(a, n, hi - i, r + k), g[i]));
            z[(size_t)(r + k) * ld + c] = value[k];
        }
#pragma unroll
        for (int k = 0; k < kRows; ++k)  // (row order kept: the sums do not depend on the grouping)
#pragma unroll
            for (int i = 0; i < kTdGroup; ++i)
                if (i < next_count) acc[i] = td_add(acc[i], td_mul(td_conj(td_reflector_entry(a, n, next_hi - i, r + k)), value[k]));
    }
    for (; r < r1; ++r) {
        Z value = z[(size_t)r * ld + c];
#pragma unroll
        for (int i = 0; i < kTdGroup; ++i)
            if (i < count) value = td_sub(value, td_mul(td_reflector_entry(a, n, hi - i, r), g[i]));
        z[(size_t)r * ld + c] = value;
#pragma unroll
        for (int i = 0; i < kTdGroup; ++i)
            if (i < next_count) acc[i] = td_add(acc[i], td_mul(td_conj(td_reflector_entry(a, n, next_hi - i, r)), value));
    }
    if (next_count > 0)
#pragma unroll
        for (int i = 0; i < kTdGroup; ++i) next_partial[((size_t)blockIdx.y * kTdGroup + i) * n_vec + c] = acc[i];
}

// real tridiagonal eigenvectors -> the arithmetic of the back-transformation
__global__ void td_widen(const double* __restrict__ in, double2* __restrict__ out, int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = make_double2(in[i], 0.0);
}
// (real: rows padded to an even number of columns - `ld` - so that a thread of the back-transformation takes two
// columns with one 16-byte access; the padding column is zero)
__global__ void td_widen(const double* __restrict__ in, double* __restrict__ out, int64_t rows, int64_t n_vec, int64_t ld) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows * ld; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld, c = i - r * ld;
        out[i] = c < n_vec ? in[r * n_vec + c] : 0.0;
    }
}

// Y (n x n_vec, row-major: eigenvectors of B = conj(H) as columns) -> out[m][i] = conj(Y[i][m]) complex,
// i.e. eigenvector m of H contiguous (the column-major layout bdg_eigh_dense returns)
template <typename T>
__global__ void td_emit_vectors(const T* __restrict__ y, int n, int ld, int n_vec, double2* __restrict__ out) {
    __shared__ double2 tile[32][33];
    const int i0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    for (int dy = threadIdx.y; dy < 32; dy += blockDim.y) {
        const int i = i0 + dy, m = m0 + threadIdx.x;
        if (i < n && m < n_vec) {
            const T v = y[(size_t)i * ld + m];
            tile[dy][threadIdx.x] = make_double2(td_re(v), -td_im(v));
        }
    }
    __syncthreads();
    for (int dy = threadIdx.y; dy < 32; dy += blockDim.y) {
        const int m = m0 + dy, i = i0 + threadIdx.x;
        if (i < n && m < n_vec) out[(size_t)m * n + i] = tile[threadIdx.x][dy];
    }
}

}  // namespace bdg

namespace {

inline void scatter_for_tridiagonal(bdg_system* sys, double2* a, hipStream_t st) {
    bdg::scatter_dense<<<(unsigned)sys->nb, 128, 0, st>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, a, (int)sys->nb);
}
inline void scatter_for_tridiagonal(bdg_system* sys, double* a, hipStream_t st) {
    bdg::scatter_dense_real<<<(unsigned)sys->nb, 128, 0, st>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, a, (int)sys->nb);
}

// All eigenvalues, ascending, of the uploaded matrix, and - if z_out is given - the eigenvectors of the
// eigenvalues above `lower_bound` (diagonalize() wants the positive half).  T = double when imag(H) = 0,
// else double2.  *n_vectors receives the number of such eigenvalues; if it exceeds `capacity` nothing is
// written to z_out and BDG_EINVAL is returned.  z_out: eigenvector m (m-th eigenvalue above the bound,
// ascending) in the 4*nb complex entries from z_out + 8*nb*m.
// K10 (twostage.hpp): a (n x n real symmetric, row-major, both triangles; overwritten) -> diagonal d and sub-diagonal e of
// an orthogonally similar tridiagonal matrix, through a band of half-width kTsBand.  Everything is enqueued on `st`.
// What the eigenvectors of the two-stage route need afterwards: the block reflectors of stage 1 (V and T of every
// panel) and the band matrix as it was before the bulge chasing, by rows.
struct TsKeep {
    DeviceBuffer<double> v_all, t_all, full;
    std::vector<int64_t> v_at;   // offset of panel p's V (m_p + 16 rows of B entries) in v_all
    std::vector<int64_t> rows;   // m_p
    void release() {
        v_all.release();
        t_all.release();
        full.release();
    }
};

int tridiagonalise_two_stage(double* a, int64_t n, double* d, double* e, hipStream_t st, TsKeep* keep = nullptr) {
    constexpr int B = bdg::kTsBand;
    DeviceBuffer<double> work, band;
    DeviceBuffer<unsigned> sync;
    hipStream_t side = nullptr;
    hipEvent_t ev_strip = nullptr, ev_qr = nullptr;
    const int64_t n_panels = n > B + 1 ? (n - B - 1 + B - 1) / B : 0;
    constexpr int kSlices = 8;                       // k slices of X = A22 V at most
    const int64_t max_parts = (n + 255) / 256;       // parts of Z = V^T X
    // work: V, X, W (n x B each), T, M (B x B), Z parts, QR partials [2][256][2B], row broadcast [2][B], the k slices of X
    const size_t work_count = (size_t)3 * n * B + 2 * B * B + (size_t)max_parts * B * B + (size_t)2 * 256 * 2 * B + 2 * B +
                              (size_t)kSlices * n * B + (size_t)((n + 127) / 128 + 1) * 2 * B * B + 2 * B + B * B + (size_t)2 * n * B + B * B + B;
    auto body = [&]() -> int {
        if (int rc = work.reserve(work_count)) return rc;
        if (int rc = band.reserve((size_t)(n + 4 * B) * bdg::kTsBandLd)) return rc;
        const size_t sync_count = ((size_t)2 * n_panels + (size_t)n + 8 + 3) / 4 * 4;
        if (int rc = sync.reserve(sync_count)) return rc;
        HIP_TRY(hipMemsetAsync(sync.ptr, 0, sync_count * sizeof(unsigned), st));
        HIP_TRY(hipMemsetAsync(work.ptr, 0, work_count * sizeof(double), st));
        double* v = work.ptr;
        double* x = v + (size_t)n * B;
        double* w = x + (size_t)n * B;
        double* t = w + (size_t)n * B;
        double* mm = t + B * B;
        double* zpart = mm + B * B;
        double* partial = zpart + (size_t)max_parts * B * B;
        double* rowi = partial + (size_t)2 * 256 * 2 * B;
        double* xpart = rowi + 2 * B;
        double* gpart = xpart + (size_t)kSlices * n * B;
        double* qr_scale = gpart + (size_t)((n + 127) / 128 + 1) * 2 * B * B;
        double* qr_beta = qr_scale + B;
        double* qr_w = qr_beta + B;
        double* v2 = qr_w + B * B;  // V of every other panel
        double* panel_copy = v2 + (size_t)n * B;  // the panel as it was (n x B), for the verification and a second factorisation
        double* qr_m = panel_copy + (size_t)n * B;
        double* qr_norms = qr_m + B * B;
        double verify_tolerance = 2e-13;  // (a sound panel: a few 1e-15; the failures seen: 1e-11 and more)
        if (const char* env = knob::raw("BODGE_AMD_EIGH_VERIFY")) verify_tolerance = atof(env);
        bool gram_qr = true;
        if (const char* env = knob::raw("BODGE_AMD_EIGH_GRAM_QR")) gram_qr = atoi(env) != 0;
        double gram_floor = bdg::kTsGramFloor;
        if (const char* env = knob::raw("BODGE_AMD_EIGH_GRAM_FLOOR")) gram_floor = atof(env);
        unsigned* panel_counter = sync.ptr;                 // one barrier counter per panel
        unsigned* unsafe = sync.ptr + n_panels + n + 8;     // [n_panels]: panels the Gram route gave up
        unsigned* progress = sync.ptr + n_panels;           // [n]
        unsigned* ticket = progress + n;
        unsigned* gave_up = ticket + 1;
        // The factorisation of panel p + 1 (a chain of small kernels, one of them a single workgroup) runs on a second
        // stream beside the bulk of panel p's rank-2B update: first the strip of the update that panel p + 1 lives in,
        // then both at once.  V alternates between two buffers (the update still reads panel p's).
        double* vbuf[2] = {v, v2};
        if (keep) {
            int64_t total = 0;
            for (int64_t j0 = 0; j0 + B + 1 < n; j0 += B) {
                keep->v_at.push_back(total);
                keep->rows.push_back(n - j0 - B);
                total += (n - j0 - B + 16) * B;
            }
            if (int rc = keep->v_all.reserve((size_t)std::max<int64_t>(total, 1))) return rc;
            if (int rc = keep->t_all.reserve((size_t)std::max<int64_t>(n_panels, 1) * B * B)) return rc;
            if (int rc = keep->full.reserve((size_t)(n + B + 2) * bdg::kTsRowLd)) return rc;
        }
        auto v_of = [&](int64_t panel) { return keep ? keep->v_all.ptr + keep->v_at[(size_t)panel] : vbuf[panel & 1]; };
        auto t_of = [&](int64_t panel) { return keep ? keep->t_all.ptr + (size_t)panel * B * B : t; };
        auto factorise = [&](int64_t panel, int64_t j0, hipStream_t on) -> int {
            const int64_t r0 = j0 + B, m = n - r0;
            bdg::TsPanelArgs q{};
            q.a = a;
            q.n = (int)n;
            q.j0 = (int)j0;
            q.r0 = (int)r0;
            q.m = (int)m;
            q.reflectors = (int)std::min<int64_t>(B, m - 1);
            q.v = v_of(panel);
            q.t = t_of(panel);
            q.partial = partial;
            q.rowi = rowi;
            q.counter = panel_counter + panel;
            const unsigned qr_grid = (unsigned)((m + 16 + 255) / 256);  // (sixteen zero rows of V behind the last one)
            if (qr_grid > 256) return fail(BDG_EINVAL, "two-stage route limited to 65000 rows");
            if (gram_qr) {
                // reflectors from the Gram matrix of the panel (no grid barrier); panels that lose too much of a column
                // to cancellation are factorised by ts_panel_qr instead (`unsafe` decides on the device)
                const unsigned gparts = (unsigned)std::max<int64_t>(1, (m - B + bdg::kTsGramRows - 1) / bdg::kTsGramRows);
                bdg::ts_gram<<<gparts, 256, 0, on>>>(a, (int)n, (int)j0, (int)r0, (int)m, gpart, B);
                bdg::TsRecurArgs rc{};
                rc.a = a;
                rc.n = (int)n;
                rc.j0 = (int)j0;
                rc.r0 = (int)r0;
                rc.m = (int)m;
                rc.reflectors = q.reflectors;
                rc.gpart = gpart;
                rc.parts = (int)gparts;
                rc.scale = qr_scale;
                rc.beta = qr_beta;
                rc.wrows = qr_w;
                rc.unsafe = unsafe + panel;
                rc.floor = gram_floor;
                rc.colnorm2 = qr_norms;
                bdg::ts_qr_recur<<<1, 256, 0, on>>>(rc);
                bdg::ts_qr_apply<<<qr_grid, 256, 0, on>>>(a, (int)n, (int)j0, (int)r0, (int)m, q.reflectors, qr_scale, qr_beta, qr_w,
                                                         unsafe + panel, q.v, panel_copy);
                // T from the stored V, and the check of Q^T P against the R that was written (flag 2 = factorise again)
                const unsigned vparts = (unsigned)((m + bdg::kTsGramRows - 1) / bdg::kTsGramRows);
                bdg::ts_vgram2<<<vparts, 256, 0, on>>>(q.v, panel_copy, (int)m, gpart);
                bdg::ts_t_from_v<<<1, 256, 0, on>>>(gpart, (int)vparts, q.reflectors, unsafe + panel, qr_scale, q.t, qr_m);
                bdg::ts_qr_verify<<<(unsigned)((m + 255) / 256), 256, 0, on>>>(a, (int)n, (int)j0, (int)r0, (int)m, q.v, panel_copy, qr_m, qr_norms,
                                                                              verify_tolerance, unsafe + panel);
                q.only_if = unsafe + panel;
                q.source = panel_copy;
            }
            bdg::ts_panel_qr<<<qr_grid, 256, 0, on>>>(q);
            return BDG_OK;
        };
        // (pays while the bulk of an update outlasts the chain: two events per panel cost more than they hide on small blocks)
        bool lookahead = n >= 5000;
        if (const char* env = knob::raw("BODGE_AMD_EIGH_LOOKAHEAD")) lookahead = atoi(env) != 0;
        if (lookahead && !side) {
            int device = 0;
            HIP_TRY(hipGetDevice(&device));
            if (int rc = pooled_stream(device, 0, &side)) return fail(rc, "stream creation failed");  // (the device's side stream: core.hpp)
            HIP_TRY(hipEventCreateWithFlags(&ev_strip, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ev_qr, hipEventDisableTiming));
        }
        int64_t panel = 0;
        if (n_panels > 0)
            if (int rc = factorise(0, 0, st)) return rc;
        for (int64_t j0 = 0; j0 + B + 1 < n; j0 += B, ++panel) {
            const int64_t r0 = j0 + B, m = n - r0;
            const double* vp = v_of(panel);
            const double* tp = t_of(panel);
            // X = A22 V in k slices: enough waves (16 rows each) for every SIMD of the device
            const int64_t row_waves = (m + 15) / 16;
            const int slices = (int)std::clamp<int64_t>((2048 + row_waves - 1) / row_waves, 1, kSlices);
            const int k_slice = (int)(((m + slices - 1) / slices + 15) / 16 * 16);
            const int used = (int)((m + k_slice - 1) / k_slice);
            bdg::ts_symm<<<dim3((unsigned)((m + 63) / 64), (unsigned)used), 256, 0, st>>>(a, (int)n, (int)r0, (int)m, vp, xpart, k_slice);
            const unsigned parts = (unsigned)((m + 255) / 256);
            bdg::ts_xz<<<parts, 256, 0, st>>>(xpart, used, vp, (int)m, x, zpart);
            bdg::ts_small<<<1, 256, 0, st>>>(zpart, (int)parts, tp, mm);
            bdg::ts_w<<<(unsigned)((m + 16 + 255) / 256), 256, 0, st>>>(x, vp, tp, mm, (int)m, w);
            const bool more = j0 + B + B + 1 < n;  // another panel follows
            const unsigned tile_rows = (unsigned)((m + 63) / 64);
            bdg::ts_rank2k<<<dim3(1, tile_rows), 256, 0, st>>>(a, (int)n, (int)r0, (int)m, vp, w, 1);
            const bool beside = more && lookahead && m >= 3072;
            if (beside) {
                HIP_TRY(hipEventRecord(ev_strip, st));
                HIP_TRY(hipStreamWaitEvent(side, ev_strip, 0));
                if (int rc = factorise(panel + 1, j0 + B, side)) return rc;
                HIP_TRY(hipEventRecord(ev_qr, side));
            }
            if (m > B) {
                const unsigned tile_cols = (unsigned)((m - B + 63) / 64);
                bdg::ts_rank2k<<<dim3(tile_cols, tile_rows), 256, 0, st>>>(a, (int)n, (int)r0, (int)m, vp, w, 0);
            }
            if (beside) HIP_TRY(hipStreamWaitEvent(st, ev_qr, 0));
            if (more && !beside)
                if (int rc = factorise(panel + 1, j0 + B, st)) return rc;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(band.ptr, 0, (size_t)(n + 4 * B) * bdg::kTsBandLd * sizeof(double), st));
        bdg::ts_extract_band<<<2048, 256, 0, st>>>(a, (int)n, band.ptr);
        if (keep) bdg::ts_expand_band<<<2048, 256, 0, st>>>(band.ptr, (int)n, keep->full.ptr);
        if (const char* env = knob::raw("BODGE_AMD_EIGH_STAGE2"); !(env && env[0] == '0')) {
            bdg::TsChaseArgs c{};
            c.ab = band.ptr;
            c.n = (int)n;
            c.progress = progress;
            c.ticket = ticket;
            c.gave_up = gave_up;
            c.timeout_ticks = 400000000u;  // 4 s of the 100 MHz clock
            c.xcd = gave_up + 1;
            HIP_TRY(hipMemsetAsync(c.xcd, 0xFF, sizeof(unsigned), st));
            int chase_grid = 2048;  // (an eighth of them find themselves on the XCD the sweeps run on)
            if (const char* env = knob::raw("BODGE_AMD_EIGH_CHASE_GRID")) chase_grid = std::max(1, atoi(env));
            DeviceBuffer<unsigned long long> phases;
            if (knob::raw("BODGE_AMD_EIGH_CHASE_PROFILE")) {
                if (int rc = phases.reserve(8)) return rc;
                HIP_TRY(hipMemsetAsync(phases.ptr, 0, 8 * sizeof(unsigned long long), st));
                c.profile = phases.ptr;
            }
            bdg::ts_chase<<<chase_grid, 64, 0, st>>>(c);
            if (c.profile) {
                unsigned long long ticks[8];
                HIP_TRY(hipMemcpyAsync(ticks, phases.ptr, sizeof ticks, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                const double steps = (double)n * n / (2.0 * B);
                fprintf(stderr, "[bdg] ts_chase, us per step: to LDS %.2f, poll + prefetch %.2f, reflector %.2f, G %.2f, S %.2f, G2 %.2f, stores %.2f, drain %.2f\n",
                        ticks[0] * 0.01 / steps, ticks[1] * 0.01 / steps, ticks[2] * 0.01 / steps, ticks[3] * 0.01 / steps,
                        ticks[4] * 0.01 / steps, ticks[5] * 0.01 / steps, ticks[6] * 0.01 / steps, ticks[7] * 0.01 / steps);
                phases.release();
            }
        }
        bdg::ts_band_to_tridiagonal<<<256, 256, 0, st>>>(band.ptr, (int)n, d, e);
        HIP_TRY(hipGetLastError());
        unsigned failed = 0;
        HIP_TRY(hipMemcpyAsync(&failed, gave_up, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (knob::raw("BODGE_AMD_TRACE") && n_panels > 0) {
            std::vector<unsigned> flags((size_t)n_panels);
            HIP_TRY(hipMemcpy(flags.data(), unsafe, sizeof(unsigned) * n_panels, hipMemcpyDeviceToHost));
            int64_t fell_back = 0;
            for (unsigned f : flags) fell_back += f != 0;
            fprintf(stderr, "[bdg] two-stage: %lld panels, %lld of them factorised with a barrier per column\n", (long long)n_panels, (long long)fell_back);
        }
        if (failed) return fail(BDG_EDEVICE, "band reduction gave up waiting for a sweep (is another kernel holding the GPU?)");
        return BDG_OK;
    };
    const int rc = body();
    if (side) {
        (void)hipStreamSynchronize(side);
        (void)hipEventDestroy(ev_strip);
        (void)hipEventDestroy(ev_qr);
    }
    work.release();
    band.release();
    sync.release();
    return rc;
}

template <typename T>
int eig_tridiagonal_typed(bdg_system* sys, double* w_out, double lower_bound, int64_t capacity, int64_t* n_vectors,
                          double* z_out) {
    const int64_t n = 4 * sys->nb;
    if (n > 46000) return fail(BDG_EINVAL, "dense path limited to 4*nb <= 46000");
    hipStream_t st = sys->stream;
    DeviceBuffer<T> a, vectors, taus, y, partial;  // vectors: v[2], q, pending v / w (n each)
    DeviceBuffer<double> diag;                    // d, e, e^2, eigenvalues (n each)
    DeviceBuffer<double> zt, scratch;             // tridiagonal eigenvectors, LU work space
    DeviceBuffer<bdg::TdCluster> clusters_dev;
    DeviceBuffer<double2> emitted;
    DeviceBuffer<bdg::TdScalars<T>> scal;
    TsKeep keep;
    bool two_stage = false;
    auto body = [&]() -> int {
        if (int rc = a.reserve((size_t)n * n + 64)) return rc;  // (+64: ts_symm reads up to 15 entries past a row's end)
        constexpr int kVectors = 4 + 2 * bdg::kTdDefer;  // (the last one holds the 2 K dot products handed from step to step)
        if (int rc = vectors.reserve((size_t)kVectors * n)) return rc;
        if (int rc = taus.reserve((size_t)n)) return rc;
        if (int rc = diag.reserve((size_t)4 * n)) return rc;
        if (int rc = scal.reserve(1)) return rc;
        HIP_TRY(hipMemsetAsync(a.ptr, 0, sizeof(T) * n * n, st));
        HIP_TRY(hipMemsetAsync(vectors.ptr, 0, sizeof(T) * kVectors * n, st));
        HIP_TRY(hipMemsetAsync(taus.ptr, 0, sizeof(T) * n, st));
        HIP_TRY(hipMemsetAsync(diag.ptr, 0, sizeof(double) * 4 * n, st));
        HIP_TRY(hipMemsetAsync(scal.ptr, 0, sizeof(bdg::TdScalars<T>), st));
        scatter_for_tridiagonal(sys, a.ptr, st);
        T* v[2] = {vectors.ptr, vectors.ptr + n};
        T* q = vectors.ptr + 2 * n;
        bdg::TdPending<T> pend{};
        for (int i = 0; i < bdg::kTdDefer; ++i) {
            pend.v[i] = vectors.ptr + (size_t)(3 + i) * n;
            pend.w[i] = vectors.ptr + (size_t)(3 + bdg::kTdDefer + i) * n;
        }
        pend.count = 0;
        T* pend_dots = vectors.ptr + (size_t)(3 + 2 * bdg::kTdDefer) * n;
        // Deferring costs the one-workgroup vector kernel ~8 us more per step (corrections of q and of row j by the pending
        // pairs) and saves passes over the block: it pays from n ~ 5000 (n = 3600: 128 against 100 ms, n = 10^4: 1.26 against
        // 1.58 s).  BODGE_AMD_EIGH_DEFER=1..4 overrides (1 = apply every update in the next pass).
        int defer = n >= 5000 ? bdg::kTdDefer : 1;
        if (const char* env = knob::raw("BODGE_AMD_EIGH_DEFER")) defer = std::clamp(atoi(env), 1, bdg::kTdDefer);
        double* d = diag.ptr;
        double* e = diag.ptr + n;
        // eigenvalues only, real matrices: through the band (K10, BLAS-3); BODGE_AMD_EIGH_STAGES=1|2 overrides
        if constexpr (std::is_same_v<T, double>) {
            // (measured, scratch/r4_twostage_check.py: 92 against 92 ms at n = 3600, 0.36 against 1.1 s at 10^4, 7.8 against 50 s at 4e4)
            // eigenpairs: 147 against 187 ms at n = 3600, 0.72 against 1.68 s at 10^4 (scratch/r4_twostage_vectors.py)
            two_stage = (z_out || n_vectors) ? n >= 3000 : n >= 5000;
            if (const char* env = knob::raw("BODGE_AMD_EIGH_STAGES")) two_stage = atoi(env) == 2 && n > 2 * bdg::kTsBand + 2;
            // (eigenvectors through the band as well: inverse iteration on the band matrix, then the block reflectors of stage 1;
            // BODGE_AMD_EIGH_BAND_VECTORS=0 keeps the one-stage route for them)
            const bool want_vectors = z_out || n_vectors;
            bool band_vectors = true;
            if (const char* env = knob::raw("BODGE_AMD_EIGH_BAND_VECTORS")) band_vectors = atoi(env) != 0;
            two_stage = two_stage && (!want_vectors || band_vectors);
            if (two_stage)
                if (int rc = tridiagonalise_two_stage(a.ptr, n, d, e, st, want_vectors ? &keep : nullptr)) return rc;
        }
        for (int64_t j = 0; j < n && !two_stage; ++j) {
            T* v_unf = v[(j + 1) & 1];  // made at step j-1; its w is finished by this step's vector kernel
            T* v_new = v[j & 1];
            bdg::td_vector_step<T><<<1, 1024, 0, st>>>(a.ptr, (int)n, (int)j, v_unf, q, pend, v_new, d, e, scal.ptr, taus.ptr, pend_dots);
            if (j > 0) ++pend.count;  // (v_{j-1}, w_{j-1}) joined the pending pairs
            if (j + 1 < n) {
                const int64_t rows = n - j - 1;
                const int apply = pend.count >= defer ? pend.count : 0;  // apply them all, or read only
                const int per_block = 4 * bdg::td_rows_for(apply);       // rows a workgroup of four waves takes at a time
                const unsigned grid = (unsigned)std::min<int64_t>(4096, (rows + per_block - 1) / per_block);
                switch (apply) {
                    case 0: bdg::td_fused_pass<T, 0><<<grid, 256, 0, st>>>(a.ptr, (int)n, (int)j, pend, v_new, q); break;
                    case 1: bdg::td_fused_pass<T, 1><<<grid, 256, 0, st>>>(a.ptr, (int)n, (int)j, pend, v_new, q); break;
                    case 2: bdg::td_fused_pass<T, 2><<<grid, 256, 0, st>>>(a.ptr, (int)n, (int)j, pend, v_new, q); break;
                    case 3: bdg::td_fused_pass<T, 3><<<grid, 256, 0, st>>>(a.ptr, (int)n, (int)j, pend, v_new, q); break;
                    default: bdg::td_fused_pass<T, 4><<<grid, 256, 0, st>>>(a.ptr, (int)n, (int)j, pend, v_new, q); break;
                }
                if (apply) pend.count = 0;
            }
        }
        static_assert(bdg::kTdDefer == 4, "the switch above lists the pass kernels for 0..4 pending pairs");
        HIP_TRY(hipGetLastError());
        std::vector<double> host((size_t)2 * n);
        HIP_TRY(hipMemcpyAsync(host.data(), diag.ptr, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        // Gershgorin interval and pivot floor of the tridiagonal matrix (as LAPACK's dstebz), then bisection
        double lo = host[0], hi = host[0], e2max = 0.0;
        std::vector<double> e2((size_t)n, 0.0);
        for (int64_t i = 0; i < n; ++i) {
            const double left = i > 0 ? std::fabs(host[(size_t)(n + i - 1)]) : 0.0;
            const double right = i + 1 < n ? std::fabs(host[(size_t)(n + i)]) : 0.0;
            if (!std::isfinite(host[(size_t)i]) || !std::isfinite(right))
                return fail(BDG_EDEVICE, "tridiagonalisation produced a non-finite entry (is the matrix finite?)");
            lo = std::min(lo, host[(size_t)i] - left - right);
            hi = std::max(hi, host[(size_t)i] + left + right);
            if (i + 1 < n) e2[(size_t)i] = right * right, e2max = std::max(e2max, right * right);
        }
        const double span = std::max(std::fabs(lo), std::fabs(hi));
        lo -= 2.0 * 2.3e-16 * span * (double)n + 1e-300;
        hi += 2.0 * 2.3e-16 * span * (double)n + 1e-300;
        const double pivmin = 2.3e-308 * std::max(1.0, e2max);
        double* e2_dev = diag.ptr + 2 * n;
        double* eig_dev = diag.ptr + 3 * n;
        HIP_TRY(hipMemcpyAsync(e2_dev, e2.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
        constexpr int kPerWave = 64 / bdg::kBisectPoints;
        bdg::td_bisect<<<(unsigned)((n + kPerWave - 1) / kPerWave), 64, 0, st>>>(d, e2_dev, (int)n, lo, hi, pivmin, eig_dev);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(w_out, eig_dev, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!z_out && !n_vectors) return BDG_OK;

        // ---- eigenvectors of the eigenvalues above the bound
        int64_t first = 0;
        while (first < n && !(w_out[first] > lower_bound)) ++first;
        const int64_t n_vec = n - first;
        if (n_vectors) *n_vectors = n_vec;
        if (!z_out || n_vec == 0) return BDG_OK;
        if (n_vec > capacity) return fail(BDG_EINVAL, "%lld eigenvalues above the bound, room for %lld eigenvectors", (long long)n_vec, (long long)capacity);
        // clusters: runs of eigenvalues closer than 1e-5 |T| are orthogonalised against each other afterwards; everything
        // further apart is orthogonal to eps |T| / gap <= 2e-11 from inverse iteration alone.  Column m of zt = eigenvalue first + m.
        const double cluster_gap = 1e-5 * std::max(span, 1e-300), nudge = 10.0 * 2.220446049250313e-16 * span;
        std::vector<bdg::TdCluster> clusters;  // (first = column, count >= 2)
        std::vector<double> shift((size_t)n_vec);
        for (int64_t k = first, run = 0; k < n; ++k) {
            run = (k > first && w_out[k] - w_out[k - 1] < cluster_gap) ? run + 1 : 0;
            // shifts of a run at least `nudge` apart (dstein): members of a degenerate level decorrelate
            shift[(size_t)(k - first)] = run == 0 ? w_out[k] : std::max(w_out[k], shift[(size_t)(k - first - 1)] + nudge);
            if (run == 1) clusters.push_back({(int)(k - first - 1), 2});
            else if (run > 1) ++clusters.back().count;
        }
        if (int rc = zt.reserve((size_t)n * n_vec)) return rc;
        if (two_stage) {
            if constexpr (std::is_same_v<T, double>) {
                // eigenvectors of the band matrix: one wave per eigenvalue, as many at once as the scratch space allows
                a.release();  // (the dense matrix is not needed any more)
                size_t free_bytes = 0, total_bytes = 0;
                HIP_TRY(hipMemGetInfo(&free_bytes, &total_bytes));
                const size_t per_wave = bdg::ts_vector_scratch((int)n) * sizeof(double);
                const int64_t room = (int64_t)((double)free_bytes * 0.6 / (double)per_wave);
                const int n_waves = (int)std::clamp<int64_t>(std::min<int64_t>(n_vec, room), 1, 9 * 256);
                if (int rc = scratch.reserve((size_t)n_waves * bdg::ts_vector_scratch((int)n) + (size_t)n_vec)) return rc;
                double* shift_dev = scratch.ptr + (size_t)n_waves * bdg::ts_vector_scratch((int)n);
                HIP_TRY(hipMemcpyAsync(shift_dev, shift.data(), sizeof(double) * n_vec, hipMemcpyHostToDevice, st));
                bdg::TsVectorArgs va{};
                va.full = keep.full.ptr;
                va.n = (int)n;
                va.shift = shift_dev;
                va.n_vec = (int)n_vec;
                va.norm = span;
                va.scratch = scratch.ptr;
                va.z = zt.ptr;
                va.ld = (int)n_vec;
                va.iterations = 3;
                if (const char* env = knob::raw("BODGE_AMD_EIGH_BAND_ITERATIONS")) va.iterations = std::clamp(atoi(env), 1, 8);
                bdg::ts_band_vectors<<<n_waves, 64, 0, st>>>(va);
            }
        } else {
            const int n_waves = (int)std::min<int64_t>(n_vec, 7168);
            if (int rc = scratch.reserve((size_t)n_waves * 6 * n + (size_t)n_vec)) return rc;
            double* shift_dev = scratch.ptr + (size_t)n_waves * 6 * n;
            HIP_TRY(hipMemcpyAsync(shift_dev, shift.data(), sizeof(double) * n_vec, hipMemcpyHostToDevice, st));
            bdg::td_inverse_iteration<<<n_waves, 64, 0, st>>>(d, e, (int)n, shift_dev, (int)n_vec, span, scratch.ptr, zt.ptr, (int)n_vec);
        }
        if (!clusters.empty()) {
            if (int rc = clusters_dev.reserve(clusters.size())) return rc;
            HIP_TRY(hipMemcpyAsync(clusters_dev.ptr, clusters.data(), sizeof(bdg::TdCluster) * clusters.size(), hipMemcpyHostToDevice, st));
            bdg::td_cluster_orthonormalise<<<(unsigned)clusters.size(), 256, 0, st>>>(clusters_dev.ptr, (int)n, zt.ptr, (int)n_vec);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));  // (the host lists leave scope; errors of the kernels surface here)
        scratch.release();
        // ---- back-transformation Y = H_0 H_1 ... H_{n-2} Z, last reflector first
        // real matrices: Z as rows of column PAIRS (double2), complex ones: of complex entries
        constexpr bool kPairs = sizeof(T) == sizeof(double);
        using Z = double2;
        const int64_t ld = kPairs ? n_vec + (n_vec & 1) : n_vec;      // columns per row of y, in units of T
        const int64_t z_cols = kPairs ? ld / 2 : n_vec, z_ld = z_cols;  // ... and in units of Z
        if (int rc = y.reserve((size_t)n * ld)) return rc;
        if constexpr (kPairs) bdg::td_widen<<<4096, 256, 0, st>>>(zt.ptr, y.ptr, n, n_vec, ld);
        else bdg::td_widen<<<4096, 256, 0, st>>>(zt.ptr, y.ptr, n * n_vec);
        Z* yz = reinterpret_cast<Z*>(y.ptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        zt.release();
        // kTdGroup reflectors per pass over Z; the pass of a group also takes the column sums of the next one
        constexpr int kChunks = 96, kGroup = bdg::kTdGroup;  // (row chunks of a pass: with column pairs a pass has half the column blocks)
        if (int rc = partial.reserve((size_t)2 * kChunks * kGroup * ld + (size_t)kGroup * kGroup)) return rc;
        Z* part[2] = {reinterpret_cast<Z*>(partial.ptr), reinterpret_cast<Z*>(partial.ptr + (size_t)kChunks * kGroup * ld)};
        T* cross = partial.ptr + (size_t)2 * kChunks * kGroup * ld;
        const unsigned col_blocks = (unsigned)((z_cols + 255) / 256);
        int chunk_cap = 48, chunk_rows = 64;  // row chunks of a pass (BODGE_AMD_EIGH_CHUNKS=cap[,rows]: A/B runs)
        if (const char* env = knob::raw("BODGE_AMD_EIGH_CHUNKS")) {
            chunk_cap = std::clamp(atoi(env), 1, kChunks);
            if (const char* comma = strchr(env, ',')) chunk_rows = std::max(1, atoi(comma + 1));
        }
        auto chunks_of = [&](int64_t row0, int* n_chunks, int* rows_per_chunk) {
            const int64_t rows = n - row0;
            *n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(chunk_cap, (rows + chunk_rows - 1) / chunk_rows));
            *rows_per_chunk = (int)((rows + *n_chunks - 1) / *n_chunks);
        };
        if (two_stage) {
            if constexpr (std::is_same_v<T, double>) {
                // Y = H_0 H_1 ... H_{P-1} Z with the block reflectors of stage 1, last panel first
                constexpr int B = bdg::kTsBand;
                const unsigned col_tiles = (unsigned)((n_vec + 31) / 32);
                const int padded = (int)col_tiles * 32;
                DeviceBuffer<double> spart;
                if (int rc = spart.reserve((size_t)((n + 255) / 256 + 1) * B * padded)) return rc;
                double* sprime = spart.ptr + (size_t)((n + 255) / 256) * B * padded;
                for (int64_t p = (int64_t)keep.rows.size() - 1; p >= 0; --p) {
                    const int64_t m = keep.rows[(size_t)p], r0 = n - m;
                    const double* vp = keep.v_all.ptr + keep.v_at[(size_t)p];
                    const double* tp = keep.t_all.ptr + (size_t)p * B * B;
                    const unsigned slices = (unsigned)((m + 255) / 256);
                    bdg::ts_vtz<<<dim3(col_tiles, slices), 64, 0, st>>>(vp, (int)m, y.ptr, (int)ld, (int)r0, (int)n_vec, spart.ptr);
                    bdg::ts_sprime<<<col_tiles, 256, 0, st>>>(tp, spart.ptr, (int)slices, padded, sprime);
                    const unsigned row_blocks = (unsigned)std::clamp<int64_t>((m + 63) / 64, 1, 32);
                    bdg::ts_zupdate<<<dim3(col_tiles, row_blocks), 256, 0, st>>>(vp, (int)m, y.ptr, (int)ld, (int)r0, (int)n_vec, sprime, padded);
                }
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(st));
                spart.release();
            }
        } else if (n >= 2) {
            int64_t hi = n - 2;
            int count = (int)std::min<int64_t>(kGroup, hi + 1);
            int64_t row0 = hi - count + 2;
            int n_chunks = 0, rows_per_chunk = 0;
            chunks_of(row0, &n_chunks, &rows_per_chunk);
            bdg::td_reflect_sums<T, Z><<<dim3(col_blocks, (unsigned)n_chunks), 256, 0, st>>>(a.ptr, (int)n, (int)hi, count, yz, (int)z_ld,
                                                                                        (int)z_cols, (int)row0, rows_per_chunk, part[0]);
            int flip = 0, written_chunks = n_chunks;
            while (hi >= 0) {
                const int64_t next_hi = hi - count;
                const int next_count = next_hi >= 0 ? (int)std::min<int64_t>(kGroup, next_hi + 1) : 0;
                const int64_t next_row0 = next_count > 0 ? next_hi - next_count + 2 : row0;
                bdg::td_reflector_gram<T><<<1, 256, 0, st>>>(a.ptr, (int)n, (int)hi, count, cross);
                chunks_of(row0, &n_chunks, &rows_per_chunk);
                bdg::td_reflect_group<T, Z><<<dim3(col_blocks, (unsigned)n_chunks), 256, 0, st>>>(
                    a.ptr, (int)n, (int)hi, count, taus.ptr, cross, yz, (int)z_ld, (int)z_cols, (int)row0, rows_per_chunk,
                    written_chunks, part[flip], next_count, (int)next_row0, part[flip ^ 1]);
                written_chunks = n_chunks;
                flip ^= 1;
                hi = next_hi;
                count = next_count;
                row0 = next_row0;
            }
        }
        HIP_TRY(hipGetLastError());
        // eigenvectors of H = conj(B): contiguous per vector, conjugated
        if (int rc = emitted.reserve((size_t)n * n_vec)) return rc;
        const dim3 tiles((unsigned)((n + 31) / 32), (unsigned)((n_vec + 31) / 32));
        bdg::td_emit_vectors<T><<<tiles, dim3(32, 8), 0, st>>>(y.ptr, (int)n, (int)ld, (int)n_vec, emitted.ptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(z_out, emitted.ptr, sizeof(double2) * n * n_vec, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return BDG_OK;
    };
    const int rc = body();
    keep.release();
    a.release();
    vectors.release();
    taus.release();
    y.release();
    partial.release();
    diag.release();
    zt.release();
    scratch.release();
    clusters_dev.release();
    emitted.release();
    scal.release();
    return rc;
}

bool tridiagonal_real_route(const bdg_system* sys) {
    bool real_route = sys->is_real;
    if (const char* env = knob::raw("BODGE_AMD_EIGH_REAL")) real_route = real_route && atoi(env) != 0;
    return real_route;
}

int eigvals_tridiagonal(bdg_system* sys, double* w_out) {
    return tridiagonal_real_route(sys) ? eig_tridiagonal_typed<double>(sys, w_out, 0.0, 0, nullptr, nullptr)
                                       : eig_tridiagonal_typed<double2>(sys, w_out, 0.0, 0, nullptr, nullptr);
}

int eig_tridiagonal_above(bdg_system* sys, double* w_out, double lower_bound, int64_t capacity, int64_t* n_vectors, double* z_out) {
    return tridiagonal_real_route(sys) ? eig_tridiagonal_typed<double>(sys, w_out, lower_bound, capacity, n_vectors, z_out)
                                       : eig_tridiagonal_typed<double2>(sys, w_out, lower_bound, capacity, n_vectors, z_out);
}

}  // namespace
