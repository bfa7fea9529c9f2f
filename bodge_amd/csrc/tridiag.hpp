// tridiag.hpp - all eigenvalues of the dense matrix without an external library: Householder
// tridiagonalisation on the GPU, then bisection on the tridiagonal matrix (Sturm counts).
// Part of the single translation unit bodge_hip.hip (included after dense.hpp).
//
// Why: free_energy's dense route (reference hamiltonian.py:282-321) and BASELINE config 5's check
// ("eigenvalues vs numpy.linalg.eigh to 1e-10") need eigenvalues only.  The Jacobi kernels (dense.hpp)
// stop at 4N = 4096 and rocSOLVER's 931 MB object takes minutes to arrive on a fresh machine; this
// path has no such wait and no size limit short of the n^2 matrix.  Eigenvectors above the Jacobi
// limit stay with rocSOLVER (back-transformation + tridiagonal eigenvectors are not built).
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

// One workgroup.  Step j: (i) finish w of step j-1 from q = B' v  (ii) row j of the lazily updated matrix:
// d_j and x  (iii) the Householder vector of step j.  v_prev / w / q are indexed by absolute row (>= j),
// v_new by absolute row (>= j+1).
template <typename T>
__global__ __launch_bounds__(1024) void td_vector_step(const T* __restrict__ a, int n, int j, const T* __restrict__ v_prev,
                                                       T* __restrict__ w, const T* __restrict__ q, T* __restrict__ v_new,
                                                       double* __restrict__ d, double* __restrict__ e, TdScalars<T>* scal) {
    __shared__ T scratch[16];
    __shared__ T shared_scalar;
    T zero;
    td_set(zero, 0.0, 0.0);
    if (j > 0) {
        // (i) p = tau q;  w = p - (tau / 2) (p^H v) v      over rows j .. n-1 (v_prev[j] = 1)
        const T tau = scal->tau;
        T dot = zero;
        for (int r = j + threadIdx.x; r < n; r += blockDim.x) dot = td_add(dot, td_mul(td_conj(td_mul(tau, q[r])), v_prev[r]));
        dot = td_block_sum(dot, scratch);
        T half_tau;
        td_set(half_tau, -0.5 * td_re(tau), -0.5 * td_im(tau));
        const T alpha2 = td_mul(half_tau, dot);
        for (int r = j + threadIdx.x; r < n; r += blockDim.x) w[r] = td_add(td_mul(tau, q[r]), td_mul(alpha2, v_prev[r]));
        __syncthreads();
    }
    // (ii) row j with the pending update B'(j, c) = B(j, c) - v_j conj(w_c) - w_j conj(v_c)
    auto entry = [&](int c) {
        T value = a[(size_t)j * n + c];
        if (j > 0) value = td_sub(value, td_add(td_mul(v_prev[j], td_conj(w[c])), td_mul(w[j], td_conj(v_prev[c]))));
        return value;
    };
    if (threadIdx.x == 0) d[j] = td_re(entry(j));
    const int m = n - j - 1;
    if (m == 0) return;
    // (iii) x_c = conj(B'(j, c)), c = j+1 .. n-1;  alpha = x_{j+1}
    double norm2 = 0.0;
    for (int c = j + 2 + threadIdx.x; c < n; c += blockDim.x) {
        const T x = td_conj(entry(c));
        v_new[c] = x;  // (scaled below)
        norm2 += td_abs2(x);
    }
    T packed;
    td_set(packed, norm2, 0.0);
    norm2 = td_re(td_block_sum(packed, scratch));
    const T alpha = td_conj(entry(j + 1));
    if (norm2 == 0.0 && td_im(alpha) == 0.0) {  // nothing to annihilate: H = I
        if (threadIdx.x == 0) {
            e[j] = td_re(alpha);
            scal->tau = zero;
            td_set(v_new[j + 1], 1.0, 0.0);
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
    for (int c = j + 2 + threadIdx.x; c < n; c += blockDim.x) v_new[c] = td_mul(v_new[c], scale);
    if (threadIdx.x == 0) {
        e[j] = beta;
        td_set(scal->tau, (beta - ar) / beta, -ai / beta);
        td_set(v_new[j + 1], 1.0, 0.0);
    }
    (void)shared_scalar;
}

// Trailing block rows / columns j+1 .. n-1: apply the pending update of step j-1, store, multiply by v_new.
// One wave per row; q[r] = sum_c B'(r, c) v_new[c].
template <typename T>
__global__ __launch_bounds__(256) void td_fused_pass(T* __restrict__ a, int n, int j, const T* __restrict__ v_prev,
                                                     const T* __restrict__ w_prev, const T* __restrict__ v_new,
                                                     T* __restrict__ q) {
    const int lane = threadIdx.x & (kWave - 1);
    const int rows_per_pass = gridDim.x * (blockDim.x / kWave);
    const bool pending = j > 0;
    for (int r = j + 1 + blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; r < n; r += rows_per_pass) {
        T vr, wr;
        td_set(vr, 0.0, 0.0);
        td_set(wr, 0.0, 0.0);
        if (pending) vr = v_prev[r], wr = w_prev[r];
        T acc;
        td_set(acc, 0.0, 0.0);
        T* row = a + (size_t)r * n;
        for (int c = j + 1 + lane; c < n; c += kWave) {
            T value = row[c];
            if (pending) {
                value = td_sub(value, td_add(td_mul(vr, td_conj(w_prev[c])), td_mul(wr, td_conj(v_prev[c]))));
                row[c] = value;
            }
            acc = td_add(acc, td_mul(value, v_new[c]));
        }
        for (int off = kWave / 2; off >= 1; off >>= 1) acc = td_add(acc, td_shfl(acc, off));
        if (lane == 0) q[r] = acc;
    }
}

// k-th smallest eigenvalue of the symmetric tridiagonal matrix (d, e) by bisection on the Sturm count
// (number of negative pivots of T - x I): one thread per eigenvalue, ascending output.
__global__ void td_bisect(const double* __restrict__ d, const double* __restrict__ e2, int n, double lo, double hi,
                          double pivmin, double* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double a = lo, b = hi;
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (a + b);
        if (!(mid > a && mid < b)) break;  // the interval is one ulp wide
        double piv = d[0] - mid;
        int count = piv < 0.0 ? 1 : 0;
        for (int i = 1; i < n; ++i) {
            if (fabs(piv) < pivmin) piv = -pivmin;
            piv = d[i] - mid - e2[i - 1] / piv;
            count += piv < 0.0 ? 1 : 0;
        }
        if (count > k) b = mid;
        else a = mid;
    }
    out[k] = 0.5 * (a + b);
}

}  // namespace bdg

namespace {

inline void scatter_for_tridiagonal(bdg_system* sys, double2* a, hipStream_t st) {
    bdg::scatter_dense<<<(unsigned)sys->nb, 128, 0, st>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, a, (int)sys->nb);
}
inline void scatter_for_tridiagonal(bdg_system* sys, double* a, hipStream_t st) {
    bdg::scatter_dense_real<<<(unsigned)sys->nb, 128, 0, st>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, a, (int)sys->nb);
}

// All eigenvalues, ascending, of the uploaded matrix.  T = double when imag(H) = 0, else double2.
template <typename T>
int eigvals_tridiagonal_typed(bdg_system* sys, double* w_out) {
    const int64_t n = 4 * sys->nb;
    if (n > 46000) return fail(BDG_EINVAL, "dense path limited to 4*nb <= 46000");
    hipStream_t st = sys->stream;
    DeviceBuffer<T> a, vectors;  // vectors: v[2], w, q (n each)
    DeviceBuffer<double> diag;   // d, e, e^2 scratch, eigenvalues (n each)
    DeviceBuffer<bdg::TdScalars<T>> scal;
    auto body = [&]() -> int {
        if (int rc = a.reserve((size_t)n * n)) return rc;
        if (int rc = vectors.reserve((size_t)4 * n)) return rc;
        if (int rc = diag.reserve((size_t)4 * n)) return rc;
        if (int rc = scal.reserve(1)) return rc;
        HIP_TRY(hipMemsetAsync(a.ptr, 0, sizeof(T) * n * n, st));
        HIP_TRY(hipMemsetAsync(vectors.ptr, 0, sizeof(T) * 4 * n, st));
        HIP_TRY(hipMemsetAsync(diag.ptr, 0, sizeof(double) * 4 * n, st));
        HIP_TRY(hipMemsetAsync(scal.ptr, 0, sizeof(bdg::TdScalars<T>), st));
        scatter_for_tridiagonal(sys, a.ptr, st);
        T* v[2] = {vectors.ptr, vectors.ptr + n};
        T* w = vectors.ptr + 2 * n;
        T* q = vectors.ptr + 3 * n;
        double* d = diag.ptr;
        double* e = diag.ptr + n;
        for (int64_t j = 0; j < n; ++j) {
            T* v_prev = v[(j + 1) & 1];  // made at step j-1
            T* v_new = v[j & 1];
            bdg::td_vector_step<T><<<1, 1024, 0, st>>>(a.ptr, (int)n, (int)j, v_prev, w, q, v_new, d, e, scal.ptr);
            if (j + 1 < n) {
                const int64_t rows = n - j - 1;
                const unsigned grid = (unsigned)std::min<int64_t>(4096, (rows + 3) / 4);
                bdg::td_fused_pass<T><<<grid, 256, 0, st>>>(a.ptr, (int)n, (int)j, v_prev, w, v_new, q);
            }
        }
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
        bdg::td_bisect<<<(unsigned)((n + 63) / 64), 64, 0, st>>>(d, e2_dev, (int)n, lo, hi, pivmin, eig_dev);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(w_out, eig_dev, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return BDG_OK;
    };
    const int rc = body();
    a.release();
    vectors.release();
    diag.release();
    scal.release();
    return rc;
}

int eigvals_tridiagonal(bdg_system* sys, double* w_out) {
    bool real_route = sys->is_real;
    if (const char* env = knob::raw("BODGE_AMD_EIGH_REAL")) real_route = real_route && atoi(env) != 0;
    return real_route ? eigvals_tridiagonal_typed<double>(sys, w_out) : eigvals_tridiagonal_typed<double2>(sys, w_out);
}

}  // namespace
