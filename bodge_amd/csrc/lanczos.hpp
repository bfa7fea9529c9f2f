// lanczos.hpp - device-resident Lanczos process on H^2
// Part of the single translation unit bodge_hip.hip (included there, in this order:
// core, plans, libraries, recurrence, lanczos, dense); everything lives in its unnamed namespace.
#pragma once

namespace {

// --------------------------------------------------------------------- Lanczos
// Lanczos process on A = H^2 for the eigenvalues of H closest to zero (the excitation gap).
// v_0 = v/|v|;  per iteration j, with unnormalised W_j = beta_j v_j kept in memory:
//   a.  U = H v_j               K1(cur = W_j,  prev = scratch, coef = 1/beta_j, pscale = 0)
//   b.  R = H U - beta_j v_{j-1}  K1(cur = U, prev = W_{j-1}, coef = 1, pscale = beta_j/beta_{j-1});
//       its d-dot |U|^2 = <v_j|H^2|v_j> = alpha_j
//   c.  W_{j+1} = R - alpha_j v_j = R - (alpha_j/beta_j) W_j,  beta_{j+1} = |W_{j+1}|
// All scalars stay on the device (lanczos_scalars); the host only enqueues and finally reads
// alpha/beta.  Several start vectors run as independent columns of the same launches.
struct LanczosState {
    Batch batch;                 // kernel plan and matrix arguments of the run
    int cols = 0, iter = 0, max_iter = 0, n_active = 0;
    DeviceBuffer<double2> work;  // third vector buffer (U = H v_j)
    DeviceBuffer<double> scalars, sums, norm_partial;
    double2 *w_cur = nullptr, *w_prev = nullptr;
    bdg::LanczosScalars z{};
};

void lanczos_free(bdg_system* sys) {
    LanczosState* lz = static_cast<LanczosState*>(sys->lanczos);
    if (!lz) return;
    lz->work.release();
    lz->scalars.release();
    lz->sums.release();
    lz->norm_partial.release();
    delete lz;
    sys->lanczos = nullptr;
}

constexpr int kNormGrid = 512;

int lanczos_norms(bdg_system* sys, LanczosState* lz, const double2* vec, const double2* other, bool combine) {
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    if (combine) {
        if (b.real)
            bdg::lanczos_combine<2><<<kNormGrid, 256, 0, st>>>(const_cast<double2*>(vec), other, lz->z.g, sys->nb,
                                                               sys->ncols, b.rl, lz->norm_partial.ptr);
        else
            bdg::lanczos_combine<1><<<kNormGrid, 256, 0, st>>>(const_cast<double2*>(vec), other, lz->z.g, sys->nb,
                                                               sys->ncols, b.rl, lz->norm_partial.ptr);
    } else {
        if (b.real)
            bdg::column_norms<2><<<kNormGrid, 256, 0, st>>>(vec, sys->nb, sys->ncols, b.rl, lz->norm_partial.ptr);
        else
            bdg::column_norms<1><<<kNormGrid, 256, 0, st>>>(vec, sys->nb, sys->ncols, b.rl, lz->norm_partial.ptr);
    }
    bdg::reduce_partials<<<1, 256, 0, st>>>(lz->norm_partial.ptr, lz->sums.ptr, kNormGrid, lz->cols);
    HIP_TRY(hipGetLastError());
    return BDG_OK;
}

int lanczos_begin(bdg_system* sys, int n_vectors, const StartSpec& start, int max_iter) {
    if (sys->ncols != sys->nb) return fail(BDG_EINVAL, "Lanczos needs a whole (square) matrix, not a slab");
    if (n_vectors < 1 || n_vectors > 64) return fail(BDG_EINVAL, "Lanczos runs 1..64 start vectors");
    if (max_iter < 1 || max_iter > (1 << 20)) return fail(BDG_EINVAL, "bad iteration limit");
    lanczos_free(sys);
    LanczosState* lz = new LanczosState();
    // the handle learns about the run only once everything below has succeeded
    struct Guard {
        LanczosState* lz;
        bool keep = false;
        ~Guard() {
            if (keep) return;
            lz->work.release();
            lz->scalars.release();
            lz->sums.release();
            lz->norm_partial.release();
            delete lz;
        }
    } guard{lz};
    Batch& b = lz->batch;
    if (int rc = b.begin(sys, 1.0, 1, n_vectors, start, -1, /*col_scalars=*/true)) return rc;  // W_0 in vec_a
    lz->cols = b.rv;
    lz->n_active = n_vectors;
    lz->max_iter = max_iter;
    lz->iter = 0;
    if (int rc = lz->work.reserve(b.vec_count)) return rc;
    const size_t cols = (size_t)lz->cols;
    if (int rc = lz->scalars.reserve((2 * ((size_t)max_iter + 1) + 5) * cols)) return rc;
    if (int rc = lz->sums.reserve(2 * cols)) return rc;
    if (int rc = lz->norm_partial.reserve((size_t)kNormGrid * cols)) return rc;
    HIP_TRY(hipMemsetAsync(lz->scalars.ptr, 0, sizeof(double) * lz->scalars.count, sys->stream));
    HIP_TRY(hipMemsetAsync(lz->work.ptr, 0, sizeof(double2) * b.vec_count, sys->stream));
    double* base = lz->scalars.ptr;
    lz->z.beta_hist = base;
    lz->z.alpha_hist = base + ((size_t)max_iter + 1) * cols;
    double* tail = base + 2 * ((size_t)max_iter + 1) * cols;
    lz->z.coef_a = tail;
    lz->z.pscale_a = tail + cols;
    lz->z.coef_b = tail + 2 * cols;
    lz->z.pscale_b = tail + 3 * cols;
    lz->z.g = tail + 4 * cols;
    lz->w_cur = sys->vec_a.ptr;
    lz->w_prev = sys->vec_b.ptr;
    if (int rc = lanczos_norms(sys, lz, lz->w_cur, nullptr, false)) return rc;
    bdg::lanczos_scalars<<<1, 128, 0, sys->stream>>>(lz->z, lz->sums.ptr, lz->cols, 0, 0);
    HIP_TRY(hipGetLastError());
    guard.keep = true;
    sys->lanczos = lz;
    return BDG_OK;
}

// One iteration j of the process (steps a-c above); W_j is in lz->w_cur on entry, W_{j+1} on exit.
int lanczos_iterate(bdg_system* sys, LanczosState* lz, bdg::StepArgs& args, int j) {
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    args.cur = lz->w_cur;  // a. U = H v_j
    args.prev = lz->work.ptr;
    args.col_coef = lz->z.coef_a;
    args.col_pscale = lz->z.pscale_a;
    b.plan.kernel<<<b.plan.grid, bdg::kBlockThreads, b.plan.lds_bytes, st>>>(args);
    args.cur = lz->work.ptr;  // b. R = H U - beta_j v_{j-1}, alpha_j = |U|^2
    args.prev = lz->w_prev;
    args.col_coef = lz->z.coef_b;
    args.col_pscale = lz->z.pscale_b;
    b.plan.kernel<<<b.plan.grid, bdg::kBlockThreads, b.plan.lds_bytes, st>>>(args);
    bdg::reduce_partials<<<1, 256, 0, st>>>(sys->partial.ptr, lz->sums.ptr, b.plan.grid, (int)b.width);
    bdg::lanczos_scalars<<<1, 128, 0, st>>>(lz->z, lz->sums.ptr, lz->cols, j, 1);
    // c. W_{j+1} = R - alpha_j v_j, beta_{j+1}
    if (int rc = lanczos_norms(sys, lz, lz->w_prev, lz->w_cur, true)) return rc;
    bdg::lanczos_scalars<<<1, 128, 0, st>>>(lz->z, lz->sums.ptr, lz->cols, j + 1, 0);
    std::swap(lz->w_cur, lz->w_prev);
    return BDG_OK;
}

// Second pass: repeat the first n_iter iterations of a freshly begun process (same start vectors:
// the Lanczos vectors are reproduced bit for bit) and accumulate, for every level l and column c,
//   y_{l,c} = Σ_j coef[j][l][c] v_j^{(c)}
// i.e. the Ritz vectors whose tridiagonal coordinates the host computed from the first pass.
// `y` receives n_levels planar buffers of b.vec_count payloads back to back (device memory).
int lanczos_ritz_block(bdg_system* sys, LanczosState* lz, int n_iter, int n_levels, const double* coef,
                       DeviceBuffer<double2>& y) {
    if (lz->iter != 0) return fail(BDG_EINVAL, "the Ritz-vector pass starts from a freshly begun process");
    if (n_iter < 1 || n_iter > lz->max_iter || n_levels < 1 || n_levels > 64)
        return fail(BDG_EINVAL, "bad iteration or level count");
    HIP_TRY(hipSetDevice(sys->device));
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    const size_t cols = (size_t)lz->cols, count = b.vec_count;
    DeviceBuffer<double> dev_coef;
    auto body = [&]() -> int {
        if (int rc = y.reserve((size_t)n_levels * count)) return rc;
        if (int rc = dev_coef.reserve((size_t)n_iter * n_levels * cols)) return rc;
        // coefficients padded to the buffer's column count (inactive columns: 0)
        std::vector<double> padded((size_t)n_iter * n_levels * cols, 0.0);
        for (int j = 0; j < n_iter; ++j)
            for (int l = 0; l < n_levels; ++l)
                for (int c = 0; c < lz->n_active; ++c)
                    padded[((size_t)j * n_levels + l) * cols + c] = coef[((size_t)j * n_levels + l) * lz->n_active + c];
        HIP_TRY(hipMemcpyAsync(dev_coef.ptr, padded.data(), sizeof(double) * padded.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(y.ptr, 0, sizeof(double2) * n_levels * count, st));
        bdg::StepArgs args = b.args;
        args.partial = sys->partial.ptr;
        const int grid = (int)std::min<size_t>(2048, (count + 255) / 256);
        for (int j = 0; j < n_iter; ++j) {
            const double* beta_j = lz->z.beta_hist + (size_t)j * cols;
            const double* coef_j = dev_coef.ptr + (size_t)j * n_levels * cols;
            if (b.real)
                bdg::lanczos_accumulate<2><<<grid, 256, 0, st>>>(lz->w_cur, beta_j, coef_j, n_levels, (int)cols, b.rl,
                                                                 (int64_t)count, y.ptr);
            else
                bdg::lanczos_accumulate<1><<<grid, 256, 0, st>>>(lz->w_cur, beta_j, coef_j, n_levels, (int)cols, b.rl,
                                                                 (int64_t)count, y.ptr);
            if (int rc = lanczos_iterate(sys, lz, args, j)) return rc;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));  // (the padded coefficients leave scope)
        lz->iter = n_iter;
        return BDG_OK;
    };
    const int rc = body();
    dev_coef.release();
    return rc;
}

// column `c` of planar block `src` -> host vector of 4*nb complex entries (site-major)
int lanczos_column_to_host(bdg_system* sys, LanczosState* lz, const double2* src, int c, double2* staging, double* out) {
    Batch& b = lz->batch;
    const size_t n = (size_t)4 * sys->nb;
    const int cgrid = (int)std::min<size_t>(4096, (n + 255) / 256);
    if (b.real)
        bdg::sitemajor_from_planar_real<<<cgrid, 256, 0, sys->stream>>>(src, staging, sys->nb, b.rl, c);
    else
        bdg::sitemajor_from_planar<<<cgrid, 256, 0, sys->stream>>>(src, staging, sys->nb, b.rl, c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, staging, sizeof(double2) * n, hipMemcpyDeviceToHost, sys->stream));
    return BDG_OK;
}

// y_out[l][c] is a site-major complex vector of 4*nb entries.
int lanczos_ritz_vectors(bdg_system* sys, int n_iter, int n_levels, const double* coef, double* y_out) {
    LanczosState* lz = static_cast<LanczosState*>(sys->lanczos);
    if (!lz) return fail(BDG_EINVAL, "bdg_lanczos_begin has not been called");
    DeviceBuffer<double2> y, host_order;
    auto body = [&]() -> int {
        if (int rc = lanczos_ritz_block(sys, lz, n_iter, n_levels, coef, y)) return rc;
        if (int rc = host_order.reserve((size_t)4 * sys->nb)) return rc;
        const size_t n = (size_t)4 * sys->nb, count = lz->batch.vec_count;
        for (int l = 0; l < n_levels; ++l)
            for (int c = 0; c < lz->n_active; ++c)
                if (int rc = lanczos_column_to_host(sys, lz, y.ptr + (size_t)l * count, c, host_order.ptr,
                                                    y_out + 2 * n * ((size_t)l * lz->n_active + c)))
                    return rc;
        HIP_TRY(hipStreamSynchronize(sys->stream));
        return BDG_OK;
    };
    const int rc = body();
    y.release();
    host_order.release();
    return rc;
}

// Eigen-decomposition of a small Hermitian matrix on the host (cyclic complex Jacobi; n <= 16 here).
// a: n x n complex, row-major, interleaved (destroyed); w: eigenvalues ascending; v: eigenvectors as COLUMNS.
void small_hermitian_eigh(int n, std::vector<double>& a, std::vector<double>& w, std::vector<double>& v) {
    auto A = [&](int r, int c) -> double* { return &a[2 * ((size_t)r * n + c)]; };
    v.assign((size_t)2 * n * n, 0.0);
    auto V = [&](int r, int c) -> double* { return &v[2 * ((size_t)r * n + c)]; };
    for (int i = 0; i < n; ++i) V(i, i)[0] = 1.0;
    double total = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) total += A(i, j)[0] * A(i, j)[0] + A(i, j)[1] * A(i, j)[1];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += A(p, q)[0] * A(p, q)[0] + A(p, q)[1] * A(p, q)[1];
        if (off <= 1e-32 * total) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double cr = A(p, q)[0], ci = A(p, q)[1], mag = std::hypot(cr, ci);
                if (mag <= 1e-300) continue;
                // J = diag(1, e^{-i phi}) * [[c, s], [-s, c]] makes (J^H A J)_pq = 0
                const double er = cr / mag, ei = ci / mag;  // e^{i phi}
                const double tau = (A(q, q)[0] - A(p, p)[0]) / (2.0 * mag);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = t * c;
                // columns: X[:, p] <- c X[:, p] - s e^{-i phi} X[:, q];  X[:, q] <- s X[:, p] + c e^{-i phi} X[:, q]
                auto rotate_columns = [&](auto X) {
                    for (int k = 0; k < n; ++k) {
                        const double pr = X(k, p)[0], pi = X(k, p)[1], qr = X(k, q)[0], qi = X(k, q)[1];
                        const double zr = er * qr + ei * qi, zi = er * qi - ei * qr;  // e^{-i phi} x_q
                        X(k, p)[0] = c * pr - sn * zr;
                        X(k, p)[1] = c * pi - sn * zi;
                        X(k, q)[0] = sn * pr + c * zr;
                        X(k, q)[1] = sn * pi + c * zi;
                    }
                };
                rotate_columns(A);
                rotate_columns(V);
                // rows of A: row p <- c row p - s e^{+i phi} row q;  row q <- s row p + c e^{+i phi} row q
                for (int k = 0; k < n; ++k) {
                    const double pr = A(p, k)[0], pi = A(p, k)[1], qr = A(q, k)[0], qi = A(q, k)[1];
                    const double zr = er * qr - ei * qi, zi = er * qi + ei * qr;  // e^{+i phi} a_qk
                    A(p, k)[0] = c * pr - sn * zr;
                    A(p, k)[1] = c * pi - sn * zi;
                    A(q, k)[0] = sn * pr + c * zr;
                    A(q, k)[1] = sn * pi + c * zi;
                }
            }
    }
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return A(x, x)[0] < A(y, y)[0]; });
    w.resize((size_t)n);
    std::vector<double> sorted((size_t)2 * n * n);
    for (int m = 0; m < n; ++m) {
        w[(size_t)m] = A(order[(size_t)m], order[(size_t)m])[0];
        for (int k = 0; k < n; ++k) {
            sorted[2 * ((size_t)k * n + m)] = V(k, order[(size_t)m])[0];
            sorted[2 * ((size_t)k * n + m) + 1] = V(k, order[(size_t)m])[1];
        }
    }
    v.swap(sorted);
}

// Rayleigh-Ritz of the partial spectrum without the Ritz block ever leaving the device (what
// observables.lowest_eigenpairs did with (levels, vectors, 4N) host arrays and host-side H products).
// Per level l, with Y = the level's block of Ritz vectors of H^2 (one column per start vector):
//   Z = (H + eps_l) Y                      candidates in the +eps eigenspace of H        (one K1 launch)
//   G = Z^H Z  -> host: rank r = multiplicity, orthonormal combinations  B = Z M            (block_gram, block_mix)
//   B <- B (B^H B)^{-1/2}                  round-off tidied (Loewdin)
//   S = B^H (H B) -> host: eigen-decomposition -> final = B R, energies                     (K1, block_gram, block_mix)
// Only r x r matrices cross PCIe on the way; at the end the `max_out` lowest states are copied out.
int lanczos_ritz_pairs(bdg_system* sys, int n_iter, int n_levels, const double* coef, const double* eps,
                       double rank_tol, int max_out, int* n_out, double* values_out, double* vectors_out) {
    LanczosState* lz = static_cast<LanczosState*>(sys->lanczos);
    if (!lz) return fail(BDG_EINVAL, "bdg_lanczos_begin has not been called");
    Batch& b = lz->batch;
    const int cols = lz->cols;
    if (cols > bdg::kBlockAlgebraMaxCols) return fail(BDG_EINVAL, "Rayleigh-Ritz on the device handles at most %d columns", bdg::kBlockAlgebraMaxCols);
    if (max_out < 0 || !n_out || (max_out > 0 && (!values_out || !vectors_out))) return fail(BDG_EINVAL, "bad output arguments");
    DeviceBuffer<double2> y, staging;
    DeviceBuffer<double> scal, gram_partial, gram, mix;
    constexpr int kGramGrid = 256;
    hipStream_t st = sys->stream;
    const size_t count = b.vec_count;
    const int64_t pairs = (int64_t)(count / b.rl);
    struct Found {
        double energy;
        int level, column;
    };
    std::vector<Found> found;

    auto gram_to_host = [&](const double2* a, const double2* bb, std::vector<double>& out) -> int {
        const dim3 grid(kGramGrid, (unsigned)cols);
        if (b.real) bdg::block_gram<2><<<grid, 256, 0, st>>>(a, bb, pairs, b.rl, gram_partial.ptr);
        else bdg::block_gram<1><<<grid, 256, 0, st>>>(a, bb, pairs, b.rl, gram_partial.ptr);
        bdg::reduce_partials<<<cols, 256, 0, st>>>(gram_partial.ptr, gram.ptr, kGramGrid, 2 * cols);
        HIP_TRY(hipGetLastError());
        out.resize((size_t)2 * cols * cols);
        HIP_TRY(hipMemcpyAsync(out.data(), gram.ptr, sizeof(double) * out.size(), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return BDG_OK;
    };
    auto mix_on_device = [&](const double2* a, const std::vector<double>& m, double2* out) -> int {
        HIP_TRY(hipMemcpyAsync(mix.ptr, m.data(), sizeof(double) * m.size(), hipMemcpyHostToDevice, st));
        const int grid = (int)std::min<int64_t>(4096, (pairs + 255) / 256);
        if (b.real) bdg::block_mix<2><<<grid, 256, 0, st>>>(a, mix.ptr, pairs, b.rl, out);
        else bdg::block_mix<1><<<grid, 256, 0, st>>>(a, mix.ptr, pairs, b.rl, out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));  // (m may leave scope)
        return BDG_OK;
    };
    // out <- coef * H x - pscale * out, all columns with the same scalars (the Lanczos batch's K1 kernel)
    auto apply_h = [&](const double2* x, double2* out, double coef_value, double pscale_value) -> int {
        std::vector<double> host((size_t)2 * cols);
        for (int c = 0; c < cols; ++c) host[(size_t)c] = coef_value, host[(size_t)cols + c] = pscale_value;
        HIP_TRY(hipMemcpyAsync(scal.ptr, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        bdg::StepArgs args = b.args;
        args.partial = sys->partial.ptr;
        args.cur = x;
        args.prev = out;
        args.col_coef = scal.ptr;
        args.col_pscale = scal.ptr + cols;
        b.plan.kernel<<<b.plan.grid, bdg::kBlockThreads, b.plan.lds_bytes, st>>>(args);
        HIP_TRY(hipGetLastError());
        return BDG_OK;
    };
    // orthonormal combinations of the columns of a block from its Gram matrix: M = U diag(w^-1/2) [U^H] restricted
    // to eigenvalues above cut * max; `symmetric` = Loewdin (M = U w^-1/2 U^H, keeps the columns where they are)
    auto combinations = [&](std::vector<double>& g, const std::vector<int>& use, double cut, bool symmetric,
                            std::vector<double>& m, int* rank) {
        const int n = (int)use.size();
        std::vector<double> small((size_t)2 * n * n), w, u;
        std::vector<double> norm((size_t)n);
        for (int i = 0; i < n; ++i) norm[(size_t)i] = std::sqrt(std::max(g[2 * ((size_t)use[(size_t)i] * cols + use[(size_t)i])], 0.0));
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double scale = 1.0 / (norm[(size_t)i] * norm[(size_t)j]);
                small[2 * ((size_t)i * n + j)] = g[2 * ((size_t)use[(size_t)i] * cols + use[(size_t)j])] * scale;
                small[2 * ((size_t)i * n + j) + 1] = g[2 * ((size_t)use[(size_t)i] * cols + use[(size_t)j]) + 1] * scale;
            }
        small_hermitian_eigh(n, small, w, u);
        m.assign((size_t)2 * cols * cols, 0.0);
        int kept = 0;
        for (int e = n - 1; e >= 0; --e) {  // largest weights first: output columns 0 .. rank-1
            if (!(w[(size_t)e] > cut * w[(size_t)n - 1])) break;
            const double inv = 1.0 / std::sqrt(w[(size_t)e]);
            for (int i = 0; i < n; ++i) {
                const double ur = u[2 * ((size_t)i * n + e)], ui = u[2 * ((size_t)i * n + e) + 1];
                if (!symmetric) {
                    m[2 * ((size_t)use[(size_t)i] * cols + kept)] = ur * inv / norm[(size_t)i];
                    m[2 * ((size_t)use[(size_t)i] * cols + kept) + 1] = ui * inv / norm[(size_t)i];
                } else {
                    for (int j = 0; j < n; ++j) {  // += u_ie w^-1/2 conj(u_je) / norm_i   (B D^-1 G'^-1/2, D = diag(norm))
                        const double vr = u[2 * ((size_t)j * n + e)], vi = -u[2 * ((size_t)j * n + e) + 1];
                        m[2 * ((size_t)use[(size_t)i] * cols + use[(size_t)j])] += (ur * vr - ui * vi) * inv / (norm[(size_t)i]);
                        m[2 * ((size_t)use[(size_t)i] * cols + use[(size_t)j]) + 1] += (ur * vi + ui * vr) * inv / (norm[(size_t)i]);
                    }
                }
            }
            ++kept;
        }
        *rank = kept;
    };

    auto body = [&]() -> int {
        if (int rc = lanczos_ritz_block(sys, lz, n_iter, n_levels, coef, y)) return rc;
        if (int rc = scal.reserve((size_t)2 * cols)) return rc;
        if (int rc = gram_partial.reserve((size_t)kGramGrid * 2 * cols * cols)) return rc;
        if (int rc = gram.reserve((size_t)2 * cols * cols)) return rc;
        if (int rc = mix.reserve((size_t)2 * cols * cols)) return rc;
        // the three vector buffers of the finished process serve as work space
        double2* z = sys->vec_a.ptr;
        double2* hb = sys->vec_b.ptr;
        std::vector<double> g, m;
        for (int l = 0; l < n_levels; ++l) {
            double2* yl = y.ptr + (size_t)l * count;
            // Z = H Y + eps Y
            HIP_TRY(hipMemcpyAsync(z, yl, sizeof(double2) * count, hipMemcpyDeviceToDevice, st));
            if (int rc = apply_h(yl, z, 1.0, -eps[l])) return rc;
            if (int rc = gram_to_host(z, z, g)) return rc;
            // a Ritz vector may lie (almost) wholly in the -eps eigenspace: its projection is round-off noise
            double longest = 0.0;
            for (int c = 0; c < cols; ++c) longest = std::max(longest, g[2 * ((size_t)c * cols + c)]);
            std::vector<int> use;
            for (int c = 0; c < cols; ++c)
                if (g[2 * ((size_t)c * cols + c)] > 1e-12 * longest && longest > 0.0) use.push_back(c);
            if (use.empty()) continue;
            int rank = 0;
            combinations(g, use, rank_tol, false, m, &rank);
            if (rank == 0) continue;
            if (int rc = mix_on_device(z, m, z)) return rc;            // B: columns 0 .. rank-1, the rest zero
            if (int rc = gram_to_host(z, z, g)) return rc;             // tidy up round-off (Loewdin)
            std::vector<int> first((size_t)rank);
            for (int c = 0; c < rank; ++c) first[(size_t)c] = c;
            int again = 0;
            combinations(g, first, 0.0, true, m, &again);
            if (int rc = mix_on_device(z, m, z)) return rc;
            // S = B^H H B
            HIP_TRY(hipMemsetAsync(hb, 0, sizeof(double2) * count, st));
            if (int rc = apply_h(z, hb, 1.0, 0.0)) return rc;
            if (int rc = gram_to_host(z, hb, g)) return rc;
            std::vector<double> small((size_t)2 * rank * rank), w, u;
            for (int i = 0; i < rank; ++i)
                for (int j = 0; j < rank; ++j) {  // Hermitian part
                    small[2 * ((size_t)i * rank + j)] = 0.5 * (g[2 * ((size_t)i * cols + j)] + g[2 * ((size_t)j * cols + i)]);
                    small[2 * ((size_t)i * rank + j) + 1] = 0.5 * (g[2 * ((size_t)i * cols + j) + 1] - g[2 * ((size_t)j * cols + i) + 1]);
                }
            small_hermitian_eigh(rank, small, w, u);
            m.assign((size_t)2 * cols * cols, 0.0);
            for (int i = 0; i < rank; ++i)
                for (int e = 0; e < rank; ++e) {
                    m[2 * ((size_t)i * cols + e)] = u[2 * ((size_t)i * rank + e)];
                    m[2 * ((size_t)i * cols + e) + 1] = u[2 * ((size_t)i * rank + e) + 1];
                }
            if (int rc = mix_on_device(z, m, yl)) return rc;            // final states of the level replace its Ritz block
            for (int e = 0; e < rank; ++e) found.push_back({w[(size_t)e], l, e});
        }
        std::stable_sort(found.begin(), found.end(), [](const Found& p, const Found& q) { return p.energy < q.energy; });
        *n_out = (int)found.size();
        const int n_copy = std::min<int>(max_out, (int)found.size());
        if (n_copy > 0)
            if (int rc = staging.reserve((size_t)4 * sys->nb)) return rc;
        const size_t n = (size_t)4 * sys->nb;
        for (int k = 0; k < n_copy; ++k) {
            values_out[k] = found[(size_t)k].energy;
            if (int rc = lanczos_column_to_host(sys, lz, y.ptr + (size_t)found[(size_t)k].level * count, found[(size_t)k].column,
                                                staging.ptr, vectors_out + 2 * n * (size_t)k))
                return rc;
            HIP_TRY(hipStreamSynchronize(st));  // (one staging buffer)
        }
        return BDG_OK;
    };
    HIP_TRY(hipSetDevice(sys->device));
    const int rc = body();
    y.release();
    staging.release();
    scal.release();
    gram_partial.release();
    gram.release();
    mix.release();
    lanczos_free(sys);  // the process's vector buffers were used as work space: the run is over
    return rc;
}

int lanczos_advance(bdg_system* sys, int n_iter, double* alpha_out, double* beta_out) {
    LanczosState* lz = static_cast<LanczosState*>(sys->lanczos);
    if (!lz) return fail(BDG_EINVAL, "bdg_lanczos_begin has not been called");
    if (n_iter < 1 || lz->iter + n_iter > lz->max_iter)
        return fail(BDG_EINVAL, "iteration count exceeds the limit given to bdg_lanczos_begin");
    HIP_TRY(hipSetDevice(sys->device));
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    bdg::StepArgs args = b.args;
    args.partial = sys->partial.ptr;
    const int first = lz->iter;
    for (int j = first; j < first + n_iter; ++j)
        if (int rc = lanczos_iterate(sys, lz, args, j)) return rc;
    HIP_TRY(hipGetLastError());
    lz->iter += n_iter;
    const size_t cols = (size_t)lz->cols;
    std::vector<double> alpha((size_t)n_iter * cols), beta((size_t)n_iter * cols);
    HIP_TRY(hipMemcpyAsync(alpha.data(), lz->z.alpha_hist + (size_t)first * cols, sizeof(double) * alpha.size(),
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(beta.data(), lz->z.beta_hist + ((size_t)first + 1) * cols, sizeof(double) * beta.size(),
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int j = 0; j < n_iter; ++j)
        for (int r = 0; r < lz->n_active; ++r) {
            alpha_out[(size_t)j * lz->n_active + r] = alpha[(size_t)j * cols + r];
            beta_out[(size_t)j * lz->n_active + r] = beta[(size_t)j * cols + r];
        }
    return BDG_OK;
}

}  // namespace
