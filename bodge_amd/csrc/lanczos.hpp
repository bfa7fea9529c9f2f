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
// y_out[l][c] is a site-major complex vector of 4*nb entries.
int lanczos_ritz_vectors(bdg_system* sys, int n_iter, int n_levels, const double* coef, double* y_out) {
    LanczosState* lz = static_cast<LanczosState*>(sys->lanczos);
    if (!lz) return fail(BDG_EINVAL, "bdg_lanczos_begin has not been called");
    if (lz->iter != 0) return fail(BDG_EINVAL, "the Ritz-vector pass starts from a freshly begun process");
    if (n_iter < 1 || n_iter > lz->max_iter || n_levels < 1 || n_levels > 64)
        return fail(BDG_EINVAL, "bad iteration or level count");
    HIP_TRY(hipSetDevice(sys->device));
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    const size_t cols = (size_t)lz->cols, count = b.vec_count;
    DeviceBuffer<double2> y, host_order;
    DeviceBuffer<double> dev_coef;
    auto body = [&]() -> int {
        if (int rc = y.reserve((size_t)n_levels * count)) return rc;
        if (int rc = host_order.reserve((size_t)4 * sys->nb)) return rc;
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
        lz->iter = n_iter;
        const size_t n = (size_t)4 * sys->nb;
        const int cgrid = (int)std::min<size_t>(4096, (n + 255) / 256);
        for (int l = 0; l < n_levels; ++l)
            for (int c = 0; c < lz->n_active; ++c) {
                const double2* src = y.ptr + (size_t)l * count;
                if (b.real)
                    bdg::sitemajor_from_planar_real<<<cgrid, 256, 0, st>>>(src, host_order.ptr, sys->nb, b.rl, c);
                else
                    bdg::sitemajor_from_planar<<<cgrid, 256, 0, st>>>(src, host_order.ptr, sys->nb, b.rl, c);
                HIP_TRY(hipMemcpyAsync(y_out + 2 * n * ((size_t)l * lz->n_active + c), host_order.ptr, sizeof(double2) * n,
                                       hipMemcpyDeviceToHost, st));
            }
        HIP_TRY(hipStreamSynchronize(st));
        return BDG_OK;
    };
    const int rc = body();
    y.release();
    host_order.release();
    dev_coef.release();
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
