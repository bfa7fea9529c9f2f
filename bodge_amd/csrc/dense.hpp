// dense.hpp - own one-sided Jacobi eigensolver (the rocSOLVER route is in bdg_eigh_dense)
// Part of the single translation unit bodge_hip.hip (included there, in this order:
// core, plans, libraries, recurrence, lanczos, dense); everything lives in its unnamed namespace.
#pragma once

namespace {

// ------------------------------------------------------------- dense eigensolver
// One-sided Jacobi on the GPU for matrices up to kJacobiLimit (no external library: the first
// use of rocSOLVER on a fresh machine pages in ~1 GB and was measured at 1.5-7.5 minutes).
constexpr int64_t kJacobiLimit = 256 * bdg::kJacobiElems;  // 2048: a column pair fits the registers of a workgroup
// With 16 elements per thread the same kernels reach 4096 (n = 3600: ~3 s against rocSOLVER's 0.13 s):
// used while the rocSOLVER object has not arrived from cold storage yet, which takes minutes.
constexpr int64_t kJacobiWideLimit = 256 * bdg::kJacobiElemsWide;

inline void scatter_for_jacobi(bdg_system* sys, double2* G, hipStream_t st) {
    bdg::scatter_dense<<<(unsigned)sys->nb, 128, 0, st>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, G,
                                                           (int)sys->nb);
}
inline void scatter_for_jacobi(bdg_system* sys, double* G, hipStream_t st) {
    bdg::scatter_dense_real<<<(unsigned)sys->nb, 128, 0, st>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr,
                                                                G, (int)sys->nb);
}

// T = double2 (Hermitian) or double (real symmetric, when imag(H) = 0: half the bytes per round)
template <typename T>
int eigh_jacobi_typed(bdg_system* sys, double* w_out, double* z_out) {
    const int64_t n = 4 * sys->nb;
    if (n > kJacobiWideLimit)
        return fail(BDG_EINVAL, "the Jacobi kernels hold a column pair in registers: 4*nb <= %d", (int)kJacobiWideLimit);
    const bool wide = n > kJacobiLimit;
    hipStream_t st = sys->stream;
    DeviceBuffer<T> G, V;
    DeviceBuffer<double> eig;
    DeviceBuffer<int> counter;
    auto body = [&]() -> int {
        if (int rc = G.reserve((size_t)n * n)) return rc;
        if (z_out)
            if (int rc = V.reserve((size_t)n * n)) return rc;
        if (int rc = eig.reserve((size_t)n)) return rc;
        if (int rc = counter.reserve(1)) return rc;
        const double shift = 1.5 * sys->gershgorin + 1.0;  // spectrum of G in [0.5 b + 1, 2.5 b + 1]
        HIP_TRY(hipMemsetAsync(G.ptr, 0, sizeof(T) * n * n, st));
        scatter_for_jacobi(sys, G.ptr, st);
        bdg::jacobi_setup<T><<<(unsigned)std::min<int64_t>(4096, (n * n + 255) / 256), 256, 0, st>>>(
            G.ptr, z_out ? V.ptr : nullptr, (int)n, shift);
        HIP_TRY(hipGetLastError());
        const int max_sweeps = 40;
        int sweep = 0;
        for (; sweep < max_sweeps; ++sweep) {
            HIP_TRY(hipMemsetAsync(counter.ptr, 0, sizeof(int), st));
            for (int round = 0; round < n - 1; ++round) {
                if (wide)
                    bdg::jacobi_round<T, bdg::kJacobiElemsWide><<<(unsigned)(n / 2), 256, 0, st>>>(
                        G.ptr, z_out ? V.ptr : nullptr, (int)n, round, 1e-15, counter.ptr);
                else
                    bdg::jacobi_round<T, bdg::kJacobiElems><<<(unsigned)(n / 2), 256, 0, st>>>(
                        G.ptr, z_out ? V.ptr : nullptr, (int)n, round, 1e-15, counter.ptr);
            }
            int rotations = 0;
            HIP_TRY(hipMemcpyAsync(&rotations, counter.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (rotations == 0) break;
        }
        if (sweep == max_sweeps) return fail(BDG_ELIBRARY, "Jacobi eigensolver did not converge in %d sweeps", max_sweeps);
        bdg::jacobi_eigenvalues<T><<<(unsigned)n, 256, 0, st>>>(G.ptr, (int)n, shift, eig.ptr);
        HIP_TRY(hipGetLastError());
        std::vector<double> vals((size_t)n);
        HIP_TRY(hipMemcpyAsync(vals.data(), eig.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<int64_t> order((size_t)n);
        for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return vals[(size_t)a] < vals[(size_t)b]; });
        for (int64_t i = 0; i < n; ++i) w_out[i] = vals[(size_t)order[(size_t)i]];
        if (z_out) {
            constexpr size_t kScalars = sizeof(T) / sizeof(double);  // 2 complex, 1 real
            std::vector<double> cols(kScalars * n * n);
            HIP_TRY(hipMemcpy(cols.data(), V.ptr, sizeof(T) * n * n, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; ++i) {
                const double* src = cols.data() + kScalars * n * order[(size_t)i];
                double* dst = z_out + 2 * n * i;
                if (kScalars == 2) {
                    memcpy(dst, src, sizeof(double) * 2 * n);
                } else {
                    for (int64_t k = 0; k < n; ++k) {
                        dst[2 * k] = src[k];
                        dst[2 * k + 1] = 0.0;
                    }
                }
            }
        }
        return BDG_OK;
    };
    const int rc = body();
    G.release();
    V.release();
    eig.release();
    counter.release();
    return rc;
}

int eigh_jacobi(bdg_system* sys, double* w_out, double* z_out) {
    bool real_route = sys->is_real;
    if (const char* env = knob::raw("BODGE_AMD_EIGH_REAL")) real_route = real_route && atoi(env) != 0;
    return real_route ? eigh_jacobi_typed<double>(sys, w_out, z_out) : eigh_jacobi_typed<double2>(sys, w_out, z_out);
}

}  // namespace
