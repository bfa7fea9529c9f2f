// host_assembly.hpp - CPU-thread helpers for the matrix that feeds the device path.  No GPU call.
//
// The reference fills its BSR blocks with one Python assignment per term (bodge/hamiltonian.py:102-118,
// 134 s for 10^6 sites, misc/benchmark.csv:40) and tests Hermiticity with a sparse M - M^H (:121-122).
// The host side of this package does the same work on whole arrays; for a 10^6-site lattice that
// is several passes over 1.28 GB of blocks, which numpy makes with one thread and index-array
// scatters.  These three functions make them with all cores:
//
//   fill_terms      scatter 2x2 spin matrices into the 4x4 Nambu blocks (:106-108, :112-116)
//   scan_blocks     one read of every block: which are all-zero (dropped at export, :142-143), the
//                   particle-hole defect max |D + A*|, |C + B*| of the blocks [[A, B], [C, D]], the
//                   Gershgorin bound, imag == 0
//   compact_blocks  the BSR triple without its all-zero blocks
//
// Results are the same numbers numpy produces: the fills are copies and sign flips, the bound sums
// in numpy's order (left to right within a block, np.add.reduceat's pairwise order over a block row).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "knobs.hpp"

namespace bdg_host {

inline int thread_count(int64_t items, int64_t items_per_thread) {
    int want = (int)std::thread::hardware_concurrency();
    if (want < 1) want = 1;
    want = std::min(want, 32);
    if (const char* env = knob::raw("BODGE_AMD_HOST_THREADS")) want = std::max(1, std::atoi(env));
    const int64_t useful = std::max<int64_t>(1, items / std::max<int64_t>(1, items_per_thread));
    return (int)std::min<int64_t>(want, useful);
}

// f(t, lo, hi) on `threads` contiguous pieces of [0, n)
template <typename F>
inline void parallel_ranges(int64_t n, int threads, F f) {
    if (threads <= 1) {
        f(0, (int64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads);
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([=]() { f(t, n * t / threads, n * (t + 1) / threads); });
    for (auto& th : pool) th.join();
}

// kind 0: hopping  H_ij -> blk[0:2,0:2] = v, blk[2:4,2:4] = -conj(v)          (ref :106-108)
// kind 1: pairing  Δ_ij -> blk[0:2,2:4] = v                                    (ref :112-113)
// kind 2: pairing, transposed block: blk[2:4,0:2] = v^†                         (ref :115-116)
// Terms are applied in order (a later term that names the same block wins, as with a numpy index
// assignment).  Threads own disjoint ranges of DESTINATION blocks and each reads the whole id list,
// so there is no race and the pages of a fresh `data` are first touched by many cores.
inline void fill_terms(double* data, int64_t nnzb, const int64_t* ids, int64_t count, const double* values,
                       bool per_term, int kind, uint8_t* touched) {
    const int threads = thread_count(count, 1 << 16);
    parallel_ranges(nnzb, threads, [=](int, int64_t lo, int64_t hi) {
        for (int64_t n = 0; n < count; ++n) {
            const int64_t k = ids[n];
            if (k < lo || k >= hi) continue;
            const double* v = values + (per_term ? 8 * n : 0);  // v00, v01, v10, v11 as (re, im)
            double* blk = data + 32 * k;
            if (kind == 0) {
                for (int r = 0; r < 2; ++r)
                    for (int c = 0; c < 2; ++c) {
                        const double re = v[2 * (2 * r + c)], im = v[2 * (2 * r + c) + 1];
                        blk[2 * (4 * r + c)] = re;
                        blk[2 * (4 * r + c) + 1] = im;
                        blk[2 * (4 * (r + 2) + c + 2)] = -re;  // -conj(v): (-re, +im)
                        blk[2 * (4 * (r + 2) + c + 2) + 1] = im;
                    }
            } else if (kind == 1) {
                for (int r = 0; r < 2; ++r)
                    for (int c = 0; c < 2; ++c) {
                        blk[2 * (4 * r + c + 2)] = v[2 * (2 * r + c)];
                        blk[2 * (4 * r + c + 2) + 1] = v[2 * (2 * r + c) + 1];
                    }
            } else {
                for (int r = 0; r < 2; ++r)
                    for (int c = 0; c < 2; ++c) {
                        blk[2 * (4 * (r + 2) + c)] = v[2 * (2 * c + r)];  // conj(v)^T
                        blk[2 * (4 * (r + 2) + c) + 1] = -v[2 * (2 * c + r) + 1];
                    }
            }
            if (touched) touched[k] = 1;
        }
    });
}

// numpy's float64 add.reduce over n strided values (its pairwise summation): the Gershgorin row
// sums below repeat np.add.reduceat's arithmetic so that the bound is the same double the numpy
// form `chebyshev.spectral_bound` gives (checked in tests/test_assembly.py; entries with both a real
// and an imaginary part may differ in the last bit: numpy's vectorised |z| is not libm's hypot).
inline double numpy_pairwise_sum(const double* a, int64_t n, int64_t stride) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i * stride];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[(i + j) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i * stride];
        return res;
    }
    int64_t half = n / 2;
    half -= half % 8;
    return numpy_pairwise_sum(a, half, stride) + numpy_pairwise_sum(a + half * stride, n - half, stride);
}

struct BlockScan {
    int64_t n_nonzero = 0;
    double ph_defect = 0.0;   // max |blk[2:4,2:4] + conj(blk[0:2,0:2])|, |blk[2:4,0:2] + conj(blk[0:2,2:4])|
    double row_sum_max = 0.0; // max over scalar rows of Σ|H_rc|
    bool all_real = true;
    bool has_nan = false;
};

inline BlockScan scan_blocks(const double* data, const int32_t* indptr, int64_t nb, uint8_t* nonzero) {
    const int64_t nnzb = indptr[nb];
    const int threads = thread_count(nnzb, 1 << 15);
    std::vector<BlockScan> part((size_t)std::max(threads, 1));
    parallel_ranges(nb, threads, [&, data, indptr, nonzero](int t, int64_t lo, int64_t hi) {
        BlockScan acc;
        std::vector<double> sums;  // per block of the row: Σ_c |blk[r, c]|, r = 0..3
        for (int64_t i = lo; i < hi; ++i) {
            sums.resize((size_t)(4 * (indptr[i + 1] - indptr[i])));
            double* sum_at = sums.data();
            for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k, sum_at += 4) {
                const double* blk = data + 32 * k;
                bool any = false, real = true;
                for (int e = 0; e < 16; ++e) {
                    any = any || blk[2 * e] != 0.0 || blk[2 * e + 1] != 0.0;
                    real = real && blk[2 * e + 1] == 0.0;
                }
                if (nonzero) nonzero[k] = any ? 1 : 0;
                acc.n_nonzero += any ? 1 : 0;
                acc.all_real = acc.all_real && real;
                for (int r = 0; r < 2; ++r)
                    for (int c = 0; c < 2; ++c) {
                        const double* a = blk + 2 * (4 * r + c);            // A
                        const double* b = blk + 2 * (4 * r + c + 2);        // B
                        const double* cc = blk + 2 * (4 * (r + 2) + c);     // C
                        const double* d = blk + 2 * (4 * (r + 2) + c + 2);  // D
                        const double da = std::hypot(d[0] + a[0], d[1] - a[1]);
                        const double db = std::hypot(cc[0] + b[0], cc[1] - b[1]);
                        if (std::isnan(da) || std::isnan(db)) acc.has_nan = true;
                        acc.ph_defect = std::max(acc.ph_defect, std::max(da, db));
                    }
                for (int r = 0; r < 4; ++r) {
                    double s = 0.0;  // numpy: abs(block).sum(axis=-1), left to right
                    for (int c = 0; c < 4; ++c) {
                        const double* z = blk + 2 * (4 * r + c);
                        const double mag = z[1] == 0.0 ? std::fabs(z[0]) : std::hypot(z[0], z[1]);
                        s = c == 0 ? mag : s + mag;
                    }
                    sum_at[r] = s;
                }
            }
            const int64_t len = indptr[i + 1] - indptr[i];
            for (int r = 0; r < 4 && len > 0; ++r) {  // np.add.reduceat: first block + pairwise sum of the others
                const double v = sums[(size_t)r] + numpy_pairwise_sum(sums.data() + 4 + r, len - 1, 4);
                if (std::isnan(v)) acc.has_nan = true;
                acc.row_sum_max = std::max(acc.row_sum_max, v);
            }
        }
        part[(size_t)t] = acc;
    });
    BlockScan out;
    for (const BlockScan& p : part) {
        out.n_nonzero += p.n_nonzero;
        out.ph_defect = std::max(out.ph_defect, p.ph_defect);
        out.row_sum_max = std::max(out.row_sum_max, p.row_sum_max);
        out.all_real = out.all_real && p.all_real;
        out.has_nan = out.has_nan || p.has_nan;
    }
    return out;
}

// (indptr, indices, data) without the blocks whose keep flag is 0; outputs sized by the caller
inline void compact_blocks(const double* data, const int32_t* indices, const int32_t* indptr, int64_t nb,
                           const uint8_t* keep, double* data_out, int32_t* indices_out, int32_t* indptr_out) {
    const int threads = thread_count(indptr[nb], 1 << 15);
    std::vector<int64_t> kept((size_t)threads + 1, 0);
    parallel_ranges(nb, threads, [&, keep, indptr](int t, int64_t lo, int64_t hi) {
        int64_t n = 0;
        for (int64_t k = indptr[lo]; k < indptr[hi]; ++k) n += keep[k] ? 1 : 0;
        kept[(size_t)t + 1] = n;
    });
    for (int t = 0; t < threads; ++t) kept[(size_t)t + 1] += kept[(size_t)t];
    indptr_out[0] = 0;
    parallel_ranges(nb, threads, [&, data, indices, indptr, keep, data_out, indices_out, indptr_out](int t, int64_t lo, int64_t hi) {
        int64_t at = kept[(size_t)t];
        for (int64_t i = lo; i < hi; ++i) {
            for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k)
                if (keep[k]) {
                    std::memcpy(data_out + 32 * at, data + 32 * k, 256);
                    indices_out[at] = indices[k];
                    ++at;
                }
            indptr_out[i + 1] = (int32_t)at;
        }
    });
}

}  // namespace bdg_host
