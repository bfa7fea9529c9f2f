// twostage.hpp - K10: real symmetric dense matrix -> band -> tridiagonal, the BLAS-3 route of the dense eigenvalue solve.
//
// Why: the one-stage tridiagonalisation of tridiag.hpp (K9) multiplies the trailing block with a vector once per column:
// n^3 / 3 matrix elements streamed from HBM, 1.1 s at n = 10^4 whatever the arithmetic costs.  The dense eigenproblem is
// the one place on the path that is a dense contraction (4 n^3 / 3 flop), so here it is run as one:
//   stage 1  dense -> band of half-width B = 32 by block Householder panels.  Per panel of B columns: a Householder QR of
//            the sub-diagonal panel (ts_panel_qr: one thread per row, the panel in registers, one grid barrier per column),
//            then the two-sided update  A22 <- A22 - V W^T - W V^T  with  X = A22 V  and the rank-2B update as fp64 MFMA
//            products (v_mfma_f64_16x16x4_f64, LDS-staged tiles): the trailing block is read twice and written once per B
//            columns instead of per column.
//   stage 2  band -> tridiagonal by bulge chasing (ts_chase): sweep j annihilates column j below the first sub-diagonal
//            with a reflector of B rows and chases the bulge down the band in steps of B rows.  One wave per sweep, the
//            three B x B blocks of a step in LDS; sweep j + 1 follows sweep j three steps behind (scratch/
//            r4_twostage_proto.py: any interleaving that keeps that distance gives the bits of the sequential order), told by
//            one progress word per sweep.  Sweeps are claimed by ticket in order, so the sweep a wave waits for is always
//            running: nothing has to be co-resident.  Band entries move through write-through (sc1) stores and sc1 loads.
// The tridiagonal matrix then goes to the bisection of tridiag.hpp.  Eigenvalues only: the eigenvectors of diagonalize()
// keep the one-stage route, whose reflectors its back-transformation knows.
// Restates LAPACK's dsytrd / dsbtrd in the two-stage form of Bischof, Lang and Sun (SBR toolbox); the reference itself
// calls scipy.linalg.eigvalsh (hamiltonian.py:302).
#pragma once

#include "kernels.hpp"

namespace bdg {

constexpr int kTsBand = 32;            // B: half-width of the band, columns per panel
constexpr int kTsBandLd = 2 * kTsBand + 2;  // doubles per column of the band storage: d = row - column = 0 .. 2B (bulge), padded

typedef double v4f64 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ grid barrier
// All workgroups of ts_panel_qr meet once per column.  counter counts arrivals over the whole launch (zeroed before it);
// what was stored before the barrier with agent-scope stores is read after it with agent-scope loads (guide: G16, the
// row "agent-scope atomic adds by one lane of each storing workgroup / sc1 poll / workgroup barrier before every load").
__device__ inline void ts_grid_barrier(unsigned* counter, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

// Hand-over data (QR partials, band entries): write-through stores and L1-bypassing loads (sc1), as raw buffer accesses -
// to the compiler ordinary memory operations that it may issue back to back (relaxed atomics are kept in program order
// one by one: 40 dependent L2 round trips per column step of the panel QR).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct TsBuffer {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ inline TsBuffer(const double* base, size_t count)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, (int)(count * sizeof(double)), 0x00020000)) {}
    __device__ inline double load(size_t index) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(index * sizeof(double)), 0, 16));
    }
    __device__ inline void store(size_t index, double v) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsrc, (int)(index * sizeof(double)), 0, 16);
    }
};
__device__ inline void ts_compiler_fence() { asm volatile("" ::: "memory"); }

// ------------------------------------------------------------------------------------------------ stage 1: panel QR
// Householder QR of the panel P = A[r0 .. n-1][j0 .. j0+B-1] (m = n - r0 rows), one thread per row with its B entries in
// registers.  Step i: reflector from column i, rows i .. m-1.  What a step needs from all rows are the sums
//   S_c = sum_{g > i} P[g][i] P[g][c]  (c = i .. B-1)   and   z_k = sum_g V[g][k] v_{i-1}[g]  (k < i-1, for the T factor)
// - one workgroup reduction, one set of partials per workgroup, one grid barrier.  Row i itself (its entries i .. B-1)
// travels the same way.  Outputs: V (m x B, unit lower trapezoidal, explicit), T (B x B upper triangular, Q = I - V T V^T),
// R into A (both triangles), zeros below it.
struct TsPanelArgs {
    double* a;       // n x n row-major symmetric, both triangles kept
    int n, j0, r0, m;
    int reflectors;  // min(B, m - 1)
    double* v;       // m x B
    double* t;       // B x B (row-major)
    double* partial; // [2][grid][B + 1 + B]: S_c, then the row-i broadcast is separate
    double* rowi;    // [2][B]
    unsigned* counter;
};

__global__ __launch_bounds__(256) void ts_panel_qr(TsPanelArgs q) {
    constexpr int B = kTsBand;
    __shared__ double red[4][2 * B];
    __shared__ double tot[2 * B];
    __shared__ double rowv[B];
    __shared__ double tmat[B][B + 1];
    __shared__ double taus[B];
    const int g = blockIdx.x * 256 + threadIdx.x;  // row of the panel
    const bool live = g < q.m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double p[B];
#pragma unroll
    for (int c = 0; c < B; ++c) p[c] = live ? q.a[(size_t)(q.r0 + g) * q.n + q.j0 + c] : 0.0;
    for (int e = threadIdx.x; e < B * (B + 1); e += 256) (&tmat[0][0])[e] = 0.0;
    double vprev = 0.0;  // this row's entry of the previous reflector
    unsigned phase = 0;
    const TsBuffer parts(q.partial, (size_t)2 * 256 * 2 * B), rows(q.rowi, 2 * B);
    // (unrolled: the column index must be a compile-time constant, or p[] lives in scratch)
#pragma unroll
    for (int i = 0; i <= B; ++i) {
        if (i > q.reflectors) continue;  // (uniform; no early exit: the loop has to unroll completely)
        // ---- partial sums of this step: S_c (c >= i) of column i, and z_k (k < i-1) of reflector i-1
#pragma unroll
        for (int c = 0; c < 2 * B; ++c) {
            double s = 0.0;
            bool wanted = false;
            if (c < B) {
                wanted = i < B && c >= i;
                if (wanted && i < q.reflectors && live && g > i) s = p[i < B ? i : 0] * p[c];
            } else {
                const int k = c - B;
                wanted = i >= 1 && k < i - 1;
                // V[g][k] for k < i-1: the stored entries of earlier reflectors (below the diagonal p[k], on it 1, above 0)
                if (wanted && live) s = (g > k ? p[k] : (g == k ? 1.0 : 0.0)) * vprev;
            }
            if (!wanted) continue;  // (known at compile time)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) red[wave][c] = s;
        }
        __syncthreads();
        const size_t part_at = ((size_t)(phase & 1) * gridDim.x + blockIdx.x) * 2 * B;
        if (threadIdx.x < 2 * B) parts.store(part_at + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (live && g == i && i < q.reflectors) {
#pragma unroll
            for (int c = 0; c < B; ++c) rows.store((phase & 1) * B + c, p[c]);
        }
        ts_grid_barrier(q.counter, (phase + 1) * gridDim.x);
        ts_compiler_fence();
        if (threadIdx.x < 2 * B) {
            double s = 0.0;
            const size_t all = (size_t)(phase & 1) * gridDim.x * 2 * B + threadIdx.x;
            for (unsigned w = 0; w < gridDim.x; ++w) s += parts.load(all + (size_t)w * 2 * B);
            tot[threadIdx.x] = s;
        } else if (threadIdx.x < 3 * B) {
            rowv[threadIdx.x - 2 * B] = rows.load((phase & 1) * B + threadIdx.x - 2 * B);
        }
        __syncthreads();
        // ---- T column of reflector i-1:  T[:i-1, i-1] = -tau_{i-1} T[:i-1, :i-1] z,  T[i-1][i-1] = tau_{i-1}
        if (i >= 1 && blockIdx.x == 0 && threadIdx.x < B) {
            const int col = i - 1, row = threadIdx.x;
            if (row < col) {
                double s = 0.0;
                for (int k = row; k < col; ++k) s += tmat[row][k] * tot[B + k];
                tmat[row][col] = -taus[col] * s;
            } else if (row == col) {
                tmat[row][col] = taus[col];
            }
        }
        if (i == q.reflectors || i == B) continue;
        // ---- reflector i
        const double alpha = rowv[i < B ? i : 0];
        const double sigma = tot[i < B ? i : 0];
        double beta = alpha, tau = 0.0, scale = 0.0;
        if (sigma > 0.0) {
            beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (threadIdx.x == 0) taus[i < B ? i : 0] = tau;
        vprev = 0.0;
        if (live && g >= i) {
            const double vg = g == i ? 1.0 : p[i < B ? i : 0] * scale;
#pragma unroll
            for (int c = 0; c < B; ++c)
                if (c > i) p[c] -= vg * tau * (rowv[c] + scale * tot[c]);  // w_c = tau (P[i][c] + scale S_c)
            p[i < B ? i : 0] = g == i ? beta : vg;
            vprev = vg;
        }
        ++phase;
        __syncthreads();
    }
    // ---- results
    if (live) {
#pragma unroll
        for (int c = 0; c < B; ++c) {
            const bool in_v = c < q.reflectors;
            q.v[(size_t)g * B + c] = !in_v ? 0.0 : (g > c ? p[c] : (g == c ? 1.0 : 0.0));
            // R (upper triangle of the first rows), zeros below; the transposed panel likewise
            const double r = (g <= c || !in_v) ? p[c] : 0.0;
            q.a[(size_t)(q.r0 + g) * q.n + q.j0 + c] = r;
            q.a[(size_t)(q.j0 + c) * q.n + q.r0 + g] = r;
        }
    }
    if (g >= q.m && g < q.m + 16) {  // sixteen zero rows behind the last one: the MFMA kernels read k in blocks of sixteen
#pragma unroll
        for (int c = 0; c < B; ++c) q.v[(size_t)g * B + c] = 0.0;
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int e = threadIdx.x; e < B * B; e += 256) q.t[e] = (e / B < q.reflectors && e % B < q.reflectors) ? tmat[e / B][e % B] : 0.0;
}

// ------------------------------------------------------------------------------------------------ stage 1: X = A22 V
// fp64 MFMA 16x16x4: lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15], and D[(l >> 4) + 4 r][l & 15] in register r.
// No LDS: the sum over k may run in any order as long as A and B agree, so lane (i, kk) takes the four consecutive
// k = 16 s + 4 kk .. + 3 of a block of sixteen - one 32-byte load of its row of A serves four MFMA steps, and a wave
// instruction reads 16 rows x 128 contiguous bytes.  V (m x 32, L2 resident) is read straight into the B fragments.
// A wave makes 16 rows x 32 columns of X over one slice of k (grid.y slices: enough waves for every SIMD); the slices
// are summed by ts_xz.  Rows and k beyond m: V and W keep sixteen zero rows behind row m - 1, and A is read inside its
// allocation (the product with a zero row of V is zero).
__global__ __launch_bounds__(256) void ts_symm(const double* __restrict__ a, int n, int r0, int m, const double* __restrict__ v,
                                               double* __restrict__ xpart, int k_slice) {
    constexpr int B = kTsBand;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= m) return;
    const int k_lo = blockIdx.y * k_slice, k_hi = min(m, k_lo + k_slice);
    const double* arow = a + (size_t)(r0 + min(row0 + i, m - 1)) * n + r0 + 4 * kk;
    const double* vcol = v + (size_t)(4 * kk) * B + i;
    v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
    for (int k = k_lo; k < k_hi; k += 16) {
        const double2 a01 = *reinterpret_cast<const double2*>(arow + k);
        const double2 a23 = *reinterpret_cast<const double2*>(arow + k + 2);
        const double av[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double* vrow = vcol + (size_t)(k + t) * B;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], vrow[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], vrow[16], acc1, 0, 0, 0);
        }
    }
    double* out = xpart + (size_t)blockIdx.y * m * B;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gr = row0 + kk + 4 * r;
        if (gr < m) {
            out[(size_t)gr * B + i] = acc0[r];
            out[(size_t)gr * B + i + 16] = acc1[r];
        }
    }
}

// X = sum of the k slices (written out), and this wave's share of Z = V^T X (32 x 32, k = the wave's 256 rows), again MFMA:
// A[i][k] = V[k][i], B[k][j] = X[k][j].
__global__ __launch_bounds__(64) void ts_xz(const double* __restrict__ xpart, int slices, const double* __restrict__ v, int m,
                                            double* __restrict__ x, double* __restrict__ zpart) {
    constexpr int B = kTsBand;
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const int g0 = blockIdx.x * 256, g1 = min(m, g0 + 256);
    v4f64 acc[2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p >> 1][p & 1] = v4f64{0.0, 0.0, 0.0, 0.0};
    for (int k = g0; k < g1; k += 4) {
        const int row = k + kk;
        double x0 = 0.0, x1 = 0.0, v0 = 0.0, v1 = 0.0;
        if (row < g1) {
            for (int sl = 0; sl < slices; ++sl) {
                x0 += xpart[((size_t)sl * m + row) * B + i];
                x1 += xpart[((size_t)sl * m + row) * B + i + 16];
            }
            x[(size_t)row * B + i] = x0;
            x[(size_t)row * B + i + 16] = x1;
            v0 = v[(size_t)row * B + i];
            v1 = v[(size_t)row * B + i + 16];
        }
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, x0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, x1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, x0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, x1, acc[1][1], 0, 0, 0);
    }
    double* out = zpart + (size_t)blockIdx.x * B * B;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)((p >> 1) * 16 + kk + 4 * r) * B + (p & 1) * 16 + i] = acc[p >> 1][p & 1][r];
}

// M = T^T (sum of the Z parts) T / 2, one workgroup
__global__ __launch_bounds__(256) void ts_small(const double* __restrict__ zpart, int parts, const double* __restrict__ t, double* __restrict__ mout) {
    constexpr int B = kTsBand;
    __shared__ double z[B][B + 1], tt[B][B + 1], u[B][B + 1];
    for (int e = threadIdx.x; e < B * B; e += 256) {
        double s = 0.0;
        for (int w = 0; w < parts; ++w) s += zpart[(size_t)w * B * B + e];
        z[e / B][e % B] = s;
        tt[e / B][e % B] = t[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < B * B; e += 256) {  // U = Z T
        const int i = e / B, j = e % B;
        double s = 0.0;
        for (int k = 0; k < B; ++k) s += z[i][k] * tt[k][j];
        u[i][j] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < B * B; e += 256) {  // M = T^T U / 2
        const int i = e / B, j = e % B;
        double s = 0.0;
        for (int k = 0; k < B; ++k) s += tt[k][i] * u[k][j];
        mout[e] = 0.5 * s;
    }
}

// W = X T - V M, rows in parallel (a thread per row); sixteen zero rows behind the last one (see ts_symm)
__global__ __launch_bounds__(256) void ts_w(const double* __restrict__ x, const double* __restrict__ v, const double* __restrict__ t,
                                            const double* __restrict__ mm, int m, double* __restrict__ w) {
    constexpr int B = kTsBand;
    __shared__ double tt[B][B], ms[B][B];
    for (int e = threadIdx.x; e < B * B; e += 256) {
        tt[e / B][e % B] = t[e];
        ms[e / B][e % B] = mm[e];
    }
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= m + 16) return;
    if (g >= m) {
#pragma unroll
        for (int c = 0; c < B; ++c) w[(size_t)g * B + c] = 0.0;
        return;
    }
    double xr[B], vr[B];
#pragma unroll
    for (int c = 0; c < B; ++c) {
        xr[c] = x[(size_t)g * B + c];
        vr[c] = v[(size_t)g * B + c];
    }
#pragma unroll 4
    for (int j = 0; j < B; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < B; ++k) s += xr[k] * tt[k][j] - vr[k] * ms[k][j];
        w[(size_t)g * B + j] = s;
    }
}

// ------------------------------------------------------------------------------------------------ stage 1: rank-2B update
// A22 <- A22 - [V | W] [W | V]^T, a wave per 32 x 32 tile (2 x 2 MFMA tiles: each fragment feeds two products), K = 2B =
// four blocks of sixteen, operands straight from V and W (L2 resident) in the k order of ts_symm.
__global__ __launch_bounds__(256) void ts_rank2k(double* __restrict__ a, int n, int r0, int m, const double* __restrict__ v,
                                                 const double* __restrict__ w) {
    constexpr int B = kTsBand;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int row0 = blockIdx.y * 64 + (wave >> 1) * 32, col0 = blockIdx.x * 64 + (wave & 1) * 32;
    if (row0 >= m || col0 >= m) return;
    v4f64 acc[2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p >> 1][p & 1] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        // k block: columns 16 (blk & 1) .. of V (blk < 2) or W for the rows, of W or V for the columns
        const double* pr = (blk < 2 ? v : w) + (blk & 1) * 16 + 4 * kk;
        const double* qc = (blk < 2 ? w : v) + (blk & 1) * 16 + 4 * kk;
        double pa[2][4], qb[2][4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const double2* ps = reinterpret_cast<const double2*>(pr + (size_t)(row0 + 16 * hh + i) * B);
            const double2* qs = reinterpret_cast<const double2*>(qc + (size_t)(col0 + 16 * hh + i) * B);
            const double2 p01 = ps[0], p23 = ps[1], q01 = qs[0], q23 = qs[1];
            pa[hh][0] = p01.x, pa[hh][1] = p01.y, pa[hh][2] = p23.x, pa[hh][3] = p23.y;
            qb[hh][0] = q01.x, qb[hh][1] = q01.y, qb[hh][2] = q23.x, qb[hh][3] = q23.y;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                acc[p >> 1][p & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[p >> 1][t], qb[p & 1][t], acc[p >> 1][p & 1], 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = row0 + (p >> 1) * 16 + kk + 4 * r, gc = col0 + (p & 1) * 16 + i;
            if (gr < m && gc < m) a[(size_t)(r0 + gr) * n + r0 + gc] -= acc[p >> 1][p & 1][r];
        }
}

// band storage of the lower triangle: ab[c * kTsBandLd + d] = A[c + d][c], d = 0 .. B (the rest zero: room for the bulge)
__global__ void ts_extract_band(const double* __restrict__ a, int n, double* __restrict__ ab) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)n * kTsBandLd; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / kTsBandLd), d = (int)(e % kTsBandLd);
        ab[e] = (d <= kTsBand && c + d < n) ? a[(size_t)(c + d) * n + c] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------ stage 2: bulge chasing
// One wave per sweep.  Step k of sweep j works on rows r = j + 1 + k B .. r + B - 1:
//   the column to annihilate: column j itself (k = 0), or the first column of the bulge block G = A[r.., r-B..] (k >= 1);
//   H = I - tau v v^T  from the left on the rest of G, two-sided on S = A[r.., r..], from the right on G2 = A[r+B.., r..],
//   which becomes the G of step k + 1 and stays in LDS.
// progress[j] = steps of sweep j finished (kTsSweepDone once it has left the matrix); a step of sweep j + 1 needs
// progress[j] >= its own index + 3.
constexpr unsigned kTsSweepDone = 0x7FFFFFFFu;
constexpr int kTsLdb = kTsBand + 1;  // doubles per LDS row of a block

struct TsChaseArgs {
    double* ab;          // band storage, n columns of kTsBandLd doubles (+ padding columns of zeros behind the matrix)
    int n;
    unsigned* progress;  // [n]
    unsigned* ticket;    // next sweep
    unsigned* gave_up;   // raised by a wave that has polled too long (the host then reports an error)
    unsigned timeout_ticks;
};

__global__ __launch_bounds__(64) void ts_chase(TsChaseArgs q) {
    constexpr int B = kTsBand, LD = kTsBandLd, LB = kTsLdb;
    constexpr int PER = B * B / 64;  // block entries per lane
    __shared__ double g[B * LB], s[B * LB], g2[B * LB];
    __shared__ double vv[B], pw[B], uu[B];
    const int lane = threadIdx.x;
    const int c = lane & 31, half = lane >> 5;
    const int n = q.n;
    const TsBuffer band(q.ab, (size_t)(n + 4 * B) * LD);
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (;;) {
        unsigned jt = 0;
        if (lane == 0) jt = __hip_atomic_fetch_add(q.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int j = (int)__builtin_amdgcn_readfirstlane(jt);
        if (j >= n - 2) return;
        // has the sweep before finished `need` steps (or left the matrix)?  blocking, or one look
        auto ready = [&](unsigned need, bool block) -> int {  // 1 yes, 0 not yet, -1 give up
            if (j == 0) return 1;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                if (__hip_atomic_load(q.progress + j - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return 1;
                if (!block) return 0;
                if (wall_clock64() - t0 > (unsigned long long)q.timeout_ticks ||
                    __hip_atomic_load(q.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) __hip_atomic_store(q.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return -1;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        };
        // entries of S (both triangles from the stored lower one) and of G2 of step k, this lane's share, to registers
        double sv[PER], gv[PER];
        auto fetch = [&](int k) {
            const int r = j + 1 + k * B, h = min(B, n - r), h2 = min(B, n - r - h);
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int idx = lane + 64 * e, i = idx % B, cc = idx / B;  // consecutive lanes walk down a column
                const int lo = min(i, cc), hi = max(i, cc);
                sv[e] = (i < h && cc < h) ? band.load((size_t)(r + lo) * LD + (hi - lo)) : 0.0;
                gv[e] = (i < h2 && cc < h) ? band.load((size_t)(r + cc) * LD + (B + i - cc)) : 0.0;
            }
        };
        if (ready(3, true) < 0) return;
        ts_compiler_fence();
        fetch(0);
        for (int k = 0;; ++k) {
            const int r = j + 1 + k * B;
            const int h = min(B, n - r);       // rows of the reflector (>= 2)
            const int h2 = min(B, n - r - h);  // rows of the block below
            const bool last = h2 <= 1;         // the next step would have a reflector of at most one row: the sweep ends
            double xi = 0.0;  // lanes 0 .. B-1: entry i of the column to annihilate
            if (k == 0 && lane < h) xi = band.load((size_t)j * LD + 1 + lane);
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int idx = lane + 64 * e, i = idx % B, cc = idx / B;
                s[i * LB + cc] = sv[e];
                g2[i * LB + cc] = gv[e];
            }
            wave_sync();
            // the blocks of the next step, if the sweep before is far enough already (else after this step)
            bool fetched = false;
            if (!last && ready((unsigned)(k + 4), false) > 0) {
                ts_compiler_fence();
                fetch(k + 1);
                fetched = true;
            }
            // ---- the reflector
            if (k > 0 && lane < B) xi = g[lane * LB + 0];
            double sq = (lane >= 1 && lane < B) ? xi * xi : 0.0;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
            const double alpha = __shfl(xi, 0);
            double beta = alpha, tau = 0.0, scale = 0.0;
            if (sq > 0.0) {
                beta = -copysign(sqrt(alpha * alpha + sq), alpha);
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            if (lane < B) vv[lane] = lane == 0 ? 1.0 : (lane < h ? xi * scale : 0.0);
            wave_sync();
            // the annihilated column: (beta, 0, ..., 0)
            if (k == 0) {
                if (lane < h) band.store((size_t)j * LD + 1 + lane, lane == 0 ? beta : 0.0);
            } else {
                if (lane < B) g[lane * LB + 0] = lane == 0 ? beta : 0.0;
                // ---- from the left on the other columns of G: column c (its half of the rows)
                if (c >= 1) {
                    double dot = 0.0;
#pragma unroll
                    for (int i = 16 * half; i < 16 * half + 16; ++i) dot += vv[i] * g[i * LB + c];
                    dot += __shfl_xor(dot, 32);
                    dot *= tau;
#pragma unroll
                    for (int i = 16 * half; i < 16 * half + 16; ++i) g[i * LB + c] -= vv[i] * dot;
                }
            }
            // ---- two-sided on S:  p = tau S v,  w = p - (tau p.v / 2) v,  S -= v w^T + w v^T
            {
                double dot = 0.0;
#pragma unroll
                for (int i = 16 * half; i < 16 * half + 16; ++i) dot += s[i * LB + c] * vv[i];  // (S symmetric: column c = row c)
                dot += __shfl_xor(dot, 32);
                const double pc = tau * dot;
                double pv = half == 0 ? pc * vv[c] : 0.0;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) pv += __shfl_xor(pv, off);
                const double wc = pc - 0.5 * tau * pv * vv[c];
                if (half == 0) pw[c] = wc;
                wave_sync();
                const double vc = vv[c];
#pragma unroll
                for (int i = 16 * half; i < 16 * half + 16; ++i) s[i * LB + c] -= vv[i] * wc + pw[i] * vc;
            }
            // ---- from the right on G2:  u = G2 v (row sums: lane = row), G2 -= tau u v^T
            if (h2 > 0) {
                double dot = 0.0;
#pragma unroll
                for (int cc = 16 * half; cc < 16 * half + 16; ++cc) dot += g2[c * LB + cc] * vv[cc];  // (lane c = row c)
                dot += __shfl_xor(dot, 32);
                if (half == 0) uu[c] = tau * dot;
                wave_sync();
                const double vc = vv[c];
#pragma unroll
                for (int i = 16 * half; i < 16 * half + 16; ++i) g2[i * LB + c] -= uu[i] * vc;
            }
            wave_sync();
            // ---- store G (done with for this sweep) and the lower triangle of S; G2 becomes the next G (kept in LDS),
            // or is stored as well when the sweep ends here
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int idx = lane + 64 * e, i = idx % B, cc = idx / B;
                if (k > 0 && i < h) band.store((size_t)(r - B + cc) * LD + (B + i - cc), g[i * LB + cc]);
                if (i < h && cc <= i) band.store((size_t)(r + cc) * LD + (i - cc), s[i * LB + cc]);
                if (last && i < h2 && cc < h) band.store((size_t)(r + cc) * LD + (B + i - cc), g2[i * LB + cc]);
            }
            if (!last) {
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const int idx = lane + 64 * e;
                    g[idx / B * LB + idx % B] = g2[idx / B * LB + idx % B];
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(q.progress + j, last ? kTsSweepDone : (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last) break;
            if (!fetched) {
                if (ready((unsigned)(k + 4), true) < 0) return;
                ts_compiler_fence();
                fetch(k + 1);
            }
        }
    }
}

// d[c] = A[c][c], e[c] = A[c + 1][c]
__global__ void ts_band_to_tridiagonal(const double* __restrict__ ab, int n, double* __restrict__ d, double* __restrict__ e) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        d[c] = ab[(size_t)c * kTsBandLd];
        e[c] = c + 1 < n ? ab[(size_t)c * kTsBandLd + 1] : 0.0;
    }
}

}  // namespace bdg
