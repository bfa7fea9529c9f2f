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
    // Between workgroups of ONE XCD the L2 is the meeting point: stores that stay in it (plain) and loads that skip the
    // L1 (nt: served by the L2) - no write-through to memory, a third of the round trip (guide: visibility table).
    __device__ inline double load_l2(size_t index) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(index * sizeof(double)), 0, 2));
    }
    __device__ inline void store_l2(size_t index, double v) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsrc, (int)(index * sizeof(double)), 0, 0);
    }
};
struct TsWords {  // the same for 32-bit words (progress counters)
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ inline TsWords(const unsigned* base, size_t count)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(base), 0, (int)(count * sizeof(unsigned)), 0x00020000)) {}
    __device__ inline unsigned load_l2(size_t index) const {
        return __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(index * sizeof(unsigned)), 0, 2);
    }
    __device__ inline void store_l2(size_t index, unsigned v) const {
        __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)(index * sizeof(unsigned)), 0, 0);
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
    const unsigned* only_if;  // nullptr, or: run only if this word is non-zero (the Gram route gave the panel up)
    const double* source;     // nullptr (the panel is read from a), or the copy ts_qr_apply made of it before overwriting it
};

__global__ __launch_bounds__(256) void ts_panel_qr(TsPanelArgs q) {
    constexpr int B = kTsBand;
    if (q.only_if && *q.only_if == 0) return;  // (uniform over the grid)
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
    for (int c = 0; c < B; ++c) {
        // (flagged by the recurrence: the panel is untouched; flagged by the verification: it holds R already, the copy is the panel)
        const bool from_copy = q.source && *q.only_if == 2u;
        p[c] = !live ? 0.0 : from_copy ? q.source[(size_t)g * B + c] : q.a[(size_t)(q.r0 + g) * q.n + q.j0 + c];
    }
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
            // R (upper triangle of the first rows), zeros below
            // (only the panel itself: its mirror image above the diagonal is never read again - the trailing block starts
            // below and to the right of it, the band is taken from the lower triangle)
            q.a[(size_t)(q.r0 + g) * q.n + q.j0 + c] = (g <= c || !in_v) ? p[c] : 0.0;
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

// ------------------------------------------------------------------------------------------------ stage 1: panel QR from the Gram matrix
// ts_panel_qr pays a grid barrier per column (~25 us each, 0.9 ms per panel).  But everything a step needs from the rows
// below the panel's top block are inner products of its CURRENT columns, and reflections leave the inner products of
// whole columns unchanged:  sum_{g >= i} P(i)[g][a] P(i)[g][b] = G[a][b] - sum_{k < i} R[k][a] R[k][b]  with G = P^T P of the
// untouched panel and R the rows already final.  So: ts_gram (G of the rows below the top block, MFMA), ts_qr_recur (ONE
// workgroup runs all B steps on the top B x B block and G: reflector scales, the rows w(i), T), ts_qr_apply (every row
// applies the B reflectors to itself) - no barrier at all.  The price is cancellation: the remaining norm of a column
// comes as a difference, with an error of eps G[i][i]; the reflector built from it is orthogonal to eps G[i][i] / (what
// remains) - and the errors of one such column feed the columns after it: a d-wave lattice of 8 x 21 sites whose worst column
// kept 3 % came out with eigenvalues 6e-10 off (tests/fuzz_dense.py, seed 7, case 45; 2e-14 with any floor from 0.1 up).
// ts_qr_recur therefore raises `unsafe` when less than a quarter of a column's squared norm remains, and the panel is then
// factorised by ts_panel_qr, which sums the rows themselves (one panel in twenty to a hundred on lattice matrices).
constexpr double kTsGramFloor = 0.25;

constexpr int kTsGramRows = 512;  // rows of a panel per workgroup of the Gram kernels (four waves of 128)
__global__ __launch_bounds__(256) void ts_gram(const double* __restrict__ a, int n, int j0, int r0, int m, double* __restrict__ gpart,
                                               int first_row) {
    constexpr int B = kTsBand;
    __shared__ double zs[4][B * B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kk = lane >> 4;
    const int g0 = first_row + blockIdx.x * kTsGramRows + wave * 128, g1 = min(m, g0 + 128);  // (the panel: its rows below the top block)
    v4f64 acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) acc[p] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = g0; k < g1; k += 4) {
        const int row = k + kk;
        double p0 = 0.0, p1 = 0.0;
        if (row < g1) {
            const double* src = a + (size_t)(r0 + row) * n + j0;
            p0 = src[i];
            p1 = src[i + 16];
        }
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, p0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, p1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(p1, p1, acc[2], 0, 0, 0);
    }
    double* mine = zs[wave];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kk + 4 * r;
        mine[row * B + i] = acc[0][r];
        mine[row * B + 16 + i] = acc[1][r];
        mine[(16 + i) * B + row] = acc[1][r];
        mine[(16 + row) * B + 16 + i] = acc[2][r];
    }
    __syncthreads();
    double* out = gpart + (size_t)blockIdx.x * B * B;
    for (int e = threadIdx.x; e < B * B; e += 256) out[e] = (zs[0][e] + zs[1][e]) + (zs[2][e] + zs[3][e]);
}

struct TsRecurArgs {
    const double* a;
    int n, j0, r0, m, reflectors;
    const double* gpart;
    int parts;
    double* scale;   // [B]  1 / (alpha - beta) of every reflector
    double* beta;    // [B]
    double* wrows;   // [B][B]  w(i)[c] = tau_i (row i of the current panel + scale_i S_c), c > i
    unsigned* unsafe;  // raised when a column keeps less than `floor` of its squared norm
    double floor;
    double* colnorm2;  // [B] squared norms of the panel's columns (the scale ts_qr_verify measures against)
};

__global__ __launch_bounds__(256) void ts_qr_recur(TsRecurArgs q) {
    constexpr int B = kTsBand, L = B + 1;
    __shared__ double top[B * L];   // the top block: rows final above the diagonal sweep (R), reflector entries below
    __shared__ double full[B * L];  // F = G + top^T top: inner products of the whole columns
    __shared__ double left[B * L];  // ... of what is left of them below the rows already final
    __shared__ double wm[B * L];
    __shared__ double sc[B], be[B];
    __shared__ int bad;
    const int t = threadIdx.x;
    if (t == 0) bad = 0;
    for (int e = t; e < B * B; e += 256) {
        const int r = e / B, c = e % B;
        double sum = 0.0;
        for (int w = 0; w < q.parts; ++w) sum += q.gpart[(size_t)w * B * B + e];
        full[r * L + c] = sum;  // (G of the rows below the top block, for now)
        top[r * L + c] = r < q.m ? q.a[(size_t)(q.r0 + r) * q.n + q.j0 + c] : 0.0;
        wm[r * L + c] = 0.0;
    }
    __syncthreads();
    for (int e = t; e < B * B; e += 256) {
        const int r = e / B, c = e % B;
        double sum = full[r * L + c];
        for (int k = 0; k < B; ++k) sum += top[k * L + r] * top[k * L + c];
        left[r * L + c] = sum;
    }
    __syncthreads();
    for (int e = t; e < B * B; e += 256) full[e / B * L + e % B] = left[e / B * L + e % B];
    __syncthreads();
    for (int i = 0; i < q.reflectors; ++i) {
        const double alpha = top[i * L + i];
        const double rem_i = left[i * L + i];
        if (t == 0 && rem_i < q.floor * full[i * L + i]) bad = 1;
        const double sigma = fmax(rem_i - alpha * alpha, 0.0);
        double beta = alpha, tau = 0.0, scale = 0.0;
        if (sigma > 0.0) {
            beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (t < B) {
            const int c = t;
            // S_c = sum over the rows below row i;  w_c = tau (P[i][c] + scale S_c)
            wm[i * L + c] = c > i ? tau * (top[i * L + c] + scale * (left[i * L + c] - alpha * top[i * L + c])) : 0.0;
        }
        if (t == 0) {
            sc[i] = scale;
            be[i] = beta;
        }
        __syncthreads();
        // the top block: rows i.. take the reflector
        for (int e = t; e < B * B; e += 256) {
            const int g = e / B, c = e % B;
            if (g >= i && c > i) top[g * L + c] -= (g == i ? 1.0 : top[g * L + i] * scale) * wm[i * L + c];
        }
        __syncthreads();
        if (t < B) {
            const int g = t;
            if (g == i) top[g * L + i] = beta;
            else if (g > i) top[g * L + i] *= scale;
        }
        // row i is final: R[i][c] = top[i][c]; the columns' remaining inner products lose it
        for (int e = t; e < B * B; e += 256) {
            const int a = e / B, c = e % B;
            if (a > i && c > i) left[a * L + c] -= top[i * L + a] * top[i * L + c];
        }
        __syncthreads();
    }
    for (int e = t; e < B * B; e += 256) q.wrows[e] = wm[e / B * L + e % B];
    if (t < B) {
        q.scale[t] = t < q.reflectors ? sc[t] : 0.0;
        q.beta[t] = t < q.reflectors ? be[t] : 0.0;
        q.colnorm2[t] = full[t * L + t];
    }
    if (t == 0) *q.unsafe = bad ? 1u : 0u;
}

// ---- T from the stored V, and the verification of the factorisation.
// ts_vgram2: V^T V and V^T P (P = the copy of the panel) in one MFMA pass over the rows.
__global__ __launch_bounds__(256) void ts_vgram2(const double* __restrict__ v, const double* __restrict__ pc, int m, double* __restrict__ gpart) {
    constexpr int B = kTsBand;
    __shared__ double zs[4][2 * B * B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kk = lane >> 4;
    const int g0 = blockIdx.x * kTsGramRows + wave * 128, g1 = min(m, g0 + 128);
    v4f64 vv[3], vp[2][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) vv[p] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < 4; ++p) vp[p >> 1][p & 1] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
    for (int k = g0; k < g1; k += 4) {
        const int row = k + kk;
        double v0 = 0.0, v1 = 0.0, p0 = 0.0, p1 = 0.0;
        if (row < g1) {
            v0 = v[(size_t)row * B + i];
            v1 = v[(size_t)row * B + i + 16];
            p0 = pc[(size_t)row * B + i];
            p1 = pc[(size_t)row * B + i + 16];
        }
        vv[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, vv[0], 0, 0, 0);
        vv[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v1, vv[1], 0, 0, 0);
        vv[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, vv[2], 0, 0, 0);
        vp[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, p0, vp[0][0], 0, 0, 0);
        vp[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, p1, vp[0][1], 0, 0, 0);
        vp[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, p0, vp[1][0], 0, 0, 0);
        vp[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, p1, vp[1][1], 0, 0, 0);
    }
    double* mine = zs[wave];  // [V^T V | V^T P]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kk + 4 * r;
        mine[row * B + i] = vv[0][r];
        mine[row * B + 16 + i] = vv[1][r];
        mine[(16 + i) * B + row] = vv[1][r];
        mine[(16 + row) * B + 16 + i] = vv[2][r];
#pragma unroll
        for (int p = 0; p < 4; ++p) mine[B * B + ((p >> 1) * 16 + row) * B + (p & 1) * 16 + i] = vp[p >> 1][p & 1][r];
    }
    __syncthreads();
    double* out = gpart + (size_t)blockIdx.x * 2 * B * B;
    for (int e = threadIdx.x; e < 2 * B * B; e += 256) out[e] = (zs[0][e] + zs[1][e]) + (zs[2][e] + zs[3][e]);
}

// T of the block reflector from the V that was stored: tau_i = 2 / |v_i|^2 and T[:i, i] = -tau_i T[:i, :i] (V^T V)[:i, i] make
// I - V T V^T the exact product of the reflectors of these very vectors - orthogonal to rounding whatever the accuracy of
// the vectors themselves (what ts_qr_recur derives from downdated inner products is not: T carried its cancellation on).
// Also M = T^T (V^T P): Q^T P = P - V M, what ts_qr_verify holds against the R that was written.
__global__ __launch_bounds__(256) void ts_t_from_v(const double* __restrict__ gpart, int parts, int reflectors, const unsigned* __restrict__ unsafe,
                                                   const double* __restrict__ scale, double* __restrict__ t, double* __restrict__ mout) {
    constexpr int B = kTsBand, L = B + 1;
    if (*unsafe) return;  // (ts_panel_qr writes T for this panel)
    __shared__ double g[B * L], tm[B * L], vp[B * L];
    const int tid = threadIdx.x;
    for (int e = tid; e < B * B; e += 256) {
        double sum = 0.0, sum2 = 0.0;
        for (int w = 0; w < parts; ++w) {
            sum += gpart[(size_t)w * 2 * B * B + e];
            sum2 += gpart[(size_t)w * 2 * B * B + B * B + e];
        }
        g[e / B * L + e % B] = sum;
        vp[e / B * L + e % B] = sum2;
        tm[e / B * L + e % B] = 0.0;
    }
    __syncthreads();
    if (tid < 64) {  // (the recurrence: one wave, column by column)
        for (int col = 0; col < reflectors; ++col) {
            const double norm2 = g[col * L + col];
            // (scale = 0: the column was already zero below the diagonal and no reflector was made; a reflector whose tail has
            // rounded away is still one - |v|^2 = 1, tau = 2: it flips the sign of the pivot, and R holds beta = -alpha)
            const double tau = scale[col] != 0.0 ? 2.0 / norm2 : 0.0;
            if (tid < col) {
                double sum = 0.0;
                for (int k = tid; k < col; ++k) sum += tm[tid * L + k] * g[k * L + col];
                tm[tid * L + col] = -tau * sum;
            } else if (tid == col) {
                tm[tid * L + col] = tau;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    __syncthreads();
    for (int e = tid; e < B * B; e += 256) {
        const int r = e / B, c = e % B;
        t[e] = (r < reflectors && c < reflectors) ? tm[r * L + c] : 0.0;
        double sum = 0.0;  // M[r][c] = sum_k T[k][r] (V^T P)[k][c], k <= r
        for (int k = 0; k <= r; ++k) sum += tm[k * L + r] * vp[k * L + c];
        mout[e] = sum;
    }
}

// Q^T P = P - V M against what was written (R in the first rows, zeros below): a panel whose worst entry is off by more than
// `tolerance` x (the largest column norm of the panel) is flagged 2, and ts_panel_qr factorises it again from the copy.
__global__ __launch_bounds__(256) void ts_qr_verify(const double* __restrict__ a, int n, int j0, int r0, int m, const double* __restrict__ v,
                                                    const double* __restrict__ pc, const double* __restrict__ mm, const double* __restrict__ colnorm2,
                                                    double tolerance, unsigned* __restrict__ unsafe) {
    constexpr int B = kTsBand;
    if (*unsafe == 1u) return;
    __shared__ double ms[B][B];
    __shared__ double limit;
    for (int e = threadIdx.x; e < B * B; e += 256) ms[e / B][e % B] = mm[e];
    if (threadIdx.x == 0) {
        double largest = 0.0;
        for (int c = 0; c < B; ++c) largest = fmax(largest, colnorm2[c]);
        limit = tolerance * sqrt(largest);
    }
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= m) return;
    double vr[B];
#pragma unroll
    for (int c = 0; c < B; ++c) vr[c] = v[(size_t)g * B + c];
    double worst = 0.0;
#pragma unroll 4
    for (int c = 0; c < B; ++c) {
        double value = pc[(size_t)g * B + c];
#pragma unroll
        for (int k = 0; k < B; ++k) value -= vr[k] * ms[k][c];
        worst = fmax(worst, fabs(value - a[(size_t)(r0 + g) * n + j0 + c]));
    }
    if (worst > limit) atomicMax(unsafe, 2u);
}

// every row applies the reflectors to itself (ts_qr_recur found them), writes its row of V, and R / zeros into A
__global__ __launch_bounds__(256) void ts_qr_apply(double* __restrict__ a, int n, int j0, int r0, int m, int reflectors,
                                                   const double* __restrict__ scale, const double* __restrict__ beta,
                                                   const double* __restrict__ wrows, const unsigned* __restrict__ unsafe,
                                                   double* __restrict__ v, double* __restrict__ panel_copy) {
    constexpr int B = kTsBand;
    if (*unsafe) return;  // (ts_panel_qr factorises this panel)
    __shared__ double wm[B][B], sc[B], be[B];
    for (int e = threadIdx.x; e < B * B; e += 256) wm[e / B][e % B] = wrows[e];
    if (threadIdx.x < B) {
        sc[threadIdx.x] = scale[threadIdx.x];
        be[threadIdx.x] = beta[threadIdx.x];
    }
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= m + 16) return;
    if (g >= m) {
#pragma unroll
        for (int c = 0; c < B; ++c) v[(size_t)g * B + c] = 0.0;
        return;
    }
    double p[B];
#pragma unroll
    for (int c = 0; c < B; ++c) {
        p[c] = a[(size_t)(r0 + g) * n + j0 + c];
        panel_copy[(size_t)g * B + c] = p[c];  // (ts_qr_verify checks the factorisation against it; ts_panel_qr starts from it if that fails)
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
        if (i < reflectors && g >= i) {
            const double vg = g == i ? 1.0 : p[i] * sc[i];
#pragma unroll
            for (int c = 0; c < B; ++c)
                if (c > i) p[c] -= vg * wm[i][c];
            p[i] = g == i ? be[i] : vg;
        }
    }
#pragma unroll
    for (int c = 0; c < B; ++c) {
        const bool in_v = c < reflectors;
        v[(size_t)g * B + c] = !in_v ? 0.0 : (g > c ? p[c] : (g == c ? 1.0 : 0.0));
        a[(size_t)(r0 + g) * n + j0 + c] = (g <= c || !in_v) ? p[c] : 0.0;
    }
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

// X = sum of the k slices (written out), and the workgroup's share of Z = V^T X (32 x 32, k = its 256 rows: 64 per
// wave, summed through LDS), again MFMA: A[i][k] = V[k][i], B[k][j] = X[k][j].
__global__ __launch_bounds__(256) void ts_xz(const double* __restrict__ xpart, int slices, const double* __restrict__ v, int m,
                                             double* __restrict__ x, double* __restrict__ zpart) {
    constexpr int B = kTsBand;
    __shared__ double zs[4][B * B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kk = lane >> 4;
    const int g0 = blockIdx.x * 256 + wave * 64, g1 = min(m, g0 + 64);
    v4f64 acc[2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p >> 1][p & 1] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = g0; k < g0 + 64; k += 4) {
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
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) zs[wave][((p >> 1) * 16 + kk + 4 * r) * B + (p & 1) * 16 + i] = acc[p >> 1][p & 1][r];
    __syncthreads();
    double* out = zpart + (size_t)blockIdx.x * B * B;
    for (int e = threadIdx.x; e < B * B; e += 256) out[e] = (zs[0][e] + zs[1][e]) + (zs[2][e] + zs[3][e]);
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
// `first_strip`: 1 = only the first B columns of the block (the next panel and its top block: its factorisation then runs
// beside the rest of the update), 0 = rows and columns from B on (the block the next panel's products read).  The first B
// rows to the right of the diagonal block are never read again and stay as they are.
__global__ __launch_bounds__(256) void ts_rank2k(double* __restrict__ a, int n, int r0, int m, const double* __restrict__ v,
                                                 const double* __restrict__ w, int first_strip) {
    constexpr int B = kTsBand;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int row0 = blockIdx.y * 64 + (wave >> 1) * 32, col0 = first_strip ? 0 : B + blockIdx.x * 64 + (wave & 1) * 32;
    if (row0 >= m || col0 >= m) return;
    if (first_strip ? (wave & 1) != 0 : row0 < B) return;
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
    // All sweeps run on ONE XCD (the first wave to arrive proposes its own, read from the hardware register; waves that
    // find themselves elsewhere leave): band entries and progress words then meet in that XCD's L2 - plain stores and
    // L1-bypassing loads, 1.5 us per hand-over instead of 4.5 through memory.  0xFFFFFFFF = not chosen yet.
    unsigned* xcd;
    unsigned long long* profile;  // nullptr, or [8] ticks (100 MHz) the sweeps spent per phase of a step, summed (measurements)
};

__global__ __launch_bounds__(64) void ts_chase(TsChaseArgs q) {
    constexpr int B = kTsBand, LD = kTsBandLd, LB = kTsLdb;
    constexpr int PER = B * B / 64;  // block entries per lane
    // The three blocks of a step live in registers.  Layout A (how the band is read and written: consecutive lanes walk
    // down a column): lane = (row i = lane & 31, parity hp = lane >> 5), register e = column 2 e + hp - row sums are sums
    // over a lane's registers.  Column sums (the reflector from the left on G) want layout B: lane = (column c = lane & 31,
    // half hp), register t = row 16 hp + t; G changes layout through LDS once per step, S is made symmetric through it.
    __shared__ double buf[B * LB];
    __shared__ double vec[B], wvec[B];
    const int lane = threadIdx.x;
    const int li = lane & 31, hp = lane >> 5;
    const int n = q.n;
    const TsBuffer band(q.ab, (size_t)(n + 4 * B) * LD);
    const TsWords progress(q.progress, (size_t)n);
    {
        const unsigned here = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu;  // HW_REG_XCC_ID
        unsigned chosen = 0;
        if (lane == 0) {
            const unsigned before = atomicCAS(q.xcd, 0xFFFFFFFFu, here);
            chosen = before == 0xFFFFFFFFu ? here : before;
        }
        if (__builtin_amdgcn_readfirstlane(chosen) != here) return;
    }
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // sum over the 32 rows (lanes of one parity), the same value in every lane of that parity afterwards
    auto sum_rows = [](double v) {
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        return v;
    };
    unsigned long long spent[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mark = 0;
    auto lap = [&](int phase) {
        if (q.profile) {
            const unsigned long long now = wall_clock64();
            spent[phase] += now - mark;
            mark = now;
        }
    };
    for (;;) {
        unsigned jt = 0;
        if (lane == 0) jt = __hip_atomic_fetch_add(q.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int j = (int)__builtin_amdgcn_readfirstlane(jt);
        if (j >= n - 2) break;
        // blocks until the sweep before has finished `need` steps (or left the matrix); false = give up
        auto wait_for = [&](unsigned need) -> bool {
            if (j == 0) return true;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                if (progress.load_l2((size_t)j - 1) >= need) return true;
                if (wall_clock64() - t0 > (unsigned long long)q.timeout_ticks ||
                    __hip_atomic_load(q.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) __hip_atomic_store(q.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        };
        // layout A: the lower triangle of S (zeros above) and G2 of step k
        double s_next[PER], g2_next[PER];
        auto fetch = [&](int k) {
            const int r = j + 1 + k * B, h = min(B, n - r), h2 = min(B, n - r - h);
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int cc = 2 * e + hp;
                s_next[e] = (li < h && cc <= li) ? band.load_l2((size_t)(r + cc) * LD + (li - cc)) : 0.0;
                g2_next[e] = (li < h2 && cc < h) ? band.load_l2((size_t)(r + cc) * LD + (B + li - cc)) : 0.0;
            }
        };
        if (!wait_for(3)) return;
        ts_compiler_fence();
        fetch(0);
        double ga[PER];  // G in layout A (from step 1 on: the G2 of the step before)
#pragma unroll
        for (int e = 0; e < PER; ++e) ga[e] = 0.0;
        if (q.profile) mark = wall_clock64();
        for (int k = 0;; ++k) {
            const int r = j + 1 + k * B;
            const int h = min(B, n - r);       // rows of the reflector (>= 2)
            const int h2 = min(B, n - r - h);  // rows of the block below
            const bool last = h2 <= 1;         // the next step would have a reflector of at most one row: the sweep ends
            double sa[PER], g2a[PER];
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                sa[e] = s_next[e];
                g2a[e] = g2_next[e];
            }
            // how far is the sweep before?  (asked now, looked at after the next LDS round trip)
            const unsigned seen = (!last && j > 0) ? progress.load_l2((size_t)j - 1) : kTsSweepDone;
            double xi = 0.0;  // lanes of parity 0: entry li of the column to annihilate
            if (k == 0) {
                if (hp == 0 && li < h) xi = band.load_l2((size_t)j * LD + 1 + li);
            } else if (hp == 0) {
                xi = ga[0];
            }
            // ---- S symmetric: the stored lower triangle goes through LDS, every lane picks up what lies above its diagonal
#pragma unroll
            for (int e = 0; e < PER; ++e) buf[li * LB + 2 * e + hp] = sa[e];
            wave_sync();
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int cc = 2 * e + hp;
                const double mirrored = buf[cc * LB + li];
                if (cc > li) sa[e] = mirrored;
            }
            lap(0);
            bool fetched = false;
            if (!last && seen >= (unsigned)(k + 4)) {
                ts_compiler_fence();
                fetch(k + 1);
                fetched = true;
            }
            lap(1);
            // ---- the reflector (lanes of parity 0 hold the column, then everybody its own row's entry)
            const double sq = sum_rows((hp == 0 && li >= 1) ? xi * xi : 0.0);
            const double sq0 = __shfl(sq, 0), alpha = __shfl(xi, 0);
            double beta = alpha, tau = 0.0, scale = 0.0;
            if (sq0 > 0.0) {
                beta = -copysign(sqrt(alpha * alpha + sq0), alpha);
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            double vi = li == 0 ? 1.0 : (li < h ? xi * scale : 0.0);
            vi = __shfl(vi, li);  // (parity 1 takes its row's entry from parity 0)
            wave_sync();          // (buf has been read)
            if (hp == 0) vec[li] = vi;
            if (k > 0) {
                // G to layout B through LDS, its first column already (beta, 0, ..., 0)
                if (hp == 0) ga[0] = li == 0 ? beta : 0.0;
#pragma unroll
                for (int e = 0; e < PER; ++e) buf[li * LB + 2 * e + hp] = ga[e];
            } else if (hp == 0 && li < h) {
                band.store_l2((size_t)j * LD + 1 + li, li == 0 ? beta : 0.0);
            }
            wave_sync();
            double vcol[PER];  // v at this lane's columns (layout A)
#pragma unroll
            for (int e = 0; e < PER; ++e) vcol[e] = vec[2 * e + hp];
            lap(2);
            if (k > 0) {
                // ---- from the left on G (layout B: lane = column li, rows 16 hp ..): column sums in the lane
                double gb[PER], vrow[PER];
#pragma unroll
                for (int t = 0; t < PER; ++t) {
                    gb[t] = buf[(16 * hp + t) * LB + li];
                    vrow[t] = vec[16 * hp + t];
                }
                double dot = 0.0;
#pragma unroll
                for (int t = 0; t < PER; ++t) dot += vrow[t] * gb[t];
                dot += __shfl_xor(dot, 32);
                dot = li >= 1 ? tau * dot : 0.0;  // (column 0 is the annihilated one)
                const size_t at = (size_t)(r - B + li) * LD + (B + 16 * hp - li);  // rows 16 hp .. of column li: contiguous
#pragma unroll
                for (int t = 0; t < PER; ++t)
                    if (16 * hp + t < h) band.store_l2(at + t, gb[t] - vrow[t] * dot);
            }
            lap(3);
            // ---- two-sided on S:  p = tau S v,  w = p - (tau p.v / 2) v,  S -= v w^T + w v^T
            {
                double dot = 0.0;
#pragma unroll
                for (int e = 0; e < PER; ++e) dot += sa[e] * vcol[e];
                dot += __shfl_xor(dot, 32);
                const double pi = tau * dot;
                const double pv = __shfl(sum_rows(hp == 0 ? pi * vi : 0.0), 0);
                const double wi = pi - 0.5 * tau * pv * vi;
                if (hp == 0) wvec[li] = wi;
                wave_sync();
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const int cc = 2 * e + hp;
                    sa[e] -= vi * wvec[cc] + wi * vcol[e];
                    if (li < h && cc <= li) band.store_l2((size_t)(r + cc) * LD + (li - cc), sa[e]);
                }
            }
            lap(4);
            // ---- from the right on G2:  u = tau G2 v (row sums),  G2 -= u v^T;  it is the G of the next step
            if (h2 > 0) {
                double dot = 0.0;
#pragma unroll
                for (int e = 0; e < PER; ++e) dot += g2a[e] * vcol[e];
                dot += __shfl_xor(dot, 32);
                const double ui = tau * dot;
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    g2a[e] -= ui * vcol[e];
                    if (last && li < h2 && 2 * e + hp < h) band.store_l2((size_t)(r + 2 * e + hp) * LD + (B + li - (2 * e + hp)), g2a[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < PER; ++e) ga[e] = g2a[e];
            lap(5);
            lap(6);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lap(7);
            if (lane == 0) progress.store_l2((size_t)j, last ? kTsSweepDone : (unsigned)(k + 1));
            if (last) break;
            if (!fetched) {
                if (!wait_for((unsigned)(k + 4))) return;
                ts_compiler_fence();
                fetch(k + 1);
            }
        }
    }
    if (q.profile && lane == 0)
        for (int phase = 0; phase < 8; ++phase) atomicAdd(q.profile + phase, spent[phase]);
}

// d[c] = A[c][c], e[c] = A[c + 1][c]
__global__ void ts_band_to_tridiagonal(const double* __restrict__ ab, int n, double* __restrict__ d, double* __restrict__ e) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        d[c] = ab[(size_t)c * kTsBandLd];
        e[c] = c + 1 < n ? ab[(size_t)c * kTsBandLd + 1] : 0.0;
    }
}

// ================================================================================================ eigenvectors
// The eigenvectors of the two-stage route do not go back through the bulge chasing (a million reflectors of 32 rows,
// each over the whole block of eigenvectors): with the eigenvalues known, the eigenvectors of the BAND matrix come from
// inverse iteration on the band itself (kept aside before the chase), and only the block reflectors of stage 1 - GEMM
// shaped - are applied afterwards.
//
// ts_expand_band: row storage of the symmetric band, full[i * kTsRowLd + k] = B(i, i - B + k), k = 0 .. 2B.
constexpr int kTsRowLd = 2 * kTsBand + 2;
__global__ void ts_expand_band(const double* __restrict__ ab, int n, double* __restrict__ full) {
    constexpr int B = kTsBand;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)(n + B + 2) * kTsRowLd; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / kTsRowLd), k = (int)(e % kTsRowLd);
        const int c = i - B + k;
        double v = 0.0;
        if (i < n && k <= 2 * B && c >= 0 && c < n) v = c <= i ? ab[(size_t)c * kTsBandLd + (i - c)] : ab[(size_t)i * kTsBandLd + (c - i)];
        full[e] = v;
    }
}

// ts_band_vectors: one wave per eigenvalue (scratch/r4_band_invit_proto.py is the numpy model).
//   LU of B - lambda with partial pivoting: at step j the candidates for the pivot are the B + 1 rows not yet used among
//   rows 0 .. j + B; all their entries lie in columns j .. j + 2B.  They sit in LDS, one SLOT per row (a row keeps its slot
//   until it is the pivot; the next row of the band then moves in), columns in a circular frame of 2B + 1 positions: lane
//   l works on column j + 1 + l.  Per step: pivot by a wave reduction, 33 multipliers, one rank-1 update of 33 x 64.
//   U's rows (65 entries) and the multipliers per slot go to global scratch.
//   Forward solve: the right-hand side of the row in slot s travels in lane s.  Backward: lane l holds x[j + 1 + l].
//   Three solves from a pseudo-random start, normalised; tiny pivots perturbed (dlagts).
struct TsVectorArgs {
    const double* full;    // ts_expand_band
    int n;
    const double* shift;   // [n_vec]
    int n_vec;
    double norm;           // of the band matrix (for the pivot floor)
    double* scratch;       // per workgroup: U [n][kTsRowLd], multipliers [n][kTsLmLd], y [n], x [n]
    double* z;             // [n][ld]: column k = eigenvector of shift[k]
    int ld;
    int iterations;        // solves from the pseudo-random start (3; dstein goes on until the vector has grown enough)
};
constexpr int kTsLmLd = kTsBand + 3;  // 33 multipliers + the pivot's slot
__host__ __device__ inline size_t ts_vector_scratch(int n) { return (size_t)n * (kTsRowLd + kTsLmLd + 2) + 64; }

// Sum / maximum over the 64 lanes without the LDS crossbar: quad permutes and row mirrors leave every lane of a row of 16
// with its row's result, row_bcast15 / row_bcast31 carry it on; the total is read from lane 63.  (A butterfly of
// ds_bpermute costs ~100 cycles a round in a dependent chain - the backward solve makes one reduction per row.)
template <int CTRL, int ROW_MASK>
__device__ inline double ts_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double ts_wave_total(double v) {
    v += ts_dpp<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += ts_dpp<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += ts_dpp<0x141, 0xF>(v);  // row_half_mirror
    v += ts_dpp<0x140, 0xF>(v);  // row_mirror
    v += ts_dpp<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += ts_dpp<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ inline double ts_wave_max_nonnegative(double v) {  // (v >= 0 or -1 for lanes that do not take part: 0 bits never win)
    v = fmax(v, ts_dpp<0xB1, 0xF>(v));
    v = fmax(v, ts_dpp<0x4E, 0xF>(v));
    v = fmax(v, ts_dpp<0x141, 0xF>(v));
    v = fmax(v, ts_dpp<0x140, 0xF>(v));
    const double r1 = ts_dpp<0x142, 0xA>(v);
    v = fmax(v, r1);
    const double r2 = ts_dpp<0x143, 0xC>(v);
    v = fmax(v, r2);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ inline double ts_readlane(double v, int from) {  // (`from` uniform)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), from), __builtin_amdgcn_readlane(__double2loint(v), from));
}

__global__ __launch_bounds__(64) void ts_band_vectors(TsVectorArgs q) {
    constexpr int B = kTsBand, SLOTS = B + 1;
    const int lane = threadIdx.x, n = q.n;
    double* u_rows = q.scratch + (size_t)blockIdx.x * ts_vector_scratch(n);
    double* lm_rows = u_rows + (size_t)n * kTsRowLd;
    double* y = lm_rows + (size_t)n * kTsLmLd;
    double* x = y + n;
    const double tiny = 2.220446049250313e-16 * q.norm;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto wave_sum = [](double v) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        return v;
    };
    for (int k_eig = blockIdx.x; k_eig < q.n_vec; k_eig += gridDim.x) {
        const double lambda = q.shift[k_eig];
        // The window in registers: wr[s] = the entry of the row in slot s at this lane's column.  The frame of step j
        // is columns j .. j + 2B = 65 columns for 64 lanes: lane L holds the column congruent to L (mod 64) among
        // j + 1 .. j + 64; column j (congruent to column j + 64) is read off first, and the one entry a row can have in
        // column j + 64 before the update - that of the row that moved in last - waits in `pending` until then.
        double wr[SLOTS];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int kk = lane - sl + B;  // column `lane` of row sl, as an index into the row's band
            wr[sl] = (sl < n && kk >= 0 && kk <= 2 * B) ? q.full[(size_t)sl * kTsRowLd + kk] - (lane == sl ? lambda : 0.0) : 0.0;
        }
        int newest = B;                                                           // slot of the row that moved in last
        double pending = B < n ? q.full[(size_t)B * kTsRowLd + 2 * B] : 0.0;       // its entry in column (its row) + B
        int row_in_slot = lane;  // lanes 0 .. B: the row their slot holds
        for (int j = 0; j < n; ++j) {
            const int pj = j & 63, i_new = j + B + 1;
            const int rel = (lane - (j + 1)) & 63;  // this lane's column is j + 1 + rel
            double incoming = 0.0, next_pending = 0.0;
            if (i_new < n) {
                incoming = q.full[(size_t)i_new * kTsRowLd + rel] - (rel == B ? lambda : 0.0);
                next_pending = q.full[(size_t)i_new * kTsRowLd + 2 * B];
            }
            // ---- column j: from lane pj's registers to lane s for slot s
            double v = 0.0;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const double entry = ts_readlane(wr[sl], pj);
                if (lane == sl) v = entry;
                if (lane == pj) wr[sl] = 0.0;  // (the lane's column is j + 64 from now on)
            }
            if (lane == pj) {
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl)
                    if (sl == newest) wr[sl] = pending;
            }
            // ---- pivot: the largest entry of column j among the slots
            const bool candidate = lane < SLOTS && row_in_slot < n;
            const double mine = candidate ? fabs(v) : -1.0;
            const double best = ts_wave_max_nonnegative(mine);
            const unsigned long long at = __ballot(candidate && mine == best);
            const int p = at ? __builtin_ctzll(at) : 0;  // (the first slot with the largest entry; uniform)
            double pv = ts_readlane(v, p);
            if (fabs(pv) < tiny) pv = pv < 0.0 ? -tiny : tiny;
            const double lm = (candidate && lane != p) ? v / pv : 0.0;
            if (lane < SLOTS) lm_rows[(size_t)j * kTsLmLd + lane] = lm;
            else if (lane == SLOTS) lm_rows[(size_t)j * kTsLmLd + SLOTS] = (double)p;
            // ---- row j of U, the update of the other rows, and row j + B + 1 into the pivot's slot
            double u = 0.0;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl)
                if (sl == p) u = wr[sl];
            u_rows[(size_t)j * kTsRowLd + 1 + rel] = u;
            if (lane == 0) u_rows[(size_t)j * kTsRowLd] = pv;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const double lms = ts_readlane(lm, sl);
                wr[sl] = sl == p ? incoming : wr[sl] - lms * u;
            }
            if (lane == p) row_in_slot = i_new;
            newest = p;
            pending = next_pending;
        }
        // ---- inverse iteration
        for (int i = lane; i < n; i += 64) {
            const uint64_t hsh = splitmix64(((uint64_t)(k_eig + 1) << 32) ^ (uint64_t)i);
            x[i] = (double)(hsh >> 11) * (2.0 / 9007199254740992.0) - 1.0;
        }
        wave_sync();
        for (int iteration = 0; iteration < q.iterations; ++iteration) {
            double norm2 = 0.0;
            for (int i = lane; i < n; i += 64) norm2 += x[i] * x[i];
            norm2 = wave_sum(norm2);
            const double scale = norm2 > 0.0 ? 1.0 / sqrt(norm2) : 1.0;
            // forward: lane s carries the right-hand side of the row in slot s.  x and y move in chunks of 64 (lane i of a
            // chunk <-> entry t + i): one load / store per 64 steps, the step's value by readlane.
            double xs = (lane < SLOTS && lane < n) ? x[lane] * scale : 0.0;
            double lm_next = lane <= SLOTS ? lm_rows[lane] : 0.0;
            double x_chunk = 0.0, y_chunk = 0.0;
            int x_base = -1000;
            for (int j = 0; j < n; ++j) {
                const double lm = lm_next;
                if (j + 1 < n && lane <= SLOTS) lm_next = lm_rows[(size_t)(j + 1) * kTsLmLd + lane];
                const int want = j + B + 1;
                if (want - x_base >= 64 || want < x_base) {
                    x_base = want;
                    x_chunk = x_base + lane < n ? x[x_base + lane] * scale : 0.0;
                }
                const double rhs_new = ts_readlane(x_chunk, want - x_base);
                const int p = (int)ts_readlane(lm, SLOTS);
                const double yj = ts_readlane(xs, p);
                xs -= lm * yj;  // (zero for the pivot's slot and for lanes that hold no slot)
                if (lane == p) xs = rhs_new;
                if (lane == (j & 63)) y_chunk = yj;
                if ((j & 63) == 63 || j == n - 1) {
                    const int at = (j & ~63) + lane;
                    if (at <= j) y[at] = y_chunk;
                }
            }
            wave_sync();
            // backward: U x = y, lane l holds x[j + 1 + l]
            double win = 0.0, out_chunk = 0.0;
            double u_next = u_rows[(size_t)(n - 1) * kTsRowLd + 1 + lane], piv_next = u_rows[(size_t)(n - 1) * kTsRowLd];
            for (int j = n - 1; j >= 0; --j) {
                const double u = u_next, piv = piv_next;
                if (j > 0) {
                    u_next = u_rows[(size_t)(j - 1) * kTsRowLd + 1 + lane];
                    piv_next = u_rows[(size_t)(j - 1) * kTsRowLd];
                }
                if ((j & 63) == 63 || j == n - 1) y_chunk = (j & ~63) + lane < n ? y[(j & ~63) + lane] : 0.0;
                const double dot = ts_wave_total(u * win);
                const double xj = (ts_readlane(y_chunk, j & 63) - dot) / piv;
                win = ts_dpp<0x138, 0xF>(win);  // wave_shr:1 - every lane takes its lower neighbour's entry
                if (lane == 0) win = xj;
                if (lane == (j & 63)) out_chunk = xj;
                if ((j & 63) == 0) {
                    const int at = j + lane;
                    if (at < n) x[at] = out_chunk;
                }
            }
            wave_sync();
        }
        double norm2 = 0.0;
        for (int i = lane; i < n; i += 64) norm2 += x[i] * x[i];
        norm2 = wave_sum(norm2);
        const double scale = norm2 > 0.0 ? 1.0 / sqrt(norm2) : 1.0;
        for (int i = lane; i < n; i += 64) q.z[(size_t)i * q.ld + k_eig] = x[i] * scale;
        wave_sync();
    }
}

// ---- Z <- (I - V T V^T) Z on the rows r0 .. of Z (n x ld, eigenvectors as columns), one panel of stage 1 at a time, last
// panel first.  ts_vtz: partial S = V^T Z per slice of 256 rows and tile of 32 columns; ts_zupdate: S = sum of the
// slices, S' = T S, Z -= V S' - both fp64 MFMA, operands straight from memory (k orders as in ts_symm).
__global__ __launch_bounds__(64) void ts_vtz(const double* __restrict__ v, int m, const double* __restrict__ z, int ld, int r0,
                                             int n_cols, double* __restrict__ spart) {
    constexpr int B = kTsBand;
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const int col0 = blockIdx.x * 32;
    const int g0 = blockIdx.y * 256, g1 = min(m, g0 + 256);
    v4f64 acc[2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p >> 1][p & 1] = v4f64{0.0, 0.0, 0.0, 0.0};
    const bool c0 = col0 + i < n_cols, c1 = col0 + 16 + i < n_cols;
#pragma unroll 4
    for (int k = g0; k < g1; k += 4) {
        const int row = k + kk;
        double v0 = 0.0, v1 = 0.0, z0 = 0.0, z1 = 0.0;
        if (row < g1) {
            v0 = v[(size_t)row * B + i];
            v1 = v[(size_t)row * B + i + 16];
            const double* zr = z + (size_t)(r0 + row) * ld + col0;
            if (c0) z0 = zr[i];
            if (c1) z1 = zr[i + 16];
        }
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, z0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, z1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, z0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, z1, acc[1][1], 0, 0, 0);
    }
    // spart[slice][32 rows of S][n_cols padded to tiles]
    double* out = spart + ((size_t)blockIdx.y * B) * (gridDim.x * 32) + col0;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)((p >> 1) * 16 + kk + 4 * r) * (gridDim.x * 32) + (p & 1) * 16 + i] = acc[p >> 1][p & 1][r];
}

// S' = T (sum of the slices of S), one workgroup per tile of 32 columns
__global__ __launch_bounds__(256) void ts_sprime(const double* __restrict__ t, const double* __restrict__ spart, int slices, int padded,
                                                 double* __restrict__ sprime) {
    constexpr int B = kTsBand;
    __shared__ double sm[B][33];
    const int col0 = blockIdx.x * 32;
    for (int e = threadIdx.x; e < B * 32; e += 256) {
        const int r = e / 32, c = e % 32;
        double sum = 0.0;
#pragma unroll 8
        for (int sl = 0; sl < slices; ++sl) sum += spart[((size_t)sl * B + r) * padded + col0 + c];
        sm[r][c] = sum;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < B * 32; e += 256) {
        const int r = e / 32, c = e % 32;
        double sum = 0.0;
        for (int k = r; k < B; ++k) sum += t[r * B + k] * sm[k][c];  // (T upper triangular)
        sprime[(size_t)r * padded + col0 + c] = sum;
    }
}

// Z[rows][tile of 32 columns] -= V[rows][0 .. 31] S', a wave per 16 rows
__global__ __launch_bounds__(256) void ts_zupdate(const double* __restrict__ v, int m, double* __restrict__ z, int ld, int r0, int n_cols,
                                                  const double* __restrict__ sprime, int padded) {
    constexpr int B = kTsBand;
    __shared__ double sp[B][33];
    const int col0 = blockIdx.x * 32;
    for (int e = threadIdx.x; e < B * 32; e += 256) sp[e / 32][e % 32] = sprime[(size_t)(e / 32) * padded + col0 + e % 32];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kk = lane >> 4;
    for (int row0 = (blockIdx.y * 4 + wave) * 16; row0 < m; row0 += gridDim.y * 64) {
        v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double* vr = v + (size_t)min(row0 + i, m - 1) * B;
#pragma unroll
        for (int k = 0; k < B; k += 4) {
            const double av = vr[k + kk];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, sp[k + kk][i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, sp[k + kk][i + 16], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = row0 + kk + 4 * r;
            if (gr < m) {
                double* zr = z + (size_t)(r0 + gr) * ld + col0;
                if (col0 + i < n_cols) zr[i] -= acc0[r];
                if (col0 + i + 16 < n_cols) zr[i + 16] -= acc1[r];
            }
        }
    }
}

}  // namespace bdg
