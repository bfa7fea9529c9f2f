// Device kernels for the BdG Chebyshev path on gfx950 (CDNA4, wave64).
//
// Data layout in HBM
//   blocks   double2[nnzb][4][4]   BSR blocks exactly as scipy stores them (256 B each)
//   indptr   int32[nb+1], indices int32[nnzb]
//   vectors  double2[4][nb][RV]    "planar": component α of every site is one plane,
//                                  and the RV vectors advanced together are adjacent,
//                                  so lanes (site s, vector r) of a wave read one
//                                  contiguous 16*RV-byte run per site and neighbouring
//                                  sites continue it.
//
// K1  cheb_step<RL>   t_next = coef * H t_cur - t_prev, fused with d = <t_cur|t_cur> and
//                     e = Re<t_next|t_cur>.  HBM-bound: 260 B of matrix per block plus
//                     192 B per (site, vector); no MFMA (0.5-2 flop/B).
// K2  reduce_partials fixed-order sum of the per-workgroup dot partials (bit reproducible).
// K3  fill_random / fill_unit / zero   start vectors from a counter-based generator.
// K5  scatter_dense   BSR -> dense column-major for the rocSOLVER path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bdg {

constexpr int kWave = 64;
constexpr int kBlockThreads = 256;
constexpr int kWavesPerBlock = kBlockThreads / kWave;
// One LDS slot is a double2 (16 B).  A staged 4x4 block takes 17 slots: the odd
// stride spreads the same element of different blocks over different 16-byte
// bank groups, so the broadcast reads below are conflict free.
constexpr int kBlockSlots = 17;

// Streaming (read-once) 16-byte load: keeps the matrix stream from evicting the
// vector planes, which are re-read by neighbouring rows, out of L2.
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ inline double2 load_stream(const double2* p) {
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));
    return make_double2(v.x, v.y);
}

// ------------------------------------------------------------------ RNG (K3)
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ inline uint64_t vector_key(uint64_t seed, uint64_t vec_id) {
    return splitmix64(seed ^ splitmix64(vec_id));
}

__host__ __device__ inline double2 start_entry(uint64_t key, uint64_t element, int kind) {
    const uint64_t h = splitmix64(key + element);
    if (kind == 0) return make_double2((h >> 63) ? -1.0 : 1.0, 0.0);
    switch (h >> 62) {
        case 0: return make_double2(1.0, 0.0);
        case 1: return make_double2(0.0, 1.0);
        case 2: return make_double2(-1.0, 0.0);
        default: return make_double2(0.0, -1.0);
    }
}

// vec[α][site][r] = entry(seed, first_id + r, 4*site + α) for r < n_active, else 0.
__global__ void fill_random(double2* __restrict__ vec, int64_t nb, int rv, int n_active,
                            uint64_t seed, uint64_t first_id, int kind) {
    const int64_t total = 4 * nb * rv;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx % rv);
        const int64_t site = (idx / rv) % nb;
        const int alpha = (int)(idx / (rv * nb));
        double2 v = make_double2(0.0, 0.0);
        if (r < n_active) v = start_entry(vector_key(seed, first_id + r), 4 * site + alpha, kind);
        vec[idx] = v;
    }
}

__global__ void fill_zero(double2* __restrict__ vec, int64_t count) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * blockDim.x)
        vec[idx] = make_double2(0.0, 0.0);
}

// vec = 0 everywhere except vec[row%4][row/4][r] = 1 for r < n_active.
__global__ void set_unit(double2* __restrict__ vec, int64_t nb, int rv, int n_active,
                         const int64_t* __restrict__ rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_active) {
        const int64_t row = rows[r];
        vec[((row & 3) * nb + (row >> 2)) * rv + r] = make_double2(1.0, 0.0);
    }
}

// ------------------------------------------------------------------------ K1
struct StepArgs {
    const int* indptr;
    const int* indices;
    const double2* blocks;
    const double2* cur;  // t_n      planar [4][nb][RL]
    double2* prev;       // t_{n-1} in, t_{n+1} out (same site, same thread: in place)
    double* partial;     // [gridDim.x][RL][2]
    double coef;
    int nb;
    int n_tiles;  // workgroup tiles of 4 * (64/RL) block rows
    int max_row_blocks;
};

__device__ inline void cmac(double2& acc, const double2 a, const double2 x) {
    acc.x = fma(a.x, x.x, acc.x);
    acc.x = fma(-a.y, x.y, acc.x);
    acc.y = fma(a.x, x.y, acc.y);
    acc.y = fma(a.y, x.x, acc.y);
}

// RL = lanes per block row = vectors advanced together (1..64, power of two).
// A wave owns 64/RL consecutive block rows; lane (s, r) produces the four
// components of row s for vector r.  The wave stages its rows' blocks into a
// private LDS region with fully coalesced 16-byte loads, then every lane reads
// the block elements it needs as LDS broadcasts (the RL lanes of a row read
// the same address).  No workgroup barrier inside the tile loop: LDS traffic of
// one wave is ordered, and nothing is shared between waves until the final dot
// reduction.
template <int RL>
__global__ __launch_bounds__(kBlockThreads) void cheb_step(StepArgs a) {
    extern __shared__ double2 lds[];
    constexpr int RW = kWave / RL;  // block rows per wave
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int s = lane / RL;
    const int r = lane % RL;
    const int region = RW * a.max_row_blocks * kBlockSlots;
    double2* stage = lds + wave * region;

    // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its
    // L2), so each such group sweeps one contiguous eighth of the tiles.
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int slots = gridDim.x >> 3;
    const int t_lo = (int)(((int64_t)a.n_tiles * xcd) >> 3);
    const int t_hi = (int)(((int64_t)a.n_tiles * (xcd + 1)) >> 3);

    const size_t plane = (size_t)a.nb * RL;
    double dsum = 0.0, esum = 0.0;

    for (int t = t_lo + slot; t < t_hi; t += slots) {
        const int row0 = (t * kWavesPerBlock + wave) * RW;
        if (row0 >= a.nb) continue;
        const int row_end = min(row0 + RW, a.nb);
        const int kb0 = a.indptr[row0];
        const int kb1 = a.indptr[row_end];

        // -- stage this wave's blocks: element e of the run goes to slot (e/16)*17 + e%16
        const int n_el = (kb1 - kb0) * 16;
        const double2* src = a.blocks + (size_t)kb0 * 16;
        for (int e0 = 0; e0 < n_el; e0 += 4 * kWave) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * kWave + lane;
                if (e < n_el) v[u] = load_stream(src + e);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * kWave + lane;
                if (e < n_el) stage[(e >> 4) * kBlockSlots + (e & 15)] = v[u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        const int i = row0 + s;
        if (i < a.nb) {
            const int kbeg = a.indptr[i];
            const int kend = a.indptr[i + 1];
            double2 acc[4];
#pragma unroll
            for (int al = 0; al < 4; ++al) acc[al] = make_double2(0.0, 0.0);

            double2 x[4], xn[4];
            if (kbeg < kend) {
                const size_t j = (size_t)a.indices[kbeg];
#pragma unroll
                for (int be = 0; be < 4; ++be) xn[be] = a.cur[be * plane + j * RL + r];
            }
            for (int k = kbeg; k < kend; ++k) {
#pragma unroll
                for (int be = 0; be < 4; ++be) x[be] = xn[be];
                if (k + 1 < kend) {
                    const size_t j = (size_t)a.indices[k + 1];
#pragma unroll
                    for (int be = 0; be < 4; ++be) xn[be] = a.cur[be * plane + j * RL + r];
                }
                const double2* blk = stage + (k - kb0) * kBlockSlots;
#pragma unroll
                for (int al = 0; al < 4; ++al)
#pragma unroll
                    for (int be = 0; be < 4; ++be) cmac(acc[al], blk[al * 4 + be], x[be]);
            }

            const size_t own = (size_t)i * RL + r;
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                const double2 p = a.prev[al * plane + own];
                const double2 c = a.cur[al * plane + own];
                double2 nx;
                nx.x = fma(a.coef, acc[al].x, -p.x);
                nx.y = fma(a.coef, acc[al].y, -p.y);
                a.prev[al * plane + own] = nx;
                dsum = fma(c.x, c.x, dsum);
                dsum = fma(c.y, c.y, dsum);
                esum = fma(nx.x, c.x, esum);
                esum = fma(nx.y, c.y, esum);
            }
        }
        // the next tile overwrites `stage`; same-wave LDS ops are ordered, the
        // fence only stops the compiler from hoisting the next stores.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }

    // -- dot products: lanes with equal r across the wave, then across the 4 waves
#pragma unroll
    for (int off = kWave / 2; off >= RL; off >>= 1) {
        dsum += __shfl_xor(dsum, off);
        esum += __shfl_xor(esum, off);
    }
    __syncthreads();  // every wave is done with its staging region
    double* red = reinterpret_cast<double*>(lds);
    if (lane < RL) {
        red[(wave * RL + lane) * 2 + 0] = dsum;
        red[(wave * RL + lane) * 2 + 1] = esum;
    }
    __syncthreads();
    if (threadIdx.x < 2 * RL) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) tot += red[w * RL * 2 + threadIdx.x];
        a.partial[(size_t)blockIdx.x * RL * 2 + threadIdx.x] = tot;
    }
}

// ------------------------------------------------------------------------ K2
// out[step][c] = Σ_g partial[step][g][c] in ascending g, c < width (= 2*RL).
__global__ void reduce_partials(const double* __restrict__ partial, double* __restrict__ out,
                                int groups, int width) {
    const int step = blockIdx.x;
    const int c = threadIdx.x;
    if (c >= width) return;
    const double* p = partial + (size_t)step * groups * width + c;
    double tot = 0.0;
    for (int g = 0; g < groups; ++g) tot += p[(size_t)g * width];
    out[(size_t)step * width + c] = tot;
}

// ------------------------------------------------------------------------ K5
// dense (column-major, n = 4 nb) gets every stored block; caller zero-fills first.
__global__ void scatter_dense(const int* __restrict__ indptr, const int* __restrict__ indices,
                              const double2* __restrict__ blocks, double2* __restrict__ dense,
                              int nb) {
    const int64_t n = 4 * (int64_t)nb;
    const int i = blockIdx.x;
    for (int k = indptr[i] + (threadIdx.x >> 4); k < indptr[i + 1]; k += blockDim.x >> 4) {
        const int el = threadIdx.x & 15;
        const int64_t row = 4 * (int64_t)i + (el >> 2);
        const int64_t col = 4 * (int64_t)indices[k] + (el & 3);
        dense[col * n + row] = blocks[(size_t)k * 16 + el];
    }
}

// planar [4][nb][rv] column r  <->  site-major [nb][4]
__global__ void planar_from_sitemajor(const double2* __restrict__ x, double2* __restrict__ planar,
                                      int64_t nb, int rv, int r) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < 4 * nb;
         idx += (int64_t)gridDim.x * blockDim.x)
        planar[((idx & 3) * nb + (idx >> 2)) * rv + r] = x[idx];
}

__global__ void sitemajor_from_planar(const double2* __restrict__ planar, double2* __restrict__ x,
                                      int64_t nb, int rv, int r) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < 4 * nb;
         idx += (int64_t)gridDim.x * blockDim.x)
        x[idx] = planar[((idx & 3) * nb + (idx >> 2)) * rv + r];
}

}  // namespace bdg
