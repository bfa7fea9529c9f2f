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
// K4  halo_pack / halo_unpack   t_n rows exchanged between row slabs (multi-GPU slab mode).
// K5  scatter_dense   BSR -> dense column-major for the dense eigensolvers.
// K6  jacobi_*        one-sided Jacobi eigensolver for small Hermitian matrices (no rocSOLVER load).
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

__device__ inline void store_stream(double2* p, double2 v) {
    v2d w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<v2d*>(p));
}

// Wave-private LDS slot of (lane, component) in the dictionary kernel.  Component-major: the 64
// lanes of one 16-byte access touch consecutive slots (no bank conflicts); lane-major (stride of
// four slots) costs 4-way conflicts (SQ_LDS_BANK_CONFLICT 1.75e7 vs 3.7e6 cycles per launch).
#ifndef BDG_SHARE_LANE_MAJOR
#define BDG_SHARE_LANE_MAJOR 0
#endif
#if BDG_SHARE_LANE_MAJOR
#define SHARE_SLOT(l, c) ((l) * 4 + (c))
#else
#define SHARE_SLOT(l, c) ((c) * kWave + (l))
#endif

// Flat slot of (component alpha, block row `site`, lane payload r) in a vector buffer
// with rv payloads per (site, component).  Planar: one plane per component.
#ifndef BDG_LAYOUT_INTERLEAVED
#define BDG_LAYOUT_INTERLEAVED 0
#endif
__host__ __device__ inline size_t vslot(int alpha, size_t site, int r, size_t nb, int rv) {
#if BDG_LAYOUT_INTERLEAVED
    return (site * 4 + alpha) * rv + r;
#else
    return ((size_t)alpha * nb + site) * rv + r;
#endif
}
// inverse of vslot / rv: (alpha, site) of the idx-th (site, component) pair
__host__ __device__ inline void vpair(int64_t pair, int64_t nb, int& alpha, int64_t& site) {
#if BDG_LAYOUT_INTERLEAVED
    site = pair >> 2;
    alpha = (int)(pair & 3);
#else
    alpha = (int)(pair / nb);
    site = pair % nb;
#endif
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

// Entry `element` = 4 * site + component of the start vector `key`: ONE hash per site, whose top bits serve the four
// components (±1: bit 63 - component; Z4: the two bits below 64 - 2 * component).  Round 4: a hash per element made the
// sweep that generates t_0 in registers the slowest launch of a run (eight 64-bit hashes per lane and plane: 192 us
// against 181 us for a sweep that moves twice the bytes) - a quarter of the hashes.
__host__ __device__ inline uint64_t start_site_hash(uint64_t key, uint64_t site) { return splitmix64(key + site); }
__host__ __device__ inline double2 start_component(uint64_t h, int component, int kind) {
    if (kind == 0) return make_double2(((h >> (63 - component)) & 1) ? -1.0 : 1.0, 0.0);
    switch ((h >> (62 - 2 * component)) & 3) {
        case 0: return make_double2(1.0, 0.0);
        case 1: return make_double2(0.0, 1.0);
        case 2: return make_double2(-1.0, 0.0);
        default: return make_double2(0.0, -1.0);
    }
}
__host__ __device__ inline double2 start_entry(uint64_t key, uint64_t element, int kind) {
    return start_component(start_site_hash(key, element >> 2), (int)(element & 3), kind);
}

// Vector buffers hold `ncols` block rows: the first `nb` are the rows this handle owns
// (global block row = row_offset + local row), the rest is halo filled by the exchange.
// vec[α][site][r] = entry(seed, first_id + r, 4*(row_offset + site) + α) for site < nb,
// r < n_active; everything else 0.
__global__ void fill_random(double2* __restrict__ vec, int64_t nb, int64_t ncols, int rv, int n_active,
                            uint64_t seed, uint64_t first_id, int kind, int64_t row_offset) {
    const int64_t total = 4 * ncols * rv;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx % rv);
        int alpha;
        int64_t site;
        vpair(idx / rv, ncols, alpha, site);
        double2 v = make_double2(0.0, 0.0);
        if (r < n_active && site < nb)
            v = start_entry(vector_key(seed, first_id + r), 4 * (row_offset + site) + alpha, kind);
        vec[idx] = v;
    }
}

// RealMode layout: double[α][site][rv]; only the Rademacher kind is real.  One thread writes one
// 16-byte lane payload (vectors 2q, 2q+1 of one (site, component)).
__global__ void fill_random_real(double* __restrict__ vec, int64_t nb, int64_t ncols, int rv,
                                 int n_active, uint64_t seed, uint64_t first_id, int64_t row_offset) {
    const int rl = rv / 2;  // payloads per (site, component)
    const int64_t total = 4 * ncols * rl;
    double2* out = reinterpret_cast<double2*>(vec);
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % rl);
        int alpha;
        int64_t site;
        vpair(idx / rl, ncols, alpha, site);
        double2 v = make_double2(0.0, 0.0);
        if (site < nb) {
            const uint64_t element = 4 * (row_offset + site) + alpha;
            if (2 * q < n_active) v.x = start_entry(vector_key(seed, first_id + 2 * q), element, 0).x;
            if (2 * q + 1 < n_active) v.y = start_entry(vector_key(seed, first_id + 2 * q + 1), element, 0).x;
        }
        out[idx] = v;
    }
}

__global__ void fill_zero(double2* __restrict__ vec, int64_t count) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * blockDim.x)
        vec[idx] = make_double2(0.0, 0.0);
}

// vec = 0 everywhere except the unit entry of GLOBAL scalar row rows[r] in column r; a
// handle sets it only if it owns that row.
__global__ void set_unit(double2* __restrict__ vec, int64_t nb, int64_t ncols, int rv, int n_active,
                         const int64_t* __restrict__ rows, int64_t row_offset) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_active) {
        const int64_t site = (rows[r] >> 2) - row_offset;
        if (site >= 0 && site < nb)
            vec[vslot((int)(rows[r] & 3), (size_t)site, r, (size_t)ncols, rv)] = make_double2(1.0, 0.0);
    }
}

__global__ void set_unit_real(double* __restrict__ vec, int64_t nb, int64_t ncols, int rv, int n_active,
                              const int64_t* __restrict__ rows, int64_t row_offset) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_active) {
        const int64_t site = (rows[r] >> 2) - row_offset;
        if (site >= 0 && site < nb) vec[vslot((int)(rows[r] & 3), (size_t)site, r, (size_t)ncols, rv)] = 1.0;
    }
}

// ------------------------------------------------------------------------ K4
// Halo exchange of t_n between row slabs.  Lane payloads are 16 B in both arithmetic
// modes, so one pair of kernels serves both.  Message layout: buf[k][α][r].
__global__ void halo_pack(const double2* __restrict__ vec, const int64_t* __restrict__ send_rows,
                          int64_t n_send, int64_t ncols, int rl, double2* __restrict__ buf) {
    const int64_t total = n_send * 4 * rl;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx % rl);
        const int alpha = (int)((idx / rl) & 3);
        const int64_t k = idx / (4 * rl);
        buf[idx] = vec[vslot(alpha, (size_t)send_rows[k], r, (size_t)ncols, rl)];
    }
}

__global__ void halo_unpack(double2* __restrict__ vec, int64_t first_col, int64_t n_recv, int64_t ncols,
                            int rl, const double2* __restrict__ buf) {
    const int64_t total = n_recv * 4 * rl;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx % rl);
        const int alpha = (int)((idx / rl) & 3);
        const int64_t k = idx / (4 * rl);
        vec[vslot(alpha, (size_t)(first_col + k), r, (size_t)ncols, rl)] = buf[idx];
    }
}

// ------------------------------------------------------------------------ K1
struct StepArgs {
    const int* indptr;
    const int* indices;
    const void* blocks;   // ComplexMode: double2[nnzb][16]; RealMode: double[nnzb][16]
    const double2* cur;   // t_n      planar [4][nb][RL] lane payloads
    double2* prev;        // t_{n-1} in, t_{n+1} out (same site, same thread: in place)
    double* partial;      // [gridDim.x][RL * kVec][2]
    const int* tile_order;  // optional permutation of workgroup tiles (nullptr = natural order)
    const int* dict_ids;    // dictionary form: per stored block, column | table index << 24
    const unsigned* dict_ell;  // the same words in rows of 4 (max_row_blocks <= 3) or 8 words, 0xFFFFFFFF = no block (cheb_step_dict)
    const void* dict_table; // dictionary form: the distinct blocks, packed for the mode
    int n_unique;
    double coef;
    // optional per-column scalars (Lanczos): t_next[col] = col_coef[col] * (H t_cur)[col]
    //                                                    - col_pscale[col] * t_prev[col]
    const double* col_coef;
    const double* col_pscale;
    int nb;       // block rows owned (computed)
    int ncols;    // block rows of the vector buffers (owned + halo); == nb without slabs
    int n_tiles;  // workgroup tiles of 4 * (64/RL) block rows
    int max_row_blocks;
    // generic kernel: blocks a wave's LDS staging region holds; a tile with more is staged in chunks
    int stage_blocks;
    // dictionary kernel: 1 = t_{n-1} loads and t_{n+1} stores carry the non-temporal hint (vector
    // buffers larger than the Infinity Cache), 0 = plain (buffers that stay cache resident)
    int stream_vectors;
    // dictionary kernel: 1 = sweep the tiles of each XCD range back to front.  Alternating the
    // direction from one launch to the next re-reads first what the previous launch touched last,
    // i.e. what the Infinity Cache still holds.
    int reverse;
    // first tile of the launch when only a band of block rows is swept (natural order only):
    // unit start vectors spread by at most the matrix bandwidth per step, the rest is still zero
    int tile_base;
    // 1 = t_{n+1} is not stored: the last step of a run, whose vectors nothing reads any more (the
    // call returns the dot products only).  The step is computed and dotted as any other.
    int discard;
};

// Arithmetic modes.  A lane's 16-byte payload is either one complex number of
// one vector (ComplexMode) or the real entries of two adjacent vectors
// (RealMode, used when imag(H) == 0 and the start vectors are real: every t_n
// is then real and half the bytes of both the matrix and the vectors vanish).
// Memory access structure is the same in both: 16 B per lane everywhere.
// Per-lane scalars of the update t_next = c * (H t_cur) - s * t_prev, one pair per payload
// component: a complex payload belongs to one column, a real payload to two.
struct LaneScalars {
    double2 c, s;
};
template <int PER_LANE>
__device__ inline LaneScalars lane_scalars(double coef, const double* col_coef, const double* col_pscale, int r) {
    LaneScalars out;
    out.c = make_double2(coef, coef);
    out.s = make_double2(1.0, 1.0);
    if (col_coef) {
        const int c0 = PER_LANE * r, c1 = PER_LANE * r + (PER_LANE - 1);
        out.c = make_double2(col_coef[c0], col_coef[c1]);
        out.s = make_double2(col_pscale[c0], col_pscale[c1]);
    }
    return out;
}

struct ComplexMode {
    static constexpr int kVec = 1;          // vectors per lane
    static constexpr int kSlotsPerBlock = 16;  // 16-byte staging slots holding one 4x4 block
    static constexpr int kBlockStride = 17;    // padded slot stride in LDS
    __device__ static inline void mac_row(double2 acc[4], const double2* blk, const double2 x[4]) {
#pragma unroll
        for (int al = 0; al < 4; ++al)
#pragma unroll
            for (int be = 0; be < 4; ++be) {
                const double2 m = blk[al * 4 + be];
                acc[al].x = fma(m.x, x[be].x, acc[al].x);
                acc[al].x = fma(-m.y, x[be].y, acc[al].x);
                acc[al].y = fma(m.x, x[be].y, acc[al].y);
                acc[al].y = fma(m.y, x[be].x, acc[al].y);
            }
    }
    // the same for a block that is diagonal as a 4x4 matrix (spin-independent hopping -t σ0 ⊗ τz,
    // the commonest block of a lattice model): 4 complex MACs instead of 16
    __device__ static inline void mac_diag(double2 acc[4], const double2* blk, const double2 x[4]) {
#pragma unroll
        for (int al = 0; al < 4; ++al) {
            const double2 m = blk[al * 5];
            acc[al].x = fma(m.x, x[al].x, acc[al].x);
            acc[al].x = fma(-m.y, x[al].y, acc[al].x);
            acc[al].y = fma(m.x, x[al].y, acc[al].y);
            acc[al].y = fma(m.y, x[al].x, acc[al].y);
        }
    }
    // compact copy of the diagonal for the stencil kernels (no saving here: four complex entries either way)
    static constexpr int kDiagSlots = 4;
    static constexpr int kSingletSlots = 0;  // (no compact form of "singlet" blocks in the unpacked modes)
    __device__ static inline void pack_diag(double2* d, const double2* blk) {
        d[0] = blk[0], d[1] = blk[5], d[2] = blk[10], d[3] = blk[15];
    }
    __device__ static inline void mac_diag_compact(double2 acc[4], const double2* d, const double2 x[4]) {
#pragma unroll
        for (int al = 0; al < 4; ++al) {
            const double2 m = d[al];
            acc[al].x = fma(m.x, x[al].x, acc[al].x);
            acc[al].x = fma(-m.y, x[al].y, acc[al].x);
            acc[al].y = fma(m.x, x[al].y, acc[al].y);
            acc[al].y = fma(m.y, x[al].x, acc[al].y);
        }
    }
    // dot[0] = <c|c>, dot[1] = Re<n|c> for the lane's single vector
    __device__ static inline void dots(double dot[4], const double2 c, const double2 n) {
        dot[0] = fma(c.x, c.x, dot[0]);
        dot[0] = fma(c.y, c.y, dot[0]);
        dot[1] = fma(n.x, c.x, dot[1]);
        dot[1] = fma(n.y, c.y, dot[1]);
    }
};

struct RealMode {
    static constexpr int kVec = 2;
    static constexpr int kSlotsPerBlock = 8;
    static constexpr int kBlockStride = 9;  // 144 B: keeps 16-B alignment, odd in 16-B units
    __device__ static inline void mac_row(double2 acc[4], const double2* blk, const double2 x[4]) {
#pragma unroll
        for (int al = 0; al < 4; ++al)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const double2 m = blk[al * 2 + pr];  // elements (al, 2pr) and (al, 2pr+1)
                acc[al].x = fma(m.x, x[2 * pr].x, acc[al].x);
                acc[al].y = fma(m.x, x[2 * pr].y, acc[al].y);
                acc[al].x = fma(m.y, x[2 * pr + 1].x, acc[al].x);
                acc[al].y = fma(m.y, x[2 * pr + 1].y, acc[al].y);
            }
    }
    __device__ static inline void mac_diag(double2 acc[4], const double2* blk, const double2 x[4]) {
        // elements (al, al): pair al>>1 of row al, component al&1
        rfma_(acc[0], blk[0].x, x[0]);
        rfma_(acc[1], blk[2].y, x[1]);
        rfma_(acc[2], blk[5].x, x[2]);
        rfma_(acc[3], blk[7].y, x[3]);
    }
    __device__ static inline void rfma_(double2& acc, const double m, const double2 x) {
        acc.x = fma(m, x.x, acc.x);
        acc.y = fma(m, x.y, acc.y);
    }
    // compact copy of the diagonal: (m00, m11), (m22, m33) - two 16-byte LDS reads per block instead of four
    static constexpr int kDiagSlots = 2;
    static constexpr int kSingletSlots = 0;
    __device__ static inline void pack_diag(double2* d, const double2* blk) {
        d[0] = make_double2(blk[0].x, blk[2].y);
        d[1] = make_double2(blk[5].x, blk[7].y);
    }
    __device__ static inline void mac_diag_compact(double2 acc[4], const double2* d, const double2 x[4]) {
        const double2 lo = d[0], hi = d[1];
        rfma_(acc[0], lo.x, x[0]);
        rfma_(acc[1], lo.y, x[1]);
        rfma_(acc[2], hi.x, x[2]);
        rfma_(acc[3], hi.y, x[3]);
    }
    // dot[0], dot[1] for the first vector (.x), dot[2], dot[3] for the second (.y)
    __device__ static inline void dots(double dot[4], const double2 c, const double2 n) {
        dot[0] = fma(c.x, c.x, dot[0]);
        dot[1] = fma(n.x, c.x, dot[1]);
        dot[2] = fma(c.y, c.y, dot[2]);
        dot[3] = fma(n.y, c.y, dot[3]);
    }
};

// Particle-hole packed storage.  Every block the assembly produces has the Nambu form
//   [[ A, B ], [ C, -conj(A) ]]   (reference hamiltonian.py:106-118: H_ij -> (+H, -H*),
// Δ_ij -> upper right, Δ_ji^† -> lower left), so the lower-right 2x2 need not be stored:
// 12 of 16 entries, order A00 A01 A10 A11 | B00 B01 B10 B11 | C00 C01 C10 C11.  Checked
// block by block at upload; matrices that violate it use the full modes.
struct ComplexPHMode {
    static constexpr int kVec = 1;
    static constexpr int kSlotsPerBlock = 12;
    static constexpr int kBlockStride = 13;
    __device__ static inline void cfma(double2& acc, const double2 m, const double2 x) {
        acc.x = fma(m.x, x.x, acc.x);
        acc.x = fma(-m.y, x.y, acc.x);
        acc.y = fma(m.x, x.y, acc.y);
        acc.y = fma(m.y, x.x, acc.y);
    }
    // acc -= conj(m) * x
    __device__ static inline void cfms_conj(double2& acc, const double2 m, const double2 x) {
        acc.x = fma(-m.x, x.x, acc.x);
        acc.x = fma(-m.y, x.y, acc.x);
        acc.y = fma(-m.x, x.y, acc.y);
        acc.y = fma(m.y, x.x, acc.y);
    }
    __device__ static inline void mac_row(double2 acc[4], const double2* blk, const double2 x[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double2 a0 = blk[2 * i], a1 = blk[2 * i + 1];
            const double2 b0 = blk[4 + 2 * i], b1 = blk[4 + 2 * i + 1];
            const double2 c0 = blk[8 + 2 * i], c1 = blk[8 + 2 * i + 1];
            cfma(acc[i], a0, x[0]);
            cfma(acc[i], a1, x[1]);
            cfma(acc[i], b0, x[2]);
            cfma(acc[i], b1, x[3]);
            cfma(acc[2 + i], c0, x[0]);
            cfma(acc[2 + i], c1, x[1]);
            cfms_conj(acc[2 + i], a0, x[2]);
            cfms_conj(acc[2 + i], a1, x[3]);
        }
    }
    __device__ static inline void mac_diag(double2 acc[4], const double2* blk, const double2 x[4]) {
        cfma(acc[0], blk[0], x[0]);       // A00
        cfma(acc[1], blk[3], x[1]);       // A11
        cfms_conj(acc[2], blk[0], x[2]);  // -conj(A00)
        cfms_conj(acc[3], blk[3], x[3]);  // -conj(A11)
    }
    // compact copy of the diagonal (A00, A11): the two entries mac_diag reads, side by side
    static constexpr int kDiagSlots = 2;
    __device__ static inline void pack_diag(double2* d, const double2* blk) { d[0] = blk[0], d[1] = blk[3]; }
    // "Singlet" blocks: A diagonal, B and C antidiagonal - on-site terms with spin-singlet pairing (mu, a Zeeman field along
    // z, Delta i sigma_2), hopping with d-wave pairing on the bond.  Half the entries of such a block are zero: the compact
    // copy holds A00, A11, B01, B10, C01, C10 and the product takes 8 complex multiply-adds instead of 16 - mac_row's
    // products on the non-zero entries, in mac_row's order (adding 0 * x changes no bit of a finite sum).
    // Measured in complex arithmetic (profiles/r04_onsite_ab.log): having this path in the kernel costs the complex forms
    // 5-6 % whether or not a block takes it (texture 42.0-42.4 against 44.4 k vector-steps/s, the headline matrix with complex
    // vectors 49.7-50.0 against 53.0-53.5 k), so it is compiled for real arithmetic only: kSingletSlots = 0 switches it off.
    static constexpr int kSingletSlots = 0;  // (6 = the compact copy below)
    __device__ static inline void pack_singlet(double2* d, const double2* blk) {
        d[0] = blk[0], d[1] = blk[3], d[2] = blk[5], d[3] = blk[6], d[4] = blk[9], d[5] = blk[10];
    }
    __device__ static inline void mac_singlet(double2 acc[4], const double2* d, const double2 x[4]) {
        cfma(acc[0], d[0], x[0]);       // A00
        cfma(acc[0], d[2], x[3]);       // B01
        cfma(acc[2], d[4], x[1]);       // C01
        cfms_conj(acc[2], d[0], x[2]);  // -conj(A00)
        cfma(acc[1], d[1], x[1]);       // A11
        cfma(acc[1], d[3], x[2]);       // B10
        cfma(acc[3], d[5], x[0]);       // C10
        cfms_conj(acc[3], d[1], x[3]);  // -conj(A11)
    }
    __device__ static inline void mac_diag_compact(double2 acc[4], const double2* d, const double2 x[4]) {
        cfma(acc[0], d[0], x[0]);
        cfma(acc[1], d[1], x[1]);
        cfms_conj(acc[2], d[0], x[2]);
        cfms_conj(acc[3], d[1], x[3]);
    }
    // Streamed on-site blocks (sweep.hpp, cheb_sweep3 with OS): a diagonal block of H is Hermitian as
    // well as particle-hole symmetric, A = A^†, C = B^†, so 12 doubles describe it:
    //   slot 0 = (Re A00, Re A11), slot 1 = A01, slots 2..5 = B00 B01 B10 B11      (96 B instead of 192)
    // (checked exactly, block by block, at upload).  The products are mac_row's, in mac_row's order,
    // on the same numbers: conjugation and negation are exact, so the results are the same bits.
    static constexpr int kOnsiteSlots = 6;
    static constexpr int kOnsiteStride = 7;
    __device__ static inline double2 cj(const double2 m) { return make_double2(m.x, -m.y); }
    __device__ static inline void mac_onsite(double2 acc[4], const double2* os, const double2 x[4]) {
        const double2 d = os[0], a01 = os[1], b00 = os[2], b01 = os[3], b10 = os[4], b11 = os[5];
        const double2 a00 = make_double2(d.x, 0.0), a11 = make_double2(d.y, 0.0), a10 = cj(a01);
        cfma(acc[0], a00, x[0]);
        cfma(acc[0], a01, x[1]);
        cfma(acc[0], b00, x[2]);
        cfma(acc[0], b01, x[3]);
        cfma(acc[2], cj(b00), x[0]);
        cfma(acc[2], cj(b10), x[1]);
        cfms_conj(acc[2], a00, x[2]);
        cfms_conj(acc[2], a01, x[3]);
        cfma(acc[1], a10, x[0]);
        cfma(acc[1], a11, x[1]);
        cfma(acc[1], b10, x[2]);
        cfma(acc[1], b11, x[3]);
        cfma(acc[3], cj(b01), x[0]);
        cfma(acc[3], cj(b11), x[1]);
        cfms_conj(acc[3], a10, x[2]);
        cfms_conj(acc[3], a11, x[3]);
    }
    // a streamed bond block diag(A00, A11, -conj A00, -conj A11) (sweep.hpp, OS = 2; two record slots A00, A11): the
    // products of mac_diag_compact on the same numbers
    static constexpr int kBondSlots = 2;
    __device__ static inline void mac_bond(double2 acc[4], const double2* bond, const double2 x[4]) {
        mac_diag_compact(acc, bond, x);
    }
    __device__ static inline void dots(double dot[4], const double2 c, const double2 n) {
        ComplexMode::dots(dot, c, n);
    }
};

struct RealPHMode {
    static constexpr int kVec = 2;
    static constexpr int kSlotsPerBlock = 6;
    static constexpr int kBlockStride = 7;
    __device__ static inline void rfma(double2& acc, const double m, const double2 x) {
        acc.x = fma(m, x.x, acc.x);
        acc.y = fma(m, x.y, acc.y);
    }
    __device__ static inline void mac_row(double2 acc[4], const double2* blk, const double2 x[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double2 a = blk[i], b = blk[2 + i], c = blk[4 + i];  // row i of A, B, C
            rfma(acc[i], a.x, x[0]);
            rfma(acc[i], a.y, x[1]);
            rfma(acc[i], b.x, x[2]);
            rfma(acc[i], b.y, x[3]);
            rfma(acc[2 + i], c.x, x[0]);
            rfma(acc[2 + i], c.y, x[1]);
            rfma(acc[2 + i], -a.x, x[2]);
            rfma(acc[2 + i], -a.y, x[3]);
        }
    }
    __device__ static inline void mac_diag(double2 acc[4], const double2* blk, const double2 x[4]) {
        rfma(acc[0], blk[0].x, x[0]);   // A00
        rfma(acc[1], blk[1].y, x[1]);   // A11
        rfma(acc[2], -blk[0].x, x[2]);  // -A00
        rfma(acc[3], -blk[1].y, x[3]);  // -A11
    }
    // the same from a compact copy of the diagonal, d = (A00, A11): one 16-byte LDS read per block instead of two
    static constexpr int kDiagSlots = 1;
    __device__ static inline void pack_diag(double2* d, const double2* blk) { d[0] = make_double2(blk[0].x, blk[1].y); }
    // "singlet" blocks (see ComplexPHMode): (A00, A11), (B01, B10), (C01, C10) - 8 multiply-adds per vector instead of 16
    static constexpr int kSingletSlots = 3;
    __device__ static inline void pack_singlet(double2* d, const double2* blk) {
        d[0] = make_double2(blk[0].x, blk[1].y);
        d[1] = make_double2(blk[2].y, blk[3].x);
        d[2] = make_double2(blk[4].y, blk[5].x);
    }
    __device__ static inline void mac_singlet(double2 acc[4], const double2* d, const double2 x[4]) {
        const double2 a = d[0], b = d[1], c = d[2];
        rfma(acc[0], a.x, x[0]);   // A00
        rfma(acc[0], b.x, x[3]);   // B01
        rfma(acc[2], c.x, x[1]);   // C01
        rfma(acc[2], -a.x, x[2]);  // -A00
        rfma(acc[1], a.y, x[1]);   // A11
        rfma(acc[1], b.y, x[2]);   // B10
        rfma(acc[3], c.y, x[0]);   // C10
        rfma(acc[3], -a.y, x[3]);  // -A11
    }
    __device__ static inline void mac_diag_compact(double2 acc[4], const double2* d, const double2 x[4]) {
        const double2 a = d[0];
        rfma(acc[0], a.x, x[0]);
        rfma(acc[1], a.y, x[1]);
        rfma(acc[2], -a.x, x[2]);
        rfma(acc[3], -a.y, x[3]);
    }
    // Streamed on-site blocks (see ComplexPHMode): A symmetric, C = B^T, 8 doubles
    //   slot 0 = (A00, A01), slot 1 = (A11, 0), slot 2 = (B00, B01), slot 3 = (B10, B11)   (64 B instead of 96)
    static constexpr int kOnsiteSlots = 4;
    static constexpr int kOnsiteStride = 5;
    __device__ static inline void mac_onsite(double2 acc[4], const double2* os, const double2 x[4]) {
        const double2 s0 = os[0], s1 = os[1], s2 = os[2], s3 = os[3];
        rfma(acc[0], s0.x, x[0]);
        rfma(acc[0], s0.y, x[1]);
        rfma(acc[0], s2.x, x[2]);
        rfma(acc[0], s2.y, x[3]);
        rfma(acc[2], s2.x, x[0]);
        rfma(acc[2], s3.x, x[1]);
        rfma(acc[2], -s0.x, x[2]);
        rfma(acc[2], -s0.y, x[3]);
        rfma(acc[1], s0.y, x[0]);
        rfma(acc[1], s1.x, x[1]);
        rfma(acc[1], s3.x, x[2]);
        rfma(acc[1], s3.y, x[3]);
        rfma(acc[3], s2.y, x[0]);
        rfma(acc[3], s3.y, x[1]);
        rfma(acc[3], -s0.y, x[2]);
        rfma(acc[3], -s1.x, x[3]);
    }
    // a streamed bond block diag(a.x, a.y, -a.x, -a.y) (sweep.hpp, OS = 2; one record slot): mac_diag's products in mac_diag's order
    static constexpr int kBondSlots = 1;
    __device__ static inline void mac_bond(double2 acc[4], const double2* bond, const double2 x[4]) {
        const double2 a = bond[0];
        rfma(acc[0], a.x, x[0]);
        rfma(acc[1], a.y, x[1]);
        rfma(acc[2], -a.x, x[2]);
        rfma(acc[3], -a.y, x[3]);
    }
    __device__ static inline void dots(double dot[4], const double2 c, const double2 n) {
        RealMode::dots(dot, c, n);
    }
};

// Wave reduction over the site lanes, then over the 4 waves through `red`,
// one partial per workgroup: partial[block][vector][{d, e}].
template <typename Mode, int RL>
__device__ inline void reduce_dots(double dot[4], double* red, double* partial, int lane, int wave) {
    constexpr int W = 2 * Mode::kVec;  // doubles per lane
#pragma unroll
    for (int off = kWave / 2; off >= RL; off >>= 1)
#pragma unroll
        for (int c = 0; c < W; ++c) dot[c] += __shfl_xor(dot[c], off);
    if (lane < RL)
#pragma unroll
        for (int c = 0; c < W; ++c) red[(wave * RL + lane) * W + c] = dot[c];
    __syncthreads();
    if ((int)threadIdx.x < W * RL) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) tot += red[w * RL * W + threadIdx.x];
        partial[(size_t)blockIdx.x * RL * W + threadIdx.x] = tot;
    }
}

// Generic form.  RL = lanes per block row (4..64); a wave owns 64/RL
// consecutive block rows; lane (s, r) produces the four components of row s
// for its vector(s).  The wave stages its rows' blocks into a private LDS
// region with fully coalesced 16-byte streaming loads, then every lane reads
// the block elements it needs as LDS broadcasts (the RL lanes of a row read the
// same address).  No workgroup barrier inside the tile loop: LDS traffic of one
// wave is ordered, and nothing is shared between waves until the dot reduction.
template <typename Mode, int RL, bool COLS = false>
__global__ __launch_bounds__(kBlockThreads) void cheb_step(StepArgs a) {
    extern __shared__ double2 lds[];
    constexpr int RW = kWave / RL;  // block rows per wave
    constexpr int SPB = Mode::kSlotsPerBlock;
    constexpr int STRIDE = Mode::kBlockStride;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int s = lane / RL;
    const int r = lane % RL;
    const int region = a.stage_blocks * STRIDE;
    double2* stage = lds + wave * region;
    const double2* all_blocks = static_cast<const double2*>(a.blocks);

    // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its
    // L2), so each such group sweeps one contiguous eighth of the tiles.
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int slots = gridDim.x >> 3;
    const int t_lo = (int)(((int64_t)a.n_tiles * xcd) >> 3);
    const int t_hi = (int)(((int64_t)a.n_tiles * (xcd + 1)) >> 3);

    double dot[4] = {0.0, 0.0, 0.0, 0.0};

    for (int t = t_lo + slot; t < t_hi; t += slots) {
        const int tt = a.reverse ? t_lo + t_hi - 1 - t : t;
        const int tile = a.tile_order ? a.tile_order[tt] : tt + a.tile_base;
        const int row0 = (tile * kWavesPerBlock + wave) * RW;
        if (row0 >= a.nb) continue;
        const int row_end = min(row0 + RW, a.nb);
        const int kb0 = a.indptr[row0];
        const int kb1 = a.indptr[row_end];

        const int i = row0 + s;
        const bool valid = i < a.nb;
        int kbeg = 0, kend = 0;
        if (valid) {
            kbeg = a.indptr[i];
            kend = a.indptr[i + 1];
        }
        double2 acc[4];
#pragma unroll
        for (int al = 0; al < 4; ++al) acc[al] = make_double2(0.0, 0.0);

        // The tile's blocks [kb0, kb1) pass through the staging region in chunks of at most
        // a.stage_blocks (one chunk for lattice matrices; long rows of general matrices take more).
        for (int c0 = kb0; c0 < kb1; c0 += a.stage_blocks) {
            const int c1 = min(c0 + a.stage_blocks, kb1);
            // -- stage: slot e of the run goes to (e / SPB) * STRIDE + e % SPB
            const int n_el = (c1 - c0) * SPB;
            const double2* src = all_blocks + (size_t)c0 * SPB;
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
                    if (e < n_el) stage[(e / SPB) * STRIDE + (e % SPB)] = v[u];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

            const int k0 = max(kbeg, c0), k1 = min(kend, c1);  // this lane's blocks in the chunk
            double2 x[4], xn[4];
            if (k0 < k1) {
                const size_t j = (size_t)a.indices[k0];
#pragma unroll
                for (int be = 0; be < 4; ++be) xn[be] = a.cur[vslot(be, j, r, a.ncols, RL)];
            }
            for (int k = k0; k < k1; ++k) {
#pragma unroll
                for (int be = 0; be < 4; ++be) x[be] = xn[be];
                if (k + 1 < k1) {
                    const size_t j = (size_t)a.indices[k + 1];
#pragma unroll
                    for (int be = 0; be < 4; ++be) xn[be] = a.cur[vslot(be, j, r, a.ncols, RL)];
                }
                Mode::mac_row(acc, stage + (k - c0) * STRIDE, x);
            }
            // the next chunk (or tile) overwrites `stage`; same-wave LDS ops are ordered, the
            // fence only stops the compiler from hoisting the next stores.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }

        if (valid) {
            // COLS instantiations (Lanczos) fetch per-column scalars here, once per tile; the
            // Chebyshev instantiations compile to the plain coef * acc - prev
            const LaneScalars ls = COLS ? lane_scalars<Mode::kVec>(a.coef, a.col_coef, a.col_pscale, r)
                                        : lane_scalars<Mode::kVec>(a.coef, nullptr, nullptr, r);
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                const size_t own = vslot(al, (size_t)i, r, a.ncols, RL);
                const double2 p = (a.stream_vectors & 1) ? load_stream(a.prev + own) : a.prev[own];
                const double2 c = a.cur[own];
                double2 nx;
                nx.x = fma(ls.c.x, acc[al].x, -(ls.s.x * p.x));
                nx.y = fma(ls.c.y, acc[al].y, -(ls.s.y * p.y));
                if (a.discard) {
                } else if (a.stream_vectors & 2) store_stream(a.prev + own, nx);
                else a.prev[own] = nx;
                Mode::dots(dot, c, nx);
            }
        }
    }

    __syncthreads();  // every wave is done with its staging region
    reduce_dots<Mode, RL>(dot, reinterpret_cast<double*>(lds), a.partial, lane, wave);
}

// ----------------------------------------------------------- K1, pipelined form
// Same arithmetic as cheb_step, restructured so that a wave keeps one tile of
// matrix data in flight while it computes the previous one:
//
//   loop over the wave's tiles:
//     issue  matrix loads of tile n+1 -> registers (ML x 16 B per lane, streaming)
//     issue  row metadata of tile n+1 (indptr pair, column indices)
//     compute tile n from LDS (t_n gathers, 16 MACs per block, epilogue)
//     write  tile n+1 registers -> LDS   (same wave: LDS ops are ordered, no barrier)
//
// MAXB (compile time) is the largest number of blocks in any block row; it
// fixes the LDS region and the register staging depth.  Matrices with longer
// rows use the generic kernel above.
template <typename Mode, int RL, int MAXB>
__global__ __launch_bounds__(kBlockThreads) void cheb_step_pipelined(StepArgs a) {
    constexpr int RW = kWave / RL;   // block rows per wave
    constexpr int NBLK = RW * MAXB;  // most blocks a wave tile can hold
    constexpr int SPB = Mode::kSlotsPerBlock;
    constexpr int STRIDE = Mode::kBlockStride;
    constexpr int ML = (NBLK * SPB + kWave - 1) / kWave;  // 16-byte loads per lane per tile
    constexpr int W = 2 * Mode::kVec;
    __shared__ double2 lds[kWavesPerBlock * NBLK * STRIDE];
    __shared__ double red[kWavesPerBlock * RL * W];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int s = lane / RL;
    const int r = lane % RL;
    double2* stage = lds + wave * (NBLK * STRIDE);
    const double2* all_blocks = static_cast<const double2*>(a.blocks);

    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int slots = gridDim.x >> 3;
    const int t_lo = (int)(((int64_t)a.n_tiles * xcd) >> 3);
    const int t_hi = (int)(((int64_t)a.n_tiles * (xcd + 1)) >> 3);

    // tile -> first block row of this wave (>= nb means "no work")
    auto first_row = [&](int t) {
        if (t >= t_hi) return a.nb;
        const int tt = a.reverse ? t_lo + t_hi - 1 - t : t;
        const int tile = a.tile_order ? a.tile_order[tt] : tt + a.tile_base;
        return (tile * kWavesPerBlock + wave) * RW;
    };

    struct RowMeta {
        int kbeg, kend;
        int col[MAXB];
    };
    auto load_meta = [&](int row0, RowMeta& m) {
        const int i = row0 + s;
        m.kbeg = m.kend = 0;
        if (i < a.nb) {
            m.kbeg = a.indptr[i];
            m.kend = a.indptr[i + 1];
        }
#pragma unroll
        for (int q = 0; q < MAXB; ++q) m.col[q] = (m.kbeg + q < m.kend) ? a.indices[m.kbeg + q] : 0;
    };
    auto tile_span = [&](int row0, int& kb0, int& n_el) {
        kb0 = 0;
        n_el = 0;
        if (row0 < a.nb) {
            kb0 = a.indptr[row0];
            n_el = (a.indptr[min(row0 + RW, a.nb)] - kb0) * SPB;
        }
    };

    double2 mreg[ML];
    auto issue_matrix = [&](int kb0, int n_el) {
        const double2* src = all_blocks + (size_t)kb0 * SPB;
#pragma unroll
        for (int u = 0; u < ML; ++u) {
            const int e = u * kWave + lane;
            if (e < n_el) mreg[u] = load_stream(src + e);
        }
    };
    auto commit_matrix = [&](int n_el) {
#pragma unroll
        for (int u = 0; u < ML; ++u) {
            const int e = u * kWave + lane;
            if (e < n_el) stage[(e / SPB) * STRIDE + (e % SPB)] = mreg[u];
        }
    };

    double dot[4] = {0.0, 0.0, 0.0, 0.0};

    // ---- prologue: tile 0 into LDS, metadata of tile 0, span of tile 1
    // `pos` walks this workgroup's positions in the (possibly permuted) tile order; a
    // position whose rows fall beyond nb (ragged last tile) simply has no work.
    int pos = t_lo + slot;
    int row0 = first_row(pos);
    int kb0, n_el;
    tile_span(row0, kb0, n_el);
    RowMeta meta;
    load_meta(row0, meta);
    issue_matrix(kb0, n_el);
    int row0_n = first_row(pos + slots);
    int kb0_n, n_el_n;
    tile_span(row0_n, kb0_n, n_el_n);
    commit_matrix(n_el);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    for (; pos < t_hi; pos += slots) {
        // ---- prefetch tile n+1 (registers) and the span of tile n+2
        issue_matrix(kb0_n, n_el_n);
        RowMeta meta_n;
        load_meta(row0_n, meta_n);
        const int row0_nn = first_row(pos + 2 * slots);
        int kb0_nn, n_el_nn;
        tile_span(row0_nn, kb0_nn, n_el_nn);

        // ---- compute tile n
        const int i = row0 + s;
        if (i < a.nb) {
            double2 acc[4], x[4], xn[4];
#pragma unroll
            for (int al = 0; al < 4; ++al) acc[al] = make_double2(0.0, 0.0);
            const int len = meta.kend - meta.kbeg;
            if (len > 0) {
#pragma unroll
                for (int be = 0; be < 4; ++be) xn[be] = a.cur[vslot(be, (size_t)meta.col[0], r, a.ncols, RL)];
            }
            const double2* blk = stage + (meta.kbeg - kb0) * STRIDE;
#pragma unroll
            for (int q = 0; q < MAXB; ++q) {
                if (q < len) {
#pragma unroll
                    for (int be = 0; be < 4; ++be) x[be] = xn[be];
                    if (q + 1 < MAXB && q + 1 < len) {
#pragma unroll
                        for (int be = 0; be < 4; ++be)
                            xn[be] = a.cur[vslot(be, (size_t)meta.col[q + 1 < MAXB ? q + 1 : 0], r, a.ncols, RL)];
                    }
                    Mode::mac_row(acc, blk + q * STRIDE, x);
                }
            }
            const LaneScalars ls = lane_scalars<Mode::kVec>(a.coef, nullptr, nullptr, r);
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                const size_t own = vslot(al, (size_t)i, r, a.ncols, RL);
                const double2 p = (a.stream_vectors & 1) ? load_stream(a.prev + own) : a.prev[own];
                const double2 c = a.cur[own];
                double2 nx;
                nx.x = fma(ls.c.x, acc[al].x, -(ls.s.x * p.x));
                nx.y = fma(ls.c.y, acc[al].y, -(ls.s.y * p.y));
                if (a.discard) {
                } else if (a.stream_vectors & 2) store_stream(a.prev + own, nx);
                else a.prev[own] = nx;
                Mode::dots(dot, c, nx);
            }
        }

        // ---- tile n+1: registers -> LDS (reads of tile n were issued earlier by this wave)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        commit_matrix(n_el_n);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        row0 = row0_n;
        kb0 = kb0_n;
        meta = meta_n;
        row0_n = row0_nn;
        kb0_n = kb0_nn;
        n_el_n = n_el_nn;
    }

    reduce_dots<Mode, RL>(dot, red, a.partial, lane, wave);
}

// ---------------------------------------------------------- K1, dictionary form
// Lattice Hamiltonians repeat a handful of distinct blocks (one per kind of site and of
// bond) millions of times.  When the upload finds few distinct blocks the matrix is kept as
// a table of those blocks plus one 32-bit word per stored block (column in the low 24 bits,
// table index in the high 8); the table is copied into LDS once per workgroup and the
// per-launch matrix stream shrinks from 96..256 B to 4 B per block.  Arithmetic, order of
// operations and results are those of the other forms (same Mode::mac_row on the same numbers).
//
// With the matrix stream gone the launch is bound by the t_n traffic, so this form also trims
// the gathers: every lane loads the t_n entries of its OWN row once (they are needed for the
// dot products anyway) and parks them in a wave-private LDS slot; a block whose column is the
// own row or a row held by a neighbouring lane of the same wave (column = row ± 1 for lattice
// neighbours along the fastest axis) is served from there instead of from L2.  The test is
// made per block on the actual column index, so any sparsity pattern stays correct.
// Row metadata of the next tile is prefetched while the current one computes.
template <typename Mode, int RL, int MAXB, bool COLS = false>
__global__ __launch_bounds__(kBlockThreads, 4) void cheb_step_dict(StepArgs a) {
    extern __shared__ double2 lds[];
    constexpr int RW = kWave / RL;
    constexpr int SPB = Mode::kSlotsPerBlock;
    constexpr int STRIDE = Mode::kBlockStride;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int s = lane / RL;
    const int r = lane % RL;

    // LDS: [table: n_unique x STRIDE slots][per wave: 64 lanes x 4 own entries]
    const double2* table = static_cast<const double2*>(a.dict_table);
    for (int e = threadIdx.x; e < a.n_unique * SPB; e += kBlockThreads)
        lds[(e / SPB) * STRIDE + (e % SPB)] = table[e];
    double2* share = lds + a.n_unique * STRIDE + wave * (kWave * 4);
    __syncthreads();

    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int slots = gridDim.x >> 3;
    const int t_lo = (int)(((int64_t)a.n_tiles * xcd) >> 3);
    const int t_hi = (int)(((int64_t)a.n_tiles * (xcd + 1)) >> 3);
    auto first_row = [&](int t) {
        if (t >= t_hi) return a.nb;
        const int tt = a.reverse ? t_lo + t_hi - 1 - t : t;
        const int tile = a.tile_order ? a.tile_order[tt] : tt + a.tile_base;
        return (tile * kWavesPerBlock + wave) * RW;
    };
    // one word per stored block: column in the low 24 bits, table index in the high 8
    struct RowMeta {
        int len;
        unsigned word[MAXB];
    };
    auto col_of = [](unsigned w) { return (size_t)(w & 0xFFFFFFu); };
    auto id_of = [](unsigned w) { return (int)(w >> 24); };
    // (fixed-width rows: one or two 16-byte loads whose address depends on the row alone - no indptr round trip first;
    // rows past the end read the last row and count as empty)
    constexpr int ELLW = MAXB <= 3 ? 4 : 8;
    auto load_meta = [&](int row0, RowMeta& m) {
        const int i = row0 + s;
        const uint4* src = reinterpret_cast<const uint4*>(a.dict_ell) + (size_t)min(i, a.nb - 1) * (ELLW / 4);
        unsigned words[8];
        const uint4 lo = src[0];
        words[0] = lo.x, words[1] = lo.y, words[2] = lo.z, words[3] = lo.w;
        if constexpr (ELLW == 8) {
            const uint4 hi = src[1];
            words[4] = hi.x, words[5] = hi.y, words[6] = hi.z, words[7] = hi.w;
        } else {
            words[4] = words[5] = words[6] = words[7] = 0xFFFFFFFFu;
        }
        m.len = 0;
#pragma unroll
        for (int q = 0; q < MAXB; ++q) {
            const bool there = i < a.nb && words[q] != 0xFFFFFFFFu;
            m.len += there ? 1 : 0;
            m.word[q] = there ? words[q] : 0u;
        }
    };

    double dot[4] = {0.0, 0.0, 0.0, 0.0};
    int pos = t_lo + slot;
    int row0 = first_row(pos);
    RowMeta meta;
    load_meta(row0, meta);
    for (; pos < t_hi; pos += slots) {
        const int row0_n = first_row(pos + slots);
        RowMeta meta_n;
        load_meta(row0_n, meta_n);

        const int i = row0 + s;
        const bool valid = i < a.nb;
        // own entries of t_n go to LDS: read back by this lane (diagonal block, dot products) and
        // by neighbouring lanes; not kept in registers (the kernel sits at the 128-VGPR step)
        {
            double2 own[4];
#pragma unroll
            for (int be = 0; be < 4; ++be)
                own[be] = valid ? a.cur[vslot(be, (size_t)i, r, a.ncols, RL)] : make_double2(0.0, 0.0);
#pragma unroll
            for (int be = 0; be < 4; ++be) share[SHARE_SLOT(lane, be)] = own[be];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        if (valid) {
            // rows held by this wave: i - s .. i - s + RW - 1 (those below nb); offset of the
            // lane that owns column c, or kWave if the column has to come from memory
            auto source = [&](unsigned w) {
                const long d = (long)col_of(w) - (long)i;
                const long ss = (long)s + d;
                return (ss >= 0 && ss < RW && (long)i + d < a.nb) ? (int)d : (int)kWave;
            };
            double2 acc[4], x[4], xn[4];
#pragma unroll
            for (int al = 0; al < 4; ++al) acc[al] = make_double2(0.0, 0.0);
            if (meta.len > 0 && source(meta.word[0]) == kWave) {
#pragma unroll
                for (int be = 0; be < 4; ++be)
                    xn[be] = a.cur[vslot(be, col_of(meta.word[0]), r, a.ncols, RL)];
            }
#pragma unroll
            for (int q = 0; q < MAXB; ++q) {
                if (q < meta.len) {
                    const int src = source(meta.word[q]);
                    if (src == kWave) {
#pragma unroll
                        for (int be = 0; be < 4; ++be) x[be] = xn[be];
                    } else {
#pragma unroll
                        for (int be = 0; be < 4; ++be) x[be] = share[SHARE_SLOT(lane + src * RL, be)];
                    }
                    if (q + 1 < MAXB && q + 1 < meta.len && source(meta.word[q + 1 < MAXB ? q + 1 : 0]) == kWave) {
#pragma unroll
                        for (int be = 0; be < 4; ++be)
                            xn[be] = a.cur[vslot(be, col_of(meta.word[q + 1 < MAXB ? q + 1 : 0]), r, a.ncols, RL)];
                    }
                    Mode::mac_row(acc, lds + id_of(meta.word[q]) * STRIDE, x);
                }
            }
            // COLS instantiations (Lanczos) fetch per-column scalars here, once per tile; the
            // Chebyshev instantiations compile to the plain coef * acc - prev
            const LaneScalars ls = COLS ? lane_scalars<Mode::kVec>(a.coef, a.col_coef, a.col_pscale, r)
                                        : lane_scalars<Mode::kVec>(a.coef, nullptr, nullptr, r);
            double2 p[4];
            if (a.stream_vectors & 1) {
#pragma unroll
                for (int al = 0; al < 4; ++al) p[al] = load_stream(a.prev + vslot(al, (size_t)i, r, a.ncols, RL));
            } else {
#pragma unroll
                for (int al = 0; al < 4; ++al) p[al] = a.prev[vslot(al, (size_t)i, r, a.ncols, RL)];
            }
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                double2 nx;
                nx.x = fma(ls.c.x, acc[al].x, -(ls.s.x * p[al].x));
                nx.y = fma(ls.c.y, acc[al].y, -(ls.s.y * p[al].y));
                p[al] = nx;
                Mode::dots(dot, share[SHARE_SLOT(lane, al)], nx);
            }
            if (a.discard) {
            } else if (a.stream_vectors & 2) {
#pragma unroll
                for (int al = 0; al < 4; ++al) store_stream(a.prev + vslot(al, (size_t)i, r, a.ncols, RL), p[al]);
            } else {
#pragma unroll
                for (int al = 0; al < 4; ++al) a.prev[vslot(al, (size_t)i, r, a.ncols, RL)] = p[al];
            }
        }
        // the next tile overwrites `share`; same-wave LDS ops are ordered
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        row0 = row0_n;
        meta = meta_n;
    }

    __syncthreads();  // table / share reads finished: the reduction reuses the front of the LDS
    reduce_dots<Mode, RL>(dot, reinterpret_cast<double*>(lds), a.partial, lane, wave);
}

// Device-side re-packing of the uploaded complex 4x4 blocks for the other storage modes.
// `entries` = 16 (full) or 12 (particle-hole packed); real_out writes doubles, else double2.
__device__ inline int packed_source(int entries, int e) {
    if (entries == 16) return e;
    const int group = e >> 2, within = e & 3;          // A, B, C groups of four
    const int row = (within >> 1) + (group == 2 ? 2 : 0);
    const int col = (within & 1) + (group == 1 ? 2 : 0);
    return row * 4 + col;
}

__global__ void pack_blocks(const double2* __restrict__ blocks, void* __restrict__ out, int64_t nnzb,
                            int entries, int real_out) {
    const int64_t total = nnzb * entries;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = idx / entries;
        const double2 v = blocks[k * 16 + packed_source(entries, (int)(idx % entries))];
        if (real_out) static_cast<double*>(out)[idx] = v.x;
        else static_cast<double2*>(out)[idx] = v;
    }
}

// Packed on-site (diagonal) blocks for the sweep kernel that streams them (Mode::mac_onsite):
// one record per block row, zeros where the row stores no diagonal block.  Square matrices whose
// diagonal blocks were verified Hermitian + particle-hole symmetric at upload.
__global__ void pack_onsite(const int* __restrict__ indptr, const int* __restrict__ indices,
                            const double2* __restrict__ blocks, int nb, int real_out, double2* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const double2 zero = make_double2(0.0, 0.0);
        double2 a00 = zero, a01 = zero, a11 = zero, b00 = zero, b01 = zero, b10 = zero, b11 = zero;
        for (int k = indptr[i]; k < indptr[i + 1]; ++k)
            if (indices[k] == i) {
                const double2* blk = blocks + (size_t)k * 16;
                a00 = blk[0], a01 = blk[1], a11 = blk[5];
                b00 = blk[2], b01 = blk[3], b10 = blk[6], b11 = blk[7];
            }
        if (real_out) {
            double2* rec = out + (size_t)i * 4;
            rec[0] = make_double2(a00.x, a01.x);
            rec[1] = make_double2(a11.x, 0.0);
            rec[2] = make_double2(b00.x, b01.x);
            rec[3] = make_double2(b10.x, b11.x);
        } else {
            double2* rec = out + (size_t)i * 6;
            rec[0] = make_double2(a00.x, a11.x);
            rec[1] = a01;
            rec[2] = b00, rec[3] = b01, rec[4] = b10, rec[5] = b11;
        }
    }
}

// -------------------------------------------------------------- Lanczos helpers
// w = rbuf - g[col] * v, written over rbuf; per-column |w|^2 over the owned rows.
// PER_LANE = 1: complex payloads (column = r); 2: real payloads (columns 2r, 2r+1).
// partial[block][column]; fixed summation order (bit reproducible).
template <int PER_LANE>
__global__ __launch_bounds__(256) void lanczos_combine(double2* __restrict__ rbuf,
                                                       const double2* __restrict__ v,
                                                       const double* __restrict__ g, int64_t nb,
                                                       int64_t ncols, int rl, double* __restrict__ partial) {
    __shared__ double scratch[2][256];
    const int r = threadIdx.x % rl;  // 256 and the grid stride are multiples of rl: r is fixed per thread
    const double g0 = g[PER_LANE * r], g1 = g[PER_LANE * r + (PER_LANE - 1)];
    double n0 = 0.0, n1 = 0.0;
    const int64_t total = 4 * ncols * rl;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int alpha;
        int64_t site;
        vpair(idx / rl, ncols, alpha, site);
        if (site >= nb) continue;  // halo rows are refreshed by the exchange, never combined
        const double2 a = rbuf[idx], b = v[idx];
        double2 w;
        w.x = fma(-g0, b.x, a.x);
        w.y = fma(-g1, b.y, a.y);
        rbuf[idx] = w;
        n0 = fma(w.x, w.x, n0);
        n1 = fma(w.y, w.y, n1);
    }
    scratch[0][threadIdx.x] = n0;
    scratch[1][threadIdx.x] = n1;
    __syncthreads();
    if ((int)threadIdx.x < rl) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = threadIdx.x; k < 256; k += rl) {
            t0 += scratch[0][k];
            t1 += scratch[1][k];
        }
        double* out = partial + (size_t)blockIdx.x * rl * PER_LANE;
        if (PER_LANE == 1) out[r] = t0 + t1;
        else {
            out[2 * r] = t0;
            out[2 * r + 1] = t1;
        }
    }
}

// Device-side scalar bookkeeping of the Lanczos process on H^2 (one thread per column).
//   phase 0 (after the norms of W_{j+1} are reduced into `sums`):
//       beta_next = sqrt(sums); coef_a = 1 / beta_next; pscale_b = beta_next / beta_cur
//   phase 1 (after step b's dots, alpha_j = |U|^2, are reduced into `sums` with stride 2):
//       alpha = sums[2 col]; g = alpha / beta_cur
struct LanczosScalars {
    double* beta_hist;   // [iter][cols]
    double* alpha_hist;  // [iter][cols]
    double* coef_a;      // 1 / beta_j              (step a: U = H W_j / beta_j)
    double* pscale_a;    // zeros
    double* coef_b;      // ones
    double* pscale_b;    // beta_j / beta_{j-1}     (step b)
    double* g;           // alpha_j / beta_j        (combine)
};

__global__ void lanczos_scalars(LanczosScalars z, const double* __restrict__ sums, int cols, int iter,
                                int phase) {
    const int c = threadIdx.x;
    if (c >= cols) return;
    if (phase == 0) {  // norms of the new vector W_iter are in sums[c]
        const double beta = sqrt(sums[c]);
        const double prev = iter > 0 ? z.beta_hist[(size_t)(iter - 1) * cols + c] : 0.0;
        z.beta_hist[(size_t)iter * cols + c] = beta;
        z.coef_a[c] = beta > 0.0 ? 1.0 / beta : 0.0;
        z.pscale_a[c] = 0.0;
        z.coef_b[c] = 1.0;
        z.pscale_b[c] = prev > 0.0 ? beta / prev : 0.0;
    } else {  // alpha_iter = |H v_iter|^2 is the d dot of step b
        const double alpha = sums[2 * c];
        const double beta = z.beta_hist[(size_t)iter * cols + c];
        z.alpha_hist[(size_t)iter * cols + c] = alpha;
        z.g[c] = beta > 0.0 ? alpha / beta : 0.0;
    }
}

// Ritz-vector accumulation (second pass of the Lanczos process): for every level l < n_levels
//   y_l += (coef[l][col] / beta[col]) * w      (w = beta_j v_j, the unnormalised Lanczos vector)
// over the whole buffers; y holds n_levels buffers of `count` payloads back to back.
template <int PER_LANE>
__global__ __launch_bounds__(256) void lanczos_accumulate(const double2* __restrict__ w,
                                                          const double* __restrict__ beta,
                                                          const double* __restrict__ coef, int n_levels,
                                                          int cols, int rl, int64_t count,
                                                          double2* __restrict__ y) {
    const int r = threadIdx.x % rl;  // fixed per thread: 256 and the grid stride are multiples of rl
    const int c0 = PER_LANE * r, c1 = PER_LANE * r + (PER_LANE - 1);
    const double b0 = beta[c0], b1 = beta[c1];
    const double i0 = b0 > 0.0 ? 1.0 / b0 : 0.0, i1 = b1 > 0.0 ? 1.0 / b1 : 0.0;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < count;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = w[idx];
        for (int l = 0; l < n_levels; ++l) {
            const double f0 = coef[(size_t)l * cols + c0] * i0, f1 = coef[(size_t)l * cols + c1] * i1;
            if (f0 == 0.0 && f1 == 0.0) continue;
            double2 acc = y[(size_t)l * count + idx];
            acc.x = fma(f0, v.x, acc.x);
            acc.y = fma(f1, v.y, acc.y);
            y[(size_t)l * count + idx] = acc;
        }
    }
}

// planar [4][nb][rl] real payloads (vectors 2r, 2r+1) -> site-major complex [nb][4] of vector `col`
__global__ void sitemajor_from_planar_real(const double2* __restrict__ planar, double2* __restrict__ x,
                                           int64_t nb, int rl, int col) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < 4 * nb;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = planar[vslot((int)(idx & 3), (size_t)(idx >> 2), col >> 1, (size_t)nb, rl)];
        x[idx] = make_double2((col & 1) ? v.y : v.x, 0.0);
    }
}

// ---- small dense algebra on blocks of vectors (Rayleigh-Ritz of the partial spectrum on the device)
// A block is a planar buffer of `pairs` (component, site) rows x rl payloads: `cols` = rl x PER_LANE columns
// (PER_LANE = 1: one complex column per payload; 2: two real columns per payload, in .x and .y).
constexpr int kBlockAlgebraMaxCols = 16;

// Gram matrix G[i][j] = sum over rows of conj(A[., i]) * B[., j].  blockIdx.y = i; every thread keeps
// row i of G for its rows, then wave shuffles + LDS give one partial per workgroup:
// partial[i][blockIdx.x][j] (complex, interleaved), summed afterwards by reduce_partials (one "step" per row i).
template <int PER_LANE>
__global__ __launch_bounds__(256) void block_gram(const double2* __restrict__ a, const double2* __restrict__ b,
                                                  int64_t pairs, int rl, double* __restrict__ partial) {
    __shared__ double red[4][2 * kBlockAlgebraMaxCols];
    const int cols = rl * PER_LANE, i = blockIdx.y;
    double2 acc[kBlockAlgebraMaxCols];
#pragma unroll
    for (int j = 0; j < kBlockAlgebraMaxCols; ++j) acc[j] = make_double2(0.0, 0.0);
    for (int64_t pair = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; pair < pairs; pair += (int64_t)gridDim.x * blockDim.x) {
        const double2 ai = a[pair * rl + i / PER_LANE];
#pragma unroll
        for (int j = 0; j < kBlockAlgebraMaxCols; ++j)
            if (j < cols) {
                const double2 bj = b[pair * rl + j / PER_LANE];
                if (PER_LANE == 1) {  // conj(ai) * bj
                    acc[j].x = fma(ai.x, bj.x, fma(ai.y, bj.y, acc[j].x));
                    acc[j].y = fma(ai.x, bj.y, fma(-ai.y, bj.x, acc[j].y));
                } else {
                    acc[j].x = fma((i & 1) ? ai.y : ai.x, (j & 1) ? bj.y : bj.x, acc[j].x);
                }
            }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int j = 0; j < kBlockAlgebraMaxCols; ++j)
        if (j < cols) {
#pragma unroll
            for (int off = kWave / 2; off >= 1; off >>= 1) {
                acc[j].x += __shfl_xor(acc[j].x, off);
                acc[j].y += __shfl_xor(acc[j].y, off);
            }
            if (lane == 0) {
                red[wave][2 * j] = acc[j].x;
                red[wave][2 * j + 1] = acc[j].y;
            }
        }
    __syncthreads();
    if ((int)threadIdx.x < 2 * cols)
        partial[((size_t)i * gridDim.x + blockIdx.x) * 2 * cols + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[., m] = sum_i A[., i] * M[i][m]  (M: cols x cols complex, row-major, interleaved; the real form uses the
// real parts).  One thread per row: it reads all of the row before it writes, so out may be A itself.
template <int PER_LANE>
__global__ __launch_bounds__(256) void block_mix(const double2* a, const double* __restrict__ m, int64_t pairs,
                                                 int rl, double2* out) {
    const int cols = rl * PER_LANE;
    for (int64_t pair = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; pair < pairs; pair += (int64_t)gridDim.x * blockDim.x) {
        double2 row[kBlockAlgebraMaxCols / PER_LANE];
#pragma unroll
        for (int r = 0; r < kBlockAlgebraMaxCols / PER_LANE; ++r)
            if (r < rl) row[r] = a[pair * rl + r];
#pragma unroll
        for (int r = 0; r < kBlockAlgebraMaxCols / PER_LANE; ++r) {
            if (r >= rl) continue;
            double2 v = make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < kBlockAlgebraMaxCols / PER_LANE; ++q) {
                if (q >= rl) continue;
                if (PER_LANE == 1) {
                    const double mr = m[2 * ((size_t)q * cols + r)], mi = m[2 * ((size_t)q * cols + r) + 1];
                    v.x = fma(row[q].x, mr, fma(-row[q].y, mi, v.x));
                    v.y = fma(row[q].x, mi, fma(row[q].y, mr, v.y));
                } else {  // columns 2q, 2q+1 in, 2r, 2r+1 out
                    v.x = fma(row[q].x, m[2 * ((size_t)(2 * q) * cols + 2 * r)], fma(row[q].y, m[2 * ((size_t)(2 * q + 1) * cols + 2 * r)], v.x));
                    v.y = fma(row[q].x, m[2 * ((size_t)(2 * q) * cols + 2 * r + 1)], fma(row[q].y, m[2 * ((size_t)(2 * q + 1) * cols + 2 * r + 1)], v.y));
                }
            }
            out[pair * rl + r] = v;
        }
    }
}

// per-column |v|^2 over the owned rows (start of the process): partial[block][column]
template <int PER_LANE>
__global__ __launch_bounds__(256) void column_norms(const double2* __restrict__ v, int64_t nb, int64_t ncols,
                                                    int rl, double* __restrict__ partial) {
    __shared__ double scratch[2][256];
    const int r = threadIdx.x % rl;
    double n0 = 0.0, n1 = 0.0;
    const int64_t total = 4 * ncols * rl;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int alpha;
        int64_t site;
        vpair(idx / rl, ncols, alpha, site);
        if (site >= nb) continue;
        const double2 a = v[idx];
        n0 = fma(a.x, a.x, n0);
        n1 = fma(a.y, a.y, n1);
    }
    scratch[0][threadIdx.x] = n0;
    scratch[1][threadIdx.x] = n1;
    __syncthreads();
    if ((int)threadIdx.x < rl) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = threadIdx.x; k < 256; k += rl) {
            t0 += scratch[0][k];
            t1 += scratch[1][k];
        }
        double* out = partial + (size_t)blockIdx.x * rl * PER_LANE;
        if (PER_LANE == 1) out[r] = t0 + t1;
        else {
            out[2 * r] = t0;
            out[2 * r + 1] = t1;
        }
    }
}

// ------------------------------------------------------------------------ K2
// out[step][c] = Σ_g partial[step][g][c], c < width (<= 128).  One block of 256 threads per
// step: thread t sums groups t/width, t/width + 256/width, ... for column t%width, then the
// 256/width partial sums of a column are added in ascending order.  The summation tree is
// fixed by (groups, width) alone, so results are bit-reproducible run to run.
__global__ __launch_bounds__(256) void reduce_partials(const double* __restrict__ partial,
                                                       double* __restrict__ out, int groups, int width) {
    __shared__ double scratch[256];
    const int step = blockIdx.x;
    const int c = threadIdx.x % width;
    const int lane_group = threadIdx.x / width;
    const int n_lane_groups = 256 / width;
    double tot = 0.0;
    if (lane_group < n_lane_groups) {
        const double* p = partial + (size_t)step * groups * width + c;
        for (int g = lane_group; g < groups; g += n_lane_groups) tot += p[(size_t)g * width];
    }
    scratch[threadIdx.x] = tot;
    __syncthreads();
    if ((int)threadIdx.x < width) {
        double sum = 0.0;
        for (int k = 0; k < n_lane_groups; ++k) sum += scratch[k * width + threadIdx.x];
        out[(size_t)step * width + threadIdx.x] = sum;
    }
}

// ------------------------------------------------------------------------ K5
// real parts only, for matrices with imag(H) = 0 (real symmetric eigensolver)
__global__ void scatter_dense_real(const int* __restrict__ indptr, const int* __restrict__ indices,
                                   const double2* __restrict__ blocks, double* __restrict__ dense, int nb) {
    const int64_t n = 4 * (int64_t)nb;
    const int i = blockIdx.x;
    for (int k = indptr[i] + (threadIdx.x >> 4); k < indptr[i + 1]; k += blockDim.x >> 4) {
        const int el = threadIdx.x & 15;
        const int64_t row = 4 * (int64_t)i + (el >> 2);
        const int64_t col = 4 * (int64_t)indices[k] + (el & 3);
        dense[col * n + row] = blocks[(size_t)k * 16 + el].x;
    }
}

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

// ------------------------------------------------------------------------ K6
// Dense Hermitian eigensolver for small matrices: one-sided (Hestenes) Jacobi on the columns
// of G = A = H + shift*I (positive definite for shift > |H|), accumulating the rotations in V.
// At convergence the columns of G are orthogonal, G = A V, so column p of V is an eigenvector
// and |g_p| - shift its eigenvalue.  One launch = one round of the round-robin tournament:
// n/2 disjoint column pairs, one workgroup per pair (no races, no atomics on the data).
//   c = g_p^H g_q = |c| e^{iφ},  ζ = (|g_q|² - |g_p|²) / (2|c|),  t = sign(ζ) / (|ζ| + sqrt(1+ζ²))
//   g_p' = cs g_p - sn e^{-iφ} g_q,   g_q' = sn e^{iφ} g_p + cs g_q          (same for V)
__device__ inline void jacobi_pair(int n, int round, int k, int& p, int& q) {
    // circle method: player n-1 stays, the others rotate
    const int m = n - 1;
    if (k == 0) {
        p = m;
        q = round % m;
    } else {
        p = (round + k) % m;
        q = (round - k + m) % m;
    }
}

// Scalar helpers so that one kernel text serves complex (double2) and real (double) matrices; the
// real form is used when imag(H) = 0 (rotations then have no phase): half the bytes.
__device__ inline double jnorm2(double2 x) { return fma(x.x, x.x, x.y * x.y); }
__device__ inline double jnorm2(double x) { return x * x; }
__device__ inline double2 jdot(double2 x, double2 y) {  // conj(x) y
    return make_double2(fma(x.x, y.x, x.y * y.y), fma(x.x, y.y, -x.y * y.x));
}
__device__ inline double2 jdot(double x, double y) { return make_double2(x * y, 0.0); }
// (nx, ny) = (cs x - sn e^{-iφ} y,  sn e^{iφ} x + cs y),  e^{iφ} = er + i ei
__device__ inline void jrotate(double2& x, double2& y, double cs, double sn, double er, double ei) {
    const double s1r = sn * er, s1i = -sn * ei, s2r = sn * er, s2i = sn * ei;
    double2 nx, ny;
    nx.x = cs * x.x - (s1r * y.x - s1i * y.y);
    nx.y = cs * x.y - (s1r * y.y + s1i * y.x);
    ny.x = (s2r * x.x - s2i * x.y) + cs * y.x;
    ny.y = (s2r * x.y + s2i * x.x) + cs * y.y;
    x = nx;
    y = ny;
}
__device__ inline void jrotate(double& x, double& y, double cs, double sn, double er, double) {
    const double s = sn * er;  // er = ±1
    const double nx = cs * x - s * y, ny = s * x + cs * y;
    x = nx;
    y = ny;
}

// One round.  The two columns of G stay in registers between the dot products and the rotation
// (n <= 256 * kJacobiElems), so G is read once and written once per round; V is read and written.
constexpr int kJacobiElems = 8;       // n <= 2048
constexpr int kJacobiElemsWide = 16;  // n <= 4096 (twice the registers per thread)

template <typename T, int ELEMS>
__global__ __launch_bounds__(256) void jacobi_round(T* __restrict__ G, T* __restrict__ V, int n, int round,
                                                    double tol, int* __restrict__ rotations) {
    constexpr int kJacobiElems = ELEMS;
    __shared__ double red[4][256];
    __shared__ double rot[4];  // cs, sn, cos φ, sin φ  (sn = 0: skip)
    int p, q;
    jacobi_pair(n, round, blockIdx.x, p, q);
    T* gp = G + (size_t)p * n;
    T* gq = G + (size_t)q * n;
    T x[kJacobiElems], y[kJacobiElems];
    double a = 0.0, b = 0.0, cr = 0.0, ci = 0.0;
#pragma unroll
    for (int u = 0; u < kJacobiElems; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < n) {
            x[u] = gp[i];
            y[u] = gq[i];
            a += jnorm2(x[u]);
            b += jnorm2(y[u]);
            const double2 c = jdot(x[u], y[u]);
            cr += c.x;
            ci += c.y;
        }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    red[2][threadIdx.x] = cr;
    red[3][threadIdx.x] = ci;
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + stride];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double aa = red[0][0], bb = red[1][0], re = red[2][0], im = red[3][0];
        const double mag = sqrt(re * re + im * im);
        if (mag <= tol * sqrt(aa * bb) || mag == 0.0) {
            rot[1] = 0.0;
        } else {
            const double zeta = (bb - aa) / (2.0 * mag);
            const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double cs = 1.0 / sqrt(1.0 + t * t);
            rot[0] = cs;
            rot[1] = cs * t;
            rot[2] = re / mag;
            rot[3] = im / mag;
            atomicAdd(rotations, 1);
        }
    }
    __syncthreads();
    const double sn = rot[1];
    if (sn == 0.0) return;
    const double cs = rot[0], er = rot[2], ei = rot[3];
#pragma unroll
    for (int u = 0; u < kJacobiElems; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < n) {
            jrotate(x[u], y[u], cs, sn, er, ei);
            gp[i] = x[u];
            gq[i] = y[u];
        }
    }
    if (V) {
        T* vp = V + (size_t)p * n;
        T* vq = V + (size_t)q * n;
        for (int i = threadIdx.x; i < n; i += 256) {
            T vx = vp[i], vy = vq[i];
            jrotate(vx, vy, cs, sn, er, ei);
            vp[i] = vx;
            vq[i] = vy;
        }
    }
}

// G += shift * I;  V = I (if given)
__device__ inline void jadd_real(double2& g, double v) { g.x += v; }
__device__ inline void jadd_real(double& g, double v) { g += v; }
__device__ inline void jset_real(double2& g, double v) { g = make_double2(v, 0.0); }
__device__ inline void jset_real(double& g, double v) { g = v; }

template <typename T>
__global__ void jacobi_setup(T* __restrict__ G, T* __restrict__ V, int n, double shift) {
    const int64_t total = (int64_t)n * n;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const bool diag = (idx / n) == (idx % n);
        if (diag) jadd_real(G[idx], shift);
        if (V) jset_real(V[idx], diag ? 1.0 : 0.0);
    }
}

// eig[p] = |g_p| - shift, one block per column
template <typename T>
__global__ __launch_bounds__(256) void jacobi_eigenvalues(const T* __restrict__ G, int n, double shift,
                                                          double* __restrict__ eig) {
    __shared__ double red[256];
    const T* g = G + (size_t)blockIdx.x * n;
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += jnorm2(g[i]);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride) red[threadIdx.x] += red[threadIdx.x + stride];
        __syncthreads();
    }
    if (threadIdx.x == 0) eig[blockIdx.x] = sqrt(red[0]) - shift;
}

// max |H - H^†| over the stored entries (reference hamiltonian.py:121-122, `M - M.getH()`): block
// (i, j) against the conjugate transpose of block (j, i), found by binary search in the sorted
// row j; a block without a stored partner counts with its own magnitude.  Sixteen lanes per block
// row, one per block element: a block is one 256-byte run read by 16 adjacent lanes, its partner
// the same 256 bytes read transposed (same cache lines), so the matrix crosses the fabric less than
// twice: 2.17 GB for the 1.28 GB matrix of 10^6 sites, 0.53 ms (profiles/r03_hermiticity.log; the
// thread-per-row form of round 2: 8.7 GB, 2.39 ms).  One partial maximum per workgroup.
__global__ __launch_bounds__(256) void hermiticity_defect(const int* __restrict__ indptr,
                                                          const int* __restrict__ indices,
                                                          const double2* __restrict__ blocks, int nb,
                                                          double* __restrict__ partial) {
    __shared__ double red[256];
    const int el = threadIdx.x & 15, twin = (el & 3) * 4 + (el >> 2);
    const int rows_per_pass = gridDim.x * (blockDim.x >> 4);
    double worst = 0.0;
    for (int i = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); i < nb; i += rows_per_pass)
        for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
            const int j = indices[k];
            int lo = indptr[j], hi = indptr[j + 1] - 1, m = -1;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const int c = indices[mid];
                if (c == i) {
                    m = mid;
                    break;
                }
                if (c < i) lo = mid + 1;
                else hi = mid - 1;
            }
            const double2 a = load_stream(blocks + (size_t)k * 16 + el);
            double2 b = make_double2(0.0, 0.0);
            if (m >= 0) {
                const double2 t = blocks[(size_t)m * 16 + twin];
                b = make_double2(t.x, -t.y);
            }
            worst = fmax(worst, hypot(a.x - b.x, a.y - b.y));
        }
    red[threadIdx.x] = worst;
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + stride]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// count of non-finite doubles in x[0..n) (eigensolver results are checked on the device, before
// any copy to the host): one atomicAdd per workgroup that saw one
__global__ void count_nonfinite(const double* __restrict__ x, int64_t n, int* __restrict__ count) {
    int bad = 0;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[idx];
        bad |= !(fabs(v) <= 1.79769313486231570815e308);
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicAdd(count, 1);
}

// planar [4][nb][rv] column r  <->  site-major [nb][4]
__global__ void planar_from_sitemajor(const double2* __restrict__ x, double2* __restrict__ planar,
                                      int64_t nb, int rv, int r) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < 4 * nb;
         idx += (int64_t)gridDim.x * blockDim.x)
        planar[vslot((int)(idx & 3), (size_t)(idx >> 2), r, (size_t)nb, rv)] = x[idx];
}

__global__ void sitemajor_from_planar(const double2* __restrict__ planar, double2* __restrict__ x,
                                      int64_t nb, int rv, int r) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < 4 * nb;
         idx += (int64_t)gridDim.x * blockDim.x)
        x[idx] = planar[vslot((int)(idx & 3), (size_t)(idx >> 2), r, (size_t)nb, rv)];
}

}  // namespace bdg
