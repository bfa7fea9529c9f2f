// K7  cheb_sweep  - the Chebyshev recurrence on a nearest-neighbour lattice stencil, TWO steps per
//                   sweep of the vectors (or one), for matrices that pass the stencil test below.
//
// Why: the one-step kernels (kernels.hpp) read t_n and t_{n-1} and write t_{n+1}: three array
// passes per step, and they already run at what a trivial 2-read-1-write stream reaches on this
// chip (tools/stream_probe2.hip).  More steps per second therefore needs fewer bytes.  Here one
// launch reads t_n and t_{n-1} once and writes t_{n+1} AND t_{n+2}: four passes per two steps
// instead of six.  t_{n+1} never comes back from memory: every value of it that step two needs is
// still in a register of the lane that made it, or of a neighbouring lane of the same wave.
//
// How: block rows are lattice sites numbered p + P*x (P = rows per x-plane; reference
// lattice.py:108 with z + Lz*y = p), and the matrix is a stencil: block (i, j) is stored only for
// j - i in {-P, -1, 0, +1, +P}, with +-1 inside one plane.  (Open boundaries in a 2-D lattice;
// anything else - periodic wrap blocks, 3-D, general sparsity - fails the test made on the device
// by `build_stencil` and runs the one-step kernels.)
//
// A wave owns a window of 12 consecutive in-plane positions and marches along x through one
// segment of planes.  Its 16 site slots are [ghost, halo, 12 owned, halo, ghost]; lane (slot s,
// vector group r) keeps, for its own position p:
//     t_n     at planes k-1, k, k+1        (rolling registers)
//     t_{n+1} at planes k-2, k-1, k        (rolling registers; plane k is made this iteration)
// so the +-P neighbours are the lane's own registers and the +-1 neighbours are the registers of
// lanes s-1 and s+1, handed over through a wave-private LDS row.  Per iteration k the wave
//     loads   t_n[plane k+2], t_{n-1}[plane k+1]   (one plane ahead of their use: prefetch)
//     step 1  t_{n+1}[k] = c1 H t_n - t_{n-1}      on slots 1..14  (halo slots redundantly)
//     step 2  t_{n+2}[k-1] = c2 H t_{n+1} - t_n    on slots 2..13
//     stores  t_{n+1}[k], t_{n+2}[k-1] of the owned slots (6 whole 128-byte lines each)
// There are no gathers: every global access is a unit-stride 1 KiB wave access, every byte of the
// four arrays is touched once per launch (plus the halo slots and the two extra planes at each
// segment end, which are served by L1/L2 when the neighbouring window runs on the same CU/XCD).
// Waves never communicate: halo values are recomputed, not exchanged, so there is no barrier, no
// flag and no ordering requirement between workgroups.  Same Mode::mac_row on the same numbers in
// the same (CSR) order as the one-step kernels => t_{n+1}, t_{n+2} are bit-identical to theirs;
// only the summation order of the dot products differs.
//
// Algorithmic bytes per launch (two steps of RV vectors, 16-byte lane payloads, RL = 4 lanes per
// site): 8 nb (stencil words) + table + 4 arrays x 256 nb.
#pragma once

#include "kernels.hpp"

namespace bdg {

constexpr int kSweepLanes = 4;                      // lanes per site (RL) of the default configuration
constexpr int kSweepSlots = kWave / kSweepLanes;    // 16 site slots per wave
constexpr int kSweepOwned = kSweepSlots - 4;        // 12 owned positions per wave window
// K7 also exists with 2 and 1 lanes per site (4 and 2 real vectors per launch, windows of 28 and
// 60 owned positions): with fewer vectors per launch the four vector buffers of a run get small
// enough to live in the 256 MB Infinity Cache from one launch to the next.
constexpr int sweep_owned(int rl) { return kWave / rl - 4; }
constexpr unsigned kNoBlock = 0xFFu;

struct SweepArgs {
    const uint2* stencil;    // per block row: table ids of the blocks at offsets -P, -1, 0, +1, +P
                             // (bytes 0..4 of the 8-byte word, kNoBlock = not stored)
    const void* dict_table;  // the distinct blocks, packed for the mode
    int n_unique;
    const double2* onsite;   // cheb_sweep3<..., OS>: packed on-site block of every row (Mode::kOnsiteSlots x 16 B)
    const double2* cur;      // t_n
    const double2* prev;     // t_{n-1}; nullptr = zero (first sweep of a run: t_{-1} = 0 is not read)
    double2* out1;           // t_{n+1}
    double2* out2;           // t_{n+2}
    double* partial1;        // [gridDim.x][RL * kVec][2]  dots of step 1: <t_n|t_n>, <t_{n+1}|t_n>
    double* partial2;        //                            dots of step 2: <t_{n+1}|t_{n+1}>, <t_{n+2}|t_{n+1}>
    double coef1, coef2;
    int nb;                  // block rows = lx * plane
    int plane;               // P
    int lx;
    int n_cols;              // windows per plane = ceil(P / 12)
    int n_segs;              // segments along x
    int two;                 // cheb_sweep: 1 = both steps, 0 = step 1 only (odd tail of a run)
    int steps;               // cheb_sweep3: steps this launch makes, 1..3 (out1 = t_{n+steps-1}, out2 = t_{n+steps})
    double* partial3;        // cheb_sweep3: dots of step 3
    int stream;              // non-temporal hints: bit 0 t_{n-1} loads, bit 1 stores, bit 2 t_n loads, bit 3 on-site records
    int zigzag;              // 1 = odd segments march against the even ones
    int wrap_p;              // 1 = the plane is a ring: position P-1 neighbours position 0 (periodic edge blocks)
    int wrap_x;              // 1 = the planes form a ring: plane lx-1 neighbours plane 0
    // cheb_sweep3<..., GEN = true>: t_n is the random start vector block and is made in registers
    // from the counter-based generator (the values fill_random would have written) instead of read
    uint64_t gen_seed, gen_first_id;
    int gen_kind;            // BDG_VEC_* (real modes: Rademacher)
    int gen_active;          // vectors of the launch that exist (the others are zero)
    // 1 = nothing is stored: the last sweep of a run, whose vectors nothing reads any more (the call returns the
    // dot products only).  Every step is computed and dotted as in any other launch.
    int discard;
    // Planes [x_lo, x_hi) are advanced by this launch (0, lx = the whole lattice).  Unit start vectors spread by one
    // plane per step: outside the band everything is zero, and the buffers hold zeros there (recurrence.hpp).
    int x_lo, x_hi;
};

// Stencil table + eligibility test, on the device from the uploaded arrays.  `words` is the
// dictionary form of the matrix (column | id << 24 per stored block).  bad[0] is raised if any
// stored block is not one of the five stencil offsets.
// Byte 5 of the word (K7; byte 7 for K8's seven offsets) is a bit mask: bit o set = the block at
// offset o is diagonal as a 4x4 matrix (`diagonal[id]`, found on the host when the blocks were
// deduplicated), so that its product takes 4 multiply-adds per vector instead of 16.
// Periodic lattices: the wrap-around blocks of the reference's `lattice.edges()` (position P-1 <->
// position 0 inside a plane, plane lx-1 <-> plane 0) are the same neighbours seen across the
// seam; they take the slots of the -1 / +1 / -P / +P neighbours they are, and bad[1] / bad[2]
// tell the kernels to close the plane / the stack of planes into rings.
// `onsite_streamed`: the words of the diagonal blocks carry no table id (the sweep reads those blocks
// from the per-site stream); their slot is only marked present.  2 = the same for every block (bond blocks
// streamed as well: pack_site_records).
// Bits 13..17 of the second word (round 4): the block at offset o is a "singlet" block (`diagonal[id]` == 2: A diagonal, B and
// C antidiagonal - kernels.hpp mac_singlet), read by cheb_sweep3 only.
__global__ void build_stencil(const int* __restrict__ indptr, const int* __restrict__ words,
                              const int* __restrict__ diagonal, int nb, int plane, int onsite_streamed,
                              uint2* __restrict__ stencil, int* __restrict__ bad) {
    const int lx = nb / plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        unsigned id[5] = {kNoBlock, kNoBlock, kNoBlock, kNoBlock, kNoBlock};
        unsigned mask = 0;
        const int p = i % plane, x = i / plane;
        bool ok = true;
        for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
            const unsigned w = (unsigned)words[k];
            const int off = (int)(w & 0xFFFFFFu) - i;
            int slot = -1;
            if (off == -plane) slot = 0;
            else if (off == -1 && p >= 1) slot = 1;
            else if (off == 0) slot = 2;
            else if (off == 1 && p <= plane - 2) slot = 3;
            else if (off == plane) slot = 4;
            else if (off == plane - 1 && p == 0) {
                slot = 1;
                atomicOr(bad + 1, 1);
            } else if (off == -(plane - 1) && p == plane - 1) {
                slot = 3;
                atomicOr(bad + 1, 1);
            } else if (off == (lx - 1) * plane && x == 0) {
                slot = 0;
                atomicOr(bad + 2, 1);
            } else if (off == -(lx - 1) * plane && x == lx - 1) {
                slot = 4;
                atomicOr(bad + 2, 1);
            }
            if (slot < 0 || (w >> 24) == kNoBlock) ok = false;
            else if (onsite_streamed == 2 || (onsite_streamed && slot == 2)) id[slot] = 0;  // (2: every block is streamed)
            else {
                id[slot] = w >> 24;
                if (diagonal[w >> 24] == 1) mask |= 1u << slot;
                // (bond blocks only: for an on-site singlet block - the s-wave models - the saving and the extra test cancel,
                // 110.2-111.1 against 110.9-111.4 k vector-steps/s; four d-wave bond blocks per site gain 5.5 %)
                if (diagonal[w >> 24] == 2 && slot != 2) mask |= 32u << slot;
            }
        }
        if (!ok) atomicOr(bad, 1);
        stencil[i] = make_uint2(id[0] | (id[1] << 8) | (id[2] << 16) | (id[3] << 24), id[4] | (mask << 8));
    }
}

// Per-site records of a matrix whose bond blocks are streamed too (bond blocks diagonal as 4x4 matrices of the Nambu
// form diag(a, b, -conj a, -conj b)): the packed on-site block of Mode::mac_onsite, then (A00, A11) of the blocks at
// the stencil offsets -P, -1, +1, +P (zeros where not stored), found by the rule of build_stencil (wrap-around blocks
// of periodic lattices included).
//   real_out:  8 slots of 16 bytes - on-site slots 0..3, one slot (A00, A11) per bond                     (128 B per site)
//   otherwise: 14 slots - on-site slots 0..5, two slots A00, A11 per bond (Peierls phases, complex hopping)  (224 B per site)
__global__ void pack_site_records(const int* __restrict__ indptr, const int* __restrict__ indices,
                                  const double2* __restrict__ blocks, int nb, int plane, int real_out, double2* __restrict__ out) {
    const int lx = nb / plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const double2 zero = make_double2(0.0, 0.0);
        double2 rec[14] = {zero, zero, zero, zero, zero, zero, zero, zero, zero, zero, zero, zero, zero, zero};
        const int p = i % plane, x = i / plane;
        for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
            const double2* blk = blocks + (size_t)k * 16;
            const int off = indices[k] - i;
            int slot = -1;
            if (off == -plane) slot = 0;
            else if (off == -1 && p >= 1) slot = 1;
            else if (off == 0) slot = 2;
            else if (off == 1 && p <= plane - 2) slot = 3;
            else if (off == plane) slot = 4;
            else if (off == plane - 1 && p == 0) slot = 1;
            else if (off == -(plane - 1) && p == plane - 1) slot = 3;
            else if (off == (lx - 1) * plane && x == 0) slot = 0;
            else if (off == -(lx - 1) * plane && x == lx - 1) slot = 4;
            const int bond = slot < 2 ? slot : slot - 1;
            if (slot == 2 && real_out) {
                rec[0] = make_double2(blk[0].x, blk[1].x);
                rec[1] = make_double2(blk[5].x, 0.0);
                rec[2] = make_double2(blk[2].x, blk[3].x);
                rec[3] = make_double2(blk[6].x, blk[7].x);
            } else if (slot == 2) {  // (pack_onsite's complex record)
                rec[0] = make_double2(blk[0].x, blk[5].x);
                rec[1] = blk[1];
                rec[2] = blk[2], rec[3] = blk[3], rec[4] = blk[6], rec[5] = blk[7];
            } else if (slot >= 0 && real_out) {
                rec[4 + bond] = make_double2(blk[0].x, blk[5].x);
            } else if (slot >= 0) {
                rec[6 + 2 * bond] = blk[0];
                rec[7 + 2 * bond] = blk[5];
            }
        }
        const int slots = real_out ? 8 : 14;
        for (int e = 0; e < slots; ++e) out[(size_t)i * slots + e] = rec[e];
    }
}

// Marching direction.  A segment marched "in reverse" runs from its far end back (plane k of the
// text below is then plane x0 + x1 - 1 - k of the lattice).  Two uses:
//  * neighbouring segments march in opposite directions (even ones up, odd ones down), so the two
//    waves on either side of a segment boundary read the planes around it at the same time - the
//    planes each recomputes for the other are then L2 / Infinity-Cache hits instead of a second
//    trip to HBM;
//  * REV flips all of them: launches alternate, so that each starts on the planes the previous
//    launch wrote last, i.e. the ones the Infinity Cache still holds.
// Wave reduction over the site lanes, then over the workgroup's waves: partial[block][vector][{d, e}].
template <typename Mode, int RL, int WAVES>
__device__ inline void sweep_reduce_dots(double dot[4], double* red, double* partial, int lane, int wave) {
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
        for (int w = 0; w < WAVES; ++w) tot += red[w * RL * W + threadIdx.x];
        partial[(size_t)blockIdx.x * RL * W + threadIdx.x] = tot;
    }
}

// kSweepWaves consecutive windows per workgroup: the windows of one workgroup share their halo
// reads through the CU's L1, those of different workgroups through L2.  (8 waves per workgroup,
// one workgroup per CU, measured the same as 4 and two: profiles/r02_sweep_experiments.log.)
#ifndef BDG_SWEEP_WAVES
#define BDG_SWEEP_WAVES 4
#endif
constexpr int kSweepWaves = BDG_SWEEP_WAVES;
constexpr int kSweepThreads = kSweepWaves * kWave;

template <typename Mode, int RL, bool REV>
__global__ __launch_bounds__(kSweepThreads, 2) void cheb_sweep(SweepArgs a) {
    extern __shared__ double2 lds[];
    constexpr int SLOTS = kWave / RL;   // site slots per wave: [ghost, halo, owned ..., halo, ghost]
    constexpr int OWNED = SLOTS - 4;
    constexpr int SPB = Mode::kSlotsPerBlock;
    constexpr int STRIDE = Mode::kBlockStride;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int s = lane / RL;
    const int r = lane % RL;

    // LDS: [table: n_unique x STRIDE slots][per wave: two rows of 64 lanes x 4 entries]
    const double2* table = static_cast<const double2*>(a.dict_table);
    for (int e = threadIdx.x; e < a.n_unique * SPB; e += kSweepThreads)
        lds[(e / SPB) * STRIDE + (e % SPB)] = table[e];
    double2* row_n = lds + a.n_unique * STRIDE + wave * (2 * kWave * 4);  // t_n of plane k
    double2* row_1 = row_n + kWave * 4;                                   // t_{n+1} of plane k-1
    __syncthreads();

    // units = (segment, window); the workgroups of one XCD (b, b+8, ...) take one contiguous
    // eighth of them, neighbouring windows on the waves of one workgroup (shared halos: same L1)
    const int n_units = a.n_cols * a.n_segs;
    const int xcd = blockIdx.x & 7;
    const int u_lo = (int)(((int64_t)n_units * xcd) >> 3);
    const int u_hi = (int)(((int64_t)n_units * (xcd + 1)) >> 3);
    const int waves_per_xcd = (gridDim.x >> 3) * kSweepWaves;

    double dot1[4] = {0.0, 0.0, 0.0, 0.0}, dot2[4] = {0.0, 0.0, 0.0, 0.0};
    const double2 zero = make_double2(0.0, 0.0);
    const size_t nb = (size_t)a.nb;
    const bool nt_prev = a.stream & 1, nt_store = a.stream & 2, nt_cur = a.stream & 4;

    for (int u = u_lo + (int)(blockIdx.x >> 3) * kSweepWaves + wave; u < u_hi; u += waves_per_xcd) {
        const int seg = u / a.n_cols, col = u - seg * a.n_cols;
        const int x0 = a.x_lo + (int)(((int64_t)(a.x_hi - a.x_lo) * seg) / a.n_segs);
        const int x1 = a.x_lo + (int)(((int64_t)(a.x_hi - a.x_lo) * (seg + 1)) / a.n_segs);
        const int p = col * OWNED - 2 + s;
        const bool inside = p >= 0 && p < a.plane;
        const bool valid = inside || a.wrap_p;  // a ring has no edge: halo slots beyond it hold the far side
        const int pw = inside ? p : ((p % a.plane) + a.plane) % a.plane;
        const bool does1 = valid && s >= 1 && s <= SLOTS - 2;
        const bool owned = inside && s >= 2 && s <= SLOTS - 3;

        const bool rev = REV != (bool)(a.zigzag & seg & 1);  // wave-uniform
        auto act = [&](int k) { return rev ? x0 + x1 - 1 - k : k; };  // lattice plane of marching index k
        // (with periodic planes the ones recomputed beyond either end of the stack are those of the far end)
        auto ring = [&](int k) { return a.wrap_x ? (k < 0 ? k + a.lx : (k >= a.lx ? k - a.lx : k)) : k; };
        auto load_plane = [&](const double2* buf, bool nt, int k, bool wanted, double2 out[4]) {
            k = ring(act(k));
            if (wanted && k >= 0 && k < a.lx) {
                const size_t site = (size_t)k * a.plane + pw;
#pragma unroll
                for (int al = 0; al < 4; ++al)
                    out[al] = nt ? load_stream(buf + vslot(al, site, r, nb, RL)) : buf[vslot(al, site, r, nb, RL)];
            } else {
#pragma unroll
                for (int al = 0; al < 4; ++al) out[al] = zero;
            }
        };
        auto load_ids = [&](int k) {
            uint2 w = make_uint2(0xFFFFFFFFu, 0xFFu);
            k = ring(act(k));
            if (does1 && k >= 0 && k < a.lx) w = a.stencil[(size_t)k * a.plane + pw];
            return w;
        };
        auto id_of = [](uint2 w, int slot) { return slot < 4 ? (w.x >> (8 * slot)) & 0xFFu : w.y & 0xFFu; };
        // acc += Σ_offsets block * x, in CSR (ascending column) order: -P, -1, 0, +1, +P
        // one block times one site's entries: 4 MACs if the block is flagged diagonal, else 16
        auto mac = [&](uint2 w, int slot, const double2 x[4], double2 acc[4]) {
            const unsigned id = id_of(w, slot);
            if (id == kNoBlock) return;
            if ((w.y >> (8 + slot)) & 1u) Mode::mac_diag(acc, lds + id * STRIDE, x);
            else Mode::mac_row(acc, lds + id * STRIDE, x);
        };
        auto apply = [&](uint2 w, const double2 lo[4], const double2* row, const double2 mid[4],
                         const double2 hi[4], double2 acc[4]) {
            double2 x[4];
            mac(w, 0, lo, acc);
            if (id_of(w, 1) != kNoBlock) {
#pragma unroll
                for (int be = 0; be < 4; ++be) x[be] = row[SHARE_SLOT(lane - RL, be)];
                mac(w, 1, x, acc);
            }
            mac(w, 2, mid, acc);
            if (id_of(w, 3) != kNoBlock) {
#pragma unroll
                for (int be = 0; be < 4; ++be) x[be] = row[SHARE_SLOT(lane + RL, be)];
                mac(w, 3, x, acc);
            }
            mac(w, 4, hi, acc);
        };

        // Step 1 runs on planes k_first..k_last: the segment itself, plus one plane on either side
        // when step 2 follows (its +-P neighbours at the segment ends).
        const int k_first = a.two ? x0 - 1 : x0, k_last = a.two ? x1 : x1 - 1;

        // ---- prologue: t_n of planes k_first-1, k_first, k_first+1; t_{n-1} and ids of plane k_first
        double2 cn_m[4], cn_0[4], cn_p[4], pv[4], c1_m[4], c1_0[4];
        load_plane(a.cur, nt_cur, k_first - 1, valid, cn_m);
        load_plane(a.cur, nt_cur, k_first, valid, cn_0);
        load_plane(a.cur, nt_cur, k_first + 1, valid, cn_p);
        load_plane(a.prev, nt_prev, k_first, does1 && a.prev != nullptr, pv);
        uint2 ids_0 = load_ids(k_first), ids_m = make_uint2(0xFFFFFFFFu, 0xFFu);
#pragma unroll
        for (int al = 0; al < 4; ++al) c1_m[al] = c1_0[al] = zero;

        for (int k = k_first; k <= k_last; ++k) {
            // ---- prefetch what the next iteration consumes
            double2 nx_cn[4], nx_pv[4];
            const bool more = k < k_last;
            load_plane(a.cur, nt_cur, k + 2, valid && more, nx_cn);
            load_plane(a.prev, nt_prev, k + 1, does1 && more && a.prev != nullptr, nx_pv);
            const uint2 nx_ids = more ? load_ids(k + 1) : make_uint2(0xFFFFFFFFu, 0xFFu);

            // ---- step 1 on plane k: t_{n+1} = c1 H t_n - t_{n-1}
#pragma unroll
            for (int be = 0; be < 4; ++be) row_n[SHARE_SLOT(lane, be)] = cn_0[be];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            double2 new1[4];
#pragma unroll
            for (int al = 0; al < 4; ++al) new1[al] = zero;
            const bool plane_ok = a.wrap_x || (act(k) >= 0 && act(k) < a.lx);
            if (does1 && plane_ok) {
                double2 acc[4];
#pragma unroll
                for (int al = 0; al < 4; ++al) acc[al] = zero;
                if (rev) apply(ids_0, cn_p, row_n, cn_0, cn_m, acc);
                else apply(ids_0, cn_m, row_n, cn_0, cn_p, acc);
#pragma unroll
                for (int al = 0; al < 4; ++al) {
                    new1[al].x = fma(a.coef1, acc[al].x, -pv[al].x);
                    new1[al].y = fma(a.coef1, acc[al].y, -pv[al].y);
                }
                if (owned && k >= x0 && k < x1) {
                    const size_t site = (size_t)act(k) * a.plane + p;
#pragma unroll
                    for (int al = 0; al < 4; ++al) {
                        if (a.discard) {
                        } else if (nt_store) store_stream(a.out1 + vslot(al, site, r, nb, RL), new1[al]);
                        else a.out1[vslot(al, site, r, nb, RL)] = new1[al];
                        Mode::dots(dot1, cn_0[al], new1[al]);
                    }
                }
            }

            // ---- step 2 on plane k-1: t_{n+2} = c2 H t_{n+1} - t_n   (row_1 holds t_{n+1}[k-1])
            if (a.two) {
                if (owned && k - 1 >= x0 && k - 1 < x1) {
                    double2 acc[4];
#pragma unroll
                    for (int al = 0; al < 4; ++al) acc[al] = zero;
                    if (rev) apply(ids_m, new1, row_1, c1_0, c1_m, acc);
                    else apply(ids_m, c1_m, row_1, c1_0, new1, acc);
                    const size_t site = (size_t)act(k - 1) * a.plane + p;
#pragma unroll
                    for (int al = 0; al < 4; ++al) {
                        double2 nx;
                        nx.x = fma(a.coef2, acc[al].x, -cn_m[al].x);
                        nx.y = fma(a.coef2, acc[al].y, -cn_m[al].y);
                        if (a.discard) {
                        } else if (nt_store) store_stream(a.out2 + vslot(al, site, r, nb, RL), nx);
                        else a.out2[vslot(al, site, r, nb, RL)] = nx;
                        Mode::dots(dot2, c1_0[al], nx);
                    }
                }
                // t_{n+1}[k] takes the place of t_{n+1}[k-1] in the hand-over row (same wave: ordered)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int be = 0; be < 4; ++be) row_1[SHARE_SLOT(lane, be)] = new1[be];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();

            // ---- roll
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                cn_m[al] = cn_0[al];
                cn_0[al] = cn_p[al];
                cn_p[al] = nx_cn[al];
                pv[al] = nx_pv[al];
                c1_m[al] = c1_0[al];
                c1_0[al] = new1[al];
            }
            ids_m = ids_0;
            ids_0 = nx_ids;
        }
    }

    __syncthreads();  // table / hand-over rows are done with: the reductions reuse the front of the LDS
    sweep_reduce_dots<Mode, RL, kSweepWaves>(dot1, reinterpret_cast<double*>(lds), a.partial1, lane, wave);
    if (a.two) {
        __syncthreads();
        sweep_reduce_dots<Mode, RL, kSweepWaves>(dot2, reinterpret_cast<double*>(lds), a.partial2, lane, wave);
    }
}


// =====================================================================================
// K7b  cheb_sweep3 - THREE recurrence steps per sweep of the vectors.
//
// A sweep that makes k steps still reads only t_n and t_{n-1} and writes only the last two
// levels t_{n+k-1}, t_{n+k}: four array passes per k steps.  k = 2 (cheb_sweep) moves 2/3 of the
// one-step kernels' bytes, k = 3 moves 4/9: the intermediate level t_{n+1} is never stored at
// all.  The price is one more plane pair of rolling state per level, a halo slot more on either
// side of the window (16 slots = [3 halo, 10 owned, 3 halo]: step j is valid on slots j..15-j),
// and one more recomputed plane at either end of a segment.  To stay within two waves per SIMD
// the current plane of every level lives in its LDS hand-over row (the lane's own slot) instead
// of a register; only the planes before and after it are registers.
//
// Iteration k of a wave (plane indices along its march):
//     step 1 on plane k     from t_n[k-1,k,k+1]                  -> level 1
//     step 2 on plane k-1   from level 1 [k-2, k-1, k]           -> level 2   (k was just made)
//     step 3 on plane k-2   from level 2 [k-3, k-2, k-1]         -> level 3
// a launch may make fewer (steps = 1 or 2: the tail of a run whose length is not a multiple of 3).
#ifndef BDG_COMPACT_DIAG
#define BDG_COMPACT_DIAG 1  // (0: A/B builds that read diagonal blocks from the full table entries)
#endif
#ifndef BDG_OS_EXPERIMENT
#define BDG_OS_EXPERIMENT 0
#endif
#ifndef BDG_SINGLET_BLOCKS
#define BDG_SINGLET_BLOCKS 1  // (0: A/B builds that multiply "singlet" blocks as full blocks)
#endif
#ifndef BDG_COMPLEX_ONSITE_STRIDE
#define BDG_COMPLEX_ONSITE_STRIDE 6  // (7 = ComplexPHMode::kOnsiteStride, the padded stride of round 3, for A/B builds)
#endif
constexpr int kSweep3Owned = kSweepSlots - 6;  // 10 owned positions per wave window (4 lanes per site)
constexpr int sweep3_owned(int rl) { return kWave / rl - 6; }

// GEN: the first sweep of a stochastic-trace run.  t_0 is a block of random vectors whose entries are
// a pure function of (seed, vector, scalar row) (K3), so the sweep makes the planes of t_0 it needs
// in registers - the halo slots and segment-end planes included - and the 128 MB of t_0 are neither
// written by a fill kernel nor read back: a 40-moment call (7 sweeps per lane group) is 7 % shorter.
//
// OS ("on-site blocks streamed"): position-dependent on-site terms - a disorder potential, a
// self-consistent gap, a magnetic texture (reference hamiltonian.py:102-118 filled per site) - make
// every diagonal block distinct and defeat the dictionary, while the bond blocks still come from a
// handful of distinct ones.  Then only the bonds sit in the LDS table; the diagonal block of every
// site is read from a per-site stream in HBM exactly once per launch (Mode::kOnsiteSlots x 16 B,
// whole-wave contiguous loads one plane ahead of their first use) into a wave-private
// LDS ring of three planes, because step j of iteration k works on plane k-j+1: plane k's blocks
// serve step 1 now, step 2 in the next iteration and step 3 in the one after.  Particle-hole modes; 4 lanes per site
// (16 slots) or, since round 4, 2 lanes (32 slots: twice the ring - in workgroups of up to seven waves, one per CU,
// where eight waves' rings do not fit the LDS any more: template parameter WAVES of cheb_sweep3).
// OS = 2: the bond blocks are streamed as well (matrices whose bond blocks are diagonal as 4x4 matrices - the
// reference's ssd() profile on a model with on-site pairing, bond disorder, Peierls phases of a position-dependent gauge):
// the record of a site is its packed on-site block plus (A00, A11) of its four bond blocks - 8 slots = 128 B in real,
// 14 slots = 224 B in complex arithmetic (pack_site_records) -, no table at all.
template <typename Mode, int OS>
constexpr int sweep3_record_slots() {
    if constexpr (OS == 2) return Mode::kOnsiteSlots + 4 * Mode::kBondSlots;
    else if constexpr (OS == 1) return Mode::kOnsiteSlots;
    else return 0;
}
// LDS stride of a site's record in 16-byte slots: such that the sites a 16-byte read of the wave touches together fall
// on different banks (real: 4 / 8 slots would put every fourth / second site on the same banks - one slot of padding;
// complex: 6 and 14 slots are 24 and 56 banks, whose multiples do not collide within eight sites - no padding, which is
// what lets the ring of a 32-slot window fit)
template <typename Mode, int OS>
constexpr int sweep3_record_stride() {
    if constexpr (OS == 0) return 0;
    else if constexpr (Mode::kVec == 2) return sweep3_record_slots<Mode, OS>() + 1;
    else return OS == 2 ? sweep3_record_slots<Mode, OS>() : BDG_COMPLEX_ONSITE_STRIDE;
}
template <typename Mode, int OS>
constexpr int sweep3_ring_slots(int slots) {
    return 3 * slots * sweep3_record_stride<Mode, OS>();
}
template <typename Mode, int OS>
constexpr int sweep3_onsite_pieces() {
    return sweep3_record_slots<Mode, OS>();
}

// What the wave of a sweep keeps in LDS: the block table, the compact block diagonals behind it, the three hand-over
// rows of the wave (level 0 plane k, level 1 plane k-1, level 2 plane k-2) and (OS) its ring of on-site records.
struct Sweep3Lds {
    double2* table;
    double2* diag;
    double2* singlet;  // compact copies of the "singlet" form of every block (Mode::kSingletSlots each; used where flagged)
    double2* row_0;
    double2* row_1;
    double2* row_2;
    double2* os_ring;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int sweep_u32x2 __attribute__((ext_vector_type(2)));
constexpr int kRawBufferFlags = 0x00020000;  // raw buffer resource of gfx9 / CDNA: 32-bit data format, no swizzle
constexpr int kAuxSc1 = 16;                  // cache-policy bits of the raw buffer builtins: sc1 = write-through
constexpr int kBufferFar = (int)0xFFFFFFE0u;     // ... and one beyond any descriptor of up to 4 GB - 32 (the one-descriptor-per-plane form)
constexpr size_t kSweepSpanLimit = 0xFFFFFF00u;  // bytes such a descriptor may span: 3 components of the whole buffer + a plane
#ifndef BDG_SWEEP_ONE_DESCRIPTOR
#define BDG_SWEEP_ONE_DESCRIPTOR 0  // (1: A/B builds - measured 6 % slower, see load_plane)
#endif
constexpr int kBufferOutOfRange = 0x7FFFFFF0;  // an offset no plane descriptor covers: such loads return zeros, such stores do nothing
constexpr int kAuxNt = 2;                    // ... nt = non-temporal (what __builtin_nontemporal_load / _store set)

// Buffer addressing (BUF): every global access of a unit goes through a raw buffer instruction whose descriptor is the
// plane it touches - base = buffer + (component * rows + plane * P) * RL * 16 bytes, made from wave-uniform values in
// scalar registers - and whose per-lane offset is (position * RL + r) * 16: ONE 32-bit register for the whole unit,
// where flat addressing keeps a 64-bit address per component and buffer.  The compiler hoisted those out of the plane
// loop and, at the 256-register limit of the streamed forms, parked them in scratch: every reload then waits with
// s_waitcnt vmcnt(0), i.e. for every prefetched plane in flight - the record loads of the complex streamed form cost
// 0.10 of its 0.37 ms that way (profiles/r04_onsite_ab.log).  Used by the forms that spilled (complex streamed forms, 2-lane
// streamed forms).
#ifndef BDG_SWEEP_BUFFER_OPS
#define BDG_SWEEP_BUFFER_OPS 1  // 0: flat addressing in the forms that never spilled (A/B builds)
#endif
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sweep_plane_rsrc(const void* buf, size_t byte_offset, int bytes) {
    const uint64_t base = reinterpret_cast<uint64_t>(buf) + byte_offset;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, bytes, kRawBufferFlags);
}

// Lane state of the generated start block (GEN): the lane's vector(s) - real modes carry vectors 2r, 2r+1 in
// (x, y), complex modes vector r.
struct Sweep3Gen {
    uint64_t key0 = 0, key1 = 0;
    bool on0 = false, on1 = false;
};
template <typename Mode>
__device__ inline Sweep3Gen sweep3_gen_keys(uint64_t seed, uint64_t first_id, int active, int r) {
    Sweep3Gen g;
    const int v0 = Mode::kVec == 2 ? 2 * r : r;
    g.on0 = v0 < active;
    g.on1 = Mode::kVec == 2 && v0 + 1 < active;
    g.key0 = vector_key(seed, first_id + v0);
    g.key1 = vector_key(seed, first_id + v0 + 1);
    return g;
}

// One unit of a three-step sweep: the wave marches its window `col` through the planes of segment `seg`
// (`rev`: from the far end back) and adds its dot products to dot1..dot3.  WT: the new planes are stored
// write-through (cheb_march3, where other workgroups read them later in the same launch).
// What changes from one sweep to the next (a kernel argument for cheb_sweep3, made per task by cheb_march3): everything
// else a unit needs stays where the kernel's arguments are - a per-task copy of all of SweepArgs costs registers.
struct Sweep3Task {
    const double2* cur;
    const double2* prev;
    double2* out1;
    double2* out2;
    double coef1;
    int steps;
    int discard;
};

template <typename Mode, int RL, bool GEN, int OS, bool WT>
__device__ __forceinline__ void sweep3_unit(const SweepArgs& a, const Sweep3Task& t, const Sweep3Lds& w, const Sweep3Gen& gen, int lane,
                                            int seg, int col, bool rev, double (&dot1)[4], double (&dot2)[4], double (&dot3)[4]) {
    constexpr int SLOTS = kWave / RL;
    constexpr int OWNED3 = SLOTS - 6;
    constexpr int STRIDE = Mode::kBlockStride;
    constexpr int RING = sweep3_ring_slots<Mode, OS>(SLOTS);  // LDS slots of the on-site ring (0 without OS)
    constexpr int RSTRIDE = sweep3_record_stride<Mode, OS>();
    constexpr int DSL = Mode::kDiagSlots;
    const int s = lane / RL;
    const int r = lane % RL;
    double2* const lds = w.table;
    double2* const diag = w.diag;
    [[maybe_unused]] double2* const singlet = w.singlet;
    double2* const row_0 = w.row_0;
    double2* const row_1 = w.row_1;
    double2* const row_2 = w.row_2;
    [[maybe_unused]] double2* const os_ring = w.os_ring;
    const double2 zero = make_double2(0.0, 0.0);
    const size_t nb = (size_t)a.nb;
    const int steps = t.steps;  // uniform
    const bool nt_prev = a.stream & 1, nt_cur = a.stream & 4;
    [[maybe_unused]] const bool nt_store = a.stream & 2;
    [[maybe_unused]] const uint64_t gen_key0 = gen.key0, gen_key1 = gen.key1;
    [[maybe_unused]] const bool gen_on0 = gen.on0, gen_on1 = gen.on1;

    const int x0 = a.x_lo + (int)(((int64_t)(a.x_hi - a.x_lo) * seg) / a.n_segs);
    const int x1 = a.x_lo + (int)(((int64_t)(a.x_hi - a.x_lo) * (seg + 1)) / a.n_segs);
    const int p = col * OWNED3 - 3 + s;
    const bool inside = p >= 0 && p < a.plane;
    const bool valid = inside || a.wrap_p;  // a ring has no edge: halo slots beyond it hold the far side
    const int pw = inside ? p : ((p % a.plane) + a.plane) % a.plane;
    const bool ok1 = valid && s >= 1 && s <= SLOTS - 2;
    const bool ok2 = valid && s >= 2 && s <= SLOTS - 3;
    const bool owned = inside && s >= 3 && s <= SLOTS - 4;

    auto act = [&](int k) { return rev ? x0 + x1 - 1 - k : k; };
    auto ring = [&](int k) { return a.wrap_x ? (k < 0 ? k + a.lx : (k >= a.lx ? k - a.lx : k)) : k; };
    auto in_lattice = [&](int k) { return a.wrap_x || (act(k) >= 0 && act(k) < a.lx); };
    // (where flat addressing does not spill it is 2-7 % faster - fewer scalar instructions per access: the dictionary forms
    // and the real 4-lane streamed forms keep it)
    constexpr bool BUF = WT || (OS != 0 && (Mode::kVec == 1 || RL == 2)) || BDG_SWEEP_BUFFER_OPS;
    const int plane_bytes = a.plane * RL * (int)sizeof(double2);
    const int lane_bytes = (pw * RL + r) * (int)sizeof(double2);  // (stores: owned lanes only, where p == pw)
    [[maybe_unused]] const unsigned comp_stride = (unsigned)(nb * RL * sizeof(double2));  // bytes from one component of a site to the next
    [[maybe_unused]] const int span_bytes = (int)(3u * comp_stride + (unsigned)plane_bytes);  // (below 4 GB: kSweepSpanLimit, checked on the host)
    auto load_plane = [&](const double2* buf, bool nt, int k, bool wanted, double2 out[4]) {
        k = ring(act(k));
        if constexpr (BUF) {
            // no branch: a plane that does not exist gets a descriptor of length zero, a lane that does not take part an
            // offset beyond any plane - out-of-range buffer loads return zeros.  The loads are then on every path, and
            // the compiler can wait for an older plane with a counted s_waitcnt vmcnt(N) that leaves these in flight
            // (with the loads behind a branch it has to assume the path without them: vmcnt(0), see the text above)
            const bool there = buf != nullptr && k >= 0 && k < a.lx;  // (uniform)
#if BDG_SWEEP_ONE_DESCRIPTOR
            // (A/B builds) one descriptor per buffer and plane: it starts at the plane of component 0 and reaches to the end of
            // the same plane of component 3; a component is a constant added to the lane's offset - one vector add instead of
            // a dozen scalar instructions per descriptor.  Measured SLOWER: headline 105.9-106.2 against 111.9-112.5 k
            // vector-steps/s on one box, complex forms -5 % (profiles/r04_onsite_ab.log): the scalar unit has room, the
            // vector unit has not.
            const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(buf, there ? (size_t)k * a.plane * RL * sizeof(double2) : 0, there ? span_bytes : 0);
#pragma unroll
            for (int al = 0; al < 4; ++al)
                out[al] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, wanted ? lane_bytes + al * (int)comp_stride : kBufferFar, 0, 0));
#else
            const int offset = wanted ? lane_bytes : kBufferOutOfRange;
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(buf, there ? ((size_t)al * nb + (size_t)k * a.plane) * RL * sizeof(double2) : 0,
                                                                     there ? plane_bytes : 0);
                out[al] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, offset, 0, 0));
            }
#endif
        } else if (wanted && k >= 0 && k < a.lx) {
            const size_t site = (size_t)k * a.plane + pw;
#pragma unroll
            for (int al = 0; al < 4; ++al)
                out[al] = nt ? load_stream(buf + vslot(al, site, r, nb, RL)) : buf[vslot(al, site, r, nb, RL)];
        } else {
#pragma unroll
            for (int al = 0; al < 4; ++al) out[al] = zero;
        }
    };
    auto store_plane = [&](double2* buf, int k, const double2 v[4]) {
        if (t.discard) return;  // (uniform: the last sweep of a run)
        const size_t site = (size_t)act(k) * a.plane + p;
        if (WT && !(a.stream & 16)) {  // (bit 4: plain stores all the same - measurements only)
            // write-through (sc1) stores: other workgroups of the same launch read these planes at the next level
            // (cheb_march3), so they must not linger in this XCD's L2
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(buf, ((size_t)al * nb + (size_t)act(k) * a.plane) * RL * sizeof(double2), plane_bytes);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[al]), rsrc, lane_bytes, 0, kAuxSc1);
            }
        } else if constexpr (BUF) {
#if BDG_SWEEP_ONE_DESCRIPTOR
            const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(buf, (size_t)act(k) * a.plane * RL * sizeof(double2), span_bytes);
#pragma unroll
            for (int al = 0; al < 4; ++al)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[al]), rsrc, lane_bytes + al * (int)comp_stride, 0, 0);
#else
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(buf, ((size_t)al * nb + (size_t)act(k) * a.plane) * RL * sizeof(double2), plane_bytes);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[al]), rsrc, lane_bytes, 0, 0);
            }
#endif
        } else {
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                if (nt_store) store_stream(buf + vslot(al, site, r, nb, RL), v[al]);
                else buf[vslot(al, site, r, nb, RL)] = v[al];
            }
        }
    };
    // plane k of t_n: read, or (GEN) made from the generator
    auto cur_plane = [&](int k, bool wanted, double2 out[4]) {
        if constexpr (GEN) {
            k = ring(act(k));
            if (wanted && k >= 0 && k < a.lx) {
                const uint64_t site = (uint64_t)k * a.plane + pw;  // (one hash per site and vector: kernels.hpp start_entry)
                const uint64_t h0 = start_site_hash(gen_key0, site);
                [[maybe_unused]] const uint64_t h1 = Mode::kVec == 2 ? start_site_hash(gen_key1, site) : 0;
#pragma unroll
                for (int al = 0; al < 4; ++al) {
                    if constexpr (Mode::kVec == 2) {
                        out[al].x = gen_on0 ? start_component(h0, al, 0).x : 0.0;
                        out[al].y = gen_on1 ? start_component(h1, al, 0).x : 0.0;
                    } else {
                        out[al] = gen_on0 ? start_component(h0, al, a.gen_kind) : zero;
                    }
                }
            } else {
#pragma unroll
                for (int al = 0; al < 4; ++al) out[al] = zero;
            }
        } else {
            load_plane(t.cur, nt_cur, k, wanted, out);
        }
    };
    auto load_ids = [&](int k, bool wanted = true) {
        uint2 w = make_uint2(0xFFFFFFFFu, 0xFFu);
        k = ring(act(k));
        if constexpr (BUF) {
            const bool there = k >= 0 && k < a.lx;  // (uniform)
            const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(a.stencil, there ? (size_t)k * a.plane * sizeof(uint2) : 0,
                                                                 there ? a.plane * (int)sizeof(uint2) : 0);
            const uint2 got = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, ok1 && wanted ? pw * (int)sizeof(uint2) : kBufferOutOfRange, 0, 0));
            if (ok1 && wanted && there) w = got;  // (a select: out-of-range loads return zeros, which would read "block 0")
        } else if (ok1 && wanted && k >= 0 && k < a.lx) {
            w = a.stencil[(size_t)k * a.plane + pw];
        }
        return w;
    };
    auto id_of = [](uint2 w, int slot) { return slot < 4 ? (w.x >> (8 * slot)) & 0xFFu : w.y & 0xFFu; };
    auto mac = [&](uint2 w, int slot, const double2 x[4], double2 acc[4]) {
        const unsigned id = id_of(w, slot);
        if (id == kNoBlock) return;
#if BDG_COMPACT_DIAG
        if ((w.y >> (8 + slot)) & 1u) Mode::mac_diag_compact(acc, diag + id * DSL, x);
#else
        if ((w.y >> (8 + slot)) & 1u) Mode::mac_diag(acc, lds + id * STRIDE, x);
#endif
        else if constexpr (Mode::kSingletSlots > 0 && BDG_SINGLET_BLOCKS) {
            if ((w.y >> (13 + slot)) & 1u) Mode::mac_singlet(acc, singlet + id * Mode::kSingletSlots, x);
            else Mode::mac_row(acc, lds + id * STRIDE, x);
        } else Mode::mac_row(acc, lds + id * STRIDE, x);
    };
    auto own_of = [&](const double2* row, double2 out[4]) {
#pragma unroll
        for (int be = 0; be < 4; ++be) out[be] = row[SHARE_SLOT(lane, be)];
    };
    auto put_own = [&](double2* row, const double2 v[4]) {
#pragma unroll
        for (int be = 0; be < 4; ++be) row[SHARE_SLOT(lane, be)] = v[be];
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // acc = Σ_offsets block * x in CSR order (-P, -1, 0, +1, +P); `before` / `after` are the planes
    // behind / ahead of the march, `mid` the lane's own entries, `row` the hand-over row of the level
    // step j (1-based) runs on plane k-j+1 and is needed on planes [x0-(steps-j), x1+(steps-j))
    const int k_first = x0 - (steps - 1), k_last = x1 + steps - 2;
    [[maybe_unused]] auto ring_entry = [&](int kk) { return os_ring + ((kk - k_first) % 3) * (RING / 3); };

    // (`kk` = marching index of the plane the step works on: selects the ring entry of its on-site blocks)
    auto apply = [&](uint2 w, const double2 before[4], const double2* row, const double2 mid[4],
                     const double2 after[4], double2 acc[4], [[maybe_unused]] int kk) {
        double2 x[4];
        [[maybe_unused]] const double2* rec = nullptr;
        if constexpr (OS != 0) rec = ring_entry(kk) + s * RSTRIDE;
        // one bond block times the neighbour's entries: from the table, or (OS = 2) from the site's record
        auto bond = [&](int slot, const double2 v[4]) {
            if constexpr (OS == 2) {
                if (id_of(w, slot) != kNoBlock) Mode::mac_bond(acc, rec + Mode::kOnsiteSlots + Mode::kBondSlots * (slot < 2 ? slot : slot - 1), v);
            } else {
                mac(w, slot, v, acc);
            }
        };
        if (rev) bond(0, after);
        else bond(0, before);
        if (id_of(w, 1) != kNoBlock) {
#pragma unroll
            for (int be = 0; be < 4; ++be) x[be] = row[SHARE_SLOT(lane - RL, be)];
            bond(1, x);
        }
        if constexpr (OS != 0) {
#if !(BDG_OS_EXPERIMENT & 4)
            if (id_of(w, 2) != kNoBlock) Mode::mac_onsite(acc, rec, mid);
#endif
        } else {
            mac(w, 2, mid, acc);
        }
        if (id_of(w, 3) != kNoBlock) {
#pragma unroll
            for (int be = 0; be < 4; ++be) x[be] = row[SHARE_SLOT(lane + RL, be)];
            bond(3, x);
        }
        if (rev) bond(4, before);
        else bond(4, after);
    };

    // OS: the wave's window of on-site records of one plane is contiguous in memory (SLOTS x
    // kOnsiteSlots 16-byte pieces; with periodic planes the halo slots wrap, so the address is
    // taken per piece): piece e of the window belongs to slot e / kOnsiteSlots.
    constexpr int OSL = OS != 0 ? (SLOTS * sweep3_onsite_pieces<Mode, OS>() + kWave - 1) / kWave : 1;  // pieces per lane
    [[maybe_unused]] auto load_onsite = [&](int k, bool wanted, double2 out[OSL]) {
        if constexpr (OS != 0) {
            constexpr int PIECES = sweep3_record_slots<Mode, OS>();
            k = ring(act(k));
#pragma unroll
            for (int j = 0; j < OSL; ++j) {
                const int e = lane + j * kWave, slot = e / PIECES, part = e - slot * PIECES;
                const int pe = col * OWNED3 - 3 + slot;
                const bool in_e = pe >= 0 && pe < a.plane;
                const int pwe = in_e ? pe : ((pe % a.plane) + a.plane) % a.plane;
                out[j] = zero;
                // (plain loads: the halo slots of the neighbouring windows read the same records - 16 slots
                // per 10 owned - and should find them in L2; a.stream bit 3 = non-temporal, for A/B runs)
                {
                    const bool there = wanted && k >= 0 && k < a.lx;  // (uniform)
                    const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(a.onsite, there ? (size_t)k * a.plane * PIECES * sizeof(double2) : 0,
                                                                         there ? a.plane * PIECES * (int)sizeof(double2) : 0);
                    const int piece_bytes = slot < SLOTS && (in_e || a.wrap_p) ? (pwe * PIECES + part) * (int)sizeof(double2) : kBufferOutOfRange;
                    out[j] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, piece_bytes, 0, 0));
                }
            }
        }
    };
    [[maybe_unused]] auto put_onsite = [&](int kk, const double2 v[OSL]) {
        if constexpr (OS != 0) {
            constexpr int PIECES = sweep3_record_slots<Mode, OS>();
            double2* dst = ring_entry(kk);
#pragma unroll
            for (int j = 0; j < OSL; ++j) {
                const int e = lane + j * kWave, slot = e / PIECES, part = e - slot * PIECES;
                if (slot < SLOTS) dst[slot * RSTRIDE + part] = v[j];
            }
        }
    };

    // ---- prologue
    // Rolling state.  t_n: `cn_m` = plane k-1; two buffers hold planes k and k+1 and swap roles
    // every iteration (the loop is unrolled by two): the centre plane goes to its LDS row at the
    // top of the iteration, which frees its registers for the load of plane k+2 - issued there and
    // first used a whole iteration later, with no register-to-register hand-over in between.
    // t_{n-1} of plane k+1 is loaded into `pv` as soon as step 1 has consumed plane k's.
    double2 cn_m[4], buf_a[4], buf_b[4], pv[4], c1_m[4], c2_m[4];
    cur_plane(k_first - 1, valid, cn_m);
    cur_plane(k_first, valid, buf_a);
    cur_plane(k_first + 1, valid, buf_b);
    load_plane(t.prev, nt_prev, k_first, !GEN && ok1 && t.prev != nullptr, pv);  // (GEN: t_{-1} = 0)
    uint2 ids_0 = load_ids(k_first), ids_1 = make_uint2(0xFFFFFFFFu, 0xFFu), ids_2 = ids_1;
#pragma unroll
    for (int al = 0; al < 4; ++al) c1_m[al] = c2_m[al] = zero;
    put_own(row_1, c1_m);  // the rows still hold the previous unit's planes
    put_own(row_2, c1_m);
    if constexpr (OS != 0) {  // on-site blocks of the first plane (the later ones arrive one iteration ahead)
        double2 first_os[OSL];
        load_onsite(k_first, true, first_os);
        put_onsite(k_first, first_os);
    }

    // one iteration: `centre` holds plane k on entry and plane k+2 (in flight) on exit, `after` plane k+1
    auto iterate = [&](int k, double2 (&centre)[4], const double2 (&after)[4]) {
        const bool more = k < k_last;
        const uint2 nx_ids = load_ids(k + 1, more);
        put_own(row_0, centre);
        wave_sync();
        cur_plane(k + 2, valid && more, centre);

        // ---- step 1 on plane k: level 1 = c1 H t_n - t_{n-1}
        double2 new1[4], new2[4];
#pragma unroll
        for (int al = 0; al < 4; ++al) new1[al] = new2[al] = zero;
        if (ok1 && in_lattice(k)) {
            double2 acc[4], mid[4];
#pragma unroll
            for (int al = 0; al < 4; ++al) acc[al] = zero;
            own_of(row_0, mid);
            apply(ids_0, cn_m, row_0, mid, after, acc, k);
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                new1[al].x = fma(t.coef1, acc[al].x, -pv[al].x);
                new1[al].y = fma(t.coef1, acc[al].y, -pv[al].y);
            }
            if (owned && k >= x0 && k < x1) {
#pragma unroll
                for (int al = 0; al < 4; ++al) Mode::dots(dot1, mid[al], new1[al]);
                if (steps == 1) store_plane(t.out2, k, new1);
                if (steps == 2) store_plane(t.out1, k, new1);
            }
        }
        load_plane(t.prev, nt_prev, k + 1, !GEN && ok1 && more && t.prev != nullptr, pv);

        // ---- step 2 on plane k-1: level 2 = c2 H level1 - t_n        (row_1 = level 1, plane k-1)
        if (steps >= 2 && ok2 && in_lattice(k - 1) && k - 1 >= x0 - (steps - 2) && k - 1 < x1 + (steps - 2)) {
            double2 acc[4], mid[4];
#pragma unroll
            for (int al = 0; al < 4; ++al) acc[al] = zero;
            own_of(row_1, mid);
            apply(ids_1, c1_m, row_1, mid, new1, acc, k - 1);
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                new2[al].x = fma(a.coef2, acc[al].x, -cn_m[al].x);
                new2[al].y = fma(a.coef2, acc[al].y, -cn_m[al].y);
            }
            if (owned && k - 1 >= x0 && k - 1 < x1) {
#pragma unroll
                for (int al = 0; al < 4; ++al) Mode::dots(dot2, mid[al], new2[al]);
                if (steps == 2) store_plane(t.out2, k - 1, new2);
                if (steps == 3) store_plane(t.out1, k - 1, new2);
            }
        }

        // (OS) on-site records of plane k+1: asked for here rather than at the top of the iteration - they are
        // not needed before its end, and 4-8 registers held across steps 1 and 2 are 4-8 registers spilled
        [[maybe_unused]] double2 nx_os[OSL];
#if BDG_OS_EXPERIMENT & 1  // (timing builds: what the record loads / the ring writes / the on-site product cost; results wrong)
        load_onsite(k + 1, false, nx_os);
#else
        load_onsite(k + 1, more, nx_os);
#endif

        // ---- step 3 on plane k-2: level 3 = c2 H level2 - level1       (row_2 = level 2, plane k-2)
        if (steps >= 3 && owned && k - 2 >= x0 && k - 2 < x1) {
            double2 acc[4], mid[4], new3[4];
#pragma unroll
            for (int al = 0; al < 4; ++al) acc[al] = zero;
            own_of(row_2, mid);
            apply(ids_2, c2_m, row_2, mid, new2, acc, k - 2);
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                new3[al].x = fma(a.coef2, acc[al].x, -c1_m[al].x);
                new3[al].y = fma(a.coef2, acc[al].y, -c1_m[al].y);
                Mode::dots(dot3, mid[al], new3[al]);
            }
            store_plane(t.out2, k - 2, new3);
        }

        // ---- roll: every level moves one plane on
        wave_sync();
        own_of(row_0, cn_m);   // level 0, plane k
        own_of(row_1, c1_m);   // level 1, plane k-1
        own_of(row_2, c2_m);   // level 2, plane k-2
        wave_sync();
        put_own(row_1, new1);  // level 1, plane k
        put_own(row_2, new2);  // level 2, plane k-1
#if !(BDG_OS_EXPERIMENT & 2)
        put_onsite(k + 1, nx_os);  // (OS) takes the ring entry of plane k-2, which step 3 has just finished with
#endif
        ids_2 = ids_1;
        ids_1 = ids_0;
        ids_0 = nx_ids;
    };
    for (int k = k_first; k <= k_last; k += 2) {
        iterate(k, buf_a, buf_b);
        if (k + 1 <= k_last) iterate(k + 1, buf_b, buf_a);
    }
}

template <typename Mode, int OS, int THREADS = kBlockThreads>
__device__ inline Sweep3Lds sweep3_stage_lds(double2* lds, const SweepArgs& a, int slots, int wave) {
    constexpr int SPB = Mode::kSlotsPerBlock;
    constexpr int STRIDE = Mode::kBlockStride;
    constexpr int DSL = Mode::kDiagSlots;
    const int ring = sweep3_ring_slots<Mode, OS>(slots);
    // LDS: [table][compact diagonals][per wave: three rows of 64 lanes x 4 entries; OS: + ring of three planes of on-site blocks]
    const double2* table = static_cast<const double2*>(a.dict_table);
    for (int e = threadIdx.x; e < a.n_unique * SPB; e += THREADS)
        lds[(e / SPB) * STRIDE + (e % SPB)] = table[e];
    // compact copies of the block diagonals behind the table: a block flagged diagonal (plain hopping: four of the
    // five blocks of a row in the s-wave models) then costs Mode::kDiagSlots 16-byte LDS reads instead of mac_diag's
    // (1 instead of 2 in real particle-hole arithmetic: a tenth of the kernel's LDS operations)
    constexpr int SSL = Mode::kSingletSlots;
    Sweep3Lds w;
    w.table = lds;
    w.diag = lds + a.n_unique * STRIDE;
    w.singlet = w.diag + a.n_unique * DSL;
    w.row_0 = w.singlet + a.n_unique * SSL + wave * (3 * kWave * 4 + ring);
    w.row_1 = w.row_0 + kWave * 4;
    w.row_2 = w.row_1 + kWave * 4;
    w.os_ring = w.row_2 + kWave * 4;
    __syncthreads();
    for (int id = threadIdx.x; id < a.n_unique; id += THREADS) {
        Mode::pack_diag(w.diag + id * DSL, lds + id * STRIDE);
        if constexpr (SSL > 0) Mode::pack_singlet(w.singlet + id * SSL, lds + id * STRIDE);
    }
    __syncthreads();
    return w;
}

// WAVES: waves per workgroup.  4 with two workgroups per CU is the default; the streamed forms whose ring of records
// leaves room for fewer than eight waves on a CU run ONE workgroup of 5..7 waves per CU instead (160 KB of LDS: seven
// waves of 12 KB rows + a ring of up to 10.5 KB), which keeps 2 lanes per site (32-slot windows) within reach.
template <typename Mode, int RL, bool REV, bool GEN = false, int OS = 0, int WAVES = kWavesPerBlock>
__global__ __launch_bounds__(WAVES * kWave, WAVES > 4 ? 1 : 2) void cheb_sweep3(SweepArgs a) {
    extern __shared__ double2 lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const Sweep3Lds w = sweep3_stage_lds<Mode, OS, WAVES * kWave>(lds, a, kWave / RL, wave);

    const int n_units = a.n_cols * a.n_segs;
    const int xcd = blockIdx.x & 7;
    const int u_lo = (int)(((int64_t)n_units * xcd) >> 3);
    const int u_hi = (int)(((int64_t)n_units * (xcd + 1)) >> 3);
    const int waves_per_xcd = (gridDim.x >> 3) * WAVES;

    double dot1[4] = {0.0, 0.0, 0.0, 0.0}, dot2[4] = {0.0, 0.0, 0.0, 0.0}, dot3[4] = {0.0, 0.0, 0.0, 0.0};
    Sweep3Gen gen;
    if constexpr (GEN) gen = sweep3_gen_keys<Mode>(a.gen_seed, a.gen_first_id, a.gen_active, lane % RL);

    const Sweep3Task task{a.cur, a.prev, a.out1, a.out2, a.coef1, a.steps, a.discard};
    for (int u = u_lo + (int)(blockIdx.x >> 3) * WAVES + wave; u < u_hi; u += waves_per_xcd) {
        const int seg = u / a.n_cols, col = u - seg * a.n_cols;
        const bool rev = REV != (bool)(a.zigzag & seg & 1);  // wave-uniform
        sweep3_unit<Mode, RL, GEN, OS, false>(a, task, w, gen, lane, seg, col, rev, dot1, dot2, dot3);
    }

    const int steps = a.steps;
    __syncthreads();
    sweep_reduce_dots<Mode, RL, WAVES>(dot1, reinterpret_cast<double*>(lds), a.partial1, lane, wave);
    if (steps >= 2) {
        __syncthreads();
        sweep_reduce_dots<Mode, RL, WAVES>(dot2, reinterpret_cast<double*>(lds), a.partial2, lane, wave);
    }
    if (steps >= 3) {
        __syncthreads();
        sweep_reduce_dots<Mode, RL, WAVES>(dot3, reinterpret_cast<double*>(lds), a.partial3, lane, wave);
    }
}

// =====================================================================================
// K7c  cheb_march3 - the three-step sweeps of a whole reduction chunk (up to 21 of them = 63 steps) in ONE launch.
//
// A launch of cheb_sweep3 is one round: every wave starts at the same moment, loads its first planes while nothing
// is written, marches, and drains while nothing is read; the next sweep cannot start before the slowest wave of the
// last one has finished and the launch has been turned around.  But sweep l+1 of a unit (segment s, window c) needs
// only the planes the nine units (s-1..s+1, c-1..c+1) wrote in sweep l (three recomputed planes / positions per
// side), and overwrites only what those nine read in sweep l.  So here the sweeps ("levels") of a chunk are tasks
// (level, lane group, unit) that the waves of one launch claim from ticket counters and run as soon as the nine
// units of the level before have published theirs:
//   * tickets (default; MarchArgs.fixed = the static assignment of cheb_sweep3 instead): per level and XCD one counter; workgroups b, b+8, ... (one XCD under round-robin placement - speed
//     only) claim units of their eighth of a level first, then whatever another eighth has left, and move on to the
//     next level only when all eight counters of the level are exhausted.  A task of level l is therefore claimed
//     only after EVERY task of level l-1 has been claimed - by a wave that is running - so whatever a task waits
//     for is finished or being worked on: no co-residency of the grid is assumed, and nothing can deadlock.
//   * hand-over (guide: G16 R1): the new planes are stored write-through (sc1 buffer stores); after its march the
//     wave waits for its stores (s_waitcnt vmcnt(0)) and lane 0 stores the unit's flag = levels done (relaxed,
//     agent scope).  A consumer polls the nine flags with ONE wave instruction (lane i loads flag i, sc1) and
//     s_sleep between polls, then makes one agent-scope acquire (buffer_inv sc1: this CU's L1) and only then loads.
//   * a bounded wait: a wave that has polled one task's flags for `timeout_ticks` of the 100 MHz clock raises the abort word, every
//     wave leaves at its next poll or claim, and the host falls back to one launch per sweep.
// Two lane groups of a call are tasks of the same launch (their units interleaved in ticket order): what two
// streams did for cheb_sweep3 - one group's waves filling the other's idle ends - now happens between any two
// tasks.  The arithmetic is sweep3_unit's: t's are the bits cheb_sweep3 makes; dot partials are kept per unit
// ([step][unit]) and summed in unit order by reduce_partials - independent of which wave ran which task.
constexpr int kMarchGroups = 4;       // lane groups one launch advances side by side
constexpr int kMarchMaxLevels = 24;   // sweeps per launch (a 63-step reduction chunk: 21)
constexpr int kMarchCounterWords = 32;  // every polled word on a 128-byte line of its own

struct MarchGroup {
    double2* buf[4];       // even levels read buf[0] (t_n), buf[1] (t_{n-1}) and write buf[2], buf[3]; odd levels the other way round
    double* partial;       // [steps of the launch][units][RL * kVec][2]
    unsigned long long gen_first_id;
    int gen_active;
    int pad;
};

struct MarchArgs {
    SweepArgs base;        // geometry, tables, coef2, generator seed / kind, zigzag, rings; buffers and partials come from the groups
    MarchGroup group[kMarchGroups];
    int n_groups;
    int n_levels;
    int last_steps;        // steps of the last level (1..3); every other level makes 3
    int discard_last;      // the last level stores nothing (end of a run)
    int first_is_start;    // level 0 is the first sweep of a run: t_{-1} = 0 is not read and coef1 = coef2 / 2
    int gen;               // ... and makes the random t_0 itself (GEN)
    int rev0;              // marching direction of level 0 (levels alternate)
    int units;             // n_cols * n_segs
    unsigned long long per_step;   // doubles between the partials of consecutive steps
    unsigned* sync;        // word 0: abort; counters at kMarchCounterWords * (1 + 8 * level + xcd); flags from flags_at on, [group][unit]
    unsigned flags_at;
    unsigned timeout_ticks;
    unsigned* gave_up;     // set (never cleared by a launch) when a wave gives up waiting: the host reads it with the results
    // 1 = no tickets: wave i of an XCD's workgroups takes units i, i + waves, ... of its eighth at every level, as
    // cheb_sweep3's waves do (neighbouring windows on the waves of one workgroup: shared halo lines in one L1).  Needs the
    // whole grid resident - a wave that waits for a unit whose wave never started gives up after the timeout and the
    // host falls back.  Launches of a single level always run this way (nothing is waited for).
    int fixed;
    // measurements and tests only (BODGE_AMD_MARCH_DEBUG): bit 0 = no acquire after a wait, bit 1 = no wait at all (WRONG
    // results: what the hand-over costs), bit 2 = plain instead of write-through stores, bit 3 = add every wave's waiting
    // and claiming time (100 MHz ticks) and task count to gave_up[1..3], bit 5 = the first wave that has to wait gives up
    // at once (the fallback path under test)
    int debug;
    int poll_sleep;        // s_sleep argument between two polls (units of 64 clocks)
};

__host__ __device__ inline size_t march_sync_words(int units, int groups) {
    return (size_t)kMarchCounterWords * (1 + 8 * kMarchMaxLevels) + (size_t)groups * units;
}

template <typename Mode, int RL, int OS>
__global__ __launch_bounds__(kBlockThreads, 2) void cheb_march3(MarchArgs m) {
    extern __shared__ double2 lds[];
    constexpr int W = 2 * Mode::kVec;  // doubles per lane of one step's dot products
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const Sweep3Lds w = sweep3_stage_lds<Mode, OS>(lds, m.base, kWave / RL, wave);

    const int n_units = m.units, n_cols = m.base.n_cols, n_segs = m.base.n_segs;
    const int home = blockIdx.x & 7;
    auto first_of = [&](int x) { return (int)(((int64_t)n_units * x) >> 3); };
    auto share_of = [&](int x) { return (unsigned)((first_of(x + 1) - first_of(x)) * m.n_groups); };
    auto counter = [&](int level, int x) { return m.sync + (size_t)kMarchCounterWords * (1 + 8 * level + x); };
    unsigned* const flags = m.sync + m.flags_at;
    unsigned long long t_waiting = 0, t_claiming = 0;
    unsigned n_tasks = 0;
    // A launch of ONE level has nothing to wait for: its waves take the units of their eighth in turn, as cheb_sweep3's do
    // (no tickets); m.fixed asks for the same at every level.
    const bool fixed = m.n_levels == 1 || m.fixed;
    const unsigned my_first = (unsigned)((blockIdx.x >> 3) * kWavesPerBlock + wave), my_stride = (unsigned)((gridDim.x >> 3) * kWavesPerBlock);
    unsigned my_next = my_first;
    for (int level = 0; level < m.n_levels;) {
        const unsigned long long t_claim = (m.debug & 8) ? wall_clock64() : 0;
        int queue = home;
        unsigned idx = 0;
        if (fixed) {
            if (my_next >= share_of(home)) {
                ++level;
                my_next = my_first;
                continue;
            }
            idx = my_next;
            my_next += my_stride;
        } else {
            // ---- claim: own eighth of the level first, then what another eighth has left, then the next level
            if (lane == 0) idx = __hip_atomic_fetch_add(counter(level, home), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx >= share_of(home)) {
                unsigned seen = 0xFFFFFFFFu;
                if (lane < 8) seen = __hip_atomic_load(counter(level, lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned stop = __hip_atomic_load(m.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(stop) != 0) break;
                const unsigned long long left = __ballot(lane < 8 && seen < share_of(lane < 8 ? lane : 0));
                if (left == 0) {
                    ++level;
                    continue;
                }
                // the nearest eighth after the own one that still has tasks
                const unsigned rot = (unsigned)(((left | (left << 8)) >> home) & 0xFFu);
                queue = (home + __builtin_ctz(rot)) & 7;
                if (lane == 0) idx = __hip_atomic_fetch_add(counter(level, queue), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                idx = __builtin_amdgcn_readfirstlane(idx);
                if (idx >= share_of(queue)) continue;  // (taken meanwhile: look again)
            }
        }
        const int g = (int)(idx % (unsigned)m.n_groups);
        const int u = first_of(queue) + (int)(idx / (unsigned)m.n_groups);
        const int seg = u / n_cols, col = u - seg * n_cols;

        if (m.debug & 8) t_claiming += wall_clock64() - t_claim;
        ++n_tasks;
        // ---- wait until the nine units around this one have published the level before
        if (level > 0 && !(m.debug & 2)) {
            const int ds = lane / 3 - 1, dc = lane % 3 - 1;
            int ns = seg + ds, nc = col + dc;
            if (m.base.wrap_x) ns = ns < 0 ? ns + n_segs : (ns >= n_segs ? ns - n_segs : ns);
            if (m.base.wrap_p) nc = nc < 0 ? nc + n_cols : (nc >= n_cols ? nc - n_cols : nc);
            const bool watch = lane < 9 && ns >= 0 && ns < n_segs && nc >= 0 && nc < n_cols;
            const unsigned* mine = watch ? flags + (size_t)g * n_units + (size_t)ns * n_cols + nc : m.sync;
            bool stop = false;
            const unsigned long long t_wait = wall_clock64();
            for (;;) {
                const unsigned v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (lanes that watch no unit read the abort word: zero unless the launch is being given up)
                if (__any(!watch && v != 0)) {
                    stop = true;
                    break;
                }
                if (__all(!watch || v >= (unsigned)level)) break;
                if (wall_clock64() - t_wait > (unsigned long long)m.timeout_ticks || (m.debug & 32)) {
                    if (lane == 0) {
                        __hip_atomic_store(m.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(m.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    stop = true;
                    break;
                }
                if (m.poll_sleep <= 4) __builtin_amdgcn_s_sleep(4);
                else if (m.poll_sleep <= 16) __builtin_amdgcn_s_sleep(16);
                else __builtin_amdgcn_s_sleep(64);
            }
            if (stop) break;
            if (m.debug & 8) t_waiting += wall_clock64() - t_wait;
            if (!(m.debug & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }

        // ---- the sweep of this unit
        const MarchGroup& grp = m.group[g];
        const bool odd = level & 1;
        const bool start = m.first_is_start && level == 0;
        const bool last = level == m.n_levels - 1;
        Sweep3Task task;
        task.cur = grp.buf[odd ? 3 : 0];
        task.prev = start ? nullptr : grp.buf[odd ? 2 : 1];
        task.out1 = grp.buf[odd ? 1 : 2];
        task.out2 = grp.buf[odd ? 0 : 3];
        task.coef1 = start ? 0.5 * m.base.coef2 : m.base.coef2;
        task.steps = last ? m.last_steps : 3;
        task.discard = last ? m.discard_last : 0;
        const bool rev = (bool)((m.rev0 + level) & 1) != (bool)(m.base.zigzag & seg & 1);
        double dot1[4] = {0.0, 0.0, 0.0, 0.0}, dot2[4] = {0.0, 0.0, 0.0, 0.0}, dot3[4] = {0.0, 0.0, 0.0, 0.0};
#ifndef BDG_MARCH_GEN
#define BDG_MARCH_GEN 1  // (0: the chunk kernel carries one unit body, runs need BODGE_AMD_SWEEP_GEN=0 - A/B builds)
#endif
        if (BDG_MARCH_GEN && start && m.gen) {
            const Sweep3Gen gen = sweep3_gen_keys<Mode>(m.base.gen_seed, grp.gen_first_id, grp.gen_active, lane % RL);
            sweep3_unit<Mode, RL, true, OS, true>(m.base, task, w, gen, lane, seg, col, rev, dot1, dot2, dot3);
        } else {
            sweep3_unit<Mode, RL, false, OS, true>(m.base, task, w, Sweep3Gen{}, lane, seg, col, rev, dot1, dot2, dot3);
        }

        // ---- publish: stores drained, then the flag
        if (m.n_levels > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(flags + (size_t)g * n_units + u, (unsigned)(level + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }

        // ---- dot products of the unit: over the site lanes, one partial per (step, unit)
        auto put_dots = [&](double (&dot)[4], int step) {
#pragma unroll
            for (int off = kWave / 2; off >= RL; off >>= 1)
#pragma unroll
                for (int c = 0; c < W; ++c) dot[c] += __shfl_xor(dot[c], off);
            if (lane < RL) {
                double* out = grp.partial + (size_t)(3 * level + step) * m.per_step + ((size_t)u * RL + lane) * W;
#pragma unroll
                for (int c = 0; c < W; ++c) out[c] = dot[c];
            }
        };
        put_dots(dot1, 0);
        if (task.steps >= 2) put_dots(dot2, 1);
        if (task.steps >= 3) put_dots(dot3, 2);
    }
    if ((m.debug & 8) && lane == 0) {
        atomicAdd(m.gave_up + 1, (unsigned)(t_waiting >> 4));
        atomicAdd(m.gave_up + 2, (unsigned)(t_claiming >> 4));
        atomicAdd(m.gave_up + 3, n_tasks);
    }
}

// =====================================================================================
// K8  cheb_roll3 - one recurrence step on a 3-D lattice stencil with the x-neighbours in registers.
//
// The one-step kernels gather all six neighbours of a site from L2.  On a 100^3 lattice the two
// x-neighbours are a whole plane (10^4 rows x 256 B per 8 real vectors = 2.5 MB) away in either
// direction, so by the time a row's +x neighbour is needed its line has often left the XCD's 4 MB
// L2: 1.15 x the algorithmic HBM traffic (profiles/r01: pmc3d).  Here a wave owns 14 consecutive
// in-plane positions p = z + Lz*y and marches along x through one segment of planes, exactly as
// cheb_sweep does, keeping t_n of planes k-1, k, k+1 of its own positions in registers: every
// row of t_n is read from HBM once (plus one ghost slot at either end of the window and one extra
// plane at either end of a segment).  The +-1 (z) neighbours come from the neighbouring lanes
// through a wave-private LDS row, the +-Lz (y) neighbours are gathered from L2, where the waves
// 7 windows away - same workgroup or the next - have just put them.
// Block (i, j) may be stored for j - i in {-P, -Lz, -1, 0, +1, +Lz, +P} with the z and y
// neighbours inside the lattice (open boundaries); anything else runs the one-step kernels.
// Two steps per sweep are not possible here: step two would need t_{n+1} of the y-neighbours,
// which other waves make.

struct RollArgs {
    const uint2* stencil;    // per block row: table ids at offsets -P, -Lz, -1, 0, +1, +Lz, +P (bytes 0..6)
    const void* dict_table;
    int n_unique;
    const double2* cur;      // t_n
    double2* prev;           // t_{n-1} in, t_{n+1} out (same lane reads and writes a row: in place)
    double* partial;         // [gridDim.x][RL * kVec][2]
    double coef;
    int nb, plane, lz, lx;
    int n_cols;              // windows per plane = ceil(P / 14)
    int n_segs;
    int stream;              // non-temporal hints: bit 0 t_{n-1} loads, bit 1 stores
    int reverse;             // 1 = march every segment from its far end (launches alternate)
    int discard;             // 1 = t_{n+1} is not stored (last step of a run: only its dot products are wanted)
    int x_lo, x_hi;          // planes advanced by this launch (0, lx = all; unit start vectors: the band that can be non-zero)
    // > 0: the units of the launch are runs of `chunk` plane-steps of the sequence (window 0: planes x_lo .. x_hi - 1, window 1:
    // ..., column-major), cut wherever a run ends - mid-column too - so that every wave slot of the device gets the same
    // number of planes (100^3: 715 windows x 100 planes on 2048 slots are 35 planes each; whole-column segments give 1430 units
    // of 50 planes - 70 % of the slots - or 2145 of 33).  A unit that crosses a column boundary marches two pieces.
    int chunk;
    // Row slabs (a stack of whole x-planes of a larger lattice, one handle per slab): the vector buffers
    // have `ld` block rows per component plane (own rows + halo rows), and t_n of the plane below plane 0 /
    // above plane lx-1 is read where the neighbouring slab keeps it - `lo_buf` / `hi_buf` point into that
    // slab's own t_n buffer (same process: no copy, no pack / unpack; block row lo_site0 + p of a buffer
    // with lo_ld rows per component plane).  nullptr = no neighbour on that side (the lattice ends there).
    int ld;
    const double2* lo_buf;
    const double2* hi_buf;
    int lo_site0, lo_ld, hi_site0, hi_ld;
};

// Row slabs: columns >= nb are halo rows.  The only ones a stencil may reference are the plane below
// the slab's first plane (local column lo_base + p, from rows of plane 0) and the plane above its last
// (hi_base + p, from rows of plane lx-1); lo_base / hi_base = -1 where the slab has no such neighbour.
__global__ void build_stencil3(const int* __restrict__ indptr, const int* __restrict__ words,
                               const int* __restrict__ diagonal, int nb, int plane, int lz, int lo_base, int hi_base,
                               uint2* __restrict__ stencil, int* __restrict__ bad) {
    const int lx = nb / plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        unsigned id[8] = {kNoBlock, kNoBlock, kNoBlock, kNoBlock, kNoBlock, kNoBlock, kNoBlock, 0u};
        const int p = i % plane, z = p % lz, y = p / lz, ly = plane / lz, x = i / plane;
        bool ok = true;
        for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
            const unsigned w = (unsigned)words[k];
            const int col = (int)(w & 0xFFFFFFu);
            const int off = col - i;
            int slot = -1;
            if (col >= nb) {
                if (x == 0 && lo_base >= 0 && col == lo_base + p) slot = 0;
                else if (x == lx - 1 && hi_base >= 0 && col == hi_base + p) slot = 6;
            } else if (off == -plane) slot = 0;
            else if (off == -lz && y >= 1) slot = 1;
            else if (off == -1 && z >= 1) slot = 2;
            else if (off == 0) slot = 3;
            else if (off == 1 && z <= lz - 2) slot = 4;
            else if (off == lz && y <= ly - 2) slot = 5;
            else if (off == plane) slot = 6;
            if (slot < 0 || (w >> 24) == kNoBlock) ok = false;
            else {
                id[slot] = w >> 24;
                if (diagonal[w >> 24] == 1) id[7] |= 1u << slot;  // byte 7: which of the seven blocks are diagonal
            }
        }
        if (!ok) atomicOr(bad, 1);
        stencil[i] = make_uint2(id[0] | (id[1] << 8) | (id[2] << 16) | (id[3] << 24), id[4] | (id[5] << 8) | (id[6] << 16) | (id[7] << 24));
    }
}

// RL = lanes per site: 4 (8 real vectors per launch, 14 owned positions per window) or 2 (4 real
// vectors, 30 owned positions: half the ghost slots, and half the bytes an XCD touches per plane).
constexpr int roll_owned(int rl) { return kWave / rl - 2; }

// (three waves per SIMD - 168 VGPRs - would fill the 2048 x 1.5 wave slots of a 100^3 launch better, but the
// seven planes of rolling state and a plane of prefetch then spill 250-360 B per lane: round 3, not kept)
// NT: non-temporal t_{n-1} loads and t_{n+1} stores (RollArgs.stream = 3: vector pairs beyond the Infinity Cache) - a
// template parameter since round 4, because every global access is a raw buffer instruction whose cache policy is an
// immediate, and because a run-time choice between two loads is a branch: see "Buffer addressing" above - loads behind
// branches make the compiler wait with vmcnt(0), i.e. for the planes it has just asked for.
template <typename Mode, int RL = kSweepLanes, bool NT = false>
__global__ __launch_bounds__(kBlockThreads, 2) void cheb_roll3(RollArgs a) {
    extern __shared__ double2 lds[];
    constexpr int SLOTS = kWave / RL;
    constexpr int OWNED = SLOTS - 2;
    constexpr int SPB = Mode::kSlotsPerBlock;
    constexpr int STRIDE = Mode::kBlockStride;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int s = lane / RL;
    const int r = lane % RL;

    const double2* table = static_cast<const double2*>(a.dict_table);
    for (int e = threadIdx.x; e < a.n_unique * SPB; e += kBlockThreads)
        lds[(e / SPB) * STRIDE + (e % SPB)] = table[e];
    double2* row_n = lds + a.n_unique * STRIDE + wave * (kWave * 4);  // t_n of plane k, all 16 slots
    __syncthreads();

    const int span = a.x_hi - a.x_lo;
    const int64_t plane_steps = (int64_t)a.n_cols * span;
    const int n_units = a.chunk > 0 ? (int)((plane_steps + a.chunk - 1) / a.chunk) : a.n_cols * a.n_segs;
    const int xcd = blockIdx.x & 7;
    const int u_lo = (int)(((int64_t)n_units * xcd) >> 3);
    const int u_hi = (int)(((int64_t)n_units * (xcd + 1)) >> 3);
    const int waves_per_xcd = (gridDim.x >> 3) * kWavesPerBlock;

    double dot[4] = {0.0, 0.0, 0.0, 0.0};
    const double2 zero = make_double2(0.0, 0.0);
    const size_t nb = (size_t)a.ld;  // block rows per component plane of this handle's vector buffers
    constexpr int kAuxVec = NT ? kAuxNt : 0;
    const int plane_bytes = a.plane * RL * (int)sizeof(double2);
    const int ly = a.plane / a.lz;
    for (int u = u_lo + (int)(blockIdx.x >> 3) * kWavesPerBlock + wave; u < u_hi; u += waves_per_xcd) {
      // the pieces of the unit: one (whole-column segments), or the parts of its run of plane-steps in up to two columns
      int64_t g0 = a.chunk > 0 ? (int64_t)u * a.chunk : 0;
      const int64_t g1 = a.chunk > 0 ? (g0 + a.chunk < plane_steps ? g0 + a.chunk : plane_steps) : 1;
      while (g0 < g1) {
        int col, x0, x1;
        if (a.chunk > 0) {
            col = (int)(g0 / span);
            const int xa = (int)(g0 - (int64_t)col * span);
            const int xb = (int)(xa + (g1 - g0) < span ? xa + (g1 - g0) : span);
            x0 = a.x_lo + xa, x1 = a.x_lo + xb;
            g0 += xb - xa;
        } else {
            const int seg = u / a.n_cols;
            col = u - seg * a.n_cols;
            x0 = a.x_lo + (int)(((int64_t)span * seg) / a.n_segs);
            x1 = a.x_lo + (int)(((int64_t)span * (seg + 1)) / a.n_segs);
            g0 = g1;
        }
        const int p = col * OWNED - 1 + s;
        const bool valid = p >= 0 && p < a.plane;
        const bool owned = valid && s >= 1 && s <= SLOTS - 2;
        const bool rev = a.reverse != 0;  // wave-uniform
        auto act = [&](int k) { return rev ? x0 + x1 - 1 - k : k; };
        // the y-neighbours are asked for wherever the lattice has them (a bond that is not stored costs a load whose
        // result nobody uses): deciding by the stencil word would make these loads wait for the word's
        const int y = valid ? p / a.lz : 0;
        const bool has_ym = owned && y >= 1, has_yp = owned && y <= ly - 2;

        // Every global access: a raw buffer instruction on the descriptor of ONE plane (base made from wave-uniform values),
        // per-lane offset (position * RL + r) * 16 or an offset out of range - no branch around any load (cheb_sweep3).
        auto plane_of = [&](const double2* buf, size_t ld, size_t row0, bool there, int offset, auto aux, double2 out[4]) {
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(buf, there ? ((size_t)al * ld + row0) * RL * sizeof(double2) : 0,
                                                                     there ? plane_bytes : 0);
                out[al] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, offset, 0, decltype(aux)::value));
            }
        };
        auto load_plane = [&](const double2* buf, auto aux, int k, int shift, bool wanted, double2 out[4]) {
            k = act(k);
            plane_of(buf, nb, (size_t)k * a.plane, k >= 0 && k < a.lx, wanted ? ((p + shift) * RL + r) * (int)sizeof(double2) : kBufferOutOfRange, aux, out);
        };
        // t_n of the lane's own position in plane k; planes -1 and lx are the neighbouring slabs' (if any)
        auto load_cur = [&](int k, bool wanted, double2 out[4]) {
            k = act(k);
            const double2* buf = a.cur;
            size_t row0 = (size_t)k * a.plane, ld = nb;
            bool there = k >= 0 && k < a.lx;
            if (k == -1 && a.lo_buf) buf = a.lo_buf, row0 = (size_t)a.lo_site0, ld = (size_t)a.lo_ld, there = true;
            if (k == a.lx && a.hi_buf) buf = a.hi_buf, row0 = (size_t)a.hi_site0, ld = (size_t)a.hi_ld, there = true;
            plane_of(buf, ld, row0, there, wanted ? (p * RL + r) * (int)sizeof(double2) : kBufferOutOfRange, std::integral_constant<int, 0>{}, out);
        };
        auto load_ids = [&](int k, bool wanted) {
            uint2 w = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            k = act(k);
            const bool there = k >= 0 && k < a.lx;
            const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(a.stencil, there ? (size_t)k * a.plane * sizeof(uint2) : 0, there ? a.plane * (int)sizeof(uint2) : 0);
            const uint2 got = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, owned && wanted ? p * (int)sizeof(uint2) : kBufferOutOfRange, 0, 0));
            if (owned && wanted && there) w = got;
            return w;
        };
        auto id_of = [](uint2 w, int slot) { return slot < 4 ? (w.x >> (8 * slot)) & 0xFFu : (w.y >> (8 * (slot - 4))) & 0xFFu; };
        constexpr std::integral_constant<int, kAuxVec> aux_vec{};
        constexpr std::integral_constant<int, 0> aux_plain{};

        // ---- prologue: t_n of planes x0-1, x0, x0+1; t_{n-1}, ids and y-neighbours of plane x0
        double2 cn_m[4], cn_0[4], cn_p[4], pv[4], ym[4], yp[4];
        load_cur(x0 - 1, valid, cn_m);
        load_cur(x0, valid, cn_0);
        load_cur(x0 + 1, valid, cn_p);
        load_plane(a.prev, aux_vec, x0, 0, owned, pv);
        uint2 ids = load_ids(x0, true);
        load_plane(a.cur, aux_plain, x0, -a.lz, has_ym, ym);
        load_plane(a.cur, aux_plain, x0, +a.lz, has_yp, yp);

        for (int k = x0; k < x1; ++k) {
            // ---- prefetch for the next iteration
            double2 nx_cn[4], nx_pv[4], nx_ym[4], nx_yp[4];
            const bool more = k + 1 < x1;
            load_cur(k + 2, valid && more, nx_cn);
            load_plane(a.prev, aux_vec, k + 1, 0, owned && more, nx_pv);
            const uint2 nx_ids = load_ids(k + 1, more);
            load_plane(a.cur, aux_plain, k + 1, -a.lz, has_ym && more, nx_ym);
            load_plane(a.cur, aux_plain, k + 1, +a.lz, has_yp && more, nx_yp);

#pragma unroll
            for (int be = 0; be < 4; ++be) row_n[SHARE_SLOT(lane, be)] = cn_0[be];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

            if (owned) {
                double2 acc[4], x[4];
#pragma unroll
                for (int al = 0; al < 4; ++al) acc[al] = zero;
                // CSR order: -P, -Lz, -1, 0, +1, +Lz, +P  (a reversed march swaps which register is -P)
                auto mac = [&](int slot, const double2 v[4]) {
                    const unsigned id = id_of(ids, slot);
                    if (id == kNoBlock) return;
                    if ((ids.y >> (24 + slot)) & 1u) Mode::mac_diag(acc, lds + id * STRIDE, v);
                    else Mode::mac_row(acc, lds + id * STRIDE, v);
                };
                if (rev) mac(0, cn_p);
                else mac(0, cn_m);
                mac(1, ym);
                if (id_of(ids, 2) != kNoBlock) {
#pragma unroll
                    for (int be = 0; be < 4; ++be) x[be] = row_n[SHARE_SLOT(lane - RL, be)];
                    mac(2, x);
                }
                mac(3, cn_0);
                if (id_of(ids, 4) != kNoBlock) {
#pragma unroll
                    for (int be = 0; be < 4; ++be) x[be] = row_n[SHARE_SLOT(lane + RL, be)];
                    mac(4, x);
                }
                mac(5, yp);
                if (rev) mac(6, cn_m);
                else mac(6, cn_p);
#pragma unroll
                for (int al = 0; al < 4; ++al) {
                    double2 nx;
                    nx.x = fma(a.coef, acc[al].x, -pv[al].x);
                    nx.y = fma(a.coef, acc[al].y, -pv[al].y);
                    if (!a.discard) {
                        const __amdgpu_buffer_rsrc_t rsrc = sweep_plane_rsrc(a.prev, ((size_t)al * nb + (size_t)act(k) * a.plane) * RL * sizeof(double2), plane_bytes);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, nx), rsrc, (p * RL + r) * (int)sizeof(double2), 0, kAuxVec);
                    }
                    Mode::dots(dot, cn_0[al], nx);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int al = 0; al < 4; ++al) {
                cn_m[al] = cn_0[al];
                cn_0[al] = cn_p[al];
                cn_p[al] = nx_cn[al];
                pv[al] = nx_pv[al];
                ym[al] = nx_ym[al];
                yp[al] = nx_yp[al];
            }
            ids = nx_ids;
        }
      }
    }

    __syncthreads();
    reduce_dots<Mode, RL>(dot, reinterpret_cast<double*>(lds), a.partial, lane, wave);
}

}  // namespace bdg
