// knobs.hpp - every run-time switch of the library, in one place.
//
// All of them are read at the point of use through knob::raw: first the table of overrides set
// with bdg_set_option (what the tests and bench.py use - calling setenv while another host thread
// of a `devices=[...]` run sits in getenv is undefined behaviour in glibc, and the library itself
// never modifies the environment), then the environment variable of the same name (read-only:
// convenient from a shell).  None is needed in normal use - the defaults are the product, the
// switches exist for the parity tests (every kernel form against every other) and for the A/B
// measurements behind DESIGN.md's tables.  "=0" disables, any other value enables, unless stated.
//
//   kernel choice
//     BODGE_AMD_KERNEL=generic|pipelined     one-step kernel form (default: by lanes per row)
//     BODGE_AMD_DICT=0                       no block dictionary: stream the blocks (also disables the stencil kernels)
//     BODGE_AMD_REAL=0 / BODGE_AMD_PH=0      complex arithmetic although H and the vectors are real / full 16-entry blocks
//     BODGE_AMD_SWEEP=0|1                    never / whenever possible use the lattice-stencil kernels (default: by size)
//     BODGE_AMD_SWEEP_STEPS=2|3              steps per sweep (cheb_sweep / cheb_sweep3)
//     BODGE_AMD_SWEEP_LANES=1|2|4            lanes per site of the sweep kernels
//     BODGE_AMD_STREAMED_SHARE=0             streamed forms (position-dependent blocks): segments cut for the whole device by every launch
//                                            (default: shared between the lane groups side by side, as for the dictionary forms)
//     BODGE_AMD_ROLL_CHUNKS=0                3-D rolling kernel: whole-column segments as units instead of runs of equal length
//     BODGE_AMD_SWEEP_GEN=0                  write the random start block with the fill kernel instead of making it in the first sweep
//     BODGE_AMD_MARCH=1|2|3                  cheb_march3 for the three-step sweeps of random-start runs (default 0: one cheb_sweep3 launch per
//                                            sweep and lane group).  1 = all sweeps of a 63-step chunk in one launch, tasks claimed by ticket,
//                                            flags between neighbouring units; 3 = the same with a fixed unit per wave (needs the grid
//                                            resident: gives up after the timeout otherwise); 2 = one launch per sweep for all lane groups
//     BODGE_AMD_MARCH_TIMEOUT_MS=ms          how long a wave polls its neighbours' flags before the launch is given up (default 2000)
//     BODGE_AMD_MARCH_SLEEP=4|16|64          s_sleep between two polls;  BODGE_AMD_MARCH_DEBUG=bits  measurements / tests (sweep.hpp: MarchArgs.debug)
//     BODGE_AMD_EIGH=jacobi|tridiagonal|rocsolver|evd|evj|ev   dense solver route (default: Jacobi up to 512 rows, own
//                                            tridiagonalisation route above);  BODGE_AMD_EIGH_REAL=0  complex arithmetic for a real matrix
//     BODGE_AMD_EIGH_DEFER=1..4              reflector pairs kept pending in the tridiagonalisation (default 4 from 5000 rows, else 1)
//     BODGE_AMD_EIGH_CHUNKS=cap[,rows]       row chunks of a back-transformation pass: at most `cap` (default 48) of at least `rows` rows (64)
//     BODGE_AMD_EIGH_STAGES=1|2              real matrices: one-stage tridiagonalisation (tridiag.hpp) / through a band (twostage.hpp); default 2 from
//                                            5000 rows (eigenvalues) / 3000 rows (eigenpairs).  Pieces of the two-stage route, for tests and A/B runs:
//                                            _GRAM_QR=0 (every panel with a grid barrier per column), _GRAM_FLOOR=x (share of a column's squared norm below
//                                            which the Gram route gives a panel up; 0.25), _VERIFY=tol (tolerance of the check of Q^T P; 2e-13),
//                                            _LOOKAHEAD=0|1, _BAND_VECTORS=0 (eigenvectors by the one-stage route), _BAND_ITERATIONS=n (3),
//                                            _STAGE2=0, _CHASE_GRID=n, _CHASE_PROFILE (phase times of the bulge chasing on stderr)
//     BODGE_AMD_ONSITE_STREAM=0              (read at upload) no bond-only dictionary + on-site stream for matrices with > 256 distinct blocks
//     BODGE_AMD_NO_DIAGONAL_BLOCKS           withhold the "diagonal as a 4x4 matrix" flag of dictionary blocks (read at upload)
//   launch shape and memory hints
//     BODGE_AMD_BLOCKS_PER_CU=n              cap on resident workgroups per CU
//     BODGE_AMD_SWEEP_SEGMENTS=n             x-segments of the sweep / rolling kernels (default: choose_segments)
//     BODGE_AMD_SWEEP_ZIGZAG=0               all segments march the same way
//     BODGE_AMD_ALTERNATE=0                  launches do not alternate their marching direction
//     BODGE_AMD_KEEP_LAST=1                  the last launch of a run stores its vectors like any other
//     BODGE_AMD_STREAMS=1..4                 streams the batches of one call run on side by side (default: 2 for the marching kernels)
//     BODGE_AMD_SWEEP_STREAM=bits            non-temporal hints of the sweep kernels (1 t_{n-1} loads, 2 stores, 4 t_n loads, 8 on-site records)
//     BODGE_AMD_STREAM_VECTORS=bits          the same for the one-step and rolling kernels
//     BODGE_AMD_L2_BUDGET=bytes              per-XCD budget behind the strip width of the tile order
//     BODGE_AMD_BATCH=n                      vectors per launch of the one-step kernels (default: batch_width)
//     BODGE_AMD_NO_BAND                      LDOS / unit starts sweep the whole matrix instead of the growing band
//     BODGE_AMD_NO_BATCH_PIPELINE            wait for every batch of a call before beginning the next
//     BODGE_AMD_OVERLAP=0|1                  slab halo exchange on the compute stream / overlapped on a second stream
//   host side
//     BODGE_AMD_HOST_THREADS=n               threads of the bdg_host_* passes (default: all cores, at most 32)
//     BODGE_AMD_NO_PREFETCH                  no background read of the rocSOLVER / RCCL shared objects
//     BODGE_AMD_TRACE                        timing lines on stderr
//   (read by the Python side: BODGE_AMD_LIBRARY, BODGE_AMD_DEVICE, BODGE_AMD_HOST_NATIVE, BODGE_AMD_FORCE_COMM)
#pragma once

#include <cstdlib>
#include <deque>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>

namespace knob {

// Overrides: name -> value, or name -> nullptr for "behave as if the variable were unset".
// Values live in a pool that is never freed, so a pointer handed out stays valid even if the
// option is changed while a call on another thread is still looking at it.
struct Table {
    std::shared_mutex mutex;
    std::unordered_map<std::string, const char*> entries;
    std::deque<std::string> pool;
};
inline Table& table() {
    static Table* t = new Table();  // (leaked on purpose: detached prefetch threads may outlive main)
    return *t;
}

// value = nullptr removes the override (the environment variable, if any, shows again)
inline void set(const char* name, const char* value) {
    Table& t = table();
    std::unique_lock<std::shared_mutex> lock(t.mutex);
    if (!value) {
        t.entries.erase(name);
        return;
    }
    for (const std::string& kept : t.pool)  // (the pool holds every distinct value once: it does not grow with the number of calls)
        if (kept == value) {
            t.entries[name] = kept.c_str();
            return;
        }
    t.pool.emplace_back(value);
    t.entries[name] = t.pool.back().c_str();
}

inline const char* raw(const char* name) {
    Table& t = table();
    {
        std::shared_lock<std::shared_mutex> lock(t.mutex);
        if (!t.entries.empty()) {
            auto it = t.entries.find(name);
            if (it != t.entries.end()) return it->second;
        }
    }
    return std::getenv(name);
}

}  // namespace knob
