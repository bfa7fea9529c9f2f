/*
 * bodge_hip.h - C ABI of the MI355X (gfx950) BdG solver library `libbodge_hip.so`.
 *
 * The reference package is pure Python and has no FFI layer; its accelerator
 * seam is the `cuda: bool` switch inside three Hamiltonian methods, each of
 * which hands ONE matrix to a GPU library call and takes arrays back:
 *
 *   free_energy()  bodge/hamiltonian.py:282-302  (dense eigvalsh; CuPy at :291-295)
 *   diagonalize()  bodge/hamiltonian.py:203-232  (dense eigh;     CuPy at :210-221)
 *   ldos()         bodge/hamiltonian.py:342-382  (sparse LU resolvent per energy)
 *
 * and the matrix they consume is the BSR triple built at :37-67 / :102-118
 * (4x4 complex128 blocks, int32 indices/indptr).  The entry points below are
 * what a ctypes binding placed at those call sites would bind: upload the BSR
 * triple once, then ask for Chebyshev dot products / moments (free energy,
 * LDOS) or for a dense Hermitian eigensolve (diagonalize, small free_energy).
 *
 * Conventions: every pointer is a caller-owned host buffer unless stated; the
 * library copies to and from HBM.  Complex numbers are interleaved (re, im)
 * doubles, i.e. numpy complex128.  Every function returns 0 on success or a
 * negative BDG_E* code; `bdg_last_error()` then describes the failure (thread
 * local).  No callbacks, no exceptions, one host thread per handle.  The handles
 * of one device share that device's HIP streams (round 4): calls on two handles
 * from two host threads are correct but ordered on the GPU, not concurrent.
 */
#ifndef BODGE_HIP_H
#define BODGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDG_OK 0
#define BDG_EINVAL (-1)   /* bad argument (shape, range, null pointer)      */
#define BDG_EDEVICE (-2)  /* HIP runtime error / no usable GPU              */
#define BDG_ENOMEM (-3)   /* device allocation failed                       */
#define BDG_ELIBRARY (-4) /* rocSOLVER / RCCL could not be loaded or failed */

/* start-vector kinds for the stochastic trace (counter-based) */
#define BDG_VEC_RADEMACHER 0 /* entries +1 / -1                  */
#define BDG_VEC_Z4 1         /* entries in {1, i, -1, -i}        */

typedef struct bdg_system bdg_system; /* one uploaded Hamiltonian on one GPU */
typedef struct bdg_comm bdg_comm;     /* one RCCL communicator rank          */
typedef struct bdg_group bdg_group;   /* several slabs driven by one process */

/* Performance record of the most recent Chebyshev call on a handle. */
typedef struct bdg_perf {
    double kernel_ms;      /* HIP-event time around the recurrence launches, summed over the streams */
    int64_t launches;      /* number of recurrence-kernel launches in that window    */
    int64_t vector_steps;  /* launches x vectors advanced per launch                 */
    double bytes_per_launch; /* algorithmic HBM bytes of one launch      */
    int32_t lanes_per_row; /* kernel configuration actually used                     */
    int32_t vectors_per_launch;
    int32_t grid;          /* workgroups per launch                                  */
    int32_t lds_bytes;     /* LDS one workgroup occupies                             */
    int32_t pipelined;     /* 1 = register-prefetch kernel, 0 = generic kernel       */
    int32_t real_arithmetic; /* 1 = real-valued specialisation (imag(H)=0, real vectors) */
    int32_t strip_rows;    /* strip width of the tile order in block rows, 0 = natural order */
    int32_t ph_packed;     /* 1 = particle-hole packed blocks (12 of 16 entries stored)      */
    int32_t dict_blocks;   /* >0 = dictionary form: number of distinct blocks in the table   */
    int32_t steps_per_launch; /* recurrence steps one launch makes: 3 or 2 = multi-step sweeps of a 2-D lattice
                                 stencil (cheb_sweep3 / cheb_sweep), else 1 */
    int32_t rolling;       /* 1 = 3-D stencil kernel with the x-neighbours in registers (cheb_roll3) */
    int32_t dict_skipped;  /* why the matrix has no block dictionary: 0 = it has one, 1 = more than 256
                              distinct blocks, 2 = more than 2^24 block columns, 3 = BODGE_AMD_DICT=0 */
    int32_t onsite_streamed; /* 1 = position-dependent on-site blocks: the diagonal block of every site is
                                streamed from HBM once per launch, only the bond blocks sit in the table
                                (dict_blocks then counts the distinct bond blocks); 2 = the bond blocks are
                                streamed as well (real matrices with spin-diagonal hopping: no table)      */
    int32_t streams;       /* HIP streams the batches of the call ran on side by side (1 = back to back) */
    double bytes_moved;    /* algorithmic HBM bytes of all `launches` together.  Less than launches x
                              bytes_per_launch: the first sweep of a run reads no t_{-1} (and no t_0 when it
                              makes the random start block itself), and the last launch of a run stores no
                              vectors - nothing reads them, the call returns dot products (BODGE_AMD_KEEP_LAST=1
                              stores them all the same).  Divide by window_ms for the achieved rate. */
    double window_ms;      /* HIP-event time from the first launch of the call to the end of its last one.  A call
                              with several batches of vectors runs them on `streams` streams side by side (the
                              idle start and end of one launch fill with another batch's work): kernel_ms then
                              sums the streams' own times (kernel_ms / launches = duration of one launch, what a
                              profiler reports) and window_ms is the elapsed time they share.  One stream:
                              window_ms = kernel_ms. */
    int64_t sweeps;        /* multi-step sweeps made, summed over the lane groups (= launches with one launch per sweep) */
    int32_t persistent;    /* 1 = cheb_march3: every launch makes all the sweeps of a reduction chunk (up to 21 = 63
                              steps) of `groups_per_launch` lane groups; its waves claim (sweep, group, unit) tasks and
                              start one as soon as the neighbouring units have published the sweep before */
    int32_t groups_per_launch; /* lane groups (batches of the call) advanced by one persistent launch, else 1 */
} bdg_perf;

const char* bdg_last_error(void);
const char* bdg_version(void);

/* Number of visible HIP devices (0 is a valid answer and not an error). */
int bdg_device_count(int* count);

/*
 * Upload a BSR matrix with 4x4 complex128 blocks (replaces the host->device
 * copy `cp.asarray(H)` of hamiltonian.py:214 / :295, but for the sparse form).
 *   nb      number of block rows (lattice sites)
 *   nnzb    number of stored blocks
 *   indptr  int32[nb+1], indices int32[nnzb] (column block per stored block)
 *   data    double[nnzb*32]: blocks in C order data[k][row][col] (re, im)
 * The triple must be canonical (sorted, no duplicates); validated on the host.
 */
int bdg_create(int device, int64_t nb, int64_t nnzb, const int32_t* indptr,
               const int32_t* indices, const double* data, bdg_system** out);
int bdg_destroy(bdg_system* sys);

/*
 * Row-slab variant (domain decomposition over lattice planes, one slab per GPU): the
 * handle owns block rows [row_offset, row_offset + nb) of the global matrix.  Column ids
 * are LOCAL: 0..nb-1 are the owned rows, nb..ncols-1 are halo rows whose t_n entries are
 * refreshed from other slabs before every recurrence launch.  Start vectors are indexed
 * globally, so the result does not depend on how the rows are cut.  bodge_amd/slab.py
 * derives these arrays from the global BSR triple.
 */
int bdg_create_slab(int device, int64_t nb, int64_t ncols, int64_t nnzb, const int32_t* indptr,
                    const int32_t* indices, const double* data, int64_t row_offset,
                    bdg_system** out);
/*
 * Exchange lists of a slab.  For peer p: send_count[p] owned rows (local ids, concatenated
 * in send_rows) go to it and recv_count[p] rows arrive from it into local columns
 * recv_col[p] .. recv_col[p]+recv_count[p]-1.  peer_rank[p] is an RCCL rank when `comm` is
 * given (one process per GPU: grouped ncclSend/ncclRecv per launch), or a member index of a
 * bdg_group when comm is NULL (one process driving several slabs: device-to-device copies).
 */
int bdg_slab_set_exchange(bdg_system* sys, bdg_comm* comm, int32_t n_peers, const int32_t* peer_rank,
                          const int64_t* send_count, const int64_t* send_rows,
                          const int64_t* recv_col, const int64_t* recv_count);
/* Same-process group of slabs: the dot products returned are summed over the members. */
int bdg_group_create(bdg_system** members, int32_t n_members, bdg_group** out);
int bdg_group_destroy(bdg_group* group);
int bdg_group_dots_random(bdg_group* group, double scale, int32_t n_steps, int32_t n_vectors,
                          uint64_t seed, uint64_t first_vec_id, int32_t vec_kind,
                          double* d_out, double* e_out);
int bdg_group_dots_unit(bdg_group* group, double scale, int32_t n_steps, int32_t n_vectors,
                        const int64_t* rows, double* d_out, double* e_out);

/* y = H x for one host vector of 4*nb complex entries (site-major, the order
 * of `H @ x` on the reference's matrix).  For parity tests. */
int bdg_spmv(bdg_system* sys, const double* x, double* y);

/*
 * Chebyshev recurrence on `n_vectors` start vectors advanced together:
 *   t_0 = v,  t_1 = H t_0 / scale,  t_{n+1} = 2 H t_n / scale - t_{n-1}
 *   d[n*n_vectors + r] = <t_n|t_n>,  e[n*n_vectors + r] = Re <t_{n+1}|t_n>,  n < n_steps
 * `scale` must bound the spectrum of H.  Random start vectors are generated on
 * the device from (seed, first_vec_id + r, element index); unit start vectors
 * are e_{rows[r]} (rows index the 4*nb scalar rows).
 */
int bdg_cheb_dots_random(bdg_system* sys, double scale, int32_t n_steps, int32_t n_vectors,
                         uint64_t seed, uint64_t first_vec_id, int32_t vec_kind,
                         double* d_out, double* e_out);
int bdg_cheb_dots_unit(bdg_system* sys, double scale, int32_t n_steps, int32_t n_vectors,
                       const int64_t* rows, double* d_out, double* e_out);

/*
 * Moments mu_m = sum_r <v_r|T_m(H/scale)|v_r>, m < n_moments (even), from the
 * dots above via mu_2n = 2 d_n - mu_0, mu_2n+1 = 2 e_n - mu_1.  If `comm` is
 * non-null the moment vector is summed over all ranks with one RCCL
 * all-reduce before it is returned (each rank passes its own first_vec_id).
 */
int bdg_cheb_moments(bdg_system* sys, bdg_comm* comm, double scale, int32_t n_moments,
                     int32_t n_vectors, uint64_t seed, uint64_t first_vec_id, int32_t vec_kind,
                     double* mu_out);
/* Per-start-vector moments for unit vectors: mu_out[m*n_vectors + r]. */
int bdg_cheb_diag_moments(bdg_system* sys, double scale, int32_t n_moments, int32_t n_vectors,
                          const int64_t* rows, double* mu_out);

/*
 * Lanczos process on H^2 (eigenvalues of H closest to zero, i.e. the excitation gap, for
 * matrices beyond dense reach).  `begin` prepares n_vectors (<= 64) independent processes from
 * counter-based start vectors; `advance` runs n_iter more iterations and returns, per
 * iteration j and vector r, alpha[j*n_vectors + r] = <v_j|H^2|v_j> and
 * beta[j*n_vectors + r] = beta_{j+1} (the tridiagonal matrix of H^2 in the Lanczos basis;
 * its lowest eigenvalues converge to the squares of the smallest |eigenvalues| of H).
 * At most max_iter iterations in total.  Whole matrices only (not slabs).
 */
int bdg_lanczos_begin(bdg_system* sys, int32_t n_vectors, uint64_t seed, uint64_t first_vec_id,
                      int32_t vec_kind, int32_t max_iter);
int bdg_lanczos_advance(bdg_system* sys, int32_t n_iter, double* alpha_out, double* beta_out);
/*
 * Ritz vectors by a second pass: on a freshly begun process (same arguments as the first pass:
 * the Lanczos vectors v_j are reproduced bit for bit) run n_iter iterations and accumulate
 *   y[l][r] = sum_j coef[(j*n_levels + l)*n_vectors + r] * v_j^{(r)},   l < n_levels,
 * the eigenvector estimates of H^2 whose coordinates in the Lanczos basis the caller computed
 * from alpha/beta.  y_out[(l*n_vectors + r)*8*nb ...] receives 4*nb complex entries (site-major)
 * per (level, start vector).  With eigenvalue eps, (H + eps) y is an eigenvector of H.
 */
int bdg_lanczos_ritz_vectors(bdg_system* sys, int32_t n_iter, int32_t n_levels, const double* coef,
                             double* y_out);

/*
 * Eigenpairs from the same second pass without the Ritz block leaving the device (Rayleigh-Ritz with H
 * on the GPU; what a caller of diagonalize() wants from a lattice too large to diagonalise: reference
 * tests/test_physics.py:105, `eigvals, eigvecs = system.diagonalize()`).  Arguments as for
 * bdg_lanczos_ritz_vectors, plus eps[l] = the level's eigenvalue estimate (square root of the converged
 * Ritz value of H^2).  Per level the candidates (H + eps_l) y span the +eps_l eigenspace; the rank of
 * their Gram matrix (eigenvalues above rank_tol x the largest) is the multiplicity; H is diagonalised
 * inside that span.  n_out receives the number of states found over all levels; the max_out lowest are
 * returned: values_out[k] ascending, vectors_out + 8*nb*k = 4*nb complex entries (site-major) of state k.
 * At most 16 columns (16 real-arithmetic or 16 complex start vectors).  Ends the Lanczos run.
 */
int bdg_lanczos_ritz_pairs(bdg_system* sys, int32_t n_iter, int32_t n_levels, const double* coef,
                           const double* eps, double rank_tol, int32_t max_out, int32_t* n_out,
                           double* values_out, double* vectors_out);

/* Write the counter-based start vector (4*nb complex entries) to a host buffer. */
int bdg_random_vector(bdg_system* sys, uint64_t seed, uint64_t vec_id, int32_t vec_kind,
                      double* v_out);

/*
 * Dense Hermitian eigensolve of the uploaded matrix on the GPU (replaces
 * cupy.linalg.eigvalsh / eigh at hamiltonian.py:295 / :215): all 4*nb
 * eigenvalues ascending in w_out; if z_out is non-null it receives the
 * eigenvectors in column-major (Fortran) order: eigenvector n occupies the
 * 4nb complex entries starting at z_out[2*n*4nb].
 * Drivers.  Eigenvalues only (z_out NULL, what free_energy needs): own Householder
 * tridiagonalisation + bisection from 4*nb > 512 on (no library; real arithmetic when imag(H) = 0),
 * own Jacobi kernels below.  With eigenvectors: own one-sided Jacobi kernels for 4*nb <= 2048 (and
 * up to 4096 for as long as the rocSOLVER object is still being read from cold storage, see
 * bdg_dense_prefetch); above, rocSOLVER dsyevd when imag(H) = 0 (real eigenvectors, widened to
 * complex in z_out) and zheevd otherwise; limit 4*nb <= 46000.  Results are scanned for non-finite
 * values on the device.
 */
int bdg_eigh_dense(bdg_system* sys, double* w_out, double* z_out);

/*
 * What diagonalize() keeps (hamiltonian.py:235-238: the eigenpairs with eigenvalue > 0): all 4*nb
 * eigenvalues ascending in w_out, and the eigenvectors of those above `lower_bound` only - vector m of
 * them (ascending) in the 4*nb complex entries from z_out + 8*nb*m; *n_vectors = their number.  If it
 * exceeds `capacity` (vectors z_out has room for) BDG_EINVAL is returned with *n_vectors set and z_out
 * untouched.  Route from 4*nb > 512 on, no library involved: Householder tridiagonalisation, bisection,
 * inverse iteration on the tridiagonal matrix (close eigenvalues orthogonalised against each other) and
 * back-transformation by the stored reflectors - half the vectors of the full solve for a BdG matrix.
 * Below, and when BODGE_AMD_EIGH names another driver, bdg_eigh_dense runs and its result is cut.
 */
int bdg_eigh_dense_above(bdg_system* sys, double lower_bound, int64_t capacity, double* w_out,
                         int64_t* n_vectors, double* z_out);

/*
 * max |H - H^†| over the stored entries of the uploaded matrix, computed on the device: the
 * Hermiticity test the reference makes on the host when a `with` block closes
 * (hamiltonian.py:121-122, `abs(M - M.getH()).max() > 1e-6` -> RuntimeError).  Whole matrices only.
 */
int bdg_hermiticity_defect(bdg_system* sys, double* defect_out);

/*
 * Host-side assembly helpers: CPU threads only, no device call, usable without a GPU.  They make
 * the passes over the BSR blocks that precede bdg_create with every core instead of one.
 *
 * bdg_host_fill_terms: scatter `count` 2x2 complex spin matrices (`values`: 8 doubles each if
 *   per_term, else one matrix used for every term) into blocks `ids[n]` of `data` (nnzb x 4 x 4
 *   complex128), in term order.  kind 0: hopping H_ij -> blk[0:2,0:2] = v, blk[2:4,2:4] = -conj(v)
 *   (hamiltonian.py:106-108); kind 1: pairing Δ_ij -> blk[0:2,2:4] = v (:112-113); kind 2: the
 *   transposed block of a pairing term, blk[2:4,0:2] = v^† (:115-116).  `touched` (nnzb bytes or
 *   NULL) gets 1 at every block written.
 * bdg_host_scan_blocks: one read of every block.  nonzero[k] = block k has a non-zero entry (the
 *   blocks `matrix("bsr")` keeps, :142-143); ph_defect = max |blk[2:4,2:4] + conj(blk[0:2,0:2])|,
 *   |blk[2:4,0:2] + conj(blk[0:2,2:4])| (0 = spectrum symmetric about zero); row_sum_max = max
 *   over scalar rows of Σ|H_rc| (Gershgorin bound); all_real = imag(data) == 0.  Any output may
 *   be NULL.  A NaN entry makes both floating-point results NaN.
 * bdg_host_compact_blocks: the BSR triple without the blocks whose keep[k] is 0; outputs are
 *   caller-allocated (indptr_out nb+1, the others n_nonzero long).
 */
int bdg_host_fill_terms(double* data, int64_t nnzb, const int64_t* ids, int64_t count, const double* values,
                        int per_term, int kind, uint8_t* touched);
int bdg_host_scan_blocks(const double* data, const int32_t* indptr, int64_t nb, uint8_t* nonzero, int64_t* n_nonzero,
                         double* ph_defect, double* row_sum_max, int32_t* all_real);
int bdg_host_compact_blocks(const double* data, const int32_t* indices, const int32_t* indptr, int64_t nb,
                            const uint8_t* keep, double* data_out, int32_t* indices_out, int32_t* indptr_out);

/*
 * Optional: start reading the rocSOLVER / rocBLAS shared objects into the page cache on a
 * background thread (file I/O only; returns at once, idempotent).  The first dense eigensolve
 * above 4*nb = 2048 loads a 931 MB library, which takes minutes from cold storage; a caller who
 * knows it will diagonalize (the reference's diagonalize() / free_energy() call sites,
 * hamiltonian.py:203-232, :282-302) calls this when the Hamiltonian is created so that the read
 * overlaps with assembly and upload.  bdg_eigh_dense waits for it.
 */
int bdg_dense_prefetch(void);
/* Wait up to timeout_seconds (negative: until done) for that read; *ready = 1 once it has finished. */
int bdg_dense_prefetch_wait(double timeout_seconds, int32_t* ready);
/* The same for the RCCL shared object (573 MB), which the first bdg_comm_* call loads.  If the
 * dense-solver read has been started too, this one follows it (one stream at a time). */
int bdg_rccl_prefetch(void);
int bdg_rccl_prefetch_wait(double timeout_seconds, int32_t* ready);

int bdg_perf_query(bdg_system* sys, bdg_perf* out);

/*
 * Optional geometry hint: block row i is site (x, y, z) of an lx x ly x lz cubic
 * lattice with i = z + lz*(y + ly*x) (reference lattice.py:108).  Lets the library
 * order its row tiles strip-major so that neighbour re-reads stay in L2.  Pure
 * performance hint: results do not depend on it.  (0,0,0) clears it.
 */
int bdg_set_lattice_shape(bdg_system* sys, int32_t lx, int32_t ly, int32_t lz);

/* Tuning override for experiments: lanes per block row (0 = automatic). */
int bdg_set_lanes_per_row(bdg_system* sys, int32_t lanes);

/*
 * Process-wide override of one of the library's run-time switches (the BODGE_AMD_* names listed
 * in csrc/knobs.hpp and DESIGN.md's appendix): `value` replaces what the environment variable of
 * that name says, value = NULL removes the override.  Thread safe, unlike setenv() next to the
 * getenv() of a running call; tests and bench.py select kernel forms this way.  The defaults
 * are the product: nothing needs to be set in normal use.
 */
int bdg_set_option(const char* name, const char* value);

/*
 * RCCL communicator, one rank per process/GPU.  Rank 0 calls
 * bdg_comm_unique_id and distributes the 128 bytes out of band; every rank
 * then calls bdg_comm_init with the same bytes.
 */
int bdg_comm_unique_id(uint8_t id_out[128]);
int bdg_comm_init(int device, const uint8_t id[128], int32_t n_ranks, int32_t rank,
                  bdg_comm** out);
/* What the communicator really spans: ncclCommCount, this rank, its device ordinal and that
 * device's PCI bus id (e.g. "0000:75:00.0") - for records that must show N distinct GPUs. */
int bdg_comm_info(bdg_comm* comm, int32_t* n_ranks, int32_t* rank, int32_t* device, char pci_bus_id[32]);
int bdg_comm_allreduce_sum(bdg_comm* comm, double* buf, int64_t count);
int bdg_comm_allreduce_max(bdg_comm* comm, double* buf, int64_t count);
int bdg_comm_destroy(bdg_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* BODGE_HIP_H */
