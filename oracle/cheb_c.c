/*
 * CPU restatement in C of the Chebyshev recurrence on a BSR matrix with 4x4 blocks.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): it checks the numpy oracle and serves as
 * the multi-threaded CPU baseline of bench.py.  Same algorithm as oracle/cheb_ref.py:
 *
 *   t_next = coef * H t_cur - t_prev,   d = <t_cur|t_cur>,   e = Re <t_next|t_cur>   per vector
 *
 * Storage follows the reference (scipy BSR: blocks[k][4][4] complex128, C order; int32 indices).
 * Vectors are row-major (n_rows x R), i.e. numpy (4N, R) C-contiguous, the layout cheb_ref.py
 * uses for `bsr @ block`.  The real variant drops the imaginary parts (valid when imag(H) = 0
 * and the start vectors are real), which is what the GPU's real-arithmetic mode does.
 *
 * Build: gcc -O3 -mavx2 -mfma -fopenmp -shared -fPIC (oracle/cheb_c.py:build; no -march=native,
 * the object travels from the build container to the GPU box's host).
 */
#define _GNU_SOURCE
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_thread_num(void) { return 0; }
static int omp_get_num_threads(void) { return 1; }
static int omp_get_max_threads(void) { return 1; }
#endif

/*
 * Dot products: every thread sums its (static, contiguous) range of block rows into its own slot
 * and the slots are added in thread order afterwards - for a given thread count the result is
 * the same bits on every run (an `omp critical` in arrival order is not).
 *
 * NUMA: the block rows are cut with schedule(static), the same cut in every call.  Arrays that
 * were first written through cheb_c_copy_rows (one memcpy per block row by the thread that will
 * later process that row) therefore have their pages on the memory node that thread ran on (threads
 * are not pinned - see cheb_c.py - but busy threads are rarely migrated).  Without the placed copy
 * every page sits on the node of the Python thread that filled the numpy array, and a 256-thread
 * host runs at the bandwidth of one node.
 */
void cheb_c_copy_rows(int64_t nb, const int64_t* row_begin /* nb + 1 byte offsets */, const char* src, char* dst) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nb; ++i) memcpy(dst + row_begin[i], src + row_begin[i], (size_t)(row_begin[i + 1] - row_begin[i]));
}

int cheb_c_threads(void) { return omp_get_max_threads(); }

/*
 * Thread placement for the timed baseline.  Unpinned, the scheduler may pile the threads of a 16-thread run onto
 * one or two core complexes of a 256-thread host (measured on the GPU box: 93 to 984 steps/s for 8 to 14 threads
 * depending on where they landed, scratch/r3_cpu_probe.py) - a memory-bound loop then runs at the bandwidth of those
 * complexes.  cheb_c_pin_threads binds thread t of the pool to cpus[t] (the caller passes CPUs spread evenly over
 * the physical cores of one memory node: spread over both sockets 16 threads are six times slower, cheb_c.py) and
 * remembers every thread's previous mask; cheb_c_unpin_threads puts the masks back - the
 * calling Python thread is thread 0 of the pool, and a mask left on it would be inherited by every BLAS thread and
 * forked worker of the process (why OMP_PROC_BIND is not used).  Returns the number of threads bound.
 */
#define CHEB_C_MAX_PINNED 1024
static cpu_set_t cheb_c_saved_mask[CHEB_C_MAX_PINNED];
static int cheb_c_saved_count = 0;

int cheb_c_pin_threads(const int* cpus, int n_cpus) {
    int bound = 0;
    if (n_cpus <= 0) return 0;
    cheb_c_saved_count = 0;
#pragma omp parallel reduction(+ : bound)
    {
        const int t = omp_get_thread_num();
        if (t < CHEB_C_MAX_PINNED && t < n_cpus && sched_getaffinity(0, sizeof(cpu_set_t), &cheb_c_saved_mask[t]) == 0) {
            cpu_set_t want;
            CPU_ZERO(&want);
            CPU_SET(cpus[t], &want);
            if (sched_setaffinity(0, sizeof(cpu_set_t), &want) == 0) bound = 1;
        }
#pragma omp single
        cheb_c_saved_count = omp_get_num_threads() < CHEB_C_MAX_PINNED ? omp_get_num_threads() : CHEB_C_MAX_PINNED;
    }
    return bound;
}

void cheb_c_unpin_threads(void) {
    const int count = cheb_c_saved_count;
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        if (t < count) (void)sched_setaffinity(0, sizeof(cpu_set_t), &cheb_c_saved_mask[t]);
    }
    cheb_c_saved_count = 0;
}

void cheb_c_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* complex128: vectors as interleaved (re, im) doubles, shape (4*nb, R, 2) */
void cheb_c_step_complex(int64_t nb, const int32_t* indptr, const int32_t* indices, const double* blocks,
                         int R, double coef, const double* cur, double* prev /* in: t_prev, out: t_next */,
                         double* d_out, double* e_out) {
    const int max_threads = omp_get_max_threads();
    double* slots = (double*)calloc((size_t)max_threads * 2 * (size_t)R, sizeof(double));  /* [thread][{d, e}][R] */
    int used_threads = 1;
#pragma omp parallel
    {
        double* d_loc = slots + (size_t)omp_get_thread_num() * 2 * (size_t)R;
        double* e_loc = d_loc + R;
#pragma omp single
        used_threads = omp_get_num_threads();
        double* acc = (double*)malloc(sizeof(double) * 8 * (size_t)R);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < nb; ++i) {
            memset(acc, 0, sizeof(double) * 8 * (size_t)R);
            for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) {
                const double* blk = blocks + (size_t)k * 32;
                const double* x = cur + (size_t)indices[k] * 4 * R * 2;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        const double mr = blk[(a * 4 + b) * 2], mi = blk[(a * 4 + b) * 2 + 1];
                        const double* xb = x + (size_t)b * R * 2;
                        double* ya = acc + (size_t)a * R * 2;
                        for (int r = 0; r < R; ++r) {
                            ya[2 * r] += mr * xb[2 * r] - mi * xb[2 * r + 1];
                            ya[2 * r + 1] += mr * xb[2 * r + 1] + mi * xb[2 * r];
                        }
                    }
            }
            for (int a = 0; a < 4; ++a) {
                const double* c = cur + ((size_t)i * 4 + a) * R * 2;
                double* p = prev + ((size_t)i * 4 + a) * R * 2;
                const double* ya = acc + (size_t)a * R * 2;
                for (int r = 0; r < R; ++r) {
                    const double nr = coef * ya[2 * r] - p[2 * r], ni = coef * ya[2 * r + 1] - p[2 * r + 1];
                    p[2 * r] = nr;
                    p[2 * r + 1] = ni;
                    d_loc[r] += c[2 * r] * c[2 * r] + c[2 * r + 1] * c[2 * r + 1];
                    e_loc[r] += nr * c[2 * r] + ni * c[2 * r + 1];
                }
            }
        }
        free(acc);
    }
    for (int r = 0; r < R; ++r) d_out[r] = e_out[r] = 0.0;
    for (int t = 0; t < used_threads; ++t) /* thread order: reproducible */
        for (int r = 0; r < R; ++r) {
            d_out[r] += slots[(size_t)t * 2 * R + r];
            e_out[r] += slots[(size_t)t * 2 * R + R + r];
        }
    free(slots);
}

/* real: blocks_re[k][4][4] doubles, vectors (4*nb, R) doubles */
void cheb_c_step_real(int64_t nb, const int32_t* indptr, const int32_t* indices, const double* blocks_re,
                      int R, double coef, const double* cur, double* prev, double* d_out, double* e_out) {
    const int max_threads = omp_get_max_threads();
    double* slots = (double*)calloc((size_t)max_threads * 2 * (size_t)R, sizeof(double));  /* [thread][{d, e}][R] */
    int used_threads = 1;
#pragma omp parallel
    {
        double* d_loc = slots + (size_t)omp_get_thread_num() * 2 * (size_t)R;
        double* e_loc = d_loc + R;
#pragma omp single
        used_threads = omp_get_num_threads();
        double* acc = (double*)malloc(sizeof(double) * 4 * (size_t)R);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < nb; ++i) {
            memset(acc, 0, sizeof(double) * 4 * (size_t)R);
            for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) {
                const double* blk = blocks_re + (size_t)k * 16;
                const double* x = cur + (size_t)indices[k] * 4 * R;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        const double m = blk[a * 4 + b];
                        const double* xb = x + (size_t)b * R;
                        double* ya = acc + (size_t)a * R;
                        for (int r = 0; r < R; ++r) ya[r] += m * xb[r];
                    }
            }
            for (int a = 0; a < 4; ++a) {
                const double* c = cur + ((size_t)i * 4 + a) * R;
                double* p = prev + ((size_t)i * 4 + a) * R;
                const double* ya = acc + (size_t)a * R;
                for (int r = 0; r < R; ++r) {
                    const double n = coef * ya[r] - p[r];
                    p[r] = n;
                    d_loc[r] += c[r] * c[r];
                    e_loc[r] += n * c[r];
                }
            }
        }
        free(acc);
    }
    for (int r = 0; r < R; ++r) d_out[r] = e_out[r] = 0.0;
    for (int t = 0; t < used_threads; ++t) /* thread order: reproducible */
        for (int r = 0; r < R; ++r) {
            d_out[r] += slots[(size_t)t * 2 * R + r];
            e_out[r] += slots[(size_t)t * 2 * R + R + r];
        }
    free(slots);
}
