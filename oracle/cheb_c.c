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
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int cheb_c_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
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
    double* d_acc = (double*)calloc((size_t)R, sizeof(double));
    double* e_acc = (double*)calloc((size_t)R, sizeof(double));
#pragma omp parallel
    {
        double* d_loc = (double*)calloc((size_t)R, sizeof(double));
        double* e_loc = (double*)calloc((size_t)R, sizeof(double));
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
#pragma omp critical
        for (int r = 0; r < R; ++r) {
            d_acc[r] += d_loc[r];
            e_acc[r] += e_loc[r];
        }
        free(d_loc);
        free(e_loc);
        free(acc);
    }
    memcpy(d_out, d_acc, sizeof(double) * (size_t)R);
    memcpy(e_out, e_acc, sizeof(double) * (size_t)R);
    free(d_acc);
    free(e_acc);
}

/* real: blocks_re[k][4][4] doubles, vectors (4*nb, R) doubles */
void cheb_c_step_real(int64_t nb, const int32_t* indptr, const int32_t* indices, const double* blocks_re,
                      int R, double coef, const double* cur, double* prev, double* d_out, double* e_out) {
    double* d_acc = (double*)calloc((size_t)R, sizeof(double));
    double* e_acc = (double*)calloc((size_t)R, sizeof(double));
#pragma omp parallel
    {
        double* d_loc = (double*)calloc((size_t)R, sizeof(double));
        double* e_loc = (double*)calloc((size_t)R, sizeof(double));
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
#pragma omp critical
        for (int r = 0; r < R; ++r) {
            d_acc[r] += d_loc[r];
            e_acc[r] += e_loc[r];
        }
        free(d_loc);
        free(e_loc);
        free(acc);
    }
    memcpy(d_out, d_acc, sizeof(double) * (size_t)R);
    memcpy(e_out, e_acc, sizeof(double) * (size_t)R);
    free(d_acc);
    free(e_acc);
}
