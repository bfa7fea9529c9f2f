// Ping-pong variant of stream_probe: the two arrays swap roles every launch, exactly as t_n and
// t_{n-1} do in the Chebyshev recurrence, so that Infinity-Cache (MALL) residency effects of the
// temporal hints on each access are the ones the real kernel would see.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v2d __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <bool NA, bool NB, bool NS>
__global__ void __launch_bounds__(256) k_recur(const v2d* __restrict__ a, v2d* __restrict__ b, double c, size_t n, size_t chunk, int reverse) {
    const size_t blk = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    size_t lo = blk * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        v2d x = NA ? __builtin_nontemporal_load(a + i) : a[i];
        v2d y = NB ? __builtin_nontemporal_load(b + i) : b[i];
        v2d r = c * x - y;
        if (NS) __builtin_nontemporal_store(r, b + i); else b[i] = r;
    }
}

template <bool NA, bool NB, bool NS>
static void run(v2d* a, v2d* b, size_t n, int grid, size_t bytes, bool alternate) {
    size_t chunk = (n + grid - 1) / grid;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 64;
    for (int i = 0; i < 4; ++i) { k_recur<NA, NB, NS><<<grid, 256>>>(a, b, 0.5, n, chunk, alternate ? (i & 1) : 0); v2d* t = a; a = b; b = t; }
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) { k_recur<NA, NB, NS><<<grid, 256>>>(a, b, 0.5, n, chunk, alternate ? (i & 1) : 0); v2d* t = a; a = b; b = t; }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("  grid %5d %s cur:%s prev:%s store:%s  %7.1f GB/s\n", grid, alternate ? "alternating" : "same dir   ", NA ? "nt" : "  ", NB ? "nt" : "  ", NS ? "nt" : "  ",
                3.0 * bytes / (ms / reps) * 1e-6);
}

int main(int argc, char** argv) {
    for (int arg = 1; arg < argc; ++arg) {
        const size_t bytes = std::strtoull(argv[arg], nullptr, 10) << 20;
        const size_t n = bytes / sizeof(v2d);
        v2d *a, *b;
        CHECK(hipMalloc(&a, bytes));
        CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 0, bytes));
        CHECK(hipMemset(b, 0, bytes));
        std::printf("arrays %zu MiB each, roles swapped every launch\n", bytes >> 20);
        for (int alt = 0; alt < 2; ++alt) {
            const int grid = 2048;
            run<false, false, false>(a, b, n, grid, bytes, alt);
            run<false, false, true>(a, b, n, grid, bytes, alt);
            run<false, true, false>(a, b, n, grid, bytes, alt);
            run<false, true, true>(a, b, n, grid, bytes, alt);
            run<true, false, false>(a, b, n, grid, bytes, alt);
            run<true, false, true>(a, b, n, grid, bytes, alt);
            run<true, true, false>(a, b, n, grid, bytes, alt);
            run<true, true, true>(a, b, n, grid, bytes, alt);
        }
        CHECK(hipFree(a));
        CHECK(hipFree(b));
    }
    return 0;
}
