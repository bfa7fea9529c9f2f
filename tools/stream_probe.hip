// HBM ceiling probe for the access pattern of the Chebyshev step: read t_n, read t_{n-1}, write
// t_{n+1} over t_{n-1} (2 reads + 1 write per element, in place), next to plain copy and read-only.
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/_build/stream_probe
// Prints GB/s per variant; used to put the kernel's roofline fraction next to what a trivial
// streaming kernel reaches on the same box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(256) k_read(const v2d* __restrict__ a, double* __restrict__ sink, size_t n) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v2d v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

template <bool NT>
__global__ void __launch_bounds__(256) k_copy(const v2d* __restrict__ a, v2d* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v2d v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}

// b = c*a - b   (read a, read b, write b)
template <bool NT>
__global__ void __launch_bounds__(256) k_recur(const v2d* __restrict__ a, v2d* __restrict__ b, double c, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v2d x = a[i];
        v2d y = NT ? __builtin_nontemporal_load(b + i) : b[i];
        v2d r = c * x - y;
        if (NT) __builtin_nontemporal_store(r, b + i); else b[i] = r;
    }
}

// same, contiguous chunk per workgroup (the tile sweep of the real kernel) instead of grid stride
__global__ void __launch_bounds__(256) k_recur_chunk(const v2d* __restrict__ a, v2d* __restrict__ b, double c, size_t n, size_t chunk) {
    size_t lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        v2d x = a[i];
        v2d y = b[i];
        b[i] = c * x - y;
    }
}

template <class F>
static double time_ms(F&& launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 256) << 20;  // per array, MiB
    const size_t n = bytes / sizeof(v2d);
    v2d *a, *b;
    double* sink;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(a, 0, bytes));
    CHECK(hipMemset(b, 0, bytes));
    const int reps = 50;
    std::printf("array %zu MiB each\n", bytes >> 20);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        double t;
        t = time_ms([&] { k_read<false><<<grid, 256>>>(a, sink, n); }, reps);
        std::printf("grid %5d read        %7.1f GB/s\n", grid, bytes / t * 1e-6);
        t = time_ms([&] { k_read<true><<<grid, 256>>>(a, sink, n); }, reps);
        std::printf("grid %5d read nt     %7.1f GB/s\n", grid, bytes / t * 1e-6);
        t = time_ms([&] { k_copy<false><<<grid, 256>>>(a, b, n); }, reps);
        std::printf("grid %5d copy        %7.1f GB/s\n", grid, 2.0 * bytes / t * 1e-6);
        t = time_ms([&] { k_copy<true><<<grid, 256>>>(a, b, n); }, reps);
        std::printf("grid %5d copy nt     %7.1f GB/s\n", grid, 2.0 * bytes / t * 1e-6);
        t = time_ms([&] { k_recur<false><<<grid, 256>>>(a, b, 0.5, n); }, reps);
        std::printf("grid %5d recur       %7.1f GB/s\n", grid, 3.0 * bytes / t * 1e-6);
        t = time_ms([&] { k_recur<true><<<grid, 256>>>(a, b, 0.5, n); }, reps);
        std::printf("grid %5d recur nt    %7.1f GB/s\n", grid, 3.0 * bytes / t * 1e-6);
        size_t chunk = (n + grid - 1) / grid;
        t = time_ms([&] { k_recur_chunk<<<grid, 256>>>(a, b, 0.5, n, chunk); }, reps);
        std::printf("grid %5d recur chunk %7.1f GB/s\n", grid, 3.0 * bytes / t * 1e-6);
    }
    return 0;
}
