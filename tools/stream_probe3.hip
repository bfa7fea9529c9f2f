// Access mix of the two-steps-per-sweep recurrence kernel: read t_n and t_{n-1}, write t_{n+1}
// and t_{n+2} (2 reads + 2 writes per element), four buffers rotating roles every launch exactly
// as the real kernel rotates them.  Gives the ceiling a trivial streaming kernel reaches for that
// mix on this machine, next to the 2R:1W mix of the one-step kernels (tools/stream_probe2.hip).
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe3.hip -o tools/_build/stream_probe3
//   tools/_build/stream_probe3 256 512
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v2d __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <bool NL, bool NS>
__global__ void __launch_bounds__(256) k_two(const v2d* __restrict__ cur, const v2d* __restrict__ prev,
                                             v2d* __restrict__ out1, v2d* __restrict__ out2, double c,
                                             size_t n, size_t chunk, int reverse) {
    const size_t blk = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    size_t lo = blk * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        v2d x = NL ? __builtin_nontemporal_load(cur + i) : cur[i];
        v2d y = NL ? __builtin_nontemporal_load(prev + i) : prev[i];
        v2d r1 = c * x - y;
        v2d r2 = c * r1 - x;
        if (NS) {
            __builtin_nontemporal_store(r1, out1 + i);
            __builtin_nontemporal_store(r2, out2 + i);
        } else {
            out1[i] = r1;
            out2[i] = r2;
        }
    }
}

template <bool NL, bool NS>
static void run(v2d* buf[4], size_t n, int grid, size_t bytes, bool alternate) {
    size_t chunk = (n + grid - 1) / grid;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 64;
    v2d *cur = buf[0], *prev = buf[1], *o1 = buf[2], *o2 = buf[3];
    auto launch = [&](int i) {
        k_two<NL, NS><<<grid, 256>>>(cur, prev, o1, o2, 0.5, n, chunk, alternate ? (i & 1) : 0);
        v2d *a = cur, *b = prev;
        cur = o2; prev = o1; o1 = b; o2 = a;
    };
    for (int i = 0; i < 4; ++i) launch(i);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(i);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("  grid %5d %s loads:%s stores:%s  %7.1f GB/s  (%.1f us per launch)\n", grid,
                alternate ? "alternating" : "same dir   ", NL ? "nt" : "  ", NS ? "nt" : "  ",
                4.0 * bytes / (ms / reps) * 1e-6, 1e3 * ms / reps);
}

int main(int argc, char** argv) {
    for (int arg = 1; arg < argc; ++arg) {
        const size_t bytes = std::strtoull(argv[arg], nullptr, 10) << 20;
        const size_t n = bytes / sizeof(v2d);
        v2d* buf[4];
        for (auto& b : buf) {
            CHECK(hipMalloc(&b, bytes));
            CHECK(hipMemset(b, 0, bytes));
        }
        std::printf("4 arrays of %zu MiB, 2 reads + 2 writes per element, roles rotate every launch\n", bytes >> 20);
        for (int grid : {1024, 2048, 8192})
            for (int alt = 0; alt < 2; ++alt) {
                run<false, false>(buf, n, grid, bytes, alt);
                run<false, true>(buf, n, grid, bytes, alt);
                run<true, false>(buf, n, grid, bytes, alt);
                run<true, true>(buf, n, grid, bytes, alt);
            }
        for (auto& b : buf) CHECK(hipFree(b));
    }
    return 0;
}
