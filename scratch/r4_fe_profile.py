"""Where the wall time of a repeated free_energy call goes (1000x1000, 512 moments, 64 vectors)."""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
system = bench.build_system([1000, 1000, 1], "swave")
call = lambda: system.free_energy(0.5, method="chebyshev", moments=512, vectors=64, trace="stochastic")
t0 = time.perf_counter(); call(); print("first", time.perf_counter() - t0)
for _ in range(3):
    t0 = time.perf_counter(); call(); print("repeat", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); call(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
p = system._solver().perf()
print({k: p[k] for k in ("kernel_ms", "window_ms", "launches", "streams", "vector_steps") if k in p})
