import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
for L in (64, 100, 128):
    system = bench.build_system([L, L, 1])
    for T in (0.1, 0.5):
        t0 = time.perf_counter(); f = system.free_energy(T); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); f = system.free_energy(T); dt2 = time.perf_counter() - t0
        fs = system.free_energy(T, trace="stochastic")
        print(f"{L}x{L} T={T}: exact-trace F = {f:.8f} in {dt2:.2f} s (first {dt:.2f} s); stochastic(64) F = {fs:.8f} rel diff {abs(fs - f) / abs(f):.1e}", flush=True)
